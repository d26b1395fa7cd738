/*
 * oracle/ref_stub_sdk/pico_stub_all.h — host stand-in for the un-vendored pico-sdk (TEST INFRASTRUCTURE ONLY).
 *
 * The reference's usb_audio.c, flash_storage.c and pdm_generator.c include pico/ and hardware/ headers from
 * firmware/pico-sdk, an empty git submodule.  None of the arithmetic on the hot path lives
 * in those headers; what the three files need from them is a handful of type names, attribute macros and
 * hardware entry points (timers, spin locks, flash program/erase, PIO/DMA setup, multicore FIFO).  This file
 * declares exactly those — written from the call sites in the reference, not from the SDK — so the three
 * files compile IN PLACE into oracle/_ref/libref_fw_*.so; ref_fw.c defines the functions (RAM-backed flash,
 * counters for time, no-ops for the rest).  Nothing here is reference code and nothing here is product code.
 */
#ifndef ORC_PICO_STUB_ALL_H
#define ORC_PICO_STUB_ALL_H
#include <stdint.h>
#include <stdbool.h>
#include <stddef.h>
#include <string.h>
#include <assert.h>

typedef unsigned int uint;
typedef uint64_t absolute_time_t;
typedef volatile uint32_t io_rw_32;
typedef volatile uint32_t spin_lock_t;

#ifndef __unused
#define __unused __attribute__((unused))
#endif
#define __not_in_flash(group)
#define __not_in_flash_func(f) f
#define __time_critical_func(f) f
#define __no_inline_not_in_flash_func(f) f
#define __scratch_x(group)
#define __scratch_y(group)
#ifndef __packed
#define __packed __attribute__((packed))
#endif
#ifndef __aligned
#define __aligned(x) __attribute__((aligned(x)))
#endif
#ifndef __force_inline
#define __force_inline inline __attribute__((always_inline))
#endif
#define __packed_aligned __attribute__((packed, aligned(4)))
#define count_of(a) (sizeof(a) / sizeof((a)[0]))
#ifndef MIN
#define MIN(a, b) ((a) < (b) ? (a) : (b))
#endif
#ifndef MAX
#define MAX(a, b) ((a) > (b) ? (a) : (b))
#endif
#define panic(...) assert(!"panic")
#define hard_assert(x) assert(x)
#define invalid_params_if(x, test) ((void)0)
#define valid_params_if(x, test) ((void)0)
#define tight_loop_contents() ((void)0)
#define PICO_OK 0
#define PICO_RP2040 (!PICO_RP2350)

/* ---- platform ---- */
#define USB_NUM_ENDPOINTS 16
#define NUM_ADC_CHANNELS 5
#define NUM_DMA_CHANNELS 12
#define NUM_BANK0_GPIOS 30
#define DMA_IRQ_0 11
#define DMA_IRQ_1 12
#define PICO_HIGHEST_IRQ_PRIORITY 0x00
#define PICO_DEFAULT_IRQ_PRIORITY 0x80
#define PICO_LOWEST_IRQ_PRIORITY 0xff
#define PICO_DEFAULT_LED_PIN 25
#define __mul_instruction(a, b) ((a) * (b))
#define __compiler_memory_barrier() __asm__ volatile("" ::: "memory")
#define remove_volatile_cast(t, x) ((t)(uintptr_t)(x))
uint get_core_num(void);

/* ---- hardware/sync.h ---- */
static inline void __dmb(void) { __asm__ volatile("" ::: "memory"); }
static inline void __dsb(void) { __asm__ volatile("" ::: "memory"); }
static inline void __isb(void) { __asm__ volatile("" ::: "memory"); }
static inline void __sev(void) {}
#ifndef ORC_WFE                      /* each wrapper TU says what "wait for the other core" means (ref_fw.c, ref_fw_core1.c) */
#define ORC_WFE() ((void)0)
#endif
static inline void __wfe(void) { ORC_WFE(); }
static inline void __wfi(void) {}
uint32_t save_and_disable_interrupts(void);
void restore_interrupts(uint32_t status);
uint32_t spin_lock_blocking(spin_lock_t *lock);
void spin_unlock(spin_lock_t *lock, uint32_t saved_irq);
spin_lock_t *spin_lock_init(uint lock_num);
spin_lock_t *spin_lock_instance(uint lock_num);
int spin_lock_claim_unused(bool required);
uint next_striped_spin_lock_num(void);

/* ---- time ---- */
uint32_t time_us_32(void);
uint64_t time_us_64(void);
void busy_wait_ms(uint32_t ms);
void busy_wait_us(uint64_t us);
void busy_wait_us_32(uint32_t us);
void sleep_ms(uint32_t ms);
void sleep_us(uint64_t us);
absolute_time_t get_absolute_time(void);
absolute_time_t make_timeout_time_ms(uint32_t ms);
uint32_t to_ms_since_boot(absolute_time_t t);
int64_t absolute_time_diff_us(absolute_time_t from, absolute_time_t to);
bool time_reached(absolute_time_t t);

/* ---- clocks / vreg / adc / gpio / irq / watchdog / bootrom ---- */
enum clock_index { clk_gpout0, clk_ref, clk_sys, clk_peri, clk_usb, clk_adc };
uint32_t clock_get_hz(uint clk);
enum vreg_voltage { VREG_VOLTAGE_0_85 = 6, VREG_VOLTAGE_1_10 = 11, VREG_VOLTAGE_1_15 = 12, VREG_VOLTAGE_1_20 = 13 };
enum vreg_voltage vreg_get_voltage(void);
void vreg_set_voltage(enum vreg_voltage v);
void adc_init(void);
void adc_select_input(uint input);
uint16_t adc_read(void);
void adc_set_temp_sensor_enabled(bool enable);
enum gpio_function { GPIO_FUNC_XIP = 0, GPIO_FUNC_SPI = 1, GPIO_FUNC_UART = 2, GPIO_FUNC_I2C = 3, GPIO_FUNC_PWM = 4, GPIO_FUNC_SIO = 5,
                     GPIO_FUNC_PIO0 = 6, GPIO_FUNC_PIO1 = 7, GPIO_FUNC_PIO2 = 8, GPIO_FUNC_GPCK = 9, GPIO_FUNC_USB = 10, GPIO_FUNC_NULL = 0x1f };
#define GPIO_OUT 1
#define GPIO_IN 0
void gpio_init(uint gpio);
void gpio_set_dir(uint gpio, bool out);
void gpio_put(uint gpio, bool value);
bool gpio_get(uint gpio);
void gpio_set_function(uint gpio, enum gpio_function fn);
void gpio_pull_up(uint gpio);
void gpio_pull_down(uint gpio);
void gpio_disable_pulls(uint gpio);
void gpio_set_drive_strength(uint gpio, int drive);
void gpio_set_slew_rate(uint gpio, int slew);
typedef void (*irq_handler_t)(void);
void irq_set_priority(uint num, uint8_t prio);
void irq_set_enabled(uint num, bool enabled);
bool irq_is_enabled(uint num);
void irq_set_exclusive_handler(uint num, irq_handler_t handler);
void irq_add_shared_handler(uint num, irq_handler_t handler, uint8_t order_priority);
void irq_remove_handler(uint num, irq_handler_t handler);
void watchdog_update(void);
void watchdog_enable(uint32_t delay_ms, bool pause_on_debug);
void watchdog_reboot(uint32_t pc, uint32_t sp, uint32_t delay_ms);
void reset_usb_boot(uint32_t gpio_mask, uint32_t disable_mask);

/* ---- hardware/flash.h: a RAM image stands behind XIP_BASE (ref_fw.c) ---- */
extern uint8_t orc_flash_image[];
#define PICO_FLASH_SIZE_BYTES (2u * 1024u * 1024u)
#define FLASH_SECTOR_SIZE 4096u
#define FLASH_PAGE_SIZE 256u
#define FLASH_BLOCK_SIZE 65536u
#define XIP_BASE ((uintptr_t)orc_flash_image)
void flash_range_erase(uint32_t flash_offs, size_t count);
void flash_range_program(uint32_t flash_offs, const uint8_t *data, size_t count);
void flash_get_unique_id(uint8_t *id_out);

/* ---- pico/multicore.h ---- */
void multicore_launch_core1(void (*entry)(void));
void multicore_reset_core1(void);
void multicore_lockout_victim_init(void);
bool multicore_lockout_victim_is_initialized(uint core_num);
void multicore_lockout_start_blocking(void);
void multicore_lockout_end_blocking(void);
bool multicore_fifo_rvalid(void);
bool multicore_fifo_wready(void);
void multicore_fifo_push_blocking(uint32_t data);
uint32_t multicore_fifo_pop_blocking(void);
void multicore_fifo_drain(void);

/* ---- hardware/pio.h + hardware/dma.h (setup calls only; no arithmetic) ---- */
typedef struct pio_hw { io_rw_32 ctrl, fstat, fdebug, flevel, txf[4], rxf[4], irq, irq_force, inte0, inte1; } pio_hw_t;
typedef pio_hw_t *PIO;
extern pio_hw_t orc_pio_hw[3];
#define pio0 (&orc_pio_hw[0])
#define pio1 (&orc_pio_hw[1])
#define pio2 (&orc_pio_hw[2])
typedef struct pio_program { const uint16_t *instructions; uint8_t length; int8_t origin; uint8_t pio_version; } pio_program_t;
typedef struct { uint32_t clkdiv, execctrl, shiftctrl, pinctrl; } pio_sm_config;
enum pio_fifo_join { PIO_FIFO_JOIN_NONE = 0, PIO_FIFO_JOIN_TX = 1, PIO_FIFO_JOIN_RX = 2 };
uint pio_add_program(PIO pio, const pio_program_t *program);
void pio_remove_program(PIO pio, const pio_program_t *program, uint loaded_offset);
bool pio_can_add_program(PIO pio, const pio_program_t *program);
int pio_claim_unused_sm(PIO pio, bool required);
void pio_sm_claim(PIO pio, uint sm);
void pio_sm_unclaim(PIO pio, uint sm);
void pio_sm_set_enabled(PIO pio, uint sm, bool enabled);
void pio_sm_init(PIO pio, uint sm, uint initial_pc, const pio_sm_config *config);
void pio_sm_set_consecutive_pindirs(PIO pio, uint sm, uint pin_base, uint pin_count, bool is_out);
void pio_sm_set_clkdiv(PIO pio, uint sm, float div);
void pio_sm_set_clkdiv_int_frac(PIO pio, uint sm, uint16_t div_int, uint8_t div_frac);
void pio_sm_clear_fifos(PIO pio, uint sm);
void pio_sm_restart(PIO pio, uint sm);
void pio_sm_clkdiv_restart(PIO pio, uint sm);
void pio_sm_exec(PIO pio, uint sm, uint instr);
void pio_sm_put_blocking(PIO pio, uint sm, uint32_t data);
void pio_sm_drain_tx_fifo(PIO pio, uint sm);
bool pio_sm_is_tx_fifo_empty(PIO pio, uint sm);
void pio_gpio_init(PIO pio, uint pin);
uint pio_get_dreq(PIO pio, uint sm, bool is_tx);
void pio_enable_sm_mask_in_sync(PIO pio, uint32_t mask);
void pio_set_sm_mask_enabled(PIO pio, uint32_t mask, bool enabled);
pio_sm_config pio_get_default_sm_config(void);
void sm_config_set_out_pins(pio_sm_config *c, uint out_base, uint out_count);
void sm_config_set_set_pins(pio_sm_config *c, uint set_base, uint set_count);
void sm_config_set_sideset_pins(pio_sm_config *c, uint sideset_base);
void sm_config_set_sideset(pio_sm_config *c, uint bit_count, bool optional, bool pindirs);
void sm_config_set_out_shift(pio_sm_config *c, bool shift_right, bool autopull, uint pull_threshold);
void sm_config_set_fifo_join(pio_sm_config *c, enum pio_fifo_join join);
void sm_config_set_clkdiv(pio_sm_config *c, float div);
void sm_config_set_clkdiv_int_frac(pio_sm_config *c, uint16_t div_int, uint8_t div_frac);
void sm_config_set_wrap(pio_sm_config *c, uint wrap_target, uint wrap);
uint pio_encode_jmp(uint addr);

typedef struct { uint32_t ctrl; } dma_channel_config;
typedef struct { io_rw_32 read_addr, write_addr, transfer_count, ctrl_trig, al1_ctrl, al3_read_addr_trig, al1_transfer_count_trig; } dma_channel_hw_t;
typedef struct { dma_channel_hw_t ch[NUM_DMA_CHANNELS]; io_rw_32 intr, inte0, intf0, ints0, inte1, intf1, ints1, abort; } dma_hw_t;
extern dma_hw_t orc_dma_hw;
#define dma_hw (&orc_dma_hw)
enum dma_channel_transfer_size { DMA_SIZE_8 = 0, DMA_SIZE_16 = 1, DMA_SIZE_32 = 2 };
int dma_claim_unused_channel(bool required);
void dma_channel_claim(uint channel);
void dma_channel_unclaim(uint channel);
dma_channel_config dma_channel_get_default_config(uint channel);
void channel_config_set_transfer_data_size(dma_channel_config *c, enum dma_channel_transfer_size size);
void channel_config_set_read_increment(dma_channel_config *c, bool incr);
void channel_config_set_write_increment(dma_channel_config *c, bool incr);
void channel_config_set_dreq(dma_channel_config *c, uint dreq);
void channel_config_set_chain_to(dma_channel_config *c, uint chain_to);
void channel_config_set_ring(dma_channel_config *c, bool write, uint size_bits);
void channel_config_set_high_priority(dma_channel_config *c, bool high_priority);
void dma_channel_configure(uint channel, const dma_channel_config *config, volatile void *write_addr, const volatile void *read_addr,
                           uint transfer_count, bool trigger);
void dma_channel_set_irq0_enabled(uint channel, bool enabled);
void dma_channel_set_irq1_enabled(uint channel, bool enabled);
void dma_channel_start(uint channel);
void dma_channel_abort(uint channel);
bool dma_channel_is_busy(uint channel);
void dma_channel_set_read_addr(uint channel, const volatile void *read_addr, bool trigger);
void dma_channel_transfer_from_buffer_now(uint channel, const volatile void *read_addr, uint32_t transfer_count);
bool dma_channel_get_irq0_status(uint channel);
void dma_channel_acknowledge_irq0(uint channel);
bool dma_channel_get_irq1_status(uint channel);
void dma_channel_acknowledge_irq1(uint channel);
dma_channel_hw_t *dma_channel_hw_addr(uint channel);
#endif
