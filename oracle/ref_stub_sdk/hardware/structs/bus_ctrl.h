/* Host stand-in for hardware/structs/bus_ctrl.h (oracle/ref_fw_main.c; test infrastructure only): main.c raises the DMA bus priority. */
#pragma once
#include "pico_stub_all.h"
typedef struct { volatile uint32_t priority; } orc_bus_ctrl_hw_t;
extern orc_bus_ctrl_hw_t orc_bus_ctrl_hw;
#define bus_ctrl_hw (&orc_bus_ctrl_hw)
#define BUSCTRL_BUS_PRIORITY_DMA_W_BITS 0x1000u
#define BUSCTRL_BUS_PRIORITY_DMA_R_BITS 0x0100u
#ifndef USBCTRL_IRQ
#define USBCTRL_IRQ 14
#endif
