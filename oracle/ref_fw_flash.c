/*
 * ref_fw_flash.c — the reference's flash_storage.c compiled IN PLACE over a RAM-backed flash image.
 * TEST INFRASTRUCTURE ONLY; part of oracle/_ref/libref_fw_*.so (see ref_fw.c).
 *
 * flash_storage.c reads flash through XIP_BASE + offset and writes it through dspi_flash_range_erase/program
 * (flash_storage.c:61-63, :336-339); ref_stub_sdk/pico_stub_all.h points XIP_BASE at orc_flash_image and
 * ref_fw_stubs.c implements erase = memset 0xFF, program = memcpy.  The file is #included (not linked) because
 * the test hooks below need three of its statics: dir_cache, dir_cache_valid, collect_live_state.
 * Pinned by running it: preset_save / preset_load / preset_delete / preset_boot_load / dir_load_cache (v1 -> v2
 * migration) / migrate_legacy / apply_factory_defaults / apply_slot_to_live (flash_storage.c:370-417, :464-742,
 * :750-849, :997-1105, :1144-1238).
 */
#include <string.h>
#include <math.h>
#include <stdint.h>
#include <stdbool.h>
#include <stddef.h>
#include "pico_stub_all.h"
static inline uint32_t __get_current_exception(void) { return 0; }     /* main-loop context (flash_storage.c:331) */

#include "flash_storage.c"

int fw_flash_slot_size(void) { return (int)sizeof(PresetSlot); }
void fw_flash_forget_dir(void) { dir_cache_valid = false; }
/* the slot sector was written behind the directory's back (tests inject slot images): mark it occupied in the cache */
void fw_flash_mark_occupied(int slot) { dir_ensure(); dir_cache.slot_occupied |= (uint16_t)(1u << slot); }
void fw_flash_collect_slot(void *image, int slot_index) {
    static PresetSlot s;
    collect_live_state(&s, (uint8_t)slot_index);
    memcpy(image, &s, sizeof s);
}
