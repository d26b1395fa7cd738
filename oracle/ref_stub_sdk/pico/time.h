/* oracle/ref_stub_sdk/pico/time.h — host stand-in for the un-vendored pico-sdk header of the same name (TEST INFRASTRUCTURE, see pico_stub_all.h). */
#pragma once
#include "pico_stub_all.h"
