// mock_prefix.h — force-included when the library's HOST sources are compiled with g++ for the CPU mock device
// (tests/mock/): supplies the two device intrinsics jv_device.h mentions so that the header parses in a host-only build.
#pragma once
#include <cstdint>
#include <cstring>
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }

// the mock device keeps the measured-and-switched-off kernel variants compiled (make EXPERIMENTAL=1 on the GPU side): they stay under test
#ifndef JV_EXPERIMENTAL
#define JV_EXPERIMENTAL 1
#endif
