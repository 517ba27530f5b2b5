// Host-side helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/ubresnet_hip.h"

extern "C" void ubr_set_error(const char* fmt, ...);

#define UBR_CHECK(cond, ...)                      \
  do {                                            \
    if (!(cond)) {                                \
      ubr_set_error(__VA_ARGS__);                 \
      return UBR_EINVAL;                          \
    }                                             \
  } while (0)

#define UBR_LAUNCH_CHECK(name)                                                   \
  do {                                                                           \
    hipError_t e__ = hipGetLastError();                                          \
    if (e__ != hipSuccess) {                                                     \
      ubr_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));      \
      return UBR_ELAUNCH;                                                        \
    }                                                                            \
  } while (0)

static inline int ubr_esize(int dtype) { return dtype == UBR_F32 ? 4 : 2; }
static inline int ubr_cpu(int dtype) { return dtype == UBR_F32 ? 4 : 8; }
static inline bool ubr_dtype_ok(int dtype) { return dtype == UBR_F32 || dtype == UBR_BF16 || dtype == UBR_F16; }
static inline int ubr_ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
static inline bool ubr_aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
