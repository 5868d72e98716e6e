// Shared helpers for the gfx950 kernels.  Wavefront = 64 lanes; no CUDA compatibility layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/sdformerflow_hip.h"

#define SDF_LAUNCH_CHECK()                         \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) return (int)e__;        \
  } while (0)

static inline hipStream_t sdf_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline bool sdf_aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }
