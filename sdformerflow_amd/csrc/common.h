// Shared helpers for the gfx950 kernels.  Wavefront = 64 lanes; no CUDA compatibility layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "../../include/sdformerflow_hip.h"
#include "launch_log.h"

#define SDF_LAUNCH_CHECK()                         \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) return (int)e__;        \
  } while (0)

static inline hipStream_t sdf_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
#include <math.h>
// LIF time constant as the C ABI carries it.  tau > 1: spikingjelly's LIFNode, v += (x - v) / tau - the reciprocal is used where it is
// exact (tau a power of two; 0 = divide).  0 < tau < 1: the multiplicative form v += (x - v) * tau of ParametricLIFNode, whose
// multiplier k = sigmoid(w) the caller passes as `tau` (reference Spiking_modules.py:75-82; spikingjelly neuron.ParametricLIFNode).
static inline bool sdf_tau_ok(int kind, float tau) { return kind != SDF_LIF || tau > 1.f || (tau > 0.f && tau < 1.f); }
static inline float sdf_inv_tau(int kind, float tau) {
  if (kind != SDF_LIF) return 0.f;
  if (tau > 0.f && tau < 1.f) return tau;
  int ex;
  return frexpf(tau, &ex) == 0.5f ? 1.0f / tau : 0.f;
}
static inline bool sdf_aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

// Dynamic LDS above 64 KiB needs an opt-in (hipFuncSetAttribute) per kernel function AND per device.  `done` = one bit per device
// ordinal, one static instance per kernel: a second GPU driven from the same process, or two host threads at their first call, both end
// up with the attribute set (setting it twice is harmless).  Returns 0 or the hipError_t.
static inline int sdf_lds_opt_in(std::atomic<uint64_t>& done, const void* fn, int bytes) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return 0;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return (int)e;
  done.fetch_or(bit, std::memory_order_release);
  return 0;
}
