// Shared definitions of the spike matrix-multiply kernels (spike_gemm.hip, spike_mm_ws.hip).
#pragma once
#include "common.h"

namespace sdfmm {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// Implicit-GEMM convolution geometry (CONV kernels): A row g = (img, oy, ox) over an OH x OW output grid; K is
// ordered (tap, channel) with taps on a KHc x KWc grid; tap (ky, kx) reads input pixel
// (oy*sy + dy[ky], ox*sx + dx[kx]) of an NHWC u8 image, zero outside [0,H) x [0,W).
struct ConvGeom {
  int H, W, Cin, OH, OW, sy, sx, KWc, kw_mul;     // kw_mul: (tap * kw_mul) >> 5 == tap / KWc for tap < 9
  int dy[3], dx[3];
};

struct GemmParams {
  SdfSpikeGemmDesc d;
  ConvGeom cv;
  int tiles_m, tiles_n, ntiles;
  float inv_tau;
  int ksplit, spc;            // split-K (warp-specialised kernel): K chunks per tile, stages per chunk
  float* partial;             // [ksplit][M][N] fp32 partial sums in the caller's workspace
};

// 8 spike bytes {0,1} -> 8 bf16 {0, 1.0}
__device__ __forceinline__ bf16x8 expand_spikes(uint2 v) {
  union { bf16x8 h; uint32_t u[4]; } r;
  r.u[0] = __builtin_amdgcn_perm(0u, v.x, 0x0c010c00u) * 0x3F80u;   // (b0 | b1 << 16) * bf16(1.0)
  r.u[1] = __builtin_amdgcn_perm(0u, v.x, 0x0c030c02u) * 0x3F80u;
  r.u[2] = __builtin_amdgcn_perm(0u, v.y, 0x0c010c00u) * 0x3F80u;
  r.u[3] = __builtin_amdgcn_perm(0u, v.y, 0x0c030c02u) * 0x3F80u;
  return r.h;
}


// warp-specialised kernel (spike_mm_ws.hip): 256 x 96 tiles, N % 96 == 0; returns 0 or an SDF_E_* / hipError code
int launch_spike_mm_ws(const GemmParams& P, bool conv, hipStream_t s);

}  // namespace sdfmm
