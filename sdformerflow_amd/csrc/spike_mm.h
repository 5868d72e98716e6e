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


// 4x4 transpose inside every quad of lanes (two DPP butterflies): in: lane q holds v[j] = X[row j][col q];
// out: lane q holds v[k] = X[row q][col k], i.e. 16 contiguous bytes of one output row.
__device__ __forceinline__ void quad_transpose(float (&v)[4], int q) {
  const bool o1 = q & 1, o2 = q & 2;
  // butterfly 1: lanes q <-> q^1 on (v0,v1) and (v2,v3)
  {
    const float s0 = o1 ? v[0] : v[1], s1 = o1 ? v[2] : v[3];
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s0), 0xB1, 0xF, 0xF, false));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s1), 0xB1, 0xF, 0xF, false));
    if (o1) { v[0] = r0; v[2] = r1; } else { v[1] = r0; v[3] = r1; }
  }
  // butterfly 2: lanes q <-> q^2 on the pairs
  {
    const float s0 = o2 ? v[0] : v[2], s1 = o2 ? v[1] : v[3];
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s0), 0x4E, 0xF, 0xF, false));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s1), 0x4E, 0xF, 0xF, false));
    if (o2) { v[0] = r0; v[1] = r1; } else { v[2] = r0; v[3] = r1; }
  }
}

// warp-specialised kernel (spike_mm_ws.hip): 256 x 96 tiles, N % 96 == 0; returns 0 or an SDF_E_* / hipError code
int launch_spike_mm_ws(const GemmParams& P, bool conv, hipStream_t s);

}  // namespace sdfmm
