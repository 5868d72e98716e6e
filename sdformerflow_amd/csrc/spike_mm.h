// Shared definitions of the spike matrix-multiply kernels (spike_gemm.hip, spike_mm_pp.hip, spike_splitk.hip).
#pragma once
#include "common.h"
#include <math.h>

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
  float acc_scale;            // multiplies the accumulator before the epilogue (1 / weight scale of the fp16 planes; else 1)
  int cb_inner = 0;           // digit convolution: workgroup -> (tile range, column block) with the column block fastest (one XCD per range)
};

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

// Weight planes: NSPLIT = 1 / 3 -> bf16 planes (W = p0 [+ p1 + p2]); NSPLIT = 2 -> two fp16 planes of wscale * W.
// Binary spikes are exact in both 16-bit formats; the accumulation is fp32 inside the matrix cores.
template <int NSPLIT>
__device__ __forceinline__ f32x16 mma(bf16x8 a, bf16x8 b, f32x16 c) {
  if constexpr (NSPLIT == 2)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// 8 activation bytes -> 8 x 16-bit values.  fp16 planes (NSPLIT == 2): ANY byte n in 0..255 becomes fp16(n) exactly - the byte
// pair {n, 0x64} is fp16(1024 + n), one packed add takes the 1024 off - so the A operand may be a spike (0 / 1) or a SUM of spikes
// (the SEW stream between blocks, reference Spiking_swin_transformer3D.py:840-845: small non-negative integers); same instruction
// count as the spike-only form.  bf16 planes: spikes only, (b0 | b1 << 16) * 0x3F80.
template <int NSPLIT>
__device__ __forceinline__ bf16x8 expand_spikes(uint2 v) {
  union { bf16x8 h; uint32_t u[4]; } r;
  if constexpr (NSPLIT == 2) {
    typedef __attribute__((ext_vector_type(2))) _Float16 h2v;
    const h2v off = {(_Float16)-1024.0f, (_Float16)-1024.0f};
    const uint32_t sel[2] = {0x04010400u, 0x04030402u};                 // result bytes {b_lo, 0x64, b_hi, 0x64}
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t w = __builtin_amdgcn_perm(0x64646464u, i < 2 ? v.x : v.y, sel[i & 1]);
      r.u[i] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2v, w) + off);
    }
  } else {
    constexpr uint32_t ONE = 0x3F80u;
    r.u[0] = __builtin_amdgcn_perm(0u, v.x, 0x0c010c00u) * ONE;
    r.u[1] = __builtin_amdgcn_perm(0u, v.x, 0x0c030c02u) * ONE;
    r.u[2] = __builtin_amdgcn_perm(0u, v.y, 0x0c010c00u) * ONE;
    r.u[3] = __builtin_amdgcn_perm(0u, v.y, 0x0c030c02u) * ONE;
  }
  return r.h;
}


// LIF / IF recurrence over the T pre-activations a lane holds (same separately-rounded op sequence as neuron.hip).
// All flags are wave-uniform; the shipped configuration (LIF, soft reset, power-of-two tau) gets a branch-free body.
template <int T>
__device__ __forceinline__ void lif_steps(const float (&xs)[T], float (&sp)[T], int kind, bool soft, float v_reset, float v_th,
                                          float tau, float inv_tau) {
  if (kind == SDF_LIF && soft && inv_tau != 0.f) {
    float v = 0.f;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const float hcur = v + (xs[t] - v) * inv_tau;
      sp[t] = (hcur - v_th >= 0.f) ? 1.f : 0.f;
      v = hcur - sp[t] * v_th;
    }
    return;
  }
  const bool reset0 = soft || v_reset == 0.f;
  float v = soft ? 0.f : v_reset;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    float hcur;
    if (kind == SDF_IF) {
      hcur = v + xs[t];
    } else {
      const float dl = reset0 ? (xs[t] - v) : (xs[t] - (v - v_reset));
      hcur = v + ((inv_tau != 0.f) ? dl * inv_tau : dl / tau);
    }
    sp[t] = (hcur - v_th >= 0.f) ? 1.f : 0.f;
    v = soft ? (hcur - sp[t] * v_th) : ((1.f - sp[t]) * hcur + sp[t] * v_reset);
  }
}

// neuron over the T values a lane holds.  NK (compile time, both neurons of an MLP share it): 0 = the shipped LIF (soft reset,
// tau a power of two: straight-line body), 1 = PSN (the k-ordered fmaf chain of neuron.hip), 2 = any other LIF / IF setting
template <int NK, int T>
__device__ __forceinline__ void neuron_T(const float (&xs)[T], float (&sp)[T], const SdfNeuronCfg& n, float inv_tau) {
  if constexpr (NK == 1 && T <= 4) {
    // a 2 x 2 (T' of the window attention) up to 4 x 4 matrix: straight-line, the coefficients are loop-invariant scalar loads the
    // compiler keeps in registers across calls (the rolled form below re-reads a row per iteration: one scalar-load latency each,
    // 70 calls per workgroup in qk_front.hip)
#pragma unroll
    for (int t = 0; t < T; ++t) {
      float hh = n.psn_b[t];
#pragma unroll
      for (int k = 0; k < T; ++k) hh = __builtin_fmaf(n.psn_w[t * T + k], xs[k], hh);
      sp[t] = hh >= 0.f ? 1.f : 0.f;
    }
  } else if constexpr (NK == 1) {
    // the row loop is kept rolled (one row of T coefficients in scalar registers at a time); decisions travel as a bit mask
    uint32_t m = 0;
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
      const float* w = n.psn_w + t * T;
      float hh = n.psn_b[t];
#pragma unroll
      for (int k = 0; k < T; ++k) hh = __builtin_fmaf(w[k], xs[k], hh);
      m |= (hh >= 0.f ? 1u : 0u) << t;
    }
#pragma unroll
    for (int t = 0; t < T; ++t) sp[t] = ((m >> t) & 1u) ? 1.f : 0.f;
  } else if constexpr (NK == 0) {
    // h = v + (x - v) / tau; s = (h - v_th >= 0); v = h - s * v_th: with s in {0, 1} the last line is h - v_th (the difference the
    // comparison already holds) or h itself, bit for bit
    float v = 0.f;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const float hcur = v + (xs[t] - v) * inv_tau;
      const float dth = hcur - n.v_th;
      const bool fire = dth >= 0.f;
      sp[t] = fire ? 1.f : 0.f;
      v = fire ? dth : hcur;
    }
  } else {
    lif_steps<T>(xs, sp, n.kind, n.soft_reset != 0, n.v_reset, n.v_th, n.tau, inv_tau);
  }
}

// PSN with its coefficients staged in LDS: table = T rows of PSN_TP(T) floats (row t = W[t][0..T-1], zero padded) followed by the T biases.
// The rolled scalar-load form of neuron_T<1, T> pays one scalar-memory latency per row and call (T = 10: the one-launch MLP ran 96 us
// with the PSN against 47 us with the LIF, the prediction heads 35-43 against 15-24); here a row is three wave-uniform 16-byte LDS
// reads (broadcasts) and the next row's reads overlap this row's chain.  Same k-ordered fmaf chain: bit-equal.
__host__ __device__ constexpr int PSN_TP(int T) { return (T + 3) & ~3; }
__host__ __device__ constexpr int PSN_TABLE(int T) { return T * PSN_TP(T) + PSN_TP(T); }          // floats
template <int T>
__device__ __forceinline__ void psn_stage(float* tbl, const SdfNeuronCfg& n, int tid, int nthreads) {
  constexpr int TP = PSN_TP(T);
  for (int i = tid; i < PSN_TABLE(T); i += nthreads) {
    const int r = i / TP, k = i - r * TP;
    tbl[i] = r < T ? (k < T ? n.psn_w[r * T + k] : 0.f) : (k < T ? n.psn_b[k] : 0.f);
  }
}
// decisions as a bit mask (bit t = step t)
template <int T, int UNR = 2>
__device__ __forceinline__ uint32_t psn_T_lds_bits(const float (&xs)[T], const float* tbl) {
  constexpr int TP = PSN_TP(T);
  uint32_t m = 0;
#pragma unroll UNR
  for (int t = 0; t < T; ++t) {
    float hh = tbl[T * TP + t];
#pragma unroll
    for (int k4 = 0; k4 < TP / 4; ++k4) {                      // (four coefficients at a time: no whole row in registers)
      const float4 q = *reinterpret_cast<const float4*>(tbl + t * TP + 4 * k4);
      if (4 * k4 < T) hh = __builtin_fmaf(q.x, xs[4 * k4 < T ? 4 * k4 : 0], hh);
      if (4 * k4 + 1 < T) hh = __builtin_fmaf(q.y, xs[4 * k4 + 1 < T ? 4 * k4 + 1 : 0], hh);
      if (4 * k4 + 2 < T) hh = __builtin_fmaf(q.z, xs[4 * k4 + 2 < T ? 4 * k4 + 2 : 0], hh);
      if (4 * k4 + 3 < T) hh = __builtin_fmaf(q.w, xs[4 * k4 + 3 < T ? 4 * k4 + 3 : 0], hh);
    }
    m |= (hh >= 0.f ? 1u : 0u) << t;
  }
  return m;
}
template <int T, int UNR = 2>
__device__ __forceinline__ void psn_T_lds(const float (&xs)[T], float (&sp)[T], const float* tbl) {
  constexpr int TP = PSN_TP(T);
  uint32_t m = 0;
#pragma unroll UNR
  for (int t = 0; t < T; ++t) {
    float w[TP];
#pragma unroll
    for (int k4 = 0; k4 < TP / 4; ++k4) {
      const float4 q = *reinterpret_cast<const float4*>(tbl + t * TP + 4 * k4);
      w[4 * k4] = q.x; w[4 * k4 + 1] = q.y; w[4 * k4 + 2] = q.z; w[4 * k4 + 3] = q.w;
    }
    float hh = tbl[T * TP + t];
#pragma unroll
    for (int k = 0; k < T; ++k) hh = __builtin_fmaf(w[k], xs[k], hh);
    m |= (hh >= 0.f ? 1u : 0u) << t;
  }
#pragma unroll
  for (int t = 0; t < T; ++t) sp[t] = ((m >> t) & 1u) ? 1.f : 0.f;
}

// neuron of class NK over the T values a lane holds, PSN coefficients from the LDS table `tbl` (psn_stage) where T > 4
template <int NK, int T, bool LEAN = false>
__device__ __forceinline__ void neuron_any(const float (&xs)[T], float (&sp)[T], const SdfNeuronCfg& n, float inv_tau, const float* tbl) {
  if constexpr (NK == 1 && (T > 4)) psn_T_lds<T, LEAN ? 1 : 2>(xs, sp, tbl);      // (LEAN: one row's coefficients in registers at a time)
  else neuron_T<NK, T>(xs, sp, n, inv_tau);
}

// the same, decisions as a bit mask (bit t = step t): the PSN never forms its T spike floats
template <int NK, int T, bool LEAN = false>
__device__ __forceinline__ uint32_t neuron_any_bits(const float (&xs)[T], const SdfNeuronCfg& n, float inv_tau, const float* tbl) {
  if constexpr (NK == 1 && (T > 4)) {
    return psn_T_lds_bits<T, LEAN ? 1 : 2>(xs, tbl);
  } else {
    float sp[T];
    neuron_T<NK, T>(xs, sp, n, inv_tau);
    uint32_t m = 0;
#pragma unroll
    for (int t = 0; t < T; ++t) m |= ((__float_as_uint(sp[t]) >> 29) & 1u) << t;      // 1.0f has bit 29 set
    return m;
  }
}

bool qk_front_supports(const SdfQkAttnDesc* d);
int launch_qk_front(const SdfQkAttnDesc* d, uint8_t* e, uint8_t* qk, bool keep, hipStream_t s);

// host side: 1 / tau where that is exact (tau a power of two; 0 = divide), and the compile-time class of a neuron setting
static inline float inv_tau_of(const SdfNeuronCfg& n) {
  return sdf_inv_tau(n.kind, n.tau);
}
static inline int neuron_class(const SdfNeuronCfg& n) {
  if (n.kind == SDF_PSN) return 1;
  return (n.kind == SDF_LIF && n.soft_reset != 0 && inv_tau_of(n) != 0.f) ? 0 : 2;
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

// ping-pong kernel, several convolutions of one family (same images / output / epilogue, own taps, weights, row map) in one launch;
// SDF_E_SHAPE when they are not one family or any of them wants split-K
int launch_spike_mm_pp_multi(const GemmParams* Ps, int n, hipStream_t s);
// ping-pong kernel (spike_mm_pp.hip): same tiles, consumer groups alternate tiles so epilogues overlap the MFMAs
bool spike_mm_pp_supports(const GemmParams& P, bool conv);
int launch_spike_mm_pp(const GemmParams& P, bool conv, hipStream_t s);
// weight-resident 3x3 / stride-1 spike convolution (spike_conv_wres.hip): one 32-column block's weights stay in LDS, the
// activations enter as halo tiles (no im2col); `supports` is false for shapes it has no instantiation for
bool spike_conv_wres_supports(const GemmParams& P, bool any_size);
int launch_spike_conv_wres(const GemmParams& P, hipStream_t s);
// MS MLP as one launch (ms_mlp_fused.hip); keep_s1 / keep_s2: optional u8 copies of the SN1 / SN2 spikes (parity tape)
bool ms_mlp_fused_supports(const SdfMsMlpDesc* d);
int launch_ms_mlp_fused(const SdfMsMlpDesc* d, uint8_t* keep_s1, uint8_t* keep_s2, hipStream_t s);
// wide stages (ms_wide.hip): the block's matrix products as position-major kernels with the neurons in their epilogues
bool ms_wide_mlp_supports(const SdfMsMlpDesc* d);
int launch_ms_wide_mlp(const SdfMsMlpDesc* d, const uint8_t* s1, bool s1_tiled, uint8_t* s2, bool s2_tiled, hipStream_t s);
bool ms_wide_attn_supports(const SdfQkAttnDesc* d);
int launch_ms_wide_front(const SdfQkAttnDesc* d, const uint8_t* xs, uint8_t* e, uint8_t* qk, bool keep, hipStream_t s);
int launch_ms_wide_proj(const SdfQkAttnDesc* d, const uint8_t* e, hipStream_t s);
bool smallm_conv_supports(const GemmParams& P);          // ms_smallm.hip: few rows against many weights, K split inside the workgroup
int launch_smallm_conv(const GemmParams& P, hipStream_t s);
bool smallm_fc2_supports(const SdfMsMlpDesc* d);          // the wide-stage MLP's second product on the small-M kernel
int launch_smallm_fc2(const SdfMsMlpDesc* d, const uint8_t* s2, hipStream_t s);
bool smallm_gemm_supports(const GemmParams& P);          // plain rows, fp32 epilogue (the decoders' stacked-tap product)
int launch_smallm_gemm(const GemmParams& P, hipStream_t s);
bool wide_merge_supports(const SdfMsMergeDesc* d);
int launch_wide_merge(const SdfMsMergeDesc* d, hipStream_t s);
// 3x3 / stride 1 convolution of few rows against many weights on the wide-stage main loop (split-K + one reduce / BN / neuron pass)
bool wide_conv_supports(const GemmParams& P);
int launch_wide_conv(const GemmParams& P, hipStream_t s);
// split-K planning (fills ksplit / spc / partial from the descriptor's workspace) and the k-ordered second pass
void plan_splitk(GemmParams& P, int kc);
int launch_splitk_reduce(const GemmParams& P, hipStream_t s);

}  // namespace sdfmm
