// MS MLP of a swin block as ONE launch for gfx950 (row a7 of SURVEY.md section 8):
//
//   x += BN2( SN2( BN1( SN1(x) W1^T ) ) W2^T )        reference Spiking_swin_transformer3D.py:164-181, :845
//
// on a (B, D = T, HW, C) fp32 channel-last buffer, neurons over the true time axis T.  The three-launch form
// (qk_attn.hip: neuron -> fc1 + BN + neuron -> fc2 + BN + shortcut) sends the spikes of SN1 (1 B/element), the hidden
// spikes (4C wide, 1 B/element) and x twice through HBM / L2; here a workgroup owns a tile of POSITIONS (pixels) with all
// T time steps of each and nothing but x itself leaves the compute unit:
//
//   1. SN1: every thread takes (position, 4 channels), reads the T float4 of that column from x, runs the neuron in
//      registers and writes the T x 4 spike bytes into the LDS image A1 [rows][C] (the fc1 A operand);
//   2. the hidden dimension is walked in chunks of CH columns.  Per chunk: fc1 on the matrix cores
//      (v_mfma_f32_16x16x32, spikes expanded to 16-bit in registers, weight planes hi / lo from LDS) -> BN1 -> SN2 over T
//      in the accumulator registers -> spike bytes into the LDS image A2 [rows][CH] -> fc2 partial products accumulated
//      in registers over all chunks (the k order of a plain K loop);
//   3. epilogue: BN2 + shortcut, 16-byte loads / stores of x through a quad transpose.
//
// Row order ("time-major in the lane"): a 16 x 16 MFMA block leaves lane l with column l % 16 of rows 4 (l / 16) + 0..3.
// A wave owns RB = 5 row blocks = 80 tile rows, i.e. 20 accumulator slots per lane and column; slot s = 4 rb + i holds
// (position pp = s / T of the lane's quarter, step t = s % T): T = 10 -> 2 positions per quarter, 8 per wave, no dead slot
// (T = 20: 4 per wave, T = 5: 16).  So the LIF / IF / PSN recurrence of SN2 runs on registers straight out of the MFMA.
//
// Waves = RG row groups x CG column groups: column group cg computes 16 NB1 of a chunk's hidden columns in fc1 and
// C / CG of the output columns in fc2, so no wave holds more than 5 x (NB1 + NB2) accumulator quads and the A2 image is the
// only thing the column groups of a row group exchange (one workgroup barrier per phase).  Weight chunks are fetched
// global -> registers one phase ahead and stored to the single LDS weight buffer between the barriers.
// Compiled with -ffp-contract=off: neuron arithmetic is the separately-rounded op sequence of neuron.hip.
#include "spike_mm.h"
#include <stdlib.h>

namespace sdfmm {
namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

struct MlpFusedParams {
  float* x;
  const uint16_t* W1;        // [NSPLIT][Ch][C]
  const uint16_t* W2;        // [NSPLIT][C][Ch]
  const float *a1, *b1, *a2, *b2;
  float asc1, asc2;
  int HW, Ch;
  int64_t P;                 // positions = B * HW
  SdfNeuronCfg sn1, sn2;
  float inv_tau1, inv_tau2;
  uint8_t* keep_s1;          // optional (parity tape): SN1 spikes [tokens][C]
  uint8_t* keep_s2;          // optional: SN2 spikes [tokens][Ch]
};

template <int NSPLIT>
__device__ __forceinline__ f32x4 mma16(bf16x8 a, bf16x8 b, f32x4 c) {
  if constexpr (NSPLIT == 2)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, a),
                                                   __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, b), c, 0, 0, 0);
}

// neuron over the T values a lane holds (LIF / IF: spike_mm.h; PSN: the k-ordered fmaf chain of neuron.hip)
template <int T>
__device__ __forceinline__ void neuron_T(const float (&xs)[T], float (&sp)[T], const SdfNeuronCfg& n, float inv_tau) {
  if (n.kind == SDF_PSN) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      float hh = n.psn_b[t];
#pragma unroll
      for (int k = 0; k < T; ++k) hh = __builtin_fmaf(n.psn_w[t * T + k], xs[k], hh);
      sp[t] = hh >= 0.f ? 1.f : 0.f;
    }
  } else {
    lif_steps<T>(xs, sp, n.kind, n.soft_reset != 0, n.v_reset, n.v_th, n.tau, inv_tau);
  }
}

template <int NSPLIT, int T, int C16, int CG, int NB1, int RG>
struct MlpGeo {
  static constexpr int C = 16 * C16, NB2 = C16 / CG, CH = 16 * NB1 * CG;
  static constexpr int NW = RG * CG, NT = 64 * NW;
  static constexpr int RB = 5, ROWS = 16 * RB, SLOTS = 4 * RB;
  static constexpr int PPG = SLOTS / T;            // positions per lane quarter
  static constexpr int PPW = 4 * PPG;              // positions per row group (wave)
  static constexpr int PPI = PPW * RG;             // positions per work item (workgroup)
  static constexpr int A1P = C + 16, A2P = CH + 16;                 // LDS row pitches (bytes): pitch / 4 = 4 x odd dwords
  static constexpr int W1P = (C + 8) * 2, W2P = (CH + 8) * 2;       // weight row pitches (bytes)
  static constexpr int W1B = NSPLIT * CH * W1P, W2B = NSPLIT * C * W2P;
  static constexpr int WB = W1B > W2B ? W1B : W2B;
  static constexpr int A1B = RG * ROWS * A1P, A2B = RG * ROWS * A2P;
  static constexpr int LDS = A1B + A2B + WB;
  static constexpr int WPIECES = NSPLIT * CH * C / 8;               // 16-byte pieces of a weight chunk (fc1 and fc2 alike)
  static constexpr int WIT = (WPIECES + NT - 1) / NT;
  static_assert(C16 % CG == 0, "output columns must split evenly over the column groups");
  static_assert(SLOTS % T == 0, "T must divide the 20 accumulator slots of a lane");
  static_assert(CH % 32 == 0 && C % 32 == 0, "K steps are 32 deep");
  static_assert((A1P / 4) % 8 == 4 && (A2P / 4) % 8 == 4, "spike image pitches must be 4 x odd dwords");
  static_assert((W1P / 4) % 8 == 4 && (W2P / 4) % 8 == 4, "weight pitches must be 4 x odd dwords");
  static_assert(LDS <= 160 * 1024, "LDS budget");
};

template <int NSPLIT, int T, int C16, int CG, int NB1, int RG>
__global__ __launch_bounds__(64 * RG * CG) void ms_mlp_fused_kernel(MlpFusedParams P) {
  using G = MlpGeo<NSPLIT, T, C16, CG, NB1, RG>;
  constexpr int C = G::C, NB2 = G::NB2, CH = G::CH, NT = G::NT, RB = G::RB, ROWS = G::ROWS;
  constexpr int PPG = G::PPG, PPW = G::PPW, PPI = G::PPI;
  constexpr int A1P = G::A1P, A2P = G::A2P, W1P = G::W1P, W2P = G::W2P, WIT = G::WIT, WPIECES = G::WPIECES;
  __shared__ __attribute__((aligned(16))) uint8_t smem[G::LDS];
  uint8_t* A1 = smem;
  uint8_t* A2 = smem + G::A1B;
  uint8_t* Wb = smem + G::A1B + G::A2B;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int rg = wave / CG, cg = wave - rg * CG;
  const int l16 = lane & 15, lq = lane >> 4;
  const int HW = P.HW, Ch = P.Ch;
  const int64_t item = blockIdx.x;
  const int64_t tstride = (int64_t)HW * C;                             // elements between two time steps of a position
  const int nchunks = Ch / CH;

  // ---------------- weight chunk loader: global -> registers now, registers -> LDS between two barriers ----------------
  u32x4 wreg[WIT];
  auto w_load = [&](bool second, int j) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < WIT; ++i) {
      const int c = tid + NT * i;
      const int cc = c < WPIECES ? c : 0;
      const uint16_t* src;
      if (!second) {                                                    // W1 chunk: rows = plane * CH + hidden column, C / 8 pieces per row
        const int row = cc / (C / 8), c8 = cc - row * (C / 8);
        const int p = row / CH, r = row - p * CH;
        src = P.W1 + ((int64_t)p * Ch + (int64_t)j * CH + r) * C + 8 * c8;
      } else {                                                          // W2 chunk: rows = plane * C + output column, CH / 8 pieces per row
        const int row = cc / (CH / 8), c8 = cc - row * (CH / 8);
        const int p = row / C, r = row - p * C;
        src = P.W2 + ((int64_t)p * C + r) * Ch + (int64_t)j * CH + 8 * c8;
      }
      wreg[i] = *reinterpret_cast<const u32x4*>(src);
    }
  };
  auto w_store = [&](bool second) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < WIT; ++i) {
      const int c = tid + NT * i;
      if (c < WPIECES) {
        if (!second) {
          const int row = c / (C / 8), c8 = c - row * (C / 8);
          *reinterpret_cast<u32x4*>(Wb + row * W1P + c8 * 16) = wreg[i];
        } else {
          const int row = c / (CH / 8), c8 = c - row * (CH / 8);
          *reinterpret_cast<u32x4*>(Wb + row * W2P + c8 * 16) = wreg[i];
        }
      }
    }
  };
  w_load(false, 0);

  // ---------------- 1. SN1 over T of every (position, channel quad) of the item -> A1 ----------------
  {
    constexpr int C4 = C / 4, NI = PPI * C4;
    const bool keep = P.keep_s1 != nullptr;
#pragma unroll 1
    for (int i = tid; i < NI; i += NT) {
      const int q = i / C4, c4 = i - q * C4;                             // position inside the item, channel quad
      const int64_t pos = item * PPI + q;
      const bool ok = pos < P.P;
      const int64_t pc = ok ? pos : 0;
      const int64_t b = pc / HW, hw = pc - b * HW;
      const float* src = P.x + ((b * T) * HW + hw) * C + 4 * c4;
      float4 v[T];
#pragma unroll
      for (int t = 0; t < T; ++t) v[t] = *reinterpret_cast<const float4*>(src + t * tstride);
      uint32_t pk[T];
#pragma unroll
      for (int t = 0; t < T; ++t) pk[t] = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float xs[T], sp[T];
#pragma unroll
        for (int t = 0; t < T; ++t) xs[t] = e == 0 ? v[t].x : (e == 1 ? v[t].y : (e == 2 ? v[t].z : v[t].w));
        neuron_T<T>(xs, sp, P.sn1, P.inv_tau1);
#pragma unroll
        for (int t = 0; t < T; ++t) pk[t] |= ((__float_as_uint(sp[t]) >> 29) & 1u) << (8 * e);      // 1.0f has bit 29 set
      }
      const int rgq = q / PPW, qq = q - rgq * PPW;
      const int g = qq / PPG, pp = qq - g * PPG;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const int slot = pp * T + t;
        const int row = 16 * (slot >> 2) + 4 * g + (slot & 3);
        *reinterpret_cast<uint32_t*>(A1 + (rgq * ROWS + row) * A1P + 4 * c4) = ok ? pk[t] : 0u;
        if (keep && ok) *reinterpret_cast<uint32_t*>(P.keep_s1 + (((b * T) + t) * HW + hw) * C + 4 * c4) = pk[t];
      }
    }
  }
  w_store(false);
  __syncthreads();                                                      // A1 and W1 chunk 0 are in LDS

  // ---------------- 2. hidden chunks ----------------
  f32x4 acc2[RB][NB2];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int nb = 0; nb < NB2; ++nb) acc2[rb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  const uint8_t* a1_lane = A1 + (rg * ROWS + l16) * A1P + 8 * lq;       // + 16 rb rows + 32 ks
  const uint8_t* a2_lane = A2 + (rg * ROWS + l16) * A2P + 8 * lq;
  const int hc0 = cg * NB1 * 16;                                        // first hidden column of this wave inside a chunk
  const int oc0 = cg * NB2 * 16;                                        // first output column of this wave
  const uint8_t* w1_lane = Wb + (hc0 + l16) * W1P + 16 * lq;            // + plane * CH rows + 16 nb rows + 64 ks bytes
  const uint8_t* w2_lane = Wb + (oc0 + l16) * W2P + 16 * lq;
  const int64_t pos_lane0 = item * PPI + rg * PPW + PPG * lq;           // first position of this lane's quarter
  const bool keep2 = P.keep_s2 != nullptr;

#pragma unroll 1
  for (int j = 0; j < nchunks; ++j) {
    w_load(true, j);                                                    // W2 chunk j: in flight during fc1
    // ---- fc1: [80 rows x C] x [C x 16 NB1] ----
    f32x4 acc1[RB][NB1];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int nb = 0; nb < NB1; ++nb) acc1[rb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < C / 32; ++ks) {
      bf16x8 a[RB];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) a[rb] = expand_spikes<NSPLIT>(*reinterpret_cast<const uint2*>(a1_lane + 16 * rb * A1P + 32 * ks));
#pragma unroll
      for (int nb = 0; nb < NB1; ++nb)
#pragma unroll
        for (int p = 0; p < NSPLIT; ++p) {
          const bf16x8 bw = *reinterpret_cast<const bf16x8*>(w1_lane + (p * CH + 16 * nb) * W1P + 64 * ks);
#pragma unroll
          for (int rb = 0; rb < RB; ++rb) acc1[rb][nb] = mma16<NSPLIT>(a[rb], bw, acc1[rb][nb]);
        }
    }
    // ---- BN1 + SN2 over T in the accumulator slots -> A2 ----
#pragma unroll
    for (int nb = 0; nb < NB1; ++nb) {
      const int nloc = hc0 + 16 * nb + l16;
      const int n = j * CH + nloc;
      const float al = P.a1[n], be = P.b1[n];
#pragma unroll
      for (int pp = 0; pp < PPG; ++pp) {
        float xs[T], sp[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const int slot = pp * T + t;
          xs[t] = __builtin_fmaf(acc1[slot >> 2][nb][slot & 3] * P.asc1, al, be);
        }
        neuron_T<T>(xs, sp, P.sn2, P.inv_tau2);
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const int slot = pp * T + t;
          const int row = 16 * (slot >> 2) + 4 * lq + (slot & 3);
          A2[(rg * ROWS + row) * A2P + nloc] = (uint8_t)(sp[t] != 0.f);
        }
        if (keep2) {
          const int64_t pos = pos_lane0 + pp;
          if (pos < P.P) {
            const int64_t b = pos / HW, hw = pos - b * HW;
#pragma unroll
            for (int t = 0; t < T; ++t) P.keep_s2[(((b * T) + t) * HW + hw) * Ch + n] = (uint8_t)(sp[t] != 0.f);
          }
        }
      }
    }
    __syncthreads();                                                    // every wave is done with W1 chunk j; A2 is complete
    w_store(true);
    __syncthreads();                                                    // W2 chunk j is in LDS
    if (j + 1 < nchunks) w_load(false, j + 1);                           // W1 chunk j + 1: in flight during fc2
    // ---- fc2 partial: [80 rows x CH] x [CH x 16 NB2] ----
#pragma unroll
    for (int ks = 0; ks < CH / 32; ++ks) {
      bf16x8 a[RB];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) a[rb] = expand_spikes<NSPLIT>(*reinterpret_cast<const uint2*>(a2_lane + 16 * rb * A2P + 32 * ks));
#pragma unroll
      for (int nb = 0; nb < NB2; ++nb)
#pragma unroll
        for (int p = 0; p < NSPLIT; ++p) {
          const bf16x8 bw = *reinterpret_cast<const bf16x8*>(w2_lane + (p * C + 16 * nb) * W2P + 64 * ks);
#pragma unroll
          for (int rb = 0; rb < RB; ++rb) acc2[rb][nb] = mma16<NSPLIT>(a[rb], bw, acc2[rb][nb]);
        }
    }
    if (j + 1 < nchunks) {
      __syncthreads();                                                  // every wave is done with W2 chunk j and A2
      w_store(false);
      __syncthreads();                                                  // W1 chunk j + 1 is in LDS
    }
  }

  // ---------------- 3. BN2 + shortcut: x += ... (quad transpose -> 16-byte accesses) ----------------
  const int ql = l16 & 3, qd = l16 >> 2;
#pragma unroll
  for (int nb = 0; nb < NB2; ++nb) {
    const int col = oc0 + 16 * nb + 4 * qd;
    const float4 al4 = *reinterpret_cast<const float4*>(P.a2 + col);
    const float4 be4 = *reinterpret_cast<const float4*>(P.b2 + col);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const int slot = 4 * rb + ql;                                     // after the transpose this lane holds tile row 16 rb + 4 lq + ql
      const int pp = slot / T, t = slot - pp * T;
      const int64_t pos = pos_lane0 + pp;
      const bool ok = pos < P.P;
      const int64_t pc = ok ? pos : 0;
      const int64_t b = pc / HW, hw = pc - b * HW;
      float* px = P.x + (((b * T) + t) * HW + hw) * C + col;
      const float4 r = *reinterpret_cast<const float4*>(px);
      float v[4] = {acc2[rb][nb][0], acc2[rb][nb][1], acc2[rb][nb][2], acc2[rb][nb][3]};
      quad_transpose(v, ql);
      float4 o = make_float4(v[0] * P.asc2, v[1] * P.asc2, v[2] * P.asc2, v[3] * P.asc2);
      o.x = __builtin_fmaf(o.x, al4.x, be4.x); o.y = __builtin_fmaf(o.y, al4.y, be4.y);
      o.z = __builtin_fmaf(o.z, al4.z, be4.z); o.w = __builtin_fmaf(o.w, al4.w, be4.w);
      o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
      if (ok) *reinterpret_cast<float4*>(px) = o;
    }
  }
}

float inv_tau_of(const SdfNeuronCfg& n) {
  int ex;
  return (n.kind == SDF_LIF && frexpf(n.tau, &ex) == 0.5f) ? 1.0f / n.tau : 0.f;
}

template <int NSPLIT, int T, int C16, int CG, int NB1, int RG>
int launch_one(const MlpFusedParams& P, hipStream_t s) {
  using G = MlpGeo<NSPLIT, T, C16, CG, NB1, RG>;
  if (P.Ch % G::CH) return SDF_E_SHAPE;
  const int64_t items = (P.P + G::PPI - 1) / G::PPI;
  if (items >= (1LL << 31)) return SDF_E_SHAPE;
  hipLaunchKernelGGL((ms_mlp_fused_kernel<NSPLIT, T, C16, CG, NB1, RG>), dim3((unsigned)items), dim3(G::NT), 0, s, P);
  return 0;
}

template <int NSPLIT, int T>
int launch_c(const MlpFusedParams& P, int C, hipStream_t s) {
  switch (C) {
    case 96: return launch_one<NSPLIT, T, 6, 3, 2, 2>(P, s);
    case 192: return launch_one<NSPLIT, T, 12, 6, 1, 1>(P, s);
    default: return SDF_E_SHAPE;
  }
}

template <int NSPLIT>
int launch_t(const MlpFusedParams& P, int T, int C, hipStream_t s) {
  switch (T) {
    case 5: return launch_c<NSPLIT, 5>(P, C, s);
    case 10: return launch_c<NSPLIT, 10>(P, C, s);
    case 20: return launch_c<NSPLIT, 20>(P, C, s);
    default: return SDF_E_SHAPE;
  }
}

}  // namespace

// true when the one-launch form has an instantiation for this MLP (the caller falls back to the three-launch form otherwise)
bool ms_mlp_fused_supports(const SdfMsMlpDesc* d) {
  if (d->C != 96 && d->C != 192) return false;
  if (d->D != 5 && d->D != 10 && d->D != 20) return false;
  if (d->Ch % 96 || d->nsplit < 1 || d->nsplit > 3) return false;
  for (const SdfNeuronCfg* n : {&d->sn1, &d->sn2}) {
    if (n->kind != SDF_LIF && n->kind != SDF_IF && n->kind != SDF_PSN) return false;
    if (n->kind == SDF_PSN && (!n->psn_w || !n->psn_b)) return false;
    if (n->kind == SDF_LIF && !(n->tau > 1.f)) return false;
  }
  if (!sdf_aligned(d->x, 16) || !sdf_aligned(d->fc1_planes, 16) || !sdf_aligned(d->fc2_planes, 16)) return false;
  if (!d->fc1_alpha || !d->fc1_beta || !d->fc2_alpha || !d->fc2_beta) return false;
  if (!sdf_aligned(d->fc2_alpha, 16) || !sdf_aligned(d->fc2_beta, 16)) return false;
  return true;
}

int launch_ms_mlp_fused(const SdfMsMlpDesc* d, uint8_t* keep_s1, uint8_t* keep_s2, hipStream_t s) {
  MlpFusedParams P;
  P.x = d->x; P.W1 = d->fc1_planes; P.W2 = d->fc2_planes;
  P.a1 = d->fc1_alpha; P.b1 = d->fc1_beta; P.a2 = d->fc2_alpha; P.b2 = d->fc2_beta;
  P.asc1 = d->nsplit == 2 ? d->fc1_acc_scale : 1.f;
  P.asc2 = d->nsplit == 2 ? d->fc2_acc_scale : 1.f;
  P.HW = (int)d->HW; P.Ch = d->Ch; P.P = (int64_t)d->B * d->HW;
  P.sn1 = d->sn1; P.sn2 = d->sn2;
  P.inv_tau1 = inv_tau_of(d->sn1); P.inv_tau2 = inv_tau_of(d->sn2);
  P.keep_s1 = keep_s1; P.keep_s2 = keep_s2;
  int rc;
  switch (d->nsplit) {
    case 1: rc = launch_t<1>(P, d->D, d->C, s); break;
    case 2: rc = launch_t<2>(P, d->D, d->C, s); break;
    default: rc = launch_t<3>(P, d->D, d->C, s); break;
  }
  if (rc) return rc;
  hipError_t e = hipGetLastError();
  return e != hipSuccess ? (int)e : 0;
}

}  // namespace sdfmm
