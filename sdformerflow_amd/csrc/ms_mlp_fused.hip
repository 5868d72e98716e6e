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
#include "switches.h"
#include <stdlib.h>
#include <type_traits>

#ifdef SDF_STAMP
// diagnostic build only (tools/stamp_mlp.sh): cycle accounting of wave 0 of workgroup 0
__device__ unsigned long long g_mlp_stamp[16];
__device__ unsigned long long g_mlp_census[3 * 8192];     // per workgroup: start, end (100 MHz real time), hardware id
#define STAMP(var) var = __builtin_readcyclecounter()
#define STAMP_ADD(acc, a, b) acc += (b) - (a)
#else
#define STAMP(var)
#define STAMP_ADD(acc, a, b)
#endif

namespace sdfmm {
#ifndef SDF_STAMP
namespace {
#endif

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

// 8 spike bytes {0,1} -> 8 x 16-bit {0, 1.0}; the 24-bit multiply is a full-rate VALU op (operands are 17 / 14 bits)
template <int NSPLIT>
__device__ __forceinline__ bf16x8 expand24(uint2 v) {
  constexpr uint32_t ONE = NSPLIT == 2 ? 0x3C00u : 0x3F80u;
  union { bf16x8 h; uint32_t u[4]; } r;
  r.u[0] = __umul24(__builtin_amdgcn_perm(0u, v.x, 0x0c010c00u), ONE);
  r.u[1] = __umul24(__builtin_amdgcn_perm(0u, v.x, 0x0c030c02u), ONE);
  r.u[2] = __umul24(__builtin_amdgcn_perm(0u, v.y, 0x0c010c00u), ONE);
  r.u[3] = __umul24(__builtin_amdgcn_perm(0u, v.y, 0x0c030c02u), ONE);
  return r.h;
}

template <int NSPLIT>
__device__ __forceinline__ f32x4 mma16(bf16x8 a, bf16x8 b, f32x4 c) {
  if constexpr (NSPLIT == 2)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, a),
                                                   __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, b), c, 0, 0, 0);
}

template <int NSPLIT, int T, int C16, int CG, int NB1, int RG, int TEAMS>
struct MlpGeo {
  static constexpr int C = 16 * C16, NB2 = C16 / CG, CH = 16 * NB1 * CG;
  static constexpr int NW = RG * CG, NT = 64 * NW;                  // waves / threads of a TEAM (one work item)
  static constexpr int NTW = TEAMS * NT;                             // threads of the workgroup
  static constexpr int RB = 5, ROWS = 16 * RB, SLOTS = 4 * RB;
  static constexpr int PPG = SLOTS / T;            // positions per lane quarter
  static constexpr int PPW = 4 * PPG;              // positions per row group (wave)
  static constexpr int PPI = PPW * RG;             // positions per work item (workgroup)
  static constexpr int A1P = C + 16, A2P = CH + 16;                 // LDS row pitches (bytes): pitch / 4 = 4 x odd dwords
  // weight row pitches (bytes): 4 m + 2 pieces of 16 bytes - conflict-free for the (row = lane % 16, piece = lane / 16) fragment
  // reads under ds_read_b128's real, non-contiguous lane groups (round 5, brute force; the "4 x odd dwords" pitch of rounds 2 - 4,
  // 13 pieces at C = 96, made every group a 2-way conflict: 36 % of this kernel's LDS cycles, profiles/r5q_pmc_forward.txt)
  static constexpr int W1P = (C + 16) * 2, W2P = (CH + 16) * 2;
  static constexpr int W1B = NSPLIT * CH * W1P, W2B = NSPLIT * C * W2P;
  static constexpr int WB = W1B > W2B ? W1B : W2B;
  static constexpr int A1B = RG * ROWS * A1P, A2B = RG * ROWS * A2P;
  static constexpr int TEAM_LDS = A1B + A2B + WB;
  static constexpr int LDS = TEAMS * TEAM_LDS + 64;                   // + the teams' barrier counters
  static constexpr int WPIECES = NSPLIT * CH * C / 8;               // 16-byte pieces of a weight chunk (fc1 and fc2 alike)
  static constexpr int WIT = (WPIECES + NT - 1) / NT;
  static constexpr int WGS = LDS <= 80 * 1024 ? 2 : 1;              // workgroups per compute unit the LDS leaves room for
  static constexpr int WPS = (WGS * TEAMS * NW + 3) / 4;            // waves per SIMD to ask the register allocator for
  static_assert((TEAMS * NW) % 4 == 0, "a workgroup must load the four SIMDs evenly (the dispatcher admits no second workgroup beside an uneven one)");
  static_assert(C16 % CG == 0, "output columns must split evenly over the column groups");
  static_assert(SLOTS % T == 0, "T must divide the 20 accumulator slots of a lane");
  static_assert(CH % 32 == 0 && C % 32 == 0, "K steps are 32 deep");
  static_assert((A1P / 4) % 8 == 4 && (A2P / 4) % 8 == 4, "spike image pitches must be 4 x odd dwords");
  static_assert((W1P / 16) % 4 == 2 && (W2P / 16) % 4 == 2, "weight pitches must be 4 m + 2 pieces");
  static_assert(LDS <= 160 * 1024, "LDS budget");
};

// barrier of one team: every wave adds 1 to the team's LDS counter once its own LDS traffic has drained, then polls for the
// epoch's total.  (s_barrier would tie the teams of a workgroup together; they are meant to drift apart so that one team's
// neuron epilogue runs under the other's MFMAs.)
__device__ __forceinline__ void team_barrier(uint32_t* cnt, uint32_t target, int lane) {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if (lane == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  while (true) {
    const uint32_t v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    if ((int32_t)(v - target) >= 0) break;
    __builtin_amdgcn_s_sleep(1);
  }
  asm volatile("" ::: "memory");
}

template <int NSPLIT, int T, int C16, int CG, int NB1, int RG, int TEAMS, int NK, bool KEEP>
__global__ __launch_bounds__(64 * RG * CG * TEAMS, (MlpGeo<NSPLIT, T, C16, CG, NB1, RG, TEAMS>::WPS)) void ms_mlp_fused_kernel(MlpFusedParams P) {
  using G = MlpGeo<NSPLIT, T, C16, CG, NB1, RG, TEAMS>;
  constexpr int C = G::C, NB2 = G::NB2, CH = G::CH, NT = G::NT, RB = G::RB, ROWS = G::ROWS, NW = G::NW;
  constexpr int PPG = G::PPG, PPW = G::PPW, PPI = G::PPI;
  constexpr int A1P = G::A1P, A2P = G::A2P, W1P = G::W1P, W2P = G::W2P, WIT = G::WIT, WPIECES = G::WPIECES;
  __shared__ __attribute__((aligned(16))) uint8_t smem[G::LDS];
  __shared__ __attribute__((aligned(16))) float psn_s[NK == 1 ? 2 * PSN_TABLE(T) : 4];   // PSN: both neurons' coefficients (spike_mm.h psn_T_lds)
  const int wtid = threadIdx.x;
  const int wwave = __builtin_amdgcn_readfirstlane(wtid >> 6);
  const int team = wwave / NW;
  uint8_t* A1 = smem + team * G::TEAM_LDS;
  uint8_t* A2 = A1 + G::A1B;
  uint8_t* Wb = A2 + G::A2B;
  uint32_t* bar = reinterpret_cast<uint32_t*>(smem + TEAMS * G::TEAM_LDS) + team;
  if (wtid < 16) reinterpret_cast<uint32_t*>(smem + TEAMS * G::TEAM_LDS)[wtid] = 0;
  if constexpr (NK == 1) {
    psn_stage<T>(psn_s, P.sn1, wtid, 64 * RG * CG * TEAMS);
    psn_stage<T>(psn_s + PSN_TABLE(T), P.sn2, wtid, 64 * RG * CG * TEAMS);
  }
  __syncthreads();                                                      // the only workgroup-wide barrier: counters are zero
  uint32_t epoch = 0;
#define TEAM_BARRIER()                                         \
  do {                                                         \
    if constexpr (TEAMS == 1) __syncthreads();                 \
    else team_barrier(bar, (epoch += NW), lane);               \
  } while (0)

  const int tid = wtid - team * NT;                                     // thread / wave index inside the team
  const int wave = wwave - team * NW, lane = wtid & 63;
  const int rg = wave / CG, cg = wave - rg * CG;
  const int l16 = lane & 15, lq = lane >> 4;
  const int HW = P.HW, Ch = P.Ch;
  const int64_t item = (int64_t)blockIdx.x * TEAMS + team;
  if (item * PPI >= P.P) return;                                        // (an odd last workgroup: its second team has nothing to do)
  const int64_t tstride = (int64_t)HW * C;                             // elements between two time steps of a position
  const int nchunks = Ch / CH;

  // ---------------- weight chunk loader: global -> registers now, registers -> LDS between two barriers ----------------
  u32x4 wreg[WIT];
  // (the thread id is laundered per call: piece addresses are recomputed where they are used instead of being hoisted out of
  // the chunk loop and kept - or spilled - across it)
  auto w_load = [&](bool second, int j) __attribute__((always_inline)) {
    int tl = tid;
    asm volatile("" : "+v"(tl));
#pragma unroll
    for (int i = 0; i < WIT; ++i) {
      const int c = tl + NT * i;
      const int cc = (WPIECES % NT == 0 || c < WPIECES) ? c : 0;
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
    int tl = tid;
    asm volatile("" : "+v"(tl));
#pragma unroll
    for (int i = 0; i < WIT; ++i) {
      const int c = tl + NT * i;
      if (WPIECES % NT == 0 || c < WPIECES) {
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
#ifdef SDF_STAMP
  unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, a_sn1 = 0, a_fc1 = 0, a_epi1 = 0, a_h1 = 0, a_fc2 = 0, a_h2 = 0, a_fin = 0;
  const unsigned long long kstart = __builtin_readcyclecounter(), rstart = __builtin_amdgcn_s_memrealtime();
#endif
  w_load(false, 0);

  // ---------------- 1. SN1 over T of every (position, channel quad) of the item -> A1 ----------------
  {
    constexpr int C4 = C / 4, NI = PPI * C4;
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
        if constexpr (NK == 1) psn_T_lds<T>(xs, sp, psn_s);
        else neuron_T<NK, T>(xs, sp, P.sn1, P.inv_tau1);
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
        if (KEEP && ok) *reinterpret_cast<uint32_t*>(P.keep_s1 + (((b * T) + t) * HW + hw) * C + 4 * c4) = pk[t];
      }
    }
  }
  w_store(false);
  TEAM_BARRIER();                                                       // A1 and W1 chunk 0 are in LDS
  STAMP(s0); STAMP_ADD(a_sn1, kstart, s0);

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

  // A GEMM phase of this wave: KS steps (32 deep) x NBK column blocks x NSPLIT planes against its 5 row blocks.  Register diet
  // (168 = three waves per SIMD): the five expanded A fragments of the current step are live, the raw A bytes of the next step
  // refill their registers as soon as they are expanded, and weight fragments are single (column block, plane) units fetched
  // one unit ahead; fences keep the scheduler from pulling every read to the front.  Per accumulator the order is the plain
  // K loop's: step ascending, planes innermost.
  auto gemm_phase = [&](auto& acc, auto nbk_tag, const uint8_t* a_lane, int a_pitch, const uint8_t* w_lane, int w_pitch, int w_prow,
                        auto ksteps_tag) __attribute__((always_inline)) {
    constexpr int NBK = decltype(nbk_tag)::value, KS = decltype(ksteps_tag)::value, U = NBK * NSPLIT;
    uint2 ar[RB];
    bf16x8 a[RB], bw[2];
    auto load_a = [&](int ks) __attribute__((always_inline)) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) ar[rb] = *reinterpret_cast<const uint2*>(a_lane + 16 * rb * a_pitch + 32 * ks);
    };
    auto load_b = [&](int g) __attribute__((always_inline)) {
      const int ks = g / U, u = g - ks * U, nb = u / NSPLIT, p = u - nb * NSPLIT;
      bw[g & 1] = *reinterpret_cast<const bf16x8*>(w_lane + (p * w_prow + 16 * nb) * w_pitch + 64 * ks);
    };
    load_a(0);
    load_b(0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) a[rb] = expand24<NSPLIT>(ar[rb]);
      if (ks + 1 < KS) load_a(ks + 1);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int g = ks * U + u;
        if (g + 1 < KS * U) load_b(g + 1);
        if (u > 0) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[rb][u / NSPLIT] = mma16<NSPLIT>(a[rb], bw[g & 1], acc[rb][u / NSPLIT]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

#pragma unroll 1
  for (int j = 0; j < nchunks; ++j) {
    STAMP(s0);
    float al1[NB1], be1[NB1];                                           // BN1 of this lane's hidden columns: requested ahead of the MFMAs
#pragma unroll
    for (int nb = 0; nb < NB1; ++nb) {
      al1[nb] = P.a1[j * CH + hc0 + 16 * nb + l16] * P.asc1;
      be1[nb] = P.b1[j * CH + hc0 + 16 * nb + l16];
    }
    // ---- fc1: [80 rows x C] x [C x 16 NB1] ----
    f32x4 acc1[RB][NB1];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int nb = 0; nb < NB1; ++nb) acc1[rb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    gemm_phase(acc1, std::integral_constant<int, NB1>{}, a1_lane, A1P, w1_lane, W1P, CH, std::integral_constant<int, C / 32>{});
    STAMP(s1);
    w_load(true, j);                                                    // W2 chunk j: in flight during the neuron epilogue (its registers are free now)
    __builtin_amdgcn_sched_barrier(0);
    // ---- BN1 + SN2 over T in the accumulator slots -> A2 ----
#pragma unroll
    for (int nb = 0; nb < NB1; ++nb) {
      const int nloc = hc0 + 16 * nb + l16;
      const int n = j * CH + nloc;
      const float al = al1[nb], be = be1[nb];
#pragma unroll
      for (int pp = 0; pp < PPG; ++pp) {
        float xs[T], sp[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const int slot = pp * T + t;
          xs[t] = __builtin_fmaf(acc1[slot >> 2][nb][slot & 3], al, be);      // al carries the accumulator scale (a power of two: exact)
        }
        if constexpr (NK == 1) psn_T_lds<T>(xs, sp, psn_s + PSN_TABLE(T));
        else neuron_T<NK, T>(xs, sp, P.sn2, P.inv_tau2);
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const int slot = pp * T + t;
          const int row = 16 * (slot >> 2) + 4 * lq + (slot & 3);
          A2[(rg * ROWS + row) * A2P + nloc] = (uint8_t)(sp[t] != 0.f);
        }
        if (KEEP) {
          const int64_t pos = pos_lane0 + pp;
          if (pos < P.P) {
            const int64_t b = pos / HW, hw = pos - b * HW;
#pragma unroll
            for (int t = 0; t < T; ++t) P.keep_s2[(((b * T) + t) * HW + hw) * Ch + n] = (uint8_t)(sp[t] != 0.f);
          }
        }
      }
    }
    STAMP(s2);
    TEAM_BARRIER();                                                     // every wave is done with W1 chunk j; A2 is complete
    w_store(true);
    TEAM_BARRIER();                                                     // W2 chunk j is in LDS
    STAMP(s3);
    if (j + 1 < nchunks) w_load(false, j + 1);                           // W1 chunk j + 1: in flight during fc2
    // ---- fc2 partial: [80 rows x CH] x [CH x 16 NB2] ----
    gemm_phase(acc2, std::integral_constant<int, NB2>{}, a2_lane, A2P, w2_lane, W2P, C, std::integral_constant<int, CH / 32>{});
    STAMP(s4);
    if (j + 1 < nchunks) {
      TEAM_BARRIER();                                                   // every wave is done with W2 chunk j and A2
      w_store(false);
      TEAM_BARRIER();                                                   // W1 chunk j + 1 is in LDS
    }
    STAMP(s5);
    STAMP_ADD(a_fc1, s0, s1); STAMP_ADD(a_epi1, s1, s2); STAMP_ADD(a_h1, s2, s3); STAMP_ADD(a_fc2, s3, s4); STAMP_ADD(a_h2, s4, s5);
  }
  STAMP(s5);

  // ---------------- 3. BN2 + shortcut: x += ... (quad transpose -> 16-byte accesses) ----------------
  // every load of the shortcut is issued before the first store (vmcnt retires in order: a load behind a store waits for it)
  int lnl = lane;                                                       // laundered: the row addresses below are computed here, after
  asm volatile("" : "+v"(lnl));                                         // the chunk loop, not hoisted above it and carried through it
  const int ql = lnl & 3, qd = (lnl & 15) >> 2;
  const int64_t pos_e0 = item * PPI + rg * PPW + PPG * (lnl >> 4);
  float* px[RB];
  bool okr[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int slot = 4 * rb + ql;                                       // after the transpose this lane holds tile row 16 rb + 4 lq + ql
    const int pp = slot / T, t = slot - pp * T;
    const int64_t pos = pos_e0 + pp;
    okr[rb] = pos < P.P;
    const int64_t pc = okr[rb] ? pos : 0;
    const int64_t b = pc / HW, hw = pc - b * HW;
    px[rb] = P.x + (((b * T) + t) * HW + hw) * C + oc0 + 4 * qd;
  }
  float4 res[RB][NB2], al4[NB2], be4[NB2];
#pragma unroll
  for (int nb = 0; nb < NB2; ++nb) {
    al4[nb] = *reinterpret_cast<const float4*>(P.a2 + oc0 + 16 * nb + 4 * qd);
    be4[nb] = *reinterpret_cast<const float4*>(P.b2 + oc0 + 16 * nb + 4 * qd);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) res[rb][nb] = *reinterpret_cast<const float4*>(px[rb] + 16 * nb);
  }
#pragma unroll
  for (int nb = 0; nb < NB2; ++nb)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) asm volatile("" :: "v"(res[rb][nb].x), "v"(res[rb][nb].w));
#pragma unroll
  for (int nb = 0; nb < NB2; ++nb) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      float v[4] = {acc2[rb][nb][0], acc2[rb][nb][1], acc2[rb][nb][2], acc2[rb][nb][3]};
      quad_transpose(v, ql);
      float4 o = make_float4(v[0] * P.asc2, v[1] * P.asc2, v[2] * P.asc2, v[3] * P.asc2);
      o.x = __builtin_fmaf(o.x, al4[nb].x, be4[nb].x); o.y = __builtin_fmaf(o.y, al4[nb].y, be4[nb].y);
      o.z = __builtin_fmaf(o.z, al4[nb].z, be4[nb].z); o.w = __builtin_fmaf(o.w, al4[nb].w, be4[nb].w);
      const float4 r = res[rb][nb];
      o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
      if (okr[rb]) *reinterpret_cast<float4*>(px[rb] + 16 * nb) = o;
    }
  }
#undef TEAM_BARRIER
#ifdef SDF_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  STAMP(s6); STAMP_ADD(a_fin, s5, s6);
  if (wtid == 0 && blockIdx.x < 8192) {
    g_mlp_census[3 * blockIdx.x] = rstart;
    g_mlp_census[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_mlp_census[3 * blockIdx.x + 2] = ((unsigned long long)xcc << 32) | hwid;
  }
  if (blockIdx.x == gridDim.x / 2 && wtid == 0) {
    unsigned long long* o = g_mlp_stamp;
    o[0] = a_sn1; o[1] = a_fc1; o[2] = a_epi1; o[3] = a_h1; o[4] = a_fc2; o[5] = a_h2; o[6] = a_fin; o[7] = nchunks;
    o[8] = __builtin_readcyclecounter() - kstart; o[9] = __builtin_amdgcn_s_memrealtime() - rstart;
  }
#endif
}

template <int NSPLIT, int T, int C16, int CG, int NB1, int RG, int TEAMS>
int launch_one(const MlpFusedParams& P, hipStream_t s) {
  using G = MlpGeo<NSPLIT, T, C16, CG, NB1, RG, TEAMS>;
  if (P.Ch % G::CH) return SDF_E_SHAPE;
  const int64_t items = (P.P + G::PPI - 1) / G::PPI, wgs = (items + TEAMS - 1) / TEAMS;
  if (wgs >= (1LL << 31)) return SDF_E_SHAPE;
  const dim3 grid((unsigned)wgs), block(G::NTW);
  const bool keep = P.keep_s1 != nullptr;                             // the parity tape: both spike tensors also go to memory
#define SDF_MLP_LAUNCH(NK)                                                                                              \
  if (keep) SDF_LAUNCH((ms_mlp_fused_kernel<NSPLIT, T, C16, CG, NB1, RG, TEAMS, NK, true>), grid, block, 0, s, P);     \
  else SDF_LAUNCH((ms_mlp_fused_kernel<NSPLIT, T, C16, CG, NB1, RG, TEAMS, NK, false>), grid, block, 0, s, P);
  switch (neuron_class(P.sn1)) {
    case 0: SDF_MLP_LAUNCH(0) break;
    case 1:
      if constexpr (T <= 10) { SDF_MLP_LAUNCH(1) break; }               // PSN: T x T coefficients in scalar registers
      return SDF_E_SHAPE;
    default: SDF_MLP_LAUNCH(2) break;
  }
#undef SDF_MLP_LAUNCH
  return 0;
}

template <int NSPLIT, int T>
int launch_c(const MlpFusedParams& P, int C, hipStream_t s) {
  switch (C) {
    // C = 96: 2 teams x (2 row groups x 3 column groups) = 12 waves, 3 per SIMD; C = 192: 2 teams x 4 column groups = 8 waves.
    // Three bf16 planes (the exact mode) need more LDS for the weight chunk: one team (of 4 row groups at C = 96)
    case 96: return launch_one<NSPLIT, T, 6, 3, 2, NSPLIT == 3 ? 4 : 2, NSPLIT == 3 ? 1 : 2>(P, s);
    case 192: return launch_one<NSPLIT, T, 12, 4, 1, 1, 1>(P, s);
    default: return SDF_E_SHAPE;
  }
}

// two fp16 planes (the default): T in {5, 10, 20}; one / three bf16 planes (bench.py --planes 1 / 3): the shipped T = 10
int launch_t(const MlpFusedParams& P, int nsplit, int T, int C, hipStream_t s) {
  if (nsplit == 2) {
    switch (T) {
      case 5: return launch_c<2, 5>(P, C, s);
      case 10: return launch_c<2, 10>(P, C, s);
      case 20: return launch_c<2, 20>(P, C, s);
      default: return SDF_E_SHAPE;
    }
  }
  if (T != 10) return SDF_E_SHAPE;
  return nsplit == 1 ? launch_c<1, 10>(P, C, s) : launch_c<3, 10>(P, C, s);
}

#ifndef SDF_STAMP
}  // namespace
#endif

// true when the one-launch form has an instantiation for this MLP (the caller falls back to the three-launch form otherwise)
bool ms_mlp_fused_supports(const SdfMsMlpDesc* d) {
  if (d->C != 96 && d->C != 192) return false;
  if (d->D != 5 && d->D != 10 && d->D != 20) return false;
  if (d->Ch % 192 || d->nsplit < 1 || d->nsplit > 3) return false;
  if (d->nsplit != 2 && d->D != 10) return false;
  if (neuron_class(d->sn1) != neuron_class(d->sn2)) return false;
  if (neuron_class(d->sn1) == 1 && d->D > 10) return false;           // PSN over T = 20: 400 coefficients do not fit the scalar registers
  for (const SdfNeuronCfg* n : {&d->sn1, &d->sn2}) {
    if (n->kind != SDF_LIF && n->kind != SDF_IF && n->kind != SDF_PSN) return false;
    if (n->kind == SDF_PSN && (!n->psn_w || !n->psn_b)) return false;
    if (!sdf_tau_ok(n->kind, n->tau)) return false;
  }
  // C = 192 (stage 1): a work item streams 1.2 MB of weights for 8 positions; with few positions (batch 1 at 288 x 384: 1 728) the
  // three-launch form is as fast (measured 65 us both), from a few thousand on the one-launch form wins (config 5: -30 %)
  if (d->C == 192 && (int64_t)d->B * d->HW < 4096 && !sdf_sw(SW_MLP_FUSED_ANY)) return false;
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
  const int rc = launch_t(P, d->nsplit, d->D, d->C, s);
  if (rc) return rc;
  hipError_t e = hipGetLastError();
  return e != hipSuccess ? (int)e : 0;
}

}  // namespace sdfmm

#ifdef SDF_STAMP
extern "C" int sdf_debug_read_stamps_mlp(unsigned long long* host16) {
  return (int)hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_mlp_stamp), 16 * sizeof(unsigned long long));
}
extern "C" int sdf_debug_occupancy_mlp(int* blocks_per_cu) {
  return (int)hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, sdfmm::ms_mlp_fused_kernel<2, 10, 6, 3, 2, 2, 2, 0, false>, 768, 0);
}
extern "C" int sdf_debug_read_census_mlp(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_mlp_census), 3 * 8192 * sizeof(unsigned long long));
}
#endif
