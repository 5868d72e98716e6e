// Narrow stages (C <= 192: swin stages 0 - 1 of the shipped model and the first patch merging) of the MS swin block for gfx950 - rows
// a5 / a6 / a7 / a8 of SURVEY.md section 8 at the shapes where the problem is LARGE IN ROWS against SMALL weights (batch 1: 69 120 /
// 17 280 token rows against 9 - 150 K weights per layer).  VERDICT r4 #1: stage 0 ran on three generations of kernels (qk_front +
// spike_gemm + ms_mlp_fused: 92 us per block, 36 % LDS bank-conflict cycles, 11 - 63 vector instructions per MFMA), stage 1 on the
// wide-stage kernels of ms_wide.hip, whose K ring + one barrier per 128-deep chunk + one wave per SIMD is built for K >= 384 and
// spends a launch of K = 192 on pipeline fill (103 us per block: the same kernels take 67 us at stage 2 for 1.5 x the flop).
//
// Here the WHOLE K of a column group's three int8 digit planes is LDS-resident (K <= 768: 12 - 74 KB per 32 columns) for the life
// of a workgroup, and the workgroup's eight waves - TWO PER SIMD - walk 80-row units of its row range independently: after the
// weights are in there is no barrier, no ring, no shared operand.  A wave's unit is the position-major tile of ms_wide.hip (80 rows
// = (20 / T) x 4 positions x all T steps: whatever neuron follows runs over T in the accumulator registers) x 32 columns:
//   * spikes are the MFMA's ROW operand, global -> registers in the MFMA's own layout, two 64-deep steps in flight (a ring of two
//     register sets, each refilled behind the MFMAs that consumed it);
//   * weight fragments come from the resident image [plane][k-piece][column ^ (k-piece & 7)] x 16 B (the conflict-free layout of
//     the wide-stage ring, whole K deep), two fragments ahead of their MFMAs;
//   * accumulators pinned to AGPRs (inline-asm MFMA, wide_common.h); 120 of them + <= 128 vector registers = two waves per SIMD, so one
//     wave's epilogue (BN, shortcut, LIF / PSN over T, byte transposes, stores) and operand latency run under the other's MFMAs;
//   * K need not be a multiple of 64 (C = 96): the resident image is zero-padded to whole steps and the lanes of the missing
//     k-pieces load nothing (out-of-range buffer offsets read zeros).
// Epilogues, operand address modes (rows / tiled hand-over / head scramble / 2x2 merge quadrants) and the tiled hand-over layout are
// those of wide_pm_kernel (ms_wide.hip): the two kernel families interoperate launch by launch and give bit-equal results (exact
// integer sums, the same fp32 epilogue expressions).
//
//   res_pm_kernel    : proj (x += BN(Z Wp^T + b) through the head scramble, + the MLP's SN1), fc1 (+ BN1 + SN2), fc2 (+ BN2 +
//                      shortcut [+ the next layer's first neuron]), patch merging    reference Spiking_swin_transformer3D.py:164-181,
//                      :709-714, :810-820, :840-845, :952-974
//   res_front_kernel : q | k = SN_q/k( BN( xs [Wq;Wk]^T ) [+ PE] ), E = k AND SN2_q( head sums of q )      reference :671-694
#include "wide_common.h"
#include "switches.h"
#include <stdlib.h>
#include <type_traits>

#ifdef SDF_STAMP
// diagnostic build only (tools/stamp_res.sh): cycle stamps of wave 0 of the middle workgroup per kernel kind (0 = front, 1 / 2 / 3 = the
// position-major product's epilogue kinds) - entry, first unit's addresses, weights in + barrier, first unit's main loop, first unit's
// epilogue, kernel end - and every workgroup's life in 100 MHz real time
__device__ unsigned long long g_res_stamp[4 * 8];
__device__ unsigned long long g_res_census[4 * 2 * 1024];
#define RSTAMP(i) do { if (rs_n[i] == 0) { rs[i] = __builtin_readcyclecounter(); rs_n[i] = 1; } } while (0)
#define RSTAMP_DECL unsigned long long rs[6] = {0, 0, 0, 0, 0, 0}; int rs_n[6] = {0, 0, 0, 0, 0, 0}; const unsigned long long rr0 = __builtin_amdgcn_s_memrealtime()
#define RSTAMP_OUT(kind, nunits)                                                                                  \
  do {                                                                                                            \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                              \
    RSTAMP(5);                                                                                                    \
    if (threadIdx.x == 0 && blockIdx.x < 1024) {                                                                  \
      g_res_census[(kind) * 2048 + 2 * blockIdx.x] = rr0;                                                         \
      g_res_census[(kind) * 2048 + 2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();                        \
    }                                                                                                             \
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) {                                                        \
      unsigned long long* o = g_res_stamp + 8 * (kind);                                                           \
      for (int i = 0; i < 5; ++i) o[i] = rs[i + 1] > rs[i] ? rs[i + 1] - rs[i] : 0;                               \
      o[5] = __builtin_amdgcn_s_memrealtime() - rr0; o[6] = gridDim.x; o[7] = (nunits);                           \
    }                                                                                                             \
  } while (0)
#else
#define RSTAMP(i)
#define RSTAMP_DECL
#define RSTAMP_OUT(kind, nunits)
#endif

namespace sdfmm {
namespace {

constexpr int NWV = 8;                                 // waves per workgroup: two per SIMD

// resident weight image: [plane][k-piece 0 .. KP-1][column ^ (k-piece & 7)] x 16 B, BN columns, KP = k-pieces padded to whole steps
__host__ __device__ constexpr int res_kp(int K) { return ((K >> 4) + 3) & ~3; }
__host__ __device__ constexpr int res_wbytes(int K, int BN) { return 3 * res_kp(K) * BN * 16; }

// whole-K digit planes of columns [n0, n0 + BN) -> LDS.  row_of(p, col) = byte offset of (plane p, column col, k = 0) or INV.
// Eight neighbouring threads read the eight 16-byte pieces of 128 contiguous bytes of a weight row.
template <int BN, class RowOf>
__device__ __forceinline__ void res_load_weights(uint8_t* Wl, const __amdgpu_buffer_rsrc_t W_rs, int K, int tid, int nthreads, int col0, int ncols,
                                                 RowOf row_of) {
  const int KP = res_kp(K), kpv = K >> 4, total = 3 * ncols * KP;
  for (int i0 = 0; i0 < total; i0 += 4 * nthreads) {
    u32x4 v[4];
    uint32_t dst[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = i0 + j * nthreads + tid;
      const int kp = i % KP, r = i / KP, col = col0 + r % ncols, p = r / ncols;
      uint32_t g = INV;
      if (i < total && kp < kpv) g = row_of(p, col);
      v[j] = __builtin_amdgcn_raw_buffer_load_b128(W_rs, g == INV ? INV : g + 16u * kp, 0, 0);      // (missing pieces / columns: zeros)
      dst[j] = i < total ? (uint32_t)(((p * KP + kp) * BN + (col ^ (kp & 7))) * 16) : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (dst[j] != 0xFFFFFFFFu) *reinterpret_cast<u32x4*>(Wl + dst[j]) = v[j];
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The main loop of one unit.  acc[digit][rb][cb] += A[this wave's 16 RB rows][K] x D_digit[16 CB columns][K]^T, K in 64-deep steps.
//   aa.get(s, rb, voff, soff) : this lane's 16-byte piece (row lane % 16 of row block rb, k-piece lane / 16) of step s, INV for a
//                               row / piece that does not exist
// Steps alternate between two register sets; a set is refilled (step s + 2) behind the last MFMAs that read it.  One step = 3 CB
// units of RB MFMAs on one weight fragment; the fragments of units g + 1, g + 2 are in flight while unit g multiplies.
// strip (STRIP builds, else null): ROW-MAJOR operands are requested as FULL LINES - a lane quad reads the 64 contiguous bytes of one row
// (lane = 4 row + piece; the caller's addresses are formed that way) - and pass through a per-wave LDS strip of RB x 1 KB into the MFMA's
// order (lane = row + 16 k-group) at the top of their step: written lane-linear, read back through the XOR swizzle of ms_smallm.hip (the
// writer stores piece p of row r at slot 4 r + (p ^ 2 (r / 8))).  A fragment-shaped load of a row-major tensor has every lane of the wave
// in a different cache line: 64 lines per instruction instead of 16 in the texture path - measured on the third decoder level's product
// (17 280 x 864 x 416, tools/res_ablate.sh, profiles/r5m_res_ablation.txt): 14 of the launch's 40 us.
template <int RB, int CB, class AAddr>
__device__ __forceinline__ void res_mainloop(i32x4 (&acc)[3][RB][CB], const __amdgpu_buffer_rsrc_t A_rs, AAddr& aa, int ksteps, const uint8_t* Wl,
                                              int KP, int lane, uint8_t* strip = nullptr) {
  constexpr int BN = 16 * CB, NU = 3 * CB;
  asm volatile("" : "+v"(lane));                         // (laundered per unit: the fragment offsets are formed here, not held across the epilogue)
  const int l16 = lane & 15, lj = lane >> 4;
  const uint32_t plane = (uint32_t)KP * BN * 16;
  uint32_t woff[2][CB];                                  // this lane's fragment of (step parity, column block) inside a plane, step 0 / 1
#pragma unroll
  for (int par = 0; par < 2; ++par)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) woff[par][cb] = (uint32_t)(((4 * par + lj) * BN + ((cb * 16 + l16) ^ (4 * par + lj))) * 16);
  i32x4 aX[RB], aY[RB];
  auto a_load1 = [&](i32x4& dst, int s, int rb) __attribute__((always_inline)) {
    uint32_t voff, soff;
    aa.get(s, rb, voff, soff);
#ifdef RES_X_NOA
    voff = INV;                                          // (diagnostic builds, tools/res_ablate.sh: the spike operand reads nothing)
#endif
    dst = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(A_rs, voff, soff, 0));      // (a step beyond K: every piece is INV)
  };
  auto step = [&](int s, auto par_c, i32x4 (&aC)[RB], auto refill_c, auto zero_c) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_c)::value;
    constexpr bool refill = decltype(refill_c)::value, zero = decltype(zero_c)::value;
    const uint8_t* wb = Wl + (uint32_t)(s >> 1) * (8u * BN * 16u);           // (steps 2 i and 2 i + 1 share the base: woff carries the parity)
    if (strip) {                                                                // (wave-uniform; same-wave LDS operations execute in order)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) *reinterpret_cast<i32x4*>(strip + rb * 1024 + 16 * lane) = aC[rb];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) aC[rb] = *reinterpret_cast<const i32x4*>(strip + rb * 1024 + (4 * l16 + (lj ^ ((l16 >> 2) & 2))) * 16);
    }
    i32x4 b[3];
    auto load_b = [&](int g) __attribute__((always_inline)) {
      const int cb = g / 3, dg = g - 3 * cb;
      b[g % 3] = *reinterpret_cast<const i32x4*>(wb + dg * plane + woff[PAR][cb]);
    };
    load_b(0);
    load_b(1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < NU; ++g) {
      if (g + 2 < NU) load_b(g + 2);
      const int cb = g / 3, dg = g - 3 * cb;
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
#ifdef RES_X_NOMFMA
        if constexpr (zero) acc[dg][rb][cb] = aC[rb] ^ b[g % 3];                    // (diagnostic builds: no matrix instruction)
        else asm volatile("" : "+v"(acc[dg][rb][cb]) : "v"(aC[rb]), "v"(b[g % 3]));
#else
        if constexpr (zero) mfma_i8_v_zero(acc[dg][rb][cb], aC[rb], b[g % 3]);      // (the tile's first step: no accumulator input)
        else mfma_i8_v(acc[dg][rb][cb], aC[rb], b[g % 3]);
#endif
        if (g == NU - 1 && refill) a_load1(aC[rb], s + 2, rb);                // (this set's last reader of row block rb: step s + 2 moves in)
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  using T0 = std::integral_constant<int, 0>;
  using T1 = std::integral_constant<int, 1>;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) a_load1(aX[rb], 0, rb);
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) a_load1(aY[rb], 1, rb);
  step(0, T0{}, aX, std::true_type{}, std::true_type{});  // step 0 starts the sums (the accumulators are never zeroed)
  int s = 1;
#pragma unroll 1
  for (; s + 1 < ksteps; s += 2) {
    step(s, T1{}, aY, std::true_type{}, std::false_type{});
    step(s + 1, T0{}, aX, std::true_type{}, std::false_type{});
    mfma_drain_v(acc);                                     // (the exit edge of the round may shuffle accumulators: wide_common.h)
  }
  if (s < ksteps) step(s, T1{}, aY, std::false_type{}, std::false_type{});
  mfma_drain_v(acc);                                       // (the accumulators are read by vector instructions from here on)
}

// the exact integer sum of the three digit sums as one fp32 number.  K <= 256: |d2| 65536 <= 256 x 127 x 65536 < 2^31, the whole sum
// is one int32 and v_cvt_f32_i32 rounds it once - the same value as fma(d2, 65536, d1 256 + d0) on exact operands, one instruction less
template <bool SMALLK>
__device__ __forceinline__ float res_digits_f32(int a0, int a1, int a2) {
  if constexpr (SMALLK) {
    const uint32_t lo = ((uint32_t)a1 << 8) + (uint32_t)a0;                  // (v_lshl_add_u32 twice)
    return (float)(int)(((uint32_t)a2 << 16) + lo);
  } else {
    return digits_f32(a0, a1, a2);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Position-major product, weight-resident.  EPI / AM as wide_pm_kernel: EPI 1 = neuron (spikes out), 2 = fp32 (+ shortcut), 3 = fp32 and
// the neuron on the updated stream; AM 0 = rows of a tensor (row-major / tiled / head scramble), 2 = the 2x2 concatenation of patch
// merging (any C % 16 == 0: the quadrant of every 16-byte k-piece is decoded per lane).  SK: K <= 256 (res_digits_f32).
template <int T, int EPI, int NK, int AM, bool SK, bool STRIP = false>
__global__ __launch_bounds__(64 * NWV) void res_pm_kernel(WidePmParams P) {
  constexpr int RB = RBW, CB = 2, ROWS = 16 * RB, SLOTS = 4 * RB, PPG = SLOTS / T, PPW = 4 * PPG, BN = 16 * CB;
  constexpr int SP = s_pitch(BN), STILE = ROWS * SP;
  static_assert(SLOTS % T == 0, "T must divide the 20 accumulator slots of a lane");
  static_assert(EPI >= 1 && EPI <= 3 && (AM == 0 || AM == 2 || AM == 3), "epilogue 1 neuron / 2 fp32 / 3 both; operand rows, merge quadrants or the transposed convolution's 2 x 2 neighbourhood");
  static_assert(AM != 3 || EPI == 2, "the transposed convolution has the fp32 epilogue only");
  extern __shared__ __attribute__((aligned(16))) uint8_t dyn[];
  const int K = P.K, N = P.N, HW = P.HW, KP = res_kp(K);
  uint8_t* Wl = dyn;                                     // resident weights
  uint8_t* Sall = dyn + res_wbytes(K, BN);               // per-wave byte tiles [80][SP]
  int32_t* rowtab_all = reinterpret_cast<int32_t*>(Sall + ((EPI & 1) ? NWV * STILE : 0));
  f32x4* coltab = reinterpret_cast<f32x4*>(rowtab_all + NWV * ROWS);           // per column: {alpha x digit scale, beta, digit scale, bias}
  float* psn_tbl = reinterpret_cast<float*>(coltab + BN);
  uint8_t* strips = reinterpret_cast<uint8_t*>(psn_tbl + (NK == 1 ? PSN_TABLE(20) : 0));      // STRIP: RB x 1 KB per wave
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, lq = lane >> 4;
  RSTAMP_DECL;
  RSTAMP(0);
  // (row range, column group), column group fastest: the workgroups of an XCD (ids equal mod 8) cover a contiguous range of rows
  int item = blockIdx.x;
  const int G = gridDim.x;
  if ((G & 7) == 0) item = (item & 7) * (G >> 3) + (item >> 3);
  if (item >= P.ncg * P.nrg) return;
  const int rr = item / P.ncg, cg = item - rr * P.ncg;
  const int n0 = cg * BN;
  const int u_lo = rr * P.passes, u_hi = min(P.nunits, u_lo + P.passes);       // this workgroup's units (passes = units per row range)
  const __amdgpu_buffer_rsrc_t A_rs = make_rsrc(P.A), x_rs = make_rsrc(P.x), o_rs = make_rsrc(P.out_spike);
  const __amdgpu_buffer_rsrc_t W_rs = make_rsrc(P.W);
  int32_t* rowtab = rowtab_all + wave * ROWS;
  const int kpv = K >> 4, ksteps = KP >> 2;
  const int c = lane & 15, q = lane >> 4;

  // row addressing of a unit: activation row (or -1) of every tile row, then this lane's operand base per row block
  uint32_t a_base[RB], a_mask[RB];
  const uint32_t a_step = __builtin_amdgcn_readfirstlane(P.a_tiled ? 4u * ROWS * 16u : (P.zsrc ? 2u * P.zg_G : 64u));
  auto prepare = [&](int unit) __attribute__((always_inline)) {
    for (int r = lane; r < ROWS; r += 64) {
      const int rb = r >> 4, i = r & 15, qq = i >> 2, slot = 4 * rb + (i & 3);
      const int pp = slot / T, t = slot - pp * T;
      const uint32_t pos = (uint32_t)unit * PPW + qq * PPG + pp;            // (positions and rows fit 31 bits: the host checks)
      int32_t g = -1;
      if (pos < (uint32_t)P.P) {
        const uint32_t b = pos / (uint32_t)HW, hw = pos - b * (uint32_t)HW;
        g = (int32_t)((b * T + t) * (uint32_t)HW + hw);
      }
      rowtab[r] = g;
    }
    asm volatile("" ::: "memory");                       // (same-wave LDS operations execute in order: no wait between the table's writes and reads)
    // STRIP: this lane requests piece lp of tile row 16 rb + lane / 4 (a lane quad = one row's 64 contiguous bytes), else piece lq of row lane % 16
    const int lrow = STRIP ? (lane >> 2) : l16;
    const int lpc = STRIP ? ((lane & 3) ^ ((lane >> 4) & 2)) : lq;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const int32_t g = rowtab[16 * rb + lrow];
      a_base[rb] = INV;
      a_mask[rb] = 0;
      if (AM == 3) {
        if (g >= 0) {                                    // input pixel (img, a, b): its 2 x 2 neighbourhood (a + dh, b + dw), zero beyond the image
          const uint32_t img = (uint32_t)g / (uint32_t)HW, pix = (uint32_t)g - img * (uint32_t)HW;
          const uint32_t a = pix / (uint32_t)P.cv_W, b = pix - a * (uint32_t)P.cv_W;
          a_base[rb] = (uint32_t)g * (uint32_t)P.cv_Cin;
          const uint32_t hin = a + 1 < (uint32_t)P.cv_H ? 0xFu : 0x5u, win = b + 1 < (uint32_t)P.cv_W ? 0xFu : 0x3u;
          a_mask[rb] = hin & win;
        }
      } else if (AM == 2) {
        if (g >= 0) {                                    // merged position (b, t, h2, w2) -> source pixel (2 h2, 2 w2) of the same (b, t)
          const uint32_t W2 = ((uint32_t)P.cv_W + 1u) >> 1, img = (uint32_t)g / (uint32_t)HW, pix = (uint32_t)g - img * (uint32_t)HW;
          const uint32_t h2 = pix / W2, w2 = pix - h2 * W2;
          a_base[rb] = ((img * (uint32_t)P.cv_H + 2 * h2) * (uint32_t)P.cv_W + 2 * w2) * (uint32_t)P.cv_Cin;
          const uint32_t hin = 2 * h2 + 1 < (uint32_t)P.cv_H ? 0xFu : 0x5u, win = 2 * w2 + 1 < (uint32_t)P.cv_W ? 0xFu : 0x3u;
          a_mask[rb] = hin & win;                        // odd sizes: the reference pads zeros in front of the neuron, SN(0) = 0
        }
      } else if (P.a_tiled) {
        a_base[rb] = (((uint32_t)unit * (uint32_t)kpv + (uint32_t)lq) * ROWS + 16 * rb + l16) * 16u;
      } else if (g >= 0) {
        a_base[rb] = P.zsrc ? (uint32_t)P.zsrc[g] + (uint32_t)(lq >> 1) * P.zg_G + 16u * (lq & 1) : (uint32_t)g * (uint32_t)K + 16u * lpc;
      }
    }
    if (AM == 3) {
      // the epilogue addresses the OUTPUT: the table now takes the pixel index of (2a, 2b) in the (imgs, 2H, 2W) grid (same-wave LDS
      // operations execute in order: every lane's reads above precede these writes)
      for (int r = lane; r < ROWS; r += 64) {
        const int32_t g = rowtab[r];
        if (g >= 0) {
          const uint32_t img = (uint32_t)g / (uint32_t)HW, pix = (uint32_t)g - img * (uint32_t)HW;
          const uint32_t a = pix / (uint32_t)P.cv_W, b = pix - a * (uint32_t)P.cv_W;
          rowtab[r] = (int32_t)(((img * 2u * (uint32_t)P.cv_H + 2u * a) * 2u * (uint32_t)P.cv_W) + 2u * b);
        }
      }
      asm volatile("" ::: "memory");
    }
  };
  // this lane's piece of (step, row block).  Plain rows: piece 4 s + lq must exist (K % 64 != 0: the last step's upper pieces do not).
  struct AddrPlain {
    const uint32_t* base; uint32_t step; int lim;        // lim = k-pieces - lq: piece 4 s + lq exists iff 4 s < lim
    __device__ __forceinline__ void get(int s, int rb, uint32_t& voff, uint32_t& soff) const {
      voff = 4 * s < lim ? base[rb] : INV;
      soff = (uint32_t)s * step;
    }
  };
  // Patch merging: K = 4 C in quadrant order (dh, dw) = (q % 2, q / 2) (reference Spiking_swin_transformer3D.py:965-970); the quadrant
  // and the channel offset of piece 4 s + lq by one multiply (cp_inv = ceil(2^16 / (C / 16)): exact for the <= 192 pieces the host admits)
  struct AddrMerge {
    const uint32_t* base; const uint32_t* mask; int lq, cp16, cp_inv, W, Cin, kpv;
    __device__ __forceinline__ void get(int s, int rb, uint32_t& voff, uint32_t& soff) const {
      const int p = 4 * s + lq, qd = (int)(((uint32_t)p * (uint32_t)cp_inv) >> 16), cpi = p - qd * cp16;
      const uint32_t toff = (uint32_t)(((qd & 1) * W + (qd >> 1)) * Cin + 16 * cpi);
      voff = (p < kpv && ((mask[rb] >> qd) & 1u)) ? base[rb] + toff : INV;
      soff = 0;
    }
  };


  // ---- first unit's row addressing, then the weights (their latency covers it), one barrier, then the waves part ways ----
  int unit = u_lo + wave;
  if (unit < u_hi) prepare(unit);
  RSTAMP(1);
  res_load_weights<BN>(Wl, W_rs, K, tid, 64 * NWV, 0, BN, [&](int p, int col) -> uint32_t {
    return n0 + col < N ? (uint32_t)((p * N + n0 + col) * K) : INV;
  });
  if constexpr (NK == 1) psn_stage<T>(psn_tbl, P.sn, tid, 64 * NWV);
  // BN / bias / digit scale of the workgroup's columns: a small LDS table, read per column block in the epilogue (held in registers
  // they cost ten across the main loop).  With a bias: h = fma(sum x scale + bias, alpha, beta); without, alpha x scale (exact: the
  // scale is a power of two) is folded on the spot: fma(sum, alpha x scale, beta) is the same rounding of the same real number.
  const bool has_bias = P.bias != nullptr;               // (wave-uniform)
  if (tid < BN) {
    const int nc = n0 + tid < N ? n0 + tid : 0;
    const float al = P.alpha ? P.alpha[nc] : 1.f, be = P.alpha ? P.beta[nc] : 0.f, csn = P.cscale[nc];
    coltab[tid] = f32x4{has_bias ? al : al * csn, be, csn, has_bias ? P.bias[nc] : 0.f};
  }
  __syncthreads();
  RSTAMP(2);

#pragma unroll 1
  for (; unit < u_hi; unit += NWV) {
    if (unit != u_lo + wave) prepare(unit);
    i32x4 acc[3][RB][CB];                                // (written by the main loop's first step)
    const int lpiece = STRIP ? ((lane & 3) ^ ((lane >> 4) & 2)) : lq;        // (the k-piece of a step this lane requests)
    uint8_t* strip = STRIP ? strips + wave * (RB * 1024) : nullptr;
    if constexpr (AM >= 2) {
      AddrMerge aa{a_base, a_mask, lpiece, P.cv_Cin >> 4, P.cv_cpt, P.cv_W, P.cv_Cin, kpv};
      res_mainloop<RB, CB>(acc, A_rs, aa, ksteps, Wl, KP, lane, strip);
    } else {
      AddrPlain aa{a_base, a_step, kpv - lpiece};
      res_mainloop<RB, CB>(acc, A_rs, aa, ksteps, Wl, KP, lane, strip);
    }

    RSTAMP(3);
    // ---------------- epilogue ----------------
    // Register budget: 128 vector registers beside the 120 accumulators (two waves per SIMD).  The accumulators are read quad by quad
    // where they are used (acc_read: left to the allocator all 120 were copied out behind the main loop), one column block at a time;
    // the shortcut values of both blocks are requested first (one latency), the store offsets are formed again per block from the row
    // table instead of being held.
    // (the PSN form with both outputs - T coefficients + T values beside everything else - requests them per block instead)
    constexpr bool RES_EARLY = !(NK == 1 && EPI == 3);
    float res[(EPI & 2) ? CB : 1][(EPI & 2) ? SLOTS : 1];
    auto request_res = [&](int cb0, int cb1) __attribute__((always_inline)) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        const i32x4 g4 = *reinterpret_cast<const i32x4*>(rowtab + 16 * rb + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t xo = (g4[e] >= 0 && n0 + c < N && !P.no_resid) ? ((uint32_t)g4[e] * (uint32_t)P.ldo + (uint32_t)(n0 + c)) * 4u : INV;
#pragma unroll
          for (int cb = 0; cb < CB; ++cb)
            if (cb >= cb0 && cb < cb1)
              res[cb][4 * rb + e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rs, n0 + 16 * cb + c < N ? xo : INV, 64u * cb, 0));
        }
      }
      asm volatile("" ::: "memory");                     // (the row table is read again below: nothing of it stays in registers)
    };
    if constexpr ((EPI & 2) != 0 && RES_EARLY) request_res(0, CB);
    uint8_t* S = Sall + wave * STILE;                    // per-wave byte tile [80][SP] (EPI & 1)
    uint32_t sel1 = 0, sel2 = 0;
    int lne = lane;
    asm volatile("" : "+v"(lne));                        // (laundered: the epilogue's lane-derived values live here only)
    if constexpr ((EPI & 1) != 0) quad_sel(lne, sel1, sel2);
    const int m4 = ((lne & 15) >> 2), ci = lne & 3;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      if constexpr ((EPI & 2) != 0 && !RES_EARLY) request_res(cb, cb + 1);
      const f32x4 cp = coltab[16 * cb + c];              // {alpha (x scale), beta, scale, bias} of this lane's column
      float h[SLOTS];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        const i32x4 a0 = acc[0][rb][cb], a1 = acc[1][rb][cb], a2 = acc[2][rb][cb];      // (vector-register accumulators: read in place)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // no bias: alf = alpha x (power-of-two digit scale) is exact, so fma(sum, alf, beta) IS fma(sum x scale + 0, alpha, beta)
          const float sum = res_digits_f32<SK>(a0[e], a1[e], a2[e]);
          h[4 * rb + e] = has_bias ? __builtin_fmaf(sum * cp[2] + cp[3], cp[0], cp[1]) : __builtin_fmaf(sum, cp[0], cp[1]);
        }
        if constexpr ((EPI & 2) != 0) {
          const i32x4 g4 = *reinterpret_cast<const i32x4*>(rowtab + 16 * rb + 4 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = h[4 * rb + e] + res[cb][4 * rb + e];
            h[4 * rb + e] = v;
            uint32_t xo = (g4[e] >= 0 && n0 + 16 * cb + c < N) ? ((uint32_t)g4[e] * (uint32_t)P.ldo + (uint32_t)(n0 + c)) * 4u : INV;
            if constexpr (AM == 3) {                     // column -> (output pixel of the row's 2 x 2 block, channel)
              const uint32_t n = (uint32_t)(n0 + 16 * cb + c), cls = n / (uint32_t)P.dc_cout, co = n - cls * (uint32_t)P.dc_cout;
              xo = (g4[e] >= 0 && n < (uint32_t)N)
                       ? (((uint32_t)g4[e] + (cls >> 1) * 2u * (uint32_t)P.cv_W + (cls & 1u)) * (uint32_t)P.dc_cout + co) * 4u - 64u * cb : INV;
            }
#ifdef RES_X_NOST
            asm volatile("" :: "v"(v), "v"(xo));         // (diagnostic builds: the fp32 stores are dropped)
#else
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), x_rs, xo, 64u * cb, 0);
#endif
          }
        }
      }
      if constexpr ((EPI & 1) != 0) {
        uint32_t bits = 0;
#pragma unroll
        for (int pp = 0; pp < PPG; ++pp) {
          float xs[T];
#pragma unroll
          for (int t = 0; t < T; ++t) xs[t] = h[pp * T + t];
          bits |= neuron_any_bits<NK, T, true>(xs, P.sn, P.inv_tau, psn_tbl) << (pp * T);      // (LEAN: the 128-register budget)
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          const uint32_t w = quad_tr_bytes(spread4(bits >> (4 * rb)), sel1, sel2);
          *reinterpret_cast<uint32_t*>(S + (16 * rb + 4 * q + ci) * SP + 16 * cb + 4 * m4) = w;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr ((EPI & 1) != 0) {
#pragma unroll
      for (int it = 0; it < (ROWS * CB + 63) / 64; ++it) {
        const int pc = lane + 64 * it;
        if (pc < ROWS * CB) {
          // tiled: piece-major (80 consecutive rows of a channel piece are 1 280 contiguous bytes); row-major: row-major pieces
          const int r = P.out_tiled ? pc % ROWS : pc / CB, k16 = P.out_tiled ? pc / ROWS : pc % CB;
          const int32_t g = rowtab[r];
          const u32x4 v = *reinterpret_cast<const u32x4*>(S + r * SP + 16 * k16);
          uint32_t off = INV;
          if (n0 + 16 * k16 < N) {
            if (P.out_tiled) off = (((uint32_t)unit * (uint32_t)(P.ldsp >> 4) + (uint32_t)((n0 >> 4) + k16)) * ROWS + r) * 16u;
            else if (g >= 0) off = (uint32_t)g * (uint32_t)P.ldsp + (uint32_t)(n0 + 16 * k16);
          }
          __builtin_amdgcn_raw_buffer_store_b128(v, o_rs, off, 0, 0);
        }
      }
    }
    RSTAMP(4);
  }
  RSTAMP_OUT(EPI, (u_hi - u_lo + NWV - 1) / NWV);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Attention front, weight-resident: q | k of one head (64 columns) per workgroup column, a wave owns 16 tokens x the T' = 2 steps.
template <int NK, bool KEEP>
__global__ __launch_bounds__(64 * NWV) void res_front_kernel(WideFrontParams P) {
  constexpr int RB = 2, CB = 4, BN = 64, ROWS = 16 * RB, SB = KEEP ? 96 : 32, SP = s_pitch(SB), STILE = ROWS * SP;
  extern __shared__ __attribute__((aligned(16))) uint8_t dyn[];
  const int C = P.C, KP = res_kp(C);
  uint8_t* Wl = dyn;
  uint8_t* Sall = dyn + res_wbytes(C, BN);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, lq = lane >> 4;
  int item = blockIdx.x;
  const int G = gridDim.x;
  if ((G & 7) == 0) item = (item & 7) * (G >> 3) + (item >> 3);
  if (item >= P.nH * P.nrg) return;
  const int rg = item / P.nH, hd = item - rg * P.nH;     // head fastest: an XCD owns a contiguous range of token tiles
  const int t_lo = rg * P.ntiles_per, t_hi = min(P.ntiles, t_lo + P.ntiles_per);
  const __amdgpu_buffer_rsrc_t A_rs = make_rsrc(P.xs);
  const int kpv = C >> 4, ksteps = KP >> 2;
  // columns 0..31 = the head's q rows, 32..63 = its k rows
  res_load_weights<BN>(Wl, make_rsrc(P.wq), C, tid, 64 * NWV, 0, 32, [&](int p, int col) -> uint32_t {
    return (uint32_t)(p * P.wq_plane + (int64_t)(hd * 32 + col) * C);
  });
  res_load_weights<BN>(Wl, make_rsrc(P.wk), C, tid, 64 * NWV, 32, 32, [&](int p, int col) -> uint32_t {
    return (uint32_t)(p * P.wk_plane + (int64_t)(hd * 32 + (col - 32)) * C);
  });
  const int c = lane & 15, q = lane >> 4;
  float qa[2], qb[2], ka[2], kb[2], qc[2], kc[2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const int ch = hd * 32 + 16 * cb + c;
    qa[cb] = P.q_al ? P.q_al[ch] : 1.f; qb[cb] = P.q_al ? P.q_be[ch] : 0.f;
    ka[cb] = P.k_al ? P.k_al[ch] : 1.f; kb[cb] = P.k_al ? P.k_be[ch] : 0.f;
    qc[cb] = P.q_cs[ch]; kc[cb] = P.k_cs[ch];
  }
  __syncthreads();

  struct AddrPlain {
    const uint32_t* base; int lim;
    __device__ __forceinline__ void get(int s, int rb, uint32_t& voff, uint32_t& soff) const {
      voff = 4 * s < lim ? base[rb] : INV;
      soff = (uint32_t)s * 64u;
    }
  };
#pragma unroll 1
  for (int tile = t_lo + wave; tile < t_hi; tile += NWV) {
    const int64_t tok0 = (int64_t)tile * (8 * RB);
    uint32_t a_base[RB];                                 // tile row r = 2 (token of the tile) + step
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const int64_t tk = tok0 + 8 * rb + (l16 >> 1);
      a_base[rb] = tk < P.rows ? (uint32_t)(((int64_t)(l16 & 1) * P.rows + tk) * C) + 16u * lq : INV;
    }
    // positional term of k (pe[(t * N1 + n) * pe_ld + channel], n = token % N1): requested ahead of the operands
    float pev[RB][2][2][2];                              // [rb][m][t][cb]
    {
      const int nbase = (int)(tok0 % P.N1);
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          int n = nbase + 8 * rb + 2 * q + m;
          n = n >= P.N1 ? n - P.N1 : n;
          n = n >= P.N1 ? n - P.N1 : n;
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
              pev[rb][m][t][cb] = P.pe ? P.pe[((int64_t)t * P.N1 + n) * P.pe_ld + hd * 32 + 16 * cb + c] : 0.f;
        }
    }
    i32x4 acc[3][RB][CB];                                // (written by the main loop's first step)
    AddrPlain aa{a_base, kpv - lq};
    res_mainloop<RB, CB>(acc, A_rs, aa, ksteps, Wl, KP, lane);

    // ---------------- epilogue: BN (+ PE) + SN_q / SN_k over T' = 2, token gate, bytes -> E ----------------
    uint8_t* S = Sall + wave * STILE;
    uint32_t sel1, sel2;
    quad_sel(lane, sel1, sel2);
    const int m4 = c >> 2, ci = c & 3;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      uint32_t qbits[2] = {0u, 0u}, kbits[2] = {0u, 0u};    // bit 2 m + t of column block cb
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          float xq[2], xk[2], sq[2], sk[2];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int e = 2 * m + t;
            xq[t] = __builtin_fmaf(res_digits_f32<true>(acc[0][rb][cb][e], acc[1][rb][cb][e], acc[2][rb][cb][e]) * qc[cb], qa[cb], qb[cb]);
            xk[t] = __builtin_fmaf(res_digits_f32<true>(acc[0][rb][2 + cb][e], acc[1][rb][2 + cb][e], acc[2][rb][2 + cb][e]) * kc[cb], ka[cb], kb[cb]);
            if (P.pe) xk[t] = xk[t] + pev[rb][m][t][cb];
          }
          neuron_T<NK, 2>(xq, sq, P.sn_q, P.it_q);
          neuron_T<NK, 2>(xk, sk, P.sn_k, P.it_k);
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            qbits[cb] |= ((__float_as_uint(sq[t]) >> 29) & 1u) << (2 * m + t);
            kbits[cb] |= ((__float_as_uint(sk[t]) >> 29) & 1u) << (2 * m + t);
          }
        }
      // token gate: head sum of q per (token, step) = sum over the 2 column blocks and the 16 lanes of the row
      const uint32_t cnt = row_sum16(spread4(qbits[0]) + spread4(qbits[1]));      // byte 2 m + t: 0..32
      uint32_t gbits = 0;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        float a2[2], gt[2];
        a2[0] = (float)((cnt >> (16 * m)) & 0xFFu);
        a2[1] = (float)((cnt >> (16 * m + 8)) & 0xFFu);
        neuron_T<NK, 2>(a2, gt, P.sn2_q, P.it_2);
        gbits |= ((__float_as_uint(gt[0]) >> 29) & 1u) << (2 * m);
        gbits |= ((__float_as_uint(gt[1]) >> 29) & 1u) << (2 * m + 1);
      }
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        uint8_t* dst = S + (16 * rb + 4 * q + ci) * SP + 16 * cb + 4 * m4;
        *reinterpret_cast<uint32_t*>(dst) = quad_tr_bytes(spread4(kbits[cb] & gbits), sel1, sel2);
        if (KEEP) {
          *reinterpret_cast<uint32_t*>(dst + 32) = quad_tr_bytes(spread4(qbits[cb]), sel1, sel2);
          *reinterpret_cast<uint32_t*>(dst + 64) = quad_tr_bytes(spread4(kbits[cb]), sel1, sel2);
        }
      }
    }
    // byte tile -> E (and the q / k tape): tile row r = 16 rb + 2 tokl + t
    constexpr int PPR = SB / 16;
#pragma unroll
    for (int it = 0; it < (ROWS * PPR + 63) / 64; ++it) {
      const int pc = lane + 64 * it;
      if (pc < ROWS * PPR) {
        const int r = pc / PPR, k16 = pc - r * PPR;
        const int64_t tk = tok0 + 8 * (r >> 4) + ((r & 15) >> 1);
        if (tk < P.rows) {
          const int64_t grow = (int64_t)(r & 1) * P.rows + tk;
          const u32x4 v = *reinterpret_cast<const u32x4*>(S + r * SP + 16 * k16);
          uint8_t* dst = k16 < 2 ? P.e + grow * C + hd * 32 + 16 * k16
                                 : (k16 < 4 ? P.qs + grow * P.ldq + hd * 32 + 16 * (k16 - 2) : P.ks + grow * P.ldk + hd * 32 + 16 * (k16 - 4));
          *reinterpret_cast<u32x4*>(dst) = v;
        }
      }
    }
  }
}

bool res_env_off() {
  const char* e = sdf_sw(SW_RES);                    // A/B: 0 = the kernels these replaced (ms_wide.hip / qk_front / ms_mlp_fused / spike_gemm)
  return e && e[0] == '0';
}

// dynamic LDS above 64 KB needs the attribute, once per kernel function (process-wide, read-only afterwards)
template <class KernelT>
int res_raise(KernelT kern) {
  static std::atomic<uint64_t> done{0};                  // (one instance per kernel type: the template parameter is the function's type; one bit per device)
  return sdf_lds_opt_in(done, reinterpret_cast<const void*>(kern), 158 * 1024);
}

template <int T, int EPI, int NK, int AM, bool STRIP = false>
int res_pm_launch(const WidePmParams& P, dim3 grid, size_t lds, hipStream_t s) {
  if (P.K <= 256) {
    auto kern = res_pm_kernel<T, EPI, NK, AM, true, STRIP>;
    if (int rc = res_raise(kern)) return rc;
    SDF_LAUNCH(kern, grid, dim3(64 * NWV), lds, s, P);
  } else {
    auto kern = res_pm_kernel<T, EPI, NK, AM, false, STRIP>;
    if (int rc = res_raise(kern)) return rc;
    SDF_LAUNCH(kern, grid, dim3(64 * NWV), lds, s, P);
  }
  return 0;
}

bool res_strip(const WidePmParams& P) {
  const char* e = sdf_sw(SW_RES_STRIP);              // A/B: 0 = fragment-shaped loads everywhere
  if (e && e[0] == '0') return false;
  return P.cv_Cin != 0 || (!P.a_tiled && !P.zsrc);
}

template <int T, int AM>
int res_pm_launch_t(const WidePmParams& P, int epi, int nk, dim3 grid, size_t lds, hipStream_t s) {
  // row-major operands through the full-line loads + LDS strip: patch merging always, the plain fp32 product (the decoders' stacked taps,
  // fc2 on the parity tape's row-major spikes) where the rows are a plain tensor
  if (epi == 2 && res_strip(P)) return res_pm_launch<T, 2, 0, AM, true>(P, grid, lds, s);
  if (epi == 2) return res_pm_launch<T, 2, 0, AM>(P, grid, lds, s);
  if constexpr (AM == 0) {
    if (epi == 1) return nk == 0 ? res_pm_launch<T, 1, 0, 0>(P, grid, lds, s) : (nk == 1 ? res_pm_launch<T, 1, 1, 0>(P, grid, lds, s) : res_pm_launch<T, 1, 2, 0>(P, grid, lds, s));
    return nk == 0 ? res_pm_launch<T, 3, 0, 0>(P, grid, lds, s) : (nk == 1 ? res_pm_launch<T, 3, 1, 0>(P, grid, lds, s) : res_pm_launch<T, 3, 2, 0>(P, grid, lds, s));
  }
  return SDF_E_SHAPE;
}

size_t res_pm_lds(int K, int epi, int T, int nk) {
  constexpr int SP = s_pitch(32);
  size_t b = (size_t)res_wbytes(K, 32) + ((epi & 1) ? NWV * 80 * SP : 0) + NWV * 80 * 4 + 32 * 16;
  if (nk == 1) b += PSN_TABLE(20) * 4;
  if (epi == 2) b += NWV * RBW * 1024;                  // (the operand strips of the STRIP builds)
  (void)T;
  return b;
}

}  // namespace

// ---- host side -------------------------------------------------------------------------------------------------------------
bool res_pm_takes(const WidePmParams& P, int T, int epi) {
  if (res_env_off() || !P.res_stage) return false;
  if (T != 10 && T != 20) return false;
  if (P.K % 16 || P.K < 32 || P.K > 1024 || P.N % 32 || P.ksplit > 1 || epi < 1 || epi > 3) return false;      // (K <= 1024: 96 KB of resident digits)
  if (P.cv_Cin && !(epi == 2 && P.cv_Cin % 16 == 0 && 4 * P.cv_Cin == P.K && !P.zsrc && !P.a_tiled)) return false;      // (patch merging, transposed convolution)
  if (P.dc_cout && !(P.cv_Cin && P.dc_cout % 4 == 0 && P.N == 4 * P.dc_cout && P.no_resid)) return false;
  if (P.zsrc && P.K % 32) return false;
  return true;
}

int launch_res_pm(WidePmParams& P, int T, int epi, hipStream_t s) {
  const int PPW = 4 * (20 / T);
  const int64_t units = (P.P + PPW - 1) / PPW;
  if (units >= (1LL << 28)) return SDF_E_SHAPE;
  P.nunits = (int)units;
  P.ncg = P.N / 32;
  // one workgroup per compute unit, all resident in one round: the smallest number r of units per wave for which
  // (column groups) x (row ranges of 8 r units) fits the chip's 256 compute units
  int r = 1;
  while ((int64_t)P.ncg * ((units + 8 * r - 1) / (8 * r)) > 256) ++r;
  if (const char* e = sdf_sw(SW_RES_UPW)) { const int v = atoi(e); if (v >= 1 && v <= 4096) r = v; }     // tuning override: units per wave
  if (const char* e = sdf_sw(SW_RES_RMUL)) { const int v = atoi(e); if (v >= 1 && v <= 16) r *= v; }       // tuning: fewer, longer-lived workgroups
  P.passes = 8 * r;                                       // units per row range
  P.nrg = (int)((units + P.passes - 1) / P.passes);
  const int64_t items = (int64_t)P.ncg * P.nrg;
  if (items >= (1LL << 31) - 8) return SDF_E_SHAPE;
  const dim3 grid((unsigned)((items + 7) / 8 * 8));
  const int nk = (epi & 1) ? neuron_class(P.sn) : 0;
  const size_t lds = res_pm_lds(P.K, epi, T, nk);
  int rc;
  if (P.cv_Cin) {                                         // patch merging: cv_cpt carries ceil(2^16 / (C / 16)) for the per-piece quadrant decode
    P.cv_cpt = (65536 + (P.cv_Cin >> 4) - 1) / (P.cv_Cin >> 4);
    for (int p = 0; p < (P.K >> 4) + 8; ++p)
      if ((int)(((uint32_t)p * (uint32_t)P.cv_cpt) >> 16) != p / (P.cv_Cin >> 4)) return SDF_E_SHAPE;
    if (P.dc_cout) rc = T == 10 ? res_pm_launch_t<10, 3>(P, epi, nk, grid, lds, s) : res_pm_launch_t<20, 3>(P, epi, nk, grid, lds, s);
    else rc = T == 10 ? res_pm_launch_t<10, 2>(P, epi, nk, grid, lds, s) : res_pm_launch_t<20, 2>(P, epi, nk, grid, lds, s);
  } else {
    rc = T == 10 ? res_pm_launch_t<10, 0>(P, epi, nk, grid, lds, s) : res_pm_launch_t<20, 0>(P, epi, nk, grid, lds, s);
  }
  if (rc) return rc;
  hipError_t e = hipGetLastError();
  return e != hipSuccess ? (int)e : 0;
}

bool res_front_takes(const WideFrontParams& P) {
  if (res_env_off()) return false;
  return P.C % 32 == 0 && P.C >= 64 && P.C <= 512;        // (the head's q | k planes, whole K: 3 x 64 x C bytes of LDS)
}

int launch_res_front(WideFrontParams& P, bool keep, int nk, hipStream_t s) {
  const int64_t ntiles = (P.rows + 15) / 16;
  if (ntiles >= (1LL << 30)) return SDF_E_SHAPE;
  P.ntiles = (int)ntiles;
  int r = 1;
  while ((int64_t)P.nH * ((ntiles + 8 * r - 1) / (8 * r)) > 256) ++r;
  if (const char* e = sdf_sw(SW_RES_UPW)) { const int v = atoi(e); if (v >= 1 && v <= 4096) r = v; }
  if (const char* e = sdf_sw(SW_RES_RMUL)) { const int v = atoi(e); if (v >= 1 && v <= 16) r *= v; }
  P.ntiles_per = 8 * r;
  P.nrg = (int)((ntiles + P.ntiles_per - 1) / P.ntiles_per);
  const int64_t items = (int64_t)P.nrg * P.nH;
  if (items >= (1LL << 31) - 8) return SDF_E_SHAPE;
  const dim3 grid((unsigned)((items + 7) / 8 * 8));
  const size_t lds = (size_t)res_wbytes(P.C, 64) + NWV * 32 * s_pitch(keep ? 96 : 32);
#define SDF_RF(NK_)                                                                                       \
  do {                                                                                                    \
    if (keep) { auto kern = res_front_kernel<NK_, true>; if (int rc = res_raise(kern)) return rc;        \
                SDF_LAUNCH(kern, grid, dim3(64 * NWV), lds, s, P); }                             \
    else { auto kern = res_front_kernel<NK_, false>; if (int rc = res_raise(kern)) return rc;            \
           SDF_LAUNCH(kern, grid, dim3(64 * NWV), lds, s, P); }                                  \
  } while (0)
  if (nk == 0) SDF_RF(0); else if (nk == 1) SDF_RF(1); else SDF_RF(2);
#undef SDF_RF
  hipError_t err = hipGetLastError();
  return err != hipSuccess ? (int)err : 0;
}

}  // namespace sdfmm

#ifdef SDF_STAMP
extern "C" int sdf_debug_read_stamps_res(unsigned long long* host32, unsigned long long* census) {
  (void)hipMemcpyFromSymbol(census, HIP_SYMBOL(g_res_census), sizeof(g_res_census));
  return (int)hipMemcpyFromSymbol(host32, HIP_SYMBOL(g_res_stamp), sizeof(g_res_stamp));
}
#endif
