// Spike products that are SMALL IN ROWS against MANY weights, for gfx950: the 3x3 convolutions of the U-Net bottleneck's MS_ResBlocks
// (reference Spiking_modules.py:906-933: 1 080 rows x 768 columns x K = 6 912 at batch 1 - 0.8 MB of spikes against 16 MB of weight
// digits) and every other layer of that kind.  Row b of SURVEY.md section 8's table for rows a9 / a10.
//
// What the streaming kernels did with it (profiles/r3s_*, r4h_*): spike_mm_pp_kernel split K over 120 workgroups, wrote fp32 partials
// and a second launch (splitk_reduce) added them, a third ran the neuron: 9.7 + 48.4 + 5.6 us per convolution, the weights streamed
// at 0.4 TB/s.  The wide-stage main loop (ms_wide.hip) in a split-K form measured no better (50 + 20 us): with few rows per column
// group its workgroups are all pipeline fill, and the partial sums are 3 x the output.
//
// Here the K split happens INSIDE a workgroup: a workgroup owns one output tile - 80 rows (= (20 / T) x 4 positions x all T steps, the
// position-major unit of ms_wide.hip) x 32 columns - and its four waves take every fourth 64-deep K step of it.  No operand is shared
// between waves, so the main loop has NO barrier.  Per step a wave loads 5 spike pieces and 6 weight-digit pieces of 1 KB, one step
// ahead, and issues 30 v_mfma_i32_16x16x64_i8.  How the pieces are addressed decides the kernel (measured, round 4: 75 us with
// fragment-shaped loads - lane = row + 16 k-group, i.e. 16 B from each of 16 different cache lines per lane quad - against the
// texture path's one line per quad and clock):
//   spikes  : a lane QUAD reads the 64 contiguous bytes of one row (lane = 4 row + piece), the wave writes the 1 KB piece to its own
//             LDS strip as it arrived (ds_write_b128, lane-linear) and reads it back in MFMA order (ds_read_b128, lane = row + 16
//             k-group); the pieces of a row sit XOR-swizzled by 2 (row / 8) so that read is conflict-free (the 16-lane groups of
//             ds_read_b128, MI355X_MICROARCH.md section LDS).  Same-wave LDS operations execute in order: no barrier, no second strip;
//   weights : pre-tiled at pack time (sdf_tile_weight_i8x3: [N / 16][K / 64][plane][lane][16 B] - every fragment is 1 KB contiguous in
//             lane order), loaded straight into the registers the MFMA reads.  Row-major digit planes are read too (fragment-shaped
//             loads: the slow form, kept for callers that only hold those).  The four partial accumulators are exact integers: they meet in LDS (the two
// low digits first folded into one int32: 40 KB instead of 90), two barriers, then wave 0 holds the complete sums and runs the same
// epilogue as the wide stages - BN, shortcut, LIF / IF over T in registers, spike bytes through the quad transpose - so one launch
// replaces neuron + GEMM + reduce, and nothing but the output leaves the chip.  The grid is (column group) x (unit) with the column
// group slowest: the workgroups of an XCD (ids equal mod 8) share a contiguous range of column groups, whose weight digits its L2
// holds once (3 groups x 0.66 MB at the bottleneck) while every workgroup re-reads them; the 0.8 MB of spikes are L2-resident anyway.
// What bounds it: 22 KB of operands per 60 MFMAs and wave, i.e. the L2 -> L1 path (64 B / clk / CU), not the matrix pipe.
//
// Results are bit-equal to any other order of the same integer sums (wide_pm_kernel, the row-major reference in the tests).
#include "wide_common.h"
#include "switches.h"
#include <stdlib.h>

#ifdef SDF_STAMP
// diagnostic build only (tools/smallm_ablate.sh stamp): cycle accounting of every wave of the middle workgroup and every workgroup's
// life in 100 MHz real time
__device__ unsigned long long g_smallm_stamp[4 * 8];
__device__ unsigned long long g_smallm_census[2 * 1024];
#define MSTAMP(i) do { st[i] = __builtin_readcyclecounter(); } while (0)
#else
#define MSTAMP(i)
#endif

namespace sdfmm {
namespace {

struct SmallMParams {
  const uint8_t* A;          // AM 0: u8 [rows][K]; AM 1: NHWC u8 images, image = (b, t), K = 9 Cin in (tap, channel) order
  const int8_t* W;           // digit planes [3][N][K], or tiled [N / 16][K / 64][3][64][16 B]
  const float* cscale;       // (N) power-of-two scale of every output channel
  int N, K, HW;
  int64_t P;                 // positions = B * HW; rows = P * T in (B, T, HW) order
  const float *bias, *alpha, *beta;
  const float* resid;        // fp32 [rows][ldo] or null
  float* out;                // fp32 [rows][ldo] (EPI & 2)
  int ldo;
  uint8_t* out_spike;        // u8 [rows][ldsp] (EPI & 1)
  int ldsp;
  SdfNeuronCfg sn;
  float inv_tau;
  int ncg, nunits;
  int cv_H, cv_W, cv_Cin, cv_spt;      // convolution: 64-deep steps per tap
  int cv_inv;                          // ceil(2^16 / cv_spt): step / cv_spt == (step * cv_inv) >> 16 for every step the kernel forms
};

// EPI: 1 = neuron on BN(...) [+ shortcut] -> spikes, 2 = the fp32 value is stored, 3 = both.  AM: 0 = rows of a tensor, 1 = 3x3 taps.
// BT: the weight digits are tiled.  CB = column blocks of a tile (2: 32 columns, two workgroups per compute unit, one operand step in
// flight per wave; 3: 48 columns, one workgroup per compute unit, two steps in flight - fewer, larger tiles re-read less).
template <int T, int EPI, int NK, int AM, bool BT, int CB>
__global__ __launch_bounds__(256, CB == 2 ? 2 : 1) void smallm_kernel(SmallMParams P) {
  constexpr int RB = RBW, NS = CB == 2 ? 1 : 2, NBUF = NS + 1, ROWS = 16 * RB, SLOTS = 4 * RB, PPG = SLOTS / T, PPW = 4 * PPG, BN = 16 * CB;
  constexpr int SP1 = s_pitch(16), REDC = 3 * 2 * RB * 64 * 16;          // per column block: three contributors x (lo, hi) x RB quads x 64 lanes
  constexpr int STRIPB = 4 * RB * 1024, REDB = CB * REDC;
  static_assert(SLOTS % T == 0, "T must divide the 20 accumulator slots of a lane");
  static_assert(ROWS * SP1 <= REDC && STRIPB <= REDB, "the byte tiles and the strips lie inside the reduction buffer");
  __shared__ __attribute__((aligned(16))) uint8_t smem[REDB];              // main loop: the waves' strips; then the reduction buffer
  __shared__ int32_t rowtab[ROWS];
  __shared__ uint32_t masktab[AM == 1 ? ROWS : 1];
  __shared__ __attribute__((aligned(16))) float psn_tbl[NK == 1 ? PSN_TABLE(T) : 4];   // PSN: W (T x T) and b (spike_mm.h: psn_T_lds)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, lq = lane >> 4;
  int item = blockIdx.x;
  const int G = gridDim.x;
  if ((G & 7) == 0) item = (item & 7) * (G >> 3) + (item >> 3);
  if (item >= P.ncg * P.nunits) return;
#ifdef SDF_STAMP
  unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
  auto stamp_out = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MSTAMP(5);
    if (lane == 0 && wave == 0 && blockIdx.x < 1024) {
      g_smallm_census[2 * blockIdx.x] = rt0;
      g_smallm_census[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    }
    if (lane == 0 && blockIdx.x == gridDim.x / 2) {
      unsigned long long* o = g_smallm_stamp + 8 * wave;
      for (int i = 0; i < 5; ++i) o[i] = st[i + 1] > st[i] ? st[i + 1] - st[i] : 0;
      o[5] = st[5] - st[0]; o[6] = __builtin_amdgcn_s_memrealtime() - rt0; o[7] = gridDim.x;
    }
  };
#endif
  MSTAMP(0);
  const int cg = item / P.nunits, unit = item - cg * P.nunits;
  const int n0 = cg * BN, K = P.K, N = P.N, HW = P.HW;
  const __amdgpu_buffer_rsrc_t A_rs = make_rsrc(P.A), W_rs = make_rsrc(P.W);
  const int nst = K >> 6;

  // L2 warm-up.  The workgroups of an XCD stream the SAME few column groups' weights (and all of the spikes) at the same pace: left to
  // the main loop, every new line is one HBM / Infinity-Cache miss that all of them wait for - the stream ran at one miss latency per
  // step (measured: 32 us at 27 steps).  Instead every wave of the XCD touches its share of those bytes once, up front, one lane per
  // 128-byte line: the whole working set (2 - 3 MB per XCD) arrives in about one latency, the main loop hits L2.  The values are not
  // used; they are "consumed" behind the first operand requests (in-order return: that wait does not cover the operands).
  uint32_t pf[6] = {0, 0, 0, 0, 0, 0};
  if constexpr (BT) {
    const int per = G >> 3, first = (blockIdx.x & 7) * per;
    const uint32_t nw = 4u * (uint32_t)per, wi = 4u * (uint32_t)((blockIdx.x >> 3)) + (uint32_t)wave;
    const int cg_lo = first / P.nunits, cg_hi = min(P.ncg - 1, (first + per - 1) / P.nunits);
    const uint32_t cgb = (uint32_t)(CB * nst) * 3072u, wbase = (uint32_t)cg_lo * cgb, wlines = ((uint32_t)(cg_hi - cg_lo + 1) * cgb) >> 7;
    const uint32_t Lw = (wlines + nw - 1) / nw;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t ln = 64u * j + lane, line = wi * Lw + ln;
      pf[j] = __builtin_amdgcn_raw_buffer_load_b32(W_rs, (ln < Lw && line < wlines) ? wbase + line * 128u : INV, 0, 0);
    }
    const uint32_t alines = (uint32_t)(((int64_t)P.P * T * (AM == 1 ? P.cv_Cin : K)) >> 7), La = (alines + nw - 1) / nw;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint32_t ln = 64u * j + lane, line = wi * La + ln;
      pf[4 + j] = __builtin_amdgcn_raw_buffer_load_b32(A_rs, (ln < La && line < alines) ? line * 128u : INV, 0, 0);
    }
  }

  // activation row (or -1) of every tile row
  if (tid < ROWS) {
    const int rb = tid >> 4, i = tid & 15, qq = i >> 2, slot = 4 * rb + (i & 3);
    const int pp = slot / T, t = slot - pp * T;
    const uint32_t pos = (uint32_t)unit * PPW + qq * PPG + pp;
    int32_t g = -1;
    if (pos < (uint32_t)P.P) {
      const uint32_t b = pos / (uint32_t)HW, hw = pos - b * (uint32_t)HW;
      g = (int32_t)((b * T + t) * (uint32_t)HW + hw);
    }
    rowtab[tid] = g;
    if constexpr (AM == 1) {                               // which of the 9 taps of the row's pixel lie inside the image (once per row)
      uint32_t m = 0;
      if (g >= 0) {
        const uint32_t pix = (uint32_t)g % (uint32_t)HW, y = pix / (uint32_t)P.cv_W, xx = pix - y * (uint32_t)P.cv_W;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
          const int yy = (int)y + tp / 3 - 1, xw = (int)xx + tp % 3 - 1;
          if (yy >= 0 && yy < P.cv_H && xw >= 0 && xw < P.cv_W) m |= 1u << tp;
        }
      }
      masktab[tid] = m;
    }
  }
  if constexpr (NK == 1) psn_stage<T>(psn_tbl, P.sn, tid, 256);
  __syncthreads();
  // spike pieces: this lane loads piece (lane & 3) ^ swz of tile row 16 rb + lane / 4, swz = 2 (row / 8): LDS slot = lane
  uint32_t a_base[RB], a_mask[RB], b_off[CB];
  const int lr = lane >> 2, lp = (lane & 3) ^ ((lr >> 2) & 2);
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int32_t g = rowtab[16 * rb + lr];
    a_base[rb] = INV;
    a_mask[rb] = 0;
    if (g >= 0) {
      if constexpr (AM == 1) {
        a_base[rb] = (uint32_t)g * (uint32_t)P.cv_Cin + 16u * lp;
        a_mask[rb] = masktab[16 * rb + lr];
      } else {
        a_base[rb] = (uint32_t)g * (uint32_t)K + 16u * lp;
      }
    }
  }
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    if constexpr (BT) b_off[cb] = n0 + 16 * cb < N ? (uint32_t)(((n0 >> 4) + cb) * nst) * 3072u + 16u * lane : INV;
    else b_off[cb] = n0 + 16 * cb + l16 < N ? (uint32_t)(n0 + 16 * cb + l16) * (uint32_t)K + 16u * lq : INV;
  }
  // MFMA-order read of the strip: lane = row + 16 k-group -> slot 4 row + (k-group ^ swz)
  uint8_t* mystrip = smem + wave * (RB * 1024);            // (one step's spike pieces on their way to MFMA order)
  const uint32_t rd_off = (uint32_t)((4 * l16 + (lq ^ ((l16 >> 2) & 2))) * 16), wr_off = 16u * lane;

  i32x4 acc[3][RB][CB];
#pragma unroll
  for (int dg = 0; dg < 3; ++dg)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) acc[dg][rb][cb] = i32x4{0, 0, 0, 0};

  // ---------------- main loop: this wave's steps s = wave, wave + 4, ... ----------------
  constexpr bool DBL = CB == 3;                            // registers for two sets of MFMA-order spike fragments (one workgroup per CU)
  i32x4 araw[NS][RB], a2[DBL ? 2 : 1][RB], b[NBUF][3][CB];
  int sa_next = wave, sb_next = wave;                      // next step whose spikes / weights are requested (scalar)
  const uint32_t plane = (uint32_t)N * (uint32_t)K;
#ifdef SMX_NOA
  constexpr bool XA = false;                               // (diagnostic builds, tools/smallm_ablate.sh: one operand's loads read nothing)
#else
  constexpr bool XA = true;
#endif
#ifdef SMX_NOB
  constexpr bool XB = false;
#else
  constexpr bool XB = true;
#endif
  // spike piece rb of step sa_next -> raw registers.  Convolution: step -> (tap, channel offset) by a multiply (cv_inv = ceil(2^16 / steps
  // per tap), exact for the step counts the host admits); a step beyond the last has tap >= 9, i.e. no mask bit - the select is
  // arithmetic, a branch here would cost a vmcnt(0)
  auto issue_a1 = [&](int ra, int rb) __attribute__((always_inline)) {
    if constexpr (AM == 1) {
      const int tap = (int)(((uint32_t)sa_next * (uint32_t)P.cv_inv) >> 16), cstep = sa_next - tap * P.cv_spt;
      const uint32_t toff = (uint32_t)(((tap / 3 - 1) * P.cv_W + (tap % 3 - 1)) * P.cv_Cin), cin0 = (uint32_t)cstep * 64u;
      const uint32_t ok = (0u - ((a_mask[rb] >> tap) & 1u)) & (0u - (uint32_t)XA);
      araw[ra][rb] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(A_rs, ((a_base[rb] + toff) & ok) | (INV & ~ok), cin0, 0));
    } else {
      const bool in = XA && sa_next < nst;
      araw[ra][rb] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(A_rs, in ? a_base[rb] : INV, (uint32_t)sa_next * 64u, 0));
    }
  };
  // weight fragment f = (digit, column block) of step sb_next -> its ring registers
  auto issue_b1 = [&](int rbuf, int f) __attribute__((always_inline)) {
    const bool in = XB && sb_next < nst;
    const int dg = f / CB, cb = f - dg * CB;
    const uint32_t so = BT ? (uint32_t)(sb_next * 3 + dg) * 1024u : (uint32_t)dg * plane + (uint32_t)sb_next * 64u;
    b[rbuf][dg][cb] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(W_rs, in ? b_off[cb] : INV, so, 0));
  };
  auto strip_write1 = [&](int ra, int rb) __attribute__((always_inline)) {
    *reinterpret_cast<i32x4*>(mystrip + rb * 1024 + wr_off) = araw[ra][rb];
  };
  auto strip_read1 = [&](int set, int rb) __attribute__((always_inline)) {
    a2[set][rb] = *reinterpret_cast<const i32x4*>(mystrip + rb * 1024 + rd_off);
  };
  // Step i: 3 CB groups of RB MFMAs on (spike fragments of step i, weight fragments requested NBUF steps ago).  One wave per SIMD (or
  // two): whatever else the wave issues between two MFMAs of 16 cycles must be SHORT or the matrix pipe drains (measured: with the
  // side work in blocks between groups a step took 1 100 cycles for 720 of MFMA, loads reading nothing).  So everything the next steps
  // need is cut into one small piece behind each MFMA, pinned there by a fence:
  //   group 0      : spike piece rb of step i + 1 (requested NS steps ago) goes raw registers -> strip
  //   group 1      : that raw register is re-requested (step i + 1 + NS)
  //   group 2      : the strip is read back in MFMA order, piece rb (second fragment set; with one set - the 32-column tile, 128
  //                  registers - behind the last MFMA instead)
  //   group g >= 1 : the weight fragment group g - 1 used is re-requested (step i + NBUF); the last one behind the last MFMA
  // Same-wave LDS operations execute in order: one strip, no barrier.
  auto step = [&](int i) __attribute__((always_inline)) {
    const int cur = DBL ? (i & 1) : 0, nxt = DBL ? (cur ^ 1) : 0;
#pragma unroll
    for (int g = 0; g < 3 * CB; ++g) {
      const int dg = g / CB, cb = g - dg * CB;
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
#ifndef SMX_NOMFMA
        mfma_i8(acc[dg][rb][cb], a2[cur][rb], b[i % NBUF][dg][cb]);
#else
        asm volatile("" :: "v"(a2[cur][rb]), "v"(b[i % NBUF][dg][cb]));
#endif
        if (g == 0) strip_write1((i + 1) % NS, rb);
        if (g == 1) issue_a1((i + 1) % NS, rb);
        if (g == 2 && DBL) strip_read1(nxt, rb);
        if (g >= 1 && rb == (g == 1 ? RB - 1 : 0)) issue_b1(i % NBUF, g - 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    sa_next += 4;
    issue_b1(i % NBUF, 3 * CB - 1);
    sb_next += 4;
    if (!DBL) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) strip_read1(nxt, rb);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  MSTAMP(1);
#pragma unroll
  for (int j = 0; j < NS; ++j) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) issue_a1(j, rb);
    sa_next += 4;
  }
#pragma unroll
  for (int j = 0; j < NBUF; ++j) {
#pragma unroll
    for (int f = 0; f < 3 * CB; ++f) issue_b1(j, f);
    sb_next += 4;
  }
  asm volatile("" :: "v"(pf[0]), "v"(pf[1]), "v"(pf[2]), "v"(pf[3]), "v"(pf[4]), "v"(pf[5]));      // (the warm-up loads end here)
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) strip_write1(0, rb);
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) issue_a1(0, rb);
  sa_next += 4;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) strip_read1(0, rb);
  __builtin_amdgcn_sched_barrier(0);
  constexpr int U = DBL ? 6 : 2;                           // steps per round: every register ring is back where it started
  static_assert(U % NS == 0 && U % NBUF == 0, "ring sizes divide the round");
  const int nrounds = ((nst + 3) / 4 + U - 1) / U;         // (uniform over the workgroup: steps beyond the wave's last multiply zeros)
#pragma unroll 1
  for (int r = 0; r < nrounds; ++r) {
#pragma unroll
    for (int j = 0; j < U; ++j) step(j);
  }
  mfma_drain(acc);                                         // (the accumulators are read by vector instructions from here on)

  MSTAMP(2);
  // ---------------- the four partial sums meet ----------------
  // (lo, hi) = (d1 256 + d0, d2) as exact int32.  Wave c < CB finishes column block c: every other wave hands it its sums of that
  // block through LDS (one barrier; a second one in front because the buffer lies over the strips), so the epilogue runs on CB waves
  // side by side instead of one (round 4 stamps: tree reduction + one-wave epilogue were 12.7 k of a workgroup's 64 k cycles).
  const int c = lane & 15, q = lane >> 4;
  const bool owner = wave < CB;
  const int mycb = owner ? wave : 0;
  uint32_t xo[SLOTS];
  float res[SLOTS];
  float al = 1.f, be = 0.f, bs = 0.f, cs = 0.f;
  const int ncol = n0 + 16 * mycb + c;
  if (owner) {                                             // the shortcut values and column parameters travel under the reduction
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int32_t g = rowtab[16 * (s >> 2) + 4 * q + (s & 3)];
      xo[s] = (g >= 0 && ncol < N) ? ((uint32_t)g * (uint32_t)P.ldo + (uint32_t)ncol) * 4u : INV;
    }
    const __amdgpu_buffer_rsrc_t r_rs = make_rsrc(P.resid);
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) res[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_rs, P.resid ? xo[s] : INV, 0, 0));
    const int nc = ncol < N ? ncol : 0;
    al = P.alpha ? P.alpha[nc] : 1.f;
    be = P.alpha ? P.beta[nc] : 0.f;
    bs = P.bias ? P.bias[nc] : 0.f;
    cs = P.cscale[nc];
  }
  i32x4* red = reinterpret_cast<i32x4*>(smem);             // [column block][contributor 0..2][lo | hi][row block][lane]
  __builtin_amdgcn_sched_barrier(0);                       // (the folding of the low digits below must not be scheduled up here: 80 registers)
  __syncthreads();                                         // (every wave is out of its strip)
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
    if (wave != cb) {
      const int j = wave < cb ? wave : wave - 1;
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        red[(((cb * 3 + j) * 2 + 0) * RB + rb) * 64 + lane] = acc_read(acc[1][rb][cb]) * 256 + acc_read(acc[0][rb][cb]);
        red[(((cb * 3 + j) * 2 + 1) * RB + rb) * 64 + lane] = acc_read(acc[2][rb][cb]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  __syncthreads();
  MSTAMP(3);
#ifdef SDF_STAMP
  if (!owner) { st[4] = st[3]; stamp_out(); return; }
#else
  if (!owner) return;
#endif
  // ---------------- the sums of this wave's column block, one row block at a time, straight into the epilogue: BN (+ bias), shortcut,
  // [store], [neuron over T -> spike bytes].  (Formed for all five row blocks at once, the 40 sum registers + 120 of reads in flight
  // + the shortcut values did not fit the two-workgroups-per-CU budget of 128 + 128 registers: 9 - 12 spilled)
  const __amdgpu_buffer_rsrc_t x_rs = make_rsrc(P.out), o_rs = make_rsrc(P.out_spike);
  float val[SLOTS];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    i32x4 lo = i32x4{0, 0, 0, 0}, hi = i32x4{0, 0, 0, 0};
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)                        // (this wave's own sums of its block: a compile-time index per branch)
      if (wave == cb) { lo = acc_read(acc[1][rb][cb]) * 256 + acc_read(acc[0][rb][cb]); hi = acc_read(acc[2][rb][cb]); }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      lo += red[(((mycb * 3 + j) * 2 + 0) * RB + rb) * 64 + lane];
      hi += red[(((mycb * 3 + j) * 2 + 1) * RB + rb) * 64 + lane];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int s = 4 * rb + e;
      float v = __builtin_fmaf((float)hi[e], 65536.f, (float)lo[e]) * cs;
      v = v + bs;
      v = __builtin_fmaf(v, al, be);
      v = v + res[s];
      val[s] = v;
      if constexpr ((EPI & 2) != 0) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), x_rs, xo[s], 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if constexpr ((EPI & 1) != 0) {
    uint8_t* S = smem + mycb * REDC;                       // byte tile [80][16 + pad] over this wave's part of the buffer (its reads are done)
    uint32_t sel1, sel2;
    quad_sel(lane, sel1, sel2);
    const int m4 = (c >> 2), ci = c & 3;
    uint32_t bits = 0;
#pragma unroll
    for (int pp = 0; pp < PPG; ++pp) {
      float xs[T], sp[T];
#pragma unroll
      for (int t = 0; t < T; ++t) xs[t] = val[pp * T + t];
      neuron_any<NK, T>(xs, sp, P.sn, P.inv_tau, psn_tbl);
#pragma unroll
      for (int t = 0; t < T; ++t) bits |= ((__float_as_uint(sp[t]) >> 29) & 1u) << (pp * T + t);      // 1.0f has bit 29 set
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const uint32_t w = quad_tr_bytes(spread4(bits >> (4 * rb)), sel1, sel2);
      *reinterpret_cast<uint32_t*>(S + (16 * rb + 4 * q + ci) * SP1 + 4 * m4) = w;
    }
#pragma unroll
    for (int it = 0; it < (ROWS + 63) / 64; ++it) {
      const int r = lane + 64 * it;
      if (r < ROWS) {
        const int32_t g = rowtab[r];
        const u32x4 v = *reinterpret_cast<const u32x4*>(S + r * SP1);
        const uint32_t off = (g >= 0 && n0 + 16 * mycb < N) ? (uint32_t)g * (uint32_t)P.ldsp + (uint32_t)(n0 + 16 * mycb) : INV;
        __builtin_amdgcn_raw_buffer_store_b128(v, o_rs, off, 0, 0);
      }
    }
  }
#ifdef SDF_STAMP
  MSTAMP(4);
  stamp_out();
#endif
}

template <int T, int AM, bool BT, int CB>
void launch_t(const SmallMParams& P, int epi, int nk, dim3 grid, hipStream_t s) {
  if (epi == 2) SDF_LAUNCH((smallm_kernel<T, 2, 0, AM, BT, CB>), grid, dim3(256), 0, s, P);
  else if (epi == 1) {
    if (nk == 0) SDF_LAUNCH((smallm_kernel<T, 1, 0, AM, BT, CB>), grid, dim3(256), 0, s, P);
    else if (nk == 1) SDF_LAUNCH((smallm_kernel<T, 1, 1, AM, BT, CB>), grid, dim3(256), 0, s, P);
    else SDF_LAUNCH((smallm_kernel<T, 1, 2, AM, BT, CB>), grid, dim3(256), 0, s, P);
  } else {
    if (nk == 0) SDF_LAUNCH((smallm_kernel<T, 3, 0, AM, BT, CB>), grid, dim3(256), 0, s, P);
    else if (nk == 1) SDF_LAUNCH((smallm_kernel<T, 3, 1, AM, BT, CB>), grid, dim3(256), 0, s, P);
    else SDF_LAUNCH((smallm_kernel<T, 3, 2, AM, BT, CB>), grid, dim3(256), 0, s, P);
  }
}

template <int T, int AM>
void launch_bt(const SmallMParams& P, int epi, int nk, bool bt, int cb, dim3 grid, hipStream_t s) {
  if (bt) { if (cb == 3) launch_t<T, AM, true, 3>(P, epi, nk, grid, s); else launch_t<T, AM, true, 2>(P, epi, nk, grid, s); }
  else launch_t<T, AM, false, 2>(P, epi, nk, grid, s);
}

// fp32 digit planes [3][N][K] -> fragment order [N / 16][K / 64][3][64 lanes][16 B]: lane l of fragment (column block, step, plane) holds
// W[plane][16 block + l % 16][64 step + 16 (l / 16) ...]
__global__ __launch_bounds__(256) void tile_weight_i8x3_kernel(const int8_t* __restrict__ planes, int8_t* __restrict__ tiled, int N, int K) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x, total = (int64_t)3 * N * (K >> 4);
  if (i >= total) return;
  const int lane = (int)(i & 63);
  int64_t f = i >> 6;
  const int p = (int)(f % 3);
  f /= 3;
  const int nst = K >> 6, st = (int)(f % nst), cbk = (int)(f / nst);
  const int8_t* src = planes + ((int64_t)p * N + 16 * cbk + (lane & 15)) * K + 64 * st + 16 * (lane >> 4);
  *reinterpret_cast<u32x4*>(tiled + i * 16) = *reinterpret_cast<const u32x4*>(src);
}

}  // namespace

static bool smallm_neuron_ok(const SdfNeuronCfg& n) {
  if (n.kind == SDF_PSN) return n.psn_w != nullptr && n.psn_b != nullptr;      // (T x T, T = the kernel's time axis: staged in LDS)
  if (n.kind != SDF_LIF && n.kind != SDF_IF) return false;
  return sdf_tau_ok(n.kind, n.tau);
}

// 3x3 / stride 1 / pad 1 spike convolution in (B, T, H, W) row order with int8 digit planes, up to SMALLM_MAX_ROWS rows: beyond that the
// weight re-reads of the (unit x column group) grid outgrow what the streaming kernels pay (they keep those shapes)
constexpr int64_t SMALLM_MAX_ROWS = 64 * 80;
// the convolution's own limit (SDF_SMALLM_CONV_ROWS: tuning override): 32 000 rows since round 5 - configs[4]'s bottleneck (24 000 rows x
// 768 x 6 912) measured 146.9 -> 151.9 samples/s on this kernel against the streaming convolution + split-K it fell back to
static int64_t smallm_conv_rows() {
  if (const char* e = sdf_sw(SW_SMALLM_CONV_ROWS)) { const long v = atol(e); if (v >= 80) return v; }
  return 400 * 80;
}

bool smallm_conv_supports(const GemmParams& P) {
  const SdfSpikeGemmDesc& d = P.d;
  const ConvGeom& cv = P.cv;
  if (const char* e = sdf_sw(SW_SMALLM)) { if (e[0] == '0') return false; }
  if ((d.nsplit != SDF_PLANES_I8X3 && d.nsplit != SDF_PLANES_I8X3_TILED) || !d.col_scale) return false;
  if (cv.KWc != 3 || d.K != 9 * cv.Cin || cv.Cin % 64 || cv.sy != 1 || cv.sx != 1 || cv.OH != cv.H || cv.OW != cv.W) return false;
  if (cv.dy[0] != -1 || cv.dy[1] != 0 || cv.dy[2] != 1 || cv.dx[0] != -1 || cv.dx[1] != 0 || cv.dx[2] != 1) return false;
  if (d.N % 32 || d.out_rowmap || d.bias || d.add || d.zg_nH) return false;
  const int64_t hw = (int64_t)cv.H * cv.W, imgs = d.M / hw;
  int T = d.sn_T;
  if (T == 0) T = imgs % 10 == 0 ? 10 : (imgs % 20 == 0 ? 20 : 0);
  if (T != 10 && T != 20) return false;
  if (imgs % T || d.M > smallm_conv_rows()) return false;
  if (d.sn_T > 0) {
    if (!smallm_neuron_ok({d.sn_kind, d.tau, d.v_th, d.v_reset, d.soft_reset, d.psn_w, d.psn_b})) return false;
    if (d.pos_inner != hw || d.t_stride != hw || d.pos_ostride != (int64_t)T * hw || d.pos_count * T != d.M) return false;   // rows (b, t, pixel)
    if (!d.out_spike) return false;
  } else if (!d.out) {
    return false;
  }
  if (d.M * (int64_t)cv.Cin >= (1LL << 31) || d.M * (int64_t)d.N * 4 >= (1LL << 31) || (int64_t)d.N * d.K * 3 >= (1LL << 31)) return false;
  return sdf_aligned(d.A, 16) && sdf_aligned(d.Wp, 16) && (!d.out || sdf_aligned(d.out, 16)) && (!d.resid || sdf_aligned(d.resid, 16)) &&
         (!d.out_spike || sdf_aligned(d.out_spike, 16)) && (!d.out || d.ldo == d.N);
}

int launch_smallm_conv(const GemmParams& G, hipStream_t s) {
  const SdfSpikeGemmDesc& d = G.d;
  const ConvGeom& cv = G.cv;
  const int64_t hw = (int64_t)cv.H * cv.W, imgs = d.M / hw;
  int T = d.sn_T;
  if (T == 0) T = imgs % 10 == 0 ? 10 : 20;
  SmallMParams P = {};
  P.A = d.A; P.W = reinterpret_cast<const int8_t*>(d.Wp); P.cscale = d.col_scale; P.N = d.N; P.K = d.K; P.HW = (int)hw; P.P = (imgs / T) * hw;
  P.alpha = d.alpha; P.beta = d.beta; P.resid = d.resid; P.out = d.out; P.ldo = d.N;
  P.out_spike = d.sn_T > 0 ? d.out_spike : nullptr; P.ldsp = d.N;
  P.sn = {d.sn_kind, d.tau, d.v_th, d.v_reset, d.soft_reset, d.psn_w, d.psn_b};
  P.inv_tau = d.sn_T > 0 ? inv_tau_of(P.sn) : 0.f;
  P.cv_H = cv.H; P.cv_W = cv.W; P.cv_Cin = cv.Cin; P.cv_spt = cv.Cin / 64;
  P.cv_inv = (65536 + P.cv_spt - 1) / P.cv_spt;
  for (int st = 0; st < 9 * P.cv_spt + 64; ++st)            // (steps up to a few rounds beyond the last are formed and must decode to tap >= 9)
    if ((int)(((uint32_t)st * (uint32_t)P.cv_inv) >> 16) != st / P.cv_spt) return SDF_E_SHAPE;
  const int PPW = 4 * (20 / T);
  P.nunits = (int)((P.P + PPW - 1) / PPW);
  const bool bt = d.nsplit == SDF_PLANES_I8X3_TILED;
  // tile width: 32 columns, two workgroups per compute unit.  The 48-column tile (one workgroup per compute unit, two operand steps in
  // flight) is 2 us faster on its own at the bottleneck shape (25.2 vs 27.5 us) but holds 224 whole compute units where this one packs
  // 336 workgroups onto 168: with three forwards in flight the headline measured +0.9 % with the narrow tile (three alternating pairs
  // of runs on one box: 687.8 / 683.5 / 685.7 against 680.7 / 677.2 / 679.6 samples/s).  SDF_SMALLM_CB=3 selects the wide tile.
  int cb = 2;
  if (const char* e = sdf_sw(SW_SMALLM_CB)) { if (bt && e[0] == '3' && d.N % 48 == 0) cb = 3; else if (e[0] == '2') cb = 2; }   // tuning override
  P.ncg = d.N / (16 * cb);
  const int64_t items = (int64_t)P.ncg * P.nunits;
  const dim3 grid((unsigned)((items + 7) / 8 * 8));
  const int epi = d.sn_T > 0 ? (d.out ? 3 : 1) : 2, nk = d.sn_T > 0 ? neuron_class(P.sn) : 0;
  if (T == 10) launch_bt<10, 1>(P, epi, nk, bt, cb, grid, s); else launch_bt<20, 1>(P, epi, nk, bt, cb, grid, s);
  hipError_t e = hipGetLastError();
  return e != hipSuccess ? (int)e : 0;
}

// fc2 of the wide-stage MLP (reference Spiking_swin_transformer3D.py:175-178, :845): x += BN2( s2 W2^T ) in place on the (B, D, HW, C)
// stream, s2 = the hidden spikes row-major [tokens][Ch]; with emit_next also SN_emit over D of the updated x.  Measured against the wide
// main loop (round 4): 30.9 -> 19.9 us at stage 3 (1 080 tokens x 768 x 3 072: 336 tiles, all resident at two per compute unit), but
// 20.1 -> 24.0 us at stage 2 (4 320 x 384 x 1 536: 648 tiles of 6 steps per wave - two rounds of mostly prologue and reduction), so the
// form is taken while the tiles fit the chip in one round (<= 512).
bool smallm_fc2_supports(const SdfMsMlpDesc* d) {
  if (const char* e = sdf_sw(SW_SMALLM)) { if (e[0] == '0') return false; }
  if (const char* e = sdf_sw(SW_SMALLM_FC2)) { if (e[0] == '0') return false; }
  if (!d->fc2_tiled || !d->fc2_cscale || (d->D != 10 && d->D != 20) || d->Ch % 64 || d->C % 32) return false;
  const int64_t tokens = (int64_t)d->B * d->D * d->HW;
  if (tokens > SMALLM_MAX_ROWS || tokens * d->Ch >= (1LL << 31) || tokens * d->C * 4 >= (1LL << 31) || (int64_t)d->C * d->Ch * 3 >= (1LL << 31)) return false;
  if (d->emit_next && !smallm_neuron_ok(d->emit_sn)) return false;
  {
    const int64_t ppw = 4 * (20 / d->D), units = ((int64_t)d->B * d->HW + ppw - 1) / ppw;
    const char* e = sdf_sw(SW_SMALLM_FC2);
    if (units * (d->C / 32) > 512 && !(e && e[0] == '2')) return false;       // (SDF_SMALLM_FC2=2: at any size, tests / A/B)
  }
  return sdf_aligned(d->fc2_tiled, 16) && sdf_aligned(d->x, 16) && (!d->emit_next || sdf_aligned(d->emit_next, 16));
}

int launch_smallm_fc2(const SdfMsMlpDesc* d, const uint8_t* s2, hipStream_t s) {
  SmallMParams P = {};
  P.A = s2; P.W = d->fc2_tiled; P.cscale = d->fc2_cscale; P.N = d->C; P.K = d->Ch; P.HW = (int)d->HW; P.P = (int64_t)d->B * d->HW;
  P.alpha = d->fc2_alpha; P.beta = d->fc2_beta; P.resid = d->x; P.out = d->x; P.ldo = d->C;        // (a lane reads its shortcut values before it writes them)
  P.out_spike = d->emit_next; P.ldsp = d->C; P.sn = d->emit_sn; P.inv_tau = d->emit_next ? inv_tau_of(d->emit_sn) : 0.f;
  const int T = d->D, PPW = 4 * (20 / T);
  P.nunits = (int)((P.P + PPW - 1) / PPW);
  P.ncg = d->C / 32;                                              // 32-column tiles: two workgroups per compute unit
  const int64_t items = (int64_t)P.ncg * P.nunits;
  if (items >= (1LL << 31) - 8) return SDF_E_SHAPE;
  const dim3 grid((unsigned)((items + 7) / 8 * 8));
  const int epi = d->emit_next ? 3 : 2, nk = d->emit_next ? neuron_class(d->emit_sn) : 0;
  if (T == 10) launch_t<10, 0, true, 2>(P, epi, nk, grid, s); else launch_t<20, 0, true, 2>(P, epi, nk, grid, s);
  hipError_t e = hipGetLastError();
  return e != hipSuccess ? (int)e : 0;
}

// out (M, N) fp32 = [BN]( A (M, K) u8 x W^T + bias ) + resid with the digits in fragment order: the stacked-tap product of the first
// decoder (reference Spiking_modules.py:461-474 as one GEMM, 1 080 rows x 3 456 columns x K = 1 536) and anything else of that shape
bool smallm_gemm_supports(const GemmParams& P) {
  const SdfSpikeGemmDesc& d = P.d;
  if (const char* e = sdf_sw(SW_SMALLM)) { if (e[0] == '0') return false; }
  if (d.nsplit != SDF_PLANES_I8X3_TILED || !d.col_scale || d.sn_T > 0 || !d.out) return false;
  if (d.K % 64 || d.K < 64 || d.N % 32 || d.lda != d.K || d.out_rowmap || d.add || d.zg_nH) return false;
  if (d.M % 10 || d.M > smallm_conv_rows()) return false;         // (rows are walked as 10 "steps" x M / 10 "positions": any order serves the fp32 form;
                                                                  //  the row bound is the convolution's since round 6: ten samples' stacked-tap product is one launch)
  if (d.M * (int64_t)d.K >= (1LL << 31) || d.M * (int64_t)d.ldo * 4 >= (1LL << 31) || (int64_t)d.N * d.K * 3 >= (1LL << 31)) return false;
  return sdf_aligned(d.A, 16) && sdf_aligned(d.Wp, 16) && sdf_aligned(d.out, 16) && (!d.resid || sdf_aligned(d.resid, 16)) && d.ldo >= d.N;
}

int launch_smallm_gemm(const GemmParams& G, hipStream_t s) {
  const SdfSpikeGemmDesc& d = G.d;
  SmallMParams P = {};
  P.A = d.A; P.W = reinterpret_cast<const int8_t*>(d.Wp); P.cscale = d.col_scale; P.N = d.N; P.K = d.K; P.HW = (int)(d.M / 10); P.P = d.M / 10;
  P.alpha = d.alpha; P.beta = d.beta; P.bias = d.bias; P.resid = d.resid; P.out = d.out; P.ldo = (int)d.ldo;
  P.nunits = (int)((P.P + 7) / 8);
  int cb = 2;                                                     // (as the convolution: the narrow tile packs two workgroups per compute unit)
  if (const char* e = sdf_sw(SW_SMALLM_CB)) { if (e[0] == '3' && d.N % 48 == 0) cb = 3; else if (e[0] == '2') cb = 2; }   // tuning override
  P.ncg = d.N / (16 * cb);
  const int64_t items = (int64_t)P.ncg * P.nunits;
  if (items >= (1LL << 31) - 8) return SDF_E_SHAPE;
  const dim3 grid((unsigned)((items + 7) / 8 * 8));
  if (cb == 3) SDF_LAUNCH((smallm_kernel<10, 2, 0, 0, true, 3>), grid, dim3(256), 0, s, P);
  else SDF_LAUNCH((smallm_kernel<10, 2, 0, 0, true, 2>), grid, dim3(256), 0, s, P);
  hipError_t e = hipGetLastError();
  return e != hipSuccess ? (int)e : 0;
}

}  // namespace sdfmm

#ifdef SDF_STAMP
extern "C" int sdf_debug_read_stamps_smallm(unsigned long long* host32, unsigned long long* census) {
  (void)hipMemcpyFromSymbol(census, HIP_SYMBOL(g_smallm_census), sizeof(g_smallm_census));
  return (int)hipMemcpyFromSymbol(host32, HIP_SYMBOL(g_smallm_stamp), sizeof(g_smallm_stamp));
}
#endif

extern "C" int sdf_tile_weight_i8x3(const int8_t* planes, int8_t* tiled, int N, int K, void* stream) {
  if (!planes || !tiled) return SDF_E_NULL;
  if (N < 16 || N % 16 || K < 64 || K % 64 || (int64_t)3 * N * K >= (1LL << 31)) return SDF_E_SHAPE;
  if (!sdf_aligned(planes, 16) || !sdf_aligned(tiled, 16)) return SDF_E_ALIGN;
  const int64_t total = (int64_t)3 * N * (K >> 4);
  SDF_LAUNCH(sdfmm::tile_weight_i8x3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, sdf_stream(stream), planes, tiled, N, K);
  SDF_LAUNCH_CHECK();
  return 0;
}
