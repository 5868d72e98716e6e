// Wide stages (C >= 192: swin stages 1 - 3 of the shipped model) of the MS swin block for gfx950 - rows a5 / a6 / a7 of
// SURVEY.md section 8 at the shapes where the problem is SMALL IN ROWS (batch 1: 4 320 / 1 080 token rows against 0.6 - 9.4 MB of
// weight planes per layer).  The general spike GEMM (spike_gemm.hip: 128 x 32 tiles, two barriers per 96-deep stage, 64-bit index
// arithmetic per element) ran these layers at 4 - 7 % of the matrix peak with 24 - 31 vector instructions per MFMA
// (profiles/r3s_pmc_forward.txt); the block's seven launches took 110 / 135 us.  Here the four matrix products of a block are
// two kernels built on one main loop, and the neurons between them run in the producing kernel's epilogue:
//
//   wide_front_kernel : q|k = SN_q/k( BN( xs [Wq;Wk]^T ) [+ PE] ),  E = k AND SN2_q( head sums of q )      reference
//                       Spiking_swin_transformer3D.py:671-694 - one launch instead of GEMM + gate; a wave owns 8 RB tokens x one head
//   wide_pm_kernel    : "position-major" GEMM - a wave owns 80 rows = (20 / T) x 4 positions x all T time steps, so whatever neuron
//                       follows runs over T in the accumulator registers:
//                         proj : x += BN( Z Wp^T + b ) through the head scramble (:709-714, :810-820, :840), emitting the MLP's SN1(x)
//                         fc1  : s2 = SN2( BN1( s1 W1^T ) )                                             (:170-174)
//                         fc2  : x += BN2( s2 W2^T )                                                   (:175-178, :845)
//
// Main loop (both kernels): v_mfma_i32_16x16x64_i8.  Spike bytes {0, 1} ARE int8 values and the weights arrive as three int8 digit
// planes + a power-of-two scale per output channel (sdf_split_weight_i8x3: w = (d2 65536 + d1 256 + d0) s_n, 22 - 23 bits + sign
// against the channel's largest weight), so a spike x weight dot product is three exact int32 sums - no expansion of the spikes to
// 16-bit floats (that cost one vector instruction per byte and MFMA: 2 per MFMA at two column blocks), three MFMAs per 64 k
// where the fp16 hi / lo planes need four, order-independent and bit-reproducible; the sums meet in fp32 as
// fma(acc2, 65536, acc1 256 + acc0) s_n.  Spikes are the ROW operand and never touch LDS: a lane loads the 16 bytes (k-piece
// j = lane / 16 of a 64-deep step) of its row straight into the registers the MFMA reads, one 128-deep chunk ahead (raw buffer
// loads, invalid rows read zeros).  The workgroup's 4 waves share BN = 16 CB weight columns: 128-deep chunks of the three planes go
// global -> registers -> LDS one chunk ahead into a two-buffer ring, ONE barrier per chunk.  LDS weight layout
// [plane][k-piece][column ^ (k-piece & 7)] x 16 B: the eight 16-byte pieces of a 128-byte weight row are written by eight
// neighbouring lanes to eight different bank quads, and every fragment read (lane = column + 16 k-group) is conflict-free.
// Workgroups are ordered row group major: an XCD owns a contiguous range of rows, whose spikes it fetches once (its L2 serves
// the column groups' re-reads) beside one copy of the weights.
//
// Epilogues: BN (+ bias, + positional term) on the accumulators; LIF / IF over T per lane (neuron_T of spike_mm.h - the separately
// rounded op sequence of neuron.hip; compiled with -ffp-contract=off); spike bits -> bytes by one 24-bit multiply per 4 rows, a
// 4 x 4 byte transpose inside each lane quad (two DPP moves + two v_perm) so that a lane holds four consecutive CHANNELS of a row,
// one ds_write_b32 into a per-wave LDS tile and 16-byte global stores.  The fp32 shortcut is read and written in the accumulator
// layout (64-byte runs per row and column block; the buffers are L2-resident at these sizes).
#include "wide_common.h"
#include "switches.h"
#include <stdlib.h>
#include <type_traits>

#ifdef SDF_STAMP
// diagnostic build only (tools/stamp_wide.sh): cycle accounting of wave 0 of the middle workgroup, per kernel kind
// (0 = front, 1 = fc1, 2 = fc2, 3 = projection), and every workgroup's life in 100 MHz real time
__device__ unsigned long long g_wide_stamp[5 * 8];
__device__ unsigned long long g_wide_census[5 * 2 * 1024];
__device__ unsigned long long g_wide_loop[8];          // main-loop phases of the LAST launch (middle workgroup, thread 0)
#define WSTAMP(var) var = __builtin_readcyclecounter()
#define WSTAMP_DECL unsigned long long ws0 = 0, ws1 = 0, ws2 = 0, ws3 = 0, ws4 = 0; const unsigned long long wr0 = __builtin_amdgcn_s_memrealtime()
#define WSTAMP_OUT(kind)                                                                                          \
  do {                                                                                                            \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                              \
    WSTAMP(ws4);                                                                                                  \
    if (threadIdx.x == 0 && blockIdx.x < 1024) {                                                                  \
      g_wide_census[(kind) * 2048 + 2 * blockIdx.x] = wr0;                                                        \
      g_wide_census[(kind) * 2048 + 2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();                       \
    }                                                                                                             \
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) {                                                        \
      unsigned long long* o = g_wide_stamp + 8 * (kind);                                                          \
      o[0] = ws1 - ws0; o[1] = ws2 - ws1; o[2] = ws3 - ws2; o[3] = ws4 - ws3; o[4] = ws4 - ws0;                   \
      o[5] = __builtin_amdgcn_s_memrealtime() - wr0; o[6] = gridDim.x;                                            \
    }                                                                                                             \
  } while (0)
#else
#define WSTAMP(var)
#define WSTAMP_DECL
#define WSTAMP_OUT(kind)
#endif

namespace sdfmm {
namespace {

// ---------------------------------------------------------------------------------------------------------------------------
// The shared main loop.  acc[digit][rb][cb] += A[rows of this wave][K] x D_digit[BN columns][K]^T over K = 128 nchunks.
//   a_base[rb] : byte offset of this lane's 16-byte piece (row lane % 16 of row block rb, k-piece lane / 16 of 64-deep step 0), or INV
//   a_step     : bytes between two 64-deep steps of that piece (64 in a row-major tensor; 4 pieces x 80 rows x 16 B in a tiled one;
//                2 G through the head scramble)
//   w_goff[i]  : byte offset of this thread's i-th weight piece of chunk 0 in its buffer (or INV); chunk c adds 128 c
//   w_lds[i]   : where that piece goes inside a ring buffer (the dump slot for a thread without an i-th piece)
// The spike operand goes global -> registers in the MFMA's own layout, one chunk ahead (measured, profiles/r4c_*: staging it through
// LDS costs more in ds_write issue - 13 cycles per 16-byte store and wave - than the lane-per-row loads lose in the L1; with the
// producer writing the TILED layout [unit][k-piece][row][16 B] the 16 lanes of a k-group read 256 contiguous bytes).
// Weights: one barrier per chunk; chunk c + 1 is committed to the other ring buffer at the top of chunk c's MFMAs and chunk c + 2
// requested.  The loop body is straight-line (requests beyond the last chunk read zeros through INV offsets, no branches): the
// compiler's vmcnt bookkeeping stays exact, so the wait for what is committed next leaves the following chunk's loads in flight.
//   a_addr(ch, h, rb, voff, soff) : the caller's address of this lane's piece of chunk ch, step h, row block rb (plain rows:
//                voff = a_base[rb], soff = (2 ch + h) a_step; the 3x3 convolution folds its tap into both)
//   c0         : first chunk of this workgroup's K range (split-K), added to the weight addressing only
//   ksteps     : 64-deep steps of the K range (2 nchunks, or 2 nchunks - 1: K = 192 at swin stage 1 - the second step of the last chunk
//                then reads zeros on both sides, through INV offsets like everything beyond the range)
template <int RB, int CB, int WIT, class AAddr, class PrepA, class Extra>
__device__ __forceinline__ void wide_mainloop(i32x4 (&acc)[3][RB][CB], const __amdgpu_buffer_rsrc_t A_rs, AAddr a_addr, int nchunks, int ksteps, int c0,
                                              uint8_t* Wlds, const __amdgpu_buffer_rsrc_t W0_rs, const __amdgpu_buffer_rsrc_t W1_rs,
                                              const uint32_t (&w_goff)[WIT], const uint32_t (&w_lds)[WIT], int lane, PrepA prepare_a,
                                              Extra extra_requests) {
  constexpr int BN = 16 * CB, WBUF = w_buf(CB);
  u32x4 wreg[WIT];
  i32x4 aX[2][RB], aY[2][RB];
  const uint32_t w_kp16 = 16u * (threadIdx.x & 7);                     // this thread's pieces start at byte w_kp16 of a chunk (wide_pieces)
  auto wreq = [&](int ch) __attribute__((always_inline)) {
#ifdef WIDE_X_NOWLOAD
    const bool in = false;
#else
    const bool in = ch < nchunks;
#endif
    const uint32_t left = in ? (uint32_t)(ksteps - 2 * ch) * 64u : 0u;          // bytes of the range from this chunk on (>= 128: all pieces)
#pragma unroll
    for (int i = 0; i < WIT; ++i)
      wreg[i] = __builtin_amdgcn_raw_buffer_load_b128((i & 1) ? W1_rs : W0_rs, w_kp16 < left ? w_goff[i] : INV, (uint32_t)(c0 + ch) * KCH, 0);
  };
  auto w_commit = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < WIT; ++i) *reinterpret_cast<u32x4*>(Wlds + buf * WBUF + w_lds[i]) = wreg[i];
  };
  auto areq = [&](i32x4 (&a)[2][RB], int ch) __attribute__((always_inline)) {
#ifdef WIDE_X_NOALOAD
    const bool in = false;
#else
    const bool in = true;
#endif
    a_addr.chunk(ch);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        uint32_t voff, soff;
        a_addr.get(h, rb, voff, soff);
        a[h][rb] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(A_rs, (in && 2 * ch + h < ksteps) ? voff : INV, soff, 0));
      }
  };
  const int l16 = lane & 15, lj = lane >> 4;
  auto a_load1 = [&](i32x4& dst, int ch, int h, int rb) __attribute__((always_inline)) {
#ifdef WIDE_X_NOALOAD
    const bool in = false;
#else
    const bool in = 2 * ch + h < ksteps;
#endif
    uint32_t voff, soff;
    a_addr.get(h, rb, voff, soff);
    dst = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(A_rs, in ? voff : INV, soff, 0));
  };
  // One chunk of this wave: 2 steps x CB column blocks x 3 digit planes = units of RB MFMAs on one weight fragment.  The fragments
  // of units g + 1 and g + 2 are in flight while unit g multiplies (an LDS read takes longer than five MFMAs).  Everything else a
  // chunk needs rides in the shadow of those MFMAs, a few instructions behind each unit (one wave per SIMD: nothing else would
  // overlap them - measured in round 4: done before / after the MFMAs, commit + copies + requests cost as much as the MFMAs):
  //   units 0 .. WIT-1      : one piece of weight chunk c + 1 each goes registers -> the other ring buffer
  //   unit  WIT             : weight chunk c + 2 is requested
  //   units UPS .. UPS+RB-1 : step 0's spike registers are dead: each is re-requested (chunk c + 2: the two register sets alternate)
  //   behind the last unit  : the same for step 1
  // The fences pin that order (the scheduler would pull every read in front of its use and push the rest behind the MFMAs).
  auto chunk = [&](int c, int cur, i32x4 (&aC)[2][RB]) __attribute__((always_inline)) {
    const uint8_t* wb = Wlds + cur * WBUF;
    uint8_t* wn = Wlds + (cur ^ 1) * WBUF;
    constexpr int UPS = CB * 3, NU = 2 * UPS;
    static_assert(WIT + 1 <= UPS && UPS + RB <= NU, "the chunk's side work must fit its units");
    i32x4 b[3];
    a_addr.chunk(c + 2);                                              // (every spike request of this chunk is for chunk c + 2)
    auto load_b = [&](int g) __attribute__((always_inline)) {
      const int h = g / UPS, r = g - h * UPS, cb = r / 3, dg = r - 3 * cb, kp = 4 * h + lj;
      b[g % 3] = *reinterpret_cast<const i32x4*>(wb + ((dg * 8 + kp) * BN + ((cb * 16 + l16) ^ kp)) * 16);
    };
    load_b(0);
    load_b(1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < NU; ++g) {
      if (g + 2 < NU) load_b(g + 2);
      const int h = g / UPS, r = g - h * UPS, cb = r / 3, dg = r - 3 * cb;
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
#ifndef WIDE_X_NOMFMA
        mfma_i8(acc[dg][rb][cb], aC[h][rb], b[g % 3]);
#else
        asm volatile("" :: "v"(aC[h][rb]), "v"(b[g % 3]));
#endif
      }
      if (g < WIT) *reinterpret_cast<u32x4*>(wn + w_lds[g]) = wreg[g];
      if (g == WIT) wreq(c + 2);
      if (g >= UPS && g < UPS + RB) a_load1(aC[0][g - UPS], c + 2, 0, g - UPS);      // (step 0's registers are dead: chunk c + 2 moves in)
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) a_load1(aC[1][rb], c + 2, 1, rb);
  };
  // A launch starts with cold caches and address translations: the first TWO weight chunks and spike chunks are requested back
  // to back at the top (one cold latency instead of two), the caller's own loads behind them.
  u32x4 wreg2[WIT];
  wreq(0);                                                            // (first in the queue: the first wait below covers nothing else)
#pragma unroll
  for (int i = 0; i < WIT; ++i)
    wreg2[i] = __builtin_amdgcn_raw_buffer_load_b128((i & 1) ? W1_rs : W0_rs, w_kp16 < (1 < nchunks ? (uint32_t)(ksteps - 2) * 64u : 0u) ? w_goff[i] : INV,
                                                     (uint32_t)(c0 + 1) * KCH, 0);
  prepare_a();                                                        // the caller's row addressing: may load (projection: the inverse map)
  extra_requests();                                                   // the caller's own loads (shortcut values, column parameters): AHEAD of the
  areq(aX, 0);                                                        // spike chunks in the in-order return queue - all of these are cold lines
  areq(aY, 1);                                                        // another launch wrote, one latency covers them together; behind the
  w_commit(0);                                                        // second chunk they stalled the third chunk's operands (round 4 stamps)
#pragma unroll
  for (int i = 0; i < WIT; ++i) wreg[i] = wreg2[i];
  __syncthreads();
  // two chunks per round - the register sets aX / aY alternate as "current" and each is refilled (chunk c + 2) as it dies: no copies
  // between the sets (the single-chunk body moves the next chunk's 40 registers into the current one's every chunk; measured at stage 2:
  // fc2 20.1 -> 17.0 us); an odd count ends with one lone chunk on aX
  int c = 0;
#pragma unroll 1
  for (; c + 1 < nchunks; c += 2) {
    chunk(c, 0, aX);                                                  // (a wave without rows multiplies zeros: no branch around the accumulators)
    __syncthreads();
    chunk(c + 1, 1, aY);
    mfma_drain(acc);                                                  // (the exit edge of the round may shuffle accumulators: wide_common.h)
    __syncthreads();
  }
  if (c < nchunks) {                                                  // odd count: the last chunk alone
    chunk(c, 0, aX);
    __syncthreads();
  }
  mfma_drain(acc);                                                    // (the accumulators are read by vector instructions from here on)
}

// this thread's weight pieces of a 128-deep chunk: piece = (digit plane p, column col of the group, k-piece kp); 256 threads take
// 32 columns x 8 k-pieces per step.  row_of(p, col, i) -> byte offset of (plane p, column col, k = 0) in the weight buffer, or INV
template <int CB, class RowOf>
__device__ __forceinline__ void wide_pieces(uint32_t (&goff)[w_steps(CB)], uint32_t (&lds)[w_steps(CB)], int tid, RowOf row_of) {
  constexpr int BN = 16 * CB, WIT = w_steps(CB);
  const int kp = tid & 7, r = tid >> 3;
#pragma unroll
  for (int i = 0; i < WIT; ++i) {
    const int p0 = (32 * i) / BN, rem0 = (32 * i) % BN;
    int p = p0, col = rem0 + r;
    if (col >= BN) { col -= BN; ++p; }
    if (col >= BN) { col -= BN; ++p; }                               // (BN = 16: a 32-column step spans two planes)
    const bool ok = p < 3;
    const uint32_t g = ok ? row_of(p, col, i) : INV;
    goff[i] = g == INV ? INV : g + 16u * kp;
    lds[i] = ok ? (uint32_t)(((p * 8 + kp) * BN + (col ^ kp)) * 16) : (uint32_t)(w_buf(CB) - 16);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Position-major GEMM.
//
// TILED spike tensors (the hand-over format between two position-major launches): [unit][channel piece K / 16][row 0..79][16 B],
// unit = the 80 tile rows (20 / T x 4 positions x T steps) one wave owns, row = its MFMA tile row.  The producer's epilogue writes
// 1 280 contiguous bytes per channel piece; the consumer's 16 lanes of a k-group read 256 contiguous bytes.  The parity tape and
// a caller that made the spikes with the neuron kernel use the row-major form [rows][K] instead (bit-equal results: the integer
// sums do not depend on the order).
// (struct WidePmParams: wide_common.h - shared with the weight-resident row-loop kernels of ms_res.hip)

// EPI: 1 = neuron (spikes out), 2 = fp32 (+ shortcut), 3 = fp32 and the neuron on the updated shortcut stream
// AM (address mode of the spike operand): 0 = rows of a tensor (row-major / tiled / head scramble), 1 = 3x3 convolution taps (EPI 4),
// 2 = the 2x2 concatenation of patch merging (K = 4 C in quadrant order (dh, dw) = (q % 2, q / 2), reference
// Spiking_swin_transformer3D.py:965-970; cv_W / cv_Cin = the SOURCE map's width and channels, HW = the merged map's positions)
template <int T, int CB, int EPI, int NK, int AM = (EPI == 4 ? 1 : 0)>
__global__ __launch_bounds__(256, 1) void wide_pm_kernel(WidePmParams P) {
  constexpr int NW = 4, RB = RBW, ROWS = 16 * RB, SLOTS = 4 * RB, PPG = SLOTS / T, PPW = 4 * PPG;
  constexpr int BN = 16 * CB, WBUF = w_buf(CB), WIT = w_steps(CB);
  constexpr int SP = s_pitch(BN), STILE = ROWS * SP;
  constexpr int LDSB = 2 * WBUF > NW * STILE ? 2 * WBUF : NW * STILE;
  static_assert(SLOTS % T == 0, "T must divide the 20 accumulator slots of a lane");
  static_assert(!(EPI & 2) || CB <= 2, "the fp32 epilogue keeps every shortcut load in flight beside the accumulators: two column blocks");
  static_assert(EPI >= 1 && EPI <= 4, "epilogue: 1 neuron, 2 fp32, 3 both, 4 split-K partial");
  __shared__ __attribute__((aligned(16))) uint8_t smem[LDSB];
  __shared__ int32_t rowtab[NW * ROWS];
  __shared__ __attribute__((aligned(16))) float psn_tbl[NK == 1 ? PSN_TABLE(T) : 4];   // PSN: W (T x T) and b staged once (spike_mm.h: psn_T_lds)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, lq = lane >> 4;
  WSTAMP_DECL;
  WSTAMP(ws0);
  if constexpr (NK == 1) psn_stage<T>(psn_tbl, P.sn, tid, 256);      // (visible behind the main loop's first barrier)
  // (row-group range, column group), column group fastest: the workgroups of an XCD (ids equal mod 8) cover a contiguous range of rows
  int item = blockIdx.x;
  const int G = gridDim.x;
  if ((G & 7) == 0) item = (item & 7) * (G >> 3) + (item >> 3);
  const int nrgp = (P.nrg + P.passes - 1) / P.passes;
  const int nks = P.ksplit > 1 ? P.ksplit : 1;
  if (item >= P.ncg * nrgp * nks) return;
  const int ks = item / (P.ncg * nrgp);                   // K range slowest: the workgroups of one range share its weight slices in L2
  item -= ks * P.ncg * nrgp;
  const int rgp = item / P.ncg, cg = item - rgp * P.ncg;
  const int n0 = cg * BN;
  const int nch = P.ksplit > 1 ? P.cps : (P.K + KCH - 1) / KCH, c0 = P.ksplit > 1 ? ks * P.cps : 0;
  const int kst = P.ksplit > 1 ? 2 * P.cps : P.K / 64;
  const int K = P.K, N = P.N, HW = P.HW;
  const __amdgpu_buffer_rsrc_t A_rs = make_rsrc(P.A), W_rs = make_rsrc(P.W), x_rs = make_rsrc(P.x), o_rs = make_rsrc(P.out_spike);
  uint32_t w_goff[WIT], w_lds[WIT];
  wide_pieces<CB>(w_goff, w_lds, tid, [&](int p, int col, int) -> uint32_t {
    return n0 + col < N ? (uint32_t)((p * N + n0 + col) * K) : INV;
  });
  const int c = lane & 15, q = lane >> 4;
  float al[CB], be[CB], bs[CB], cs[CB];                   // BN / bias / digit scale of this lane's columns: requested in pass 0, behind the operands
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) al[cb] = be[cb] = bs[cb] = cs[cb] = 0.f;

#pragma unroll 1
  for (int pass = 0; pass < P.passes; ++pass) {
    const int rg = rgp * P.passes + pass;
    if (rg >= P.nrg) break;                                // (uniform over the workgroup)
    const int unit = rg * NW + wave;
    const bool active = unit < P.nunits;
    uint32_t a_base[RB], a_mask[RB];                      // (a_mask: convolution - which of the 9 taps of the row's pixel lie inside the image)
    const uint32_t a_step = __builtin_amdgcn_readfirstlane(P.a_tiled ? 4u * ROWS * 16u : (P.zsrc ? 2u * P.zg_G : 64u));
    uint32_t xo[(EPI & 2) ? SLOTS : 1];
    float res[(EPI & 2) ? CB : 1][(EPI & 2) ? SLOTS : 1];
    // row addressing of this wave's tile (runs behind the first weight request of the main loop)
    auto prepare_a = [&]() __attribute__((always_inline)) {
      // activation row (or -1) of every tile row of this wave
      for (int r = lane; r < ROWS; r += 64) {
        const int rb = r >> 4, i = r & 15, qq = i >> 2, slot = 4 * rb + (i & 3);
        const int pp = slot / T, t = slot - pp * T;
        const uint32_t pos = (uint32_t)unit * PPW + qq * PPG + pp;     // (positions and rows fit 31 bits: the host checks)
        int32_t g = -1;
        if (active && pos < (uint32_t)P.P) {
          const uint32_t b = pos / (uint32_t)HW, hw = pos - b * (uint32_t)HW;
          g = (int32_t)((b * T + t) * (uint32_t)HW + hw);
        }
        rowtab[wave * ROWS + r] = g;
      }
      asm volatile("" ::: "memory");                       // (same-wave LDS operations execute in order: no wait between the table's writes and reads)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        const int32_t g = rowtab[wave * ROWS + 16 * rb + l16];
        a_base[rb] = INV;
        a_mask[rb] = 0;
        if (AM == 2) {
          if (g >= 0) {                                      // merged position (b, t, h2, w2) -> source pixel (2 h2, 2 w2) of the same (b, t)
            const uint32_t W2 = ((uint32_t)P.cv_W + 1u) >> 1, img = (uint32_t)g / (uint32_t)HW, pix = (uint32_t)g - img * (uint32_t)HW;
            const uint32_t h2 = pix / W2, w2 = pix - h2 * W2;
            a_base[rb] = ((img * (uint32_t)P.cv_H + 2 * h2) * (uint32_t)P.cv_W + 2 * w2) * (uint32_t)P.cv_Cin + 16u * lq;
            const uint32_t hin = 2 * h2 + 1 < (uint32_t)P.cv_H ? 0xFu : 0x5u, win = 2 * w2 + 1 < (uint32_t)P.cv_W ? 0xFu : 0x3u;
            a_mask[rb] = hin & win;                          // odd sizes: the reference pads zeros in front of the neuron, SN(0) = 0
          }
        } else if (AM == 1) {
          if (g >= 0) {
            const uint32_t pix = (uint32_t)g % (uint32_t)HW, y = pix / (uint32_t)P.cv_W, xx = pix - y * (uint32_t)P.cv_W;
            a_base[rb] = (uint32_t)g * (uint32_t)P.cv_Cin + 16u * lq;
            uint32_t m = 0;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
              const int yy = (int)y + tap / 3 - 1, xw = (int)xx + tap % 3 - 1;
              if (yy >= 0 && yy < P.cv_H && xw >= 0 && xw < P.cv_W) m |= 1u << tap;
            }
            a_mask[rb] = m;
          }
        } else if (P.a_tiled) {
          if (active) a_base[rb] = (((uint32_t)unit * (uint32_t)(K >> 4) + (uint32_t)lq) * ROWS + 16 * rb + l16) * 16u;
        } else if (g >= 0) {
          a_base[rb] = P.zsrc ? (uint32_t)P.zsrc[g] + (uint32_t)(lq >> 1) * P.zg_G + 16u * (lq & 1) : (uint32_t)g * (uint32_t)K + 16u * lq;
        }
      }
    };
    // the shortcut values of this lane's outputs (and, once, its column parameters) are requested behind the first operand chunks
    // and arrive under the main loop
    auto late_requests = [&]() __attribute__((always_inline)) {
      if constexpr ((EPI & 2) != 0) {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
          const int32_t g = rowtab[wave * ROWS + 16 * (s >> 2) + 4 * q + (s & 3)];
          xo[s] = (g >= 0 && n0 + c < N) ? ((uint32_t)g * (uint32_t)P.ldo + (uint32_t)(n0 + c)) * 4u : INV;
        }
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
          for (int s = 0; s < SLOTS; ++s)
            res[cb][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rs, (n0 + 16 * cb + c < N && !P.no_resid) ? xo[s] : INV, 64u * cb, 0));
      }
      if (pass == 0) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
          const int n = n0 + 16 * cb + c;
          const int nc = n < N ? n : 0;
          al[cb] = P.alpha ? P.alpha[nc] : 1.f;
          be[cb] = P.alpha ? P.beta[nc] : 0.f;
          bs[cb] = P.bias ? P.bias[nc] : 0.f;
          cs[cb] = P.cscale[nc];
        }
      }
    };

    i32x4 acc[3][RB][CB];
#pragma unroll
    for (int dg = 0; dg < 3; ++dg)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) acc[dg][rb][cb] = i32x4{0, 0, 0, 0};
    WSTAMP(ws1);
    // this lane's piece address of (chunk, step, row block).  The convolution form (EPI 4 only) decodes the chunk once - chunk ->
    // (tap, channel offset), the tap moves the pixel - in scalar registers; the plain forms carry no trace of it.
    struct AddrPlain {
      const uint32_t* base; uint32_t step, ch2;
      __device__ __forceinline__ void chunk(int ch) { ch2 = 2u * (uint32_t)ch; }
      __device__ __forceinline__ void get(int h, int rb, uint32_t& voff, uint32_t& soff) const { voff = base[rb]; soff = (ch2 + (uint32_t)h) * step; }
    };
    struct AddrConv {
      const uint32_t* base; const uint32_t* mask; int cpt, W, Cin, c0; int tap[2]; uint32_t toff[2], cin0[2];
      __device__ __forceinline__ void chunk(int ch) {
        if constexpr (AM == 2) {
          // patch merging: `cpt` counts 64-deep STEPS per quadrant (C = 192: a 128-deep chunk straddles two quadrants) - each step
          // decodes its own quadrant (dh, dw) = (q % 2, q / 2)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int st = __builtin_amdgcn_readfirstlane(2 * (c0 + ch) + h);
            tap[h] = st / cpt;
            cin0[h] = (uint32_t)((st - tap[h] * cpt) * 64);
            toff[h] = (uint32_t)(((tap[h] & 1) * W + (tap[h] >> 1)) * Cin);
          }
        } else {
          const int cg_ = __builtin_amdgcn_readfirstlane(c0 + ch);
          tap[0] = tap[1] = cg_ / cpt;
          cin0[0] = (uint32_t)((cg_ - tap[0] * cpt) * KCH);
          cin0[1] = cin0[0] + 64u;
          toff[0] = toff[1] = (uint32_t)(((tap[0] / 3 - 1) * W + (tap[0] % 3 - 1)) * Cin);
        }
      }
      __device__ __forceinline__ void get(int h, int rb, uint32_t& voff, uint32_t& soff) const {
        voff = ((mask[rb] >> tap[h]) & 1u) ? base[rb] + toff[h] : INV;
        soff = cin0[h];
      }
    };
    typename std::conditional<AM != 0, AddrConv, AddrPlain>::type a_addr;
    if constexpr (AM != 0) a_addr = AddrConv{a_base, a_mask, P.cv_cpt, P.cv_W, P.cv_Cin, c0, {0, 0}, {0u, 0u}, {0u, 0u}};
    else a_addr = AddrPlain{a_base, a_step, 0u};
    wide_mainloop<RB, CB, WIT>(acc, A_rs, a_addr, nch, kst, c0, smem, W_rs, W_rs, w_goff, w_lds, lane, prepare_a, late_requests);
    WSTAMP(ws2);

    // ---------------- epilogue (an inactive wave has no valid row: its stores are dropped) ----------------
    float val[(EPI & 2) ? CB : 1][(EPI & 2) ? SLOTS : 1];   // fp32 forms: the updated shortcut stream, kept for the neuron
    if constexpr ((EPI & 2) != 0) {
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
          float v = digits_f32(acc[0][s >> 2][cb][s & 3], acc[1][s >> 2][cb][s & 3], acc[2][s >> 2][cb][s & 3]) * cs[cb];
          v = v + bs[cb];
          v = __builtin_fmaf(v, al[cb], be[cb]);
          v = v + res[cb][s];
          val[cb][s] = v;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), x_rs, (n0 + 16 * cb + c < N) ? xo[s] : INV, 64u * cb, 0);
        }
    }
    if constexpr (EPI == 4) {
      // split-K partial: the raw sums (digit scale applied) of this K range, fp32 [ks][row][N]; BN / shortcut / neuron happen in
      // wide_reduce_kernel on the k-ordered total
      const __amdgpu_buffer_rsrc_t p_rs = make_rsrc(P.partial + (int64_t)ks * P.P * T * N);
#pragma unroll
      for (int s = 0; s < SLOTS; ++s) {
        const int32_t g = rowtab[wave * ROWS + 16 * (s >> 2) + 4 * q + (s & 3)];
        const uint32_t po = (g >= 0 && n0 + c < N) ? ((uint32_t)g * (uint32_t)N + (uint32_t)(n0 + c)) * 4u : INV;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
          const float v = digits_f32(acc[0][s >> 2][cb][s & 3], acc[1][s >> 2][cb][s & 3], acc[2][s >> 2][cb][s & 3]) * cs[cb];
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), p_rs, (n0 + 16 * cb + c < N) ? po : INV, 64u * cb, 0);
        }
      }
    }
    WSTAMP(ws3);
    if constexpr ((EPI & 1) != 0) {
      uint8_t* S = smem + wave * STILE;                    // per-wave byte tile [80][SP]; aliases the weight ring (the main loop ends with a barrier)
      uint32_t sel1, sel2;
      quad_sel(lane, sel1, sel2);
      const int m4 = (c >> 2), ci = c & 3;
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        uint32_t bits = 0;
#pragma unroll
        for (int pp = 0; pp < PPG; ++pp) {
          float xs[T], sp[T];
#pragma unroll
          for (int t = 0; t < T; ++t) {
            const int s = pp * T + t;
            if constexpr (EPI == 1)
              xs[t] = __builtin_fmaf(digits_f32(acc[0][s >> 2][cb][s & 3], acc[1][s >> 2][cb][s & 3], acc[2][s >> 2][cb][s & 3]) * cs[cb] + bs[cb], al[cb], be[cb]);
            else
              xs[t] = val[cb][s];
          }
          neuron_any<NK, T>(xs, sp, P.sn, P.inv_tau, psn_tbl);
#pragma unroll
          for (int t = 0; t < T; ++t) bits |= ((__float_as_uint(sp[t]) >> 29) & 1u) << (pp * T + t);      // 1.0f has bit 29 set
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          const uint32_t w = quad_tr_bytes(spread4(bits >> (4 * rb)), sel1, sel2);
          *reinterpret_cast<uint32_t*>(S + (16 * rb + 4 * q + ci) * SP + 16 * cb + 4 * m4) = w;
        }
      }
#pragma unroll
      for (int it = 0; it < (ROWS * CB + 63) / 64; ++it) {
        const int pc = lane + 64 * it;
        if (pc < ROWS * CB) {
          // tiled: piece-major (80 consecutive rows of a channel piece are 1 280 contiguous bytes); row-major: row-major pieces
          const int r = P.out_tiled ? pc % ROWS : pc / CB, k16 = P.out_tiled ? pc / ROWS : pc % CB;
          const int32_t g = rowtab[wave * ROWS + r];
          const u32x4 v = *reinterpret_cast<const u32x4*>(S + r * SP + 16 * k16);
          uint32_t off = INV;
          if (n0 + 16 * k16 < N) {
            if (P.out_tiled) { if (active) off = (((uint32_t)unit * (uint32_t)(P.ldsp >> 4) + (uint32_t)((n0 >> 4) + k16)) * ROWS + r) * 16u; }
            else if (g >= 0) off = (uint32_t)g * (uint32_t)P.ldsp + (uint32_t)(n0 + 16 * k16);
          }
          __builtin_amdgcn_raw_buffer_store_b128(v, o_rs, off, 0, 0);
        }
      }
    }
    if (pass + 1 < P.passes) __syncthreads();              // the byte tiles alias the weight ring of the next pass
  }
  WSTAMP_OUT(EPI);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Attention front: q | k projection + BN (+ positional term) + SN_q / SN_k over the T' = 2 steps + token gate -> E.
// (struct WideFrontParams: wide_common.h)

template <int RB, int NK, bool KEEP>
__global__ __launch_bounds__(256, 1) void wide_front_kernel(WideFrontParams P) {
  constexpr int NW = 4, CB = 4, WBUF = w_buf(CB), WIT = w_steps(CB);           // 64 columns: 6 piece steps = 3 planes x (q | k)
  constexpr int ROWS = 16 * RB, SB = KEEP ? 96 : 32, SP = s_pitch(SB), STILE = ROWS * SP;
  constexpr int LDSB = 2 * WBUF > NW * STILE ? 2 * WBUF : NW * STILE;
  static_assert(WIT == 6, "piece step i = 2 p + (q | k)");
  __shared__ __attribute__((aligned(16))) uint8_t smem[LDSB];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, lq = lane >> 4;
  WSTAMP_DECL;
  WSTAMP(ws0);
  int item = blockIdx.x;
  const int G = gridDim.x;
  if ((G & 7) == 0) item = (item & 7) * (G >> 3) + (item >> 3);
  if (item >= P.nH * P.nrg) return;
  const int rg = item / P.nH, hd = item - rg * P.nH;                 // head fastest: an XCD owns a contiguous range of token tiles
  const int tile = rg * NW + wave;
  const bool active = tile < P.ntiles;
  const int C = P.C;
  const int64_t tok0 = (int64_t)tile * (8 * RB);

  const __amdgpu_buffer_rsrc_t A_rs = make_rsrc(P.xs), Wq_rs = make_rsrc(P.wq), Wk_rs = make_rsrc(P.wk);
  uint32_t a_base[RB];                                               // tile row r = 2 (token of the tile) + step
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int64_t tk = tok0 + 8 * rb + (l16 >> 1);
    a_base[rb] = (active && tk < P.rows) ? (uint32_t)(((int64_t)(l16 & 1) * P.rows + tk) * C) + 16u * lq : INV;
  }
  i32x4 acc[3][RB][CB];
#pragma unroll
  for (int dg = 0; dg < 3; ++dg)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) acc[dg][rb][cb] = i32x4{0, 0, 0, 0};
  // weight pieces: step i = 2 p + (0: the head's 32 q rows, 1: its 32 k rows) - the buffer of a piece is a compile-time choice
  uint32_t w_goff[WIT], w_lds[WIT];
  wide_pieces<CB>(w_goff, w_lds, tid, [&](int p, int col, int i) -> uint32_t {
    return (uint32_t)(p * ((i & 1) ? P.wk_plane : P.wq_plane) + (int64_t)(hd * 32 + (col & 31)) * C);
  });
  // BN / digit scales of this lane's columns and the positional term of k (pe[(t * N1 + n) * pe_ld + channel], n = token % N1):
  // requested behind the first operand chunks, they arrive under the main loop
  const int c = lane & 15, q = lane >> 4;
  float qa[2], qb[2], ka[2], kb[2], qc[2], kc[2];
  float pev[RB][2][2][2];                                 // [rb][m][t][cb]
  auto late_requests = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int ch = hd * 32 + 16 * cb + c;
      qa[cb] = P.q_al ? P.q_al[ch] : 1.f; qb[cb] = P.q_al ? P.q_be[ch] : 0.f;
      ka[cb] = P.k_al ? P.k_al[ch] : 1.f; kb[cb] = P.k_al ? P.k_be[ch] : 0.f;
      qc[cb] = P.q_cs[ch]; kc[cb] = P.k_cs[ch];
    }
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
  };
  WSTAMP(ws1);
  struct AddrPlain {
    const uint32_t* base; uint32_t ch2;
    __device__ __forceinline__ void chunk(int ch) { ch2 = 2u * (uint32_t)ch; }
    __device__ __forceinline__ void get(int h, int rb, uint32_t& voff, uint32_t& soff) const { voff = base[rb]; soff = (ch2 + (uint32_t)h) * 64u; }
  } a_addr{a_base, 0u};
  wide_mainloop<RB, CB, WIT>(acc, A_rs, a_addr, (C + KCH - 1) / KCH, C / 64, 0, smem, Wq_rs, Wk_rs, w_goff, w_lds, lane, [] {}, late_requests);
  WSTAMP(ws2);
  if (!active) return;                                    // (the main loop ends with a barrier: the per-wave byte tiles may alias the weight ring)

  // ---------------- epilogue ----------------
  const int lnl = lane;
  uint8_t* S = smem + wave * STILE;
  uint32_t sel1, sel2;
  quad_sel(lnl, sel1, sel2);
  const int m4 = c >> 2, ci = c & 3;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    uint32_t qbits[2] = {0u, 0u}, kbits[2] = {0u, 0u};      // bit 2 m + t of column block cb
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        float xq[2], xk[2], sq[2], sk[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int e = 2 * m + t;
          xq[t] = __builtin_fmaf(digits_f32(acc[0][rb][cb][e], acc[1][rb][cb][e], acc[2][rb][cb][e]) * qc[cb], qa[cb], qb[cb]);
          xk[t] = __builtin_fmaf(digits_f32(acc[0][rb][2 + cb][e], acc[1][rb][2 + cb][e], acc[2][rb][2 + cb][e]) * kc[cb], ka[cb], kb[cb]);
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
  WSTAMP(ws3);
  // byte tile -> E (and the q / k tape): tile row r = 16 rb + 2 tokl + t
  constexpr int PPR = SB / 16;
#pragma unroll
  for (int it = 0; it < (ROWS * PPR + 63) / 64; ++it) {
    const int pc = lnl + 64 * it;
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
  WSTAMP_OUT(0);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Second pass of the split-K convolution: out = fmaf(sum over the K ranges (in range order: deterministic), alpha, beta) (+ resid),
// stored as fp32 and / or turned into the spikes of the neuron over T.  A thread owns (position, 4 channels) with all T steps: all
// its loads are issued before the first use.  rows are (b, t, pixel) as everywhere in this file.
struct WideReduceParams {
  const float* partial;      // [ksplit][rows][N]
  int ksplit, N, HW;
  int64_t P;                 // positions = B * HW
  const float *alpha, *beta, *resid;
  float* out;                // fp32 [rows][N] or null
  uint8_t* out_spike;        // u8 [rows][N] or null
  SdfNeuronCfg sn;
  float inv_tau;
};

template <int T, int NK>
__global__ __launch_bounds__(256) void wide_reduce_kernel(WideReduceParams P) {
  const int n4 = P.N >> 2;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= P.P * n4) return;
  const int64_t pos = i / n4;
  const int c4 = (int)(i - pos * n4);
  const int64_t b = pos / P.HW, hw = pos - b * P.HW;
  const int64_t row0 = (b * T) * P.HW + hw, tstride = (int64_t)P.HW * P.N, kstride = P.P * T * P.N;
  const float* src = P.partial + row0 * P.N + 4 * c4;
  float4 v[T];
#pragma unroll
  for (int t = 0; t < T; ++t) v[t] = *reinterpret_cast<const float4*>(src + t * tstride);
  for (int k = 1; k < P.ksplit; ++k) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const float4 w = *reinterpret_cast<const float4*>(src + k * kstride + t * tstride);
      v[t].x += w.x; v[t].y += w.y; v[t].z += w.z; v[t].w += w.w;
    }
  }
  float4 al = make_float4(1.f, 1.f, 1.f, 1.f), be = make_float4(0.f, 0.f, 0.f, 0.f);
  if (P.alpha) { al = *reinterpret_cast<const float4*>(P.alpha + 4 * c4); be = *reinterpret_cast<const float4*>(P.beta + 4 * c4); }
  uint32_t pk[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    v[t].x = __builtin_fmaf(v[t].x, al.x, be.x); v[t].y = __builtin_fmaf(v[t].y, al.y, be.y);
    v[t].z = __builtin_fmaf(v[t].z, al.z, be.z); v[t].w = __builtin_fmaf(v[t].w, al.w, be.w);
    if (P.resid) {
      const float4 r = *reinterpret_cast<const float4*>(P.resid + row0 * P.N + 4 * c4 + t * tstride);
      v[t].x += r.x; v[t].y += r.y; v[t].z += r.z; v[t].w += r.w;
    }
    if (P.out) *reinterpret_cast<float4*>(P.out + row0 * P.N + 4 * c4 + t * tstride) = v[t];
    pk[t] = 0;
  }
  if (P.out_spike) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float xs[T], sp[T];
#pragma unroll
      for (int t = 0; t < T; ++t) xs[t] = e == 0 ? v[t].x : (e == 1 ? v[t].y : (e == 2 ? v[t].z : v[t].w));
      neuron_T<NK, T>(xs, sp, P.sn, P.inv_tau);
#pragma unroll
      for (int t = 0; t < T; ++t) pk[t] |= ((__float_as_uint(sp[t]) >> 29) & 1u) << (8 * e);
    }
#pragma unroll
    for (int t = 0; t < T; ++t) *reinterpret_cast<uint32_t*>(P.out_spike + row0 * P.N + 4 * c4 + t * tstride) = pk[t];
  }
}

__global__ __launch_bounds__(256) void zsrc_kernel(const int32_t* __restrict__ map, int32_t* __restrict__ zsrc, int64_t B_, int Tq, int N1,
                                                   int nH, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;                   // slice-map index: attention step t' of window b' is slice t' B_ + b'
  if (i >= total) return;
  const int32_t r = map[i];
  if (r < 0) return;
  const int n = (int)(i % N1);
  const int64_t sl = i / N1, t = sl / B_, b = sl - t * B_;
  // Z[t, b, n, g * 32 + d] = E_flat[((((b nH + g) T' + t) N1 + n) 32 + d]: offset of (g = 0, d = 0)
  zsrc[r] = (int32_t)((((b * nH) * Tq + t) * N1 + n) * 32);
}

// neurons the epilogues run: LIF / IF (classes 0 and 2 of spike_mm.h) and the PSN (class 1: its T x T matrix and bias, T = the
// kernel's time axis, staged in LDS or - T' = 2 - read as scalars)
bool neuron_ok(const SdfNeuronCfg& n) {
  if (n.kind == SDF_PSN) return n.psn_w != nullptr && n.psn_b != nullptr;
  if (n.kind != SDF_LIF && n.kind != SDF_IF) return false;
  return sdf_tau_ok(n.kind, n.tau);
}

// column blocks per wave: the widest tile that still leaves >= 600 waves (the chip has 1 024 SIMDs), else the narrowest
int pick_cb(int64_t units, int N, const int* cbs, int ncb) {
  for (int i = ncb - 1; i >= 0; --i)
    if (N % (16 * cbs[i]) == 0 && units * (N / (16 * cbs[i])) >= 600) return cbs[i];
  for (int i = 0; i < ncb; ++i)
    if (N % (16 * cbs[i]) == 0) return cbs[i];
  return 0;
}

template <int T, int CB, int EPI>
int launch_pm_nk(const WidePmParams& P, int nk, dim3 grid, hipStream_t s) {
  if constexpr (EPI == 2) {
    SDF_LAUNCH((wide_pm_kernel<T, CB, EPI, 0>), grid, dim3(256), 0, s, P);
  } else {
    if (nk == 0) SDF_LAUNCH((wide_pm_kernel<T, CB, EPI, 0>), grid, dim3(256), 0, s, P);
    else if (nk == 1) SDF_LAUNCH((wide_pm_kernel<T, CB, EPI, 1>), grid, dim3(256), 0, s, P);
    else SDF_LAUNCH((wide_pm_kernel<T, CB, EPI, 2>), grid, dim3(256), 0, s, P);
  }
  return 0;
}

template <int T>
int launch_pm_t(WidePmParams& P, int epi, hipStream_t s) {
  constexpr int PPW = 4 * (20 / T);
  const int64_t units = (P.P + PPW - 1) / PPW;
  // the fp32 epilogues hold the shortcut values and the updated stream beside the accumulators: two column blocks; the neuron-only
  // epilogue (fc1) takes three where that still fills the chip
  static const int cbs[2] = {2, 3};
  int cb = pick_cb(units, P.N, cbs, epi == 1 ? 2 : 1);
  if (const char* e = sdf_sw(SW_WIDE_CB)) {                     // tuning override (fc1): 2 / 3
    const int v = e[0] - '0';
    if (epi == 1 && (v == 2 || v == 3) && P.N % (16 * v) == 0) cb = v;
  }
  if (!cb || units >= (1LL << 28)) return SDF_E_SHAPE;
  P.nunits = (int)units;
  P.nrg = (int)((units + 3) / 4);
  P.ncg = P.N / (16 * cb);
  // one workgroup per compute unit (the kernels take most of its registers): a launch of up to four rounds runs as ONE round of
  // workgroups that walk several row groups (the prologue is paid once, no second dispatch wave)
  const int64_t all = (int64_t)P.ncg * P.nrg;
  P.passes = all > 256 && all <= 1024 ? (int)((all + 255) / 256) : 1;
  if (const char* e = sdf_sw(SW_WIDE_PASSES)) { const int v = atoi(e); if (v >= 1 && v <= 8) P.passes = v; }     // tuning override
  const int64_t items = (int64_t)P.ncg * ((P.nrg + P.passes - 1) / P.passes);
  if (items >= (1LL << 31) - 8) return SDF_E_SHAPE;
  const dim3 grid((unsigned)((items + 7) / 8 * 8));
  const int nk = neuron_class(P.sn);
  if (epi == 1) return cb == 3 ? launch_pm_nk<T, 3, 1>(P, nk, grid, s) : launch_pm_nk<T, 2, 1>(P, nk, grid, s);
  if (epi == 2) return launch_pm_nk<T, 2, 2>(P, 0, grid, s);
  return launch_pm_nk<T, 2, 3>(P, nk, grid, s);
}

int launch_pm(WidePmParams& P, int T, int epi, hipStream_t s) {
  if (res_pm_takes(P, T, epi)) return launch_res_pm(P, T, epi, s);      // narrow stages: weights LDS-resident, row loop (ms_res.hip)
  int rc;
  switch (T) {
    case 10: rc = launch_pm_t<10>(P, epi, s); break;
    case 20: rc = launch_pm_t<20>(P, epi, s); break;
    default: return SDF_E_SHAPE;
  }
  if (rc) return rc;
  hipError_t e = hipGetLastError();
  return e != hipSuccess ? (int)e : 0;
}

// These kernels are built for problems that are small in rows (one workgroup per compute unit, a prologue per 320 rows): up to 400
// 80-row units (configs[4]'s stage 3 - 24 000 rows - measured +1 % on them: 119.5 against 118.1 samples/s).  Larger problems
// (configs[4]: 96 000 rows at stage 2) keep the streaming kernels, which they fill: measured 119 -> 109 samples/s with the limit lifted
// (SDF_WIDE=2: tests, A/B).  Round 5: 131 072 rows - with the row-loop kernels of ms_res.hip taking the front, the projection and fc1
// of a block at any row count, the limit only decides whether fc2 of the C >= 384 stages (K > 1 024: not theirs) runs here or drags the
// WHOLE block back to the streaming kernels: configs[4] (96 000 rows at stage 2) 129.7 -> 143.4 samples/s with it lifted
// (profiles/r5ao_config5_kernel_stats.txt, same box A/B).
static int64_t wide_max_rows() {                      // (SDF_WIDE_MAXROWS: tuning override)
  if (const char* e = sdf_sw(SW_WIDE_MAXROWS)) { const long v = atol(e); if (v >= 80) return v; }
  return 131072;
}
#define WIDE_MAX_ROWS wide_max_rows()
bool wide_env_any() {
  const char* e = sdf_sw(SW_WIDE);
  return e && e[0] == '2';
}
bool wide_env_off() {
  const char* e = sdf_sw(SW_WIDE);
  return e && e[0] == '0';
}
// narrow stages (C <= 192: swin stages 0 - 1): the weight-resident row-loop kernels of ms_res.hip (SDF_RES=0: off - A/B)
// Which stages take them is a measured choice, and the measure is CHIP TIME, not latency: bench.py keeps three forwards in flight, so a
// launch costs the headline (compute units held) x (duration).  Same box, alternating runs (round 5, profiles/r5o_routing_ab.txt,
// tools/res_routing_ab.sh): round 4's kernels everywhere 705 - 707 samples/s; these kernels at stage 1 only 734 - 737; at stages 1 - 2
// 751 - 759; at stages 1 - 3 756 - 761 (the default) - although at C = 384 / 768 a launch is no faster alone (proj 19 against 15 us: but on
// 88 compute units instead of 168).  At C = 96 (stage 0, K = one and a half steps) the block is 112 us against 92 us on qk_front +
// spike_gemm + ms_mlp_fused alone and equal in flight (754 - 758): it keeps round 3's kernels (lower latency); SDF_RES_MINC lowers the
// bound (tests, A/B).  Units per wave beyond what fills the chip once (SDF_RES_RMUL = 2, 3) trade latency for nothing (753, 711 - 723).
int res_minc() {
  if (const char* e = sdf_sw(SW_RES_MINC)) { const int v = atoi(e); if (v >= 32 && v <= 192) return v; }
  return 128;
}
int res_maxc() {                                         // (SDF_RES_MAXC: tuning override - K = C <= 768 fits the resident image)
  if (const char* e = sdf_sw(SW_RES_MAXC)) { const int v = atoi(e); if (v >= 32 && v <= 768) return v; }
  return 768;
}
bool res_stage_ok(int C, bool merge = false) {
  const char* e = sdf_sw(SW_RES);
  if ((e && e[0] == '0') || C < 64 || C % 32) return false;
  // (the merge of a 192-channel stage, K = 768: 24.6 us on 88 compute units against 16.1 us on 168 with the K-ring kernel - slower alone,
  //  less chip time with three forwards in flight; K = 4 C <= 1024 is what the resident image admits)
  if (merge) return C <= 256;
  return C >= res_minc() && C <= res_maxc();
}

}  // namespace

// ---- host side -------------------------------------------------------------------------------------------------------------
// Shapes the wide forms are built for: int8 digit planes beside the 16-bit ones (the caller packs both; the default two-plane
// mode only - the exact three-plane and the one-plane bf16 modes keep the general kernels), LIF / IF / PSN neurons (round 5: the PSN's
// T x T matrix is staged in LDS, its rows read as broadcasts), C >= 192 in steps of 64 with Ch % 96 == 0, T in {10, 20}, at most
// WIDE_MAX_ROWS (131 072, SDF_WIDE_MAXROWS) rows, operands within the kernels' 31-bit buffer offsets.
bool ms_wide_mlp_supports(const SdfMsMlpDesc* d) {
  if (wide_env_off() || (d->flags & SDF_MLP_NARROW)) return false;
  if (!d->fc1_digits || !d->fc1_cscale || !d->fc2_digits || !d->fc2_cscale) return false;
  const bool res = res_stage_ok(d->C) && d->C <= 192 && d->Ch % 32 == 0 && d->Ch <= 768;      // (ms_res.hip: any row count, K % 16 == 0)
  if (d->nsplit != 2 || (!res && (d->C < 192 || d->C % 64 || d->Ch % 64 || d->Ch % 96))) return false;
  if (d->D != 10 && d->D != 20) return false;
  if (!neuron_ok(d->sn1) || !neuron_ok(d->sn2) || (d->emit_next && !neuron_ok(d->emit_sn))) return false;
  const int64_t tokens = (int64_t)d->B * d->D * d->HW;
  if (tokens * d->Ch >= (1LL << 31) || tokens * d->C * 4 >= (1LL << 31)) return false;
  if (tokens > WIDE_MAX_ROWS && !wide_env_any() && !res) return false;
  if (!d->fc1_alpha || !d->fc1_beta || !d->fc2_alpha || !d->fc2_beta) return false;
  return sdf_aligned(d->x, 16) && sdf_aligned(d->fc1_digits, 16) && sdf_aligned(d->fc2_digits, 16);
}

// s1 = SN1(x) must already be in `s1` (row-major u8 [tokens][C], or tiled); s2 receives the hidden spikes (row-major for the
// parity tape, else tiled); x is updated in place
int launch_ms_wide_mlp(const SdfMsMlpDesc* d, const uint8_t* s1, bool s1_tiled, uint8_t* s2, bool s2_tiled, hipStream_t s) {
  const bool fc2_small = smallm_fc2_supports(d);            // few tokens against a long K: fc2 on the small-M kernel (reads s2 row-major)
  if (fc2_small) s2_tiled = false;
  WidePmParams P = {};
  P.A = s1; P.a_tiled = s1_tiled; P.out_tiled = s2_tiled; P.W = d->fc1_digits; P.cscale = d->fc1_cscale; P.N = d->Ch; P.K = d->C; P.HW = (int)d->HW; P.P = (int64_t)d->B * d->HW;
  P.alpha = d->fc1_alpha; P.beta = d->fc1_beta;
  P.out_spike = s2; P.ldsp = d->Ch; P.sn = d->sn2; P.inv_tau = inv_tau_of(d->sn2);
  P.res_stage = res_stage_ok(d->C);                        // (per launch: res_pm_takes admits K <= 768)
  int rc = launch_pm(P, d->D, 1, s);
  if (rc) return rc;
  if (fc2_small) return launch_smallm_fc2(d, s2, s);
  WidePmParams Q = {};
  Q.A = s2; Q.a_tiled = s2_tiled;
  Q.W = d->fc2_digits; Q.cscale = d->fc2_cscale; Q.N = d->C; Q.K = d->Ch; Q.HW = (int)d->HW; Q.P = P.P;
  Q.alpha = d->fc2_alpha; Q.beta = d->fc2_beta; Q.x = d->x; Q.ldo = d->C;
  Q.sn = d->emit_next ? d->emit_sn : d->sn2; Q.inv_tau = inv_tau_of(Q.sn);
  Q.out_spike = d->emit_next; Q.ldsp = d->C; Q.out_tiled = 0;     // the next layer's first neuron on the updated stream, row-major
  Q.res_stage = P.res_stage;
  return launch_pm(Q, d->D, d->emit_next ? 3 : 2, s);
}

// ---- patch merging on the wide main loop: out = BN( [2x2 concat of the spikes] W^T ) -------------------------------------------------
bool wide_merge_supports(const SdfMsMergeDesc* d) {
  if (wide_env_off()) return false;
  if (d->B < 1 || d->H < 1 || d->W < 1 || (d->D != 10 && d->D != 20)) return false;
  const bool res = res_stage_ok(d->C, true);                // (ms_res.hip: the quadrant of every 16-byte piece decoded per lane, any row count)
  if ((d->C % 64 && !res) || d->C < 64 || d->N % 32) return false;
  const int64_t rows = (int64_t)d->B * d->D * ((d->H + 1) / 2) * ((d->W + 1) / 2);
  if (rows > WIDE_MAX_ROWS && !wide_env_any() && !res) return false;
  if (rows * 4 * d->C >= (1LL << 31) || rows * (int64_t)d->N * 4 >= (1LL << 31) || (int64_t)d->N * 4 * d->C * 3 >= (1LL << 31)) return false;
  return sdf_aligned(d->spikes, 16) && sdf_aligned(d->digits, 16) && sdf_aligned(d->out, 16);
}

int launch_wide_merge(const SdfMsMergeDesc* d, hipStream_t s) {
  WidePmParams P = {};
  P.A = d->spikes; P.W = d->digits; P.cscale = d->cscale; P.N = d->N; P.K = 4 * d->C;
  P.HW = ((d->H + 1) / 2) * ((d->W + 1) / 2); P.P = (int64_t)d->B * P.HW;
  P.alpha = d->alpha; P.beta = d->beta; P.x = d->out; P.ldo = d->N; P.no_resid = 1;
  P.cv_H = d->H; P.cv_W = d->W; P.cv_Cin = d->C; P.cv_cpt = d->C / 64;          // (64-deep steps per quadrant)
  const int T = d->D, PPW = 4 * (20 / T);
  P.res_stage = res_stage_ok(d->C, true);
  if (res_pm_takes(P, T, 2)) return launch_res_pm(P, T, 2, s);
  const int64_t units = (P.P + PPW - 1) / PPW;
  if (units >= (1LL << 28)) return SDF_E_SHAPE;
  P.nunits = (int)units; P.nrg = (int)((units + 3) / 4); P.ncg = d->N / 32;
  const int64_t all = (int64_t)P.ncg * P.nrg;
  P.passes = all > 256 && all <= 1024 ? (int)((all + 255) / 256) : 1;
  const int64_t items = (int64_t)P.ncg * ((P.nrg + P.passes - 1) / P.passes);
  const dim3 grid((unsigned)((items + 7) / 8 * 8));
  if (T == 10) SDF_LAUNCH((wide_pm_kernel<10, 2, 2, 0, 2>), grid, dim3(256), 0, s, P);
  else SDF_LAUNCH((wide_pm_kernel<20, 2, 2, 0, 2>), grid, dim3(256), 0, s, P);
  hipError_t e = hipGetLastError();
  return e != hipSuccess ? (int)e : 0;
}

bool ms_wide_attn_supports(const SdfQkAttnDesc* d) {
  if (wide_env_off() || (d->flags & SDF_QK_NARROW)) return false;
  const bool res = res_stage_ok(d->C) && d->C <= 192;
  if (d->nsplit != 2 || d->Tq != 2 || (!res && (d->C < 192 || d->C % 64)) || d->C != d->nH * 32 || d->N1 < 24) return false;
  if (!d->x_src || d->xB < 1 || d->xHW < 1 || (d->xD != 10 && d->xD != 20)) return false;
  if ((int64_t)d->xB * d->xD * d->xHW != d->x_rows) return false;
  const SdfNeuronCfg* ns[4] = {&d->sn_proj, &d->sn_q, &d->sn_k, &d->sn2_q};
  const int nk = neuron_class(*ns[0]);
  for (const SdfNeuronCfg* n : ns)
    if (!neuron_ok(*n) || neuron_class(*n) != nk) return false;
  if (d->emit_s1 && !neuron_ok(d->emit_sn)) return false;
  const int64_t M = d->B_ * d->N1 * d->Tq;
  if (M * d->C >= (1LL << 31) || d->x_rows * d->C * 4 >= (1LL << 31) || M >= (1LL << 31)) return false;
  if (d->x_rows > WIDE_MAX_ROWS && !wide_env_any() && !res) return false;
  const bool fused = d->qk_planes != nullptr;
  if (fused ? (!d->qk_digits || !d->qk_cscale) : (!d->q_digits || !d->q_cscale || !d->k_digits || !d->k_cscale)) return false;
  if (!d->p_digits || !d->p_cscale) return false;
  const float* al[4] = {fused ? d->qk_alpha : d->q_alpha, fused ? d->qk_beta : d->q_beta, fused ? d->qk_alpha : d->k_alpha,
                        fused ? d->qk_beta : d->k_beta};
  if ((al[0] == nullptr) != (al[1] == nullptr) || (al[2] == nullptr) != (al[3] == nullptr)) return false;
  if (d->p_alpha && !d->p_beta) return false;
  return sdf_aligned(d->x, 16) && sdf_aligned(d->p_digits, 16) && sdf_aligned(fused ? (const void*)d->qk_digits : (const void*)d->q_digits, 16) &&
         (fused || sdf_aligned(d->k_digits, 16));
}

// xs (2, rows, C) = SN_proj of the gathered slices -> E (2, rows, C) [+ q | k tape] ; then x += proj(E) (+ emit)
int launch_ms_wide_front(const SdfQkAttnDesc* d, const uint8_t* xs, uint8_t* e, uint8_t* qk, bool keep, hipStream_t s) {
  WideFrontParams P = {};
  const int C = d->C;
  const int64_t rows = d->B_ * d->N1, M = rows * d->Tq;
  P.xs = xs; P.rows = rows; P.N1 = d->N1; P.C = C; P.nH = d->nH;
  if (d->qk_planes) {
    P.wq = d->qk_digits; P.wk = d->qk_digits + (int64_t)C * C; P.wq_plane = P.wk_plane = 2LL * C * C;
    P.q_cs = d->qk_cscale; P.k_cs = d->qk_cscale + C;
    P.q_al = d->qk_alpha; P.q_be = d->qk_beta;
    P.k_al = d->qk_alpha ? d->qk_alpha + C : nullptr; P.k_be = d->qk_beta ? d->qk_beta + C : nullptr;
    P.pe = d->qk_add ? d->qk_add + C : nullptr; P.pe_ld = 2LL * C;
    P.qs = qk; P.ks = qk + C; P.ldq = P.ldk = 2LL * C;
  } else {
    P.wq = d->q_digits; P.wk = d->k_digits; P.wq_plane = P.wk_plane = (int64_t)C * C;
    P.q_cs = d->q_cscale; P.k_cs = d->k_cscale;
    P.q_al = d->q_alpha; P.q_be = d->q_beta; P.k_al = d->k_alpha; P.k_be = d->k_beta;
    P.pe = d->k_add; P.pe_ld = C;
    P.qs = qk; P.ks = qk + M * C; P.ldq = P.ldk = C;
  }
  P.sn_q = d->sn_q; P.sn_k = d->sn_k; P.sn2_q = d->sn2_q;
  P.it_q = inv_tau_of(d->sn_q); P.it_k = inv_tau_of(d->sn_k); P.it_2 = inv_tau_of(d->sn2_q);
  P.e = e;
  if (res_stage_ok(C) && res_front_takes(P)) return launch_res_front(P, keep, neuron_class(d->sn_q), s);      // (ms_res.hip)
  // token tiles of 8 RB tokens: 32 (RB = 4: the accumulators leave room for the early positional-term loads) while that leaves
  // >= 600 waves, else 16 (RB = 2)
  const int64_t t5 = (rows + 31) / 32, t2 = (rows + 15) / 16;
  const bool big = t5 * d->nH >= 800;
  const int64_t ntiles = big ? t5 : t2;
  P.ntiles = (int)ntiles;
  P.nrg = (int)((ntiles + 3) / 4);
  const int64_t items = (int64_t)P.nrg * d->nH;
  if (items >= (1LL << 31) - 8) return SDF_E_SHAPE;
  const dim3 grid((unsigned)((items + 7) / 8 * 8));
  const int nk = neuron_class(d->sn_q);
#define SDF_WF(RB_, NK_)                                                                                  \
  do {                                                                                                    \
    if (keep) SDF_LAUNCH((wide_front_kernel<RB_, NK_, true>), grid, dim3(256), 0, s, P);          \
    else SDF_LAUNCH((wide_front_kernel<RB_, NK_, false>), grid, dim3(256), 0, s, P);              \
  } while (0)
  if (big) { if (nk == 0) SDF_WF(4, 0); else if (nk == 1) SDF_WF(4, 1); else SDF_WF(4, 2); }
  else { if (nk == 0) SDF_WF(2, 0); else if (nk == 1) SDF_WF(2, 1); else SDF_WF(2, 2); }
#undef SDF_WF
  hipError_t err = hipGetLastError();
  return err != hipSuccess ? (int)err : 0;
}

int launch_ms_wide_proj(const SdfQkAttnDesc* d, const uint8_t* e, hipStream_t s) {
  WidePmParams P = {};
  P.A = e; P.zsrc = d->x_src; P.zg_G = (uint32_t)d->Tq * (uint32_t)d->N1 * 32u;
  P.W = d->p_digits; P.cscale = d->p_cscale; P.N = d->C; P.K = d->C; P.HW = (int)d->xHW; P.P = (int64_t)d->xB * d->xHW;
  P.bias = d->p_bias; P.alpha = d->p_alpha; P.beta = d->p_beta;
  P.x = d->x; P.ldo = d->C;
  P.out_spike = d->emit_s1; P.ldsp = d->C; P.out_tiled = (d->flags & SDF_QK_KEEP_SPIKES) ? 0 : 1;
  P.sn = d->emit_s1 ? d->emit_sn : d->sn_proj; P.inv_tau = inv_tau_of(P.sn);
  P.res_stage = res_stage_ok(d->C);
  return launch_pm(P, d->xD, d->emit_s1 ? 3 : 2, s);
}

// ---- 3x3 convolution of few rows against many weights (the U-Net bottleneck's res-blocks: 1 080 rows x 768 x 6 912) -----------------
// K is split over workgroups so that the weight digits (16 MB) leave HBM once, spread over the whole chip; the partial sums meet in
// wide_reduce_kernel together with BN, the shortcut and the neuron.
bool wide_conv_supports(const GemmParams& P) {
  const SdfSpikeGemmDesc& d = P.d;
  const ConvGeom& cv = P.cv;
  if (wide_env_off()) return false;
  // Measured on MI355X (round 4, tools/stamp_wide.sh conv): 50 us + 20 us for the reduce pass on the bottleneck's 1 080 x 768 x 6 912
  // problem, against 50 + 5 us of the streaming ping-pong kernel with its split-K - no gain (four row passes per workgroup each pay
  // the pipeline fill; the reduce pass is latency-bound on 81 workgroups), so the engine does not take it: SDF_WIDE_CONV=1 opts in
  // (tests, A/B).
  {
    const char* e = sdf_sw(SW_WIDE_CONV);
    if (!(e && e[0] == '1')) return false;
  }
  if (d.nsplit != SDF_PLANES_I8X3 || !d.col_scale) return false;
  if (cv.KWc != 3 || d.K != 9 * cv.Cin || cv.Cin % KCH || cv.sy != 1 || cv.sx != 1 || cv.OH != cv.H || cv.OW != cv.W) return false;
  if (cv.dy[0] != -1 || cv.dy[1] != 0 || cv.dy[2] != 1 || cv.dx[0] != -1 || cv.dx[1] != 0 || cv.dx[2] != 1) return false;
  if (d.N % 32 || d.out_rowmap || d.bias || d.add || d.zg_nH) return false;
  const int64_t hw = (int64_t)cv.H * cv.W, imgs = d.M / hw;
  int T = d.sn_T;
  if (T == 0) T = imgs % 10 == 0 ? 10 : (imgs % 20 == 0 ? 20 : 0);
  if (T != 10 && T != 20) return false;
  if (imgs % T || d.M > WIDE_MAX_ROWS) return false;
  if (d.sn_T > 0) {
    if (!neuron_ok({d.sn_kind, d.tau, d.v_th, d.v_reset, d.soft_reset, nullptr, nullptr})) return false;
    if (d.pos_inner != hw || d.t_stride != hw || d.pos_ostride != (int64_t)T * hw || d.pos_count * T != d.M) return false;   // rows (b, t, pixel)
  }
  if (d.M * (int64_t)cv.Cin >= (1LL << 31) || d.M * (int64_t)d.N * 4 >= (1LL << 31) || (int64_t)d.N * d.K * 3 >= (1LL << 31)) return false;
  // split-K plan: >= 200 workgroups of whole chunks
  const int nchunks = d.K / KCH, ncg = d.N / 32;
  int ks = 1;
  while (ks < nchunks && (ncg * ks < 200 || nchunks % ks)) ++ks;
  if (nchunks % ks) return false;
  if (!d.workspace || d.workspace_bytes < (int64_t)ks * d.M * d.N * 4) return false;
  return sdf_aligned(d.A, 16) && sdf_aligned(d.Wp, 16) && sdf_aligned(d.workspace, 16) && (!d.out || sdf_aligned(d.out, 16)) &&
         (!d.resid || sdf_aligned(d.resid, 16)) && (!d.alpha || (sdf_aligned(d.alpha, 16) && sdf_aligned(d.beta, 16))) &&
         (!d.out_spike || sdf_aligned(d.out_spike, 4)) && d.ldo == d.N;
}

int launch_wide_conv(const GemmParams& G, hipStream_t s) {
  const SdfSpikeGemmDesc& d = G.d;
  const ConvGeom& cv = G.cv;
  const int64_t hw = (int64_t)cv.H * cv.W, imgs = d.M / hw;
  int T = d.sn_T;
  if (T == 0) T = imgs % 10 == 0 ? 10 : 20;
  const int nchunks = d.K / KCH, ncg = d.N / 32;
  int ks = 1;
  while (ks < nchunks && (ncg * ks < 200 || nchunks % ks)) ++ks;
  WidePmParams P = {};
  P.A = d.A; P.W = reinterpret_cast<const int8_t*>(d.Wp); P.cscale = d.col_scale; P.N = d.N; P.K = d.K; P.HW = (int)hw; P.P = (imgs / T) * hw;
  P.cv_H = cv.H; P.cv_W = cv.W; P.cv_Cin = cv.Cin; P.cv_cpt = cv.Cin / KCH;
  P.ksplit = ks;                                            // (one range: the same path, slot 0 of the partial buffer)
  P.cps = nchunks / ks;
  P.partial = reinterpret_cast<float*>(d.workspace);
  const int PPW = 4 * (20 / T);
  const int64_t units = (P.P + PPW - 1) / PPW;
  P.nunits = (int)units; P.nrg = (int)((units + 3) / 4); P.ncg = ncg;
  P.passes = P.nrg;                                         // one workgroup per (column group, K range) walks every row group
  const int64_t items = (int64_t)ncg * ks;
  const dim3 grid((unsigned)((items + 7) / 8 * 8));
  if (T == 10) SDF_LAUNCH((wide_pm_kernel<10, 2, 4, 0>), grid, dim3(256), 0, s, P);
  else SDF_LAUNCH((wide_pm_kernel<20, 2, 4, 0>), grid, dim3(256), 0, s, P);
  WideReduceParams R = {};
  R.partial = P.partial; R.ksplit = ks; R.N = d.N; R.HW = (int)hw; R.P = P.P;
  R.alpha = d.alpha; R.beta = d.beta; R.resid = d.resid; R.out = d.out; R.out_spike = d.sn_T > 0 ? d.out_spike : nullptr;
  R.sn = {d.sn_kind, d.tau, d.v_th, d.v_reset, d.soft_reset, nullptr, nullptr};
  R.inv_tau = d.sn_T > 0 ? inv_tau_of(R.sn) : 0.f;
  const int64_t nthr = R.P * (d.N / 4);
  const dim3 rgrid((unsigned)((nthr + 255) / 256));
  const int nk = d.sn_T > 0 ? neuron_class(R.sn) : 0;
  if (T == 10) { if (nk == 0) SDF_LAUNCH((wide_reduce_kernel<10, 0>), rgrid, dim3(256), 0, s, R); else SDF_LAUNCH((wide_reduce_kernel<10, 2>), rgrid, dim3(256), 0, s, R); }
  else { if (nk == 0) SDF_LAUNCH((wide_reduce_kernel<20, 0>), rgrid, dim3(256), 0, s, R); else SDF_LAUNCH((wide_reduce_kernel<20, 2>), rgrid, dim3(256), 0, s, R); }
  hipError_t e = hipGetLastError();
  return e != hipSuccess ? (int)e : 0;
}

}  // namespace sdfmm

#ifdef SDF_STAMP
extern "C" int sdf_debug_read_stamps_wide(unsigned long long* host32, unsigned long long* census) {
  (void)hipMemcpyFromSymbol(census, HIP_SYMBOL(g_wide_census), sizeof(g_wide_census));
  (void)hipMemcpyFromSymbol(host32, HIP_SYMBOL(g_wide_stamp), sizeof(g_wide_stamp));
  return (int)hipMemcpyFromSymbol(host32 + 40, HIP_SYMBOL(g_wide_loop), sizeof(g_wide_loop));
}
#endif

extern "C" int sdf_ms_patch_merge_fwd(const SdfMsMergeDesc* d, void* stream) {
  if (!d) return SDF_E_NULL;
  if (!d->spikes || !d->digits || !d->cscale || !d->out) return SDF_E_NULL;
  if (d->alpha && !d->beta) return SDF_E_NULL;
  if (!sdfmm::wide_merge_supports(d)) return SDF_E_SHAPE;
  return sdfmm::launch_wide_merge(d, sdf_stream(stream));
}

extern "C" int sdf_window_zsrc_map(const int32_t* slice_map, int64_t B_, int Tq, int N1, int nH, int32_t* x_src, void* stream) {
  if (!slice_map || !x_src) return SDF_E_NULL;
  if (B_ < 1 || Tq < 1 || N1 < 1 || nH < 1) return SDF_E_SHAPE;
  const int64_t total = B_ * Tq * N1;
  if (total * nH * 32 >= (1LL << 31)) return SDF_E_SHAPE;
  SDF_LAUNCH(sdfmm::zsrc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, sdf_stream(stream), slice_map, x_src, B_, Tq,
                     N1, nH, total);
  SDF_LAUNCH_CHECK();
  return 0;
}
