// Wide stages (C >= 288: swin stages 2 and 3 of the shipped model) of the MS swin block for gfx950 - rows a5 / a6 / a7 of
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
// Main loop (both kernels): v_mfma_f32_16x16x32 (f16 / bf16 planes), spikes are the ROW operand.  The wave's A operand never
// touches LDS: a lane loads 16 bytes (the k-pieces 2j, 2j + 1 of a 64-deep K pair, j = lane / 16) of its row straight into
// registers one pair ahead (raw buffer loads, invalid rows read zeros), and expands 8 bytes to 8 halves per MFMA step.  The
// workgroup's NW waves share BN = 16 CB weight columns: 64-deep K chunks of both planes go global -> registers -> LDS one chunk
// ahead into a two-buffer ring, ONE barrier per chunk.  LDS weight layout [plane][k-piece][column ^ (k-piece & 7)] x 16 B: the
// eight 16-byte pieces of a 128-byte weight row are written by eight neighbouring lanes to eight different bank quads, and every
// fragment read (lane = column + 16 k-group) is conflict-free (tools/probes: /tmp bank model in DESIGN.md section 5).
// Workgroups that share a column group are neighbours on one XCD (its L2 serves the weight re-reads).
//
// Epilogues: BN (+ bias, + positional term) on the accumulators; LIF / IF over T per lane (neuron_T of spike_mm.h - the separately
// rounded op sequence of neuron.hip; compiled with -ffp-contract=off); spike bits -> bytes by one 24-bit multiply per 4 rows, a
// 4 x 4 byte transpose inside each lane quad (two DPP moves + two v_perm) so that a lane holds four consecutive CHANNELS of a row,
// one ds_write_b32 into a per-wave LDS tile and 16-byte global stores.  The fp32 shortcut is read and written in the accumulator
// layout (64-byte runs per row and column block; the buffers are L2-resident at these sizes).
#include "spike_mm.h"
#include <stdlib.h>

namespace sdfmm {
namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
constexpr uint32_t INV = 0x80000000u;               // buffer offset of "no such row": loads return zeros, stores are dropped

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)INV, 0x00020000);
}

// 8 spike bytes {0, 1} -> 8 x 16-bit {0, 1.0} (fp16 for two planes, bf16 otherwise); the 24-bit multiply is a full-rate op
template <int NSPLIT>
__device__ __forceinline__ bf16x8 expand01(uint32_t lo, uint32_t hi) {
  constexpr uint32_t ONE = NSPLIT == 2 ? 0x3C00u : 0x3F80u;
  union { bf16x8 h; uint32_t u[4]; } r;
  r.u[0] = __umul24(__builtin_amdgcn_perm(0u, lo, 0x0c010c00u), ONE);
  r.u[1] = __umul24(__builtin_amdgcn_perm(0u, lo, 0x0c030c02u), ONE);
  r.u[2] = __umul24(__builtin_amdgcn_perm(0u, hi, 0x0c010c00u), ONE);
  r.u[3] = __umul24(__builtin_amdgcn_perm(0u, hi, 0x0c030c02u), ONE);
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

// 4 bits -> 4 bytes {0, 1}: bit i lands in byte i (i + 7 i = 8 i; the cross terms i + 7 k, k != i, miss every byte's bit 0)
__device__ __forceinline__ uint32_t spread4(uint32_t nib) { return __umul24(nib & 0xFu, 0x204081u) & 0x01010101u; }

// 4 x 4 byte transpose inside every quad of lanes: in: lane i of the quad holds bytes (rows 0..3) of column i; out: lane i holds
// bytes (columns 0..3) of row i.  sel1 / sel2 are the lane's v_perm selectors (quad_sel).
__device__ __forceinline__ void quad_sel(int lane, uint32_t& sel1, uint32_t& sel2) {
  sel1 = (lane & 1) ? 0x03070105u : 0x06020400u;
  sel2 = (lane & 2) ? 0x03020706u : 0x05040100u;
}
__device__ __forceinline__ uint32_t quad_tr_bytes(uint32_t w, uint32_t sel1, uint32_t sel2) {
  const uint32_t t1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0xB1, 0xF, 0xF, false);     // lane ^ 1
  const uint32_t a = __builtin_amdgcn_perm(t1, w, sel1);
  const uint32_t t2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, 0x4E, 0xF, 0xF, false);     // lane ^ 2
  return __builtin_amdgcn_perm(t2, a, sel2);
}

// sum over the 16 lanes of a DPP row (all lanes end with the total)
__device__ __forceinline__ uint32_t row_sum16(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);                    // quad_perm [1,0,3,2]
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);                    // quad_perm [2,3,0,1]
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false);                   // row_half_mirror
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false);                   // row_mirror
  return v;
}

constexpr int KCH = 64;                              // K chunk = one "pair" of MFMA steps
constexpr int RBW = 5;                               // row blocks of a wave in the position-major kernel: 80 rows = 20 slots per lane

__host__ __device__ constexpr int s_pitch(int bytes) { return ((bytes + 16) / 4) % 8 == 4 ? bytes + 16 : bytes + 32; }

// ---------------------------------------------------------------------------------------------------------------------------
// The shared main loop.  acc[RB][CB] += A[rows of this wave][K] x W[BN columns][K]^T over K = 64 npairs, npairs EVEN.
//   a_base[rb] : byte offset of this lane's first 16-byte piece of row block rb's row (lane % 16) in the A buffer, or INV
//   a_step     : bytes between two K pairs of a row (64 for plain rows; 2 G through the head scramble)
//   w_goff[i]  : byte offset of this thread's i-th weight piece of chunk 0 in its buffer (or INV); chunk c adds 128 c
//   w_lds[i]   : where that piece goes inside a ring buffer
// One barrier per chunk; chunk c + 1 is committed to the other LDS buffer at the top of chunk c's MFMAs and chunk c + 2 requested.
// The loop body is straight-line (requests beyond the last chunk read zeros through INV offsets, no branches): the compiler's
// vmcnt bookkeeping stays exact, so a wait for the pair about to be multiplied leaves the next pair's loads in flight.
template <int WIT>
struct WPieces {
  uint32_t goff[WIT], lds[WIT];
  int which[WIT];                                     // buffer resource of piece i (0 / 1), compile-time after unrolling
};

template <int NSPLIT, int RB, int CB, int WIT>
__device__ __forceinline__ void wide_mainloop(f32x4 (&acc)[RB][CB], const __amdgpu_buffer_rsrc_t A_rs, const uint32_t (&a_base)[RB],
                                              uint32_t a_step, int npairs, bool active, uint8_t* Wlds, const __amdgpu_buffer_rsrc_t W0_rs,
                                              const __amdgpu_buffer_rsrc_t W1_rs, const uint32_t (&w_goff)[WIT], const uint32_t (&w_lds)[WIT],
                                              int lane) {
  constexpr int BN = 16 * CB, WBUF = NSPLIT * BN * 8 * 16;
  u32x4 wreg[WIT];
  auto wreq = [&](int ch) __attribute__((always_inline)) {
    const bool in = ch < npairs;
#pragma unroll
    for (int i = 0; i < WIT; ++i)
      wreg[i] = __builtin_amdgcn_raw_buffer_load_b128((i & 1) ? W1_rs : W0_rs, in ? w_goff[i] : INV, (uint32_t)ch * (KCH * 2), 0);
  };
  auto w_commit = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < WIT; ++i) *reinterpret_cast<u32x4*>(Wlds + buf * WBUF + w_lds[i]) = wreg[i];
  };
  const int l16 = lane & 15, lj = lane >> 4;
  // this lane's fragment of (column block cb, plane p, step h): piece kp = 2 lj + h of the chunk
  auto compute_pair = [&](const u32x4 (&a)[RB], int buf) __attribute__((always_inline)) {
    const uint8_t* wb = Wlds + buf * WBUF;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      bf16x8 ax[RB];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) ax[rb] = h == 0 ? expand01<NSPLIT>(a[rb][0], a[rb][1]) : expand01<NSPLIT>(a[rb][2], a[rb][3]);
      const int kp = 2 * lj + h;
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
#pragma unroll
        for (int p = 0; p < NSPLIT; ++p) {
          const bf16x8 b = *reinterpret_cast<const bf16x8*>(wb + ((p * 8 + kp) * BN + ((cb * 16 + l16) ^ kp)) * 16);
#pragma unroll
          for (int rb = 0; rb < RB; ++rb) acc[rb][cb] = mma16<NSPLIT>(ax[rb], b, acc[rb][cb]);
        }
      }
    }
  };
  auto a_load = [&](u32x4 (&a)[RB], int s) __attribute__((always_inline)) {
    const bool in = s < npairs;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) a[rb] = __builtin_amdgcn_raw_buffer_load_b128(A_rs, in ? a_base[rb] : INV, (uint32_t)s * a_step, 0);
  };
  u32x4 aA[RB], aB[RB];
  wreq(0);
  a_load(aA, 0);
  w_commit(0);
  wreq(1);
  __syncthreads();
#pragma unroll 1
  for (int s = 0; s < npairs; s += 2) {
    w_commit(1);
    wreq(s + 2);
    a_load(aB, s + 1);
    if (active) compute_pair(aA, 0);
    __syncthreads();
    w_commit(0);
    wreq(s + 3);
    a_load(aA, s + 2);
    if (active) compute_pair(aB, 1);
    __syncthreads();
  }
}

// this thread's weight pieces of a 64-deep chunk: piece = (plane p, column col of the group, k-piece kp); 256 threads take 32 columns
// x 8 k-pieces per step.  row_of(p, col) -> element offset of (plane p, column col, k = 0) in the weight buffer
template <int NSPLIT, int CB, class RowOf>
__device__ __forceinline__ void wide_pieces(uint32_t (&goff)[NSPLIT * CB / 2], uint32_t (&lds)[NSPLIT * CB / 2], int tid, RowOf row_of) {
  constexpr int BN = 16 * CB, WIT = NSPLIT * CB / 2;
  const int kp = tid & 7, r = tid >> 3;
#pragma unroll
  for (int i = 0; i < WIT; ++i) {
    constexpr int dummy = 0;
    (void)dummy;
    const int p0 = (32 * i) / BN, rem0 = (32 * i) % BN;
    const bool wrap = rem0 + r >= BN;
    const int p = p0 + (wrap ? 1 : 0), col = rem0 + r - (wrap ? BN : 0);
    goff[i] = row_of(p, col, i) + 16u * kp;
    lds[i] = (uint32_t)(((p * 8 + kp) * BN + (col ^ kp)) * 16);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Position-major GEMM.
struct WidePmParams {
  const uint8_t* A;          // plain: u8 [rows][K];  head scramble: the gated spikes E, flat
  const int32_t* zsrc;       // head scramble: per activation row the byte offset of (k-group 0, byte 0) in E; null = plain rows
  uint32_t zg_G;             // head scramble: bytes between two k-groups (T' * N1 * 32)
  const uint16_t* W;         // [NSPLIT][N][K]
  int N, K, HW;
  int64_t P;                 // positions = B * HW; rows = P * T in (B, T, HW) order
  float asc;
  const float *bias, *alpha, *beta;
  float* x;                  // fp32 epilogue: out = resid, row stride ldo
  int ldo;
  uint8_t* out_spike;        // neuron epilogue: u8 [rows][ldsp]
  int ldsp;
  SdfNeuronCfg sn;
  float inv_tau;
  int ncg, nrg, nunits;
};

// EPI: 1 = neuron (spikes out), 2 = fp32 (+ shortcut), 3 = fp32 and the neuron on the updated shortcut stream
template <int NSPLIT, int T, int CB, int NW, int EPI, int NK>
__global__ __launch_bounds__(64 * NW, (CB >= 6 ? 1 : 2)) void wide_pm_kernel(WidePmParams P) {
  constexpr int RB = RBW, ROWS = 16 * RB, SLOTS = 4 * RB, PPG = SLOTS / T, PPW = 4 * PPG;
  constexpr int BN = 16 * CB, NT = 64 * NW, PIECES = NSPLIT * BN * 8, WBUF = PIECES * 16;
  constexpr int SP = s_pitch(BN), STILE = ROWS * SP;
  constexpr int LDSB = 2 * WBUF > NW * STILE ? 2 * WBUF : NW * STILE;
  static_assert(SLOTS % T == 0, "T must divide the 20 accumulator slots of a lane");
  static_assert(!(EPI & 2) || CB <= 3, "the fp32 epilogue keeps every shortcut load in flight: three column blocks at most");
  __shared__ __attribute__((aligned(16))) uint8_t smem[LDSB];
  __shared__ int32_t rowtab[NW * ROWS];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, lq = lane >> 4;
  // (column group, row group): workgroups of one column group are neighbours on one XCD
  int item = blockIdx.x;
  const int G = gridDim.x;
  if ((G & 7) == 0) item = (item & 7) * (G >> 3) + (item >> 3);
  if (item >= P.ncg * P.nrg) return;
  const int cg = item / P.nrg, rg = item - cg * P.nrg;
  const int n0 = cg * BN;
  const int unit = rg * NW + wave;
  const bool active = unit < P.nunits;
  const int K = P.K, N = P.N, HW = P.HW;

  // activation row (or -1) of every tile row of this wave
  for (int r = lane; r < ROWS; r += 64) {
    const int rb = r >> 4, i = r & 15, q = i >> 2, slot = 4 * rb + (i & 3);
    const int pp = slot / T, t = slot - pp * T;
    const int64_t pos = (int64_t)unit * PPW + q * PPG + pp;
    int32_t g = -1;
    if (active && pos < P.P) {
      const int64_t b = pos / HW, hw = pos - b * HW;
      g = (int32_t)((b * T + t) * HW + hw);
    }
    rowtab[wave * ROWS + r] = g;
  }
  asm volatile("" ::: "memory");                          // (same-wave LDS operations execute in order: no wait between the table's writes and reads)
  const __amdgpu_buffer_rsrc_t A_rs = make_rsrc(P.A), W_rs = make_rsrc(P.W);
  uint32_t a_base[RB];
  uint32_t a_step = KCH;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int32_t g = rowtab[wave * ROWS + 16 * rb + l16];
    a_base[rb] = INV;
    if (g >= 0) a_base[rb] = P.zsrc ? (uint32_t)P.zsrc[g] + (uint32_t)(lq >> 1) * P.zg_G + 16u * (lq & 1) : (uint32_t)g * (uint32_t)K + 16u * lq;
  }
  if (P.zsrc) a_step = 2 * P.zg_G;

  f32x4 acc[RB][CB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) acc[rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int WIT = PIECES / NT;
  static_assert(PIECES % NT == 0 && NT == 256, "weight pieces: 256 threads x whole steps");
  uint32_t w_goff[WIT], w_lds[WIT];
  wide_pieces<NSPLIT, CB>(w_goff, w_lds, tid, [&](int p, int col, int) -> uint32_t {
    return n0 + col < N ? (uint32_t)((p * N + n0 + col) * K) * 2u : INV;
  });
  wide_mainloop<NSPLIT, RB, CB, WIT>(acc, A_rs, a_base, a_step, K / KCH, active, smem, W_rs, W_rs, w_goff, w_lds, lane);
  if (!active) return;                                    // (every barrier is behind this wave)

  // ---------------- epilogue ----------------
  int lnl = lane;
  asm volatile("" : "+v"(lnl));                           // row offsets are computed here, not hoisted above the main loop
  const int c = lnl & 15, q = lnl >> 4;
  float al[CB], be[CB], bs[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    const int n = n0 + 16 * cb + c;
    const int nc = n < N ? n : 0;
    al[cb] = P.alpha ? P.alpha[nc] : 1.f;
    be[cb] = P.alpha ? P.beta[nc] : 0.f;
    bs[cb] = P.bias ? P.bias[nc] : 0.f;
  }
  if constexpr ((EPI & 2) != 0) {
    const __amdgpu_buffer_rsrc_t x_rs = make_rsrc(P.x);
    uint32_t xo[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int32_t g = rowtab[wave * ROWS + 16 * (s >> 2) + 4 * q + (s & 3)];
      xo[s] = (g >= 0 && n0 + c < N) ? ((uint32_t)g * (uint32_t)P.ldo + (uint32_t)(n0 + c)) * 4u : INV;
    }
    float res[CB][SLOTS];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int s = 0; s < SLOTS; ++s)
        res[cb][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rs, (n0 + 16 * cb + c < N) ? xo[s] : INV, 64u * cb, 0));
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int s = 0; s < SLOTS; ++s) {
        float v = acc[s >> 2][cb][s & 3] * P.asc;
        v = v + bs[cb];
        v = __builtin_fmaf(v, al[cb], be[cb]);
        v = v + res[cb][s];
        acc[s >> 2][cb][s & 3] = v;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), x_rs, (n0 + 16 * cb + c < N) ? xo[s] : INV, 64u * cb, 0);
      }
  }
  if constexpr ((EPI & 1) != 0) {
    uint8_t* S = smem + wave * STILE;                    // per-wave byte tile [80][SP]; aliases the weight ring (the main loop ends with a barrier)
    uint32_t sel1, sel2;
    quad_sel(lnl, sel1, sel2);
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
          if constexpr (EPI == 1) xs[t] = __builtin_fmaf(acc[s >> 2][cb][s & 3] * P.asc + bs[cb], al[cb], be[cb]);
          else xs[t] = acc[s >> 2][cb][s & 3];
        }
        neuron_T<NK, T>(xs, sp, P.sn, P.inv_tau);
#pragma unroll
        for (int t = 0; t < T; ++t) bits |= ((__float_as_uint(sp[t]) >> 29) & 1u) << (pp * T + t);      // 1.0f has bit 29 set
      }
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        const uint32_t w = quad_tr_bytes(spread4(bits >> (4 * rb)), sel1, sel2);
        *reinterpret_cast<uint32_t*>(S + (16 * rb + 4 * q + ci) * SP + 16 * cb + 4 * m4) = w;
      }
    }
    const __amdgpu_buffer_rsrc_t o_rs = make_rsrc(P.out_spike);
#pragma unroll
    for (int it = 0; it < (ROWS * CB + 63) / 64; ++it) {
      const int pc = lnl + 64 * it;
      if (pc < ROWS * CB) {
        const int r = pc / CB, k16 = pc - r * CB;
        const int32_t g = rowtab[wave * ROWS + r];
        const u32x4 v = *reinterpret_cast<const u32x4*>(S + r * SP + 16 * k16);
        __builtin_amdgcn_raw_buffer_store_b128(v, o_rs, (g >= 0 && n0 + 16 * k16 < N) ? (uint32_t)g * (uint32_t)P.ldsp + (uint32_t)(n0 + 16 * k16) : INV, 0, 0);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Attention front: q | k projection + BN (+ positional term) + SN_q / SN_k over the T' = 2 steps + token gate -> E.
struct WideFrontParams {
  const uint8_t* xs;         // (2, rows, C) u8: SN_proj of the gathered slices
  int64_t rows;              // B_ * N1
  int N1, C, nH;
  const uint16_t* wq; const uint16_t* wk;  // planes, row pitch C; plane strides (elements)
  int64_t wq_plane, wk_plane;
  const float *q_al, *q_be, *k_al, *k_be;
  const float* pe; int64_t pe_ld;          // k's additive term pe[(t * N1 + n) * pe_ld + c] or null
  float q_asc, k_asc;
  SdfNeuronCfg sn_q, sn_k, sn2_q;
  float it_q, it_k, it_2;
  uint8_t* e;                // (2, rows, C)
  uint8_t* qs; uint8_t* ks;  // KEEP: q / k spikes with row strides ldq / ldk
  int64_t ldq, ldk;
  int nrg, ntiles;           // row groups (NW tiles each), token tiles
};

template <int NSPLIT, int RB, int NK, bool KEEP>
__global__ __launch_bounds__(256, 2) void wide_front_kernel(WideFrontParams P) {
  constexpr int NW = 4, CB = 4, BN = 64, NT = 256, PIECES = NSPLIT * BN * 8, WBUF = PIECES * 16, WIT = PIECES / NT;
  static_assert(NSPLIT == 2, "piece step i = 2 p + (q | k) needs 64 columns = two steps per plane");
  constexpr int ROWS = 16 * RB, SB = KEEP ? 96 : 32, SP = s_pitch(SB), STILE = ROWS * SP;
  constexpr int LDSB = 2 * WBUF > NW * STILE ? 2 * WBUF : NW * STILE;
  static_assert(PIECES % NT == 0, "weight pieces must divide over the threads");
  __shared__ __attribute__((aligned(16))) uint8_t smem[LDSB];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, lq = lane >> 4;
  int item = blockIdx.x;
  const int G = gridDim.x;
  if ((G & 7) == 0) item = (item & 7) * (G >> 3) + (item >> 3);
  if (item >= P.nH * P.nrg) return;
  const int hd = item / P.nrg, rg = item - hd * P.nrg;
  const int tile = rg * NW + wave;
  const bool active = tile < P.ntiles;
  const int C = P.C;
  const int64_t tok0 = (int64_t)tile * (8 * RB);

  const __amdgpu_buffer_rsrc_t A_rs = make_rsrc(P.xs), Wq_rs = make_rsrc(P.wq), Wk_rs = make_rsrc(P.wk);
  uint32_t a_base[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int64_t tk = tok0 + 8 * rb + (l16 >> 1);
    a_base[rb] = (active && tk < P.rows) ? (uint32_t)(((int64_t)(l16 & 1) * P.rows + tk) * C) + 16u * lq : INV;
  }
  f32x4 acc[RB][CB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) acc[rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
  // weight pieces: step i = 2 p + (0: the head's 32 q rows, 1: its 32 k rows) - the buffer of a piece is a compile-time choice
  uint32_t w_goff[WIT], w_lds[WIT];
  wide_pieces<NSPLIT, CB>(w_goff, w_lds, tid, [&](int p, int col, int i) -> uint32_t {
    return (uint32_t)((p * ((i & 1) ? P.wk_plane : P.wq_plane) + (int64_t)(hd * 32 + (col & 31)) * C) * 2);
  });
  wide_mainloop<NSPLIT, RB, CB, WIT>(acc, A_rs, a_base, (uint32_t)KCH, C / KCH, active, smem, Wq_rs, Wk_rs, w_goff, w_lds, lane);
  if (!active) return;                                    // (the main loop ends with a barrier: the per-wave byte tiles may alias the weight ring)

  // ---------------- epilogue ----------------
  int lnl = lane;
  asm volatile("" : "+v"(lnl));
  const int c = lnl & 15, q = lnl >> 4;
  float qa[2], qb[2], ka[2], kb[2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const int ch = hd * 32 + 16 * cb + c;
    qa[cb] = P.q_al ? P.q_al[ch] : 1.f; qb[cb] = P.q_al ? P.q_be[ch] : 0.f;
    ka[cb] = P.k_al ? P.k_al[ch] : 1.f; kb[cb] = P.k_al ? P.k_be[ch] : 0.f;
  }
  // positional term of k: pe[(t * N1 + n) * pe_ld + channel], n = token % N1
  const int nbase = (int)(tok0 % P.N1);
  float pev[RB][2][2][2];                                 // [rb][m][t][cb]
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
          xq[t] = __builtin_fmaf(acc[rb][cb][2 * m + t] * P.q_asc, qa[cb], qb[cb]);
          xk[t] = __builtin_fmaf(acc[rb][2 + cb][2 * m + t] * P.k_asc, ka[cb], kb[cb]);
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

bool neuron_ok(const SdfNeuronCfg& n) {
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

template <int NSPLIT, int T, int CB, int EPI>
int launch_pm_nk(const WidePmParams& P, int nk, dim3 grid, hipStream_t s) {
  if constexpr (EPI == 2) {
    hipLaunchKernelGGL((wide_pm_kernel<NSPLIT, T, CB, 4, EPI, 0>), grid, dim3(256), 0, s, P);
  } else {
    if (nk == 0) hipLaunchKernelGGL((wide_pm_kernel<NSPLIT, T, CB, 4, EPI, 0>), grid, dim3(256), 0, s, P);
    else hipLaunchKernelGGL((wide_pm_kernel<NSPLIT, T, CB, 4, EPI, 2>), grid, dim3(256), 0, s, P);
  }
  return 0;
}

template <int T>
int launch_pm_t(WidePmParams& P, int epi, hipStream_t s) {
  constexpr int PPW = 4 * (20 / T);
  const int64_t units = (P.P + PPW - 1) / PPW;
  static const int cbs_n[2] = {3, 6}, cbs_f[2] = {2, 3};
  const int cb = pick_cb(units, P.N, epi == 1 ? cbs_n : cbs_f, 2);
  if (!cb || units >= (1LL << 28)) return SDF_E_SHAPE;
  P.nunits = (int)units;
  P.nrg = (int)((units + 3) / 4);
  P.ncg = P.N / (16 * cb);
  const int64_t items = (int64_t)P.ncg * P.nrg;
  if (items >= (1LL << 31) - 8) return SDF_E_SHAPE;
  const dim3 grid((unsigned)((items + 7) / 8 * 8));
  const int nk = neuron_class(P.sn);
  if (epi == 1) return cb == 6 ? launch_pm_nk<2, T, 6, 1>(P, nk, grid, s) : launch_pm_nk<2, T, 3, 1>(P, nk, grid, s);
  if (epi == 2) return cb == 3 ? launch_pm_nk<2, T, 3, 2>(P, 0, grid, s) : launch_pm_nk<2, T, 2, 2>(P, 0, grid, s);
  return cb == 3 ? launch_pm_nk<2, T, 3, 3>(P, nk, grid, s) : launch_pm_nk<2, T, 2, 3>(P, nk, grid, s);
}

int launch_pm(WidePmParams& P, int T, int epi, hipStream_t s) {
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

bool wide_env_off() {
  const char* e = getenv("SDF_WIDE");
  return e && e[0] == '0';
}

}  // namespace

// ---- host side -------------------------------------------------------------------------------------------------------------
// Shapes the wide forms are built for: two fp16 planes, LIF / IF neurons (the PSN keeps the general kernels), C a multiple of 128
// from 256 on, T in {10, 20}, operands within the kernels' 31-bit buffer offsets.
bool ms_wide_mlp_supports(const SdfMsMlpDesc* d) {
  if (wide_env_off() || (d->flags & SDF_MLP_NARROW)) return false;
  if (d->nsplit != 2 || d->C < 256 || d->C % 128 || d->Ch % 128 || d->Ch % 96) return false;
  if (d->D != 10 && d->D != 20) return false;
  if (!neuron_ok(d->sn1) || !neuron_ok(d->sn2)) return false;
  const int64_t tokens = (int64_t)d->B * d->D * d->HW;
  if (tokens * d->Ch >= (1LL << 31) || tokens * d->C * 4 >= (1LL << 31)) return false;
  if (!d->fc1_alpha || !d->fc1_beta || !d->fc2_alpha || !d->fc2_beta) return false;
  return sdf_aligned(d->x, 16) && sdf_aligned(d->fc1_planes, 16) && sdf_aligned(d->fc2_planes, 16);
}

// s1 = SN1(x) must already be in `s1` (u8 [tokens][C]); s2 receives the hidden spikes; x is updated in place
int launch_ms_wide_mlp(const SdfMsMlpDesc* d, const uint8_t* s1, uint8_t* s2, hipStream_t s) {
  WidePmParams P = {};
  P.A = s1; P.W = d->fc1_planes; P.N = d->Ch; P.K = d->C; P.HW = (int)d->HW; P.P = (int64_t)d->B * d->HW;
  P.asc = d->fc1_acc_scale; P.alpha = d->fc1_alpha; P.beta = d->fc1_beta;
  P.out_spike = s2; P.ldsp = d->Ch; P.sn = d->sn2; P.inv_tau = inv_tau_of(d->sn2);
  int rc = launch_pm(P, d->D, 1, s);
  if (rc) return rc;
  WidePmParams Q = {};
  Q.A = s2; Q.W = d->fc2_planes; Q.N = d->C; Q.K = d->Ch; Q.HW = (int)d->HW; Q.P = P.P;
  Q.asc = d->fc2_acc_scale; Q.alpha = d->fc2_alpha; Q.beta = d->fc2_beta; Q.x = d->x; Q.ldo = d->C;
  Q.sn = d->sn2;
  return launch_pm(Q, d->D, 2, s);
}

bool ms_wide_attn_supports(const SdfQkAttnDesc* d) {
  if (wide_env_off() || (d->flags & SDF_QK_NARROW)) return false;
  if (d->nsplit != 2 || d->Tq != 2 || d->C < 256 || d->C % 128 || d->C != d->nH * 32 || d->N1 < 24) return false;
  if (!d->x_src || d->xB < 1 || d->xHW < 1 || (d->xD != 10 && d->xD != 20)) return false;
  if ((int64_t)d->xB * d->xD * d->xHW != d->x_rows) return false;
  const SdfNeuronCfg* ns[4] = {&d->sn_proj, &d->sn_q, &d->sn_k, &d->sn2_q};
  const int nk = neuron_class(*ns[0]);
  for (const SdfNeuronCfg* n : ns)
    if (!neuron_ok(*n) || neuron_class(*n) != nk) return false;
  if (d->emit_s1 && !neuron_ok(d->emit_sn)) return false;
  const int64_t M = d->B_ * d->N1 * d->Tq;
  if (M * d->C >= (1LL << 31) || d->x_rows * d->C * 4 >= (1LL << 31) || M >= (1LL << 31)) return false;
  const bool fused = d->qk_planes != nullptr;
  if (!fused && (!d->q_planes || !d->k_planes)) return false;
  const float* al[4] = {fused ? d->qk_alpha : d->q_alpha, fused ? d->qk_beta : d->q_beta, fused ? d->qk_alpha : d->k_alpha,
                        fused ? d->qk_beta : d->k_beta};
  if ((al[0] == nullptr) != (al[1] == nullptr) || (al[2] == nullptr) != (al[3] == nullptr)) return false;
  if (d->p_alpha && !d->p_beta) return false;
  return sdf_aligned(d->x, 16) && sdf_aligned(d->p_planes, 16) && sdf_aligned(fused ? (const void*)d->qk_planes : (const void*)d->q_planes, 16) &&
         (fused || sdf_aligned(d->k_planes, 16));
}

// xs (2, rows, C) = SN_proj of the gathered slices -> E (2, rows, C) [+ q | k tape] ; then x += proj(E) (+ emit)
int launch_ms_wide_front(const SdfQkAttnDesc* d, const uint8_t* xs, uint8_t* e, uint8_t* qk, bool keep, hipStream_t s) {
  WideFrontParams P = {};
  const int C = d->C;
  const int64_t rows = d->B_ * d->N1, M = rows * d->Tq;
  P.xs = xs; P.rows = rows; P.N1 = d->N1; P.C = C; P.nH = d->nH;
  if (d->qk_planes) {
    P.wq = d->qk_planes; P.wk = d->qk_planes + (int64_t)C * C; P.wq_plane = P.wk_plane = 2LL * C * C;
    P.q_al = d->qk_alpha; P.q_be = d->qk_beta;
    P.k_al = d->qk_alpha ? d->qk_alpha + C : nullptr; P.k_be = d->qk_beta ? d->qk_beta + C : nullptr;
    P.pe = d->qk_add ? d->qk_add + C : nullptr; P.pe_ld = 2LL * C;
    P.q_asc = P.k_asc = d->qk_acc_scale;
    P.qs = qk; P.ks = qk + C; P.ldq = P.ldk = 2LL * C;
  } else {
    P.wq = d->q_planes; P.wk = d->k_planes; P.wq_plane = P.wk_plane = (int64_t)C * C;
    P.q_al = d->q_alpha; P.q_be = d->q_beta; P.k_al = d->k_alpha; P.k_be = d->k_beta;
    P.pe = d->k_add; P.pe_ld = C;
    P.q_asc = d->q_acc_scale; P.k_asc = d->k_acc_scale;
    P.qs = qk; P.ks = qk + M * C; P.ldq = P.ldk = C;
  }
  if (P.q_asc == 0.f) P.q_asc = 1.f;
  if (P.k_asc == 0.f) P.k_asc = 1.f;
  P.sn_q = d->sn_q; P.sn_k = d->sn_k; P.sn2_q = d->sn2_q;
  P.it_q = inv_tau_of(d->sn_q); P.it_k = inv_tau_of(d->sn_k); P.it_2 = inv_tau_of(d->sn2_q);
  P.e = e;
  // token tiles of 8 RB tokens: 40 (RB = 5) while that leaves >= 600 waves, else 16 (RB = 2)
  const int64_t t5 = (rows + 39) / 40, t2 = (rows + 15) / 16;
  const bool big = t5 * d->nH >= 600;
  const int64_t ntiles = big ? t5 : t2;
  P.ntiles = (int)ntiles;
  P.nrg = (int)((ntiles + 3) / 4);
  const int64_t items = (int64_t)P.nrg * d->nH;
  if (items >= (1LL << 31) - 8) return SDF_E_SHAPE;
  const dim3 grid((unsigned)((items + 7) / 8 * 8));
  const int nk = neuron_class(d->sn_q);
#define SDF_WF(RB_, NK_)                                                                                  \
  do {                                                                                                    \
    if (keep) hipLaunchKernelGGL((wide_front_kernel<2, RB_, NK_, true>), grid, dim3(256), 0, s, P);       \
    else hipLaunchKernelGGL((wide_front_kernel<2, RB_, NK_, false>), grid, dim3(256), 0, s, P);           \
  } while (0)
  if (big) { if (nk == 0) SDF_WF(5, 0); else SDF_WF(5, 2); }
  else { if (nk == 0) SDF_WF(2, 0); else SDF_WF(2, 2); }
#undef SDF_WF
  hipError_t err = hipGetLastError();
  return err != hipSuccess ? (int)err : 0;
}

int launch_ms_wide_proj(const SdfQkAttnDesc* d, const uint8_t* e, hipStream_t s) {
  WidePmParams P = {};
  P.A = e; P.zsrc = d->x_src; P.zg_G = (uint32_t)d->Tq * (uint32_t)d->N1 * 32u;
  P.W = d->p_planes; P.N = d->C; P.K = d->C; P.HW = (int)d->xHW; P.P = (int64_t)d->xB * d->xHW;
  P.asc = d->p_acc_scale; P.bias = d->p_bias; P.alpha = d->p_alpha; P.beta = d->p_beta;
  P.x = d->x; P.ldo = d->C;
  P.out_spike = d->emit_s1; P.ldsp = d->C;
  P.sn = d->emit_s1 ? d->emit_sn : d->sn_proj; P.inv_tau = inv_tau_of(P.sn);
  return launch_pm(P, d->xD, d->emit_s1 ? 3 : 2, s);
}

}  // namespace sdfmm

extern "C" int sdf_window_zsrc_map(const int32_t* slice_map, int64_t B_, int Tq, int N1, int nH, int32_t* x_src, void* stream) {
  if (!slice_map || !x_src) return SDF_E_NULL;
  if (B_ < 1 || Tq < 1 || N1 < 1 || nH < 1) return SDF_E_SHAPE;
  const int64_t total = B_ * Tq * N1;
  if (total * nH * 32 >= (1LL << 31)) return SDF_E_SHAPE;
  hipLaunchKernelGGL(sdfmm::zsrc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, sdf_stream(stream), slice_map, x_src, B_, Tq,
                     N1, nH, total);
  SDF_LAUNCH_CHECK();
  return 0;
}
