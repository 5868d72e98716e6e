// Ping-pong spike matrix multiply / implicit-GEMM convolution for gfx950.
//
//   out = epilogue( A (binary u8 spikes; a matrix, or an NHWC image batch read through im2col) x W^T )
//
// One persistent workgroup of 12 wavefronts per CU, three waves per SIMD, split by ROLE:
//
//   waves 0-3   consumer group 0 : tiles 0, 2, 4, ... of the workgroup's tile list
//   waves 4-7   consumer group 1 : tiles 1, 3, 5, ...
//   waves 8-11  producers        : address arithmetic (row decode, im2col taps, bounds), global loads two stages
//                                  ahead (two register sets), LDS writes of the K stages of ALL tiles in order
//
// A consumer wave owns 64 rows x 96 columns of a 256 x 96 tile (2 x 3 accumulators of v_mfma_f32_32x32x16).  K
// advances in stages of 64 through an LDS ring.  The hand-over is NOT a workgroup barrier: every ring slot has a
// `full` and an `empty` counter in LDS (4 producer waves bump `full` after their writes have landed, the 4 consumer
// waves of the owning group bump `empty` after their last read), and the waiting side polls.  So while one group
// runs the epilogue of its tile (BN / residual / neuron, global loads and stores - long and latency-bound), the
// other group is already multiplying the next tile: the matrix pipe of every SIMD stays fed across epilogues, which
// a barrier-synchronised kernel (rounds 1 - 5: spike_mm_ws.hip, deleted in round 6) cannot do.
//
// Weights: NSPLIT = 2 -> two fp16 planes of wscale * W (hi + lo = 22 significant bits; wscale is a power of two
// that places the largest |w| just under the fp16 range, the accumulator is multiplied by 1 / wscale - exact -
// before anything else touches it); NSPLIT = 3 -> three bf16 planes (all 24 bits); NSPLIT = 1 -> one bf16 plane.
// Binary spikes are exact in either 16-bit format; accumulation is fp32 in the matrix cores.
//
// Epilogues (consumer waves only):
//   F32   : acc*ascale (+bias) -> fmaf(., alpha, beta) -> (+resid) -> fp32 store, optional output row scatter
//   SPIKE : fmaf(acc*ascale, alpha, beta) (+ positional term) -> LIF / IF / PSN over the T steps of each position,
//           taken straight from the accumulator slots a lane holds (slot = 16*rowblock + reg -> position = slot / T,
//           t = slot % T) -> 1-byte spikes, staged through a private LDS region to leave as 16-byte stores.
// LDS rows are padded (A 18-dword stride for ds_read_b64, W 36-dword stride for ds_read_b128): conflict-free.
// Compiled with -ffp-contract=off (the neuron arithmetic is the separately-rounded op sequence of neuron.hip).
#include "spike_mm.h"
#include "switches.h"
#include <stdlib.h>
#include <type_traits>

#ifdef SDF_STAMP
// diagnostic build only (tools/stamp_conv.sh): per-role cycle accounting of workgroup 0
__device__ unsigned long long g_sdf_stamp[16];
#define STAMP(var) var = __builtin_readcyclecounter()
#define STAMP_ADD(acc, a, b) acc += (b) - (a)
#else
#define STAMP(var)
#define STAMP_ADD(acc, a, b)
#endif

namespace sdfmm {
namespace {

constexpr int BM = 256, BN = 96, KC = 64;
constexpr int A_LD = KC + 8;                 // bytes per A row   (72 B  = 18 dwords; 80-byte rows with ds_write_b128 measured no faster)
constexpr int W_LD = KC + 8;                 // 16-bit elements per W row (144 B = 36 dwords)
constexpr int A_BYTES = BM * A_LD;           // 18432

// Raw buffer access (32-bit byte offset against a wave-uniform descriptor of 2^31 records): an offset with bit 31 set is
// out of range, so the hardware returns zeros for such a load and drops such a store - row / tap / K bounds become an
// offset select instead of an exec-masked branch (hipcc puts a vmcnt wait behind every one of those).
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
constexpr uint32_t INV = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)INV, 0x00020000);
}
__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t r, uint32_t off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 buf_load16f(__amdgpu_buffer_rsrc_t r, uint32_t off) {
  // (element copies first: __builtin_bit_cast applied directly to a vector-element expression reads element 0 - clang bug)
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
  const uint32_t x = v.x, y = v.y, z = v.z, w = v.w;
  return make_float4(__uint_as_float(x), __uint_as_float(y), __uint_as_float(z), __uint_as_float(w));
}
__device__ __forceinline__ void buf_store16f(__amdgpu_buffer_rsrc_t r, uint32_t off, float4 o) {
  u32x4 v;
  v.x = __float_as_uint(o.x); v.y = __float_as_uint(o.y); v.z = __float_as_uint(o.z); v.w = __float_as_uint(o.w);
  __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 0);
}

// spin until the LDS counter reaches `target` (wave-uniform); later LDS accesses are not hoisted above it
__device__ __forceinline__ void wait_ge(uint32_t* p, uint32_t target) {
  while (true) {
    const uint32_t v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    if ((int32_t)(v - target) >= 0) break;
    __builtin_amdgcn_s_sleep(1);
  }
  asm volatile("" ::: "memory");
}
// all LDS operations of this wave have completed -> bump the counter (one lane).  vmcnt is deliberately not waited
// for: the producers' prefetch and the consumers' epilogue stores stay in flight.
__device__ __forceinline__ void signal(uint32_t* p, int lane) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane == 0) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// The kernel's body: `blk` / `G` = this workgroup's index and the number of workgroups that share P's work items (the plain kernel's
// blockIdx.x / gridDim.x; the multi-problem kernel below deals a range of its grid to every problem).
template <int NSPLIT, int TT, bool CONV>
__device__ __forceinline__ void spike_mm_pp_body(const GemmParams& P, const int blk, const int G) {
  constexpr bool SPIKE = TT > 0;
  constexpr int T = SPIKE ? TT : 1;
  constexpr int NPOS = 32 / T;                                   // positions per lane-half (2 row blocks = 32 slots)
  constexpr int W_BYTES = NSPLIT * BN * W_LD * 2;
  constexpr int BUF = A_BYTES + W_BYTES;                         // 46080 (2 planes) / 59904 (3 planes)
  constexpr int NSLOT = (SPIKE || NSPLIT == 3) ? 2 : 3;
  constexpr int STG = SPIKE ? 8 * 32 * BN : 0;                   // per consumer wave: 32 x 96 spike bytes (one row block at a time)
  constexpr int WCH = NSPLIT * BN * (KC / 8);                    // 16-byte chunks of a W stage
  constexpr int WIT = WCH / 256;                                 // 3 / 6 / 9 per producer lane
  static_assert(WCH % 256 == 0, "W stage must divide over the producer lanes");
  constexpr int DEC = 256 * 4 * 8 + 64;                          // producers: per lane 4 x (row offset 4 B, tap mask 4 B); 16 tap shifts
  constexpr int PAR = SPIKE ? (T * T + T + 3) / 4 * 16           // SPIKE epilogue: the PSN matrix and bias
                            : 8 * 3 * BN * 4;                    // F32 epilogue: per consumer wave bias / alpha / beta of its 96 columns
  static_assert(NSLOT * BUF + STG + DEC + PAR + 64 <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(16))) uint8_t smem[NSLOT * BUF + STG + DEC + PAR + 64];
  uint32_t* full = reinterpret_cast<uint32_t*>(smem + NSLOT * BUF + STG + DEC + PAR);
  uint32_t* empty = full + NSLOT;

  const SdfSpikeGemmDesc& d = P.d;
  // convolution geometry as individual scalars (constant member indices only: a struct copy indexed by a computed tap
  // row / column ends up in scratch, and every scratch access is a vmcnt-ordered vector memory operation)
  const int cH = P.cv.H, cW = P.cv.W, cCin = P.cv.Cin, cOH = P.cv.OH, cOW = P.cv.OW, csy = P.cv.sy, csx = P.cv.sx;
  const int cKW = P.cv.KWc, ckwm = P.cv.kw_mul;
  const int dy0 = P.cv.dy[0], dy1 = P.cv.dy[1], dy2 = P.cv.dy[2], dx0 = P.cv.dx[0], dx1 = P.cv.dx[1], dx2 = P.cv.dx[2];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;   // scalar: role branches are uniform
  const int K = d.K, N = d.N;
  const int nstages = P.spc;                                     // stages per work item (= all of K unless split-K)
  const int ksplit = P.ksplit;

  if (tid < 2 * NSLOT) full[tid] = 0;
  if (SPIKE && d.sn_kind == SDF_PSN && tid >= 128 && tid < 128 + T * T + T) {   // PSN weights: read as LDS broadcasts in the epilogue
    const int i = tid - 128;
    reinterpret_cast<float*>(smem + NSLOT * BUF + STG + DEC)[i] = i < T * T ? d.psn_w[i] : d.psn_b[i - T * T];
  }
  if (CONV && tid >= 64 && tid < 80) {                           // byte shift of tap tp in the NHWC image (taps >= KH*KW: unused)
    const int tp = tid - 64;
    const int ky = (tp * ckwm) >> 5, kx = tp - ky * cKW;
    const int ddy = ky == 0 ? dy0 : (ky == 1 ? dy1 : dy2), ddx = kx == 0 ? dx0 : (kx == 1 ? dx1 : dx2);
    reinterpret_cast<int*>(smem + NSLOT * BUF + STG + 256 * 4 * 8)[tp] = tp < 9 ? (ddy * cW + ddx) * cCin : 0;
  }
  __syncthreads();

  // contiguous range of work items of this workgroup; item = tile * ksplit + kchunk, tiles column-block-major
  // (t = cb * tiles_m + rt).  Workgroups are dealt round-robin to the 8 XCDs, so consecutive ranges go to the
  // workgroups of ONE XCD: neighbouring row tiles (which share their im2col halo rows) meet in the same L2.
  int wg = blk;
  if ((G & 7) == 0) wg = (wg & 7) * (G >> 3) + (wg >> 3);
  const int nitems = P.ntiles * ksplit;
  const int base = nitems / G, rem = nitems % G;
  const int t_begin = wg * base + (wg < rem ? wg : rem);
  const int n_my = base + (wg < rem ? 1 : 0);
  if (n_my == 0) return;
  const int Q = n_my * nstages;                                  // stages this workgroup streams through

  // global row (or -1) of tile-row R of row-tile rt
  auto tile_row = [&](int rt, int R) __attribute__((always_inline)) -> int64_t {
    if (SPIKE) {
      const int w = R >> 6, rloc = R & 63;
      const int rb = rloc >> 5, rr = rloc & 31;
      const int h = (rr >> 2) & 1, r = (rr & 3) + 4 * (rr >> 3);
      const int slot = rb * 16 + r;
      const int pl = slot / T, t = slot - pl * T;
      const int64_t pos = (int64_t)rt * (8 * NPOS) + (w * 2 + h) * NPOS + pl;
      if (pl >= NPOS || pos >= d.pos_count) return -1;
      const uint32_t po = (uint32_t)pos / (uint32_t)d.pos_inner;
      return (int64_t)po * d.pos_ostride + ((uint32_t)pos - po * (uint32_t)d.pos_inner) + (int64_t)t * d.t_stride;
    }
    const int64_t m = (int64_t)rt * BM + R;
    return m < d.M ? m : -1;
  };

  if (wave >= 8) {
    // =============================== PRODUCERS ===============================
    // A: four lanes per tile row (16 bytes each = the row's 64-byte K slice is one contiguous segment), rows
    // r4, r4+64, r4+128, r4+192.  W: eight lanes per 128-byte row slice.
    const int ptid = tid - 512;
    const int r4 = ptid >> 2, j4 = ptid & 3;
    struct Regs { uint4 a[4]; uint4 w[WIT]; };
    Regs R0, R1;
    uint32_t a_off[4] = {INV, INV, INV, INV};                    // row base (byte offset into A) or INV
    const __amdgpu_buffer_rsrc_t A_rs = make_rsrc(d.A);
    // exact size: the last K stage may be partial - its out-of-row reads hit the next weight row (finite, multiplied by
    // zero spikes) or, past the last row, the end of the buffer (zeros) - so the weight loads need no per-lane bound
    const __amdgpu_buffer_rsrc_t W_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(d.Wp), 0, NSPLIT * N * K * 2, 0x00020000);
    uint32_t a_valid[4] = {0, 0, 0, 0};                          // CONV: bit tap = that tap's input pixel is inside the image
    const uint32_t zg_gstride = (uint32_t)(d.zg_T * d.zg_N1 * 32);
    const int* tsh_s = reinterpret_cast<const int*>(smem + NSLOT * BUF + STG + 256 * 4 * 8);
    uint32_t w_off[WIT];
    int w_lds[WIT];
#pragma unroll
    for (int i = 0; i < WIT; ++i) {
      const int c = ptid + 256 * i;
      const int row = c >> 3, cc = c & 7;                        // row = p*96 + n
      const int p = row / BN, n = row - p * BN;
      w_off[i] = (((uint32_t)p * (uint32_t)N + (uint32_t)n) * (uint32_t)K + 8u * cc) * 2u;     // bytes
      w_lds[i] = A_BYTES + (row * W_LD + 8 * cc) * 2;
    }
    // stage iterator of the LOADER (no divisions in steady state)
    int ld_st = 0, ld_kc = t_begin % ksplit;
    int ld_cb = (t_begin / ksplit) / P.tiles_m, ld_rt = (t_begin / ksplit) - ld_cb * P.tiles_m;
    int ld_tap = 0, ld_c0 = 0;                                   // CONV: first tap of the stage and the channel offset inside it
    bool ld_new = true;
    // new tile: decode this lane's four rows.  A ROLLED loop (small code, few live registers) that parks its results
    // in a private LDS strip, from where they are read back into statically indexed registers.
    uint32_t* dec_off = reinterpret_cast<uint32_t*>(smem + NSLOT * BUF + STG) + ptid * 4;
    uint32_t* dec_val = reinterpret_cast<uint32_t*>(smem + NSLOT * BUF + STG + 256 * 4 * 4) + ptid * 4;
    auto decode = [&]() __attribute__((always_inline)) {
#pragma unroll 1
      for (int i = 0; i < 4; ++i) {
        const int64_t g = tile_row(ld_rt, r4 + 64 * i);
        uint32_t off = g < 0 ? INV : (uint32_t)g * (uint32_t)d.lda;
        uint32_t vm = 0;
        if (CONV) {
          if (g >= 0) {
            const uint32_t ohw = (uint32_t)(cOH * cOW);
            const uint32_t img = (uint32_t)g / ohw;
            const uint32_t r2 = (uint32_t)g - img * ohw;
            const uint32_t oy = r2 / (uint32_t)cOW, ox = r2 - oy * (uint32_t)cOW;
            const int iy0 = (int)oy * csy, ix0 = (int)ox * csx;
            off = ((img * (uint32_t)cH + (uint32_t)iy0) * (uint32_t)cW + (uint32_t)ix0) * (uint32_t)cCin;   // un-shifted pixel
            // rows / columns of the tap grid that fall inside the image, then one bit per tap
            const uint32_t ym = ((unsigned)(iy0 + dy0) < (unsigned)cH ? 1u : 0u) | ((unsigned)(iy0 + dy1) < (unsigned)cH ? 2u : 0u) |
                                ((unsigned)(iy0 + dy2) < (unsigned)cH ? 4u : 0u);
            const uint32_t xm = ((unsigned)(ix0 + dx0) < (unsigned)cW ? 1u : 0u) | ((unsigned)(ix0 + dx1) < (unsigned)cW ? 2u : 0u) |
                                ((unsigned)(ix0 + dx2) < (unsigned)cW ? 4u : 0u);
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
              const int ky = (tp * ckwm) >> 5, kx = tp - ky * cKW;       // wave-uniform
              vm |= (((ym >> ky) & (xm >> kx)) & 1u) << tp;
            }
          }
        } else if (!SPIKE && d.zg_nH > 0 && g >= 0) {
          const uint32_t bn = (uint32_t)(d.zg_B * d.zg_N1);
          const uint32_t zt = (uint32_t)g / bn;
          const uint32_t r2 = (uint32_t)g - zt * bn;
          const uint32_t zb = r2 / (uint32_t)d.zg_N1;
          const uint32_t zn = r2 - zb * (uint32_t)d.zg_N1;
          off = (((zb * (uint32_t)d.zg_nH) * (uint32_t)d.zg_T + zt) * (uint32_t)d.zg_N1 + zn) * 32u;
          if (d.zg_rep > 0) {                                        // independent replicas (SdfSpikeGemmDesc.zg_rep)
            const uint32_t rr = zb / (uint32_t)d.zg_rep, zbr = zb - rr * (uint32_t)d.zg_rep;
            const uint32_t half = (uint32_t)d.zg_rep * (uint32_t)d.zg_N1 * (uint32_t)d.K;
            const uint32_t o = (((zbr * (uint32_t)d.zg_nH) * (uint32_t)d.zg_T + zt) * (uint32_t)d.zg_N1 + zn) * 32u;
            const uint32_t te = o / half;
            off = (te * (uint32_t)d.zg_B + rr * (uint32_t)d.zg_rep) * (uint32_t)d.zg_N1 * (uint32_t)d.K + (o - te * half);
          }
        }
        dec_off[i] = off;
        dec_val[i] = vm;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // own writes, same wave: in order
#pragma unroll
      for (int i = 0; i < 4; ++i) { a_off[i] = dec_off[i]; a_valid[i] = dec_val[i]; }
    };
    auto load = [&](Regs& R) __attribute__((always_inline)) {
      const int k0 = (ld_kc * nstages + ld_st) * KC;
      const int k = k0 + 16 * j4;                                // this lane's K offset inside the stage
      if (CONV) {
        // first tap of the stage and its channel offset are wave-uniform and advance incrementally; this lane's 16-byte
        // chunk lies in that tap or the next one (Cin >= 48)
        if (ld_st == 0) { ld_tap = k0 / cCin; ld_c0 = k0 - ld_tap * cCin; }
        int c = ld_c0 + 16 * j4;
        const bool nxt = c >= cCin;
        if (nxt) c -= cCin;
        const int tp = ld_tap + (nxt ? 1 : 0);
        const uint32_t sh = (uint32_t)(tsh_s[tp] + c);            // may be "negative": wraps back into range
        const uint32_t bit = 1u << tp;
#pragma unroll
        for (int i = 0; i < 4; ++i) R.a[i] = buf_load16(A_rs, (k < K && (a_valid[i] & bit)) ? a_off[i] + sh : INV);
        ld_c0 += KC;
        if (ld_c0 >= cCin) { ld_c0 -= cCin; ++ld_tap; }
        if (ld_c0 >= cCin) { ld_c0 -= cCin; ++ld_tap; }
      } else {
        const uint32_t ko = (!SPIKE && d.zg_nH > 0) ? (uint32_t)(k >> 5) * zg_gstride + (uint32_t)(k & 31) : (uint32_t)k;
#pragma unroll
        for (int i = 0; i < 4; ++i) R.a[i] = buf_load16(A_rs, (a_off[i] != INV && k < K) ? a_off[i] + ko : INV);
      }
      const uint32_t wbase = ((uint32_t)(ld_cb * BN) * (uint32_t)K + (uint32_t)k0) * 2u;     // wave-uniform bytes: the soffset
#pragma unroll
      for (int i = 0; i < WIT; ++i) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(W_rs, w_off[i], wbase, 0);
        R.w[i] = make_uint4(v.x, v.y, v.z, v.w);
      }
      // advance
      if (++ld_st == nstages) {
        ld_st = 0;
        if (++ld_kc == ksplit) {
          ld_kc = 0; ld_new = true;
          if (++ld_rt == P.tiles_m) { ld_rt = 0; ++ld_cb; }
        }
      }
    };
    auto store = [&](const Regs& R, int slot) __attribute__((always_inline)) {
      uint8_t* B = smem + slot * BUF;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        uint2* ap = reinterpret_cast<uint2*>(B + (r4 + 64 * i) * A_LD + 16 * j4);      // rows are 8-byte aligned
        ap[0] = make_uint2(R.a[i].x, R.a[i].y);
        ap[1] = make_uint2(R.a[i].z, R.a[i].w);
      }
#pragma unroll
      for (int i = 0; i < WIT; ++i) *reinterpret_cast<uint4*>(B + w_lds[i]) = R.w[i];
    };

    // Stage q is written to the ring at step q and its loads were issued at step q - 2 into register set q & 1 (the
    // first two steps only load).  Unrolled by two so that each register set has its own straight-line code.
    int slot = 0;
    uint32_t use = 0;
#ifdef SDF_STAMP
    unsigned long long p0 = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0, s_wait = 0, s_store = 0, s_dec = 0, s_load = 0;
#endif
    auto step = [&](Regs& R, int q) __attribute__((always_inline)) {
      STAMP(p0);
      if (q >= 0 && q < Q) {
        if (use) wait_ge(&empty[slot], 4 * use);
        STAMP(p1);
        store(R, slot);
        signal(&full[slot], lane);
        if (++slot == NSLOT) { slot = 0; ++use; }
        STAMP(p2);
        STAMP_ADD(s_wait, p0, p1); STAMP_ADD(s_store, p1, p2);
      }
      STAMP(p2);
      if (q + 2 < Q) {
        if (ld_new) { ld_new = false; decode(); }
        STAMP(p3);
        load(R);
        STAMP(p4);
        STAMP_ADD(s_dec, p2, p3); STAMP_ADD(s_load, p3, p4);
      }
    };
    for (int q = -2; q < Q; q += 2) {
      step(R0, q);
      step(R1, q + 1);
    }
#ifdef SDF_STAMP
    if (blk == 0 && tid == 512) { g_sdf_stamp[0] = s_wait; g_sdf_stamp[1] = s_store; g_sdf_stamp[2] = s_dec; g_sdf_stamp[3] = s_load; g_sdf_stamp[4] = Q; }
#endif
    return;
  }

  // =============================== CONSUMERS ===============================
  const int grp = wave >> 2, cw = wave & 3;
  const float asc = P.acc_scale;
  f32x16 acc[2][3];
  const bool soft = d.soft_reset != 0;
#ifdef SDF_STAMP
  unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0, s_cwait = 0, s_mma = 0, s_epi = 0, kstart = __builtin_readcyclecounter();
#endif
  for (int it = grp; it < n_my; it += 2) {
    // lane-derived values are re-derived per tile from a laundered copy: otherwise the compiler hoists every address
    // and mask of the epilogue out of the tile loop, keeps them live across the main loop and spills them - and a
    // scratch reload between the epilogue's stores is a vmcnt-ordered round trip
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int l31 = ln & 31, lh = ln >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jn = 0; jn < 3; ++jn)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.f;
    for (int st = 0; st < nstages; ++st) {
      const int q = it * nstages + st;
      const int slot = q % NSLOT;
      const uint32_t use = (uint32_t)(q / NSLOT);
      STAMP(c0);
      wait_ge(&full[slot], 4 * (use + 1));
      STAMP(c1);
      const uint8_t* A_s = smem + slot * BUF;
      const uint16_t* W_s = reinterpret_cast<const uint16_t*>(A_s + A_BYTES);
      // Fragment reads run one (k-step, plane) unit ahead of the MFMAs that use them: the 3 ds_read_b128 of the next unit's
      // weight fragments (and, at a new k-step, the 2 ds_read_b64 of its spike fragments) are in flight while the 6 MFMAs of the
      // current unit execute (192 cycles > LDS latency), so the s_waitcnt lgkmcnt in front of an MFMA finds its operands
      // landed.  Two sets of 3 x 4 + 2 x 2 registers - small enough for every variant of the kernel.
      constexpr int UNITS = (KC / 16) * NSPLIT;
      uint2 fa[2][2];
      bf16x8 fb[2][3];
      auto frag_load = [&](int u, uint2 (&xa)[2], bf16x8 (&xb)[3], bool with_a) __attribute__((always_inline)) {
        const int ks = u / NSPLIT, p = u - ks * NSPLIT;
        if (with_a) {
#pragma unroll
          for (int rb = 0; rb < 2; ++rb)
            xa[rb] = *reinterpret_cast<const uint2*>(&A_s[(cw * 64 + rb * 32 + l31) * A_LD + ks * 16 + 8 * lh]);
        }
#pragma unroll
        for (int nb = 0; nb < 3; ++nb)
          xb[nb] = *reinterpret_cast<const bf16x8*>(&W_s[(p * BN + nb * 32 + l31) * W_LD + ks * 16 + 8 * lh]);
      };
      frag_load(0, fa[0], fb[0], true);
      bf16x8 a0, a1;
#pragma unroll
      for (int u = 0; u < UNITS; ++u) {
        const int ks = u / NSPLIT, p = u - ks * NSPLIT;
        if (u + 1 < UNITS) frag_load(u + 1, fa[((u + 1) / NSPLIT) & 1], fb[(u + 1) & 1], (u + 1) % NSPLIT == 0);
        __builtin_amdgcn_sched_barrier(0);                       // keep the prefetch above this unit's MFMAs
        if (p == 0) { a0 = expand_spikes<NSPLIT>(fa[ks & 1][0]); a1 = expand_spikes<NSPLIT>(fa[ks & 1][1]); }
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) {
          if (SPIKE) {                                             // rows = (position, t) slots of a lane: the neuron runs on registers
            acc[0][nb] = mma<NSPLIT>(a0, fb[u & 1][nb], acc[0][nb]);
            acc[1][nb] = mma<NSPLIT>(a1, fb[u & 1][nb], acc[1][nb]);
          } else {                                                 // fp32 epilogue: weights as the row operand - accumulator quads are
            acc[0][nb] = mma<NSPLIT>(fb[u & 1][nb], a0, acc[0][nb]);   // four consecutive columns of one output row (no transpose)
            acc[1][nb] = mma<NSPLIT>(fb[u & 1][nb], a1, acc[1][nb]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      signal(&empty[slot], lane);                                // every fragment of this slot is in registers
      STAMP(c2);
      STAMP_ADD(s_cwait, c0, c1); STAMP_ADD(s_mma, c1, c2);
    }
    STAMP(c2);

    // ------------------------------- epilogue of this tile -------------------------------
    const int item = t_begin + it;
    const int t = item / ksplit, kc = item - t * ksplit;
    const int cb = t / P.tiles_m, rt = t - cb * P.tiles_m;
    const int n0 = cb * BN;
    if (!SPIKE) {
      // The weights were the MFMA's row operand: lane (l31, lh) holds output row l31 of each of its two row blocks and, in
      // accumulator quad q4 of column block nb, the four consecutive columns n0 + 32 nb + 8 q4 + 4 lh + 0..3 -> 16-byte loads /
      // stores with no transpose (round 2 multiplied the other way round and transposed 24 quads per tile).
      // vmcnt counts loads AND stores in order on CDNA4, so a load placed between stores makes its s_waitcnt drain
      // the stores in front of it.  Hence: per batch, ALL residual loads are issued and waited for (pinned in straight-line
      // code) before its first store, and there are no per-lane "load or constant" selects (hipcc branches around those and waits).
      const int mrow0 = rt * BM + cw * 64 + l31;                 // + rb*32
      const bool has_map = d.out_rowmap != nullptr, has_res = d.resid != nullptr;
      if (ksplit > 1) {
        // split-K: raw fp32 partial sums; scale / bias / BN / residual / scatter happen in splitk_reduce_kernel
        float* pbase = P.partial + (int64_t)kc * d.M * N;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          const int m = mrow0 + rb * 32;
          float* op = pbase + (int64_t)m * N + n0 + 4 * lh;
#pragma unroll
          for (int nb = 0; nb < 3; ++nb)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4)
              if (m < (int)d.M)
                *reinterpret_cast<float4*>(op + nb * 32 + 8 * q4) =
                    make_float4(acc[rb][nb][q4 * 4 + 0], acc[rb][nb][q4 * 4 + 1], acc[rb][nb][q4 * 4 + 2], acc[rb][nb][q4 * 4 + 3]);
        }
      } else {
        // byte offset of each of this lane's 2 rows in out / resid, or INV (row past M, or dropped by the row map)
        const uint32_t ldo4 = (uint32_t)d.ldo * 4u;
        uint32_t dstm[2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          const int m = mrow0 + rb * 32;
          int r = m;
          if (has_map) r = d.out_rowmap[m < (int)d.M ? m : 0];
          dstm[rb] = (m < (int)d.M && r >= 0) ? (uint32_t)r * ldo4 : INV;
        }
        {
        // Six batches (column block nb, row block rb) of 4 quads per lane.  The residual loads of batch
        // k + 1 are issued BEFORE the stores of batch k: vmcnt retires in order, so waiting for them never waits for a
        // store, and one memory round trip overlaps the next.  Bias / alpha / beta of the tile's 96 columns are parked
        // in a private LDS strip (lgkmcnt, not vmcnt).
        float* par_s = reinterpret_cast<float*>(smem + NSLOT * BUF + STG + DEC) + wave * (3 * BN);
        {
          const int c4 = ln < 24 ? ln : 0;                        // 24 lanes x 4 columns per parameter
          float4 p0 = make_float4(0.f, 0.f, 0.f, 0.f), p1 = make_float4(1.f, 1.f, 1.f, 1.f), p2 = make_float4(0.f, 0.f, 0.f, 0.f);
          if (d.bias) p0 = *reinterpret_cast<const float4*>(d.bias + n0 + 4 * c4);
          if (d.alpha) { p1 = *reinterpret_cast<const float4*>(d.alpha + n0 + 4 * c4); p2 = *reinterpret_cast<const float4*>(d.beta + n0 + 4 * c4); }
          if (ln < 24) {
            *reinterpret_cast<float4*>(par_s + 4 * c4) = p0;
            *reinterpret_cast<float4*>(par_s + BN + 4 * c4) = p1;
            *reinterpret_cast<float4*>(par_s + 2 * BN + 4 * c4) = p2;
          }
        }
        const __amdgpu_buffer_rsrc_t res_rs = make_rsrc(d.resid), out_rs = make_rsrc(d.out);   // offsets < 2^31: launcher
        float4 rs[2][4];
        auto load_rs = [&](int nb, int rb, float4 (&rr)[4]) __attribute__((always_inline)) {
          const uint32_t cb4 = (uint32_t)(n0 + nb * 32 + 4 * lh) * 4u;
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) rr[q4] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (has_res) {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) rr[q4] = buf_load16f(res_rs, dstm[rb] + cb4 + 32u * q4);      // INV + ... stays out of range
          }
        };
        load_rs(0, 0, rs[0]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // parameter strip written (same wave: in order)
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          const int nb = k >> 1, rb = k & 1;
          __builtin_amdgcn_sched_barrier(0);                    // keep the batches apart: bounded live ranges, no spills
          if (k + 1 < 6) load_rs((k + 1) >> 1, (k + 1) & 1, rs[(k + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) asm volatile("" :: "v"(rs[k & 1][q4].x), "v"(rs[k & 1][q4].w));    // waits pinned here
          const uint32_t cb4 = (uint32_t)(n0 + nb * 32 + 4 * lh) * 4u;
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const float4 bs = *reinterpret_cast<const float4*>(par_s + nb * 32 + 8 * q4 + 4 * lh);
            const float4 al = *reinterpret_cast<const float4*>(par_s + BN + nb * 32 + 8 * q4 + 4 * lh);
            const float4 be = *reinterpret_cast<const float4*>(par_s + 2 * BN + nb * 32 + 8 * q4 + 4 * lh);
            float4 o = make_float4(acc[rb][nb][q4 * 4 + 0] * asc, acc[rb][nb][q4 * 4 + 1] * asc, acc[rb][nb][q4 * 4 + 2] * asc,
                                   acc[rb][nb][q4 * 4 + 3] * asc);
            o.x += bs.x; o.y += bs.y; o.z += bs.z; o.w += bs.w;
            o.x = __builtin_fmaf(o.x, al.x, be.x); o.y = __builtin_fmaf(o.y, al.y, be.y);
            o.z = __builtin_fmaf(o.z, al.z, be.z); o.w = __builtin_fmaf(o.w, al.w, be.w);
            const float4 r = rs[k & 1][q4];
            o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
            buf_store16f(out_rs, dstm[rb] + cb4 + 32u * q4, o);
          }
        }
        }
      }
    } else {
      uint8_t* S_s = smem + NSLOT * BUF + wave * 32 * BN;        // private 32 x 96 byte staging strip of this wave
      const int64_t pos0 = (int64_t)rt * (8 * NPOS) + (cw * 2 + lh) * NPOS;
      uint32_t bits[3];                                          // per column block: bit slot = spike of accumulator slot
      // Membrane output (d.out != NULL): the pre-activation (+ residual) is ALSO stored as fp32 - the MS shortcut stream
      // and the spikes of the next layer's neuron leave one kernel (MS_ResBlock: conv2 -> BN -> + identity, then the next
      // block's sn1, reference Spiking_modules.py:922-933).  Row of (position, t) as for the spikes; 4-byte stores, 32 lanes
      // = one 128-byte line.
      const bool memb = d.out != nullptr;
      const __amdgpu_buffer_rsrc_t mo_rs = make_rsrc(memb ? d.out : nullptr), mr_rs = make_rsrc(d.resid);
      uint32_t grow[NPOS];                                       // byte offset of row (position pl, t = 0) in out / resid, or INV
      if (memb) {
#pragma unroll
        for (int pl = 0; pl < NPOS; ++pl) {
          const int64_t pos = pos0 + pl;
          const uint32_t pc = pos < d.pos_count ? (uint32_t)pos : 0u;
          const uint32_t po = pc / (uint32_t)d.pos_inner;
          const uint32_t g0 = po * (uint32_t)d.pos_ostride + (pc - po * (uint32_t)d.pos_inner);
          grow[pl] = pos < d.pos_count ? g0 * (uint32_t)d.ldo * 4u : INV;
        }
      }
      const uint32_t tstep = (uint32_t)d.t_stride * (uint32_t)d.ldo * 4u;
      // residual of (column block, position) j + 1 is requested before the stores of j go out (vmcnt retires in order)
      float rsd[2][T];
      auto load_rsd = [&](int j, float (&r)[T]) __attribute__((always_inline)) {
        const int nb = j / NPOS, pl = j - nb * NPOS;
        const uint32_t base = grow[pl] + (uint32_t)(n0 + nb * 32 + l31) * 4u;
#pragma unroll
        for (int t2 = 0; t2 < T; ++t2) r[t2] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(mr_rs, base + (uint32_t)t2 * tstep, 0, 0));
      };
      const bool has_rsd = memb && d.resid != nullptr;
      if (has_rsd) load_rsd(0, rsd[0]);
#pragma unroll
      for (int nb = 0; nb < 3; ++nb) {
        const int n = n0 + nb * 32 + l31;
        const float al = d.alpha ? d.alpha[n] : 1.f;
        const float be = d.alpha ? d.beta[n] : 0.f;
        uint32_t bm = 0;
#pragma unroll
        for (int pl = 0; pl < NPOS; ++pl) {
          const int j = nb * NPOS + pl;
          const int64_t pos = pos0 + pl;
          float xs[T], sp[T];
#pragma unroll
          for (int t2 = 0; t2 < T; ++t2) {
            const int slot = pl * T + t2;
            xs[t2] = __builtin_fmaf(acc[slot >> 4][nb][slot & 15] * asc, al, be);   // al = 1, be = 0 when there is no BN
          }
          if (memb) {
            if (has_rsd) {
              if (j + 1 < 3 * NPOS) load_rsd(j + 1, rsd[(j + 1) & 1]);
#pragma unroll
              for (int t2 = 0; t2 < T; ++t2) xs[t2] = xs[t2] + rsd[j & 1][t2];
            }
#pragma unroll
            for (int t2 = 0; t2 < T; ++t2)
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(xs[t2]), mo_rs, grow[pl] + (uint32_t)t2 * tstep + (uint32_t)n * 4u, 0, 0);
          }
          if (d.add) {                                           // wave-uniform; loads unconditional (clamped position)
            const int64_t pc = pos < d.pos_count ? pos : 0;
            const float* addp = d.add + ((uint32_t)pc % (uint32_t)d.add_prows) * (int64_t)N + n;
#pragma unroll
            for (int t2 = 0; t2 < T; ++t2) xs[t2] = xs[t2] + addp[(int64_t)t2 * d.add_prows * N];
          }
          if (d.sn_kind == SDF_PSN) {
            const float* psn_s = reinterpret_cast<const float*>(smem + NSLOT * BUF + STG + DEC);
#pragma unroll
            for (int t2 = 0; t2 < T; ++t2) {
              float hh = psn_s[T * T + t2];
#pragma unroll
              for (int k = 0; k < T; ++k) hh = __builtin_fmaf(psn_s[t2 * T + k], xs[k], hh);
              sp[t2] = hh >= 0.f ? 1.f : 0.f;
            }
          } else {
            lif_steps<T>(xs, sp, d.sn_kind, soft, d.v_reset, d.v_th, d.tau, P.inv_tau);
          }
#pragma unroll
          for (int t2 = 0; t2 < T; ++t2) bm |= ((__float_as_uint(sp[t2]) >> 29) & 1u) << (pl * T + t2);   // 1.0f has bit 29 set
        }
        bits[nb] = bm;
      }
      // spikes leave as 16-byte row segments: one 32-row block at a time through the wave's private staging strip (its
      // own LDS operations complete in order: wave-level fences, no workgroup barrier)
      const __amdgpu_buffer_rsrc_t sp_rs = make_rsrc(d.out_spike);
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
        for (int nb = 0; nb < 3; ++nb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int rowl = (r & 3) + 8 * (r >> 2) + 4 * lh;
            S_s[rowl * BN + nb * 32 + l31] = (uint8_t)((bits[nb] >> (rb * 16 + r)) & 1u);
          }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int c = ln; c < 32 * (BN / 16); c += 64) {
          const int rowl = c / (BN / 16), c16 = c - rowl * (BN / 16);
          const int64_t g = tile_row(rt, cw * 64 + rb * 32 + rowl);
          const uint4 sv = *reinterpret_cast<const uint4*>(&S_s[rowl * BN + 16 * c16]);
          u32x4 v;
          v.x = sv.x; v.y = sv.y; v.z = sv.z; v.w = sv.w;
          __builtin_amdgcn_raw_buffer_store_b128(v, sp_rs, g >= 0 ? (uint32_t)g * (uint32_t)N + (uint32_t)(n0 + 16 * c16) : INV, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // reads done before the strip is overwritten
        __builtin_amdgcn_wave_barrier();
      }
    }
    STAMP(c3);
    STAMP_ADD(s_epi, c2, c3);
  }
#ifdef SDF_STAMP
  if (blk == 0 && (tid == 0 || tid == 256)) {
    const int o = tid == 0 ? 5 : 10;
    g_sdf_stamp[o] = s_cwait; g_sdf_stamp[o + 1] = s_mma; g_sdf_stamp[o + 2] = s_epi; g_sdf_stamp[o + 3] = __builtin_readcyclecounter() - kstart;
    g_sdf_stamp[o + 4] = (n_my + 1 - grp) / 2;
  }
#endif
}

template <int NSPLIT, int TT, bool CONV>
__global__ __launch_bounds__(768) void spike_mm_pp_kernel(GemmParams P) {
  spike_mm_pp_body<NSPLIT, TT, CONV>(P, (int)blockIdx.x, (int)gridDim.x);
}

// Up to four convolutions that differ only in their taps, weights and output row map - the output-parity classes of a stride-2
// transposed convolution (reference Spiking_modules.py:461-474; engine.py: deconv_classes) - as ONE launch: every class gets a range of
// the grid and walks its own work items.  Alone, each class fills about half of the chip (135 two-item workgroups at the last decoder
// level); together they fill it, and three launch boundaries go.  The per-class fields are scalar selects on the class index.
constexpr int PP_MULTI_MAX = 4;
struct GemmMulti {
  GemmParams base;
  const uint16_t* Wp[PP_MULTI_MAX];
  const int32_t* rowmap[PP_MULTI_MAX];
  int K[PP_MULTI_MAX], spc[PP_MULTI_MAX], KWc[PP_MULTI_MAX], kw_mul[PP_MULTI_MAX];
  int dy[PP_MULTI_MAX][3], dx[PP_MULTI_MAX][3];
  float acc_scale[PP_MULTI_MAX];
  int first[PP_MULTI_MAX + 1];        // first workgroup of class i; first[n] = grid size
  int n;
};

template <int NSPLIT>
__global__ __launch_bounds__(768) void spike_mm_pp_multi_kernel(GemmMulti M) {
  int i = 0;
#pragma unroll
  for (int k = 1; k < PP_MULTI_MAX; ++k)
    if (k < M.n && (int)blockIdx.x >= M.first[k]) i = k;
  i = __builtin_amdgcn_readfirstlane(i);
#define SDF_PICK(f) (i == 0 ? M.f[0] : (i == 1 ? M.f[1] : (i == 2 ? M.f[2] : M.f[3])))
#define SDF_PICK2(f, j) (i == 0 ? M.f[0][j] : (i == 1 ? M.f[1][j] : (i == 2 ? M.f[2][j] : M.f[3][j])))
  GemmParams P = M.base;
  P.d.Wp = SDF_PICK(Wp);
  P.d.out_rowmap = SDF_PICK(rowmap);
  P.d.K = SDF_PICK(K);
  P.spc = SDF_PICK(spc);
  P.cv.KWc = SDF_PICK(KWc);
  P.cv.kw_mul = SDF_PICK(kw_mul);
  P.acc_scale = SDF_PICK(acc_scale);
#pragma unroll
  for (int j = 0; j < 3; ++j) { P.cv.dy[j] = SDF_PICK2(dy, j); P.cv.dx[j] = SDF_PICK2(dx, j); }
  const int f0 = SDF_PICK(first), f1 = i == 0 ? M.first[1] : (i == 1 ? M.first[2] : (i == 2 ? M.first[3] : M.first[4]));
#undef SDF_PICK
#undef SDF_PICK2
  spike_mm_pp_body<NSPLIT, 0, true>(P, (int)blockIdx.x - f0, f1 - f0);
}

template <int NSPLIT, bool CONV>
int launch_t(const GemmParams& P, dim3 grid, hipStream_t s) {
  switch (P.d.sn_T) {
    case 0: SDF_LAUNCH((spike_mm_pp_kernel<NSPLIT, 0, CONV>), grid, dim3(768), 0, s, P); return 0;
    case 2:
      if constexpr (!CONV) { SDF_LAUNCH((spike_mm_pp_kernel<NSPLIT, 2, CONV>), grid, dim3(768), 0, s, P); return 0; }
      return SDF_E_SHAPE;
    case 10: SDF_LAUNCH((spike_mm_pp_kernel<NSPLIT, 10, CONV>), grid, dim3(768), 0, s, P); return 0;
    case 20:                                                       // one position per lane half (20 of its 32 accumulator slots)
      if constexpr (!CONV && NSPLIT == 2) { SDF_LAUNCH((spike_mm_pp_kernel<NSPLIT, 20, CONV>), grid, dim3(768), 0, s, P); return 0; }
      return SDF_E_SHAPE;
    default: return SDF_E_SHAPE;
  }
}

}  // namespace

// true when the ping-pong kernel has an instantiation for this problem (the caller refuses the shape otherwise)
bool spike_mm_pp_supports(const GemmParams& P, bool conv) {
  const SdfSpikeGemmDesc& d = P.d;
  if (d.N % BN) return false;
  if (d.sn_T != 0 && d.sn_T != 10 && !(d.sn_T == 2 && !conv) && !(d.sn_T == 20 && !conv && d.nsplit == 2)) return false;
  // every operand is addressed with a 31-bit byte offset against a raw buffer descriptor
  const int64_t lim = (int64_t)1 << 31;
  int64_t rows = d.M;                                           // highest row index + 1 touched in A / out_spike
  if (d.sn_T > 0) rows = ((d.pos_count + d.pos_inner - 1) / d.pos_inner) * d.pos_ostride + d.pos_inner + d.sn_T * d.t_stride;
  const int64_t a_bytes = conv ? (d.M / ((int64_t)P.cv.OH * P.cv.OW)) * P.cv.H * P.cv.W * P.cv.Cin
                               : (d.zg_nH > 0 ? d.M * d.K : rows * d.lda);
  if (a_bytes >= lim || (int64_t)d.nsplit * d.N * d.K * 2 >= lim || (d.sn_T > 0 && rows * d.N >= lim)) return false;
  if (d.sn_T > 0 && d.out && rows * d.ldo * 4 >= lim) return false;          // membrane output of the fused-neuron epilogue
  if (d.sn_T == 0) {
    const int64_t max_rows = d.out_rowmap ? (d.out_rows > 0 ? d.out_rows : lim) : d.M;   // scattered rows need the caller's bound
    if (max_rows * d.ldo * 4 >= lim) return false;
  }
  return true;
}

// tiles, split-K plan and the number of workgroups of one problem (P is completed in place); < 0 = an SDF_E_* code
static int pp_plan(GemmParams& P) {
  const SdfSpikeGemmDesc& d = P.d;
  if (d.N % BN) return SDF_E_SHAPE;
  const bool spike = d.sn_T > 0;
  if (!spike && (d.ldo % 4 || !sdf_aligned(d.out, 16) || (d.resid && !sdf_aligned(d.resid, 16)) ||
                 (d.bias && !sdf_aligned(d.bias, 16)) || (d.alpha && (!sdf_aligned(d.alpha, 16) || !sdf_aligned(d.beta, 16)))))
    return SDF_E_ALIGN;                                          // the fp32 epilogue moves 16 bytes per lane
  const int npos = spike ? 32 / d.sn_T : 0;
  P.tiles_m = (int)(spike ? (d.pos_count + 8 * npos - 1) / (8 * npos) : (d.M + BM - 1) / BM);
  P.tiles_n = d.N / BN;
  P.ntiles = P.tiles_m * P.tiles_n;
  plan_splitk(P, KC);
  const int nitems = P.ntiles * P.ksplit;
  // one workgroup per compute unit (the kernel fills a CU's registers and LDS).  With more than 256 items the grid is sized for
  // EQUAL rounds - 288 items: 144 workgroups x 2 items, not 256 of which 32 run a second item while 224 compute units idle:
  // same duration, and the other in-flight forwards' kernels get the compute units this launch does not need
  // A workgroup's two consumer groups alternate items, so a workgroup wants an EVEN number of items: with one item per workgroup
  // half of its matrix-pipe time is idle.  SDF_PP_PAIR=0: the round-2 rule (one item per workgroup up to 256 workgroups).
  const bool pair = !(sdf_sw(SW_PP_PAIR) && sdf_sw(SW_PP_PAIR)[0] == '0');
  int G;
  if (pair && nitems >= 16) {
    const int rounds = (nitems + 511) / 512;                     // items per workgroup = 2 * rounds
    G = (nitems + 2 * rounds - 1) / (2 * rounds);
  } else {
    const int rounds = (nitems + 255) / 256;
    G = (nitems + rounds - 1) / rounds;
  }
  return G;
}

int launch_spike_mm_pp(const GemmParams& Pin, bool conv, hipStream_t s) {
  GemmParams P = Pin;
  const SdfSpikeGemmDesc& d = P.d;
  const int G = pp_plan(P);
  if (G < 0) return G;
  dim3 grid((unsigned)G);
  int rc;
  if (conv)
    rc = d.nsplit == 1 ? launch_t<1, true>(P, grid, s) : (d.nsplit == 2 ? launch_t<2, true>(P, grid, s) : launch_t<3, true>(P, grid, s));
  else
    rc = d.nsplit == 1 ? launch_t<1, false>(P, grid, s) : (d.nsplit == 2 ? launch_t<2, false>(P, grid, s) : launch_t<3, false>(P, grid, s));
  if (rc) return rc;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  return launch_splitk_reduce(P, s);
}

int launch_spike_mm_pp_multi(const GemmParams* Ps, int n, hipStream_t s) {
  if (n < 2 || n > PP_MULTI_MAX) return SDF_E_SHAPE;
  GemmMulti M = {};
  M.n = n;
  int wgs = 0;
  // longest K first: workgroups start in grid order, and the tail of the launch should be the short classes
  int order[PP_MULTI_MAX] = {0, 1, 2, 3};
  for (int a = 0; a < n; ++a)
    for (int b = a + 1; b < n; ++b)
      if (Ps[order[b]].d.K > Ps[order[a]].d.K) { const int t = order[a]; order[a] = order[b]; order[b] = t; }
  for (int i = 0; i < n; ++i) {
    GemmParams P = Ps[order[i]];
    if (P.d.sn_T != 0) return SDF_E_SHAPE;
    const int G = pp_plan(P);
    if (G < 0) return G;
    if (P.ksplit != 1) return SDF_E_SHAPE;                       // (a split-K member needs its own reduce pass: the caller launches one by one)
    if (i == 0) {
      M.base = P;
    } else {                                                     // one family: everything but taps / weights / row map / K is shared
      const SdfSpikeGemmDesc &a = M.base.d, &b = P.d;
      const ConvGeom &u = M.base.cv, &v = P.cv;
      if (a.A != b.A || a.out != b.out || a.M != b.M || a.N != b.N || a.ldo != b.ldo || a.nsplit != b.nsplit || a.bias != b.bias ||
          a.alpha != b.alpha || a.beta != b.beta || a.resid != b.resid || a.out_rows != b.out_rows || a.zg_nH != b.zg_nH ||
          u.H != v.H || u.W != v.W || u.Cin != v.Cin || u.OH != v.OH || u.OW != v.OW || u.sy != v.sy || u.sx != v.sx ||
          M.base.tiles_m != P.tiles_m || M.base.tiles_n != P.tiles_n)
        return SDF_E_SHAPE;
    }
    M.Wp[i] = P.d.Wp; M.rowmap[i] = P.d.out_rowmap; M.K[i] = P.d.K; M.spc[i] = P.spc; M.KWc[i] = P.cv.KWc; M.kw_mul[i] = P.cv.kw_mul;
    M.acc_scale[i] = P.acc_scale;
    for (int j = 0; j < 3; ++j) { M.dy[i][j] = P.cv.dy[j]; M.dx[i][j] = P.cv.dx[j]; }
    M.first[i] = wgs;
    wgs += G;
  }
  for (int i = n; i <= PP_MULTI_MAX; ++i) M.first[i] = wgs;
  for (int i = n; i < PP_MULTI_MAX; ++i) { M.Wp[i] = M.Wp[0]; M.rowmap[i] = M.rowmap[0]; M.K[i] = M.K[0]; M.spc[i] = M.spc[0]; M.KWc[i] = M.KWc[0];
                                            M.kw_mul[i] = M.kw_mul[0]; M.acc_scale[i] = M.acc_scale[0]; }
  const dim3 grid((unsigned)wgs);
  switch (M.base.d.nsplit) {
    case 1: SDF_LAUNCH((spike_mm_pp_multi_kernel<1>), grid, dim3(768), 0, s, M); break;
    case 2: SDF_LAUNCH((spike_mm_pp_multi_kernel<2>), grid, dim3(768), 0, s, M); break;
    case 3: SDF_LAUNCH((spike_mm_pp_multi_kernel<3>), grid, dim3(768), 0, s, M); break;
    default: return SDF_E_SHAPE;
  }
  hipError_t e = hipGetLastError();
  return e != hipSuccess ? (int)e : 0;
}

}  // namespace sdfmm

#ifdef SDF_STAMP
extern "C" int sdf_debug_read_stamps_pp(unsigned long long* host16) {
  return (int)hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_sdf_stamp), 16 * sizeof(unsigned long long));
}
#endif
