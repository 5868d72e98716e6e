// Warp-specialised spike matrix multiply / implicit-GEMM convolution for gfx950.
//
//   out = epilogue( A (binary u8 spikes; a matrix, or an NHWC image batch read through im2col) x W^T )
//
// One persistent workgroup of 8 wavefronts per CU, split by ROLE so that every SIMD hosts exactly one MFMA wave
// and one load wave (the matrix pipe and the vector pipe of a SIMD run concurrently):
//
//   waves 0-3  consumers : own 64 rows x 96 columns of the 256 x 96 tile each (2 x 3 accumulators of
//                          v_mfma_f32_32x32x16_bf16), read fragments from LDS, expand the 1-byte spikes to bf16 in
//                          registers, multiply against up to three bf16 weight planes (hi/mid/lo split of the fp32
//                          weights: fp32-grade results at bf16 MFMA rate), then run the epilogue;
//   waves 4-7  producers : all address arithmetic (row decode, im2col taps, bounds), the global loads of the stage
//                          after next, and the LDS writes of the next stage.
//
// K advances in stages of 64 through a 2-deep LDS ring; ONE s_barrier per stage hands buffers over in both
// directions (producers fill buf[(q+1)%2] while consumers multiply buf[q%2]).  The stage stream runs across tile
// boundaries, so the producers are already filling the next tile while the consumers run an epilogue.
//
// Epilogues (consumer waves only, no workgroup barrier inside):
//   F32   : (+bias) -> fmaf(., alpha, beta) -> (+resid) -> fp32 store, optional output row scatter
//   SPIKE : fmaf(., alpha, beta) (+ positional term) -> LIF / IF / PSN over the T steps of each position, taken
//           straight from the accumulator slots a lane holds (slot = 16*rowblock + reg -> position = slot / T,
//           t = slot % T) -> 1-byte spikes, staged through a private LDS region to leave as 16-byte stores.
// LDS rows are padded (A 18-dword stride for ds_read_b64, W 36-dword stride for ds_read_b128): conflict-free.
// Compiled with -ffp-contract=off (the neuron arithmetic is the separately-rounded op sequence of neuron.hip).
#include "spike_mm.h"
#include <stdlib.h>

#ifdef SDF_STAMP
// diagnostic build only: per-role cycle accounting of workgroup 0 (never compiled into the product library)
__device__ unsigned long long g_sdf_stamp[16];
#define STAMP(var) var = __builtin_readcyclecounter()
#else
#define STAMP(var)
#endif

namespace sdfmm {
namespace {

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it would make the
// producers wait for the prefetch they have just issued and the consumers for their epilogue stores.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

constexpr int BM = 256, BN = 96, KC = 64;
constexpr int A_LD = KC + 8;                 // bytes per A row   (72 B  = 18 dwords)
constexpr int W_LD = KC + 8;                 // bf16 per W row    (144 B = 36 dwords)
constexpr int A_BYTES = BM * A_LD;           // 18432

template <int NSPLIT, int TT, bool CONV>
__global__ __launch_bounds__(512) void spike_mm_ws_kernel(GemmParams P) {
  constexpr bool SPIKE = TT > 0;
  constexpr int T = SPIKE ? TT : 1;
  constexpr int NPOS = 32 / T;                                   // positions per lane-half (2 row blocks = 32 slots)
  constexpr int W_BYTES = NSPLIT * BN * W_LD * 2;
  constexpr int BUF = A_BYTES + W_BYTES;
  constexpr int WCH = NSPLIT * BN * (KC / 8);                    // 16-byte chunks of a W stage
  constexpr int WIT = WCH / 256;                                 // 9 (3 planes) or 3 (1 plane) per producer lane
  static_assert(WCH % 256 == 0, "W stage must divide over the producer lanes");
  __shared__ __attribute__((aligned(16))) uint8_t smem[2 * BUF + (SPIKE ? 4 * 64 * BN : 16)];

  const SdfSpikeGemmDesc& d = P.d;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int K = d.K, N = d.N;
  const int nstages = P.spc;                                     // stages per work item (= all of K unless split-K)
  const int ksplit = P.ksplit;

  // contiguous range of work items of this workgroup; item = tile * ksplit + kchunk, tiles column-block-major
  // (t = cb * tiles_m + rt).  Every item streams `spc` stages; stages past K load zeros.
  const int G = gridDim.x, wg = blockIdx.x;
  const int nitems = P.ntiles * ksplit;
  const int base = nitems / G, rem = nitems % G;
  const int t_begin = wg * base + (wg < rem ? wg : rem);
  const int n_my = base + (wg < rem ? 1 : 0);
  if (n_my == 0) return;
  const int Q = n_my * nstages;                                  // stages this workgroup streams through

  // global row (or -1) of tile-row R of row-tile rt
  auto tile_row = [&](int rt, int R) -> int64_t {
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

  if (wave >= 4) {
    // =============================== PRODUCERS ===============================
    const int ptid = tid - 256;                                  // one A row per lane, 4 x 16 B per stage
    uint4 areg[4], wreg[WIT];
    int64_t a_off = -1;                                          // row base (bytes / scramble base)
    const int64_t zg_gstride = (int64_t)d.zg_T * d.zg_N1 * 32;
    // per-lane weight chunk coordinates: element offset inside the plane-major [p][N][K] array, minus n0*K and k0
    uint32_t w_off[WIT];
    int w_lds[WIT];
#pragma unroll
    for (int i = 0; i < WIT; ++i) {
      const int c = ptid + 256 * i;
      const int row = c >> 3, cc = c & 7;                        // row = p*96 + n
      const int p = row / BN, n = row - p * BN;
      w_off[i] = ((uint32_t)p * (uint32_t)N + (uint32_t)n) * (uint32_t)K + 8u * cc;
      w_lds[i] = A_BYTES + (row * W_LD + 8 * cc) * 2;
    }
    // stage iterator of the LOADER (no divisions in steady state)
    int ld_st = 0, ld_kc = t_begin % ksplit;
    int ld_cb = (t_begin / ksplit) / P.tiles_m, ld_rt = (t_begin / ksplit) - ld_cb * P.tiles_m;
    bool ld_new = true;
    uint32_t a_valid = 0;                                        // CONV: bit tap = input pixel of that tap is inside the image
    auto load = [&]() {
      const int k0 = (ld_kc * nstages + ld_st) * KC;
      if (ld_new) {                                              // new tile: decode this lane's row once
        ld_new = false;
        const int64_t g = tile_row(ld_rt, ptid);
        a_off = g < 0 ? -1 : g * d.lda;
        if (CONV) {
          a_valid = 0;
          if (g >= 0) {
            const ConvGeom& cv = P.cv;
            const uint32_t ohw = (uint32_t)(cv.OH * cv.OW);
            const uint32_t img = (uint32_t)g / ohw;
            const uint32_t r2 = (uint32_t)g - img * ohw;
            const uint32_t oy = r2 / (uint32_t)cv.OW, ox = r2 - oy * (uint32_t)cv.OW;
            const int iy0 = (int)oy * cv.sy, ix0 = (int)ox * cv.sx;
            a_off = (((int64_t)img * cv.H + iy0) * cv.W + ix0) * cv.Cin;     // byte offset of the un-shifted pixel
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
              const int ky = (tp * cv.kw_mul) >> 5, kx = tp - ky * cv.KWc;
              const int iy = iy0 + (ky == 0 ? cv.dy[0] : (ky == 1 ? cv.dy[1] : cv.dy[2]));
              const int ix = ix0 + (kx == 0 ? cv.dx[0] : (kx == 1 ? cv.dx[1] : cv.dx[2]));
              if ((unsigned)iy < (unsigned)cv.H && (unsigned)ix < (unsigned)cv.W) a_valid |= 1u << tp;
            }
          }
        } else if (!SPIKE && d.zg_nH > 0 && g >= 0) {
          const uint32_t bn = (uint32_t)(d.zg_B * d.zg_N1);
          const uint32_t zt = (uint32_t)g / bn;
          const uint32_t r2 = (uint32_t)g - zt * bn;
          const uint32_t zb = r2 / (uint32_t)d.zg_N1;
          const uint32_t zn = r2 - zb * (uint32_t)d.zg_N1;
          a_off = (((int64_t)zb * d.zg_nH * d.zg_T + zt) * d.zg_N1 + zn) * 32;
        }
      }
      if (CONV) {
        // wave-uniform: first tap of the stage, its channel offset, and the byte shift of up to 3 consecutive taps
        const ConvGeom& cv = P.cv;
        const int tap0 = k0 / cv.Cin, c0 = k0 - tap0 * cv.Cin;
        int tshift[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const int tp = tap0 + u;
          const int ky = (tp * cv.kw_mul) >> 5, kx = tp - ky * cv.KWc;
          const int ddy = ky == 0 ? cv.dy[0] : (ky == 1 ? cv.dy[1] : cv.dy[2]);
          const int ddx = kx == 0 ? cv.dx[0] : (kx == 1 ? cv.dx[1] : cv.dx[2]);
          tshift[u] = (ddy * cv.W + ddx) * cv.Cin;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int u = 0, c = c0 + 16 * j;                            // all wave-uniform
          if (c >= cv.Cin) { c -= cv.Cin; u = 1; }
          if (c >= cv.Cin) { c -= cv.Cin; u = 2; }
          areg[j] = make_uint4(0, 0, 0, 0);
          if (k0 + 16 * j < K && ((a_valid >> (tap0 + u)) & 1u))
            areg[j] = *reinterpret_cast<const uint4*>(d.A + (a_off + (u == 0 ? tshift[0] : (u == 1 ? tshift[1] : tshift[2])) + c));
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int k = k0 + 16 * j;
          areg[j] = make_uint4(0, 0, 0, 0);
          if (a_off >= 0 && k < K) {
            const uint8_t* src = (!SPIKE && d.zg_nH > 0) ? d.A + a_off + (k >> 5) * zg_gstride + (k & 31) : d.A + a_off + k;
            areg[j] = *reinterpret_cast<const uint4*>(src);
          }
        }
      }
      const uint32_t wbase = (uint32_t)(ld_cb * BN) * (uint32_t)K + (uint32_t)k0;     // wave-uniform
#pragma unroll
      for (int i = 0; i < WIT; ++i) {
        const int cc = (ptid + 256 * i) & 7;
        wreg[i] = make_uint4(0, 0, 0, 0);
        if (k0 + 8 * cc < K) wreg[i] = *reinterpret_cast<const uint4*>(d.Wp + (w_off[i] + wbase));
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
    auto store = [&](int buf) {
      uint8_t* B = smem + buf * BUF;
      uint2* ap = reinterpret_cast<uint2*>(B + ptid * A_LD);     // rows are 8-byte aligned
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ap[2 * j] = make_uint2(areg[j].x, areg[j].y);
        ap[2 * j + 1] = make_uint2(areg[j].z, areg[j].w);
      }
#pragma unroll
      for (int i = 0; i < WIT; ++i) *reinterpret_cast<uint4*>(B + w_lds[i]) = wreg[i];
    };

    load();
    store(0);
    if (Q > 1) load();
    lds_barrier();                                             // stage 0 is in LDS
#ifdef SDF_STAMP
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, s_store = 0, s_load = 0, s_bar = 0;
#endif
    for (int q = 0; q < Q; ++q) {
      STAMP(t0);
      if (q + 1 < Q) {
        store((q + 1) & 1);                                      // consumers left this buffer at the previous barrier
        STAMP(t1);
        if (q + 2 < Q) load();
      }
      STAMP(t2);
      lds_barrier();
      STAMP(t3);
#ifdef SDF_STAMP
      s_store += t1 - t0; s_load += t2 - t1; s_bar += t3 - t2;
#endif
    }
#ifdef SDF_STAMP
    if (blockIdx.x == 0 && tid == 256) { g_sdf_stamp[0] = s_store; g_sdf_stamp[1] = s_load; g_sdf_stamp[2] = s_bar; g_sdf_stamp[3] = Q; }
#endif
    return;
  }

  // =============================== CONSUMERS ===============================
  const int l31 = lane & 31, lh = lane >> 5;
  const float asc = P.acc_scale;
  f32x16 acc[2][3];
  const bool soft = d.soft_reset != 0;
  lds_barrier();                                               // matches the producers' "stage 0 is in LDS"
#ifdef SDF_STAMP
  unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0, s_mma = 0, s_epi = 0, s_cbar = 0;
#endif
  for (int q = 0; q < Q; ++q) {
    STAMP(c0);
    const int tl = q / nstages, st = q - tl * nstages;
    const uint8_t* A_s = smem + (q & 1) * BUF;
    const uint16_t* W_s = reinterpret_cast<const uint16_t*>(A_s + A_BYTES);
    if (st == 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    }
#pragma unroll
    for (int ks = 0; ks < KC / 16; ++ks) {
      bf16x8 a[2];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
        a[rb] = expand_spikes<NSPLIT>(*reinterpret_cast<const uint2*>(&A_s[(wave * 64 + rb * 32 + l31) * A_LD + ks * 16 + 8 * lh]));
#pragma unroll
      for (int nb = 0; nb < 3; ++nb) {
#pragma unroll
        for (int p = 0; p < NSPLIT; ++p) {
          const bf16x8 b = *reinterpret_cast<const bf16x8*>(&W_s[(p * BN + nb * 32 + l31) * W_LD + ks * 16 + 8 * lh]);
          acc[0][nb] = mma<NSPLIT>(a[0], b, acc[0][nb]);
          acc[1][nb] = mma<NSPLIT>(a[1], b, acc[1][nb]);
        }
      }
    }

    STAMP(c1);
    if (st == nstages - 1) {
      const int item = t_begin + tl;
      const int t = item / ksplit, kc = item - t * ksplit;
      const int cb = t / P.tiles_m, rt = t - cb * P.tiles_m;
      const int n0 = cb * BN;
      if (!SPIKE) {
        // Each lane ends up with 4 consecutive columns of one row (quad transpose) -> 16-byte loads / stores.
        // vmcnt counts loads AND stores in order on CDNA4, so a load placed between stores makes its s_waitcnt drain
        // the stores in front of it.  Hence two wave-uniform code paths: (1) no row map / residual: no loads at all,
        // the 24 stores stream out back to back; (2) otherwise ALL row-map and residual loads are issued before the
        // first store.  No per-lane "load or constant" selects anywhere (hipcc branches around those and waits).
        const int qd = l31 >> 2, ql = l31 & 3;
        const int mrow0 = rt * BM + wave * 64 + 4 * lh + ql;       // + rb*32 + 8*q4
        const bool has_map = d.out_rowmap != nullptr, has_res = d.resid != nullptr;
        float4 bs[3], al[3], be[3];
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) {
          const int n = n0 + nb * 32 + 4 * qd;
          bs[nb] = make_float4(0.f, 0.f, 0.f, 0.f);
          al[nb] = make_float4(1.f, 1.f, 1.f, 1.f);
          be[nb] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (d.bias) bs[nb] = *reinterpret_cast<const float4*>(d.bias + n);
          if (d.alpha) { al[nb] = *reinterpret_cast<const float4*>(d.alpha + n); be[nb] = *reinterpret_cast<const float4*>(d.beta + n); }
        }
        // Pin the waits for these loads HERE, in straight-line code.  hipcc sinks the epilogue arithmetic into the
        // exec-masked store blocks below; a first use inside such a block gets its own s_waitcnt vmcnt(0) in every
        // block (the wait is not known on the skip path), and that wait also drains the previous block's store.
#pragma unroll
        for (int nb = 0; nb < 3; ++nb)
          asm volatile("" :: "v"(bs[nb].x), "v"(bs[nb].w), "v"(al[nb].x), "v"(al[nb].w), "v"(be[nb].x), "v"(be[nb].w));
        auto finish = [&](int nb, int rb, int q4, float4 r) -> float4 {
          float v[4] = {acc[rb][nb][q4 * 4 + 0], acc[rb][nb][q4 * 4 + 1], acc[rb][nb][q4 * 4 + 2], acc[rb][nb][q4 * 4 + 3]};
          quad_transpose(v, ql);
          float4 o = make_float4(v[0] * asc, v[1] * asc, v[2] * asc, v[3] * asc);
          o.x += bs[nb].x; o.y += bs[nb].y; o.z += bs[nb].z; o.w += bs[nb].w;
          o.x = __builtin_fmaf(o.x, al[nb].x, be[nb].x); o.y = __builtin_fmaf(o.y, al[nb].y, be[nb].y);
          o.z = __builtin_fmaf(o.z, al[nb].z, be[nb].z); o.w = __builtin_fmaf(o.w, al[nb].w, be[nb].w);
          o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
          return o;
        };
        if (ksplit > 1) {
          // split-K: raw fp32 partial sums; bias / BN / residual / scatter happen in splitk_reduce_kernel
          float* pbase = P.partial + (int64_t)kc * d.M * N;
#pragma unroll
          for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
              const int m = mrow0 + rb * 32 + 8 * q4;
              float* op = pbase + (int64_t)m * N + n0 + 4 * qd;
#pragma unroll
              for (int nb = 0; nb < 3; ++nb) {
                float v[4] = {acc[rb][nb][q4 * 4 + 0], acc[rb][nb][q4 * 4 + 1], acc[rb][nb][q4 * 4 + 2], acc[rb][nb][q4 * 4 + 3]};
                quad_transpose(v, ql);
                if (m < (int)d.M) *reinterpret_cast<float4*>(op + nb * 32) = make_float4(v[0], v[1], v[2], v[3]);
              }
            }
        } else if (!has_map && !has_res) {
#pragma unroll
          for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
              const int m = mrow0 + rb * 32 + 8 * q4;
              float* op = d.out + (int64_t)m * d.ldo + n0 + 4 * qd;
#pragma unroll
              for (int nb = 0; nb < 3; ++nb) {
                const float4 o = finish(nb, rb, q4, make_float4(0.f, 0.f, 0.f, 0.f));
                if (m < (int)d.M) *reinterpret_cast<float4*>(op + nb * 32) = o;
              }
            }
        } else {
          int dst[2][4];
          unsigned okm = 0;
#pragma unroll
          for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
              const int m = mrow0 + rb * 32 + 8 * q4;
              const bool in = m < (int)d.M;
              dst[rb][q4] = in ? m : 0;
              if (in) okm |= 1u << (rb * 4 + q4);
            }
          if (has_map) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
              for (int q4 = 0; q4 < 4; ++q4) {
                const int r = d.out_rowmap[dst[rb][q4]];
                if (r < 0) okm &= ~(1u << (rb * 4 + q4));
                dst[rb][q4] = r >= 0 ? r : 0;
              }
          }
          // residual prefetch in two batches (one per row block, 48 VGPRs): two drain points instead of 24
#pragma unroll
          for (int rb = 0; rb < 2; ++rb) {
            float4 rs[4][3];
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
              for (int nb = 0; nb < 3; ++nb) rs[q4][nb] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (has_res) {
#pragma unroll
              for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
                for (int nb = 0; nb < 3; ++nb)
                  rs[q4][nb] = *reinterpret_cast<const float4*>(d.resid + (int64_t)dst[rb][q4] * d.ldo + n0 + nb * 32 + 4 * qd);
#pragma unroll
              for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
                for (int nb = 0; nb < 3; ++nb) asm volatile("" :: "v"(rs[q4][nb].x), "v"(rs[q4][nb].w));   // wait here, once
            }
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
              for (int nb = 0; nb < 3; ++nb) {
                const float4 o = finish(nb, rb, q4, rs[q4][nb]);
                if ((okm >> (rb * 4 + q4)) & 1u)
                  *reinterpret_cast<float4*>(d.out + (int64_t)dst[rb][q4] * d.ldo + n0 + nb * 32 + 4 * qd) = o;
              }
          }
        }
      } else {
        uint8_t* S_s = smem + 2 * BUF + wave * 64 * BN;          // private 64 x 96 byte staging tile of this wave
        const int64_t pos0 = (int64_t)rt * (8 * NPOS) + (wave * 2 + lh) * NPOS;
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) {
          const int n = n0 + nb * 32 + l31;
          const float al = d.alpha ? d.alpha[n] : 1.f;
          const float be = d.alpha ? d.beta[n] : 0.f;
#pragma unroll
          for (int pl = 0; pl < NPOS; ++pl) {
            const int64_t pos = pos0 + pl;
            float xs[T], sp[T];
#pragma unroll
            for (int t2 = 0; t2 < T; ++t2) {
              const int slot = pl * T + t2;
              xs[t2] = __builtin_fmaf(acc[slot >> 4][nb][slot & 15] * asc, al, be);   // al = 1, be = 0 when there is no BN
            }
            if (d.add) {                                         // wave-uniform; loads unconditional (clamped position)
              const int64_t pc = pos < d.pos_count ? pos : 0;
              const float* addp = d.add + ((uint32_t)pc % (uint32_t)d.add_prows) * (int64_t)N + n;
#pragma unroll
              for (int t2 = 0; t2 < T; ++t2) xs[t2] = xs[t2] + addp[(int64_t)t2 * d.add_prows * N];
            }
            if (d.sn_kind == SDF_PSN) {
#pragma unroll
              for (int t2 = 0; t2 < T; ++t2) {
                float hh = d.psn_b[t2];
#pragma unroll
                for (int k = 0; k < T; ++k) hh = __builtin_fmaf(d.psn_w[t2 * T + k], xs[k], hh);
                sp[t2] = hh >= 0.f ? 1.f : 0.f;
              }
            } else {
              lif_steps<T>(xs, sp, d.sn_kind, soft, d.v_reset, d.v_th, d.tau, P.inv_tau);
            }
#pragma unroll
            for (int t2 = 0; t2 < T; ++t2) {
              const int slot = pl * T + t2;
              const int rowl = (slot >> 4) * 32 + (slot & 3) + 8 * ((slot & 15) >> 2) + 4 * lh;
              S_s[rowl * BN + nb * 32 + l31] = (uint8_t)(sp[t2] != 0.f);
            }
          }
        }
        // the staging tile is private to this wave: its own LDS operations complete in order, no workgroup barrier
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int c = lane; c < 64 * (BN / 16); c += 64) {
          const int rowl = c / (BN / 16), c16 = c - rowl * (BN / 16);
          const int64_t g = tile_row(rt, wave * 64 + rowl);
          if (g >= 0)
            *reinterpret_cast<uint4*>(d.out_spike + g * N + n0 + 16 * c16) = *reinterpret_cast<const uint4*>(&S_s[rowl * BN + 16 * c16]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // reads done before the next tile overwrites S_s
        __builtin_amdgcn_wave_barrier();
      }
    }
    STAMP(c2);
    lds_barrier();                                             // hand buf[q&1] back, receive buf[(q+1)&1]
    STAMP(c3);
#ifdef SDF_STAMP
    s_mma += c1 - c0; s_epi += c2 - c1; s_cbar += c3 - c2;
#endif
  }
#ifdef SDF_STAMP
  if (blockIdx.x == 0 && tid == 0) { g_sdf_stamp[4] = s_mma; g_sdf_stamp[5] = s_epi; g_sdf_stamp[6] = s_cbar; g_sdf_stamp[7] = Q; }
#endif
}

// Second pass of split-K: out = epilogue( sum_k partial[k] ), chunks added in k order (deterministic).
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, int ksplit, int64_t M, int N,
                                                            float asc, const float* bias, const float* alpha, const float* beta,
                                                            const float* resid, const int* rowmap, float* out, int64_t ldo) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int nq = N / 4;
  if (q >= M * nq) return;
  const int64_t m = q / nq;
  const int n = (int)(q - m * nq) * 4;
  float4 a = *reinterpret_cast<const float4*>(partial + m * N + n);
  for (int k = 1; k < ksplit; ++k) {
    const float4 b = *reinterpret_cast<const float4*>(partial + ((int64_t)k * M + m) * N + n);
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  a.x *= asc; a.y *= asc; a.z *= asc; a.w *= asc;                // power of two: exact
  if (bias) { const float4 b = *reinterpret_cast<const float4*>(bias + n); a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
  if (alpha) {
    const float4 al = *reinterpret_cast<const float4*>(alpha + n), be = *reinterpret_cast<const float4*>(beta + n);
    a.x = __builtin_fmaf(a.x, al.x, be.x); a.y = __builtin_fmaf(a.y, al.y, be.y);
    a.z = __builtin_fmaf(a.z, al.z, be.z); a.w = __builtin_fmaf(a.w, al.w, be.w);
  }
  int64_t dst = m;
  if (rowmap) dst = rowmap[m];
  if (dst < 0) return;
  if (resid) { const float4 r = *reinterpret_cast<const float4*>(resid + dst * ldo + n); a.x += r.x; a.y += r.y; a.z += r.z; a.w += r.w; }
  *reinterpret_cast<float4*>(out + dst * ldo + n) = a;
}

template <int NSPLIT, bool CONV>
int launch_t(const GemmParams& P, dim3 grid, hipStream_t s) {
  switch (P.d.sn_T) {
    case 0: hipLaunchKernelGGL((spike_mm_ws_kernel<NSPLIT, 0, CONV>), grid, dim3(512), 0, s, P); return 0;
    case 2: if constexpr (!CONV) { hipLaunchKernelGGL((spike_mm_ws_kernel<NSPLIT, 2, CONV>), grid, dim3(512), 0, s, P); return 0; } return SDF_E_SHAPE;
    case 10: hipLaunchKernelGGL((spike_mm_ws_kernel<NSPLIT, 10, CONV>), grid, dim3(512), 0, s, P); return 0;
    default: return SDF_E_SHAPE;
  }
}

}  // namespace

// split-K when the tiles alone cannot occupy the chip (small M, large K): needs the fp32 epilogue and a
// caller-provided workspace of ksplit*M*N floats; partial sums are combined in k order by a second kernel
void plan_splitk(GemmParams& P, int kc) {
  const SdfSpikeGemmDesc& d = P.d;
  const int S = (d.K + kc - 1) / kc;
  P.ksplit = 1;
  P.spc = S;
  P.partial = nullptr;
  if (d.sn_T == 0 && P.ntiles <= 128 && S >= 4 && d.workspace) {
    // K chunks per tile: the count that minimises (rounds of the 256 workgroups) x (stages per item + a fixed per-item cost of
    // about three stages: prologue, epilogue, partial store).  Round 2 took ceil(256 / tiles), which on 40 tiles is 7 chunks =
    // 280 items = TWO rounds of 16 stages where 6 chunks = 240 items run ONE round of 18 (U-Net res-blocks: 49 -> 3x us).
    int kmax = S / 2 < 32 ? S / 2 : 32;
    if (const char* e = getenv("SDF_KSPLIT_MULT")) {                // tuning override: the old rule, oversubscribed
      int ks = (256 + P.ntiles - 1) / P.ntiles * (atoi(e) > 0 ? atoi(e) : 1);
      kmax = ks < kmax ? ks : kmax;
    }
    int best = 1;
    int64_t best_cost = -1;
    for (int ks = 1; ks <= kmax; ++ks) {
      if ((int64_t)ks * d.M * d.N * 4 > d.workspace_bytes) break;
      const int64_t rounds = ((int64_t)P.ntiles * ks + 255) / 256, spc = (S + ks - 1) / ks;
      const int64_t cost = rounds * (spc + 3);
      if (best_cost < 0 || cost < best_cost || (getenv("SDF_KSPLIT_MULT") && ks == kmax)) { best_cost = cost; best = ks; }
    }
    if (best > 1 && sdf_aligned(d.workspace, 16)) {
      P.ksplit = best;
      P.spc = (S + best - 1) / best;
      P.partial = reinterpret_cast<float*>(d.workspace);
    }
  }
}

int launch_splitk_reduce(const GemmParams& P, hipStream_t s) {
  if (P.ksplit <= 1) return 0;
  const SdfSpikeGemmDesc& d = P.d;
  const int64_t quads = d.M * (d.N / 4);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, P.partial, P.ksplit, d.M,
                     d.N, P.acc_scale, d.bias, d.alpha, d.beta, d.resid, d.out_rowmap, d.out, d.ldo);
  const hipError_t e = hipGetLastError();
  return e != hipSuccess ? (int)e : 0;
}

int launch_spike_mm_ws(const GemmParams& Pin, bool conv, hipStream_t s) {
  GemmParams P = Pin;
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
  const int G = nitems < 256 ? nitems : 256;
  dim3 grid((unsigned)G);
  int rc;
  if (conv)
    rc = d.nsplit == 1 ? launch_t<1, true>(P, grid, s) : launch_t<3, true>(P, grid, s);
  else
    rc = d.nsplit == 1 ? launch_t<1, false>(P, grid, s) : launch_t<3, false>(P, grid, s);
  if (rc) return rc;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  return launch_splitk_reduce(P, s);
}

}  // namespace sdfmm

#ifdef SDF_STAMP
extern "C" int sdf_debug_read_stamps(unsigned long long* host16) {
  return (int)hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_sdf_stamp), 16 * sizeof(unsigned long long));
}
#endif
