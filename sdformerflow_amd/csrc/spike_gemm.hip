// Spike GEMM for gfx950:  out = epilogue( A[M,K] (binary u8 spikes) x W[N,K]^T ).
//
// MFMA-tiled (v_mfma_f32_32x32x16_bf16, 64-lane wavefronts).  A binary operand is exact in bf16, so the fp32
// weights are carried as up to three bf16 planes (hi/mid/lo residual split) and the product is accumulated in
// fp32 by the matrix cores: fp32-grade results at bf16 MFMA rate.  The 1-byte spikes are expanded to bf16 in
// registers (v_perm + one multiply per pair).
//
// Structure: PERSISTENT workgroups (4 waves) walk a contiguous range of output tiles; tile = 128*RB rows x 32*NB
// columns, K in stages of 96.  Each wave owns 32*RB rows (RB MFMA row blocks) x NB column blocks.  The next
// stage - of this tile or of the NEXT tile - is prefetched global -> VGPR while the current one is multiplied, so
// the load latency, the index prologue and the epilogue of a tile overlap the neighbouring tiles' work instead
// of being paid once per workgroup launch.  Tiles are ordered column-block-major, so a workgroup keeps its
// weight tile in LDS across row tiles whenever K fits one stage (weight-stationary for the K = 96 layers, which
// carry most of the rows).  LDS rows are padded (A: 26-dword stride for ds_read_b64, W: 52-dword stride for
// ds_read_b128) so fragment reads are bank-conflict free.
//
// Two epilogues:
//   F32   : (+bias) -> fmaf(., alpha, beta) -> (+resid) -> fp32 store, optional output row scatter
//           (window_reverse + roll + crop as a row map) and the reference's head-scramble on the A side.
//   SPIKE : fmaf(., alpha, beta) (+ positional term) -> LIF / IF / PSN over the T time steps of each position
//           -> 1-byte spikes.  The tile's rows are laid out so that all T steps of a position sit in the 16*RB
//           accumulator slots one lane holds per column (slot = 16*rowblock + reg; position = slot / T,
//           t = slot % T), i.e. the recurrence runs in registers straight out of the MFMA accumulators and the
//           fp32 pre-activation never touches HBM.  Spikes are staged through LDS to leave as 16-byte stores.
// Compiled with -ffp-contract=off: the neuron arithmetic is the same separately-rounded op sequence as neuron.hip.
#include "wide_common.h"
#include "switches.h"
#include <stdlib.h>

namespace {
using namespace sdfmm;

constexpr int KC = 96;
constexpr int A_LD = KC + 8;          // bytes per A row in LDS (104 B = 26 dwords), written as 2 x 8 B
constexpr int W_LD = KC + 8;          // bf16 elements per W row in LDS (208 B = 52 dwords)

// WAVES = 8 (512 threads): two waves per SIMD share one weight tile, so one wave's address / epilogue VALU work
// overlaps the other's MFMAs.  WAVES = 4 (256 threads) with small tiles is for problems with few rows.
template <int NSPLIT, int NB, int RB, int TT, int WAVES, bool CONV>
__global__ __launch_bounds__(64 * WAVES) void spike_gemm_kernel(GemmParams P) {
  constexpr bool SPIKE = TT > 0;
  constexpr int T = SPIKE ? TT : 1;
  constexpr int NT = 64 * WAVES;                                 // threads per workgroup
  constexpr int BM = 32 * RB * WAVES, BN = 32 * NB, WR = 32 * RB;   // tile rows, tile cols, rows per wave
  constexpr int NPOS = (16 * RB) / T;                            // SPIKE: positions per lane-half
  constexpr int A_CH = BM * (KC / 16);                           // 16-byte chunks of an A stage
  constexpr int AIT = A_CH / NT;                                 // 3 * RB per thread
  constexpr int WCH = NSPLIT * BN * (KC / 8);                    // 16-byte chunks of a W stage
  constexpr int WIT = (WCH + NT - 1) / NT;
  static_assert(!SPIKE || NPOS >= 1, "T does not fit the accumulator slots of one lane");
  __shared__ __attribute__((aligned(16))) uint8_t smem[BM * A_LD + NSPLIT * BN * W_LD * 2];
  uint8_t* A_s = smem;
  uint16_t* W_s = reinterpret_cast<uint16_t*>(smem + BM * A_LD);

  const SdfSpikeGemmDesc& d = P.d;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int l31 = lane & 31, lh = lane >> 5;
  const int K = d.K, N = d.N;
  const int nstages = (K + KC - 1) / KC;

  // contiguous tile range of this workgroup (column-block-major tile order: t = cb * tiles_m + rt)
  const int G = gridDim.x, wg = blockIdx.x;
  const int base = P.ntiles / G, rem = P.ntiles % G;
  const int t_begin = wg * base + (wg < rem ? wg : rem);
  const int t_end = t_begin + base + (wg < rem ? 1 : 0);
  if (t_begin >= t_end) return;

  // ---- per-thread staging coordinates (tile independent) ----
  int a_row[AIT], a_c16[AIT];
#pragma unroll
  for (int i = 0; i < AIT; ++i) {
    const int c = tid + NT * i;
    a_row[i] = c / (KC / 16);
    a_c16[i] = c % (KC / 16);
  }

  // global row (or -1) of tile-row R of row-tile rt
  auto tile_row = [&](int rt, int R) -> int64_t {
    if (SPIKE) {
      const int w = R / WR, rloc = R % WR;
      const int rb = rloc >> 5, rr = rloc & 31;
      const int h = (rr >> 2) & 1, r = (rr & 3) + 4 * (rr >> 3);
      const int slot = rb * 16 + r;
      const int pl = slot / T, t = slot - pl * T;
      const int64_t pos = (int64_t)rt * (2 * WAVES * NPOS) + (w * 2 + h) * NPOS + pl;
      if (pl >= NPOS || pos >= d.pos_count) return -1;
      const uint32_t po = (uint32_t)pos / (uint32_t)d.pos_inner;         // positions < 2^31
      return (int64_t)po * d.pos_ostride + ((uint32_t)pos - po * (uint32_t)d.pos_inner) + (int64_t)t * d.t_stride;
    }
    const int64_t m = (int64_t)rt * BM + R;
    return m < d.M ? m : -1;
  };

  uint4 areg[AIT], wreg[WIT];
  int64_t a_off[AIT];                // source offset of each staged chunk's row for the tile being LOADED
  int a_iyx[CONV ? AIT : 1];         // CONV: (oy*sy) << 16 | (ox*sx) of that row
  const int64_t zg_gstride = (int64_t)d.zg_T * d.zg_N1 * 32;     // head-scramble: offset between channel groups
  auto set_rows = [&](int rt) {
#pragma unroll
    for (int i = 0; i < AIT; ++i) {
      const int64_t g = tile_row(rt, a_row[i]);
      a_off[i] = g < 0 ? -1 : g * d.lda;
      if (CONV) {
        if (g >= 0) {
          const ConvGeom& cv = P.cv;
          const uint32_t ohw = (uint32_t)(cv.OH * cv.OW);
          const uint32_t img = (uint32_t)g / ohw;
          const uint32_t r2 = (uint32_t)g - img * ohw;
          const uint32_t oy = r2 / (uint32_t)cv.OW, ox = r2 - oy * (uint32_t)cv.OW;
          a_off[i] = (int64_t)img * cv.H * cv.W;                       // pixel index of the image origin
          a_iyx[i] = (int)((oy * cv.sy) << 16 | (ox * cv.sx));         // un-offset input coordinates
        }
      } else if (!SPIKE && d.zg_nH > 0 && g >= 0) {
        // Z[t,b,n,g*32+d] = E_flat[((((b*nH+g)*T+t)*N1+n)*32+d]  ->  base(t,b,n) + g*(T*N1*32) + d
        const uint32_t bn = (uint32_t)(d.zg_B * d.zg_N1);
        const uint32_t zt = (uint32_t)g / bn;
        const uint32_t r2 = (uint32_t)g - zt * bn;
        const uint32_t zb = r2 / (uint32_t)d.zg_N1;
        const uint32_t zn = r2 - zb * (uint32_t)d.zg_N1;
        if (d.zg_rep > 0) {                                          // independent replicas: the scramble stays inside replica zb / zg_rep
          const uint32_t rr = zb / (uint32_t)d.zg_rep, zbr = zb - rr * (uint32_t)d.zg_rep;
          const uint32_t half = (uint32_t)d.zg_rep * (uint32_t)d.zg_N1 * (uint32_t)d.K;        // bytes of one attention step of a replica
          const uint32_t o = (((zbr * (uint32_t)d.zg_nH) * (uint32_t)d.zg_T + zt) * (uint32_t)d.zg_N1 + zn) * 32u;
          const uint32_t te = o / half;
          a_off[i] = ((int64_t)te * d.zg_B + (int64_t)rr * d.zg_rep) * d.zg_N1 * d.K + (o - te * half);
        } else
        a_off[i] = (((int64_t)zb * d.zg_nH * d.zg_T + zt) * d.zg_N1 + zn) * 32;
      }
    }
  };
  auto load_stage = [&](int cb, int k0, bool with_w) {
    // CONV: wave-uniform part of the (tap, channel) split of this stage; Cin >= 48 so a chunk wraps at most twice
    const int tap0 = CONV ? k0 / P.cv.Cin : 0, c0 = CONV ? k0 - tap0 * P.cv.Cin : 0;
#pragma unroll
    for (int i = 0; i < AIT; ++i) {
      const int k = k0 + 16 * a_c16[i];
      areg[i] = make_uint4(0, 0, 0, 0);
      if (a_off[i] >= 0 && k < K) {
        const uint8_t* src;
        if (CONV) {
          const ConvGeom& cv = P.cv;
          int tap = tap0, c = c0 + 16 * a_c16[i];
          if (c >= cv.Cin) { c -= cv.Cin; ++tap; }
          if (c >= cv.Cin) { c -= cv.Cin; ++tap; }
          const int ky = (tap * cv.kw_mul) >> 5, kx = tap - ky * cv.KWc;
          const int iy = (a_iyx[i] >> 16) + (ky == 0 ? cv.dy[0] : (ky == 1 ? cv.dy[1] : cv.dy[2]));
          const int ix = (a_iyx[i] & 0xffff) + (kx == 0 ? cv.dx[0] : (kx == 1 ? cv.dx[1] : cv.dx[2]));
          if ((unsigned)iy >= (unsigned)cv.H || (unsigned)ix >= (unsigned)cv.W) continue;
          src = d.A + (a_off[i] + (int64_t)iy * cv.W + ix) * cv.Cin + c;
        } else if (!SPIKE && d.zg_nH > 0) {
          src = d.A + a_off[i] + (k >> 5) * zg_gstride + (k & 31);
        } else {
          src = d.A + a_off[i] + k;
        }
        areg[i] = *reinterpret_cast<const uint4*>(src);
      }
    }
    if (with_w) {
      const int n0 = cb * BN;
#pragma unroll
      for (int i = 0; i < WIT; ++i) {
        const int c = tid + NT * i;
        wreg[i] = make_uint4(0, 0, 0, 0);
        if (c < WCH) {
          const int p = c / (BN * (KC / 8));
          const int r2 = c - p * (BN * (KC / 8));
          const int n = r2 / (KC / 8), c16 = r2 % (KC / 8);
          const int k = k0 + 8 * c16;
          if (n0 + n < N && k < K) wreg[i] = *reinterpret_cast<const uint4*>(d.Wp + ((int64_t)p * N + n0 + n) * K + k);
        }
      }
    }
  };
  auto store_stage = [&](bool with_w) {
#pragma unroll
    for (int i = 0; i < AIT; ++i) {
      uint2* dst = reinterpret_cast<uint2*>(&A_s[a_row[i] * A_LD + 16 * a_c16[i]]);   // rows are 8-byte aligned
      dst[0] = make_uint2(areg[i].x, areg[i].y);
      dst[1] = make_uint2(areg[i].z, areg[i].w);
    }
    if (with_w) {
#pragma unroll
      for (int i = 0; i < WIT; ++i) {
        const int c = tid + NT * i;
        if (c < WCH) {
          const int p = c / (BN * (KC / 8));
          const int r2 = c - p * (BN * (KC / 8));
          const int n = r2 / (KC / 8), c16 = r2 % (KC / 8);
          *reinterpret_cast<uint4*>(&W_s[(p * BN + n) * W_LD + 8 * c16]) = wreg[i];
        }
      }
    }
  };

  f32x16 acc[RB][NB];
  const float asc = P.acc_scale;                     // 1 / weight scale of the fp16 planes (a power of two), else 1
  const bool soft = d.soft_reset != 0;

  // ---- software pipeline over (tile, stage) ----
  int tl = t_begin, st = 0;                         // stage being computed
  int cb = tl / P.tiles_m, rt = tl - cb * P.tiles_m;
  int w_cb = -1;                                    // column block whose single-stage W tile sits in LDS
  set_rows(rt);
  bool cur_w = true;
  load_stage(cb, 0, true);
  while (true) {
    __syncthreads();                                // previous stage's LDS reads (and spike staging) are done
    store_stage(cur_w);
    if (nstages == 1) w_cb = cb;
    __syncthreads();
    // advance the loader to the next (tile, stage) and prefetch it
    int ntl = tl, nst = st + 1;
    if (nst == nstages) { nst = 0; ntl = tl + 1; }
    const bool more = ntl < t_end;
    int ncb = cb, nrt = rt;
    bool nxt_w = true;
    if (more) {
      if (ntl != tl) {
        ncb = ntl / P.tiles_m;
        nrt = ntl - ncb * P.tiles_m;
        set_rows(nrt);
        nxt_w = !(nstages == 1 && ncb == w_cb);
      }
      load_stage(ncb, nst * KC, nxt_w);
    }
    if (st == 0) {
#pragma unroll
      for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    }
#pragma unroll
    for (int ks = 0; ks < KC / 16; ++ks) {
      bf16x8 a[RB];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
        a[rb] = expand_spikes<NSPLIT>(*reinterpret_cast<const uint2*>(&A_s[(wave * WR + rb * 32 + l31) * A_LD + ks * 16 + 8 * lh]));
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
        for (int p = 0; p < NSPLIT; ++p) {
          const bf16x8 b = *reinterpret_cast<const bf16x8*>(&W_s[(p * BN + nb * 32 + l31) * W_LD + ks * 16 + 8 * lh]);
#pragma unroll
          for (int rb = 0; rb < RB; ++rb) acc[rb][nb] = mma<NSPLIT>(a[rb], b, acc[rb][nb]);
        }
      }
    }

    if (st == nstages - 1) {
      const int n0 = cb * BN;
      if (!SPIKE) {
        // ---- fp32 epilogue ----
        // Quad transpose -> each lane owns 4 consecutive columns of one row -> 16-byte loads / stores.  Two wave-uniform
        // paths (vmcnt counts loads AND stores in order on CDNA4): without row map / residual there is no load at all
        // and the stores stream; otherwise every row-map / residual load of a row block is issued before its first
        // store.  Parameter loads are pinned in straight-line code; no per-lane "load or constant" selects.
        const int qd = l31 >> 2, ql = l31 & 3;
        const int mrow0 = rt * BM + wave * WR + 4 * lh + ql;       // + rb*32 + 8*q4
        const bool has_map = d.out_rowmap != nullptr, has_res = d.resid != nullptr;
        float4 bs[NB], al[NB], be[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const int n = n0 + nb * 32 + 4 * qd;                       // N % 32 == 0: a column block is whole or absent
          const int nc = n < N ? n : 0;
          bs[nb] = make_float4(0.f, 0.f, 0.f, 0.f);
          al[nb] = make_float4(1.f, 1.f, 1.f, 1.f);
          be[nb] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (d.bias) bs[nb] = *reinterpret_cast<const float4*>(d.bias + nc);
          if (d.alpha) { al[nb] = *reinterpret_cast<const float4*>(d.alpha + nc); be[nb] = *reinterpret_cast<const float4*>(d.beta + nc); }
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
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
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          int dst[4];
          unsigned okm = 0;
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const int m = mrow0 + rb * 32 + 8 * q4;
            const bool in = m < (int)d.M;
            dst[q4] = in ? m : 0;
            if (in) okm |= 1u << q4;
          }
          if (has_map) {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
              const int r = d.out_rowmap[dst[q4]];
              if (r < 0) okm &= ~(1u << q4);
              dst[q4] = r >= 0 ? r : 0;
            }
          }
          float4 rs[4][NB];
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) rs[q4][nb] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (has_res) {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
              for (int nb = 0; nb < NB; ++nb) {
                const int n = n0 + nb * 32 + 4 * qd;
                rs[q4][nb] = *reinterpret_cast<const float4*>(d.resid + (int64_t)dst[q4] * d.ldo + (n < N ? n : 0));
              }
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
              for (int nb = 0; nb < NB; ++nb) asm volatile("" :: "v"(rs[q4][nb].x), "v"(rs[q4][nb].w));
          }
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
              const int n = n0 + nb * 32 + 4 * qd;
              const float4 o = finish(nb, rb, q4, rs[q4][nb]);
              if (((okm >> q4) & 1u) && n < N) *reinterpret_cast<float4*>(d.out + (int64_t)dst[q4] * d.ldo + n) = o;
            }
        }
      } else {
        // ---- spike epilogue: BN (+add) -> neuron over T -> bytes, staged through LDS ----
        __syncthreads();                                     // every wave is done reading A_s for this stage
        uint8_t* S_s = smem + wave * WR * BN;                // this wave's WR x BN byte tile (aliases A_s)
        const int64_t pos0 = (int64_t)rt * (2 * WAVES * NPOS) + (wave * 2 + lh) * NPOS;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const int n = n0 + nb * 32 + l31;
          const bool ncol = n < N;
          const float al = (d.alpha && ncol) ? d.alpha[n] : 1.f;
          const float be = (d.alpha && ncol) ? d.beta[n] : 0.f;
#pragma unroll
          for (int pl = 0; pl < NPOS; ++pl) {
            const int64_t pos = pos0 + pl;
            float xs[T], sp[T];
#pragma unroll
            for (int t = 0; t < T; ++t) {
              const int slot = pl * T + t;                   // compile-time after unrolling
              xs[t] = __builtin_fmaf(acc[slot >> 4][nb][slot & 15] * asc, al, be);      // al = 1, be = 0 without BN
            }
            if (d.add) {                                     // wave-uniform; loads unconditional (clamped position / column)
              const int64_t pc = pos < d.pos_count ? pos : 0;
              const float* addp = d.add + ((uint32_t)pc % (uint32_t)d.add_prows) * (int64_t)N + (ncol ? n : 0);
#pragma unroll
              for (int t = 0; t < T; ++t) xs[t] = xs[t] + addp[(int64_t)t * d.add_prows * N];
            }
            if (d.sn_kind == SDF_PSN) {
#pragma unroll
              for (int t = 0; t < T; ++t) {
                float hh = d.psn_b[t];
#pragma unroll
                for (int k = 0; k < T; ++k) hh = __builtin_fmaf(d.psn_w[t * T + k], xs[k], hh);
                sp[t] = hh >= 0.f ? 1.f : 0.f;
              }
            } else {
              lif_steps<T>(xs, sp, d.sn_kind, soft, d.v_reset, d.v_th, d.tau, P.inv_tau);
            }
#pragma unroll
            for (int t = 0; t < T; ++t) {
              const int slot = pl * T + t;
              const int rowl = (slot >> 4) * 32 + (slot & 3) + 8 * ((slot & 15) >> 2) + 4 * lh;
              S_s[rowl * BN + nb * 32 + l31] = (uint8_t)(sp[t] != 0.f);
            }
          }
        }
        __syncthreads();                                     // spike bytes of the whole tile are in LDS
        constexpr int CPR = BN / 16;                         // 16-byte chunks per row
#pragma unroll
        for (int c = lane; c < WR * CPR; c += 64) {
          const int rowl = c / CPR, c16 = c - rowl * CPR;
          const int64_t g = tile_row(rt, wave * WR + rowl);
          const int n = n0 + 16 * c16;
          if (g >= 0 && n < N)
            *reinterpret_cast<uint4*>(d.out_spike + g * N + n) = *reinterpret_cast<const uint4*>(&S_s[rowl * BN + 16 * c16]);
        }
      }
    }
    if (!more) break;
    tl = ntl; st = nst; cb = ncb; rt = nrt; cur_w = nxt_w;
  }
}

__global__ __launch_bounds__(256) void split_weight_kernel(const float* __restrict__ W, uint16_t* __restrict__ planes,
                                                           int64_t n, int nsplit) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float r = W[i];
  for (int p = 0; p < nsplit; ++p) {
    uint32_t u = __float_as_uint(r);                       // round-to-nearest-even fp32 -> bf16 (finite weights)
    uint32_t h = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
    planes[(int64_t)p * n + i] = (uint16_t)h;
    r = r - __uint_as_float(h << 16);                      // exact: the residual fits fp32
  }
}

// W * scale (fp32, n elements; scale a power of two) -> two fp16 planes hi + lo (round-to-nearest-even residual split)
__global__ __launch_bounds__(256) void split_weight_f16_kernel(const float* __restrict__ W, uint16_t* __restrict__ planes,
                                                               int64_t n, float scale) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float w = W[i] * scale;                            // exact
  const _Float16 hi = (_Float16)w;                         // RNE
  const _Float16 lo = (_Float16)(w - (float)hi);           // the residual is exact in fp32
  planes[i] = __builtin_bit_cast(uint16_t, hi);
  planes[n + i] = __builtin_bit_cast(uint16_t, lo);
}

template <int NSPLIT, int NB, int RB, int WAVES, bool CONV = false>
int launch(const GemmParams& P, dim3 grid, hipStream_t s) {
#define SDF_GEMM_T(TT)                                                                                      \
  case TT:                                                                                                  \
    if constexpr (TT == 0 || (16 * RB) / (TT ? TT : 1) >= 1) {                                              \
      SDF_LAUNCH((spike_gemm_kernel<NSPLIT, NB, RB, TT, WAVES, CONV>), grid, dim3(64 * WAVES), 0, s, P);  \
      return 0;                                                                                             \
    }                                                                                                       \
    return SDF_E_SHAPE;
  switch (P.d.sn_T) {
    SDF_GEMM_T(0) SDF_GEMM_T(2) SDF_GEMM_T(4) SDF_GEMM_T(5) SDF_GEMM_T(10) SDF_GEMM_T(20)
    default: return SDF_E_SHAPE;
  }
#undef SDF_GEMM_T
}

// configurations built: big = 8 waves x (RB=2, NB=3) [512 x 96 tiles]; mid = 8 waves x (RB=1, NB=3) [256 x 96];
// small = 4 waves x (RB=2 | 1, NB=1) [256|128 x 32]
template <int NSPLIT>
int launch_cfg(const GemmParams& P, int cfg, dim3 grid, hipStream_t s) {
  switch (cfg) {
    case 0: return launch<NSPLIT, 3, 2, 8>(P, grid, s);
    case 1: return launch<NSPLIT, 3, 1, 8>(P, grid, s);
    case 2: return launch<NSPLIT, 1, 2, 4>(P, grid, s);
    default: return launch<NSPLIT, 1, 1, 4>(P, grid, s);
  }
}

// acc_scale: only the fp16 planes carry a weight scale; it must be a power of two (exact rescaling of the accumulator)
bool sdf_scale_ok(const SdfSpikeGemmDesc* d) {
  if (d->nsplit != 2) return d->acc_scale == 0.f || d->acc_scale == 1.f;
  int ex;
  return d->acc_scale > 0.f && frexpf(d->acc_scale, &ex) == 0.5f;
}
float sdf_acc_scale(const SdfSpikeGemmDesc* d) { return d->nsplit == 2 ? d->acc_scale : 1.f; }

}  // namespace

extern "C" int sdf_spike_gemm_fwd(const SdfSpikeGemmDesc* d, void* stream) {
  if (!d) return SDF_E_NULL;
  if (!d->A || !d->Wp) return SDF_E_NULL;
  const bool spike = d->sn_T > 0;
  if (spike ? !d->out_spike : !d->out) return SDF_E_NULL;
  if (d->M < 1 || d->M >= (1LL << 31) || d->N < 32 || d->K < 32 || d->K % 32 || d->N % 32) return SDF_E_SHAPE;
  if (d->nsplit == SDF_PLANES_I8X3_TILED) {                      // digit planes in fragment order: few rows against many weights, fp32 out
    if (!d->col_scale) return SDF_E_NULL;
    if (d->alpha && !d->beta) return SDF_E_NULL;
    GemmParams Q;
    Q.d = *d;
    return smallm_gemm_supports(Q) ? launch_smallm_gemm(Q, sdf_stream(stream)) : SDF_E_SHAPE;
  }
  if (d->nsplit == SDF_PLANES_I8X3) {
    // row-major digit planes [3][N][K]: a plain product with the fp32 epilogue on the weight-resident row-loop kernel (ms_res.hip) - the
    // stacked-tap products of the middle decoder levels (reference Spiking_modules.py:461-474 written as one GEMM + col2im): rows are
    // walked as 10 "steps" x M / 10 "positions" (any order serves the fp32 form), the shortcut, if any, is the output buffer itself
    if (!d->col_scale) return SDF_E_NULL;
    if (d->alpha && !d->beta) return SDF_E_NULL;
    if (spike || d->M % 10 || d->K % 16 || d->K > 1024 || d->lda != d->K || d->ldo < d->N || d->out_rowmap || d->add || d->zg_nH) return SDF_E_SHAPE;
    if (d->resid && d->resid != d->out) return SDF_E_SHAPE;
    if (d->M * (int64_t)d->K >= (1LL << 31) || d->M * (int64_t)d->ldo * 4 >= (1LL << 31) || (int64_t)d->N * d->K * 3 >= (1LL << 31)) return SDF_E_SHAPE;
    if (!sdf_aligned(d->A, 16) || !sdf_aligned(d->Wp, 16) || !sdf_aligned(d->out, 16)) return SDF_E_ALIGN;
    WidePmParams P = {};
    P.A = d->A; P.W = reinterpret_cast<const int8_t*>(d->Wp); P.cscale = d->col_scale; P.N = d->N; P.K = d->K;
    P.HW = (int)(d->M / 10); P.P = d->M / 10;
    P.bias = d->bias; P.alpha = d->alpha; P.beta = d->beta; P.x = d->out; P.ldo = (int)d->ldo; P.no_resid = d->resid ? 0 : 1;
    P.res_stage = 1;
    if (!res_pm_takes(P, 10, 2)) return SDF_E_SHAPE;
    return launch_res_pm(P, 10, 2, sdf_stream(stream));
  }
  if (d->nsplit < 1 || d->nsplit > 3) return SDF_E_DTYPE;         // 1 = bf16, 2 = fp16 hi/lo (scaled), 3 = bf16 hi/mid/lo
  if (!sdf_scale_ok(d)) return SDF_E_DTYPE;
  if (d->alpha && !d->beta) return SDF_E_NULL;
  if (d->lda % 16) return SDF_E_SHAPE;
  GemmParams P;
  P.d = *d;
  P.inv_tau = 0.f;
  P.acc_scale = sdf_acc_scale(d);
  if (spike) {
    if (d->pos_count < 1 || d->pos_inner < 1 || d->pos_count * d->sn_T != d->M) return SDF_E_SHAPE;
    if (d->sn_kind != SDF_LIF && d->sn_kind != SDF_PSN && d->sn_kind != SDF_IF) return SDF_E_DTYPE;
    if (d->sn_kind == SDF_PSN && (!d->psn_w || !d->psn_b)) return SDF_E_NULL;
    if (!sdf_tau_ok(d->sn_kind, d->tau)) return SDF_E_SHAPE;
    if (d->add && d->add_prows < 1) return SDF_E_SHAPE;
    if (d->zg_nH > 0 || d->out_rowmap || d->resid || d->bias) return SDF_E_SHAPE;   // F32-only features
    if (!sdf_aligned(d->out_spike, 16) || d->N % 16) return SDF_E_ALIGN;
    P.inv_tau = sdf_inv_tau(d->sn_kind, d->tau);
  } else if (d->zg_nH > 0) {
    if (d->K != d->zg_nH * 32 || d->zg_T < 1 || d->zg_B < 1 || d->zg_N1 < 1) return SDF_E_SHAPE;
    if ((int64_t)d->zg_T * d->zg_B * d->zg_N1 != d->M) return SDF_E_SHAPE;
    if (d->zg_rep < 0 || (d->zg_rep > 0 && (d->zg_B % d->zg_rep || d->zg_rep % d->zg_T))) return SDF_E_SHAPE;
  }
  if (!sdf_aligned(d->A, 16) || !sdf_aligned(d->Wp, 16)) return SDF_E_ALIGN;
  if (!spike && (d->ldo % 4 || !sdf_aligned(d->out, 16) || (d->resid && !sdf_aligned(d->resid, 16)) ||
                 (d->bias && !sdf_aligned(d->bias, 16)) || (d->alpha && (!sdf_aligned(d->alpha, 16) || !sdf_aligned(d->beta, 16)))))
    return SDF_E_ALIGN;                                          // the fp32 epilogue moves 16 bytes per lane

  // ---- tile configuration ----
  // cfg 0: 512x96 (8 waves, 2 row blocks)   cfg 1: 256x96 (8 waves)   cfg 2: 256x32 (4 waves)   cfg 3: 128x32 (4 waves)
  // 96-wide tiles reuse every A fragment 3x; the fused neuron needs 2 row blocks when T > 8 (16*RB slots per lane).
  static const int CFG_NB[4] = {3, 3, 1, 1}, CFG_RB[4] = {2, 1, 2, 1}, CFG_WAVES[4] = {8, 8, 4, 4};
  auto ok = [&](int c) { return (CFG_NB[c] == 1 || d->N % 96 == 0) && (!spike || (16 * CFG_RB[c]) / d->sn_T >= 1); };
  auto ntiles_for = [&](int c) -> int64_t {
    const int64_t per = spike ? 2 * CFG_WAVES[c] * ((16 * CFG_RB[c]) / d->sn_T) : 32 * CFG_RB[c] * CFG_WAVES[c];
    const int64_t rows = spike ? d->pos_count : d->M;
    return ((rows + per - 1) / per) * ((d->N + 32 * CFG_NB[c] - 1) / (32 * CFG_NB[c]));
  };
  // Measured on MI355X (tools/gemm_microbench.py, profiles/r1_gemm_microbench.txt): these layers have K = 96..3072
  // and are bound by the per-element epilogue / addressing VALU work, not by the MFMAs, so the smallest tile
  // (most waves in flight) wins for the fp32 and T' = 2 epilogues; the T >= 5 neuron epilogue needs the 96-wide
  // 8-wave tile to amortise its staging.  (ntiles_for() is kept for the tuning override.)
  int cfg = (spike && d->sn_T >= 5) ? (ok(1) ? 1 : 2) : 3;
  if (!spike) {
    // fp32 epilogue: 256 x 32 or 128 x 32 tiles, whichever needs less time in whole rounds of the 768 resident workgroups
    // (a 256-row tile costs two 128-row ones); ties go to the SMALLER tile since round 6: with many rows (10 samples per launch
    // sequence: 712 800 rows of the stage-0 projection) the count always ties and the 256-row tile measured 44.5 us per sample
    // against 29.5 (gpurun_out -> profiles/r6_launch_table_R4.txt / _R10.txt; same-box A/B +0.4 % of the headline)
    const int64_t r2 = (ntiles_for(2) + 767) / 768 * 2, r3 = (ntiles_for(3) + 767) / 768;
    cfg = r2 < r3 ? 2 : 3;
  }
  if (!ok(cfg)) cfg = ok(2) ? 2 : 0;
  if (const char* e = sdf_sw(SW_GEMM_CFG)) {                  // tuning override: 0..3
    const int c = e[0] - '0';
    if (c >= 0 && c < 4 && ok(c)) cfg = c;
  }
  // Ping-pong 256 x 96 kernel (spike_mm_pp.hip) where it measured faster than the small-tile kernel on the model's
  // layers (tools/gemm_shapes.py, profiles/r1_gemm_shapes.txt): the fused T = 10 neuron epilogue (the MLP's fc1 - long
  // epilogues that the other consumer group hides) and fp32 epilogues with a long K loop over many rows, a very
  // long one, or many column tiles per row tile (the decoders' stacked tap matrices).  SDF_GEMM_WS: 0 = never, 2 = whenever legal.
  {
    const char* e = sdf_sw(SW_GEMM_WS);
    const bool legal = d->N % 96 == 0 && spike_mm_pp_supports(P, false);
    bool use_pp = legal && ((spike && (d->sn_T == 10 || d->sn_T == 20)) ||
                           (!spike && ((d->K >= 384 && d->M >= 32768) || d->K >= 2048 || (d->K >= 384 && d->N >= 864))));
    if (e && e[0] == '0') use_pp = false;
    if (e && e[0] == '2') use_pp = legal;
    if (use_pp) return launch_spike_mm_pp(P, false, sdf_stream(stream));
  }
  if (!ok(cfg)) return SDF_E_SHAPE;
  const int nb = CFG_NB[cfg], rb = CFG_RB[cfg], waves = CFG_WAVES[cfg];
  const int npos = spike ? (16 * rb) / d->sn_T : 0;
  P.tiles_m = (int)(spike ? (d->pos_count + 2 * waves * npos - 1) / (2 * waves * npos)
                          : (d->M + 32 * rb * waves - 1) / (32 * rb * waves));
  P.tiles_n = (d->N + 32 * nb - 1) / (32 * nb);
  P.ntiles = P.tiles_m * P.tiles_n;
  const size_t lds = (size_t)32 * rb * waves * A_LD + (size_t)d->nsplit * 32 * nb * W_LD * 2;
  int wg_per_cu = lds > 80 * 1024 ? 1 : (lds > 53 * 1024 ? 2 : 3);   // more resident workgroups measured no faster (profiles/r1_gemm_shapes.txt)
  if (const char* e = sdf_sw(SW_GEMM_WGS)) { const int v = atoi(e); if (v >= 1 && v <= 8) wg_per_cu = v; }   // tuning override
  const int G = P.ntiles < 256 * wg_per_cu ? P.ntiles : 256 * wg_per_cu;
  dim3 grid((unsigned)G);
  hipStream_t s = sdf_stream(stream);
  const int rc = d->nsplit == 1 ? launch_cfg<1>(P, cfg, grid, s) : (d->nsplit == 2 ? launch_cfg<2>(P, cfg, grid, s) : launch_cfg<3>(P, cfg, grid, s));
  if (rc) return rc;
  SDF_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
// Implicit-GEMM convolution on spike images (NHWC u8): the same persistent MFMA kernel with an im2col loader.
// argument checks of sdf_spike_conv2d_fwd + the kernel-side parameter block (no launch)
static int conv2d_build(const SdfSpikeConvDesc* c, GemmParams& P, bool& i8x3, bool& tiled) {
  if (!c) return SDF_E_NULL;
  const SdfSpikeGemmDesc* d = &c->g;
  if (!d->A || !d->Wp) return SDF_E_NULL;
  const bool spike = d->sn_T > 0;
  if (spike ? !d->out_spike : !d->out) return SDF_E_NULL;
  if (c->H < 1 || c->W < 1 || c->H > 32767 || c->W > 32767 || c->OH < 1 || c->OW < 1) return SDF_E_SHAPE;
  if (c->Cin < 48 || c->Cin % 16 || c->KH < 1 || c->KH > 3 || c->KW < 1 || c->KW > 3 || c->sy < 1 || c->sx < 1) return SDF_E_SHAPE;
  if (d->K != c->KH * c->KW * c->Cin || d->N % 32 || d->M % ((int64_t)c->OH * c->OW) || d->M >= (1LL << 31)) return SDF_E_SHAPE;
  if ((d->M / ((int64_t)c->OH * c->OW)) * c->H * c->W >= (1LL << 31)) return SDF_E_SHAPE;
  tiled = d->nsplit == SDF_PLANES_I8X3_TILED;          // digit planes in fragment order: only the small-M kernel reads them
  i8x3 = d->nsplit == SDF_PLANES_I8X3 || tiled;       // int8 digit planes: the weight-resident / wide / small-M kernels
  if (!i8x3 && (d->nsplit < 1 || d->nsplit > 3)) return SDF_E_DTYPE;
  if (!i8x3 && !sdf_scale_ok(d)) return SDF_E_DTYPE;
  if (i8x3 && !d->col_scale) return SDF_E_NULL;
  if (d->alpha && !d->beta) return SDF_E_NULL;
  if (d->zg_nH > 0) return SDF_E_SHAPE;
  P.d = *d;
  P.d.lda = 0;
  P.inv_tau = 0.f;
  P.acc_scale = i8x3 ? 1.f : sdf_acc_scale(d);
  if (spike) {
    if ((d->sn_T != 10 && !(i8x3 && (d->sn_T == 5 || d->sn_T == 20))) || d->pos_count < 1 || d->pos_inner < 1 ||
        d->pos_count * d->sn_T != d->M) return SDF_E_SHAPE;          // streaming kernels: T = 10; the digit kernel rolls its time loop
    if (d->sn_kind != SDF_LIF && d->sn_kind != SDF_PSN && d->sn_kind != SDF_IF) return SDF_E_DTYPE;
    if (d->sn_kind == SDF_PSN && (!d->psn_w || !d->psn_b)) return SDF_E_NULL;
    if (!sdf_tau_ok(d->sn_kind, d->tau)) return SDF_E_SHAPE;
    if (d->out_rowmap || d->bias || d->add) return SDF_E_SHAPE;
    if (d->resid && !d->out) return SDF_E_NULL;                   // a residual only exists for the membrane output
    if (d->out && (d->ldo < d->N || !sdf_aligned(d->out, 4))) return SDF_E_SHAPE;
    if (!sdf_aligned(d->out_spike, 16)) return SDF_E_ALIGN;
    P.inv_tau = sdf_inv_tau(d->sn_kind, d->tau);
  }
  if (!sdf_aligned(d->A, 16) || !sdf_aligned(d->Wp, 16)) return SDF_E_ALIGN;
  ConvGeom& cv = P.cv;
  cv.H = c->H; cv.W = c->W; cv.Cin = c->Cin; cv.OH = c->OH; cv.OW = c->OW; cv.sy = c->sy; cv.sx = c->sx;
  cv.KWc = c->KW;
  cv.kw_mul = c->KW == 1 ? 32 : (c->KW == 2 ? 16 : 11);
  for (int i = 0; i < 3; ++i) { cv.dy[i] = c->dy[i]; cv.dx[i] = c->dx[i]; }
  return 0;
}

// the stride-2 3x3 transposed convolution as one product over the 2 x 2 input neighbourhood (header; csrc/ms_res.hip, AM = 3)
extern "C" int sdf_spike_deconv3x3s2_fwd(const SdfSpikeDeconvDesc* d, void* stream) {
  if (!d) return SDF_E_NULL;
  if (!d->spikes || !d->digits || !d->cscale || !d->out) return SDF_E_NULL;
  if ((d->alpha != nullptr) != (d->beta != nullptr)) return SDF_E_NULL;
  if (d->imgs < 1 || d->H < 1 || d->W < 1 || (d->T != 10 && d->T != 20) || d->imgs % d->T) return SDF_E_SHAPE;
  if (d->Cin < 16 || d->Cin % 16 || 4 * d->Cin > 1024 || d->Cout < 8 || d->Cout % 8) return SDF_E_SHAPE;
  const int64_t rows = (int64_t)d->imgs * d->H * d->W;
  if (rows * d->Cin >= (1LL << 31) || rows * 4 * d->Cout * 4 >= (1LL << 31)) return SDF_E_SHAPE;      // 32-bit buffer offsets
  if (!sdf_aligned(d->spikes, 16) || !sdf_aligned(d->digits, 16) || !sdf_aligned(d->out, 16)) return SDF_E_ALIGN;
  // the decoder's last level (208 channels, tens of thousands of pixels): halo tiles in LDS, weights resident (spike_deconv_wres.hip);
  // other shapes: the row-loop kernel's form of the same product (ms_res.hip, AM = 3)
  if (spike_deconv_wres_supports(d->imgs, d->H, d->W, d->Cin, d->Cout))
    return launch_spike_deconv_wres(d->spikes, d->digits, d->cscale, d->alpha, d->beta, d->out, d->imgs, d->H, d->W, d->Cout, sdf_stream(stream));
  WidePmParams P = {};
  P.A = d->spikes; P.W = d->digits; P.cscale = d->cscale; P.N = 4 * d->Cout; P.K = 4 * d->Cin;
  P.HW = d->H * d->W; P.P = (int64_t)(d->imgs / d->T) * P.HW;
  P.alpha = d->alpha; P.beta = d->beta; P.x = d->out; P.ldo = d->Cout; P.no_resid = 1;
  P.cv_H = d->H; P.cv_W = d->W; P.cv_Cin = d->Cin; P.dc_cout = d->Cout;
  P.res_stage = 1;
  if (!res_pm_takes(P, d->T, 2)) return SDF_E_SHAPE;
  return launch_res_pm(P, d->T, 2, sdf_stream(stream));
}

extern "C" int sdf_spike_conv2d_fwd(const SdfSpikeConvDesc* c, void* stream) {
  GemmParams P;
  bool i8x3 = false, tiled = false;
  {
    const int rc = conv2d_build(c, P, i8x3, tiled);
    if (rc) return rc;
  }
  const SdfSpikeGemmDesc* d = &c->g;
  const bool spike = d->sn_T > 0;
  // few rows against many weights (the U-Net bottleneck's res-blocks) with digit planes: one launch, K split over the waves of a
  // workgroup (csrc/ms_smallm.hip); SDF_WIDE_CONV=1 selects the split-K-over-workgroups form it replaced (csrc/ms_wide.hip, A/B)
  if (i8x3 && wide_conv_supports(P)) return launch_wide_conv(P, sdf_stream(stream));
  if (i8x3 && smallm_conv_supports(P)) return launch_smallm_conv(P, sdf_stream(stream));
  if (tiled) return SDF_E_SHAPE;
  // 3x3 / stride 1 on 96 channels with enough tiles to fill the chip: weights resident in LDS, halo tiles instead of im2col
  const char* ewr = sdf_sw(SW_CONV_WRES);                  // A/B override: 0 = always the streaming kernels below, 2 = at any size
  if (!(ewr && ewr[0] == '0') && spike_conv_wres_supports(P, i8x3 || (ewr && ewr[0] == '2'))) return launch_spike_conv_wres(P, sdf_stream(stream));
  if (i8x3) return SDF_E_SHAPE;                                 // digit planes have no streaming-kernel form: the caller packs per shape
  // 256 x 96 tiles, producer waves do the im2col addressing; the ping-pong kernel overlaps epilogues with the MFMAs
  if (spike_mm_pp_supports(P, true)) return launch_spike_mm_pp(P, true, sdf_stream(stream));
  // operands beyond the kernel's 31-bit buffer offsets (e.g. 80 images of 240 x 320 x 96 fp32 out): images are
  // independent, so the fp32 epilogue form is launched in image chunks that fit
  const int64_t imgs = d->M / ((int64_t)c->OH * c->OW);
  if (!spike && imgs > 1) {
    for (int64_t nch = 2; nch <= imgs; ++nch) {
      const int64_t per = (imgs + nch - 1) / nch;
      GemmParams Q = P;
      Q.d.M = per * c->OH * c->OW;
      if (!spike_mm_pp_supports(Q, true)) continue;
      for (int64_t i0 = 0; i0 < imgs; i0 += per) {
        const int64_t n = imgs - i0 < per ? imgs - i0 : per, r0 = i0 * c->OH * c->OW;
        Q = P;
        Q.d.M = n * c->OH * c->OW;
        Q.d.A = d->A + i0 * c->H * c->W * c->Cin;
        if (d->out_rowmap) {
          Q.d.out_rowmap = d->out_rowmap + r0;
        } else {
          Q.d.out = d->out + r0 * d->ldo;
          if (d->resid) Q.d.resid = d->resid + r0 * d->ldo;
        }
        const int rc = launch_spike_mm_pp(Q, true, sdf_stream(stream));
        if (rc) return rc;
      }
      return 0;
    }
  }
  return SDF_E_SHAPE;                                           // no streaming-kernel form of this problem (round 6: the barrier-synchronised predecessor of the ping-pong kernel is gone)
}

// n convolutions on the same images that differ only in taps / weights / output row map (the four output-parity classes of a
// stride-2 transposed convolution): ONE launch of the ping-pong kernel where every one of them would take it, else one call each.
extern "C" int sdf_spike_conv2d_multi_fwd(const SdfSpikeConvDesc* cs, int n, void* stream) {
  if (!cs) return SDF_E_NULL;
  if (n < 1) return SDF_E_SHAPE;
  const char* emu = sdf_sw(SW_CONV_MULTI);                 // A/B override: 0 = one launch per convolution
  if (n >= 2 && n <= 4 && !(emu && emu[0] == '0')) {
    GemmParams Ps[4];
    bool one = true;
    for (int i = 0; i < n; ++i) {
      bool i8x3 = false, tiled = false;
      const int rc = conv2d_build(cs + i, Ps[i], i8x3, tiled);
      if (rc) return rc;
      const char* ewr = sdf_sw(SW_CONV_WRES);
      const bool wres = !(ewr && ewr[0] == '0') && spike_conv_wres_supports(Ps[i], i8x3 || (ewr && ewr[0] == '2'));
      one = one && !i8x3 && cs[i].g.sn_T == 0 && !wres && spike_mm_pp_supports(Ps[i], true);
    }
    if (one) {
      const int rc = launch_spike_mm_pp_multi(Ps, n, sdf_stream(stream));
      if (rc != SDF_E_SHAPE) return rc;                        // (SDF_E_SHAPE: not one family after all - e.g. a split-K plan)
    }
  }
  for (int i = 0; i < n; ++i) {
    const int rc = sdf_spike_conv2d_fwd(cs + i, stream);
    if (rc) return rc;
  }
  return 0;
}

extern "C" int sdf_split_weight_f16x2(const float* W, uint16_t* planes, int64_t n, float scale, void* stream) {
  if (!W || !planes) return SDF_E_NULL;
  if (n < 1) return SDF_E_SHAPE;
  int ex;
  if (!(scale > 0.f) || frexpf(scale, &ex) != 0.5f) return SDF_E_DTYPE;
  SDF_LAUNCH(split_weight_f16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, sdf_stream(stream), W, planes, n,
                     scale);
  SDF_LAUNCH_CHECK();
  return 0;
}

extern "C" int sdf_split_weight_bf16(const float* W, uint16_t* planes, int64_t n, int nsplit, void* stream) {
  if (!W || !planes) return SDF_E_NULL;
  if (n < 1) return SDF_E_SHAPE;
  if (nsplit < 1 || nsplit > 3) return SDF_E_DTYPE;
  SDF_LAUNCH(split_weight_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, sdf_stream(stream), W, planes, n,
                     nsplit);
  SDF_LAUNCH_CHECK();
  return 0;
}
