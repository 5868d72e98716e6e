// Weight-resident 3x3 convolution of REAL-valued activations for gfx950 (stride 1, pad 1): the dense convolutions of the ANN
// patch embedding (reference models/STSwinNet/PatchEmbed.py:166-196 `PatchEmbedLocal`: head conv + 4 BatchNorm residual blocks,
// models/submodules.py:160-229 `ResidualBlock`), which the library served with an fp32 Winograd kernel at 2.7 ms per layer
// (profiles/r1o_config3.txt: 58 % of BASELINE config 3).
//
//   y[img, n, y, x] = act( alpha[n] * sum_{ky,kx,c} a[img, c, y+ky-1, x+kx-1] * w[n, c, ky, kx] + beta[n] (+ resid) )
//
// fp32 values on the 16-bit matrix pipe: every operand is the sum of two fp16 numbers, a = a_hi + a_lo (a_hi = fp16(a),
// a_lo = fp16(a - a_hi): 22 significant bits), and a product is three MFMAs, a_hi*w_hi + a_lo*w_hi + a_hi*w_lo, accumulated
// in fp32 inside the matrix cores (the dropped a_lo*w_lo term is 2^-22 of the product).  The activations LIVE in that form
// between the layers - the epilogue of one convolution writes the hi / lo pairs the next one multiplies, 4 bytes per value like
// fp32 - so no layer pays a conversion pass and BatchNorm (folded to alpha / beta), the residual add and the ReLU never
// touch memory on their own.
//
// Same division of labour as spike_conv_wres.hip: a workgroup owns one 32-column block of the output and keeps that block's
// weights - both planes, the whole K = 9 * Cin - in LDS (2 x 32 x 864 fp16 = 110 KB); activations enter as halo images
// (10 x 18 pixels for a tile of 8 x 16 outputs), the nine taps are nine constant offsets into the image.  A halo image of all
// 96 channels in two planes would be 69 KB, so the image is cut by CHANNEL: one step = 16 input channels (15 KB image, 9 taps x 3
// MFMAs per wave), the accumulators stay in registers across the steps of a tile and the epilogue runs after the last one.
// 8 wavefronts = 2 groups of 4; a group owns one halo image, the next image is requested from memory before the MFMAs of the
// current one and written after them, the two groups run out of phase (LDS counters, no workgroup barrier in the steady state).
//
// Activation layout ("planes", sdf_pack_planes): [img][Cin/16][H][W] records of 64 bytes = 4 x { 4 x fp16 hi, 4 x fp16 lo } for
// 16 channels: a 16-byte piece is four channels complete, which is what one lane holds after the epilogue's quad transpose
// (one 16-byte store per four channels) and what the halo loader splits into the hi and lo halves of the LDS pixel record.
// LDS is conflict-free by construction: pixel stride 80 bytes (16-byte slots 5 apart), row pitch a multiple of 256 bytes, so
// the 16 lanes of every ds_read_b128 group cover the 16 slots of a bank row once; weight rows 2K + 16 bytes (4 x odd dwords).
#include "spike_mm.h"

namespace sdfmm {
namespace {

constexpr int TH = 8, TW = 16;                  // output pixels of a tile: 4 waves x (2 rows x 16 pixels)
constexpr int HH = TH + 2, HWID = TW + 2;       // halo image
constexpr int NB = 32;                          // output columns of a workgroup
constexpr int REC = 64;                         // bytes of a pixel record in memory (16 channels, hi + lo)
constexpr int PS = 80;                          // pixel stride in the halo image: 32 B hi, 32 B lo, 16 B pad
constexpr int RPB = 1536;                       // halo row pitch: 18 * 80 = 1440 -> next multiple of 256
constexpr int HALO = HH * RPB;
constexpr int PIECES = HH * HWID * 4;           // 16-byte pieces of a halo image
constexpr int CPL = (PIECES + 255) / 256;       // pieces per lane of a group
constexpr uint32_t INV = 0x80000000u;

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) _Float16 h2;

struct DenseParams {
  SdfDenseConvDesc d;
  int tiles_m, ntiles;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)INV, 0x00020000);
}
__device__ __forceinline__ void wait_ge(uint32_t* p, uint32_t target) {
  while (true) {
    const uint32_t v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    if ((int32_t)(v - target) >= 0) break;
    __builtin_amdgcn_s_sleep(1);
  }
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void signal(uint32_t* p, int lane) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane == 0) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// fp32 -> (hi, lo) fp16 pair, round to nearest; values beyond the fp16 range saturate instead of becoming infinities
__device__ __forceinline__ void split_f16(float x, _Float16& hi, _Float16& lo) {
  x = __builtin_fminf(__builtin_fmaxf(x, -65000.f), 65000.f);
  hi = (_Float16)x;
  lo = (_Float16)(x - (float)hi);
}
__device__ __forceinline__ uint32_t pack2(_Float16 a, _Float16 b) {
  h2 v; v.x = a; v.y = b;
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float2 unpack2(uint32_t u) {
  const h2 v = __builtin_bit_cast(h2, u);
  return make_float2((float)v.x, (float)v.y);
}
// one 16-byte piece {hi0..3, lo0..3} <-> four fp32 values
__device__ __forceinline__ u32x4 piece_from(float4 o) {
  _Float16 h[4], l[4];
  split_f16(o.x, h[0], l[0]); split_f16(o.y, h[1], l[1]); split_f16(o.z, h[2], l[2]); split_f16(o.w, h[3], l[3]);
  u32x4 v;
  v.x = pack2(h[0], h[1]); v.y = pack2(h[2], h[3]); v.z = pack2(l[0], l[1]); v.w = pack2(l[2], l[3]);
  return v;
}
__device__ __forceinline__ float4 piece_to(u32x4 v) {
  const float2 h01 = unpack2(v.x), h23 = unpack2(v.y), l01 = unpack2(v.z), l23 = unpack2(v.w);
  return make_float4(h01.x + l01.x, h01.y + l01.y, h23.x + l23.x, h23.y + l23.y);
}

template <int CCH>
__global__ __launch_bounds__(512) void dense_conv_wres_kernel(DenseParams P) {
  constexpr int K = 9 * 16 * CCH;
  constexpr int WP = 2 * K + 16;                                      // weight row pitch (bytes): 4 x odd dwords
  constexpr int W_BYTES = 2 * NB * WP;
  constexpr int PAR = 2 * NB * 4;
  static_assert((WP / 4) % 8 == 4, "weight row pitch must be 4 x odd dwords");
  static_assert(W_BYTES + 2 * HALO + PAR + 64 <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(16))) uint8_t smem[W_BYTES + 2 * HALO + PAR + 64];
  uint8_t* W_s = smem;
  float* par_s = reinterpret_cast<float*>(smem + W_BYTES + 2 * HALO);
  uint32_t* cnt = reinterpret_cast<uint32_t*>(smem + W_BYTES + 2 * HALO + PAR);     // [g]: halo written, [2 + g]: halo read

  const SdfDenseConvDesc& d = P.d;
  const int H = d.H, W = d.W, N = d.N;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int grp = wave >> 2, cw = wave & 3;
  const int gl = tid & 255;                                           // lane inside the group
  const bool has_res = d.resid != nullptr;
  const bool relu = d.relu != 0, out_f32 = d.out_f32 != 0;
  const int nch = N >> 4;                                             // channel records of the output / residual planes

  if (tid < 4) cnt[tid] = 0;

  // work items: item = cb * tiles_m + rt, rt = img * tiles_img + tile; contiguous ranges per workgroup, workgroups dealt
  // XCD-contiguously (neighbouring tiles share halo rows in one L2)
  const int tiles_x = (W + TW - 1) / TW, tiles_img = tiles_x * ((H + TH - 1) / TH);
  const int Gd = gridDim.x;
  int wg = blockIdx.x;
  if ((Gd & 7) == 0) wg = (wg & 7) * (Gd >> 3) + (wg >> 3);
  const int nitems = P.ntiles;
  const int base = nitems / Gd, rem = nitems % Gd;
  const int t_begin = wg * base + (wg < rem ? wg : rem);
  const int n_my = base + (wg < rem ? 1 : 0);

  // this lane's pieces of a halo image: LDS offset of the hi half, byte offset relative to the tile's origin record, (dy, dx)
  uint32_t h_lds[CPL];
  int h_rel[CPL], h_yx[CPL];
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const int c = gl + 256 * i;
    const int cc = c < PIECES ? c : 0;
    const int hy = cc / (HWID * 4), r = cc - hy * (HWID * 4);
    const int px = r >> 2, j = r & 3;
    h_lds[i] = c < PIECES ? (uint32_t)(hy * RPB + px * PS + 8 * j) : 0xFFFFFFFFu;
    h_rel[i] = ((hy - 1) * W + (px - 1)) * REC + 16 * j;
    h_yx[i] = ((hy - 1) << 16) | ((px - 1) & 0xFFFF);
  }
  const __amdgpu_buffer_rsrc_t A_rs = make_rsrc(d.x);
  u32x4 hreg[CPL];
  auto halo_load = [&](int img, int ch, int y0, int x0) __attribute__((always_inline)) {
    const uint32_t org = (uint32_t)((((img * CCH + ch) * H + y0) * W + x0) * REC);
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      const int yy = y0 + (h_yx[i] >> 16), xx = x0 + (int)(int16_t)(h_yx[i] & 0xFFFF);
      const bool ok = h_lds[i] != 0xFFFFFFFFu && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
      hreg[i] = __builtin_amdgcn_raw_buffer_load_b128(A_rs, ok ? org + (uint32_t)h_rel[i] : INV, 0, 0);
    }
  };
  uint8_t* H_s = smem + W_BYTES + grp * HALO;
  auto halo_store = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      if (h_lds[i] != 0xFFFFFFFFu) {
        *reinterpret_cast<uint2*>(H_s + h_lds[i]) = make_uint2(hreg[i].x, hreg[i].y);          // 4 x hi
        *reinterpret_cast<uint2*>(H_s + h_lds[i] + 32) = make_uint2(hreg[i].z, hreg[i].w);     // 4 x lo
      }
    }
  };

  auto item_decode = [&](int item, int& cb, int& img, int& y0, int& x0) __attribute__((always_inline)) {
    cb = item / P.tiles_m;
    const int rt = item - cb * P.tiles_m;
    img = rt / tiles_img;
    const int tl = rt - img * tiles_img;
    const int ty = tl / tiles_x, tx = tl - ty * tiles_x;
    y0 = ty * TH; x0 = tx * TW;
  };

  // fragment addresses of this lane: 32 pixels of the wave (2 rows x 16) as MFMA rows, 32 weight rows as MFMA columns
  int ln = lane;
  asm volatile("" : "+v"(ln));
  const int l31 = ln & 31, lh = ln >> 5;
  const uint32_t a_lane = (uint32_t)((2 * cw + (l31 >> 4)) * RPB + (l31 & 15) * PS + 16 * lh);
  const uint32_t w_lane = (uint32_t)(l31 * WP + 16 * lh);
  const int qd = l31 >> 2, ql = l31 & 3;
  const __amdgpu_buffer_rsrc_t out_rs = make_rsrc(d.out), res_rs = make_rsrc(d.resid);

  uint32_t nstep = 0;                                                 // halo steps this group has been through
  int seg_begin = 0;
  while (seg_begin < n_my) {
    // ---------------- a segment of items that share the column block: (re)load its weights ----------------
    int cb, img, y0, x0;
    item_decode(t_begin + seg_begin, cb, img, y0, x0);
    int seg_end = (cb + 1) * P.tiles_m - t_begin;
    if (seg_end > n_my) seg_end = n_my;
    const int n0 = cb * NB;
    __syncthreads();                                                  // every wave is done with the previous block's weights
    {
      constexpr int KC8 = K / 8;                                       // 16-byte pieces per weight row
      constexpr int WCH = 2 * NB * KC8;
      const __amdgpu_buffer_rsrc_t W_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(d.w), 0, 2 * N * K * 2, 0x00020000);
      constexpr int WB = 7, NBATCH = (WCH + 512 * WB - 1) / (512 * WB);   // batches of 7 pieces per lane in flight
#pragma unroll 1
      for (int b = 0; b < NBATCH; ++b) {
        u32x4 wv[WB];
#pragma unroll
        for (int i = 0; i < WB; ++i) {
          const int c = tid + 512 * (b * WB + i);
          const int cc = c < WCH ? c : 0;
          const int row = cc / KC8, kc = cc - row * KC8;               // row = p * 32 + n
          const int p = row / NB, n = row - p * NB;
          wv[i] = __builtin_amdgcn_raw_buffer_load_b128(W_rs, c < WCH ? (uint32_t)(((p * N + n0 + n) * K + kc * 8) * 2) : INV, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < WB; ++i) {
          const int c = tid + 512 * (b * WB + i);
          const int cc = c < WCH ? c : 0;
          const int row = cc / KC8, kc = cc - row * KC8;
          if (c < WCH) *reinterpret_cast<u32x4*>(W_s + row * WP + kc * 16) = wv[i];
        }
      }
      if (tid < 2 * NB) {
        const int which = tid / NB, n = tid - which * NB;
        float v = which == 0 ? 1.f : 0.f;
        if (which == 0 && d.alpha) v = d.alpha[n0 + n];
        if (which == 1 && d.beta) v = d.beta[n0 + n];
        par_s[tid] = v;
      }
    }
    __syncthreads();
    const float4 al4 = *reinterpret_cast<const float4*>(par_s + 4 * qd);
    const float4 be4 = *reinterpret_cast<const float4*>(par_s + NB + 4 * qd);

    // ---------------- this group's items of the segment: seg_begin + grp, + 2, ... ----------------
    int it = seg_begin + grp;
    if (it < seg_end) {
      item_decode(t_begin + it, cb, img, y0, x0);
      halo_load(img, 0, y0, x0);
      if (nstep) wait_ge(&cnt[2 + grp], 4 * nstep);                   // previous halo fully read by the group
      halo_store();
      signal(&cnt[grp], lane);
    }
    for (; it < seg_end; it += 2) {
      item_decode(t_begin + it, cb, img, y0, x0);
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      uint32_t rowoff[4];                                             // byte offset of (pixel, first channel of the lane) in out, or INV
      uint32_t resoff[4];
      u32x4 rs[4];
#pragma unroll 1
      for (int ch = 0; ch < CCH; ++ch) {
        // request the NEXT step's halo now: its latency hides behind this step's MFMAs
        bool have_next = true;
        {
          int ni = img, nc = ch + 1, ny = y0, nx = x0;
          if (nc == CCH) {
            nc = 0;
            if (it + 2 < seg_end) { int ncb; item_decode(t_begin + it + 2, ncb, ni, ny, nx); }
            else have_next = false;
          }
          if (have_next) halo_load(ni, nc, ny, nx);
        }
        if (ch == CCH - 1) {
          // rows of the epilogue and its residual, requested before the last MFMAs.  quad transpose: lane (qd, ql) ends with
          // columns 4qd..4qd+3 of row rr = 8*q4 + 4*lh + ql of the wave's 32 pixels
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const int rr = 8 * q4 + 4 * lh + ql;
            const int y = y0 + 2 * cw + (rr >> 4), x = x0 + (rr & 15);
            const bool ok = y < H && x < W;
            const uint32_t rec = (uint32_t)(((img * nch + 2 * cb + (qd >> 2)) * H + y) * W + x) * REC + (uint32_t)(qd & 3) * 16u;
            resoff[q4] = ok ? rec : INV;
            rowoff[q4] = !ok ? INV : out_f32 ? (uint32_t)((img * H + y) * W + x) * (uint32_t)N * 4u + (uint32_t)(n0 + 4 * qd) * 4u : rec;
            rs[q4] = u32x4{0u, 0u, 0u, 0u};
          }
          if (has_res) {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) rs[q4] = __builtin_amdgcn_raw_buffer_load_b128(res_rs, resoff[q4], 0, 0);
          }
        }
        ++nstep;
        wait_ge(&cnt[grp], 4 * nstep);                                // this step's halo is in LDS (all four waves' pieces)

        // ------------------------------ MFMA phase: 9 taps x 3 products ------------------------------
        constexpr int PF = 3;                                         // taps of fragments in flight
        bf16x8 fa[PF + 1][2], fb[PF + 1][2];
        const uint32_t w_step = w_lane + (uint32_t)ch * (9 * 32);
        auto frag = [&](int tap, int set) __attribute__((always_inline)) {
          const int ky = tap / 3, kx = tap - 3 * ky;
          fa[set][0] = *reinterpret_cast<const bf16x8*>(H_s + a_lane + (ky * RPB + kx * PS));
          fa[set][1] = *reinterpret_cast<const bf16x8*>(H_s + a_lane + (ky * RPB + kx * PS + 32));
          fb[set][0] = *reinterpret_cast<const bf16x8*>(W_s + w_step + tap * 32);
          fb[set][1] = *reinterpret_cast<const bf16x8*>(W_s + w_step + (NB * WP + tap * 32));
        };
#pragma unroll
        for (int i = 0; i < PF; ++i) frag(i, i);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          if (tap + PF < 9) frag(tap + PF, (tap + PF) % (PF + 1));
          __builtin_amdgcn_sched_barrier(0);
          const int s = tap % (PF + 1);
          acc = mma<2>(fa[s][1], fb[s][0], acc);                      // small terms first
          acc = mma<2>(fa[s][0], fb[s][1], acc);
          acc = mma<2>(fa[s][0], fb[s][0], acc);
          __builtin_amdgcn_sched_barrier(0);
        }
        signal(&cnt[2 + grp], lane);                                  // every fragment of this halo is in registers

        // ------------------------------ epilogue after the last channel step ------------------------------
        if (ch == CCH - 1) {
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            float v[4] = {acc[q4 * 4 + 0], acc[q4 * 4 + 1], acc[q4 * 4 + 2], acc[q4 * 4 + 3]};
            quad_transpose(v, ql);
            float4 o;
            o.x = __builtin_fmaf(v[0], al4.x, be4.x); o.y = __builtin_fmaf(v[1], al4.y, be4.y);
            o.z = __builtin_fmaf(v[2], al4.z, be4.z); o.w = __builtin_fmaf(v[3], al4.w, be4.w);
            if (has_res) {
              const float4 r = piece_to(rs[q4]);
              o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
            }
            if (relu) {
              o.x = __builtin_fmaxf(o.x, 0.f); o.y = __builtin_fmaxf(o.y, 0.f);
              o.z = __builtin_fmaxf(o.z, 0.f); o.w = __builtin_fmaxf(o.w, 0.f);
            }
            u32x4 st;
            if (out_f32) {
              st.x = __float_as_uint(o.x); st.y = __float_as_uint(o.y); st.z = __float_as_uint(o.z); st.w = __float_as_uint(o.w);
            } else {
              st = piece_from(o);
            }
            __builtin_amdgcn_raw_buffer_store_b128(st, out_rs, rowoff[q4], 0, 0);
          }
        }
        // ------------------------------ hand the next halo over ------------------------------
        if (have_next) {
          wait_ge(&cnt[2 + grp], 4 * nstep);                          // all four waves have this step's fragments in registers
          halo_store();
          signal(&cnt[grp], lane);
        }
      }
    }
    seg_begin = seg_end;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// NCHW fp32 <-> planes.  One thread per (pixel, 4-channel piece); channels beyond C are zero in the planes.
__global__ __launch_bounds__(256) void pack_planes_kernel(const float* __restrict__ x, u32x4* __restrict__ planes, int imgs, int C,
                                                          int nch, int H, int W) {
  const int64_t hw = (int64_t)H * W;
  const int64_t total = (int64_t)imgs * nch * 4 * hw;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t pix = i % hw;                                        // pixel fastest: the four strided reads are coalesced
    const int64_t r = i / hw;
    const int j = (int)(r & 3);
    const int64_t ic = r >> 2;                                         // img * nch + record
    const int img = (int)(ic / nch), rec = (int)(ic - (int64_t)img * nch);
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = rec * 16 + 4 * j + e;
      v[e] = c < C ? x[((int64_t)img * C + c) * hw + pix] : 0.f;
    }
    planes[(ic * hw + pix) * 4 + j] = piece_from(make_float4(v[0], v[1], v[2], v[3]));
  }
}

__global__ __launch_bounds__(256) void unpack_planes_kernel(const u32x4* __restrict__ planes, float* __restrict__ x, int imgs, int C,
                                                            int nch, int H, int W) {
  const int64_t hw = (int64_t)H * W;
  const int64_t total = (int64_t)imgs * nch * 4 * hw;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t pix = i % hw;
    const int64_t r = i / hw;
    const int j = (int)(r & 3);
    const int64_t ic = r >> 2;
    const int img = (int)(ic / nch), rec = (int)(ic - (int64_t)img * nch);
    const float4 o = piece_to(planes[(ic * hw + pix) * 4 + j]);
    const float v[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = rec * 16 + 4 * j + e;
      if (c < C) x[((int64_t)img * C + c) * hw + pix] = v[e];
    }
  }
}

}  // namespace
}  // namespace sdfmm

extern "C" int sdf_dense_conv3x3_fwd(const SdfDenseConvDesc* d, void* stream) {
  using namespace sdfmm;
  if (!d || !d->x || !d->w || !d->out) return SDF_E_NULL;
  if (d->imgs <= 0 || d->H <= 0 || d->W <= 0 || d->N <= 0 || d->N % NB) return SDF_E_SHAPE;
  if (d->cin_records != 1 && d->cin_records != 6) return SDF_E_SHAPE;            // instantiated: Cin <= 16 (head), Cin = 96
  if (!sdf_aligned(d->x, 16) || !sdf_aligned(d->w, 16) || !sdf_aligned(d->out, 16) || (d->resid && !sdf_aligned(d->resid, 16)))
    return SDF_E_ALIGN;
  const int64_t lim = (int64_t)1 << 31, px = (int64_t)d->imgs * d->H * d->W;
  if (px * d->cin_records * REC >= lim || px * d->N * 4 >= lim) return SDF_E_SHAPE;   // 31-bit buffer offsets
  DenseParams P;
  P.d = *d;
  P.tiles_m = (int)((int64_t)d->imgs * ((d->H + TH - 1) / TH) * ((d->W + TW - 1) / TW));
  P.ntiles = P.tiles_m * (d->N / NB);
  const int G = P.ntiles < 256 ? P.ntiles : 256;
  hipStream_t s = sdf_stream(stream);
  if (d->cin_records == 1) hipLaunchKernelGGL((dense_conv_wres_kernel<1>), dim3(G), dim3(512), 0, s, P);
  else hipLaunchKernelGGL((dense_conv_wres_kernel<6>), dim3(G), dim3(512), 0, s, P);
  SDF_LAUNCH_CHECK();
  return 0;
}

static int planes_grid(int64_t total) {
  const int64_t b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : b > 65536 ? 65536 : b);
}

extern "C" int sdf_pack_planes(const float* x, void* planes, int imgs, int C, int H, int W, void* stream) {
  if (!x || !planes) return SDF_E_NULL;
  if (imgs <= 0 || C <= 0 || H <= 0 || W <= 0) return SDF_E_SHAPE;
  if (!sdf_aligned(planes, 16)) return SDF_E_ALIGN;
  const int nch = (C + 15) / 16;
  const int64_t total = (int64_t)imgs * nch * 4 * H * W;
  hipLaunchKernelGGL(sdfmm::pack_planes_kernel, dim3(planes_grid(total)), dim3(256), 0, sdf_stream(stream), x,
                     reinterpret_cast<sdfmm::u32x4*>(planes), imgs, C, nch, H, W);
  SDF_LAUNCH_CHECK();
  return 0;
}

extern "C" int sdf_unpack_planes(const void* planes, float* x, int imgs, int C, int H, int W, void* stream) {
  if (!x || !planes) return SDF_E_NULL;
  if (imgs <= 0 || C <= 0 || H <= 0 || W <= 0) return SDF_E_SHAPE;
  if (!sdf_aligned(planes, 16)) return SDF_E_ALIGN;
  const int nch = (C + 15) / 16;
  const int64_t total = (int64_t)imgs * nch * 4 * H * W;
  hipLaunchKernelGGL(sdfmm::unpack_planes_kernel, dim3(planes_grid(total)), dim3(256), 0, sdf_stream(stream),
                     reinterpret_cast<const sdfmm::u32x4*>(planes), x, imgs, C, nch, H, W);
  SDF_LAUNCH_CHECK();
  return 0;
}
