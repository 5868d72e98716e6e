// Stride-2 3x3 transposed convolution on spikes, weight-resident (gfx950): the last U-Net decoder level of the SNN forward
// (reference MS_SpikingTransposeDecoderLayer, Spiking_modules.py:461-474: SN -> ConvTranspose2d(3, stride 2, padding 1, output_padding 1)
// -> BatchNorm; 10 images of 72 x 96 x 208 spike bytes -> 144 x 192 x 48 fp32 at BASELINE config 2).
//
// out(2a + py, 2b + px) only receives input pixels (a + dh, b + dw), dh <= py, dw <= px: written as a product, a row is an input
// pixel (img, a, b), its K = 4 Cin the 2 x 2 input neighbourhood in quadrant order q = dh + 2 dw, its N = 4 Cout columns the four
// output pixels of the row's 2 x 2 output block (column (2 py + px) Cout + co; 7 of the 16 (quadrant, class) weight blocks are zero).
// Four parity-class convolutions on the streaming kernel ran this before (one launch: 78 us): 96-column tiles for 48 columns, fp16
// planes with the spike bytes expanded in registers, and every class writing every other pixel of an output row - half-written lines,
// 106 MB to HBM for a 53 MB tensor (profiles/r5f_pmc_forward.txt).  The row-loop kernel's form of the same product (ms_res.hip, AM = 3)
// measured 105 us: its operand rows come from L2 once per 32-column group and step.  Here the digit convolution's structure
// (spike_conv_wres.hip) is cut for this shape:
//   * a workgroup owns one 32-column block of the 4 Cout columns: its digit planes - the whole K = 4 Cin - are LDS-resident (80 KB);
//   * a group of 4 waves owns a tile of 8 x 16 input pixels: the 9 x 17 pixel halo (one extra row and column: the neighbourhood only
//     reaches down and right) enters LDS once and serves the four quadrants; a wave multiplies 32 pixels x 32 columns x K
//     (26 K steps x 3 digits of v_mfma_i32_32x32x32_i8, exact int32 sums);
//   * Cin = 208 is 13 sixteen-byte pieces per quadrant - an odd number, and a K step is two pieces (one per half wave): pieces are
//     paired INSIDE a quadrant (16 bytes apart), the four left-over pieces as (q0, q2) and (q1, q3) (one pixel apart), so that the
//     half wave's share is one of two lane bases and every step's offset an immediate (the K order of an exact sum is free);
//   * two groups per workgroup run out of phase (hand-over through LDS counters as in the convolution kernel);
//   * epilogue: digits -> fp32, BatchNorm, and a 16-byte piece of 4 channels goes to output pixel (2a + py, 2b + px) of its column's
//     class: the two pixels of a row's line are written by neighbouring pieces of the same wave;
//   * Cout % 32 == 0: a 32-column block lies inside ONE class, and a class only receives the quadrants with dh <= py, dw <= px - the
//     workgroup runs just those K steps (7 / 13 / 14 / 26 of 26: the zero blocks are never multiplied), and the tile ranges per class
//     are sized by a tile's cost in that class so that all workgroups finish together.
// Compiled with -ffp-contract=off.
#include "spike_mm.h"
#include "switches.h"
#include "wide_common.h"
#include <type_traits>

namespace sdfmm {
namespace {

constexpr int CIN = 208, C16 = CIN / 16, NB = 32;
constexpr int TH = 8, TW = 16, HH = TH + 1, HWD = TW + 1;
constexpr int PS = CIN;                                   // pixel stride (13 pieces: odd - conflict-free for the 32-pixel fragment reads)
constexpr int RPB = (HWD * PS + 255) / 256 * 256;         // halo row pitch: a whole number of bank rounds (the wave's two pixel rows)
constexpr int HALO = HH * RPB;
constexpr int KS = 2 * C16;                               // 26 K steps of 32 (4 quadrants x 13 pieces / 2)
constexpr int WP = 32 * KS + 16;                          // digit-plane row pitch: 53 pieces (odd)
constexpr int W_BYTES = 3 * NB * WP;
constexpr int PAR = 2 * NB * 4;
constexpr int NGRP = 2, NT = 256 * NGRP;
constexpr int LDS_BYTES = W_BYTES + NGRP * HALO + PAR + 64;
constexpr int RCH = HWD * C16;                            // 16-byte pieces of a halo row (221): one per lane of the group

static_assert(LDS_BYTES <= 160 * 1024 && RCH <= 256 && C16 % 2 == 1, "geometry");

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(16))) int i32x16;

struct DeconvParams {
  const uint8_t* A; const int8_t* W; const float* cscale; const float* alpha; const float* beta; float* out;
  int imgs, H, W_, Cout, N;
  int tiles_m, tiles_n, ntiles;
  // Cout % 32 == 0: a column block lies inside ONE output-parity class and only multiplies the quadrants that class receives; the
  // workgroups of class c are ids [cls_start[c], cls_start[c + 1]), cls_nr[c] tile ranges x (Cout / 32) column blocks, the ranges sized
  // by the class's K steps (7 / 13 / 14 / 26 of 26) so that every workgroup has about the same work.  cls_nr[0] = 0: the uniform form
  int cls_start[5], cls_nr[4];
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p) {
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
template <int CTRL>
__device__ __forceinline__ float dpp_quad(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// 4 x 4 transpose of dwords among the four lanes of a quad (spike_conv_wres.hip)
__device__ __forceinline__ void qt4(float& a0, float& a1, float& a2, float& a3, bool o1, bool o2) {
  float r = dpp_quad<0xB1>(o1 ? a0 : a1);
  a0 = o1 ? r : a0; a1 = o1 ? a1 : r;
  r = dpp_quad<0xB1>(o1 ? a2 : a3);
  a2 = o1 ? r : a2; a3 = o1 ? a3 : r;
  r = dpp_quad<0x4E>(o2 ? a0 : a2);
  a0 = o2 ? r : a0; a2 = o2 ? a2 : r;
  r = dpp_quad<0x4E>(o2 ? a1 : a3);
  a1 = o2 ? r : a1; a3 = o2 ? a3 : r;
}
// halo offset of the lower half wave's piece of K step ks, and which lane base it goes with
__device__ __forceinline__ constexpr int step_off(int ks) {
  if (ks < 4 * (C16 / 2)) {                                  // pairs inside quadrant q = dh + 2 dw
    const int q = ks / (C16 / 2), j = ks - q * (C16 / 2);
    return (q & 1) * RPB + (q >> 1) * PS + 32 * j;
  }
  return (ks - 4 * (C16 / 2)) * RPB + 16 * (C16 - 1);        // left-over pieces: (q0 | q2) then (q1 | q3), one pixel apart
}
__device__ __forceinline__ constexpr int step_base(int ks) { return ks < 4 * (C16 / 2) ? 0 : 1; }
// the K steps an output-parity class needs (class = 2 py + px receives quadrants with dh <= py, dw <= px; 4 = all: the uniform form):
// class 0: q0; class 1: q0, q2; class 2: q0, q1; class 3: all four.  Steps [6 q, 6 q + 6) are quadrant q's pairs, 24 = left-overs (q0 | q2),
// 25 = (q1 | q3).
__host__ __device__ constexpr int cls_nsteps(int cls) { return cls == 0 ? 7 : (cls == 1 ? 13 : (cls == 2 ? 14 : KS)); }
__device__ __forceinline__ constexpr int cls_step(int cls, int i) {
  constexpr int Q = C16 / 2;
  if (cls >= 3) return i;
  if (cls == 0) return i < Q ? i : 4 * Q;
  if (cls == 1) return i < Q ? i : (i < 2 * Q ? 2 * Q + (i - Q) : 4 * Q);
  return i < 2 * Q ? i : 4 * Q + (i - 2 * Q);
}
// LDS slot (16-byte pieces from the row start) of piece kc = q * C16 + c of a weight row
__device__ __forceinline__ int slot(int kc) {
  const int q = kc / C16, c = kc - q * C16;
  if (c < C16 - 1) return 2 * (q * (C16 / 2) + (c >> 1)) + (c & 1);
  return 2 * (4 * (C16 / 2) + (q & 1)) + (q >> 1);
}

__global__ __launch_bounds__(NT) void spike_deconv_wres_kernel(DeconvParams P) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* W_s = smem;
  float* par_s = reinterpret_cast<float*>(smem + W_BYTES + NGRP * HALO);
  uint32_t* cnt = reinterpret_cast<uint32_t*>(smem + W_BYTES + NGRP * HALO + PAR);   // [g]: halo written, [NGRP + g]: halo read
  const int H = P.H, W = P.W_;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int grp = wave >> 2, cw = wave & 3;
  const int gl = tid & 255;
  const int N = P.N, K = 4 * CIN;
  if (tid < 2 * NGRP) cnt[tid] = 0;

  // work items: item = cb * tiles_m + tile; the tiles_n workgroups that serve the column blocks of ONE tile range are neighbours on one
  // XCD and walk the range side by side (the halo leaves HBM once and comes out of that L2 for the other column blocks)
  const int tiles_x = (W + TW - 1) / TW, tiles_img = tiles_x * ((H + TH - 1) / TH);
  const int Gd = gridDim.x;
  int wg = blockIdx.x, nr, r, cbw, kcls = 4;
  if (P.cls_nr[0] > 0) {                                   // class-balanced form
    const int cbc = P.Cout / NB;
    kcls = wg >= P.cls_start[3] ? 3 : (wg >= P.cls_start[2] ? 2 : (wg >= P.cls_start[1] ? 1 : 0));
    kcls = __builtin_amdgcn_readfirstlane(kcls);
    const int wl = wg - P.cls_start[kcls];
    nr = P.cls_nr[kcls]; r = wl / cbc; cbw = kcls * cbc + (wl - r * cbc);
  } else {
    if ((Gd & 7) == 0) wg = (wg & 7) * (Gd >> 3) + (wg >> 3);
    nr = Gd / P.tiles_n; r = wg / P.tiles_n; cbw = wg - r * P.tiles_n;
  }
  if (r >= nr) return;
  const int base = P.tiles_m / nr, rem = P.tiles_m % nr;
  const int t_begin = r * base + (r < rem ? r : rem), n_my = base + (r < rem ? 1 : 0);
  const int n0 = cbw * NB;

  // ---- the column block's digit planes: 3 x 32 rows x 52 pieces, K order permuted (slot), once per workgroup ----
  {
    constexpr int KC16 = 4 * C16, WCH = 3 * NB * KC16;
    const __amdgpu_buffer_rsrc_t W_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(P.W), 0, 3 * N * K, 0x00020000);
    constexpr int WB = 5, NBATCH = (WCH + NT * WB - 1) / (NT * WB);
#pragma unroll 1
    for (int b = 0; b < NBATCH; ++b) {
      u32x4 wv[WB];
#pragma unroll
      for (int i = 0; i < WB; ++i) {
        const int c = tid + NT * (b * WB + i);
        const int cc = c < WCH ? c : 0;
        const int row = cc / KC16, kc = cc - row * KC16;               // row = digit * 32 + n
        const int dg = row / NB, n = row - dg * NB;
        wv[i] = __builtin_amdgcn_raw_buffer_load_b128(W_rs, (c < WCH && n0 + n < N) ? (uint32_t)((dg * N + n0 + n) * K + kc * 16) : INV, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < WB; ++i) {
        const int c = tid + NT * (b * WB + i);
        const int cc = c < WCH ? c : 0;
        const int row = cc / KC16, kc = cc - row * KC16;
        if (c < WCH) *reinterpret_cast<u32x4*>(W_s + row * WP + slot(kc) * 16) = wv[i];
      }
    }
    // BN folded with the column's digit scale: fma(D * s, alpha, beta) == fma(D, s * alpha, beta) exactly (s is a power of two)
    if (tid < 2 * NB) {
      const int which = tid / NB, n = tid - which * NB;
      const int nc = n0 + n < N ? n0 + n : 0;
      const float sc = P.cscale[nc];
      par_s[tid] = which == 0 ? (P.alpha ? P.alpha[nc] * sc : sc) : (P.alpha ? P.beta[nc] : 0.f);
    }
  }
  __syncthreads();

  // halo image: pass i of the group's 256 lanes moves halo row i, lane j the j-th 16-byte piece of it
  const bool hj_ok = gl < RCH;
  const int hpx = (hj_ok ? gl : 0) / C16, hc16 = (hj_ok ? gl : 0) - hpx * C16;
  const uint32_t h_lds0 = (uint32_t)(hpx * PS + hc16 * 16);
  const int h_rel0 = hpx * CIN + hc16 * 16;
  const __amdgpu_buffer_rsrc_t A_rs = rsrc(P.A), out_rs = rsrc(P.out);
  u32x4 hreg[HH];
  auto halo_load = [&](int img, int y0, int x0) __attribute__((always_inline)) {
    const uint32_t org = (uint32_t)(((img * H + y0) * W + x0) * CIN) + (uint32_t)h_rel0;
    const bool xok = hj_ok && x0 + hpx < W;
#pragma unroll
    for (int i = 0; i < HH; ++i) hreg[i] = __builtin_amdgcn_raw_buffer_load_b128(A_rs, (xok && y0 + i < H) ? org + (uint32_t)(i * W * CIN) : INV, 0, 0);
  };
  uint8_t* H_s = smem + W_BYTES + grp * HALO;
  auto halo_store = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < HH; ++i)
      if (hj_ok) *reinterpret_cast<u32x4*>(H_s + h_lds0 + i * RPB) = hreg[i];
  };
  auto item_decode = [&](int tile, int& img, int& y0, int& x0) __attribute__((always_inline)) {
    img = tile / tiles_img;
    const int tl = tile - img * tiles_img, ty = tl / tiles_x, tx = tl - ty * tiles_x;
    y0 = ty * TH; x0 = tx * TW;
  };

  int ln = lane;
  asm volatile("" : "+v"(ln));
  const int l31 = ln & 31, lh = ln >> 5;
  const uint32_t a_lane = (uint32_t)((2 * cw + (l31 >> 4)) * RPB + (l31 & 15) * PS);
  const uint32_t a_base[2] = {a_lane + (uint32_t)(16 * lh), a_lane + (uint32_t)(PS * lh)};
  const uint32_t w_lane = (uint32_t)(l31 * WP + 16 * lh);
  // this lane's store column after the quad transpose: channels n0 + 8 ql + 4 lh + 0..3 -> (class, channel): a byte offset inside the
  // row's 2 x 2 output block
  const int ql = l31 & 3;
  const int nst = n0 + 8 * ql + 4 * lh;
  const int cls = nst / P.Cout, co = nst - cls * P.Cout;
  const uint32_t cls_off = (uint32_t)((((cls >> 1) * 2 * W + (cls & 1)) * P.Cout + co) * 4);
  const bool col_ok = nst < N;

  uint32_t nstep = 0;
  int it = grp;
  int img, y0, x0;
  if (it < n_my) {
    item_decode(t_begin + it, img, y0, x0);
    halo_load(img, y0, x0);
    halo_store();
    signal(&cnt[grp], lane);
  }
  for (; it < n_my; it += NGRP) {
    item_decode(t_begin + it, img, y0, x0);
    // the next tile's halo is requested ahead of this tile's MFMAs (two groups: 256 registers, the 36 in flight are affordable)
    const bool have_next = it + NGRP < n_my;
    if (have_next) {
      int ni, ny, nx;
      item_decode(t_begin + it + NGRP, ni, ny, nx);
      halo_load(ni, ny, nx);
    }
    ++nstep;
    wait_ge(&cnt[grp], 4 * nstep);
    i32x16 acc[3];
#pragma unroll
    for (int dg = 0; dg < 3; ++dg)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[dg][e] = 0;
    constexpr int PF = 2;
    auto mfma_phase = [&](auto tag) __attribute__((always_inline)) {
      constexpr int CLS = decltype(tag)::value, NS = cls_nsteps(CLS);
      i32x4 fa[PF + 1], fb[PF + 1][3];
      auto frag = [&](int i, int set) __attribute__((always_inline)) {
        const int ks = cls_step(CLS, i);
        fa[set] = *reinterpret_cast<const i32x4*>(H_s + a_base[step_base(ks)] + step_off(ks));
#pragma unroll
        for (int dg = 0; dg < 3; ++dg) fb[set][dg] = *reinterpret_cast<const i32x4*>(W_s + w_lane + (dg * NB * WP + ks * 32));
      };
#pragma unroll
      for (int i = 0; i < PF; ++i) frag(i, i);
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        if (i + PF < NS) frag(i + PF, (i + PF) % (PF + 1));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int dg = 0; dg < 3; ++dg)
          acc[dg] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb[i % (PF + 1)][dg], fa[i % (PF + 1)], acc[dg], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    switch (kcls) {                                          // (wave-uniform: a workgroup's column block lies in one class)
      case 0: mfma_phase(std::integral_constant<int, 0>{}); break;
      case 1: mfma_phase(std::integral_constant<int, 1>{}); break;
      case 2: mfma_phase(std::integral_constant<int, 2>{}); break;
      default: mfma_phase(std::integral_constant<int, 4>{}); break;
    }
    signal(&cnt[NGRP + grp], lane);

    // ---- epilogue: this lane's pixel (ybase, xbase); accumulator quad q4 = columns n0 + 8 q4 + 4 lh + 0..3 ----
    const int ybase = y0 + 2 * cw + (l31 >> 4), xbase = x0 + (l31 & 15);
    float4 om[4];
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e = q4 * 4 + j;
        const int lo = acc[1][e] * 256 + acc[0][e];                    // the two low digits meet as integers (< 2^26), the high one in the fma
        v[j] = __builtin_fmaf((float)acc[2][e], 65536.f, (float)lo);
      }
      const float4 al4 = *reinterpret_cast<const float4*>(par_s + 8 * q4 + 4 * lh);
      const float4 be4 = *reinterpret_cast<const float4*>(par_s + NB + 8 * q4 + 4 * lh);
      om[q4] = make_float4(__builtin_fmaf(v[0], al4.x, be4.x), __builtin_fmaf(v[1], al4.y, be4.y), __builtin_fmaf(v[2], al4.z, be4.z),
                           __builtin_fmaf(v[3], al4.w, be4.w));
    }
    // the four lanes of a quad (four consecutive pixels) exchange their quads: lane i ends with quad i of pixel j in om[j]
    const bool o1 = (l31 & 1) != 0, o2 = (l31 & 2) != 0;
    qt4(om[0].x, om[1].x, om[2].x, om[3].x, o1, o2);
    qt4(om[0].y, om[1].y, om[2].y, om[3].y, o1, o2);
    qt4(om[0].z, om[1].z, om[2].z, om[3].z, o1, o2);
    qt4(om[0].w, om[1].w, om[2].w, om[3].w, o1, o2);
    const bool yok = ybase < H;
    // output pixel (2 y, 2 x) of input pixel (y, x): index ((img * 2H + 2 y) * 2W + 2 x); the quad's first pixel is xbase - ql
    const uint32_t z0 = (uint32_t)(((img * 2 * H + 2 * ybase) * 2 * W + 2 * (xbase - ql)) * P.Cout * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool ok = yok && col_ok && xbase - ql + j < W;
      const u32x4 o = {__float_as_uint(om[j].x), __float_as_uint(om[j].y), __float_as_uint(om[j].z), __float_as_uint(om[j].w)};
      __builtin_amdgcn_raw_buffer_store_b128(o, out_rs, ok ? z0 + (uint32_t)(2 * j * P.Cout * 4) + cls_off : INV, 0, 0);
    }
    if (have_next) {
      wait_ge(&cnt[NGRP + grp], 4 * nstep);
      halo_store();
      signal(&cnt[grp], lane);
    }
  }
}

}  // namespace

bool spike_deconv_wres_supports(int imgs, int H, int W, int Cin, int Cout) {
  const char* e = sdf_sw(SW_DECONV_WRES);                   // A/B: 0 = the row-loop kernel's form (ms_res.hip)
  if (e && e[0] == '0') return false;
  if (Cin != CIN || Cout < 8 || Cout % 4 || (4 * Cout) % 4) return false;
  const int64_t rows = (int64_t)imgs * H * W;
  return rows * CIN < (1LL << 31) && rows * 16 * Cout < (1LL << 31) && rows >= 4096;
}

int launch_spike_deconv_wres(const uint8_t* A, const int8_t* Wd, const float* cscale, const float* alpha, const float* beta, float* out,
                             int imgs, int H, int W, int Cout, hipStream_t s) {
  DeconvParams P;
  P.A = A; P.W = Wd; P.cscale = cscale; P.alpha = alpha; P.beta = beta; P.out = out;
  P.imgs = imgs; P.H = H; P.W_ = W; P.Cout = Cout; P.N = 4 * Cout;
  P.tiles_m = imgs * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
  P.tiles_n = (P.N + NB - 1) / NB;
  P.ntiles = P.tiles_m * P.tiles_n;
  int grid;
  const char* eb = sdf_sw(SW_DECONV_BALANCE);           // A/B: 0 = every column block multiplies all four quadrants
  if (Cout % NB == 0 && P.tiles_m >= 64 && !(eb && eb[0] == '0')) {
    // class-balanced: ranges per class in proportion to a tile's cost in that class - its K steps plus the part every tile pays (halo,
    // epilogue, stores: 16 K steps' worth measured best; SDF_DECONV_EPI: tuning override) - 250 - 256 workgroups in all
    const int cbc = Cout / NB, budget = 256 / cbc;
    int epi = 16;
    if (const char* e = sdf_sw(SW_DECONV_EPI)) { const int v = atoi(e); if (v >= 0 && v <= 100) epi = v; }
    int tot = 0, nrc[4];
    for (int c = 0; c < 4; ++c) tot += cls_nsteps(c) + epi;
    int used = 0;
    for (int c = 0; c < 4; ++c) {
      nrc[c] = budget * (cls_nsteps(c) + epi) / tot;
      if (nrc[c] < 1) nrc[c] = 1;
      if (nrc[c] > P.tiles_m) nrc[c] = P.tiles_m;
      used += nrc[c];
    }
    while (used < budget && nrc[3] < P.tiles_m) { ++nrc[3]; ++used; }
    P.cls_start[0] = 0;
    for (int c = 0; c < 4; ++c) { P.cls_nr[c] = nrc[c]; P.cls_start[c + 1] = P.cls_start[c] + nrc[c] * cbc; }
    grid = P.cls_start[4];
  } else {
    // tile ranges x column blocks, column block fastest: as many ranges as fill the chip once with the grid a multiple of 8 (a range's
    // column blocks then share an XCD: the kernel's id remap)
    int nr = 256 / P.tiles_n;
    while (nr > 1 && (nr * P.tiles_n) % 8) --nr;
    if (nr > P.tiles_m) nr = P.tiles_m;
    if (nr < 1) nr = 1;
    for (int c = 0; c < 4; ++c) P.cls_nr[c] = 0;
    for (int c = 0; c < 5; ++c) P.cls_start[c] = 0;
    grid = nr * P.tiles_n;
  }
  static std::atomic<uint64_t> raised{0};                  // > 64 KiB of dynamic LDS: opt-in once per device
  if (const int e = sdf_lds_opt_in(raised, reinterpret_cast<const void*>(spike_deconv_wres_kernel), LDS_BYTES)) return e;
  SDF_LAUNCH(spike_deconv_wres_kernel, dim3((unsigned)grid), dim3(NT), LDS_BYTES, s, P);
  hipError_t e = hipGetLastError();
  return e != hipSuccess ? (int)e : 0;
}

}  // namespace sdfmm
