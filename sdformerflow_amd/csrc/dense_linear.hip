// Linear layer on REAL-valued activations for gfx950: the qkv / proj / fc1 / fc2 projections of the ANN swin blocks
// (reference models/STSwinNet/swin_transformer3D_v2.py:176-202 `WindowAttention3D.forward`, :15-34 `Mlp.forward`, :272-313
// `forward_part1 / forward_part2`), which the library serves with fp32 GEMMs at ~75 TFLOP/s.
//
//   C[m, n] = act( sum_k A[m, k] * W[n, k] + bias[n] ) (+ resid[m, n])
//
// fp32 values on the 16-bit matrix pipe exactly as in dense_conv_wres.hip: a = a_hi + a_lo, w = w_hi + w_lo in fp16,
// a * w = a_hi*w_hi + a_lo*w_hi + a_hi*w_lo (three v_mfma_f32_32x32x16_f16, fp32 accumulation).  The weights arrive as fp16
// planes [2][N][K] made once per layer; the activation is split where it enters LDS (v_cvt_pk_f16_f32 on the loader's
// registers), so it stays plain fp32 row-major in memory on both sides and the layer is a drop-in for F.linear.
//
// A workgroup (4 waves) owns a 128 x 96 tile of C: K advances in chunks of 32 through a double-buffered LDS image of both
// operands (the next chunk is requested from memory before the MFMAs of the current one and split / written after them: one
// barrier per chunk); a wave multiplies 32 rows x 96 columns (9 MFMAs per 16 k from 8 ds_read_b128).  The weights are the
// MFMA's row operand, so a lane's accumulator quads are four consecutive columns of one row of C: bias, GELU (erf form, as
// F.gelu) and the residual are applied on 16-byte pieces and stored as such.  LDS rows are 64 data bytes + 16 pad (slots 5
// apart: conflict-free ds_read_b128 over 16 consecutive rows).
#include "common.h"
#include "switches.h"
#include <math.h>
#include <stdlib.h>

namespace {

constexpr int BM = 128, BN = 96, KC = 32;
constexpr int RS = 80;                                   // LDS row stride in bytes: 32 fp16 + 16 pad
constexpr int A_PLANE = BM * RS, B_PLANE = BN * RS;
constexpr int BUF = 2 * A_PLANE + 2 * B_PLANE;           // one stage: A hi, A lo, W hi, W lo
constexpr uint32_t INV = 0x80000000u;

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) _Float16 h2;
typedef __attribute__((ext_vector_type(2))) float f2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

struct LinearParams {
  SdfDenseLinearDesc d;
  int tiles_n;
  int xcd;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

// four fp32 values -> {hi0..3} and {lo0..3} as two 8-byte words
__device__ __forceinline__ void split4(u32x4 v, uint2& hi, uint2& lo) {
  f2 a, b;
  // no clamp: a value beyond the fp16 range converts to inf, its residual to -inf / NaN, and the products to NaN - an
  // out-of-range or NaN activation shows in the output instead of being replaced by a finite number
  a.x = __uint_as_float(v.x); a.y = __uint_as_float(v.y);
  b.x = __uint_as_float(v.z); b.y = __uint_as_float(v.w);
  const h2 ha = __builtin_convertvector(a, h2), hb = __builtin_convertvector(b, h2);
  f2 ra, rb;
  ra.x = a.x - (float)ha.x; ra.y = a.y - (float)ha.y; rb.x = b.x - (float)hb.x; rb.y = b.y - (float)hb.y;
  const h2 la = __builtin_convertvector(ra, h2), lb = __builtin_convertvector(rb, h2);
  hi = make_uint2(__builtin_bit_cast(uint32_t, ha), __builtin_bit_cast(uint32_t, hb));
  lo = make_uint2(__builtin_bit_cast(uint32_t, la), __builtin_bit_cast(uint32_t, lb));
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }

__global__ __launch_bounds__(256) void dense_linear_kernel(LinearParams P) {
  __shared__ __attribute__((aligned(16))) uint8_t smem[2 * BUF];
  const SdfDenseLinearDesc& d = P.d;
  const int M = d.M, N = d.N, K = d.K;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  // column tiles fastest: the A tile is shared through L2 - which is per XCD, and consecutive workgroup ids go round the 8 XCDs: ids are
  // re-dealt so that consecutive LOGICAL ids are neighbours on one XCD (round 5; SDF_DENSE_LINEAR_XCD=0 in the launcher: the old order)
  const int G = gridDim.x;
  const int wg = (P.xcd && (G & 7) == 0) ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int tn = wg % P.tiles_n, tm = wg / P.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  // convolution form: row m = (img, oy, ox) of a 3x3 / pad 1 / stride cv_stride convolution over a channels-last fp32 image,
  // k = (ky * 3 + kx) * C + c: a chunk of 32 k is 128 contiguous bytes of one tap (C % 32 == 0), zero outside the image
  const bool conv = d.cv_C > 0;
  const int cvH = d.cv_H, cvW = d.cv_W, cvC = d.cv_C, ohw = d.cv_OH * d.cv_OW;
  const __amdgpu_buffer_rsrc_t A_rs = rsrc(d.a, conv ? (uint32_t)(M / ohw) * (uint32_t)(cvH * cvW) * (uint32_t)cvC * 4u
                                                     : (uint32_t)M * (uint32_t)K * 4u);
  const __amdgpu_buffer_rsrc_t W_rs = rsrc(d.w, 2u * (uint32_t)N * (uint32_t)K * 2u);

  // loader pieces of this thread.  A: 128 rows x 8 float4 per chunk, 4 per thread; W: 2 planes x 96 rows x 4 sixteen-byte pieces, 3 per thread
  uint32_t a_off[4], a_lds[4], w_off[3], w_lds[3];
  int a_y[4], a_x[4];                                                  // convolution form: input row / column of tap (0, 0)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = tid + 256 * i, row = p >> 3, c4 = p & 7;
    a_lds[i] = (uint32_t)(row * RS + c4 * 8);
    a_y[i] = a_x[i] = 0;
    if (!conv) {
      a_off[i] = m0 + row < M ? (uint32_t)((m0 + row) * K + c4 * 4) * 4u : INV;
    } else if (m0 + row < M) {
      const int img = (m0 + row) / ohw, pix = (m0 + row) - img * ohw;
      const int oy = pix / d.cv_OW, ox = pix - oy * d.cv_OW;
      a_y[i] = oy * d.cv_stride - 1; a_x[i] = ox * d.cv_stride - 1;
      a_off[i] = (uint32_t)(img * cvH * cvW * cvC + c4 * 4) * 4u;       // image base + the piece's channel offset
    } else {
      a_off[i] = INV;
    }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int q = tid + 256 * i, pl = q / (BN * 4), r = (q - pl * BN * 4) >> 2, c = q & 3;
    w_off[i] = (uint32_t)((pl * N + n0 + r) * K + c * 8) * 2u;
    w_lds[i] = (uint32_t)(2 * A_PLANE + pl * B_PLANE + r * RS + c * 16);
  }
  u32x4 areg[4], wreg[3];
  auto request = [&](int kc) __attribute__((always_inline)) {
    if (conv) {
      const int tap = (kc * KC) / cvC, c0 = kc * KC - tap * cvC, ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int iy = a_y[i] + ky, ix = a_x[i] + kx;
        const bool ok = a_off[i] != INV && (unsigned)iy < (unsigned)cvH && (unsigned)ix < (unsigned)cvW;
        areg[i] = __builtin_amdgcn_raw_buffer_load_b128(A_rs, ok ? a_off[i] + (uint32_t)((iy * cvW + ix) * cvC + c0) * 4u : INV, 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) areg[i] = __builtin_amdgcn_raw_buffer_load_b128(A_rs, a_off[i] != INV ? a_off[i] + (uint32_t)kc * (KC * 4) : INV, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) wreg[i] = __builtin_amdgcn_raw_buffer_load_b128(W_rs, w_off[i] + (uint32_t)kc * (KC * 2), 0, 0);
  };
  auto deposit = [&](uint8_t* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint2 hi, lo;
      split4(areg[i], hi, lo);
      *reinterpret_cast<uint2*>(buf + a_lds[i]) = hi;
      *reinterpret_cast<uint2*>(buf + A_PLANE + a_lds[i]) = lo;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) *reinterpret_cast<u32x4*>(buf + w_lds[i]) = wreg[i];
  };

  int ln = lane;
  asm volatile("" : "+v"(ln));
  const int l31 = ln & 31, lh = ln >> 5;
  const uint32_t a_frag = (uint32_t)((wave * 32 + l31) * RS + 16 * lh);
  const uint32_t w_frag = (uint32_t)(2 * A_PLANE + l31 * RS + 16 * lh);

  f32x16 acc[3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

  const int nk = K / KC;
  request(0);
  deposit(smem);
  __syncthreads();
#pragma unroll 1
  for (int kc = 0; kc < nk; ++kc) {
    uint8_t* cur = smem + (kc & 1) * BUF;
    if (kc + 1 < nk) request(kc + 1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const f16x8 ah = *reinterpret_cast<const f16x8*>(cur + a_frag + s * 32);
      const f16x8 al = *reinterpret_cast<const f16x8*>(cur + A_PLANE + a_frag + s * 32);
      f16x8 wh[3], wl[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        wh[j] = *reinterpret_cast<const f16x8*>(cur + w_frag + j * 32 * RS + s * 32);
        wl[j] = *reinterpret_cast<const f16x8*>(cur + B_PLANE + w_frag + j * 32 * RS + s * 32);
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j], al, acc[j], 0, 0, 0);   // small terms first
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[j], ah, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j], ah, acc[j], 0, 0, 0);
      }
    }
    if (kc + 1 < nk) deposit(smem + ((kc + 1) & 1) * BUF);
    __syncthreads();
  }

  // epilogue: accumulator quad q of block j = columns n0 + 32j + 8q + 4lh .. + 3 of row m0 + 32*wave + l31
  int m = m0 + wave * 32 + l31;
  const bool m_ok = m < M;
  if (conv && d.out_T > 1 && m_ok) {                                   // images arrive (t, b)-major and leave (b, t)-major
    const int img = m / ohw, pix = m - img * ohw, nb = (M / ohw) / d.out_T;
    m = ((img % nb) * d.out_T + img / nb) * ohw + pix;
  }
  const __amdgpu_buffer_rsrc_t C_rs = rsrc(d.out, (uint32_t)M * (uint32_t)N * 4u);
  const __amdgpu_buffer_rsrc_t R_rs = rsrc(d.resid, (uint32_t)M * (uint32_t)N * 4u);
  const bool gelu = d.gelu != 0, has_res = d.resid != nullptr;
  const float asc = d.acc_scale != 0.f ? d.acc_scale : 1.f;             // 1 / weight scale (a power of two)
#pragma unroll
  for (int j = 0; j < 3; ++j) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = n0 + 32 * j + 8 * q + 4 * lh;
      const uint32_t off = m_ok ? (uint32_t)(m * N + n) * 4u : INV;
      float4 o = make_float4(acc[j][4 * q + 0] * asc, acc[j][4 * q + 1] * asc, acc[j][4 * q + 2] * asc, acc[j][4 * q + 3] * asc);
      if (d.bias) {
        const float4 b = *reinterpret_cast<const float4*>(d.bias + n);
        o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w;
      }
      if (gelu) { o.x = gelu_erf(o.x); o.y = gelu_erf(o.y); o.z = gelu_erf(o.z); o.w = gelu_erf(o.w); }
      if (has_res) {
        const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(R_rs, off, 0, 0);
        o.x += __uint_as_float(r.x); o.y += __uint_as_float(r.y); o.z += __uint_as_float(r.z); o.w += __uint_as_float(r.w);
      }
      u32x4 st;
      st.x = __float_as_uint(o.x); st.y = __float_as_uint(o.y); st.z = __float_as_uint(o.z); st.w = __float_as_uint(o.w);
      __builtin_amdgcn_raw_buffer_store_b128(st, C_rs, off, 0, 0);
    }
  }
}

}  // namespace

extern "C" int sdf_dense_linear_fwd(const SdfDenseLinearDesc* d, void* stream) {
  if (!d || !d->a || !d->w || !d->out) return SDF_E_NULL;
  if (d->M <= 0 || d->N <= 0 || d->K <= 0 || d->N % BN || d->K % KC) return SDF_E_SHAPE;
  if (d->acc_scale != 0.f) {
    int ex;
    if (!(d->acc_scale > 0.f) || frexpf(d->acc_scale, &ex) != 0.5f) return SDF_E_DTYPE;      // a power of two (exact rescaling)
  }
  if (d->cv_C > 0) {
    if (d->cv_C % KC || d->K != 9 * d->cv_C || d->cv_H <= 0 || d->cv_W <= 0 || d->cv_stride <= 0 || d->cv_OH <= 0 || d->cv_OW <= 0) return SDF_E_SHAPE;
    if (d->cv_OH != (d->cv_H - 1) / d->cv_stride + 1 || d->cv_OW != (d->cv_W - 1) / d->cv_stride + 1) return SDF_E_SHAPE;   // 3x3, pad 1
    if (d->M % (d->cv_OH * d->cv_OW) || d->resid) return SDF_E_SHAPE;
    const int imgs = d->M / (d->cv_OH * d->cv_OW);
    if (d->out_T > 1 && imgs % d->out_T) return SDF_E_SHAPE;
    if ((int64_t)imgs * d->cv_H * d->cv_W * d->cv_C * 4 >= ((int64_t)1 << 31)) return SDF_E_SHAPE;
  }
  const int64_t lim = (int64_t)1 << 31;
  if ((d->cv_C == 0 && (int64_t)d->M * d->K * 4 >= lim) || (int64_t)d->M * d->N * 4 >= lim || (int64_t)d->N * d->K * 4 >= lim) return SDF_E_SHAPE;
  if (!sdf_aligned(d->a, 16) || !sdf_aligned(d->w, 16) || !sdf_aligned(d->out, 16) || (d->resid && !sdf_aligned(d->resid, 16)) ||
      (d->bias && !sdf_aligned(d->bias, 16)))
    return SDF_E_ALIGN;
  LinearParams P;
  P.d = *d;
  P.tiles_n = d->N / BN;
  const bool xcd = [] { const char* e = sdf_sw(SW_DENSE_LINEAR_XCD); return !e || e[0] != '0'; }();
  P.xcd = xcd ? 1 : 0;
  const int64_t tiles = (int64_t)((d->M + BM - 1) / BM) * P.tiles_n;
  SDF_LAUNCH(dense_linear_kernel, dim3((unsigned)tiles), dim3(256), 0, sdf_stream(stream), P);
  SDF_LAUNCH_CHECK();
  return 0;
}
