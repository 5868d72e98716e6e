// 1x1 strided convolution of a REAL-valued fp32 NHWC image on the exact fp32 matrix pipe (gfx950):
//
//   out[img, oy, ox, n] = sum_c x[img, oy * s, ox * s, c] * w[n, c] (+ bias[n])
//
// the membrane shortcut of the stride-2 patch-embedding projection (SpikingPEDLayer.conv_res on the fp32 membrane,
// reference Spiking_modules.py:772-826, call :819) - the one layer of the SNN forward whose input is not spikes.  It ran on
// MIOpen's implicit-GEMM kernel + a layout copy (43 us); here each wave multiplies 32 output pixels against all N channels
// with v_mfma_f32_32x32x2_f32 (exact fp32 products and sums: the numerics of an fmaf chain, no 16-bit split):
//   * the WEIGHTS are the MFMA's row operand (LDS-resident, row pitch Cin + 1 floats: conflict-free ds_read_b32), the pixels
//     its column operand, so a lane's accumulator quads are four consecutive channels of one pixel -> 16-byte stores;
//   * the K dimension is dealt to the two lane halves as contiguous halves (k-step ks multiplies channels ks and Cin/2 + ks):
//     a lane loads the Cin/2 channels it will ever need of its pixel as Cin/8 float4 from global memory, once.
#include "common.h"

namespace {
typedef __attribute__((ext_vector_type(16))) float f32x16;

struct PwParams { SdfPointwiseConvDesc d; };

template <int CIN, int NBLK>
__global__ __launch_bounds__(256) void pointwise_conv_f32_kernel(PwParams P) {
  constexpr int KH = CIN / 2, WP = CIN + 1, N = 32 * NBLK;
  const SdfPointwiseConvDesc& d = P.d;
  __shared__ float W_s[N * WP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  for (int i = tid; i < N * CIN; i += 256) {
    const int n = i / CIN, c = i - n * CIN;
    W_s[n * WP + c] = d.w[i];
  }
  const int64_t M = (int64_t)d.imgs * d.OH * d.OW;
  const int64_t m = ((int64_t)blockIdx.x * 4 + wave) * 32 + l31;       // this lane's output pixel
  const bool ok = m < M;
  const int64_t mc = ok ? m : 0;
  const int64_t img = mc / ((int64_t)d.OH * d.OW), r = mc - img * (int64_t)d.OH * d.OW;
  const int oy = (int)(r / d.OW), ox = (int)(r - (int64_t)oy * d.OW);
  const float* src = d.x + ((img * d.H + (int64_t)oy * d.stride) * d.W + (int64_t)ox * d.stride) * CIN + KH * lh;
  float4 xv[KH / 4];
#pragma unroll
  for (int i = 0; i < KH / 4; ++i) xv[i] = *reinterpret_cast<const float4*>(src + 4 * i);
  __syncthreads();
  f32x16 acc[NBLK];
#pragma unroll
  for (int cb = 0; cb < NBLK; ++cb)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[cb][e] = 0.f;
#pragma unroll
  for (int ks = 0; ks < KH; ++ks) {
    const float4 q = xv[ks >> 2];
    const float b = (ks & 3) == 0 ? q.x : ((ks & 3) == 1 ? q.y : ((ks & 3) == 2 ? q.z : q.w));
#pragma unroll
    for (int cb = 0; cb < NBLK; ++cb) {
      const float a = W_s[(32 * cb + l31) * WP + ks + KH * lh];
      acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[cb], 0, 0, 0);
    }
  }
  if (!ok) return;
  float* dst = d.out + m * N;
#pragma unroll
  for (int cb = 0; cb < NBLK; ++cb)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int n = 32 * cb + 8 * q4 + 4 * lh;
      float4 o = make_float4(acc[cb][4 * q4 + 0], acc[cb][4 * q4 + 1], acc[cb][4 * q4 + 2], acc[cb][4 * q4 + 3]);
      if (d.bias) {
        const float4 bv = *reinterpret_cast<const float4*>(d.bias + n);
        o.x += bv.x; o.y += bv.y; o.z += bv.z; o.w += bv.w;
      }
      *reinterpret_cast<float4*>(dst + n) = o;
    }
}
}  // namespace

extern "C" int sdf_pointwise_conv_f32_fwd(const SdfPointwiseConvDesc* d, void* stream) {
  if (!d || !d->x || !d->w || !d->out) return SDF_E_NULL;
  if (d->imgs < 1 || d->H < 1 || d->W < 1 || d->stride < 1) return SDF_E_SHAPE;
  if (d->OH != (d->H - 1) / d->stride + 1 || d->OW != (d->W - 1) / d->stride + 1) return SDF_E_SHAPE;
  if (d->Cin != 96 || (d->N != 96 && d->N != 192)) return SDF_E_SHAPE;              // instantiations: the 96-channel patch embedding
  if (!sdf_aligned(d->x, 16) || !sdf_aligned(d->out, 16) || !sdf_aligned(d->w, 4) || (d->bias && !sdf_aligned(d->bias, 16))) return SDF_E_ALIGN;
  const int64_t M = (int64_t)d->imgs * d->OH * d->OW, wgs = (M + 127) / 128;
  if (wgs >= (1LL << 31)) return SDF_E_SHAPE;
  PwParams P;
  P.d = *d;
  hipStream_t s = sdf_stream(stream);
  if (d->N == 96) SDF_LAUNCH((pointwise_conv_f32_kernel<96, 3>), dim3((unsigned)wgs), dim3(256), 0, s, P);
  else SDF_LAUNCH((pointwise_conv_f32_kernel<96, 6>), dim3((unsigned)wgs), dim3(256), 0, s, P);
  SDF_LAUNCH_CHECK();
  return 0;
}
