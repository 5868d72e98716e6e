// Eval-BatchNorm affine + residual add, the glue between MIOpen convolutions and the neuron kernels:
//   out = fmaf(x, alpha[c], beta[c]) (+ resid),  c = (i / inner) % C
// (reference: SpikingNormLayer after a conv + the MS shortcut, Spiking_modules.py:922-926, 816-818).
// HBM-bound: one 16-byte load per operand per lane, 16-byte store.
#include "common.h"

namespace {
__global__ __launch_bounds__(256) void affine_resid_kernel(const float* __restrict__ x, const float* __restrict__ alpha,
                                                           const float* __restrict__ beta, const float* resid, float* out,
                                                           int64_t quads, int C, int64_t inner) {
  int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= quads) return;
  int64_t i = q * 4;
  float4 v = *reinterpret_cast<const float4*>(x + i);
  float4 a, b;
  if (inner == 1) {
    int c = (int)(i % C);
    a = *reinterpret_cast<const float4*>(alpha + c);
    b = *reinterpret_cast<const float4*>(beta + c);
  } else {
    int c = (int)((i / inner) % C);
    float aa = alpha[c], bb = beta[c];
    a = make_float4(aa, aa, aa, aa);
    b = make_float4(bb, bb, bb, bb);
  }
  v.x = __builtin_fmaf(v.x, a.x, b.x);
  v.y = __builtin_fmaf(v.y, a.y, b.y);
  v.z = __builtin_fmaf(v.z, a.z, b.z);
  v.w = __builtin_fmaf(v.w, a.w, b.w);
  if (resid) {
    float4 r = *reinterpret_cast<const float4*>(resid + i);
    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
  }
  *reinterpret_cast<float4*>(out + i) = v;
}
}  // namespace

extern "C" int sdf_affine_resid_fwd(const float* x, const float* alpha, const float* beta, const float* resid, float* out,
                                    int64_t n, int C, int64_t inner, void* stream) {
  if (!x || !alpha || !beta || !out) return SDF_E_NULL;
  if (n < 4 || n % 4 || C < 1 || inner < 1) return SDF_E_SHAPE;
  if (inner == 1 ? (C % 4 != 0) : (inner % 4 != 0)) return SDF_E_SHAPE;
  if (!sdf_aligned(x, 16) || !sdf_aligned(out, 16) || (resid && !sdf_aligned(resid, 16))) return SDF_E_ALIGN;
  if (inner == 1 && (!sdf_aligned(alpha, 16) || !sdf_aligned(beta, 16))) return SDF_E_ALIGN;
  int64_t quads = n / 4;
  SDF_LAUNCH(affine_resid_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, sdf_stream(stream), x,
                     alpha, beta, resid, out, quads, C, inner);
  SDF_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
// Window partition / reverse for the TRAINING path as index arithmetic (the inference path folds the same table into the
// neuron kernel's gather and the GEMM's scatter): rows of a channel-last fp32 buffer moved through the int32 slice map of
// sdf_window_slice_map - no materialised pad, roll, permute or crop (reference Spiking_swin_transformer3D.py:789-820).
//   gather : out[i, :] = map[i] >= 0 ? x[map[i], :] : 0          (pad + roll(-shift) + window_partition_v2, and the
//                                                                  backward of `scatter`)
//   scatter: out[map[i], :] = y[i, :] for map[i] >= 0            (window_reverse + roll(+shift) + crop, and the backward
//            of `gather`; every source row occurs exactly once in the map, so there is no accumulation and no atomic)
namespace {
template <bool SCATTER>
__global__ __launch_bounds__(256) void rows_move_kernel(const float* __restrict__ src, const int32_t* __restrict__ map,
                                                        float* __restrict__ dst, int64_t M, int Q) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;     // float4 index into the (M, C) side
  if (i >= M * Q) return;
  const int64_t row = i / Q;
  const int q = (int)(i - row * Q);
  const int32_t m = map[row];
  if (SCATTER) {
    if (m >= 0) *reinterpret_cast<float4*>(dst + ((int64_t)m * Q + q) * 4) = *reinterpret_cast<const float4*>(src + i * 4);
  } else {
    *reinterpret_cast<float4*>(dst + i * 4) =
        m >= 0 ? *reinterpret_cast<const float4*>(src + ((int64_t)m * Q + q) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
}  // namespace

extern "C" int sdf_rows_gather_fwd(const float* x, const int32_t* map, float* out, int64_t M, int C, void* stream) {
  if (!x || !map || !out) return SDF_E_NULL;
  if (M < 1 || C < 4 || C % 4) return SDF_E_SHAPE;
  if (!sdf_aligned(x, 16) || !sdf_aligned(out, 16)) return SDF_E_ALIGN;
  SDF_LAUNCH((rows_move_kernel<false>), dim3((unsigned)((M * (C / 4) + 255) / 256)), dim3(256), 0, sdf_stream(stream), x, map,
                     out, M, C / 4);
  SDF_LAUNCH_CHECK();
  return 0;
}

extern "C" int sdf_rows_scatter_fwd(const float* y, const int32_t* map, float* out, int64_t M, int C, void* stream) {
  if (!y || !map || !out) return SDF_E_NULL;
  if (M < 1 || C < 4 || C % 4) return SDF_E_SHAPE;
  if (!sdf_aligned(y, 16) || !sdf_aligned(out, 16)) return SDF_E_ALIGN;
  SDF_LAUNCH((rows_move_kernel<true>), dim3((unsigned)((M * (C / 4) + 255) / 256)), dim3(256), 0, sdf_stream(stream), y, map,
                     out, M, C / 4);
  SDF_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// LayerNorm over the last dim of (rows, C) fp32 (nn.LayerNorm of the ANN swin blocks, reference
// models/STSwinNet/swin_transformer3D_v2.py:231-233, 272, 312, 356, 622-624): L lanes per row hold the row in registers (V float4
// each), mean and centred variance by xor shuffles inside the L lanes, one read and one write of the tensor.
namespace {
template <int L, int V>
__global__ __launch_bounds__(256) void layer_norm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ out, int64_t rows, int C,
                                                         float eps) {
  const int sub = threadIdx.x % L;
  const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) / L;
  if (row >= rows) return;                                            // whole L-lane groups leave together (256 % L == 0)
  const float* xr = x + row * C;
  float4 v[V];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int c = (i * L + sub) * 4;
    v[i] = c < C ? *reinterpret_cast<const float4*>(xr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
#pragma unroll
  for (int m = L / 2; m >= 1; m >>= 1) s += __shfl_xor(s, m, L);
  const float mean = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int c = (i * L + sub) * 4;
    if (c < C) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  }
#pragma unroll
  for (int m = L / 2; m >= 1; m >>= 1) q += __shfl_xor(q, m, L);
  const float rstd = 1.f / sqrtf(q / (float)C + eps);
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int c = (i * L + sub) * 4;
    if (c < C) {
      const float4 g = *reinterpret_cast<const float4*>(gamma + c), b = *reinterpret_cast<const float4*>(beta + c);
      float4 o;
      o.x = (v[i].x - mean) * rstd * g.x + b.x; o.y = (v[i].y - mean) * rstd * g.y + b.y;
      o.z = (v[i].z - mean) * rstd * g.z + b.z; o.w = (v[i].w - mean) * rstd * g.w + b.w;
      *reinterpret_cast<float4*>(out + row * C + c) = o;
    }
  }
}

template <int L>
int launch_layer_norm(const float* x, const float* g, const float* b, float* out, int64_t rows, int C, float eps, hipStream_t s) {
  const int V = (C + 4 * L - 1) / (4 * L);
  const dim3 grid((unsigned)((rows * L + 255) / 256));
  switch (V) {
    case 1: SDF_LAUNCH((layer_norm_kernel<L, 1>), grid, dim3(256), 0, s, x, g, b, out, rows, C, eps); break;
    case 2: SDF_LAUNCH((layer_norm_kernel<L, 2>), grid, dim3(256), 0, s, x, g, b, out, rows, C, eps); break;
    case 3: SDF_LAUNCH((layer_norm_kernel<L, 3>), grid, dim3(256), 0, s, x, g, b, out, rows, C, eps); break;
    case 4: SDF_LAUNCH((layer_norm_kernel<L, 4>), grid, dim3(256), 0, s, x, g, b, out, rows, C, eps); break;
    case 5: case 6: SDF_LAUNCH((layer_norm_kernel<L, 6>), grid, dim3(256), 0, s, x, g, b, out, rows, C, eps); break;
    case 7: case 8: SDF_LAUNCH((layer_norm_kernel<L, 8>), grid, dim3(256), 0, s, x, g, b, out, rows, C, eps); break;
    default: return SDF_E_SHAPE;
  }
  return 0;
}
}  // namespace

extern "C" int sdf_layer_norm_fwd(const float* x, const float* gamma, const float* beta, float* out, int64_t rows, int C, float eps,
                                  void* stream) {
  if (!x || !gamma || !beta || !out) return SDF_E_NULL;
  if (rows <= 0 || C <= 0 || C % 4 || C > 2048) return SDF_E_SHAPE;
  if (!sdf_aligned(x, 16) || !sdf_aligned(out, 16) || !sdf_aligned(gamma, 16) || !sdf_aligned(beta, 16)) return SDF_E_ALIGN;
  const int rc = C <= 512 ? launch_layer_norm<16>(x, gamma, beta, out, rows, C, eps, sdf_stream(stream))
                          : launch_layer_norm<64>(x, gamma, beta, out, rows, C, eps, sdf_stream(stream));
  if (rc) return rc;
  SDF_LAUNCH_CHECK();
  return 0;
}
