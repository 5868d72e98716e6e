// Batch-statistics BatchNorm over the LAST dim of a channel-last (R, C) fp32 buffer, forward and backward - the training
// form of row a3 (reference: SpikingNormLayer "BN" in train mode = spikingjelly layer.BatchNorm2d multi-step ->
// nn.BatchNorm2d on the (T*B, C, H, W) view the reference makes with `permute(0, 1, 4, 2, 3)`, Spiking_modules.py:101-146,
// Spiking_swin_transformer3D.py:172, 178, 673, 677, 714, 972).  The library BatchNorm wants channels second: on this
// framework's channel-last activations that costs a permute copy before and after every call and again in backward (240 of
// the 453 copy launches of a training step).  Here the reduction runs down the rows of the (R, C) matrix in place.
//   stats   : per-channel sum and sum of squares in fp64 (row-strided threads, LDS tree, per-block partials, fixed-order finish)
//             -> mean, biased var, invstd = 1 / sqrt(var + eps); running_mean / running_var updated with the unbiased variance
//   forward : y = (x - mean) * invstd * w + b                       (the operation order of ATen's CPU/GPU train kernels)
//   backward: gb = sum gy, gw = sum gy * xhat;  gx = (gy - gb / R - xhat * gw / R) * invstd * w
// All passes are HBM-bound streams of float4; nothing is saved by the forward except mean / invstd (2 C floats).
#include "common.h"

namespace {

constexpr int BN_BLOCKS = 512;     // two workgroups per CU for the reductions

struct BnParams {
  const float* x; const float* gy; float* y; float* gx;
  const float* w; const float* b;
  const float* mean; const float* invstd;
  double* partial;                 // [nblk][2][C]
  int64_t R; int C;                // channel-last: R rows of C; NCHW: R = N * HW elements per channel
  int64_t NB; int HW;              // NCHW only: N images, HW = H * W (HW % 4 == 0)
};

// per-channel sums of (a, a*a) [stats] or (gy, gy * xhat) [backward] over a row-strided slice; partial[blk][0|1][c]
template <bool BWD>
__global__ __launch_bounds__(256) void bn_reduce_kernel(BnParams P) {
  extern __shared__ double sm[];                                  // [RL][2][4 * QB]
  const int Q = P.C / 4;                                          // channel quads
  const int QB = Q < 64 ? Q : 64;                                 // quads per channel block
  const int RL = 256 / QB;                                        // row lanes
  const int cq = threadIdx.x % QB, rl = threadIdx.x / QB;
  for (int q0 = 0; q0 < Q; q0 += QB) {
    const int q = q0 + cq;
    double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
    if (rl < RL && q < Q) {
      float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), is = mu;
      if (BWD) {
        mu = *reinterpret_cast<const float4*>(P.mean + 4 * q);
        is = *reinterpret_cast<const float4*>(P.invstd + 4 * q);
      }
      auto accum = [&](const float4& a, const float4& g) __attribute__((always_inline)) {
        if (!BWD) {
          s0[0] += a.x; s0[1] += a.y; s0[2] += a.z; s0[3] += a.w;
          s1[0] += (double)a.x * a.x; s1[1] += (double)a.y * a.y; s1[2] += (double)a.z * a.z; s1[3] += (double)a.w * a.w;
        } else {
          s0[0] += g.x; s0[1] += g.y; s0[2] += g.z; s0[3] += g.w;
          s1[0] += (double)g.x * ((a.x - mu.x) * is.x); s1[1] += (double)g.y * ((a.y - mu.y) * is.y);
          s1[2] += (double)g.z * ((a.z - mu.z) * is.z); s1[3] += (double)g.w * ((a.w - mu.w) * is.w);
        }
      };
      // four rows' loads in flight per thread (one at a time left 2 MB in flight on the whole chip: a latency-bound stream); the
      // additions keep their order, the sums their bits
      const int64_t step = (int64_t)gridDim.x * RL;
      int64_t r = (int64_t)blockIdx.x * RL + rl;
      for (; r + 3 * step < P.R; r += 4 * step) {
        float4 a[4], g[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          a[u] = *reinterpret_cast<const float4*>(P.x + (r + u * step) * P.C + 4 * q);
          g[u] = BWD ? *reinterpret_cast<const float4*>(P.gy + (r + u * step) * P.C + 4 * q) : a[u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) accum(a[u], g[u]);
      }
      for (; r < P.R; r += step) {
        const float4 a = *reinterpret_cast<const float4*>(P.x + r * P.C + 4 * q);
        const float4 g = BWD ? *reinterpret_cast<const float4*>(P.gy + r * P.C + 4 * q) : a;
        accum(a, g);
      }
    }
    __syncthreads();
    if (rl < RL) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        sm[(rl * 2 + 0) * 4 * QB + 4 * cq + j] = s0[j];
        sm[(rl * 2 + 1) * 4 * QB + 4 * cq + j] = s1[j];
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * 4 * QB; i += 256) {         // fixed-order sum over the row lanes
      const int which = i / (4 * QB), c = i % (4 * QB);
      if (4 * q0 + c < P.C) {
        double t = 0;
        for (int l = 0; l < RL; ++l) t += sm[(l * 2 + which) * 4 * QB + c];
        P.partial[((int64_t)blockIdx.x * 2 + which) * P.C + 4 * q0 + c] = t;
      }
    }
  }
}

// one workgroup per channel: strided fp64 sums of the per-block partials + a fixed-shape LDS tree (run-to-run bit-equal)
template <bool BWD>
__global__ __launch_bounds__(64) void bn_finish_kernel(const double* partial, int nblk, int C, int64_t R, float eps, float momentum,
                                                       float* out0, float* out1, float* running_mean, float* running_var) {
  __shared__ double sm[2][64];
  const int c = blockIdx.x;
  double s = 0, ss = 0;
  for (int b = threadIdx.x; b < nblk; b += 64) { s += partial[((int64_t)b * 2) * C + c]; ss += partial[((int64_t)b * 2 + 1) * C + c]; }
  sm[0][threadIdx.x] = s;
  sm[1][threadIdx.x] = ss;
  __syncthreads();
  for (int o = 32; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o) { sm[0][threadIdx.x] += sm[0][threadIdx.x + o]; sm[1][threadIdx.x] += sm[1][threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  s = sm[0][0]; ss = sm[1][0];
  if (BWD) {
    out0[c] = (float)ss;                                          // grad_weight = sum gy * xhat
    out1[c] = (float)s;                                           // grad_bias   = sum gy
    return;
  }
  const double m = s / (double)R;
  double var = ss / (double)R - m * m;
  if (var < 0) var = 0;
  out0[c] = (float)m;
  out1[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    const double unb = R > 1 ? var * (double)R / (double)(R - 1) : var;
    running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * m);
    running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unb);
  }
}

template <bool BWD>
__global__ __launch_bounds__(256) void bn_apply_kernel(BnParams P, const float* gw, const float* gb) {
  const int Q = P.C / 4;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= P.R * Q) return;
  const int q = (int)(i % Q);
  const float4 a = *reinterpret_cast<const float4*>(P.x + i * 4);
  const float4 mu = *reinterpret_cast<const float4*>(P.mean + 4 * q), is = *reinterpret_cast<const float4*>(P.invstd + 4 * q);
  const float4 w = *reinterpret_cast<const float4*>(P.w + 4 * q);
  float4 o;
  if (!BWD) {
    const float4 b = *reinterpret_cast<const float4*>(P.b + 4 * q);
    o.x = (a.x - mu.x) * is.x * w.x + b.x; o.y = (a.y - mu.y) * is.y * w.y + b.y;
    o.z = (a.z - mu.z) * is.z * w.z + b.z; o.w = (a.w - mu.w) * is.w * w.w + b.w;
    *reinterpret_cast<float4*>(P.y + i * 4) = o;
  } else {
    const float4 g = *reinterpret_cast<const float4*>(P.gy + i * 4);
    const float4 sw = *reinterpret_cast<const float4*>(gw + 4 * q), sb = *reinterpret_cast<const float4*>(gb + 4 * q);
    const float inv_r = 1.0f / (float)P.R;
    o.x = (g.x - sb.x * inv_r - (a.x - mu.x) * is.x * (sw.x * inv_r)) * is.x * w.x;
    o.y = (g.y - sb.y * inv_r - (a.y - mu.y) * is.y * (sw.y * inv_r)) * is.y * w.y;
    o.z = (g.z - sb.z * inv_r - (a.z - mu.z) * is.z * (sw.z * inv_r)) * is.z * w.z;
    o.w = (g.w - sb.w * inv_r - (a.w - mu.w) * is.w * (sw.w * inv_r)) * is.w * w.w;
    *reinterpret_cast<float4*>(P.gx + i * 4) = o;
  }
}

// ---- NCHW layout (the conv outputs of the patch embedding / U-Net: (T*B, C, H, W)) ----
// reduce: a workgroup walks (image, channel) planes, sums each plane with float4 lanes + wave shuffles, and accumulates the
// plane sums per channel in LDS; partial[blk][0|1][c] as in the channel-last kernel.
template <bool BWD>
__global__ __launch_bounds__(256) void bn_reduce_nchw_kernel(BnParams P) {
  extern __shared__ double sm[];                                  // [2][C] + [4][2] scratch
  double* acc = sm;
  double* wsum = sm + 2 * P.C;
  for (int i = threadIdx.x; i < 2 * P.C; i += 256) acc[i] = 0;
  __syncthreads();
  const int64_t planes = P.NB * P.C;
  const int Q = P.HW / 4;
  for (int64_t pl = blockIdx.x; pl < planes; pl += gridDim.x) {
    const int c = (int)(pl % P.C);
    const float* xp = P.x + pl * P.HW;
    const float* gp = BWD ? P.gy + pl * P.HW : nullptr;
    const float mu = BWD ? P.mean[c] : 0.f, is = BWD ? P.invstd[c] : 0.f;
    double s0 = 0, s1 = 0;
    auto accum = [&](const float4& a, const float4& g) __attribute__((always_inline)) {
      if (!BWD) {
        s0 += ((double)a.x + a.y) + ((double)a.z + a.w);
        s1 += ((double)a.x * a.x + (double)a.y * a.y) + ((double)a.z * a.z + (double)a.w * a.w);
      } else {
        s0 += ((double)g.x + g.y) + ((double)g.z + g.w);
        s1 += ((double)g.x * ((a.x - mu) * is) + (double)g.y * ((a.y - mu) * is)) +
              ((double)g.z * ((a.z - mu) * is) + (double)g.w * ((a.w - mu) * is));
      }
    };
    int q = threadIdx.x;                                            // four loads in flight per thread, additions in the same order
    for (; q + 768 < Q; q += 1024) {
      float4 a[4], g[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = *reinterpret_cast<const float4*>(xp + 4 * (q + 256 * u));
        g[u] = BWD ? *reinterpret_cast<const float4*>(gp + 4 * (q + 256 * u)) : a[u];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) accum(a[u], g[u]);
    }
    for (; q < Q; q += 256) {
      const float4 a = *reinterpret_cast<const float4*>(xp + 4 * q);
      const float4 g = BWD ? *reinterpret_cast<const float4*>(gp + 4 * q) : a;
      accum(a, g);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { wsum[wave * 2] = s0; wsum[wave * 2 + 1] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
      acc[c] += (wsum[0] + wsum[2]) + (wsum[4] + wsum[6]);
      acc[P.C + c] += (wsum[1] + wsum[3]) + (wsum[5] + wsum[7]);
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < 2 * P.C; i += 256) P.partial[(int64_t)blockIdx.x * 2 * P.C + i] = acc[i];
}

template <bool BWD>
__global__ __launch_bounds__(256) void bn_apply_nchw_kernel(BnParams P, const float* gw, const float* gb) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;     // float4 index
  const int Q = P.HW / 4;
  if (i >= P.NB * P.C * Q) return;
  const int c = (int)((i / Q) % P.C);
  const float4 a = *reinterpret_cast<const float4*>(P.x + i * 4);
  const float mu = P.mean[c], is = P.invstd[c], w = P.w[c];
  float4 o;
  if (!BWD) {
    const float b = P.b[c];
    o.x = (a.x - mu) * is * w + b; o.y = (a.y - mu) * is * w + b; o.z = (a.z - mu) * is * w + b; o.w = (a.w - mu) * is * w + b;
    *reinterpret_cast<float4*>(P.y + i * 4) = o;
  } else {
    const float4 g = *reinterpret_cast<const float4*>(P.gy + i * 4);
    const float inv_r = 1.0f / (float)P.R, sb = gb[c] * inv_r, sw = gw[c] * inv_r;
    o.x = (g.x - sb - (a.x - mu) * is * sw) * is * w; o.y = (g.y - sb - (a.y - mu) * is * sw) * is * w;
    o.z = (g.z - sb - (a.z - mu) * is * sw) * is * w; o.w = (g.w - sb - (a.w - mu) * is * sw) * is * w;
    *reinterpret_cast<float4*>(P.gx + i * 4) = o;
  }
}

int nblocks(int64_t R, int C) {
  const int QB = C / 4 < 64 ? C / 4 : 64, RL = 256 / QB;
  const int64_t need = (R + RL - 1) / RL;
  return (int)(need < BN_BLOCKS ? need : BN_BLOCKS);
}

size_t reduce_lds(int C) {
  const int QB = C / 4 < 64 ? C / 4 : 64, RL = 256 / QB;
  return (size_t)RL * 2 * 4 * QB * sizeof(double);
}

}  // namespace

extern "C" int64_t sdf_bn_train_workspace_bytes(int64_t R, int C) {
  if (R < 1 || C < 4 || C % 4) return 0;
  return (int64_t)BN_BLOCKS * 2 * C * (int64_t)sizeof(double);
}

extern "C" int sdf_bn_train_fwd(const float* x, const float* weight, const float* bias, float* y, float* save_mean, float* save_invstd,
                                float* running_mean, float* running_var, int64_t R, int C, float eps, float momentum, void* workspace,
                                int64_t workspace_bytes, void* stream) {
  if (!x || !weight || !bias || !y || !save_mean || !save_invstd || !workspace) return SDF_E_NULL;
  if (R < 1 || C < 4 || C % 4 || R * (int64_t)C >= (1LL << 40)) return SDF_E_SHAPE;
  if ((running_mean == nullptr) != (running_var == nullptr)) return SDF_E_NULL;
  if (workspace_bytes < sdf_bn_train_workspace_bytes(R, C)) return SDF_E_SHAPE;
  if (!sdf_aligned(x, 16) || !sdf_aligned(y, 16) || !sdf_aligned(weight, 16) || !sdf_aligned(bias, 16) || !sdf_aligned(save_mean, 16) ||
      !sdf_aligned(save_invstd, 16) || !sdf_aligned(workspace, 8))
    return SDF_E_ALIGN;
  BnParams P = {};
  P.x = x; P.y = y; P.w = weight; P.b = bias; P.mean = save_mean; P.invstd = save_invstd; P.partial = reinterpret_cast<double*>(workspace);
  P.R = R; P.C = C;
  hipStream_t s = sdf_stream(stream);
  const int nblk = nblocks(R, C);
  SDF_LAUNCH((bn_reduce_kernel<false>), dim3(nblk), dim3(256), reduce_lds(C), s, P);
  SDF_LAUNCH((bn_finish_kernel<false>), dim3(C), dim3(64), 0, s, P.partial, nblk, C, R, eps, momentum, save_mean, save_invstd,
                     running_mean, running_var);
  SDF_LAUNCH((bn_apply_kernel<false>), dim3((unsigned)((R * (C / 4) + 255) / 256)), dim3(256), 0, s, P, nullptr, nullptr);
  SDF_LAUNCH_CHECK();
  return 0;
}

extern "C" int sdf_bn_train_bwd(const float* x, const float* grad_y, const float* weight, const float* save_mean, const float* save_invstd,
                                float* grad_x, float* grad_weight, float* grad_bias, int64_t R, int C, void* workspace,
                                int64_t workspace_bytes, void* stream) {
  if (!x || !grad_y || !weight || !save_mean || !save_invstd || !grad_x || !grad_weight || !grad_bias || !workspace) return SDF_E_NULL;
  if (R < 1 || C < 4 || C % 4 || R * (int64_t)C >= (1LL << 40)) return SDF_E_SHAPE;
  if (workspace_bytes < sdf_bn_train_workspace_bytes(R, C)) return SDF_E_SHAPE;
  if (!sdf_aligned(x, 16) || !sdf_aligned(grad_y, 16) || !sdf_aligned(grad_x, 16) || !sdf_aligned(weight, 16) ||
      !sdf_aligned(save_mean, 16) || !sdf_aligned(save_invstd, 16) || !sdf_aligned(grad_weight, 16) || !sdf_aligned(grad_bias, 16) ||
      !sdf_aligned(workspace, 8))
    return SDF_E_ALIGN;
  BnParams P = {};
  P.x = x; P.gy = grad_y; P.gx = grad_x; P.w = weight; P.mean = save_mean; P.invstd = save_invstd;
  P.partial = reinterpret_cast<double*>(workspace); P.R = R; P.C = C;
  hipStream_t s = sdf_stream(stream);
  const int nblk = nblocks(R, C);
  SDF_LAUNCH((bn_reduce_kernel<true>), dim3(nblk), dim3(256), reduce_lds(C), s, P);
  SDF_LAUNCH((bn_finish_kernel<true>), dim3(C), dim3(64), 0, s, P.partial, nblk, C, R, 0.f, 0.f, grad_weight, grad_bias,
                     nullptr, nullptr);
  SDF_LAUNCH((bn_apply_kernel<true>), dim3((unsigned)((R * (C / 4) + 255) / 256)), dim3(256), 0, s, P, grad_weight, grad_bias);
  SDF_LAUNCH_CHECK();
  return 0;
}

// NCHW variants: x (N, C, H, W) contiguous, HW = H * W a multiple of 4; same statistics, same outputs, same workspace size.
extern "C" int sdf_bn_train_nchw_fwd(const float* x, const float* weight, const float* bias, float* y, float* save_mean,
                                     float* save_invstd, float* running_mean, float* running_var, int64_t N, int C, int HW, float eps,
                                     float momentum, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!x || !weight || !bias || !y || !save_mean || !save_invstd || !workspace) return SDF_E_NULL;
  if (N < 1 || C < 1 || HW < 4 || HW % 4 || N * (int64_t)C * HW >= (1LL << 40) || C > 2048) return SDF_E_SHAPE;
  if ((running_mean == nullptr) != (running_var == nullptr)) return SDF_E_NULL;
  if (workspace_bytes < (int64_t)BN_BLOCKS * 2 * C * (int64_t)sizeof(double)) return SDF_E_SHAPE;
  if (!sdf_aligned(x, 16) || !sdf_aligned(y, 16) || !sdf_aligned(workspace, 8)) return SDF_E_ALIGN;
  BnParams P = {};
  P.x = x; P.y = y; P.w = weight; P.b = bias; P.mean = save_mean; P.invstd = save_invstd; P.partial = reinterpret_cast<double*>(workspace);
  P.R = N * HW; P.C = C; P.NB = N; P.HW = HW;
  hipStream_t s = sdf_stream(stream);
  const int64_t planes = N * C;
  const int nblk = (int)(planes < BN_BLOCKS ? planes : BN_BLOCKS);
  const size_t lds = (size_t)(2 * C + 8) * sizeof(double);
  SDF_LAUNCH((bn_reduce_nchw_kernel<false>), dim3(nblk), dim3(256), lds, s, P);
  SDF_LAUNCH((bn_finish_kernel<false>), dim3(C), dim3(64), 0, s, P.partial, nblk, C, P.R, eps, momentum, save_mean, save_invstd,
                     running_mean, running_var);
  SDF_LAUNCH((bn_apply_nchw_kernel<false>), dim3((unsigned)((N * C * (HW / 4) + 255) / 256)), dim3(256), 0, s, P, nullptr, nullptr);
  SDF_LAUNCH_CHECK();
  return 0;
}

extern "C" int sdf_bn_train_nchw_bwd(const float* x, const float* grad_y, const float* weight, const float* save_mean,
                                     const float* save_invstd, float* grad_x, float* grad_weight, float* grad_bias, int64_t N, int C,
                                     int HW, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!x || !grad_y || !weight || !save_mean || !save_invstd || !grad_x || !grad_weight || !grad_bias || !workspace) return SDF_E_NULL;
  if (N < 1 || C < 1 || HW < 4 || HW % 4 || N * (int64_t)C * HW >= (1LL << 40) || C > 2048) return SDF_E_SHAPE;
  if (workspace_bytes < (int64_t)BN_BLOCKS * 2 * C * (int64_t)sizeof(double)) return SDF_E_SHAPE;
  if (!sdf_aligned(x, 16) || !sdf_aligned(grad_y, 16) || !sdf_aligned(grad_x, 16) || !sdf_aligned(workspace, 8)) return SDF_E_ALIGN;
  BnParams P = {};
  P.x = x; P.gy = grad_y; P.gx = grad_x; P.w = weight; P.mean = save_mean; P.invstd = save_invstd;
  P.partial = reinterpret_cast<double*>(workspace); P.R = N * HW; P.C = C; P.NB = N; P.HW = HW;
  hipStream_t s = sdf_stream(stream);
  const int64_t planes = N * C;
  const int nblk = (int)(planes < BN_BLOCKS ? planes : BN_BLOCKS);
  const size_t lds = (size_t)(2 * C + 8) * sizeof(double);
  SDF_LAUNCH((bn_reduce_nchw_kernel<true>), dim3(nblk), dim3(256), lds, s, P);
  SDF_LAUNCH((bn_finish_kernel<true>), dim3(C), dim3(64), 0, s, P.partial, nblk, C, P.R, 0.f, 0.f, grad_weight, grad_bias, nullptr,
                     nullptr);
  SDF_LAUNCH((bn_apply_nchw_kernel<true>), dim3((unsigned)((N * C * (HW / 4) + 255) / 256)), dim3(256), 0, s, P, grad_weight, grad_bias);
  SDF_LAUNCH_CHECK();
  return 0;
}
