// Backward of the spiking neurons (LIF / IF with BPTT through the membrane, PSN) for gfx950 - the training path's
// counterpart of neuron.hip.
//
// Reference semantics: spikingjelly's multi-step LIFNode run under autograd (the reference trains with the torch
// backend, Spiking_modules.py:40-47) with the ATan surrogate `g'(u) = alpha/2 / (1 + (pi/2 alpha u)^2)` (configs/*.yml
// `surrogate_fun: surrogate.ATan()`), `detach_reset` from the YAML; PSN.forward (Spiking_submodules.py:207-211) is
// `H = b + W X`, `S = surrogate(H)`.
//
// Same streaming shape as the forward: a lane owns 4 consecutive neurons, issues all 2T 16-byte loads (x and dL/dS) up
// front, REcomputes the membrane trajectory h_t in registers with the forward's exact arithmetic (nothing is saved by
// the forward: 8 B in + 4 B out per neuron-step instead of 12 + 4 with a stored h), then walks time backwards:
//     gh_t = gv_t * dv_t/dh_t + gs_t * g'(h_t - v_th)      gx_t = gh_t / tau      gv_{t-1} = gh_t - gh_t / tau
// Every operation is a separately rounded fp32 op in the order torch's autograd applies them, so the LIF gradient is
// bit-equal to the CPU reference for detach_reset = true (two-term sums commute).
// PSN: gh = gs * g'(h), gx = W^T gh per lane; dW = gh x^T and db = sum gh are reduced lane -> wave (shuffles) ->
// workgroup (LDS) -> per-workgroup partials in the caller's workspace, summed in a fixed order by a second tiny kernel
// (deterministic, no atomics).
#include "common.h"

namespace {

struct BwdParams {
  const float* x;
  const float* gs;
  float* gx;
  int64_t N;
  int T, kind, soft, detach;
  float tau, inv_tau, v_th, v_reset;
  float c_atan, half_alpha;        // (float)(pi/2 * alpha), (float)(alpha/2)
  const float* W;
  const float* b;
  float* partial;                  // [nblk][T*T + T]
  float* gh_out;                   // optional (T, N)
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// torch: alpha / 2 / (1 + (pi / 2 * alpha * u).pow(2)) * g   ==   ((1 + t*t).reciprocal() * (alpha/2)) * g
__device__ __forceinline__ float sg_atan(float u, float g, float c, float ha) {
  const float t = c * u;
  const float y = 1.f + t * t;
  return ((1.f / y) * ha) * g;
}

template <int TT>
__global__ __launch_bounds__(256) void lif_bwd_kernel(BwdParams P) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q * 4 >= P.N) return;
  const int64_t e = q * 4;
  float4 xv[TT], gv_[TT];
#pragma unroll
  for (int t = 0; t < TT; ++t) xv[t] = ld4(P.x + (int64_t)t * P.N + e);
#pragma unroll
  for (int t = 0; t < TT; ++t) gv_[t] = ld4(P.gs + (int64_t)t * P.N + e);
  const bool soft = P.soft != 0, reset0 = soft || P.v_reset == 0.f, is_if = P.kind == SDF_IF;
  const float v0 = soft ? 0.f : P.v_reset;
  float hx[TT][4];                                             // membrane before fire, per step and neuron
  {
    float v[4] = {v0, v0, v0, v0};
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      const float xs[4] = {xv[t].x, xv[t].y, xv[t].z, xv[t].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float h;
        if (is_if) {
          h = v[j] + xs[j];
        } else {
          const float d = reset0 ? (xs[j] - v[j]) : (xs[j] - (v[j] - P.v_reset));
          h = v[j] + ((P.inv_tau != 0.f) ? d * P.inv_tau : d / P.tau);
        }
        const float s = (h - P.v_th >= 0.f) ? 1.f : 0.f;
        v[j] = soft ? (h - s * P.v_th) : ((1.f - s) * h + s * P.v_reset);
        hx[t][j] = h;
      }
    }
  }
  float gv[4] = {0.f, 0.f, 0.f, 0.f};                          // dL/dv_t flowing back from step t+1 (v_T is unused)
#pragma unroll
  for (int t = TT - 1; t >= 0; --t) {
    const float gs[4] = {gv_[t].x, gv_[t].y, gv_[t].z, gv_[t].w};
    float gx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float h = hx[t][j], u = h - P.v_th;
      const float s = (u >= 0.f) ? 1.f : 0.f;
      float gspike = gs[j];                                    // dL/ds_t, plus the reset path unless it is detached
      float gh;
      if (soft) {                                              // v_t = h - s * v_th
        if (!P.detach) gspike = gspike + (-(gv[j] * P.v_th));
        gh = gv[j] + sg_atan(u, gspike, P.c_atan, P.half_alpha);
      } else {                                                 // v_t = (1 - s) * h + s * v_reset
        if (!P.detach) gspike = gspike + (gv[j] * P.v_reset + (-(gv[j] * h)));
        gh = gv[j] * (1.f - s) + sg_atan(u, gspike, P.c_atan, P.half_alpha);
      }
      if (is_if) {                                             // h = v + x
        gx[j] = gh;
        gv[j] = gh;
      } else {                                                 // h = v + (x - v) / tau  (v_reset is a constant)
        const float qd = (P.inv_tau != 0.f) ? gh * P.inv_tau : gh / P.tau;
        gx[j] = qd;
        gv[j] = gh - qd;
      }
    }
    st4(P.gx + (int64_t)t * P.N + e, make_float4(gx[0], gx[1], gx[2], gx[3]));
  }
}

// VEC neurons per lane: 4 (16-byte accesses) while the T*T + T reduction accumulators leave room for them, 2 for the
// T >= 8 reducing variants (110 accumulators + 2 x 10 x VEC operands must stay under ~170 VGPRs for 3 waves per SIMD).
template <int TT, bool REDUCE, int VEC>
__global__ __launch_bounds__(256) void psn_bwd_kernel(BwdParams P) {
  constexpr int NACC = REDUCE ? TT * TT + TT : 1;
  __shared__ float Ws[TT * TT + TT];
  __shared__ float red[4][NACC];
  for (int i = threadIdx.x; i < TT * TT + TT; i += 256) Ws[i] = i < TT * TT ? P.W[i] : P.b[i - TT * TT];
  __syncthreads();
  float acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.f;
  const int64_t groups = P.N / VEC;
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < groups; q += (int64_t)gridDim.x * 256) {
    const int64_t e = q * VEC;
    asm volatile("" ::: "memory");     // W / b are re-read from LDS every iteration: hoisted they would pin T*T + T registers
    float xv[TT][VEC], gh[TT][VEC];
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      if (VEC == 4) {
        const float4 a = ld4(P.x + (int64_t)t * P.N + e);
        xv[t][0] = a.x; xv[t][1] = a.y; xv[t][VEC - 2] = a.z; xv[t][VEC - 1] = a.w;
      } else {
        const float2 a = *reinterpret_cast<const float2*>(P.x + (int64_t)t * P.N + e);
        xv[t][0] = a.x; xv[t][1] = a.y;
      }
    }
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      if (VEC == 4) {
        const float4 a = ld4(P.gs + (int64_t)t * P.N + e);
        gh[t][0] = a.x; gh[t][1] = a.y; gh[t][VEC - 2] = a.z; gh[t][VEC - 1] = a.w;
      } else {
        const float2 a = *reinterpret_cast<const float2*>(P.gs + (int64_t)t * P.N + e);
        gh[t][0] = a.x; gh[t][1] = a.y;
      }
    }
#pragma unroll
    for (int t = 0; t < TT; ++t) {                              // h_t with the forward's fma chain, gh_t = gs_t g'(h_t)
      if (REDUCE) asm volatile("" ::: "memory");               // one row of W live at a time beside the accumulators
      const float bt = Ws[TT * TT + t];
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float h = bt;
#pragma unroll
        for (int k = 0; k < TT; ++k) h = __builtin_fmaf(Ws[t * TT + k], xv[k][j], h);
        gh[t][j] = sg_atan(h, gh[t][j], P.c_atan, P.half_alpha);
      }
      if (P.gh_out) {
        if (VEC == 4) st4(P.gh_out + (int64_t)t * P.N + e, make_float4(gh[t][0], gh[t][1], gh[t][VEC - 2], gh[t][VEC - 1]));
        else *reinterpret_cast<float2*>(P.gh_out + (int64_t)t * P.N + e) = make_float2(gh[t][0], gh[t][1]);
      }
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int k = 0; k < TT; ++k) {                              // gx_k = sum_t W[t][k] gh_t
      float g[VEC];
      if (REDUCE) asm volatile("" ::: "memory");
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        g[j] = 0.f;
#pragma unroll
        for (int t = 0; t < TT; ++t) g[j] = __builtin_fmaf(Ws[t * TT + k], gh[t][j], g[j]);
      }
      if (VEC == 4) st4(P.gx + (int64_t)k * P.N + e, make_float4(g[0], g[1], g[VEC - 2], g[VEC - 1]));
      else *reinterpret_cast<float2*>(P.gx + (int64_t)k * P.N + e) = make_float2(g[0], g[1]);
    }
    if (REDUCE) {
#pragma unroll
      for (int t = 0; t < TT; ++t) {
#pragma unroll
        for (int k = 0; k < TT; ++k) {
          float d = gh[t][0] * xv[k][0];
#pragma unroll
          for (int j = 1; j < VEC; ++j) d = __builtin_fmaf(gh[t][j], xv[k][j], d);
          acc[t * TT + k] += d;
        }
        float d = gh[t][0];
#pragma unroll
        for (int j = 1; j < VEC; ++j) d += gh[t][j];
        acc[TT * TT + t] += d;
      }
    }
  }
  if (REDUCE) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // butterfly over the 64 lanes, step-major: the T*T + T exchanges of one step are independent and stay in flight
    // together (value-major order makes every exchange wait for the one before it: 660 serial LDS round trips)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      float other[NACC];
#pragma unroll
      for (int i = 0; i < NACC; ++i) other[i] = __shfl_xor(acc[i], o);
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] += other[i];
    }
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) red[wave][i] = acc[i];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NACC; i += 256)
      P.partial[(int64_t)blockIdx.x * NACC + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
  }
}

// one workgroup per accumulator: strided partial sums, then a fixed-shape LDS tree - run-to-run bit-equal
__global__ __launch_bounds__(256) void psn_bwd_finish_kernel(const float* partial, int nblk, int nacc, int tt, float* gW, float* gb) {
  __shared__ float sm[256];
  const int i = blockIdx.x;
  float s = 0.f;
  for (int b = threadIdx.x; b < nblk; b += 256) s += partial[(int64_t)b * nacc + i];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (i < tt * tt) gW[i] = sm[0]; else gb[i - tt * tt] = sm[0];
  }
}

int psn_vec(int T, bool reduce) { return (reduce && T >= 8) ? 2 : 4; }

int psn_blocks(int T, int64_t N, bool reduce) {
  const int64_t need = (N / psn_vec(T, reduce) + 255) / 256;
  return (int)(need < 512 ? need : 512);     // two resident workgroups per CU; the end-of-block reduction is amortised over the grid-stride loop
}

}  // namespace

extern "C" int sdf_lif_bwd(const float* x, const float* grad_spike, float* grad_x, int T, int64_t N, int kind, float tau,
                           float v_th, int soft_reset, float v_reset, int detach_reset, int surrogate, float alpha,
                           void* stream) {
  if (!x || !grad_spike || !grad_x) return SDF_E_NULL;
  if (N < 4 || N % 4) return SDF_E_SHAPE;
  if (kind != SDF_LIF && kind != SDF_IF) return SDF_E_DTYPE;
  if (surrogate != SDF_SURROGATE_ATAN) return SDF_E_DTYPE;
  if (kind == SDF_LIF && !(tau > 1.f)) return SDF_E_SHAPE;        // the multiplicative (PLIF) form has no backward here
  if (!sdf_aligned(x, 16) || !sdf_aligned(grad_spike, 16) || !sdf_aligned(grad_x, 16)) return SDF_E_ALIGN;
  BwdParams P = {};
  P.x = x; P.gs = grad_spike; P.gx = grad_x; P.N = N; P.T = T; P.kind = kind; P.soft = soft_reset; P.detach = detach_reset;
  P.tau = tau; P.v_th = v_th; P.v_reset = soft_reset ? 0.f : v_reset;
  int ex;
  P.inv_tau = (kind == SDF_LIF && frexpf(tau, &ex) == 0.5f) ? 1.0f / tau : 0.f;
  P.c_atan = (float)(3.14159265358979323846 / 2 * (double)alpha);
  P.half_alpha = (float)((double)alpha / 2);
  dim3 grid((unsigned)((N / 4 + 255) / 256)), block(256);
  hipStream_t s = sdf_stream(stream);
#define SDF_T_CASE(TT) case TT: SDF_LAUNCH(lif_bwd_kernel<TT>, grid, block, 0, s, P); break;
  switch (T) {
    SDF_T_CASE(1) SDF_T_CASE(2) SDF_T_CASE(4) SDF_T_CASE(5) SDF_T_CASE(8) SDF_T_CASE(10) SDF_T_CASE(16) SDF_T_CASE(20)
    default: return SDF_E_SHAPE;
  }
#undef SDF_T_CASE
  SDF_LAUNCH_CHECK();
  return 0;
}

extern "C" int64_t sdf_psn_bwd_workspace_bytes(int T, int64_t N) {
  if (T < 1 || T > 10 || N < 4) return 0;
  return (int64_t)psn_blocks(T, N, true) * (T * T + T) * (int64_t)sizeof(float);
}

extern "C" int sdf_psn_bwd(const float* x, const float* W, const float* b, const float* grad_spike, float* grad_x,
                           float* grad_W, float* grad_b, float* grad_h, void* workspace, int64_t workspace_bytes, int T,
                           int64_t N, int surrogate, float alpha, void* stream) {
  if (!x || !W || !b || !grad_spike || !grad_x) return SDF_E_NULL;
  if (N < 4 || N % 4) return SDF_E_SHAPE;
  if (surrogate != SDF_SURROGATE_ATAN) return SDF_E_DTYPE;
  if (!sdf_aligned(x, 16) || !sdf_aligned(grad_spike, 16) || !sdf_aligned(grad_x, 16) || (grad_h && !sdf_aligned(grad_h, 16)))
    return SDF_E_ALIGN;
  const bool reduce = grad_W != nullptr;
  if (reduce && (!grad_b || !workspace)) return SDF_E_NULL;
  if (reduce && T > 10) return SDF_E_SHAPE;                      // larger T: take grad_h and form dW = grad_h x^T outside
  if (reduce && workspace_bytes < sdf_psn_bwd_workspace_bytes(T, N)) return SDF_E_SHAPE;
  BwdParams P = {};
  P.x = x; P.gs = grad_spike; P.gx = grad_x; P.N = N; P.T = T; P.W = W; P.b = b;
  P.partial = reinterpret_cast<float*>(workspace); P.gh_out = grad_h;
  P.c_atan = (float)(3.14159265358979323846 / 2 * (double)alpha);
  P.half_alpha = (float)((double)alpha / 2);
  const int nblk = psn_blocks(T, N, reduce);
  dim3 grid((unsigned)nblk), block(256);
  hipStream_t s = sdf_stream(stream);
#define SDF_T_CASE(TT, V)                                                                  \
  case TT:                                                                                 \
    if (reduce) SDF_LAUNCH((psn_bwd_kernel<TT, true, V>), grid, block, 0, s, P);   \
    else SDF_LAUNCH((psn_bwd_kernel<TT, false, 4>), grid, block, 0, s, P);         \
    break;
#define SDF_T_CASE_NR(TT) case TT: SDF_LAUNCH((psn_bwd_kernel<TT, false, 4>), grid, block, 0, s, P); break;
  switch (T) {
    SDF_T_CASE(1, 4) SDF_T_CASE(2, 4) SDF_T_CASE(4, 4) SDF_T_CASE(5, 4) SDF_T_CASE(8, 2) SDF_T_CASE(10, 2)
    SDF_T_CASE_NR(16) SDF_T_CASE_NR(20)
    default: return SDF_E_SHAPE;
  }
#undef SDF_T_CASE
#undef SDF_T_CASE_NR
  SDF_LAUNCH_CHECK();
  if (reduce) {
    const int nacc = T * T + T;
    SDF_LAUNCH(psn_bwd_finish_kernel, dim3(nacc), dim3(256), 0, s, P.partial, nblk, nacc, T, grad_W, grad_b);
    SDF_LAUNCH_CHECK();
  }
  return 0;
}
