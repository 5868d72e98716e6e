// Training-path token gate of Spiking_QK_WindowAttention3D (reference Spiking_swin_transformer3D.py:687-694) on fp32 spike
// tensors, forward and backward ("backward kernels for ... QK-attention"):
//   s[t,row,g] = sum_{d<32} q[t,row,g*32+d]          A = SN2_q(s) over the T' attention steps (LIF / IF)
//   e[t,row,c] = k[t,row,c] * A[t,row,c/32]
// backward, given dL/de:
//   gk[t,row,c] = ge[t,row,c] * A[t,row,c/32]        gA[t,row,g] = sum_{d<32} ge * k
//   gs = BPTT of the neuron over t (ATan surrogate, detach_reset; same recurrence as neuron_bwd.hip)      gq[t,row,c] = gs[t,row,c/32]
// Replaces, per block and direction, the chain reshape -> sum(-1) -> neuron -> repeat_interleave -> mul that autograd runs
// as ~8 elementwise / reduction launches.  HBM-bound: 8 lanes own one (row, head) - a 128-byte line of each tensor per step -
// and combine their partial sums with three xor shuffles; everything else is per-lane.  fp32 in / out so that the tensors
// carry gradients (the inference kernel of qk_gate.hip moves bytes).  PSN gates carry a learnable T' x T' matrix and bias: their
// gradients are reduced wave -> workgroup -> per-workgroup partials -> fixed-order finish (deterministic, no atomics).
#include "common.h"

namespace {

struct GateTrainParams {
  const float* q; const float* k; const float* ge;
  float* e; float* gq; float* gk;
  int64_t rows; int C; int G;
  int kind, soft, detach;
  float tau, inv_tau, v_th, v_reset, c_atan, half_alpha;
  const float* psn_w; const float* psn_b;   // PSN gate: (T', T') and (T')
  float* partial;                           // PSN backward: [nblk][T'*T' + T'] partial sums of dW | db
};

__device__ __forceinline__ float sum8(float v) {          // over the 8 lanes of one (row, head)
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  return v;
}

template <int TQ>
__device__ __forceinline__ void gate_neuron(const GateTrainParams& P, const float (&s)[TQ], float (&h)[TQ], float (&A)[TQ]) {
  if (P.kind == SDF_PSN) {                                       // h = b + W s (the forward kernel's fma chain), A = (h >= 0)
#pragma unroll
    for (int t = 0; t < TQ; ++t) {
      float hh = P.psn_b[t];
#pragma unroll
      for (int k = 0; k < TQ; ++k) hh = __builtin_fmaf(P.psn_w[t * TQ + k], s[k], hh);
      h[t] = hh;
      A[t] = hh >= 0.f ? 1.f : 0.f;
    }
    return;
  }
  const bool soft = P.soft != 0, reset0 = soft || P.v_reset == 0.f;
  float v = soft ? 0.f : P.v_reset;
#pragma unroll
  for (int t = 0; t < TQ; ++t) {
    if (P.kind == SDF_IF) {
      h[t] = v + s[t];
    } else {
      const float d = reset0 ? (s[t] - v) : (s[t] - (v - P.v_reset));
      h[t] = v + ((P.inv_tau != 0.f) ? d * P.inv_tau : d / P.tau);
    }
    A[t] = (h[t] - P.v_th >= 0.f) ? 1.f : 0.f;
    v = soft ? (h[t] - A[t] * P.v_th) : ((1.f - A[t]) * h[t] + A[t] * P.v_reset);
  }
}

template <int TQ, bool BWD>
__global__ __launch_bounds__(256) void qk_gate_train_kernel(GateTrainParams P) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;     // lane i owns channels 4 (i % 8) .. +3 of (row, head) i / 8
  const int64_t pairs = P.rows * P.G;
  const bool live = i < pairs * 8;
  const int64_t off = live ? i * 4 : 0;                          // ((row*G + g)*32 + 4*(i%8)) == row*C + g*32 + 4*(i%8)
  const int64_t step = P.rows * (int64_t)P.C;
  float4 qv[TQ], kv[TQ], gv[TQ];
  float s[TQ], h[TQ], A[TQ];
#pragma unroll
  for (int t = 0; t < TQ; ++t) {
    qv[t] = live ? *reinterpret_cast<const float4*>(P.q + t * step + off) : make_float4(0.f, 0.f, 0.f, 0.f);
    kv[t] = live ? *reinterpret_cast<const float4*>(P.k + t * step + off) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (BWD) gv[t] = live ? *reinterpret_cast<const float4*>(P.ge + t * step + off) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int t = 0; t < TQ; ++t) s[t] = sum8((qv[t].x + qv[t].y) + (qv[t].z + qv[t].w));      // spikes: an exact integer 0..32
  gate_neuron<TQ>(P, s, h, A);
  if (!BWD) {
    if (live) {
#pragma unroll
      for (int t = 0; t < TQ; ++t)
        *reinterpret_cast<float4*>(P.e + t * step + off) = make_float4(kv[t].x * A[t], kv[t].y * A[t], kv[t].z * A[t], kv[t].w * A[t]);
    }
    return;
  }
  float gA[TQ], gs[TQ];
#pragma unroll
  for (int t = 0; t < TQ; ++t)
    gA[t] = sum8((gv[t].x * kv[t].x + gv[t].y * kv[t].y) + (gv[t].z * kv[t].z + gv[t].w * kv[t].w));
  if (P.kind == SDF_PSN) {
    // gh = gA * g'(h);  gs_k = sum_t W[t][k] gh_t;  dW[t][k] += gh_t s_k, db[t] += gh_t over all (row, head) pairs
    float gh[TQ];
#pragma unroll
    for (int t = 0; t < TQ; ++t) {
      const float tt = P.c_atan * h[t], y = 1.f + tt * tt;
      gh[t] = ((1.f / y) * P.half_alpha) * gA[t];
    }
#pragma unroll
    for (int k = 0; k < TQ; ++k) {
      float g = 0.f;
#pragma unroll
      for (int t = 0; t < TQ; ++t) g = __builtin_fmaf(P.psn_w[t * TQ + k], gh[t], g);
      gs[k] = g;
    }
    if (live) {
#pragma unroll
      for (int t = 0; t < TQ; ++t) {
        *reinterpret_cast<float4*>(P.gq + t * step + off) = make_float4(gs[t], gs[t], gs[t], gs[t]);
        *reinterpret_cast<float4*>(P.gk + t * step + off) = make_float4(gv[t].x * A[t], gv[t].y * A[t], gv[t].z * A[t], gv[t].w * A[t]);
      }
    }
    // parameter gradients: one lane per (row, head) contributes (all 8 hold the same sums); wave butterfly -> LDS -> partial
    constexpr int NACC = TQ * TQ + TQ;
    __shared__ float red[4][NACC];
    const bool lead = live && (threadIdx.x & 7) == 0;
    float acc[NACC];
#pragma unroll
    for (int t = 0; t < TQ; ++t) {
#pragma unroll
      for (int k = 0; k < TQ; ++k) acc[t * TQ + k] = lead ? gh[t] * s[k] : 0.f;
      acc[TQ * TQ + t] = lead ? gh[t] : 0.f;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
#pragma unroll
      for (int a = 0; a < NACC; ++a) acc[a] += __shfl_xor(acc[a], o);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
      for (int a = 0; a < NACC; ++a) red[wave][a] = acc[a];
    }
    __syncthreads();
    if ((int)threadIdx.x < NACC)
      P.partial[(int64_t)blockIdx.x * NACC + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    return;
  }
  // BPTT through the gate neuron (the recurrence of neuron_bwd.hip on the TQ head sums)
  const bool soft = P.soft != 0;
  float gvm = 0.f;
#pragma unroll
  for (int t = TQ - 1; t >= 0; --t) {
    const float u = h[t] - P.v_th;
    float gsp = gA[t], gh;
    const float tt = P.c_atan * u, y = 1.f + tt * tt;
    if (soft) {
      if (!P.detach) gsp = gsp + (-(gvm * P.v_th));
      gh = gvm + ((1.f / y) * P.half_alpha) * gsp;
    } else {
      if (!P.detach) gsp = gsp + (gvm * P.v_reset + (-(gvm * h[t])));
      gh = gvm * (1.f - A[t]) + ((1.f / y) * P.half_alpha) * gsp;
    }
    if (P.kind == SDF_IF) {
      gs[t] = gh;
      gvm = gh;
    } else {
      const float qd = (P.inv_tau != 0.f) ? gh * P.inv_tau : gh / P.tau;
      gs[t] = qd;
      gvm = gh - qd;
    }
  }
  if (live) {
#pragma unroll
    for (int t = 0; t < TQ; ++t) {
      *reinterpret_cast<float4*>(P.gq + t * step + off) = make_float4(gs[t], gs[t], gs[t], gs[t]);
      *reinterpret_cast<float4*>(P.gk + t * step + off) = make_float4(gv[t].x * A[t], gv[t].y * A[t], gv[t].z * A[t], gv[t].w * A[t]);
    }
  }
}

int fill(GateTrainParams& P, int64_t rows, int C, int kind, float tau, float v_th, int soft_reset, float v_reset) {
  if (rows < 1 || C < 32 || C % 32 || rows * (int64_t)C >= (1LL << 40)) return SDF_E_SHAPE;
  if (kind != SDF_LIF && kind != SDF_IF && kind != SDF_PSN) return SDF_E_DTYPE;
  if (kind == SDF_LIF && !(tau > 1.f)) return SDF_E_SHAPE;
  P.rows = rows; P.C = C; P.G = C / 32; P.kind = kind; P.soft = soft_reset; P.tau = tau; P.v_th = v_th;
  P.v_reset = soft_reset ? 0.f : v_reset;
  int ex;
  P.inv_tau = (kind == SDF_LIF && frexpf(tau, &ex) == 0.5f) ? 1.0f / tau : 0.f;
  return 0;
}

// fixed-order sum of the per-workgroup partials (double accumulation, one thread per parameter-gradient entry group)
__global__ __launch_bounds__(256) void gate_finish_kernel(const float* partial, int64_t nblk, int nacc, int tt, float* gW, float* gb) {
  __shared__ double sm[256];
  const int a = blockIdx.x;
  double t = 0;
  for (int64_t b = threadIdx.x; b < nblk; b += 256) t += partial[b * nacc + a];
  sm[threadIdx.x] = t;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (a < tt * tt) gW[a] = (float)sm[0]; else gb[a - tt * tt] = (float)sm[0];
  }
}

template <bool BWD>
int launch(const GateTrainParams& P, int Tq, hipStream_t s) {
  const int64_t lanes = P.rows * P.G * 8;
  dim3 grid((unsigned)((lanes + 255) / 256)), block(256);
  switch (Tq) {
    case 1: SDF_LAUNCH((qk_gate_train_kernel<1, BWD>), grid, block, 0, s, P); break;
    case 2: SDF_LAUNCH((qk_gate_train_kernel<2, BWD>), grid, block, 0, s, P); break;
    case 4: SDF_LAUNCH((qk_gate_train_kernel<4, BWD>), grid, block, 0, s, P); break;
    default: return SDF_E_SHAPE;
  }
  SDF_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" int sdf_qk_gate_f32_fwd(const float* q, const float* k, float* e, int Tq, int64_t rows, int C, int kind, float tau,
                                   float v_th, int soft_reset, float v_reset, const float* psn_w, const float* psn_b, void* stream) {
  if (!q || !k || !e) return SDF_E_NULL;
  if (kind == SDF_PSN && (!psn_w || !psn_b)) return SDF_E_NULL;
  if (!sdf_aligned(q, 16) || !sdf_aligned(k, 16) || !sdf_aligned(e, 16)) return SDF_E_ALIGN;
  GateTrainParams P = {};
  const int rc = fill(P, rows, C, kind, tau, v_th, soft_reset, v_reset);
  if (rc) return rc;
  P.q = q; P.k = k; P.e = e; P.psn_w = psn_w; P.psn_b = psn_b;
  return launch<false>(P, Tq, sdf_stream(stream));
}

extern "C" int64_t sdf_qk_gate_bwd_workspace_bytes(int Tq, int64_t rows, int C) {
  if (Tq < 1 || rows < 1 || C < 32) return 0;
  const int64_t nblk = (rows * (C / 32) * 8 + 255) / 256;
  return nblk * (Tq * Tq + Tq) * (int64_t)sizeof(float);
}

extern "C" int sdf_qk_gate_bwd(const float* q, const float* k, const float* grad_e, float* grad_q, float* grad_k, int Tq,
                               int64_t rows, int C, int kind, float tau, float v_th, int soft_reset, float v_reset,
                               int detach_reset, int surrogate, float alpha, const float* psn_w, const float* psn_b,
                               float* grad_psn_w, float* grad_psn_b, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!q || !k || !grad_e || !grad_q || !grad_k) return SDF_E_NULL;
  if (kind == SDF_PSN) {
    if (!psn_w || !psn_b || !grad_psn_w || !grad_psn_b || !workspace) return SDF_E_NULL;
    if (workspace_bytes < sdf_qk_gate_bwd_workspace_bytes(Tq, rows, C) || !sdf_aligned(workspace, 4)) return SDF_E_SHAPE;
  }
  if (surrogate != SDF_SURROGATE_ATAN) return SDF_E_DTYPE;
  if (!sdf_aligned(q, 16) || !sdf_aligned(k, 16) || !sdf_aligned(grad_e, 16) || !sdf_aligned(grad_q, 16) || !sdf_aligned(grad_k, 16))
    return SDF_E_ALIGN;
  GateTrainParams P = {};
  const int rc = fill(P, rows, C, kind, tau, v_th, soft_reset, v_reset);
  if (rc) return rc;
  P.q = q; P.k = k; P.ge = grad_e; P.gq = grad_q; P.gk = grad_k; P.detach = detach_reset;
  P.psn_w = psn_w; P.psn_b = psn_b; P.partial = reinterpret_cast<float*>(workspace);
  P.c_atan = (float)(3.14159265358979323846 / 2 * (double)alpha);
  P.half_alpha = (float)((double)alpha / 2);
  const int rc2 = launch<true>(P, Tq, sdf_stream(stream));
  if (rc2 || kind != SDF_PSN) return rc2;
  const int64_t nblk = (rows * (C / 32) * 8 + 255) / 256;
  const int nacc = Tq * Tq + Tq;
  SDF_LAUNCH(gate_finish_kernel, dim3(nacc), dim3(256), 0, sdf_stream(stream), P.partial, nblk, nacc, Tq, grad_psn_w, grad_psn_b);
  SDF_LAUNCH_CHECK();
  return 0;
}
