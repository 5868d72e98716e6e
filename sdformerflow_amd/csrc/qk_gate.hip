// Token gate of Spiking_QK_WindowAttention3D (reference Spiking_swin_transformer3D.py:687-694).
//   a[t,row,g] = sum_{d<32} q[t,row,g*32+d]   (an integer 0..32, exact in fp32)
//   A = neuron over the Tq attention-time steps (LIF or PSN, same arithmetic as neuron.hip)
//   e[t,row,c] = k[t,row,c] * A[t,row,c/32]
// One lane per (row, 32-channel group): 32 B of q and k per step, all byte traffic.
// Compiled with -ffp-contract=off (LIF ops are separately rounded; PSN uses explicit fmaf).
#include "common.h"

namespace {
constexpr int TQ_MAX = 4;

struct GateParams {
  const uint8_t* q; const uint8_t* k; uint8_t* e;
  int Tq; int64_t rows; int C; int G;
  int64_t ldq, ldk;              // row strides of q and k in bytes (>= C; the fused q|k GEMM writes both into one buffer)
  int kind; float tau, inv_tau, v_th, v_reset; int soft;
  const float* psn_w; const float* psn_b;
};

__global__ __launch_bounds__(256) void qk_gate_kernel(GateParams P) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= P.rows * P.G) return;
  const int64_t step = P.rows * (int64_t)P.C;
  const int64_t off = i * 32;                       // (row*G + g)*32 == row*C + g*32
  const int64_t row = i / P.G;
  const int g32 = (int)(i - row * P.G) * 32;
  const int64_t qoff = row * P.ldq + g32, koff = row * P.ldk + g32;
  float a[TQ_MAX];
  uint4 k0[TQ_MAX], k1[TQ_MAX];
#pragma unroll
  for (int t = 0; t < TQ_MAX; ++t) {
    if (t < P.Tq) {
      const uint4* qp = reinterpret_cast<const uint4*>(P.q + t * P.rows * P.ldq + qoff);
      uint4 x0 = qp[0], x1 = qp[1];
      int s = __popc(x0.x) + __popc(x0.y) + __popc(x0.z) + __popc(x0.w) + __popc(x1.x) + __popc(x1.y) + __popc(x1.z) +
              __popc(x1.w);
      a[t] = (float)s;
      const uint4* kp = reinterpret_cast<const uint4*>(P.k + t * P.rows * P.ldk + koff);
      k0[t] = kp[0];
      k1[t] = kp[1];
    }
  }
  float gate[TQ_MAX];
  if (P.kind == SDF_PSN) {
#pragma unroll
    for (int t = 0; t < TQ_MAX; ++t) {
      if (t < P.Tq) {
        float h = P.psn_b[t];
#pragma unroll
        for (int kk = 0; kk < TQ_MAX; ++kk)
          if (kk < P.Tq) h = __builtin_fmaf(P.psn_w[t * P.Tq + kk], a[kk], h);
        gate[t] = h >= 0.f ? 1.f : 0.f;
      }
    }
  } else {
    const bool soft = P.soft != 0;
    const bool reset0 = soft || P.v_reset == 0.f;
    float v = soft ? 0.f : P.v_reset;
#pragma unroll
    for (int t = 0; t < TQ_MAX; ++t) {
      if (t < P.Tq) {
        float h;
        if (P.kind == SDF_IF) {
          h = v + a[t];
        } else {
          float dlt = reset0 ? (a[t] - v) : (a[t] - (v - P.v_reset));
          h = v + ((P.inv_tau != 0.f) ? dlt * P.inv_tau : dlt / P.tau);
        }
        float s = (h - P.v_th >= 0.f) ? 1.f : 0.f;
        v = soft ? (h - s * P.v_th) : ((1.f - s) * h + s * P.v_reset);
        gate[t] = s;
      }
    }
  }
#pragma unroll
  for (int t = 0; t < TQ_MAX; ++t) {
    if (t < P.Tq) {
      const uint32_t msk = gate[t] != 0.f ? 0xFFFFFFFFu : 0u;
      uint4* ep = reinterpret_cast<uint4*>(P.e + t * step + off);
      ep[0] = make_uint4(k0[t].x & msk, k0[t].y & msk, k0[t].z & msk, k0[t].w & msk);
      ep[1] = make_uint4(k1[t].x & msk, k1[t].y & msk, k1[t].z & msk, k1[t].w & msk);
    }
  }
}
}  // namespace

extern "C" int sdf_qk_gate_strided_fwd(const uint8_t* q, const uint8_t* k, uint8_t* e, int Tq, int64_t rows, int C, int64_t ldq,
                                       int64_t ldk, int kind, float tau, float v_th, float v_reset, int soft_reset,
                                       const float* psn_w, const float* psn_b, void* stream) {
  if (!q || !k || !e) return SDF_E_NULL;
  if (Tq < 1 || Tq > TQ_MAX || rows < 1 || C < 32 || C % 32) return SDF_E_SHAPE;
  if (ldq < C || ldk < C || ldq % 16 || ldk % 16) return SDF_E_SHAPE;
  if (kind != SDF_LIF && kind != SDF_PSN && kind != SDF_IF) return SDF_E_DTYPE;
  if (kind == SDF_PSN && (!psn_w || !psn_b)) return SDF_E_NULL;
  if (!sdf_tau_ok(kind, tau)) return SDF_E_SHAPE;
  if (!sdf_aligned(q, 16) || !sdf_aligned(k, 16) || !sdf_aligned(e, 16)) return SDF_E_ALIGN;
  GateParams P;
  P.q = q; P.k = k; P.e = e; P.Tq = Tq; P.rows = rows; P.C = C; P.G = C / 32;
  P.ldq = ldq; P.ldk = ldk;
  P.kind = kind; P.tau = tau; P.v_th = v_th; P.v_reset = v_reset; P.soft = soft_reset;
  P.inv_tau = sdf_inv_tau(kind, tau);
  P.psn_w = psn_w; P.psn_b = psn_b;
  int64_t n = rows * P.G;
  SDF_LAUNCH(qk_gate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, sdf_stream(stream), P);
  SDF_LAUNCH_CHECK();
  return 0;
}

extern "C" int sdf_qk_gate_fwd(const uint8_t* q, const uint8_t* k, uint8_t* e, int Tq, int64_t rows, int C, int kind,
                               float tau, float v_th, float v_reset, int soft_reset, const float* psn_w,
                               const float* psn_b, void* stream) {
  return sdf_qk_gate_strided_fwd(q, k, e, Tq, rows, C, C, C, kind, tau, v_th, v_reset, soft_reset, psn_w, psn_b, stream);
}
