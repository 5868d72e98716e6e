// Spiking-neuron state update kernels (LIF / IF / PSN) for gfx950.
//
// HBM-bound streaming kernels: every lane owns 4 consecutive neurons, issues all T of its 16-byte
// loads before the first use (T <= 20 loads in flight per lane hide HBM latency without LDS), runs
// the charge / fire / reset recurrence in registers and writes each spike once (fp32 or 1 byte).
// The eval-BatchNorm affine, the positional-encoding add and the pad / roll / window-partition
// gather that precede the neuron in the reference are folded into the load.
//
// Numerics contract (checked bit-for-bit against oracle/): every LIF operation is a separately
// rounded IEEE fp32 op - this file is compiled with -ffp-contract=off - and the only fused
// operations are the explicit fmaf() of the BN prologue and of the PSN chain.
#include "common.h"

namespace {

struct NeuronParams {
  SdfNeuronDesc d;
  int64_t quads;     // nb*ni/4
  int64_t rows;      // gather mode: nb*ni/rowlen
  float inv_tau;     // exact reciprocal when tau is a power of two, else 0
};

__device__ __forceinline__ float4 load4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <int KIND>
struct Step;  // charge/fire/reset of one time step on a 4-vector

__device__ __forceinline__ float lif_charge(float v, float x, float tau, float inv_tau, float v_reset, bool reset0) {
  float d = reset0 ? (x - v) : (x - (v - v_reset));
  float q = (inv_tau != 0.f) ? d * inv_tau : d / tau;   // power-of-two tau: multiplication is exact
  return v + q;
}

__device__ __forceinline__ float fire_reset(float& v, float h, float v_th, float v_reset, bool soft) {
  float s = (h - v_th >= 0.f) ? 1.f : 0.f;
  v = soft ? (h - s * v_th) : ((1.f - s) * h + s * v_reset);
  return s;
}

__device__ __forceinline__ void store_spikes(const NeuronParams& P, int64_t ooff, float4 s) {
  if (P.d.out_dtype == SDF_F32) {
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(P.d.out) + ooff) = s;
  } else {
    uint32_t w = (uint32_t)(s.x != 0.f) | ((uint32_t)(s.y != 0.f) << 8) | ((uint32_t)(s.z != 0.f) << 16) |
                 ((uint32_t)(s.w != 0.f) << 24);
    *reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(P.d.out) + ooff) = w;
  }
}

// Address of the quad in x for step t, or nullptr for a padding row.
__device__ __forceinline__ const float* x_addr(const NeuronParams& P, int t, int64_t e, int64_t b, int64_t r, int64_t xrep = 0) {
  if (P.d.rowmap == nullptr) return P.d.x + xrep + b * P.d.x_sb + (int64_t)t * P.d.x_st + r;
  int64_t row = e / P.d.rowlen;
  int64_t col = e - row * P.d.rowlen;
  int32_t src = P.d.rowmap[(int64_t)t * P.rows + row];
  return src < 0 ? nullptr : P.d.x + (int64_t)src * P.d.rowlen + col;
}

__device__ __forceinline__ float4 prologue(const NeuronParams& P, float4 x, int t, int64_t r, float4 al, float4 be) {
  if (P.d.alpha) {
    x.x = __builtin_fmaf(x.x, al.x, be.x);
    x.y = __builtin_fmaf(x.y, al.y, be.y);
    x.z = __builtin_fmaf(x.z, al.z, be.z);
    x.w = __builtin_fmaf(x.w, al.w, be.w);
  }
  if (P.d.add) {
    float4 a = load4(P.d.add + (int64_t)t * P.d.add_st + (r % P.d.add_period));
    x.x += a.x; x.y += a.y; x.z += a.z; x.w += a.w;
  }
  return x;
}

// T is a compile-time constant (register-resident x) when TT > 0, else runtime (streaming, LIF/IF only).
template <int TT>
__device__ __forceinline__ void neuron_body(const NeuronParams& P, int64_t q) {
  if (q >= P.quads) return;
  const int T = TT > 0 ? TT : P.d.T;
  int64_t e = q * 4, xrep = 0, orep = 0;
  if (P.d.nrep > 1) {                                            // outermost dimension: nrep equally laid out problems (dense mode)
    const int64_t per = P.d.nb * P.d.ni, rp = e / per;
    e -= rp * per;
    xrep = rp * P.d.x_srep;
    orep = rp * P.d.o_srep;
  }
  const int64_t b = e / P.d.ni;
  const int64_t r = e - b * P.d.ni;
  const int64_t obase = orep + b * P.d.o_sb + r;

  float4 al = make_float4(1.f, 1.f, 1.f, 1.f), be = make_float4(0.f, 0.f, 0.f, 0.f);
  if (P.d.alpha) {
    if (P.d.inner == 1) {
      int c = (int)(r % P.d.C);
      al = load4(P.d.alpha + c);
      be = load4(P.d.beta + c);
    } else {
      int c = (int)((r / P.d.inner) % P.d.C);
      float a = P.d.alpha[c], bb = P.d.beta[c];
      al = make_float4(a, a, a, a);
      be = make_float4(bb, bb, bb, bb);
    }
  }

  if constexpr (TT > 0) {
    float4 xv[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      const float* p = x_addr(P, t, e, b, r, xrep);
      xv[t] = p ? load4(p) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int t = 0; t < TT; ++t) xv[t] = prologue(P, xv[t], t, r, al, be);

    if (P.d.kind == SDF_PSN) {
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        const float bt = P.d.psn_b[t];
        float4 h = make_float4(bt, bt, bt, bt);
#pragma unroll
        for (int k = 0; k < TT; ++k) {
          const float w = P.d.psn_w[t * TT + k];
          h.x = __builtin_fmaf(w, xv[k].x, h.x);
          h.y = __builtin_fmaf(w, xv[k].y, h.y);
          h.z = __builtin_fmaf(w, xv[k].z, h.z);
          h.w = __builtin_fmaf(w, xv[k].w, h.w);
        }
        float4 s = make_float4(h.x >= 0.f ? 1.f : 0.f, h.y >= 0.f ? 1.f : 0.f, h.z >= 0.f ? 1.f : 0.f,
                               h.w >= 0.f ? 1.f : 0.f);
        store_spikes(P, obase + (int64_t)t * P.d.o_st, s);
      }
      return;
    }
    const bool soft = P.d.soft_reset != 0;
    const bool reset0 = soft || P.d.v_reset == 0.f;
    const float v0 = soft ? 0.f : P.d.v_reset;
    float4 v = make_float4(v0, v0, v0, v0);
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      float4 h, s;
      if (P.d.kind == SDF_IF) {
        h.x = v.x + xv[t].x; h.y = v.y + xv[t].y; h.z = v.z + xv[t].z; h.w = v.w + xv[t].w;
      } else {
        h.x = lif_charge(v.x, xv[t].x, P.d.tau, P.inv_tau, P.d.v_reset, reset0);
        h.y = lif_charge(v.y, xv[t].y, P.d.tau, P.inv_tau, P.d.v_reset, reset0);
        h.z = lif_charge(v.z, xv[t].z, P.d.tau, P.inv_tau, P.d.v_reset, reset0);
        h.w = lif_charge(v.w, xv[t].w, P.d.tau, P.inv_tau, P.d.v_reset, reset0);
      }
      s.x = fire_reset(v.x, h.x, P.d.v_th, P.d.v_reset, soft);
      s.y = fire_reset(v.y, h.y, P.d.v_th, P.d.v_reset, soft);
      s.z = fire_reset(v.z, h.z, P.d.v_th, P.d.v_reset, soft);
      s.w = fire_reset(v.w, h.w, P.d.v_th, P.d.v_reset, soft);
      store_spikes(P, obase + (int64_t)t * P.d.o_st, s);
    }
    if (P.d.v_last) *reinterpret_cast<float4*>(P.d.v_last + q * 4) = v;
  } else {
    // runtime T: sequential neurons only (PSN with an unlisted T is rejected on the host)
    const bool soft = P.d.soft_reset != 0;
    const bool reset0 = soft || P.d.v_reset == 0.f;
    const float v0 = soft ? 0.f : P.d.v_reset;
    float4 v = make_float4(v0, v0, v0, v0);
    const float* p = x_addr(P, 0, e, b, r, xrep);
    float4 nxt = p ? load4(p) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = 0; t < T; ++t) {
      float4 x = nxt;
      if (t + 1 < T) {
        const float* pn = x_addr(P, t + 1, e, b, r, xrep);
        nxt = pn ? load4(pn) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      x = prologue(P, x, t, r, al, be);
      float4 h, s;
      if (P.d.kind == SDF_IF) {
        h.x = v.x + x.x; h.y = v.y + x.y; h.z = v.z + x.z; h.w = v.w + x.w;
      } else {
        h.x = lif_charge(v.x, x.x, P.d.tau, P.inv_tau, P.d.v_reset, reset0);
        h.y = lif_charge(v.y, x.y, P.d.tau, P.inv_tau, P.d.v_reset, reset0);
        h.z = lif_charge(v.z, x.z, P.d.tau, P.inv_tau, P.d.v_reset, reset0);
        h.w = lif_charge(v.w, x.w, P.d.tau, P.inv_tau, P.d.v_reset, reset0);
      }
      s.x = fire_reset(v.x, h.x, P.d.v_th, P.d.v_reset, soft);
      s.y = fire_reset(v.y, h.y, P.d.v_th, P.d.v_reset, soft);
      s.z = fire_reset(v.z, h.z, P.d.v_th, P.d.v_reset, soft);
      s.w = fire_reset(v.w, h.w, P.d.v_th, P.d.v_reset, soft);
      store_spikes(P, obase + (int64_t)t * P.d.o_st, s);
    }
    if (P.d.v_last) *reinterpret_cast<float4*>(P.d.v_last + q * 4) = v;
  }
}

template <int TT>
__global__ __launch_bounds__(256) void neuron_kernel(NeuronParams P) {
  neuron_body<TT>(P, (int64_t)blockIdx.x * 256 + threadIdx.x);
}

// Several independent neuron calls of the same T as ONE launch (the decoders' skip inputs: four small tensors, each a launch of
// its own before): workgroup ranges are dealt to the descriptors by `first[]`
constexpr int MULTI_MAX = 6;
struct NeuronMulti {
  NeuronParams p[MULTI_MAX];
  int first[MULTI_MAX + 1];       // first workgroup of descriptor i; first[n] = grid size
  int n;
};

template <int TT>
__global__ __launch_bounds__(256) void neuron_multi_kernel(NeuronMulti M) {
  int i = 0;
#pragma unroll
  for (int k = 1; k < MULTI_MAX; ++k)
    if (k < M.n && (int)blockIdx.x >= M.first[k]) i = k;
  i = __builtin_amdgcn_readfirstlane(i);
  // (select by a chain of wave-uniform compares: an indexed copy of a by-value struct would live in scratch)
#define SDF_MULTI_CASE(k) if (i == k) { neuron_body<TT>(M.p[k], (int64_t)((int)blockIdx.x - M.first[k]) * 256 + threadIdx.x); return; }
  SDF_MULTI_CASE(0) SDF_MULTI_CASE(1) SDF_MULTI_CASE(2) SDF_MULTI_CASE(3) SDF_MULTI_CASE(4) SDF_MULTI_CASE(5)
#undef SDF_MULTI_CASE
}

// One neuron per lane, any N: the contiguous (T, N) entry points use it when N is not a multiple of 4 (rows of such a
// buffer are not 16-byte aligned, so the streaming kernel's float4 accesses do not apply).  Same op sequence.
__global__ __launch_bounds__(256) void neuron_scalar_kernel(const float* __restrict__ x, void* __restrict__ out, float* __restrict__ v_last,
                                                            int T, int64_t N, int kind, float tau, float inv_tau, float v_th,
                                                            float v_reset, int soft_i, int out_dtype, const float* __restrict__ psn_w,
                                                            const float* __restrict__ psn_b) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const bool soft = soft_i != 0;
  auto put = [&](int t, float sp) {
    if (out_dtype == SDF_F32) reinterpret_cast<float*>(out)[(int64_t)t * N + n] = sp;
    else reinterpret_cast<uint8_t*>(out)[(int64_t)t * N + n] = sp != 0.f ? 1 : 0;
  };
  if (kind == SDF_PSN) {
    for (int t = 0; t < T; ++t) {
      float h = psn_b[t];
      for (int k = 0; k < T; ++k) h = __builtin_fmaf(psn_w[t * T + k], x[(int64_t)k * N + n], h);
      put(t, h >= 0.f ? 1.f : 0.f);
    }
    return;
  }
  const bool reset0 = soft || v_reset == 0.f;
  float v = soft ? 0.f : v_reset;
  for (int t = 0; t < T; ++t) {
    const float xt = x[(int64_t)t * N + n];
    const float h = kind == SDF_IF ? v + xt : lif_charge(v, xt, tau, inv_tau, v_reset, reset0);
    put(t, fire_reset(v, h, v_th, v_reset, soft));
  }
  if (v_last) v_last[n] = v;
}

int launch_scalar(const float* x, void* spike, float* v_last, int T, int64_t N, int kind, float tau, float v_th, int soft_reset,
                  float v_reset, int spike_dtype, const float* W, const float* b, void* stream) {
  if (!x || !spike) return SDF_E_NULL;
  if (T < 1 || N < 1) return SDF_E_SHAPE;
  if (spike_dtype != SDF_F32 && spike_dtype != SDF_U8) return SDF_E_DTYPE;
  if (kind == SDF_PSN && (!W || !b)) return SDF_E_NULL;
  if (!sdf_tau_ok(kind, tau)) return SDF_E_SHAPE;
  const float inv_tau = sdf_inv_tau(kind, tau);
  SDF_LAUNCH(neuron_scalar_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, sdf_stream(stream), x, spike, v_last, T, N,
                     kind, tau, inv_tau, v_th, v_reset, soft_reset, spike_dtype, W, b);
  SDF_LAUNCH_CHECK();
  return 0;
}

int validate(const SdfNeuronDesc& d) {
  if (!d.x || !d.out) return SDF_E_NULL;
  if (d.T < 1 || d.nb < 1 || d.ni < 4) return SDF_E_SHAPE;
  if (d.out_dtype != SDF_F32 && d.out_dtype != SDF_U8) return SDF_E_DTYPE;
  if (d.kind != SDF_LIF && d.kind != SDF_PSN && d.kind != SDF_IF) return SDF_E_DTYPE;
  if (d.ni % 4 || d.o_sb % 4 || d.o_st % 4) return SDF_E_SHAPE;
  if (!sdf_aligned(d.x, 16) || !sdf_aligned(d.out, d.out_dtype == SDF_F32 ? 16 : 4)) return SDF_E_ALIGN;
  if (d.nrep < 0 || (d.nrep > 1 && (d.rowmap || d.x_srep % 4 || d.o_srep % 4))) return SDF_E_SHAPE;      // (the outer dimension is a dense-mode feature)
  if (d.rowmap) {
    if (d.rowlen < 4 || d.rowlen % 4 || (d.nb * d.ni) % d.rowlen) return SDF_E_SHAPE;
  } else if (d.x_sb % 4 || d.x_st % 4) {
    return SDF_E_SHAPE;
  }
  if (d.alpha) {
    if (!d.beta) return SDF_E_NULL;
    if (d.C < 1 || d.inner < 1) return SDF_E_SHAPE;
    if (d.inner == 1 ? (d.C % 4 != 0) : (d.inner % 4 != 0)) return SDF_E_SHAPE;
    if (d.inner == 1 && (!sdf_aligned(d.alpha, 16) || !sdf_aligned(d.beta, 16))) return SDF_E_ALIGN;
  }
  if (d.add && (d.add_period < 4 || d.add_period % 4 || d.add_st % 4 || !sdf_aligned(d.add, 16))) return SDF_E_SHAPE;
  if (d.kind == SDF_PSN) {
    if (!d.psn_w || !d.psn_b) return SDF_E_NULL;
    if (d.v_last) return SDF_E_SHAPE;
  } else if (!sdf_tau_ok(d.kind, d.tau)) {
    return SDF_E_SHAPE;
  }
  if (d.v_last && !sdf_aligned(d.v_last, 16)) return SDF_E_ALIGN;
  return 0;
}

}  // namespace

extern "C" int sdf_neuron_fwd(const SdfNeuronDesc* dp, void* stream) {
  if (!dp) return SDF_E_NULL;
  int rc = validate(*dp);
  if (rc) return rc;
  NeuronParams P;
  P.d = *dp;
  P.quads = (dp->nrep > 1 ? dp->nrep : 1) * dp->nb * dp->ni / 4;
  P.rows = dp->rowmap ? dp->nb * dp->ni / dp->rowlen : 0;
  P.inv_tau = sdf_inv_tau(dp->kind, dp->tau);
  dim3 grid((unsigned)((P.quads + 255) / 256)), block(256);
  hipStream_t s = sdf_stream(stream);
#define SDF_T_CASE(TT) case TT: SDF_LAUNCH(neuron_kernel<TT>, grid, block, 0, s, P); break;
  switch (dp->T) {
    SDF_T_CASE(1) SDF_T_CASE(2) SDF_T_CASE(4) SDF_T_CASE(5) SDF_T_CASE(8) SDF_T_CASE(10) SDF_T_CASE(16) SDF_T_CASE(20)
    default:
      if (dp->kind == SDF_PSN) return SDF_E_SHAPE;
      SDF_LAUNCH(neuron_kernel<0>, grid, block, 0, s, P);
  }
#undef SDF_T_CASE
  SDF_LAUNCH_CHECK();
  return 0;
}

extern "C" int sdf_neuron_multi_fwd(const SdfNeuronDesc* descs, int n, void* stream) {
  if (!descs) return SDF_E_NULL;
  if (n < 1) return SDF_E_SHAPE;
  if (n == 1) return sdf_neuron_fwd(descs, stream);
  const int T = descs[0].T;
  bool one = n <= MULTI_MAX && (T == 2 || T == 4 || T == 5 || T == 10 || T == 20);
  for (int i = 0; i < n; ++i) {
    const int rc = validate(descs[i]);
    if (rc) return rc;
    one = one && descs[i].T == T;
  }
  if (!one) {                                                    // mixed T or more than MULTI_MAX descriptors: one launch each
    for (int i = 0; i < n; ++i) {
      const int rc = sdf_neuron_fwd(descs + i, stream);
      if (rc) return rc;
    }
    return 0;
  }
  NeuronMulti M;
  M.n = n;
  int64_t wgs = 0;
  for (int i = 0; i < n; ++i) {
    NeuronParams& P = M.p[i];
    P.d = descs[i];
    P.quads = (descs[i].nrep > 1 ? descs[i].nrep : 1) * descs[i].nb * descs[i].ni / 4;
    P.rows = descs[i].rowmap ? descs[i].nb * descs[i].ni / descs[i].rowlen : 0;
    P.inv_tau = sdf_inv_tau(descs[i].kind, descs[i].tau);
    M.first[i] = (int)wgs;
    wgs += (P.quads + 255) / 256;
    if (wgs >= (1LL << 31)) return SDF_E_SHAPE;
  }
  for (int i = n; i <= MULTI_MAX; ++i) M.first[i] = (int)wgs;
  dim3 grid((unsigned)wgs), block(256);
  hipStream_t s = sdf_stream(stream);
  switch (T) {
    case 2: SDF_LAUNCH(neuron_multi_kernel<2>, grid, block, 0, s, M); break;
    case 4: SDF_LAUNCH(neuron_multi_kernel<4>, grid, block, 0, s, M); break;
    case 5: SDF_LAUNCH(neuron_multi_kernel<5>, grid, block, 0, s, M); break;
    case 10: SDF_LAUNCH(neuron_multi_kernel<10>, grid, block, 0, s, M); break;
    default: SDF_LAUNCH(neuron_multi_kernel<20>, grid, block, 0, s, M); break;
  }
  SDF_LAUNCH_CHECK();
  return 0;
}

extern "C" int sdf_lif_fwd(const float* x, void* spike, float* v_last, int T, int64_t N, float tau, float v_th,
                           int soft_reset, float v_reset, int spike_dtype, void* stream) {
  if (N % 4) return launch_scalar(x, spike, v_last, T, N, SDF_LIF, tau, v_th, soft_reset, v_reset, spike_dtype, nullptr, nullptr, stream);
  SdfNeuronDesc d = {};
  d.x = x; d.out = spike; d.v_last = v_last; d.T = T; d.out_dtype = spike_dtype;
  d.nb = 1; d.ni = N; d.x_sb = 0; d.x_st = N; d.o_sb = 0; d.o_st = N;
  d.kind = SDF_LIF; d.tau = tau; d.v_th = v_th; d.v_reset = v_reset; d.soft_reset = soft_reset;
  return sdf_neuron_fwd(&d, stream);
}

extern "C" int sdf_psn_fwd(const float* x, const float* W, const float* b, void* spike, int T, int64_t N,
                           int spike_dtype, void* stream) {
  if (N % 4 || (T != 1 && T != 2 && T != 4 && T != 5 && T != 8 && T != 10 && T != 16 && T != 20))
    return launch_scalar(x, spike, nullptr, T, N, SDF_PSN, 2.f, 0.f, 1, 0.f, spike_dtype, W, b, stream);
  SdfNeuronDesc d = {};
  d.x = x; d.out = spike; d.T = T; d.out_dtype = spike_dtype;
  d.nb = 1; d.ni = N; d.x_sb = 0; d.x_st = N; d.o_sb = 0; d.o_st = N;
  d.kind = SDF_PSN; d.psn_w = W; d.psn_b = b;
  return sdf_neuron_fwd(&d, stream);
}

extern "C" int sdf_version(void) { return SDF_VERSION; }
