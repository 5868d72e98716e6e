// The spiking QK window attention as ONE entry point of the C ABI (`sdf_qk_attn_fwd`), and the window index
// table it consumes, so that a host in any language drives rows a4 - a6 with two calls and no index code of its own:
//
//   sdf_window_slice_map : pad + roll(-shift) + window_partition_v2 + the raw .view(Wd, B_, Wh, Ww, C) of the reference
//                          (Spiking_swin_transformer3D.py:789-804, :100-113) as an int32 gather table, built on the device;
//                          read backwards it is window_reverse + roll(+shift) + crop (:810-820)
//   sdf_qk_attn_fwd      : x += SSA(x) for Spiking_QK_WindowAttention3D (:661-717, :781-821, :840) - four launches on the
//                          caller's stream: neuron over the gathered slices -> [Wq;Wk] spike GEMM with BN (+ positional
//                          term) and the q / k neurons fused -> token gate (group sums, sn2_q, AND) -> projection spike
//                          GEMM reading through the head scramble, bias + BN + scatter + residual in its epilogue.
// The intermediates (three u8 spike tensors) live in the caller's workspace; nothing is allocated here.
#include "spike_mm.h"
#include "switches.h"
#include <stdlib.h>

namespace {

__global__ __launch_bounds__(256) void slice_map_kernel(int32_t* __restrict__ map, int B, int D, int H, int W, int Wd, int Wh,
                                                        int Ww, int sd, int sh, int sw, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int N1 = Wh * Ww;
  const int Dp = (D + Wd - 1) / Wd * Wd, Hp = (H + Wh - 1) / Wh * Wh, Wp = (W + Ww - 1) / Ww * Ww;
  const int nD = Dp / Wd, nHb = Hp / Wh, nWb = Wp / Ww;
  const int tok = (int)(i % N1);
  const int64_t j = i / N1;                                   // slice = window * Wd + frame inside the window
  const int wd = (int)(j % Wd);
  const int64_t win = j / Wd;
  const int wb = (int)(win % nWb), hb = (int)((win / nWb) % nHb), db = (int)((win / ((int64_t)nWb * nHb)) % nD);
  const int b = (int)(win / ((int64_t)nWb * nHb * nD));
  const int d = (db * Wd + wd + sd) % Dp, h = (hb * Wh + tok / Ww + sh) % Hp, w = (wb * Ww + tok % Ww + sw) % Wp;
  map[i] = (d < D && h < H && w < W) ? (int32_t)((((int64_t)b * D + d) * H + h) * W + w) : -1;
}

void fill_neuron(SdfNeuronDesc& n, const SdfNeuronCfg& c) {
  n.kind = c.kind; n.tau = c.tau; n.v_th = c.v_th; n.v_reset = c.v_reset; n.soft_reset = c.soft_reset;
  n.psn_w = c.psn_w; n.psn_b = c.psn_b;
}

void fill_gemm_neuron(SdfSpikeGemmDesc& g, const SdfNeuronCfg& c, int T) {
  g.sn_T = T; g.sn_kind = c.kind; g.tau = c.tau; g.v_th = c.v_th; g.v_reset = c.v_reset; g.soft_reset = c.soft_reset;
  g.psn_w = c.psn_w; g.psn_b = c.psn_b;
}

}  // namespace

extern "C" int sdf_window_slice_map(int32_t* map, int B, int D, int H, int W, int Wd, int Wh, int Ww, int shift_d, int shift_h,
                                    int shift_w, int64_t* n_windows, void* stream) {
  if (!map) return SDF_E_NULL;
  if (B < 1 || D < 1 || H < 1 || W < 1 || Wd < 1 || Wh < 1 || Ww < 1 || shift_d < 0 || shift_h < 0 || shift_w < 0) return SDF_E_SHAPE;
  const int64_t nD = (D + Wd - 1) / Wd, nHb = (H + Wh - 1) / Wh, nWb = (W + Ww - 1) / Ww;
  const int64_t B_ = (int64_t)B * nD * nHb * nWb, total = B_ * Wd * Wh * Ww;
  if ((int64_t)B * D * H * W >= (1LL << 31) || total >= (1LL << 40)) return SDF_E_SHAPE;
  if (n_windows) *n_windows = B_;
  SDF_LAUNCH(slice_map_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, sdf_stream(stream), map, B, D, H, W,
                     Wd, Wh, Ww, shift_d, shift_h, shift_w, total);
  SDF_LAUNCH_CHECK();
  return 0;
}

extern "C" int64_t sdf_qk_attn_workspace_bytes(int64_t B_, int Tq, int N1, int C) {
  if (B_ < 1 || Tq < 1 || N1 < 1 || C < 1) return 0;
  const int64_t M = B_ * Tq * N1;
  // spikes of the slices (later E), then q | k; the wide-stage form (ms_wide.hip) keeps the slice spikes in a third region
  return ((M * C + 255) / 256 * 256) + ((M * 2 * C + 255) / 256 * 256) + M * C;
}

extern "C" int sdf_qk_attn_is_wide(const SdfQkAttnDesc* d) {
  if (!d || !d->x || !d->slice_map || !d->p_planes) return 0;
  if (d->B_ < 1 || d->Tq < 1 || d->N1 < 1 || d->C < 32 || d->C % 32 || d->nH < 1 || d->C != d->nH * 32) return 0;
  return sdfmm::ms_wide_attn_supports(d) ? 1 : 0;
}

extern "C" int sdf_ms_mlp_is_wide(const SdfMsMlpDesc* d) {
  if (!d || !d->x || !d->fc1_planes || !d->fc2_planes) return 0;
  if (d->B < 1 || d->D < 1 || d->HW < 1 || d->C < 32 || d->C % 32 || d->Ch < 32 || d->Ch % 32) return 0;
  return sdfmm::ms_wide_mlp_supports(d) ? 1 : 0;
}

extern "C" int sdf_qk_attn_fwd(const SdfQkAttnDesc* d, void* stream) {
  if (!d) return SDF_E_NULL;
  if (!d->x || !d->slice_map || !d->workspace || !d->p_planes) return SDF_E_NULL;
  const bool fused = d->qk_planes != nullptr;
  if (!fused && (!d->q_planes || !d->k_planes)) return SDF_E_NULL;
  if (d->B_ < 1 || d->Tq < 1 || d->N1 < 1 || d->C < 32 || d->C % 32 || d->nH < 1 || d->C != d->nH * 32) return SDF_E_SHAPE;
  if (d->workspace_bytes < sdf_qk_attn_workspace_bytes(d->B_, d->Tq, d->N1, d->C)) return SDF_E_SHAPE;
  if (d->rep_windows < 0 || (d->rep_windows > 0 && (d->B_ % d->rep_windows || d->rep_windows % d->Tq))) return SDF_E_SHAPE;
  if (!sdf_aligned(d->workspace, 256)) return SDF_E_ALIGN;
  const int C = d->C, Tq = d->Tq;
  const int64_t rows = d->B_ * d->N1, M = rows * Tq;
  uint8_t* xs = reinterpret_cast<uint8_t*>(d->workspace);
  uint8_t* qk = xs + (M * C + 255) / 256 * 256;

  // wide stages: slice neuron -> [q | k + BN + PE + neurons + gate] -> position-major projection (+ the MLP's first neuron)
  if (sdfmm::ms_wide_attn_supports(d)) {
    uint8_t* e = xs;
    uint8_t* xsw = qk + (M * 2 * C + 255) / 256 * 256;
    SdfNeuronDesc n = {};
    n.x = d->x; n.out = xsw; n.T = Tq; n.out_dtype = SDF_U8;
    n.nb = 1; n.ni = rows * C; n.x_sb = 0; n.x_st = 0; n.o_sb = 0; n.o_st = rows * C;
    n.rowmap = d->slice_map; n.rowlen = C;
    fill_neuron(n, d->sn_proj);
    int rcw = sdf_neuron_fwd(&n, stream);
    if (rcw) return rcw;
    rcw = sdfmm::launch_ms_wide_front(d, xsw, e, qk, (d->flags & SDF_QK_KEEP_SPIKES) != 0, sdf_stream(stream));
    if (rcw) return rcw;
    return sdfmm::launch_ms_wide_proj(d, e, sdf_stream(stream));
  }
  if (d->emit_s1) return SDF_E_SHAPE;                           // only the wide-stage form emits the next neuron's spikes

  // steps 1 - 3 as one launch where the kernel has an instantiation (SDF_QK_FRONT=0 / SDF_QK_FOUR_LAUNCHES: the A/B reference below)
  bool front = false;
  {
    const char* e_front = sdf_sw(SW_QK_FRONT);                    // (read per call, like SDF_MLP_FUSED)
    const bool off = e_front && e_front[0] == '0';
    if (!off && !(d->flags & SDF_QK_FOUR_LAUNCHES) && sdfmm::qk_front_supports(d)) {
      const int rc0 = sdfmm::launch_qk_front(d, xs, qk, (d->flags & SDF_QK_KEEP_SPIKES) != 0, sdf_stream(stream));
      if (rc0) return rc0;
      front = true;
    }
  }
  int rc = 0;
  if (!front) {
  // 1. proj_sn over the T' frames of every window slice, gathered through the slice map (pad / roll / partition folded in)
  SdfNeuronDesc n = {};
  n.x = d->x; n.out = xs; n.T = Tq; n.out_dtype = SDF_U8;
  n.nb = 1; n.ni = rows * C; n.x_sb = 0; n.x_st = 0; n.o_sb = 0; n.o_st = rows * C;
  n.rowmap = d->slice_map; n.rowlen = C;
  fill_neuron(n, d->sn_proj);
  rc = sdf_neuron_fwd(&n, stream);
  if (rc) return rc;

  // 2. q = SN(BN(xs Wq^T)), k = SN(BN(xs Wk^T) + PE): the neuron runs in the GEMM epilogue, q / k never exist in fp32
  auto qk_gemm = [&](const uint16_t* planes, float acc_scale, int N, const float* alpha, const float* beta, const float* add,
                     const SdfNeuronCfg& sn, uint8_t* out) {
    SdfSpikeGemmDesc g = {};
    g.A = xs; g.Wp = planes; g.out_spike = out; g.M = M; g.N = N; g.K = C; g.lda = C; g.ldo = N; g.nsplit = d->nsplit;
    g.acc_scale = acc_scale; g.alpha = alpha; g.beta = beta; g.add = add; g.add_prows = d->N1;
    g.pos_count = rows; g.pos_inner = rows; g.pos_ostride = 0; g.t_stride = rows;
    fill_gemm_neuron(g, sn, Tq);
    return sdf_spike_gemm_fwd(&g, stream);
  };
  int64_t ldq, ldk;
  const uint8_t *qp, *kp;
  if (fused) {
    rc = qk_gemm(d->qk_planes, d->qk_acc_scale, 2 * C, d->qk_alpha, d->qk_beta, d->qk_add, d->sn_q, qk);
    if (rc) return rc;
    qp = qk; kp = qk + C; ldq = ldk = 2 * C;
  } else {
    rc = qk_gemm(d->q_planes, d->q_acc_scale, C, d->q_alpha, d->q_beta, nullptr, d->sn_q, qk);
    if (rc) return rc;
    rc = qk_gemm(d->k_planes, d->k_acc_scale, C, d->k_alpha, d->k_beta, d->k_add, d->sn_k, qk + M * C);
    if (rc) return rc;
    qp = qk; kp = qk + M * C; ldq = ldk = C;
  }

  // 3. token gate: A = sn2_q(sum over each head's 32 channels of q), E = k AND A  (E overwrites xs)
  rc = sdf_qk_gate_strided_fwd(qp, kp, xs, Tq, rows, C, ldq, ldk, d->sn2_q.kind, d->sn2_q.tau, d->sn2_q.v_th, d->sn2_q.v_reset,
                               d->sn2_q.soft_reset, d->sn2_q.psn_w, d->sn2_q.psn_b, stream);
  if (rc) return rc;
  }

  // 4. x[slice_map] += BN(Z Wp^T + b), Z = E read through the reference's raw head reshape
  SdfSpikeGemmDesc g = {};
  g.A = xs; g.Wp = d->p_planes; g.out = d->x; g.M = M; g.N = C; g.K = C; g.lda = C; g.ldo = C; g.nsplit = d->nsplit;
  g.acc_scale = d->p_acc_scale; g.bias = d->p_bias; g.alpha = d->p_alpha; g.beta = d->p_beta; g.resid = d->x;
  g.out_rowmap = d->slice_map; g.out_rows = d->x_rows;
  g.zg_nH = d->nH; g.zg_T = Tq; g.zg_B = (int32_t)d->B_; g.zg_N1 = d->N1; g.zg_rep = d->rep_windows;
  g.workspace = d->gemm_workspace; g.workspace_bytes = d->gemm_workspace_bytes;
  return sdf_spike_gemm_fwd(&g, stream);
}

// ---------------------------------------------------------------------------------------------------------
// MS MLP, whole (row a7): x += BN2(W2 SN2(BN1(W1 SN1(x))))  over the true time axis D of a (B, D, HW, C) buffer.
extern "C" int64_t sdf_ms_mlp_workspace_bytes(int64_t tokens, int C, int Ch) {
  if (tokens < 1 || C < 1 || Ch < 1) return 0;
  // (+ 80 rows each: the tiled hand-over layout of the wide-stage kernels rounds the rows up to whole 80-row units)
  return (((tokens + 80) * C + 255) / 256 * 256) + (tokens + 80) * Ch;
}

extern "C" int sdf_ms_mlp_fwd(const SdfMsMlpDesc* d, void* stream) {
  if (!d) return SDF_E_NULL;
  if (!d->x || !d->fc1_planes || !d->fc2_planes || !d->workspace) return SDF_E_NULL;
  if (d->B < 1 || d->D < 1 || d->HW < 1 || d->C < 32 || d->C % 32 || d->Ch < 32 || d->Ch % 32) return SDF_E_SHAPE;
  const int64_t tokens = (int64_t)d->B * d->D * d->HW;
  if (d->workspace_bytes < sdf_ms_mlp_workspace_bytes(tokens, d->C, d->Ch)) return SDF_E_SHAPE;
  if (!sdf_aligned(d->workspace, 256)) return SDF_E_ALIGN;
  const int C = d->C, Ch = d->Ch, D = d->D;
  const int64_t hw = d->HW;
  uint8_t* s1 = reinterpret_cast<uint8_t*>(d->workspace);
  uint8_t* s2 = s1 + (tokens * C + 255) / 256 * 256;
  // wide stages: [SN1 unless the attention's projection already emitted it] -> fc1 + BN1 + SN2 -> fc2 + BN2 + shortcut.  With the
  // tape both spike tensors are row-major at the documented offsets; without it they travel in the tiled hand-over layout
  if (sdfmm::ms_wide_mlp_supports(d)) {
    if (d->s1_in && d->s1_in != s1) return SDF_E_SHAPE;
    const bool tape = (d->flags & SDF_MLP_KEEP_SPIKES) != 0;
    uint8_t* s2w = tape ? s2 : s1 + ((tokens + 80) * C + 255) / 256 * 256;
    if (!d->s1_in) {
      SdfNeuronDesc n = {};
      n.x = d->x; n.out = s1; n.T = D; n.out_dtype = SDF_U8;
      n.nb = d->B; n.ni = hw * C; n.x_sb = (int64_t)D * hw * C; n.x_st = hw * C; n.o_sb = (int64_t)D * hw * C; n.o_st = hw * C;
      fill_neuron(n, d->sn1);
      const int rcw = sdf_neuron_fwd(&n, stream);
      if (rcw) return rcw;
    }
    return sdfmm::launch_ms_wide_mlp(d, s1, d->s1_in != nullptr && !tape, s2w, !tape, sdf_stream(stream));
  }
  if (d->s1_in || d->emit_next) return SDF_E_SHAPE;            // wide-stage inputs / outputs
  // one launch where the kernel has an instantiation (SDF_MLP_FUSED=0 / SDF_MLP_THREE_LAUNCHES: the A/B reference below)
  {
    const char* e = sdf_sw(SW_MLP_FUSED);
    if (!(d->flags & SDF_MLP_THREE_LAUNCHES) && !(e && e[0] == '0') && sdfmm::ms_mlp_fused_supports(d)) {
      const bool keep = (d->flags & SDF_MLP_KEEP_SPIKES) != 0;
      return sdfmm::launch_ms_mlp_fused(d, keep ? s1 : nullptr, keep ? s2 : nullptr, sdf_stream(stream));
    }
  }
  // 1. sn1 over the D steps of every (b, hw, c)
  SdfNeuronDesc n = {};
  n.x = d->x; n.out = s1; n.T = D; n.out_dtype = SDF_U8;
  n.nb = d->B; n.ni = hw * C; n.x_sb = (int64_t)D * hw * C; n.x_st = hw * C; n.o_sb = (int64_t)D * hw * C; n.o_st = hw * C;
  fill_neuron(n, d->sn1);
  int rc = sdf_neuron_fwd(&n, stream);
  if (rc) return rc;
  // 2. s2 = SN2(BN1(s1 W1^T)): neuron fused into the GEMM epilogue, the 4C hidden tensor never exists in fp32
  SdfSpikeGemmDesc g = {};
  g.A = s1; g.Wp = d->fc1_planes; g.out_spike = s2; g.M = tokens; g.N = Ch; g.K = C; g.lda = C; g.ldo = Ch; g.nsplit = d->nsplit;
  g.acc_scale = d->fc1_acc_scale; g.alpha = d->fc1_alpha; g.beta = d->fc1_beta;
  g.pos_count = (int64_t)d->B * hw; g.pos_inner = hw; g.pos_ostride = (int64_t)D * hw; g.t_stride = hw;
  fill_gemm_neuron(g, d->sn2, D);
  rc = sdf_spike_gemm_fwd(&g, stream);
  if (rc) return rc;
  // 3. x += BN2(s2 W2^T)
  SdfSpikeGemmDesc h = {};
  h.A = s2; h.Wp = d->fc2_planes; h.out = d->x; h.M = tokens; h.N = C; h.K = Ch; h.lda = Ch; h.ldo = C; h.nsplit = d->nsplit;
  h.acc_scale = d->fc2_acc_scale; h.alpha = d->fc2_alpha; h.beta = d->fc2_beta; h.resid = d->x;
  h.workspace = d->gemm_workspace; h.workspace_bytes = d->gemm_workspace_bytes;
  return sdf_spike_gemm_fwd(&h, stream);
}

extern "C" int sdf_spike_gemm_bn_fwd(const uint8_t* A_spike, const uint16_t* W_planes, int nsplit, float acc_scale,
                                     const float* bn_a, const float* bn_b, float* out, int64_t M, int K, int N, void* stream) {
  SdfSpikeGemmDesc g = {};
  g.A = A_spike; g.Wp = W_planes; g.out = out; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldo = N; g.nsplit = nsplit;
  g.acc_scale = acc_scale; g.alpha = bn_a; g.beta = bn_b;
  return sdf_spike_gemm_fwd(&g, stream);
}
