// Flow-prediction head of a U-Net level for gfx950, one launch:
//
//   p[t]  = conv1x1( SN_pred(z)[t] ) + bias          MS_SpikingPredLayer, reference Spiking_modules.py:605-640
//   flow  = nearest_upsample( sum_t p[t] )           reference Spiking_STSwinNet.py:289-303
//   and, for the NEXT decoder level (whose input is cat(p, z, skip) on channels, Spiking_STSwinNet.py:168-172, behind ITS
//   neuron, Spiking_modules.py:467-474): the spike bytes of SN_next(z) and SN_next(p) written into their channel slices
//   of that level's NHWC spike image.
//
// The three-launch form (neuron kernel -> spike GEMM with the 2 output columns padded to a 32-column block -> flow_out)
// writes z's spikes (1 B / element), a (rows, 32) fp32 product of which 2 columns are real, and re-reads it to sum T;
// the next level's neuron then reads z a second time.  Here z is read ONCE: a group of LPP lanes owns one position
// (pixel) with all T steps; each lane takes three channel quads (Cin = 12 LPP), runs the neuron over T in registers and
// accumulates its part of the two dot products; a butterfly over the group's lanes finishes them.
// Compiled with -ffp-contract=off (neuron arithmetic = the separately-rounded op sequence of neuron.hip).
#include "spike_mm.h"

namespace {
using sdfmm::neuron_T;

struct PredParams {
  SdfPredHeadDesc d;
  float inv_tau_p, inv_tau_n;
  int64_t P;                       // positions = B * h * w
  int same_next;                   // SN_next has SN_pred's settings: z's spikes are computed once
};

template <int T, int LPP, int NK>
__global__ __launch_bounds__(256) void pred_head_kernel(PredParams P) {
  constexpr int Cin = 12 * LPP, PPW = 64 / LPP;
  const SdfPredHeadDesc& d = P.d;
  __shared__ __attribute__((aligned(16))) float psn_s[NK == 1 ? 2 * sdfmm::PSN_TABLE(T) : 4];   // PSN: coefficients of SN_pred and SN_next
  if constexpr (NK == 1) {
    sdfmm::psn_stage<T>(psn_s, d.sn_pred, threadIdx.x, 256);
    if (d.next_spikes) sdfmm::psn_stage<T>(psn_s + sdfmm::PSN_TABLE(T), d.sn_next, threadIdx.x, 256);
    __syncthreads();
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane % LPP, pw = lane / LPP;
  const int64_t pos = ((int64_t)blockIdx.x * 4 + wave) * PPW + pw;
  const bool ok = pos < P.P;
  const int64_t pc = ok ? pos : 0;
  const int64_t HW = (int64_t)d.h * d.w;
  const int64_t b = pc / HW, hw = pc - b * HW;
  const int64_t row0 = (b * T) * HW + hw;                               // row of step 0; + t * HW
  float acc[T][2];
#pragma unroll
  for (int t = 0; t < T; ++t) acc[t][0] = acc[t][1] = 0.f;
  const bool to_next = d.next_spikes != nullptr, keep = d.keep_spikes != nullptr;
  // all three quads' loads are requested up front where the registers allow it (T <= 10: 30 x 16 bytes in flight per lane): the
  // small levels are a few hundred waves, each of which would otherwise pay the memory latency three times in a row
  constexpr bool AHEAD = T <= 10;
  float4 vall[AHEAD ? 3 : 1][T];
  if (AHEAD) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int t = 0; t < T; ++t) vall[AHEAD ? i : 0][t] = *reinterpret_cast<const float4*>(d.z + (row0 + t * HW) * Cin + 4 * (g + LPP * i));
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int c = 4 * (g + LPP * i);
    float4 v[T];
#pragma unroll
    for (int t = 0; t < T; ++t) v[t] = AHEAD ? vall[AHEAD ? i : 0][t] : *reinterpret_cast<const float4*>(d.z + (row0 + t * HW) * Cin + c);
    const float4 w0 = *reinterpret_cast<const float4*>(d.wgt + c), w1 = *reinterpret_cast<const float4*>(d.wgt + Cin + c);
    uint32_t pk[T], pk2[T];
#pragma unroll
    for (int t = 0; t < T; ++t) pk[t] = pk2[t] = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float xs[T], sp[T];
#pragma unroll
      for (int t = 0; t < T; ++t) xs[t] = e == 0 ? v[t].x : (e == 1 ? v[t].y : (e == 2 ? v[t].z : v[t].w));
      if constexpr (NK == 1) sdfmm::psn_T_lds<T>(xs, sp, psn_s); else neuron_T<NK, T>(xs, sp, d.sn_pred, P.inv_tau_p);
      const float we0 = e == 0 ? w0.x : (e == 1 ? w0.y : (e == 2 ? w0.z : w0.w));
      const float we1 = e == 0 ? w1.x : (e == 1 ? w1.y : (e == 2 ? w1.z : w1.w));
#pragma unroll
      for (int t = 0; t < T; ++t) {
        acc[t][0] = __builtin_fmaf(sp[t], we0, acc[t][0]);
        acc[t][1] = __builtin_fmaf(sp[t], we1, acc[t][1]);
        pk[t] |= ((__float_as_uint(sp[t]) >> 29) & 1u) << (8 * e);          // 1.0f has bit 29 set
      }
      if (to_next && !P.same_next) {
        if constexpr (NK == 1) sdfmm::psn_T_lds<T>(xs, sp, psn_s + sdfmm::PSN_TABLE(T)); else neuron_T<NK, T>(xs, sp, d.sn_next, P.inv_tau_n);
#pragma unroll
        for (int t = 0; t < T; ++t) pk2[t] |= ((__float_as_uint(sp[t]) >> 29) & 1u) << (8 * e);
      }
    }
    if (ok && keep) {
#pragma unroll
      for (int t = 0; t < T; ++t) *reinterpret_cast<uint32_t*>(d.keep_spikes + (row0 + t * HW) * Cin + c) = pk[t];
    }
    if (ok && to_next) {
#pragma unroll
      for (int t = 0; t < T; ++t)
        *reinterpret_cast<uint32_t*>(d.next_spikes + (row0 + t * HW) * d.next_ld + d.next_z_off + c) = P.same_next ? pk[t] : pk2[t];
    }
  }
  // the group's lanes each hold a partial of the 2 T dot products: xor butterfly (every lane ends with the totals)
#pragma unroll
  for (int m = 1; m < LPP; m <<= 1) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      acc[t][0] += __shfl_xor(acc[t][0], m, 64);
      acc[t][1] += __shfl_xor(acc[t][1], m, 64);
    }
  }
  const float b0 = d.bias ? d.bias[0] : 0.f, b1 = d.bias ? d.bias[1] : 0.f;
  float f0 = 0.f, f1 = 0.f;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    acc[t][0] += b0; acc[t][1] += b1;
    f0 += acc[t][0]; f1 += acc[t][1];
  }
  if (!ok) return;
  if (g == 0) {
    if (d.pred) {
#pragma unroll
      for (int t = 0; t < T; ++t) *reinterpret_cast<float4*>(d.pred + (row0 + t * HW) * 4) = make_float4(acc[t][0], acc[t][1], 0.f, 0.f);
    }
    if (to_next) {
      // SN_next over the two prediction channels (the slice is 4 wide in the image; channels 2, 3 carry zero weights downstream)
      float sp0[T], sp1[T], xs[T];
#pragma unroll
      for (int t = 0; t < T; ++t) xs[t] = acc[t][0];
      if constexpr (NK == 1) sdfmm::psn_T_lds<T>(xs, sp0, psn_s + sdfmm::PSN_TABLE(T)); else neuron_T<NK, T>(xs, sp0, d.sn_next, P.inv_tau_n);
#pragma unroll
      for (int t = 0; t < T; ++t) xs[t] = acc[t][1];
      if constexpr (NK == 1) sdfmm::psn_T_lds<T>(xs, sp1, psn_s + sdfmm::PSN_TABLE(T)); else neuron_T<NK, T>(xs, sp1, d.sn_next, P.inv_tau_n);
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const uint32_t w = ((__float_as_uint(sp0[t]) >> 29) & 1u) | (((__float_as_uint(sp1[t]) >> 29) & 1u) << 8);
        uint8_t* dst = d.next_spikes + (row0 + t * HW) * d.next_ld;
        *reinterpret_cast<uint32_t*>(dst + d.next_pred_off) = w;
        for (int zo = 0; zo < d.next_zero_len; zo += 4) *reinterpret_cast<uint32_t*>(dst + d.next_zero_off + zo) = 0u;
      }
    }
  }
  if (d.flow) {
    // nearest upsampling by whole factors (sy, sx): this position's sy x sx block of both channels, dealt over the group's lanes
    const int sy = d.H / d.h, sx = d.W / d.w, blk = sy * sx;
    const int y = (int)(hw / d.w), x = (int)(hw - (int64_t)y * d.w);
    for (int i = g; i < 2 * blk; i += LPP) {
      const int o = i / blk, r = i - o * blk;
      const int dy = r / sx, dx = r - dy * sx;
      d.flow[((b * 2 + o) * d.H + (y * sy + dy)) * (int64_t)d.W + x * sx + dx] = o ? f1 : f0;
    }
  }
}

template <int T, int LPP>
int launch_nk(const PredParams& P, int nk, dim3 grid, hipStream_t s) {
  switch (nk) {
    case 0: SDF_LAUNCH((pred_head_kernel<T, LPP, 0>), grid, dim3(256), 0, s, P); return 0;
    case 1:
      if constexpr (T <= 10) { SDF_LAUNCH((pred_head_kernel<T, LPP, 1>), grid, dim3(256), 0, s, P); return 0; }
      return SDF_E_SHAPE;
    default: SDF_LAUNCH((pred_head_kernel<T, LPP, 2>), grid, dim3(256), 0, s, P); return 0;
  }
}

template <int T>
int launch_lpp(const PredParams& P, int nk, hipStream_t s) {
  const int lpp = P.d.Cin / 12;
  const int64_t per_wg = 4 * (64 / lpp), wgs = (P.P + per_wg - 1) / per_wg;
  if (wgs >= (1LL << 31)) return SDF_E_SHAPE;
  const dim3 grid((unsigned)wgs);
  switch (lpp) {
    case 8: return launch_nk<T, 8>(P, nk, grid, s);
    case 16: return launch_nk<T, 16>(P, nk, grid, s);
    case 32: return launch_nk<T, 32>(P, nk, grid, s);
    default: return SDF_E_SHAPE;
  }
}

bool same_neuron(const SdfNeuronCfg& a, const SdfNeuronCfg& b) {
  return a.kind == b.kind && a.tau == b.tau && a.v_th == b.v_th && a.v_reset == b.v_reset && a.soft_reset == b.soft_reset &&
         a.psn_w == b.psn_w && a.psn_b == b.psn_b;
}

int check_neuron(const SdfNeuronCfg& n) {
  if (n.kind != SDF_LIF && n.kind != SDF_PSN && n.kind != SDF_IF) return SDF_E_DTYPE;
  if (n.kind == SDF_PSN && (!n.psn_w || !n.psn_b)) return SDF_E_NULL;
  if (!sdf_tau_ok(n.kind, n.tau)) return SDF_E_SHAPE;
  return 0;
}
}  // namespace

extern "C" int sdf_pred_head_fwd(const SdfPredHeadDesc* d, void* stream) {
  if (!d) return SDF_E_NULL;
  if (!d->z || !d->wgt) return SDF_E_NULL;
  if (!d->pred && !d->flow && !d->next_spikes) return SDF_E_NULL;
  if (d->B < 1 || d->h < 1 || d->w < 1) return SDF_E_SHAPE;
  if (d->Cin != 96 && d->Cin != 192 && d->Cin != 384) return SDF_E_SHAPE;          // 12 channels per lane, 8 / 16 / 32 lanes per position
  if (d->D != 5 && d->D != 10 && d->D != 20) return SDF_E_SHAPE;
  int rc = check_neuron(d->sn_pred);
  if (rc) return rc;
  const int nk = sdfmm::neuron_class(d->sn_pred);
  if (nk == 1 && d->D > 10) return SDF_E_SHAPE;
  if (d->flow && (d->H < d->h || d->W < d->w || d->H % d->h || d->W % d->w)) return SDF_E_SHAPE;   // whole upsampling factors only
  if (d->next_spikes) {
    rc = check_neuron(d->sn_next);
    if (rc) return rc;
    if (sdfmm::neuron_class(d->sn_next) != nk) return SDF_E_SHAPE;
    if (d->next_ld < 4 || d->next_ld % 4 || d->next_z_off % 4 || d->next_pred_off % 4 || d->next_zero_off % 4 || d->next_zero_len % 4 ||
        d->next_z_off < 0 || d->next_pred_off < 0 || d->next_zero_len < 0 || d->next_z_off + d->Cin > d->next_ld ||
        d->next_pred_off + 4 > d->next_ld || (d->next_zero_len > 0 && d->next_zero_off + d->next_zero_len > d->next_ld))
      return SDF_E_SHAPE;
    if (!sdf_aligned(d->next_spikes, 4)) return SDF_E_ALIGN;
  }
  if (!sdf_aligned(d->z, 16) || !sdf_aligned(d->wgt, 16) || (d->pred && !sdf_aligned(d->pred, 16)) ||
      (d->keep_spikes && !sdf_aligned(d->keep_spikes, 4)))
    return SDF_E_ALIGN;
  PredParams P;
  P.d = *d;
  P.inv_tau_p = sdfmm::inv_tau_of(d->sn_pred);
  P.inv_tau_n = d->next_spikes ? sdfmm::inv_tau_of(d->sn_next) : 0.f;
  P.P = (int64_t)d->B * d->h * d->w;
  P.same_next = d->next_spikes && same_neuron(d->sn_pred, d->sn_next);
  hipStream_t s = sdf_stream(stream);
  switch (d->D) {
    case 5: rc = launch_lpp<5>(P, nk, s); break;
    case 10: rc = launch_lpp<10>(P, nk, s); break;
    default: rc = launch_lpp<20>(P, nk, s); break;
  }
  if (rc) return rc;
  SDF_LAUNCH_CHECK();
  return 0;
}
