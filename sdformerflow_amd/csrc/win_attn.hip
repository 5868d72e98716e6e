// Fused 3-D window attention for gfx950: QK^T + bias (+mask) [+softmax] . V over one (T,H,W) window and head.
//
//   mode SDF_ATTN_ANN : cosine attention of the ANN WindowAttention3D (reference
//        models/STSwinNet/swin_transformer3D_v2.py:169-205): normalize(q) normalize(k)^T * logit_scale[g]
//        + bias[g] (+ mask[w]) -> softmax -> @ v, q/k/v read straight out of the packed qkv GEMM output.
//   mode SDF_ATTN_SEW : spiking Spiking_BN_WindowAttention3D (reference
//        models/STSwinNet_SNN/Spiking_swin_transformer3D.py:320-363): binary q/k/v in the reference's raw
//        (B_,nH,N,hd) head view, (q*scale) k^T + bias[g] (+ mask[w]), NO softmax, @ v, output written through
//        the reference's (B_,nH,T',N1,hd) -> (T',B_,N1,C) scramble.
//
// One workgroup (4 wavefronts of 64) per (window, head).  q/k/v (N<=192 tokens x 32 dims, fp32) are staged
// once in LDS (k and q L2-normalised on the way in for the ANN mode).  Each wave owns 16-query tiles and keeps
// the whole 16 x N score strip in registers: S^T = K Q^T is computed with v_mfma_f32_16x16x4_f32 (exact fp32,
// k-ordered fmaf chain) so that the accumulator of key tile jt *is* the A operand of the P.V product
// (keys permuted inside each 4-deep MFMA step: step s of tile jt covers keys 16jt + 4*(lane>>4) + s), i.e. no
// LDS round trip and no shuffle between the two matrix products.  Row max / sum reduce in-lane over the 4*NT
// registers and across the 4 lane groups with two DPP-free shuffles.  Bias and mask are read as float4
// (4 consecutive keys) from L2-resident tables.
#include "common.h"

namespace {

constexpr int HD = 32;
constexpr int NT_MAX = 12;            // up to 192 tokens per window
constexpr int LDW = HD + 4;           // padded LDS row (floats)

typedef __attribute__((ext_vector_type(4))) float f32x4;

struct AttnParams {
  SdfWinAttnDesc d;
};

template <int MODE>
__global__ __launch_bounds__(256) void win_attn_kernel(AttnParams P) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const SdfWinAttnDesc& d = P.d;
  const int N = d.N, NT = (N + 15) / 16, NP = NT * 16;
  float* Ks = lds;
  float* Qs = Ks + NP * LDW;
  float* Vs = Qs + NP * LDW;

  const int bg = blockIdx.x;
  const int b = bg / d.nH, g = bg - b * d.nH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int C = d.nH * HD;

  // ---- stage q, k, v rows (one row of 32 dims per thread) ----
  for (int r = tid; r < NP; r += 256) {
    float4 qv[8], kv[8], vv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) qv[i] = kv[i] = vv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < N) {
      if (MODE == SDF_ATTN_ANN) {
        const float* base = reinterpret_cast<const float*>(d.q) + ((int64_t)b * N + r) * 3 * C + g * HD;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          qv[i] = *reinterpret_cast<const float4*>(base + 4 * i);
          kv[i] = *reinterpret_cast<const float4*>(base + C + 4 * i);
          vv[i] = *reinterpret_cast<const float4*>(base + 2 * C + 4 * i);
        }
        // F.normalize(x, dim=-1): x / max(||x||_2, 1e-12)
        float sq = 0.f, sk = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          sq += qv[i].x * qv[i].x + qv[i].y * qv[i].y + qv[i].z * qv[i].z + qv[i].w * qv[i].w;
          sk += kv[i].x * kv[i].x + kv[i].y * kv[i].y + kv[i].z * kv[i].z + kv[i].w * kv[i].w;
        }
        const float iq = 1.f / fmaxf(sqrtf(sq), 1e-12f), ik = 1.f / fmaxf(sqrtf(sk), 1e-12f);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          qv[i].x *= iq; qv[i].y *= iq; qv[i].z *= iq; qv[i].w *= iq;
          kv[i].x *= ik; kv[i].y *= ik; kv[i].z *= ik; kv[i].w *= ik;
        }
      } else {
        const int64_t off = (((int64_t)b * d.nH + g) * N + r) * HD;
        const uint8_t* qp = reinterpret_cast<const uint8_t*>(d.q) + off;
        const uint8_t* kp = reinterpret_cast<const uint8_t*>(d.k) + off;
        const uint8_t* vp = reinterpret_cast<const uint8_t*>(d.v) + off;
        const float sc = d.scale[g];                      // q * scale (reference :26)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          uint32_t wq = *reinterpret_cast<const uint32_t*>(qp + 4 * i);
          uint32_t wk = *reinterpret_cast<const uint32_t*>(kp + 4 * i);
          uint32_t wv = *reinterpret_cast<const uint32_t*>(vp + 4 * i);
          qv[i] = make_float4((wq & 0xff) ? sc : 0.f, ((wq >> 8) & 0xff) ? sc : 0.f, ((wq >> 16) & 0xff) ? sc : 0.f,
                              (wq >> 24) ? sc : 0.f);
          kv[i] = make_float4((float)(wk & 0xff), (float)((wk >> 8) & 0xff), (float)((wk >> 16) & 0xff), (float)(wk >> 24));
          vv[i] = make_float4((float)(wv & 0xff), (float)((wv >> 8) & 0xff), (float)((wv >> 16) & 0xff), (float)(wv >> 24));
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      *reinterpret_cast<float4*>(&Qs[r * LDW + 4 * i]) = qv[i];
      *reinterpret_cast<float4*>(&Ks[r * LDW + 4 * i]) = kv[i];
      *reinterpret_cast<float4*>(&Vs[r * LDW + 4 * i]) = vv[i];
    }
  }
  __syncthreads();

  const int l15 = lane & 15, lg = lane >> 4;
  const float ls = (MODE == SDF_ATTN_ANN) ? d.scale[g] : 1.f;
  const float* bias_g = d.bias + (int64_t)g * N * N;
  const float* mask_w = d.mask ? d.mask + (int64_t)(b % d.nW) * N * N : nullptr;

  for (int qt = wave; qt < NT; qt += 4) {
    const int qi = qt * 16 + l15;                          // this lane's query
    // B operand of S^T = K Q^T: Q[qi][8*lg + s], s = 0..7
    float qreg[8];
    {
      float4 a = *reinterpret_cast<const float4*>(&Qs[qi * LDW + 8 * lg]);
      float4 c = *reinterpret_cast<const float4*>(&Qs[qi * LDW + 8 * lg + 4]);
      qreg[0] = a.x; qreg[1] = a.y; qreg[2] = a.z; qreg[3] = a.w;
      qreg[4] = c.x; qreg[5] = c.y; qreg[6] = c.z; qreg[7] = c.w;
    }
    f32x4 st[NT_MAX];
#pragma unroll
    for (int jt = 0; jt < NT_MAX; ++jt) {
      if (jt < NT) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const int kj = jt * 16 + l15;                      // A operand row: key
        float4 a = *reinterpret_cast<const float4*>(&Ks[kj * LDW + 8 * lg]);
        float4 c = *reinterpret_cast<const float4*>(&Ks[kj * LDW + 8 * lg + 4]);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, qreg[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, qreg[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, qreg[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, qreg[3], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(c.x, qreg[4], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(c.y, qreg[5], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(c.z, qreg[6], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(c.w, qreg[7], acc, 0, 0, 0);
        // lane holds S[qi][kb + r], kb = 16*jt + 4*lg : scale, bias, mask
        const int kb = jt * 16 + 4 * lg;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), mv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (qi < N && kb < N) {            // N % 2 == 0 is enough for the pairwise tail below; N % 4 != 0 handled
          const float* bp = bias_g + (int64_t)qi * N + kb;
          bv.x = bp[0];
          if (kb + 1 < N) bv.y = bp[1];
          if (kb + 2 < N) bv.z = bp[2];
          if (kb + 3 < N) bv.w = bp[3];
          if (mask_w) {
            const float* mp = mask_w + (int64_t)qi * N + kb;
            mv.x = mp[0];
            if (kb + 1 < N) mv.y = mp[1];
            if (kb + 2 < N) mv.z = mp[2];
            if (kb + 3 < N) mv.w = mp[3];
          }
        }
        const float NEG = (MODE == SDF_ATTN_ANN) ? -INFINITY : 0.f;
        // attn = qk * logit_scale + bias (+ mask): separately rounded, as the reference computes it
        acc[0] = (kb + 0 < N) ? (acc[0] * ls + bv.x) + mv.x : NEG;
        acc[1] = (kb + 1 < N) ? (acc[1] * ls + bv.y) + mv.y : NEG;
        acc[2] = (kb + 2 < N) ? (acc[2] * ls + bv.z) + mv.z : NEG;
        acc[3] = (kb + 3 < N) ? (acc[3] * ls + bv.w) + mv.w : NEG;
        st[jt] = acc;
      }
    }
    if (MODE == SDF_ATTN_ANN) {
      float m = -INFINITY;
#pragma unroll
      for (int jt = 0; jt < NT_MAX; ++jt)
        if (jt < NT) m = fmaxf(fmaxf(fmaxf(m, st[jt][0]), fmaxf(st[jt][1], st[jt][2])), st[jt][3]);
      m = fmaxf(m, __shfl_xor(m, 16));
      m = fmaxf(m, __shfl_xor(m, 32));
      float sum = 0.f;
#pragma unroll
      for (int jt = 0; jt < NT_MAX; ++jt)
        if (jt < NT) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float e = expf(st[jt][r] - m);
            st[jt][r] = e;
            sum += e;
          }
        }
      sum += __shfl_xor(sum, 16);
      sum += __shfl_xor(sum, 32);
#pragma unroll
      for (int jt = 0; jt < NT_MAX; ++jt)
        if (jt < NT) {
#pragma unroll
          for (int r = 0; r < 4; ++r) st[jt][r] = st[jt][r] / sum;
        }
    }
    // ---- O = P V : A = P (registers), B = V[key = 16jt + 4*lg + s][d = 16dt + l15] ----
    f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jt = 0; jt < NT_MAX; ++jt) {
      if (jt < NT) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int key = jt * 16 + 4 * lg + s;
          const float v0 = Vs[key * LDW + l15], v1 = Vs[key * LDW + 16 + l15];
          o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(st[jt][s], v0, o0, 0, 0, 0);
          o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(st[jt][s], v1, o1, 0, 0, 0);
        }
      }
    }
    // ---- store: o[r] is query qt*16 + 4*lg + r, dim l15 (o0) / 16 + l15 (o1) ----
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = qt * 16 + 4 * lg + r;
      if (i < N) {
        int64_t off;
        if (MODE == SDF_ATTN_ANN) {
          off = ((int64_t)b * N + i) * C + g * HD;                           // (attn@v).transpose(1,2).reshape(B_,N,C)
        } else {
          const int t = i / d.N1, n1 = i - t * d.N1;                         // (B_,nH,T',N1,hd) -> (T',B_,N1,C)
          off = (((int64_t)t * d.B_ + b) * d.N1 + n1) * C + g * HD;
        }
        d.out[off + l15] = o0[r];
        d.out[off + 16 + l15] = o1[r];
      }
    }
  }
}

}  // namespace

extern "C" int sdf_win_attn_fwd(const SdfWinAttnDesc* d, void* stream) {
  if (!d) return SDF_E_NULL;
  if (!d->q || !d->k || !d->v || !d->out || !d->scale || !d->bias) return SDF_E_NULL;
  if (d->mode != SDF_ATTN_ANN && d->mode != SDF_ATTN_SEW) return SDF_E_DTYPE;
  if (d->hd != HD || d->B_ < 1 || d->nH < 1 || d->N < 1 || d->N > 16 * NT_MAX) return SDF_E_SHAPE;
  if (d->mask && (d->nW < 1 || d->B_ % d->nW)) return SDF_E_SHAPE;
  if (d->mode == SDF_ATTN_SEW && (d->Tq < 1 || d->N1 < 1 || d->Tq * d->N1 != d->N)) return SDF_E_SHAPE;
  if (!sdf_aligned(d->q, d->mode == SDF_ATTN_ANN ? 16 : 4) || !sdf_aligned(d->out, 4)) return SDF_E_ALIGN;
  AttnParams P;
  P.d = *d;
  if (!d->mask) P.d.nW = 1;
  const int NP = ((d->N + 15) / 16) * 16;
  const size_t lds = (size_t)3 * NP * LDW * sizeof(float);
  dim3 grid((unsigned)(d->B_ * d->nH)), block(256);
  hipStream_t s = sdf_stream(stream);
  static bool lds_opt_in = false;       // > 64 KiB of dynamic LDS needs a one-time opt-in (read-only afterwards)
  if (!lds_opt_in) {
    hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(win_attn_kernel<SDF_ATTN_ANN>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 16 * NT_MAX * LDW * 4);
    hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(win_attn_kernel<SDF_ATTN_SEW>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 16 * NT_MAX * LDW * 4);
    if (e1 != hipSuccess) return (int)e1;
    if (e2 != hipSuccess) return (int)e2;
    lds_opt_in = true;
  }
  if (d->mode == SDF_ATTN_ANN) {
    hipLaunchKernelGGL(win_attn_kernel<SDF_ATTN_ANN>, grid, block, lds, s, P);
  } else {
    hipLaunchKernelGGL(win_attn_kernel<SDF_ATTN_SEW>, grid, block, lds, s, P);
  }
  SDF_LAUNCH_CHECK();
  return 0;
}
