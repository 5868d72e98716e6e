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
// Two kernels.  win_attn_kernel (below) is the general one (any N <= 192): q/k/v (N tokens x 32 dims, fp32) staged
// once in LDS (k and q L2-normalised on the way in for the ANN mode).  Each wave owns 16-query tiles and keeps
// the whole 16 x N score strip in registers: S^T = K Q^T is computed with v_mfma_f32_16x16x4_f32 (exact fp32,
// k-ordered fmaf chain) so that the accumulator of key tile jt *is* the A operand of the P.V product
// (keys permuted inside each 4-deep MFMA step: step s of tile jt covers keys 16jt + 4*(lane>>4) + s), i.e. no
// LDS round trip and no shuffle between the two matrix products.  Row max / sum reduce in-lane over the 4*NT
// registers and across the 4 lane groups with two DPP-free shuffles.  Bias and mask are read as float4
// (4 consecutive keys) from L2-resident tables.
#include "common.h"
#include "switches.h"

namespace {

constexpr int HD = 32;
constexpr int NT_MAX = 12;            // up to 192 tokens per window
constexpr int LDW = HD + 4;           // padded LDS row (floats)

typedef __attribute__((ext_vector_type(4))) float f32x4;

struct AttnParams {
  SdfWinAttnDesc d;
};

// ANN mode: first float of the q | k | v row of token r of window b.  With a row map the window partition (pad, roll, reshape)
// is this lookup; a padding token reads the qkv Linear's image of a zero row.
__device__ __forceinline__ const float* qkv_row(const SdfWinAttnDesc& d, int b, int r, int C) {
  const float* q = reinterpret_cast<const float*>(d.q);
  if (!d.row_map) return q + ((int64_t)b * d.N + r) * 3 * C;
  const int row = d.row_map[(int64_t)b * d.N + r];
  return row >= 0 ? q + (int64_t)row * 3 * C : d.pad_qkv;
}

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
        const float* base = qkv_row(d, b, r, C) + g * HD;
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
          const int64_t orow = d.row_map ? d.row_map[(int64_t)b * N + i] : (int64_t)b * N + i;
          if (orow < 0) continue;                                            // padding token: cropped away by the reference
          off = orow * C + g * HD;                                           // (attn@v).transpose(1,2).reshape(B_,N,C) [+ window reverse]
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

// ---------------------------------------------------------------------------------------------------------
// win_attn_tiled_kernel: the same arithmetic for even N with a compile-time tile count NTC = ceil(N / 16), built so
// that the MFMA pipe is the limiter rather than load latency:
//   * only K (row-major) and V (TRANSPOSED, Vt[d][key]) live in LDS - 48 KB for N = 162, three workgroups per CU;
//     the P.V B operand is then one ds_read_b128 per 4 MFMAs instead of one ds_read_b32 per MFMA;
//   * the Q fragment of a 16-query tile (8 floats per lane) comes straight from global memory, one tile ahead;
//   * bias and mask of the whole 16 x N strip are requested with raw-buffer 8-byte loads (out-of-range offset ->
//     zeros, no branches) BEFORE the K Q^T MFMAs, so their L2 latency is covered by those MFMAs;
//   * no control flow inside the strip (NTC is a template parameter): the compiler software-pipelines LDS reads;
//   * softmax normalisation multiplies by one reciprocal per row.
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
constexpr uint32_t INV_OFF = 0x80000000u;

#ifndef SDF_ATTN_WPE
#define SDF_ATTN_WPE 3
#endif
template <int MODE, int NTC, bool HAS_MASK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SDF_ATTN_WPE, 4))) void win_attn_tiled_kernel(AttnParams P) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const SdfWinAttnDesc& d = P.d;
  constexpr int NP = NTC * 16, LDT = NP + 4;
  const int N = d.N;
  float* Ks = lds;                     // [NP][LDW]
  float* Vt = Ks + NP * LDW;           // [HD][LDT]

  // Workgroup -> (window b, head g).  Consecutive workgroup ids go round-robin over the 8 XCDs (one L2 each), so the
  // id is first turned into an XCD-contiguous sequence number L; L then walks mask-window-major: the B_/nW batch
  // copies x nH heads that add the same (N x N) mask are neighbours on one XCD and re-read it from that L2.
  const int total = gridDim.x, per_xcd = total >> 3;
  int L = blockIdx.x;
  if (L < per_xcd * 8) L = (L & 7) * per_xcd + (L >> 3);
  const int share = (d.B_ / d.nW) * d.nH;                   // workgroups per mask window
  const int w = L / share, rest = L - w * share;
  const int bc = rest / d.nH, g = rest - bc * d.nH;
  const int b = bc * d.nW + w;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int C = d.nH * HD;

  // Q fragment of query tile qt for this lane: Q[qt*16 + l15][8 lg .. 8 lg + 7] (normalised / scaled)
  auto load_q = [&](int qt, float (&qr)[8]) __attribute__((always_inline)) {
    const int qi = qt * 16 + l15;
#pragma unroll
    for (int i = 0; i < 8; ++i) qr[i] = 0.f;
    if (MODE == SDF_ATTN_ANN) {
      if (qi < N) {
        const float* base = qkv_row(d, b, qi, C) + g * HD + 8 * lg;
        const float4 a = *reinterpret_cast<const float4*>(base), c = *reinterpret_cast<const float4*>(base + 4);
        qr[0] = a.x; qr[1] = a.y; qr[2] = a.z; qr[3] = a.w; qr[4] = c.x; qr[5] = c.y; qr[6] = c.z; qr[7] = c.w;
      }
    } else {
      if (qi < N) {
        const uint8_t* qp = reinterpret_cast<const uint8_t*>(d.q) + (((int64_t)b * d.nH + g) * N + qi) * HD + 8 * lg;
        const uint32_t w0 = *reinterpret_cast<const uint32_t*>(qp), w1 = *reinterpret_cast<const uint32_t*>(qp + 4);
        const float sc = d.scale[g];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          qr[i] = ((w0 >> (8 * i)) & 0xff) ? sc : 0.f;
          qr[4 + i] = ((w1 >> (8 * i)) & 0xff) ? sc : 0.f;
        }
      }
    }
  };
  auto norm_q = [&](float (&qr)[8]) __attribute__((always_inline)) {
    if (MODE == SDF_ATTN_ANN) {                          // F.normalize(q, dim=-1)
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) ss += qr[i] * qr[i];
      ss += __shfl_xor(ss, 16);
      ss += __shfl_xor(ss, 32);
      const float iq = 1.f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
      for (int i = 0; i < 8; ++i) qr[i] *= iq;
    }
  };

  // 11 query tiles over 4 waves leaves one wave a tile short; rotating the start by the workgroup number moves that
  // light wave from SIMD to SIMD across the co-resident workgroups
  const int wv = (wave + L) & 3;
  float qnext[8];
  load_q(wv, qnext);                                   // in flight while K / V are staged

  // this window's mask is shared by B_/nW x nH workgroups but cold in L2 for the first of them: touch every 128-byte
  // line of it now, so that the miss latency is paid together with the K / V loads below, not in front of a softmax
  float warm = 0.f;
  if (HAS_MASK) {
    const float* mw = d.mask + (int64_t)(b % d.nW) * N * N;
    for (int i = tid * 32; i < N * N; i += 256 * 32) warm += mw[i];
  }
  // ---- stage K rows and V^T: 8 lanes per token row (one float4 each) - whole 128-byte lines per request ----
  {
    constexpr int IT = (NP * 8 + 255) / 256;
    const int piece = tid & 7;
    float4 kq[IT], vq[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int r = (tid >> 3) + 32 * it;
      kq[it] = vq[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < N) {
        if (MODE == SDF_ATTN_ANN) {
          const float* base = qkv_row(d, b, r, C) + g * HD + 4 * piece;
          kq[it] = *reinterpret_cast<const float4*>(base + C);
          vq[it] = *reinterpret_cast<const float4*>(base + 2 * C);
        } else {
          const int64_t off = (((int64_t)b * d.nH + g) * N + r) * HD + 4 * piece;
          const uint32_t wk = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(d.k) + off);
          const uint32_t wv = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(d.v) + off);
          kq[it] = make_float4((float)(wk & 0xff), (float)((wk >> 8) & 0xff), (float)((wk >> 16) & 0xff), (float)(wk >> 24));
          vq[it] = make_float4((float)(wv & 0xff), (float)((wv >> 8) & 0xff), (float)((wv >> 16) & 0xff), (float)(wv >> 24));
        }
      }
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int r = (tid >> 3) + 32 * it;
      float4 kv = kq[it];
      if (MODE == SDF_ATTN_ANN) {                        // F.normalize(k, dim=-1): the row's 8 lanes are neighbours
        float sk = kv.x * kv.x + kv.y * kv.y + kv.z * kv.z + kv.w * kv.w;
        sk += __shfl_xor(sk, 1);
        sk += __shfl_xor(sk, 2);
        sk += __shfl_xor(sk, 4);
        const float ik = 1.f / fmaxf(sqrtf(sk), 1e-12f);
        kv.x *= ik; kv.y *= ik; kv.z *= ik; kv.w *= ik;
      }
      if (r < NP) {
        *reinterpret_cast<float4*>(&Ks[r * LDW + 4 * piece]) = kv;
        Vt[(4 * piece + 0) * LDT + r] = vq[it].x;
        Vt[(4 * piece + 1) * LDT + r] = vq[it].y;
        Vt[(4 * piece + 2) * LDT + r] = vq[it].z;
        Vt[(4 * piece + 3) * LDT + r] = vq[it].w;
      }
    }
  }
  asm volatile("" ::"v"(warm));                                  // the touches only have to have been issued
  __syncthreads();

  const float ls = (MODE == SDF_ATTN_ANN) ? d.scale[g] : 1.f;
  const uint32_t tbytes = (uint32_t)N * (uint32_t)N * 4u;
  const __amdgpu_buffer_rsrc_t bias_rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.bias + (int64_t)g * N * N), 0, (int)tbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t mask_rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(HAS_MASK ? d.mask + (int64_t)(b % d.nW) * N * N : d.bias), 0, HAS_MASK ? (int)tbytes : 0, 0x00020000);
  const float NEG = (MODE == SDF_ATTN_ANN) ? -INFINITY : 0.f;

  for (int qt = wv; qt < NTC; qt += 4) {
    if (qt * 16 >= N) break;
    const int qi = qt * 16 + l15;
    float qreg[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) qreg[i] = qnext[i];
    norm_q(qreg);
    if (qt + 4 < NTC) load_q(qt + 4, qnext);
    // ---- request bias / mask of the strip: lane needs [qi][16 jt + 4 lg .. + 3]; whole tiles with one 16-byte load
    //      (rows are only 8-byte aligned - fine for buffer loads), the ragged last tile with two 8-byte loads ----
    u32x4 bb[NTC], mm[NTC];
    const uint32_t rowoff = (uint32_t)qi * (uint32_t)N * 4u;
    auto load_strip = [&](const __amdgpu_buffer_rsrc_t& rs, u32x4 (&dst)[NTC]) __attribute__((always_inline)) {
#pragma unroll
      for (int jt = 0; jt < NTC; ++jt) {
        const int kb = jt * 16 + 4 * lg;
        if (jt < NTC - 1) {                                        // 16 (NTC-1) < N: every tile but the last is whole
          dst[jt] = __builtin_amdgcn_raw_buffer_load_b128(rs, qi < N ? rowoff + (uint32_t)kb * 4u : INV_OFF, 0, 0);
        } else {
          const uint32_t o0 = (qi < N && kb < N) ? rowoff + (uint32_t)kb * 4u : INV_OFF;
          const uint32_t o1 = (qi < N && kb + 2 < N) ? rowoff + (uint32_t)kb * 4u + 8u : INV_OFF;
          const u32x2 lo = __builtin_amdgcn_raw_buffer_load_b64(rs, o0, 0, 0);
          const u32x2 hi = __builtin_amdgcn_raw_buffer_load_b64(rs, o1, 0, 0);
          dst[jt] = u32x4{lo.x, lo.y, hi.x, hi.y};
        }
      }
    };
#ifdef SDF_ATTN_NOSTRIP
    // timing experiment only (tools/attn_strip_cost.sh): what do the bias / mask strip loads from L2 cost?  (wrong results)
#pragma unroll
    for (int jt = 0; jt < NTC; ++jt) { bb[jt] = u32x4{0u, 0u, 0u, 0u}; mm[jt] = u32x4{0u, 0u, 0u, 0u}; }
#else
    load_strip(bias_rs, bb);
    if (HAS_MASK) load_strip(mask_rs, mm);
#endif
    // ---- S^T = K Q^T ----
    f32x4 st[NTC];
#pragma unroll
    for (int jt = 0; jt < NTC; ++jt) {
      const int kj = jt * 16 + l15;
      const float4 a = *reinterpret_cast<const float4*>(&Ks[kj * LDW + 8 * lg]);
      const float4 c = *reinterpret_cast<const float4*>(&Ks[kj * LDW + 8 * lg + 4]);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, qreg[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, qreg[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, qreg[2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, qreg[3], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(c.x, qreg[4], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(c.y, qreg[5], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(c.z, qreg[6], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(c.w, qreg[7], acc, 0, 0, 0);
      st[jt] = acc;
    }
    // ---- attn = qk * logit_scale + bias (+ mask), separately rounded as the reference computes it ----
#pragma unroll
    for (int jt = 0; jt < NTC; ++jt) {
      const int kb = jt * 16 + 4 * lg;
      const uint32_t b0 = bb[jt].x, b1 = bb[jt].y, b2 = bb[jt].z, b3 = bb[jt].w;
      float s0 = st[jt][0] * ls + __uint_as_float(b0), s1 = st[jt][1] * ls + __uint_as_float(b1);
      float s2 = st[jt][2] * ls + __uint_as_float(b2), s3 = st[jt][3] * ls + __uint_as_float(b3);
      if (HAS_MASK) {
        const uint32_t m0 = mm[jt].x, m1 = mm[jt].y, m2 = mm[jt].z, m3 = mm[jt].w;
        s0 += __uint_as_float(m0); s1 += __uint_as_float(m1); s2 += __uint_as_float(m2); s3 += __uint_as_float(m3);
      }
      if (jt == NTC - 1) {                                         // only the last tile can be ragged
        s0 = (kb + 0 < N) ? s0 : NEG; s1 = (kb + 1 < N) ? s1 : NEG;
        s2 = (kb + 2 < N) ? s2 : NEG; s3 = (kb + 3 < N) ? s3 : NEG;
      }
      st[jt][0] = s0; st[jt][1] = s1; st[jt][2] = s2; st[jt][3] = s3;
    }
    if (MODE == SDF_ATTN_ANN) {
      float m = -INFINITY;
#pragma unroll
      for (int jt = 0; jt < NTC; ++jt) m = fmaxf(fmaxf(fmaxf(m, st[jt][0]), fmaxf(st[jt][1], st[jt][2])), st[jt][3]);
      m = fmaxf(m, __shfl_xor(m, 16));
      m = fmaxf(m, __shfl_xor(m, 32));
      // exp(x - m) = 2^(x log2e - m log2e): one fma + the hardware exp2 (1 ulp) instead of the ~14-instruction expf;
      // the argument rounding adds <= |x - m| * 2^-24 relative error - far inside the 1e-3 flow tolerance
      const float L2E = 1.4426950408889634f, mneg = -m * L2E;
      float sum = 0.f;
#pragma unroll
      for (int jt = 0; jt < NTC; ++jt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __builtin_amdgcn_exp2f(fmaf(st[jt][r], L2E, mneg));
          st[jt][r] = e;
          sum += e;
        }
      }
      sum += __shfl_xor(sum, 16);
      sum += __shfl_xor(sum, 32);
      const float inv = 1.f / sum;
#pragma unroll
      for (int jt = 0; jt < NTC; ++jt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) st[jt][r] *= inv;
      }
    }
    // ---- O = P V : A = P (registers), B = Vt[d = 16 dt + l15][key = 16 jt + 4 lg + s] ----
    f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jt = 0; jt < NTC; ++jt) {
      const float4 v0 = *reinterpret_cast<const float4*>(&Vt[l15 * LDT + jt * 16 + 4 * lg]);
      const float4 v1 = *reinterpret_cast<const float4*>(&Vt[(16 + l15) * LDT + jt * 16 + 4 * lg]);
      o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(st[jt][0], v0.x, o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(st[jt][0], v1.x, o1, 0, 0, 0);
      o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(st[jt][1], v0.y, o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(st[jt][1], v1.y, o1, 0, 0, 0);
      o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(st[jt][2], v0.z, o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(st[jt][2], v1.z, o1, 0, 0, 0);
      o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(st[jt][3], v0.w, o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(st[jt][3], v1.w, o1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = qt * 16 + 4 * lg + r;
      if (i < N) {
        int64_t off;
        if (MODE == SDF_ATTN_ANN) {
          const int64_t orow = d.row_map ? d.row_map[(int64_t)b * N + i] : (int64_t)b * N + i;   // window reverse + roll back + crop
          if (orow < 0) continue;
          off = orow * C + g * HD;
        } else {
          const int t = i / d.N1, n1 = i - t * d.N1;
          off = (((int64_t)t * d.B_ + b) * d.N1 + n1) * C + g * HD;
        }
        d.out[off + l15] = o0[r];
        d.out[off + 16 + l15] = o1[r];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// win_attn_tiled_f16_kernel: the ANN arithmetic of win_attn_tiled_kernel on the 16-bit matrix pipe.  Every fp32 operand is
// the sum of two fp16 numbers (x = hi + lo, hi = fp16(x), lo = fp16(x - hi): 22 significant bits) and every product is three
// MFMAs (hi*hi + lo*hi + hi*lo, fp32 accumulation): K Q^T is 3 x v_mfma_f32_16x16x32_f16 per 16 x 16 score tile (the whole
// head dimension in one instruction) instead of 8 x v_mfma_f32_16x16x4_f32, P V is 2 x 3 x v_mfma_f32_16x16x16_f16 instead
// of 8 - 96 matrix-pipe cycles per tile pair instead of 512.  The register choreography is unchanged: S^T = K Q^T leaves
// keys 4*(lane/16) + 0..3 of query lane%16 in a lane's accumulator, which is exactly the A operand layout of the 16-key P V
// step (split into hi / lo on the spot), so the probabilities never touch LDS.  K rows live in LDS as two fp16 planes
// (80-byte rows: conflict-free ds_read_b128 over 16 consecutive keys), V transposed as [key quad][dim]{4 x hi, 4 x lo}
// (one ds_read_b128 per P V B operand, 512-byte key-quad pitch: conflict-free).
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
constexpr int KRS = 64;                // K row pitch in bytes: 32 fp16, no padding - the 16-byte piece p of row r sits at piece p ^ kswz(r)
// ds_read_b128 is served in four NON-contiguous 16-lane groups ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS): for (row = lane % 16,
// piece = lane / 16) fragment reads no padded pitch of a 4-piece row is conflict-free below 6 pieces; this XOR is at 4 (brute force), and
// the image shrinks from 80 to 64 bytes per row (round 2's 80-byte pitch cost 33 % extra LDS cycles, profiles/r3i_*)
__device__ __forceinline__ int kswz(int row) { return ((row >> 2) & 1) * 2; }

// two fp32 -> their hi and lo fp16 halves, packed (round to nearest; beyond the fp16 range: inf / NaN, visible in the output):
// v_cvt_pk_f16_f32, two conversions back, one packed subtraction (exact), v_cvt_pk_f16_f32.
// (Round 5 tried lo = v_fma_mixlo/hi_f16(hi, -1.0, x) - three instructions instead of five - as inline assembly: the results were
// corrupted, because the compiler's hazard recogniser does not see an assembly statement's register write as a vector write, so
// nothing kept it the required wait states behind an MFMA still reading that register as its accumulator input; hipcc does not
// select the mix instructions from C.)
__device__ __forceinline__ void split2_f16(float x, float y, uint32_t& hi, uint32_t& lo) {
  const f32x2 v = {x, y};
  const f16x2 h = __builtin_convertvector(v, f16x2);
  const f32x2 r = v - __builtin_convertvector(h, f32x2);
  const f16x2 l = __builtin_convertvector(r, f16x2);
  hi = __builtin_bit_cast(uint32_t, h);
  lo = __builtin_bit_cast(uint32_t, l);
}
// four fp32 -> four hi and four lo fp16
__device__ __forceinline__ void split4_f16(float x, float y, float z, float w, uint2& hi, uint2& lo) {
  split2_f16(x, y, hi.x, lo.x);
  split2_f16(z, w, hi.y, lo.y);
}

template <int MODE, int NTC, bool HAS_MASK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SDF_ATTN_WPE, 4))) void win_attn_tiled_f16_kernel(AttnParams P) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const SdfWinAttnDesc& d = P.d;
  constexpr int NP = NTC * 16;
  const int N = d.N;
  uint8_t* Khi = reinterpret_cast<uint8_t*>(lds);                  // [NP][KRS]
  uint8_t* Klo = Khi + NP * KRS;
  uint8_t* Vt2 = Klo + NP * KRS;                                   // [NP / 4][HD][16 B]

  const int total = gridDim.x, per_xcd = total >> 3;               // workgroup -> (window, head): as win_attn_tiled_kernel
  int L = blockIdx.x;
  if (L < per_xcd * 8) L = (L & 7) * per_xcd + (L >> 3);
  const int share = (d.B_ / d.nW) * d.nH;
  const int w = L / share, rest = L - w * share;
  const int bc = rest / d.nH, g = rest - bc * d.nH;
  const int b = bc * d.nW + w;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int C = d.nH * HD;

  auto load_q = [&](int qt, float (&qr)[8]) __attribute__((always_inline)) {
    const int qi = qt * 16 + l15;
#pragma unroll
    for (int i = 0; i < 8; ++i) qr[i] = 0.f;
    if (qi < N) {
      if (MODE == SDF_ATTN_ANN) {
        const float* base = qkv_row(d, b, qi, C) + g * HD + 8 * lg;
        const float4 a = *reinterpret_cast<const float4*>(base), c = *reinterpret_cast<const float4*>(base + 4);
        qr[0] = a.x; qr[1] = a.y; qr[2] = a.z; qr[3] = a.w; qr[4] = c.x; qr[5] = c.y; qr[6] = c.z; qr[7] = c.w;
      } else {                                                     // binary q: 0 / 1 (the scale multiplies the exact count afterwards)
        const uint8_t* qp = reinterpret_cast<const uint8_t*>(d.q) + (((int64_t)b * d.nH + g) * N + qi) * HD + 8 * lg;
        const uint32_t w0 = *reinterpret_cast<const uint32_t*>(qp), w1 = *reinterpret_cast<const uint32_t*>(qp + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          qr[i] = ((w0 >> (8 * i)) & 0xff) ? 1.f : 0.f;
          qr[4 + i] = ((w1 >> (8 * i)) & 0xff) ? 1.f : 0.f;
        }
      }
    }
  };
  const int wv = (wave + L) & 3;
  float qnext[8];
  load_q(wv, qnext);

  float warm = 0.f;
  if (HAS_MASK) {
    const float* mw = d.mask + (int64_t)(b % d.nW) * N * N;
    for (int i = tid * 32; i < N * N; i += 256 * 32) warm += mw[i];
  }
  // ---- stage K (normalised, hi / lo planes) and V (transposed by key quads): a unit = (key quad, 4 dims), 8 units per quad ----
  {
    constexpr int UNITS = (NP / 4) * 8, IT = (UNITS + 255) / 256;
    float4 kq[IT][4], vq[IT][4];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int u = tid + 256 * it, kgp = u >> 3, piece = u & 7;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = 4 * kgp + j;
        kq[it][j] = vq[it][j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (u < UNITS && r < N) {
          if (MODE == SDF_ATTN_ANN) {
            const float* base = qkv_row(d, b, r, C) + g * HD + 4 * piece;
            kq[it][j] = *reinterpret_cast<const float4*>(base + C);
            vq[it][j] = *reinterpret_cast<const float4*>(base + 2 * C);
          } else {
            const int64_t off = (((int64_t)b * d.nH + g) * N + r) * HD + 4 * piece;
            const uint32_t wk = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(d.k) + off);
            const uint32_t wv = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(d.v) + off);
            kq[it][j] = make_float4((float)(wk & 0xff), (float)((wk >> 8) & 0xff), (float)((wk >> 16) & 0xff), (float)(wk >> 24));
            vq[it][j] = make_float4((float)(wv & 0xff), (float)((wv >> 8) & 0xff), (float)((wv >> 16) & 0xff), (float)(wv >> 24));
          }
        }
      }
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int u = tid + 256 * it, kgp = u >> 3, piece = u & 7;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float4 kv = kq[it][j];
        float ik = 1.f;
        if (MODE == SDF_ATTN_ANN) {                                // F.normalize(k, dim=-1): the key's 8 units are neighbouring lanes
          float sk = kv.x * kv.x + kv.y * kv.y + kv.z * kv.z + kv.w * kv.w;
          sk += __shfl_xor(sk, 1);
          sk += __shfl_xor(sk, 2);
          sk += __shfl_xor(sk, 4);
          ik = 1.f / fmaxf(sqrtf(sk), 1e-12f);
        }
        uint2 hi, lo;
        split4_f16(kv.x * ik, kv.y * ik, kv.z * ik, kv.w * ik, hi, lo);
        if (u < UNITS) {
          const int kr = 4 * kgp + j, ko = kr * KRS + 16 * ((piece >> 1) ^ kswz(kr)) + 8 * (piece & 1);
          *reinterpret_cast<uint2*>(Khi + ko) = hi;
          *reinterpret_cast<uint2*>(Klo + ko) = lo;
        }
      }
      if (u < UNITS) {
        const float vx[4][4] = {{vq[it][0].x, vq[it][1].x, vq[it][2].x, vq[it][3].x}, {vq[it][0].y, vq[it][1].y, vq[it][2].y, vq[it][3].y},
                                {vq[it][0].z, vq[it][1].z, vq[it][2].z, vq[it][3].z}, {vq[it][0].w, vq[it][1].w, vq[it][2].w, vq[it][3].w}};
#pragma unroll
        for (int c = 0; c < 4; ++c) {                              // dim 4*piece + c: its four keys, hi then lo
          uint2 hi, lo;
          split4_f16(vx[c][0], vx[c][1], vx[c][2], vx[c][3], hi, lo);
          *reinterpret_cast<uint4*>(Vt2 + ((kgp * HD) + 4 * piece + c) * 16) = make_uint4(hi.x, hi.y, lo.x, lo.y);
        }
      }
    }
  }
  asm volatile("" ::"v"(warm));
  __syncthreads();

  const float ls = d.scale[g];                          // ANN: logit scale on cosines; SEW: q's scale on the exact spike count
  const uint32_t tbytes = (uint32_t)N * (uint32_t)N * 4u;
  const __amdgpu_buffer_rsrc_t bias_rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.bias + (int64_t)g * N * N), 0, (int)tbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t mask_rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(HAS_MASK ? d.mask + (int64_t)(b % d.nW) * N * N : d.bias), 0, HAS_MASK ? (int)tbytes : 0, 0x00020000);

  const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(d.out, 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t map_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(d.row_map ? d.row_map : reinterpret_cast<const int*>(d.out)), 0,
                                                                          d.row_map ? 0x7FFFFFFF : 0, 0x00020000);
  for (int qt = wv; qt < NTC; qt += 4) {
    if (qt * 16 >= N) break;
    const int qi = qt * 16 + l15;
    float qreg[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) qreg[i] = qnext[i];
    if (MODE == SDF_ATTN_ANN) {
      // F.normalize(q, dim=-1), with the head's logit scale and log2(e) folded into the same multiplier: the scores leave the
      // matrix pipe in the log2 domain of the softmax (K Q^T * ls * log2 e), 8 multiplications per lane instead of 4 NTC
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) ss += qreg[i] * qreg[i];
      ss += __shfl_xor(ss, 16);
      ss += __shfl_xor(ss, 32);
      const float iq = (ls * 1.4426950408889634f) / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
      for (int i = 0; i < 8; ++i) qreg[i] *= iq;
    }
    if (qt + 4 < NTC) load_q(qt + 4, qnext);
    uint2 qh0, ql0, qh1, ql1;
    split4_f16(qreg[0], qreg[1], qreg[2], qreg[3], qh0, ql0);
    split4_f16(qreg[4], qreg[5], qreg[6], qreg[7], qh1, ql1);
    const f16x8 q_hi = __builtin_bit_cast(f16x8, make_uint4(qh0.x, qh0.y, qh1.x, qh1.y));
    const f16x8 q_lo = __builtin_bit_cast(f16x8, make_uint4(ql0.x, ql0.y, ql1.x, ql1.y));

    u32x4 bb[NTC], mm[NTC];
    // one lane offset per strip (its query's row + the lane group's four keys; out of range as a whole for a padding query), the
    // key tile is the instruction's scalar offset: no address arithmetic per load
    const uint32_t rowoff = (uint32_t)qi * (uint32_t)N * 4u;
    const uint32_t rbase = qi < N ? rowoff + 16u * (uint32_t)lg : INV_OFF;
    auto load_strip = [&](const __amdgpu_buffer_rsrc_t& rs, u32x4 (&dst)[NTC]) __attribute__((always_inline)) {
#pragma unroll
      for (int jt = 0; jt < NTC; ++jt) {
        const int kb = jt * 16 + 4 * lg;
        if (jt < NTC - 1) {
          dst[jt] = __builtin_amdgcn_raw_buffer_load_b128(rs, rbase, jt * 64, 0);
        } else {
          const uint32_t o0 = (qi < N && kb < N) ? rowoff + (uint32_t)kb * 4u : INV_OFF;
          const uint32_t o1 = (qi < N && kb + 2 < N) ? rowoff + (uint32_t)kb * 4u + 8u : INV_OFF;
          const u32x2 lo = __builtin_amdgcn_raw_buffer_load_b64(rs, o0, 0, 0);
          const u32x2 hi = __builtin_amdgcn_raw_buffer_load_b64(rs, o1, 0, 0);
          dst[jt] = u32x4{lo.x, lo.y, hi.x, hi.y};
        }
      }
    };
    load_strip(bias_rs, bb);
    if (HAS_MASK) load_strip(mask_rs, mm);
    // ---- S^T = K Q^T: three 16x16x32 products per key tile ----
    // ANN: the accumulator STARTS as (bias + mask) * log2 e - the additions of the score come out of the matrix pipe for free and
    // the strip is in the softmax's log2 domain when it lands; keys beyond N start at -inf (their K rows are zero: -inf + 0)
    f32x4 st[NTC];
    constexpr float L2E = 1.4426950408889634f;
#pragma unroll
    for (int jt = 0; jt < NTC; ++jt) {
      const int kj = jt * 16 + l15;
      const f16x8 k_hi = *reinterpret_cast<const f16x8*>(Khi + kj * KRS + 16 * (lg ^ kswz(l15)));
      const f16x8 k_lo = *reinterpret_cast<const f16x8*>(Klo + kj * KRS + 16 * (lg ^ kswz(l15)));
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if (MODE == SDF_ATTN_ANN) {                                  // binary operands have no lo halves: one exact product
        f32x2 b01 = {__uint_as_float(bb[jt].x), __uint_as_float(bb[jt].y)}, b23 = {__uint_as_float(bb[jt].z), __uint_as_float(bb[jt].w)};
        if (HAS_MASK) {
          const f32x2 m01 = {__uint_as_float(mm[jt].x), __uint_as_float(mm[jt].y)}, m23 = {__uint_as_float(mm[jt].z), __uint_as_float(mm[jt].w)};
          b01 += m01; b23 += m23;
        }
        const f32x2 l2 = {L2E, L2E};
        b01 *= l2; b23 *= l2;
        if (jt == NTC - 1) {
          const int kb = jt * 16 + 4 * lg;
          b01.x = (kb + 0 < N) ? b01.x : -INFINITY; b01.y = (kb + 1 < N) ? b01.y : -INFINITY;
          b23.x = (kb + 2 < N) ? b23.x : -INFINITY; b23.y = (kb + 3 < N) ? b23.y : -INFINITY;
        }
        acc = f32x4{b01.x, b01.y, b23.x, b23.y};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(k_hi, q_lo, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(k_lo, q_hi, acc, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(k_hi, q_hi, acc, 0, 0, 0);
      st[jt] = acc;
    }
    float rinv[4] = {1.f, 1.f, 1.f, 1.f};                          // ANN: 1 / (row sum) of this lane's four OUTPUT rows (queries 4 lg + r)
    if (MODE == SDF_ATTN_ANN) {
      float m = st[0][0];
#pragma unroll
      for (int jt = 0; jt < NTC; ++jt) {
        m = fmaxf(fmaxf(m, st[jt][0]), st[jt][1]);
        m = fmaxf(fmaxf(m, st[jt][2]), st[jt][3]);
      }
      m = fmaxf(m, __shfl_xor(m, 16));
      m = fmaxf(m, __shfl_xor(m, 32));
      const f32x2 m2 = {m, m};
      f32x2 sum2 = {0.f, 0.f};
#pragma unroll
      for (int jt = 0; jt < NTC; ++jt) {
        const f32x2 a01 = f32x2{st[jt][0], st[jt][1]} - m2, a23 = f32x2{st[jt][2], st[jt][3]} - m2;
        const f32x2 e01 = {__builtin_amdgcn_exp2f(a01.x), __builtin_amdgcn_exp2f(a01.y)};
        const f32x2 e23 = {__builtin_amdgcn_exp2f(a23.x), __builtin_amdgcn_exp2f(a23.y)};
        sum2 += e01; sum2 += e23;
        st[jt] = f32x4{e01.x, e01.y, e23.x, e23.y};
      }
      float sum = sum2.x + sum2.y;
      sum += __shfl_xor(sum, 16);
      sum += __shfl_xor(sum, 32);
      // the probabilities go to the matrix pipe UNNORMALISED (e in (0, 1], the row maximum is 1: hi + lo carry 22 bits of every
      // term that matters) and the finished rows are divided instead: 8 multiplications per lane for 4 NTC.  The output rows of a
      // lane are queries 4 lg + r, the sums sit with lanes l15 = query: one cross-lane read each
      const float inv = 1.f / sum;
#pragma unroll
      for (int r = 0; r < 4; ++r) rinv[r] = __shfl(inv, (lane & 48) | (4 * lg + r));
    } else {
#pragma unroll
      for (int jt = 0; jt < NTC; ++jt) {
        const int kb = jt * 16 + 4 * lg;
        const uint32_t b0 = bb[jt].x, b1 = bb[jt].y, b2 = bb[jt].z, b3 = bb[jt].w;
        float s0 = st[jt][0] * ls + __uint_as_float(b0), s1 = st[jt][1] * ls + __uint_as_float(b1);
        float s2 = st[jt][2] * ls + __uint_as_float(b2), s3 = st[jt][3] * ls + __uint_as_float(b3);
        if (HAS_MASK) {
          const uint32_t m0 = mm[jt].x, m1 = mm[jt].y, m2 = mm[jt].z, m3 = mm[jt].w;
          s0 += __uint_as_float(m0); s1 += __uint_as_float(m1); s2 += __uint_as_float(m2); s3 += __uint_as_float(m3);
        }
        if (jt == NTC - 1) {
          s0 = (kb + 0 < N) ? s0 : 0.f; s1 = (kb + 1 < N) ? s1 : 0.f;
          s2 = (kb + 2 < N) ? s2 : 0.f; s3 = (kb + 3 < N) ? s3 : 0.f;
        }
        st[jt][0] = s0; st[jt][1] = s1; st[jt][2] = s2; st[jt][3] = s3;
      }
    }
    // ---- O = P V: the lane's four probabilities of key tile jt are the A operand (keys 16 jt + 4 lg + 0..3) ----
    f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jt = 0; jt < NTC; ++jt) {
      uint2 ph, pl;
      split4_f16(st[jt][0], st[jt][1], st[jt][2], st[jt][3], ph, pl);
      const f16x4 p_hi = __builtin_bit_cast(f16x4, ph), p_lo = __builtin_bit_cast(f16x4, pl);
      const uint4 v0 = *reinterpret_cast<const uint4*>(Vt2 + (((4 * jt + lg) * HD) + l15) * 16);
      const uint4 v1 = *reinterpret_cast<const uint4*>(Vt2 + (((4 * jt + lg) * HD) + 16 + l15) * 16);
      const f16x4 v0h = __builtin_bit_cast(f16x4, make_uint2(v0.x, v0.y)), v0l = __builtin_bit_cast(f16x4, make_uint2(v0.z, v0.w));
      const f16x4 v1h = __builtin_bit_cast(f16x4, make_uint2(v1.x, v1.y)), v1l = __builtin_bit_cast(f16x4, make_uint2(v1.z, v1.w));
      o0 = __builtin_amdgcn_mfma_f32_16x16x16f16(p_lo, v0h, o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_16x16x16f16(p_lo, v1h, o1, 0, 0, 0);
      if (MODE == SDF_ATTN_ANN) {                                  // binary v has no lo half
        o0 = __builtin_amdgcn_mfma_f32_16x16x16f16(p_hi, v0l, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_16x16x16f16(p_hi, v1l, o1, 0, 0, 0);
      }
      o0 = __builtin_amdgcn_mfma_f32_16x16x16f16(p_hi, v0h, o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_16x16x16f16(p_hi, v1h, o1, 0, 0, 0);
    }
    // 32-bit buffer offsets (the host admits this kernel only while B_ * N * C * 4 < 2^31): no 64-bit address arithmetic
    int orow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = qt * 16 + 4 * lg + r;
      if (MODE == SDF_ATTN_ANN) {                                    // window reverse + roll back + crop: the row map (< 0: a padding token)
        orow[r] = b * N + i;
        if (d.row_map) orow[r] = (int)__builtin_amdgcn_raw_buffer_load_b32(map_rs, i < N ? (uint32_t)(b * N + i) * 4u : INV_OFF, 0, 0);
        if (i >= N) orow[r] = -1;
      } else {
        const int t = i / d.N1, n1 = i - t * d.N1;
        orow[r] = i < N ? (t * d.B_ + b) * d.N1 + n1 : -1;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t off = orow[r] >= 0 ? ((uint32_t)orow[r] * (uint32_t)C + (uint32_t)(g * HD + l15)) * 4u : INV_OFF;
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o0[r] * rinv[r]), out_rs, off, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o1[r] * rinv[r]), out_rs, off, 64, 0);
    }
  }
}

template <int MODE, int NTC>
int launch_tiled_f16(const AttnParams& P, hipStream_t s) {
  constexpr size_t lds = (size_t)(2 * NTC * 16 * KRS + (NTC * 4) * HD * 16);
  dim3 grid((unsigned)(P.d.B_ * P.d.nH)), block(256);
  if (P.d.mask) {
    SDF_LAUNCH((win_attn_tiled_f16_kernel<MODE, NTC, true>), grid, block, lds, s, P);
  } else {
    SDF_LAUNCH((win_attn_tiled_f16_kernel<MODE, NTC, false>), grid, block, lds, s, P);
  }
  SDF_LAUNCH_CHECK();
  return 0;
}

template <int MODE, int NTC>
int launch_tiled(const AttnParams& P, hipStream_t s) {
  constexpr size_t lds = (size_t)(NTC * 16 * LDW + HD * (NTC * 16 + 4)) * sizeof(float);
  dim3 grid((unsigned)(P.d.B_ * P.d.nH)), block(256);
  if (P.d.mask) {
    SDF_LAUNCH((win_attn_tiled_kernel<MODE, NTC, true>), grid, block, lds, s, P);
  } else {
    SDF_LAUNCH((win_attn_tiled_kernel<MODE, NTC, false>), grid, block, lds, s, P);
  }
  SDF_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" int sdf_win_attn_fwd(const SdfWinAttnDesc* d, void* stream) {
  if (!d) return SDF_E_NULL;
  if (!d->q || !d->k || !d->v || !d->out || !d->scale || !d->bias) return SDF_E_NULL;
  if (d->mode != SDF_ATTN_ANN && d->mode != SDF_ATTN_SEW) return SDF_E_DTYPE;
  if (d->hd != HD || d->B_ < 1 || d->nH < 1 || d->N < 1 || d->N > 16 * NT_MAX) return SDF_E_SHAPE;
  if (d->mask && (d->nW < 1 || d->B_ % d->nW)) return SDF_E_SHAPE;
  if (d->mode == SDF_ATTN_SEW && (d->Tq < 1 || d->N1 < 1 || d->Tq * d->N1 != d->N)) return SDF_E_SHAPE;
  if (d->row_map && (d->mode != SDF_ATTN_ANN || !d->pad_qkv)) return SDF_E_NULL;   // windowing through the map: ANN form, needs the pad row
  if (!sdf_aligned(d->q, d->mode == SDF_ATTN_ANN ? 16 : 4) || !sdf_aligned(d->out, 4)) return SDF_E_ALIGN;
  AttnParams P;
  P.d = *d;
  if (!d->mask) P.d.nW = 1;
  const int NP = ((d->N + 15) / 16) * 16;
  const size_t lds = (size_t)3 * NP * LDW * sizeof(float);
  dim3 grid((unsigned)(d->B_ * d->nH)), block(256);
  hipStream_t s = sdf_stream(stream);
  // even N that fits a compiled tile count: the MFMA-paced kernel (8-byte bias / mask loads need N % 2 == 0)
  const char* ge = sdf_sw(SW_ATTN_GENERIC);                 // A/B override: 1 = always the general kernel
  if (d->N % 2 == 0 && (int64_t)d->N * d->N * 4 < (1LL << 31) && !(ge && ge[0] == '1')) {
    const int nt = (d->N + 15) / 16;                         // compiled tile counts: windows (2,8,8) and (2,9,9)
    const char* f32 = sdf_sw(SW_ATTN_F32);                // A/B override: 1 = the fp32-pipe kernels
    // (the 16-bit-pipe kernels address the output and the row map with 32-bit byte offsets)
    const bool pipe16 = !(f32 && f32[0] == '1') && (int64_t)d->B_ * d->N * d->nH * HD * 4 < (1LL << 31);
    if (d->mode == SDF_ATTN_ANN) {
      if (pipe16 && nt == 8) return launch_tiled_f16<SDF_ATTN_ANN, 8>(P, s);
      if (pipe16 && nt == 11) return launch_tiled_f16<SDF_ATTN_ANN, 11>(P, s);
      if (nt == 8) return launch_tiled<SDF_ATTN_ANN, 8>(P, s);
      if (nt == 11) return launch_tiled<SDF_ATTN_ANN, 11>(P, s);
    } else {
      if (pipe16 && nt == 8) return launch_tiled_f16<SDF_ATTN_SEW, 8>(P, s);
      if (pipe16 && nt == 11) return launch_tiled_f16<SDF_ATTN_SEW, 11>(P, s);
      if (nt == 8) return launch_tiled<SDF_ATTN_SEW, 8>(P, s);
      if (nt == 11) return launch_tiled<SDF_ATTN_SEW, 11>(P, s);
    }
  }
  static std::atomic<uint64_t> opt_ann{0}, opt_sew{0};      // > 64 KiB of dynamic LDS: opt-in once per kernel and device
  if (const int e1 = sdf_lds_opt_in(opt_ann, reinterpret_cast<const void*>(win_attn_kernel<SDF_ATTN_ANN>), 3 * 16 * NT_MAX * LDW * 4)) return e1;
  if (const int e2 = sdf_lds_opt_in(opt_sew, reinterpret_cast<const void*>(win_attn_kernel<SDF_ATTN_SEW>), 3 * 16 * NT_MAX * LDW * 4)) return e2;
  if (d->mode == SDF_ATTN_ANN) {
    SDF_LAUNCH(win_attn_kernel<SDF_ATTN_ANN>, grid, block, lds, s, P);
  } else {
    SDF_LAUNCH(win_attn_kernel<SDF_ATTN_SEW>, grid, block, lds, s, P);
  }
  SDF_LAUNCH_CHECK();
  return 0;
}
