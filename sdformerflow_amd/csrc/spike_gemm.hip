// Spike GEMM for gfx950:  out[M,N] = epilogue( A[M,K] (binary u8 spikes) x W[N,K]^T ).
//
// MFMA-tiled (v_mfma_f32_32x32x16_bf16, 64-lane wavefronts).  A binary operand is exact in bf16, so the
// fp32 weights are carried as up to three bf16 planes (hi/mid/lo residual split) and the product is
// accumulated in fp32 by the matrix cores: fp32-grade results at bf16 MFMA rate (3 MFMAs per 32x32x16
// sub-product instead of 8 f32-input MFMAs).  The 1-byte spikes are expanded to bf16 in registers.
//
// Workgroup = 4 waves; tile 128 (M) x 96 (N) x 96 (K stage); wave w owns rows [32w, 32w+32) and all
// three 32-column sub-tiles (3 accumulators of 16 VGPRs).  LDS rows are padded so that the ds_read_b64
// A-fragment reads (26-dword stride) and the ds_read_b128 B-fragment reads (52-dword stride) are
// bank-conflict free.  Epilogue fuses bias, eval-BN affine (fmaf), residual add and an optional output
// row scatter (window_reverse + roll + crop of the reference as a precomputed row map).
#include "common.h"

namespace {

constexpr int BM = 128, BN = 96, KC = 96;
constexpr int A_LD = KC + 8;          // bytes per A row in LDS (104 B = 26 dwords)
constexpr int W_LD = KC + 8;          // bf16 elements per W row in LDS (208 B = 52 dwords)

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

struct GemmParams {
  SdfSpikeGemmDesc d;
};

// 8 spike bytes {0,1} -> 8 bf16 {0, 1.0}
__device__ __forceinline__ bf16x8 expand_spikes(uint2 v) {
  union { bf16x8 h; uint32_t u[4]; } r;
  // bytes [b0 b1 b2 b3] -> (b0 | b1 << 16) * 0x3F80 ; (b2 | b3 << 16) * 0x3F80
  r.u[0] = __builtin_amdgcn_perm(0u, v.x, 0x0c010c00u) * 0x3F80u;
  r.u[1] = __builtin_amdgcn_perm(0u, v.x, 0x0c030c02u) * 0x3F80u;
  r.u[2] = __builtin_amdgcn_perm(0u, v.y, 0x0c010c00u) * 0x3F80u;
  r.u[3] = __builtin_amdgcn_perm(0u, v.y, 0x0c030c02u) * 0x3F80u;
  return r.h;
}

template <int NSPLIT>
__global__ __launch_bounds__(256) void spike_gemm_kernel(GemmParams P) {
  __shared__ __attribute__((aligned(16))) uint8_t A_s[BM * A_LD];
  __shared__ __attribute__((aligned(16))) uint16_t W_s[NSPLIT * BN * W_LD];

  const SdfSpikeGemmDesc& d = P.d;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int l31 = lane & 31, lh = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int K = d.K, N = d.N;

  f32x16 acc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;

  // per-thread A staging coordinates: 3 chunks of 16 B; chunk id c -> row c/6, 16-byte column c%6
  int a_row[3], a_c16[3];
  int64_t a_base[3];            // row base offset (normal) or -1 when row >= M
  int zt[3], zb[3], zn[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    int c = tid + 256 * i;
    a_row[i] = c / 6;
    a_c16[i] = c % 6;
    int64_t m = m0 + a_row[i];
    if (m < d.M) {
      a_base[i] = m * d.lda;
      if (d.zg_nH > 0) {
        int64_t bn = (int64_t)d.zg_B * d.zg_N1;
        zt[i] = (int)(m / bn);
        int64_t rem = m - (int64_t)zt[i] * bn;
        zb[i] = (int)(rem / d.zg_N1);
        zn[i] = (int)(rem - (int64_t)zb[i] * d.zg_N1);
      }
    } else {
      a_base[i] = -1;
    }
  }

  for (int k0 = 0; k0 < K; k0 += KC) {
    // ---- stage A tile (u8) ----
    uint4 areg[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      int k = k0 + 16 * a_c16[i];
      areg[i] = make_uint4(0, 0, 0, 0);
      if (a_base[i] >= 0 && k < K) {
        const uint8_t* src;
        if (d.zg_nH > 0) {
          int g = k >> 5, dd = k & 31;
          src = d.A + ((((int64_t)zb[i] * d.zg_nH + g) * d.zg_T + zt[i]) * d.zg_N1 + zn[i]) * 32 + dd;
        } else {
          src = d.A + a_base[i] + k;
        }
        areg[i] = *reinterpret_cast<const uint4*>(src);
      }
    }
    // ---- stage W tile (bf16 planes) ----
    constexpr int WCH = NSPLIT * BN * (KC / 8);                 // 16-byte chunks
    constexpr int WIT = (WCH + 255) / 256;
    uint4 wreg[WIT];
#pragma unroll
    for (int i = 0; i < WIT; ++i) {
      int c = tid + 256 * i;
      wreg[i] = make_uint4(0, 0, 0, 0);
      if (c < WCH) {
        int p = c / (BN * (KC / 8));
        int rem = c - p * (BN * (KC / 8));
        int n = rem / (KC / 8), c16 = rem % (KC / 8);
        int k = k0 + 8 * c16;
        if (n0 + n < N && k < K)
          wreg[i] = *reinterpret_cast<const uint4*>(d.Wp + ((int64_t)p * N + n0 + n) * K + k);
      }
    }
    __syncthreads();                                             // previous stage's reads are done
#pragma unroll
    for (int i = 0; i < 3; ++i)
      *reinterpret_cast<uint4*>(&A_s[a_row[i] * A_LD + 16 * a_c16[i]]) = areg[i];
#pragma unroll
    for (int i = 0; i < WIT; ++i) {
      int c = tid + 256 * i;
      if (c < WCH) {
        int p = c / (BN * (KC / 8));
        int rem = c - p * (BN * (KC / 8));
        int n = rem / (KC / 8), c16 = rem % (KC / 8);
        *reinterpret_cast<uint4*>(&W_s[(p * BN + n) * W_LD + 8 * c16]) = wreg[i];
      }
    }
    __syncthreads();

    // ---- MFMA over the stage ----
#pragma unroll
    for (int ks = 0; ks < KC / 16; ++ks) {
      uint2 av = *reinterpret_cast<const uint2*>(&A_s[(wave * 32 + l31) * A_LD + ks * 16 + 8 * lh]);
      bf16x8 a = expand_spikes(av);
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) {
#pragma unroll
        for (int p = 0; p < NSPLIT; ++p) {
          bf16x8 b = *reinterpret_cast<const bf16x8*>(&W_s[(p * BN + nt * 32 + l31) * W_LD + ks * 16 + 8 * lh]);
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[nt], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue ----
#pragma unroll
  for (int nt = 0; nt < 3; ++nt) {
    const int n = n0 + nt * 32 + l31;
    if (n >= N) continue;
    const float bs = d.bias ? d.bias[n] : 0.f;
    const float al = d.alpha ? d.alpha[n] : 1.f;
    const float be = d.alpha ? d.beta[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int64_t m = m0 + row;
      if (m >= d.M) continue;
      int64_t dst = d.out_rowmap ? (int64_t)d.out_rowmap[m] : m;
      if (dst < 0) continue;
      float v = acc[nt][r];
      if (d.bias) v = v + bs;
      if (d.alpha) v = __builtin_fmaf(v, al, be);
      if (d.resid) v = v + d.resid[dst * d.ldo + n];
      d.out[dst * d.ldo + n] = v;
    }
  }
}

__global__ __launch_bounds__(256) void split_weight_kernel(const float* __restrict__ W, uint16_t* __restrict__ planes,
                                                           int64_t n, int nsplit) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float r = W[i];
  for (int p = 0; p < nsplit; ++p) {
    // round-to-nearest-even fp32 -> bf16 (finite weights)
    uint32_t u = __float_as_uint(r);
    uint32_t h = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
    planes[(int64_t)p * n + i] = (uint16_t)h;
    r = r - __uint_as_float(h << 16);       // exact: the residual fits fp32
  }
}

}  // namespace

extern "C" int sdf_spike_gemm_fwd(const SdfSpikeGemmDesc* d, void* stream) {
  if (!d) return SDF_E_NULL;
  if (!d->A || !d->Wp || !d->out) return SDF_E_NULL;
  if (d->M < 1 || d->N < 32 || d->K < 32 || d->K % 32 || d->N % 32) return SDF_E_SHAPE;
  if (d->nsplit < 1 || d->nsplit > 3) return SDF_E_DTYPE;
  if (d->alpha && !d->beta) return SDF_E_NULL;
  if (d->zg_nH > 0) {
    if (d->K != d->zg_nH * 32 || d->zg_T < 1 || d->zg_B < 1 || d->zg_N1 < 1) return SDF_E_SHAPE;
    if ((int64_t)d->zg_T * d->zg_B * d->zg_N1 != d->M) return SDF_E_SHAPE;
  } else if (d->lda % 16) {
    return SDF_E_SHAPE;
  }
  if (!sdf_aligned(d->A, 16) || !sdf_aligned(d->Wp, 16) || !sdf_aligned(d->out, 4)) return SDF_E_ALIGN;
  GemmParams P;
  P.d = *d;
  dim3 grid((unsigned)((d->M + BM - 1) / BM), (unsigned)((d->N + BN - 1) / BN)), block(256);
  hipStream_t s = sdf_stream(stream);
  switch (d->nsplit) {
    case 1: hipLaunchKernelGGL(spike_gemm_kernel<1>, grid, block, 0, s, P); break;
    case 2: hipLaunchKernelGGL(spike_gemm_kernel<2>, grid, block, 0, s, P); break;
    default: hipLaunchKernelGGL(spike_gemm_kernel<3>, grid, block, 0, s, P); break;
  }
  SDF_LAUNCH_CHECK();
  return 0;
}

extern "C" int sdf_split_weight_bf16(const float* W, uint16_t* planes, int64_t n, int nsplit, void* stream) {
  if (!W || !planes) return SDF_E_NULL;
  if (n < 1) return SDF_E_SHAPE;
  if (nsplit < 1 || nsplit > 3) return SDF_E_DTYPE;
  hipLaunchKernelGGL(split_weight_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, sdf_stream(stream), W, planes, n,
                     nsplit);
  SDF_LAUNCH_CHECK();
  return 0;
}
