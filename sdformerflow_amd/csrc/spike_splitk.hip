// Split-K plan and its deterministic second pass, shared by the spike matrix-multiply kernels (spike_mm_pp.hip, spike_gemm.hip).
#include "spike_mm.h"
#include "switches.h"
#include <stdlib.h>

namespace sdfmm {
namespace {

// Second pass of split-K: out = epilogue( sum_k partial[k] ), chunks added in k order (deterministic).
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, int ksplit, int64_t M, int N,
                                                            float asc, const float* bias, const float* alpha, const float* beta,
                                                            const float* resid, const int* rowmap, float* out, int64_t ldo) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int nq = N / 4;
  if (q >= M * nq) return;
  const int64_t m = q / nq;
  const int n = (int)(q - m * nq) * 4;
  float4 a = *reinterpret_cast<const float4*>(partial + m * N + n);
  for (int k = 1; k < ksplit; ++k) {
    const float4 b = *reinterpret_cast<const float4*>(partial + ((int64_t)k * M + m) * N + n);
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  a.x *= asc; a.y *= asc; a.z *= asc; a.w *= asc;                // power of two: exact
  if (bias) { const float4 b = *reinterpret_cast<const float4*>(bias + n); a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
  if (alpha) {
    const float4 al = *reinterpret_cast<const float4*>(alpha + n), be = *reinterpret_cast<const float4*>(beta + n);
    a.x = __builtin_fmaf(a.x, al.x, be.x); a.y = __builtin_fmaf(a.y, al.y, be.y);
    a.z = __builtin_fmaf(a.z, al.z, be.z); a.w = __builtin_fmaf(a.w, al.w, be.w);
  }
  int64_t dst = m;
  if (rowmap) dst = rowmap[m];
  if (dst < 0) return;
  if (resid) { const float4 r = *reinterpret_cast<const float4*>(resid + dst * ldo + n); a.x += r.x; a.y += r.y; a.z += r.z; a.w += r.w; }
  *reinterpret_cast<float4*>(out + dst * ldo + n) = a;
}

}  // namespace

// split-K when the tiles alone cannot occupy the chip (small M, large K): needs the fp32 epilogue and a
// caller-provided workspace of ksplit*M*N floats; partial sums are combined in k order by a second kernel
void plan_splitk(GemmParams& P, int kc) {
  const SdfSpikeGemmDesc& d = P.d;
  const int S = (d.K + kc - 1) / kc;
  P.ksplit = 1;
  P.spc = S;
  P.partial = nullptr;
  if (d.sn_T == 0 && P.ntiles <= 128 && S >= 4 && d.workspace) {
    // K chunks per tile: the count that minimises (rounds of the 256 workgroups) x (stages per item + a fixed per-item cost of
    // about three stages: prologue, epilogue, partial store).  Round 2 took ceil(256 / tiles), which on 40 tiles is 7 chunks =
    // 280 items = TWO rounds of 16 stages where 6 chunks = 240 items run ONE round of 18 (U-Net res-blocks: 49 -> 3x us).
    int kmax = S / 2 < 32 ? S / 2 : 32;
    if (const char* e = sdf_sw(SW_KSPLIT_MULT)) {                // tuning override: the old rule, oversubscribed
      int ks = (256 + P.ntiles - 1) / P.ntiles * (atoi(e) > 0 ? atoi(e) : 1);
      kmax = ks < kmax ? ks : kmax;
    }
    int best = 1;
    int64_t best_cost = -1;
    for (int ks = 1; ks <= kmax; ++ks) {
      if ((int64_t)ks * d.M * d.N * 4 > d.workspace_bytes) break;
      const int64_t rounds = ((int64_t)P.ntiles * ks + 255) / 256, spc = (S + ks - 1) / ks;
      const int64_t cost = rounds * (spc + 3);
      if (best_cost < 0 || cost < best_cost || (sdf_sw(SW_KSPLIT_MULT) && ks == kmax)) { best_cost = cost; best = ks; }
    }
    if (best > 1 && sdf_aligned(d.workspace, 16)) {
      P.ksplit = best;
      P.spc = (S + best - 1) / best;
      P.partial = reinterpret_cast<float*>(d.workspace);
    }
  }
}

int launch_splitk_reduce(const GemmParams& P, hipStream_t s) {
  if (P.ksplit <= 1) return 0;
  const SdfSpikeGemmDesc& d = P.d;
  const int64_t quads = d.M * (d.N / 4);
  SDF_LAUNCH(splitk_reduce_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, P.partial, P.ksplit, d.M,
                     d.N, P.acc_scale, d.bias, d.alpha, d.beta, d.resid, d.out_rowmap, d.out, d.ldo);
  const hipError_t e = hipGetLastError();
  return e != hipSuccess ? (int)e : 0;
}

}  // namespace sdfmm
