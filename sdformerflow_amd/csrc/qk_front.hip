// First half of the spiking QK window attention as ONE launch for gfx950 (rows a5 / a6 of SURVEY.md section 8):
//
//   xs  = SN_proj( x gathered through the slice map )                          reference Spiking_swin_transformer3D.py:670, 789-804
//   q|k = SN_q/k( BN( xs [Wq;Wk]^T ) [+ positional encoding on k] )            :671-680
//   E   = k AND SN2_q( sum over each head's 32 channels of q )                 :687-694   (token gate)
//
// which `sdf_qk_attn_fwd` ran as three launches (neuron over the slices, q|k spike GEMM with fused neurons, gate: 49 us of a 69 us
// half-block at stage 0, 32 of 44 us at stage 2 - three latency-bound launches with the slice spikes and q|k round-tripping
// through L2 / HBM between them, profiles/r3p_forward_sequence.txt).  All three steps are local to one PAIR of window slices
// (the T' = 2 "time" steps of the reference's raw (T', B_, Wh, Ww, C) view are slices b' and B_ + b') and, from the GEMM on, to
// one HEAD: a workgroup owns (slice pair b', head g) - 162 rows x the head's 32 q and 32 k columns.
//   * the slice spikes are made on the fly: per K chunk of 96 channels the workgroup loads the 2 x N1 rows of x through the slice
//     map (fp32, 16-byte loads), runs SN_proj over the pair in registers and writes the bytes into an LDS image - the slice
//     spike tensor never exists in memory (the nH workgroups of a pair repeat this; they are neighbours on one XCD and share
//     the x rows in its L2);
//   * v_mfma_f32_32x32x16_f16 with the WEIGHTS as the row operand and the 32 tokens of a wave as the column operand, one
//     accumulator per (q | k, time step): a lane then holds, for its token, 16 of the head's channels at both time steps - BN,
//     the positional term and SN_q / SN_k run on the accumulators in registers; same k order and plane order as the GEMM kernel
//     it replaces, so q and k are bit-equal to the four-launch form;
//   * the token gate is a 16-register sum + one cross-half shuffle (the head's 32 channels sit in the two lane halves), SN2_q over
//     the two steps, and an AND on the k bits still in registers; E leaves as one dword per 4 channels.
// The second half (projection through the head scramble + BN + scatter + shortcut) reads E exactly as before.
// Compiled with -ffp-contract=off (the neuron arithmetic is the separately-rounded op sequence of neuron.hip).
#include "spike_mm.h"
#include "switches.h"
#include <stdlib.h>

namespace sdfmm {
namespace {

constexpr int QF_KC = 96;                 // channels per K chunk
constexpr int QF_AP = QF_KC + 8;          // bytes per token row of the spike image (26 dwords: conflict-free ds_read_b64)
constexpr int QF_WP = QF_KC + 8;          // halves per weight row (52 dwords: conflict-free ds_read_b128)
constexpr int QF_NT = 96;                 // token slots (3 waves x 32 MFMA columns)

struct QkFrontParams {
  const float* x;
  const int32_t* map;
  int64_t rows;                           // B_ * N1
  int B_, N1, C, nH;
  const uint16_t* wq; const uint16_t* wk; // fp16 planes; row pitch C, plane strides below (elements)
  int64_t wq_plane, wk_plane;
  const float *q_al, *q_be, *k_al, *k_be; // (C) each, or null (alpha 1, beta 0)
  const float* pe; int64_t pe_ld;         // k's additive term pe[(t * N1 + n) * pe_ld + c], or null
  float q_asc, k_asc;
  SdfNeuronCfg sn_proj, sn_q, sn_k, sn2_q;
  float it_proj, it_q, it_k, it_2;
  uint8_t* e;                             // (2, rows, C) gated k spikes
  uint8_t* qs; uint8_t* ks;               // KEEP: q / k spikes, row strides ldq / ldk
  int64_t ldq, ldk;
};

template <int NK, bool KEEP>
__global__ __launch_bounds__(256, 3) void qk_front_kernel(QkFrontParams P) {
  __shared__ __attribute__((aligned(16))) uint8_t A_s[2 * QF_NT * QF_AP];            // [t][token][QF_AP]
  __shared__ __attribute__((aligned(16))) uint16_t W_s[2 * 64 * QF_WP];               // [plane][q 0-31 | k 32-63][QF_WP]
  __shared__ __attribute__((aligned(16))) float par_s[4 * 32];                        // BN of the head: q alpha | q beta | k alpha | k beta
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  // (slice pair, head): the nH workgroups of a pair are neighbours on one XCD (workgroup id b sits on XCD b % 8)
  int item = blockIdx.x;
  const int G = gridDim.x;
  if ((G & 7) == 0) item = (item & 7) * (G >> 3) + (item >> 3);
  const int bp = item / P.nH, g = item - bp * P.nH;
  const int C = P.C, N1 = P.N1;

  // staging coordinates: (token, 4-channel group) pairs of the chunk, 24 groups per token
  constexpr int XIT = (QF_NT * 24 + 255) / 256;                                       // 9 (81 tokens: 8 passes carry work)
  f32x16 aq[2], ak[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) aq[t][e] = ak[t][e] = 0.f;

  // source rows of this thread's (token, 4-channel group) pairs, both time steps: loaded once (the chunk loop would re-load the map
  // in front of every x load - two dependent global-memory latencies per chunk)
  int srow0[XIT], srow1[XIT];
#pragma unroll
  for (int i = 0; i < XIT; ++i) {
    const int tk = (tid + 256 * i) / 24;
    srow0[i] = srow1[i] = -2;                                                          // -2: no such token (nothing to write)
    if (tk < N1) {
      srow0[i] = P.map[(int64_t)bp * N1 + tk];
      srow1[i] = P.map[((int64_t)P.B_ + bp) * N1 + tk];
    }
  }
  // the head's BatchNorm pairs go to LDS now (no global-memory latency per accumulator quad in the epilogue)
  if (tid < 128) {
    const int which = tid >> 5, c = g * 32 + (tid & 31);
    const float* src = which == 0 ? P.q_al : which == 1 ? P.q_be : which == 2 ? P.k_al : P.k_be;
    par_s[tid] = src ? src[c] : ((which & 1) ? 0.f : 1.f);
  }
  const int nchunks = C / QF_KC;
  // one chunk's operands: request (global loads into registers) and commit (SN_proj, bytes and weights into LDS).  Three workgroups
  // per CU hide each other's load latency (168 registers: two workgroups per CU ran stage 0 in 36 us, three in 28); requesting
  // chunk c + 1 under the MFMAs of chunk c instead measured SLOWER (its registers cost the third workgroup)
  constexpr int XB = 5;                                                               // x pairs per request batch (two batches a chunk)
  int tl = tid;                                                                       // staging coordinates are re-derived from this (laundered
                                                                                      // per chunk), not kept in ~60 registers across the MFMAs
  auto request_w = [&](uint4 (&wreg)[6], int kc) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int pc = tl + 256 * i;                                                    // 1536 pieces: 2 planes x 64 rows x 12
      const int p = pc / 768, r = (pc - p * 768) / 12, c8 = pc - p * 768 - r * 12;
      const uint16_t* src = r < 32 ? P.wq + p * P.wq_plane + (int64_t)(g * 32 + r) * C
                                   : P.wk + p * P.wk_plane + (int64_t)(g * 32 + r - 32) * C;
      wreg[i] = *reinterpret_cast<const uint4*>(src + kc + 8 * c8);
    }
  };
  auto commit_w = [&](const uint4 (&wreg)[6]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int pc = tl + 256 * i;
      const int p = pc / 768, r = (pc - p * 768) / 12, c8 = pc - p * 768 - r * 12;
      *reinterpret_cast<uint4*>(&W_s[(p * 64 + r) * QF_WP + 8 * c8]) = wreg[i];
    }
  };
  auto request_x = [&](float4 (&x0)[XB], float4 (&x1)[XB], int kc, const int (&r0)[XB], const int (&r1)[XB], int i0) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < XB; ++j) {
      const int pr = tl + 256 * (i0 + j);
      const int c4 = pr % 24;
      x0[j] = x1[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r0[j] >= 0) x0[j] = *reinterpret_cast<const float4*>(P.x + (int64_t)r0[j] * C + kc + 4 * c4);
      if (r1[j] >= 0) x1[j] = *reinterpret_cast<const float4*>(P.x + (int64_t)r1[j] * C + kc + 4 * c4);
    }
  };
  auto commit_x = [&](const float4 (&x0)[XB], const float4 (&x1)[XB], const int (&r0)[XB], int i0) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < XB; ++j) {
      if (r0[j] != -2) {
        const int pr = tl + 256 * (i0 + j);
        const int tok = pr / 24, c4 = pr - tok * 24;
        const float a0[4] = {x0[j].x, x0[j].y, x0[j].z, x0[j].w}, a1[4] = {x1[j].x, x1[j].y, x1[j].z, x1[j].w};
        uint32_t w0 = 0, w1 = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xs[2] = {a0[e], a1[e]};
          float sp[2];
          neuron_T<NK, 2>(xs, sp, P.sn_proj, P.it_proj);
          w0 |= (sp[0] != 0.f ? 1u : 0u) << (8 * e);
          w1 |= (sp[1] != 0.f ? 1u : 0u) << (8 * e);
        }
        *reinterpret_cast<uint32_t*>(&A_s[tok * QF_AP + 4 * c4]) = w0;
        *reinterpret_cast<uint32_t*>(&A_s[(QF_NT + tok) * QF_AP + 4 * c4]) = w1;
      }
    }
  };
  // source rows of request batch i0 (-2 beyond the last pass)
  auto batch_rows = [&](int (&r0)[XB], int (&r1)[XB], int i0) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < XB; ++j) {
      r0[j] = i0 + j < XIT ? srow0[i0 + j < XIT ? i0 + j : 0] : -2;
      r1[j] = i0 + j < XIT ? srow1[i0 + j < XIT ? i0 + j : 0] : -2;
    }
  };
#pragma unroll 1
  for (int ch = 0; ch < nchunks; ++ch) {
    const int kc = ch * QF_KC;
    if (ch > 0) __syncthreads();                                                      // the previous chunk's fragments are read
    asm volatile("" : "+v"(tl));
    {
      uint4 wreg[6];
      request_w(wreg, kc);
      commit_w(wreg);
    }
#pragma unroll
    for (int i0 = 0; i0 < XIT; i0 += XB) {
      float4 x0[XB], x1[XB];
      int r0[XB], r1[XB];
      batch_rows(r0, r1, i0);
      request_x(x0, x1, kc, r0, r1, i0);
      commit_x(x0, x1, r0, i0);
    }
    __syncthreads();
    // ---- MFMAs: waves 0..2 own tokens 32 w .. 32 w + 31 (the fourth wave only stages) ----
    if (wave < 3 && 32 * wave < N1) {
#pragma unroll
      for (int ks = 0; ks < QF_KC / 16; ++ks) {
        const bf16x8 b0 = expand_spikes<2>(*reinterpret_cast<const uint2*>(&A_s[(32 * wave + l31) * QF_AP + 16 * ks + 8 * lh]));
        const bf16x8 b1 = expand_spikes<2>(*reinterpret_cast<const uint2*>(&A_s[(QF_NT + 32 * wave + l31) * QF_AP + 16 * ks + 8 * lh]));
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const bf16x8 wq = *reinterpret_cast<const bf16x8*>(&W_s[(p * 64 + l31) * QF_WP + 16 * ks + 8 * lh]);
          const bf16x8 wk = *reinterpret_cast<const bf16x8*>(&W_s[(p * 64 + 32 + l31) * QF_WP + 16 * ks + 8 * lh]);
          aq[0] = mma<2>(wq, b0, aq[0]);
          aq[1] = mma<2>(wq, b1, aq[1]);
          ak[0] = mma<2>(wk, b0, ak[0]);
          ak[1] = mma<2>(wk, b1, ak[1]);
        }
        if (ks & 1) __builtin_amdgcn_sched_barrier(0);                                // at most two k-steps of fragments in flight
      }
    }
  }

  // ---- epilogue: BN (+ positional term) -> SN_q / SN_k over the pair -> token gate -> E ----
  const int n = 32 * wave + l31;
  if (wave >= 3 || n >= N1) return;
  float4 pe0[4], pe1[4];                                                               // the lane's positional terms: one round of loads
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) {
    pe0[q4] = pe1[q4] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (P.pe) {
      pe0[q4] = *reinterpret_cast<const float4*>(P.pe + (int64_t)n * P.pe_ld + g * 32 + 8 * q4 + 4 * lh);
      pe1[q4] = *reinterpret_cast<const float4*>(P.pe + (int64_t)(N1 + n) * P.pe_ld + g * 32 + 8 * q4 + 4 * lh);
    }
  }
  uint32_t kbits[2] = {0u, 0u}, qbits[2] = {0u, 0u};                                   // bit 4 q4 + i = channel 8 q4 + 4 lh + i of the head
  float cnt[2] = {0.f, 0.f};
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) {
    const int cl = 8 * q4 + 4 * lh;
    const float4 qa = *reinterpret_cast<const float4*>(&par_s[cl]), qb = *reinterpret_cast<const float4*>(&par_s[32 + cl]);
    const float4 ka = *reinterpret_cast<const float4*>(&par_s[64 + cl]), kb = *reinterpret_cast<const float4*>(&par_s[96 + cl]);
    const float qa4[4] = {qa.x, qa.y, qa.z, qa.w}, qb4[4] = {qb.x, qb.y, qb.z, qb.w};
    const float ka4[4] = {ka.x, ka.y, ka.z, ka.w}, kb4[4] = {kb.x, kb.y, kb.z, kb.w};
    const float p04[4] = {pe0[q4].x, pe0[q4].y, pe0[q4].z, pe0[q4].w}, p14[4] = {pe1[q4].x, pe1[q4].y, pe1[q4].z, pe1[q4].w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float xq[2], xk[2], sq[2], sk[2];
      xq[0] = __builtin_fmaf(aq[0][4 * q4 + i] * P.q_asc, qa4[i], qb4[i]);
      xq[1] = __builtin_fmaf(aq[1][4 * q4 + i] * P.q_asc, qa4[i], qb4[i]);
      xk[0] = __builtin_fmaf(ak[0][4 * q4 + i] * P.k_asc, ka4[i], kb4[i]);
      xk[1] = __builtin_fmaf(ak[1][4 * q4 + i] * P.k_asc, ka4[i], kb4[i]);
      if (P.pe) { xk[0] = xk[0] + p04[i]; xk[1] = xk[1] + p14[i]; }
      neuron_T<NK, 2>(xq, sq, P.sn_q, P.it_q);
      neuron_T<NK, 2>(xk, sk, P.sn_k, P.it_k);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        cnt[t] += sq[t];
        qbits[t] |= (sq[t] != 0.f ? 1u : 0u) << (4 * q4 + i);
        kbits[t] |= (sk[t] != 0.f ? 1u : 0u) << (4 * q4 + i);
      }
    }
  }
  // the head's other 16 channels live in the other lane half
  float a2[2], gate[2];
  a2[0] = cnt[0] + __shfl_xor(cnt[0], 32);
  a2[1] = cnt[1] + __shfl_xor(cnt[1], 32);
  neuron_T<NK, 2>(a2, gate, P.sn2_q, P.it_2);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const uint32_t eb = gate[t] != 0.f ? kbits[t] : 0u;
    const int64_t row = (int64_t)t * P.rows + (int64_t)bp * N1 + n;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int c = g * 32 + 8 * q4 + 4 * lh;
      const uint32_t nib = (eb >> (4 * q4)) & 0xFu;
      *reinterpret_cast<uint32_t*>(P.e + row * C + c) = (nib & 1u) | ((nib & 2u) << 7) | ((nib & 4u) << 14) | ((nib & 8u) << 21);
      if (KEEP) {
        const uint32_t nq = (qbits[t] >> (4 * q4)) & 0xFu, nk = (kbits[t] >> (4 * q4)) & 0xFu;
        *reinterpret_cast<uint32_t*>(P.qs + row * P.ldq + c) = (nq & 1u) | ((nq & 2u) << 7) | ((nq & 4u) << 14) | ((nq & 8u) << 21);
        *reinterpret_cast<uint32_t*>(P.ks + row * P.ldk + c) = (nk & 1u) | ((nk & 2u) << 7) | ((nk & 4u) << 14) | ((nk & 8u) << 21);
      }
    }
  }
}

template <bool KEEP>
int launch_nk(const QkFrontParams& P, int nk, dim3 grid, hipStream_t s) {
  switch (nk) {
    case 0: SDF_LAUNCH((qk_front_kernel<0, KEEP>), grid, dim3(256), 0, s, P); return 0;
    case 1: SDF_LAUNCH((qk_front_kernel<1, KEEP>), grid, dim3(256), 0, s, P); return 0;
    default: SDF_LAUNCH((qk_front_kernel<2, KEEP>), grid, dim3(256), 0, s, P); return 0;
  }
}

}  // namespace

// Shapes and settings the one-launch first half is built for; everything else keeps the three launches.
bool qk_front_supports(const SdfQkAttnDesc* d) {
  if (d->nsplit != 2 || d->Tq != 2 || d->N1 < 1 || d->N1 > QF_NT || d->C % QF_KC || d->C != d->nH * 32) return false;
  // Wins where there are many slice pairs and one or two K chunks (config 2 stage 0: 49 -> 28 us, stage 1: 37 -> 24 us).  From three
  // chunks on (C >= 288: few pairs, every head's workgroup repeats the pair's SN_proj, a chain of chunk latencies per workgroup)
  // the three pipelined launches are as fast or faster (stage 2: 32 vs 32 us, stage 3: 33 vs 55 us) and stay.  SDF_QK_FRONT_ANY=1
  // lifts the limit (tests).
  const char* e_any = sdf_sw(SW_QK_FRONT_ANY);                     // (read per call: a test may scope it)
  const bool any = e_any && e_any[0] == '1';
  if (!any && d->C / QF_KC > 2) return false;
  const SdfNeuronCfg* ns[4] = {&d->sn_proj, &d->sn_q, &d->sn_k, &d->sn2_q};
  const int nk = neuron_class(*ns[0]);
  for (const SdfNeuronCfg* n : ns) {
    if (n->kind != SDF_LIF && n->kind != SDF_IF && n->kind != SDF_PSN) return false;
    if (n->kind == SDF_PSN && (!n->psn_w || !n->psn_b)) return false;
    if (!sdf_tau_ok(n->kind, n->tau)) return false;
    if (neuron_class(*n) != nk) return false;
  }
  if (d->B_ * d->nH >= (1LL << 31) || d->B_ * 2 * d->N1 >= (1LL << 31)) return false;
  const bool fused = d->qk_planes != nullptr;
  const float* al[4] = {fused ? d->qk_alpha : d->q_alpha, fused ? d->qk_beta : d->q_beta, fused ? d->qk_alpha : d->k_alpha,
                        fused ? d->qk_beta : d->k_beta};
  if ((al[0] == nullptr) != (al[1] == nullptr) || (al[2] == nullptr) != (al[3] == nullptr)) return false;
  return sdf_aligned(d->x, 16) && sdf_aligned(fused ? (const void*)d->qk_planes : (const void*)d->q_planes, 16) &&
         (fused || sdf_aligned(d->k_planes, 16));
}

int launch_qk_front(const SdfQkAttnDesc* d, uint8_t* e, uint8_t* qk, bool keep, hipStream_t s) {
  QkFrontParams P = {};
  const int C = d->C;
  const int64_t rows = d->B_ * d->N1, M = rows * d->Tq;
  P.x = d->x; P.map = d->slice_map; P.rows = rows; P.B_ = (int)d->B_; P.N1 = d->N1; P.C = C; P.nH = d->nH;
  if (d->qk_planes) {
    P.wq = d->qk_planes; P.wk = d->qk_planes + (int64_t)C * C; P.wq_plane = P.wk_plane = 2LL * C * C;
    P.q_al = d->qk_alpha; P.q_be = d->qk_beta;
    P.k_al = d->qk_alpha ? d->qk_alpha + C : nullptr; P.k_be = d->qk_beta ? d->qk_beta + C : nullptr;
    P.pe = d->qk_add ? d->qk_add + C : nullptr; P.pe_ld = 2LL * C;
    P.q_asc = P.k_asc = d->qk_acc_scale;
    P.qs = qk; P.ks = qk + C; P.ldq = P.ldk = 2LL * C;
  } else {
    P.wq = d->q_planes; P.wk = d->k_planes; P.wq_plane = P.wk_plane = (int64_t)C * C;
    P.q_al = d->q_alpha; P.q_be = d->q_beta; P.k_al = d->k_alpha; P.k_be = d->k_beta;
    P.pe = d->k_add; P.pe_ld = C;
    P.q_asc = d->q_acc_scale; P.k_asc = d->k_acc_scale;
    P.qs = qk; P.ks = qk + M * C; P.ldq = P.ldk = C;
  }
  if (P.q_asc == 0.f) P.q_asc = 1.f;
  if (P.k_asc == 0.f) P.k_asc = 1.f;
  P.sn_proj = d->sn_proj; P.sn_q = d->sn_q; P.sn_k = d->sn_k; P.sn2_q = d->sn2_q;
  P.it_proj = inv_tau_of(d->sn_proj); P.it_q = inv_tau_of(d->sn_q); P.it_k = inv_tau_of(d->sn_k); P.it_2 = inv_tau_of(d->sn2_q);
  P.e = e;
  const dim3 grid((unsigned)(d->B_ * d->nH));
  const int nk = neuron_class(d->sn_proj);
  const int rc = keep ? launch_nk<true>(P, nk, grid, s) : launch_nk<false>(P, nk, grid, s);
  if (rc) return rc;
  hipError_t err = hipGetLastError();
  return err != hipSuccess ? (int)err : 0;
}

}  // namespace sdfmm
