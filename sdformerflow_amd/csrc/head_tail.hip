// The two ends of the forward that are not spike matmuls (gfx950).
//
//  sdf_head_conv_sn_fwd : the patch embedding's head - 3x3 / pad 1 convolution of the REAL-valued event voxel (2 input
//      channels) -> eval BatchNorm -> neuron over the T time steps - as one kernel.  The fp32 pre-activation
//      (imgs x H x W x Cout, 212 MB at 288 x 384) never exists; the kernel reads the 8.8 MB voxel and writes 1-byte spikes.
//      (reference Spiking_modules.py:1782: head = conv -> SpikingNormLayer -> Spiking_neuron.)
//      A workgroup = 16 pixels of one image row x 16 channel lanes (Cout / 16 channels each), all T steps: the input
//      patch goes through LDS, the 18 * Cout weights live in registers, the accumulation is one fmaf chain per output
//      in (ky, kx, cin) order, the recurrence runs in registers, spikes leave through LDS as 16-byte stores.
//
//  sdf_flow_out_fwd : sum of the per-step flow predictions over time + nearest-neighbour upsampling to the input
//      resolution (reference Spiking_STSwinNet.py:289-303: flow.sum(0) then F.interpolate(scale_factor=H/h, W/w)).
// Compiled with -ffp-contract=off (the neuron arithmetic is the separately-rounded op sequence of neuron.hip).
#include "spike_mm.h"
#include "switches.h"
#include <stdlib.h>

namespace {
using sdfmm::lif_steps;

struct HeadParams {
  SdfHeadConvDesc d;
  float inv_tau;
};

template <int T, int CPT, int CIN>
__global__ __launch_bounds__(256) void head_conv_sn_kernel(HeadParams P) {
  const SdfHeadConvDesc& d = P.d;
  constexpr int PX = 16, COLS = PX + 2;
  __shared__ float xs_s[T][3][COLS][CIN];                       // input patch of the 16 pixels, all T steps
  __shared__ __attribute__((aligned(16))) uint8_t sp_s[T][PX][16 * CPT];
  const int tid = threadIdx.x;
  const int px = tid >> 4, cl = tid & 15;
  const int xt = d.W / PX + (d.W % PX ? 1 : 0);                 // pixel tiles per row
  const int tile = blockIdx.x;
  const int x0 = (tile % xt) * PX;
  const int y = (tile / xt) % d.H;
  const int b = tile / (xt * d.H);
  const int Cout = 16 * CPT;

  // ---- stage the patch (zero outside the image) ----
  const bool strided = d.x_sy != 0;
  for (int i = tid; i < T * 3 * COLS * CIN; i += 256) {
    const int ci = i % CIN, c = (i / CIN) % COLS, r = (i / (CIN * COLS)) % 3, t = i / (CIN * COLS * 3);
    const int yy = y + r - 1, xx = x0 + c - 1;
    float v = 0.f;
    if ((unsigned)yy < (unsigned)d.H && (unsigned)xx < (unsigned)d.W)
      v = strided ? d.x[(int64_t)b * d.x_sb + (int64_t)t * d.x_st + (int64_t)yy * d.x_sy + (int64_t)xx * d.x_sx + d.x_sc[ci]]
                  : d.x[((((int64_t)b * T + t) * d.H + yy) * d.W + xx) * CIN + ci];
    (&xs_s[0][0][0][0])[i] = v;
  }
  // ---- this lane's weights: w[cout][cin][ky][kx] (the module's own layout) ----
  float wr[CPT][9][CIN];
#pragma unroll
  for (int c = 0; c < CPT; ++c)
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci) wr[c][tp][ci] = d.w[((cl * CPT + c) * CIN + ci) * 9 + tp];
  __syncthreads();

  float acc[T][CPT];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int c = 0; c < CPT; ++c) acc[t][c] = 0.f;
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) {
    const int ky = tp / 3, kx = tp % 3;
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const float xv = xs_s[t][ky][px + kx][ci];
#pragma unroll
        for (int c = 0; c < CPT; ++c) acc[t][c] = __builtin_fmaf(xv, wr[c][tp][ci], acc[t][c]);
      }
  }

  // ---- BN -> neuron over T, per channel ----
  const bool soft = d.soft_reset != 0;
#pragma unroll
  for (int c = 0; c < CPT; ++c) {
    const int n = cl * CPT + c;
    const float al = d.alpha ? d.alpha[n] : 1.f, be = d.alpha ? d.beta[n] : 0.f;
    float xs[T], sp[T];
#pragma unroll
    for (int t = 0; t < T; ++t) xs[t] = __builtin_fmaf(acc[t][c], al, be);
    if (d.sn_kind == SDF_PSN) {
#pragma unroll
      for (int t = 0; t < T; ++t) {
        float hh = d.psn_b[t];
#pragma unroll
        for (int k = 0; k < T; ++k) hh = __builtin_fmaf(d.psn_w[t * T + k], xs[k], hh);
        sp[t] = hh >= 0.f ? 1.f : 0.f;
      }
    } else {
      lif_steps<T>(xs, sp, d.sn_kind, soft, d.v_reset, d.v_th, d.tau, P.inv_tau);
    }
#pragma unroll
    for (int t = 0; t < T; ++t) sp_s[t][px][n] = (uint8_t)(sp[t] != 0.f);
  }
  __syncthreads();
  // ---- 16 pixels x Cout bytes are contiguous in the NHWC spike image: 16-byte stores ----
  constexpr int CH16 = PX * CPT;                                // 16-byte chunks per time step
  for (int i = tid; i < T * CH16; i += 256) {
    const int t = i / CH16, c16 = i - t * CH16;
    const int p = (c16 * 16) / Cout;                            // pixel of this chunk
    if (x0 + p < d.W)
      *reinterpret_cast<uint4*>(d.out + ((((int64_t)b * T + t) * d.H + y) * d.W + x0) * Cout + c16 * 16) =
          *reinterpret_cast<const uint4*>(&sp_s[t][0][0] + c16 * 16);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// The same head on the exact fp32 matrix pipe (LIF / IF neurons; the PSN keeps the kernel above: it needs all T pre-activations of a
// channel at once).  A wave owns 32 consecutive pixels of one image row for all T steps:
//   * v_mfma_f32_32x32x2_f32, the WEIGHTS as the row operand (9 * Cin / 2 registers per 32-channel block, loaded once per wave), the
//     pixels as the column operand: k-step s multiplies (tap, channel) pairs k = 2s + (lane / 32), i.e. with Cin = 2 one tap per
//     step, lane half = polarity.  fp32 products and sums - the numerics of an fmaf chain, no 16-bit split;
//   * the operand values come straight from the voxel by buffer loads (lanes = consecutive x: coalesced; the 9 taps are L1 hits;
//     the time step is the instruction's scalar offset; outside the image the lane's offset is out of range -> 0), requested one
//     step ahead of the MFMAs that consume them;
//   * a lane's accumulator quads are 4 consecutive channels of its pixel: BatchNorm and the neuron step run on them in registers
//     (the membrane of 16 channels per block lives in the lane across the T steps), four spikes pack to one dword;
//   * spikes leave through a wave-private LDS tile (T' steps x 32 pixels x Cout bytes) as whole 16-byte pieces of the NHWC image.
// The first version (above) spent 108 us on config 2's 288 x 384 head - 4.5 x its own FMA issue time, on per-lane weight loads,
// index arithmetic and byte traffic through LDS; this one takes 50 us (profiles/r3p_forward_sequence.txt): 2.5 GFLOP incl. the
// padding of 48 channels to 64 rows is 16 us of the fp32 matrix pipe, the BatchNorm + LIF + packing of 24 outputs per lane and
// step is about as many vector cycles, and two waves per SIMD (186 registers) overlap the two only partly.
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4h;

template <int T, int CIN, int NOUT, bool FAST>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void head_conv_mfma_kernel(HeadParams P, int tiles) {
  constexpr int KS = 9 * CIN / 2, NBLK = (NOUT + 31) / 32, TF = 5, ROW = 32 * NOUT;        // ROW: spike bytes of a tile per time step
  static_assert(T % TF == 0, "flush period");
  const SdfHeadConvDesc& d = P.d;
  __shared__ __attribute__((aligned(16))) uint8_t sp_s[4][TF][ROW];
  __shared__ __attribute__((aligned(16))) float par_s[2][32 * NBLK];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  if (tid < 32 * NBLK) {
    par_s[0][tid] = (tid < NOUT && d.alpha) ? d.alpha[tid] : 1.f;
    par_s[1][tid] = (tid < NOUT && d.alpha) ? d.beta[tid] : 0.f;
  }
  // this lane's weights: row = channel 32 blk + l31, k = 2s + lh -> (tap, ci); module layout w[cout][cin][ky][kx]
  float wa[NBLK][KS];
#pragma unroll
  for (int blk = 0; blk < NBLK; ++blk)
#pragma unroll
    for (int s_ = 0; s_ < KS; ++s_) {
      const int k = 2 * s_ + lh, tap = k / CIN, ci = k - tap * CIN, n = 32 * blk + l31;
      wa[blk][s_] = n < NOUT ? d.w[(n * CIN + ci) * 9 + tap] : 0.f;
    }
  __syncthreads();
  const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x), 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t o_rs = __builtin_amdgcn_make_buffer_rsrc(d.out, 0, 0x7FFFFFFF, 0x00020000);
  const int xt = d.W >> 5;
  const bool soft = d.soft_reset != 0, is_if = d.sn_kind == SDF_IF;
  constexpr bool fast = FAST;                                          // LIF, soft reset, multiplicative charge (the shipped setting)
  const bool reset0 = soft || d.v_reset == 0.f;
  const float v_th = d.v_th, v_reset = d.v_reset, tau = d.tau, inv_tau = P.inv_tau;
  uint8_t* my_s = &sp_s[wave][0][0];

#pragma unroll 1
  for (int tile = blockIdx.x * 4 + wave; tile < tiles; tile += gridDim.x * 4) {
    const int x0 = (tile % xt) << 5, y = (tile / xt) % d.H, b = tile / (xt * d.H);
    // byte offsets of this lane's KS operand values at t = 0 (0x80000000 = outside the image: the load returns 0)
    uint32_t off[KS];
#pragma unroll
    for (int s_ = 0; s_ < KS; ++s_) {
      const int k = 2 * s_ + lh, tap = k / CIN, ci = k - tap * CIN;
      const int yy = y + tap / 3 - 1, xx = x0 + l31 + tap % 3 - 1;
      const bool ok = (unsigned)yy < (unsigned)d.H && (unsigned)xx < (unsigned)d.W;
      off[s_] = ok ? (uint32_t)(((int64_t)b * d.x_sb + (int64_t)yy * d.x_sy + (int64_t)xx * d.x_sx + d.x_sc[ci]) * 4) : 0x80000000u;
    }
    const uint32_t st4 = (uint32_t)(d.x_st * 4);
    float xin[2][KS];
#pragma unroll
    for (int s_ = 0; s_ < KS; ++s_) xin[0][s_] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rs, off[s_], 0, 0));
    float v[NBLK][16];
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk)
#pragma unroll
      for (int e = 0; e < 16; ++e) v[blk][e] = soft ? 0.f : v_reset;

    auto one_step = [&](int t, const float (&xc)[KS], float (&xn)[KS]) __attribute__((always_inline)) {
      if (t + 1 < T) {
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_)
          xn[s_] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rs, off[s_], (uint32_t)(t + 1) * st4, 0));
      }
      f32x16 acc[NBLK];
#pragma unroll
      for (int blk = 0; blk < NBLK; ++blk)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[blk][e] = 0.f;
#pragma unroll
      for (int s_ = 0; s_ < KS; ++s_)
#pragma unroll
        for (int blk = 0; blk < NBLK; ++blk) acc[blk] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[blk][s_], xc[s_], acc[blk], 0, 0, 0);
      uint8_t* row = my_s + (t % TF) * ROW + l31 * NOUT + 4 * lh;
      int pofs = 4 * lh;                                                // the BatchNorm pairs are re-read every step (not kept in 48 registers)
      asm volatile("" : "+v"(pofs));
#pragma unroll
      for (int blk = 0; blk < NBLK; ++blk)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (32 * blk + 8 * q >= NOUT) continue;                       // padding rows of the last block
          const float4 al = *reinterpret_cast<const float4*>(&par_s[0][32 * blk + 8 * q + pofs]);
          const float4 be = *reinterpret_cast<const float4*>(&par_s[1][32 * blk + 8 * q + pofs]);
          const float a4[4] = {al.x, al.y, al.z, al.w}, b4[4] = {be.x, be.y, be.z, be.w};
          uint32_t word = 0;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float xv = __builtin_fmaf(acc[blk][4 * q + i], a4[i], b4[i]);
            float& vm = v[blk][4 * q + i];
            bool fire;
            if constexpr (fast) {
              const float h = vm + (xv - vm) * inv_tau;
              fire = h - v_th >= 0.f;
              vm = fire ? h - v_th : h;
            } else {
              float h;
              if (is_if) h = vm + xv;
              else {
                const float dl = reset0 ? (xv - vm) : (xv - (vm - v_reset));
                h = vm + ((inv_tau != 0.f) ? dl * inv_tau : dl / tau);
              }
              fire = h - v_th >= 0.f;
              const float sp = fire ? 1.f : 0.f;
              vm = soft ? (h - sp * v_th) : ((1.f - sp) * h + sp * v_reset);
            }
            word |= fire ? (1u << (8 * i)) : 0u;
          }
          *reinterpret_cast<uint32_t*>(row + 32 * blk + 8 * q) = word;
        }
      if (t % TF == TF - 1) {                                           // whole 16-byte pieces of TF rows of the NHWC spike image
        constexpr int PT = ROW / 16, PIECES = TF * PT;
        const int t0 = t - (TF - 1);
        const uint32_t obase = (uint32_t)(((((int64_t)b * T + t0) * d.H + y) * d.W + x0) * NOUT), ostep = (uint32_t)(d.H * d.W * NOUT);
#pragma unroll
        for (int j = 0; j < (PIECES + 63) / 64; ++j) {
          const int pc = lane + 64 * j;
          const int tt = pc / PT, o = (pc - tt * PT) * 16;
          const u32x4h val = *reinterpret_cast<const u32x4h*>(my_s + (pc < PIECES ? tt * ROW + o : 0));
          __builtin_amdgcn_raw_buffer_store_b128(val, o_rs, pc < PIECES ? obase + (uint32_t)tt * ostep + (uint32_t)o : 0x80000000u, 0, 0);
        }
      }
    };
#pragma unroll 1
    for (int t = 0; t < T; t += 2) {                                    // T is odd only for T = 5: the guard is uniform
      one_step(t, xin[0], xin[1]);
      if (t + 1 < T) one_step(t + 1, xin[1], xin[0]);
    }
  }
}

// The PSN form of the same kernel (T <= 10).  A PSN decision needs all T pre-activations of its channel: H[t'] = b[t'] + sum_t W[t'][t] x_t
// is accumulated on the fly - step t adds W[.][t] x_t to the T accumulators of every channel, the k-ordered fmaf chain of neuron.hip -
// one 32-channel block at a time (16 channels x T accumulators = 160 registers per lane at T = 10): per block the T steps' MFMAs, then
// the decisions.  The matrix work is the same as the LIF form's (each block's products are issued once); the voxel is read once per
// block (L1 / L2 hits the second time).  W is staged transposed in LDS (column t contiguous), read as broadcasts.
template <int T, int CIN, int NOUT>
__global__ __launch_bounds__(256) void head_conv_mfma_psn_kernel(HeadParams P, int tiles) {
  constexpr int KS = 9 * CIN / 2, NBLK = (NOUT + 31) / 32, ROW = 32 * NOUT, TP = sdfmm::PSN_TP(T);
  const SdfHeadConvDesc& d = P.d;
  __shared__ __attribute__((aligned(16))) uint8_t sp_s[4][T][ROW];
  __shared__ __attribute__((aligned(16))) float par_s[2][32 * NBLK];
  __shared__ __attribute__((aligned(16))) float wt_s[T * TP + TP];                     // wt_s[t * TP + t'] = W[t'][t]; then the biases
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  if (tid < 32 * NBLK) {
    par_s[0][tid] = (tid < NOUT && d.alpha) ? d.alpha[tid] : 1.f;
    par_s[1][tid] = (tid < NOUT && d.alpha) ? d.beta[tid] : 0.f;
  }
  for (int i = tid; i < T * TP + TP; i += 256) {
    const int r = i / TP, k = i - r * TP;
    wt_s[i] = r < T ? (k < T ? d.psn_w[k * T + r] : 0.f) : (k < T ? d.psn_b[k] : 0.f);
  }
  float wa[NBLK][KS];
#pragma unroll
  for (int blk = 0; blk < NBLK; ++blk)
#pragma unroll
    for (int s_ = 0; s_ < KS; ++s_) {
      const int k = 2 * s_ + lh, tap = k / CIN, ci = k - tap * CIN, n = 32 * blk + l31;
      wa[blk][s_] = n < NOUT ? d.w[(n * CIN + ci) * 9 + tap] : 0.f;
    }
  __syncthreads();
  const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x), 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t o_rs = __builtin_amdgcn_make_buffer_rsrc(d.out, 0, 0x7FFFFFFF, 0x00020000);
  const int xt = d.W >> 5;
  uint8_t* my_s = &sp_s[wave][0][0];

#pragma unroll 1
  for (int tile = blockIdx.x * 4 + wave; tile < tiles; tile += gridDim.x * 4) {
    const int x0 = (tile % xt) << 5, y = (tile / xt) % d.H, b = tile / (xt * d.H);
    uint32_t off[KS];
#pragma unroll
    for (int s_ = 0; s_ < KS; ++s_) {
      const int k = 2 * s_ + lh, tap = k / CIN, ci = k - tap * CIN;
      const int yy = y + tap / 3 - 1, xx = x0 + l31 + tap % 3 - 1;
      const bool ok = (unsigned)yy < (unsigned)d.H && (unsigned)xx < (unsigned)d.W;
      off[s_] = ok ? (uint32_t)(((int64_t)b * d.x_sb + (int64_t)yy * d.x_sy + (int64_t)xx * d.x_sx + d.x_sc[ci]) * 4) : 0x80000000u;
    }
    const uint32_t st4 = (uint32_t)(d.x_st * 4);
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk) {
      float hacc[T][16];
#pragma unroll
      for (int t2 = 0; t2 < T; ++t2) {
        const float bt = wt_s[T * TP + t2];
#pragma unroll
        for (int e = 0; e < 16; ++e) hacc[t2][e] = bt;
      }
      float xin[2][KS];
#pragma unroll
      for (int s_ = 0; s_ < KS; ++s_) xin[0][s_] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rs, off[s_], 0, 0));
      auto one_step = [&](int t, const float (&xc)[KS], float (&xn)[KS]) __attribute__((always_inline)) {
        if (t + 1 < T) {
#pragma unroll
          for (int s_ = 0; s_ < KS; ++s_)
            xn[s_] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rs, off[s_], (uint32_t)(t + 1) * st4, 0));
        }
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[blk][s_], xc[s_], acc, 0, 0, 0);
        float xv[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 al = *reinterpret_cast<const float4*>(&par_s[0][32 * blk + 8 * q + 4 * lh]);
          const float4 be = *reinterpret_cast<const float4*>(&par_s[1][32 * blk + 8 * q + 4 * lh]);
          xv[4 * q + 0] = __builtin_fmaf(acc[4 * q + 0], al.x, be.x); xv[4 * q + 1] = __builtin_fmaf(acc[4 * q + 1], al.y, be.y);
          xv[4 * q + 2] = __builtin_fmaf(acc[4 * q + 2], al.z, be.z); xv[4 * q + 3] = __builtin_fmaf(acc[4 * q + 3], al.w, be.w);
        }
        float wc[TP];                                                   // column t of W: W[t'][t], t' = 0..T-1
#pragma unroll
        for (int k4 = 0; k4 < TP / 4; ++k4) {
          const float4 q = *reinterpret_cast<const float4*>(&wt_s[t * TP + 4 * k4]);
          wc[4 * k4] = q.x; wc[4 * k4 + 1] = q.y; wc[4 * k4 + 2] = q.z; wc[4 * k4 + 3] = q.w;
        }
#pragma unroll
        for (int t2 = 0; t2 < T; ++t2)
#pragma unroll
          for (int e = 0; e < 16; ++e) hacc[t2][e] = __builtin_fmaf(wc[t2], xv[e], hacc[t2][e]);
      };
#pragma unroll 1
      for (int t = 0; t < T; t += 2) {
        one_step(t, xin[0], xin[1]);
        if (t + 1 < T) one_step(t + 1, xin[1], xin[0]);
      }
      // decisions of this block: channel 32 blk + 8 q + 4 lh + i of pixel l31, all T steps
#pragma unroll
      for (int t2 = 0; t2 < T; ++t2)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (32 * blk + 8 * q >= NOUT) continue;
          uint32_t word = 0;
#pragma unroll
          for (int i = 0; i < 4; ++i) word |= (hacc[t2][4 * q + i] >= 0.f ? 1u : 0u) << (8 * i);
          *reinterpret_cast<uint32_t*>(my_s + t2 * ROW + l31 * NOUT + 4 * lh + 32 * blk + 8 * q) = word;
        }
    }
    constexpr int PT = ROW / 16, PIECES = T * PT;
    const uint32_t obase = (uint32_t)(((((int64_t)b * T) * d.H + y) * d.W + x0) * NOUT), ostep = (uint32_t)(d.H * d.W * NOUT);
#pragma unroll
    for (int j = 0; j < (PIECES + 63) / 64; ++j) {
      const int pc = lane + 64 * j;
      const int tt = pc / PT, o = (pc - tt * PT) * 16;
      const u32x4h val = *reinterpret_cast<const u32x4h*>(my_s + (pc < PIECES ? tt * ROW + o : 0));
      __builtin_amdgcn_raw_buffer_store_b128(val, o_rs, pc < PIECES ? obase + (uint32_t)tt * ostep + (uint32_t)o : 0x80000000u, 0, 0);
    }
  }
}

__global__ __launch_bounds__(256) void flow_out_kernel(const float* __restrict__ pred, float* __restrict__ out, int B, int D,
                                                       int h, int w, int64_t ldp, int C, int H, int W, float sy, float sx) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)B * C * H * W) return;
  const int X = (int)(i % W), Y = (int)((i / W) % H), c = (int)((i / ((int64_t)W * H)) % C), b = (int)(i / ((int64_t)W * H * C));
  // nearest source index exactly as ATen computes it for a given scale_factor: min(floor(dst * (1 / scale)), in - 1)
  int ys = (int)floorf((float)Y * sy), xsrc = (int)floorf((float)X * sx);
  ys = ys < h - 1 ? ys : h - 1;
  xsrc = xsrc < w - 1 ? xsrc : w - 1;
  float s = 0.f;
  for (int t = 0; t < D; ++t) s += pred[((((int64_t)b * D + t) * h + ys) * w + xsrc) * ldp + c];
  out[i] = s;
}

// col2im of a stride-2 3x3 transposed convolution whose nine per-tap products Y[(img, iy, ix)][tap][co] came out of ONE plain
// spike GEMM over the stacked tap weights: out[img, oy, ox, co] = BN( sum of the 1, 2 or 4 taps that reach (oy, ox) ),
// oy = 2 iy - 1 + ky.  Taps are added in a fixed order (ky = 2 before ky = 0, kx likewise): deterministic.
__global__ __launch_bounds__(256) void deconv_col2im_kernel(const float* __restrict__ Y, const float* __restrict__ alpha,
                                                            const float* __restrict__ beta, float* __restrict__ out, int imgs, int H,
                                                            int W, int Cout) {
  const int cq = Cout / 4;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)imgs * 4 * H * W * cq) return;
  const int c4 = (int)(i % cq) * 4;
  const int64_t pix = i / cq;
  const int ox = (int)(pix % (2 * W)), oy = (int)((pix / (2 * W)) % (2 * H)), img = (int)(pix / ((int64_t)4 * H * W));
  const int y0 = oy >> 1, x0 = ox >> 1;
  // (input row, kernel row) pairs reaching oy: even -> (y0, 1); odd -> (y0, 2), (y0 + 1, 0)
  const int ny = (oy & 1) ? ((y0 + 1 < H) ? 2 : 1) : 1, nx = (ox & 1) ? ((x0 + 1 < W) ? 2 : 1) : 1;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int a = 0; a < ny; ++a) {
    const int iy = y0 + a, ky = (oy & 1) ? (a == 0 ? 2 : 0) : 1;
    for (int b = 0; b < nx; ++b) {
      const int ix = x0 + b, kx = (ox & 1) ? (b == 0 ? 2 : 0) : 1;
      const float4 v = *reinterpret_cast<const float4*>(Y + (((int64_t)img * H + iy) * W + ix) * (9 * (int64_t)Cout) + (ky * 3 + kx) * Cout + c4);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  if (alpha) {
    const float4 al = *reinterpret_cast<const float4*>(alpha + c4), be = *reinterpret_cast<const float4*>(beta + c4);
    acc.x = __builtin_fmaf(acc.x, al.x, be.x); acc.y = __builtin_fmaf(acc.y, al.y, be.y);
    acc.z = __builtin_fmaf(acc.z, al.z, be.z); acc.w = __builtin_fmaf(acc.w, al.w, be.w);
  }
  *reinterpret_cast<float4*>(out + pix * Cout + c4) = acc;
}

template <int T>
int launch_head_mfma(const HeadParams& P, hipStream_t s) {
  const int tiles = P.d.B * P.d.H * (P.d.W / 32);
  const int want = (tiles + 3) / 4;
  dim3 grid((unsigned)(want < 768 ? want : 768));                  // three workgroups per CU are resident; the rest of the tiles loop
  if (P.d.sn_kind == SDF_PSN) {
    if constexpr (T <= 10) {
      if (P.d.Cin == 2 && P.d.Cout == 48) { SDF_LAUNCH((head_conv_mfma_psn_kernel<T, 2, 48>), grid, dim3(256), 0, s, P, tiles); return 0; }
      if (P.d.Cin == 2 && P.d.Cout == 32) { SDF_LAUNCH((head_conv_mfma_psn_kernel<T, 2, 32>), grid, dim3(256), 0, s, P, tiles); return 0; }
    }
    return SDF_E_SHAPE;
  }
  const bool fast = P.d.sn_kind == SDF_LIF && P.d.soft_reset != 0 && P.inv_tau != 0.f;
#define SDF_HEAD_CASE(CI, CO)                                                                                              \
  if (P.d.Cin == CI && P.d.Cout == CO) {                                                                                   \
    if (fast) SDF_LAUNCH((head_conv_mfma_kernel<T, CI, CO, true>), grid, dim3(256), 0, s, P, tiles);               \
    else SDF_LAUNCH((head_conv_mfma_kernel<T, CI, CO, false>), grid, dim3(256), 0, s, P, tiles);                   \
    return 0;                                                                                                              \
  }
  SDF_HEAD_CASE(2, 48) SDF_HEAD_CASE(2, 32) SDF_HEAD_CASE(2, 64) SDF_HEAD_CASE(4, 48)
#undef SDF_HEAD_CASE
  return SDF_E_SHAPE;
}

template <int T>
int launch_head(const HeadParams& P, dim3 grid, hipStream_t s) {
  if (P.d.Cin == 2 && P.d.Cout == 48) { SDF_LAUNCH((head_conv_sn_kernel<T, 3, 2>), grid, dim3(256), 0, s, P); return 0; }
  if (P.d.Cin == 2 && P.d.Cout == 32) { SDF_LAUNCH((head_conv_sn_kernel<T, 2, 2>), grid, dim3(256), 0, s, P); return 0; }
  if (P.d.Cin == 2 && P.d.Cout == 64) { SDF_LAUNCH((head_conv_sn_kernel<T, 4, 2>), grid, dim3(256), 0, s, P); return 0; }
  if (P.d.Cin == 4 && P.d.Cout == 48) { SDF_LAUNCH((head_conv_sn_kernel<T, 3, 4>), grid, dim3(256), 0, s, P); return 0; }
  return SDF_E_SHAPE;
}
}  // namespace

extern "C" int sdf_head_conv_sn_fwd(const SdfHeadConvDesc* d, void* stream) {
  if (!d || !d->x || !d->w || !d->out) return SDF_E_NULL;
  if (d->B < 1 || d->H < 1 || d->W < 1 || d->W % 16) return SDF_E_SHAPE;      // whole 16-pixel tiles (16-byte spike stores)
  if (d->alpha && !d->beta) return SDF_E_NULL;
  if (d->sn_kind != SDF_LIF && d->sn_kind != SDF_PSN && d->sn_kind != SDF_IF) return SDF_E_DTYPE;
  if (d->sn_kind == SDF_PSN && (!d->psn_w || !d->psn_b)) return SDF_E_NULL;
  if (!sdf_tau_ok(d->sn_kind, d->tau)) return SDF_E_SHAPE;
  if (!sdf_aligned(d->out, 16)) return SDF_E_ALIGN;
  HeadParams P;
  P.d = *d;
  P.inv_tau = sdf_inv_tau(d->sn_kind, d->tau);
  const int64_t tiles = (int64_t)d->B * d->H * (d->W / 16);
  if (tiles >= (1LL << 31)) return SDF_E_SHAPE;
  dim3 grid((unsigned)tiles);
  hipStream_t s = sdf_stream(stream);
  if (d->x_sy == 0) {                                                  // the packed NHWC layout as strides
    P.d.x_sb = (int64_t)d->T * d->H * d->W * d->Cin; P.d.x_st = (int64_t)d->H * d->W * d->Cin;
    P.d.x_sy = (int64_t)d->W * d->Cin; P.d.x_sx = d->Cin;
    for (int ci = 0; ci < 4; ++ci) P.d.x_sc[ci] = ci;
  }
  // fp32 matrix-pipe kernel: LIF / IF, 32-pixel tiles, 31-bit byte offsets into the voxel
  const char* e_hm = sdf_sw(SW_HEAD_MFMA);                        // (read per call)
  const bool no_mfma = e_hm && e_hm[0] == '0';
  int64_t span = (int64_t)(d->B - 1) * P.d.x_sb + (int64_t)(d->T - 1) * P.d.x_st + (int64_t)(d->H - 1) * P.d.x_sy + (int64_t)(d->W - 1) * P.d.x_sx;
  int64_t scmax = 0;
  for (int ci = 0; ci < d->Cin && ci < 4; ++ci) scmax = P.d.x_sc[ci] > scmax ? P.d.x_sc[ci] : scmax;
  const bool psn_mfma = d->sn_kind == SDF_PSN && d->T <= 10 && d->Cin == 2 && (d->Cout == 48 || d->Cout == 32);
  const bool mfma_ok = !no_mfma && (d->sn_kind != SDF_PSN || psn_mfma) && d->W % 32 == 0 && (span + scmax + 1) * 4 < (1LL << 31) && d->Cin <= 4 &&
                       (int64_t)d->B * d->T * d->H * d->W * d->Cout < (1LL << 31) &&
                       P.d.x_sb >= 0 && P.d.x_st >= 0 && P.d.x_sy >= 0 && P.d.x_sx >= 0;
  int rc;
  if (mfma_ok) {
    switch (d->T) {
      case 5: rc = launch_head_mfma<5>(P, s); break;
      case 10: rc = launch_head_mfma<10>(P, s); break;
      case 20: rc = launch_head_mfma<20>(P, s); break;
      default: rc = SDF_E_SHAPE;
    }
    if (rc) return rc;
    SDF_LAUNCH_CHECK();
    return 0;
  }
  P.d = *d;
  switch (d->T) {
    case 5: rc = launch_head<5>(P, grid, s); break;
    case 10: rc = launch_head<10>(P, grid, s); break;
    case 20: rc = launch_head<20>(P, grid, s); break;
    default: rc = SDF_E_SHAPE;
  }
  if (rc) return rc;
  SDF_LAUNCH_CHECK();
  return 0;
}

extern "C" int sdf_flow_out_fwd(const float* pred, float* out, int B, int D, int h, int w, int64_t ldp, int C, int H, int W,
                                float scale_y, float scale_x, void* stream) {
  if (!pred || !out) return SDF_E_NULL;
  if (B < 1 || D < 1 || h < 1 || w < 1 || C < 1 || H < 1 || W < 1 || ldp < C || !(scale_y > 0.f) || !(scale_x > 0.f)) return SDF_E_SHAPE;
  const int64_t n = (int64_t)B * C * H * W;
  SDF_LAUNCH(flow_out_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, sdf_stream(stream), pred, out, B, D, h, w,
                     ldp, C, H, W, 1.0f / scale_y, 1.0f / scale_x);
  SDF_LAUNCH_CHECK();
  return 0;
}

extern "C" int sdf_deconv_col2im_fwd(const float* Y, const float* alpha, const float* beta, float* out, int imgs, int H, int W,
                                     int Cout, void* stream) {
  if (!Y || !out) return SDF_E_NULL;
  if (alpha && !beta) return SDF_E_NULL;
  if (imgs < 1 || H < 1 || W < 1 || Cout < 4 || Cout % 4) return SDF_E_SHAPE;
  if (!sdf_aligned(Y, 16) || !sdf_aligned(out, 16) || (alpha && (!sdf_aligned(alpha, 16) || !sdf_aligned(beta, 16)))) return SDF_E_ALIGN;
  const int64_t n = (int64_t)imgs * 4 * H * W * (Cout / 4);
  if ((n + 255) / 256 >= (1LL << 31)) return SDF_E_SHAPE;
  SDF_LAUNCH(deconv_col2im_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, sdf_stream(stream), Y, alpha, beta, out,
                     imgs, H, W, Cout);
  SDF_LAUNCH_CHECK();
  return 0;
}
