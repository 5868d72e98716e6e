// The attention half of an ANN video-swin block as ONE launch (gfx950):
//
//   out = x + proj( softmax( normalize(q) normalize(k)^T * logit_scale + bias (+ mask) ) v ),   q | k | v = LayerNorm(x) Wqkv^T + b
//
// over the (shifted) 3-D windows of x - reference models/STSwinNet/swin_transformer3D_v2.py:272-310 (`forward_part1`: norm1, pad,
// roll, window_partition, the attention, window_reverse, roll back, crop) with :169-205 (`WindowAttention3D.forward`) inside and
// the shortcut of :331.  Four launches before (LayerNorm, the qkv Linear, win_attn.hip, the proj Linear): q | k | v made a round trip
// through HBM - 3 C floats per token written and read back, 131 + 131 MB on BASELINE config 3's first stage - and that round
// trip, not the matrix pipe, bounded the attention kernel (profiles/r5x_*).  Here a workgroup owns one window and nothing but x
// and out touches HBM:
//
//   * 12 wavefronts; wave w < 11 owns the 16 tokens 16 w .. 16 w + 15 of the window (N = 162: 11 tiles) for the whole kernel.
//     It reads their rows of x through the window's slice map (pad / roll / partition are this lookup), normalises them
//     (LayerNorm in registers: a token's 96 channels are spread over the 4 lanes l, l+16, l+32, l+48) and keeps LN(x) as the
//     fp16 hi / lo MFMA operand of the three projections - the activations never exist in memory in any other form;
//   * per head: the head's 96 rows of Wqkv (q, k, v: 32 each; fp16 hi / lo planes, packed once on the host) are staged in LDS;
//     every wave computes Q^T, K^T (weights as the row operand: a lane ends with 8 dims of ITS token - after the cosine
//     normalisation exactly the K Q^T operand layout) and V (tokens as rows: a lane ends with 4 consecutive keys of one dim -
//     exactly a unit of the transposed V image) of its 16 tokens, 3 x v_mfma_f32_16x16x32_f16 per product (hi*hi + hi*lo +
//     lo*hi, fp32 accumulation: 22 significant bits, the numerics of dense_linear.hip); K and V go to LDS, Q stays in registers;
//   * the attention of the wave's 16 queries against the window's keys as in win_attn_tiled_f16_kernel (scores in the log2
//     domain straight out of the matrix pipe - the accumulator starts as the row of the (bias + mask) * log2 e table the host
//     combined once, requested ahead of the projections - unnormalised probabilities, P V on the 16-bit pipe), but as O^T = V^T P^T: a lane
//     ends with 8 dims of ITS query - divided by the row sum and split, that is the column operand of the output projection,
//     whose accumulators (16 tokens x 96 channels per wave) collect the heads one by one;
//   * the projection weights sit in LDS for the life of the workgroup with their input channels in the order the O^T
//     accumulators leave them (permuted once on the host); the epilogue adds the projection bias and the shortcut x and
//     writes the rows back through the slice map (padding tokens have no row: dropped, as the reference crops them).
//
// LDS: qkv weights of one head 42 KB + projection weights 42 KB + K hi / lo 33 KB + V^T 22 KB = 139 KB, one workgroup (three
// wavefronts per SIMD) per compute unit.  Built for C = 96 (three heads of 32) and windows of 162 tokens - BASELINE config 3's
// first stage, where 83 % of the model's window-attention tokens are; other shapes keep the four-launch path.
// Compiled with -ffp-contract=off.
#include "common.h"

#ifdef SDF_STAMP
// diagnostic build only (tools/stamp_ann_block.sh): 100 MHz timestamps of wave 0 of three workgroups along the kernel's phases
__device__ unsigned long long g_ann_stamp[3 * 24];
#define STAMP(i) do { if (st_on) st_buf[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define STAMP(i)
#endif

namespace {

constexpr int C = 96, NH = 3, HD = 32, N = 162, NTC = 11, NP = NTC * 16;
constexpr int NWAVE = 12, NTHR = 64 * NWAVE;
// row pitches for (row = lane % 16, 16-byte piece = lane / 16) fragment reads: ds_read_b128 is served in four NON-contiguous 16-lane
// groups ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS) - 14 pieces per 12-piece row and 6 per 4-piece row are conflict-free for
// them (brute force over pitches; the "4 x odd dwords" pitches 13 and 5 of win_attn.hip's first layout cost 38 % extra LDS cycles)
constexpr int WPB = 2 * C + 32;                 // weight row pitch in bytes (fp16)
constexpr int KRS = 96;                         // K row pitch in bytes: 32 fp16 + 32 pad
constexpr int WG_BYTES = 2 * 96 * WPB;          // one head's q | k | v rows, hi then lo plane
constexpr int WP_BYTES = 2 * C * WPB;           // projection rows, hi then lo plane
constexpr int K_BYTES = 2 * NP * KRS;
constexpr int V_BYTES = (NP / 4) * HD * 16;
constexpr int LDS_BYTES = WG_BYTES + WP_BYTES + K_BYTES + V_BYTES;
constexpr uint32_t INV_OFF = 0x80000000u;

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

struct BlockParams {
  SdfAnnAttnBlockDesc d;
};

// two fp32 -> their hi and lo fp16 halves, packed (win_attn.hip: split2_f16)
__device__ __forceinline__ void split2(float x, float y, uint32_t& hi, uint32_t& lo) {
  const f32x2 v = {x, y};
  const f16x2 h = __builtin_convertvector(v, f16x2);
  const f32x2 r = v - __builtin_convertvector(h, f32x2);
  const f16x2 l = __builtin_convertvector(r, f16x2);
  hi = __builtin_bit_cast(uint32_t, h);
  lo = __builtin_bit_cast(uint32_t, l);
}
// eight fp32 -> one hi and one lo operand of v_mfma_f32_16x16x32_f16
__device__ __forceinline__ void split8(const float (&x)[8], f16x8& hi, f16x8& lo) {
  uint32_t h[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) split2(x[2 * i], x[2 * i + 1], h[i], l[i]);
  hi = __builtin_bit_cast(f16x8, u32x4{h[0], h[1], h[2], h[3]});
  lo = __builtin_bit_cast(f16x8, u32x4{l[0], l[1], l[2], l[3]});
}
// a += A_hi B_hi + A_hi B_lo + A_lo B_hi (the smallest products first)
__device__ __forceinline__ f32x4 mma3(const f16x8& ah, const f16x8& al, const f16x8& bh, const f16x8& bl, f32x4 a) {
  a = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, a, 0, 0, 0);
  a = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, a, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, a, 0, 0, 0);
}

__global__ __launch_bounds__(NTHR) void ann_attn_block_kernel(BlockParams P) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const SdfAnnAttnBlockDesc& d = P.d;
  uint8_t* Wg = smem;                            // [hi | lo][96 rows: q 0..31, k 32..63, v 64..95][WPB]
  uint8_t* Wp = Wg + WG_BYTES;                   // [hi | lo][96 out channels][WPB], input channels in accumulator order
  uint8_t* Khi = Wp + WP_BYTES;                  // [NP][KRS], dims in accumulator order
  uint8_t* Klo = Khi + NP * KRS;
  uint8_t* Vt = Klo + NP * KRS;                  // [NP / 4][HD][16 B]: {4 keys hi, 4 keys lo}

  // workgroup -> window: the windows that share a mask table are neighbours on one XCD (win_attn_tiled_kernel)
  const int total = gridDim.x, per_xcd = total >> 3;
  int L = blockIdx.x;
  if (L < per_xcd * 8) L = (L & 7) * per_xcd + (L >> 3);
  const int nW = d.nW;
  const int share = d.B_ / nW;
  const int wmask = L / share, b = (L - wmask * share) * nW + wmask;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lg = lane >> 4;
  const bool worker = wave < NTC;                // the twelfth wave only helps to stage weights
#ifdef SDF_STAMP
  const int st_slot = blockIdx.x == 0 ? 0 : (blockIdx.x == 300 ? 1 : (blockIdx.x == 600 ? 2 : -1));
  const bool st_on = st_slot >= 0 && tid == 0;
  unsigned long long st_buf[24];
  for (int i = 0; i < 24; ++i) st_buf[i] = 0;
#endif
  STAMP(0);

  const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x), 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t o_rs = __builtin_amdgcn_make_buffer_rsrc(d.out, 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t wq_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(d.wqkv), 0, 2 * 3 * C * C * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t wp_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(d.wproj), 0, 2 * C * C * 2, 0x00020000);

  // ---- requests first, in the order of their dependency depth: the slice-map entry of this lane's token (the rows of x hang on it),
  // the projection weights (2 planes x 96 rows x 12 pieces of 16 bytes = 3 per thread, stored to LDS behind the x requests) ----
  const int tok = wave * 16 + l15;                                  // token of this lane in the window
  int row = -1;
  if (worker && tok < N) row = d.row_map[(int64_t)b * N + tok];
  u32x4 wpv[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int i = tid + NTHR * j, prow = i / 12, pc = i - prow * 12;   // prow = plane * 96 + out channel
    wpv[j] = __builtin_amdgcn_raw_buffer_load_b128(wp_rs, (uint32_t)(prow * C * 2 + pc * 16), 0, 0);
  }
  // the LayerNorm's weight / bias of this lane's 24 channels: requested now - behind the rows of x each of the three channel groups
  // paid its own L2 round trip (stamps: 3 - 4 us of a window's 37)
  float4 lng[3][2], lnb[3][2];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    lng[s][0] = *reinterpret_cast<const float4*>(d.ln_w + 32 * s + 8 * lg); lng[s][1] = *reinterpret_cast<const float4*>(d.ln_w + 32 * s + 8 * lg + 4);
    lnb[s][0] = *reinterpret_cast<const float4*>(d.ln_b + 32 * s + 8 * lg); lnb[s][1] = *reinterpret_cast<const float4*>(d.ln_b + 32 * s + 8 * lg + 4);
  }
  // ---- this wave's 16 tokens: rows of x -> LayerNorm -> fp16 hi / lo operand (channels 32 s + 8 lg + 0..7 of token l15) ----
  f16x8 yh[3], yl[3];
  {
    float xv[3][8];
    const uint32_t xo = row >= 0 ? (uint32_t)row * (uint32_t)(C * 4) + (uint32_t)(32 * lg) : INV_OFF;
#ifdef SDF_STAMP
    asm volatile("" ::"v"(xo));
    STAMP(18);
#endif
    u32x4 xa[3], xc[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      xa[s] = __builtin_amdgcn_raw_buffer_load_b128(x_rs, xo, s * 128, 0);
      xc[s] = __builtin_amdgcn_raw_buffer_load_b128(x_rs, xo, s * 128 + 16, 0);
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const u32x4 a = xa[s], c = xc[s];
      xv[s][0] = __uint_as_float(a.x); xv[s][1] = __uint_as_float(a.y); xv[s][2] = __uint_as_float(a.z); xv[s][3] = __uint_as_float(a.w);
      xv[s][4] = __uint_as_float(c.x); xv[s][5] = __uint_as_float(c.y); xv[s][6] = __uint_as_float(c.z); xv[s][7] = __uint_as_float(c.w);
    }
#ifdef SDF_STAMP
    STAMP(19);
    asm volatile("" ::"v"(xv[2][7]));
    STAMP(20);
#endif
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int i = 0; i < 8; ++i) sum += xv[s][i];
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float mean = sum * (1.f / C);
    float sq = 0.f;
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float t = xv[s][i] - mean;
        sq += t * t;
      }
    sq += __shfl_xor(sq, 16);
    sq += __shfl_xor(sq, 32);
    const float rstd = 1.f / sqrtf(sq * (1.f / C) + d.ln_eps);      // biased variance (torch.nn.LayerNorm)
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const float4 g0 = lng[s][0], g1 = lng[s][1], b0 = lnb[s][0], b1 = lnb[s][1];
      const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      float y[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) y[i] = row >= 0 ? (xv[s][i] - mean) * rstd * gg[i] + bb[i] : 0.f;   // a padding token is a zero row BEHIND the norm (:283-284)
      split8(y, yh[s], yl[s]);
    }
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {                                     // the projection weights (long since arrived) into LDS
    const int i = tid + NTHR * j, prow = i / 12, pc = i - prow * 12;
    *reinterpret_cast<u32x4*>(Wp + prow * WPB + pc * 16) = wpv[j];
  }

#ifdef SDF_STAMP
  asm volatile("" ::"v"(yl[2]));
  STAMP(21);
#endif
  f32x4 pacc[6];                                                    // out^T: channels 16 ct + 4 lg + r of token l15
#pragma unroll
  for (int ct = 0; ct < 6; ++ct) pacc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};

  const uint32_t tbytes = (uint32_t)N * (uint32_t)N * 4u;
  const uint32_t rowoff = (uint32_t)tok * (uint32_t)N * 4u;
  const uint32_t rbase = (worker && tok < N) ? rowoff + 16u * (uint32_t)lg : INV_OFF;

  // the head's q | k | v rows: 2 planes x 96 rows x 12 pieces of 16 bytes = 3 pieces per thread, requested one head ahead (during
  // the attention of the head before) and written to LDS between two barriers - no wave ever waits for a weight load
  u32x4 wpre[3];
  auto request_head = [&](int g) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int i = tid + NTHR * j;
      const int prow = i / 12, pc = i - prow * 12;                  // prow = plane * 96 + (0..31 q, 32..63 k, 64..95 v)
      const int pl = prow / 96, r = prow - pl * 96;
      const int grow = pl * 3 * C + (r >> 5) * C + g * HD + (r & 31);
      wpre[j] = __builtin_amdgcn_raw_buffer_load_b128(wq_rs, (uint32_t)(grow * C * 2 + pc * 16), 0, 0);
    }
  };
  request_head(0);
  STAMP(1);

#pragma unroll 1
  for (int g = 0; g < NH; ++g) {
    // the score strip of this wave's queries starts as the head's table row ((bias + mask) * log2 e, combined once on the host):
    // requested here, ahead of the projections that hide its latency; keys beyond N start at -inf
    f32x4 st[NTC];
    {
      const __amdgpu_buffer_rsrc_t tab_rs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(d.table + ((int64_t)wmask * NH + g) * N * N), 0, (int)tbytes, 0x00020000);
#pragma unroll
      for (int jt = 0; jt < NTC; ++jt) {
        if (jt < NTC - 1) {
          const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(tab_rs, rbase, jt * 64, 0);
          st[jt] = f32x4{__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w)};
        } else {                                                    // N % 4 == 2: the last piece of a row is 8 bytes
          const int kb = jt * 16 + 4 * lg;
          const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(tab_rs, (rbase != INV_OFF && kb < N) ? rowoff + (uint32_t)kb * 4u : INV_OFF, 0, 0);
          st[jt] = f32x4{kb + 0 < N ? __uint_as_float(t.x) : -INFINITY, kb + 1 < N ? __uint_as_float(t.y) : -INFINITY, -INFINITY, -INFINITY};
        }
      }
    }
    __syncthreads();                                                // every wave is done with the previous head's weights, K and V
    STAMP(2 + 5 * g);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int i = tid + NTHR * j;
      const int prow = i / 12, pc = i - prow * 12;
      *reinterpret_cast<u32x4*>(Wg + prow * WPB + pc * 16) = wpre[j];
    }
    __syncthreads();
    STAMP(3 + 5 * g);
    f16x8 q_hi, q_lo;
    if (worker) {
      // ---- Q^T and K^T of the wave's tokens: weights as rows, a lane ends with dims 4 lg + r (dt = 0) and 16 + 4 lg + r (dt = 1) ----
      float qv[8], kv[8];
#pragma unroll
      for (int which = 0; which < 2; ++which) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          if (d.qkv_bias) acc = *reinterpret_cast<const f32x4*>(d.qkv_bias + which * C + g * HD + 16 * dt + 4 * lg);
          const uint8_t* wr = Wg + (32 * which + 16 * dt + l15) * WPB + 16 * lg;
#pragma unroll
          for (int s = 0; s < 3; ++s) {
            const f16x8 wh = *reinterpret_cast<const f16x8*>(wr + 64 * s), wl = *reinterpret_cast<const f16x8*>(wr + 96 * WPB + 64 * s);
            acc = mma3(wh, wl, yh[s], yl[s], acc);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) (which ? kv : qv)[4 * dt + r] = acc[r];
        }
      }
      // F.normalize(., dim=-1) (:179); q also takes the head's logit scale and log2 e: the scores land in the softmax's log2 domain
      float sq = 0.f, sk = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) { sq += qv[i] * qv[i]; sk += kv[i] * kv[i]; }
      sq += __shfl_xor(sq, 16); sk += __shfl_xor(sk, 16);
      sq += __shfl_xor(sq, 32); sk += __shfl_xor(sk, 32);
      const float iq = d.scale[g] / fmaxf(sqrtf(sq), 1e-12f);        // (scale: the head's logit scale * log2 e)
      const float ik = tok < N ? 1.f / fmaxf(sqrtf(sk), 1e-12f) : 0.f;   // rows beyond the window: zero keys
#pragma unroll
      for (int i = 0; i < 8; ++i) { qv[i] *= iq; kv[i] *= ik; }
      split8(qv, q_hi, q_lo);
      f16x8 k_hi, k_lo;
      split8(kv, k_hi, k_lo);
      *reinterpret_cast<f16x8*>(Khi + tok * KRS + 16 * lg) = k_hi;
      *reinterpret_cast<f16x8*>(Klo + tok * KRS + 16 * lg) = k_lo;
      // ---- V: tokens as rows, a lane ends with keys 16 w + 4 lg + 0..3 of dim 16 dt + l15: one unit of the transposed image ----
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (d.qkv_bias) {
          const float bv = d.qkv_bias[2 * C + g * HD + 16 * dt + l15];
          acc = f32x4{bv, bv, bv, bv};
        }
        const uint8_t* wr = Wg + (64 + 16 * dt + l15) * WPB + 16 * lg;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          const f16x8 wh = *reinterpret_cast<const f16x8*>(wr + 64 * s), wl = *reinterpret_cast<const f16x8*>(wr + 96 * WPB + 64 * s);
          acc = mma3(yh[s], yl[s], wh, wl, acc);
        }
        const int k0 = wave * 16 + 4 * lg;
        uint32_t h01, l01, h23, l23;
        split2(k0 + 0 < N ? acc[0] : 0.f, k0 + 1 < N ? acc[1] : 0.f, h01, l01);
        split2(k0 + 2 < N ? acc[2] : 0.f, k0 + 3 < N ? acc[3] : 0.f, h23, l23);
        *reinterpret_cast<u32x4*>(Vt + (((4 * wave + lg) * HD) + 16 * dt + l15) * 16) = u32x4{h01, h23, l01, l23};
      }
    }
    STAMP(4 + 5 * g);
    __syncthreads();                                                // K and V of the window are complete
    STAMP(5 + 5 * g);
    if (g + 1 < NH) request_head(g + 1);
    if (worker && wave * 16 < N) {
      // ---- S^T = K Q^T on top of the table row: the scores land in the softmax's log2 domain ----
#pragma unroll
      for (int jt = 0; jt < NTC; ++jt) {
        const int kj = jt * 16 + l15;
        const f16x8 k_hi = *reinterpret_cast<const f16x8*>(Khi + kj * KRS + 16 * lg);
        const f16x8 k_lo = *reinterpret_cast<const f16x8*>(Klo + kj * KRS + 16 * lg);
        st[jt] = mma3(k_hi, k_lo, q_hi, q_lo, st[jt]);
      }
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
      // ---- O^T = V^T P^T, probabilities unnormalised (e in (0, 1]); a lane ends with dims 4 lg + r | 16 + 4 lg + r of query l15 ----
      f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int jt = 0; jt < NTC; ++jt) {
        const f32x2 a01 = f32x2{st[jt][0], st[jt][1]} - m2, a23 = f32x2{st[jt][2], st[jt][3]} - m2;
        const f32x2 e01 = {__builtin_amdgcn_exp2f(a01.x), __builtin_amdgcn_exp2f(a01.y)};
        const f32x2 e23 = {__builtin_amdgcn_exp2f(a23.x), __builtin_amdgcn_exp2f(a23.y)};
        sum2 += e01; sum2 += e23;
        uint32_t ph0, pl0, ph1, pl1;
        split2(e01.x, e01.y, ph0, pl0);
        split2(e23.x, e23.y, ph1, pl1);
        const f16x4 p_hi = __builtin_bit_cast(f16x4, u32x2{ph0, ph1}), p_lo = __builtin_bit_cast(f16x4, u32x2{pl0, pl1});
        const u32x4 v0 = *reinterpret_cast<const u32x4*>(Vt + (((4 * jt + lg) * HD) + l15) * 16);
        const u32x4 v1 = *reinterpret_cast<const u32x4*>(Vt + (((4 * jt + lg) * HD) + 16 + l15) * 16);
        const f16x4 v0h = __builtin_bit_cast(f16x4, u32x2{v0.x, v0.y}), v0l = __builtin_bit_cast(f16x4, u32x2{v0.z, v0.w});
        const f16x4 v1h = __builtin_bit_cast(f16x4, u32x2{v1.x, v1.y}), v1l = __builtin_bit_cast(f16x4, u32x2{v1.z, v1.w});
        o0 = __builtin_amdgcn_mfma_f32_16x16x16f16(v0h, p_lo, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_16x16x16f16(v1h, p_lo, o1, 0, 0, 0);
        o0 = __builtin_amdgcn_mfma_f32_16x16x16f16(v0l, p_hi, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_16x16x16f16(v1l, p_hi, o1, 0, 0, 0);
        o0 = __builtin_amdgcn_mfma_f32_16x16x16f16(v0h, p_hi, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_16x16x16f16(v1h, p_hi, o1, 0, 0, 0);
      }
      float sum = sum2.x + sum2.y;
      sum += __shfl_xor(sum, 16);
      sum += __shfl_xor(sum, 32);
      const float inv = 1.f / sum;                                  // the row sum belongs to query l15 - this lane's own output column
      const float ov[8] = {o0[0] * inv, o0[1] * inv, o0[2] * inv, o0[3] * inv, o1[0] * inv, o1[1] * inv, o1[2] * inv, o1[3] * inv};
      f16x8 o_hi, o_lo;
      split8(ov, o_hi, o_lo);
      // ---- out^T += Wp[:, head g] O^T: the head's 32 input channels are one K step, in the order the accumulators left them ----
#pragma unroll
      for (int ct = 0; ct < 6; ++ct) {
        const uint8_t* wr = Wp + (16 * ct + l15) * WPB + 64 * g + 16 * lg;
        const f16x8 wh = *reinterpret_cast<const f16x8*>(wr), wl = *reinterpret_cast<const f16x8*>(wr + C * WPB);
        pacc[ct] = mma3(wh, wl, o_hi, o_lo, pacc[ct]);
      }
    }
    STAMP(6 + 5 * g);
  }
  // ---- + projection bias + shortcut, rows back through the slice map ----
  if (row >= 0) {
    const uint32_t ro = (uint32_t)row * (uint32_t)(C * 4) + (uint32_t)(16 * lg);
#pragma unroll
    for (int ct = 0; ct < 6; ++ct) {
      const u32x4 xr = __builtin_amdgcn_raw_buffer_load_b128(x_rs, ro, 64 * ct, 0);
      f32x4 o = pacc[ct];
      if (d.proj_bias) {
        const f32x4 pb = *reinterpret_cast<const f32x4*>(d.proj_bias + 16 * ct + 4 * lg);
        o += pb;
      }
      o += f32x4{__uint_as_float(xr.x), __uint_as_float(xr.y), __uint_as_float(xr.z), __uint_as_float(xr.w)};
      __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3])},
                                             o_rs, ro, 64 * ct, 0);
    }
  }
#ifdef SDF_STAMP
  __builtin_amdgcn_s_waitcnt(0);
  STAMP(17);
  if (st_on) for (int i = 0; i < 24; ++i) g_ann_stamp[st_slot * 24 + i] = st_buf[i];
#endif
}

}  // namespace

#ifdef SDF_STAMP
extern "C" int sdf_debug_read_stamps_ann(unsigned long long* host72) {
  return (int)hipMemcpyFromSymbol(host72, HIP_SYMBOL(g_ann_stamp), 72 * sizeof(unsigned long long));
}
#endif

extern "C" int sdf_ann_attn_block_supported(int C_, int nH, int N_) { return C_ == C && nH == NH && N_ == N; }

extern "C" int sdf_ann_attn_block_fwd(const SdfAnnAttnBlockDesc* d, void* stream) {
  if (!d) return SDF_E_NULL;
  if (!d->x || !d->out || !d->row_map || !d->ln_w || !d->ln_b || !d->wqkv || !d->wproj || !d->scale || !d->table) return SDF_E_NULL;
  if (d->C != C || d->nH != NH || d->N != N || d->B_ < 1 || d->rows < 1) return SDF_E_SHAPE;
  if (d->nW < 1 || d->B_ % d->nW) return SDF_E_SHAPE;
  if ((int64_t)d->rows * C * 4 >= (1LL << 31)) return SDF_E_SHAPE;              // 32-bit buffer offsets
  if (!sdf_aligned(d->x, 16) || !sdf_aligned(d->out, 16) || !sdf_aligned(d->wqkv, 16) || !sdf_aligned(d->wproj, 16) ||
      !sdf_aligned(d->ln_w, 16) || !sdf_aligned(d->ln_b, 16) || (d->qkv_bias && !sdf_aligned(d->qkv_bias, 16)) ||
      (d->proj_bias && !sdf_aligned(d->proj_bias, 16)))
    return SDF_E_ALIGN;
  BlockParams P;
  P.d = *d;
  static std::atomic<uint64_t> raised{0};                                        // > 64 KiB of dynamic LDS: opt-in once per device
  if (const int e1 = sdf_lds_opt_in(raised, reinterpret_cast<const void*>(ann_attn_block_kernel), LDS_BYTES)) return e1;
  hipStream_t s = sdf_stream(stream);
  const dim3 grid((unsigned)d->B_), block(NTHR);
  SDF_LAUNCH(ann_attn_block_kernel, grid, block, LDS_BYTES, s, P);
  SDF_LAUNCH_CHECK();
  return 0;
}
