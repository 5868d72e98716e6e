// The MLP half of an ANN video-swin block as ONE launch (gfx950):
//
//   out = x + fc2( GELU( fc1( LayerNorm(x) ) ) )
//
// - reference models/STSwinNet/swin_transformer3D_v2.py:312-336 (`forward_part2` + the shortcut) with `Mlp.forward` (:15-34) inside.
// Three launches before (LayerNorm, fc1 + GELU, fc2 + shortcut): the hidden activations - 4 C floats per token, 170 MB on BASELINE config
// 3's first stage - were written and read back, and both Linear launches were HBM-bound on them (docs/history/DESIGN_rounds1-5.md section 5).  The companion of
// ann_block.hip, same ownership: a wave keeps 16 tokens for the whole kernel, LayerNorm(x) lives in its registers as the fp16 hi / lo
// operand of fc1; the hidden dimension is walked in chunks of 96: H^T chunk = W1[chunk] LN(x)^T with the weights as the MFMA's row
// operand - a lane ends with hidden units of ITS token - GELU in registers, split into hi / lo: that IS the column operand of
// out^T += W2[:, chunk] H^T (W2's hidden channels permuted once on the host into the order the accumulators leave them), whose
// accumulators (16 tokens x 96 channels per wave) collect the four chunks.  Nothing but x and out touches HBM.
// A chunk's weights (96 rows of W1, 96 columns of W2; fp16 hi / lo planes, three products per fp32 product: the numerics of
// dense_linear.hip) are staged in LDS - requested a chunk ahead into registers, written between two barriers.
// Built for C = 96, hidden 384 (config 3's first stage).  Compiled with -ffp-contract=off.
#include "common.h"

namespace {

constexpr int C = 96, CH = 384, NCHUNK = CH / 96;
constexpr int NWAVE = 12, NTHR = 64 * NWAVE, ROWS_WG = 16 * NWAVE;
constexpr int WPB = 2 * 96 + 32;                // weight row pitch in bytes (12 pieces + 2: conflict-free for ds_read_b128's real lane groups)
constexpr int W1_BYTES = 2 * 96 * WPB, W2_BYTES = 2 * C * WPB;
constexpr int LDS_BYTES = W1_BYTES + W2_BYTES;
constexpr uint32_t INV_OFF = 0x80000000u;

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

struct MlpParams {
  SdfAnnMlpBlockDesc d;
};

__device__ __forceinline__ void split2(float x, float y, uint32_t& hi, uint32_t& lo) {
  const f32x2 v = {x, y};
  const f16x2 h = __builtin_convertvector(v, f16x2);
  const f32x2 r = v - __builtin_convertvector(h, f32x2);
  const f16x2 l = __builtin_convertvector(r, f16x2);
  hi = __builtin_bit_cast(uint32_t, h);
  lo = __builtin_bit_cast(uint32_t, l);
}
__device__ __forceinline__ void split8(const float (&x)[8], f16x8& hi, f16x8& lo) {
  uint32_t h[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) split2(x[2 * i], x[2 * i + 1], h[i], l[i]);
  hi = __builtin_bit_cast(f16x8, u32x4{h[0], h[1], h[2], h[3]});
  lo = __builtin_bit_cast(f16x8, u32x4{l[0], l[1], l[2], l[3]});
}
__device__ __forceinline__ f32x4 mma3(const f16x8& ah, const f16x8& al, const f16x8& bh, const f16x8& bl, f32x4 a) {
  a = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, a, 0, 0, 0);
  a = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, a, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, a, 0, 0, 0);
}
// erf-form GELU (F.gelu) with erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 absolute, the size of an fp32 rounding of erf itself):
//   erfc(u) = t (a1 + t (a2 + t (a3 + t (a4 + t a5)))) exp(-u^2), t = 1 / (1 + p u), u = |x| / sqrt 2
//   gelu(x) = x q for x < 0, x - x q for x >= 0, q = erfc(u) / 2   (no cancellation for large negative x)
// One reciprocal, one exp2 and ten multiply-adds, branch-free: the library erff (two masked branches, ~ 50 instructions per value) made
// this kernel's 96 GELUs per lane three times its matrix work (117 us per launch; with this form: see docs/history/DESIGN_rounds1-5.md section 6).
__device__ __forceinline__ float gelu_erf(float x) {
  const float u = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, u, 1.f));
  float p = __builtin_fmaf(t, 0.5f * 1.061405429f, 0.5f * -1.453152027f);
  p = __builtin_fmaf(t, p, 0.5f * 1.421413741f);
  p = __builtin_fmaf(t, p, 0.5f * -0.284496736f);
  p = __builtin_fmaf(t, p, 0.5f * 0.254829592f);
  const float q = p * t * __builtin_amdgcn_exp2f(u * u * -1.4426950408889634f);
  const float xq = x * q;
  return x < 0.f ? xq : x - xq;
}

__global__ __launch_bounds__(NTHR) void ann_mlp_block_kernel(MlpParams P) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const SdfAnnMlpBlockDesc& d = P.d;
  uint8_t* W1s = smem;                            // [hi | lo][96 hidden units of the chunk][WPB]
  uint8_t* W2s = smem + W1_BYTES;                 // [hi | lo][96 out channels][WPB]: the chunk's 96 hidden channels, accumulator order
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lg = lane >> 4;
  const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x), 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t o_rs = __builtin_amdgcn_make_buffer_rsrc(d.out, 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t w1_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(d.w1), 0, 2 * CH * C * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t w2_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(d.w2), 0, 2 * C * CH * 2, 0x00020000);

  // a chunk's weights: W1 rows 96 c .. 96 c + 95 (12 pieces each) and W2 columns 96 c .. (12 pieces of every row), two planes each:
  // 2 x 2 304 pieces = 6 per thread, requested a chunk ahead
  // (the two halves are requested at different points of the chunk before: 24 registers in flight at once spilled)
  u32x4 wpre[6];
  auto request_w1 = [&](int c) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int i = tid + NTHR * j, prow = i / 12, pc = i - prow * 12;            // prow = plane * 96 + row
      const int pl = prow / 96, r = prow - pl * 96;
      wpre[j] = __builtin_amdgcn_raw_buffer_load_b128(w1_rs, (uint32_t)(((pl * CH + 96 * c + r) * C) * 2 + pc * 16), 0, 0);
    }
  };
  auto request_w2 = [&](int c) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int i = tid + NTHR * j, prow = i / 12, pc = i - prow * 12;
      const int pl = prow / 96, r = prow - pl * 96;
      wpre[3 + j] = __builtin_amdgcn_raw_buffer_load_b128(w2_rs, (uint32_t)(((pl * C + r) * CH + 96 * c) * 2 + pc * 16), 0, 0);
    }
  };
  request_w1(0);
  request_w2(0);

  // ---- this wave's 16 tokens: LayerNorm -> fp16 hi / lo operand (channels 32 s + 8 lg + 0..7 of token l15), as in ann_block.hip ----
  const int64_t tok = (int64_t)blockIdx.x * ROWS_WG + wave * 16 + l15;
  const bool tok_ok = tok < d.rows;
  float4 lng[3][2], lnb[3][2];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    lng[s][0] = *reinterpret_cast<const float4*>(d.ln_w + 32 * s + 8 * lg); lng[s][1] = *reinterpret_cast<const float4*>(d.ln_w + 32 * s + 8 * lg + 4);
    lnb[s][0] = *reinterpret_cast<const float4*>(d.ln_b + 32 * s + 8 * lg); lnb[s][1] = *reinterpret_cast<const float4*>(d.ln_b + 32 * s + 8 * lg + 4);
  }
  f16x8 yh[3], yl[3];
  {
    float xv[3][8];
    const uint32_t xo = tok_ok ? (uint32_t)tok * (uint32_t)(C * 4) + (uint32_t)(32 * lg) : INV_OFF;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(x_rs, xo, s * 128, 0);
      const u32x4 c = __builtin_amdgcn_raw_buffer_load_b128(x_rs, xo, s * 128 + 16, 0);
      xv[s][0] = __uint_as_float(a.x); xv[s][1] = __uint_as_float(a.y); xv[s][2] = __uint_as_float(a.z); xv[s][3] = __uint_as_float(a.w);
      xv[s][4] = __uint_as_float(c.x); xv[s][5] = __uint_as_float(c.y); xv[s][6] = __uint_as_float(c.z); xv[s][7] = __uint_as_float(c.w);
    }
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
    const float rstd = 1.f / sqrtf(sq * (1.f / C) + d.ln_eps);
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const float4 g0 = lng[s][0], g1 = lng[s][1], b0 = lnb[s][0], b1 = lnb[s][1];
      const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      float y[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) y[i] = (xv[s][i] - mean) * rstd * gg[i] + bb[i];
      split8(y, yh[s], yl[s]);
    }
  }

  f32x4 pacc[6];                                                    // out^T: channels 16 ot + 4 lg + r of token l15
#pragma unroll
  for (int ot = 0; ot < 6; ++ot) pacc[ot] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
  for (int c = 0; c < NCHUNK; ++c) {
    __syncthreads();                                                // every wave is done with the previous chunk's weights
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int i = tid + NTHR * j, prow = i / 12, pc = i - prow * 12;
      *reinterpret_cast<u32x4*>(W1s + prow * WPB + pc * 16) = wpre[j];
      *reinterpret_cast<u32x4*>(W2s + prow * WPB + pc * 16) = wpre[3 + j];
    }
    __syncthreads();
    if (c + 1 < NCHUNK) request_w1(c + 1);
    // ---- H^T chunk = W1[chunk] LN(x)^T + b1: six tiles of 16 hidden units; a lane ends with units 16 ht + 4 lg + r of ITS token ----
    f16x8 hh[3], hl[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float hv[8];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int ht = 2 * j + u;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (d.b1) acc = *reinterpret_cast<const f32x4*>(d.b1 + 96 * c + 16 * ht + 4 * lg);
        const uint8_t* wr = W1s + (16 * ht + l15) * WPB + 16 * lg;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          const f16x8 wh = *reinterpret_cast<const f16x8*>(wr + 64 * s), wl = *reinterpret_cast<const f16x8*>(wr + 96 * WPB + 64 * s);
          acc = mma3(wh, wl, yh[s], yl[s], acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) hv[4 * u + r] = gelu_erf(acc[r]);
      }
      // the pair of tiles (2 j, 2 j + 1) is K step j of the second product: slots 8 lg + i = units 32 j + (i < 4 ? 4 lg + i : 16 + 4 lg + i - 4)
      split8(hv, hh[j], hl[j]);
    }
    // ---- out^T += W2[:, chunk] H^T ----
    if (c + 1 < NCHUNK) request_w2(c + 1);
#pragma unroll
    for (int ot = 0; ot < 6; ++ot) {
      const uint8_t* wr = W2s + (16 * ot + l15) * WPB + 16 * lg;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const f16x8 wh = *reinterpret_cast<const f16x8*>(wr + 64 * j), wl = *reinterpret_cast<const f16x8*>(wr + C * WPB + 64 * j);
        pacc[ot] = mma3(wh, wl, hh[j], hl[j], pacc[ot]);
      }
    }
  }
  // ---- + fc2 bias + shortcut ----
  if (tok_ok) {
    const uint32_t ro = (uint32_t)tok * (uint32_t)(C * 4) + (uint32_t)(16 * lg);
#pragma unroll
    for (int ot = 0; ot < 6; ++ot) {
      const u32x4 xr = __builtin_amdgcn_raw_buffer_load_b128(x_rs, ro, 64 * ot, 0);
      f32x4 o = pacc[ot];
      if (d.b2) o += *reinterpret_cast<const f32x4*>(d.b2 + 16 * ot + 4 * lg);
      o += f32x4{__uint_as_float(xr.x), __uint_as_float(xr.y), __uint_as_float(xr.z), __uint_as_float(xr.w)};
      __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3])},
                                             o_rs, ro, 64 * ot, 0);
    }
  }
}

}  // namespace

extern "C" int sdf_ann_mlp_block_supported(int C_, int Ch_) { return C_ == C && Ch_ == CH; }

extern "C" int sdf_ann_mlp_block_fwd(const SdfAnnMlpBlockDesc* d, void* stream) {
  if (!d) return SDF_E_NULL;
  if (!d->x || !d->out || !d->ln_w || !d->ln_b || !d->w1 || !d->w2) return SDF_E_NULL;
  if (d->C != C || d->Ch != CH || d->rows < 1) return SDF_E_SHAPE;
  if ((int64_t)d->rows * C * 4 >= (1LL << 31)) return SDF_E_SHAPE;              // 32-bit buffer offsets
  if (!sdf_aligned(d->x, 16) || !sdf_aligned(d->out, 16) || !sdf_aligned(d->w1, 16) || !sdf_aligned(d->w2, 16) || !sdf_aligned(d->ln_w, 16) ||
      !sdf_aligned(d->ln_b, 16) || (d->b1 && !sdf_aligned(d->b1, 16)) || (d->b2 && !sdf_aligned(d->b2, 16)))
    return SDF_E_ALIGN;
  MlpParams P;
  P.d = *d;
  static std::atomic<uint64_t> raised{0};                                        // > 64 KiB of dynamic LDS: opt-in once per device
  if (const int e = sdf_lds_opt_in(raised, reinterpret_cast<const void*>(ann_mlp_block_kernel), LDS_BYTES)) return e;
  const int64_t wgs = (d->rows + ROWS_WG - 1) / ROWS_WG;
  SDF_LAUNCH(ann_mlp_block_kernel, dim3((unsigned)wgs), dim3(NTHR), LDS_BYTES, sdf_stream(stream), P);
  SDF_LAUNCH_CHECK();
  return 0;
}
