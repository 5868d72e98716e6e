// The two activation-side products of a Linear layer in the TRAINING path, for gfx950, straight from the fp32 tensors autograd holds
// (no packed weight planes to keep in step with the optimiser, no scale to choose, no host synchronisation):
//
//   mode 0  forward   out[m, n] = sum_k x[m, k] * W[n, k] + bias[n]       x (M, K) holding spikes (exact in bf16), W (N, K) fp32
//   mode 1  dX        out[m, k] = sum_n dY[m, n] * W[n, k]                dY (M, N) fp32 gradients, W (N, K) fp32
//
// (reference: nn.Linear forward / autograd in the training step, train_flow_parallel_supervised_SNN.py:233-336; layers
// Spiking_swin_transformer3D.py:661-717, :164-181, :952-974.)  Real operands are split where they enter LDS into THREE bf16 planes by
// truncation (hi + mid + lo == value exactly, fp32's exponent range - gradients span decades the fp16 hi / lo split of the inference
// kernels cannot hold): the forward multiplies the spike plane with the three weight planes (3 products, exact products, fp32 sums);
// dX keeps the eight plane pairs down to 2^-24 of a product (all but lo x lo, which is 2^-32 of it).
//
// A workgroup owns 128 rows x 96 output columns, a wave 32 x 96 as 2 x 6 blocks of v_mfma_f32_16x16x32_bf16 with the WEIGHTS as the
// row operand (a lane ends with four consecutive output columns of one row: 16-byte stores).  The reduction index advances in chunks
// of 32; inside a chunk lane group g takes indices 4g..4g+3 and 16+4g..16+4g+3 for both operands - for dX the weight tile lies in
// LDS as it lies in memory ([n][k], the reduction index is the row) and is read with ds_read_b64_tr_b16, and that dealing is what makes
// those reads conflict-free at a 224-byte pitch (csrc/linear_dw.hip); row-major operands read the same indices as two 8-byte pieces.
#include "common.h"

namespace {

constexpr int BM = 128, BC = 96, RC = 32;
constexpr int RS = 80;                                   // row-major images: 32 x 16 bit + 16 pad (16 rows x 8 bytes: conflict-free)
constexpr int RP = 224;                                  // [reduction][column] weight image of mode 1: 96 x 16 bit + 32 pad
constexpr uint32_t INV = 0x80000000u;

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

struct TrainParams {
  SdfLinearTrainDesc d;
  int tiles_c;
  int cchunks;                                           // convolution form (mode 0): chunks of 32 per tap = cv_C / 32, else 0
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint32_t top2(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }
__device__ __forceinline__ uint2 top4(u32x4 v) { return make_uint2(top2(v.x, v.y), top2(v.z, v.w)); }

// four fp32 values -> three 8-byte words of bf16 planes, hi + mid + lo == v exactly (truncation splits)
__device__ __forceinline__ void split3(u32x4 v, uint2& hi, uint2& mid, uint2& lo) {
  uint32_t h[4], m[4], l[4];
  const uint32_t x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    h[i] = x[i] & 0xFFFF0000u;
    const float r1 = __uint_as_float(x[i]) - __uint_as_float(h[i]);
    m[i] = __float_as_uint(r1) & 0xFFFF0000u;
    l[i] = __float_as_uint(r1 - __uint_as_float(m[i]));
  }
  hi = make_uint2(top2(h[0], h[1]), top2(h[2], h[3]));
  mid = make_uint2(top2(m[0], m[1]), top2(m[2], m[3]));
  lo = make_uint2(top2(l[0], l[1]), top2(l[2], l[3]));
}

// reduction indices 4g..4g+3 | 16+4g..16+4g+3 of one row of a row-major image (p points at index 4g of the row)
__device__ __forceinline__ bf16x8 row_frag(const uint8_t* p) {
  const uint2 a = *reinterpret_cast<const uint2*>(p), b = *reinterpret_cast<const uint2*>(p + 32);
  const u32x4 r = {a.x, a.y, b.x, b.y};
  return __builtin_bit_cast(bf16x8, r);
}
// the same indices of 16 COLUMNS of a [reduction][column] image through the transposing read (csrc/linear_dw.hip)
__device__ __forceinline__ bf16x8 tr_frag(const uint8_t* p) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 16 * RP));
  s16x8 r;
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3]; r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
  return __builtin_bit_cast(bf16x8, r);
}

template <int MODE>
__global__ __launch_bounds__(256) void linear_train_kernel(TrainParams P) {
  constexpr int AP = MODE == 0 ? 1 : 3;                               // planes of the activation-side operand
  constexpr int A_PLANE = BM * RS;
  constexpr int W_PLANE = MODE == 0 ? BC * RS : RC * RP;
  constexpr int W_BASE = AP * A_PLANE;
  __shared__ __attribute__((aligned(16))) uint8_t smem[AP * A_PLANE + 3 * W_PLANE];
  const SdfLinearTrainDesc& d = P.d;
  const int M = d.M, N = d.N, K = d.K;
  const int R = MODE == 0 ? K : N, Cn = MODE == 0 ? N : K;            // reduction length, output columns
  // convolution form (mode 0, cv_C > 0): a is (M, cv_C) zero-ringed channels-last pixel rows, K = 9 cv_C ordered (ky, kx, c), and the chunk
  // of tap (ky, kx) reads the rows (ky - 1) * cv_Wp + (kx - 1) further on - a shift of the loader's offset, nothing else (rows before the
  // tensor wrap to offsets beyond the descriptor and read as zero, like those past its end)
  const int lda = (MODE == 0 && P.cchunks > 0) ? d.cv_C : R;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  // the column tiles of one row tile read the same activation rows: consecutive logical ids share an XCD (its L2)
  const int G = gridDim.x;
  const int wg = (G & 7) == 0 ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int tm = wg / P.tiles_c, tc = wg - tm * P.tiles_c;
  const int m0 = tm * BM, c0 = tc * BC;

  const __amdgpu_buffer_rsrc_t A_rs = rsrc(d.a, (uint32_t)M * (uint32_t)lda * 4u);
  const __amdgpu_buffer_rsrc_t W_rs = rsrc(d.w, (uint32_t)N * (uint32_t)K * 4u);

  // loader.  A: 128 rows x 8 float4 per chunk, 4 per thread.  W: mode 0 - 96 rows (n) x 8 float4; mode 1 - 32 rows (n) x 24 float4: 3 per thread
  uint32_t a_off[4], a_lds[4], w_off[3], w_lds[3];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = tid + 256 * i, row = p >> 3, c4 = p & 7;
    a_off[i] = m0 + row < M ? (uint32_t)((m0 + row) * lda + c4 * 4) * 4u : INV;
    a_lds[i] = (uint32_t)(row * RS + c4 * 8);
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int p = tid + 256 * i;
    if (MODE == 0) {
      const int row = p >> 3, c4 = p & 7;
      w_off[i] = (uint32_t)((c0 + row) * K + c4 * 4) * 4u;
      w_lds[i] = (uint32_t)(W_BASE + row * RS + c4 * 8);
    } else {
      const int row = p / 24, c4 = p - row * 24;
      w_off[i] = (uint32_t)(row * K + c0 + c4 * 4) * 4u;
      w_lds[i] = (uint32_t)(W_BASE + row * RP + c4 * 8);
    }
  }
  const uint32_t a_step = RC * 4u, w_step = MODE == 0 ? RC * 4u : (uint32_t)(RC * K) * 4u;
  u32x4 areg[4], wreg[3];
  auto request = [&](int c) __attribute__((always_inline)) {
    uint32_t a_add = (uint32_t)c * a_step;
    if (MODE == 0 && P.cchunks > 0) {
      const int tap = c / P.cchunks, cc = c - tap * P.cchunks, ky = tap / 3, kx = tap - 3 * ky;
      a_add = (uint32_t)(((ky - 1) * d.cv_Wp + (kx - 1)) * lda * 4 + cc * (int)a_step);   // (two's complement: may be "negative")
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) areg[i] = __builtin_amdgcn_raw_buffer_load_b128(A_rs, a_off[i] != INV ? a_off[i] + a_add : INV, 0, 0);
#pragma unroll
    for (int i = 0; i < 3; ++i) wreg[i] = __builtin_amdgcn_raw_buffer_load_b128(W_rs, w_off[i] + (uint32_t)c * w_step, 0, 0);
  };
  auto deposit = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (MODE == 0) {
        *reinterpret_cast<uint2*>(smem + a_lds[i]) = top4(areg[i]);    // spikes: the top half is the value
      } else {
        uint2 hi, mid, lo;
        split3(areg[i], hi, mid, lo);
        *reinterpret_cast<uint2*>(smem + a_lds[i]) = hi;
        *reinterpret_cast<uint2*>(smem + A_PLANE + a_lds[i]) = mid;
        *reinterpret_cast<uint2*>(smem + 2 * A_PLANE + a_lds[i]) = lo;
      }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      uint2 hi, mid, lo;
      split3(wreg[i], hi, mid, lo);
      *reinterpret_cast<uint2*>(smem + w_lds[i]) = hi;
      *reinterpret_cast<uint2*>(smem + W_PLANE + w_lds[i]) = mid;
      *reinterpret_cast<uint2*>(smem + 2 * W_PLANE + w_lds[i]) = lo;
    }
  };

  int ln = lane;
  asm volatile("" : "+v"(ln));
  const int g = ln >> 4, li = ln & 15;
  const uint32_t a_frag = (uint32_t)((wave * 32 + li) * RS + 8 * g);                 // + 16 rows for the second row block
  const uint32_t w_frag = MODE == 0 ? (uint32_t)(W_BASE + li * RS + 8 * g)          // + 16 rows per column block
                                    : (uint32_t)(W_BASE + (4 * g + (li >> 2)) * RP + (li & 3) * 8);   // + 32 bytes per column block

  f32x4 acc[2][6], acs[2][6];                                          // hi x hi products | every smaller plane pair
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) acc[i][j] = acs[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nchunk = R / RC;
  request(0);
#pragma unroll 1
  for (int c = 0; c < nchunk; ++c) {
    __syncthreads();                                                   // every wave is done with the previous chunk's image
    deposit();
    __syncthreads();
    if (c + 1 < nchunk) request(c + 1);
    bf16x8 a[AP][2];
#pragma unroll
    for (int pl = 0; pl < AP; ++pl)
#pragma unroll
      for (int i = 0; i < 2; ++i) a[pl][i] = row_frag(smem + pl * A_PLANE + a_frag + i * 16 * RS);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      bf16x8 w[3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        w[pl] = MODE == 0 ? row_frag(smem + pl * W_PLANE + w_frag + j * 16 * RS) : tr_frag(smem + pl * W_PLANE + w_frag + j * 32);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        // the hi x hi products have their own accumulator: the matrix pipe aligns a product to the running sum with a bounded number
        // of guard bits and drops the rest one-sidedly - per element 1e-8 of the result for the 768 small addends of a 96-long
        // reduction, invisible per element and systematic over a sum of 10^7 elements (tools/linear_train_bias.py; a PSN bias
        // gradient moved 2.4e-4 of its largest element against the fp64-checked oracle).  Small planes meet small sums.
        if (MODE == 0) {
          acs[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[2], a[0][i], acs[i][j], 0, 0, 0);
          acs[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], a[0][i], acs[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], a[0][i], acc[i][j], 0, 0, 0);
        } else {
          acs[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[2], a[AP / 2][i], acs[i][j], 0, 0, 0); // w lo  a mid
          acs[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], a[AP - 1][i], acs[i][j], 0, 0, 0); // w mid a lo
          acs[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[2], a[0][i], acs[i][j], 0, 0, 0);      // w lo  a hi
          acs[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], a[AP - 1][i], acs[i][j], 0, 0, 0); // w hi  a lo
          acs[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], a[AP / 2][i], acs[i][j], 0, 0, 0); // w mid a mid
          acs[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], a[0][i], acs[i][j], 0, 0, 0);      // w mid a hi
          acs[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], a[AP / 2][i], acs[i][j], 0, 0, 0); // w hi  a mid
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], a[0][i], acc[i][j], 0, 0, 0);      // w hi  a hi
        }
      }
    }
  }

  // accumulator register r of block (i, j): out[m0 + 32 wave + 16 i + li][c0 + 16 j + 4 g + r]
  const __amdgpu_buffer_rsrc_t C_rs = rsrc(d.out, (uint32_t)M * (uint32_t)Cn * 4u);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + wave * 32 + 16 * i + li;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int col = c0 + 16 * j + 4 * g;
      f32x4 o = acc[i][j] + acs[i][j];
      if (MODE == 0 && d.bias) {
        const float4 b = *reinterpret_cast<const float4*>(d.bias + col);
        o[0] += b.x; o[1] += b.y; o[2] += b.z; o[3] += b.w;
      }
      u32x4 st;
      st.x = __float_as_uint(o[0]); st.y = __float_as_uint(o[1]); st.z = __float_as_uint(o[2]); st.w = __float_as_uint(o[3]);
      __builtin_amdgcn_raw_buffer_store_b128(st, C_rs, m < M ? (uint32_t)(m * Cn + col) * 4u : INV, 0, 0);
    }
  }
}

}  // namespace

extern "C" int sdf_linear_train_fwd(const SdfLinearTrainDesc* d, void* stream) {
  if (!d || !d->a || !d->w || !d->out) return SDF_E_NULL;
  if (d->M <= 0 || d->N <= 0 || d->K <= 0 || (d->mode != 0 && d->mode != 1)) return SDF_E_SHAPE;
  if (d->cv_C < 0 || (d->cv_C > 0 && (d->mode != 0 || d->cv_C % RC || d->K != 9 * d->cv_C || d->cv_Wp < 3))) return SDF_E_SHAPE;
  const int R = d->mode == 0 ? d->K : d->N, Cn = d->mode == 0 ? d->N : d->K;
  if (R % RC || Cn % BC) return SDF_E_SHAPE;
  const int64_t lim = (int64_t)1 << 31;
  const int64_t a_cols = d->cv_C > 0 ? d->cv_C : d->K;                // (convolution form: a is (M, cv_C), K counts the nine taps)
  if ((int64_t)d->M * a_cols * 4 >= lim || (int64_t)d->M * d->N * 4 >= lim || (int64_t)d->N * d->K * 4 >= lim) return SDF_E_SHAPE;
  if (!sdf_aligned(d->a, 16) || !sdf_aligned(d->w, 16) || !sdf_aligned(d->out, 16) || (d->bias && !sdf_aligned(d->bias, 16))) return SDF_E_ALIGN;
  TrainParams P;
  P.d = *d;
  P.tiles_c = Cn / BC;
  P.cchunks = d->cv_C > 0 ? d->cv_C / RC : 0;
  const int64_t wgs = (int64_t)((d->M + BM - 1) / BM) * P.tiles_c;
  hipStream_t s = sdf_stream(stream);
  if (d->mode == 0) SDF_LAUNCH(linear_train_kernel<0>, dim3((unsigned)wgs), dim3(256), 0, s, P);
  else SDF_LAUNCH(linear_train_kernel<1>, dim3((unsigned)wgs), dim3(256), 0, s, P);
  SDF_LAUNCH_CHECK();
  return 0;
}
