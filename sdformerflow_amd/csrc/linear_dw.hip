// Weight gradient of a Linear layer fed by SPIKES, for gfx950 (training path, BASELINE configs[3]):
//
//   dW[n, k] = sum_m dY[m, n] * X[m, k]          dY (M, N) fp32 row-major, X (M, K) fp32 row-major holding 0 / 1
//
// what torch.autograd computes for every nn.Linear of the MS swin blocks in the reference's training step
// (train_flow_parallel_supervised_SNN.py:233-336 `loss.backward()`; the layers: Spiking_swin_transformer3D.py:661-717 linear_q /
// linear_k / proj, :164-181 fc1 / fc2, :952-974 reduction) - until round 5 a library fp32 GEMM with a 276 480-long reduction
// and a 96 x 96 result at stage 0.
//
// Both operands are m-major in memory and the matrix pipe wants the reduction index contiguous per lane: the tiles go into LDS as
// they lie in memory (row = m, 16-bit elements) and are read back with ds_read_b64_tr_b16, gfx950's transposing LDS read - a 16-lane
// group takes a 4-row x 16-column block and each lane receives one COLUMN of it, which is the operand layout of
// v_mfma_f32_16x16x32_bf16 (lane = (column, k group), 8 reduction indices per lane in two such reads).  The order of the reduction
// inside a 32-row chunk is free as long as both operands agree: k group g takes rows 4g..4g+3 and 16+4g..16+4g+3, so that the two
// groups of a 32-lane half read rows 0..7 of a 224-byte pitch = 8 disjoint runs of 8 banks (conflict-free; the natural 8g..8g+7
// cannot be: rows r and r + 8 share banks at every pitch whose runs are 8-bank aligned).
//
// Numerics: the spikes are exact in bf16.  dY is a gradient - its range is not the activations' - and is split, where it enters LDS,
// into THREE bf16 planes by truncation (hi = top 16 bits, mid = top 16 bits of the exact remainder, lo = the rest: 8 + 8 + 8
// mantissa bits, hi + mid + lo == dY exactly, fp32's exponent range, no scale to choose), three products per block, fp32 accumulation.
// The kernel is HBM-bound at N = K = 96 (480 B of operands per row against 55 kflop x 3) and the third product is free there.
// All three planes of a block share ONE accumulator (linear_train.hip keeps hi x hi apart: the 16-bit pipe drops alignment bits
// one-sidedly when a small product meets a large running sum).  Measured for this kernel (tools/linear_train_bias.py,
// profiles/r5bd_linear_train_bias.txt): per element of dW - a 276 480-long sum - rms error 2.1e-7 of mean |dw| against the
// library's 2.2e-6, bias -1.4e-7 of mean |dw|; a weight gradient is consumed element by element (clip, AdamW), nothing sums over it the
// way a PSN bias gradient sums over dX, so the second accumulator (27 more registers per tap set: the convolution form has none
// left) buys nothing here.
//
// A workgroup (4 waves) owns a 96 x 96 tile of dW over a contiguous range of m; a wave 48 x 48 (3 x 3 blocks of 16 x 16).  m advances
// in chunks of 32 through a double-buffered LDS image (next chunk requested before the MFMAs, split / written after them, one
// barrier per chunk).  The m ranges of a tile (split count chosen for ~3 workgroups per compute unit) leave fp32 partial tiles that a
// second kernel adds in a fixed order - no atomics, the same bits every run.
#include "common.h"

namespace {

constexpr int TN = 96, TK = 96, MC = 32;
constexpr int RP = 224;                                  // LDS row pitch in bytes: 96 x 16 bit + 32 pad = 7 x 32 (see above)
constexpr int PLANE = MC * RP;
constexpr uint32_t INV = 0x80000000u;

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

struct DwParams {
  SdfLinearDwDesc d;
  int tiles_n, tiles_k, rows_per_split;
  int x_ld;                                              // row pitch of X in floats (Linear: K; convolution form: C)
  int cblocks, Wp;                                       // convolution form: 96-channel blocks of C, padded image width
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

// two fp32 values -> the dword {bf16(a), bf16(b)} of their top halves
__device__ __forceinline__ uint32_t top2(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// four fp32 values -> three 8-byte words of bf16 planes, hi + mid + lo == v exactly (truncation splits)
__device__ __forceinline__ void split3(u32x4 v, uint2& hi, uint2& mid, uint2& lo) {
  uint32_t h[4], m[4], l[4];
  const uint32_t x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    h[i] = x[i] & 0xFFFF0000u;
    const float r1 = __uint_as_float(x[i]) - __uint_as_float(h[i]);
    m[i] = __float_as_uint(r1) & 0xFFFF0000u;
    l[i] = __float_as_uint(r1 - __uint_as_float(m[i]));            // <= 8 significant bits: its top half is all of it
  }
  hi = make_uint2(top2(h[0], h[1]), top2(h[2], h[3]));
  mid = make_uint2(top2(m[0], m[1]), top2(m[2], m[3]));
  lo = make_uint2(top2(l[0], l[1]), top2(l[2], l[3]));
}

__device__ __forceinline__ bf16x8 tr_frag(const uint8_t* p) {
  // rows 4g + q (this read) and 16 + 4g + q (the next): element j of the lane = reduction index 4g + j, 16 + 4g + (j - 4)
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 16 * RP));
  s16x8 r;
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3]; r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
  return __builtin_bit_cast(bf16x8, r);
}

// TAPS = 1: the Linear form.  TAPS = 3: the 3x3 / stride 1 / pad 1 CONVOLUTION form on zero-ringed channels-last images - row m is a
// pixel of the (H + 2) x (W + 2) grid, dY is zero on the ring, and tap (ky, kx) multiplies dY[m] with X[m + (ky - 1) Wp + kx - 1]: a
// column tile is (ky, 96 input channels) and its three kx taps are ONE staged X image of 34 rows read at row offsets 0, 1, 2 - in an
// m-major image a pixel shift is an address offset of the transposing read, nothing else.
template <int TAPS>
__global__ __launch_bounds__(256, 2) void linear_dw_kernel(DwParams P) {
  constexpr int XR = MC + TAPS - 1;                                   // rows of the X image
  constexpr int XPLANE = XR * RP, BUF = 3 * PLANE + XPLANE;
  constexpr int NXP = (XR * 24 + 255) / 256;                          // X pieces per thread
  __shared__ __attribute__((aligned(16))) uint8_t smem[2 * BUF];
  const SdfLinearDwDesc& d = P.d;
  const int N = d.N;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int tiles = P.tiles_n * P.tiles_k;
  // the tiles of one m range read the same rows: they must share an L2, i.e. an XCD - consecutive workgroup ids go round the 8 XCDs,
  // so ids are re-dealt to make consecutive LOGICAL ids neighbours on one XCD (measured without: the N = 96, K = 384 layer read
  // 846 MB for 531 MB of operands, the convolution form 2.5 GB for 0.87, profiles/r5bb_pmc_linear_dw.txt)
  const int G = gridDim.x;
  const int wg = (G & 7) == 0 ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int split = wg / tiles, tile = wg - split * tiles;
  const int tn = tile / P.tiles_k, tk = tile - tn * P.tiles_k;
  const int ky = TAPS == 3 ? tk / P.cblocks : 0, cb = TAPS == 3 ? tk - ky * P.cblocks : tk;
  const int n0 = tn * TN;
  const int64_t m_begin = (int64_t)split * P.rows_per_split;
  int64_t m_end = m_begin + P.rows_per_split;
  if (m_end > d.M) m_end = d.M;
  const int nchunk = m_end > m_begin ? (int)((m_end - m_begin + MC - 1) / MC) : 0;
  const int64_t xshift = TAPS == 3 ? (int64_t)(ky - 1) * P.Wp - 1 : 0;  // X row of tap 0 for dY row 0 (may be before the tensor)

  const __amdgpu_buffer_rsrc_t Y_rs = rsrc(d.dy, (uint32_t)(d.M * N * 4));
  const __amdgpu_buffer_rsrc_t X_rs = rsrc(d.x, (uint32_t)(d.M * P.x_ld * 4));

  // loader: a chunk is 32 rows x 24 float4 of dY (3 pieces per thread) and XR rows x 24 of X.  No bounds arithmetic: a range is a
  // whole number of chunks, so only the tensors' ends matter and the buffer descriptor returns zero past them - also for the rows
  // BEFORE the tensor that the convolution form's first chunks ask for (their 32-bit offsets wrap far beyond the descriptor's size)
  uint32_t y_off[3], lds_off[3], x_off[NXP], xlds_off[NXP];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int p = tid + 256 * i, row = p / 24, c4 = p - row * 24;
    y_off[i] = (uint32_t)(((m_begin + row) * N + n0 + c4 * 4) * 4);
    lds_off[i] = (uint32_t)(row * RP + c4 * 8);
  }
#pragma unroll
  for (int i = 0; i < NXP; ++i) {
    const int p = tid + 256 * i, row = p / 24, c4 = p - row * 24;
    x_off[i] = (uint32_t)(((m_begin + row + xshift) * P.x_ld + cb * TK + c4 * 4) * 4);
    xlds_off[i] = (uint32_t)(3 * PLANE + row * RP + c4 * 8);
  }
  const bool xlast = XR * 24 % 256 == 0 || tid + 256 * (NXP - 1) < XR * 24;   // the last X piece of this thread exists
  const uint32_t ystep = (uint32_t)(MC * N * 4), xstep = (uint32_t)(MC * P.x_ld * 4);
  u32x4 yA[3], xA[NXP], yB[3], xB[NXP];
  auto request = [&](int c, u32x4 (&yreg)[3], u32x4 (&xreg)[NXP]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 3; ++i) yreg[i] = __builtin_amdgcn_raw_buffer_load_b128(Y_rs, y_off[i] + (uint32_t)c * ystep, 0, 0);
#pragma unroll
    for (int i = 0; i < NXP; ++i)
      xreg[i] = __builtin_amdgcn_raw_buffer_load_b128(X_rs, (i + 1 < NXP || xlast) ? x_off[i] + (uint32_t)c * xstep : INV, 0, 0);
  };
  auto deposit = [&](uint8_t* buf, u32x4 (&yreg)[3], u32x4 (&xreg)[NXP]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      uint2 hi, mid, lo;
      split3(yreg[i], hi, mid, lo);
      *reinterpret_cast<uint2*>(buf + lds_off[i]) = hi;
      *reinterpret_cast<uint2*>(buf + PLANE + lds_off[i]) = mid;
      *reinterpret_cast<uint2*>(buf + 2 * PLANE + lds_off[i]) = lo;
    }
#pragma unroll
    for (int i = 0; i < NXP; ++i)
      if (i + 1 < NXP || xlast) *reinterpret_cast<uint2*>(buf + xlds_off[i]) = make_uint2(top2(xreg[i].x, xreg[i].y), top2(xreg[i].z, xreg[i].w));
  };

  // transposed reads: lane (g, q, p) of a fragment supplies the address of row 4g + q, columns 4p .. 4p + 3 of the block
  int ln = lane;
  asm volatile("" : "+v"(ln));
  const int g = ln >> 4, li = ln & 15;
  const uint32_t tr_lane = (uint32_t)((4 * g + (li >> 2)) * RP + (li & 3) * 8);
  const int wn = wave >> 1, wk = wave & 1;
  const uint32_t a_base = tr_lane + (uint32_t)(48 * wn * 2);           // dY columns n0 + 48 wn + ...
  const uint32_t b_base = 3 * PLANE + tr_lane + (uint32_t)(48 * wk * 2);

  f32x4 acc[TAPS][3][3];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[t][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto multiply = [&](const uint8_t* cur) __attribute__((always_inline)) {
#pragma unroll
    for (int pl = 2; pl >= 0; --pl) {                                   // small terms first
      bf16x8 a[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) a[i] = tr_frag(cur + pl * PLANE + a_base + i * 32);
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        bf16x8 b[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) b[j] = tr_frag(cur + b_base + t * RP + j * 32);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[t][i][j], 0, 0, 0);
      }
    }
  };
  // (requests past the range read the next range's rows, or zeros past the tensor: deposited, never multiplied)
  if constexpr (TAPS == 1) {
    // two register sets: chunks c + 1 AND c + 2 are in flight while chunk c is multiplied (a chunk's MFMAs are ~0.5 us, a load ~2)
    request(0, yA, xA);
    request(1, yB, xB);
    deposit(smem, yA, xA);
    __syncthreads();
#pragma unroll 1
    for (int c = 0; c < nchunk; c += 2) {
      request(c + 2, yA, xA);
      multiply(smem);
      deposit(smem + BUF, yB, xB);                                      // chunk c + 1
      __syncthreads();
      if (c + 1 < nchunk) {
        request(c + 3, yB, xB);
        multiply(smem + BUF);
      }
      deposit(smem, yA, xA);                                            // chunk c + 2
      __syncthreads();
    }
  } else {
    // three taps: 81 MFMAs per chunk and 108 accumulator registers - one chunk in flight, the other workgroup of the compute unit
    // covers the rest of the latency
    request(0, yA, xA);
    deposit(smem, yA, xA);
    __syncthreads();
#pragma unroll 1
    for (int c = 0; c < nchunk; ++c) {
      request(c + 1, yA, xA);
      multiply(smem + (c & 1) * BUF);
      deposit(smem + ((c + 1) & 1) * BUF, yA, xA);
      __syncthreads();
    }
  }

  // accumulator register r of block (t, i, j): dW[n0 + 48 wn + 16 i + 4 g + r][column of (tap, 96 cb + 48 wk + 16 j + li)]
  const int ld = d.K;
  float* out = d.nsplit > 1 ? d.partial + (int64_t)split * N * ld : d.dw;
#pragma unroll
  for (int t = 0; t < TAPS; ++t) {
    const int k0 = TAPS == 3 ? (ky * 3 + t) * P.x_ld + cb * TK : cb * TK;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float* row = out + (int64_t)(n0 + 48 * wn + 16 * i + 4 * g + r) * ld + k0 + 48 * wk + li;
#pragma unroll
        for (int j = 0; j < 3; ++j) row[16 * j] = acc[t][i][j][r];
      }
  }
}

// dw[i] = sum over the m ranges in a FIXED order: 16 lanes take the ranges s = l, l + 16, ... of a float4 column in turn and their
// sums meet in a fixed tree (a 96 x 96 layer has 768 ranges and only 2 304 float4 columns: one thread per column walked them for
// longer than the product took)
__global__ __launch_bounds__(256) void linear_dw_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int n4, int nsplit) {
  __shared__ float4 red[256];
  const int col = blockIdx.x * 16 + (threadIdx.x & 15), sl = threadIdx.x >> 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col < n4) {
    const float4* p = reinterpret_cast<const float4*>(partial) + col;
    int k = sl;
    for (; k + 48 < nsplit; k += 64) {                                 // four loads in flight, additions in range order
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = p[(int64_t)(k + 16 * u) * n4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    for (; k < nsplit; k += 16) {
      const float4 v = p[(int64_t)k * n4];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  red[threadIdx.x] = s;
  __syncthreads();
#pragma unroll
  for (int h = 8; h >= 1; h >>= 1) {
    if (sl < h) {
      const float4 a = red[threadIdx.x], b = red[threadIdx.x + 16 * h];
      red[threadIdx.x] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
    __syncthreads();
  }
  if (sl == 0 && col < n4) reinterpret_cast<float4*>(dw)[col] = red[threadIdx.x];
}

// (imgs, C, H, W) fp32 -> zero-ringed channels-last pixel rows (imgs, H + 2, W + 2, C): what the convolution form reads.  One
// workgroup moves 64 pixels of a padded row x 96 channels through LDS (global reads run along x, global writes along c).
__global__ __launch_bounds__(256) void ringed_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int H, int W) {
  __shared__ float tile[64 * 97];
  const int Wp = W + 2, cblocks = C / 96;
  const int img = blockIdx.z / cblocks, c0 = (blockIdx.z - img * cblocks) * 96;
  const int yy = blockIdx.y, px0 = blockIdx.x * 64;
  const int tid = threadIdx.x, lx = tid & 63, cw = tid >> 6;
  const int px = px0 + lx;
  const bool inside = yy >= 1 && yy <= H && px >= 1 && px <= W;
  const float* s = src + (((int64_t)img * C + c0) * H + (yy - 1)) * W + (px - 1);
#pragma unroll 4
  for (int c = cw; c < 96; c += 4) tile[lx * 97 + c] = inside ? s[(int64_t)c * H * W] : 0.f;
  __syncthreads();
  float* d = dst + (((int64_t)img * (H + 2) + yy) * Wp + px0) * C + c0;
  const int npx = Wp - px0 < 64 ? Wp - px0 : 64;
  for (int i = tid; i < npx * 96; i += 256) {
    const int p = i / 96, c = i - p * 96;
    d[(int64_t)p * C + c] = tile[p * 97 + c];
  }
}

// the way back: ringed channels-last rows (imgs, H + 2, W + 2, C) -> (imgs, C, H, W) (+ bias): reads run along c, writes along x
__global__ __launch_bounds__(256) void unring_rows_kernel(const float* __restrict__ src, const float* __restrict__ bias, float* __restrict__ dst,
                                                          int C, int H, int W) {
  __shared__ float tile[64 * 97];
  const int Wp = W + 2, cblocks = C / 96;
  const int img = blockIdx.z / cblocks, c0 = (blockIdx.z - img * cblocks) * 96;
  const int y = blockIdx.y, x0 = blockIdx.x * 64;
  const int tid = threadIdx.x;
  const int npx = W - x0 < 64 ? W - x0 : 64;
  const float* s = src + (((int64_t)img * (H + 2) + y + 1) * Wp + x0 + 1) * C + c0;
  for (int i = tid; i < npx * 96; i += 256) {
    const int p = i / 96, c = i - p * 96;
    tile[p * 97 + c] = s[(int64_t)p * C + c];
  }
  __syncthreads();
  const int lx = tid & 63, cw = tid >> 6;
  if (lx < npx) {
    float* d = dst + (((int64_t)img * C + c0) * H + y) * W + x0 + lx;
#pragma unroll 4
    for (int c = cw; c < 96; c += 4) d[(int64_t)c * H * W] = tile[lx * 97 + c] + (bias ? bias[c0 + c] : 0.f);
  }
}

}  // namespace

// the number of m ranges sdf_linear_dw_fwd wants for this shape (the caller provides `partial` of that many N x K fp32 tiles when > 1)
extern "C" int sdf_linear_dw_splits(int64_t M, int N, int K, int cv_C) {
  if (M <= 0 || N <= 0 || K <= 0 || N % TN || K % TK || cv_C < 0 || (cv_C > 0 && (cv_C % TK || K != 9 * cv_C))) return 0;
  const int64_t tiles = (int64_t)(N / TN) * (cv_C > 0 ? 3 * (cv_C / TK) : K / TK);
  int64_t want = (768 + tiles - 1) / tiles;                           // ~3 workgroups per compute unit in all
  const int64_t most = (M + 8 * MC - 1) / (8 * MC);                   // a range is at least 8 chunks long
  if (want > most) want = most;
  if (want < 1) want = 1;
  return (int)want;
}

extern "C" int sdf_linear_dw_fwd(const SdfLinearDwDesc* d, void* stream) {
  if (!d || !d->dy || !d->x || !d->dw) return SDF_E_NULL;
  if (d->M <= 0 || d->N <= 0 || d->K <= 0 || d->N % TN || d->K % TK || d->nsplit < 1) return SDF_E_SHAPE;
  const bool conv = d->cv_C > 0;
  if (d->cv_C < 0 || (conv && (d->cv_C % TK || d->K != 9 * d->cv_C || d->cv_Wp < 3))) return SDF_E_SHAPE;
  if (d->nsplit > 1 && !d->partial) return SDF_E_NULL;
  const int64_t lim = (int64_t)1 << 31;
  const int x_ld = conv ? d->cv_C : d->K;
  if (d->M * d->N * 4 >= lim || d->M * x_ld * 4 >= lim || (int64_t)d->N * d->K * 4 >= lim) return SDF_E_SHAPE;
  if (!sdf_aligned(d->dy, 16) || !sdf_aligned(d->x, 16) || !sdf_aligned(d->dw, 16) || (d->partial && !sdf_aligned(d->partial, 16))) return SDF_E_ALIGN;
  DwParams P;
  P.d = *d;
  P.tiles_n = d->N / TN;
  P.x_ld = x_ld;
  P.cblocks = conv ? d->cv_C / TK : 0;
  P.Wp = d->cv_Wp;
  P.tiles_k = conv ? 3 * P.cblocks : d->K / TK;
  const int64_t chunks = (d->M + MC - 1) / MC;
  P.rows_per_split = (int)((chunks + d->nsplit - 1) / d->nsplit) * MC;
  const int64_t wgs = (int64_t)d->nsplit * P.tiles_n * P.tiles_k;
  if (wgs >= lim) return SDF_E_SHAPE;
  hipStream_t s = sdf_stream(stream);
  if (conv) SDF_LAUNCH(linear_dw_kernel<3>, dim3((unsigned)wgs), dim3(256), 0, s, P);
  else SDF_LAUNCH(linear_dw_kernel<1>, dim3((unsigned)wgs), dim3(256), 0, s, P);
  SDF_LAUNCH_CHECK();
  if (d->nsplit > 1) {
    const int n4 = d->N * d->K / 4;
    SDF_LAUNCH(linear_dw_reduce_kernel, dim3((unsigned)((n4 + 15) / 16)), dim3(256), 0, s, d->partial, d->dw, n4, d->nsplit);
    SDF_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int sdf_ringed_rows_fwd(const float* src, float* dst, int imgs, int C, int H, int W, void* stream) {
  if (!src || !dst) return SDF_E_NULL;
  if (imgs < 1 || C < 96 || C % 96 || H < 1 || W < 1 || H + 2 > 65535 || (int64_t)imgs * (C / 96) > 65535) return SDF_E_SHAPE;
  const dim3 grid((unsigned)((W + 2 + 63) / 64), (unsigned)(H + 2), (unsigned)(imgs * (C / 96)));
  SDF_LAUNCH(ringed_rows_kernel, grid, dim3(256), 0, sdf_stream(stream), src, dst, C, H, W);
  SDF_LAUNCH_CHECK();
  return 0;
}

extern "C" int sdf_unring_rows_fwd(const float* src, const float* bias, float* dst, int imgs, int C, int H, int W, void* stream) {
  if (!src || !dst) return SDF_E_NULL;
  if (imgs < 1 || C < 96 || C % 96 || H < 1 || W < 1 || H > 65535 || (int64_t)imgs * (C / 96) > 65535) return SDF_E_SHAPE;
  const dim3 grid((unsigned)((W + 63) / 64), (unsigned)H, (unsigned)(imgs * (C / 96)));
  SDF_LAUNCH(unring_rows_kernel, grid, dim3(256), 0, sdf_stream(stream), src, bias, dst, C, H, W);
  SDF_LAUNCH_CHECK();
  return 0;
}
