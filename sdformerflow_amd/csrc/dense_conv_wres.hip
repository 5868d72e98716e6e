// Weight-resident 3x3 convolution of REAL-valued activations for gfx950 (stride 1, pad 1): the dense convolutions of the ANN
// patch embedding (reference models/STSwinNet/PatchEmbed.py:166-196 `PatchEmbedLocal`: head conv + 4 BatchNorm residual blocks,
// models/submodules.py:160-229 `ResidualBlock`), which the library served with an fp32 Winograd kernel at 2.7 ms per layer
// (profiles/r1o_config3.txt: 58 % of BASELINE config 3).
//
//   y[img, n, y, x] = act( alpha[n] * sum_{ky,kx,c} a[img, c, y+ky-1, x+kx-1] * w[n, c, ky, kx] + beta[n] (+ resid) )
//
// fp32 values on the 16-bit matrix pipe: every operand is the sum of two fp16 numbers, a = a_hi + a_lo (a_hi = fp16(a),
// a_lo = fp16(a - a_hi): 22 significant bits), and a product is three MFMAs, a_hi*w_hi + a_lo*w_hi + a_hi*w_lo, accumulated
// in fp32 inside the matrix cores (the dropped a_lo*w_lo term is 2^-22 of the product).  The activations LIVE in that form
// between the layers - the epilogue of one convolution writes the hi / lo pairs the next one multiplies, 4 bytes per value like
// fp32 - so no layer pays a conversion pass and BatchNorm (folded to alpha / beta), the residual add and the ReLU never
// touch memory on their own.
//
// A workgroup keeps the weights of one 32-column block of the output - both planes, the whole K = 9 * Cin - in LDS
// (2 x 32 x 864 fp16 = 110 KB), as spike_conv_wres.hip does.  The activations are handled per WAVE: each of the 12 wavefronts
// (three per SIMD) owns a tile of 4 x 8 output pixels (its 32 MFMA columns) and a private 6 x 10 pixel halo image of it in
// LDS; the nine taps are nine constant offsets into that image.  An image of all 96 channels in two planes would not fit beside
// the weights, so it is cut by CHANNEL: one step = 16 input channels (3.75 KB image, 9 taps x 3 MFMAs), the accumulators stay in
// registers across the steps of a tile and the epilogue runs after the last one.  Nothing is shared between waves but the
// weights, so the steady state has no barrier, no counter and no polling: a wave requests the image of step s + 2 from memory,
// multiplies step s, and writes the image of step s + 1 behind its own last fragment reads (a wave's LDS operations execute in
// order); the waves a SIMD hosts fill each other's gaps on the matrix pipe.  (The first version shared a 10 x 18 image between
// four waves: two LDS hand-overs per step cost as much as the step's 27 MFMAs - profiles/r2l_stamps_dense.txt.  Eight waves
// with padded 80-byte pixel records ran the pipe 56 % busy at 1.71 GHz, twelve with the swizzled 64-byte records below 63 %
// at 1.59 GHz: busy x clock is what the chip's power limit fixes, r2m / r2p_pmc_dense.txt.)
// The N / 32 workgroups that serve the column blocks of one tile range sit on one XCD and walk the range side by side: its
// activations leave HBM once (column blocks as an outer loop over the tensor read 2.7 x the algorithmic bytes, r2l_pmc_dense.txt).
// The weights are the MFMA's ROW operand and the pixels its column operand, so a lane's accumulator quads are four consecutive
// channels of one pixel - a 16-byte piece of the output as it stands, no transpose in the epilogue.
//
// Activation layout ("planes", sdf_pack_planes): [img][Cin/16][H][W] records of 64 bytes = 4 x { 4 x fp16 hi, 4 x fp16 lo } for
// 16 channels: a 16-byte piece is four channels complete, which is what one lane holds after the epilogue's quad transpose
// (one 16-byte store per four channels) and what the halo loader splits into the hi and lo halves of the LDS pixel record.
// LDS is conflict-free by construction without padding: a pixel record is 64 bytes = four 16-byte slots (hi 0-7, hi 8-15,
// lo 0-7, lo 8-15) stored at slot ^ (halo row & 3).  The 16 lanes of a ds_read_b128 group are four 4-pixel row segments on four
// consecutive halo rows; four consecutive pixels cover the four 64-byte quarters of a 256-byte bank row, and the four rows put
// the slot they all ask for at four different places of its quarter.  Weight rows are 2K + 16 bytes (4 x odd dwords).
#include "spike_mm.h"
#include "switches.h"
#include <stdlib.h>

#ifdef SDF_STAMP
// diagnostic build only (tools/stamp_dense.sh): cycle accounting of wave 0 of each group of workgroup 0
__device__ unsigned long long g_dense_stamp[32];
#define STAMP(var) var = __builtin_readcyclecounter()
#define STAMP_ADD(acc, a, b) acc += (b) - (a)
#else
#define STAMP(var)
#define STAMP_ADD(acc, a, b)
#endif

namespace sdfmm {
namespace {

constexpr int TW = 8;                           // a pixel block = 32 MFMA columns = 4 rows x 8 pixels
constexpr int HWID = TW + 2;                    // halo width
constexpr int NB = 32;                          // output columns of a workgroup pass
constexpr int REC = 64;                         // bytes of a pixel record in memory (16 channels, hi + lo)
constexpr int PS = 64;                          // pixel stride in the halo image: 4 slots of 16 B = hi 0-7, hi 8-15, lo 0-7, lo 8-15,
                                                // stored at slot ^ (halo row & 3) - see the bank note above
constexpr int RPB = HWID * PS;                  // halo row pitch (640 bytes)
constexpr uint32_t INV = 0x80000000u;
// A wave's tile is TPW pixel blocks stacked vertically (4 TPW rows x 8 pixels) over ONE halo image.  TPW = 1: twelve waves (three per
// SIMD), 3.75 KB images.  TPW = 2: eight waves (two per SIMD), 6.25 KB images; a weight fragment read from LDS feeds two MFMAs (one
// per block), the halo rows the two blocks share are loaded once, and every per-step instruction is amortised over 54 MFMAs instead
// of 27: 1.0 instead of 1.33 ds_read_b128 per MFMA (the LDS array was the busiest unit of the TPW = 1 kernel: 4 reads + 8 halo
// stores per 3 MFMAs x 12 waves is ~90 % of its cycles at full matrix rate, profiles/r2p_pmc_dense.txt).
template <int TPW>
struct Geo {
  static constexpr int TH = 4 * TPW, HH = TH + 2;
  static constexpr int NW = TPW == 1 ? 12 : 8;
  static constexpr int HALO = HH * RPB;
  static constexpr int PIECES = HH * HWID * 4;  // 16-byte pieces of a halo image (240 / 400)
  static constexpr int CPL = (PIECES + 63) / 64;
};

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) _Float16 h2;

struct DenseParams {
  SdfDenseConvDesc d;
  int wtiles;                                   // wave tiles of the whole output: imgs * ceil(H/4) * ceil(W/8)
  int ranges;                                   // contiguous tile ranges; each is served by N / 32 workgroups, one per column block
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)INV, 0x00020000);
}

typedef __attribute__((ext_vector_type(2))) float f2;
// four fp32 values -> one 16-byte piece {hi0..3, lo0..3}: hi = fp16(x), lo = fp16(x - hi), both round to nearest
// (v_cvt_pk_f16_f32); a value beyond the fp16 range becomes inf / NaN and stays visible downstream (no silent clamp)
__device__ __forceinline__ u32x4 piece_from(float4 o) {
  f2 a, b;
  a.x = o.x; a.y = o.y;
  b.x = o.z; b.y = o.w;
  const h2 ha = __builtin_convertvector(a, h2), hb = __builtin_convertvector(b, h2);
  f2 ra, rb;
  ra.x = a.x - (float)ha.x; ra.y = a.y - (float)ha.y; rb.x = b.x - (float)hb.x; rb.y = b.y - (float)hb.y;
  const h2 la = __builtin_convertvector(ra, h2), lb = __builtin_convertvector(rb, h2);
  u32x4 v;
  v.x = __builtin_bit_cast(uint32_t, ha); v.y = __builtin_bit_cast(uint32_t, hb);
  v.z = __builtin_bit_cast(uint32_t, la); v.w = __builtin_bit_cast(uint32_t, lb);
  return v;
}
__device__ __forceinline__ float4 piece_to(u32x4 v) {
  const h2 ha = __builtin_bit_cast(h2, (uint32_t)v.x), hb = __builtin_bit_cast(h2, (uint32_t)v.y);
  const h2 la = __builtin_bit_cast(h2, (uint32_t)v.z), lb = __builtin_bit_cast(h2, (uint32_t)v.w);
  return make_float4((float)ha.x + (float)la.x, (float)ha.y + (float)la.y, (float)hb.x + (float)lb.x, (float)hb.y + (float)lb.y);
}

template <int CCH, int TPW>
__global__ __launch_bounds__(64 * Geo<TPW>::NW) void dense_conv_wres_kernel(DenseParams P) {
  constexpr int TH = Geo<TPW>::TH, NW = Geo<TPW>::NW, HALO = Geo<TPW>::HALO, PIECES = Geo<TPW>::PIECES,
                CPL = Geo<TPW>::CPL;
  constexpr int K = 9 * 16 * CCH;
  constexpr int WP = 2 * K + 16;                                      // weight row pitch (bytes): 4 x odd dwords
  constexpr int W_BYTES = 2 * NB * WP;
  constexpr int PAR = 2 * NB * 4;
  static_assert((WP / 4) % 8 == 4, "weight row pitch must be 4 x odd dwords");
  static_assert(W_BYTES + NW * HALO + PAR <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(16))) uint8_t smem[W_BYTES + NW * HALO + PAR];
  uint8_t* W_s = smem;
  float* par_s = reinterpret_cast<float*>(smem + W_BYTES + NW * HALO);

  const SdfDenseConvDesc& d = P.d;
  const int H = d.H, W = d.W, N = d.N;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const bool has_res = d.resid != nullptr;
  const bool relu = d.relu != 0, out_f32 = d.out_f32 != 0;
  const int nch = N >> 4;                                             // channel records of the output / residual planes
  const int ncb = N / NB;
  const int xrec = d.x_records > 0 ? d.x_records : CCH;               // records per image of the tensor x is a slice of

  // workgroup -> (tile range, column block).  The ncb workgroups of a range run side by side on ONE XCD, so the activations
  // they all read come out of HBM once and out of that L2 afterwards (column blocks as an outer loop over the whole tensor read
  // 2.7 x the algorithmic bytes, profiles/r2l_pmc_dense.txt).  Full grid: workgroup p sits on XCD p % 8 as its (p / 8)-th.
  int range, cb;
  if (gridDim.x == 256 && ncb <= 32) {
    const int x = blockIdx.x & 7, k = blockIdx.x >> 3, m = 32 / ncb;   // m whole groups per XCD, the rest pooled across XCDs
    if (k < m * ncb) { range = x * m + k / ncb; cb = k % ncb; }
    else { const int e = x * (32 - m * ncb) + (k - m * ncb); range = 8 * m + e / ncb; cb = e % ncb; }
  } else {
    range = blockIdx.x / ncb; cb = blockIdx.x % ncb;
  }
  if (range >= P.ranges) return;
  const int n0 = cb * NB;
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH, tiles_img = tiles_x * tiles_y;
  const int base = P.wtiles / P.ranges, rem = P.wtiles % P.ranges;
  const int t_begin = range * base + (range < rem ? range : rem);
  const int n_my = base + (range < rem ? 1 : 0);

  // ---------------- the column block's weights, once ----------------
  {
    constexpr int KC8 = K / 8;                                         // 16-byte pieces per weight row
    constexpr int WCH = 2 * NB * KC8;
    const __amdgpu_buffer_rsrc_t W_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(d.w), 0, 2 * N * K * 2, 0x00020000);
    constexpr int NT = 64 * NW;
    constexpr int WB = 7, NBATCH = (WCH + NT * WB - 1) / (NT * WB);     // batches of 7 pieces per lane in flight
#pragma unroll 1
    for (int b = 0; b < NBATCH; ++b) {
      u32x4 wv[WB];
#pragma unroll
      for (int i = 0; i < WB; ++i) {
        const int c = tid + NT * (b * WB + i);
        const int cc = c < WCH ? c : 0;
        const int row = cc / KC8, kc = cc - row * KC8;                 // row = p * 32 + n
        const int p = row / NB, n = row - p * NB;
        wv[i] = __builtin_amdgcn_raw_buffer_load_b128(W_rs, c < WCH ? (uint32_t)(((p * N + n0 + n) * K + kc * 8) * 2) : INV, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < WB; ++i) {
        const int c = tid + NT * (b * WB + i);
        const int cc = c < WCH ? c : 0;
        const int row = cc / KC8, kc = cc - row * KC8;
        if (c < WCH) *reinterpret_cast<u32x4*>(W_s + row * WP + kc * 16) = wv[i];
      }
    }
    if (tid < 2 * NB) {
      const int which = tid / NB, n = tid - which * NB;
      const float asc = d.acc_scale != 0.f ? d.acc_scale : 1.f;          // 1 / weight scale, a power of two: folds into alpha exactly
      float v = which == 0 ? asc : 0.f;
      if (which == 0 && d.alpha) v = d.alpha[n0 + n] * asc;
      if (which == 1 && d.beta) v = d.beta[n0 + n];
      par_s[tid] = v;
    }
  }
  __syncthreads();

  // this lane's pieces of its wave's halo image: LDS offset of the hi half, byte offset relative to the tile's origin record,
  // (dy, dx).  Lanes without a fourth piece request a copy of their third and write nothing.
  uint32_t h_lds[CPL];
  int h_rel[CPL], h_yx[CPL];
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const int c = lane + 64 * i;
    const int cc = c < PIECES ? c : c - 64;
    const int hy = cc / (HWID * 4), r = cc - hy * (HWID * 4);
    const int px = r >> 2, j = r & 3;
    h_lds[i] = c < PIECES ? (uint32_t)(hy * RPB + px * PS + (((j >> 1) ^ (hy & 3)) << 4) + 8 * (j & 1)) : 0xFFFFFFFFu;
    h_rel[i] = ((hy - 1) * W + (px - 1)) * REC + 16 * j;
    h_yx[i] = ((hy - 1) << 16) | ((px - 1) & 0xFFFF);
  }
  const __amdgpu_buffer_rsrc_t A_rs = make_rsrc(d.x);
  // two register sets: the halo of step s + 2 is requested at the start of step s and written to LDS at the end of step s + 1
  // (a step is ~0.5 us of MFMAs, a load under a full chip takes longer than that)
  u32x4 hreg[2][CPL];
  auto halo_load = [&](u32x4 (&hr)[CPL], int img, int ch, int y0, int x0) __attribute__((always_inline)) {
    const uint32_t org = (uint32_t)((((img * xrec + ch) * H + y0) * W + x0) * REC);
    if (y0 >= 1 && x0 >= 1 && y0 + TH + 1 <= H && x0 + TW + 1 <= W) {  // interior tile: every halo pixel exists
#pragma unroll
      for (int i = 0; i < CPL; ++i) hr[i] = __builtin_amdgcn_raw_buffer_load_b128(A_rs, org + (uint32_t)h_rel[i], 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < CPL; ++i) {
        const int yy = y0 + (h_yx[i] >> 16), xx = x0 + (int)(int16_t)(h_yx[i] & 0xFFFF);
        const bool ok = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
        hr[i] = __builtin_amdgcn_raw_buffer_load_b128(A_rs, ok ? org + (uint32_t)h_rel[i] : INV, 0, 0);
      }
    }
  };
  uint8_t* H_s = smem + W_BYTES + wave * HALO;                        // this wave's own image: no other wave reads or writes it
  auto halo_store = [&](const u32x4 (&hr)[CPL]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      if (h_lds[i] != 0xFFFFFFFFu) {
        *reinterpret_cast<uint2*>(H_s + h_lds[i]) = make_uint2(hr[i].x, hr[i].y);          // 4 x hi
        *reinterpret_cast<uint2*>(H_s + (h_lds[i] ^ 32u)) = make_uint2(hr[i].z, hr[i].w);  // 4 x lo: slot ^ 2
      }
    }
  };

  // a position in the walk over (image, tile row, tile column); the tiles of a wave are NW apart.  (Handing the tiles out
  // through an LDS counter instead measured no faster - the waves of a workgroup already finish together.)
  struct Pos { int img, ty, tx; };
  auto pos_of = [&](int tile) __attribute__((always_inline)) {
    Pos p;
    p.img = tile / tiles_img;
    const int tl = tile - p.img * tiles_img;
    p.ty = tl / tiles_x; p.tx = tl - p.ty * tiles_x;
    return p;
  };
  auto advance = [&](Pos& p) __attribute__((always_inline)) {
    p.tx += NW;
    while (p.tx >= tiles_x) { p.tx -= tiles_x; ++p.ty; }
    while (p.ty >= tiles_y) { p.ty -= tiles_y; ++p.img; }
  };

  // fragment addresses of this lane.  The WEIGHTS are the MFMA's row operand and the wave's 32 pixels (4 rows x 8) its column
  // operand: accumulator slots 4q..4q+3 of a lane are then output channels 8q + 4*(lane / 32) + 0..3 of pixel lane % 32 -
  // four consecutive channels, i.e. one 16-byte piece of the planes (or of an fp32 channels-last row) with no transpose.
  int ln = lane;
  asm volatile("" : "+v"(ln));
  const int l31 = ln & 31, lh = ln >> 5;
  // hi fragment of tap row ky: slot lh ^ ((tile row + ky) & 3) of the lane's pixel record; the lo fragment is that address ^ 32
  uint32_t a_row[3];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) a_row[ky] = (uint32_t)(((l31 >> 3) + ky) * RPB + (l31 & 7) * PS + ((lh ^ (((l31 >> 3) + ky) & 3)) << 4));
  const uint32_t w_lane = (uint32_t)(l31 * WP + 16 * lh);
  const __amdgpu_buffer_rsrc_t out_rs = make_rsrc(d.out), res_rs = make_rsrc(d.resid);

#ifdef SDF_STAMP
  unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, a_issue = 0, a_mfma = 0, a_epi = 0, a_store = 0, nstep = 0;
  const unsigned long long kstart = __builtin_readcyclecounter(), rstart = __builtin_amdgcn_s_memrealtime();
#endif
  // ---------------- this wave's tiles of the range: wave, wave + 8, ...; steps = tiles x channel records ----------------
  const int my_items = n_my > wave ? (n_my - wave + NW - 1) / NW : 0;
  const int S = my_items * CCH;
  Pos pc = pos_of(t_begin + wave);                                     // position of the step being multiplied
  Pos pp = pc;                                                         // position / record / count of the next step to REQUEST
  int chp = 0, sp = 0;
  auto request = [&](u32x4 (&hr)[CPL]) __attribute__((always_inline)) {
    if (sp < S) {
      halo_load(hr, pp.img, chp, pp.ty * TH, pp.tx * TW);
      ++sp;
      if (++chp == CCH) { chp = 0; advance(pp); }
    }
  };
  if (S > 0) {
    request(hreg[0]);
    halo_store(hreg[0]);
    request(hreg[1]);
  }
  f32x16 acc[TPW];
  uint32_t pixoff[TPW], pixres[TPW];                                  // byte offset of this lane's pixel in out / resid (first piece), or INV
  u32x4 rs[TPW][4];
#pragma unroll
  for (int t = 0; t < TPW; ++t) pixoff[t] = pixres[t] = INV;
  int ch = 0;
  // one step: request step s + 2 into `hnew`, multiply step s, then write step s + 1 (requested a step ago, in `hold`)
  auto step = [&](int s, u32x4 (&hnew)[CPL], const u32x4 (&hold)[CPL]) __attribute__((always_inline)) {
    STAMP(s0);
    request(hnew);
    if (ch == 0) {
#pragma unroll
      for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    }
    if (ch == CCH - 1) {
      // the epilogue's addresses and its residual, requested before the last MFMAs: piece q of the lane = channels
      // n0 + 8q + 4*lh .. + 3 = record 2cb + (q >> 1), piece 2(q & 1) + lh of that record
      const int img = pc.img;
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const int y = pc.ty * TH + 4 * t + (l31 >> 3), x = pc.tx * TW + (l31 & 7);
        const bool ok = y < H && x < W;
        const uint32_t pix = (uint32_t)((img * H + y) * W + x);
        pixres[t] = ok ? (uint32_t)(img * nch + 2 * cb) * (uint32_t)(H * W) * REC + (uint32_t)(y * W + x) * REC + (uint32_t)lh * 16u : INV;
        pixoff[t] = !ok ? INV : out_f32 ? pix * (uint32_t)N * 4u + (uint32_t)(n0 + 4 * lh) * 4u : pixres[t];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          rs[t][q] = u32x4{0u, 0u, 0u, 0u};
          if (has_res)
            rs[t][q] = __builtin_amdgcn_raw_buffer_load_b128(res_rs, ok ? pixres[t] + (uint32_t)((q >> 1) * H * W * REC + (q & 1) * 32) : INV, 0, 0);
        }
      }
    }
    STAMP(s1);

    // ------------------------------ MFMA phase: 9 taps x 3 products ------------------------------
    constexpr int PF = 2;                                             // taps of fragments in flight
    bf16x8 fa[PF + 1][TPW][2], fb[PF + 1][2];
    const uint32_t w_step = w_lane + (uint32_t)ch * (9 * 32);
    auto frag = [&](int tap, int set) __attribute__((always_inline)) {
      const int ky = tap / 3, kx = tap - 3 * ky;
      fb[set][0] = *reinterpret_cast<const bf16x8*>(W_s + w_step + tap * 32);
      fb[set][1] = *reinterpret_cast<const bf16x8*>(W_s + w_step + (NB * WP + tap * 32));
#pragma unroll
      for (int t = 0; t < TPW; ++t) {                                  // block t sits 4 halo rows lower: same swizzle (4 = 0 mod 4)
        fa[set][t][0] = *reinterpret_cast<const bf16x8*>(H_s + a_row[ky] + (kx * PS + t * 4 * RPB));
        fa[set][t][1] = *reinterpret_cast<const bf16x8*>(H_s + (a_row[ky] ^ 32u) + (kx * PS + t * 4 * RPB));
      }
    };
#pragma unroll
    for (int i = 0; i < PF; ++i) frag(i, i);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap + PF < 9) frag(tap + PF, (tap + PF) % (PF + 1));
      __builtin_amdgcn_sched_barrier(0);
      const int f = tap % (PF + 1);
#pragma unroll
      for (int t = 0; t < TPW; ++t) acc[t] = mma<2>(fb[f][0], fa[f][t][1], acc[t]);   // w_hi * a_lo: small terms first
#pragma unroll
      for (int t = 0; t < TPW; ++t) acc[t] = mma<2>(fb[f][1], fa[f][t][0], acc[t]);   // w_lo * a_hi
#pragma unroll
      for (int t = 0; t < TPW; ++t) acc[t] = mma<2>(fb[f][0], fa[f][t][0], acc[t]);   // w_hi * a_hi
      __builtin_amdgcn_sched_barrier(0);
    }
    STAMP(s2);
    // the next image goes into LDS behind this step's last fragment reads (a wave's LDS operations execute in order)
    if (s + 1 < S) halo_store(hold);
    STAMP(s3);

    // ------------------------------ epilogue after the last channel step ------------------------------
    if (ch == CCH - 1) {
#pragma unroll
      for (int t = 0; t < TPW; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 al4 = *reinterpret_cast<const float4*>(par_s + 8 * q + 4 * lh);
        const float4 be4 = *reinterpret_cast<const float4*>(par_s + NB + 8 * q + 4 * lh);
        float4 o;
        o.x = __builtin_fmaf(acc[t][q * 4 + 0], al4.x, be4.x); o.y = __builtin_fmaf(acc[t][q * 4 + 1], al4.y, be4.y);
        o.z = __builtin_fmaf(acc[t][q * 4 + 2], al4.z, be4.z); o.w = __builtin_fmaf(acc[t][q * 4 + 3], al4.w, be4.w);
        if (has_res) {
          const float4 r = piece_to(rs[t][q]);
          o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        if (relu) {
          o.x = __builtin_fmaxf(o.x, 0.f); o.y = __builtin_fmaxf(o.y, 0.f);
          o.z = __builtin_fmaxf(o.z, 0.f); o.w = __builtin_fmaxf(o.w, 0.f);
        }
        u32x4 st;
        uint32_t off;
        if (out_f32) {
          st.x = __float_as_uint(o.x); st.y = __float_as_uint(o.y); st.z = __float_as_uint(o.z); st.w = __float_as_uint(o.w);
          off = pixoff[t] + (uint32_t)(q * 32);
        } else {
          st = piece_from(o);
          off = pixoff[t] + (uint32_t)((q >> 1) * H * W * REC + (q & 1) * 32);
        }
        __builtin_amdgcn_raw_buffer_store_b128(st, out_rs, pixoff[t] != INV ? off : INV, 0, 0);
      }
    }
    if (++ch == CCH) { ch = 0; advance(pc); }
    STAMP(s4);
    STAMP_ADD(a_issue, s0, s1); STAMP_ADD(a_mfma, s1, s2); STAMP_ADD(a_store, s2, s3); STAMP_ADD(a_epi, s3, s4);
#ifdef SDF_STAMP
    ++nstep;
#endif
  };
#pragma unroll 1
  for (int s = 0; s < S; s += 2) {
    step(s, hreg[0], hreg[1]);
    if (s + 1 < S) step(s + 1, hreg[1], hreg[0]);
  }
#ifdef SDF_STAMP
  if (blockIdx.x == 0 && (tid == 0 || tid == 256)) {
    unsigned long long* o = g_dense_stamp + (tid ? 16 : 0);
    o[0] = a_issue; o[1] = 0; o[2] = a_mfma; o[3] = a_epi; o[4] = 0; o[5] = a_store; o[6] = nstep;
    o[7] = __builtin_readcyclecounter() - kstart; o[8] = __builtin_amdgcn_s_memrealtime() - rstart;
  }
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// NCHW fp32 <-> planes.  One thread per (pixel, 4-channel piece); channels beyond C are zero in the planes.
__global__ __launch_bounds__(256) void pack_planes_kernel(const float* __restrict__ x, u32x4* __restrict__ planes, int imgs, int C,
                                                          int nch, int H, int W) {
  const int64_t hw = (int64_t)H * W;
  const int64_t total = (int64_t)imgs * nch * 4 * hw;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t pix = i % hw;                                        // pixel fastest: the four strided reads are coalesced
    const int64_t r = i / hw;
    const int j = (int)(r & 3);
    const int64_t ic = r >> 2;                                         // img * nch + record
    const int img = (int)(ic / nch), rec = (int)(ic - (int64_t)img * nch);
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = rec * 16 + 4 * j + e;
      v[e] = c < C ? x[((int64_t)img * C + c) * hw + pix] : 0.f;
    }
    planes[(ic * hw + pix) * 4 + j] = piece_from(make_float4(v[0], v[1], v[2], v[3]));
  }
}

__global__ __launch_bounds__(256) void unpack_planes_kernel(const u32x4* __restrict__ planes, float* __restrict__ x, int imgs, int C,
                                                            int nch, int H, int W) {
  const int64_t hw = (int64_t)H * W;
  const int64_t total = (int64_t)imgs * nch * 4 * hw;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t pix = i % hw;
    const int64_t r = i / hw;
    const int j = (int)(r & 3);
    const int64_t ic = r >> 2;
    const int img = (int)(ic / nch), rec = (int)(ic - (int64_t)img * nch);
    const float4 o = piece_to(planes[(ic * hw + pix) * 4 + j]);
    const float v[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = rec * 16 + 4 * j + e;
      if (c < C) x[((int64_t)img * C + c) * hw + pix] = v[e];
    }
  }
}

// Bilinear x2 upsampling (align_corners false: F.interpolate(scale_factor=2, mode="bilinear"), reference models/submodules.py:117-157
// `UpsampleConvLayer`) of an (imgs, C, h, w) fp32 tensor with arbitrary strides, written as records rec0 .. of planes
// [imgs][rec_total][2h][2w]: output pixel 2k takes 0.25 / 0.75 of inputs k-1 / k, pixel 2k+1 takes 0.75 / 0.25 of k / k+1,
// indices clamped to the image.
// ZI (round 6): ZERO INSERTION instead of interpolation - output pixel (2k, 2l) = input (k, l), every other pixel 0: the input of a stride-2
// transposed convolution written as a stride-1 correlation (the SEW decoders, engine_sew._deconv_dense_fwd), straight into the planes.
template <bool CL, bool ZI = false>
__global__ __launch_bounds__(256) void pack_planes_up2_kernel(const float* __restrict__ x, u32x4* __restrict__ planes, int imgs, int C,
                                                              int h, int w, int64_t sn, int64_t sc, int64_t sh, int64_t sw, int rec0,
                                                              int rec_total) {
  // CL (channel stride 1, C % 4 == 0): one thread per (pixel, piece), pieces fastest - four 16-byte loads, one 16-byte store,
  // both contiguous across the wave.  Otherwise one thread per (pixel, record), pixels fastest: every scalar load is contiguous
  // across the wave and a thread writes its record's 64 bytes whole.
  const int H = 2 * h, W = 2 * w, nch = (C + 15) / 16;
  const int64_t hw = (int64_t)H * W;
  const int64_t total = (int64_t)imgs * nch * hw * (CL ? 4 : 1);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int64_t pix, ic;
    int j0 = 0;
    if (CL) { j0 = (int)(i & 3); pix = (i >> 2) % hw; ic = (i >> 2) / hw; }
    else { pix = i % hw; ic = i / hw; }
    const int img = (int)(ic / nch), rec = (int)(ic - (int64_t)img * nch);
    const int oy = (int)(pix / W), ox = (int)(pix - (int64_t)oy * W);
    const int ky = oy >> 1, kx = ox >> 1;
    const int y0 = (oy & 1) ? ky : (ky > 0 ? ky - 1 : 0), y1 = (oy & 1) ? (ky + 1 < h ? ky + 1 : h - 1) : ky;
    const int x0 = (ox & 1) ? kx : (kx > 0 ? kx - 1 : 0), x1 = (ox & 1) ? (kx + 1 < w ? kx + 1 : w - 1) : kx;
    const float wy1 = (oy & 1) ? 0.25f : (ky > 0 ? 0.75f : 0.f), wx1 = (ox & 1) ? 0.25f : (kx > 0 ? 0.75f : 0.f);
    const float wy0 = 1.f - wy1, wx0 = 1.f - wx1;
    const int64_t o00 = y0 * sh + x0 * sw, o01 = y0 * sh + x1 * sw, o10 = y1 * sh + x0 * sw, o11 = y1 * sh + x1 * sw;
    u32x4* dst = planes + (((int64_t)img * rec_total + rec0 + rec) * hw + pix) * 4;
    const float* bimg = x + img * sn;
    if constexpr (ZI) {
      const bool on = !(oy & 1) && !(ox & 1);
      const int64_t o = ky * sh + kx * sw;
      if (CL) {
        const int c = rec * 16 + 4 * j0;
        dst[j0] = piece_from((on && c < C) ? *reinterpret_cast<const float4*>(bimg + c + o) : make_float4(0.f, 0.f, 0.f, 0.f));
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int c = rec * 16 + 4 * j + e;
            v[e] = (on && c < C) ? bimg[c * sc + o] : 0.f;
          }
          dst[j] = piece_from(make_float4(v[0], v[1], v[2], v[3]));
        }
      }
      continue;
    }
    if (CL) {
      const int c = rec * 16 + 4 * j0;
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < C) {
        const float* b = bimg + c;
        const float4 a00 = *reinterpret_cast<const float4*>(b + o00), a01 = *reinterpret_cast<const float4*>(b + o01);
        const float4 a10 = *reinterpret_cast<const float4*>(b + o10), a11 = *reinterpret_cast<const float4*>(b + o11);
        o.x = wy0 * (wx0 * a00.x + wx1 * a01.x) + wy1 * (wx0 * a10.x + wx1 * a11.x);
        o.y = wy0 * (wx0 * a00.y + wx1 * a01.y) + wy1 * (wx0 * a10.y + wx1 * a11.y);
        o.z = wy0 * (wx0 * a00.z + wx1 * a01.z) + wy1 * (wx0 * a10.z + wx1 * a11.z);
        o.w = wy0 * (wx0 * a00.w + wx1 * a01.w) + wy1 * (wx0 * a10.w + wx1 * a11.w);
      }
      dst[j0] = piece_from(o);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = rec * 16 + 4 * j + e;
          float o = 0.f;
          if (c < C) {
            const float* b = bimg + c * sc;
            o = wy0 * (wx0 * b[o00] + wx1 * b[o01]) + wy1 * (wx0 * b[o10] + wx1 * b[o11]);
          }
          v[e] = o;
        }
        dst[j] = piece_from(make_float4(v[0], v[1], v[2], v[3]));
      }
    }
  }
}

}  // namespace
}  // namespace sdfmm

extern "C" int sdf_dense_conv3x3_fwd(const SdfDenseConvDesc* d, void* stream) {
  using namespace sdfmm;
  if (!d || !d->x || !d->w || !d->out) return SDF_E_NULL;
  if (d->imgs <= 0 || d->H <= 0 || d->W <= 0 || d->N <= 0 || d->N % NB) return SDF_E_SHAPE;
  if (d->cin_records != 1 && d->cin_records != 6) return SDF_E_SHAPE;            // instantiated: Cin <= 16 (head), Cin = 96
  if (!sdf_aligned(d->x, 16) || !sdf_aligned(d->w, 16) || !sdf_aligned(d->out, 16) || (d->resid && !sdf_aligned(d->resid, 16)))
    return SDF_E_ALIGN;
  if (d->x_records != 0 && d->x_records < d->cin_records) return SDF_E_SHAPE;
  const int64_t lim = (int64_t)1 << 31, px = (int64_t)d->imgs * d->H * d->W;
  if (px * (d->x_records > 0 ? d->x_records : d->cin_records) * REC >= lim || px * d->N * 4 >= lim) return SDF_E_SHAPE;   // 31-bit buffer offsets
  DenseParams P;
  P.d = *d;
  const int ncb = d->N / NB;
  if (ncb > 32) return SDF_E_SHAPE;
  // two pixel blocks per wave where the map is tall enough for 8-row tiles to waste little (a 36-row map would multiply 40 rows)
  const int tpw_env = [] { const char* e = sdf_sw(SW_DENSE_TPW); return e ? atoi(e) : 0; }();
  const int waste8 = ((d->H + 7) / 8) * 8 - d->H, waste4 = ((d->H + 3) / 4) * 4 - d->H;
  int tpw = (d->cin_records == 6 && (waste8 == waste4 || d->H >= 128)) ? 2 : 1;
  if (d->cin_records == 6 && (tpw_env == 1 || tpw_env == 2)) tpw = tpw_env;
  const int TH = 4 * tpw, NW = tpw == 1 ? Geo<1>::NW : Geo<2>::NW;
  P.wtiles = (int)((int64_t)d->imgs * ((d->H + TH - 1) / TH) * ((d->W + TW - 1) / TW));
  const int want = (P.wtiles + NW - 1) / NW;                           // a workgroup wants at least one tile per wave
  P.ranges = want < 256 / ncb ? want : 256 / ncb;
  const int G = P.ranges == 256 / ncb ? 256 : P.ranges * ncb;
  hipStream_t s = sdf_stream(stream);
  if (d->cin_records == 1) SDF_LAUNCH((dense_conv_wres_kernel<1, 1>), dim3(G), dim3(64 * NW), 0, s, P);
  else if (tpw == 1) SDF_LAUNCH((dense_conv_wres_kernel<6, 1>), dim3(G), dim3(64 * NW), 0, s, P);
  else SDF_LAUNCH((dense_conv_wres_kernel<6, 2>), dim3(G), dim3(64 * NW), 0, s, P);
  SDF_LAUNCH_CHECK();
  return 0;
}

#ifdef SDF_STAMP
extern "C" int sdf_debug_read_stamps_dense(unsigned long long* host32) {
  return (int)hipMemcpyFromSymbol(host32, HIP_SYMBOL(g_dense_stamp), 32 * sizeof(unsigned long long));
}
#endif

static int planes_grid(int64_t total) {
  const int64_t b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : b > 65536 ? 65536 : b);
}

extern "C" int sdf_pack_planes(const float* x, void* planes, int imgs, int C, int H, int W, void* stream) {
  if (!x || !planes) return SDF_E_NULL;
  if (imgs <= 0 || C <= 0 || H <= 0 || W <= 0) return SDF_E_SHAPE;
  if (!sdf_aligned(planes, 16)) return SDF_E_ALIGN;
  const int nch = (C + 15) / 16;
  const int64_t total = (int64_t)imgs * nch * 4 * H * W;
  SDF_LAUNCH(sdfmm::pack_planes_kernel, dim3(planes_grid(total)), dim3(256), 0, sdf_stream(stream), x,
                     reinterpret_cast<sdfmm::u32x4*>(planes), imgs, C, nch, H, W);
  SDF_LAUNCH_CHECK();
  return 0;
}

extern "C" int sdf_pack_planes_up2(const float* x, void* planes, int imgs, int C, int h, int w, int64_t sn, int64_t sc, int64_t sh,
                                   int64_t sw, int rec0, int rec_total, void* stream) {
  if (!x || !planes) return SDF_E_NULL;
  const int nch = (C + 15) / 16;
  if (imgs <= 0 || C <= 0 || h <= 0 || w <= 0 || rec0 < 0 || rec0 + nch > rec_total) return SDF_E_SHAPE;
  if (!sdf_aligned(planes, 16)) return SDF_E_ALIGN;
  const bool cl = sc == 1 && C % 4 == 0 && sn % 4 == 0 && sh % 4 == 0 && sw % 4 == 0 && sdf_aligned(x, 16);
  const int64_t total = (int64_t)imgs * nch * (2 * h) * (2 * w) * (cl ? 4 : 1);
  if (cl)
    SDF_LAUNCH(sdfmm::pack_planes_up2_kernel<true>, dim3(planes_grid(total)), dim3(256), 0, sdf_stream(stream), x,
                       reinterpret_cast<sdfmm::u32x4*>(planes), imgs, C, h, w, sn, sc, sh, sw, rec0, rec_total);
  else
    SDF_LAUNCH(sdfmm::pack_planes_up2_kernel<false>, dim3(planes_grid(total)), dim3(256), 0, sdf_stream(stream), x,
                       reinterpret_cast<sdfmm::u32x4*>(planes), imgs, C, h, w, sn, sc, sh, sw, rec0, rec_total);
  SDF_LAUNCH_CHECK();
  return 0;
}

extern "C" int sdf_pack_planes_zero_up2(const float* x, void* planes, int imgs, int C, int h, int w, int64_t sn, int64_t sc, int64_t sh,
                                        int64_t sw, int rec0, int rec_total, void* stream) {
  if (!x || !planes) return SDF_E_NULL;
  const int nch = (C + 15) / 16;
  if (imgs <= 0 || C <= 0 || h <= 0 || w <= 0 || rec0 < 0 || rec0 + nch > rec_total) return SDF_E_SHAPE;
  if (!sdf_aligned(planes, 16)) return SDF_E_ALIGN;
  const bool cl = sc == 1 && C % 4 == 0 && sn % 4 == 0 && sh % 4 == 0 && sw % 4 == 0 && sdf_aligned(x, 16);
  const int64_t total = (int64_t)imgs * nch * (2 * h) * (2 * w) * (cl ? 4 : 1);
  if (cl)
    SDF_LAUNCH((sdfmm::pack_planes_up2_kernel<true, true>), dim3(planes_grid(total)), dim3(256), 0, sdf_stream(stream), x,
               reinterpret_cast<sdfmm::u32x4*>(planes), imgs, C, h, w, sn, sc, sh, sw, rec0, rec_total);
  else
    SDF_LAUNCH((sdfmm::pack_planes_up2_kernel<false, true>), dim3(planes_grid(total)), dim3(256), 0, sdf_stream(stream), x,
               reinterpret_cast<sdfmm::u32x4*>(planes), imgs, C, h, w, sn, sc, sh, sw, rec0, rec_total);
  SDF_LAUNCH_CHECK();
  return 0;
}

extern "C" int sdf_unpack_planes(const void* planes, float* x, int imgs, int C, int H, int W, void* stream) {
  if (!x || !planes) return SDF_E_NULL;
  if (imgs <= 0 || C <= 0 || H <= 0 || W <= 0) return SDF_E_SHAPE;
  if (!sdf_aligned(planes, 16)) return SDF_E_ALIGN;
  const int nch = (C + 15) / 16;
  const int64_t total = (int64_t)imgs * nch * 4 * H * W;
  SDF_LAUNCH(sdfmm::unpack_planes_kernel, dim3(planes_grid(total)), dim3(256), 0, sdf_stream(stream),
                     reinterpret_cast<const sdfmm::u32x4*>(planes), x, imgs, C, nch, H, W);
  SDF_LAUNCH_CHECK();
  return 0;
}
