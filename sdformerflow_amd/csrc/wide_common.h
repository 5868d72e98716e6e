// Helpers shared by the int8-digit spike kernels of the wide stages (ms_wide.hip) and of the small-M products (ms_smallm.hip):
// buffer resources, the in-place int8 MFMA, the digit recombination, spike bits -> bytes and the quad transpose.
#pragma once
#include "spike_mm.h"

namespace sdfmm {

// ---- launch parameters of the position-major product and of the attention front (ms_wide.hip, ms_res.hip) ----
struct WidePmParams {
  const uint8_t* A;          // row-major u8 [rows][K], or tiled, or (zsrc) the gated spikes E, flat
  int a_tiled;
  const int32_t* zsrc;       // head scramble: per activation row the byte offset of (k-group 0, byte 0) in E; null = plain rows
  uint32_t zg_G;             // head scramble: bytes between two k-groups (T' * N1 * 32)
  const int8_t* W;           // digit planes [3][N][K]
  const float* cscale;       // (N) power-of-two scale of every output channel
  int N, K, HW;
  int64_t P;                 // positions = B * HW; rows = P * T in (B, T, HW) order
  const float *bias, *alpha, *beta;
  float* x;                  // fp32 epilogue: out = resid, row stride ldo
  int ldo;
  uint8_t* out_spike;        // neuron epilogue: u8 [rows][ldsp] or tiled (K = ldsp)
  int ldsp, out_tiled;
  SdfNeuronCfg sn;
  float inv_tau;
  int ncg, nrg, nunits, passes;   // a workgroup walks `passes` row groups (grid = ncg x ceil(nrg / passes))
  // 3x3 / stride 1 / pad 1 convolution on an NHWC u8 image batch (imgs = B * T, position = pixel): A = the image, K = 9 Cin in
  // (tap, channel) order; cv_cpt = 128-deep chunks per tap (0 = not a convolution)
  int cv_H, cv_W, cv_Cin, cv_cpt;
  // split-K: the workgroup grid has a third factor, K range ks covers chunks [ks * cps, (ks + 1) * cps); EPI 4 stores the raw fp32
  // sums (digit scale applied) to partial[ks][row][N] for wide_reduce_kernel
  int ksplit, cps;
  float* partial;
  int no_resid;              // fp32 epilogue without a shortcut: out = BN(...) (patch merging writes a new tensor)
  int res_stage;             // set by the host: a narrow stage (C <= 192) - the weight-resident row-loop kernel (ms_res.hip) may take it
  // stride-2 3x3 transposed convolution as ONE product (ms_res.hip, AM = 3): a row = input pixel (img, a, b), K = the 2 x 2 input
  // neighbourhood (a..a+1, b..b+1) in patch-merging's quadrant order (cv_H / cv_W / cv_Cin = the INPUT image), N = 4 dc_cout columns =
  // the four output pixels (2a + py, 2b + px) of that row's 2 x 2 output block (column = (2 py + px) * dc_cout + co): x = the
  // (imgs, 2 cv_H, 2 cv_W, dc_cout) fp32 output.  0 = not this form
  int dc_cout;
};

struct WideFrontParams {
  const uint8_t* xs;         // (2, rows, C) u8: SN_proj of the gathered slices
  int64_t rows;              // B_ * N1
  int N1, C, nH;
  const int8_t* wq; const int8_t* wk;      // digit planes, row pitch C; plane strides (bytes)
  int64_t wq_plane, wk_plane;
  const float *q_cs, *k_cs;                // (C) power-of-two channel scales
  const float *q_al, *q_be, *k_al, *k_be;
  const float* pe; int64_t pe_ld;          // k's additive term pe[(t * N1 + n) * pe_ld + c] or null
  SdfNeuronCfg sn_q, sn_k, sn2_q;
  float it_q, it_k, it_2;
  uint8_t* e;                // (2, rows, C)
  uint8_t* qs; uint8_t* ks;  // KEEP: q / k spikes with row strides ldq / ldk
  int64_t ldq, ldk;
  int nrg, ntiles;           // row groups (NW tiles each), token tiles
  int ntiles_per;            // ms_res.hip: token tiles per row group
};

// weight-resident row-loop forms (ms_res.hip): whole-K digit planes of a column group stay in LDS, the waves of a workgroup walk row
// units independently (two waves per SIMD, no barrier after the weights are in)
// stride-2 3x3 transposed convolution, weight-resident halo-tile kernel (spike_deconv_wres.hip): Cin = 208
bool spike_deconv_wres_supports(int imgs, int H, int W, int Cin, int Cout);
int launch_spike_deconv_wres(const uint8_t* A, const int8_t* Wd, const float* cscale, const float* alpha, const float* beta, float* out,
                             int imgs, int H, int W, int Cout, hipStream_t s);
bool res_pm_takes(const WidePmParams& P, int T, int epi);
int launch_res_pm(WidePmParams& P, int T, int epi, hipStream_t s);
bool res_front_takes(const WideFrontParams& P);
int launch_res_front(WideFrontParams& P, bool keep, int nk, hipStream_t s);

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
constexpr uint32_t INV = 0x80000000u;               // buffer offset of "no such row": loads return zeros, stores are dropped

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)INV, 0x00020000);
}

// 4 bits -> 4 bytes {0, 1}: bit i lands in byte i (i + 7 i = 8 i; the cross terms i + 7 k, k != i, miss every byte's bit 0)
__device__ __forceinline__ uint32_t spread4(uint32_t nib) { return __umul24(nib & 0xFu, 0x204081u) & 0x01010101u; }

// 4 x 4 byte transpose inside every quad of lanes: in: lane i of the quad holds bytes (rows 0..3) of column i; out: lane i holds
// bytes (columns 0..3) of row i.  sel1 / sel2 are the lane's v_perm selectors (quad_sel).
__device__ __forceinline__ void quad_sel(int lane, uint32_t& sel1, uint32_t& sel2) {
  sel1 = (lane & 1) ? 0x03070105u : 0x06020400u;
  sel2 = (lane & 2) ? 0x03020706u : 0x05040100u;
}
__device__ __forceinline__ uint32_t quad_tr_bytes(uint32_t w, uint32_t sel1, uint32_t sel2) {
  const uint32_t t1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0xB1, 0xF, 0xF, false);     // lane ^ 1
  const uint32_t a = __builtin_amdgcn_perm(t1, w, sel1);
  const uint32_t t2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, 0x4E, 0xF, 0xF, false);     // lane ^ 2
  return __builtin_amdgcn_perm(t2, a, sel2);
}

// sum over the 16 lanes of a DPP row (all lanes end with the total)
__device__ __forceinline__ uint32_t row_sum16(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);                    // quad_perm [1,0,3,2]
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);                    // quad_perm [2,3,0,1]
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false);                   // row_half_mirror
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false);                   // row_mirror
  return v;
}

typedef __attribute__((ext_vector_type(4))) int i32x4;
constexpr int KCH = 128;                             // K chunk = two 64-deep MFMA steps
constexpr int RBW = 5;                               // row blocks of a wave in the position-major kernel: 80 rows = 20 slots per lane

__host__ __device__ constexpr int s_pitch(int bytes) { return ((bytes + 16) / 4) % 8 == 4 ? bytes + 16 : bytes + 32; }
__host__ __device__ constexpr int w_pieces(int CB) { return 3 * 16 * CB * 8; }                   // 16-byte pieces of a weight chunk
__host__ __device__ constexpr int w_steps(int CB) { return (w_pieces(CB) + 255) / 256; }         // per thread (256 threads)
__host__ __device__ constexpr int w_buf(int CB) { return w_pieces(CB) * 16 + 16; }               // ring buffer bytes (+ a dump slot)

// acc += A x B on the int8 matrix pipe, accumulating IN PLACE in the accumulator registers.  Written as inline assembly with the
// accumulator tied to an AGPR quad: left to the register allocator (the builtin), the 120 - 240 accumulators of a wave were split
// between VGPRs and AGPRs and shuttled through a scratch quad around every MFMA (4 copies + wait states each: the matrix pipe ran
// at half rate, /tmp ISA of round 4).  The same accumulator is not touched again for >= 30 MFMAs (no software wait states needed).
__device__ __forceinline__ void mfma_i8(i32x4& acc, const i32x4& a, const i32x4& b) {
  asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

// The same with the accumulator in VECTOR registers (ms_res.hip: 120 accumulators + the rest fit the 256 unified registers of two
// waves per SIMD when no AGPR is used at all - and the epilogue reads them without one v_accvgpr_read per register)
__device__ __forceinline__ void mfma_i8_v(i32x4& acc, const i32x4& a, const i32x4& b) {
  asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_i8_v_zero(i32x4& acc, const i32x4& a, const i32x4& b) {
  asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "v"(b));
}
template <int A, int B, int C_>
__device__ __forceinline__ void mfma_drain_v(i32x4 (&acc)[A][B][C_]) {
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
#pragma unroll
  for (int a = 0; a < A; ++a)
#pragma unroll
    for (int b = 0; b < B; ++b)
#pragma unroll
      for (int c = 0; c < C_; ++c) asm volatile("" : "+v"(acc[a][b][c]));
}

// acc = A x B (no accumulator input: the first k-step of a tile - saves zeroing the accumulators, one v_accvgpr_write per register)
__device__ __forceinline__ void mfma_i8_zero(i32x4& acc, const i32x4& a, const i32x4& b) {
  asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, 0" : "=a"(acc) : "v"(a), "v"(b));
}

// The MFMAs above are inline assembly: hipcc's hazard recogniser does not see that they write the accumulators, so nothing pads the
// wait states an MFMA result needs before any OTHER kind of instruction may read or move it (cdna_hip_programming.md section 5.7 item 2:
// "an MFMA's D -> any reader ... including compiler code after the asm", 12 states for an 8-pass MFMA).  mfma_drain() is that pad: the
// nops, then every accumulator quad named as "+a" in an (empty) volatile statement behind them - volatile statements keep their order, so
// no compiler-made read or copy of a quad can land between its last MFMA and the nops.  Call it behind the last MFMA of a main loop (and
// of every loop ROUND where the compiler may shuffle accumulators on the loop's exit edge).  Found in round 4: a lone chunk behind a
// two-chunk loop round gave wrong sums in some instantiations - accumulator copies on the exit edge, a few cycles behind the last MFMA.
template <int A, int B, int C_>
__device__ __forceinline__ void mfma_drain(i32x4 (&acc)[A][B][C_]) {
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
#pragma unroll
  for (int a = 0; a < A; ++a)
#pragma unroll
    for (int b = 0; b < B; ++b)
#pragma unroll
      for (int c = 0; c < C_; ++c) asm volatile("" : "+a"(acc[a][b][c]));
}

// One accumulator quad AGPR -> VGPR, HERE (volatile, ordered against memory operations): left to the register allocator, the copies of
// ALL of a wave's accumulators were placed right behind the main loop - 120 registers live across the reduction's barriers, which
// pushed the small-M kernel's two-workgroups-per-CU builds (128 + 128 registers) into scratch.
__device__ __forceinline__ i32x4 acc_read(const i32x4& a) {
  const int x0 = a[0], x1 = a[1], x2 = a[2], x3 = a[3];
  int r0, r1, r2, r3;
  asm volatile("v_accvgpr_read_b32 %0, %4\n\tv_accvgpr_read_b32 %1, %5\n\tv_accvgpr_read_b32 %2, %6\n\tv_accvgpr_read_b32 %3, %7"
               : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "a"(x0), "a"(x1), "a"(x2), "a"(x3) : "memory");
  return i32x4{r0, r1, r2, r3};
}

// the three exact integer sums of an output -> one fp32 number (the two low digits meet as integers: < 2^27 at K = 3072)
__device__ __forceinline__ float digits_f32(int a0, int a1, int a2) { return __builtin_fmaf((float)a2, 65536.f, (float)(a1 * 256 + a0)); }

}  // namespace
}  // namespace sdfmm
