/* sdformerflow_hip.h - C ABI of libsdformerflow_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the SDformerFlow forward hot path.  The reference is
 * pure Python; the only native code on its path is SpikingJelly's CuPy multi-step neuron kernel,
 * switched on with `functional.set_backend(model, "cupy", neurontype)` (reference
 * eval_DSEC_flow_SNN.py:118-119).  Every entry point below names the reference call it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller
 *   - the library never allocates or frees device memory and never synchronises; all work is
 *     enqueued on `stream` (a hipStream_t passed as void*; NULL = the legacy default stream)
 *   - return value: 0 = launched, <0 = argument error detected before any launch
 *     (SDF_E_*), >0 = the hipError_t of the failed launch.  No C++ exception crosses the ABI.
 *   - spikes are {0,1}; `*_dtype` selects their storage: SDF_F32 (the reference's own format)
 *     or SDF_U8 (1 byte/spike, the operand format of the spike GEMM)
 */
#ifndef SDFORMERFLOW_HIP_H
#define SDFORMERFLOW_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDF_VERSION 107   /* round 6 (107): + sdf_switches_reload, sdf_launch_log / sdf_launch_log_read, SdfQkAttnDesc.rep_windows, SdfSpikeGemmDesc.zg_rep; the diagnostic SDF_* switches are read once, not per call.  106 / round 5: + sdf_ann_attn_block_fwd, sdf_ann_mlp_block_fwd, sdf_spike_deconv3x3s2_fwd, row-major digit planes in sdf_spike_gemm_fwd; 106: + sdf_linear_dw_fwd, sdf_ringed_rows_fwd, sdf_linear_train_fwd */

enum { SDF_F32 = 0, SDF_U8 = 1 };
enum { SDF_LIF = 0, SDF_PSN = 1, SDF_IF = 2 };
enum {
  SDF_E_NULL = -1,    /* required pointer is NULL            */
  SDF_E_SHAPE = -2,   /* size / divisibility requirement     */
  SDF_E_DTYPE = -3,   /* unknown dtype / kind selector       */
  SDF_E_ALIGN = -4    /* pointer or stride alignment         */
};

int sdf_version(void);

/* The library's diagnostic switches (the SDF_* environment variables listed in INTEGRATION.md's appendix: A/B and tuning overrides,
 * none needed in production) are read from the environment ONCE, at the first call that asks for one, into a read-only table - no
 * entry point calls getenv() on its per-call path.  A test harness that changes the environment between calls re-reads the table
 * with this call (not concurrently with other calls into the library).  No reference counterpart. */
void sdf_switches_reload(void);

/* Diagnostic per-launch log (round 6; off by default, one relaxed atomic load per launch while off).  sdf_launch_log(1) clears the
 * log and switches it on: every kernel launch of every entry point is then bracketed by two HIP events on its own stream and noted
 * with its kernel, grid and block; sdf_launch_log(0) switches it off.  sdf_launch_log_read synchronises with the recorded launches
 * and copies up to `max_records` records in launch order; returns the number of launches logged (may exceed max_records).
 * `workgroups * us / 256` = the CHIP time of a launch (compute units x time it can hold), the measure bench.py's `roofline.by_time`
 * reports beside the duration.  Not for use during stream capture (events).  No reference counterpart. */
typedef struct SdfLaunchRecord {
  char kernel[192];         /* demangled kernel name with its template arguments */
  uint32_t workgroups, threads, lds_bytes;   /* grid size in workgroups, threads per workgroup, dynamic LDS bytes */
  float us;                 /* duration between the two events around the launch; -1: unavailable */
} SdfLaunchRecord;
void sdf_launch_log(int enable);
int sdf_launch_log_read(SdfLaunchRecord* out, int max_records);

/* ---------------------------------------------------------------------------------------------
 * Multi-step LIF over the leading (time) axis of a contiguous fp32 (T, N) tensor.
 * Replaces: neuron.LIFNode multi-step forward (reference Spiking_modules.py:40-47,98-99; the
 * CuPy backend of eval_DSEC_flow_SNN.py:118-119).  Arithmetic (each op separately rounded):
 *   h = v + (x_t - v)/tau ; s = (h - v_th >= 0) ; v = h - s*v_th (soft) | (1-s)*h + s*v_reset
 *   tau > 1: spikingjelly LIFNode.  0 < tau < 1 (every forward entry point and descriptor that carries a neuron): the
 *   multiplicative charge h = v + (x_t - v)*tau of ParametricLIFNode, tau = sigmoid(w) (reference Spiking_modules.py:75-82).
 *   tau <= 0 or tau == 1: SDF_E_SHAPE.  The backward entry points take tau > 1 only.
 * v starts at 0 (soft) or v_reset (hard).  `v_last` (fp32, N) receives the final membrane or is NULL.
 */
int sdf_lif_fwd(const float* x, void* spike, float* v_last, int T, int64_t N, float tau, float v_th,
                int soft_reset, float v_reset, int spike_dtype, void* stream);

/* Parallel Spiking Neuron: H = b + W X over time, S = (H >= 0).
 * Replaces: PSN.forward (reference Spiking_submodules.py:207-211).  H[t] is the fp32 fmaf chain
 *   h = b[t]; for k = 0..T-1: h = fmaf(W[t,k], x[k], h)
 * (an order the oracle restates exactly; torch.addmm's own order is BLAS-defined).
 * W is (T,T) row-major fp32, b is (T) fp32, both device memory.  T <= 32.
 */
int sdf_psn_fwd(const float* x, const float* W, const float* b, void* spike, int T, int64_t N,
                int spike_dtype, void* stream);

/* ---- training path: backward of the neurons -------------------------------------------------
 * Replaces: autograd through spikingjelly's multi-step LIFNode / IFNode (torch backend; the reference builds them in
 * Spiking_modules.py:40-66 with `surrogate_function=ATan()`, `detach_reset` from the YAML) - BPTT over the T steps:
 *   gh_t = gv_t * dv_t/dh_t + (gs_t [+ reset path unless detach_reset]) * g'(h_t - v_th)
 *   gx_t = gh_t / tau ;  gv_{t-1} = gh_t - gh_t / tau           (IF: gx_t = gv_{t-1} = gh_t)
 *   g'(u) = alpha/2 / (1 + (pi/2 * alpha * u)^2)                (surrogate.ATan, the only one the configs use)
 * x, grad_spike, grad_x: (T, N) fp32 contiguous, N % 4 == 0, 16-byte aligned.  The membrane trajectory is recomputed
 * from x with the forward's exact arithmetic - the forward saves nothing.  T in {1,2,4,5,8,10,16,20}.
 * Bit-equal to the CPU autograd of the reference formulas for detach_reset = 1. */
#define SDF_SURROGATE_ATAN 0
int sdf_lif_bwd(const float* x, const float* grad_spike, float* grad_x, int T, int64_t N, int kind, float tau,
                float v_th, int soft_reset, float v_reset, int detach_reset, int surrogate, float alpha, void* stream);

/* Backward of PSN.forward (reference Spiking_submodules.py:207-211): gh = grad_spike * g'(H), grad_x = W^T gh,
 * grad_W = gh X^T, grad_b = sum_n gh.  grad_W / grad_b (T*T, T fp32) are optional (both or neither, T <= 10): they are
 * reduced deterministically through `workspace` (sdf_psn_bwd_workspace_bytes, caller-owned).  grad_h (T, N) is an
 * optional copy of gh (for T > 10 form grad_W = grad_h X^T with a library GEMM). */
int64_t sdf_psn_bwd_workspace_bytes(int T, int64_t N);
int sdf_psn_bwd(const float* x, const float* W, const float* b, const float* grad_spike, float* grad_x, float* grad_W,
                float* grad_b, float* grad_h, void* workspace, int64_t workspace_bytes, int T, int64_t N, int surrogate,
                float alpha, void* stream);

/* Batch-statistics BatchNorm over the last dim of a channel-last (R, C) fp32 buffer, forward and backward (training form of
 * the SpikingNormLayer "BN": spikingjelly layer.BatchNorm2d multi-step -> nn.BatchNorm2d on the view the reference makes with
 * permute(0,1,4,2,3); Spiking_modules.py:101-146, Spiking_swin_transformer3D.py:172, 178, 673, 677, 714, 972) - no permute
 * copies.  forward: y = (x - mean) * invstd * w + b with batch mean / biased variance (fp64 sums), save_mean / save_invstd (C)
 * for the backward, running stats updated in place with momentum and the unbiased variance (pass NULL, NULL to skip).
 * backward: grad_bias = sum gy, grad_weight = sum gy * xhat, grad_x = (gy - gb/R - xhat * gw/R) * invstd * w.
 * C % 4 == 0; workspace: sdf_bn_train_workspace_bytes(R, C) bytes (8-byte aligned), caller-owned; deterministic reductions. */
int64_t sdf_bn_train_workspace_bytes(int64_t R, int C);
int sdf_bn_train_fwd(const float* x, const float* weight, const float* bias, float* y, float* save_mean, float* save_invstd,
                     float* running_mean, float* running_var, int64_t R, int C, float eps, float momentum, void* workspace,
                     int64_t workspace_bytes, void* stream);
int sdf_bn_train_bwd(const float* x, const float* grad_y, const float* weight, const float* save_mean, const float* save_invstd,
                     float* grad_x, float* grad_weight, float* grad_bias, int64_t R, int C, void* workspace, int64_t workspace_bytes,
                     void* stream);

/* The same for an NCHW buffer (N, C, H, W), HW = H * W a multiple of 4, C <= 2048: the conv outputs of the patch embedding
 * and the U-Net tail (layer.BatchNorm2d on (T*B, C, H, W), Spiking_modules.py:118, :291-296, :339-347, :467-474, :811-819,
 * :906-933).  Same statistics / outputs / workspace size as the channel-last pair. */
int sdf_bn_train_nchw_fwd(const float* x, const float* weight, const float* bias, float* y, float* save_mean, float* save_invstd,
                          float* running_mean, float* running_var, int64_t N, int C, int HW, float eps, float momentum,
                          void* workspace, int64_t workspace_bytes, void* stream);
int sdf_bn_train_nchw_bwd(const float* x, const float* grad_y, const float* weight, const float* save_mean,
                          const float* save_invstd, float* grad_x, float* grad_weight, float* grad_bias, int64_t N, int C, int HW,
                          void* workspace, int64_t workspace_bytes, void* stream);

/* Token gate of Spiking_QK_WindowAttention3D for the training path (reference Spiking_swin_transformer3D.py:687-694:
 * `q.sum(-1)` over each head's 32 channels -> sn2_q over the T' steps -> `k.mul(...)`), fp32 spike tensors (T', rows, C):
 *   forward   e = k * A,  A = SN2_q(head sums of q)
 *   backward  grad_k = grad_e * A ;  grad_q = BPTT of SN2_q applied to sum_d(grad_e * k), broadcast over the head's channels
 * LIF / IF / PSN gates (ATan surrogate; detach_reset as in sdf_lif_bwd); T' in {1, 2, 4}; C % 32 == 0; 16-byte aligned pointers.
 * PSN: psn_w (T',T') / psn_b (T') in, grad_psn_w / grad_psn_b out, reduced deterministically through `workspace`
 * (sdf_qk_gate_bwd_workspace_bytes); all five NULL otherwise.
 * One launch each instead of the ~8 elementwise / reduction launches autograd runs for the composed expression. */
int sdf_qk_gate_f32_fwd(const float* q, const float* k, float* e, int Tq, int64_t rows, int C, int kind, float tau, float v_th,
                        int soft_reset, float v_reset, const float* psn_w, const float* psn_b, void* stream);
int64_t sdf_qk_gate_bwd_workspace_bytes(int Tq, int64_t rows, int C);
int sdf_qk_gate_bwd(const float* q, const float* k, const float* grad_e, float* grad_q, float* grad_k, int Tq, int64_t rows, int C,
                    int kind, float tau, float v_th, int soft_reset, float v_reset, int detach_reset, int surrogate, float alpha,
                    const float* psn_w, const float* psn_b, float* grad_psn_w, float* grad_psn_b, void* workspace,
                    int64_t workspace_bytes, void* stream);

/* General neuron launch: strided / gathered input, fused eval-BatchNorm and additive prologue.
 * Replaces the reference idiom  SN( BN( y.permute(..) ).permute(..) [+ positional_encoding] )
 * (Spiking_swin_transformer3D.py:670-680, 168-174, 970) and the pad / roll / window_partition_v2
 * gather in front of it (:789-804), without materialising any of the permutes.
 *
 * One time step holds nb*ni logical elements; element (b, r) of step t lives at
 *     x   + b*x_sb + t*x_st + r          (dense mode,  rowmap == NULL)
 *     x   + rowmap[t*rows + row]*rowlen + col ,  row = (b*ni + r)/rowlen, col = (b*ni+r)%rowlen,
 *           rows = nb*ni/rowlen, a negative map entry reads as 0.0      (gather mode)
 *     out + b*o_sb + t*o_st + r
 * Prologue (in this order, all optional):  y = fmaf(x, alpha[c], beta[c]) with c = (r/inner) % C;
 *     y = y + add[t*add_st + r % add_period].
 * ni, all strides, rowlen, inner*C-blocks and add_period must be multiples of 4 unless inner == 1
 * is replaced by C % 4 == 0 (the kernel moves 4 elements per lane).
 */
typedef struct SdfNeuronDesc {
  const float* x;
  void* out;
  float* v_last;            /* LIF/IF only, dense (nb*ni) or NULL */
  int32_t T;
  int32_t out_dtype;        /* SDF_F32 | SDF_U8 */
  int64_t nb, ni;
  int64_t x_sb, x_st;
  int64_t o_sb, o_st;
  const int32_t* rowmap;    /* NULL = dense */
  int32_t rowlen;
  int32_t kind;             /* SDF_LIF | SDF_PSN | SDF_IF */
  const float* alpha;       /* NULL = no affine */
  const float* beta;
  int32_t C;
  int32_t inner;
  const float* add;         /* NULL = none */
  int64_t add_st, add_period;
  float tau, v_th, v_reset;
  int32_t soft_reset;
  const float* psn_w;       /* (T,T) */
  const float* psn_b;       /* (T)   */
  /* round 6: an outermost dimension - nrep > 1 equally laid out problems (dense mode only), problem p at x + p*x_srep / out + p*o_srep
   * (v_last dense over all nrep*nb*ni): the batch elements of a pixel-strided channel slice in one descriptor.  0 / 1 = none. */
  int64_t nrep, x_srep, o_srep;
} SdfNeuronDesc;

int sdf_neuron_fwd(const SdfNeuronDesc* d, void* stream);
/* n independent neuron calls as ONE launch when they share T in {2, 4, 5, 10, 20} and n <= 6 (otherwise one launch each): the
 * U-Net decoders' skip inputs - four small tensors that each paid a launch of their own (reference Spiking_modules.py:467-474). */
int sdf_neuron_multi_fwd(const SdfNeuronDesc* descs, int n, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Spike GEMM:  out[M,N] = epilogue( A[M,K] (binary, u8) x W[N,K]^T ).
 * Replaces: sj_layer.Linear on spike tensors + the BatchNorm that follows it + the residual add
 * (reference Spiking_swin_transformer3D.py:170-178, 671-677, 712-714, 840, 845, 971-972).
 *
 * W is passed as `nsplit` bf16 planes (device, plane-major [nsplit][N][K]) with
 * W = plane0 + plane1 + plane2 (hi / mid / lo split made by sdf_split_weight_bf16); binary A is
 * exact in bf16, so nsplit = 3 reproduces the fp32 product to fp32 rounding on the bf16 MFMA path.
 * Epilogue, in order:  acc (+ bias[n]) ; fmaf(., alpha[n], beta[n]) if alpha ; (+ resid[row][n]) ;
 * stored fp32 at out[row*ldo + n] where row = out_rowmap ? out_rowmap[m] : m (negative = dropped)
 * and resid uses the same row.  `resid` may alias `out` (in-place residual update).
 *
 * A addressing: zg_nH == 0: A[m*lda + k].  zg_nH > 0: the reference's head-scramble
 *   Z[t,b,n,g*32+d] = E_flat[((((b*nH+g)*Tq+t)*N1+n)*32+d]   (Spiking_swin_transformer3D.py:709-710)
 * with m = (t*zg_B + b)*zg_N1 + n, k = g*32 + d, Tq = zg_T; requires K == nH*32.
 * zg_rep > 0 (round 6): the zg_B windows are zg_B / zg_rep INDEPENDENT problems of zg_rep windows each - replica r owns windows
 * r*zg_rep .. and its E_flat is the (Tq, zg_rep, N1, C) sub-tensor of the (Tq, zg_B, N1, C) buffer at those windows - so that one
 * launch sequence serves several batch-1 forwards with the reference's batch-1 semantics each (zg_rep % Tq == 0, zg_B % zg_rep == 0).
 * Requirements: K % 32 == 0, N % 32 == 0, lda % 16 == 0.
 */
typedef struct SdfSpikeGemmDesc {
  const uint8_t* A;
  const uint16_t* Wp;       /* bf16 bits, [nsplit][N][K] */
  float* out;
  int64_t M;
  int32_t N, K;
  int64_t lda, ldo;
  int32_t nsplit;           /* 1: one bf16 plane; 2: two fp16 planes of wscale*W (22 bits); 3: three bf16 planes (24 bits) */
  const float* bias;        /* NULL ok */
  const float* alpha;       /* NULL ok */
  const float* beta;
  const float* resid;       /* NULL ok */
  const int32_t* out_rowmap;/* NULL ok */
  int32_t zg_nH, zg_T, zg_B, zg_N1;
  /* Fused neuron epilogue (sn_T > 0):  out_spike = SN_T( fmaf(acc, alpha, beta) [+ add] ), 1-byte spikes, the fp32
   * pre-activation never leaves the chip.  Replaces Linear -> BN -> [+ positional_encoding] -> Spiking_neuron
   * (reference Spiking_swin_transformer3D.py:170-174, 671-680).  The M = pos_count*sn_T rows are (position, t)
   * pairs: row(P, t) = (P / pos_inner)*pos_ostride + P % pos_inner + t*t_stride, used for A and out_spike alike
   * (MLP on a (B,D,HW,.) tensor: pos_inner = HW, pos_ostride = D*HW, t_stride = HW; attention (T',rows,.):
   * pos_inner = rows, t_stride = rows).  add is (sn_T, add_prows, N) fp32 indexed [t][P % add_prows][n] or NULL.
   * sn_T in {2,4,5,10,20}; bias / resid / out_rowmap / zg_* must be unset; N % 16 == 0. */
  int32_t sn_T, sn_kind;    /* sn_T == 0: fp32 epilogue */
  float tau, v_th, v_reset;
  int32_t soft_reset;
  const float* psn_w;
  const float* psn_b;
  int64_t pos_count, pos_inner, pos_ostride, t_stride;
  const float* add;
  int64_t add_prows;
  uint8_t* out_spike;       /* (rows, N) u8 */
  /* Optional scratch for split-K (small M, large K): caller-owned device memory; when it holds at least
   * 2*M*N floats the library may run the product as K chunks + a deterministic k-ordered reduction. */
  void* workspace;
  int64_t workspace_bytes;
  /* nsplit == 2 only: 1 / wscale of the planes made by sdf_split_weight_f16x2 (a power of two); the accumulator is
   * multiplied by it - exactly - before the epilogue.  0 or 1 otherwise. */
  float acc_scale;
  /* With out_rowmap: number of rows of out / resid (an upper bound of the scattered row indices + 1); 0 = unknown.
   * Lets the library pick kernels that address the output with 32-bit offsets. */
  int64_t out_rows;
  /* nsplit == SDF_PLANES_I8X3 only: Wp is then int8_t[3][N][K] digit planes and col_scale[n] the power-of-two scale of output channel
   * n, both made by sdf_split_weight_i8x3: w[n][k] = (d2*65536 + d1*256 + d0) * col_scale[n].  NULL otherwise.  Taken by
   * sdf_spike_conv2d_fwd (3x3 / stride 1 / Cin 96 launches of the weight-resident convolution; the small-M convolution) and, since
   * round 5, by sdf_spike_gemm_fwd itself: a plain product with the fp32 epilogue (bias, alpha / beta, resid == out or NULL) on the
   * weight-resident row-loop kernel (csrc/ms_res.hip) for M % 10 == 0, K % 16 == 0, K <= 1024, lda == K - the stacked-tap products
   * of the middle decoder levels (reference Spiking_modules.py:461-474 as one product + sdf_deconv_col2im_fwd); SDF_E_SHAPE otherwise. */
  const float* col_scale;
  int32_t zg_rep;           /* head scramble: windows per independent replica (0 = zg_B: one problem) */
} SdfSpikeGemmDesc;

#define SDF_PLANES_I8X3 4
/* the same digits in MFMA fragment order (sdf_tile_weight_i8x3 below): what the small-M convolution kernel streams at full rate */
#define SDF_PLANES_I8X3_TILED 5

/* Weight-format helper for the 3x3 `layer.Conv2d` weights of MS_ResBlock / SEWResBlock (reference Spiking_modules.py:845-846,
 * 898-899; the reference has no counterpart - it multiplies fp32 weights with fp32 spike tensors in MIOpen / cuDNN).
 * fp32 weights (N, K) -> three int8 digit planes [3][N][K] + per-row power-of-two scales: q = rint(w / s) with
 * s = 2^(e - 23) with 2^e the power of two above max_k |w[n][k]| (23 bits + sign; 22 where the row maximum exceeds 0.996 * 2^e and the
 * top digit would leave int8), balanced base-256 digits.
 * Spikes are int8 values already: the spike x weight dot product becomes three exact int32 MFMA sums. */
int sdf_split_weight_i8x3(const float* W, int8_t* planes, float* col_scale, int N, int K, void* stream);

/* Digit planes [3][N][K] (sdf_split_weight_i8x3) -> fragment order [N / 16][K / 64][3][64][16 B]: the 1 KB a wave reads for one
 * 16-column x 64-deep x one-plane MFMA operand is contiguous and in lane order (lane l: column 16 j + l % 16, k = 64 s + 16 (l / 16) ..).
 * Same size as the planes; passed to sdf_spike_conv2d_fwd with nsplit = SDF_PLANES_I8X3_TILED and the same col_scale.  Only the
 * small-M kernels read it: the convolution (3x3 / stride 1, Cin % 64 == 0, at most 32 000 rows in (B, T, H, W) order: the U-Net
 * bottleneck of reference Spiking_modules.py:906-933) and, through sdf_spike_gemm_fwd with the same nsplit, the plain product with the
 * fp32 epilogue (M % 10 == 0, M <= 5 120, lda == K, K % 64 == 0, no out_rowmap / zg / add / neuron: the first decoder's stacked-tap GEMM,
 * reference Spiking_modules.py:461-474); other shapes return SDF_E_SHAPE.  N % 16 == 0, K % 64 == 0. */
int sdf_tile_weight_i8x3(const int8_t* planes, int8_t* tiled, int N, int K, void* stream);

int sdf_spike_gemm_fwd(const SdfSpikeGemmDesc* d, void* stream);

/* The plain form of the spike GEMM the hot-path contract names: out[M,N] = fmaf(A W^T, bn_a[n], bn_b[n])
 * with A (M,K) u8 spikes and W as planes (see above); a thin wrapper over sdf_spike_gemm_fwd. */
int sdf_spike_gemm_bn_fwd(const uint8_t* A_spike, const uint16_t* W_planes, int nsplit, float acc_scale, const float* bn_a,
                          const float* bn_b, float* out, int64_t M, int K, int N, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Window index table (row a4 / a6).  Replaces: F.pad to window multiples + torch.roll(-shift) + window_partition_v2 +
 * the raw `.view(Wd, B_, Wh, Ww, C)` (reference Spiking_swin_transformer3D.py:789-804, :100-113) and, read backwards,
 * window_reverse + roll(+shift) + crop (:810-820; swin_transformer3D_v2.py:52-65).
 * map[(j*Wh*Ww) + tok] = flat (b, d, h, w) index of token `tok` of slice j, or -1 for padding; slice j = window*Wd + frame,
 * windows ordered (b, d-block, h-block, w-block); attention step t' of window b' is slice t'*B_ + b'.  map holds
 * B_*Wd*Wh*Ww int32 (device memory of the caller); *n_windows (host, optional) receives B_.  Built by a kernel on `stream`. */
int sdf_window_slice_map(int32_t* map, int B, int D, int H, int W, int Wd, int Wh, int Ww, int shift_d, int shift_h, int shift_w,
                         int64_t* n_windows, void* stream);

/* Rows of a channel-last fp32 buffer moved through that table - the training path's window partition / reverse with no
 * materialised pad, roll, permute or crop:  gather  out[i,:] = map[i] >= 0 ? x[map[i],:] : 0   (M = map length rows out);
 * scatter  out[map[i],:] = y[i,:] for map[i] >= 0 (rows of `out` that no entry names are left untouched - every valid source
 * row occurs exactly once in a slice map, so each is the other's backward).  C % 4 == 0, 16-byte aligned. */
int sdf_rows_gather_fwd(const float* x, const int32_t* map, float* out, int64_t M, int C, void* stream);
int sdf_rows_scatter_fwd(const float* y, const int32_t* map, float* out, int64_t M, int C, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Spiking QK window attention, whole (rows a5 + a6):  x += SSA(x) in place.
 * Replaces: Spiking_SwinTransformerBlock3D.SSA + Spiking_QK_WindowAttention3D.forward + the shortcut add
 * (reference Spiking_swin_transformer3D.py:781-821, :661-717, :840).  TWO launches on `stream` where the first half has its
 * one-launch kernel (csrc/qk_front.hip: T' = 2, N1 <= 96, C % 96 == 0, fp16 planes, the four neurons of one class - the first
 * three steps below on a (slice pair, head) per workgroup, xs and q | k never in memory), else four:
 *   xs = SN_proj(x gathered through slice_map)                                  (T', B_*N1, C) u8
 *   q | k = SN_q/k( BN( xs [Wq;Wk]^T ) [+ positional_encoding on the k half] )  fused into the GEMM epilogue
 *   E = k AND SN2_q( sum over each head's 32 channels of q )                    (token gate)
 *   x[slice_map] += BN( Z Wp^T + b ),  Z = E read through the raw (B_,nH,T',N1,hd) head reshape (:709-710)
 * Weights are planes from sdf_split_weight_*; BN as (alpha, beta) per channel.  Give EITHER the stacked projection
 * (qk_*: planes [nsplit][2C][C], alpha / beta (2C), add (T', N1, 2C) with zeros in the q half, or NULL) - valid when the
 * q and k neurons are parameter-free and equal - OR the separate q_* / k_* sets (PSN: each neuron owns a matrix).
 * workspace: sdf_qk_attn_workspace_bytes(B_, T', N1, C) bytes, 256-byte aligned, caller-owned; gemm_workspace is the
 * optional split-K scratch of sdf_spike_gemm_fwd.  C = nH * 32. */
typedef struct SdfNeuronCfg {
  int32_t kind;             /* SDF_LIF | SDF_PSN | SDF_IF */
  float tau, v_th, v_reset;
  int32_t soft_reset;
  const float* psn_w;       /* (T,T), PSN only */
  const float* psn_b;       /* (T)          */
} SdfNeuronCfg;

typedef struct SdfQkAttnDesc {
  float* x;                 /* (B, D, H, W, C) fp32 channel-last, updated in place */
  const int32_t* slice_map; /* sdf_window_slice_map */
  int64_t B_;               /* windows */
  int64_t x_rows;           /* B*D*H*W */
  int32_t Tq, N1, C, nH;    /* window depth, Wh*Ww, channels, heads */
  int32_t nsplit;
  const uint16_t* qk_planes; const float* qk_alpha; const float* qk_beta; const float* qk_add; float qk_acc_scale;
  const uint16_t* q_planes;  const float* q_alpha;  const float* q_beta;  float q_acc_scale;
  const uint16_t* k_planes;  const float* k_alpha;  const float* k_beta;  const float* k_add; float k_acc_scale;
  const uint16_t* p_planes;  const float* p_bias;   const float* p_alpha; const float* p_beta; float p_acc_scale;
  SdfNeuronCfg sn_proj, sn_q, sn_k, sn2_q;
  void* workspace;      int64_t workspace_bytes;
  void* gemm_workspace; int64_t gemm_workspace_bytes;
  int32_t flags;        /* SDF_QK_* bits */
  /* Digit-plane stages (T' = 2, LIF / IF / PSN neurons, digit planes given - see below; round 5: csrc/ms_res.hip - whole-K weights LDS-resident, a
   * row loop per workgroup - for C in 64..768 in steps of 32 wherever K <= 1024, by default from 128 channels on; csrc/ms_wide.hip for C >= 192 in
   * steps of 64 beside it): with x_src set the call is THREE launches -
   * slice neuron, one kernel for q | k + BN + neurons + token gate, and the projection as a "position-major" product whose waves own
   * all xD time steps of a few positions of the (xB, xD, xHW, C) buffer x (xB * xD * xHW == x_rows):
   *   x_src   : int32 per row of x, made by sdf_window_zsrc_map from slice_map (where that row's gated spikes start in E); NULL = the
   *             general kernels above
   *   emit_s1 : optional u8 [x_rows][C]: receives SN_emit( x after the update ) over the xD steps of every position - the first
   *             neuron of the MLP that follows (reference Spiking_swin_transformer3D.py:168; hand it to SdfMsMlpDesc.s1_in), so the
   *             updated x is not read again; emit_sn = that neuron (LIF / IF / PSN with its T = xD matrix).  Layout: the TILED hand-over form
   *             ([80-row unit][C / 16][row][16 B], (x_rows + 80) * C bytes) - opaque to the caller - unless SDF_QK_KEEP_SPIKES is
   *             set: then plain row-major [x_rows][C] (the parity tape). */
  const int32_t* x_src;
  int32_t xB, xD;
  int64_t xHW;
  uint8_t* emit_s1;
  SdfNeuronCfg emit_sn;
  /* The wide-stage kernels multiply on the int8 matrix pipe (spike bytes are int8 values): the same weights again as digit planes
   * int8_t[3][rows][C] + one power-of-two scale per output channel, made by sdf_split_weight_i8x3 (stacked [Wq; Wk] as 2C rows for
   * the qk_* form, else q_* and k_*; p_* = the output projection).  All NULL = the general kernels on the 16-bit planes above. */
  const int8_t* qk_digits; const float* qk_cscale;
  const int8_t* q_digits;  const float* q_cscale;
  const int8_t* k_digits;  const float* k_cscale;
  const int8_t* p_digits;  const float* p_cscale;
  /* Round 6 - several independent batch-1 forwards in one call ("replicas"): the B_ windows are B_ / rep_windows problems of
   * rep_windows windows each.  slice_map (and x_src) are the caller's concatenation of the per-replica tables - entry
   * (t' * B_ + r * rep_windows + b') * N1 + n = r * x_rows_1 + map_1[(t' * rep_windows + b') * N1 + n] - so steps 1 - 3 run unchanged on
   * the (T', B_, N1, C) workspace; only the head scramble of step 4 is per replica (SdfSpikeGemmDesc.zg_rep).  0 = one problem (the
   * reference's own batch semantics, which couples the samples of a batch through window_partition_v2's raw view).  rep_windows % T' == 0. */
  int32_t rep_windows;
} SdfQkAttnDesc;

enum {
  SDF_QK_KEEP_SPIKES = 1,    /* leave the q | k spikes in `workspace` behind E (what the four-launch form always does; the parity tape) */
  SDF_QK_FOUR_LAUNCHES = 2,  /* never take the one-launch first half (A/B reference; same as SDF_QK_FRONT=0 in the environment) */
  SDF_QK_NARROW = 4          /* never take the wide-stage kernels (A/B reference; same as SDF_WIDE=0 in the environment) */
};

/* Inverse of the slice map for the projection's head scramble (reference Spiking_swin_transformer3D.py:709-710): x_src[r] = byte offset
 * in the gated-spike tensor E (T', B_, N1, C) of Z[t, b, n, 0] for the slice-map entry (t, b, n) that names row r of x; rows no entry
 * names keep their old value (there are none for a map made by sdf_window_slice_map over the same x).  x_src: int32 [x_rows]. */
int sdf_window_zsrc_map(const int32_t* slice_map, int64_t B_, int Tq, int N1, int nH, int32_t* x_src, void* stream);

int64_t sdf_qk_attn_workspace_bytes(int64_t B_, int Tq, int N1, int C);
int sdf_qk_attn_fwd(const SdfQkAttnDesc* d, void* stream);
/* 1 when sdf_qk_attn_fwd will run this descriptor on the wide-stage kernels (and therefore honours emit_s1), else 0; host-only. */
int sdf_qk_attn_is_wide(const SdfQkAttnDesc* d);

/* ---------------------------------------------------------------------------------------------
 * MS MLP, whole (row a7):  x += BN2( SN2( BN1( SN1(x) W1^T ) ) W2^T )  in place on a (B, D, HW, C) fp32 buffer, neurons over
 * the true time axis D.  Replaces: MS_Spiking_Mlp.forward + the block's second shortcut add (reference
 * Spiking_swin_transformer3D.py:164-181, :845).  ONE launch (csrc/ms_mlp_fused.hip: SN1, fc1, BN1, SN2, fc2, BN2 and the
 * shortcut on a tile of positions x all D steps; neither spike tensor leaves the compute unit) for C in {96, 192},
 * D in {5,10,20}, Ch % 96 == 0; otherwise three launches with the hidden (.., 4C) tensor as 1-byte spikes in `workspace`
 * (D in {2,4,5,10,20}, the fused-neuron GEMM epilogue).  workspace: sdf_ms_mlp_workspace_bytes(B*D*HW, C, Ch), 256-byte aligned. */
typedef struct SdfMsMlpDesc {
  float* x;
  int32_t B, D;
  int64_t HW;
  int32_t C, Ch, nsplit;
  const uint16_t* fc1_planes; const float* fc1_alpha; const float* fc1_beta; float fc1_acc_scale;
  const uint16_t* fc2_planes; const float* fc2_alpha; const float* fc2_beta; float fc2_acc_scale;
  SdfNeuronCfg sn1, sn2;
  void* workspace;      int64_t workspace_bytes;
  void* gemm_workspace; int64_t gemm_workspace_bytes;
  int32_t flags;        /* SDF_MLP_* bits */
  /* Digit-plane stages (C >= 64 in steps of 32 up to 192 and Ch <= 768 - csrc/ms_res.hip - or C >= 192 in steps of 64; LIF / IF / PSN, digit planes
   * given - see below, D in {10, 20}; by default taken from 128 channels on): fc1 + BN1 + SN2 and fc2 + BN2 + shortcut as two
   * position-major launches.  s1_in != NULL: the SN1 spikes are already at the head of `workspace` (written there by
   * SdfQkAttnDesc.emit_s1 == workspace; tiled, or row-major when both calls carry their KEEP_SPIKES flag): no neuron launch, x is
   * only read by the last launch's shortcut.  Without SDF_MLP_KEEP_SPIKES the hidden spikes travel in the tiled form too. */
  const uint8_t* s1_in;
  /* the weights again as int8 digit planes int8_t[3][N][K] + a power-of-two scale per output channel (sdf_split_weight_i8x3): what
   * the wide-stage kernels multiply by; NULL = the general kernels on the 16-bit planes above. */
  const int8_t* fc1_digits; const float* fc1_cscale;
  const int8_t* fc2_digits; const float* fc2_cscale;
  /* wide-stage form only: emit_next != NULL (u8 [tokens][C], row-major) also receives SN_emit( x after the update ) over the D steps of
   * every position - the first neuron of whatever reads x next (the patch merging's `sn`, reference Spiking_swin_transformer3D.py:970,
   * or MS_ResBlock.sn1 of the U-Net bottleneck, Spiking_modules.py:922) - from the last launch's epilogue, so x is not read again. */
  uint8_t* emit_next;
  SdfNeuronCfg emit_sn;
  /* wide-stage form, optional: fc2's digits again in MFMA fragment order (sdf_tile_weight_i8x3 of fc2_digits; fc2_cscale serves both).
   * With it fc2 runs on the small-M kernel (csrc/ms_smallm.hip: K split over the waves of a workgroup, two workgroups per compute
   * unit) instead of the wide main loop where its (80-row unit x 32-column) tiles fit the chip in one round (<= 512 tiles: swin stage 3 at
   * batch 1, 30.9 -> 19.9 us) - bit-equal results; NULL = the wide main loop. */
  const int8_t* fc2_tiled;
} SdfMsMlpDesc;

enum {
  SDF_MLP_KEEP_SPIKES = 1,   /* leave SN1's and SN2's spikes in `workspace` (u8 [tokens][C], then [tokens][Ch] at the next 256-byte
                                boundary): what the three-launch form always does; the one-launch form only on request */
  SDF_MLP_THREE_LAUNCHES = 2,/* never take the one-launch kernel (A/B reference; same as SDF_MLP_FUSED=0 in the environment) */
  SDF_MLP_NARROW = 4         /* never take the wide-stage kernels (A/B reference; same as SDF_WIDE=0 in the environment) */
};

int64_t sdf_ms_mlp_workspace_bytes(int64_t tokens, int C, int Ch);
int sdf_ms_mlp_fwd(const SdfMsMlpDesc* d, void* stream);
/* 1 when sdf_ms_mlp_fwd will run this descriptor on the wide-stage kernels (and therefore accepts s1_in), else 0; host-only. */
int sdf_ms_mlp_is_wide(const SdfMsMlpDesc* d);

/* ---------------------------------------------------------------------------------------------
 * MS patch merging on spikes (row a8):  out = BN( cat_2x2( S ) W^T ),  S = SN(x) as u8 (B, D, H, W, C), W (N, 4C) given as int8 digit
 * planes, out (B, D, ceil(H/2), ceil(W/2), N) fp32 (odd sizes read zero spikes, the reference's F.pad in front of the neuron).  Replaces MS_SpikingPatchMerging.forward behind its neuron (reference
 * Spiking_swin_transformer3D.py:965-972: the four strided slices, the concatenation along channels in quadrant order
 * (dh, dw) = (q % 2, q / 2), sj_layer.Linear, the BatchNorm) - the concatenation is index arithmetic in the operand loads of the
 * digit main loops (csrc/ms_res.hip for C <= 256 in steps of 32: any row count; csrc/ms_wide.hip for C % 64 == 0: at most 32 000 output
 * rows), nothing is materialised.  N % 32 == 0, D in {10, 20}; SDF_E_SHAPE otherwise (the caller keeps its gather map + sdf_spike_gemm_fwd). */
typedef struct SdfMsMergeDesc {
  const uint8_t* spikes;
  const int8_t* digits; const float* cscale;   /* sdf_split_weight_i8x3 of the (N, 4C) reduction weight */
  const float* alpha; const float* beta;       /* BatchNorm as (alpha, beta) per output channel, or NULL */
  float* out;
  int32_t B, D, H, W, C, N;
} SdfMsMergeDesc;

int sdf_ms_patch_merge_fwd(const SdfMsMergeDesc* d, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Stride-2 3x3 transposed convolution on spikes as ONE product (round 5; csrc/ms_res.hip):
 *   out[img, oy, ox, co] = alpha[co] * sum_{ky, kx, c : oy = 2 iy - 1 + ky, ox = 2 ix - 1 + kx} S[img, iy, ix, c] W[c, co, ky, kx] + beta[co]
 * - `layer.ConvTranspose2d(3, stride 2, padding 1, output_padding 1)` + the decoder's BatchNorm behind its neuron (reference
 * MS_SpikingTransposeDecoderLayer, Spiking_modules.py:461-474).  A row of the product is an input pixel (img, a, b), its K = 4 Cin the
 * 2 x 2 input neighbourhood (a..a+1, b..b+1; zero beyond the image) in patch merging's quadrant order (dh, dw) = (q % 2, q / 2), its
 * N = 4 Cout columns the four output pixels (2a + py, 2b + px) of the row's 2 x 2 output block, column (2 py + px) Cout + co: the weight
 * matrix holds W[., ., ky, kx] in the blocks a tap reaches and zeros elsewhere (7 of 16 blocks), as int8 digit planes
 * (sdf_split_weight_i8x3 of the (4 Cout, 4 Cin) matrix).  The four parity-class convolutions it replaces wrote every other pixel of a
 * row each (half-written 128-byte lines: twice the output's bytes reached HBM); here a row's two output pixels of a line are stored together.
 * S (imgs, H, W, Cin) u8 NHWC, imgs % T == 0 with T in {10, 20}; Cin % 16 == 0, 4 Cin <= 1024; Cout % 8 == 0; out (imgs, 2H, 2W, Cout)
 * fp32; alpha / beta (4 Cout): the BatchNorm pair of channel co at every column (2 py + px) Cout + co, or NULL.  SDF_E_SHAPE otherwise
 * (the caller keeps sdf_spike_conv2d_multi_fwd). */
typedef struct SdfSpikeDeconvDesc {
  const uint8_t* spikes;
  const int8_t* digits; const float* cscale;   /* [3][4 Cout][4 Cin] + (4 Cout) */
  const float* alpha; const float* beta;       /* (4 Cout) or NULL */
  float* out;
  int32_t imgs, T, H, W, Cin, Cout;
} SdfSpikeDeconvDesc;

int sdf_spike_deconv3x3s2_fwd(const SdfSpikeDeconvDesc* d, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Spike convolution (implicit GEMM):  out = epilogue( im2col(X) x W^T ) with X an NHWC u8 spike image batch.
 * Replaces: layer.Conv2d / nn.Conv2d / ConvTranspose2d on spikes + the SpikingNormLayer, shortcut add and
 * Spiking_neuron around it in the patch embedding and the U-Net tail (reference Spiking_modules.py:339-347,
 * 467-474, 811-818, 906-926).  `g` is used as in sdf_spike_gemm_fwd with:
 *   g.A  = X (imgs, H, W, Cin) u8, Cin % 16 == 0, Cin >= 48;   g.M = imgs*OH*OW;   g.N = Cout (multiple of 96)
 *   g.K  = KH*KW*Cin, weights packed [nsplit][Cout][(ky, kx, cin)] bf16 planes
 *   row m = (img, oy, ox) reads input pixel (oy*sy + dy[ky], ox*sx + dx[kx]); zero outside the image
 *     (a 3x3 / pad 1 convolution: dy = dx = {-1, 0, 1}; the parity classes of a stride-2 transposed convolution
 *     are 1- or 2-tap grids with their own offsets and an out_rowmap that places the class's outputs)
 *   epilogue: the fp32 one (alpha/beta/resid/out_rowmap, out is (rows, Cout) fp32 NHWC) or, with sn_T == 10 (int8 digit
 *     planes on the weight-resident kernel: also 5 and 20, its time loop is rolled),
 *     the fused neuron (out_spike u8 NHWC); pos_* describe how the imgs dimension factors into (time, position).
 *     With sn_T > 0 and g.out != NULL the pre-activation  fmaf(acc, alpha, beta) (+ g.resid)  is ALSO stored to g.out
 *     (fp32, row stride g.ldo) and the neuron runs on that sum: MS_ResBlock's conv2 -> BN -> + identity and the next
 *     block's sn1 in one launch (reference Spiking_modules.py:922-933).
 */
typedef struct SdfSpikeConvDesc {
  SdfSpikeGemmDesc g;
  int32_t H, W, Cin, OH, OW;
  int32_t KH, KW, sy, sx;
  int32_t dy[3], dx[3];
} SdfSpikeConvDesc;

int sdf_spike_conv2d_fwd(const SdfSpikeConvDesc* c, void* stream);
/* n (1 .. 4 fused) convolutions with the fp32 epilogue on the SAME images and output buffer that differ only in taps (KH, KW, dy, dx),
 * weights and g.out_rowmap - the four output-parity classes of ConvTranspose2d(k 3, s 2, p 1, op 1) in MS_SpikingTransposeDecoderLayer
 * (reference Spiking_modules.py:461-474), whose row maps write disjoint rows: ONE launch where the library's streaming kernel serves
 * every one of them (each class alone fills about half of the chip), otherwise exactly n calls of sdf_spike_conv2d_fwd in order. */
int sdf_spike_conv2d_multi_fwd(const SdfSpikeConvDesc* cs, int n, void* stream);

/* W (fp32, n elements) -> nsplit bf16 planes (round-to-nearest-even residual split). */
int sdf_split_weight_bf16(const float* W, uint16_t* planes, int64_t n, int nsplit, void* stream);

/* W (fp32, n elements) -> two fp16 planes [2][n] with  scale * W = hi + lo  (|error| <= 2^-22 |W|).  `scale` is a
 * power of two chosen by the caller so that scale * max|W| < 2^15 (fp16 range; small weights stay far above the
 * fp16 subnormal spacing); pass acc_scale = 1 / scale with nsplit = 2.  Spikes are exact in fp16, the MFMA
 * accumulates in fp32 and the rescaling is exact, so the result differs from the 3-plane one only by the 2^-22
 * weight truncation - below the rounding noise of the fp32 accumulation itself - at 2/3 of its MFMA work. */
int sdf_split_weight_f16x2(const float* W, uint16_t* planes, int64_t n, float scale, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Token gate of Spiking_QK_WindowAttention3D (reference Spiking_swin_transformer3D.py:687-694):
 *   a[t,b,n,g] = sum_{d<32} q[t,b,n,g*32+d] ; A = neuron_T'(a) ; E = k * A (broadcast over d).
 * q, k, e: u8 spikes laid out (Tq, rows, C) with rows = B_*N1, C % 32 == 0; the neuron runs over Tq
 * with the same kind/params semantics as sdf_neuron_fwd (LIF | PSN).
 */
int sdf_qk_gate_fwd(const uint8_t* q, const uint8_t* k, uint8_t* e, int Tq, int64_t rows, int C,
                    int kind, float tau, float v_th, float v_reset, int soft_reset,
                    const float* psn_w, const float* psn_b, void* stream);
/* Same, with q and k rows ldq / ldk bytes apart (multiples of 16, >= C): q and k may be the two halves of the
 * (Tq*rows, 2C) output of ONE fused-neuron spike GEMM over the stacked weights [Wq; Wk].  e stays (Tq, rows, C). */
int sdf_qk_gate_strided_fwd(const uint8_t* q, const uint8_t* k, uint8_t* e, int Tq, int64_t rows, int C,
                            int64_t ldq, int64_t ldk, int kind, float tau, float v_th, float v_reset,
                            int soft_reset, const float* psn_w, const float* psn_b, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Head of the patch embedding: conv3x3 / pad 1 (no bias) on the real-valued event voxel -> eval BatchNorm -> neuron over T,
 * one kernel, spikes out.  Replaces MS_PED_Spiking_PatchEmbed_Conv_sfn.head (reference Spiking_modules.py:1782).
 *   x   (B*T, H, W, Cin) fp32, NHWC, image index = b*T + t          w (Cout, Cin, 3, 3) fp32 (the module's layout)
 *   out (B*T, H, W, Cout) u8 spikes                                  alpha / beta (Cout) or NULL
 * Accumulation: fp32 products and sums in (ky, kx, cin) order - on the exact fp32 matrix pipe (v_mfma_f32_32x32x2_f32) when
 * W % 32 == 0 (LIF / IF; the PSN for T <= 10 and Cin = 2: its H is accumulated on the fly per 32-channel block), else one fmaf
 * chain per output on the vector pipe.  Built for (Cin, Cout) in {(2,32), (2,48), (2,64), (4,48)}, T in {5, 10, 20}, W % 16 == 0; anything else returns
 * SDF_E_SHAPE and the caller keeps its library convolution + sdf_neuron_fwd pair.
 */
typedef struct SdfHeadConvDesc {
  const float* x;
  const float* w;
  const float* alpha;
  const float* beta;
  uint8_t* out;
  int32_t B, T, H, W, Cin, Cout;
  int32_t sn_kind;
  float tau, v_th, v_reset;
  int32_t soft_reset;
  const float* psn_w;
  const float* psn_b;
  /* strided input (all zero = the packed NHWC layout above): element (b, t, y, x, ci) of the input is
   * x[b * x_sb + t * x_st + y * x_sy + x * x_sx + x_sc[ci]] - the event voxel (B, bins, 2, H, W) is then read IN PLACE
   * (reference patch embedding :1775-1784: time step t takes bin t of each polarity; channel ci = (bin group, polarity)),
   * no re-layout copies in front of the kernel */
  int64_t x_sb, x_st, x_sy, x_sx;
  int64_t x_sc[4];
} SdfHeadConvDesc;

int sdf_head_conv_sn_fwd(const SdfHeadConvDesc* d, void* stream);

/* ---------------------------------------------------------------------------------------------
 * 1x1 strided convolution of a real-valued fp32 NHWC image on the exact fp32 matrix pipe (csrc/pointwise_conv.hip):
 *   out[img, oy, ox, n] = sum_c x[img, oy*stride, ox*stride, c] * w[n, c] (+ bias[n])
 * Replaces SpikingPEDLayer.conv_res - the membrane shortcut of the stride-2 patch-embedding projection, the one layer of the
 * SNN forward that reads a membrane instead of spikes (reference Spiking_modules.py:772-826).
 *   x (imgs, H, W, Cin) fp32; w (N, Cin) fp32 (the module's (N, Cin, 1, 1)); out (imgs, OH, OW, N) fp32, OH = (H-1)/stride + 1.
 * Built for Cin = 96, N in {96, 192}; anything else returns SDF_E_SHAPE and the caller keeps its library convolution. */
typedef struct SdfPointwiseConvDesc {
  const float* x;
  const float* w;
  const float* bias;        /* (N) or NULL */
  float* out;
  int32_t imgs, H, W, Cin, N, stride, OH, OW;
} SdfPointwiseConvDesc;

int sdf_pointwise_conv_f32_fwd(const SdfPointwiseConvDesc* d, void* stream);

/* ---------------------------------------------------------------------------------------------
 * col2im + eval BatchNorm of a ConvTranspose2d(k=3, s=2, p=1, output_padding=1) on spikes whose nine per-tap products were
 * computed by ONE plain sdf_spike_gemm_fwd over the stacked tap weights (N = 9*Cout, column = (ky*3+kx)*Cout + co):
 *   out[img, oy, ox, co] = fmaf( sum_{taps reaching (oy,ox)} Y[(img, iy, ix)][(ky*3+kx)*Cout + co], alpha[co], beta[co] ),
 *   oy = 2*iy - 1 + ky, ox = 2*ix - 1 + kx.   Replaces MS_SpikingTransposeDecoderLayer's deconv + norm
 * (reference Spiking_modules.py:461-474) for the small decoder levels, where one GEMM fills the chip and four
 * parity-class convolutions (the other form this library offers, see sdf_spike_conv2d_fwd) do not.
 *   Y (imgs*H*W, 9*Cout) fp32;  out (imgs, 2H, 2W, Cout) fp32;  alpha / beta (Cout) or NULL;  Cout % 4 == 0.
 */
int sdf_deconv_col2im_fwd(const float* Y, const float* alpha, const float* beta, float* out, int imgs, int H, int W,
                          int Cout, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Flow read-out: out[b,c,Y,X] = sum_t pred[b,t,ys,xs,c] with (ys, xs) the nearest-neighbour source pixel of (Y, X) for
 * scale factors (scale_y, scale_x) - exactly F.interpolate(pred.sum(time), scale_factor=...)
 * (reference Spiking_STSwinNet.py:289-303).  pred is (B, D, h, w, ldp) fp32 of which the first C columns are read;
 * out is (B, C, H, W) fp32.
 */
int sdf_flow_out_fwd(const float* pred, float* out, int B, int D, int h, int w, int64_t ldp, int C, int H, int W,
                     float scale_y, float scale_x, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Flow-prediction head of one U-Net level, one launch (csrc/pred_head.hip):
 *   p[t] = conv1x1(SN_pred(z)[t]) + bias (2 outputs; MS_SpikingPredLayer, reference Spiking_modules.py:605-640),
 *   flow = nearest upsampling by whole factors (H / h, W / w) of sum_t p[t] (reference Spiking_STSwinNet.py:289-303), and - the
 *   next decoder level reads cat(p, z, skip) behind its own neuron (Spiking_STSwinNet.py:168-172, Spiking_modules.py:467-474) -
 *   the spike bytes of SN_next(z) / SN_next(p) written into their channel slices of that level's NHWC u8 spike image.
 * Replaces sdf_neuron_fwd + sdf_spike_gemm_fwd (2 columns padded to 32) + sdf_flow_out_fwd of this level and two
 * sdf_neuron_fwd launches (+ the zero fill of the image's padding channels) of the next one; z is read once.
 *   z (B, D, h*w, Cin) fp32 channel-last; Cin in {96, 192, 384}; D in {5, 10, 20} (PSN: D <= 10); wgt (2, Cin) fp32; bias (2) or NULL
 *   pred (B, D, h*w, 4) fp32 = (p0, p1, 0, 0) or NULL;  flow (B, 2, H, W) fp32 or NULL (H % h == 0, W % w == 0)
 *   next_spikes (B, D, h*w, next_ld) u8 or NULL: bytes [next_z_off, +Cin) = SN_next(z), [next_pred_off, +4) = SN_next(p) (channels
 *     2, 3 zero), [next_zero_off, +next_zero_len) zeroed (the image's padding channels); all offsets / lengths multiples of 4
 *   keep_spikes (B, D, h*w, Cin) u8 or NULL: SN_pred(z) (parity tape).  Anything else returns SDF_E_SHAPE: the caller keeps the
 *   three-launch form. */
typedef struct SdfPredHeadDesc {
  const float* z;
  int32_t B, D, h, w, Cin;
  SdfNeuronCfg sn_pred;
  const float* wgt;                 /* (2, Cin) fp32 */
  const float* bias;
  float* pred;
  float* flow;
  int32_t H, W;
  uint8_t* next_spikes;
  int32_t next_ld, next_z_off, next_pred_off, next_zero_off, next_zero_len;
  SdfNeuronCfg sn_next;
  uint8_t* keep_spikes;
} SdfPredHeadDesc;

int sdf_pred_head_fwd(const SdfPredHeadDesc* d, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Eval BatchNorm affine (+ residual):  out = fmaf(x, alpha[c], beta[c]) (+ resid), c = (i/inner) % C.
 * Replaces: SpikingNormLayer after a convolution and the membrane (MS) shortcut add
 * (reference Spiking_modules.py:922-926, 816-818).  n % 4 == 0; inner % 4 == 0 or (inner == 1, C % 4 == 0).
 * `resid` may be NULL; `out` may alias `x` or `resid`.
 */
int sdf_affine_resid_fwd(const float* x, const float* alpha, const float* beta, const float* resid, float* out,
                         int64_t n, int C, int64_t inner, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused 3-D window attention over (window, head): scores + relative-position bias (+ shift mask)
 * [+ softmax] times V, head_dim 32, up to 192 tokens per window.
 *   SDF_ATTN_ANN replaces WindowAttention3D.forward's core (reference
 *     models/STSwinNet/swin_transformer3D_v2.py:176-202): `q` points at the packed qkv projection output
 *     (B_, N, 3, nH, 32) fp32 (k, v are ignored but must be non-NULL); cosine attention
 *     normalize(q) normalize(k)^T * scale[g] + bias[g] (+ mask[b % nW]) -> softmax -> @ v;
 *     out (B_, N, nH*32) fp32.  scale[g] = exp(min(logit_scale[g], ln 100)); bias = 16*sigmoid(cpb(...)).
 *   SDF_ATTN_SEW replaces Spiking_BN_WindowAttention3D.forward's core (reference
 *     models/STSwinNet_SNN/Spiking_swin_transformer3D.py:320-363): q, k, v are u8 spikes in the reference's
 *     raw head view (B_, nH, N, 32) of the (T', B_, N1, C) buffers; (q*scale[g]) k^T + bias[g] (+ mask), NO
 *     softmax, @ v; out fp32 (T', B_, N1, C) through the (B_,nH,T',N1,hd) -> (T',B_,N1,C) scramble.
 * bias is (nH, N, N) fp32, mask (nW, N, N) fp32 or NULL.  All matrix products run on the exact fp32 MFMA.
 */
enum { SDF_ATTN_ANN = 0, SDF_ATTN_SEW = 1 };
typedef struct SdfWinAttnDesc {
  int32_t mode;
  const void* q;
  const void* k;
  const void* v;
  float* out;
  int32_t B_, nW, nH, N, hd;
  int32_t Tq, N1;           /* SEW only: N == Tq*N1 */
  const float* scale;       /* (nH) */
  const float* bias;        /* (nH,N,N) */
  const float* mask;        /* (nW,N,N) or NULL */
  /* ANN mode, optional in-kernel windowing (replaces F.pad + torch.roll + window_partition in front of the attention and
   * window_reverse + roll back + crop behind it, reference swin_transformer3D_v2.py:286-310): row_map[b_*N + n] = row of
   * token n of window b_ in the un-partitioned (rows, 3C) qkv buffer and (rows, C) output buffer (the table of
   * sdf_window_slice_map), or -1 for a padding token; a padding token's q | k | v is pad_qkv (3C floats: what the qkv
   * Linear makes of a zero row, i.e. its bias) and its output row is dropped.  NULL = q and out are window-major. */
  const int32_t* row_map;
  const float* pad_qkv;
} SdfWinAttnDesc;

int sdf_win_attn_fwd(const SdfWinAttnDesc* d, void* stream);

/* ---- the attention half of an ANN video-swin block as one launch (BASELINE config 3, first stage) -----------------
 * Replaces `SwinTransformerBlock3D.forward_part1` + the shortcut (reference models/STSwinNet/swin_transformer3D_v2.py:272-310, :331)
 * with `WindowAttention3D.forward` (:169-205) inside:
 *   out = x + proj(softmax(normalize(q) normalize(k)^T * logit_scale + bias (+ mask)) v),  q | k | v = LayerNorm(x) Wqkv^T + qkv_bias
 * over the windows `row_map` describes (the table of sdf_window_slice_map: row_map[b_ * N + n] = row of token n of window b_ in the
 * (rows, C) tensors, -1 = a padding token: a zero row behind the norm, its output dropped).  q | k | v never exist in memory.
 * Built for C = 96, nH = 3, N = 162 (sdf_ann_attn_block_supported); other shapes: SDF_E_SHAPE, the caller keeps
 * sdf_layer_norm_fwd + sdf_dense_linear_fwd + sdf_win_attn_fwd + sdf_dense_linear_fwd.  x == out is allowed (windows are disjoint).
 * wqkv: fp16 planes [2][3C][C] with W = plane0 + plane1 (plane0 = fp16(W), plane1 = fp16(W - plane0)), rows in the module's
 * order (q | k | v).  wproj: fp16 planes [2][C][C] of the projection weight with the input channels of every head g in the
 * kernel's accumulator order: wproj[., o, 32 g + 8 a + i] = Wproj[o, 32 g + (i < 4 ? 4 a + i : 16 + 4 a + i - 4)], a = 0..3, i = 0..7. */
typedef struct SdfAnnAttnBlockDesc {
  const float* x;            /* (rows, C) */
  float* out;                /* (rows, C) */
  const int32_t* row_map;    /* (B_ * N) */
  int32_t B_, nW, nH, N, C;
  int64_t rows;
  const float* ln_w;         /* (C) */
  const float* ln_b;         /* (C) */
  float ln_eps;
  const uint16_t* wqkv;      /* fp16 planes [2][3C][C] */
  const float* qkv_bias;     /* (3C) or NULL */
  const float* scale;        /* (nH): exp(min(logit_scale, ln 100)) * log2(e) */
  const float* table;        /* (nW, nH, N, N): (16 sigmoid(cpb_mlp(...))[h] + mask[w]) * log2(e), combined once per parameter version
                              * (nW = 1 and no mask for un-shifted windows): the softmax runs in the log2 domain and the score's
                              * additions are the matrix pipe's accumulator input */
  const uint16_t* wproj;     /* fp16 planes [2][C][C], input channels permuted (above) */
  const float* proj_bias;    /* (C) or NULL */
} SdfAnnAttnBlockDesc;

int sdf_ann_attn_block_supported(int C, int nH, int N);
int sdf_ann_attn_block_fwd(const SdfAnnAttnBlockDesc* d, void* stream);

/* ---- the MLP half of an ANN video-swin block as one launch (BASELINE config 3, first stage) ---------------------------
 * Replaces `SwinTransformerBlock3D.forward_part2` + the shortcut (reference models/STSwinNet/swin_transformer3D_v2.py:312-336) with
 * `Mlp.forward` (:15-34) inside:  out = x + fc2(GELU(fc1(LayerNorm(x)))), erf-form GELU (F.gelu), on rows x (rows, C) fp32; the hidden
 * activations never exist in memory.  Built for C = 96, hidden 384 (sdf_ann_mlp_block_supported); other shapes: SDF_E_SHAPE, the caller
 * keeps sdf_layer_norm_fwd + sdf_dense_linear_fwd (GELU) + sdf_dense_linear_fwd (shortcut).  x == out is allowed.
 * w1: fp16 planes [2][Ch][C] of fc1's weight (plane0 = fp16(W), plane1 = fp16(W - plane0)); w2: fp16 planes [2][C][Ch] of fc2's weight
 * with the hidden channels of every group of 32 in the kernel's accumulator order: w2[., o, 32 g + 8 a + i] =
 * W2[o, 32 g + (i < 4 ? 4 a + i : 16 + 4 a + i - 4)], a = 0..3, i = 0..7. */
typedef struct SdfAnnMlpBlockDesc {
  const float* x;            /* (rows, C) */
  float* out;                /* (rows, C) */
  int64_t rows;
  int32_t C, Ch;
  const float* ln_w;         /* (C) */
  const float* ln_b;         /* (C) */
  float ln_eps;
  const uint16_t* w1;        /* fp16 planes [2][Ch][C] */
  const float* b1;           /* (Ch) or NULL */
  const uint16_t* w2;        /* fp16 planes [2][C][Ch], hidden channels permuted (above) */
  const float* b2;           /* (C) or NULL */
} SdfAnnMlpBlockDesc;

int sdf_ann_mlp_block_supported(int C, int Ch);
int sdf_ann_mlp_block_fwd(const SdfAnnMlpBlockDesc* d, void* stream);

/* ---- dense 3x3 convolution of real-valued activations (ANN patch embedding, BASELINE config 3) ----------------------
 * Replaces the library convolutions of reference models/STSwinNet/PatchEmbed.py:166-196 (`PatchEmbedLocal`: head Conv2d +
 * four `ResidualBlock`s, models/submodules.py:160-229) together with their BatchNorm2d (eval: folded to alpha / beta), residual
 * add and ReLU:  out = act(alpha[n] * conv3x3(x, w)[n] + beta[n] (+ resid)), stride 1, zero padding 1.
 * Activations are "planes": [imgs][ceil(C/16)][H][W] records of 64 bytes, a record = 4 pieces of { 4 x fp16 hi, 4 x fp16 lo }
 * for 16 channels (value = hi + lo, 22 significant bits; sdf_pack_planes / sdf_unpack_planes convert from / to NCHW fp32).
 * w: fp16 planes [2][N][K] with w = plane0 + plane1, K = 9 * 16 * cin_records ordered (record, ky, kx, channel in record);
 * channels beyond Cin carry zero weights.  N % 32 == 0; cin_records in {1, 6}; every tensor below 2^31 bytes. */
typedef struct SdfDenseConvDesc {
  const void* x;            /* planes [imgs][cin_records][H][W][64 B] */
  const uint16_t* w;        /* fp16 planes [2][N][K] */
  const float* alpha;       /* (N) or NULL = 1 */
  const float* beta;        /* (N) or NULL = 0 */
  const void* resid;        /* planes [imgs][N/16][H][W][64 B] or NULL */
  void* out;                /* planes [imgs][N/16][H][W][64 B], or fp32 NHWC (imgs,H,W,N) when out_f32 */
  int32_t imgs, H, W, cin_records, N;
  int32_t relu;             /* max(., 0) after the residual add */
  int32_t out_f32;
  int32_t x_records;        /* records per image of the tensor x points into when x is a channel slice of a wider planes tensor
                             * (x = base + first_record * H * W * 64 bytes); 0 = cin_records.  Wide convolutions are chains of
                             * slices: out_k = conv(slice_k) + out_{k-1} through `resid`, the last link carries beta and relu */
  float acc_scale;          /* the planes hold scale * w (scale a power of two chosen at pack time so that max|w| fills the fp16
                             * range: small weights keep their 22 bits instead of sinking into fp16 subnormals); the accumulator is
                             * multiplied by acc_scale = 1 / scale (folded into alpha: exact).  0 = 1 */
} SdfDenseConvDesc;

int sdf_dense_conv3x3_fwd(const SdfDenseConvDesc* d, void* stream);
/* x (imgs,C,H,W) fp32 -> planes (channels C .. 16*ceil(C/16)-1 zero) and back */
int sdf_pack_planes(const float* x, void* planes, int imgs, int C, int H, int W, void* stream);
int sdf_unpack_planes(const void* planes, float* x, int imgs, int C, int H, int W, void* stream);
/* bilinear x2 upsampling (align_corners false; reference models/submodules.py:117-157 `UpsampleConvLayer`) of x (imgs,C,h,w) fp32
 * with element strides (sn,sc,sh,sw), written as records rec0 .. rec0+ceil(C/16)-1 of planes [imgs][rec_total][2h][2w] */
int sdf_pack_planes_up2(const float* x, void* planes, int imgs, int C, int h, int w, int64_t sn, int64_t sc, int64_t sh, int64_t sw,
                        int rec0, int rec_total, void* stream);
/* round 6: the same record addressing with ZERO INSERTION instead of interpolation - output pixel (2k, 2l) = input (k, l), all other
 * pixels 0: the input of ConvTranspose2d(3, stride 2, padding 1, output_padding 1) as the stride-1 correlation sdf_dense_conv3x3_fwd
 * runs with the flipped kernel (the SEW decoders, reference Spiking_modules.py:449-456), without a zero-filled fp32 image in between. */
int sdf_pack_planes_zero_up2(const float* x, void* planes, int imgs, int C, int h, int w, int64_t sn, int64_t sc, int64_t sh, int64_t sw,
                             int rec0, int rec_total, void* stream);

/* ---- Linear layer on real-valued activations (ANN swin blocks, BASELINE config 3) --------------------------------
 * Replaces F.linear (+ F.gelu, + the residual add) of reference models/STSwinNet/swin_transformer3D_v2.py:176-202
 * (`WindowAttention3D`: qkv, proj), :15-34 (`Mlp`: fc1 -> GELU -> fc2) and :272-313 (the block's two shortcut adds):
 *   out[m, n] = act(sum_k a[m, k] * w[n, k] + bias[n]) (+ resid[m, n]),  act = erf-form GELU when `gelu`, else identity.
 * a, resid, out: fp32 row-major (M, K) / (M, N); out may alias resid.  w: fp16 planes [2][N][K], w = plane0 + plane1
 * (hi = fp16(s w), lo = fp16(s w - hi), s = 1 / acc_scale).  N % 96 == 0, K % 32 == 0, every tensor below 2^31 bytes.
 * Activations are split into hi + lo fp16 in the loader: a value beyond the fp16 range (|x| > 65504) or a NaN becomes inf / NaN
 * and the affected outputs come out NaN - it is never silently replaced by a finite number. */
typedef struct SdfDenseLinearDesc {
  const float* a;
  const uint16_t* w;
  const float* bias;        /* (N) or NULL */
  const float* resid;       /* (M, N) or NULL */
  float* out;
  int32_t M, N, K;
  int32_t gelu;
  /* convolution form (cv_C > 0; all zero = plain Linear): `a` is a channels-last image (imgs, cv_H, cv_W, cv_C) fp32 and row m =
   * (img, oy, ox) gathers the 3x3 / pad 1 / stride cv_stride neighbourhood, K = 9 * cv_C ordered (ky, kx, c) - the strided
   * projection of reference models/STSwinNet/PatchEmbed.py:191 (`PatchEmbedLocal.proj`).  cv_OH / cv_OW = (H - 1) / stride + 1;
   * out_T > 1 writes image t * B + b of the input as image b * out_T + t of the output ((T,B) -> (B,T): the layout the swin
   * stages consume).  No residual in this form. */
  int32_t cv_H, cv_W, cv_C, cv_stride, cv_OH, cv_OW, out_T;
  float acc_scale;          /* 1 / (power-of-two scale the weight planes were packed with); multiplies the accumulator before the
                             * bias (exact).  0 = 1 */
} SdfDenseLinearDesc;

int sdf_dense_linear_fwd(const SdfDenseLinearDesc* d, void* stream);

/* Weight gradient of a Linear layer fed by spikes (training path, BASELINE configs[3]): dw[n, k] = sum_m dy[m, n] * x[m, k] - what
 * autograd computes for every nn.Linear of the MS swin blocks in the reference's step (train_flow_parallel_supervised_SNN.py:233-336
 * `loss.backward()`; layers Spiking_swin_transformer3D.py:661-717, :164-181, :952-974).  dy (M, N) and x (M, K) fp32 row-major, x
 * holding values exact in bf16 (spikes: 0 / 1); dy is split into three bf16 planes inside the kernel (exact), fp32 accumulation.
 * N % 96 == 0, K % 96 == 0.  nsplit = sdf_linear_dw_splits(M, N, K, cv_C) ranges of m are summed in a fixed order through `partial`
 * (nsplit x N x K fp32; may be NULL when nsplit == 1).  dw is overwritten.
 *
 * Convolution form (cv_C > 0): the weight gradient of a 3x3 / stride 1 / pad 1 convolution fed by spikes (the patch embedding's and
 * the bottleneck's MS_ResBlock convolutions, reference Spiking_modules.py:291-347; autograd's `conv2d_weight`) on ZERO-RINGED
 * channels-last images: row m is a pixel of the (H + 2) x cv_Wp grid of an image (cv_Wp = W + 2), dy (M, N) is zero on the ring, x is
 * (M, cv_C) with the ring zero (the convolution's padding), K = 9 cv_C and
 *   dw[n, (ky * 3 + kx) * cv_C + c] = sum_m dy[m, n] * x[m + (ky - 1) * cv_Wp + (kx - 1), c]       (rows outside [0, M) read as zero)
 * cv_C % 96 == 0. */
typedef struct SdfLinearDwDesc {
  const float* dy;
  const float* x;
  float* dw;
  float* partial;
  int64_t M;
  int32_t N, K, nsplit;
  int32_t cv_C, cv_Wp;
} SdfLinearDwDesc;

int sdf_linear_dw_splits(int64_t M, int N, int K, int cv_C);
int sdf_linear_dw_fwd(const SdfLinearDwDesc* d, void* stream);
/* src (imgs, C, H, W) fp32 -> dst (imgs, H + 2, W + 2, C) fp32, the ring zero: the layout of the convolution form above.  C % 96 == 0. */
int sdf_ringed_rows_fwd(const float* src, float* dst, int imgs, int C, int H, int W, void* stream);
/* the way back: src (imgs, H + 2, W + 2, C) ringed channels-last rows -> dst (imgs, C, H, W), + bias[c] when bias != NULL.  C % 96 == 0. */
int sdf_unring_rows_fwd(const float* src, const float* bias, float* dst, int imgs, int C, int H, int W, void* stream);

/* The two activation-side products of a Linear layer in the training path, from the fp32 tensors autograd holds (reference: nn.Linear
 * forward / autograd in train_flow_parallel_supervised_SNN.py:233-336; layers Spiking_swin_transformer3D.py:661-717, :164-181, :952-974):
 *   mode 0  out (M, N) = a (M, K) * w^T + bias     a holds spikes (values exact in bf16), w (N, K) fp32;   K % 32 == 0, N % 96 == 0
 *   mode 1  out (M, K) = a (M, N) * w              a = dY fp32 (any range), w (N, K) fp32;                 N % 32 == 0, K % 96 == 0
 * Real operands are split into three bf16 planes inside the kernel (exact; fp32 exponent range), fp32 accumulation.  bias: mode 0 only,
 * may be NULL.
 * Convolution form of mode 0 (cv_C > 0): the FORWARD of a 3x3 / stride 1 / pad 1 convolution fed by spikes (MS_ResBlock, reference
 * Spiking_modules.py:291-347) on the zero-ringed channels-last pixel rows of sdf_ringed_rows_fwd: a is (M, cv_C), w is (N, 9 cv_C) ordered
 * (ky, kx, c) [= weight.permute(0, 2, 3, 1)], K = 9 cv_C, out (M, N) on the same ringed grid (ring rows hold no result: sdf_unring_rows_fwd
 * drops them on the way back to NCHW). */
typedef struct SdfLinearTrainDesc {
  const float* a;
  const float* w;
  const float* bias;
  float* out;
  int32_t M, N, K;
  int32_t mode;
  int32_t cv_C, cv_Wp;
} SdfLinearTrainDesc;

int sdf_linear_train_fwd(const SdfLinearTrainDesc* d, void* stream);

/* nn.LayerNorm over the last dim of x (rows, C) fp32, elementwise affine (reference models/STSwinNet/swin_transformer3D_v2.py:
 * `norm1` / `norm2` of the blocks :231-233, `PatchMerging.norm` :356, the per-stage output norms :622-624).  C % 4 == 0, C <= 2048. */
int sdf_layer_norm_fwd(const float* x, const float* gamma, const float* beta, float* out, int64_t rows, int C, float eps, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SDFORMERFLOW_HIP_H */
