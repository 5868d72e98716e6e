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
int launch_head(const HeadParams& P, dim3 grid, hipStream_t s) {
  if (P.d.Cin == 2 && P.d.Cout == 48) { hipLaunchKernelGGL((head_conv_sn_kernel<T, 3, 2>), grid, dim3(256), 0, s, P); return 0; }
  if (P.d.Cin == 2 && P.d.Cout == 32) { hipLaunchKernelGGL((head_conv_sn_kernel<T, 2, 2>), grid, dim3(256), 0, s, P); return 0; }
  if (P.d.Cin == 2 && P.d.Cout == 64) { hipLaunchKernelGGL((head_conv_sn_kernel<T, 4, 2>), grid, dim3(256), 0, s, P); return 0; }
  if (P.d.Cin == 4 && P.d.Cout == 48) { hipLaunchKernelGGL((head_conv_sn_kernel<T, 3, 4>), grid, dim3(256), 0, s, P); return 0; }
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
  int rc;
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
  hipLaunchKernelGGL(flow_out_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, sdf_stream(stream), pred, out, B, D, h, w,
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
  hipLaunchKernelGGL(deconv_col2im_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, sdf_stream(stream), Y, alpha, beta, out,
                     imgs, H, W, Cout);
  SDF_LAUNCH_CHECK();
  return 0;
}
