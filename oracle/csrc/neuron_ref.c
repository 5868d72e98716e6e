/* CPU oracle (TEST INFRASTRUCTURE ONLY) - plain C restatement of the neuron arithmetic the HIP
 * kernels are checked against bit-for-bit.  Compiled by gcc with -ffp-contract=off so that the only
 * fused operations are the explicit fmaf() calls.
 *
 * lif:  reference call site models/STSwinNet_SNN/Spiking_modules.py:40-47 (spikingjelly LIFNode,
 *       third-party; equations per its 0.0.0.0.14 docs, see oracle/sdformer_oracle.py:lif_multistep)
 * psn:  models/STSwinNet_SNN/Spiking_submodules.py:207-211, H fixed to the k-ordered fmaf chain
 * bn :  eval BatchNorm as fmaf(x, alpha, beta) (ATen CPU kernel, see sdformer_oracle.py:bn_affine)
 */
#include <math.h>
#include <stdint.h>

/* x (T,N) fp32; optional affine per channel c = (n / inner) % C; optional add[t*add_st + n % add_period] */
static inline float prologue(float x, int64_t n, int t, const float* alpha, const float* beta, int C, int64_t inner,
                             const float* add, int64_t add_st, int64_t add_period) {
  if (alpha) {
    int c = (int)((n / inner) % C);
    x = fmaf(x, alpha[c], beta[c]);
  }
  if (add) x = x + add[(int64_t)t * add_st + n % add_period];
  return x;
}

void ref_lif(const float* x, float* s, float* v_last, int T, int64_t N, float tau, float v_th, int soft, float v_reset,
             const float* alpha, const float* beta, int C, int64_t inner, const float* add, int64_t add_st,
             int64_t add_period, int is_if) {
  for (int64_t n = 0; n < N; ++n) {
    float v = soft ? 0.f : v_reset;
    for (int t = 0; t < T; ++t) {
      float xi = prologue(x[(int64_t)t * N + n], n, t, alpha, beta, C, inner, add, add_st, add_period);
      float h;
      if (is_if) {
        h = v + xi;
      } else if (tau < 1.f) {     /* ParametricLIFNode: tau carries k = sigmoid(w), the charge multiplies (Spiking_modules.py:75-82) */
        h = (soft || v_reset == 0.f) ? v + (xi - v) * tau : v + (xi - (v - v_reset)) * tau;
      } else if (soft || v_reset == 0.f) {
        h = v + (xi - v) / tau;
      } else {
        h = v + (xi - (v - v_reset)) / tau;
      }
      float sp = (h - v_th >= 0.f) ? 1.f : 0.f;
      v = soft ? (h - sp * v_th) : ((1.f - sp) * h + sp * v_reset);
      s[(int64_t)t * N + n] = sp;
    }
    if (v_last) v_last[n] = v;
  }
}

void ref_psn(const float* x, const float* W, const float* b, float* s, float* hout, int T, int64_t N, const float* alpha,
             const float* beta, int C, int64_t inner, const float* add, int64_t add_st, int64_t add_period) {
  float xi[64];
  for (int64_t n = 0; n < N; ++n) {
    for (int t = 0; t < T; ++t)
      xi[t] = prologue(x[(int64_t)t * N + n], n, t, alpha, beta, C, inner, add, add_st, add_period);
    for (int t = 0; t < T; ++t) {
      float h = b[t];
      for (int k = 0; k < T; ++k) h = fmaf(W[t * T + k], xi[k], h);
      if (hout) hout[(int64_t)t * N + n] = h;
      s[(int64_t)t * N + n] = h >= 0.f ? 1.f : 0.f;
    }
  }
}
