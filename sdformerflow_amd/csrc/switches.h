// Diagnostic switches of the library (the SDF_* environment variables of INTEGRATION.md's appendix; none is needed in production).
// They are read from the environment ONCE - at the first C-ABI call that asks for one - into a process-wide read-only table, so no
// entry point calls getenv() on its per-call path ("no global mutable state except a lazily-built, read-only cache").
// A test harness that changes the environment between calls says so explicitly: sdf_switches_reload() (include/sdformerflow_hip.h).
#pragma once
#define SDF_SWITCH_LIST(X)                                                                                                        \
  X(DENSE_TPW) X(DENSE_LINEAR_XCD) X(HEAD_MFMA) X(MLP_FUSED_ANY) X(MLP_FUSED) X(RES) X(RES_STRIP) X(RES_UPW) X(RES_RMUL)            \
  X(RES_MINC) X(RES_MAXC) X(SMALLM) X(SMALLM_CONV_ROWS) X(SMALLM_CB) X(SMALLM_FC2) X(WIDE) X(WIDE_CB) X(WIDE_PASSES)                \
  X(WIDE_MAXROWS) X(WIDE_CONV) X(QK_FRONT) X(QK_FRONT_ANY) X(CONV_WRES) X(CONV_WRES_RB) X(CONV_WRES_GROUPS) X(CONV_WRES_CB_INNER)   \
  X(CONV_WRES_NOSPK) X(CONV_MULTI) X(DECONV_WRES) X(DECONV_BALANCE) X(DECONV_EPI) X(GEMM_CFG) X(GEMM_WGS) X(PP_PAIR)     \
  X(ATTN_GENERIC) X(ATTN_F32) X(KSPLIT_MULT) X(GEMM_WS)
enum SdfSwitch {
#define X(n) SW_##n,
  SDF_SWITCH_LIST(X)
#undef X
      SW_COUNT
};
// Value of SDF_<name> as it stood when the table was (re)built, or nullptr when unset: a drop-in for getenv("SDF_<name>").
const char* sdf_sw(SdfSwitch s);
