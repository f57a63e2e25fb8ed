// extern "C" surface of libbvi_hip.so (declared in include/bvi.h): argument
// checks, workspace sizing and dispatch to the per-family launchers.
#include "bvi_common.h"

extern "C" {

int bvi_version(void) { return BVI_VERSION; }

const char* bvi_status_string(int status) {
  switch (status) {
    case BVI_OK: return "ok";
    case BVI_ERR_ARG: return "invalid argument (null pointer, negative size, misaligned X or bad enum)";
    case BVI_ERR_UNSUPPORTED: return "shape or precision not supported by the gfx950 kernels";
    case BVI_ERR_WORKSPACE: return "workspace missing or smaller than bvi_workspace_bytes()";
    case BVI_ERR_TOO_LARGE:
      return "size beyond the fixed device scratch of the allocation-free entry point (at most 256 chains)";
    default: return status > 0 ? hipGetErrorString((hipError_t)status) : "unknown status";
  }
}

int bvi_workspace_bytes(int family, int64_t n_rows, int n_chains, int d, int k, size_t* out_bytes) {
  if (!out_bytes || n_rows < 0 || n_chains < 1) return BVI_ERR_ARG;
  switch (family) {
    case BVI_FAMILY_NORMAL_IID: *out_bytes = bvi_ws_normal_iid(n_rows, n_chains); break;
    case BVI_FAMILY_LINREG:
      if (d < 1 || d > 64) return BVI_ERR_UNSUPPORTED;
      *out_bytes = bvi_ws_linreg(n_rows, n_chains, d);
      break;
    case BVI_FAMILY_GMM1D:
      if (k < 1 || k > 16) return BVI_ERR_UNSUPPORTED;
      *out_bytes = bvi_ws_gmm1d(n_rows, n_chains, k);
      break;
    case BVI_FAMILY_GMM_FULLCOV:
      if (k < 1 || k > 16 || d < 1 || d > 32) return BVI_ERR_UNSUPPORTED;
      *out_bytes = bvi_ws_gmm_fullcov(n_rows, n_chains, d, k);
      break;
    case BVI_FAMILY_MVN_TRIL:
      if (d < 1 || d > 32) return BVI_ERR_UNSUPPORTED;
      *out_bytes = bvi_ws_mvn_tril(n_rows, n_chains, d);
      break;
    default: return BVI_ERR_UNSUPPORTED;
  }
  return BVI_OK;
}

int bvi_normal_iid_logp_grad(const float* y, int64_t n, const float* theta, int n_chains, float* logp,
                             float* grad, const double* logp_ref, void* ws, size_t ws_bytes, void* stream) {
  if ((!y && n > 0) || !theta || !logp || n < 0 || n_chains < 1) return BVI_ERR_ARG;
  if (!ws || ws_bytes < bvi_ws_normal_iid(n, n_chains)) return BVI_ERR_WORKSPACE;
  return bvi_launch_normal_iid(y, n, theta, n_chains, logp, grad, (float*)ws, (hipStream_t)stream, logp_ref);
}

int bvi_linreg_logp_grad(const float* X, const float* y, int64_t n, int d, const float* theta, int n_chains,
                         int precision, float* logp, float* grad, const double* logp_ref, void* ws, size_t ws_bytes,
                         void* stream) {
  if (((!X || !y) && n > 0) || !theta || !logp || n < 0 || n_chains < 1) return BVI_ERR_ARG;
  if (precision < BVI_PREC_AUTO || precision > BVI_PREC_BF16X6) return BVI_ERR_ARG;
  if (d < 1 || d > 64) return BVI_ERR_UNSUPPORTED;
  if (!ws || ws_bytes < bvi_ws_linreg(n, n_chains, d)) return BVI_ERR_WORKSPACE;
  return bvi_launch_linreg(X, y, n, d, theta, n_chains, precision, logp, grad, (float*)ws, (hipStream_t)stream,
                           logp_ref);
}

int bvi_linreg_packed_bytes(int64_t n, int d, int format, size_t* out_bytes) {
  if (!out_bytes || n < 0) return BVI_ERR_ARG;
  if (format == BVI_PACK_F16 ? (d < 1 || d > 64) : (d != 32 && d != 64)) return BVI_ERR_UNSUPPORTED;
  if (format == BVI_PACK_BF16X3) *out_bytes = bvi_linreg_packed_size(n, d);
  else if (format == BVI_PACK_F16) *out_bytes = bvi_linreg_f16_packed_size(n, d);
  else return BVI_ERR_ARG;
  return BVI_OK;
}

int bvi_linreg_recommended_format(int64_t n, int d, int n_chains, int* format) {
  if (!format || n < 0 || n_chains < 1) return BVI_ERR_ARG;
  if (d < 1 || d > 64) return BVI_ERR_UNSUPPORTED;
  const int few = d > 8 ? 16 : 32;
  *format = (n >= 16384 && (n_chains > few || (d > 8 && n >= 400000))) ? BVI_PACK_F16 : -1;
  return BVI_OK;
}

int bvi_linreg_pack_scratch_bytes(int d, int format, size_t* out_bytes) {
  if (!out_bytes) return BVI_ERR_ARG;
  if (format == BVI_PACK_F16 ? (d < 1 || d > 64) : (d != 32 && d != 64)) return BVI_ERR_UNSUPPORTED;
  if (format == BVI_PACK_BF16X3) *out_bytes = 0;
  else if (format == BVI_PACK_F16) *out_bytes = bvi_linreg_f16_pack_scratch(d);
  else return BVI_ERR_ARG;
  return BVI_OK;
}

int bvi_linreg_pack(const float* X, const float* y, int64_t n, int d, int format, void* packed,
                    size_t packed_bytes, void* ws, size_t ws_bytes, void* stream) {
  if (((!X || !y) && n > 0) || n < 0) return BVI_ERR_ARG;
  if (format == BVI_PACK_F16 ? (d < 1 || d > 64) : (d != 32 && d != 64)) return BVI_ERR_UNSUPPORTED;
  if (format == BVI_PACK_BF16X3) {
    if ((!packed && n > 0) || packed_bytes < bvi_linreg_packed_size(n, d)) return BVI_ERR_WORKSPACE;
    return bvi_launch_linreg_pack(X, y, n, d, packed, (hipStream_t)stream);
  }
  if (format == BVI_PACK_F16) {
    if (!packed || packed_bytes < bvi_linreg_f16_packed_size(n, d)) return BVI_ERR_WORKSPACE;
    if (!ws || ws_bytes < bvi_linreg_f16_pack_scratch(d)) return BVI_ERR_WORKSPACE;
    return bvi_launch_linreg_f16_pack(X, y, n, d, packed, ws, (hipStream_t)stream);
  }
  return BVI_ERR_ARG;
}

int bvi_linreg_packed_logp_grad(const void* packed, int64_t n, int d, int format, const float* theta,
                                int n_chains, int precision, float* logp, float* grad, const double* logp_ref,
                                void* ws, size_t ws_bytes, void* stream) {
  if ((!packed && n > 0) || !theta || !logp || n < 0 || n_chains < 1) return BVI_ERR_ARG;
  if (precision < BVI_PREC_AUTO || precision > BVI_PREC_BF16X6) return BVI_ERR_ARG;
  if (format == BVI_PACK_F16 ? (d < 1 || d > 64) : (d != 32 && d != 64)) return BVI_ERR_UNSUPPORTED;
  if (!ws || ws_bytes < bvi_ws_linreg(n, n_chains, d)) return BVI_ERR_WORKSPACE;
  if (format == BVI_PACK_BF16X3)
    return bvi_launch_linreg_packed(packed, n, d, theta, n_chains, precision, logp, grad, (float*)ws,
                                    (hipStream_t)stream, logp_ref);
  if (format == BVI_PACK_F16) {
    if (precision != BVI_PREC_AUTO) return BVI_ERR_UNSUPPORTED;
    return bvi_launch_linreg_f16(packed, n, d, theta, n_chains, logp, grad, (float*)ws, (hipStream_t)stream,
                                 logp_ref);
  }
  return BVI_ERR_ARG;
}

int bvi_gmm1d_logp_grad(const float* y, int64_t n, int k, const float* theta, int n_chains, float* logp,
                        float* grad, const double* logp_ref, void* ws, size_t ws_bytes, void* stream) {
  if ((!y && n > 0) || !theta || !logp || n < 0 || n_chains < 1) return BVI_ERR_ARG;
  if (k < 1 || k > 16) return BVI_ERR_UNSUPPORTED;
  if (!ws || ws_bytes < bvi_ws_gmm1d(n, n_chains, k)) return BVI_ERR_WORKSPACE;
  return bvi_launch_gmm1d(y, n, k, theta, n_chains, logp, grad, (float*)ws, (hipStream_t)stream, logp_ref);
}

int bvi_gmm_fullcov_logp_grad(const float* Y, int64_t n, int d, int k, const float* theta, int n_chains,
                              float* logp, float* grad, const double* logp_ref, void* ws, size_t ws_bytes,
                              void* stream) {
  if ((!Y && n > 0) || !theta || !logp || n < 0 || n_chains < 1) return BVI_ERR_ARG;
  if (k < 1 || k > 16 || d < 1 || d > 32) return BVI_ERR_UNSUPPORTED;
  if (!ws || ws_bytes < bvi_ws_gmm_fullcov(n, n_chains, d, k)) return BVI_ERR_WORKSPACE;
  return bvi_launch_gmm_fullcov(Y, n, d, k, theta, n_chains, logp, grad, (float*)ws, (hipStream_t)stream,
                                logp_ref);
}

int bvi_gmm_fullcov_packed_bytes(int64_t n, int d, size_t* out_bytes) {
  if (!out_bytes || n < 0) return BVI_ERR_ARG;
  if (d < 1 || d > 32) return BVI_ERR_UNSUPPORTED;
  *out_bytes = bvi_gmm_fullcov_packed_size(n, d);
  return BVI_OK;
}

int bvi_gmm_fullcov_pack_scratch_bytes(int d, size_t* out_bytes) {
  if (!out_bytes) return BVI_ERR_ARG;
  if (d < 1 || d > 32) return BVI_ERR_UNSUPPORTED;
  *out_bytes = bvi_gmm_fullcov_pack_scratch(d);
  return BVI_OK;
}

int bvi_gmm_fullcov_pack(const float* Y, int64_t n, int d, void* packed, size_t packed_bytes, void* ws,
                         size_t ws_bytes, void* stream) {
  if ((!Y && n > 0) || n < 0) return BVI_ERR_ARG;
  if (d < 1 || d > 32) return BVI_ERR_UNSUPPORTED;
  if (!packed || packed_bytes < bvi_gmm_fullcov_packed_size(n, d)) return BVI_ERR_WORKSPACE;
  if (!ws || ws_bytes < bvi_gmm_fullcov_pack_scratch(d)) return BVI_ERR_WORKSPACE;
  return bvi_launch_gmm_fullcov_pack(Y, n, d, packed, ws, (hipStream_t)stream);
}

int bvi_gmm_fullcov_packed_logp_grad(const float* Y, const void* packed, int64_t n, int d, int k, const float* theta,
                                     int n_chains, float* logp, float* grad, const double* logp_ref, void* ws,
                                     size_t ws_bytes, void* stream) {
  if ((!Y && n > 0) || !packed || !theta || !logp || n < 0 || n_chains < 1) return BVI_ERR_ARG;
  if (k < 1 || k > 16 || d < 1 || d > 32) return BVI_ERR_UNSUPPORTED;
  if (!ws || ws_bytes < bvi_ws_gmm_fullcov(n, n_chains, d, k)) return BVI_ERR_WORKSPACE;
  return bvi_launch_gmm_fullcov(Y, n, d, k, theta, n_chains, logp, grad, (float*)ws, (hipStream_t)stream, logp_ref,
                                packed);
}

int bvi_mvn_tril_packed_logp_grad(const float* Y, const void* packed, int64_t n, int d, int parameterisation,
                                  const float* theta, int n_chains, float* logp, float* grad, const double* logp_ref,
                                  void* ws, size_t ws_bytes, void* stream) {
  if ((!Y && n > 0) || !packed || !theta || !logp || n < 0 || n_chains < 1) return BVI_ERR_ARG;
  if (parameterisation != BVI_MVN_SCALE_TRIL && parameterisation != BVI_MVN_CHOL_PRECISION_SOLVE) return BVI_ERR_ARG;
  if (d < 1 || d > 32) return BVI_ERR_UNSUPPORTED;
  if (!ws || ws_bytes < bvi_ws_mvn_tril(n, n_chains, d)) return BVI_ERR_WORKSPACE;
  return bvi_launch_mvn_tril(Y, n, d, parameterisation, theta, n_chains, logp, grad, (float*)ws, (hipStream_t)stream,
                             logp_ref, packed);
}

int bvi_cluster_rows_scratch_bytes(int64_t n, int d, int n_centres, size_t* out_bytes) {
  if (!out_bytes || n < 0) return BVI_ERR_ARG;
  if (d < 1 || d > 32 || n_centres < 1 || n_centres > BVI_CLUSTER_MAX_CENTRES) return BVI_ERR_UNSUPPORTED;
  *out_bytes = bvi_ws_cluster_rows(n, d, n_centres);
  return BVI_OK;
}

int bvi_cluster_rows(const float* Y, int64_t n, int d, int n_centres, float* Y_out, int64_t* order, void* ws,
                     size_t ws_bytes, void* stream) {
  if (((!Y || !Y_out) && n > 0) || n < 0 || Y == Y_out) return BVI_ERR_ARG;
  if (d < 1 || d > 32 || n_centres < 1 || n_centres > BVI_CLUSTER_MAX_CENTRES) return BVI_ERR_UNSUPPORTED;
  if (n > 0 && (!ws || ws_bytes < bvi_ws_cluster_rows(n, d, n_centres))) return BVI_ERR_WORKSPACE;
  return bvi_launch_cluster_rows(Y, n, d, n_centres, Y_out, order, ws, (hipStream_t)stream);
}

int bvi_mvn_tril_logp_grad(const float* Y, int64_t n, int d, int parameterisation, const float* theta, int n_chains,
                           float* logp, float* grad, const double* logp_ref, void* ws, size_t ws_bytes,
                           void* stream) {
  if ((!Y && n > 0) || !theta || !logp || n < 0 || n_chains < 1) return BVI_ERR_ARG;
  if (parameterisation != BVI_MVN_SCALE_TRIL && parameterisation != BVI_MVN_CHOL_PRECISION_SOLVE) return BVI_ERR_ARG;
  if (d < 1 || d > 32) return BVI_ERR_UNSUPPORTED;
  if (!ws || ws_bytes < bvi_ws_mvn_tril(n, n_chains, d)) return BVI_ERR_WORKSPACE;
  return bvi_launch_mvn_tril(Y, n, d, parameterisation, theta, n_chains, logp, grad, (float*)ws, (hipStream_t)stream,
                             logp_ref);
}

}  // extern "C"
