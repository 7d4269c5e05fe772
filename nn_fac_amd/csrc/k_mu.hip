// beta-divergence multiplicative updates and cost (mu.py:79-97, beta_divergence.py:45-52).  Placeholder entry points
// until the fused two-GEMM kernels land (they return NNF_ERR_UNSUPPORTED; the Python host raises).
#include "k_stream_common.h"

extern "C" int nnf_mu_left_f32(nnf_ctx*, const float*, int64_t, int64_t, int64_t, const float*, int64_t, const float*,
                               int64_t, int, double, float*, int64_t, void*) { return NNF_ERR_UNSUPPORTED; }
extern "C" int nnf_mu_right_f32(nnf_ctx*, const float*, int64_t, int64_t, int64_t, const float*, int64_t, const float*,
                                int64_t, int, double, float*, int64_t, void*) { return NNF_ERR_UNSUPPORTED; }
extern "C" int nnf_betadiv_f32(nnf_ctx*, const float*, int64_t, int64_t, int64_t, const float*, int64_t, const float*,
                               int64_t, int, double, double*, void*) { return NNF_ERR_UNSUPPORTED; }
