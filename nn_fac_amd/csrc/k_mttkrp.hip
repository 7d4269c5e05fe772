// MTTKRP of a dense 3-way tensor (ntf.py:448-449).  Placeholder entry point until the kernel lands.
#include "k_stream_common.h"

extern "C" int nnf_mttkrp3_f32(nnf_ctx*, const float*, int64_t, int64_t, int64_t, const float*, int64_t, const float*,
                               int64_t, const float*, int64_t, int, int, float*, int64_t, void*) { return NNF_ERR_UNSUPPORTED; }
