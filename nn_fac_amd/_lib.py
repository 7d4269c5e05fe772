"""ctypes binding of libnnfac_hip.so (C ABI declared in include/nnfac_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C nn_fac_amd/csrc``.  There is no CPU fallback:
if the shared object is missing or cannot be loaded, every compute entry point raises ``EngineError``.
"""
import ctypes as C
import os

from .utils.errors import EngineError

_HERE = os.path.dirname(os.path.abspath(__file__))
# NNF_LIBRARY: another build of the same library (tools/: timing-only ablation builds side by side with the product's)
LIB_PATH = os.environ.get("NNF_LIBRARY") or os.path.join(_HERE, "libnnfac_hip.so")

_i64, _i32, _u32 = C.c_int64, C.c_int, C.c_uint
_p, _f32, _f64 = C.c_void_p, C.c_float, C.c_double

# name -> (restype, argtypes); mirrors include/nnfac_hip.h one to one
SIGNATURES = {
    "nnf_version": (_i32, []),
    "nnf_build_flags": (C.c_size_t, [C.c_char_p, C.c_size_t]),
    "nnf_status_string": (C.c_char_p, [_i32]),
    "nnf_ctx_create": (_i32, [C.POINTER(_p), _i32, C.c_size_t]),
    "nnf_ctx_destroy": (_i32, [_p]),
    "nnf_ctx_workspace_bytes": (C.c_size_t, [_p]),
    "nnf_comm_unique_id": (_i32, [_p]),
    "nnf_comm_create": (_i32, [C.POINTER(_p), _p, _i32, _i32, _p]),
    "nnf_comm_destroy": (_i32, [_p]),
    "nnf_comm_size": (_i32, [_p]),
    "nnf_comm_rank": (_i32, [_p]),
    "nnf_allreduce_f32": (_i32, [_p, _p, _i64, _p]),
    "nnf_allreduce_f64": (_i32, [_p, _p, _i64, _p]),
    "nnf_gram_f32": (_i32, [_p, _p, _i32, _i64, _i64, _p, _i64, _p]),
    "nnf_xht_f32": (_i32, [_p, _p, _i64, _i64, _i64, _p, _i32, _i64, _p, _i64, _p]),
    "nnf_xty_f32": (_i32, [_p, _p, _i64, _i64, _i64, _p, _i32, _i64, _p, _i64, _p]),
    "nnf_ctx_set_scratch": (_i32, [_p, _p, C.c_size_t]),
    "nnf_frob_resid_f32": (_i32, [_p, _p, _i64, _i64, _i64, _p, _i64, _p, _i64, _i32, _p, _p]),
    "nnf_nmf_gram_cost_f32": (_i32, [_p, _p, _i64, _p, _i64, _p, _p, _i64, _i32, _i64, _p, _p, _p]),
    "nnf_nmf_gram_cost_cal_f32": (_i32, [_p, _p, _i64, _p, _i64, _p, _p, _i64, _i32, _i64, _p, _f64, _f64, _p, _p]),
    "nnf_nmf_gram_cost_g64_f32": (_i32, [_p, _p, _i64, _p, _i64, _p, _p, _i64, _i32, _i64, _p, _f64, _f64, _f64, _p, _p]),
    "nnf_gram_f64_f32": (_i32, [_p, _p, _i32, _i64, _i64, _p, _i64, _p, _p]),
    "nnf_hals_solve_f32": (_i32, [_p, _p, _i64, _p, _i64, _p, _i64, _i32, _i64, _i32, _f64, _f32, _u32, _p, _p]),
    "nnf_hals_solve_cross_f32": (_i32, [_p, _p, _i64, _p, _p, _i64, _p, _i64, _p, _i64, _i32, _i64, _i32, _f64, _f32, _u32, _p,
                                        _p]),
    "nnf_hals_solve_continue_f32": (_i32, [_p, _p, _i64, _p, _i64, _p, _i64, _i32, _i64, _i32, _i32, _f64, _f32, _u32, _p,
                                           _p]),
    "nnf_hals_resident_columns": (_i32, [_p, _i32, C.POINTER(_i64)]),
    "nnf_hals_row_update_f32": (_i32, [_p, _p, _i64, _p, _i64, _p, _i64, _i32, _i64, _i32, _f32, _u32, _p, _p]),
    "nnf_hals_row_scale_f32": (_i32, [_p, _p, _i64, _i64, _i32, _p, _i64, _p]),
    "nnf_hals_stop_restore_f32": (_i32, [_p, _p, _i32, _i32, _i32, _f64, _p, _i64, _i32, _i64, _p, _i64, _p, _p]),
    "nnf_hals_sweeps_f32": (_i32, [_p, _p, _i64, _p, _i64, _p, _i64, _i32, _i64, _i32, _f32, _u32, _p, _p, _i64, _p]),
    "nnf_hals_sweeps_ex_f32": (_i32, [_p, _p, _i64, _p, _i64, _p, _i64, _i32, _i64, _i32, _i32, _f32, _u32, _p, _p, _i64, _i32,
                                      _p, _p, _p]),
    "nnf_hals_resid_floats": (_i32, [_p, _i32, _i64, C.POINTER(_i64)]),
    "nnf_ctx_set_probe": (_i32, [_p, _p, _p]),
    "nnf_ctx_set_probe_kernel": (_i32, [_p, _i32]),
    "nnf_ctx_set_probe_ring": (_i32, [_p, C.POINTER(_p), _i32]),
    "nnf_ctx_probe_ring_count": (_i32, [_p]),
    "nnf_mu_left_f32": (_i32, [_p, _p, _i64, _i64, _i64, _p, _i64, _p, _i64, _i32, _f64, _p, _i64, _p]),
    "nnf_mu_left_kl_cost_f32": (_i32, [_p, _p, _i64, _i64, _i64, _p, _i64, _p, _i64, _i32, _p, _i64, _p, _p]),
    "nnf_mu_right_f32": (_i32, [_p, _p, _i64, _i64, _i64, _p, _i64, _p, _i64, _i32, _f64, _p, _i64, _p]),
    "nnf_mu_right_accum_f32": (_i32, [_p, _p, _i64, _i64, _i64, _p, _i64, _p, _i64, _i32, _f64, _p, _i64, _p, _i64, _p, _p]),
    "nnf_ttm3_f32": (_i32, [_p, _p, _i64, _i64, _i64, _p, _i64, _i32, _i32, _p, _p]),
    "nnf_ntd_core_pg_f32": (_i32, [_p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _f64, _f64, _i32, _f64, _p, _p]),
    "nnf_mu_ratio_f32": (_i32, [_p, _p, _i64, _i64, _i64, _p, _i64, _p, _i64, _i32, _f64, _p, _p, _i64, _p]),
    "nnf_mu_apply_f32": (_i32, [_p, _p, _i64, _i32, _i64, _p, _i64, _p, _i64, _p, _f64, _p, _i64, _p]),
    "nnf_mu_left_num_f32": (_i32, [_p, _p, _i64, _i64, _i64, _p, _i64, _p, _i64, _i32, _p, _i64, _p]),
    "nnf_small_gemm_f32": (_i32, [_p, _p, _i64, _i32, _i32, _p, _i64, _i64, _p, _i64, _p]),
    "nnf_deep_kl_apply_f32": (_i32, [_p, _p, _i64, _i32, _i64, _p, _i64, _p, _p, _i64, _f64, _p, _i64, _p]),
    "nnf_betadiv_f32": (_i32, [_p, _p, _i64, _i64, _i64, _p, _i64, _p, _i64, _i32, _f64, _p, _p]),
    "nnf_mttkrp3_f32": (_i32, [_p, _p, _i64, _i64, _i64, _p, _i64, _p, _i64, _p, _i64, _i32, _i32, _p, _i64, _p]),
    "nnf_mttkrp3_from_partial_f32": (_i32, [_p, _p, _i64, _i64, _p, _i64, _i32, _i32, _p, _i64, _p]),
    "nnf_cp3_partial_cost_f32": (_i32, [_p, _p, _i64, _i64, _i64, _p, _i64, _p, _i64, _p, _i64, _i32, _p, _p, _p]),
    "nnf_cp3_betadiv_f32": (_i32, [_p, _p, _i64, _i64, _i64, _p, _i64, _p, _i64, _p, _i64, _i32, _f64, _p, _p]),
    "nnf_dot_f32": (_i32, [_p, _p, _i64, _p, _i64, _i64, _i64, _p, _p]),
    "nnf_hadamard_f32": (_i32, [_p, _p, _p, _p, _i64, _p]),
}

_lib = None


def load():
    """Load the shared library (once) and declare every prototype.  Raises EngineError when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C nn_fac_amd/csrc -j8` (there is no CPU fallback)")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise EngineError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(status, what=""):
    if status != 0:
        msg = load().nnf_status_string(status).decode()
        raise EngineError(f"{what}: libnnfac_hip status {status} ({msg})")
