/*
 * nnfac_hip.h -- C ABI of libnnfac_hip.so, the MI355X (gfx950) inner-update engine behind
 * nn_fac's HALS-NNLS / beta-divergence MU hot path.
 *
 * The reference (ax-le/nn-fac) is pure Python: it has no FFI layer.  Each entry point below
 * names the reference statement(s) it replaces (file:line under /root/reference); the Python
 * host in nn_fac_amd/ keeps the reference's function signatures and calls these through ctypes
 * (INTEGRATION.md shows the stub a maintainer of the reference would add).
 *
 * Conventions
 *   - every pointer except `ctx`, `out_ctx` is a DEVICE pointer owned by the caller (fp32 unless
 *     the name says f64); the library never frees or keeps them;
 *   - matrices are row-major with an explicit leading dimension in ELEMENTS;
 *   - factors are passed "transposed": an m-by-r factor U is handed over as Ut (r-by-m, ld >= m), which is the
 *     layout hals_nnls_acc itself works on (nnls.py:147 takes U_in^T, V_in as r-by-n_cols);
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream) and
 *     allocates nothing: scratch comes from the context workspace (sized at nnf_ctx_create);
 *   - return value: 0 = NNF_OK, negative = error (nnf_status_string); no exception crosses the ABI;
 *   - a context is bound to one device and must not be used from two threads at once; one context
 *     per stream if calls on different streams may overlap (they share the workspace otherwise).
 *   - rank limit: 1 <= r <= 128 (NNF_ERR_UNSUPPORTED above).
 */
#ifndef NNFAC_HIP_H
#define NNFAC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NNF_OK 0
#define NNF_ERR_ARG (-1)         /* bad size / null pointer / bad flag combination            */
#define NNF_ERR_LAUNCH (-2)      /* HIP runtime error on launch (hipGetLastError)              */
#define NNF_ERR_UNSUPPORTED (-3) /* shape outside the built kernels (tensor rank > 128, ...)      */
#define NNF_ERR_WORKSPACE (-4)   /* context workspace too small for this call                  */
#define NNF_ERR_DEVICE (-5)      /* wrong / unavailable device                                 */

/* One launch per product / cost pass and the register- or LDS-resident sweep kernels up to this rank.  The matrix and CP entry
 * points (nnf_gram / xty / xht / frob_resid / betadiv / mu_ratio / nmf_gram_cost / hals_solve / hals_sweeps / hals_row_* /
 * mttkrp3 / mttkrp3_from_partial / cp3_betadiv / ttm3 along the first and last axis) go on above it, as the reference does
 * (nn_fac/nmf.py:175-178: any rank <= min(shape); nnls.py:158 loops range(r)): the contractions and cost passes walk the rank
 * in chunks of 128, the sweeps run in the generic kernel on columns in global memory.  The fused MU kernels, nnf_ttm3_f32
 * along the middle axis and the Tucker core update return NNF_ERR_UNSUPPORTED above it. */
#define NNF_MAX_RANK 128

/* hals flags */
#define NNF_HALS_SPARSITY 1u  /* subtract `sparsity` in the row update (nnls.py:162-164)       */
#define NNF_HALS_NORMALIZE 2u /* l2-normalise each row after its update (nnls.py:179-185)      */
#define NNF_HALS_NONZERO 4u   /* all-zero row -> 1e-16*max(V) (nnls.py:173-174)                */

typedef struct nnf_ctx nnf_ctx;

/* status block written by nnf_hals_solve_f32 (device memory, 8 doubles) */
#define NNF_HALS_ST_EPS 0     /* nodelta of the last executed sweep        (nnls.py:195)       */
#define NNF_HALS_ST_CNT 1     /* cnt as returned by the reference = sweeps done + 1 (:196)     */
#define NNF_HALS_ST_EPS0 2    /* nodelta of the first sweep                (nnls.py:188)       */
#define NNF_HALS_ST_ERR 3     /* 0 ok; 1 = grid barrier timed out (result invalid);
                                 2 = zero Gram diagonal met with NONZERO set (nnls.py:176-177) */
#define NNF_HALS_ST_WORDS 8

int nnf_version(void);
/* The values the timing-only ablation / A-B switches of the kernel sources were COMPILED with, one "unit: NAME=value ..."
 * entry per translation unit, sorted by unit, separated by "; " (e.g. "k_stream: XHT_ABL=0; ...").  Every switch has a
 * product default; a stray -D in a build would silently ship a kernel that skips work, so tests/test_abi_and_host.py
 * compares this string with the defaults.  Writes at most `cap` bytes (NUL-terminated), returns the full length. */
size_t nnf_build_flags(char* buf, size_t cap);
const char* nnf_status_string(int status);

/* Create a context on HIP device `device` with `workspace_bytes` of scratch (0 = default 256 MiB). */
int nnf_ctx_create(nnf_ctx** out_ctx, int device, size_t workspace_bytes);
int nnf_ctx_destroy(nnf_ctx* ctx);
size_t nnf_ctx_workspace_bytes(const nnf_ctx* ctx);
/* Caller-owned device scratch for temporaries that grow with the DATA: nnf_frob_resid_f32 / nnf_betadiv_f32 at a rank above
 * NNF_MAX_RANK build the m x n model over rank chunks in 4*m*ldp bytes, ldp = n rounded up to 4 (taken from here, else from
 * the tail of the context workspace, else NNF_ERR_WORKSPACE).  16-byte aligned; not freed by the library; must stay alive
 * until the calls using it have finished on their streams.  (NULL, 0) withdraws it. */
int nnf_ctx_set_scratch(nnf_ctx* ctx, void* device_buf, size_t bytes);

/* Measurement hook: two caller-owned hipEvent_t (passed as void*; NULL, NULL removes them) that nnf_xty_f32 records on its
 * launch stream immediately before and after its main kernel, so that a benchmark can time the dominant kernel alone --
 * without the slab reduction that follows it -- with HIP events on the stream the kernel runs on. */
int nnf_ctx_set_probe(nnf_ctx* ctx, void* ev_begin, void* ev_end);
/* Which main kernel the two events bracket (default NNF_PROBE_XTY): the streaming kernel of nnf_xty_f32 / nnf_xht_f32 /
 * the cost entry points / nnf_mu_left_f32 / nnf_mu_right_f32 / nnf_mttkrp3_f32, or the persistent sweep kernel of
 * nnf_hals_solve_f32 / nnf_hals_sweeps_f32 -- each without the small preparation / reduction kernels around it. */
#define NNF_PROBE_XTY 0
#define NNF_PROBE_XHT 1
#define NNF_PROBE_COST 2
#define NNF_PROBE_HALS 3
#define NNF_PROBE_MU_LEFT 4
#define NNF_PROBE_MU_RIGHT 5
#define NNF_PROBE_MTTKRP 6
#define NNF_PROBE_COUNT 7
int nnf_ctx_set_probe_kernel(nnf_ctx* ctx, int kernel_id);
/* The same hook for a whole timed region: `npairs` (begin, end) pairs of caller-owned events, events[2i], events[2i + 1];
 * the i-th launch of the selected kernel after this call records pair i, launches beyond `npairs` record nothing (no wrap).
 * The library copies the pointers.  NULL / 0 removes the ring.  While a ring is set it takes precedence over the single pair.
 * This is how bench.py times the dominant kernel on the launches INSIDE its timed loop (next to whatever shares the chip
 * with it there) rather than in a stand-alone loop. */
int nnf_ctx_set_probe_ring(nnf_ctx* ctx, void* const* events, int npairs);
/* Number of pairs of the current ring that have been recorded so far (0 without a ring). */
int nnf_ctx_probe_ring_count(const nnf_ctx* ctx);

/* ---- multi-GPU exchange of the row-sharded path: RCCL all-reduce (sum, in place) over xGMI -------------
 * One process per GPU.  Rank 0 draws a 128-byte id (nnf_comm_unique_id) and hands it to the other ranks by any channel of
 * the host application; every rank then calls nnf_comm_create(its ctx, nranks, rank, id) -- collective, like
 * ncclCommInitRank.  What a sharded NMF iteration exchanges (nn_fac_amd/nmf.py does the same through torch.distributed):
 *   V update: UtM (r x n) | UtU (r x r) in ONE buffer -> nnf_allreduce_f32;  U update: the per-sweep stopping sums
 *   (nnf_hals_sweeps_f32 -> nnf_allreduce_f64 -> nnf_hals_stop_restore_f32);  cost: one double -> nnf_allreduce_f64.
 * RCCL is bound at run time (dlopen); NNF_ERR_DEVICE when it is not available. */
typedef struct nnf_comm nnf_comm;
int nnf_comm_unique_id(void* id_out_128_bytes);
int nnf_comm_create(nnf_comm** out_comm, nnf_ctx* ctx, int nranks, int rank, const void* id_128_bytes);
int nnf_comm_destroy(nnf_comm* comm);
int nnf_comm_size(const nnf_comm* comm);
int nnf_comm_rank(const nnf_comm* comm);
int nnf_allreduce_f32(nnf_comm* comm, float* buf, int64_t count, void* stream);
int nnf_allreduce_f64(nnf_comm* comm, double* buf, int64_t count, void* stream);

/* G[r x r] = A[r x K] * A^T.   Replaces VVt = np.dot(V, V.T) (nmf.py:407), UtU = np.dot(U.T, U) (nmf.py:432),
 * and each factor Gram in ntf.py:442-445.  Split-K partials are summed in fp64 in a fixed order. */
int nnf_gram_f32(nnf_ctx* ctx, const float* A, int r, int64_t K, int64_t lda, float* G, int64_t ldg, void* stream);
/* The same Gram and, next to it, the sums BEFORE they are rounded to fp32 (G64: r x r doubles, contiguous) -- the split-K
 * partials are added in fp64 anyway.  For nnf_nmf_gram_cost_g64_f32: fp32 storage of U^T U alone (3.4e-8 relative rms per
 * entry) bounds the Gram-identity cost at ~1e-4 of a late-run cost at 10^6 x 4000 rank 100. */
int nnf_gram_f64_f32(nnf_ctx* ctx, const float* A, int r, int64_t K, int64_t lda, float* G, int64_t ldg, double* G64, void* stream);

/* out[r x m] = V[r x n] * X[m x n]^T.   Replaces VMt = np.dot(V, data.T) (nmf.py:408), the "X H^T" product. */
int nnf_xht_f32(nnf_ctx* ctx, const float* X, int64_t m, int64_t n, int64_t ldx, const float* V, int r, int64_t ldv,
                float* out, int64_t ldo, void* stream);

/* out[r x n] = Ut[r x m] * X[m x n].   Replaces UtM = np.dot(U.T, data) (nmf.py:433), the "W^T X" product.
 * Split over m across workgroups; partial slabs summed in a fixed order (bitwise reproducible). */
int nnf_xty_f32(nnf_ctx* ctx, const float* X, int64_t m, int64_t n, int64_t ldx, const float* Ut, int r, int64_t ldu,
                float* out, int64_t ldo, void* stream);

/* *out_f64 = sum_ij (X[i,j] - sum_k Ut[k,i] V[k,j])^2.   Replaces np.linalg.norm(data - U@V, 'fro')**2 (nmf.py:452)
 * without materialising U@V.  Per-lane fp32, then fp64 from the wave level up, fixed order. */
int nnf_frob_resid_f32(nnf_ctx* ctx, const float* X, int64_t m, int64_t n, int64_t ldx, const float* Ut, int64_t ldu,
                       const float* V, int64_t ldv, int r, double* out_f64, void* stream);

/* The same cost (nmf.py:452) from what a HALS iteration has on hand after its V update, without another pass over X:
 *     ||X - U V||^2 = ||X||^2 - 2 <V, U^T X> + sum_j v_j^T (U^T U) v_j
 * UtM = U^T X (r x n) and UtU (r x r) are the fp32 operands of the V update (nmf.py:432-433: they belong to the final U of
 * the iteration), V (r x n) is its result, *normx2_f64 = ||X||^2 (device scalar: computed once per run, all-reduced once in
 * a row-sharded run -- every other operand is replicated there, so this cost needs no collective).  The three inner
 * products and the quadratic forms are taken in fp64; the fp32 rounding of UtM and UtU leaves an absolute error of
 * ~1e-9 ||X||^2 (measured: tools/probes/gram_cost_probe.py), which is only acceptable while the
 * residual is not that small.  The kernel estimates it and says so:
 *   out_f64[0] = cost,  out_f64[1] = 0 if the estimate is below 5e-4 of the cost, else 1 (the caller then evaluates
 *   nnf_frob_resid_f32 for this iterate),  out_f64[2] = the estimate (4 sigma).
 * UtU_b (may be NULL): the Gram is the Hadamard product UtU .* UtU_b -- the `cross` of one_ntf_step (ntf.py:442-445), whose
 * own cost line IS this identity (ntf.py:462-470): V = the last updated factor (transposed), UtM = its MTTKRP right-hand side. */
int nnf_nmf_gram_cost_f32(nnf_ctx* ctx, const float* V, int64_t ldv, const float* UtM, int64_t ldm, const float* UtU, const float* UtU_b,
                          int64_t ldg, int r, int64_t n, const double* normx2_f64, double* out_f64, void* stream);
/* The same with the caller's figures for the rounding of the cross term: sigma_a = relative rms error of a UtM entry, bias_a =
 * |relative mean error| (both >= 0).  nnf_nmf_gram_cost_f32 assumes 6e-8 / 0 -- what the W^T X kernel leaves at 100000 rows;
 * the error grows with the rows one workgroup sums in fp32 (1e6 x 4000 rank 100 on one device: 9.5e-7 rms, -2.2e-7 mean,
 * tools/probes/accum_error_probe.py), and a mean error does not average down: the estimate becomes
 *     4 sqrt((2 sigma_a ||V .* UtM||_F)^2 + sigma_B^2) + 4 bias_a |<V, UtM>|.
 * A driver measures the two once per run (nn_fac_amd/nmf.py: the cross product summed in one piece against the same summed
 * in 16 row blocks). */
int nnf_nmf_gram_cost_cal_f32(nnf_ctx* ctx, const float* V, int64_t ldv, const float* UtM, int64_t ldm, const float* UtU,
                              const float* UtU_b, int64_t ldg, int r, int64_t n, const double* normx2_f64, double sigma_a,
                              double bias_a, double* out_f64, void* stream);
/* The same with the quadratic form taken on UtU64 (nnf_gram_f64_f32; UtU, the fp32 Gram the solve used, still provides max|UtU|)
 * and sigma_g = the caller's figure for the relative rms error of a UtU64 entry (what the fp32 accumulation inside a split
 * leaves) in place of the 4e-8 of fp32 storage in sigma_B = sigma_g max|UtU| ||V||_F^2.  No Hadamard form (NMF loop only). */
int nnf_nmf_gram_cost_g64_f32(nnf_ctx* ctx, const float* V, int64_t ldv, const float* UtM, int64_t ldm, const float* UtU,
                              const double* UtU64, int64_t ldg, int r, int64_t n, const double* normx2_f64, double sigma_a,
                              double bias_a, double sigma_g, double* out_f64, void* stream);

/* hals_nnls_acc (nnls.py:147-198) on device: V (r x ncols, in/out) is swept in place until
 *   eps >= delta*eps0 fails, or sweeps == max_sweeps                         (nnls.py:156)
 * max_sweeps is min(maxiter, floor(1+alpha*rho)) resolved by the host; the wall-clock rule of nnls.py:190-194 stays
 * on the host side (nn_fac_amd/update_rules/nnls.py).  UtU is r x r (ldg), UtM r x ncols (ldm).
 * status_f64: NNF_HALS_ST_WORDS doubles.  The sweep loop, the global sum of squared steps and the stopping decision
 * all run inside one persistent launch (grid barrier per sweep); nothing is read back by the host. */
int nnf_hals_solve_f32(nnf_ctx* ctx, const float* UtM, int64_t ldm, const float* UtU, int64_t ldg, float* V, int64_t ldv,
                       int r, int64_t ncols, int max_sweeps, double delta, float sparsity, unsigned flags,
                       double* status_f64, void* stream);

/* hals_nnls_acc as one_ntf_step calls it (ntf.py:442-456): Gram = UtU_a .* UtU_b (the `cross` of two factor Grams; UtU_b may be
 * NULL), start values V_in, result in V_out (may alias V_in) -- nnf_hals_solve_f32 without the Hadamard launch and the copy
 * in front of it.  max_sweeps <= 1000. */
int nnf_hals_solve_cross_f32(nnf_ctx* ctx, const float* UtM, int64_t ldm, const float* UtU_a, const float* UtU_b, int64_t ldg,
                             const float* V_in, int64_t ldvi, float* V_out, int64_t ldvo, int r, int64_t ncols, int max_sweeps,
                             double delta, float sparsity, unsigned flags, double* status_f64, void* stream);

/* max_sweeps above 1000 (one launch tags at most 1000 sweeps; nnf_hals_solve_f32 answers NNF_ERR_UNSUPPORTED): chain
 * nnf_hals_solve_f32(..., 1000, ...) with nnf_hals_solve_continue_f32(..., sweeps_done = 1000, 2000, ..., max_sweeps = the
 * next slice <= 1000, ...) on the same stream and status block.  A continuation whose predecessor already ended the solve
 * (nnls.py:156) returns at once; the block finally holds eps / cnt / eps0 of the whole solve.  No host round trip. */
int nnf_hals_solve_continue_f32(nnf_ctx* ctx, const float* UtM, int64_t ldm, const float* UtU, int64_t ldg, float* V,
                                int64_t ldv, int r, int64_t ncols, int sweeps_done, int max_sweeps, double delta,
                                float sparsity, unsigned flags, double* status_f64, void* stream);

/* Same sweeps, fixed count, no stopping rule: runs exactly `nsweeps` sweeps and writes the LOCAL sum of squared steps of
 * each sweep to nodelta_f64[0..nsweeps).  Building block of the row-sharded solve (the stopping scalar is all-reduced by
 * the host between chunks).
 * snapshots (may be NULL): nsweeps blocks of r x ncols floats (row stride ncols, block stride snap_stride elements); block s
 * receives V after sweep s+1, so an overshoot of the global stopping rule is undone by copying one block (no replay). */
int nnf_hals_sweeps_f32(nnf_ctx* ctx, const float* UtM, int64_t ldm, const float* UtU, int64_t ldg, float* V, int64_t ldv,
                        int r, int64_t ncols, int nsweeps, float sparsity, unsigned flags, double* nodelta_f64,
                        float* snapshots, int64_t snap_stride, void* stream);

/* The same blind sweeps as a CONTINUATION of a solve (the chunks of the row-sharded protocol; nnls.py:156-196 run in pieces):
 * `sweeps_done` sweeps of this solve have run in earlier calls.  At ranks 64..100 and many columns the sweeps run on the matrix
 * cores (k_hals_mfma.hip) and keep a scaled residual per column next to V; resid_in / resid_out (nnf_hals_resid_floats() floats
 * each, opaque layout, caller-owned; distinct buffers) hand it from call to call, and then the chunks of a solve leave bit for
 * bit what ONE call of all the sweeps leaves.  resid_in NULL (first chunk, or a caller that does not care): the residual is
 * formed from V; resid_out NULL: not kept.  The other kernel layouts carry no state and ignore both.
 * snap_first: the first sweep of this call (0-based) that writes a snapshot -- `head` blind sweeps and a window of snapshots
 * are ONE launch; snapshot block j receives V after sweep snap_first + j + 1 of this call (nsweeps - snap_first blocks). */
int nnf_hals_sweeps_ex_f32(nnf_ctx* ctx, const float* UtM, int64_t ldm, const float* UtU, int64_t ldg, float* V, int64_t ldv,
                           int r, int64_t ncols, int nsweeps, int sweeps_done, float sparsity, unsigned flags,
                           double* nodelta_f64, float* snapshots, int64_t snap_stride, int snap_first, const float* resid_in,
                           float* resid_out, void* stream);
/* Floats per residual-state buffer of nnf_hals_sweeps_ex_f32 for an r x ncols factor; 0 when the layout that runs keeps none. */
int nnf_hals_resid_floats(nnf_ctx* ctx, int r, int64_t ncols, int64_t* floats_out);

/* Row-sharded solves that normalise the SHARDED factor (nmf(normalize=[True, .]) over several ranks): the row
 * norm of nnls.py:179-185 runs over the columns of all ranks, once per row update, so the host walks the rows:
 *   nnf_hals_row_update_f32  row k of V (r x ncols: this rank's columns) gets the update of nnls.py:162-170;
 *                            out2_f64 = {sum of squared steps, sum of squares of the updated row} over the local columns
 *   (the caller all-reduces the two doubles)
 *   nnf_hals_row_scale_f32   row k /= sqrt(*normsq_f64), or := 1/sqrt(ncols_total) when the norm is 0 (nnls.py:181-185) */
int nnf_hals_row_update_f32(nnf_ctx* ctx, const float* UtM, int64_t ldm, const float* UtU, int64_t ldg, float* V, int64_t ldv, int r,
                            int64_t ncols, int k, float sparsity, unsigned flags, double* out2_f64, void* stream);
int nnf_hals_row_scale_f32(nnf_ctx* ctx, float* V, int64_t ldv, int64_t ncols, int k, const double* normsq_f64, int64_t ncols_total,
                           void* stream);

/* Columns the register-resident sweep kernel of rank r keeps on this device (one lane per column, all workgroups co-resident).
 * More columns than that: nnf_hals_solve_f32 / nnf_hals_sweeps_f32 stream the factor through HBM once per sweep and
 * nnf_hals_sweeps_f32 refuses `snapshots`; a caller that runs blind chunks of sweeps (the row-sharded protocol of nnls.py:156, or
 * the solve of a 10^6-column factor on one device) splits the columns into blocks of at most this many and adds the blocks'
 * per-sweep sums. */
int nnf_hals_resident_columns(nnf_ctx* ctx, int r, int64_t* columns_out);

/* Row-sharded solve, device-side stopping decision (no host round trip): after `nsweeps` blind sweeps of
 * nnf_hals_sweeps_f32 (snapshots for the last nsweeps - head of them) and an all-reduce of their per-sweep sums over the
 * ranks, replay nnls.py:156 over the sums: stop = first s with !(sum[s] >= delta*sum[0]) or s + 1 == budget.  A stop inside the
 * snapshot window restores V from that snapshot and writes {eps, cnt, eps0, 0} to status_f64; a stop before the window writes
 * error 3, no stop within these sweeps error 4 (the caller redoes the solve with a host-synchronous protocol). */
int nnf_hals_stop_restore_f32(nnf_ctx* ctx, const double* sums_f64, int nsweeps, int head, int budget, double delta, float* V,
                              int64_t ldv, int r, int64_t ncols, const float* snapshots, int64_t snap_stride,
                              double* status_f64, void* stream);

/* mu_betadivmin (mu.py:79-97) for the left factor, transposed storage:
 *   Ut_out[k,i] = max(Ut[k,i] * (num[k,i]/den[k,i])^gamma(beta), 1e-12),
 *   num = ((UV)^(beta-2) .* X) V^T, den = (UV)^(beta-1) V^T       (beta=1: den = rowsum(V); beta=2: Gram form)
 * One pass over X; U@V is never materialised.  Ut_out must not alias Ut.  beta != 2 needs r <= 64 (else
 * NNF_ERR_UNSUPPORTED). */
int nnf_mu_left_f32(nnf_ctx* ctx, const float* X, int64_t m, int64_t n, int64_t ldx, const float* Ut, int64_t ldu,
                    const float* V, int64_t ldv, int r, double beta, float* Ut_out, int64_t lduo, void* stream);

/* nnf_mu_left_f32 with beta = 1 that also returns *cost_f64 = beta_divergence(X, U V, 1) of the factors it STARTS from
 * (mu.py:84-88 + nmf.py:455): the update forms every entry of U V anyway, so the cost of outer iteration i is a by-product of
 * the left update of iteration i+1 and the separate pass over X (nnf_betadiv_f32) is only needed after the last one.
 * Same update as nnf_mu_left_f32 bit for bit.  r <= 64. */
int nnf_mu_left_kl_cost_f32(nnf_ctx* ctx, const float* X, int64_t m, int64_t n, int64_t ldx, const float* Ut, int64_t ldu,
                            const float* V, int64_t ldv, int r, float* Ut_out, int64_t lduo, double* cost_f64, void* stream);

/* switch_alternate_mu(..., "V") (mu.py:26-27): V_out[k,j] = max(V[k,j] * (num/den)^gamma, 1e-12) with
 *   num = U^T((UV)^(beta-2) .* X), den = U^T (UV)^(beta-1); split over m, fixed-order slab reduction. */
int nnf_mu_right_f32(nnf_ctx* ctx, const float* X, int64_t m, int64_t n, int64_t ldx, const float* Ut, int64_t ldu,
                     const float* V, int64_t ldv, int r, double beta, float* V_out, int64_t ldvo, void* stream);

/* Row-sharded right update (mu.py:79-97 on the transposed problem): the sums over the rows of X are
 * additive over row blocks, so every rank accumulates
 *   num = U^T((UV)^(beta-2) .* X)  (r x n)   and   den = U^T (UV)^(beta-1)  (r x n)
 * for its block (beta = 1: den_vec_f64[k] = sum_i U[i,k], r doubles, `den` unused; beta = 2: num = U^T X,
 * den = (U^T U) V), the caller all-reduces them, and nnf_mu_apply_f32 finishes:
 *   out[k,j] = max(F[k,j] * (num[k,j]/den[k,j])^gamma(beta), 1e-12)    (den_vec_f64 != NULL: den[k,j] = den_vec_f64[k]). */
int nnf_mu_right_accum_f32(nnf_ctx* ctx, const float* X, int64_t m, int64_t n, int64_t ldx, const float* Ut, int64_t ldu,
                           const float* V, int64_t ldv, int r, double beta, float* num, int64_t ldnum, float* den,
                           int64_t ldden, double* den_vec_f64, void* stream);
int nnf_mu_apply_f32(nnf_ctx* ctx, const float* F, int64_t ldf, int r, int64_t cols, const float* num, int64_t ldnum,
                     const float* den, int64_t ldden, const double* den_vec_f64, double beta, float* out, int64_t ldo,
                     void* stream);

/* Ranks beyond the fused kernels (64 < r <= 128, beta != 2): the element-wise operands of mu_betadivmin (mu.py:84-97),
 *   R1 = X .* (UV)^(beta-2)   and, unless beta == 1,   R2 = (UV)^(beta-1)        (both m x n, row stride ldr, caller-owned),
 * written in one pass over X; the two contractions are then plain nnf_xht_f32 / nnf_xty_f32 calls on R1 / R2 and
 * nnf_mu_apply_f32 finishes (nn_fac_amd/engine.py composes them when the fused entry points return NNF_ERR_UNSUPPORTED). */
int nnf_mu_ratio_f32(nnf_ctx* ctx, const float* X, int64_t m, int64_t n, int64_t ldx, const float* Ut, int64_t ldu,
                     const float* V, int64_t ldv, int r, double beta, float* R1, float* R2, int64_t ldr, void* stream);

/* Deep KL-NMF (deep_nmf.py:84-113, update_rules/deep_mu.py:8-14), the three device pieces of deep_KL_mu:
 *   nnf_mu_left_num_f32   num[k,i] = sum_j (X[i,j]/(UV)[i,j]) V[k,j]  -- the raw KL numerator of the left update (the fused
 *                         kernel of nnf_mu_left_f32 without its division by rowsum(V)); b = U .* num  (deep_mu.py:10); r <= 64;
 *   nnf_small_gemm_f32    out[p x cols] = A[p x q] B[q x cols], q <= 2048 (eight rows of A at a time in LDS) -- (W_{l+1} H_{l+1})^T = H_{l+1}^T W_{l+1}^T
 *                         (deep_nmf.py:93,109), also the rank-sized links of the NTD contraction chains (ntd.py:539-557);
 *   nnf_deep_kl_apply_f32 out[k,i] = max(1e-12, (b/lambda) / (W0(b exp(a/lambda)/lambda) + 1e-12)),  b = F .* num,
 *                         a[k,i] = hsum_f64[k] - lambda log(WHnext[k,i])   (deep_mu.py:9-12; hsum = row sums of H_l, i.e.
 *                         ONES @ H_l^T; W0 = principal Lambert W, evaluated in fp64 from the LOGARITHM of its argument so
 *                         that exp(a/lambda) cannot overflow). */
int nnf_mu_left_num_f32(nnf_ctx* ctx, const float* X, int64_t m, int64_t n, int64_t ldx, const float* Ut, int64_t ldu,
                        const float* V, int64_t ldv, int r, float* num, int64_t ldnum, void* stream);
int nnf_small_gemm_f32(nnf_ctx* ctx, const float* A, int64_t lda, int p, int q, const float* B, int64_t ldb, int64_t cols,
                       float* out, int64_t ldo, void* stream);
int nnf_deep_kl_apply_f32(nnf_ctx* ctx, const float* F, int64_t ldf, int r, int64_t cols, const float* num, int64_t ldnum,
                          const double* hsum_f64, const float* WHnext, int64_t ldw, double lambda, float* out, int64_t ldo,
                          void* stream);

/* beta_divergence(X, U@V, beta) (beta_divergence.py:45-52) fused with the product; *out_f64 = the sum. */
int nnf_betadiv_f32(nnf_ctx* ctx, const float* X, int64_t m, int64_t n, int64_t ldx, const float* Ut, int64_t ldu,
                    const float* V, int64_t ldv, int r, double beta, double* out_f64, void* stream);

/* MTTKRP of a dense 3-way tensor T[I x J x K] (C order) with the Khatri-Rao product of the two other factors
 * generated on the fly: replaces khatri_rao + np.dot(unfolded[mode], krao) (ntf.py:448-449).
 * Factors are passed transposed (Ft_a: R x dim_a, ld = ld_a).  out is R x dim_mode (the rhs^T hals_nnls_acc wants). */
int nnf_mttkrp3_f32(nnf_ctx* ctx, const float* T, int64_t I, int64_t J, int64_t K, const float* Ft0, int64_t ld0,
                    const float* Ft1, int64_t ld1, const float* Ft2, int64_t ld2, int R, int mode, float* out,
                    int64_t ldo, void* stream);

/* MTTKRP from a shared partial product (dimension tree).  F2 does not change between the mode-0 and the mode-1 update of
 * one_ntf_step (ntf.py:437-456), so both right-hand sides are contractions of Y[r][i][j] = sum_k T[i][j][k] F2[k][r]
 * (= nnf_ttm3_f32(T, F2t, mode 2): ONE pass over T):  axis 2: out[r][a] = sum_b Y[r][a][b] Ft[r][b] (mode 0, Ft = F1t);
 * axis 1: out[r][b] = sum_a Y[r][a][b] Ft[r][a] (mode 1, Ft = the updated F0t).  Y is R x A x B, contiguous. */
int nnf_mttkrp3_from_partial_f32(nnf_ctx* ctx, const float* Y, int64_t A, int64_t B, const float* Ft, int64_t ldf, int R,
                                 int axis, float* out, int64_t ldo, void* stream);

/* Fused pass over T for an NTF iteration loop: the squared residual of the CURRENT CP model (ntf.py:470, evaluated
 * directly) and the partial product Y[r][i][j] = sum_k T[i][j][k] F2[k][r] the NEXT iteration's mode-0 / mode-1 right-hand
 * sides are contracted from -- both need the final factors and the whole tensor.  With the mode-2 MTTKRP an iteration then
 * reads T twice instead of four times.  R <= 64 (NNF_ERR_UNSUPPORTED above: use nnf_cp3_betadiv_f32 + nnf_ttm3_f32). */
int nnf_cp3_partial_cost_f32(nnf_ctx* ctx, const float* T, int64_t I, int64_t J, int64_t K, const float* Ft0, int64_t ld0,
                             const float* Ft1, int64_t ld1, const float* Ft2, int64_t ld2, int R, float* Y, double* cost_f64,
                             void* stream);

/* beta_divergence(T, [[F0,F1,F2]], beta) for a dense 3-way tensor and its CP model (factors transposed, R x dim): the cost
 * of ntf.py:470 (HALS: 2x the beta=2 value = ||T - model||^2) and ntf.py:473 (MU), Khatri-Rao operand generated on the fly. */
int nnf_cp3_betadiv_f32(nnf_ctx* ctx, const float* T, int64_t I, int64_t J, int64_t K, const float* Ft0, int64_t ld0,
                        const float* Ft1, int64_t ld1, const float* Ft2, int64_t ld2, int R, double beta, double* out_f64,
                        void* stream);

/* small helpers used by the drivers (all deterministic, fixed-order) */
/* Mode-n product of the 3-way data tensor with a transposed factor: out = T x_mode F^T
 * (tl.tenalg.mode_dot(T, F.T, mode); the contractions of ntd.py:550,581 and of mu_tensorial, mu.py:159, are chains of
 * these).  T is I x J x K row-major, Ft is r x I_mode (row stride ldf).  The new axis comes out FIRST for the two outer
 * modes -- mode 0: out[r][J][K], mode 2: out[r][I][J] -- and in place for the middle one -- mode 1: out[I][r][K]. */
int nnf_ttm3_f32(nnf_ctx* ctx, const float* T, int64_t I, int64_t J, int64_t K, const float* Ft, int64_t ldf, int r, int mode,
                 float* out, void* stream);

/* Projected-gradient update of the NTD core (ntd.py:588-619) and the Gram-form reconstruction error (ntd.py:639), one
 * single-workgroup launch, fp64 arithmetic in LDS:
 *   step = round(prod_i 1/sigma_max(M_i), 6);  repeat (<= max_iter, while update >= delta * first update):
 *       grad = core x_0 M0 x_1 M1 x_2 M2 - MtX + sparse;  core -= min(step*grad, core)
 * core (d0 x d1 x d2, updated in place), MtX same shape, M_i = F_i^T F_i (d_i x d_i, dense).
 * status_f64[6] = {iterations, last update norm, first update norm, step, norm_sq - 2<MtX,core> + <core x M, core>, 0}.
 * Cores whose four fp64 copies exceed one workgroup's LDS (~4500 entries) work out of the context workspace instead. */
int nnf_ntd_core_pg_f32(nnf_ctx* ctx, float* core, const float* MtX, const float* M0, const float* M1, const float* M2, int d0,
                        int d1, int d2, double sparse, double delta, int max_iter, double norm_sq, double* status_f64,
                        void* stream);

/* *out_f64 = sum_ij A[i,j]*B[i,j]   (fp64 accumulate)  -- inner products of ntf.py:470 */
int nnf_dot_f32(nnf_ctx* ctx, const float* A, int64_t lda, const float* B, int64_t ldb, int64_t rows, int64_t cols,
                double* out_f64, void* stream);
/* C = A .* B elementwise, r x r  (Hadamard of Grams, ntf.py:442-445) */
int nnf_hadamard_f32(nnf_ctx* ctx, const float* A, const float* B, float* C, int64_t count, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NNFAC_HIP_H */
