// Accelerated HALS NNLS sweeps (nn_fac/update_rules/nnls.py:147-198) as ONE persistent launch.
//
// Data: V (r x ncols, row-major) is updated in place; UtM (r x ncols) and UtU (r x r) are read-only.
// Every column of V is an independent Gauss-Seidel problem; the only cross-column coupling is the
// stopping scalar sum(step^2) (and, with NORMALIZE / NONZERO, a row norm / row-all-zero / max(V)).
//
// Fast path (flags subset of {SPARSITY}):
//   one lane = one column; the column of V (and of UtM when RP <= 64) lives in VGPRs for the whole solve;
//   the padded Gram (RP x RP) and 1/diag are read through the scalar cache (wave-uniform operands -> s_load + v_fmac with
//   an SGPR source), so a sweep costs RP^2 VALU FMAs per column and no LDS or HBM traffic at all.
//   Per sweep the workgroups exchange ONE double each through a grid barrier (write-through store of the partial,
//   vmcnt drain, agent-scope counter add, relaxed poll, every workgroup re-sums all partials in index order so all
//   reach the same decision bit for bit; cdna_hip_programming.md Guideline 16, R1 form).  Spins are bounded.
//   If ncols exceeds the resident thread count the same kernel strides over column sets, re-reading its own
//   stores (no cross-workgroup hand-off of V is ever needed).
// Generic path (NORMALIZE / NONZERO, any r): columns in LDS, run-time rank loop, one grid reduction per row.
#include "k_hals_common.h"
#include <cstdlib>

// prep: padded Gram, 1/diag, zeroed barrier words and status
// One workgroup per Gram row (a single workgroup walked the 50 x 64 image in 13 dependent round trips: 11.5 us in front of
// every solve); block 0 also writes the per-row pairs, the all-live flag, the barrier word and the status block.
__global__ void nnf_hals_prep_kernel(const float* __restrict__ UtU, int64_t ldg, int r, int RP, float* __restrict__ Gp,
                                     float* __restrict__ dinv, float* __restrict__ Gs, unsigned* counter, double* status) {
    const int RS = 32 * ((RP + 31) / 32);      // row stride of the padded Gram (32-float blocks, k_hals_fast.hip)
    const int a = blockIdx.x;                   // 0 .. RP-1
    const float da = (a < r) ? UtU[(int64_t)a * ldg + a] : 0.f;
    const float dia = (da != 0.f) ? (float)(1.0 / (double)da) : 0.f;
    for (int b = threadIdx.x; b < RS; b += blockDim.x) {
        const float g = (a < r && b < r) ? UtU[(int64_t)a * ldg + b] : 0.f;
        Gp[a * RS + b] = g;
        if (Gs) Gs[a * RS + b] = g * dia;      // rows scaled by 1/diag (rows with a zero diagonal: all zero), same padding
    }
    if (blockIdx.x != 0) return;
    for (int k = threadIdx.x; k < RP; k += blockDim.x) {
        const float d = (k < r) ? UtU[(int64_t)k * ldg + k] : 0.f;
        dinv[2 * k] = (d != 0.f) ? (float)(1.0 / (double)d) : 0.f;   // pair (1/diag, nz): nz = 0 = leave the row alone
        dinv[2 * k + 1] = (d != 0.f) ? 1.f : 0.f;
    }
    {   // all-live flag: no zero on the diagonal of the r x r Gram
        int dead = 0;
        for (int i = threadIdx.x; i < r; i += blockDim.x) dead |= (UtU[(int64_t)i * ldg + i] == 0.f) ? 1 : 0;
        const int any_dead = __syncthreads_or(dead);
        if (threadIdx.x == 0) dinv[2 * RP] = any_dead ? 0.f : 1.f;
    }
    if (threadIdx.x == 0) {
        *counter = 0u;
        if (status) {
            status[NNF_HALS_ST_EPS] = 1.0;
            status[NNF_HALS_ST_CNT] = 1.0;
            status[NNF_HALS_ST_EPS0] = 0.0;
            status[NNF_HALS_ST_ERR] = 0.0;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Generic path: any r, all flags.  128 threads per workgroup, one column per thread, column in LDS
// (vl[k*128 + tid]); each row update may need a grid reduction (row sum of squares, row non-zero count, max V).
// Requires all workgroups resident (grid sized by the host) -- columns beyond the resident set are strided.
// ---------------------------------------------------------------------------------------------------------
// GCOL (ranks above NNF_MAX_RANK, where r x 128 columns no longer fit the LDS): a thread's column is the column of V itself
// in global memory -- read and written in place, coalesced across the threads of a wave, served by L1/L2 (a workgroup's
// r x 128 block of V is 100 KB at rank 200); no load / store phase.  Same arithmetic in the same order as the LDS form.
// BIG (ranks above NNF_MAX_RANK): the row's dot product in eight independent partial sums over batches of eight entries -- the
// loads of a batch go out together (a plain `dot = fmaf(G[i], v[i], dot)` waits for one LDS / memory round trip per entry: 43 ns
// per entry at rank 200, 1.7 ms per sweep).  Up to NNF_MAX_RANK the sum keeps the reference's order (nnls.py:162: np.dot).
template <int MODE, bool GCOL, bool BIG = GCOL>
__global__ __launch_bounds__(128) void nnf_hals_generic_kernel(const float* __restrict__ UtM, int64_t ldm,
                                                               const float* __restrict__ Gp, const float* __restrict__ dinv,
                                                               int RP, float* __restrict__ V, int64_t ldv, int r,
                                                               int64_t ncols, int max_sweeps, double delta, float sp,
                                                               unsigned flags, hals_sync sy, double* __restrict__ status,
                                                               double* __restrict__ sweep_partials, int sweep0,
                                                               float* __restrict__ snapshots, int64_t snap_stride, int snap_first) {
    // LPC lanes per column.  BIG with the column in LDS: FOUR (lane q of a quad takes the entries i = q (mod 4) of a row's dot
    // product, two shuffles add the quarters -- the same bits in all four lanes -- every lane forms the step, lane 0 keeps it):
    // a thread's dependent chain is r*r/4 entries instead of r*r, and a workgroup's 32 columns are r x 32 floats of LDS, so
    // that several workgroups share a CU.  The lanes of a column sit in one wave, whose LDS operations execute in order.
    constexpr int LPC = (BIG && !GCOL) ? 4 : 1;
    constexpr int CW = 128 / LPC;                                     // columns per workgroup
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* vl = reinterpret_cast<float*>(smem);                       // [r][CW]  (GCOL: unused)
    double* red = reinterpret_cast<double*>(smem + (GCOL ? (size_t)0 : (size_t)r * CW * 4) + 16);
    __shared__ unsigned lds_flag;
    const int nblocks = gridDim.x;
    const int q = threadIdx.x % LPC;                                  // this lane's share of a column (0: the lane that keeps it)
    const int64_t gcolumn = (int64_t)blockIdx.x * CW + threadIdx.x / LPC;
    const bool rowsync = (flags & (NNF_HALS_NORMALIZE | NNF_HALS_NONZERO)) != 0;
    // with row-level grid reductions every thread must take part in every exchange: one column per thread (quad) only
    const bool active = gcolumn < ncols;
    const bool owner = active && q == 0;
    unsigned epoch = 0;
    double eps0 = 0.0, eps = 1.0;
    int done = 0, err = 0;
    bool ok = true;
    if (MODE == 0 && sweep0 > 0 && !hals_take_over(status, sweep0, delta, eps0, eps)) return;
    const int64_t col0 = active ? gcolumn : 0;
    float* mycol = GCOL ? V + col0 : vl + threadIdx.x / LPC;         // element k of the column: mycol[k * cs]
    const int64_t cs = GCOL ? ldv : CW;
    if constexpr (!GCOL) {
        for (int k = q; k < r; k += LPC) mycol[k * CW] = active ? V[(int64_t)k * ldv + col0] : 0.f;
        if constexpr (LPC > 1) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
    for (int s = 1; s <= max_sweeps && ok; ++s) {
        double nd = 0.0;
        for (int k = 0; k < r; ++k) {
            const float di = dinv[2 * k];
            if (di != 0.f) {
                float dot = 0.f;
                if constexpr (BIG) {
                    const float* gk = Gp + (size_t)k * RP;
                    float d[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    int i = q;
                    for (; i + 7 * LPC < r; i += 8 * LPC) {                    // eight entries of this lane: i, i + LPC, ...
                        float gv[8], vv[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) { gv[u] = gk[i + u * LPC]; vv[u] = mycol[(int64_t)(i + u * LPC) * cs]; }
#pragma unroll
                        for (int u = 0; u < 8; ++u) d[u] = fmaf(gv[u], vv[u], d[u]);
                    }
                    for (int u = 0; i < r; i += LPC, ++u) d[u & 7] = fmaf(gk[i], mycol[(int64_t)i * cs], d[u & 7]);
                    dot = ((d[0] + d[1]) + (d[2] + d[3])) + ((d[4] + d[5]) + (d[6] + d[7]));
                    if constexpr (LPC > 1) {
                        dot += __shfl_xor(dot, 1, 64);
                        dot += __shfl_xor(dot, 2, 64);
                    }
                } else
                for (int i = 0; i < r; ++i) dot = fmaf(Gp[k * RP + i], mycol[i * cs], dot);
                const float vk = mycol[k * cs];
                float step = fmaxf((UtM[(int64_t)k * ldm + col0] - dot - sp) * di, -vk);
                if (!active) step = 0.f;
                if constexpr (LPC > 1) {
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // (every lane of the quad has read v[k])
                    if (q == 0) mycol[k * cs] = vk + step;
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                    if (q == 0) nd += (double)step * (double)step;
                } else {
                    if (!GCOL || active) mycol[k * cs] = vk + step;
                    nd += (double)step * (double)step;
                }
            } else if (flags & NNF_HALS_NONZERO) {
                err = 2;   // nnls.py:176-177
            }
            if (rowsync) {
                const float vk = mycol[k * cs];
                double vmax = 0.0;
                if (flags & NNF_HALS_NONZERO) {
                    for (int i = q; i < r; i += LPC) vmax = fmax(vmax, (double)mycol[i * cs]);
                    if constexpr (LPC > 1) {
                        vmax = fmax(vmax, __shfl_xor(vmax, 1, 64));
                        vmax = fmax(vmax, __shfl_xor(vmax, 2, 64));
                    }
                }
                double mine[3] = {owner ? (double)vk * (double)vk : 0.0, (owner && vk != 0.f) ? 1.0 : 0.0,
                                  active ? vmax : -1.0e300};
                double tot[3];
                const double b0 = nnf_block_sum_f64(mine[0], red);
                const double b1 = nnf_block_sum_f64(mine[1], red);
                // block max of mine[2]
                double bm = mine[2];
                for (int o = 32; o > 0; o >>= 1) { const double y = __shfl_down(bm, o, 64); bm = y > bm ? y : bm; }
                if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = bm;
                __syncthreads();
                bm = red[0] > red[1] ? red[0] : red[1];
                __syncthreads();
                double pub[3] = {b0, b1, bm};
                if (threadIdx.x != 0) { pub[0] = 0; pub[1] = 0; }
                ok = grid_exchange<3>(sy, ++epoch, nblocks, pub, tot, red, &lds_flag);
                if (!ok) break;
                if ((flags & NNF_HALS_NONZERO) && di != 0.f && tot[1] == 0.0 && owner)
                    mycol[k * cs] = (float)(1e-16 * tot[2]);            // nnls.py:173-174
                if constexpr (LPC > 1) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                if (flags & NNF_HALS_NORMALIZE) {
                    // the norm is taken after the NONZERO refill (nnls.py:179-185)
                    double nsq = tot[0];
                    if ((flags & NNF_HALS_NONZERO) && di != 0.f && tot[1] == 0.0) {
                        const double f = 1e-16 * tot[2];
                        nsq = f * f * (double)ncols;
                    }
                    if (owner) {
                        if (nsq != 0.0) mycol[k * cs] = (float)((double)mycol[k * cs] / sqrt(nsq));
                        else mycol[k * cs] = (float)(1.0 / sqrt((double)ncols));
                    }
                    if constexpr (LPC > 1) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                }
            }
        }
        if (!ok) break;
        done = s;
        const double bs = nnf_block_sum_f64(nd, red);
        if (MODE == 1) {
            if (threadIdx.x == 0) sweep_partials[(size_t)(s - 1) * nblocks + blockIdx.x] = bs;
            if (snapshots != nullptr && s > snap_first && active) {   // block s - 1 - snap_first: V after sweep s
                float* sn = snapshots + (int64_t)(s - 1 - snap_first) * snap_stride + col0;
                for (int k = q; k < r; k += LPC) sn[(int64_t)k * ncols] = mycol[k * cs];
            }
        } else {
            double mine[1] = {bs}, tot[1];
            ok = grid_exchange<1>(sy, ++epoch, nblocks, mine, tot, red, &lds_flag);
            if (!ok) break;
            if (s == 1 && sweep0 == 0) eps0 = tot[0];
            eps = tot[0];
            if (!(eps >= delta * eps0)) break;
        }
    }
    if (!GCOL && active)
        for (int k = q; k < r; k += LPC) V[(int64_t)k * ldv + col0] = mycol[k * CW];
    if (blockIdx.x == 0 && threadIdx.x == 0 && status) {
        if (MODE == 0 && max_sweeps >= 1) {
            status[NNF_HALS_ST_EPS] = eps;
            status[NNF_HALS_ST_CNT] = (double)(sweep0 + done + 1);
            status[NNF_HALS_ST_EPS0] = eps0;
        }
        if (!ok) status[NNF_HALS_ST_ERR] = 1.0;
        else if (err) status[NNF_HALS_ST_ERR] = (double)err;
    }
}

__global__ void nnf_hals_hadamard_kernel(const float* __restrict__ A, const float* __restrict__ B, int64_t ld, int r,
                                        float* __restrict__ out) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < r * r) out[e] = A[(int64_t)(e / r) * ld + (e % r)] * B[(int64_t)(e / r) * ld + (e % r)];
}

// out[s] = sum_b partials[s][b]  (index order)
__global__ __launch_bounds__(256) void nnf_hals_sum_sweeps_kernel(const double* __restrict__ partials, int nblocks,
                                                                  double* __restrict__ out) {
    __shared__ double red[4];
    const double* p = partials + (size_t)blockIdx.x * nblocks;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) s += p[b];
    const double t = nnf_block_sum_f64(s, red);
    if (threadIdx.x == 0) out[blockIdx.x] = t;
}

// ---------------------------------------------------------------------------------------------------------
static int pick_rp(int r) {
    static const int opts[] = {8, 16, 24, 32, 40, 48, 50, 52, 56, 64, 80, 96, 100, 104, 112, 128};   // 100: config E's rank
    for (int o : opts)
        if (r <= o) return o;
    return (r + 7) & ~7;      // above NNF_MAX_RANK: the generic kernel only (its padded Gram has one row per 8)
}

// workgroups of 256 columns the register-resident lane kernel of padded rank RP keeps on the chip
static int64_t lane_resident_blocks(nnf_ctx* ctx, int RP) {
    hals_args q{};
    q.ncols = -1;
    int nb = 0, rc;
    if (RP <= 48) rc = nnf_hals_fast_part0(ctx, RP, q, NNF_HALS_MAX_BLOCKS, &nb, nullptr);
    else if (RP <= 64) rc = nnf_hals_fast_part1(ctx, RP, q, NNF_HALS_MAX_BLOCKS, &nb, nullptr);
    else if (RP <= 104) rc = nnf_hals_fast_part2(ctx, RP, q, NNF_HALS_MAX_BLOCKS, &nb, nullptr);
    else rc = nnf_hals_fast_part3(ctx, RP, q, NNF_HALS_MAX_BLOCKS, &nb, nullptr);
    return rc == NNF_OK ? nb : 0;
}

// Many columns at ranks 64..100: the push form on the matrix cores (k_hals_mfma.hip); below rank 64 a k-block has too few
// MFMAs to cover its own gather -> update -> scatter chain (measured: 9.7-10.6 against 9.4-9.6 us per sweep at rank 50).
static bool hals_mfma_default(int RP, int64_t ncols) {
    const char* force = getenv("NNF_HALS_FORCE");
    if (force && (force[0] == 'l' || force[0] == 'q' || force[0] == 'w')) return false;
    if (!nnf_hals_mfma_supported(RP)) return false;
    if (force && force[0] == 'm') return true;
    return RP >= 64 && ncols > 32768;
}

template <int MODE>
static int hals_entry(nnf_ctx* ctx, const float* UtM, int64_t ldm, const float* UtU, int64_t ldg, float* V, int64_t ldv,
                      int r, int64_t ncols, int nsweeps, double delta, float sparsity, unsigned flags, double* status,
                      double* nodelta_out, hipStream_t st, float* snapshots = nullptr, int64_t snap_stride = 0,
                      int sweep0 = 0, const float* UtU2 = nullptr, const float* Vsrc = nullptr, int64_t ldvs = 0,
                      int snap_first = 0, const float* resid_in = nullptr, float* resid_out = nullptr) {
    if (!ctx || !UtM || !UtU || !V || r < 1 || ncols < 1 || ldm < ncols || ldv < ncols || ldg < r || nsweeps < 0)
        return NNF_ERR_ARG;
    if (MODE == 0 && !status) return NNF_ERR_ARG;
    if (MODE == 1 && !nodelta_out && nsweeps > 0) return NNF_ERR_ARG;
    if (nsweeps > NNF_HALS_MAX_SWEEPS) return NNF_ERR_UNSUPPORTED;   // (longer solves: chained by the caller)
    const bool big_rank = r > NNF_MAX_RANK;   // the generic kernel on columns in global memory (no snapshots, no residual state)

    if (flags & ~(NNF_HALS_SPARSITY | NNF_HALS_NORMALIZE | NNF_HALS_NONZERO)) return NNF_ERR_ARG;
    const int RP = pick_rp(r);
    const float sp = (flags & NNF_HALS_SPARSITY) ? sparsity : 0.f;
    const int max_blocks = NNF_HALS_MAX_BLOCKS;
    nnf_ws_cursor cur(ctx);
    const int RS = 32 * ((RP + 31) / 32);
    const bool generic = big_rank || (flags & (NNF_HALS_NORMALIZE | NNF_HALS_NONZERO)) != 0;
    // few columns: four lanes per column (k_hals_quad.hip); many: one lane per column (k_hals_fast.hip)
    // NNF_HALS_FORCE=lane|quad pins the column layout (tests exercise both kernels on the same fixtures)
    const char* force = getenv("NNF_HALS_FORCE");
    const bool force_lane = force && force[0] == 'l', force_quad = force && force[0] == 'q', force_wave = force && force[0] == 'w';
    const bool force_mfma = force && force[0] == 'm';
    // fewer still (<= 4800, a persistent solve from its first sweep): one wave per column, push form (k_hals_wave.hip)
    const bool want_mfma = !generic && nsweeps > 0 && hals_mfma_default(RP, ncols);
    const bool wave = !generic && MODE == 0 && sweep0 == 0 && !force_lane && !force_quad && !(force_mfma && want_mfma) && nsweeps <= NNF_HALS_MAX_SWEEPS &&
                      nnf_hals_wave_fits(ctx, r, ncols, max_blocks);
    // NNF_HALS_FORCE=wave: a solve this layout could take but does not fit is refused instead of moving to another layout
    if (force_wave && !generic && MODE == 0 && sweep0 == 0 && nsweeps <= NNF_HALS_MAX_SWEEPS && !wave) return NNF_ERR_UNSUPPORTED;
    const bool quad = !wave && !generic && !force_lane && !(force_mfma && want_mfma) && (ncols <= 32768 || force_quad) && (int64_t)(r + 16) * (ldv > ldm ? ldv : ldm) * 4 < (int64_t)0x7fff0000 &&
                      nnf_hals_quad_fits(ctx, r, ncols, max_blocks);
    if (getenv("NNF_HALS_DEBUG"))
        fprintf(stderr, "[nnf hals] r=%d ncols=%lld mode=%d sweeps=%d layout=%s\n", r, (long long)ncols, MODE, nsweeps,
                wave ? "wave" : quad ? "quad" : generic ? "generic" : "lane");
    const size_t gs_off = (((size_t)RP * RS + 2 * RP + 1) + 15) & ~(size_t)15;   // scaled image, 64-byte aligned
    const bool want_gs = !generic && RP > 32 && RP <= 52;
    size_t gfloats = gs_off + (want_gs ? (size_t)RP * RS : 0);
    if (quad && nnf_hals_quad_gram_floats(r) > gfloats) gfloats = nnf_hals_quad_gram_floats(r);
    if (wave && nnf_hals_wave_gram_floats(r) > gfloats) gfloats = nnf_hals_wave_gram_floats(r);
    // many columns, ranks 48..100: the push form on the matrix cores (k_hals_mfma.hip) when every column stays resident
    const bool try_mfma = want_mfma && !quad && !wave;
    float* Gm = try_mfma ? (float*)cur.take(nnf_hals_mfma_gram_floats(RP) * 4) : nullptr;
    if (try_mfma && !Gm) return NNF_ERR_WORKSPACE;
    float* Gp = (float*)cur.take(gfloats * 4);   // padded Gram, then the (1/diag, nz) pairs (quad: scaled Gram, 1/diag)
    float* dinv = Gp ? Gp + (size_t)RP * RS : nullptr;
    unsigned* counter = (unsigned*)cur.take(256);
    double* slots = (double*)cur.take((size_t)2 * max_blocks * 4 * 8);
    // tagged granules of the fast paths: the context's dedicated region (never shared with another kernel's scratch)
    double* sslots = (MODE == 0) ? (double*)ctx->xch : slots;
    if (MODE == 0 && (size_t)(nsweeps + 2) * max_blocks * 16 > ctx->xch_bytes) return NNF_ERR_WORKSPACE;
    double* sweep_partials = nullptr;
    if (MODE == 1) sweep_partials = (double*)cur.take((size_t)(nsweeps > 0 ? nsweeps : 1) * max_blocks * 8);
    if (!Gp || !dinv || !counter || !slots || !sslots || (MODE == 1 && !sweep_partials))
        return NNF_ERR_WORKSPACE;
    if (Vsrc == nullptr || Vsrc == V) { Vsrc = V; ldvs = ldv; }
    if (!quad && !wave && (UtU2 != nullptr || Vsrc != V)) {
        // the Hadamard Gram and the separate start values are native to the few-column (quad) kernel -- the shape they were
        // made for (NTF factors); the other layouts get them from two small element-wise launches
        if (UtU2 != nullptr) {
            float* Gh = (float*)cur.take((size_t)r * r * 4);
            if (!Gh) return NNF_ERR_WORKSPACE;
            hipLaunchKernelGGL(nnf_hals_hadamard_kernel, dim3((r * r + 255) / 256), dim3(256), 0, st, UtU, UtU2, ldg, r, Gh);
            NNF_CHECK_LAUNCH();
            UtU = Gh;
            ldg = r;
            UtU2 = nullptr;
        }
        // separate start values: the resident lane kernel reads them itself (once); the generic and the streaming kernels
        // work in place on a copy
        const bool lane_resident = !generic && nsweeps > 0 && RP > 0 && nnf_cdiv(ncols, 256) <= lane_resident_blocks(ctx, RP) &&
                                   (((int64_t)(r - 1) * ldvs + ncols) * 4) < (int64_t)0x7fff0000;
        if (Vsrc != V && !lane_resident) {
            if (hipMemcpy2DAsync(V, (size_t)ldv * 4, Vsrc, (size_t)ldvs * 4, (size_t)ncols * 4, (size_t)r, hipMemcpyDeviceToDevice,
                                 st) != hipSuccess)
                return NNF_ERR_LAUNCH;
            Vsrc = V;
            ldvs = ldv;
        }
    }
    if (!quad && !wave) {
        hipLaunchKernelGGL(nnf_hals_prep_kernel, dim3(RP), dim3(128), 0, st, UtU, ldg, r, RP, Gp, dinv,
                           want_gs ? Gp + gs_off : (float*)nullptr, counter,
                           (MODE == 0 && sweep0 == 0) ? status : (double*)nullptr);
        NNF_CHECK_LAUNCH();
        if (nsweeps == 0) return NNF_OK;
    }
    ctx->hals_epoch = (ctx->hals_epoch + 1u) & 0x3fffffu;   // tag = epoch*1024 + sweep stays below 2^32
    if (ctx->hals_epoch == 0u) {   // wrapped (2^22 solves): clear the region so that tags of the previous round cannot match
        if (hipMemsetAsync(ctx->xch, 0, ctx->xch_bytes, st) != hipSuccess) return NNF_ERR_LAUNCH;
        ctx->hals_epoch = 1u;
    }
    hals_sync sy{counter, slots, sslots, ctx->hals_epoch};
    int nblocks = 0, rc = NNF_OK;
    if (wave) {
        hals_args a{UtM, ldm, nullptr, nullptr, nullptr, V, ldv, r, ncols, nsweeps, delta, sp, MODE, sy, status, sweep_partials,
                    snapshots, snap_stride, sweep0, Vsrc, ldvs};
        float* snap = (float*)cur.take(nnf_hals_wave_snap_floats(r, ncols) * 4);
        if (!snap) return NNF_ERR_WORKSPACE;
        rc = nnf_hals_wave_run(ctx, UtU, UtU2, ldg, Gp, snap, counter, a, &nblocks, st);
        if (rc != NNF_OK) return rc;
        if (nsweeps == 0) return NNF_OK;
    } else if (quad) {
        if ((((int64_t)(r - 1) * ldv + ncols) * 4) >= (int64_t)0x7fff0000 || (((int64_t)(r - 1) * ldm + ncols) * 4) >= (int64_t)0x7fff0000)
            return NNF_ERR_UNSUPPORTED;   // 32-bit buffer offsets
        hals_args a{UtM, ldm, nullptr, nullptr, nullptr, V, ldv, r, ncols, nsweeps, delta, sp, MODE, sy, status, sweep_partials,
                    snapshots, snap_stride, sweep0, Vsrc, ldvs};
        a.snap_first = snap_first;
        rc = nnf_hals_quad_run(ctx, UtU, UtU2, ldg, Gp, counter, a, &nblocks, st);
        if (rc != NNF_OK) return rc;
        if (nsweeps == 0) return NNF_OK;
    } else if (generic) {
        // one column per thread, all workgroups resident (row-level grid reductions)
        // blind sweeps without row-level reductions exchange nothing: no residency needed (any number of columns)
        const bool exchanges = MODE == 0 || (flags & (NNF_HALS_NORMALIZE | NNF_HALS_NONZERO)) != 0;
        // the column in LDS (r x 128 floats per workgroup) while that fits and -- when the workgroups exchange -- all of them are
        // resident with it; else (ranks above ~300, or more columns than one LDS-bound workgroup per CU holds) the column stays
        // in global memory (GCOL).  Measured at rank 200 (tools/probes/bigrank_sweep_probe.py): LDS 3-5x faster per sweep.
        static const int force_gcol = [] { const char* e = getenv("NNF_HALS_GCOL"); return e ? atoi(e) : 0; }();   // A/B knob
        const int cw_lds = big_rank ? 32 : 128;            // columns per workgroup with the column in LDS (four lanes per column above rank 128)
        const size_t shm_lds = (size_t)r * cw_lds * 4 + 16 + 3 * 2 * 8 + 64, shm_g = 16 + 3 * 2 * 8 + 64;
        bool gcol = big_rank && (force_gcol || shm_lds > (size_t)150 * 1024);
        int64_t grid = nnf_cdiv(ncols, cw_lds);
        int nb = 0;
        hipError_t he = hipSuccess;
        if (!gcol && big_rank) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(&nnf_hals_generic_kernel<MODE, false, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_lds);
            he = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, nnf_hals_generic_kernel<MODE, false, true>, 128, shm_lds);
        } else if (!gcol) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(&nnf_hals_generic_kernel<MODE, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_lds);
            he = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, nnf_hals_generic_kernel<MODE, false>, 128, shm_lds);
            if (big_rank && (he != hipSuccess || nb < 1 ||
                             (exchanges && (grid > max_blocks || grid > (int64_t)(nb >= 3 ? nb - 1 : nb) * ctx->num_cus)))) gcol = true;
        }
        if (gcol) {
            grid = nnf_cdiv(ncols, 128);
            he = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, nnf_hals_generic_kernel<MODE, true>, 128, shm_g);
        }
        if (he != hipSuccess || nb < 1) return NNF_ERR_LAUNCH;
        const size_t shm = gcol ? shm_g : shm_lds;
        int bpc = nb >= 3 ? nb - 1 : nb;
        if (bpc > 4 && !(big_rank && !gcol)) bpc = 4;
        if ((exchanges && grid > (int64_t)bpc * ctx->num_cus) || grid > (exchanges ? (int64_t)max_blocks : (int64_t)0x7fffffff))
            return NNF_ERR_UNSUPPORTED;
        if (!exchanges && MODE == 1 && grid > max_blocks) {   // the per-sweep partial sums: one double per workgroup and sweep
            sweep_partials = (double*)cur.take((size_t)(nsweeps > 0 ? nsweeps : 1) * (size_t)grid * 8);
            if (!sweep_partials) return NNF_ERR_WORKSPACE;
        }
        nblocks = (int)grid;
        if (getenv("NNF_HALS_DEBUG")) fprintf(stderr, "[nnf hals] generic: %s, %d workgroups, %d per CU\n", gcol ? "columns in global memory" : "columns in LDS", nblocks, bpc);
        if (gcol)
            hipLaunchKernelGGL((nnf_hals_generic_kernel<MODE, true>), dim3(nblocks), dim3(128), shm, st, UtM, ldm, Gp, dinv, RS, V,
                               ldv, r, ncols, nsweeps, delta, sp, flags, sy, status, sweep_partials, sweep0, snapshots, snap_stride, snap_first);
        else if (big_rank)
            hipLaunchKernelGGL((nnf_hals_generic_kernel<MODE, false, true>), dim3(nblocks), dim3(128), shm, st, UtM, ldm, Gp, dinv, RS, V,
                               ldv, r, ncols, nsweeps, delta, sp, flags, sy, status, sweep_partials, sweep0, snapshots, snap_stride, snap_first);
        else
            hipLaunchKernelGGL((nnf_hals_generic_kernel<MODE, false>), dim3(nblocks), dim3(128), shm, st, UtM, ldm, Gp, dinv, RS, V,
                               ldv, r, ncols, nsweeps, delta, sp, flags, sy, status, sweep_partials, sweep0, snapshots, snap_stride, snap_first);
        NNF_CHECK_LAUNCH();
    } else {
        if ((((int64_t)(r - 1) * ldv + ncols) * 4) >= (int64_t)0x7fff0000 || (((int64_t)(r - 1) * ldm + ncols) * 4) >= (int64_t)0x7fff0000)
            return NNF_ERR_UNSUPPORTED;   // 32-bit buffer offsets
        hals_args a{UtM, ldm, Gp, dinv, want_gs ? Gp + gs_off : nullptr, V, ldv, r, ncols, nsweeps, delta, sp, MODE, sy, status,
                    sweep_partials, snapshots, snap_stride, sweep0, Vsrc, ldvs};
        a.snap_first = snap_first;
        a.resid_in = resid_in;
        a.resid_out = resid_out;
        nnf_probe(ctx, NNF_PROBE_HALS, 0, st);
        rc = NNF_ERR_UNSUPPORTED;
        if (try_mfma) rc = nnf_hals_mfma_run(ctx, RP, UtU, ldg, Gm, a, max_blocks, &nblocks, st);
        if (rc == NNF_OK) {
            if (getenv("NNF_HALS_DEBUG")) fprintf(stderr, "[nnf hals] -> mfma kernel, %d workgroups\n", nblocks);
        } else if (rc != NNF_ERR_UNSUPPORTED) {
            return rc;
        } else
        if (RP <= 48) rc = nnf_hals_fast_part0(ctx, RP, a, max_blocks, &nblocks, st);
        else if (RP <= 64) rc = nnf_hals_fast_part1(ctx, RP, a, max_blocks, &nblocks, st);
        else if (RP <= 104) rc = nnf_hals_fast_part2(ctx, RP, a, max_blocks, &nblocks, st);
        else rc = nnf_hals_fast_part3(ctx, RP, a, max_blocks, &nblocks, st);
        if (rc != NNF_OK) return rc;
        nnf_probe(ctx, NNF_PROBE_HALS, 1, st);
    }
    if (MODE == 1) {
        hipLaunchKernelGGL(nnf_hals_sum_sweeps_kernel, dim3(nsweeps), dim3(256), 0, st, sweep_partials, nblocks,
                           nodelta_out);
        NNF_CHECK_LAUNCH();
    }
    return NNF_OK;
}

extern "C" int nnf_hals_solve_f32(nnf_ctx* ctx, const float* UtM, int64_t ldm, const float* UtU, int64_t ldg, float* V,
                                  int64_t ldv, int r, int64_t ncols, int max_sweeps, double delta, float sparsity,
                                  unsigned flags, double* status_f64, void* stream) {
    return hals_entry<0>(ctx, UtM, ldm, UtU, ldg, V, ldv, r, ncols, max_sweeps, delta, sparsity, flags, status_f64,
                         nullptr, (hipStream_t)stream);
}

extern "C" int nnf_hals_sweeps_f32(nnf_ctx* ctx, const float* UtM, int64_t ldm, const float* UtU, int64_t ldg, float* V,
                                   int64_t ldv, int r, int64_t ncols, int nsweeps, float sparsity, unsigned flags,
                                   double* nodelta_f64, float* snapshots, int64_t snap_stride, void* stream) {
    if (snapshots) {
        // snapshots are written by the resident fast path only
        if ((flags & (NNF_HALS_NORMALIZE | NNF_HALS_NONZERO)) || snap_stride < (int64_t)r * ncols) return NNF_ERR_ARG;
        // (whether the columns fit the resident kernel is decided where the kernel is picked: launch_rp / the quad path)
    }
    return hals_entry<1>(ctx, UtM, ldm, UtU, ldg, V, ldv, r, ncols, nsweeps, 0.0, sparsity, flags, nullptr, nodelta_f64,
                         (hipStream_t)stream, snapshots, snap_stride);
}

// Blind sweeps that CONTINUE a solve (the chunks of the row-sharded protocol, dist.py): `sweeps_done` sweeps of this solve
// have run already in earlier calls.  The matrix-core kernel (k_hals_mfma.hip) keeps a per-column residual next to V; handed
// from call to call through resid_in / resid_out (nnf_hals_resid_floats() floats each, opaque layout; NULL: the launch forms
// its residual from scratch) the chunks of a solve give bit for bit what one launch of all the sweeps gives.  The other
// layouts carry no state and ignore the three arguments.  snap_first: the first sweep of this call (0-based) that writes a
// snapshot -- a chunk = `head` blind sweeps + a window of snapshots is ONE launch; block j holds V after sweep snap_first + j + 1.
extern "C" int nnf_hals_sweeps_ex_f32(nnf_ctx* ctx, const float* UtM, int64_t ldm, const float* UtU, int64_t ldg, float* V,
                                      int64_t ldv, int r, int64_t ncols, int nsweeps, int sweeps_done, float sparsity,
                                      unsigned flags, double* nodelta_f64, float* snapshots, int64_t snap_stride, int snap_first,
                                      const float* resid_in, float* resid_out, void* stream) {
    if (sweeps_done < 0 || snap_first < 0 || (snapshots && snap_first >= nsweeps && nsweeps > 0)) return NNF_ERR_ARG;
    if (snapshots && ((flags & (NNF_HALS_NORMALIZE | NNF_HALS_NONZERO)) || snap_stride < (int64_t)r * ncols)) return NNF_ERR_ARG;
    return hals_entry<1>(ctx, UtM, ldm, UtU, ldg, V, ldv, r, ncols, nsweeps, 0.0, sparsity, flags, nullptr, nodelta_f64,
                         (hipStream_t)stream, snapshots, snap_stride, sweeps_done, nullptr, nullptr, 0, snap_first, resid_in,
                         resid_out);
}

// floats of residual state per buffer for nnf_hals_sweeps_ex_f32 on an r x ncols factor (0: the layout that runs carries none)
extern "C" int nnf_hals_resid_floats(nnf_ctx* ctx, int r, int64_t ncols, int64_t* floats_out) {
    if (!ctx || r < 1 || ncols < 1 || !floats_out) return NNF_ERR_ARG;
    const int RP = pick_rp(r);
    *floats_out = (RP > 0 && hals_mfma_default(RP, ncols)) ? (int64_t)nnf_hals_mfma_resid_floats(RP, ncols) : 0;
    return NNF_OK;
}

// Solves longer than one launch can tag (max_sweeps > 1000, e.g. hals_nnls_acc(maxiter=5000)): the caller chains launches of
// at most 1000 sweeps; every launch after the first is this entry with sweeps_done = the budget already spent.  The launch
// reads the status block its predecessor left (same pointer): if that one already ended the solve (stopping rule of
// nnls.py:156, or an error) it returns at once and leaves V and the block untouched, else it carries eps0 and the count on.
// Nothing is read back by the host between the launches.
extern "C" int nnf_hals_solve_continue_f32(nnf_ctx* ctx, const float* UtM, int64_t ldm, const float* UtU, int64_t ldg,
                                           float* V, int64_t ldv, int r, int64_t ncols, int sweeps_done, int max_sweeps,
                                           double delta, float sparsity, unsigned flags, double* status_f64, void* stream) {
    if (sweeps_done < 1 || max_sweeps < 1) return NNF_ERR_ARG;
    return hals_entry<0>(ctx, UtM, ldm, UtU, ldg, V, ldv, r, ncols, max_sweeps, delta, sparsity, flags, status_f64, nullptr,
                         (hipStream_t)stream, nullptr, 0, sweeps_done);
}

// ---------------------------------------------------------------------------------------------------------
// Row-sharded solve without a host round trip (the stopping scalar of nnls.py:156 is global, the sweeps are
// local).  A rank runs `nsweeps` blind sweeps (nnf_hals_sweeps_f32; the last nsweeps - head of them leave snapshots), the
// per-sweep sums are all-reduced, and this kernel -- every workgroup redundantly, from the same doubles -- replays the
// reference's loop condition over them:   stop = first s with  !(sum[s] >= delta * sum[0])  or  s + 1 == budget.
//   stop inside the snapshot window  -> V := snapshot of that sweep (unless it is the last one run), status = {eps, cnt, eps0, 0}
//   stop before the window           -> status error 3 (V holds too many sweeps and no snapshot of the right one)
//   no stop within these sweeps      -> status error 4 (more sweeps needed)
// The host looks at the status block one or two outer iterations later (the pipelined loop of nmf.py); 3 / 4 make it redo
// that iteration with the synchronous chunked protocol (dist.sharded_hals_solve), which also re-centres the sweep guess.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nnf_hals_stop_restore_kernel(const double* __restrict__ sums, int nsweeps, int head,
                                                                   int budget, double delta, float* __restrict__ V, int64_t ldv,
                                                                   int r, int64_t ncols, const float* __restrict__ snapshots,
                                                                   int64_t snap_stride, double* __restrict__ status) {
    __shared__ int s_stop;
    if (threadIdx.x == 0) {
        int stop = -1;
        const double eps0 = sums[0];
        for (int s = 0; s < nsweeps; ++s)
            if (!(sums[s] >= delta * eps0) || s + 1 >= budget) { stop = s; break; }
        s_stop = stop;
        if (blockIdx.x == 0) {
            status[NNF_HALS_ST_EPS0] = eps0;
            if (stop < 0) {
                status[NNF_HALS_ST_EPS] = sums[nsweeps - 1];
                status[NNF_HALS_ST_CNT] = (double)(nsweeps + 1);
                status[NNF_HALS_ST_ERR] = 4.0;
            } else {
                status[NNF_HALS_ST_EPS] = sums[stop];
                status[NNF_HALS_ST_CNT] = (double)(stop + 2);
                status[NNF_HALS_ST_ERR] = (stop < head) ? 3.0 : 0.0;
            }
        }
    }
    __syncthreads();
    const int stop = s_stop;
    if (stop < head || stop >= nsweeps - 1) return;     // nothing to restore (error, or the last sweep run is the right one)
    const float* src = snapshots + (int64_t)(stop - head) * snap_stride;
    const int64_t total = (int64_t)r * ncols;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t k = e / ncols, j = e - k * ncols;
        V[k * ldv + j] = src[k * ncols + j];
    }
}

// Columns the register-resident sweep kernel of rank r holds on this device (every column = one lane, its V and right-hand
// side entries in registers, all workgroups co-resident).  Beyond it nnf_hals_solve_f32 / nnf_hals_sweeps_f32 stream the
// factor through HBM once per sweep and nnf_hals_sweeps_f32 writes no snapshots: callers that run blind chunks of sweeps
// (the row-sharded protocol, the solve of a 10^6-column factor on one device) split the columns into blocks of this size.
extern "C" int nnf_hals_resident_columns(nnf_ctx* ctx, int r, int64_t* columns_out) {
    if (!ctx || r < 1 || !columns_out) return NNF_ERR_ARG;
    const int RP = pick_rp(r);
    if (r > NNF_MAX_RANK) {   // the generic kernel: four 128-column workgroups per CU
        *columns_out = (int64_t)4 * ctx->num_cus * 128;
        return NNF_OK;
    }
    hals_args a{};
    a.ncols = -1;
    int nblocks = 0, rc;
    if (RP <= 48) rc = nnf_hals_fast_part0(ctx, RP, a, NNF_HALS_MAX_BLOCKS, &nblocks, nullptr);
    else if (RP <= 64) rc = nnf_hals_fast_part1(ctx, RP, a, NNF_HALS_MAX_BLOCKS, &nblocks, nullptr);
    else if (RP <= 104) rc = nnf_hals_fast_part2(ctx, RP, a, NNF_HALS_MAX_BLOCKS, &nblocks, nullptr);
    else rc = nnf_hals_fast_part3(ctx, RP, a, NNF_HALS_MAX_BLOCKS, &nblocks, nullptr);
    if (rc != NNF_OK) return rc;
    *columns_out = (int64_t)nblocks * 256;
    return NNF_OK;
}

extern "C" int nnf_hals_stop_restore_f32(nnf_ctx* ctx, const double* sums_f64, int nsweeps, int head, int budget, double delta,
                                         float* V, int64_t ldv, int r, int64_t ncols, const float* snapshots,
                                         int64_t snap_stride, double* status_f64, void* stream) {
    if (!ctx || !sums_f64 || !V || !status_f64 || nsweeps < 1 || head < 0 || head >= nsweeps || budget < 1 || r < 1 ||
        ncols < 1 || ldv < ncols)
        return NNF_ERR_ARG;
    if (nsweeps - head > 1 && (!snapshots || snap_stride < (int64_t)r * ncols)) return NNF_ERR_ARG;
    int64_t grid = nnf_cdiv((int64_t)r * ncols, 256 * 8);
    if (grid > 2048) grid = 2048;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(nnf_hals_stop_restore_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, sums_f64, nsweeps, head,
                       budget, delta, V, ldv, r, ncols, snapshots, snap_stride, status_f64);
    NNF_CHECK_LAUNCH();
    return NNF_OK;
}

// hals_nnls_acc as one_ntf_step calls it (ntf.py:442-456): the Gram is the Hadamard product of two factor Grams
// (`cross`), the start values are the current factor and the result is a NEW factor.  Same solve as nnf_hals_solve_f32
// with UtU := UtU_a .* UtU_b (UtU_b may be NULL) and V_out := V_in before the first sweep -- without the Hadamard launch and
// the copy in front of it (the few-column kernel forms the product while it stages the Gram and reads its start values
// from V_in; the other layouts do both with small launches of their own).  V_in may equal V_out.
extern "C" int nnf_hals_solve_cross_f32(nnf_ctx* ctx, const float* UtM, int64_t ldm, const float* UtU_a, const float* UtU_b,
                                        int64_t ldg, const float* V_in, int64_t ldvi, float* V_out, int64_t ldvo, int r,
                                        int64_t ncols, int max_sweeps, double delta, float sparsity, unsigned flags,
                                        double* status_f64, void* stream) {
    if (!V_in || ldvi < ncols) return NNF_ERR_ARG;
    if (max_sweeps > NNF_HALS_MAX_SWEEPS) return NNF_ERR_UNSUPPORTED;
    return hals_entry<0>(ctx, UtM, ldm, UtU_a, ldg, V_out, ldvo, r, ncols, max_sweeps, delta, sparsity, flags, status_f64,
                         nullptr, (hipStream_t)stream, nullptr, 0, 0, UtU_b, V_in, ldvi);
}

// ---------------------------------------------------------------------------------------------------------
// Row-sharded solves that NORMALISE the sharded factor (nmf(normalize=[True, .]) over several ranks): the row
// norm of nnls.py:179-185 runs over the columns of ALL ranks, once per row update, so the sweep cannot stay inside one launch.
// The host walks the rows (nn_fac_amd/dist.py: sharded_hals_solve_rownorm): this launch updates row k on the local columns and
// leaves {sum of squared steps, sum of squares of the updated row}; the two are all-reduced together; the second launch scales
// the row.  r collectives per sweep: correct and slow -- the option is not on any BASELINE configuration.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nnf_hals_row_update_kernel(const float* __restrict__ UtM, int64_t ldm, const float* __restrict__ UtU,
                                                                  int64_t ldg, float* __restrict__ V, int64_t ldv, int r, int64_t ncols,
                                                                  int k, float sp, double* __restrict__ partial) {
    __shared__ double red[4];
    const float d = UtU[(int64_t)k * ldg + k];
    const float di = (d != 0.f) ? (float)(1.0 / (double)d) : 0.f;
    double nd = 0.0, nv = 0.0;
    for (int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x; col < ncols; col += (int64_t)gridDim.x * 256) {
        float vk = V[(int64_t)k * ldv + col];
        if (di != 0.f) {                                  // nnls.py:160: a row with a zero Gram diagonal is left alone
            float dot = 0.f;
            for (int i = 0; i < r; ++i) dot = fmaf(UtU[(int64_t)k * ldg + i], V[(int64_t)i * ldv + col], dot);
            const float step = fmaxf((UtM[(int64_t)k * ldm + col] - dot - sp) * di, -vk);
            vk += step;
            V[(int64_t)k * ldv + col] = vk;
            nd += (double)step * (double)step;
        }
        nv += (double)vk * (double)vk;
    }
    const double bd = nnf_block_sum_f64(nd, red), bv = nnf_block_sum_f64(nv, red);
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = bd;
        partial[gridDim.x + blockIdx.x] = bv;
    }
}
__global__ __launch_bounds__(256) void nnf_hals_row_sums_kernel(const double* __restrict__ partial, int nwg, double* __restrict__ out2) {
    __shared__ double red[4];
    for (int q = 0; q < 2; ++q) {
        double s = 0.0;
        for (int e = threadIdx.x; e < nwg; e += 256) s += partial[(size_t)q * nwg + e];
        const double t = nnf_block_sum_f64(s, red);
        if (threadIdx.x == 0) out2[q] = t;
    }
}
__global__ __launch_bounds__(256) void nnf_hals_row_scale_kernel(float* __restrict__ V, int64_t ncols, const double* __restrict__ normsq,
                                                                 double fill) {
    const double nsq = normsq[0];
    for (int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x; col < ncols; col += (int64_t)gridDim.x * 256)
        V[col] = (nsq != 0.0) ? (float)((double)V[col] / sqrt(nsq)) : (float)fill;      // nnls.py:181-185
}
extern "C" int nnf_hals_row_update_f32(nnf_ctx* ctx, const float* UtM, int64_t ldm, const float* UtU, int64_t ldg, float* V, int64_t ldv,
                                       int r, int64_t ncols, int k, float sparsity, unsigned flags, double* out2_f64, void* stream) {
    if (!ctx || !UtM || !UtU || !V || !out2_f64 || r < 1 || ncols < 1 || k < 0 || k >= r || ldm < ncols || ldv < ncols || ldg < r)
        return NNF_ERR_ARG;
    if (flags & ~NNF_HALS_SPARSITY) return NNF_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    int64_t grid = nnf_cdiv(ncols, 256);
    if (grid > 2048) grid = 2048;
    nnf_ws_cursor cur(ctx);
    double* partial = (double*)cur.take((size_t)2 * grid * 8);
    if (!partial) return NNF_ERR_WORKSPACE;
    hipLaunchKernelGGL(nnf_hals_row_update_kernel, dim3((int)grid), dim3(256), 0, st, UtM, ldm, UtU, ldg, V, ldv, r, ncols, k,
                       (flags & NNF_HALS_SPARSITY) ? sparsity : 0.f, partial);
    NNF_CHECK_LAUNCH();
    hipLaunchKernelGGL(nnf_hals_row_sums_kernel, dim3(1), dim3(256), 0, st, partial, (int)grid, out2_f64);
    NNF_CHECK_LAUNCH();
    return NNF_OK;
}
extern "C" int nnf_hals_row_scale_f32(nnf_ctx* ctx, float* V, int64_t ldv, int64_t ncols, int k, const double* normsq_f64,
                                      int64_t ncols_total, void* stream) {
    if (!ctx || !V || !normsq_f64 || ncols < 1 || k < 0 || ldv < ncols || ncols_total < ncols) return NNF_ERR_ARG;
    int64_t grid = nnf_cdiv(ncols, 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(nnf_hals_row_scale_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, V + (int64_t)k * ldv, ncols,
                       normsq_f64, 1.0 / sqrt((double)ncols_total));
    NNF_CHECK_LAUNCH();
    return NNF_OK;
}
