// MTTKRP of a dense 3-way tensor T[I x J x K] (C order, k fastest) with the Khatri-Rao product of the two other factors
// generated on the fly -- replaces khatri_rao + np.dot(unfolded[mode], krao) of nn_fac/ntf.py:448-449.
//
// The reference materialises all three unfoldings (ntf.py:309-311) and the (250000 x R) Khatri-Rao matrix.  Here the
// tensor is read in place, ONCE per mode, and a Khatri-Rao row is two factor entries multiplied while the small
// operand is staged into LDS:
//   mode 0:  out[r][i] = sum_{j,k} T[i][j][k] F1t[r][j] F2t[r][k]     rows i (ld J*K), segments j (offset j*K), inner k
//   mode 1:  out[r][j] = sum_{i,k} T[i][j][k] F0t[r][i] F2t[r][k]     rows j (ld K),   segments i (offset i*J*K), inner k
//            -> both are "V X^T"-shaped (reduction along the contiguous axis): one segmented kernel, split over segments
//   mode 2:  out[r][k] = sum_{i,j} T[i][j][k] F0t[r][i] F1t[r][j]     T seen as an (I*J) x K matrix, reduction over rows
//            -> "W^T X"-shaped, split over rows
// Same MFMA / buffer-load / fragment-order machinery as k_stream.hip; partial slabs are summed in fp64 in a fixed order.
// unfold/khatri_rao index conventions follow tensorly 0.6.0 (first remaining mode slowest), see SURVEY.md appendix B.
#include "k_stream_common.h"
#ifndef SEG_ABL
#define SEG_ABL 0      // the same for nnf_mttkrp_seg_kernel (tools/mttkrp_ablate.sh seg)
#endif
#ifndef MTTKRP_ABL
#define MTTKRP_ABL 0   // timing-only ablations of nnf_mttkrp_rows_kernel (tools/mttkrp_ablate.sh); 0 = the product
#endif
NNF_BUILD_FLAGS(k_mttkrp, "SEG_ABL=" NNF_STR(SEG_ABL) " MTTKRP_ABL=" NNF_STR(MTTKRP_ABL))

// ---------------------------------------------------------------------------------------------------------
// segmented V X^T:  out[rk][row] = sum_{s in split} Fs[rk][s] * sum_k Fk[rk][k] * T[row*ldrow + s*segstride + k]
//   grid = row blocks (256 rows) x segment splits; slab[split][rk][row]
// ---------------------------------------------------------------------------------------------------------
template <int MT, bool VEC>
__global__ __launch_bounds__(256, (MT <= 4 ? 2 : 1)) void nnf_mttkrp_seg_kernel(
    const float* __restrict__ T, int64_t nrows, int64_t ldrow, int64_t nseg, int64_t segstride, int64_t klen,
    const float* __restrict__ Fs, int64_t lds_, const float* __restrict__ Fk, int64_t ldk, int r,
    float* __restrict__ slabs, int64_t ldp, int nrb, int nsplit, int64_t seg_per_split, int fk_vec_ok) {
    __shared__ f32x4 ldsA[2][MT * 256];
    int sp, rb;
    nnf_xcd_map(blockIdx.x, nrb, sp, rb);
    if (sp >= nsplit) return;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ii = lane & 15, g = lane >> 4;
    const int64_t s_begin = (int64_t)sp * seg_per_split;
    const int64_t s_end = (s_begin + seg_per_split < nseg) ? (s_begin + seg_per_split) : nseg;
    const int64_t i0w = (int64_t)rb * 256 + 64 * w;
    int64_t rows = nrows - i0w;
    if (rows > 64) rows = 64;
    const int cps = (int)((klen + 63) >> 6);               // chunks per segment
    const int nchunk = (int)(s_end - s_begin) * cps;
    const int ldr4 = (int)(ldrow * 4);
    const int voff = (int)(((int64_t)ii * ldrow + 4 * g) * 4);

    f32x4 acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 xb[4][4];
    f32x4 areg[MT];

    // A tile of chunk q: KR values Fs[row][s] * Fk[row][k0 + ...], zero past the segment end.  The loads go out at the start
    // of the chunk before (genA), the product is formed in front of the LDS store (genA_finish): multiplied at once, the
    // Fs entry was waited for with s_waitcnt vmcnt(0) -- a drain of the X ring per chunk (tools/check_loop_drains.py).
    const rsrc_t rfs = nnf_make_rsrc(Fs, (uint32_t)((((int64_t)r - 1) * lds_ + nseg) * 4));   // rank rows >= r: zero
    const int rows4 = (int)((int64_t)(threadIdx.x & 15) * lds_ * 4), lds64 = (int)(lds_ * 64);
    float fs[MT];
    auto genA = [&](int q) {
        const int64_t s = s_begin + q / cps;
        const int64_t k0 = (int64_t)(q % cps) * 64;
#if SEG_ABL == 1
        return;
#endif
        stageA_load<MT>(Fk, ldk, r, (q < nchunk) ? klen : 0, k0, fk_vec_ok, areg);
        const int off = (q < nchunk) ? rows4 + (int)(s * 4) : (int)0x7ffffff0;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            fs[mt] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rfs, off, mt * lds64, 0));
    };
    auto genA_finish = [&]() {
#if SEG_ABL == 1
        for (int mt = 0; mt < MT; ++mt) areg[mt] = f32x4{1.f, 1.f, 1.f, 1.f};
        return;
#endif
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) areg[mt] *= fs[mt];
    };
    // one descriptor per segment (32-bit offsets stay inside the wave's 64 rows of one segment)
    auto seg_rsrc = [&](int q) -> rsrc_t {
        const int64_t s = s_begin + q / cps;
        const bool live = (q < nchunk) && rows > 0;
        const float* base = T + (live ? (i0w * ldrow + s * segstride) : 0);
        const uint32_t bytes = live ? (uint32_t)(((rows - 1) * ldrow + klen) * 4) : 0u;
        return nnf_make_rsrc(base, bytes);
    };
    // X of a whole chunk is requested at ONE point of the loop -- 16 loads back to back, the four 64-byte pieces of a row's
    // 256 bytes by consecutive instructions -- into the register set the previous chunk has finished with (two sets,
    // ping-pong).  Refilling each 16-column group right after its MFMAs (one set) spread the four pieces of a row over a
    // whole chunk of MFMAs: tools/mttkrp_ablate.sh seg showed the kernel bound by its X stream alone (142 us, 141 without its
    // MFMAs, 81 without the stream: 3.5 TB/s from 64-byte pieces of rows a megabyte apart).
    auto loadX = [&](f32x4 (&dst)[4][4], int q) {
        const rsrc_t rs = seg_rsrc(q);
        const int kb0 = (q % cps) * 256;                     // byte offset of the chunk inside the segment
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int t = 0; t < 4; ++t) dst[t][nt] = nnf_bload4<VEC>(rs, voff, nt * 16 * ldr4 + kb0 + 64 * t);
    };
    // (two sets only while they fit: rank tiles MT <= 2 -- 226 VGPRs at rank 30; with the 64 accumulators of MT = 4 the second
    //  set spills, tests/test_abi_and_host.py -- larger ranks refill each group's registers right after its MFMAs)
    constexpr bool PP = MT <= 2;
    f32x4 xb2[PP ? 4 : 1][4];
    auto chunk = [&](int q, f32x4 (&cur)[4][4], f32x4 (&nxt)[4][4]) {
        const f32x4* img = ldsA[q & 1];
        genA(q + 1);
#if SEG_ABL != 3
        if constexpr (PP) loadX(nxt, q + 1);
#endif
        const int64_t kbase = (int64_t)(q % cps) * 64;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4 af[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) af[mt] = img[(mt * 4 + t) * 64 + lane];
            // ragged segment tail: the bytes past klen belong to the next tensor row
#if SEG_ABL != 5
            const int64_t krem = klen - (kbase + 16 * t + 4 * g);
            if (krem < 4) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (c >= krem) cur[t][nt][c] = 0.f;
            }
#endif
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
#if SEG_ABL == 2
                        if (mt == 0) acc[0][nt][c] += af[0][c] * cur[t][nt][c];
#else
                        acc[mt][nt] = MFMA16(af[mt][c], cur[t][nt][c], acc[mt][nt]);
#endif
                    }
#if SEG_ABL != 3
            if constexpr (!PP) {   // one register set: this group's registers are free again
                const rsrc_t rs = seg_rsrc(q + 1);
                const int kb = ((q + 1) % cps) * 256 + 64 * t;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) cur[t][nt] = nnf_bload4<VEC>(rs, voff, nt * 16 * ldr4 + kb);
            }
#endif
        }
#if SEG_ABL != 4
        genA_finish();
        stageA_store<MT>(const_cast<f32x4*>(ldsA[(q + 1) & 1]), areg);
        __syncthreads();
#endif
    };

    genA(0);
    loadX(xb, 0);
    genA_finish();
    stageA_store<MT>(ldsA[0], areg);
    __syncthreads();

    if constexpr (PP) {
        auto& other = reinterpret_cast<f32x4 (&)[4][4]>(xb2);
        for (int q = 0; q < nchunk; q += 2) {
            chunk(q, xb, other);
            if (q + 1 < nchunk) chunk(q + 1, other, xb);
        }
    } else {
        for (int q = 0; q < nchunk; ++q) chunk(q, xb, xb);
    }
    float* sl = slabs + (int64_t)sp * r * ldp;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int64_t i = i0w + 16 * nt + ii;
        if (i < nrows) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int rk = 16 * mt + 4 * g + reg;
                    if (rk < r) sl[(int64_t)rk * ldp + i] = acc[mt][nt][reg];
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// mode 2:  slab[split][rk][k] = sum_{row in split} Fa[rk][row / nb] * Fb[rk][row % nb] * M[row][k],  M = T as (I*J) x K
// ---------------------------------------------------------------------------------------------------------
template <int MT, bool VEC, bool KRF>
__global__ __launch_bounds__(256, (MT <= 4 ? 2 : 1)) void nnf_mttkrp_rows_kernel(
    const float* __restrict__ M, int64_t m, int64_t n, int64_t ldx, const float* __restrict__ Fa, int64_t lda,
    const float* __restrict__ Fb, int64_t ldb, int64_t nb, int r, float* __restrict__ slabs, int64_t ldp, int ncb,
    int nsplit, int64_t rows_per_split) {
    __shared__ f32x4 ldsA[2][MT * 256];
    int ks, cb;
    nnf_xcd_map(blockIdx.x, ncb, ks, cb);
    if (ks >= nsplit) return;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int jj = lane & 15, g = lane >> 4;
    const int64_t i_begin = (int64_t)ks * rows_per_split;
    const int64_t i_end = (i_begin + rows_per_split < m) ? (i_begin + rows_per_split) : m;
    const int nchunk = (int)((i_end - i_begin + 63) >> 6);
    const int64_t jl = (int64_t)cb * 256 + w * 64 + 4 * jj;
    const rsrc_t rs = nnf_make_rsrc(M + i_begin * ldx, (uint32_t)(((i_end - i_begin - 1) * ldx + n) * 4));
    const int voff = (jl < n) ? (int)(((int64_t)4 * g * ldx + jl) * 4) : (int)0x7ffffff0;
    const int ldx4 = (int)(ldx * 4);

    f32x4 acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) acc[mt][cc] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 xb[4][4];
    f32x4 areg[MT];
    // Khatri-Rao operand of a chunk: areg[mt].c = Fa[row][i] * Fb[row][j] for tensor row (i*nb + j) = i_begin + 64q + 16t + 4g + c.
    // Issued as plain loads at the start of the chunk BEFORE (genA_issue), multiplied right before the LDS store
    // (genA_finish): the raw factor entries sit in registers across the chunk's MFMAs, so the only wait is the in-order
    // vmcnt(N) in front of the store with the whole X ring still in flight.  (The first form multiplied inside the
    // generation loop: hipcc turned the per-entry wrap test into real loops, each followed by s_waitcnt vmcnt(0) -- eight
    // full drains of the X prefetch ring per chunk, 32 of the kernel's 125 us, tools/mttkrp_ablate.sh.)
    // Buffer loads with hardware bounds checking: rank rows >= r and entries of tensor rows >= i_end read as zero / are
    // masked; (ia_c, ib_c) = (row / nb, row % nb) of the chunk issued next, carried from chunk to chunk (one division per
    // thread up front).  nb < 4 (several wraps inside a lane's four rows) or factors beyond 31-bit offsets: the slow form.
    const rsrc_t rfa = nnf_make_rsrc(Fa, (uint32_t)((((int64_t)r - 1) * lda + (m + nb - 1) / nb) * 4));
    const rsrc_t rfb = nnf_make_rsrc(Fb, (uint32_t)((((int64_t)r - 1) * ldb + nb) * 4));
    const int rowa4 = (int)((int64_t)(threadIdx.x & 15) * lda * 4), rowb4 = (int)((int64_t)(threadIdx.x & 15) * ldb * 4);
    const int lda64 = (int)(lda * 64), ldb64 = (int)(ldb * 64);   // byte step of one 16-row rank tile
    int64_t ia_c, ib_c;
    {
        const int64_t rr0 = i_begin + 16 * (threadIdx.x >> 6) + 4 * ((threadIdx.x & 63) >> 4);
        ia_c = rr0 / nb;
        ib_c = rr0 - ia_c * nb;
    }
    float fa0[MT], fa1[MT], fb[MT][4];
    int kr_wrap = 0, kr_valid = 0;   // bit c: row c of the lane's four wrapped into the next i / exists
    auto genA_slow = [&](int q) {
        const int t = threadIdx.x >> 6, L = threadIdx.x & 63;
        const int64_t rr = i_begin + 64 * (int64_t)q + 16 * t + 4 * (L >> 4);
        const int64_t ia0 = rr / nb, ib0 = rr - ia0 * nb;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int row = 16 * mt + (L & 15);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (row < r) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (rr + c < i_end) {
                        int64_t ia = ia0, ib = ib0 + c;
                        while (ib >= nb) { ib -= nb; ++ia; }
                        v[c] = Fa[(int64_t)row * lda + ia] * Fb[(int64_t)row * ldb + ib];
                    }
                }
            }
            areg[mt] = v;
        }
    };
    auto genA_issue = [&](int q) {
#if MTTKRP_ABL == 1
        return;
#endif
        if constexpr (!KRF) { genA_slow(q); return; }
        const int t = threadIdx.x >> 6, L = threadIdx.x & 63;
        const int64_t rr = i_begin + 64 * (int64_t)q + 16 * t + 4 * (L >> 4);
        const int ia0 = (int)ia_c, ib0 = (int)ib_c, nbi = (int)nb;
        if (nb >= 64) {
            ib_c += 64;
            if (ib_c >= nb) { ib_c -= nb; ++ia_c; }
        } else {
            const int64_t rn = rr + 64;
            ia_c = rn / nb;
            ib_c = rn - ia_c * nb;
        }
        const int64_t left = i_end - rr;                       // rows of this lane's four that exist (<= 0: none)
        kr_valid = left >= 4 ? 15 : left <= 0 ? 0 : ((1 << (int)left) - 1);
        kr_wrap = 0;
        int offb[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            int ib = ib0 + c;
            const bool wr = ib >= nbi;
            ib -= wr ? nbi : 0;
            kr_wrap |= wr ? (1 << c) : 0;
            offb[c] = ((kr_valid >> c) & 1) ? rowb4 + 4 * ib : (int)0x7ffffff0;
        }
        const int offa0 = (kr_valid & 1) ? rowa4 + 4 * ia0 : (int)0x7ffffff0;
        const int offa1 = (kr_valid & kr_wrap) ? rowa4 + 4 * ia0 + 4 : (int)0x7ffffff0;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            fa0[mt] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rfa, offa0, mt * lda64, 0));
            fa1[mt] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rfa, offa1, mt * lda64, 0));
#pragma unroll
            for (int c = 0; c < 4; ++c)
                fb[mt][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rfb, offb[c], mt * ldb64, 0));
        }
    };
    auto genA_finish = [&]() {
#if MTTKRP_ABL == 1
        for (int mt = 0; mt < MT; ++mt) areg[mt] = f32x4{1.f, 1.f, 1.f, 1.f};
        return;
#endif
        if constexpr (!KRF) return;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f32x4 v;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float fa = ((kr_wrap >> c) & 1) ? fa1[mt] : fa0[mt];
                v[c] = ((kr_valid >> c) & 1) ? fa * fb[mt][c] : 0.f;
            }
            areg[mt] = v;
        }
    };
    genA_issue(0);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) xb[t][c] = nnf_bload4<VEC>(rs, voff, (16 * t + c) * ldx4);
    genA_finish();
    stageA_store<MT>(ldsA[0], areg);
    __syncthreads();
    for (int q = 0; q < nchunk; ++q) {
        const f32x4* img = ldsA[q & 1];
        genA_issue(q + 1);
        const int soff_next = (q + 1) * 64 * ldx4;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4 af[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) af[mt] = img[(mt * 4 + t) * 64 + lane];
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
#if MTTKRP_ABL == 2
                        if (mt == 0 && cc == 0) acc[0][0] += xb[t][c] * af[0][c];
#else
                        acc[mt][cc] = MFMA16(af[mt][c], xb[t][c][cc], acc[mt][cc]);
#endif
                    }
#if MTTKRP_ABL != 3
#pragma unroll
            for (int c = 0; c < 4; ++c) xb[t][c] = nnf_bload4<VEC>(rs, voff, soff_next + (16 * t + c) * ldx4);
#endif
            __builtin_amdgcn_sched_barrier(0);   // keep this group's loads inside the group (see nnf_xty_kernel)
        }
#if MTTKRP_ABL != 4
        genA_finish();
        stageA_store<MT>(const_cast<f32x4*>(ldsA[(q + 1) & 1]), areg);
        __syncthreads();
#endif
    }
    if (jl < ldp) {
        float* sl = slabs + (int64_t)ks * r * ldp;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int rk = 16 * mt + 4 * g + reg;
                if (rk < r)
                    *reinterpret_cast<f32x4*>(sl + (int64_t)rk * ldp + jl) =
                        f32x4{acc[mt][0][reg], acc[mt][1][reg], acc[mt][2][reg], acc[mt][3][reg]};
            }
    }
}

template <int MT, bool VEC>
static int launch_seg(nnf_ctx* ctx, const float* T, int64_t nrows, int64_t ldrow, int64_t nseg, int64_t segstride,
                      int64_t klen, const float* Fs, int64_t lds_, const float* Fk, int64_t ldk, int r, float* out,
                      int64_t ldo, hipStream_t st) {
    if ((64 * ldrow + klen + 256) * 4 >= (int64_t)0x7fff0000 || (int64_t)(16 * MT) * lds_ * 4 >= (int64_t)0x7fff0000) return NNF_ERR_UNSUPPORTED;
    const int nrb = (int)nnf_cdiv(nrows, 256);
    const int64_t ldp = nnf_rup(nrows, 4);
    int64_t nsplit = 2 * (int64_t)ctx->num_cus / nrb;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > nseg) nsplit = nseg;
    nnf_ws_cursor cur(ctx);
    const int64_t slab_elems = (int64_t)r * ldp;
    const int64_t ws_max = (int64_t)(cur.remaining() / 4) / slab_elems;
    if (ws_max < 1) return NNF_ERR_WORKSPACE;
    if (nsplit > ws_max) nsplit = ws_max;
    const int64_t sps = nnf_cdiv(nseg, nsplit);
    nsplit = nnf_cdiv(nseg, sps);
    float* slabs = (float*)cur.take((size_t)nsplit * slab_elems * 4);
    if (!slabs) return NNF_ERR_WORKSPACE;
    const int fk_vec_ok = ((((uintptr_t)Fk) & 15) == 0 && (ldk & 3) == 0) ? 1 : 0;
    const int grid = 8 * (int)nnf_cdiv(nsplit, 8) * nrb;
    nnf_probe(ctx, NNF_PROBE_MTTKRP, 0, st);
    hipLaunchKernelGGL((nnf_mttkrp_seg_kernel<MT, VEC>), dim3(grid), dim3(256), 0, st, T, nrows, ldrow, nseg, segstride,
                       klen, Fs, lds_, Fk, ldk, r, slabs, ldp, nrb, (int)nsplit, sps, fk_vec_ok);
    NNF_CHECK_LAUNCH();
    nnf_probe(ctx, NNF_PROBE_MTTKRP, 1, st);
    return nnf_launch_reduce_slabs(slabs, (int)nsplit, slab_elems, r, nrows, ldp, out, ldo, st);
}

template <int MT, bool VEC>
static int launch_rows(nnf_ctx* ctx, const float* M, int64_t m, int64_t n, const float* Fa, int64_t lda, const float* Fb,
                       int64_t ldb, int64_t nb, int r, float* out, int64_t ldo, hipStream_t st) {
    const int ncb = (int)nnf_cdiv(n, 256);
    const int64_t ldp = nnf_rup(n, 4);
    int64_t nsplit = 2 * (int64_t)ctx->num_cus / ncb;
    if (nsplit < 1) nsplit = 1;
    const int64_t max_split = nnf_cdiv(m, 64);
    if (nsplit > max_split) nsplit = max_split;
    nnf_ws_cursor cur(ctx);
    const int64_t slab_elems = (int64_t)r * ldp;
    const int64_t ws_max = (int64_t)(cur.remaining() / 4) / slab_elems;
    if (ws_max < 1) return NNF_ERR_WORKSPACE;
    if (nsplit > ws_max) nsplit = ws_max;
    int64_t rps = nnf_rup(nnf_cdiv(m, nsplit), 64);
    while ((rps + 128) * n * 4 >= (int64_t)0x7fff0000) {
        if (rps <= 64) return NNF_ERR_UNSUPPORTED;
        rps = nnf_rup(rps / 2, 64);
    }
    nsplit = nnf_cdiv(m, rps);
    if (nsplit > ws_max) return NNF_ERR_WORKSPACE;
    float* slabs = (float*)cur.take((size_t)nsplit * slab_elems * 4);
    if (!slabs) return NNF_ERR_WORKSPACE;
    const int grid = 8 * (int)nnf_cdiv(nsplit, 8) * ncb;
    nnf_probe(ctx, NNF_PROBE_MTTKRP, 0, st);
    // buffer-addressed Khatri-Rao generation: 31-bit byte offsets into both factors, at most one wrap inside four rows
    const bool kr_fast = nb >= 4 && (int64_t)(16 * MT) * lda * 4 < (int64_t)0x7fff0000 && (int64_t)(16 * MT) * ldb * 4 < (int64_t)0x7fff0000;   // (padded rank rows: their offsets must not wrap either)
    if (kr_fast)
        hipLaunchKernelGGL((nnf_mttkrp_rows_kernel<MT, VEC, true>), dim3(grid), dim3(256), 0, st, M, m, n, n, Fa, lda, Fb, ldb, nb,
                           r, slabs, ldp, ncb, (int)nsplit, rps);
    else
        hipLaunchKernelGGL((nnf_mttkrp_rows_kernel<MT, VEC, false>), dim3(grid), dim3(256), 0, st, M, m, n, n, Fa, lda, Fb, ldb, nb,
                           r, slabs, ldp, ncb, (int)nsplit, rps);
    NNF_CHECK_LAUNCH();
    nnf_probe(ctx, NNF_PROBE_MTTKRP, 1, st);
    return nnf_launch_reduce_slabs(slabs, (int)nsplit, slab_elems, r, n, ldp, out, ldo, st);
}

#define MTTKRP_MT(FN, VEC, ...)                       \
    switch (MT) {                                     \
        case 1: return FN<1, VEC>(__VA_ARGS__);       \
        case 2: return FN<2, VEC>(__VA_ARGS__);       \
        case 3: return FN<3, VEC>(__VA_ARGS__);       \
        case 4: return FN<4, VEC>(__VA_ARGS__);       \
        case 5: return FN<5, VEC>(__VA_ARGS__);       \
        case 6: return FN<6, VEC>(__VA_ARGS__);       \
        case 7: return FN<7, VEC>(__VA_ARGS__);       \
        default: return FN<8, VEC>(__VA_ARGS__);      \
    }

static int seg_dispatch(nnf_ctx* ctx, const float* T, int64_t nrows, int64_t ldrow, int64_t nseg, int64_t segstride,
                        int64_t klen, const float* Fs, int64_t lds_, const float* Fk, int64_t ldk, int R, float* out,
                        int64_t ldo, hipStream_t st) {
    const int MT = (R + 15) / 16;
    const bool vec = ((((uintptr_t)T) & 15) == 0) && (ldrow % 4 == 0) && (segstride % 4 == 0);
    if (vec) { MTTKRP_MT(launch_seg, true, ctx, T, nrows, ldrow, nseg, segstride, klen, Fs, lds_, Fk, ldk, R, out, ldo, st) }
    else { MTTKRP_MT(launch_seg, false, ctx, T, nrows, ldrow, nseg, segstride, klen, Fs, lds_, Fk, ldk, R, out, ldo, st) }
}

static int rows_dispatch(nnf_ctx* ctx, const float* M, int64_t m, int64_t n, const float* Fa, int64_t lda,
                         const float* Fb, int64_t ldb, int64_t nb, int R, float* out, int64_t ldo, hipStream_t st) {
    const int MT = (R + 15) / 16;
    if (x_vec_ok(M, n)) { MTTKRP_MT(launch_rows, true, ctx, M, m, n, Fa, lda, Fb, ldb, nb, R, out, ldo, st) }
    else { MTTKRP_MT(launch_rows, false, ctx, M, m, n, Fa, lda, Fb, ldb, nb, R, out, ldo, st) }
}

extern "C" int nnf_mttkrp3_f32(nnf_ctx* ctx, const float* T, int64_t I, int64_t J, int64_t K, const float* Ft0,
                               int64_t ld0, const float* Ft1, int64_t ld1, const float* Ft2, int64_t ld2, int R, int mode,
                               float* out, int64_t ldo, void* stream) {
    if (!ctx || !T || !Ft0 || !Ft1 || !Ft2 || !out || I < 1 || J < 1 || K < 1 || R < 1 || ld0 < I || ld1 < J || ld2 < K)
        return NNF_ERR_ARG;
    if (mode < 0 || mode > 2) return NNF_ERR_ARG;
    if (R > NNF_MAX_RANK) {
        // ranks above 128: the rank rows of the result are independent of each other -- passes of <= 128 rows of the three
        // transposed factors (the tensor is read once per pass)
        for (int k0 = 0; k0 < R; k0 += NNF_MAX_RANK) {
            const int rc = nnf_mttkrp3_f32(ctx, T, I, J, K, Ft0 + (int64_t)k0 * ld0, ld0, Ft1 + (int64_t)k0 * ld1, ld1,
                                           Ft2 + (int64_t)k0 * ld2, ld2, R - k0 < NNF_MAX_RANK ? R - k0 : NNF_MAX_RANK, mode,
                                           out + (int64_t)k0 * ldo, ldo, stream);
            if (rc != NNF_OK) return rc;
        }
        return NNF_OK;
    }
    hipStream_t st = (hipStream_t)stream;
    if (mode == 0) {
        if (ldo < I) return NNF_ERR_ARG;
        return seg_dispatch(ctx, T, I, J * K, J, K, K, Ft1, ld1, Ft2, ld2, R, out, ldo, st);
    }
    if (mode == 1) {
        if (ldo < J) return NNF_ERR_ARG;
        return seg_dispatch(ctx, T, J, K, I, J * K, K, Ft0, ld0, Ft2, ld2, R, out, ldo, st);
    }
    if (ldo < K) return NNF_ERR_ARG;
    return rows_dispatch(ctx, T, I * J, K, Ft0, ld0, Ft1, ld1, J, R, out, ldo, st);
}

// ---------------------------------------------------------------------------------------------------------
// MTTKRP from a shared partial product (dimension tree).  Between the mode-0 and the mode-1 update of one_ntf_step
// (ntf.py:437-456) the last factor does not change, so both right-hand sides are contractions of the SAME
//   Y[r][i][j] = sum_k T[i][j][k] F2[k][r]           (= nnf_ttm3_f32(T, F2t, mode 2): one pass over T)
//   mode 0:  rhs[r][i] = sum_j Y[r][i][j] F1t[r][j]   (axis 2)      mode 1:  rhs[r][j] = sum_i Y[r][i][j] F0t[r][i]   (axis 1)
// which are R independent matrix-vector products over Y (R*I*J floats: 30 MB at 500^3, rank 30) instead of a second pass
// over the 500 MB tensor.  Same value as unfolded[mode] @ khatri_rao (the sums are re-associated, fp32 partials).
// ---------------------------------------------------------------------------------------------------------
// axis 2: one wave per (r, a) row of Y; lanes stride the row (coalesced), DPP-free shuffle reduction in a fixed order
__global__ __launch_bounds__(256) void nnf_partial_last_kernel(const float* __restrict__ Y, int64_t A, int64_t B,
                                                               const float* __restrict__ Ft, int64_t ldf, int r,
                                                               float* __restrict__ out, int64_t ldo) {
    const int lane = threadIdx.x & 63;
    const int64_t rows = (int64_t)r * A;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
        const int64_t k = row / A, a = row - k * A;
        const float* y = Y + row * B;
        const float* f = Ft + k * ldf;
        float s0 = 0.f, s1 = 0.f;   // even / odd 64-element pieces of the row, each in increasing order
        // eight pieces (both operands) in flight per trip, from clamped addresses: a load per trip waited for alone made a
        // 500-entry row four dependent memory round trips
        for (int64_t b0 = lane; b0 < B; b0 += 512) {
            float yv[8], fv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t b = b0 + 64 * u < B ? b0 + 64 * u : B - 1;
                yv[u] = y[b];
                fv[u] = f[b];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (b0 + 64 * u < B) {
                    if (u & 1) s1 = fmaf(yv[u], fv[u], s1);
                    else s0 = fmaf(yv[u], fv[u], s0);
                }
            }
        }
        float s = s0 + s1;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
        if (lane == 0) out[k * ldo + a] = s;
    }
}
// axis 1: grid (column blocks of 256, a-chunks, r); thread = one column b, sums its chunk of a in order -> slab[chunk][r][ldp]
__global__ __launch_bounds__(256) void nnf_partial_mid_kernel(const float* __restrict__ Y, int64_t A, int64_t B,
                                                              const float* __restrict__ Ft, int64_t ldf, int64_t a_per,
                                                              float* __restrict__ slabs, int64_t ldp, int r) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t chunk = blockIdx.y, k = blockIdx.z;
    const int64_t a0 = chunk * a_per, a1 = (a0 + a_per < A) ? a0 + a_per : A;
    if (b >= B) return;
    const float* y = Y + (k * A) * B + b;
    const float* f = Ft + k * ldf;     // wave-uniform operand
    float s = 0.f;
    for (int64_t a = a0; a < a1; a += 8) {   // eight rows in flight per trip (clamped addresses), added in row order
        float yv[8], fv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t aa = a + u < a1 ? a + u : a1 - 1;
            yv[u] = y[aa * B];
            fv[u] = f[aa];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (a + u < a1) s = fmaf(yv[u], fv[u], s);
    }
    slabs[(chunk * r + k) * ldp + b] = s;
}

extern "C" int nnf_mttkrp3_from_partial_f32(nnf_ctx* ctx, const float* Y, int64_t A, int64_t B, const float* Ft, int64_t ldf,
                                            int R, int axis, float* out, int64_t ldo, void* stream) {
    if (!ctx || !Y || !Ft || !out || A < 1 || B < 1 || R < 1 || (axis != 1 && axis != 2)) return NNF_ERR_ARG;
    if (ldf < (axis == 1 ? A : B) || ldo < (axis == 1 ? B : A)) return NNF_ERR_ARG;
    if (R > NNF_MAX_RANK) {   // R independent matrix-vector products: passes of <= 128 of them
        for (int k0 = 0; k0 < R; k0 += NNF_MAX_RANK) {
            const int rc = nnf_mttkrp3_from_partial_f32(ctx, Y + (int64_t)k0 * A * B, A, B, Ft + (int64_t)k0 * ldf, ldf,
                                                        R - k0 < NNF_MAX_RANK ? R - k0 : NNF_MAX_RANK, axis, out + (int64_t)k0 * ldo,
                                                        ldo, stream);
            if (rc != NNF_OK) return rc;
        }
        return NNF_OK;
    }
    hipStream_t st = (hipStream_t)stream;
    if (axis == 2) {
        int64_t grid = nnf_cdiv((int64_t)R * A, 4);
        if (grid > 8192) grid = 8192;
        hipLaunchKernelGGL(nnf_partial_last_kernel, dim3((int)grid), dim3(256), 0, st, Y, A, B, Ft, ldf, R, out, ldo);
        NNF_CHECK_LAUNCH();
        return NNF_OK;
    }
    // axis 1: enough a-chunks to fill the chip, at least 16 rows each
    const int64_t cb = nnf_cdiv(B, 256);
    int64_t nchunk = nnf_cdiv((int64_t)4 * ctx->num_cus, cb * R);
    if (nchunk < 1) nchunk = 1;
    if (nchunk > nnf_cdiv(A, 16)) nchunk = nnf_cdiv(A, 16);
    if (nchunk > 65535) nchunk = 65535;
    const int64_t a_per = nnf_cdiv(A, nchunk);
    nchunk = nnf_cdiv(A, a_per);
    const int64_t ldp = nnf_rup(B, 4);
    nnf_ws_cursor cur(ctx);
    float* slabs = (float*)cur.take((size_t)nchunk * R * ldp * 4);
    if (!slabs) return NNF_ERR_WORKSPACE;
    if (cb > 65535) return NNF_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(nnf_partial_mid_kernel, dim3((unsigned)cb, (unsigned)nchunk, (unsigned)R), dim3(256), 0, st, Y, A, B, Ft,
                       ldf, a_per, slabs, ldp, R);
    NNF_CHECK_LAUNCH();
    return nnf_launch_reduce_slabs(slabs, (int)nchunk, (int64_t)R * ldp, R, B, ldp, out, ldo, st);
}
