// potrf_lookahead.hpp -- k_potrf_solve_la: the Cholesky of M and the triangular solve in one launch, with the trailing update
// spread over several compute units ("look-ahead" form): the chain workgroup and the far workgroups.  The kernel itself
// (k_potrf_solve_la) is in msckf_kernels.hpp behind the solver workgroups it shares with k_potrf_solve; k_front runs the same two
// bodies on the prior.
//
// k_potrf_solve keeps the whole trailing matrix in the registers of ONE workgroup: its first block steps are bound by the FP64
// matrix rate of one CU (66 + 55 + 45 + ... tile products on three SIMDs, 8-13 k cycles per step against the 4.6 k of the
// pivot chain) and its prologue by one CU's memory pipeline (190 KB of tiles).  Here the chain workgroup only ever holds
//   * the block row it is about to turn into a panel (sRow), and the last LA panels (sPan),
// and every other block row a is brought forward by a FAR workgroup of its own (one workgroup per row a = LA+1 .. nb-2: the
// tiles (a, b > a) and the diagonal tile (a+1, a+1)): it applies the panels 0 .. a-1-LA as they are published (the step counter
// the solver workgroups poll as well), stores the row into R's own tiles and raises rdy[a].  The chain workgroup's workers fetch row a
// in the middle of step a-2 (raw loads that stay in flight until step a-1), apply the LA panels the far workgroup has not seen
// during step a-1, and the row is the panel of step a.  Nothing of this sits on the chain: a far workgroup has LA-1 block steps (+ what is left of the step in
// which the panel was published) for two hand-offs through L2 (~1.3 us each) and LA panels' products.
//
// Roles of the chain workgroup (8 wavefronts): wave 0 the pivot chain (as potrf_reg_body: own panel tile, next diagonal tile,
// DPP sweep); waves 1-3, 5-7 the workers (panel tiles of the step, then the arriving row); wave 4 the publisher: it alone writes
// R and inv(L11) to memory (from LDS), drains its stores and raises the step counter -- the workers' prefetches stay in flight
// across the block steps (no vmcnt wait in the step barrier).  (The long early rows take the publisher 2-3 k cycles to drain: it
// raises their counter at the top of the next step instead of holding the step barrier; the last rows at once.)
//
// Inter-workgroup hand-offs follow MI355X_MICROARCH.md "Valid forms" (one lane of the storing workgroup signals for all of
// that workgroup's sc1 stores, behind every storing wave's vmcnt(0) and a workgroup barrier; the consumer's polling wave loads
// sc1 after its poll matched).  Every wait is bounded; a wait that gives up sets *lost (the host runs the update again through
// k_potrf_solve).
#pragma once
// (included from msckf_kernels.hpp INSIDE namespace orcvio_amd, behind potrf_reg_body: it uses DiagStep, the LDS barriers, st_tile)

#define LA_NBMAX 14
template <int LA>
__host__ __device__ constexpr int la_lds_doubles() { return 1360 + 16 + (LA + 1) * LA_NBMAX * 256 + 2 * 256; }
__host__ __device__ inline int la_far_workgroups(int nb, int LA) { return nb - 2 - LA > 0 ? nb - 2 - LA : 0; }

// Tile loads are RAW (four sc1 loads through 32-bit byte offsets from a wave-uniform base, no edge handling where they are issued):
// nothing waits for them there, the consumer masks the rows / columns beyond the matrix where it uses them (unit diagonal, zero
// elsewhere).  The offsets of a tile of the INPUT clamp row and column into the matrix (the prior is n x n with ld = n: nothing
// beyond it may be read) and, with rev, count down from the far corner: tile (a, b) of X'(i, j) = X(n-1-i, n-1-j) (potrf_reg_body).
struct LaIn { const double* X; int ld; int n; int rev; };
__device__ __forceinline__ void la_x_offsets(unsigned (&off)[4], const LaIn& in, int a, int b, int kk, int cc) {
    int j = 16 * b + cc;
    j = j < in.n ? j : in.n - 1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int i = 16 * a + kk + 4 * r;
        i = i < in.n ? i : in.n - 1;
        off[r] = (unsigned)((in.rev ? (in.n - 1 - i) * in.ld + (in.n - 1 - j) : i * in.ld + j) * 8);
    }
}
__device__ __forceinline__ void la_r_offsets(unsigned (&off)[4], int ldr, int a, int b, unsigned lane_br) {
#pragma unroll
    for (int r = 0; r < 4; ++r) off[r] = lane_br + (unsigned)(((16 * a + 4 * r) * ldr + 16 * b) * 8);
}
__device__ __forceinline__ void la_load_off(d4& v, const double* __restrict__ base, const unsigned (&off)[4]) {
    const char* t = reinterpret_cast<const char*>(base);   // (wave-uniform: the saddr form of global_load, no 64-bit VALU address arithmetic)
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = ld_pub(reinterpret_cast<const double*>(t + off[r]));
}
__device__ __forceinline__ d4 la_mask_edge(d4 v, int n, int a, int b, int kk, int cc) {
    const int j = 16 * b + cc;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = 16 * a + kk + 4 * r;
        v[r] = (i < n && j < n) ? v[r] : ((i == j) ? 1.0 : 0.0);
    }
    return v;
}
__device__ __forceinline__ d4 la_load_x_raw(const LaIn& in, int a, int b, int kk, int cc) {
    unsigned off[4];
    la_x_offsets(off, in, a, b, kk, cc);
    d4 v;
    la_load_off(v, in.X, off);
    return v;
}
__device__ __forceinline__ d4 la_load_x(const LaIn& in, int a, int b, int kk, int cc) { return la_mask_edge(la_load_x_raw(in, a, b, kk, cc), in.n, a, b, kk, cc); }
// tile (a, b) of R as another workgroup stored it
__device__ __forceinline__ d4 la_load_r(const double* __restrict__ R, int ldr, int a, int b, int kk, int cc) {
    unsigned off[4];
    la_r_offsets(off, ldr, a, b, (unsigned)((kk * ldr + cc) * 8));
    d4 v;
    la_load_off(v, R, off);
    return v;
}
__device__ __forceinline__ bool la_wait_ge(const int* p, int need, int limit) {   // bounded poll of a word another workgroup raises
#pragma unroll 1
    for (int it = 0; it < limit; ++it) {
        if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >= need) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

// c - A^T-operand * b: the FP64 matrix instruction negates its A operand itself (the BLGP field of v_mfma_f64 is its NEG mask on
// gfx940+: bit 0 = A) -- four v_xor of the sign bit per product less on an issue-bound instruction stream
__device__ __forceinline__ d4 mfma_f64_na(double a, double b, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 1); }

// stamps (diagnostic, scripts/gpu_potrf_la_stamps.py): [0..63] chain wave, [64..127] worker 0, [128..191] wave 4, [192..255] the far
// workgroup of row min(LA+4, nb-2), [256], [257] wall clock at start / end, [320 + 16 wi + kb] / [416 + 16 wi + kb] every worker's end of products / arrival at the barrier
#define LA_STAMP(ptr, idx) do { if constexpr (ST) { if ((ptr) && l == 0) (ptr)[idx] = clock64(); } } while (0)

// PRIOR: the input is a covariance (semi-definite: pivots <= tol_rel * largest diagonal entry are dropped, DiagStep<.., true>; n x n
// with any ld, optionally read reversed) and nobody trails the factorisation but the far workgroups: flag and rdy are put back to
// zero at the end (no kernel ahead of k_front clears them).  !PRIOR: M = s2 I + ..., positive definite, NP x NP.
template <int LA, bool ST, bool PRIOR, bool KEEP_WORDS = false>   // KEEP_WORDS: the caller puts flag / rdy back to zero (others still read them)
__device__ __forceinline__ void potrf_la_chain_wg(double* __restrict__ lds, const LaIn in, const double tol_rel,
                                                  double* __restrict__ R, int ldr, double* __restrict__ Dinv, int* __restrict__ info,
                                                  int* __restrict__ flag, int* __restrict__ rdy, int* __restrict__ lost,
                                                  const int spin, unsigned long long* __restrict__ stamps) {
    const int n = in.n;
    double (*sD)[17] = reinterpret_cast<double (*)[17]>(lds);                      // diagonal tile being factored (row view)
    double (*sDi)[16][17] = reinterpret_cast<double (*)[16][17]>(lds + 272);       // inv(L11) of block step kb in sDi[kb & 1]
    double (*sL)[16][17] = reinterpret_cast<double (*)[16][17]>(lds + 816);        // L11 of block step kb (rows) in sL[kb & 1]
    int* sCnt = reinterpret_cast<int*>(lds + 1360);                                // [16] waves that have written their panel tiles of step kb
    double* sRow = lds + 1376;                                                     // [NBMAX][256] block row kb, up to date (accumulator layout)
    double* sPan = sRow + LA_NBMAX * 256;                                          // [LA][NBMAX][256] panels of the last LA steps
    double* sDiag = sPan + LA * LA_NBMAX * 256;                                    // [2][256] diagonal tile k, up to date through panel k-2, in [k & 1]
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
    const int kk = l >> 4, cc = l & 15;
    const int nb = (n + 15) >> 4;
    const int nb2 = (nb + 1) & ~1;   // every wave passes nb2 step barriers
    if constexpr (ST) { if (stamps && tid == 0) stamps[256] = wall_clock64(); }
    if (tid < 16) sCnt[tid] = 0;   // first use is behind barrier A of step 0

    if (wave == 0) {
        // ================================ the pivot chain ================================
        LA_STAMP(stamps, 0);
        double dmin = INFINITY;
        double tol = 0.0;
        {
            const d4 d0 = la_load_x(in, 0, 0, kk, cc);
            if (PRIOR && tol_rel > 0.0) {   // pivot tolerance: relative to the largest diagonal entry (one gather)
                double mx = 0.0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int e = l + 64 * r;
                    const double dv = in.X[(size_t)(e < n ? e : 0) * (in.ld + 1)];
                    mx = fmax(mx, e < n ? dv : 0.0);
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
                tol = tol_rel * mx;
            }
            int z = 0;
            asm volatile("" : "+v"(z));
#pragma unroll
            for (int r = 0; r < 4; ++r) (&sD[0][0] + z)[(kk + 4 * r) * 17 + cc] = d0[r];
        }
        double v[16];
        auto sweep_tile = [&](int kb) {
            int z = 0;
            asm volatile("" : "+v"(z));
            const double* pD = &sD[0][0] + z;
            double* pDi = &sDi[0][0][0] + z + (kb & 1) * 272;
            double y[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const double a = pD[cc * 17 + c];
                v[c] = (c <= cc) ? a : 0.0;
                y[c] = (c == cc) ? 1.0 : 0.0;
            }
            DiagStep<0, PRIOR>::run(v, y, tol, dmin);   // (M is positive definite by construction: no pivot test on its chain)
            if (l < 16) {
                double* pL = &sL[0][0][0] + z + (kb & 1) * 272;
#pragma unroll
                for (int c = 0; c < 16; ++c) pDi[c * 17 + l] = y[c];
#pragma unroll
                for (int c = 0; c < 16; ++c) pL[l * 17 + c] = v[c];
            }
        };
        __builtin_amdgcn_s_setprio(3);
        wave_sync();
        LA_STAMP(stamps, 1);
        sweep_tile(0);
        LA_STAMP(stamps, 2);
        for (int kb = 0; kb < nb2; ++kb) {
            lds_barrier();   // A: row kb and the diagonal tile kb+1 are up to date in LDS, inv(L11) of step kb is in sDi
            if (kb >= nb) break;   // (the padding step of an odd nb: the workers' loop is unrolled by two)
            LA_STAMP(stamps, 3 + 3 * kb);
            const int kn = kb + 1;
            if (kn < nb) {
                int z = 0;
                asm volatile("" : "+v"(z));
                const double* pDi = &sDi[0][0][0] + z + (kb & 1) * 272;
                const double* st = sRow + z + kn * 256 + l;
                const double* pGn = sDiag + z + (kn & 1) * 256 + l;
                double li[4], sv[4];
                d4 t;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) { li[s4] = pDi[cc * 17 + kk + 4 * s4]; sv[s4] = st[s4 * 64]; t[s4] = pGn[s4 * 64]; }
                d4 x = {0, 0, 0, 0};
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) x = mfma_f64(li[s4], sv[s4], x);
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) t = mfma_f64_na(x[s4], x[s4], t);
                __builtin_amdgcn_sched_barrier(0);
                double* pPan = sPan + z + ((kb % LA) * LA_NBMAX + kn) * 256 + l;
#pragma unroll
                for (int r = 0; r < 4; ++r) pPan[r * 64] = x[r];
                lds_publish_count(&sCnt[kb], l);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 4; ++r) (&sD[0][0] + z)[(kk + 4 * r) * 17 + cc] = t[r];
                wave_sync();
                sweep_tile(kn);
            } else {
                lds_publish_count(&sCnt[kb], l);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if (l == 0) {   // (M: any pivot that is not positive is a failure; sCnt[30..31]: the two counters for the launch's own finish, LaFin)
            const int bad = PRIOR ? ((dmin < -tol) ? 1 : 0) : (!(dmin > 0.0) ? 1 : 0);
            info[1] = bad;
            sCnt[31] = bad;
        }
        LA_STAMP(stamps, 63);
    } else if (wave == 4) {
        // ================================ the publisher ================================
        const unsigned lane_b = (unsigned)((kk * ldr + cc) * 8);
        unsigned long long* st4 = stamps ? stamps + 128 : nullptr;
        if (nb > 1) {
            const d4 d1 = la_load_x(in, 1, 1, kk, cc);
#pragma unroll
            for (int r = 0; r < 4; ++r) sDiag[256 + r * 64 + l] = d1[r];
        }
        int ndrop = 0;
        bool pend = false;   // the stores of the previous step are not signalled yet
        for (int kb = 0; kb < nb2; ++kb) {
            int z = 0;
            asm volatile("" : "+v"(z));
            lds_barrier();   // A
            if (kb >= nb) break;
            if (pend) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (l == 0) __hip_atomic_store(flag, kb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                LA_STAMP(st4, 2 * (kb - 1) + 1);
                pend = false;
            }
            const double* pDi = &sDi[0][0][0] + z + (kb & 1) * 272;
#pragma unroll
            for (int r = 0; r < 4; ++r) st_pub<true>(Dinv + (size_t)kb * 256 + l + 64 * r, pDi[(kk + 4 * r) * 17 + cc]);
            const double* pL = &sL[0][0][0] + z + (kb & 1) * 272;
            double* ub = R + (size_t)(16 * kb) * ldr + 16 * kb;
#pragma unroll
            for (int r = 0; r < 4; ++r) st_tile<true>(ub + (size_t)(4 * r) * ldr, lane_b, (kk + 4 * r <= cc) ? pL[cc * 17 + kk + 4 * r] : 0.0);
            {
                const bool dropped = l < 16 && pL[(l & 15) * 18] == 0.0;
                ndrop += __builtin_popcountll(__ballot(dropped));
            }
            // the panel of this step, from LDS to memory, as soon as every wave has written its tiles
            while (__hip_atomic_load(&sCnt[kb], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < POTRF_NW + 1) __builtin_amdgcn_s_sleep(1);
            const double* pp = sPan + z + (kb % LA) * LA_NBMAX * 256 + l;
            double* urow = R + (size_t)(16 * kb) * ldr;
#pragma unroll 2
            for (int b = kb + 1; b < nb; ++b) {
                double x[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) x[r] = pp[b * 256 + r * 64];
#pragma unroll
                for (int r = 0; r < 4; ++r) st_tile<true>(urow + 16 * b + (size_t)(4 * r) * ldr, lane_b, x[r]);
            }
            // The counter rises once this wave's stores have landed.  For the long early rows that takes 2-3 k cycles: the wave does not
            // hold the step barrier for it, it raises the counter at the top of the next step (the far workgroups have LA-1 steps of
            // slack, the solver workgroups trail anyway); the last rows are short and are signalled at once (the solve's tail).
            if (nb - 1 - kb <= 3) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (l == 0) __hip_atomic_store(flag, kb + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                LA_STAMP(st4, 2 * kb + 1);
            } else {
                pend = true;
            }
        }
        if (l == 0) { info[0] = ndrop; sCnt[30] = ndrop; }
    } else {
        // ================================ the workers ================================
        const int wi = (wave < 4) ? wave - 1 : 10 - wave;   // 0 .. POTRF_NW-1; the SIMDs 1, 2, 3 hold the pairs (0, 5), (1, 4), (2, 3)
        // Dealing of a block row's tiles b = k+2 .. nb-1 (same for the panel of step k and the row k+1 that arrives in it): the first six
        // one each, the next ones to the workers 4, 3, 0, 1, 2, 5 in that order; the arriving diagonal tile is always worker 5's (it has
        // one row tile unless the row has twelve): the three SIMDs stay within a tile product or two of each other.
        const int w2 = (wi == 4) ? 0 : (wi == 3) ? 1 : (wi == 0) ? 2 : (wi == 1) ? 3 : (wi == 2) ? 4 : 5;
        unsigned long long* stw = (stamps && wi == POTRF_NW - 1) ? stamps + 64 : nullptr;   // (detailed stamps: worker 5, the second wavefront of SIMD 1)
        bool gone = false;                                  // a wait gave up
        {   // block row 0 -> sRow
            d4 t0[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int b = 1 + wi + POTRF_NW * q;
                if (b < nb) t0[q] = la_load_x_raw(in, 0, b, kk, cc);
            }
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int b = 1 + wi + POTRF_NW * q;
                if (b < nb) {
                    const d4 tm = la_mask_edge(t0[q], n, 0, b, kk, cc);
#pragma unroll
                    for (int r = 0; r < 4; ++r) sRow[b * 256 + r * 64 + l] = tm[r];
                }
            }
        }
        // what arrives in step k: the tiles (k+1, b) of block row k+1, b = k+2+wi (and one more: see the dealing above), and the diagonal tile k+2 (worker 5);
        // from X while no far workgroup has touched them (rows <= LA, diagonal tiles <= LA+1), else from R behind rdy[]
        const unsigned lane_br = (unsigned)((kk * ldr + cc) * 8);
        auto fetch = [&](int k, d4& f0, d4& f1, d4& fd) {   // (raw: masked where they are used)
            // ONE site per step, and a tile this worker does not have keeps its register as it is (no second definition from another
            // load site): the compiler then leaves the loads in flight over the step instead of copying them -- behind a vmcnt(0) --
            // where two definitions would meet
            const int ra = k + 1, da = k + 2, b0 = k + 2 + wi, b1 = k + 2 + POTRF_NW + w2;
            const bool rx = ra <= LA, dx = da <= LA + 1;   // (from the input while no far workgroup has touched them)
            unsigned o[4];
            if (b0 < nb) {
                if (rx) la_x_offsets(o, in, ra, b0, kk, cc); else la_r_offsets(o, ldr, ra, b0, lane_br);
                la_load_off(f0, rx ? in.X : R, o);
            }
            if (b1 < nb) {
                if (rx) la_x_offsets(o, in, ra, b1, kk, cc); else la_r_offsets(o, ldr, ra, b1, lane_br);
                la_load_off(f1, rx ? in.X : R, o);
            }
            if (da < nb && wi == POTRF_NW - 1) {
                if (dx) la_x_offsets(o, in, da, da, kk, cc); else la_r_offsets(o, ldr, da, da, lane_br);
                la_load_off(fd, dx ? in.X : R, o);
            }
        };
        const bool ragged = (n & 15) != 0;   // the last block row / column reaches past the matrix
        // the word step k's arrivals of THIS worker wait for (-1: none): row k+1 and the diagonal tile k+2 are one far workgroup's
        auto need_word = [&](int k) -> int {
            const int ra = k + 1;
            const bool row = ra > LA && k + 2 + wi < nb;
            const bool dg = k + 2 > LA + 1 && k + 2 < nb && wi == POTRF_NW - 1;
            return (row || dg) ? ra : -1;
        };
        // Two register sets for the arriving tiles, used alternately (the loop below is unrolled by two): a prefetch writes the set the
        // NEXT step reads, and nothing is ever copied between them (a copy would wait for the loads at the end of the step that
        // issued them).
        d4 ea0 = {0, 0, 0, 0}, ea1 = ea0, ead = ea0, eb0 = ea0, eb1 = ea0, ebd = ea0;
        fetch(0, ea0, ea1, ead);
        LA_STAMP(stw, 0);
        auto step = [&](const int kb, d4& c0, d4& c1, d4& cd, d4& n0, d4& n1, d4& nd) {
            int z = 0;
            asm volatile("" : "+v"(z));
            lds_barrier();   // A
            const bool live = kb < nb;   // (false in the padding step of an odd nb: barrier and fetch only)
            // the word the NEXT step's arrivals wait for: sampled here, looked at behind the panel phase (a poll costs a round trip to L2)
            const int w = (kb + 1 < nb) ? need_word(kb + 1) : -1;
            int rv = 1;
            if (w >= 0) rv = __hip_atomic_load(&rdy[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // ---- LDS reads first: the panel phase's (they gate the step), then the operands of the older panels kb-LA+1 .. kb-1 of
            //      the arriving tiles (in LDS since the last barrier).  Everything below is written for a SHORT instruction stream --
            //      two workers share a SIMD and the step is bound by what it can issue, not by the matrix pipe: wave-uniform branches
            //      around whole groups of four products, no per-product selects, no zero operands. ----
            const int b0 = kb + 2 + wi, b1 = kb + 2 + POTRF_NW + w2;
            const bool one = b0 < nb, two = b1 < nb;
            const int ra = kb + 1, da = kb + 2;
            const bool dg = da < nb && wi == POTRF_NW - 1;
            const double* pDi = &sDi[0][0][0] + z + (kb & 1) * 272;
            double* pPan = sPan + z + (kb % LA) * LA_NBMAX * 256 + l;
            double li[4], sa[4], sbv[4];
            if (one) {
                const double* st0 = sRow + z + b0 * 256 + l;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) { li[s4] = pDi[cc * 17 + kk + 4 * s4]; sa[s4] = st0[s4 * 64]; }
            }
            if (two) {
                const double* st1 = sRow + z + b1 * 256 + l;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) sbv[s4] = st1[s4 * 64];
            }
            double oa[LA - 1][4], o0[LA - 1][4], o1[LA - 1][4], od[LA - 1][4];
#pragma unroll
            for (int j = 0; j < LA - 1; ++j) {
                const int p = kb - 1 - j;
                if (p >= 0) {   // (the first steps have fewer panels behind them)
                    const double* q = sPan + z + (p % LA) * LA_NBMAX * 256 + l;
                    if (one) {
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4) { oa[j][s4] = q[ra * 256 + s4 * 64]; o0[j][s4] = q[b0 * 256 + s4 * 64]; }
                    }
                    if (two) {
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4) o1[j][s4] = q[b1 * 256 + s4 * 64];
                    }
                    if (dg) {
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4) od[j][s4] = q[da * 256 + s4 * 64];
                    }
                }
            }
            // ---- panel tiles (kb, b), b = kb+2+wi (and one more): inv(L11) * tile -> sPan ----
            d4 x0 = {0, 0, 0, 0}, x1 = {0, 0, 0, 0};
            if (one) {
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) x0 = mfma_f64(li[s4], sa[s4], x0);
#pragma unroll
                for (int r = 0; r < 4; ++r) pPan[b0 * 256 + r * 64] = x0[r];
            }
            if (two) {
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) x1 = mfma_f64(li[s4], sbv[s4], x1);
#pragma unroll
                for (int r = 0; r < 4; ++r) pPan[b1 * 256 + r * 64] = x1[r];
            }
            if (live) lds_publish_count(&sCnt[kb], l);
            LA_STAMP(stw, 1 + 4 * kb);
            // ---- the arriving row kb+1 and diagonal tile kb+2: the older panels now (ascending: the order of k_potrf_solve -- the two
            //      kernels agree bit for bit), the panel of this step behind the count of its tiles ----
            d4 a0 = c0, a1 = c1, ad = cd;
            if (ragged) {   // (tiles of X as they lie in memory; what a far workgroup stored is masked already)
                if (ra <= LA && b0 == nb - 1) a0 = la_mask_edge(a0, n, ra, b0, kk, cc);
                if (ra <= LA && b1 == nb - 1) a1 = la_mask_edge(a1, n, ra, b1, kk, cc);
                if (da <= LA + 1 && da == nb - 1) ad = la_mask_edge(ad, n, da, da, kk, cc);
            }
#pragma unroll
            for (int j = LA - 2; j >= 0; --j) {
                if (kb - 1 - j >= 0) {
                    if (one) {
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4) a0 = mfma_f64_na(oa[j][s4], o0[j][s4], a0);
                    }
                    if (two) {
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4) a1 = mfma_f64_na(oa[j][s4], o1[j][s4], a1);
                    }
                    if (dg) {
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4) ad = mfma_f64_na(od[j][s4], od[j][s4], ad);
                    }
                }
            }
            LA_STAMP(stw, 2 + 4 * kb);
            // ---- the arrivals of the next step: fetched here, more than half a step ahead of their use (ONE site, unconditional); the
            //      word sampled at the top of the step has come back under the products above ----
            if (w >= 0 && __builtin_amdgcn_readfirstlane(rv) < 1 && !gone && !la_wait_ge(&rdy[w], 1, spin)) gone = true;
            fetch(kb + 1 < nb ? kb + 1 : nb - 1, n0, n1, nd);
            LA_STAMP(stw, 3 + 4 * kb);
            if (live) {
                while (__hip_atomic_load(&sCnt[kb], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < POTRF_NW + 1) __builtin_amdgcn_s_sleep(1);
            }
            LA_STAMP(stw, 4 + 4 * kb);
            // panel kb: the row operand is the chain's tile (kb, kb+1), the column operands this worker's own x0 / x1
            double qa[4], qd[4];
            if (one) {
                const double* q = sPan + z + ((kb % LA) * LA_NBMAX + ra) * 256 + l;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) qa[s4] = q[s4 * 64];
            }
            if (dg) {
                const double* q = sPan + z + ((kb % LA) * LA_NBMAX + da) * 256 + l;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) qd[s4] = q[s4 * 64];
            }
            if (one) {
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) a0 = mfma_f64_na(qa[s4], x0[s4], a0);
#pragma unroll
                for (int r = 0; r < 4; ++r) sRow[z + b0 * 256 + r * 64 + l] = a0[r];
            }
            if (two) {
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) a1 = mfma_f64_na(qa[s4], x1[s4], a1);
#pragma unroll
                for (int r = 0; r < 4; ++r) sRow[z + b1 * 256 + r * 64 + l] = a1[r];
            }
            if (dg) {
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) ad = mfma_f64_na(qd[s4], qd[s4], ad);
#pragma unroll
                for (int r = 0; r < 4; ++r) sDiag[z + (da & 1) * 256 + r * 64 + l] = ad[r];
            }
            if constexpr (ST) { if (stamps && l == 0 && live) stamps[320 + 16 * wi + kb] = clock64(); }   // (every worker: products of step kb done)
        };
        for (int kb = 0; kb < nb2; kb += 2) {
            step(kb, ea0, ea1, ead, eb0, eb1, ebd);
            step(kb + 1, eb0, eb1, ebd, ea0, ea1, ead);
        }
        if (gone && l == 0) atomicExch(lost, 1);
        LA_STAMP(stw, 63);
    }
    if constexpr (ST) { if (stamps && tid == 0) stamps[257] = wall_clock64(); }
    if constexpr (PRIOR && !KEEP_WORDS) {   // every far workgroup has handed its row over (they were all picked up): nobody reads the words any more
        __syncthreads();
        if (wave == 4) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (this wave's last store of the step counter has landed)
            if (l < 16) __hip_atomic_store(rdy + l, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (l == 16) __hip_atomic_store(flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// The far workgroup of block row a (LA+1 <= a <= nb-2): tiles (a, b), b = a+1 .. nb-1, and the diagonal tile (a+1, a+1); the
// panels 0 .. a-1-LA; one or two tiles per wavefront.
template <int LA, bool ST>
__device__ __forceinline__ void potrf_la_far_wg(const int a, const LaIn in, double* __restrict__ R, int ldr,
                                                const int* __restrict__ flag, int* __restrict__ rdy, int* __restrict__ lost,
                                                const int spin, unsigned long long* __restrict__ stamps) {
    const int n = in.n;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
    const int kk = l >> 4, cc = l & 15;
    const int nb = (n + 15) >> 4;
    const int ntile = nb - a;   // nb-1-a off-diagonal tiles + the diagonal tile a+1
    unsigned long long* stf = (stamps && a == ((LA + 4 < nb - 2) ? LA + 4 : nb - 2)) ? stamps + 192 : nullptr;   // (one far workgroup's timeline)
    // tile q of the group: q < nb-1-a: (a, a+1+q); q == nb-1-a: (a+1, a+1)
    int ta[2], tb[2];
    bool live[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int q = wave + 8 * u;
        live[u] = q < ntile;
        ta[u] = (q < nb - 1 - a) ? a : a + 1;
        tb[u] = (q < nb - 1 - a) ? a + 1 + q : a + 1;
    }
    d4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
    for (int u = 0; u < 2; ++u)
        if (live[u]) acc[u] = la_load_x(in, ta[u], tb[u], kk, cc);
    const int plast = a - 1 - LA;
    bool gone = false;
    if (wave == 0) LA_STAMP(stf, 0);
    for (int p = 0; p <= plast; ++p) {
        if (wave == 0) {   // one wave polls the step counter for the workgroup
            if (!gone && !la_wait_ge(flag, p + 1, spin)) gone = true;
            LA_STAMP(stf, 1 + 3 * p);
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (live[u]) {
                const d4 qa = la_load_r(R, ldr, p, ta[u], kk, cc);
                const d4 qb = (tb[u] == ta[u]) ? qa : la_load_r(R, ldr, p, tb[u], kk, cc);
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) acc[u] = mfma_f64_na(qa[s4], qb[s4], acc[u]);
            }
        }
        if (wave == 0) LA_STAMP(stf, 2 + 3 * p);
    }
    const unsigned lane_b = (unsigned)((kk * ldr + cc) * 8);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if (live[u]) {
            double* ub = R + (size_t)(16 * ta[u]) * ldr + 16 * tb[u];
#pragma unroll
            for (int r = 0; r < 4; ++r) st_tile<true>(ub + (size_t)(4 * r) * ldr, lane_b, acc[u][r]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_store(&rdy[a], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gone) atomicExch(lost, 1);
    }
    if (wave == 0) LA_STAMP(stf, 3 * (plast + 1) + 1);
}

#undef LA_STAMP
