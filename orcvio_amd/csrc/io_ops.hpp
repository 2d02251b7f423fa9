// io_ops.hpp -- the host side of an update without the host: inputs are pulled out of the handle's pinned arena by the
// first kernel of the launch graph, results are pushed into host-coherent memory by the last one, and the caller waits on
// a word in that memory instead of a stream synchronisation (SURVEY.md 8d "host-visible" latency; call sites
// src/orcvio.cpp:2497-2560, :2803-2851).  HBM-/PCIe-bound byte movers: 16 bytes per lane, coalesced.
#pragma once
#include <hip/hip_runtime.h>

namespace orcvio_amd {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // 16 bytes per lane

// d_in <- pinned host arena (device-visible).  One pass, every lane one 16-byte load in flight per iteration; the grid is
// sized by the caller so that an iteration or two covers the block (PCIe reads want many requests outstanding).
// zero / zero_mask: workgroup 0 also clears the words zero[i] whose bit i of zero_mask is set (the pivot counters and shard status
// words an object update accumulates into: no fill launch in front of its compression).
__global__ __launch_bounds__(256) void k_ingest(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16, int* __restrict__ zero = nullptr,
                                                unsigned zero_mask = 0u) {
    if (zero && blockIdx.x == 0 && threadIdx.x < 32 && (zero_mask >> threadIdx.x & 1u)) zero[threadIdx.x] = 0;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) dst[i] = __builtin_nontemporal_load(src + i);
}

// The epilogue of a zero-copy update, ONE launch behind k_finish_sqrt:
//   workgroup 0                 the small result block [info | dx | gamma | accept] -> host-coherent memory
//   workgroups 1 .. nb_P        P+ -> host-coherent memory (want_P)
//   the others                  the device-side twin of orcvio_msckf_cov_commit (commit): S+ = sigma Z^T (or the prior's own
//                               factor) into the spare factor buffer, P+ over the resident covariance -- refused by every
//                               workgroup for itself if the update was (info[2..3] pivot counters of chol(M): k_finish_sqrt
//                               kept P; info[8] a hand-off inside a launch timed out: the results are garbage and the host
//                               re-runs the update; a non-finite dx, which no pivot test sees)
// Every wave that stores to host memory waits for its stores, the workgroup meets, its lane 0 makes them visible at system scope
// and counts itself in; the one that arrives last bumps the device-side sequence number and stores it to the host flag
// (release, system scope).  The host spins on that word.  The commit workgroups are not waited for: see below.
struct EpilogueArgs {
    const u32x4* small_src; u32x4* small_dst; size_t small16;
    const u32x4* P_src; u32x4* P_dst; size_t P16; int nb_P;
    // commit
    int commit;                   // 0: none, 1: P+ only, 2: P+ and the factor
    const double* Pout; double* Pres; size_t nn;
    const double* Z; int ldz, kf, n; double sigma;
    const double* prior; long sLi, sLj; double* Sout; int ldo;
    const double* dx; const int* info;
    const int* accept;            // optional (gated object update): *accept == 0 -> the update was rejected by its chi-square gate: P+ = P
                                  // stands in Pout already (k_finish_sqrt), the factor to keep is the prior's own
    int* counter;                 // device memory, zero between launches
    unsigned long long* seq;      // device memory: publications so far
    unsigned long long* flag;     // host-coherent memory: the caller waits for *flag >= its expected sequence number
    const int* info_also;         // optional: a second status block whose refusal refuses this commit too (the feature half's, for the chained object solve)
    int* info_keep;               // optional: a copy of info[0..15] for a LATER update's commit to look at (its info_also) once this one's status words
                                  // have been overwritten: the frame call's object half on the stream behind this launch
    int pub_all;                  // 1: the flag rises behind the COMMIT workgroups too (the frame call's chained object solve runs on a stream of its
                                  // own: whatever the caller does next on the handle's stream must find the commit done)
};
__global__ __launch_bounds__(256) void k_epilogue(EpilogueArgs a) {
    const int nb = gridDim.x, b = blockIdx.x, t = threadIdx.x;
    if (b == 0) {
        if (a.info_keep && t < 16) a.info_keep[t] = a.info[t];
        for (size_t i = t; i < a.small16; i += 256) a.small_dst[i] = a.small_src[i];
    } else if (b <= a.nb_P) {
        const size_t stride = (size_t)a.nb_P * 256;
        for (size_t i = (size_t)(b - 1) * 256 + t; i < a.P16; i += stride) a.P_dst[i] = a.P_src[i];
    } else if (a.commit) {
        __shared__ int s_bad;
        if (t == 0) s_bad = 0;
        __syncthreads();
        int bad = 0;
        for (int i = t; i < a.n; i += 256) { const double v = a.dx[i]; bad |= !(v - v == 0.0); }
        if (bad) s_bad = 1;
        __syncthreads();
        const bool also = a.info_also && (a.info_also[2] != 0 || a.info_also[3] != 0 || a.info_also[8] != 0);
        const bool refused = a.info[2] != 0 || a.info[3] != 0 || a.info[8] != 0 || s_bad != 0 || also;
        const bool applied = a.info[2] == 0 && a.info[3] == 0 && !also && (a.accept == nullptr || *a.accept != 0);
        const int nbc = nb - 1 - a.nb_P, bc = b - 1 - a.nb_P;
        const size_t stride = (size_t)nbc * 256;
        if (!refused)
            for (size_t i = (size_t)bc * 256 + t; i < a.nn; i += stride) a.Pres[i] = a.Pout[i];
        if (a.commit == 2) {
            const size_t kn = (size_t)a.kf * a.n;
            for (size_t idx = (size_t)bc * 256 + t; idx < kn; idx += stride) {
                const int i = (int)(idx / a.n), j = (int)(idx - (size_t)i * a.n);
                a.Sout[(size_t)i * a.ldo + j] = applied ? a.sigma * a.Z[(size_t)i * a.ldz + j] : a.prior[(long)j * a.sLi + (long)i * a.sLj];
            }
        }
    }
    // the flag rises when what the HOST reads is out (workgroups 0 .. nb_P); the commit workgroups write device memory only, which
    // whatever comes next on the stream sees by stream order -- the caller has its results 3-4 us before they are done
    const int npub = a.pub_all ? nb : 1 + a.nb_P;
    if (b >= npub) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) {
        int old = npub - 1;
        if (npub > 1) {   // (this workgroup's stores are out before it counts itself in)
            __threadfence_system();
            old = atomicAdd(a.counter, 1);
        } else {
            __threadfence_system();
        }
        if (old == npub - 1) {
            if (npub > 1) atomicExch(a.counter, 0);
            const unsigned long long v = atomicAdd(a.seq, 1ull) + 1ull;
            __hip_atomic_store(a.flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // (release: behind everything above)
        }
    }
}

// ---- helpers of the 2 x 2 block factorisation of windows beyond the register-resident kernels (capi_update.inc: blk2) ----
// lower factor (row-major, ld ldl) from the upper factor the register kernels write (R row-major ld ldr, X = R^T R): L = R^T, zero above
__global__ __launch_bounds__(256) void k_factor_to_lower(const double* __restrict__ Rf, int ldr, int n, double* __restrict__ L, int ldl) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * n) return;
    const int i = idx / n, j = idx - i * n;
    L[(size_t)i * ldl + j] = (j <= i) ? Rf[(size_t)j * ldr + i] : 0.0;
}
// rows [i0, i0 + rows) of the right-hand sides [B1 (strided) | bx] into Z
__global__ __launch_bounds__(256) void k_copy_rhs(const double* __restrict__ B1, long sB1i, long sB1c, int nc1, const double* __restrict__ bx, long sbx,
                                                  int i0, int rows, double* __restrict__ Z, int ldz) {
    const int nc = nc1 + 1;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * nc) return;
    const int r = idx / nc, c = idx - r * nc, i = i0 + r;
    Z[(size_t)i * ldz + c] = c < nc1 ? B1[(long)i * sB1i + (long)c * sB1c] : (bx ? bx[(long)i * sbx] : 0.0);
}
__global__ void k_info_merge(int* __restrict__ dst, const int* __restrict__ a) {   // dst[0..1] += a[0..1] (pivot counters of the two diagonal blocks)
    if (threadIdx.x < 2) dst[threadIdx.x] += a[threadIdx.x];
}

// Block [status | dof | accepted rows | accepted tracks | sequence number | rank + 1 | 0 ...] behind a rank's compressed block in the all-gather
// slot: what the other ranks must know about this rank's share (sharded calls; ORCVIO_ERR_PEER).  The sequence number is the ipc
// transport's update counter q (0 under RCCL): the receiving rank checks it against its own before it sums the slot.
#define ORCVIO_SHARD_META 16
#define ORCVIO_SHARD_META_SEQ 4
#define ORCVIO_SHARD_META_RANK 5   /* the sending rank + 1: the receiver counts the slots that carry their own number (ranks_seen) */
__global__ __launch_bounds__(64) void k_shard_meta(double* __restrict__ meta, int status, int dof, const int* __restrict__ accept,
                                                   const int* __restrict__ row_ptr, int F, unsigned long long q = 0ull, int rank = 0) {
    int rows = 0, cnt = 0;
    if (accept)
        for (int j = threadIdx.x; j < F; j += 64)
            if (accept[j]) { rows += row_ptr[j + 1] - row_ptr[j]; ++cnt; }
    for (int o = 32; o > 0; o >>= 1) { rows += __shfl_down(rows, o); cnt += __shfl_down(cnt, o); }
    if (threadIdx.x == 0) { meta[0] = (double)status; meta[1] = (double)dof; meta[2] = (double)rows; meta[3] = (double)cnt; }
    if (threadIdx.x >= 4 && threadIdx.x < ORCVIO_SHARD_META) meta[threadIdx.x] = threadIdx.x == ORCVIO_SHARD_META_SEQ ? (double)q : (threadIdx.x == ORCVIO_SHARD_META_RANK ? (double)(rank + 1) : 0.0);
}
}  // namespace orcvio_amd
