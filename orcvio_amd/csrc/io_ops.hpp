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
__global__ __launch_bounds__(256) void k_ingest(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) dst[i] = __builtin_nontemporal_load(src + i);
}

// k_ingest with the derived index arrays made on the way: the observations grouped by clone (a stable counting sort of obs_clone;
// the sparse part of the compression reads every clone's rows contiguously) -- position of every observation in the clone-sorted
// order (clone_obs) and the first sparse row of every clone (clone_ptr = 2 x the exclusive prefix of the clone counts).  The host
// used to make them in two scalar passes over the observations (~6 us at 12 000 observations, on the critical path of a
// zero-copy update); here block 0 does it while the other blocks copy, out of the same pinned bytes.
//   block 0:  keys -> per-thread counts per clone (LDS) -> exclusive scan over the threads, per clone -> clone bases -> positions
//   blocks 1..: the three byte ranges of the arena that travel (the two derived arrays lie between them and are not copied)
// Limits (the caller checks them): nobs <= 255 * 256, N <= 64.  Dynamic LDS: 256 * 64 * 3 bytes + the keys as bytes.
struct IngestSortArgs {
    const u32x4* src[3]; u32x4* dst[3]; unsigned long long n16[3];
    const int* obs_clone_host;    // [nobs] in the pinned arena (16-byte aligned)
    int* clone_obs;               // [nobs] device
    int* clone_ptr;               // [N + 1] device
    int nobs, N;
};
#define INGEST_SORT_T 256
__host__ __device__ inline size_t ingest_sort_lds(int nobs) { return (size_t)INGEST_SORT_T * 64 * 3 + (((size_t)nobs + 15) & ~(size_t)15); }
__global__ __launch_bounds__(INGEST_SORT_T) void k_ingest_sort(IngestSortArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_sort[];
    if (blockIdx.x > 0) {
        const size_t stride = (size_t)(gridDim.x - 1) * INGEST_SORT_T;
        for (int q = 0; q < 3; ++q)
            for (size_t i = (size_t)(blockIdx.x - 1) * INGEST_SORT_T + threadIdx.x; i < a.n16[q]; i += stride) a.dst[q][i] = __builtin_nontemporal_load(a.src[q] + i);
        return;
    }
    unsigned char* cnt = smem_sort;                                                            // [T][64] observations of clone b among this thread's keys
    unsigned short* off = reinterpret_cast<unsigned short*>(smem_sort + INGEST_SORT_T * 64);   // [T][64] ... among the threads in front of it
    unsigned char* key = smem_sort + INGEST_SORT_T * 64 * 3;                                   // [nobs] the clone indices as bytes
    __shared__ int base[65];
    const int t = threadIdx.x;
    {   // the keys out of pinned memory once, 16 bytes per lane (a scalar read of host memory is a PCIe round trip each)
        const u32x4* k4 = reinterpret_cast<const u32x4*>(a.obs_clone_host);
        const int n4 = (a.nobs + 3) / 4;
        for (int i = t; i < n4; i += INGEST_SORT_T) {
            const u32x4 v = __builtin_nontemporal_load(k4 + i);
            key[4 * i] = (unsigned char)v.x; key[4 * i + 1] = (unsigned char)v.y; key[4 * i + 2] = (unsigned char)v.z; key[4 * i + 3] = (unsigned char)v.w;
        }
    }
    for (int b = 0; b < a.N; ++b) cnt[t * 64 + b] = 0;
    __syncthreads();
    const int per = (a.nobs + INGEST_SORT_T - 1) / INGEST_SORT_T;
    const int k0 = t * per, k1 = (k0 + per < a.nobs) ? k0 + per : a.nobs;
    for (int k = k0; k < k1; ++k) cnt[t * 64 + key[k]]++;
    __syncthreads();
    const int wave = t >> 6, l = t & 63;
    for (int b = wave; b < a.N; b += INGEST_SORT_T / 64) {   // exclusive scan of cnt[.][b] over the threads: 4 per lane, then across the lanes
        int c[4], s = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) { c[i] = cnt[(4 * l + i) * 64 + b]; s += c[i]; }
        int inc = s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o); if (l >= o) inc += v; }
        int run = inc - s;
#pragma unroll
        for (int i = 0; i < 4; ++i) { off[(4 * l + i) * 64 + b] = (unsigned short)run; run += c[i]; }
        if (l == 63) base[b + 1] = inc;   // observations of clone b
    }
    __syncthreads();
    if (t == 0) {
        base[0] = 0;
        for (int b = 0; b < a.N; ++b) base[b + 1] += base[b];
    }
    __syncthreads();
    if (t <= a.N) a.clone_ptr[t] = 2 * base[t];
    for (int k = k0; k < k1; ++k) {
        const int b = key[k];
        a.clone_obs[k] = base[b] + (int)off[t * 64 + b]++;
    }
}

// The epilogue of a zero-copy update, ONE launch behind k_finish_sqrt:
//   workgroup 0                 the small result block [info | dx | gamma | accept] -> host-coherent memory
//   workgroups 1 .. nb_P        P+ -> host-coherent memory (want_P)
//   the others                  the device-side twin of orcvio_msckf_cov_commit (commit): S+ = sigma Z^T (or the prior's own
//                               factor) into the spare factor buffer, P+ over the resident covariance -- refused by every
//                               workgroup for itself if the update was (info[2..3] pivot counters of chol(M): k_finish_sqrt
//                               kept P; info[8] a hand-off inside a launch timed out: the results are garbage and the host
//                               re-runs the update; a non-finite dx, which no pivot test sees)
// Every storing wave waits for its stores, the workgroup meets, its lane 0 makes them visible at system scope and counts
// itself in; the workgroup that arrives last bumps the device-side sequence number and stores it to the host flag (release,
// system scope).  The host spins on that word.
struct EpilogueArgs {
    const u32x4* small_src; u32x4* small_dst; size_t small16;
    const u32x4* P_src; u32x4* P_dst; size_t P16; int nb_P;
    // commit
    int commit;                   // 0: none, 1: P+ only, 2: P+ and the factor
    const double* Pout; double* Pres; size_t nn;
    const double* Z; int ldz, kf, n; double sigma;
    const double* prior; long sLi, sLj; double* Sout; int ldo;
    const double* dx; const int* info;
    int* counter;                 // device memory, zero between launches
    unsigned long long* seq;      // device memory: publications so far
    unsigned long long* flag;     // host-coherent memory: the caller waits for *flag >= its expected sequence number
};
__global__ __launch_bounds__(256) void k_epilogue(EpilogueArgs a) {
    const int nb = gridDim.x, b = blockIdx.x, t = threadIdx.x;
    if (b == 0) {
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
        const bool refused = a.info[2] != 0 || a.info[3] != 0 || a.info[8] != 0 || s_bad != 0;
        const bool applied = a.info[2] == 0 && a.info[3] == 0;
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) {
        int old = nb - 1;
        if (nb > 1) {   // (this workgroup's stores are out before it counts itself in)
            __threadfence_system();
            old = atomicAdd(a.counter, 1);
        }
        if (old == nb - 1) {
            if (nb > 1) atomicExch(a.counter, 0);
            const unsigned long long v = atomicAdd(a.seq, 1ull) + 1ull;
            __hip_atomic_store(a.flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // (release: behind everything above)
        }
    }
}

// Block [status | dof | accepted rows | accepted tracks | 0 ...] behind a rank's compressed block in the all-gather slot:
// what the other ranks must know about this rank's share (sharded calls; ORCVIO_ERR_PEER).
#define ORCVIO_SHARD_META 16
__global__ __launch_bounds__(64) void k_shard_meta(double* __restrict__ meta, int status, int dof, const int* __restrict__ accept,
                                                   const int* __restrict__ row_ptr, int F) {
    int rows = 0, cnt = 0;
    if (accept)
        for (int j = threadIdx.x; j < F; j += 64)
            if (accept[j]) { rows += row_ptr[j + 1] - row_ptr[j]; ++cnt; }
    for (int o = 32; o > 0; o >>= 1) { rows += __shfl_down(rows, o); cnt += __shfl_down(cnt, o); }
    if (threadIdx.x == 0) { meta[0] = (double)status; meta[1] = (double)dof; meta[2] = (double)rows; meta[3] = (double)cnt; }
    if (threadIdx.x >= 4 && threadIdx.x < ORCVIO_SHARD_META) meta[threadIdx.x] = 0.0;
}
}  // namespace orcvio_amd
