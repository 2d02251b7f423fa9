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

// resident covariance <- P+ of the update that has just run, unless that update was refused: info[2..3] the pivot counters of
// chol(M) (k_finish_sqrt kept P), info[8] a hand-off inside a launch timed out (the results are garbage and the host re-runs the
// update), info[13] a non-finite dx (k_check_finite).  The device-side twin of orcvio_msckf_cov_commit's copy.
__global__ __launch_bounds__(256) void k_commit_copy(const double* __restrict__ src, double* __restrict__ dst, size_t n,
                                                     const int* __restrict__ info) {
    if (info[2] != 0 || info[3] != 0 || info[8] != 0 || info[13] != 0) return;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// dx finite?  One wavefront, in front of the commit kernels and the publication: *bad = 1 if any entry of dx is NaN / Inf
// (a NaN pivot does not show in the smallest-pivot test of chol(M)).
__global__ __launch_bounds__(64) void k_check_finite(const double* __restrict__ dx, int n, int* __restrict__ bad) {
    int b = 0;
    for (int i = threadIdx.x; i < n; i += 64) {
        const double v = dx[i];
        b |= !(v - v == 0.0);
    }
    const unsigned long long any = __ballot(b != 0);
    if (threadIdx.x == 0) *bad = any != 0ull ? 1 : 0;
}

// Results -> host-coherent memory, then the flag.  Workgroup b < nb_small copies the small block [info | dx | gamma |
// accept]; the others copy P+ (if wanted).  Every storing wave waits for its stores, the workgroup meets, its lane 0 makes
// them visible at system scope and counts itself in; the workgroup that arrives last bumps the device-side sequence
// number and stores it to the host flag (release, system scope).  The host spins on that word.
struct PublishArgs {
    const u32x4* small_src; u32x4* small_dst; size_t small16;
    const u32x4* P_src; u32x4* P_dst; size_t P16;
    int* counter;                 // device memory, zero between launches
    unsigned long long* seq;      // device memory: publications so far
    unsigned long long* flag;     // host-coherent memory: the caller waits for *flag == its expected sequence number
};
__global__ __launch_bounds__(256) void k_publish(PublishArgs a) {
    const int nb = gridDim.x;
    if (blockIdx.x == 0) {
        for (size_t i = threadIdx.x; i < a.small16; i += 256) a.small_dst[i] = a.small_src[i];
    }
    if (a.P16 > 0 && nb > 1 && blockIdx.x > 0) {
        const size_t stride = (size_t)(nb - 1) * 256;
        for (size_t i = (size_t)(blockIdx.x - 1) * 256 + threadIdx.x; i < a.P16; i += stride) a.P_dst[i] = a.P_src[i];
    } else if (a.P16 > 0 && nb == 1) {
        for (size_t i = threadIdx.x; i < a.P16; i += 256) a.P_dst[i] = a.P_src[i];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();
        const int old = atomicAdd(a.counter, 1);
        if (old == nb - 1) {
            atomicExch(a.counter, 0);
            const unsigned long long v = atomicAdd(a.seq, 1ull) + 1ull;
            __threadfence_system();
            __hip_atomic_store(a.flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
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
