// ipc_protocol.hpp -- the parts of the handle's second transport (capi_ipc.inc: HIP IPC + shared memory) that are plain
// host / element arithmetic, in a header without HIP so that the library AND a CPU test (tests/cpp/test_ipc_protocol.cpp: four to
// eight threads standing in for ranks) run the same code.  Multi-GPU form of the update: SURVEY.md 8e -- tracks
// dealt over ranks, ONE exchange of the compressed blocks, rank-ordered sum, replicated solve.
//   * the two-generation slot rule: update q uses generation q & 1 of every double-buffered slot.  A rank writes its slot of
//     update q + 2 (the same generation as q) only after it has seen every peer's q + 1, and a peer publishes q + 1 only after it has
//     finished reading q: no slot is rewritten while somebody still reads it.
//   * the host exchanges through the shared segment: all-reduce(max) of eight doubles, and the sum of the ranks' degrees of
//     freedom stamped with the sharded update they belong to (a rank whose counter has slipped is DETECTED, not summed).
//   * the rank-ordered sum of the gathered blocks (k_gram_reduce): the same order on every rank, hence the same bits.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>

#ifndef ORCVIO_IPC_MAX_RANKS
#define ORCVIO_IPC_MAX_RANKS 16
#endif
#if defined(__HIPCC__)
#define ORC_IPC_HD __host__ __device__
#else
#define ORC_IPC_HD
#endif

namespace orcvio_amd {

// first double of slot `rank` of update q in a gather buffer of two generations (gen_stride doubles apart, `slot` doubles per rank)
ORC_IPC_HD inline size_t ipc_slot_offset(unsigned long long q, int rank, size_t slot, size_t gen_stride) {
    return (size_t)(q & 1ull) * gen_stride + slot * (size_t)rank;
}

// entry `src` of the rank-ordered sum of `nparts` gathered blocks (part_stride doubles apart): four partial sums over the ranks
// c = 0, 4, 8 .. / 1, 5 .. / 2, 6 .. / 3, 7 .. combined as (s0 + s1) + (s2 + s3) -- a FIXED order, so every rank gets the same bits
ORC_IPC_HD inline double rank_ordered_sum(const double* parts, int nparts, size_t part_stride, size_t src) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int c = 0;
    for (; c + 4 <= nparts; c += 4) {   // four loads in flight; fixed summation order (deterministic)
        const double v0 = parts[(size_t)c * part_stride + src], v1 = parts[(size_t)(c + 1) * part_stride + src];
        const double v2 = parts[(size_t)(c + 2) * part_stride + src], v3 = parts[(size_t)(c + 3) * part_stride + src];
        s0 += v0; s1 += v1; s2 += v2; s3 += v3;
    }
    for (; c < nparts; ++c) s0 += parts[(size_t)c * part_stride + src];
    return (s0 + s1) + (s2 + s3);
}

// ---- host exchanges: double-buffered value slots + one sequence word per rank ----------------------------------------------------
struct IpcHostSlots {
    alignas(128) volatile unsigned long long red_seq[ORCVIO_IPC_MAX_RANKS];
    double red_val[2][ORCVIO_IPC_MAX_RANKS][8];
    alignas(128) volatile unsigned long long dof_seq[ORCVIO_IPC_MAX_RANKS];
    double dof_val[2][ORCVIO_IPC_MAX_RANKS];
    unsigned long long dof_q[2][ORCVIO_IPC_MAX_RANKS];   // the sharded update (seq) each dof_val belongs to
};

// all-reduce (max) number q of this rank (q = 1, 2, ..: the caller's own count, taken BEFORE anything that can fail rank-locally).
// wait(pred) spins on pred() with the caller's bound and returns false if it gave up.  Returns -1, or the rank that never arrived.
template <typename Wait>
inline int ipc_slots_allreduce_max(IpcHostSlots* s, int rank, int world, unsigned long long q, double* values, int count, Wait&& wait) {
    double* mine = s->red_val[q & 1ull][rank];
    for (int i = 0; i < 8; ++i) mine[i] = i < count ? values[i] : 0.0;
    __atomic_store_n(&s->red_seq[rank], q, __ATOMIC_RELEASE);
    for (int p = 0; p < world; ++p)
        if (!wait([&] { return __atomic_load_n(&s->red_seq[p], __ATOMIC_ACQUIRE) >= q; })) return p;
    for (int i = 0; i < count; ++i) {
        double m = s->red_val[q & 1ull][0][i];
        for (int p = 1; p < world; ++p) m = std::max(m, s->red_val[q & 1ull][p][i]);
        values[i] = m;
    }
    return -1;
}

// the sum of the ranks' degrees of freedom for object update number q (this rank's count of such updates), which is sharded update
// `useq` of the handle.  Returns -1 (sum in *total), p >= 0: rank p never arrived, -2 - p: rank p's value belongs to another update.
template <typename Wait>
inline int ipc_slots_sum_dofs(IpcHostSlots* s, int rank, int world, unsigned long long q, unsigned long long useq, int dof, int* total, Wait&& wait) {
    s->dof_val[q & 1ull][rank] = (double)dof;
    s->dof_q[q & 1ull][rank] = useq;
    __atomic_store_n(&s->dof_seq[rank], q, __ATOMIC_RELEASE);
    int sum = 0;
    for (int p = 0; p < world; ++p) {
        if (!wait([&] { return __atomic_load_n(&s->dof_seq[p], __ATOMIC_ACQUIRE) >= q; })) return p;
        if (s->dof_q[q & 1ull][p] != useq) return -2 - p;   // a value of another update: the ranks' counters disagree
        sum += (int)s->dof_val[q & 1ull][p];
    }
    *total = sum;
    return -1;
}

}  // namespace orcvio_amd
