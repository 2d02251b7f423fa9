// CPU test of the ipc transport's protocol code (orcvio_amd/csrc/host/ipc_protocol.hpp -- the very functions capi_ipc.inc and
// k_gram_reduce run): 4 .. 8 threads stand in for ranks on ONE shared IpcHostSlots / gather buffer (multi-GPU form
// SURVEY.md 8e).  No GPU, no HIP.
//   1. rank-ordered sum: the same bits whichever "rank" forms it, equal to the documented order, and NOT the bits of a naive
//      left-to-right sum for blocks chosen to show it (the order is part of the contract)
//   2. two-generation slots: thousands of all-reduce rounds and dof sums with ranks running at different speeds -- every rank reads,
//      in every round, exactly the values of THAT round (a slot rewritten too early would show as a stale / future value)
//   3. a rank whose update counter has slipped is detected by its peers (value of another update), not summed
//   4. a rank that never arrives: the bounded wait reports WHICH rank
// Build: g++ -O2 -std=c++17 -pthread tests/cpp/test_ipc_protocol.cpp -o test_ipc_protocol
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <random>
#include <thread>
#include <vector>

#include "../../orcvio_amd/csrc/host/ipc_protocol.hpp"

using namespace orcvio_amd;

static bool spin(const std::function<bool()>& pred, double limit_s) {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned it = 0;; ++it) {
        if (pred()) return true;
        if ((it & 255u) == 255u) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit_s) return false;
            std::this_thread::yield();
        }
    }
}

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); ++fails; } } while (0)

static void test_rank_ordered_sum() {
    for (int world : {1, 2, 3, 4, 5, 7, 8, 16}) {
        const size_t n = 64, stride = n + 16;
        std::vector<double> parts(stride * world);
        std::mt19937_64 rng(world);
        std::uniform_real_distribution<double> U(-1.0, 1.0);
        for (auto& v : parts) v = U(rng) * std::pow(10.0, (double)(rng() % 12) - 6.0);   // (wildly different magnitudes: the order shows)
        bool differs_from_naive = false;
        for (size_t i = 0; i < n; ++i) {
            // the documented order, written out independently
            double s[4] = {0, 0, 0, 0};
            int c = 0;
            for (; c + 4 <= world; c += 4) for (int k = 0; k < 4; ++k) s[k] += parts[(size_t)(c + k) * stride + i];
            for (; c < world; ++c) s[0] += parts[(size_t)c * stride + i];
            const double want = (s[0] + s[1]) + (s[2] + s[3]);
            const double got = rank_ordered_sum(parts.data(), world, stride, i);
            CHECK(std::memcmp(&want, &got, 8) == 0);
            double naive = 0.0;
            for (int p = 0; p < world; ++p) naive += parts[(size_t)p * stride + i];
            if (std::memcmp(&naive, &got, 8) != 0) differs_from_naive = true;
        }
        if (world >= 5) CHECK(differs_from_naive);
    }
    // slots of the two generations never overlap, and generation q + 2 is generation q
    const size_t slot = 1000, gen = slot * 8;
    for (unsigned long long q = 1; q < 6; ++q)
        for (int r = 0; r < 8; ++r) {
            CHECK(ipc_slot_offset(q, r, slot, gen) == ipc_slot_offset(q + 2, r, slot, gen));
            const size_t o = ipc_slot_offset(q, r, slot, gen), g0 = (size_t)(q & 1ull) * gen;
            CHECK(o >= g0 && o + slot <= g0 + gen);   // inside its own generation: the other generation's slots are never touched
        }
}

static void test_rounds(int world, int rounds) {
    IpcHostSlots* s = new IpcHostSlots();
    std::memset((void*)s, 0, sizeof(*s));
    std::atomic<int> bad{0};
    std::vector<std::thread> th;
    for (int rank = 0; rank < world; ++rank)
        th.emplace_back([&, rank] {
            std::mt19937 rng(100 + rank);
            auto wait = [&](const std::function<bool()>& pred) { return spin(pred, 20.0); };
            unsigned long long dofq = 0;
            for (int it = 1; it <= rounds; ++it) {
                if ((rng() & 7u) == 0) std::this_thread::sleep_for(std::chrono::microseconds(rng() % 50));   // ranks at different speeds
                // all-reduce round `it`: rank r contributes it * 1000 + r in entry 0, -(it * 1000 + r) in entry 1
                double v[8] = {it * 1000.0 + rank, -(it * 1000.0 + rank), (double)it, 0, 0, 0, 0, 0};
                const int miss = ipc_slots_allreduce_max(s, rank, world, (unsigned long long)it, v, 3, wait);
                if (miss != -1 || v[0] != it * 1000.0 + (world - 1) || v[1] != -(it * 1000.0) || v[2] != (double)it) bad++;
                if (it % 3 == 0) {   // every third update is an object update: its own counter, stamped with the update's number
                    int total = -1;
                    const int r = ipc_slots_sum_dofs(s, rank, world, ++dofq, (unsigned long long)it, 10 * it + rank, &total, wait);
                    if (r != -1 || total != 10 * it * world + world * (world - 1) / 2) bad++;
                }
            }
        });
    for (auto& t : th) t.join();
    CHECK(bad.load() == 0);
    delete s;
}

static void test_slipped_counter_and_missing_rank() {
    const int world = 4;
    IpcHostSlots* s = new IpcHostSlots();
    std::memset((void*)s, 0, sizeof(*s));
    std::vector<int> res(world, 99);
    std::vector<std::thread> th;
    for (int rank = 0; rank < world; ++rank)
        th.emplace_back([&, rank] {
            auto wait = [&](const std::function<bool()>& pred) { return spin(pred, 5.0); };
            int total = 0;
            // rank 2 believes this is sharded update 8, the others 7: same dof round, another update
            res[rank] = ipc_slots_sum_dofs(s, rank, world, 1ull, rank == 2 ? 8ull : 7ull, 5, &total, wait);
        });
    for (auto& t : th) t.join();
    CHECK(res[0] == -2 - 2 && res[1] == -2 - 2 && res[3] == -2 - 2);   // the peers name rank 2
    CHECK(res[2] <= -2);                                                 // ... and rank 2 sees that ITS peers disagree with it
    // rank 3 never arrives: the bounded wait names it
    std::memset((void*)s, 0, sizeof(*s));
    std::vector<int> miss(3, 99);
    th.clear();
    for (int rank = 0; rank < 3; ++rank)
        th.emplace_back([&, rank] {
            auto wait = [&](const std::function<bool()>& pred) { return spin(pred, 0.2); };
            double v[8] = {1.0 * rank};
            miss[rank] = ipc_slots_allreduce_max(s, rank, world, 1ull, v, 1, wait);
        });
    for (auto& t : th) t.join();
    CHECK(miss[0] == 3 && miss[1] == 3 && miss[2] == 3);
    delete s;
}

int main() {
    test_rank_ordered_sum();
    for (int world : {4, 6, 8}) test_rounds(world, 3000);
    test_slipped_counter_and_missing_rank();
    std::printf(fails ? "test_ipc_protocol: %d FAILURES\n" : "test_ipc_protocol: ok\n", fails);
    return fails ? 1 : 0;
}
