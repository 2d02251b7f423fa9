// stream_bench.cpp -- a stream of filter frames through the C-ABI from C++, no Python in the loop.
//
// The reference's caller is C++ (app/orcvioMain.cpp:106-198 -> OrcVIO::processFeatures, src/orcvio.cpp:567-594): per image
//   batchImuProcessing (covariance: :800-816) -> stateAugmentation (:962-1010) -> removeLostFeatures (:2497-2560)
//   -> pruneImuStateBuffer (update :2803-2851, marginalisation :2874-2956).
// This harness replays a pre-generated cycle of such frames (orcvio_amd/synth.py make_stream / write_stream: 19 / 20 clones,
// 20-200 ragged tracks, 12 in-state features, the prune update + marginalisation on the 20-clone frames) on the resident
// covariance, in one of two forms:
//   --mode calls   the separate C calls a caller makes today (cov_propagate, cov_augment, [cov_prefactor], io_begin + the tracks
//                  written into the arena + upload_slam_features + io_update(commit), the same for the prune update,
//                  cov_remove_clones)
//   --mode step    ONE call per frame (orcvio_msckf_io_step_frame)
// and prints one JSON line: frames/s, median / p95 per frame class, a checksum of every dx and of the final covariance (the two
// modes must agree bit for bit: tests/test_gpu_stream.py).
//
// Build: g++ -O2 -std=c++17 stream_bench.cpp -I include -L orcvio_amd/lib -lorcvio_msckf -Wl,-rpath,<lib dir>
#include <algorithm>
#include <chrono>
#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/orcvio_msckf.h"

struct Frame {
    int N = 0, F = 0, nobs = 0, has_prune = 0, F2 = 0, nobs2 = 0, n_remove = 0, remove[4] = {0, 0, 0, 0};
    std::vector<double> poses, p_w, obs_z, Phi, Q;
    std::vector<int32_t> obs_ptr, obs_clone;
    std::vector<int32_t> s_anchor, s_state, s_slot;
    std::vector<double> s_param, s_rho, s_pw, s_pfej, s_z, s_zvel;
    std::vector<double> p_w2, obs_z2;
    std::vector<int32_t> obs_ptr2, obs_clone2;
};

struct Stream {
    int n_frames = 0, n0 = 0, idp = 1, n_slam = 0;
    orcvio_msckf_flags flags{};
    std::vector<double> P0;
    std::vector<Frame> frames;
};

template <typename T>
static bool rd(FILE* f, std::vector<T>& v, size_t n) {
    v.resize(n);
    return n == 0 || fread(v.data(), sizeof(T), n, f) == n;
}

static bool load_stream(const char* path, Stream& s) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    char magic[8];
    if (fread(magic, 1, 8, f) != 8 || std::memcmp(magic, "ORCSTRM2", 8) != 0) { fclose(f); return false; }
    std::vector<int32_t> hi;
    std::vector<double> hd;
    bool ok = rd(f, hi, 10) && rd(f, hd, 2);
    if (!ok) { fclose(f); return false; }
    s.n_frames = hi[0]; s.n0 = hi[1]; s.idp = hi[2]; s.n_slam = hi[3];
    s.flags.leg_dim = hi[4]; s.flags.use_larvio = hi[5]; s.flags.use_left_perturbation = hi[6]; s.flags.if_fej = hi[7];
    s.flags.estimate_td = hi[8]; s.flags.discard_large_update = hi[9];
    s.flags.noise_feature = hd[0]; s.flags.chi2_prob = hd[1];
    ok = rd(f, s.P0, (size_t)s.n0 * s.n0);
    const int leg = s.flags.leg_dim, ns = s.n_slam;
    s.frames.resize(s.n_frames);
    for (int k = 0; k < s.n_frames && ok; ++k) {
        Frame& fr = s.frames[k];
        std::vector<int32_t> h;
        ok = rd(f, h, 11);
        if (!ok) break;
        fr.N = h[0]; fr.F = h[1]; fr.nobs = h[2]; fr.has_prune = h[3]; fr.F2 = h[4]; fr.nobs2 = h[5]; fr.n_remove = h[6];
        for (int i = 0; i < 4; ++i) fr.remove[i] = h[7 + i];
        ok = rd(f, fr.poses, (size_t)fr.N * ORCVIO_POSE_STRIDE) && rd(f, fr.p_w, (size_t)3 * fr.F) && rd(f, fr.obs_ptr, (size_t)fr.F + 1) &&
             rd(f, fr.obs_clone, (size_t)fr.nobs) && rd(f, fr.obs_z, (size_t)2 * fr.nobs) && rd(f, fr.s_anchor, ns) && rd(f, fr.s_state, ns) &&
             rd(f, fr.s_slot, ns) && rd(f, fr.s_param, (size_t)3 * ns) && rd(f, fr.s_rho, ns) && rd(f, fr.s_pw, (size_t)3 * ns) &&
             rd(f, fr.s_pfej, (size_t)3 * ns) && rd(f, fr.s_z, (size_t)2 * ns) && rd(f, fr.s_zvel, (size_t)2 * ns);
        if (ok && fr.has_prune)
            ok = rd(f, fr.p_w2, (size_t)3 * fr.F2) && rd(f, fr.obs_ptr2, (size_t)fr.F2 + 1) && rd(f, fr.obs_clone2, (size_t)fr.nobs2) &&
                 rd(f, fr.obs_z2, (size_t)2 * fr.nobs2);
        ok = ok && rd(f, fr.Phi, (size_t)leg * leg) && rd(f, fr.Q, (size_t)leg * leg);
    }
    fclose(f);
    return ok;
}

static uint64_t fnv(uint64_t h, const void* p, size_t bytes) {
    const unsigned char* b = static_cast<const unsigned char*>(p);
    for (size_t i = 0; i < bytes; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

#define CHK(call)                                                                                          \
    do {                                                                                                   \
        const int32_t rc_ = (call);                                                                        \
        if (rc_ != ORCVIO_OK) {                                                                            \
            std::fprintf(stderr, "%s -> %d (%s)\n", #call, (int)rc_, orcvio_msckf_last_error());           \
            return rc_;                                                                                    \
        }                                                                                                  \
    } while (0)

struct FrameOut { std::vector<double> dx1, dx2; int32_t st1[8], st2[8]; };

// the separate calls, as a C++ caller of the round-5 ABI makes them
static int32_t frame_calls(orcvio_msckf_handle* h, const Stream& s, const Frame& fr, bool prefactor, FrameOut& out) {
    const int leg = s.flags.leg_dim;
    CHK(orcvio_msckf_cov_propagate(h, leg, fr.Phi.data(), fr.Q.data()));
    CHK(orcvio_msckf_cov_augment(h));
    if (prefactor) CHK(orcvio_msckf_cov_prefactor(h));
    orcvio_msckf_io io{};
    CHK(orcvio_msckf_io_begin(h, &s.flags, fr.N, fr.F, fr.nobs, 0, &io));
    std::memcpy(io.poses, fr.poses.data(), sizeof(double) * fr.poses.size());
    std::memcpy(io.obs_ptr, fr.obs_ptr.data(), sizeof(int32_t) * fr.obs_ptr.size());
    if (fr.F) {
        std::memcpy(io.p_w, fr.p_w.data(), sizeof(double) * fr.p_w.size());
        std::memcpy(io.obs_clone, fr.obs_clone.data(), sizeof(int32_t) * fr.obs_clone.size());
        std::memcpy(io.obs_z, fr.obs_z.data(), sizeof(double) * fr.obs_z.size());
    }
    orcvio_msckf_slam_features sl{};
    sl.n_features = s.n_slam; sl.idp_dim = s.idp; sl.anchor = fr.s_anchor.data(); sl.state = fr.s_state.data(); sl.slot = fr.s_slot.data();
    sl.param = fr.s_param.data(); sl.inv_depth = fr.s_rho.data(); sl.p_w = fr.s_pw.data(); sl.p_fej = fr.s_pfej.data(); sl.z = fr.s_z.data();
    sl.z_vel = fr.s_zvel.data();
    if (s.n_slam > 0) CHK(orcvio_msckf_upload_slam_features(h, &sl));
    CHK(orcvio_msckf_io_update(h, 0, 1, out.st1));
    out.dx1.assign(io.dx, io.dx + io.n);
    out.dx2.clear();
    if (fr.has_prune) {
        CHK(orcvio_msckf_io_begin(h, &s.flags, fr.N, fr.F2, fr.nobs2, 0, &io));
        std::memcpy(io.poses, fr.poses.data(), sizeof(double) * fr.poses.size());
        std::memcpy(io.obs_ptr, fr.obs_ptr2.data(), sizeof(int32_t) * fr.obs_ptr2.size());
        std::memcpy(io.p_w, fr.p_w2.data(), sizeof(double) * fr.p_w2.size());
        std::memcpy(io.obs_clone, fr.obs_clone2.data(), sizeof(int32_t) * fr.obs_clone2.size());
        std::memcpy(io.obs_z, fr.obs_z2.data(), sizeof(double) * fr.obs_z2.size());
        CHK(orcvio_msckf_io_update(h, 0, 1, out.st2));
        out.dx2.assign(io.dx, io.dx + io.n);
    }
    if (fr.n_remove > 0) CHK(orcvio_msckf_cov_remove_clones(h, leg, fr.remove, fr.n_remove));
    return ORCVIO_OK;
}

#ifdef ORCVIO_HAVE_STEP_FRAME
// ONE call per frame
static int32_t frame_step(orcvio_msckf_handle* h, const Stream& s, const Frame& fr, bool apply_dx, FrameOut& out) {
    const int leg = s.flags.leg_dim;
    orcvio_msckf_io io{};
    CHK(orcvio_msckf_io_begin(h, &s.flags, fr.N, fr.F, fr.nobs, 2, &io));   // (2: the resident covariance AFTER this frame's augmentation)
    std::memcpy(io.poses, fr.poses.data(), sizeof(double) * fr.poses.size());
    std::memcpy(io.obs_ptr, fr.obs_ptr.data(), sizeof(int32_t) * fr.obs_ptr.size());
    if (fr.F) {
        std::memcpy(io.p_w, fr.p_w.data(), sizeof(double) * fr.p_w.size());
        std::memcpy(io.obs_clone, fr.obs_clone.data(), sizeof(int32_t) * fr.obs_clone.size());
        std::memcpy(io.obs_z, fr.obs_z.data(), sizeof(double) * fr.obs_z.size());
    }
    orcvio_msckf_slam_features sl{};
    sl.n_features = s.n_slam; sl.idp_dim = s.idp; sl.anchor = fr.s_anchor.data(); sl.state = fr.s_state.data(); sl.slot = fr.s_slot.data();
    sl.param = fr.s_param.data(); sl.inv_depth = fr.s_rho.data(); sl.p_w = fr.s_pw.data(); sl.p_fej = fr.s_pfej.data(); sl.z = fr.s_z.data();
    sl.z_vel = fr.s_zvel.data();
    orcvio_msckf_tracks pt{};
    pt.n_features = fr.F2; pt.p_w = fr.p_w2.data(); pt.obs_ptr = fr.obs_ptr2.data(); pt.obs_clone = fr.obs_clone2.data(); pt.obs_z = fr.obs_z2.data();
    orcvio_msckf_frame_step st{};
    st.leg_dim = leg; st.Phi = fr.Phi.data(); st.Q = fr.Q.data(); st.augment = 1;
    st.slam_features = s.n_slam > 0 ? &sl : nullptr;
    st.prune_tracks = fr.has_prune ? &pt : nullptr;
    st.prune_apply_dx = apply_dx ? 1 : 0;
    st.remove_clones = fr.n_remove > 0 ? fr.remove : nullptr; st.n_remove = fr.n_remove;
    orcvio_msckf_frame_result res{};
    CHK(orcvio_msckf_io_step_frame(h, &st, &res));
    std::memcpy(out.st1, res.stats, sizeof(out.st1));
    std::memcpy(out.st2, res.prune_stats, sizeof(out.st2));
    out.dx1.assign(res.dx, res.dx + io.n);
    out.dx2.clear();
    if (fr.has_prune) out.dx2.assign(res.prune_dx, res.prune_dx + io.n);
    return ORCVIO_OK;
}
#endif

static double pct(std::vector<double> v, double q) {
    if (v.empty()) return 0.0;
    std::sort(v.begin(), v.end());
    const double pos = q * (v.size() - 1);
    const size_t i = (size_t)pos;
    const double fr = pos - i;
    return i + 1 < v.size() ? v[i] * (1 - fr) + v[i + 1] * fr : v[i];
}

int main(int argc, char** argv) {
    std::string path, mode = "calls", dump;
    int frames = 480, warm = 32, device = 0;
    bool prefactor = true, apply_dx = false, sync_each = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() { return i + 1 < argc ? std::string(argv[++i]) : std::string(); };
        if (a == "--stream") path = next();
        else if (a == "--mode") mode = next();
        else if (a == "--frames") frames = atoi(next().c_str());
        else if (a == "--warmup") warm = atoi(next().c_str());
        else if (a == "--device") device = atoi(next().c_str());
        else if (a == "--no-prefactor") prefactor = false;
        else if (a == "--apply-dx") apply_dx = true;
        else if (a == "--sync-each") sync_each = true;
        else if (a == "--dump") dump = next();
        else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    Stream s;
    if (path.empty() || !load_stream(path.c_str(), s)) { std::fprintf(stderr, "stream_bench: cannot read --stream %s\n", path.c_str()); return 2; }
    orcvio_msckf_handle* h = nullptr;
    CHK(orcvio_msckf_create(device, 24, 256, 4096, &h));
    CHK(orcvio_msckf_set_option(h, ORCVIO_OPT_EXTRA_STATES, s.idp * s.n_slam));
    CHK(orcvio_msckf_set_option(h, ORCVIO_OPT_EKF_ROWS, 1));
    CHK(orcvio_msckf_cov_set(h, s.n0, s.P0.data()));
    std::vector<double> t_all, t_prune, t_plain;
    std::vector<std::vector<double>> per_frame(s.n_frames);
    uint64_t hash = 1469598103934665603ull;
    FrameOut out;
    FILE* df = dump.empty() ? nullptr : fopen(dump.c_str(), "wb");
    int n_updates = 0, discards = 0;
    auto t_loop0 = std::chrono::steady_clock::now();
    for (int it = 0; it < frames + warm; ++it) {
        const Frame& fr = s.frames[it % s.n_frames];
        const auto t0 = std::chrono::steady_clock::now();
        int32_t rc;
        if (mode == "calls") rc = frame_calls(h, s, fr, prefactor, out);
#ifdef ORCVIO_HAVE_STEP_FRAME
        else if (mode == "step") rc = frame_step(h, s, fr, apply_dx, out);
#endif
        else { std::fprintf(stderr, "unknown --mode %s\n", mode.c_str()); return 2; }
        if (rc != ORCVIO_OK) return rc;
        if (sync_each) CHK(orcvio_msckf_sync(h, nullptr));   // (the results are in host memory when the calls return; what is still in flight -- the marginalisation -- is ordered in front of the next frame by the stream)
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (it == warm - 1) { CHK(orcvio_msckf_sync(h, nullptr)); t_loop0 = std::chrono::steady_clock::now(); }
        n_updates += 1 + (fr.has_prune ? 1 : 0);
        discards += out.st1[4];
        for (double v : out.dx1) if (!std::isfinite(v)) { std::fprintf(stderr, "non-finite dx in frame %d\n", it); return 3; }
        hash = fnv(hash, out.dx1.data(), sizeof(double) * out.dx1.size());
        hash = fnv(hash, out.dx2.data(), sizeof(double) * out.dx2.size());
        if (df) {
            const int32_t l1 = (int32_t)out.dx1.size(), l2 = (int32_t)out.dx2.size();
            fwrite(&l1, 4, 1, df); fwrite(out.dx1.data(), 8, l1, df); fwrite(&l2, 4, 1, df); fwrite(out.dx2.data(), 8, l2, df);
        }
        if (it >= warm) {
            t_all.push_back(ms);
            (fr.has_prune ? t_prune : t_plain).push_back(ms);
            per_frame[it % s.n_frames].push_back(ms);
        }
    }
    CHK(orcvio_msckf_sync(h, nullptr));   // (inside the timed loop: frames_per_s counts everything the frames enqueued)
    const double loop_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_loop0).count();
    int32_t n_end = 0;
    CHK(orcvio_msckf_cov_get(h, &n_end, nullptr));
    std::vector<double> P((size_t)n_end * n_end);
    CHK(orcvio_msckf_cov_get(h, &n_end, P.data()));
    const uint64_t hash_P = fnv(1469598103934665603ull, P.data(), sizeof(double) * P.size());
    if (df) { fwrite(&n_end, 4, 1, df); fwrite(P.data(), 8, P.size(), df); fclose(df); }
    int64_t cnt[ORCVIO_COUNTERS] = {0};
    (void)orcvio_msckf_counters(h, cnt, ORCVIO_COUNTERS);
    double mean = 0.0;
    for (double v : t_all) mean += v;
    mean /= t_all.empty() ? 1 : t_all.size();
    double worst_jit = 0.0;
    for (auto& v : per_frame) if (v.size() >= 8) worst_jit = std::max(worst_jit, pct(v, 0.95) / pct(v, 0.5));
    std::printf("{\"harness\": \"tests/cpp/stream_bench.cpp\", \"mode\": \"%s\", \"prefactor\": %d, \"apply_dx\": %d, \"frames\": %d, \"frames_per_s\": %.1f, "
                "\"mean_ms\": %.5f, \"median_ms\": %.5f, \"p95_ms\": %.5f, "
                "\"with_prune\": {\"median_ms\": %.5f, \"p95_ms\": %.5f, \"p95_over_median\": %.4f, \"n\": %zu}, "
                "\"without_prune\": {\"median_ms\": %.5f, \"p95_ms\": %.5f, \"p95_over_median\": %.4f, \"n\": %zu}, "
                "\"worst_p95_over_median_same_frame\": %.4f, \"updates_per_frame\": %.3f, \"large_update_flags\": %d, \"front_fallbacks\": %" PRId64 ", "
                "\"graph_captures\": %" PRId64 ", \"graph_replays\": %" PRId64 ", \"plain_runs\": %" PRId64 ", "
                "\"n_end\": %d, \"dx_hash\": \"%016" PRIx64 "\", \"P_hash\": \"%016" PRIx64 "\"}\n",
                mode.c_str(), prefactor ? 1 : 0, apply_dx ? 1 : 0, frames, frames / loop_s, mean, pct(t_all, 0.5), pct(t_all, 0.95),
                pct(t_prune, 0.5), pct(t_prune, 0.95), t_prune.empty() ? 0.0 : pct(t_prune, 0.95) / pct(t_prune, 0.5), t_prune.size(),
                pct(t_plain, 0.5), pct(t_plain, 0.95), t_plain.empty() ? 0.0 : pct(t_plain, 0.95) / pct(t_plain, 0.5), t_plain.size(),
                worst_jit, (double)n_updates / (frames + warm), discards, cnt[0], cnt[1], cnt[2], cnt[3], n_end, hash, hash_P);
    orcvio_msckf_destroy(h);
    return 0;
}
