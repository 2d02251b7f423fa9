// GPU test of the C++ host mirror end to end: std::map containers -> MsckfBackend::msckfUpdate (HIP path
// through the C-ABI) against the plain-C oracle on the same window.  Also the pruneImuStateBuffer variant.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../../orcvio_amd/csrc/host/orcvio_msckf_host.hpp"

extern "C" int orc_oracle_msckf_update(int N, int F, const int* flags, double sigma, double chi2_prob, const double* chi2_table,
                                       int chi2_table_len, const double* R_b2w, const double* t_b_w, const double* t_fej,
                                       const double* R_b2c, const double* t_c_b, const double* p_w, const int* obs_ptr,
                                       const int* obs_clone, const double* obs_z, const double* obs_zvel, const int* clone_mask,
                                       const double* P, double* dx, double* P_out, int* accept, double* gamma, double* H_all,
                                       double* r_all, int* block_ptr, double* H_thin, double* r_thin, double* K, double* G, int* info);
extern "C" double orc_oracle_chi2_quantile(int dof, double p);

using namespace orcvio_amd;

static void rotz(double a, double* R) { R[0] = cos(a); R[1] = -sin(a); R[2] = 0; R[3] = sin(a); R[4] = cos(a); R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1; }

static double relerr(const std::vector<double>& a, const std::vector<double>& b) {
    double d = 0, n = 0;
    for (size_t i = 0; i < a.size(); ++i) { d += (a[i] - b[i]) * (a[i] - b[i]); n += b[i] * b[i]; }
    return std::sqrt(d / (n > 0 ? n : 1));
}

// where the threads of this process sit (a bounded wait of the library has fired): name, kernel wait channel, system call
#include <dirent.h>
#include <string>
static void dump_threads() {
    DIR* d = opendir("/proc/self/task");
    if (!d) return;
    while (dirent* e = readdir(d)) {
        if (e->d_name[0] == '.') continue;
        const std::string base = std::string("/proc/self/task/") + e->d_name + "/";
        char buf[3][256] = {{0}, {0}, {0}};
        const char* files[3] = {"comm", "wchan", "syscall"};
        for (int i = 0; i < 3; ++i) {
            FILE* f = fopen((base + files[i]).c_str(), "r");
            if (f) { if (fgets(buf[i], 255, f)) { char* nl = strchr(buf[i], '\n'); if (nl) *nl = 0; } fclose(f); }
        }
        std::printf("thread %s comm=%s wchan=%s syscall=%s\n", e->d_name, buf[0], buf[1], buf[2]);
    }
    closedir(d);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);   // (the harness reports how far the run got if it ever hangs)
    const int N = 8, F = 40;
    std::mt19937 rng(3);
    std::normal_distribution<double> G(0, 1);
    std::uniform_real_distribution<double> U(0, 1);
    StateServer ss;
    for (int i = 0; i < N; ++i) {
        IMUState_Aug a; a.id = 100 + 2 * i;
        rotz(0.03 * i, a.orientation);
        a.position[0] = 0.2 * i; a.position[1] = 0.05 * std::sin(0.4 * i); a.position[2] = 0.02 * i;
        for (int k = 0; k < 3; ++k) a.position_FEJ[k] = a.position[k];
        // camera looks along body x: R_b2c rows = (cam x = -body y, cam y = -body z, cam z = body x)
        const double Rbc[9] = {0, -1, 0, 0, 0, -1, 1, 0, 0};
        std::memcpy(a.R_imu_cam0, Rbc, sizeof(Rbc));
        a.t_cam0_imu[0] = 0.05; a.t_cam0_imu[1] = 0.02; a.t_cam0_imu[2] = -0.01;
        ss.imu_states_augment[a.id] = a;
    }
    ss.imu_state = IMUState();
    std::memcpy(ss.imu_state.R_imu_cam0, ss.imu_states_augment.begin()->second.R_imu_cam0, 72);
    std::memcpy(ss.imu_state.t_cam0_imu, ss.imu_states_augment.begin()->second.t_cam0_imu, 24);
    const int n = 22 + 6 * N;
    ss.state_cov.assign((size_t)n * n, 0.0);
    {   // SPD prior with the extrinsic / td block zeroed
        std::vector<double> A((size_t)n * n);
        for (auto& v : A) v = G(rng) / std::sqrt((double)n);
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double s = 0;
                for (int k = 0; k < n; ++k) s += A[(size_t)i * n + k] * A[(size_t)j * n + k];
                ss.state_cov[(size_t)i * n + j] = 1e-4 * s + (i == j ? 1e-3 : 0.0);
            }
        for (int i = 15; i < 22; ++i)
            for (int j = 0; j < n; ++j) ss.state_cov[(size_t)i * n + j] = ss.state_cov[(size_t)j * n + i] = 0.0;
    }
    MapServer map_server;
    std::vector<FeatureIDType> ids;
    for (int j = 0; j < F; ++j) {
        Feature f; f.id = 1000 + 3 * j;
        const double pt[3] = {6.0 + 6.0 * U(rng), -2.0 + 4.0 * U(rng), -1.0 + 2.0 * U(rng)};
        const int M = 3 + (int)(U(rng) * (N - 3)), s0 = (int)(U(rng) * (N - M + 1));
        int i = 0;
        for (auto& kv : ss.imu_states_augment) {
            if (i >= s0 && i < s0 + M) {
                const IMUState_Aug& a = kv.second;
                double tcw[3], d[3], pc[3];
                for (int k = 0; k < 3; ++k) tcw[k] = a.position[k] + a.orientation[k * 3] * a.t_cam0_imu[0] + a.orientation[k * 3 + 1] * a.t_cam0_imu[1] + a.orientation[k * 3 + 2] * a.t_cam0_imu[2];
                for (int k = 0; k < 3; ++k) d[k] = pt[k] - tcw[k];
                double Rwc[9];
                for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Rwc[r * 3 + c] = a.R_imu_cam0[r * 3] * a.orientation[c * 3] + a.R_imu_cam0[r * 3 + 1] * a.orientation[c * 3 + 1] + a.R_imu_cam0[r * 3 + 2] * a.orientation[c * 3 + 2];
                for (int k = 0; k < 3; ++k) pc[k] = Rwc[k * 3] * d[0] + Rwc[k * 3 + 1] * d[1] + Rwc[k * 3 + 2] * d[2];
                const double noise = (j % 7 == 0) ? 0.1 : 0.008;   // every 7th track is an outlier
                f.observations[kv.first] = {pc[0] / pc[2] + noise * G(rng), pc[1] / pc[2] + noise * G(rng)};
                f.observations_vel[kv.first] = {0.01 * G(rng), 0.01 * G(rng)};
            }
            ++i;
        }
        for (int k = 0; k < 3; ++k) f.position[k] = pt[k] + 0.02 * G(rng);
        map_server[f.id] = f;
        ids.push_back(f.id);
    }
    MsckfBackend be(0, 16, 256, 8192);
    int fails = 0;
    for (int variant = 0; variant < 2; ++variant) {
        std::vector<StateIDType> only;
        if (variant == 1) { auto it = ss.imu_states_augment.begin(); ++it; only.push_back(it->first); ++it; only.push_back(it->first); }
        // oracle on the same flat data
        std::vector<double> R, t, tf, Rc, tc, p_w, z, zv;
        std::vector<int32_t> ptr, clone;
        std::map<StateIDType, int> index_of;
        MsckfBackend::flattenWindow(ss, R, t, tf, Rc, tc, index_of);
        MsckfBackend::flattenTracks(map_server, ids, index_of, only, p_w, ptr, clone, z, zv);
        std::vector<double> table(500, 0.0);
        for (int d = 1; d < 500; ++d) table[d] = orc_oracle_chi2_quantile(d, 0.95);
        const int fl[5] = {22, 1, 0, 0, 0};
        std::vector<double> dx(n), Pn((size_t)n * n), gam(F), Gm((size_t)n * n), Ht((size_t)n * n), rt(n);
        std::vector<int> acc(F);
        int info[2];
        StateServer ss_copy = ss;
        int rc = orc_oracle_msckf_update(N, F, fl, 0.008, 0.95, table.data(), 500, R.data(), t.data(), tf.data(), Rc.data(), tc.data(),
                                         p_w.data(), ptr.data(), clone.data(), z.data(), zv.data(), nullptr, ss.state_cov.data(), dx.data(),
                                         Pn.data(), acc.data(), gam.data(), nullptr, nullptr, nullptr, Ht.data(), rt.data(), nullptr, Gm.data(), info);
        UpdateOutcome out = be.msckfUpdate(ss_copy, map_server, ids, only);
        if (rc != 0 || out.status != ORCVIO_OK) { std::printf("status oracle %d gpu %d (%s)\n", rc, out.status, orcvio_msckf_last_error()); return 1; }
        int nacc = 0, same = 1;
        for (int j = 0; j < F; ++j) { nacc += acc[j]; same &= (acc[j] == out.accepted[j]); }
        const double e_dx = relerr(out.delta_x, dx), e_P = relerr(ss_copy.state_cov, Pn);
        std::printf("variant %d: accepted %d/%d same mask %d  dx rel %.2e  P rel %.2e  incremented %d\n", variant, nacc, F, same, e_dx, e_P,
                    (int)out.state_incremented);
        if (!same || !(e_dx < 1e-6) || !(e_P < 1e-6) || nacc == 0 || (variant == 0 && nacc == F)) ++fails;
        // the state was incremented: clone 0 position moved by delta_x[22+3..22+5]
        const IMUState_Aug& c0 = ss_copy.imu_states_augment.begin()->second;
        const IMUState_Aug& b0 = ss.imu_states_augment.begin()->second;
        for (int k = 0; k < 3; ++k)
            if (std::fabs((c0.position[k] - b0.position[k]) - out.delta_x[22 + 3 + k]) > 1e-12) ++fails;
    }
    {   // triangulation through the host mirror: positions wiped, recovered from the observations alone
        MapServer m2 = map_server;
        for (auto& kv : m2) { kv.second.is_initialized = false; for (int k = 0; k < 3; ++k) kv.second.position[k] = 0.0; }
        int st = ORCVIO_OK;
        // this scene moves the camera ALONG its viewing direction (little parallax): the reference's 0.2 m motion test
        // rejects most tracks, so the threshold is lowered to exercise the solver; depth is then only loosely determined
        be.optimization_config.translation_threshold = 0.01;
        const StateIDType curr_id = ss.imu_states_augment.rbegin()->first;   // observations of the newest clone are left out
        std::vector<bool> ok = be.initializePositions(ss, m2, ids, curr_id, &st);
        int nok = 0;
        double worst = 0.0;
        for (size_t k = 0; k < ids.size(); ++k) {
            if (!ok[k]) continue;
            ++nok;
            const Feature& a = m2.at(ids[k]);
            const Feature& b = map_server.at(ids[k]);
            double d = 0;
            for (int c = 0; c < 3; ++c) d += (a.position[c] - b.position[c]) * (a.position[c] - b.position[c]);
            if (k % 7 != 0 && std::sqrt(d) > worst) worst = std::sqrt(d);
            if (!a.is_initialized || a.id_anchor < 0 || !(a.invDepth > 0)) ++fails;
        }
        std::printf("triangulation: status %d, %d/%d valid, worst inlier distance to the generator's point %.3f m\n", st, nok, F, worst);
        if (st != ORCVIO_OK || nok < F / 2) ++fails;   // (positions are checked against the oracle in tests/test_gpu_triangulate.py)
    }
    {   // the same two call sites with the covariance resident in HBM, through the SHARDED entry points (communicator of one
        // rank: the RCCL all-gather and the rank-ordered sum run, the result must equal the plain call), then the prune
        // update on the factor the first update left, then the rows / columns of the pruned clones dropped on the device
        StateServer a = ss, b = ss;
        UpdateOutcome o1 = be.msckfUpdate(a, map_server, ids);                   // host covariance, single GPU
        std::vector<StateIDType> only;
        { auto it = a.imu_states_augment.begin(); only.push_back(it->first); ++it; only.push_back(it->first); }
        UpdateOutcome o2 = be.msckfUpdate(a, map_server, ids, only);
        MsckfBackend sh(0, 16, 256, 8192);
        // ORCVIO_TEST_SKIP_COMM: the harness's second attempt after a first one that sat in the creation of the communicator (a
        // second process initialising RCCL on a GPU whose first process holds a communicator): the same two updates through
        // the plain entry point, resident covariance all the same
        const bool skip_comm = std::getenv("ORCVIO_TEST_SKIP_COMM") != nullptr;
        int rc = ORCVIO_OK;
        if (!skip_comm) {
            std::printf("sharded: creating the communicator\n");
            try {
                rc = sh.commInit(MsckfBackend::commUniqueId(), 0, 1);
            } catch (const std::exception& e) {
                std::printf("sharded: %s\n", e.what());
                rc = ORCVIO_ERR_TIMEOUT;
            }
            std::printf("sharded: communicator ready (rc %d)\n", rc);
            if (rc == ORCVIO_ERR_TIMEOUT) {   // the library's bounded wait fired (ORCVIO_COMM_TIMEOUT_S): say where the threads sit
                std::printf("sharded: %s\n", orcvio_msckf_last_error());
                dump_threads();
                return 77;
            }
        } else std::printf("sharded: SKIPPED (ORCVIO_TEST_SKIP_COMM), plain entry point on the resident covariance\n");
        if (rc == ORCVIO_OK) rc = sh.covarianceToDevice(b);
        UpdateOutcome s1 = skip_comm ? sh.msckfUpdate(b, map_server, ids) : sh.msckfUpdateSharded(b, map_server, ids);   // resident covariance, sharded path
        UpdateOutcome s2 = skip_comm ? sh.msckfUpdate(b, map_server, ids, only) : sh.msckfUpdateSharded(b, map_server, ids, only);   // ... on the resident factor of the first
        if (rc == ORCVIO_OK) rc = sh.removeClonesFromCovariance(b, only);
        if (rc == ORCVIO_OK) rc = sh.covarianceToHost(b);
        if (rc != ORCVIO_OK || s1.status != ORCVIO_OK || s2.status != ORCVIO_OK) { std::printf("sharded / resident: rc %d %d %d (%s)\n", rc, s1.status, s2.status, orcvio_msckf_last_error()); ++fails; }
        // host side of the marginalisation for the comparison (src/orcvio.cpp:2935-2951)
        const int n2 = n - 12;
        std::vector<double> Pm((size_t)n2 * n2);
        for (int i = 0, ii = 0; i < n; ++i) {
            if (i >= 22 && i < 34) continue;
            for (int j = 0, jj = 0; j < n; ++j) {
                if (j >= 22 && j < 34) continue;
                Pm[(size_t)ii * n2 + jj] = a.state_cov[(size_t)i * n + j];
                ++jj;
            }
            ++ii;
        }
        const double e1 = relerr(s1.delta_x, o1.delta_x), e2 = relerr(s2.delta_x, o2.delta_x), eP = relerr(b.state_cov, Pm);
        int same = (int)(s1.accepted == o1.accepted && s2.accepted == o2.accepted);
        std::printf("resident + sharded (world 1): dx rel %.2e / %.2e, P rel after marginalisation %.2e, same masks %d\n", e1, e2, eP, same);
        if (!(e1 < 1e-9) || !(e2 < 1e-6) || !(eP < 1e-6) || !same) ++fails;
        // dealing: every id lands on exactly one rank, loads balanced
        const std::vector<int> owner = MsckfBackend::dealFeatures(map_server, ids, 4);
        long load[4] = {0, 0, 0, 0};
        for (size_t k = 0; k < ids.size(); ++k) load[owner[k]] += 2 * (long)map_server.at(ids[k]).observations.size() - 3;
        long lo = load[0], hi = load[0];
        for (long v : load) { lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
        if (hi - lo > 13) { std::printf("dealFeatures: unbalanced %ld..%ld\n", lo, hi); ++fails; }
    }
    {   // one frame: the feature update, then processObjects' update (System.cpp:548-554), covariance resident -- the two call
        // sites one behind the other against MsckfBackend::frameUpdate (ONE library call, the objects' compression beside the
        // features' solve): same corrections, same covariance
        const int K = 4, FR = 6;
        const double centre[3] = {9.0, 0.4, 0.1}, shape[3] = {0.9, 0.6, 0.5};
        double wTo[16] = {1, 0, 0, centre[0], 0, 1, 0, centre[1], 0, 0, 1, centre[2], 0, 0, 0, 1};
        const double kps[K * 3] = {0.6, 0.3, 0.2, -0.5, 0.35, 0.1, 0.4, -0.3, -0.25, -0.45, -0.2, 0.3};
        std::vector<double> wTc((size_t)FR * 16), zs((size_t)FR * K * 2), bb((size_t)FR * 4);
        std::vector<int32_t> fclone(FR);
        int i = 0, fidx = 0;
        for (auto& kv : ss.imu_states_augment) {
            if (i >= 1 && fidx < FR) {
                const IMUState_Aug& a = kv.second;
                double tcw[3], Rwc[9];
                for (int k = 0; k < 3; ++k) tcw[k] = a.position[k] + a.orientation[k * 3] * a.t_cam0_imu[0] + a.orientation[k * 3 + 1] * a.t_cam0_imu[1] + a.orientation[k * 3 + 2] * a.t_cam0_imu[2];
                for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Rwc[r * 3 + c] = a.R_imu_cam0[r * 3] * a.orientation[c * 3] + a.R_imu_cam0[r * 3 + 1] * a.orientation[c * 3 + 1] + a.R_imu_cam0[r * 3 + 2] * a.orientation[c * 3 + 2];
                double* T = &wTc[(size_t)fidx * 16];   // camera -> world: [Rwc^T | tcw]
                for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) T[r * 4 + c] = Rwc[c * 3 + r]; T[r * 4 + 3] = tcw[r]; }
                T[12] = T[13] = T[14] = 0; T[15] = 1;
                for (int k = 0; k < K; ++k) {
                    double d[3], pc[3];
                    for (int c = 0; c < 3; ++c) d[c] = centre[c] + kps[k * 3 + c] - tcw[c];
                    for (int c = 0; c < 3; ++c) pc[c] = Rwc[c * 3] * d[0] + Rwc[c * 3 + 1] * d[1] + Rwc[c * 3 + 2] * d[2];
                    zs[((size_t)fidx * K + k) * 2] = pc[0] / pc[2] + 0.004 * G(rng);
                    zs[((size_t)fidx * K + k) * 2 + 1] = pc[1] / pc[2] + 0.004 * G(rng);
                }
                // bounding box of the ellipsoid's outline: dual conic C = Pm Q Pm^T, Q = wTo diag(a^2, b^2, c^2, -1) wTo^T, Pm = [Rwc | -Rwc tcw]
                double Pm[12], Q[16] = {0}, C[9] = {0};
                for (int r = 0; r < 3; ++r) {
                    for (int c = 0; c < 3; ++c) Pm[r * 4 + c] = Rwc[r * 3 + c];
                    Pm[r * 4 + 3] = -(Rwc[r * 3] * tcw[0] + Rwc[r * 3 + 1] * tcw[1] + Rwc[r * 3 + 2] * tcw[2]);
                }
                const double dg[4] = {shape[0] * shape[0], shape[1] * shape[1], shape[2] * shape[2], -1.0};
                for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) for (int k = 0; k < 4; ++k) Q[r * 4 + c] += wTo[r * 4 + k] * dg[k] * wTo[c * 4 + k];
                for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) for (int k = 0; k < 4; ++k) for (int l = 0; l < 4; ++l) C[r * 3 + c] += Pm[r * 4 + k] * Q[k * 4 + l] * Pm[c * 4 + l];
                const double sx = std::sqrt(C[2] * C[2] - C[0] * C[8]), sy = std::sqrt(C[5] * C[5] - C[4] * C[8]);
                double x0 = (C[2] - sx) / C[8], x1 = (C[2] + sx) / C[8], y0 = (C[5] - sy) / C[8], y1 = (C[5] + sy) / C[8];
                if (x0 > x1) std::swap(x0, x1);
                if (y0 > y1) std::swap(y0, y1);
                bb[(size_t)fidx * 4] = x0 + 0.002 * G(rng); bb[(size_t)fidx * 4 + 1] = y0 + 0.002 * G(rng);
                bb[(size_t)fidx * 4 + 2] = x1 + 0.002 * G(rng); bb[(size_t)fidx * 4 + 3] = y1 + 0.002 * G(rng);
                fclone[fidx] = i;
                ++fidx;
            }
            ++i;
        }
        orcvio_object_track trk{};
        trk.n_keypoints = K; trk.n_frames = FR; trk.wTo = wTo; trk.shape = shape; trk.kps = kps;
        trk.frame_wTc = wTc.data(); trk.frame_zs = zs.data(); trk.frame_bbox = bb.data(); trk.frame_clone = fclone.data();
        std::vector<orcvio_object_track> trks(1, trk);
        orcvio_object_eval_flags ef{};
        ef.use_left_perturbation = 1; ef.use_new_bbox_residual = 0; ef.vio_use_left_perturbation = 0; ef.fix_dcampose_dimupose_to_identity = 0;
        std::memcpy(ef.R_b2c, ss.imu_state.R_imu_cam0, sizeof(ef.R_b2c));
        std::memcpy(ef.t_c_b, ss.imu_state.t_cam0_imu, sizeof(ef.t_c_b));
        MsckfBackend fr(0, 16, 256, 8192);
        StateServer a = ss, b = ss;
        int rc = fr.covarianceToDevice(a);
        UpdateOutcome f1 = fr.msckfUpdate(a, map_server, ids);
        UpdateOutcome o1 = fr.removeLostObjectTracks(a, ef, trks);
        if (rc == ORCVIO_OK) rc = fr.covarianceToHost(a);
        int rc2 = fr.covarianceToDevice(b);
        MsckfBackend::FrameOutcome fo = fr.frameUpdate(b, map_server, ids, {}, ef, trks);
        if (rc2 == ORCVIO_OK) rc2 = fr.covarianceToHost(b);
        if (rc != ORCVIO_OK || rc2 != ORCVIO_OK || f1.status != ORCVIO_OK || o1.status != ORCVIO_OK || fo.features.status != ORCVIO_OK || fo.objects.status != ORCVIO_OK) {
            std::printf("frame: status %d %d / %d %d / %d %d (%s)\n", rc, rc2, f1.status, o1.status, fo.features.status, fo.objects.status, orcvio_msckf_last_error());
            ++fails;
        } else {
            const double e1 = relerr(fo.features.delta_x, f1.delta_x), e2 = o1.updated ? relerr(fo.objects.delta_x, o1.delta_x) : 0.0, eP = relerr(b.state_cov, a.state_cov);
            double epos = 0;
            for (auto& kv : a.imu_states_augment)
                for (int k = 0; k < 3; ++k) epos = std::fmax(epos, std::fabs(kv.second.position[k] - b.imu_states_augment.at(kv.first).position[k]));
            std::printf("frame in one call: object update accepted %d / %d (gamma %.3f), dx rel %.2e / %.2e, P rel %.2e, clone positions differ by %.2e\n",
                        (int)o1.updated, (int)fo.objects.updated, o1.gamma[0], e1, e2, eP, epos);
            if (!(e1 < 1e-12) || !(e2 < 1e-12) || !(eP < 1e-12) || !(epos < 1e-13) || o1.updated != fo.objects.updated || f1.accepted != fo.features.accepted) ++fails;
            if (!o1.updated) { std::printf("frame: the object update of the test was rejected\n"); ++fails; }
        }
    }
    std::printf(fails ? "FAILED\n" : "host gpu ok\n");
    return fails;
}
