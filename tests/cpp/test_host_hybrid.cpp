// GPU test of MsckfBackend::hybridUpdate (hybrid filter, existing EKF-SLAM features): the case is written by
// tests/test_host_shim.py (numpy), run here through the std::map containers and the C-ABI, and the outcome is written
// back for comparison with the literal restatement (oracle/mirror_hybrid.py + the write-back of src/orcvio.cpp:1836-1887).
//   usage: test_host_hybrid <case.bin> <out.bin>
#include <cstdio>
#include <cstdlib>

#include "../../orcvio_amd/csrc/host/orcvio_msckf_host.hpp"

using namespace orcvio_amd;

static std::vector<double> D;
static size_t pos = 0;
static double rd() { return D[pos++]; }
static void rdv(double* dst, int n) { for (int i = 0; i < n; ++i) dst[i] = rd(); }

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 3;
    fseek(f, 0, SEEK_END);
    const long bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    D.resize(bytes / 8);
    if (fread(D.data(), 8, D.size(), f) != D.size()) return 4;
    fclose(f);
    const int N = (int)rd(), F = (int)rd(), nf = (int)rd(), d = (int)rd(), estimate_td = (int)rd(), if_fej = (int)rd(), nnew = (int)rd(), use_larvio = (int)rd();
    const int knui = (int)rd();   // Schmidt nuisance states (poses follow the clones'; anchors >= N address them)
    StateServer ss;
    MapServer map;
    std::vector<StateIDType> ids(N);
    for (int i = 0; i < N; ++i) {
        IMUState_Aug a; a.id = 100 + 2 * i; ids[i] = a.id;
        rdv(a.orientation, 9); rdv(a.position, 3); rdv(a.position_FEJ, 3); rdv(a.R_imu_cam0, 9); rdv(a.t_cam0_imu, 3);
        // orientation_cam / position_cam as src/orcvio.cpp:954-961
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                double s = 0;
                for (int k = 0; k < 3; ++k) s += a.orientation[3 * r + k] * a.R_imu_cam0[3 * c + k];
                a.orientation_cam[3 * r + c] = s;
            }
        for (int r = 0; r < 3; ++r) {
            double s = a.position[r];
            for (int k = 0; k < 3; ++k) s += a.orientation[3 * r + k] * a.t_cam0_imu[k];
            a.position_cam[r] = s;
        }
        ss.imu_states_augment[a.id] = a;
    }
    auto cam_pose = [](IMUState_Aug& a) {   // orientation_cam / position_cam as src/orcvio.cpp:954-961
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                double s = 0;
                for (int k = 0; k < 3; ++k) s += a.orientation[3 * r + k] * a.R_imu_cam0[3 * c + k];
                a.orientation_cam[3 * r + c] = s;
            }
        for (int r = 0; r < 3; ++r) {
            double s = a.position[r];
            for (int k = 0; k < 3; ++k) s += a.orientation[3 * r + k] * a.t_cam0_imu[k];
            a.position_cam[r] = s;
        }
    };
    for (int j = 0; j < knui; ++j) {   // clones that left the window but stay in state_cov (use_schmidt, :2881-2920)
        IMUState_Aug a; a.id = 10 + j;
        rdv(a.orientation, 9); rdv(a.position, 3); rdv(a.position_FEJ, 3); rdv(a.R_imu_cam0, 9); rdv(a.t_cam0_imu, 3);
        cam_pose(a);
        ss.nui_ids.push_back(a.id);
        ss.nui_imu_states[a.id] = a;
        ids.push_back(a.id);   // (anchor index N + j)
    }
    ss.imu_state = IMUState();
    ss.imu_state.id = ids[N - 1];   // the current state is the newest clone
    std::memcpy(ss.imu_state.orientation, ss.imu_states_augment[ids[N - 1]].orientation, 72);
    std::memcpy(ss.imu_state.position, ss.imu_states_augment[ids[N - 1]].position, 24);
    std::memcpy(ss.imu_state.R_imu_cam0, ss.imu_states_augment[ids[0]].R_imu_cam0, 72);
    std::memcpy(ss.imu_state.t_cam0_imu, ss.imu_states_augment[ids[0]].t_cam0_imu, 24);
    std::vector<FeatureIDType> msckf_ids, ekf_ids, new_ids;
    for (int j = 0; j < F; ++j) {
        Feature ft; ft.id = 1000 + j;
        rdv(ft.position, 3);
        const int M = (int)rd();
        for (int k = 0; k < M; ++k) {
            const int c = (int)rd();
            Vec2 z, zv; z.x = rd(); z.y = rd(); zv.x = rd(); zv.y = rd();
            ft.observations[ids[c]] = z; ft.observations_vel[ids[c]] = zv;
        }
        map[ft.id] = ft; msckf_ids.push_back(ft.id);
    }
    for (int j = 0; j < nf; ++j) {
        Feature ft; ft.id = 5000 + j;
        ft.id_anchor = ids[(int)rd()];
        rdv(ft.invParam, 3); rdv(ft.obs_anchor, 3); ft.invDepth = rd(); rdv(ft.position, 3); rdv(ft.position_FEJ, 3);
        Vec2 z, zv; z.x = rd(); z.y = rd(); zv.x = rd(); zv.y = rd();
        ft.observations[ss.imu_state.id] = z; ft.observations_vel[ss.imu_state.id] = zv;
        map[ft.id] = ft; ekf_ids.push_back(ft.id); ss.feature_states.push_back(ft.id);
    }
    for (int j = 0; j < nnew; ++j) {   // features about to enter the state: anchor, parameters, position, all observations
        Feature ft; ft.id = 9000 + j;
        ft.id_anchor = ids[(int)rd()];
        rdv(ft.invParam, 3); rdv(ft.position, 3);
        ft.obs_anchor[0] = ft.invParam[0]; ft.obs_anchor[1] = ft.invParam[1]; ft.obs_anchor[2] = 1.0; ft.invDepth = ft.invParam[2];
        const int M = (int)rd();
        for (int k = 0; k < M; ++k) {
            const int c = (int)rd();
            Vec2 z, zv; z.x = rd(); z.y = rd(); zv.x = rd(); zv.y = rd();
            ft.observations[ids[c]] = z; ft.observations_vel[ids[c]] = zv;
        }
        map[ft.id] = ft; new_ids.push_back(ft.id);
    }
    const int n = 22 + 6 * N + d * nf + 6 * knui;
    ss.state_cov.resize((size_t)n * n);
    rdv(ss.state_cov.data(), n * n);
    if (pos != D.size()) { printf("case file: %zu of %zu doubles read\n", pos, D.size()); return 5; }

    MsckfBackend be(0, 32, 2048, 65536);
    be.flags.estimate_td = estimate_td; be.flags.if_fej = if_fej; be.flags.use_larvio = use_larvio;
    be.feature_idp_dim = d;
    std::vector<int> new_acc;
    UpdateOutcome o = be.hybridUpdate(ss, map, msckf_ids, ekf_ids, new_ids, &new_acc);
    if (o.status != ORCVIO_OK) { printf("hybridUpdate: %d %s\n", o.status, orcvio_msckf_last_error()); return 6; }
    if (!o.updated || !o.state_incremented) { printf("no update applied\n"); return 7; }

    std::vector<double> out;
    out.insert(out.end(), o.delta_x.begin(), o.delta_x.end());
    out.insert(out.end(), ss.state_cov.begin(), ss.state_cov.end());
    out.insert(out.begin(), (double)o.delta_x.size());   // size of the state after the update first
    for (int j = 0; j < F; ++j) out.push_back(o.accepted[j]);
    for (int a : o.ekf_accepted) out.push_back(a);
    for (int a : new_acc) out.push_back(a);
    for (FeatureIDType id : ss.feature_states) {
        const Feature& ft = map.at(id);
        out.insert(out.end(), ft.invParam, ft.invParam + 3);
        out.push_back(ft.invDepth);
        out.insert(out.end(), ft.position, ft.position + 3);
    }
    for (const auto& kv : ss.imu_states_augment) {
        out.insert(out.end(), kv.second.orientation, kv.second.orientation + 9);
        out.insert(out.end(), kv.second.position, kv.second.position + 3);
    }
    FILE* g = fopen(argv[2], "wb");
    if (!g) return 8;
    fwrite(out.data(), 8, out.size(), g);
    fclose(g);
    printf("host hybrid ok\n");
    return 0;
}
