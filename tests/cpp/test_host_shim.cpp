// CPU test of the C++ host mirror (orcvio_amd/csrc/host/orcvio_msckf_host.hpp): flattening of the
// std::map containers and the row re-indexing of constructObjectResidualJacobians -- the layout the
// reference pins in src/tests/test_state_update.cpp:16-103 (2 frames x (1 keypoint + bbox), D = I).
// No device call is made (the backend is never constructed).
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../../orcvio_amd/csrc/host/orcvio_msckf_host.hpp"

using namespace orcvio_amd;

#define CHECK(c)                                                        \
    do {                                                                \
        if (!(c)) { std::printf("FAILED: %s (line %d)\n", #c, __LINE__); return 1; } \
    } while (0)

// access to the non-device parts without creating a handle
struct Probe : MsckfBackend {
    using MsckfBackend::flattenTracks;
    using MsckfBackend::flattenWindow;
};

int main() {
    // ---- flatten: clone ids 7, 3, 11 -> indices by id order 3->0, 7->1, 11->2 -------------------
    StateServer ss;
    for (StateIDType id : {7LL, 3LL, 11LL}) {
        IMUState_Aug a; a.id = id; a.position[0] = (double)id;
        ss.imu_states_augment[id] = a;
    }
    std::vector<double> R, t, tf, Rc, tc;
    std::map<StateIDType, int> index_of;
    MsckfBackend::flattenWindow(ss, R, t, tf, Rc, tc, index_of);
    CHECK(index_of[3] == 0 && index_of[7] == 1 && index_of[11] == 2);
    CHECK(t[0] == 3.0 && t[3] == 7.0 && t[6] == 11.0);

    MapServer map_server;
    Feature f1; f1.id = 5; f1.position[0] = 1;
    f1.observations[11] = {0.1, 0.2}; f1.observations[3] = {0.3, 0.4}; f1.observations[99] = {9, 9};   // 99 left the window
    f1.observations_vel[3] = {1, 2};
    Feature f2; f2.id = 2; f2.observations[7] = {0.5, 0.6}; f2.observations[11] = {0.7, 0.8};
    map_server[5] = f1; map_server[2] = f2;
    std::vector<double> p_w, z, zv;
    std::vector<int32_t> ptr, clone;
    MsckfBackend::flattenTracks(map_server, {5, 2}, index_of, {}, p_w, ptr, clone, z, zv);
    CHECK(ptr.size() == 3 && ptr[1] == 2 && ptr[2] == 4);
    CHECK(clone[0] == 0 && clone[1] == 2 && clone[2] == 1 && clone[3] == 2);   // ascending by clone id within a track
    CHECK(z[0] == 0.3 && z[1] == 0.4 && zv[0] == 1 && zv[1] == 2 && zv[2] == 0);
    // prune variant: only clones 3 and 7
    MsckfBackend::flattenTracks(map_server, {5, 2}, index_of, {3, 7}, p_w, ptr, clone, z, zv);
    CHECK(ptr[1] == 1 && ptr[2] == 2 && clone[0] == 0 && clone[1] == 1);

    // ---- constructObjectResidualJacobians layout (test_state_update.cpp:16-103) --------------------
    // The method only needs `flags` and the state server; build the object without a device through a
    // zero-initialised buffer (no constructor run, no destructor run).
    alignas(MsckfBackend) static unsigned char buf[sizeof(MsckfBackend)];
    std::memset(buf, 0, sizeof(buf));
    MsckfBackend* be = reinterpret_cast<MsckfBackend*>(buf);
    be->flags.leg_dim = 22;
    std::mt19937 rng(1);
    std::uniform_real_distribution<double> U(-1, 1);
    const int F = 2, rows = F * 2 + F * 4, oc = 45;
    std::vector<double> J(rows * 6), Hf(rows * oc), r(rows), Hf0, r0;
    for (auto& v : J) v = U(rng);
    for (auto& v : Hf) v = U(rng);
    for (auto& v : r) v = U(rng);
    Hf0 = Hf; r0 = r;
    std::vector<double> wTc(16 * F, 0.0);
    std::vector<int32_t> row_clone;
    std::vector<double> Hx6;
    bool ok = be->constructObjectResidualJacobians(ss, {0.0, 1.0}, J, {0.0, 1.0}, {1, 1}, wTc, oc, Hf, r, row_clone, Hx6, true);
    CHECK(ok && (int)row_clone.size() == rows);
    for (int i = 0; i < rows; ++i) {
        int nr, nc;
        if (i < F * 2) { nr = (i / 2) * 6 + (i % 2); nc = i / 2; }
        else { int j = i - F * 2; nr = (j / 4) * 6 + (j % 4) + 2; nc = j / 4; }
        CHECK(row_clone[nr] == nc);
        CHECK(r[nr] == r0[i]);
        for (int c = 0; c < 6; ++c) CHECK(Hx6[nr * 6 + c] == J[i * 6 + c]);
        for (int c = 0; c < oc; ++c) CHECK(Hf[nr * oc + c] == Hf0[i * oc + c]);
    }
    // frame 0 not in the window: only frame 1's six rows survive, all on clone index 0 of {1.0}
    Hf = Hf0; r = r0;
    ok = be->constructObjectResidualJacobians(ss, {1.0}, J, {0.0, 1.0}, {1, 1}, wTc, oc, Hf, r, row_clone, Hx6, true);
    CHECK(ok && row_clone.size() == 6 && r[0] == r0[2] && r[2] == r0[8]);
    Hf = Hf0; r = r0;
    ok = be->constructObjectResidualJacobians(ss, {5.0}, J, {0.0, 1.0}, {1, 1}, wTc, oc, Hf, r, row_clone, Hx6, true);
    CHECK(!ok);
    std::printf("host shim ok\n");
    return 0;
}
