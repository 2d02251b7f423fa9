// Host-compiled view of the device inline math (tests only): lets the CPU test-suite
// check orcvio_amd/csrc/msckf_math.hpp against the oracle without a GPU.
#include "../../orcvio_amd/csrc/msckf_math.hpp"
extern "C" void orc_test_obs_jacobian(const double* pose, const double* p_w, const double* z,
                                      int use_larvio, int use_left, int if_fej,
                                      double* Hx, double* He, double* Hf, double* r) {
    orcvio_amd::ObsFlags f{use_larvio, use_left, if_fej};
    orcvio_amd::obs_jacobian(pose, p_w, z, f, Hx, He, Hf, r);
}
