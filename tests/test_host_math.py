"""The device inline math (orcvio_amd/csrc/msckf_math.hpp), compiled for the host, against the oracle."""
import ctypes as C
import os

import numpy as np
import pytest

from orcvio_amd import synth
from oracle import oracle

_dp = C.POINTER(C.c_double)


@pytest.mark.parametrize('larvio,left,fej', [(1, 0, 0), (1, 0, 1), (0, 0, 0), (0, 0, 1), (0, 1, 0), (0, 1, 1)])
def test_obs_jacobian_closed_forms(built, larvio, left, fej):
    lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'cpp', 'libhostmath.so'))
    f = synth.Flags(use_larvio=larvio, use_left_perturbation=left, if_fej=fej)
    w = synth.make_window(N=6, F=10, seed=3, flags=f, track_len=(3, 6))
    worst = 0.0
    for j in range(w.F):
        for k in range(w.obs_ptr[j], w.obs_ptr[j + 1]):
            i = int(w.obs_clone[k])
            ref = oracle.measurement_jacobian(w, i, w.p_w[j], w.obs_z[k])
            pose = np.zeros(28)
            pose[0:9] = w.R_b2w[i].ravel(); pose[9:12] = w.t_b_w[i]; pose[12:15] = w.t_fej[i]
            pose[15:24] = w.R_b2c[i].ravel(); pose[24:27] = w.t_c_b[i]
            Hx = np.zeros(12); He = np.zeros(12); Hf = np.zeros(6); r = np.zeros(2)
            pw = np.ascontiguousarray(w.p_w[j]); z = np.ascontiguousarray(w.obs_z[k])
            lib.orc_test_obs_jacobian(pose.ctypes.data_as(_dp), pw.ctypes.data_as(_dp), z.ctypes.data_as(_dp),
                                      larvio, left, fej, Hx.ctypes.data_as(_dp), He.ctypes.data_as(_dp),
                                      Hf.ctypes.data_as(_dp), r.ctypes.data_as(_dp))
            for a, b in zip((Hx, He, Hf, r), ref):
                worst = max(worst, float(np.abs(a - b.ravel()).max() / max(1.0, np.abs(b).max())))
    assert worst < 1e-13
