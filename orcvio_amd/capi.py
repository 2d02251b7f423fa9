"""ctypes binding of the C-ABI in include/orcvio_msckf.h (plumbing only).

The library is built in-tree by ``__graft_entry__.build()`` /
``orcvio_amd.build.build_library()`` into ``orcvio_amd/lib/liborcvio_msckf.so``.
There is no CPU fallback: if the shared object is missing, or no gfx950 device is
visible, the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'liborcvio_msckf.so')
LIB_DBG_PATH = os.path.join(_HERE, 'lib', 'liborcvio_msckf_dbg.so')   # diagnostics build: + orcvio_msckf_debug_* (tests only)

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)

STATUS = {0: 'OK', 1: 'ERR_INVALID', 2: 'ERR_NO_DEVICE', 3: 'ERR_CAPACITY', 4: 'ERR_TRACK_TOO_LONG',
          5: 'ERR_HIP', 6: 'ERR_NOT_SPD', 7: 'ERR_TIMEOUT', 8: 'ERR_PEER'}

# every symbol include/orcvio_msckf.h declares (checked by tests/test_abi.py)
EXPORTS = [
    'orcvio_msckf_abi_version', 'orcvio_msckf_last_error', 'orcvio_msckf_chi2_quantile',
    'orcvio_msckf_create', 'orcvio_msckf_destroy', 'orcvio_msckf_update_features',
    'orcvio_msckf_update_objects', 'orcvio_msckf_upload', 'orcvio_msckf_run_local',
    'orcvio_msckf_block_ptr', 'orcvio_msckf_run_finish', 'orcvio_msckf_run_update',
    'orcvio_msckf_sync', 'orcvio_msckf_download', 'orcvio_msckf_profile_update',
    'orcvio_msckf_increment_state', 'orcvio_msckf_set_option', 'orcvio_msckf_run_local_to',
    'orcvio_msckf_object_rows_eval', 'orcvio_msckf_triangulation_config_default', 'orcvio_msckf_triangulate',
    'orcvio_msckf_triangulate_uploaded', 'orcvio_msckf_objects_local', 'orcvio_msckf_objects_finish',
    'orcvio_msckf_objects_download', 'orcvio_msckf_cov_set', 'orcvio_msckf_cov_get', 'orcvio_msckf_cov_propagate',
    'orcvio_msckf_cov_augment', 'orcvio_msckf_cov_remove_clones', 'orcvio_msckf_cov_commit', 'orcvio_msckf_cov_prefactor', 'orcvio_msckf_upload_new_features', 'orcvio_msckf_download_new_feature_blocks', 'orcvio_msckf_upload_nuisance_poses',
    'orcvio_msckf_augment_state_nuisance', 'orcvio_msckf_cov_clones_to_nuisance', 'orcvio_msckf_cov_commit_new_features',
    'orcvio_msckf_update_object_tracks', 'orcvio_msckf_objects_local_tracks',
    'orcvio_msckf_upload_ekf_rows', 'orcvio_msckf_download_ekf', 'orcvio_msckf_upload_slam_features', 'orcvio_msckf_upload_dense_rows', 'orcvio_msckf_augment_new_features', 'orcvio_msckf_gate_tracks', 'orcvio_msckf_new_feature_rows', 'orcvio_msckf_augment_state',
    'orcvio_msckf_profile_stages', 'orcvio_msckf_update_object_lm_msgs',
    'orcvio_msckf_comm_unique_id', 'orcvio_msckf_comm_init', 'orcvio_msckf_comm_destroy', 'orcvio_msckf_comm_info',
    'orcvio_msckf_run_update_sharded', 'orcvio_msckf_update_features_sharded', 'orcvio_msckf_update_object_tracks_sharded',
    'orcvio_msckf_comm_barrier', 'orcvio_msckf_comm_allreduce_max', 'orcvio_msckf_io_begin', 'orcvio_msckf_io_update',
    'orcvio_msckf_augment_state_ref_ldlt', 'orcvio_msckf_io_update_frame', 'orcvio_msckf_io_stage_object_tracks', 'orcvio_msckf_io_submit', 'orcvio_msckf_io_collect',
    'orcvio_msckf_objects_refined', 'orcvio_msckf_counters', 'orcvio_msckf_comm_details', 'orcvio_msckf_profile_sharded',
    'orcvio_msckf_io_step_frame',
]


class MsckfFlags(C.Structure):
    _fields_ = [('leg_dim', C.c_int32), ('use_larvio', C.c_int32), ('use_left_perturbation', C.c_int32),
                ('if_fej', C.c_int32), ('estimate_td', C.c_int32), ('discard_large_update', C.c_int32),
                ('noise_feature', C.c_double), ('chi2_prob', C.c_double)]


class MsckfWindow(C.Structure):
    _fields_ = [('n_clones', C.c_int32), ('R_b2w', _dp), ('t_b_w', _dp), ('t_fej', _dp), ('R_b2c', _dp), ('t_c_b', _dp)]


class MsckfTracks(C.Structure):
    _fields_ = [('n_features', C.c_int32), ('p_w', _dp), ('obs_ptr', _ip), ('obs_clone', _ip), ('obs_z', _dp),
                ('obs_zvel', _dp)]


class MsckfObjectRows(C.Structure):
    _fields_ = [('n_rows', C.c_int32), ('n_obj_cols', C.c_int32), ('row_clone', _ip), ('Hx6', _dp), ('Hf', _dp),
                ('res', _dp)]


class ObjectEvalFlags(C.Structure):
    _fields_ = [('use_left_perturbation', C.c_int32), ('use_new_bbox_residual', C.c_int32),
                ('vio_use_left_perturbation', C.c_int32), ('fix_dcampose_dimupose_to_identity', C.c_int32),
                ('R_b2c', C.c_double * 9), ('t_c_b', C.c_double * 3)]


class ObjectTrackC(C.Structure):
    _fields_ = [('n_keypoints', C.c_int32), ('n_frames', C.c_int32), ('wTo', _dp), ('shape', _dp), ('kps', _dp),
                ('frame_wTc', _dp), ('frame_zs', _dp), ('frame_bbox', _dp), ('frame_clone', _ip)]


class ObjectLMMsg(C.Structure):
    """orcvio_object_lm_msg: one orcvio_ros_msgs/ObjectLM message (include/orcvio_msckf.h)."""
    _fields_ = [('object_id', C.c_int64), ('n_rows', C.c_int32), ('n_obj_cols', C.c_int32), ('n_frames', C.c_int32),
                ('residual', _dp), ('jacobian_wrt_object_state', _dp), ('jacobian_wrt_sensor_state', _dp),
                ('valid_camera_pose_mat', _dp), ('timestamps', _dp), ('zs_num_wrt_timestamps', _ip)]


class MsckfIo(C.Structure):
    """orcvio_msckf_io: the handle's pinned arena, written and read in place (include/orcvio_msckf.h)."""
    _fields_ = [('n', C.c_int32), ('poses', _dp), ('p_w', _dp), ('obs_ptr', _ip), ('obs_clone', _ip), ('obs_z', _dp),
                ('obs_zvel', _dp), ('P', _dp), ('dx', _dp), ('gamma', _dp), ('accept', _ip), ('P_out', _dp)]


POSE_STRIDE = 28   # ORCVIO_POSE_STRIDE: R_b2w 9 | t_b_w 3 | t_fej 3 | R_b2c 9 | t_c_b 3 | 1 unused


class MsckfResult(C.Structure):
    _fields_ = [('dx', _dp), ('P_out', _dp), ('accept', _ip), ('gamma', _dp), ('H_thin', _dp), ('r_thin', _dp),
                ('K', _dp), ('G', _dp), ('stats', C.c_int32 * 8)]


class TriangulationConfig(C.Structure):
    _fields_ = [('translation_threshold', C.c_double), ('huber_epsilon', C.c_double), ('estimation_precision', C.c_double),
                ('initial_damping', C.c_double), ('outer_loop_max_iteration', C.c_int32),
                ('inner_loop_max_iteration', C.c_int32), ('cost_threshold', C.c_double),
                ('init_final_dist_threshold', C.c_double)]


class TriangulationResult(C.Structure):
    _fields_ = [('valid', _ip), ('p_w', _dp), ('inv_param', _dp), ('flags', _ip), ('cost', _dp)]


class MsckfState(C.Structure):
    _fields_ = [('R_b2w_imu', C.c_double * 9), ('v', C.c_double * 3), ('p', C.c_double * 3), ('bg', C.c_double * 3),
                ('ba', C.c_double * 3), ('R_b2c', C.c_double * 9), ('t_c_b', C.c_double * 3), ('td', C.c_double),
                ('imu_intrinsics', C.c_double * 24), ('n_clones', C.c_int32), ('clone_R_b2w', _dp),
                ('clone_t_b_w', _dp), ('clone_R_c2w', _dp), ('clone_t_c_w', _dp)]


_LIB = None
_LIB_DBG = None


def load(debug_hooks=False):
    """Loads the HIP library; raises if it has not been built (no fallback).  debug_hooks: the diagnostics build, which adds the
    orcvio_msckf_debug_* test hooks (a separate shared object with its own state: handles must stay with the library that
    created them)."""
    global _LIB, _LIB_DBG
    try:
        # PyTorch ships its own libamdhip64; importing it first makes this process use ONE HIP
        # runtime (two runtimes in one process cannot both own the device).
        import torch  # noqa: F401
    except Exception:
        pass
    if debug_hooks:
        if _LIB_DBG is None:
            if not os.path.exists(LIB_DBG_PATH):
                raise RuntimeError(f'{LIB_DBG_PATH} is missing: run __graft_entry__.build()')
            _LIB_DBG = _bind(C.CDLL(LIB_DBG_PATH))
        return _LIB_DBG
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f'{LIB_PATH} is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950)')
    _LIB = _bind(C.CDLL(LIB_PATH))
    return _LIB


def _bind(lib):
    lib.orcvio_msckf_abi_version.restype = C.c_int32
    lib.orcvio_msckf_last_error.restype = C.c_char_p
    lib.orcvio_msckf_chi2_quantile.restype = C.c_double
    lib.orcvio_msckf_chi2_quantile.argtypes = [C.c_int32, C.c_double]
    lib.orcvio_msckf_create.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    lib.orcvio_msckf_destroy.argtypes = [C.c_void_p]
    lib.orcvio_msckf_destroy.restype = None
    lib.orcvio_msckf_update_features.argtypes = [C.c_void_p, C.POINTER(MsckfFlags), C.POINTER(MsckfWindow),
                                                 C.POINTER(MsckfTracks), _dp, C.POINTER(MsckfResult)]
    lib.orcvio_msckf_update_objects.argtypes = [C.c_void_p, C.POINTER(MsckfFlags), C.c_int32,
                                                C.POINTER(MsckfObjectRows), C.c_int32, _dp, C.POINTER(MsckfResult)]
    lib.orcvio_msckf_upload.argtypes = [C.c_void_p, C.POINTER(MsckfFlags), C.POINTER(MsckfWindow),
                                        C.POINTER(MsckfTracks), _dp]
    lib.orcvio_msckf_run_local.argtypes = [C.c_void_p, C.c_void_p]
    lib.orcvio_msckf_block_ptr.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    lib.orcvio_msckf_run_finish.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    lib.orcvio_msckf_run_update.argtypes = [C.c_void_p, C.c_void_p]
    lib.orcvio_msckf_sync.argtypes = [C.c_void_p, C.c_void_p]
    lib.orcvio_msckf_download.argtypes = [C.c_void_p, C.POINTER(MsckfResult)]
    lib.orcvio_msckf_profile_update.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_char_p), _dp,
                                                C.POINTER(C.c_int32)]
    lib.orcvio_msckf_increment_state.argtypes = [C.POINTER(MsckfFlags), _dp, C.POINTER(MsckfState)]
    lib.orcvio_msckf_set_option.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    lib.orcvio_msckf_run_local_to.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.orcvio_msckf_triangulation_config_default.argtypes = [C.POINTER(TriangulationConfig)]
    lib.orcvio_msckf_triangulation_config_default.restype = None
    lib.orcvio_msckf_triangulate.argtypes = [C.c_void_p, C.POINTER(TriangulationConfig), C.POINTER(MsckfWindow),
                                             C.POINTER(MsckfTracks), _ip, C.POINTER(TriangulationResult)]
    lib.orcvio_msckf_triangulate_uploaded.argtypes = [C.c_void_p, C.POINTER(TriangulationConfig), _ip, C.c_void_p]
    lib.orcvio_msckf_objects_local.argtypes = [C.c_void_p, C.POINTER(MsckfFlags), C.c_int32, C.POINTER(MsckfObjectRows),
                                               C.c_int32, _dp, C.c_void_p, _ip, C.c_void_p]
    lib.orcvio_msckf_objects_finish.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    lib.orcvio_msckf_objects_download.argtypes = [C.c_void_p, C.POINTER(MsckfResult)]
    lib.orcvio_msckf_cov_set.argtypes = [C.c_void_p, C.c_int32, _dp]
    lib.orcvio_msckf_cov_get.argtypes = [C.c_void_p, _ip, _dp]
    lib.orcvio_msckf_cov_propagate.argtypes = [C.c_void_p, C.c_int32, _dp, _dp]
    lib.orcvio_msckf_cov_augment.argtypes = [C.c_void_p]
    lib.orcvio_msckf_cov_remove_clones.argtypes = [C.c_void_p, C.c_int32, _ip, C.c_int32]
    lib.orcvio_msckf_cov_commit.argtypes = [C.c_void_p]
    lib.orcvio_msckf_cov_prefactor.argtypes = [C.c_void_p]
    lib.orcvio_msckf_cov_clones_to_nuisance.argtypes = [C.c_void_p, C.c_int32, _ip, C.c_int32]
    lib.orcvio_msckf_upload_nuisance_poses.argtypes = [C.c_void_p, C.c_void_p]
    lib.orcvio_msckf_update_object_tracks.argtypes = [C.c_void_p, C.POINTER(MsckfFlags), C.POINTER(ObjectEvalFlags), C.c_int32,
                                                      C.POINTER(ObjectTrackC), C.c_int32, _dp, C.POINTER(MsckfResult)]
    lib.orcvio_msckf_objects_local_tracks.argtypes = [C.c_void_p, C.POINTER(MsckfFlags), C.POINTER(ObjectEvalFlags), C.c_int32,
                                                      C.POINTER(ObjectTrackC), C.c_int32, _dp, C.c_void_p, _ip, C.c_void_p]
    lib.orcvio_msckf_io_update_frame.argtypes = [C.c_void_p, C.POINTER(MsckfResult), C.POINTER(MsckfFlags), C.POINTER(ObjectEvalFlags),
                                                 C.POINTER(ObjectTrackC), C.c_int32, C.c_int32, C.POINTER(MsckfResult)]
    lib.orcvio_msckf_io_stage_object_tracks.argtypes = [C.c_void_p, C.POINTER(MsckfFlags), C.POINTER(ObjectEvalFlags), C.POINTER(ObjectTrackC), C.c_int32]
    lib.orcvio_msckf_comm_unique_id.argtypes = [C.c_char_p]
    lib.orcvio_msckf_comm_init.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_int32]
    lib.orcvio_msckf_comm_destroy.argtypes = [C.c_void_p]
    lib.orcvio_msckf_comm_info.argtypes = [C.c_void_p, _ip, _ip]
    lib.orcvio_msckf_run_update_sharded.argtypes = [C.c_void_p, C.c_void_p]
    lib.orcvio_msckf_comm_barrier.argtypes = [C.c_void_p]
    lib.orcvio_msckf_comm_allreduce_max.argtypes = [C.c_void_p, _dp, C.c_int32]
    lib.orcvio_msckf_io_begin.argtypes = [C.c_void_p, C.POINTER(MsckfFlags), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(MsckfIo)]
    lib.orcvio_msckf_io_update.argtypes = [C.c_void_p, C.c_int32, C.c_int32, _ip]
    lib.orcvio_msckf_io_submit.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    lib.orcvio_msckf_io_collect.argtypes = [C.c_void_p, _ip]
    lib.orcvio_msckf_io_step_frame.argtypes = [C.c_void_p, C.POINTER(FrameStep), C.POINTER(FrameResult)]
    lib.orcvio_msckf_update_features_sharded.argtypes = lib.orcvio_msckf_update_features.argtypes
    lib.orcvio_msckf_update_object_tracks_sharded.argtypes = lib.orcvio_msckf_update_object_tracks.argtypes
    return lib


COMM_ID_BYTES = 128


def comm_unique_id() -> bytes:
    """ncclGetUniqueId (rank 0); ship the bytes to the other ranks, then every rank calls MsckfUpdater.comm_init."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = load().orcvio_msckf_comm_unique_id(buf)
    if rc != 0:
        raise MsckfError(rc, 'orcvio_msckf_comm_unique_id')
    return buf.raw


class MsckfError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        msg = load().orcvio_msckf_last_error().decode(errors='replace')
        super().__init__(f'{where}: {STATUS.get(code, code)}: {msg}')


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _i(a):
    return None if a is None else a.ctypes.data_as(_ip)


class EkfRows(C.Structure):
    """orcvio_msckf_ekf_rows (include/orcvio_msckf.h)."""
    _fields_ = [('n_features', C.c_int32), ('idp_dim', C.c_int32),
                ('anchor', C.POINTER(C.c_int32)), ('state', C.POINTER(C.c_int32)), ('slot', C.POINTER(C.c_int32)),
                ('H_e', C.POINTER(C.c_double)), ('H_a', C.POINTER(C.c_double)), ('H_x', C.POINTER(C.c_double)),
                ('H_f', C.POINTER(C.c_double)), ('z_vel', C.POINTER(C.c_double)), ('r', C.POINTER(C.c_double))]


class SlamFeatures(C.Structure):
    """orcvio_msckf_slam_features (include/orcvio_msckf.h)."""
    _fields_ = [('n_features', C.c_int32), ('idp_dim', C.c_int32),
                ('anchor', C.POINTER(C.c_int32)), ('state', C.POINTER(C.c_int32)), ('slot', C.POINTER(C.c_int32)),
                ('param', C.POINTER(C.c_double)), ('inv_depth', C.POINTER(C.c_double)), ('p_w', C.POINTER(C.c_double)),
                ('p_fej', C.POINTER(C.c_double)), ('z', C.POINTER(C.c_double)), ('z_vel', C.POINTER(C.c_double))]


class FrameStep(C.Structure):
    """orcvio_msckf_frame_step (include/orcvio_msckf.h)."""
    _fields_ = [('leg_dim', C.c_int32), ('Phi', C.POINTER(C.c_double)), ('Q', C.POINTER(C.c_double)), ('augment', C.c_int32),
                ('slam_features', C.POINTER(SlamFeatures)), ('prune_tracks', C.POINTER(MsckfTracks)), ('prune_apply_dx', C.c_int32),
                ('remove_clones', C.POINTER(C.c_int32)), ('n_remove', C.c_int32)]


class FrameResult(C.Structure):
    """orcvio_msckf_frame_result (include/orcvio_msckf.h)."""
    _fields_ = [('stats', C.c_int32 * 8), ('prune_stats', C.c_int32 * 8), ('dx', C.POINTER(C.c_double)), ('gamma', C.POINTER(C.c_double)),
                ('accept', C.POINTER(C.c_int32)), ('prune_dx', C.POINTER(C.c_double)), ('prune_gamma', C.POINTER(C.c_double)),
                ('prune_accept', C.POINTER(C.c_int32)), ('n_after', C.c_int32), ('status_first', C.c_int32), ('status_prune', C.c_int32),
                ('repaired', C.c_int32)]


class MsckfNewFeatures(C.Structure):
    """orcvio_msckf_new_features (include/orcvio_msckf.h)."""
    _fields_ = [('n_features', C.c_int32), ('idp_dim', C.c_int32), ('anchor', C.POINTER(C.c_int32)),
                ('param', C.POINTER(C.c_double)), ('inv_depth', C.POINTER(C.c_double)), ('p_w', C.POINTER(C.c_double)),
                ('p_fej', C.POINTER(C.c_double)), ('obs_ptr', C.POINTER(C.c_int32)), ('obs_clone', C.POINTER(C.c_int32)),
                ('obs_z', C.POINTER(C.c_double)), ('obs_zvel', C.POINTER(C.c_double))]


def make_flags(f) -> MsckfFlags:
    return MsckfFlags(int(f.leg_dim), int(f.use_larvio), int(f.use_left_perturbation), int(f.if_fej),
                      int(f.estimate_td), int(f.discard_large_update), float(f.noise_feature), float(f.chi2_prob))


class MsckfUpdater:
    """Thin owner of one ``orcvio_msckf_handle`` taking ``synth.Window``-shaped inputs."""

    def __init__(self, device=0, max_clones=32, max_features=2048, max_observations=65536, debug_hooks=False):
        self.lib = load(debug_hooks)
        self.debug_hooks = bool(debug_hooks)
        self.h = C.c_void_p()
        rc = self.lib.orcvio_msckf_create(device, max_clones, max_features, max_observations, C.byref(self.h))
        if rc != 0:
            raise MsckfError(rc, 'orcvio_msckf_create')
        self._keep = None
        self.n = None
        self.F = None
        self.n_extra = 0

    def set_materialize_stack(self, on: bool):
        """ORCVIO_OPT_MATERIALIZE_STACK: also write the stacked projected blocks [H' | r'] (debug_read 'Hs')."""
        rc = self.lib.orcvio_msckf_set_option(self.h, 1, int(bool(on)))
        if rc != 0:
            raise MsckfError(rc, 'orcvio_msckf_set_option')

    def set_fused_solve(self, on: bool):
        """ORCVIO_OPT_FUSED_SOLVE: chol(M) and the triangular solve in one launch (default) or in two."""
        rc = self.lib.orcvio_msckf_set_option(self.h, 2, int(bool(on)))
        if rc != 0:
            raise MsckfError(rc, 'orcvio_msckf_set_option')

    def set_lookahead_solve(self, depth: int):
        """ORCVIO_OPT_LOOKAHEAD_SOLVE: 3 (default) / 2 = k_potrf_solve_la with that look-ahead depth, 0 = k_potrf_solve (one workgroup)."""
        rc = self.lib.orcvio_msckf_set_option(self.h, 14, int(depth))
        if rc != 0:
            raise MsckfError(rc, 'orcvio_msckf_set_option')

    def set_fused_front(self, on: bool):
        """ORCVIO_OPT_FUSED_FRONT: chol(P) as workgroup 0 of the feature launch (default) or forked to a side stream."""
        rc = self.lib.orcvio_msckf_set_option(self.h, 3, int(bool(on)))
        if rc != 0:
            raise MsckfError(rc, 'orcvio_msckf_set_option')

    def set_ekf_rows_mode(self, on: bool):
        """ORCVIO_OPT_EKF_ROWS: the extra states become active columns (upload_ekf_rows may follow an upload)."""
        rc = self.lib.orcvio_msckf_set_option(self.h, 5, int(bool(on)))
        if rc != 0:
            raise MsckfError(rc, 'orcvio_msckf_set_option')

    def upload_ekf_rows(self, idp_dim, anchor, state, slot, H_e, H_a, H_x, H_f, r, z_vel=None):
        """Row pairs of the SLAM features the current state observes, as featureJacobian_ekf produces them."""
        F = len(anchor)
        ia = [np.ascontiguousarray(a, dtype=np.int32) for a in (anchor, state, slot)]
        da = [np.ascontiguousarray(a, dtype=np.float64) for a in (H_e, H_a, H_x, H_f, r)]
        zv = None if z_vel is None else np.ascontiguousarray(z_vel, dtype=np.float64)
        ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
        rows = EkfRows(F, int(idp_dim), ip(ia[0]), ip(ia[1]), ip(ia[2]), _d(da[0]), _d(da[1]), _d(da[2]), _d(da[3]), _d(zv), _d(da[4]))
        rc = self.lib.orcvio_msckf_upload_ekf_rows(self.h, C.byref(rows))
        if rc != 0:
            raise MsckfError(rc, 'orcvio_msckf_upload_ekf_rows')
        self._ekf_F = F

    def upload_slam_features(self, idp_dim, slam, slots=None):
        """SLAM features (synth.SlamFeature-shaped) observed by the current state: rows evaluated on the device."""
        F = len(slam)
        ia = [np.ascontiguousarray(a, dtype=np.int32) for a in ([f.anchor for f in slam], [f.state for f in slam],
                                                                 list(range(F)) if slots is None else slots)]
        param = np.ascontiguousarray([f.inv_param if idp_dim == 3 else f.obs_anchor for f in slam], dtype=np.float64).reshape(F, 3)
        rho = np.ascontiguousarray([f.inv_depth for f in slam], dtype=np.float64)
        pw = np.ascontiguousarray([f.p_w for f in slam], dtype=np.float64).reshape(F, 3)
        pf = np.ascontiguousarray([f.p_fej if f.p_fej is not None else f.p_w for f in slam], dtype=np.float64).reshape(F, 3)
        z = np.ascontiguousarray([f.z for f in slam], dtype=np.float64).reshape(F, 2)
        zv = np.ascontiguousarray([f.z_vel for f in slam], dtype=np.float64).reshape(F, 2)
        ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
        st = SlamFeatures(F, int(idp_dim), ip(ia[0]), ip(ia[1]), ip(ia[2]), _d(param), _d(rho), _d(pw), _d(pf), _d(z), _d(zv))
        rc = self.lib.orcvio_msckf_upload_slam_features(self.h, C.byref(st))
        if rc != 0:
            raise MsckfError(rc, 'orcvio_msckf_upload_slam_features')
        self._ekf_F = F

    def make_slam_call(self, idp_dim, slam, slots=None):
        """orcvio_msckf_upload_slam_features with the records marshalled once: returns call()."""
        F = len(slam)
        ia = [np.ascontiguousarray(a, dtype=np.int32) for a in ([f.anchor for f in slam], [f.state for f in slam],
                                                                 list(range(F)) if slots is None else slots)]
        param = np.ascontiguousarray([f.inv_param if idp_dim == 3 else f.obs_anchor for f in slam], dtype=np.float64).reshape(F, 3)
        rho = np.ascontiguousarray([f.inv_depth for f in slam], dtype=np.float64)
        pw = np.ascontiguousarray([f.p_w for f in slam], dtype=np.float64).reshape(F, 3)
        pf = np.ascontiguousarray([f.p_fej if f.p_fej is not None else f.p_w for f in slam], dtype=np.float64).reshape(F, 3)
        z = np.ascontiguousarray([f.z for f in slam], dtype=np.float64).reshape(F, 2)
        zv = np.ascontiguousarray([f.z_vel for f in slam], dtype=np.float64).reshape(F, 2)
        ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
        st = SlamFeatures(F, int(idp_dim), ip(ia[0]), ip(ia[1]), ip(ia[2]), _d(param), _d(rho), _d(pw), _d(pf), _d(z), _d(zv))
        lib, h = self.lib, self.h

        def call():
            rc = lib.orcvio_msckf_upload_slam_features(h, C.byref(st))
            if rc != 0:
                raise MsckfError(rc, 'orcvio_msckf_upload_slam_features')
            self._ekf_F = F
        call.hold = (ia, param, rho, pw, pf, z, zv, st)
        return call

    def upload_new_features(self, win, idp_dim, feats):
        """Features entering the state (objects as for new_feature_rows): featureJacobian_ekf_new and the W split on the
        device; the V parts join the dense rows of this upload.  Call download_new_feature_blocks() after the update."""
        k, d = len(feats), int(idp_dim)
        ptr, cl, zz, zv = [0], [], [], []
        for ft in feats:
            for (c, z, v) in ft.obs:
                cl.append(c); zz.append(z); zv.append(v)
            ptr.append(len(cl))
        ia = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        da = lambda a, shape: np.ascontiguousarray(a, dtype=np.float64).reshape(shape)
        keep = [ia([f.anchor for f in feats]), da([f.inv_param if d == 3 else f.obs_anchor for f in feats], (k, 3)),
                da([f.inv_depth for f in feats], (k,)), da([f.p_w for f in feats], (k, 3)),
                da([f.p_fej if f.p_fej is not None else f.p_w for f in feats], (k, 3)), ia(ptr), ia(cl), da(zz, (-1, 2)), da(zv, (-1, 2))]
        nf = MsckfNewFeatures(k, d, _i(keep[0]), _d(keep[1]), _d(keep[2]), _d(keep[3]), _d(keep[4]), _i(keep[5]), _i(keep[6]),
                              _d(keep[7]), _d(keep[8]))
        rc = self.lib.orcvio_msckf_upload_new_features(self.h, C.byref(nf))
        if rc != 0:
            raise MsckfError(rc, 'orcvio_msckf_upload_new_features')
        self._new = (k, d, win.n)

    def download_new_feature_blocks(self):
        k, d, n = self._new
        H_1, H_2, r_1 = np.zeros((d * k, n)), np.zeros((k, d, d)), np.zeros(d * k)
        rc = self.lib.orcvio_msckf_download_new_feature_blocks(self.h, _d(H_1), _d(H_2), _d(r_1))
        if rc != 0:
            raise MsckfError(rc, 'orcvio_msckf_download_new_feature_blocks')
        return H_1, H_2, r_1

    def upload_dense_rows(self, H, r):
        """Caller-projected dense rows over the whole state, stacked as they are (no gate)."""
        H = np.ascontiguousarray(H, dtype=np.float64)
        r = np.ascontiguousarray(r, dtype=np.float64)
        rc = self.lib.orcvio_msckf_upload_dense_rows(self.h, H.shape[0], _d(H), _d(r))
        if rc != 0:
            raise MsckfError(rc, 'orcvio_msckf_upload_dense_rows')

    def augment_new_features(self, win, track, anchor, inv_param, dx, P_upd):
        """New 3-d SLAM features that were listed as tracks: (dx_new [3k], P_aug [(n+3k)^2]) after the update."""
        fl, wn, tr, keep = self._structs(win)
        k = len(track)
        n = P_upd.shape[0]
        ti = np.ascontiguousarray(track, dtype=np.int32)
        ai = np.ascontiguousarray(anchor, dtype=np.int32)
        ip = np.ascontiguousarray(inv_param, dtype=np.float64).reshape(k, 3)
        dxc = np.ascontiguousarray(dx, dtype=np.float64)
        Pc = np.ascontiguousarray(P_upd, dtype=np.float64)
        dx_new = np.zeros(3 * k)
        P_aug = np.zeros((n + 3 * k, n + 3 * k))
        rc = self.lib.orcvio_msckf_augment_new_features(self.h, C.byref(wn), k, ti.ctypes.data_as(C.POINTER(C.c_int32)),
                                                        ai.ctypes.data_as(C.POINTER(C.c_int32)), _d(ip), _d(dxc), _d(Pc), _d(dx_new), _d(P_aug))
        if rc != 0:
            raise MsckfError(rc, 'orcvio_msckf_augment_new_features')
        return dx_new, P_aug

    def gate_tracks(self, win):
        """gatingTestFeature of every track of `win` against win.P, no update: (gamma [F], accept [F])."""
        fl, wn, tr, keep = self._structs(win)
        gamma = np.zeros(win.F)
        accept = np.zeros(win.F, dtype=np.int32)
        rc = self.lib.orcvio_msckf_gate_tracks(self.h, C.byref(fl), C.byref(wn), C.byref(tr), _d(keep['P']), _d(gamma),
                                               accept.ctypes.data_as(C.POINTER(C.c_int32)))
        if rc != 0:
            raise MsckfError(rc, 'orcvio_msckf_gate_tracks')
        return gamma, accept

    def download_ekf(self):
        F = getattr(self, '_ekf_F', 0)
        gamma = np.zeros(F)
        accept = np.zeros(F, dtype=np.int32)
        rc = self.lib.orcvio_msckf_download_ekf(self.h, _d(gamma), accept.ctypes.data_as(C.POINTER(C.c_int32)))
        if rc != 0:
            raise MsckfError(rc, 'orcvio_msckf_download_ekf')
        return gamma, accept

    def set_extra_states(self, k: int):
        """ORCVIO_OPT_EXTRA_STATES: k state columns behind the clones (EKF-SLAM feature states) untouched by the rows."""
        rc = self.lib.orcvio_msckf_set_option(self.h, 4, int(k))
        if rc != 0:
            raise MsckfError(rc, 'orcvio_msckf_set_option')
        self.n_extra = int(k)

    # -- multi-GPU: the handle's own RCCL communicator ---------------------------------------------
    def comm_init(self, unique_id: bytes, rank: int, world: int):
        assert len(unique_id) == COMM_ID_BYTES
        self._chk(self.lib.orcvio_msckf_comm_init(self.h, unique_id, int(rank), int(world)), 'orcvio_msckf_comm_init')

    def comm_destroy(self):
        self._chk(self.lib.orcvio_msckf_comm_destroy(self.h), 'orcvio_msckf_comm_destroy')

    def counters(self):
        """Cumulative counters of the handle: front_fallbacks (fused front end re-run on the forked path because another tenant of the
        device held compute units), launch sequences captured / replayed from a graph / run as plain launches."""
        v = (C.c_int64 * 10)()
        self.lib.orcvio_msckf_counters.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_int32]
        self._chk(self.lib.orcvio_msckf_counters(self.h, v, 10), 'orcvio_msckf_counters')
        return dict(front_fallbacks=int(v[0]), graph_captures=int(v[1]), graph_replays=int(v[2]), plain_runs=int(v[3]),
                    front_blocked_by_comm=int(v[4]), obj_fused=int(v[5]), chained_frames=int(v[6]), prestaged_frames=int(v[7]),
                    step_frames=int(v[8]), step_repairs=int(v[9]))

    def comm_info(self):
        r, w = C.c_int32(0), C.c_int32(0)
        self._chk(self.lib.orcvio_msckf_comm_info(self.h, C.byref(r), C.byref(w)), 'orcvio_msckf_comm_info')
        return r.value, w.value

    def comm_details(self):
        v = (C.c_int32 * 8)()
        self.lib.orcvio_msckf_comm_details.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int32]
        self._chk(self.lib.orcvio_msckf_comm_details(self.h, v, 8), 'orcvio_msckf_comm_details')
        return dict(transport={0: 'none', 1: 'rccl', 2: 'ipc'}[int(v[0])], rank=int(v[1]), world=int(v[2]), ranks_seen=int(v[3]),
                    shared_device=bool(v[4]), gather_uncached=bool(v[5]), ipc_across_devices_unverified=bool(v[6]))

    def profile_sharded(self, reps=20):
        """COLLECTIVE.  Device microseconds of this rank's sharded update: local, exchange, replicated solve, total (medians)."""
        us = (C.c_double * 4)()
        self.lib.orcvio_msckf_profile_sharded.argtypes = [C.c_void_p, C.c_int32, _dp]
        self._chk(self.lib.orcvio_msckf_profile_sharded(self.h, int(reps), us), 'orcvio_msckf_profile_sharded')
        return dict(local_us=us[0], exchange_us=us[1], replicated_solve_us=us[2], total_us=us[3])

    def comm_barrier(self):
        """Everything enqueued on the handle's stream is finished on every rank (bounded wait: ERR_TIMEOUT, never a hang)."""
        self._chk(self.lib.orcvio_msckf_comm_barrier(self.h), 'orcvio_msckf_comm_barrier')

    def comm_allreduce_max(self, values):
        """max over the ranks of up to 8 doubles, through the handle's communicator."""
        v = np.ascontiguousarray(values, dtype=np.float64).copy()
        self._chk(self.lib.orcvio_msckf_comm_allreduce_max(self.h, _d(v), int(v.size)), 'orcvio_msckf_comm_allreduce_max')
        return v

    def run_update_sharded(self, stream=None):
        """This rank's uploaded tracks -> block -> RCCL all-gather -> rank-ordered sum -> replicated solve."""
        self._chk(self.lib.orcvio_msckf_run_update_sharded(self.h, C.c_void_p(stream) if stream else None),
                  'orcvio_msckf_run_update_sharded')

    def update_features_sharded(self, win, want_G=False, resident_cov=False, want_P=True):
        """Host buffers in and out; `win` holds THIS RANK's share of the tracks (sharding.shard_window)."""
        fl, w, t, arrs = self._structs(win)
        out, res = self._result(win.n, win.F, False, want_G, False)
        if not want_P:
            res.P_out = None
        rc = self.lib.orcvio_msckf_update_features_sharded(self.h, C.byref(fl), C.byref(w), C.byref(t),
                                                           None if resident_cov else _d(arrs['P']), C.byref(res))
        if rc != 0:
            raise MsckfError(rc, 'orcvio_msckf_update_features_sharded')
        return self._finish(out, res, win.F)

    def update_object_tracks_sharded(self, flags, n_clones, objs, P, R_b2c, t_c_b, obj_left, new_bbox, vio_left, fix_D=False, want_G=False):
        """System::processObjects over the ranks: `objs` are THIS RANK's object tracks."""
        fl = make_flags(flags)
        ef, arr, keep = self._object_tracks(objs, R_b2c, t_c_b, obj_left, new_bbox, vio_left, fix_D)
        n = flags.leg_dim + 6 * n_clones + self.n_extra
        Pc = None if P is None else np.ascontiguousarray(P, dtype=np.float64)
        out, res = self._result(n, 1, False, want_G, False)
        rc = self.lib.orcvio_msckf_update_object_tracks_sharded(self.h, C.byref(fl), C.byref(ef), n_clones, arr, len(objs), _d(Pc), C.byref(res))
        if rc != 0:
            raise MsckfError(rc, 'orcvio_msckf_update_object_tracks_sharded')
        out = self._finish(out, res, 1)
        out['gamma'] = float(out['gamma'][0])
        out['accept'] = int(out['accept'][0])
        return out

    def close(self):
        if self.h:
            self.lib.orcvio_msckf_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- argument marshalling -------------------------------------------------------------
    def _structs(self, win):
        fl = make_flags(win.flags)
        arrs = dict(R_b2w=np.ascontiguousarray(win.R_b2w, dtype=np.float64),
                    t_b_w=np.ascontiguousarray(win.t_b_w, dtype=np.float64),
                    t_fej=np.ascontiguousarray(win.t_fej, dtype=np.float64),
                    R_b2c=np.ascontiguousarray(win.R_b2c, dtype=np.float64),
                    t_c_b=np.ascontiguousarray(win.t_c_b, dtype=np.float64),
                    p_w=np.ascontiguousarray(win.p_w, dtype=np.float64),
                    obs_ptr=np.ascontiguousarray(win.obs_ptr, dtype=np.int32),
                    obs_clone=np.ascontiguousarray(win.obs_clone, dtype=np.int32),
                    obs_z=np.ascontiguousarray(win.obs_z, dtype=np.float64),
                    obs_zvel=np.ascontiguousarray(win.obs_zvel, dtype=np.float64),
                    P=np.ascontiguousarray(win.P, dtype=np.float64))
        w = MsckfWindow(win.N, _d(arrs['R_b2w']), _d(arrs['t_b_w']), _d(arrs['t_fej']), _d(arrs['R_b2c']),
                        _d(arrs['t_c_b']))
        t = MsckfTracks(win.F, _d(arrs['p_w']), _i(arrs['obs_ptr']), _i(arrs['obs_clone']), _d(arrs['obs_z']),
                        _d(arrs['obs_zvel']))
        return fl, w, t, arrs

    def _result(self, n, F, want_K=False, want_G=False, want_thin=False):
        out = dict(dx=np.zeros(n), P_new=np.zeros((n, n)), accept=np.zeros(max(F, 1), dtype=np.int32),
                   gamma=np.zeros(max(F, 1)))
        if want_thin:
            out['H_thin'] = np.zeros((n - 15, n))
            out['r_thin'] = np.zeros(n - 15)
        if want_K:
            out['K'] = np.zeros((n, n - 15))
        if want_G:
            out['G'] = np.zeros((n, n))
        res = MsckfResult(_d(out['dx']), _d(out['P_new']), _i(out['accept']), _d(out['gamma']),
                          _d(out.get('H_thin')), _d(out.get('r_thin')), _d(out.get('K')), _d(out.get('G')))
        return out, res

    @staticmethod
    def _finish(out, res, F):
        out['accept'] = out['accept'][:F]
        out['gamma'] = out['gamma'][:F]
        out['stats'] = np.array(list(res.stats), dtype=np.int32)
        out['updated'] = bool(out['stats'][3])
        return out

    # -- one-shot host-buffer update (the reference call sites) ---------------------------
    def update_features(self, win, want_K=False, want_G=False, want_thin=False, resident_cov=False, want_P=True):
        """resident_cov: the prior is the device-resident covariance (P is not sent); want_P=False: P+ stays in HBM
        (cov_commit makes it the resident covariance) and only dx, gamma, accept come back."""
        fl, w, t, arrs = self._structs(win)
        out, res = self._result(win.n, win.F, want_K, want_G, want_thin)
        if not want_P:
            res.P_out = None
        rc = self.lib.orcvio_msckf_update_features(self.h, C.byref(fl), C.byref(w), C.byref(t),
                                                   None if resident_cov else _d(arrs['P']), C.byref(res))
        if rc != 0:
            raise MsckfError(rc, 'orcvio_msckf_update_features')
        return self._finish(out, res, win.F)

    # -- the zero-copy boundary: the handle's pinned arena written and read in place ----------------
    def io_begin(self, flags, N, F, nobs, with_P=True):
        """orcvio_msckf_io_begin: returns numpy VIEWS of the handle's pinned arena -- inputs to fill (poses [N, 28], p_w, obs_ptr,
        obs_clone, obs_z, obs_zvel or None, P or None) and outputs to read after io_update (dx, gamma, accept, P_out)."""
        fl = make_flags(flags)
        io = MsckfIo()
        self._chk(self.lib.orcvio_msckf_io_begin(self.h, C.byref(fl), int(N), int(F), int(nobs), 2 if with_P == 2 and with_P is not True else int(bool(with_P)), C.byref(io)), 'orcvio_msckf_io_begin')
        n = int(io.n)
        view = lambda ptr, shape: None if not ptr else np.ctypeslib.as_array(ptr, shape=shape)
        self.n, self.F = n, int(F)
        self._io_keep = (fl, io)
        return dict(n=n, poses=view(io.poses, (N, POSE_STRIDE)), p_w=view(io.p_w, (max(F, 1), 3))[:F], obs_ptr=view(io.obs_ptr, (F + 1,)),
                    obs_clone=view(io.obs_clone, (max(nobs, 1),))[:nobs], obs_z=view(io.obs_z, (max(nobs, 1), 2))[:nobs],
                    obs_zvel=None if not io.obs_zvel else view(io.obs_zvel, (max(nobs, 1), 2))[:nobs], P=view(io.P, (n, n)),
                    dx=view(io.dx, (n,)), gamma=view(io.gamma, (max(F, 1),))[:F], accept=view(io.accept, (max(F, 1),))[:F],
                    P_out=view(io.P_out, (n, n)))

    def io_fill(self, io, win, with_P=True):
        """Writes a synth.Window into the arena views of io_begin (what a caller's flatten step does in place)."""
        io['poses'][:, 0:9] = win.R_b2w.reshape(win.N, 9)
        io['poses'][:, 9:12] = win.t_b_w
        io['poses'][:, 12:15] = win.t_fej
        io['poses'][:, 15:24] = win.R_b2c.reshape(win.N, 9)
        io['poses'][:, 24:27] = win.t_c_b
        io['poses'][:, 27] = 0.0
        io['obs_ptr'][:] = win.obs_ptr
        if win.F:
            io['p_w'][:] = win.p_w
            io['obs_clone'][:] = win.obs_clone
            io['obs_z'][:] = win.obs_z
            if io['obs_zvel'] is not None:
                io['obs_zvel'][:] = win.obs_zvel
        if with_P:
            io['P'][:] = win.P

    def io_update(self, want_P=True, commit=False):
        """orcvio_msckf_io_update on what stands in the arena; returns the stats (the results are in the io_begin views)."""
        stats = np.zeros(8, dtype=np.int32)
        self._chk(self.lib.orcvio_msckf_io_update(self.h, int(bool(want_P)), int(bool(commit)), _i(stats)), 'orcvio_msckf_io_update')
        return stats

    def io_submit(self, want_P=True, commit=False):
        """orcvio_msckf_io_submit: the update on what stands in the arena is launched; io_collect() waits for it."""
        self._chk(self.lib.orcvio_msckf_io_submit(self.h, int(bool(want_P)), int(bool(commit))), 'orcvio_msckf_io_submit')

    def io_collect(self):
        stats = np.zeros(8, dtype=np.int32)
        self._chk(self.lib.orcvio_msckf_io_collect(self.h, _i(stats)), 'orcvio_msckf_io_collect')
        return stats

    def make_io_call(self, win, resident_cov=False, want_P=True, commit=False):
        """io_begin + the window written into the arena ONCE: returns (call, views) where call() is orcvio_msckf_io_update alone --
        the per-frame cost of a caller that flattens its containers straight into the arena (bench.py's latency modes)."""
        io = self.io_begin(win.flags, win.N, win.F, int(win.obs_ptr[-1]), with_P=not resident_cov)
        self.io_fill(io, win, with_P=not resident_cov)
        lib, h = self.lib, self.h
        stats = np.zeros(8, dtype=np.int32)
        args = (h, int(bool(want_P)), int(bool(commit)), _i(stats))

        def call():
            rc = lib.orcvio_msckf_io_update(*args)
            if rc != 0:
                raise MsckfError(rc, 'orcvio_msckf_io_update')
        call._keep = (io, stats)
        io['stats'] = stats
        return call, io

    def make_update_call(self, win, resident_cov=False, want_P=True, commit=False):
        """The one-shot update with the argument structs marshalled ONCE: returns (call, out) where call() runs
        orcvio_msckf_update_features (and orcvio_msckf_cov_commit if `commit`) on the same buffers -- what a C++ caller's
        per-frame cost is, without this binding's numpy / ctypes marshalling (bench.py's latency modes)."""
        fl, w, t, arrs = self._structs(win)
        out, res = self._result(win.n, win.F)
        if not want_P:
            res.P_out = None
        Pp = None if resident_cov else _d(arrs['P'])
        lib, h = self.lib, self.h
        args = (h, C.byref(fl), C.byref(w), C.byref(t), Pp, C.byref(res))
        keep = (fl, w, t, arrs, res)

        def call():
            rc = lib.orcvio_msckf_update_features(*args)
            if rc == 0 and commit:
                rc = lib.orcvio_msckf_cov_commit(h)
            if rc != 0:
                raise MsckfError(rc, 'orcvio_msckf_update_features')
        call._keep = keep
        self.n, self.F = win.n, win.F
        return call, out

    # -- object blocks (OrcVIO::removeLostObjects) ---------------------------------------------
    def update_objects(self, flags, n_clones, blocks, P, want_G=False):
        """blocks: list of dict(row_clone [rows] int32, Hx6 [rows,6], Hf [rows,no], res [rows])."""
        fl = make_flags(flags)
        n = flags.leg_dim + 6 * n_clones + self.n_extra
        keep = []
        arr = (MsckfObjectRows * max(len(blocks), 1))()
        for k, b in enumerate(blocks):
            rc_ = np.ascontiguousarray(b['row_clone'], dtype=np.int32)
            hx = np.ascontiguousarray(b['Hx6'], dtype=np.float64)
            hf = np.ascontiguousarray(b['Hf'], dtype=np.float64)
            rs = np.ascontiguousarray(b['res'], dtype=np.float64)
            keep += [rc_, hx, hf, rs]
            arr[k] = MsckfObjectRows(len(rs), hf.shape[1] if hf.ndim == 2 else 0, _i(rc_), _d(hx), _d(hf), _d(rs))
        Pc = np.ascontiguousarray(P, dtype=np.float64)
        out, res = self._result(n, 1, False, want_G, False)
        rc = self.lib.orcvio_msckf_update_objects(self.h, C.byref(fl), n_clones, arr, len(blocks), _d(Pc), C.byref(res))
        if rc != 0:
            raise MsckfError(rc, 'orcvio_msckf_update_objects')
        out = self._finish(out, res, 1)
        out['gamma'] = float(out['gamma'][0])
        out['accept'] = int(out['accept'][0])
        return out

    def update_object_lm_msgs(self, flags, n_clones, window_timestamps, R_b2c, t_c_b, msgs, P, fix_D=False, wire_row_major=True):
        """msgs: list of dict(object_id, residual [rows], jacobian_wrt_object_state [rows, no], jacobian_wrt_sensor_state [rows, 6],
        valid_camera_pose_mat [6, frames], timestamps [frames], zs_num_wrt_timestamps [frames]) -- the fields of ObjectLM.msg; the
        2-d arrays are flattened in the wire's order (row-major, or column-major when wire_row_major is False)."""
        fl = make_flags(flags)
        order = 'C' if wire_row_major else 'F'
        arr = (ObjectLMMsg * max(len(msgs), 1))()
        keep = []
        for k, m in enumerate(msgs):
            res = np.ascontiguousarray(m['residual'], dtype=np.float64)
            jo = np.asarray(m['jacobian_wrt_object_state'], dtype=np.float64)
            js = np.asarray(m['jacobian_wrt_sensor_state'], dtype=np.float64)
            pm = np.asarray(m['valid_camera_pose_mat'], dtype=np.float64)
            ts = np.ascontiguousarray(m['timestamps'], dtype=np.float64)
            zn = np.ascontiguousarray(m['zs_num_wrt_timestamps'], dtype=np.int32)
            flat = [np.ascontiguousarray(a.ravel(order=order)) for a in (jo, js, pm)]
            keep += [res, ts, zn] + flat
            arr[k] = ObjectLMMsg(int(m.get('object_id', k)), len(res), jo.shape[1], len(ts), _d(res), _d(flat[0]), _d(flat[1]), _d(flat[2]), _d(ts), _i(zn))
        wt = np.ascontiguousarray(window_timestamps, dtype=np.float64)
        Rb = np.ascontiguousarray(R_b2c, dtype=np.float64)
        tb = np.ascontiguousarray(t_c_b, dtype=np.float64)
        Pc = None if P is None else np.ascontiguousarray(P, dtype=np.float64)
        n = flags.leg_dim + 6 * n_clones + self.n_extra
        out, res_ = self._result(n, 1)
        self.lib.orcvio_msckf_update_object_lm_msgs.argtypes = [C.c_void_p, C.POINTER(MsckfFlags), C.c_int32, _dp, _dp, _dp, C.c_int32, C.c_int32,
                                                                C.POINTER(ObjectLMMsg), C.c_int32, _dp, C.POINTER(MsckfResult)]
        rc = self.lib.orcvio_msckf_update_object_lm_msgs(self.h, C.byref(fl), n_clones, _d(wt), _d(Rb), _d(tb), int(fix_D), int(wire_row_major),
                                                         arr, len(msgs), _d(Pc), C.byref(res_))
        if rc != 0:
            raise MsckfError(rc, 'orcvio_msckf_update_object_lm_msgs')
        out = self._finish(out, res_, 1)
        out['gamma'] = float(out['gamma'][0])
        out['accept'] = int(out['accept'][0])
        return out

    def _object_blocks(self, blocks):
        keep = []
        arr = (MsckfObjectRows * max(len(blocks), 1))()
        for k, b in enumerate(blocks):
            rc_ = np.ascontiguousarray(b['row_clone'], dtype=np.int32)
            hx = np.ascontiguousarray(b['Hx6'], dtype=np.float64)
            hf = np.ascontiguousarray(b['Hf'], dtype=np.float64)
            rs = np.ascontiguousarray(b['res'], dtype=np.float64)
            keep += [rc_, hx, hf, rs]
            arr[k] = MsckfObjectRows(len(rs), hf.shape[1] if hf.ndim == 2 else 0, _i(rc_), _d(hx), _d(hf), _d(rs))
        return arr, keep

    def objects_local(self, flags, n_clones, blocks, P, d_dst=None, stream=None):
        """This rank's objects -> its compressed block (in d_dst or the handle's own block); returns the local dof."""
        fl = make_flags(flags)
        arr, keep = self._object_blocks(blocks)
        Pc = np.ascontiguousarray(P, dtype=np.float64)
        dof = C.c_int32(0)
        rc = self.lib.orcvio_msckf_objects_local(self.h, C.byref(fl), n_clones, arr, len(blocks), _d(Pc),
                                                 C.c_void_p(d_dst) if d_dst else None, C.byref(dof),
                                                 C.c_void_p(stream) if stream else None)
        if rc != 0:
            raise MsckfError(rc, 'orcvio_msckf_objects_local')
        self.n = flags.leg_dim + 6 * n_clones + self.n_extra
        return int(dof.value)

    def objects_finish(self, d_blocks, n_blocks, dof_total, stream=None):
        rc = self.lib.orcvio_msckf_objects_finish(self.h, C.c_void_p(d_blocks), n_blocks, int(dof_total),
                                                  C.c_void_p(stream) if stream else None)
        if rc != 0:
            raise MsckfError(rc, 'orcvio_msckf_objects_finish')

    def objects_download(self, want_G=False):
        out, res = self._result(self.n, 1, False, want_G, False)
        rc = self.lib.orcvio_msckf_objects_download(self.h, C.byref(res))
        if rc != 0:
            raise MsckfError(rc, 'orcvio_msckf_objects_download')
        out = self._finish(out, res, 1)
        out['gamma'] = float(out['gamma'][0])
        out['accept'] = int(out['accept'][0])
        return out

    def object_rows_eval(self, obj, R_b2c, t_c_b, obj_left, new_bbox, vio_left, fix_D=False):
        """Residual rows of one object track (synth.ObjectTrack-shaped) in window coordinates, evaluated on the GPU.
        Returns dict(row_clone, Hx6, Hf, res) ready for update_objects, or None if no frame is in the window."""
        K = obj.kps.shape[0]
        F = len(obj.frames)
        fl = ObjectEvalFlags(int(obj_left), int(new_bbox), int(vio_left), int(fix_D))
        fl.R_b2c[:] = list(np.asarray(R_b2c, dtype=np.float64).ravel())
        fl.t_c_b[:] = list(np.asarray(t_c_b, dtype=np.float64).ravel())
        wTo = np.ascontiguousarray(obj.wTo, dtype=np.float64)
        shape = np.ascontiguousarray(obj.shape, dtype=np.float64)
        kps = np.ascontiguousarray(obj.kps, dtype=np.float64)
        wTc = np.ascontiguousarray(np.stack([fr['wTc'] for fr in obj.frames]), dtype=np.float64)
        zs = np.ascontiguousarray(np.stack([fr['zs'] for fr in obj.frames]), dtype=np.float64)
        bb = np.ascontiguousarray(np.stack([fr['bbox'] for fr in obj.frames]), dtype=np.float64)
        cl = np.ascontiguousarray([fr['clone'] for fr in obj.frames], dtype=np.int32)
        tr = ObjectTrackC(K, F, _d(wTo), _d(shape), _d(kps), _d(wTc), _d(zs), _d(bb), _i(cl))
        cap = F * (2 * K + 4)
        ncol = 9 + 3 * K
        row_clone = np.zeros(cap, dtype=np.int32)
        Hx6 = np.zeros((cap, 6)); Hf = np.zeros((cap, ncol)); res = np.zeros(cap)
        n_rows = C.c_int32(0)
        self.lib.orcvio_msckf_object_rows_eval.argtypes = [C.c_void_p, C.POINTER(ObjectEvalFlags), C.POINTER(ObjectTrackC),
                                                           C.c_int32, C.POINTER(C.c_int32), _ip, _dp, _dp, _dp]
        rc = self.lib.orcvio_msckf_object_rows_eval(self.h, C.byref(fl), C.byref(tr), cap, C.byref(n_rows), _i(row_clone),
                                                    _d(Hx6), _d(Hf), _d(res))
        if rc != 0:
            raise MsckfError(rc, 'orcvio_msckf_object_rows_eval')
        m = n_rows.value
        if m == 0:
            return None
        return dict(row_clone=row_clone[:m].copy(), Hx6=Hx6[:m].copy(), Hf=Hf[:m].copy(), res=res[:m].copy())

    def _object_tracks(self, objs, R_b2c, t_c_b, obj_left, new_bbox, vio_left, fix_D):
        fl = ObjectEvalFlags(int(obj_left), int(new_bbox), int(vio_left), int(fix_D))
        fl.R_b2c[:] = list(np.asarray(R_b2c, dtype=np.float64).ravel())
        fl.t_c_b[:] = list(np.asarray(t_c_b, dtype=np.float64).ravel())
        arr = (ObjectTrackC * max(len(objs), 1))()
        keep = []
        for k, obj in enumerate(objs):
            wTo = np.ascontiguousarray(obj.wTo, dtype=np.float64)
            shape = np.ascontiguousarray(obj.shape, dtype=np.float64)
            kps = np.ascontiguousarray(np.asarray(obj.kps, dtype=np.float64).reshape(-1, 3))
            K = kps.shape[0]
            wTc = np.ascontiguousarray(np.stack([fr['wTc'] for fr in obj.frames]), dtype=np.float64)
            bb = np.ascontiguousarray(np.stack([fr['bbox'] for fr in obj.frames]), dtype=np.float64)
            cl = np.ascontiguousarray([fr['clone'] for fr in obj.frames], dtype=np.int32)
            if K == 0:   # bbox-only track: no keypoints (the library reads neither kps nor frame_zs; one-element dummies keep the pointers valid)
                kps = np.zeros((1, 3))
                zs = np.zeros((len(obj.frames), 1, 2))
            else:
                zs = np.ascontiguousarray(np.stack([np.asarray(fr['zs'], dtype=np.float64).reshape(-1, 2) for fr in obj.frames]))
            keep += [wTo, shape, kps, wTc, zs, bb, cl]
            arr[k] = ObjectTrackC(K, len(obj.frames), _d(wTo), _d(shape), _d(kps), _d(wTc), _d(zs), _d(bb), _i(cl))
        return fl, arr, keep

    def update_object_tracks(self, flags, n_clones, objs, P, R_b2c, t_c_b, obj_left, new_bbox, vio_left, fix_D=False, want_G=False):
        """removeLostObjects straight from object tracks (synth.ObjectTrack-shaped): rows evaluated on the device."""
        fl = make_flags(flags)
        ef, arr, keep = self._object_tracks(objs, R_b2c, t_c_b, obj_left, new_bbox, vio_left, fix_D)
        n = flags.leg_dim + 6 * n_clones + self.n_extra
        Pc = None if P is None else np.ascontiguousarray(P, dtype=np.float64)
        out, res = self._result(n, 1, False, want_G, False)
        rc = self.lib.orcvio_msckf_update_object_tracks(self.h, C.byref(fl), C.byref(ef), n_clones, arr, len(objs), _d(Pc), C.byref(res))
        if rc != 0:
            raise MsckfError(rc, 'orcvio_msckf_update_object_tracks')
        out = self._finish(out, res, 1)
        out['gamma'] = float(out['gamma'][0])
        out['accept'] = int(out['accept'][0])
        return out

    def update_frame(self, win, oflags, objs, R_b2c, t_c_b, obj_left, new_bbox, vio_left, fix_D=False, commit_objects=True):
        """orcvio_msckf_io_update_frame: the feature update of `win` on the resident covariance (arena written in place, committed
        inside the launch) and the object update of `objs` on the covariance it leaves, the objects' compression beside the
        features' solve.  Returns (features, objects) result dicts."""
        io = self.io_begin(win.flags, win.N, win.F, int(win.obs_ptr[-1]), with_P=False)
        self.io_fill(io, win, with_P=False)
        call, outs = self.make_frame_call(win, oflags, objs, R_b2c, t_c_b, obj_left, new_bbox, vio_left, fix_D, commit_objects)
        call()
        return outs()

    def make_frame_call(self, win, oflags, objs, R_b2c, t_c_b, obj_left, new_bbox, vio_left, fix_D=False, commit_objects=True):
        """Arguments of orcvio_msckf_io_update_frame marshalled once (the arena must have been laid out and filled: io_begin /
        io_fill with with_P=False): returns (call, outs) -- call() is the C call alone, outs() the two result dicts."""
        ofl = make_flags(oflags)
        ef, arr, keep = self._object_tracks(objs, R_b2c, t_c_b, obj_left, new_bbox, vio_left, fix_D)
        out_f, res_f = self._result(win.n, win.F)
        res_f.P_out = None
        out_o, res_o = self._result(win.n, 1)
        res_o.P_out = None
        lib, h, nobj, co = self.lib, self.h, len(objs), int(bool(commit_objects))
        hold = (ofl, ef, arr, keep, out_f, res_f, out_o, res_o)

        def call():
            rc = lib.orcvio_msckf_io_update_frame(h, C.byref(res_f), C.byref(ofl), C.byref(ef), arr, nobj, co, C.byref(res_o))
            if rc != 0:
                raise MsckfError(rc, 'orcvio_msckf_io_update_frame')

        def stage():
            # orcvio_msckf_io_stage_object_tracks: the same tracks staged AHEAD of the call (between io_fill and call())
            rc = lib.orcvio_msckf_io_stage_object_tracks(h, C.byref(ofl), C.byref(ef), arr, nobj)
            if rc != 0:
                raise MsckfError(rc, 'orcvio_msckf_io_stage_object_tracks')
        call.stage = stage

        def outs():
            f = self._finish(dict(out_f), res_f, win.F)
            o = self._finish(dict(out_o), res_o, 1)
            o['gamma'] = float(o['gamma'][0])
            o['accept'] = int(o['accept'][0])
            return f, o
        call.hold = hold
        return call, outs

    # -- staged, device-resident form -----------------------------------------------------
    def upload(self, win, without_positions=False, resident_cov=False):
        """without_positions: p_w is not sent (NULL); triangulate_uploaded() must run before the update.
        resident_cov: P is not sent (NULL): the device-resident covariance (cov_*) is the prior."""
        fl, w, t, arrs = self._structs(win)
        if without_positions:
            t.p_w = None
        rc = self.lib.orcvio_msckf_upload(self.h, C.byref(fl), C.byref(w), C.byref(t), None if resident_cov else _d(arrs['P']))
        if rc != 0:
            raise MsckfError(rc, 'orcvio_msckf_upload')
        self.n, self.F = win.n, win.F

    # -- device-resident covariance ------------------------------------------------------------------
    def _chk(self, rc, where):
        if rc != 0:
            raise MsckfError(rc, where)

    def cov_set(self, P):
        Pc = np.ascontiguousarray(P, dtype=np.float64)
        self._chk(self.lib.orcvio_msckf_cov_set(self.h, Pc.shape[0], _d(Pc)), 'orcvio_msckf_cov_set')

    def cov_get(self):
        n = C.c_int32(0)
        self._chk(self.lib.orcvio_msckf_cov_get(self.h, C.byref(n), None), 'orcvio_msckf_cov_get')
        P = np.zeros((n.value, n.value))
        self._chk(self.lib.orcvio_msckf_cov_get(self.h, C.byref(n), _d(P)), 'orcvio_msckf_cov_get')
        return P

    def io_step_frame(self, win, Phi=None, Q=None, augment=True, slam=None, idp_dim=1, prune=None, prune_apply_dx=False, remove=(),
                      raise_on_refusal=True):
        """orcvio_msckf_io_step_frame: ONE filter frame on the resident covariance -- propagation, augmentation, the update on `win`'s
        tracks (+ the in-state features `slam`), the prune update on `prune`'s tracks (a synth.Window sharing win's poses), the
        marginalisation of `remove`.  Returns dict(dx, gamma, accept, stats, prune_dx, prune_gamma, prune_accept, prune_stats, n_after,
        status_first, status_prune, repaired) with copies of the results."""
        io = self.io_begin(win.flags, win.N, win.F, int(win.obs_ptr[-1]), with_P=2)
        self.io_fill(io, win, with_P=False)
        keep = []
        st = FrameStep()
        st.leg_dim = int(win.flags.leg_dim)
        if Phi is not None:
            Ph = np.ascontiguousarray(Phi, dtype=np.float64); Qc = np.ascontiguousarray(Q, dtype=np.float64)
            keep += [Ph, Qc]
            st.Phi = _d(Ph); st.Q = _d(Qc)
        st.augment = int(bool(augment))
        if slam is not None and len(slam) > 0:
            call = self.make_slam_call(idp_dim, slam)
            keep.append(call)
            st.slam_features = C.pointer(call.hold[-1])
        F2 = 0
        if prune is not None:
            F2 = int(prune.F)
            arrs = [np.ascontiguousarray(prune.p_w, dtype=np.float64), np.ascontiguousarray(prune.obs_ptr, dtype=np.int32),
                    np.ascontiguousarray(prune.obs_clone, dtype=np.int32), np.ascontiguousarray(prune.obs_z, dtype=np.float64),
                    np.ascontiguousarray(prune.obs_zvel, dtype=np.float64)]
            tr = MsckfTracks(F2, _d(arrs[0]), _i(arrs[1]), _i(arrs[2]), _d(arrs[3]), _d(arrs[4]) if win.flags.estimate_td else None)
            keep += arrs + [tr]
            st.prune_tracks = C.pointer(tr)
        st.prune_apply_dx = int(bool(prune_apply_dx))
        rm = np.ascontiguousarray(list(remove), dtype=np.int32)
        keep.append(rm)
        if len(rm):
            st.remove_clones = _i(rm)
        st.n_remove = len(rm)
        res = FrameResult()
        rc = self.lib.orcvio_msckf_io_step_frame(self.h, C.byref(st), C.byref(res))
        if rc != 0 and (raise_on_refusal or rc != 6):
            raise MsckfError(rc, 'orcvio_msckf_io_step_frame')
        n = int(io['n'])
        arr = lambda p, k, dt: np.ctypeslib.as_array(p, shape=(max(k, 1),))[:k].astype(dt, copy=True) if p else None
        return dict(rc=rc, dx=arr(res.dx, n, np.float64), gamma=arr(res.gamma, win.F, np.float64), accept=arr(res.accept, win.F, np.int32),
                    stats=np.array(res.stats[:], dtype=np.int32), prune_dx=arr(res.prune_dx, n, np.float64),
                    prune_gamma=arr(res.prune_gamma, F2, np.float64), prune_accept=arr(res.prune_accept, F2, np.int32),
                    prune_stats=np.array(res.prune_stats[:], dtype=np.int32), n_after=int(res.n_after), status_first=int(res.status_first),
                    status_prune=int(res.status_prune), repaired=int(res.repaired))

    def cov_propagate(self, Phi, Q):
        Ph = np.ascontiguousarray(Phi, dtype=np.float64)
        Qc = np.ascontiguousarray(Q, dtype=np.float64)
        self._chk(self.lib.orcvio_msckf_cov_propagate(self.h, Ph.shape[0], _d(Ph), _d(Qc)), 'orcvio_msckf_cov_propagate')

    def cov_augment(self):
        self._chk(self.lib.orcvio_msckf_cov_augment(self.h), 'orcvio_msckf_cov_augment')

    def cov_remove_clones(self, leg_dim, indices):
        ix = np.ascontiguousarray(indices, dtype=np.int32)
        self._chk(self.lib.orcvio_msckf_cov_remove_clones(self.h, leg_dim, _i(ix), len(ix)), 'orcvio_msckf_cov_remove_clones')

    def cov_commit(self):
        self._chk(self.lib.orcvio_msckf_cov_commit(self.h), 'orcvio_msckf_cov_commit')

    def cov_clones_to_nuisance(self, leg_dim, indices):
        ix = np.ascontiguousarray(indices, dtype=np.int32)
        self._chk(self.lib.orcvio_msckf_cov_clones_to_nuisance(self.h, leg_dim, _i(ix), len(ix)), 'orcvio_msckf_cov_clones_to_nuisance')

    def set_object_dof_rank(self, on: bool):
        """ORCVIO_OPT_OBJECT_DOF: 1 = the object gate counts rows - rank(H_f) degrees of freedom (default rows - columns)."""
        self._chk(self.lib.orcvio_msckf_set_option(self.h, 11, int(bool(on))), 'orcvio_msckf_set_option')

    def set_object_refine(self, mode: int):
        """ORCVIO_OPT_OBJECT_REFINE: 0 never, 1 objects with an ill-conditioned H_f (default), 2 every object take the explicit-basis projection."""
        self._chk(self.lib.orcvio_msckf_set_option(self.h, 13, int(mode)), 'orcvio_msckf_set_option')

    def objects_refined(self) -> int:
        """Objects of the last downloaded object update that took the explicit-basis projection."""
        c = C.c_int32(0)
        self.lib.orcvio_msckf_objects_refined.argtypes = [C.c_void_p, _ip]
        self._chk(self.lib.orcvio_msckf_objects_refined(self.h, C.byref(c)), 'orcvio_msckf_objects_refined')
        return int(c.value)

    def set_ref_h2_ldlt(self, on: bool):
        """ORCVIO_OPT_REF_H2_LDLT: the reference's literal H_2.ldlt() in the tail of the hybrid update (default: triangular solve)."""
        self._chk(self.lib.orcvio_msckf_set_option(self.h, 12, int(bool(on))), 'orcvio_msckf_set_option')

    def set_schmidt_states(self, k):
        """ORCVIO_OPT_SCHMIDT_STATES: the last 6 k extra states are Schmidt nuisance states."""
        self._chk(self.lib.orcvio_msckf_set_option(self.h, 10, int(k)), 'orcvio_msckf_set_option')

    def upload_nuisance_poses(self, nui):
        """Poses of the nuisance states (synth.Window.nui-shaped dict), after upload()."""
        arrs = [np.ascontiguousarray(nui[k], dtype=np.float64) for k in ('R_b2w', 't_b_w', 't_fej', 'R_b2c', 't_c_b')]
        wn = MsckfWindow(arrs[0].shape[0], *[_d(a) for a in arrs])
        self._chk(self.lib.orcvio_msckf_upload_nuisance_poses(self.h, C.byref(wn)), 'orcvio_msckf_upload_nuisance_poses')

    def cov_commit_new_features(self):
        """After an update with upload_new_features: the resident covariance becomes the augmented one; returns dx_new."""
        k, d, _ = self._new
        dx_new = np.zeros(d * k)
        self.lib.orcvio_msckf_cov_commit_new_features.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        self._chk(self.lib.orcvio_msckf_cov_commit_new_features(self.h, _d(dx_new)), 'orcvio_msckf_cov_commit_new_features')
        return dx_new

    def cov_prefactor(self):
        """Factor the resident covariance now (asynchronously): the next update finds its prior's square-root factor resident."""
        self._chk(self.lib.orcvio_msckf_cov_prefactor(self.h), 'orcvio_msckf_cov_prefactor')

    # -- feature triangulation (Feature::checkMotion + ::initializePosition) ---------------------
    def _tri_config(self, cfg=None):
        c = TriangulationConfig()
        self.lib.orcvio_msckf_triangulation_config_default(C.byref(c))
        if cfg is not None:   # any object / dict with the reference's OptimizationConfig field names
            get = (lambda k: cfg[k]) if isinstance(cfg, dict) else (lambda k: getattr(cfg, k))
            for k, _ in TriangulationConfig._fields_:
                try:
                    setattr(c, k, get(k))
                except (KeyError, AttributeError):
                    pass
        return c

    def triangulate(self, win, cfg=None, is_initialized=None):
        """Host-buffer call: returns dict(valid, p_w, solution, flags, cost) over the tracks of ``win``."""
        _, w, t, arrs = self._structs(win)
        c = self._tri_config(cfg)
        F = win.F
        ini = None if is_initialized is None else np.ascontiguousarray(is_initialized, dtype=np.int32)
        if ini is None:
            t.p_w = None
        out = dict(valid=np.zeros(max(F, 1), np.int32), p_w=np.zeros((max(F, 1), 3)), solution=np.zeros((max(F, 1), 3)),
                   flags=np.zeros(max(F, 1), np.int32), cost=np.zeros(max(F, 1)))
        res = TriangulationResult(_i(out['valid']), _d(out['p_w']), _d(out['solution']), _i(out['flags']), _d(out['cost']))
        rc = self.lib.orcvio_msckf_triangulate(self.h, C.byref(c), C.byref(w), C.byref(t), _i(ini), C.byref(res))
        if rc != 0:
            raise MsckfError(rc, 'orcvio_msckf_triangulate')
        return {k: v[:F] for k, v in out.items()}

    def triangulate_uploaded(self, cfg=None, is_initialized=None, stream=None):
        """Triangulates the uploaded tracks in place on the device; invalid ones are excluded from the update."""
        c = self._tri_config(cfg)
        ini = None if is_initialized is None else np.ascontiguousarray(is_initialized, dtype=np.int32)
        rc = self.lib.orcvio_msckf_triangulate_uploaded(self.h, C.byref(c), _i(ini), C.c_void_p(stream) if stream else None)
        if rc != 0:
            raise MsckfError(rc, 'orcvio_msckf_triangulate_uploaded')

    def run_update(self, stream=None):
        rc = self.lib.orcvio_msckf_run_update(self.h, C.c_void_p(stream) if stream else None)
        if rc != 0:
            raise MsckfError(rc, 'orcvio_msckf_run_update')

    def run_local(self, stream=None):
        rc = self.lib.orcvio_msckf_run_local(self.h, C.c_void_p(stream) if stream else None)
        if rc != 0:
            raise MsckfError(rc, 'orcvio_msckf_run_local')

    def run_local_to(self, d_dst, stream=None):
        rc = self.lib.orcvio_msckf_run_local_to(self.h, C.c_void_p(d_dst), C.c_void_p(stream) if stream else None)
        if rc != 0:
            raise MsckfError(rc, 'orcvio_msckf_run_local_to')

    def block_ptr(self):
        p = C.c_void_p()
        ne = C.c_int64()
        rc = self.lib.orcvio_msckf_block_ptr(self.h, C.byref(p), C.byref(ne))
        if rc != 0:
            raise MsckfError(rc, 'orcvio_msckf_block_ptr')
        return p.value, ne.value

    def run_finish(self, d_blocks, n_blocks, stream=None):
        rc = self.lib.orcvio_msckf_run_finish(self.h, C.c_void_p(d_blocks), n_blocks,
                                              C.c_void_p(stream) if stream else None)
        if rc != 0:
            raise MsckfError(rc, 'orcvio_msckf_run_finish')

    def sync(self, stream=None):
        rc = self.lib.orcvio_msckf_sync(self.h, C.c_void_p(stream) if stream else None)
        if rc != 0:
            raise MsckfError(rc, 'orcvio_msckf_sync')

    def download(self, want_K=False, want_G=False, want_thin=False):
        out, res = self._result(self.n, self.F, want_K, want_G, want_thin)
        rc = self.lib.orcvio_msckf_download(self.h, C.byref(res))
        if rc != 0:
            raise MsckfError(rc, 'orcvio_msckf_download')
        return self._finish(out, res, self.F)

    def download_dx(self):
        """dx alone of the finished update (P+ stays in HBM: orcvio_msckf_cov_commit makes it the resident covariance)."""
        if getattr(self, '_dx_n', None) != self.n:
            self._dx_buf = np.zeros(self.n)
            self._dx_res = MsckfResult(_d(self._dx_buf), None, None, None, None, None, None, None)
            self._dx_n = self.n
        rc = self.lib.orcvio_msckf_download(self.h, C.byref(self._dx_res))
        if rc != 0:
            raise MsckfError(rc, 'orcvio_msckf_download')
        return self._dx_buf

    def set_stage_profile(self, on: bool):
        """ORCVIO_OPT_STAGE_PROFILE: HIP events between the stages of the object update."""
        self._chk(self.lib.orcvio_msckf_set_option(self.h, 6, int(bool(on))), 'orcvio_msckf_set_option')

    def profile_stages(self):
        """[(stage, ms)] of the last object update run with the stage profile on."""
        names = (C.c_char_p * 32)()
        ms = (C.c_double * 32)()
        cnt = C.c_int32(32)
        self.lib.orcvio_msckf_profile_stages.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), _dp, C.POINTER(C.c_int32)]
        self._chk(self.lib.orcvio_msckf_profile_stages(self.h, names, ms, C.byref(cnt)), 'orcvio_msckf_profile_stages')
        return [(names[i].decode(), ms[i]) for i in range(cnt.value)]

    def profile(self, reps=20, stream=None):
        names = (C.c_char_p * 16)()
        ms = (C.c_double * 16)()
        cnt = C.c_int32(16)
        rc = self.lib.orcvio_msckf_profile_update(self.h, C.c_void_p(stream) if stream else None, reps, names, ms,
                                                  C.byref(cnt))
        if rc != 0:
            raise MsckfError(rc, 'orcvio_msckf_profile_update')
        return {names[i].decode(): ms[i] for i in range(cnt.value)}


def new_feature_rows(win, idp_dim, feats):
    """Host arithmetic (no device): featureJacobian_ekf_new + the W = [V | U] split for features entering the state.
    feats: objects with anchor, inv_param / obs_anchor / inv_depth, p_w, p_fej (optional), obs = [(clone, z, z_vel)].
    Returns (H_top [rows, n], r_top, H_1 [d k, n], H_2 [k, d, d], r_1)."""
    lib = load()
    fl = make_flags(win.flags)
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (win.R_b2w, win.t_b_w, win.t_fej, win.R_b2c, win.t_c_b)]
    wn = MsckfWindow(win.N, *[_d(a) for a in arrs])
    k, d, n = len(feats), int(idp_dim), win.n
    ptr, cl, zz, zv = [0], [], [], []
    for ft in feats:
        for (c, z, v) in ft.obs:
            cl.append(c); zz.append(z); zv.append(v)
        ptr.append(len(cl))
    ia = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    da = lambda a, shape: np.ascontiguousarray(a, dtype=np.float64).reshape(shape)
    anchor, obs_ptr, obs_clone = ia([f.anchor for f in feats]), ia(ptr), ia(cl)
    param = da([f.inv_param if d == 3 else f.obs_anchor for f in feats], (k, 3))
    rho = da([f.inv_depth for f in feats], (k,))
    pw = da([f.p_w for f in feats], (k, 3))
    pf = da([f.p_fej if f.p_fej is not None else f.p_w for f in feats], (k, 3))
    oz, ozv = da(zz, (-1, 2)), da(zv, (-1, 2))
    cap = 2 * len(cl)
    H_top, r_top = np.zeros((cap, n)), np.zeros(cap)
    H_1, H_2, r_1 = np.zeros((d * k, n)), np.zeros((k, d, d)), np.zeros(d * k)
    rows = C.c_int32(0)
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    rc = lib.orcvio_msckf_new_feature_rows(C.byref(fl), C.byref(wn), d, n, k, ip(anchor), _d(param), _d(rho), _d(pw), _d(pf), ip(obs_ptr),
                                           ip(obs_clone), _d(oz), _d(ozv), C.byref(rows), _d(H_top), _d(r_top), _d(H_1), _d(H_2), _d(r_1))
    if rc != 0:
        raise MsckfError(rc, 'orcvio_msckf_new_feature_rows')
    return H_top[:rows.value].copy(), r_top[:rows.value].copy(), H_1, H_2, r_1


def augment_state_nuisance(idp_dim, nui_rows, H_1, H_2, r_1, sigma2, dx, P_upd, ref_ldlt=False):
    """Host arithmetic: augment_state with nui_rows Schmidt nuisance rows at the end (new states go in front of them).
    ref_ldlt: the reference's literal H_2.ldlt() (orcvio_msckf_augment_state_ref_ldlt)."""
    lib = load()
    fn = lib.orcvio_msckf_augment_state_ref_ldlt if ref_ldlt else lib.orcvio_msckf_augment_state_nuisance
    fn.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                                        C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                                        C.POINTER(C.c_double), C.POINTER(C.c_double)]
    d, k, n = int(idp_dim), H_2.shape[0], P_upd.shape[0]
    a = [np.ascontiguousarray(x, dtype=np.float64) for x in (H_1, H_2, r_1, dx, P_upd)]
    dx_new = np.zeros(d * k)
    P_aug = np.zeros((n + d * k, n + d * k))
    rc = fn(n, k, d, int(nui_rows), _d(a[0]), _d(a[1]), _d(a[2]), float(sigma2), _d(a[3]), _d(a[4]), _d(dx_new), _d(P_aug))
    if rc != 0:
        raise MsckfError(rc, 'orcvio_msckf_augment_state_nuisance')
    return dx_new, P_aug


def augment_state(idp_dim, H_1, H_2, r_1, sigma2, dx, P_upd):
    """Host arithmetic (no device): dx_new and the augmented covariance of measurementUpdate_hybrid's tail."""
    lib = load()
    lib.orcvio_msckf_augment_state.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                               C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                               C.POINTER(C.c_double), C.POINTER(C.c_double)]
    d = int(idp_dim)
    k = H_2.shape[0]
    n = P_upd.shape[0]
    a = [np.ascontiguousarray(x, dtype=np.float64) for x in (H_1, H_2, r_1, dx, P_upd)]
    dx_new = np.zeros(d * k)
    P_aug = np.zeros((n + d * k, n + d * k))
    rc = lib.orcvio_msckf_augment_state(n, k, d, _d(a[0]), _d(a[1]), _d(a[2]), float(sigma2), _d(a[3]), _d(a[4]), _d(dx_new), _d(P_aug))
    if rc != 0:
        raise MsckfError(rc, 'orcvio_msckf_augment_state')
    return dx_new, P_aug


def increment_window(win, dx):
    """incrementState_IMUCam (orcvio_msckf_increment_state, host arithmetic) applied to the clone poses and the extrinsics of a
    synth.Window: the window a caller flattens for the NEXT update of the frame.  Returns (window, applied)."""
    import dataclasses
    N = win.N
    s = MsckfState()
    s.R_b2w_imu[:] = list(np.eye(3).reshape(-1))
    s.R_b2c[:] = list(np.ascontiguousarray(win.R_b2c[0]).reshape(-1))
    s.t_c_b[:] = list(win.t_c_b[0])
    s.n_clones = N
    R = np.ascontiguousarray(win.R_b2w, dtype=np.float64).copy()
    t = np.ascontiguousarray(win.t_b_w, dtype=np.float64).copy()
    s.clone_R_b2w = _d(R); s.clone_t_b_w = _d(t)
    fl = make_flags(win.flags)
    dxc = np.ascontiguousarray(dx, dtype=np.float64)
    rc = load().orcvio_msckf_increment_state(C.byref(fl), _d(dxc), C.byref(s))
    if rc < 0:
        raise MsckfError(1, 'orcvio_msckf_increment_state')
    Rc = np.array(s.R_b2c[:]).reshape(3, 3)
    tc = np.array(s.t_c_b[:])
    return dataclasses.replace(win, R_b2w=R, t_b_w=t, R_b2c=np.ascontiguousarray(np.repeat(Rc[None], N, 0)),
                               t_c_b=np.ascontiguousarray(np.repeat(tc[None], N, 0))), rc == 1


def chi2_quantile(dof, prob=0.95):
    return float(load().orcvio_msckf_chi2_quantile(int(dof), float(prob)))


def debug_read(upd: MsckfUpdater, which: str):
    """Test helper: copies an intermediate device buffer of the last run to the host."""
    lib = upd.lib
    lib.orcvio_msckf_debug_read.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
    dims = np.zeros(8, dtype=np.int32)
    rc = lib.orcvio_msckf_debug_read(upd.h, 7, dims.ctypes.data_as(C.c_void_p), dims.nbytes)
    if rc != 0:
        raise MsckfError(rc, 'debug_read dims')
    n, NA, NAP, NP, m_tot = (int(x) for x in dims[:5])
    ldz = int(dims[6])
    if which == 'dims':
        return dict(n=n, NA=NA, NAP=NAP, NP=NP, m_tot=m_tot, Mmax=int(dims[5]), ldz=ldz, reg_path=int(dims[7]))
    shapes = {'Hs': (0, (m_tot, NAP)), 'Ab': (1, (NAP, NAP)), 'A': (2, (NAP, NAP)), 'RP': (3, (NP, NP)),
              'M': (4, (NP, NP)), 'RM': (5, (NP, NP)), 'Z': (6, (n, ldz)), 'U': (8, (NP, NP))}
    code, shape = shapes[which]
    out = np.zeros(shape)
    if out.size:
        rc = lib.orcvio_msckf_debug_read(upd.h, code, out.ctypes.data_as(C.c_void_p), out.nbytes)
        if rc != 0:
            raise MsckfError(rc, f'debug_read {which}')
    return out


def debug_potrf(upd: MsckfUpdater, X, tol_rel=0.0, force_lds_path=False):
    """Test helper: Cholesky factor (lower), block inverses and pivot info of a host matrix."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    n = X.shape[0]
    L = np.zeros((n, n))
    nb = (n + 15) // 16
    Dinv = np.zeros((nb, 16, 16))
    info = np.zeros(2, dtype=np.int32)
    lib = upd.lib
    lib.orcvio_msckf_debug_potrf.argtypes = [C.c_void_p, _dp, C.c_int32, C.c_double, C.c_int32, _dp, _dp, _ip]
    rc = lib.orcvio_msckf_debug_potrf(upd.h, _d(X), n, float(tol_rel), int(force_lds_path), _d(L), _d(Dinv), _i(info))
    if rc != 0:
        raise MsckfError(rc, 'debug_potrf')
    return L, Dinv, info


def debug_trsm(upd: MsckfUpdater, X, B):
    """Test helper: Z = chol(X)^-1 B."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    B = np.ascontiguousarray(B, dtype=np.float64)
    n, nrhs = B.shape
    Z = np.zeros((n, nrhs))
    lib = upd.lib
    lib.orcvio_msckf_debug_trsm.argtypes = [C.c_void_p, _dp, C.c_int32, _dp, C.c_int32, _dp]
    rc = lib.orcvio_msckf_debug_trsm(upd.h, _d(X), n, _d(B), nrhs, _d(Z))
    if rc != 0:
        raise MsckfError(rc, 'debug_trsm')
    return Z


def debug_potrf_solve(upd: MsckfUpdater, X, B, la=0, stamps=False, reps=0):
    """Test helper / diagnostic: L = chol(X) and Z = L^-1 B in ONE launch -- k_potrf_solve (la = 0) or k_potrf_solve_la
    (la = 2, 3: the trailing update spread over far workgroups).  Returns dict(L, Z, info[dropped, non-positive, lost], stamps, us)."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    B = np.ascontiguousarray(B, dtype=np.float64)
    n, nrhs = B.shape
    L = np.zeros((n, n))
    Z = np.zeros((n, nrhs))
    info = np.zeros(3, dtype=np.int32)
    st = np.zeros(512, dtype=np.uint64)
    us = C.c_double(0.0)
    lib = upd.lib
    lib.orcvio_msckf_debug_potrf_solve.argtypes = [C.c_void_p, _dp, C.c_int32, _dp, C.c_int32, C.c_int32, _dp, _dp, C.c_void_p, C.c_void_p,
                                                   C.c_int32, C.POINTER(C.c_double)]
    rc = lib.orcvio_msckf_debug_potrf_solve(upd.h, _d(X), n, _d(B), nrhs, int(la), _d(L), _d(Z), info.ctypes.data,
                                            st.ctypes.data if stamps else None, int(reps), C.byref(us))
    if rc != 0:
        raise MsckfError(rc, 'debug_potrf_solve')
    return dict(L=L, Z=Z, info=info, stamps=st.astype(np.int64) if stamps else None, us=us.value)
