"""ctypes binding of libvo_mi355x.so (C ABI declared in include/vo_mi355x.h).

The library is the product: hand-written HIP kernels for gfx950.  There is no
CPU fallback -- if the shared object is missing or no HIP device is present the
calls raise.  Nothing in this package imports anything from `oracle/`.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(_HERE)
# (VO_MI355X_LIB: another build of the same sources -- an experiment compiled beside the shipped library, csrc/Makefile -- for A/B runs in one call)
LIB_PATH = os.environ.get("VO_MI355X_LIB") or os.path.join(PKG_ROOT, "lib", "libvo_mi355x.so")
CSRC_DIR = os.path.join(PKG_ROOT, "csrc")

VO_OK = 0
ERRORS = {-1: "VO_E_INVALID", -2: "VO_E_HIP", -3: "VO_E_NOMEM", -4: "VO_E_STATE", -5: "VO_E_CAPACITY",
          -6: "VO_E_NUMERIC"}


class VoError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (ERRORS.get(code, "VO_E_?"), code, msg))
        self.code = code


class KltParams(C.Structure):
    _fields_ = [("win", C.c_int32), ("max_level", C.c_int32), ("max_count", C.c_int32),
                ("epsilon", C.c_double), ("min_eig_threshold", C.c_float), ("_pad", C.c_int32)]


class StParams(C.Structure):
    _fields_ = [("max_corners", C.c_int32), ("block_size", C.c_int32), ("quality_level", C.c_double),
                ("min_distance", C.c_double), ("use_harris", C.c_int32), ("_pad", C.c_int32), ("harris_k", C.c_double)]


class BaParams(C.Structure):
    _fields_ = [("max_iters", C.c_int32), ("_pad", C.c_int32), ("ftol", C.c_double), ("xtol", C.c_double),
                ("gtol", C.c_double), ("lambda0", C.c_double), ("huber_delta", C.c_double), ("lambda_min", C.c_double)]


TUNING_FIELDS = ("ba_kernels", "ba_lanes", "ba_threads", "ba_pitch_pad", "ba_chunks", "ba_workgroups", "ba_workgroup_cap", "ba_fold", "klt_waves", "klt_pair",
                 "st_two_kernels", "st_band_rows", "st_separate_nms", "st_keep_eig", "st_host_limit", "xcd_remap_off", "gate_groups", "reserve_cus",
                 "gather_workgroups")


class Tuning(C.Structure):
    """vo_tuning: forced forms for parity tests and A/B measurements; 0 = the library's rule"""
    _fields_ = [(k, C.c_int32) for k in TUNING_FIELDS] + [("reserved", C.c_int32 * 13)]


class PnpParams(C.Structure):
    _fields_ = [("reproj_err", C.c_double), ("confidence", C.c_double), ("max_iters", C.c_int32), ("seed", C.c_int32)]


class PnpStats(C.Structure):
    _fields_ = [("cost", C.c_double), ("n_inliers", C.c_int32), ("hypotheses", C.c_int32), ("best", C.c_int32),
                ("status", C.c_int32)]


class EssParams(C.Structure):
    _fields_ = [("threshold", C.c_double), ("prob", C.c_double), ("distance_thresh", C.c_double), ("max_iters", C.c_int32),
                ("seed", C.c_int32)]


class EssStats(C.Structure):
    _fields_ = [("n_inliers", C.c_int32), ("n_good", C.c_int32), ("hypotheses", C.c_int32), ("best", C.c_int32),
                ("status", C.c_int32), ("_pad", C.c_int32)]


class SiftKp(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("size", C.c_float), ("angle", C.c_float), ("response", C.c_float),
                ("octave", C.c_int32)]


class BaStats(C.Structure):
    _fields_ = [("cost0", C.c_double), ("cost", C.c_double), ("lam", C.c_double), ("iters", C.c_int32),
                ("accepted", C.c_int32), ("status", C.c_int32), ("n_obs", C.c_int32)]


class PipeParams(C.Structure):
    _fields_ = [("ba_window", C.c_int32), ("min_track_length", C.c_int32), ("mask_radius", C.c_int32), ("max_new", C.c_int32),
                ("pnp_blind_batches", C.c_int32), ("ba_budget", C.c_int32), ("resurrect", C.c_int32), ("reserved", C.c_int32),
                ("max_reproj_err", C.c_double),
                ("min_bearing_angle", C.c_double), ("klt", KltParams), ("st", StParams), ("ba", BaParams), ("pnp", PnpParams)]


class PipeRecord(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("t", "status", "overflow", "n_landmarks", "n_candidates", "n_dead", "n_dead_total", "n_tracked",
                                         "pnp_inliers", "pnp_hypotheses", "pnp_bound_reached", "n_ripe", "n_new", "n_resurrected",
                                         "n_detected", "ba_landmarks", "ba_observations", "ba_iters", "ba_accepted", "ba_status", "ba_done",
                                         "t_final")] + [("ba_cost0", C.c_double), ("ba_cost", C.c_double), ("H", C.c_double * 12),
                                                        ("H_final", C.c_double * 12)]


_u8p, _i16p, _i32p = C.POINTER(C.c_uint8), C.POINTER(C.c_int16), C.POINTER(C.c_int32)
_f32p, _f64p = C.POINTER(C.c_float), C.POINTER(C.c_double)
_ctx = C.c_void_p

# name -> (restype, argtypes); must list every symbol include/vo_mi355x.h declares
SIGNATURES = {
    "vo_abi_version": (C.c_int32, []),
    "vo_device_count": (C.c_int32, [_i32p]),
    "vo_ctx_create": (C.c_int32, [C.c_int32] * 6 + [C.POINTER(_ctx)]),
    "vo_ctx_create_batched": (C.c_int32, [C.c_int32] * 7 + [C.POINTER(_ctx)]),
    "vo_pyramid_read_seq": (C.c_int32, [_ctx, C.c_int32, C.c_int32, C.c_int32, _u8p, _i16p]),
    "vo_ctx_destroy": (C.c_int32, [_ctx]),
    "vo_last_error": (C.c_char_p, [_ctx]),
    "vo_sync": (C.c_int32, [_ctx]),
    "vo_frame_push": (C.c_int32, [_ctx, _u8p, C.c_int32]),
    "vo_set_prefilter": (C.c_int32, [_ctx, C.c_int32, C.c_double, C.c_double]),
    "vo_seq_upload": (C.c_int32, [_ctx, _u8p, C.c_int32]),
    "vo_frame_push_resident": (C.c_int32, [_ctx, C.c_int32]),
    "vo_pyramid_level_size": (C.c_int32, [_ctx, C.c_int32, _i32p, _i32p]),
    "vo_pyramid_read": (C.c_int32, [_ctx, C.c_int32, C.c_int32, _u8p, _i16p]),
    "vo_klt_default_params": (C.c_int32, [C.POINTER(KltParams)]),
    "vo_klt_track": (C.c_int32, [_ctx, _f32p, C.c_int32, C.POINTER(KltParams), _f32p, _u8p, _f32p, _i32p]),
    "vo_points_upload": (C.c_int32, [_ctx, _f32p, C.c_int32]),
    "vo_points_download": (C.c_int32, [_ctx, _f32p, _u8p, _f32p, _i32p, C.c_int32]),
    "vo_klt_track_resident": (C.c_int32, [_ctx, C.c_int32, C.POINTER(KltParams)]),
    "vo_st_default_params": (C.c_int32, [C.POINTER(StParams)]),
    "vo_shi_tomasi": (C.c_int32, [_ctx, _f32p, C.c_int32, C.c_int32, _u8p, C.POINTER(StParams), _f32p, _i32p]),
    "vo_shi_tomasi_resident": (C.c_int32, [_ctx, C.c_int32, C.c_int32, C.POINTER(StParams)]),
    "vo_shi_tomasi_fetch": (C.c_int32, [_ctx, _f32p, _i32p]),
    "vo_shi_tomasi_read": (C.c_int32, [_ctx, _f32p, _u8p, _i32p]),
    "vo_triangulate_dlt": (C.c_int32, [_ctx, _f32p, _f32p, _f32p, _f32p, C.c_int32, _f32p, _f64p, _f64p, _f64p,
                                       _f64p, _f64p]),
    "vo_dlt_upload": (C.c_int32, [_ctx, _f32p, _f32p, _f32p, _f32p, C.c_int32, _f64p, _f64p, _f64p]),
    "vo_dlt_resident": (C.c_int32, [_ctx]),
    "vo_dlt_fetch": (C.c_int32, [_ctx, _f32p, _f64p, _f64p]),
    "vo_frame_step_resident": (C.c_int32, [_ctx, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                           C.POINTER(KltParams), C.POINTER(StParams), C.POINTER(BaParams)]),
    "vo_frame_step_host": (C.c_int32, [_ctx, C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                       C.POINTER(KltParams), C.POINTER(StParams), C.POINTER(BaParams)]),
    "vo_host_alloc": (C.c_int32, [C.c_uint64, C.POINTER(C.c_void_p)]),
    "vo_host_free": (C.c_int32, [C.c_void_p]),
    "vo_frame_fetch": (C.c_int32, [_ctx, C.c_int32, _f32p, _u8p, _f32p, _f32p, _f64p, _f64p, _f64p, _f64p,
                                   C.POINTER(BaStats), _f32p, _i32p]),
    "vo_get_tuning": (C.c_int32, [_ctx, C.POINTER(Tuning)]),
    "vo_set_tuning": (C.c_int32, [_ctx, C.POINTER(Tuning)]),
    "vo_set_graph_mode": (C.c_int32, [_ctx, C.c_int32]),
    "vo_set_side_stream": (C.c_int32, [_ctx, C.c_int32]),
    "vo_step_layout": (C.c_int32, [_ctx, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "vo_profile_enable": (C.c_int32, [_ctx, C.c_int32]),
    "vo_profile_read": (C.c_int32, [_ctx, C.c_int32, _f64p, _i32p]),
    "vo_debug_cycles": (C.c_int32, [_ctx, C.c_int32, C.POINTER(C.c_int64)]),
    "vo_ba_default_params": (C.c_int32, [C.POINTER(BaParams)]),
    "vo_ba_adjust": (C.c_int32, [_ctx, _f64p, _f64p, _f64p, _f64p, C.c_int32, C.c_int32, C.POINTER(BaParams),
                                 _f64p, _f64p, C.POINTER(BaStats)]),
    "vo_ba_upload": (C.c_int32, [_ctx, _f64p, _f64p, _f64p, _f64p, C.c_int32, C.c_int32]),
    "vo_ba_solve_resident": (C.c_int32, [_ctx, C.POINTER(BaParams)]),
    "vo_ba_upload_bank": (C.c_int32, [_ctx, _f64p, _f64p, _f64p, _f64p, C.c_int32, C.c_int32, C.c_int32]),
    "vo_ba_select_problem": (C.c_int32, [_ctx, C.c_int32]),
    "vo_ba_fetch": (C.c_int32, [_ctx, _f64p, _f64p, C.POINTER(BaStats)]),
    "vo_comm_unique_id": (C.c_int32, [_u8p]),
    "vo_comm_init": (C.c_int32, [_ctx, C.c_int32, C.c_int32, _u8p]),
    "vo_comm_destroy": (C.c_int32, [_ctx]),
    "vo_ba_set_sharded": (C.c_int32, [_ctx, C.c_int32]),
    "vo_ba_gather_points": (C.c_int32, [_ctx, _f64p]),
    "vo_pnp_default_params": (C.c_int32, [C.POINTER(PnpParams)]),
    "vo_pnp_ransac": (C.c_int32, [_ctx, _f64p, _f32p, _f32p, C.c_int32, C.POINTER(PnpParams), _f64p, _f64p, _u8p,
                                  C.POINTER(PnpStats)]),
    "vo_essential_default_params": (C.c_int32, [C.POINTER(EssParams)]),
    "vo_essential_ransac": (C.c_int32, [_ctx, _f64p, _f32p, _f32p, C.c_int32, C.POINTER(EssParams), _f64p, _f64p, _f64p, _u8p,
                                        C.POINTER(EssStats)]),
    "vo_sift_detect_compute": (C.c_int32, [_ctx, _u8p, C.c_int32, _u8p, C.c_int32, C.c_int32, C.POINTER(SiftKp), _f32p, _i32p]),
    "vo_match_knn2": (C.c_int32, [_ctx, _f32p, C.c_int32, _f32p, C.c_int32, C.c_int32, _i32p, _f32p]),
    "vo_pnp_upload": (C.c_int32, [_ctx, _f64p, _f32p, _f32p, C.c_int32]),
    "vo_pnp_solve_resident": (C.c_int32, [_ctx, C.POINTER(PnpParams), C.c_int32]),
    "vo_pnp_fetch": (C.c_int32, [_ctx, _f64p, _f64p, _u8p, C.POINTER(PnpStats)]),
    "vo_tracks_seed": (C.c_int32, [_ctx, _f32p, C.c_int32, C.c_int32]),
    "vo_tracks_track": (C.c_int32, [_ctx, C.c_int32, C.POINTER(KltParams)]),
    "vo_tracks_detect": (C.c_int32, [_ctx, C.c_int32, C.c_int32, C.POINTER(StParams), C.c_int32]),
    "vo_tracks_read": (C.c_int32, [_ctx, _i32p, _f32p, _f32p, _i32p, _i32p, _i32p, _i32p, _i32p]),
    "vo_tracks_obs": (C.c_int32, [_ctx, C.c_int32, C.c_int32, _f64p]),
    "vo_ba_obs_from_tracks": (C.c_int32, [_ctx, C.c_int32]),
    "vo_pipe_default_params": (C.c_int32, [C.POINTER(PipeParams)]),
    "vo_pipe_create": (C.c_int32, [_ctx, _f64p, C.POINTER(PipeParams)]),
    "vo_pipe_table_bytes": (C.c_int32, [_ctx, C.c_int32, C.POINTER(C.c_uint64)]),
    "vo_pipe_table_write": (C.c_int32, [_ctx, C.c_int32, C.c_void_p]),
    "vo_pipe_table_read": (C.c_int32, [_ctx, C.c_int32, C.c_void_p]),
    "vo_pipe_commit": (C.c_int32, [_ctx]),
    "vo_pipe_step_host": (C.c_int32, [_ctx, C.POINTER(C.c_void_p), C.c_int32, C.c_int32]),
    "vo_pipe_step": (C.c_int32, [_ctx, C.c_int32, C.c_int32]),
    "vo_pipe_fetch": (C.c_int32, [_ctx, C.POINTER(PipeRecord)]),
    "vo_pipe_set_ba_budget": (C.c_int32, [_ctx, C.c_int32]),
    "vo_pipe_lists_bytes": (C.c_int32, [_ctx, C.POINTER(C.c_uint64)]),
    "vo_pipe_lists_read": (C.c_int32, [_ctx, C.c_void_p]),
    "vo_pipe_rows_read": (C.c_int32, [_ctx, C.c_int32, _i32p, C.c_int32, C.c_void_p]),
    "vo_pipe_inliers_read": (C.c_int32, [_ctx, _u8p, C.c_int32]),
    "vo_ba_probe": (C.c_int32, [_ctx, C.c_double, C.c_double, _f64p, _i32p, _f64p, _f64p, _f64p, _f64p, _f64p,
                                _f64p, _f64p, _f64p, _f64p]),
}

_lib = None


def load():
    """Load the shared library (raises if it has not been built -- run __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VoError(-2, "libvo_mi355x.so not built (%s); run `python -c 'import __graft_entry__ as g; "
                              "g.build()'` -- there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        missing = [n for n in SIGNATURES if not hasattr(L, n)]
        if missing:
            raise VoError(-2, "libvo_mi355x.so lacks symbols %s -- stale build?" % missing)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def ptr(a, t):
    if a is None:
        return None
    return a.ctypes.data_as(C.POINTER(t))


def as_c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)
