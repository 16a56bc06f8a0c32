"""ctypes doorway onto the CPU oracle (oracle/_build/liblfx_oracle.so) and, when built, the
reference pieces (oracle/_ref/libref_pieces.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# LFX_ORACLE_LIB: another build of the same sources (the sanitizer build, `make -C oracle asan`)
_LIB = os.environ.get("LFX_ORACLE_LIB") or os.path.join(_HERE, "_build", "liblfx_oracle.so")
_REF = os.path.join(_HERE, "_ref", "libref_pieces.so")

LABEL_NAMES = ["Default", "Edge", "EdgeNeighbor", "Surface", "SurfaceNeighbor", "OutOfRange",
               "Occluded", "ParallelBeam"]
LABEL = {n: i for i, n in enumerate(LABEL_NAMES)}


class Params(C.Structure):
    """extraction/include/lidar_feature_extraction/hyper_parameter.hpp:32-65"""
    _fields_ = [("padding", C.c_int32), ("neighbor_degree_threshold", C.c_double),
                ("distance_diff_threshold", C.c_double), ("parallel_beam_min_range_ratio", C.c_double),
                ("edge_threshold", C.c_double), ("surface_threshold", C.c_double),
                ("min_range", C.c_double), ("max_range", C.c_double), ("n_blocks", C.c_int32)]


def default_params():
    """code defaults, hyper_parameter.hpp:35-43"""
    return Params(5, 2.0, 0.3, 0.02, 0.05, 0.05, 0.1, 100.0, 6)


def launch_params():
    """lidar_feature_launch/config/lidar_feature_extraction.param.yaml:3-10"""
    return Params(2, 3.0, 0.3, 0.02, 50.0, 0.05, 0.1, 1000.0, 6)


def build(force=False):
    if os.environ.get("LFX_ORACLE_LIB"):
        return _LIB
    srcs = [os.path.join(_HERE, f) for f in ("lfx_oracle.cpp", "lfx_oracle_loc.cpp", "lfx_oracle.h")]
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []) + ["all"])
    return _LIB


def build_ref():
    """Compile the reference's own math.cpp / convolution.cpp / index_range.cpp where they lie
    (only possible in the build container, where /root/reference exists)."""
    if os.path.isdir("/root/reference/extraction/src"):
        subprocess.check_call(["make", "-C", _HERE, "-s", "ref"])
    return _REF if os.path.exists(_REF) else None


_lib = None
_ref = None

_d, _i, _f, _u8 = C.c_double, C.c_int, C.c_float, C.c_uint8
_pd, _pi, _pf, _pu8 = (C.POINTER(t) for t in (_d, _i, _f, _u8))


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.orc_xy_norm.restype = _d
        L.orc_xy_norm.argtypes = [_d, _d]
        L.orc_calc_radian.argtypes = [_d, _d, _d, _d, _pd]
        L.orc_inner_product.restype = _d
        L.orc_inner_product.argtypes = [_pd, _pd, _i]
        L.orc_convolution1d.argtypes = [_pd, _i, _pd, _i, _pd]
        L.orc_make_weight.argtypes = [_i, _pd]
        L.orc_make_weight.restype = None
        L.orc_calc_curvature.argtypes = [_pd, _i, _i, _pd]
        L.orc_argsort.argtypes = [_pd, _i, _pi]
        L.orc_argsort.restype = None
        L.orc_index_range.argtypes = [_i, _i, _i, _pi]
        L.orc_padded_index_range.argtypes = [_i, _i, _i, _pi]
        L.orc_loc_triplet_cross.argtypes = [_pd, _pd, _pd, _pd]
        L.orc_loc_triplet_cross.restype = None
        L.orc_loc_mean_cov.argtypes = [_pd, _i, _pd, _pd]
        L.orc_loc_mean_cov.restype = None
        L.orc_loc_principal.argtypes = [_pd, _pd, _pd]
        L.orc_loc_principal.restype = None
        L.orc_loc_principal_is_reliable.argtypes = [_pd]
        L.orc_loc_solve_linear.argtypes = [_pd, _i, _i, _pd, _pd]
        L.orc_loc_solve_linear.restype = None
        L.orc_loc_quaternion.argtypes = [_pd, _pd]
        L.orc_loc_quaternion.restype = None
        L.orc_loc_edge_residuals.argtypes = [_pf, _i, _pd, _i, _pf, _i, _pd, _pd]
        L.orc_loc_edge_residuals.restype = None
        L.orc_loc_surface_residuals.argtypes = [_pf, _i, _pd, _i, _pf, _i, _pd, _pd]
        L.orc_loc_surface_residuals.restype = None
        L.orc_loc_drp_dq.argtypes = [_pd, _pd, _pd]
        L.orc_loc_drp_dq.restype = None
        L.orc_loc_nearest.argtypes = [_pf, _i, _pd, _i, _pd, _pd, _pi]
        L.orc_loc_nearest.restype = None
        for name in ("orc_loc_median", "orc_loc_mad", "orc_loc_scale"):
            getattr(L, name).argtypes = [_pd, _i]
            getattr(L, name).restype = _d
        for name in ("orc_loc_huber", "orc_loc_huber_derivative"):
            getattr(L, name).argtypes = [_d, _d]
            getattr(L, name).restype = _d
        L.orc_loc_is_degenerate.argtypes = [_pd, _i, _d]
        for name in ("orc_loc_angle_axis_to_quaternion", "orc_loc_rotation_matrix", "orc_loc_make_m"):
            getattr(L, name).argtypes = [_pd, _pd]
            getattr(L, name).restype = None
        L.orc_loc_pairs_update.argtypes = [_pd, _pd, _i, _pd, _pd, _pd]
        L.orc_loc_pairs_update.restype = None
        L.orc_loc_optimize_pairs.argtypes = [_pd, _pd, _i, _pd, _i, _pd, _pd, _pd, _pi, _pi]
        L.orc_loc_optimize_scan.argtypes = [_pf, _i, _pf, _i, _i, _pf, _i, _pf, _i, _pd, _i, _pd, _pd, _pd, _pi, _pi]
        L.orc_voxel_downsample.argtypes = [_pf, _i, _f, _pf, _pi]
        L.orc_range_message.argtypes = [_i, C.c_char_p, C.c_char_p, C.c_longlong, C.c_longlong, C.c_char_p, C.c_size_t]
        L.orc_irange.argtypes = [_i, _pi]
        L.orc_irange.restype = None
        L.orc_mapped_points_at.argtypes = [_pd, _i, _pi, _i, _i, _i, _i, _pd, _pi]
        L.orc_ring_message.argtypes = [_i, _i, C.POINTER(Params), C.c_char_p, C.c_size_t]
        L.orc_polar_less_f64.argtypes = [_d, _d, _d, _d]
        L.orc_polar_less_f32.argtypes = [_f, _f, _f, _f]
        L.orc_sort_by_atan2_f64.argtypes = [_pd, _pd, _i, _pi]
        L.orc_sort_by_atan2_f64.restype = None
        L.orc_is_neighbor_xy.argtypes = [_f, _f, _f, _f, _d, _pi]
        L.orc_is_in_inclusive_range.argtypes = [_d, _d, _d]
        for name in ("orc_fill_from_left", "orc_fill_from_right", "orc_fill_neighbors"):
            getattr(L, name).argtypes = [_pu8, _i, _pi, _pf, _pf, _d, _i, _i, _u8]
        for name in ("orc_edge_label_assign", "orc_surface_label_assign"):
            getattr(L, name).argtypes = [_pu8, _pd, _i, _pi, _pf, _pf, _d, _i, _d]
        L.orc_assign_label.argtypes = [_pu8, _pd, _i, _pf, _pf, _d, _i, _i, _d, _d]
        L.orc_label_occluded.argtypes = [_pu8, _i, _pf, _pf, _d, _i, _d]
        L.orc_label_out_of_range.argtypes = [_pu8, _i, _pf, _pf, _d, _d]
        L.orc_label_out_of_range.restype = None
        L.orc_label_parallel_beam.argtypes = [_pu8, _i, _pf, _pf, _d]
        L.orc_label_parallel_beam.restype = None
        L.orc_label_to_color.argtypes = [_u8, _pu8]
        L.orc_label_to_color.restype = None
        L.orc_extract.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t,
                                  C.c_size_t, C.POINTER(Params), _i] + [C.c_void_p] * 6 + [C.c_int32] + \
                                 [C.c_void_p] * 8
        _lib = L
    return _lib


def ref_pieces():
    """The reference's own compiled math.cpp / convolution.cpp / index_range.cpp, or None."""
    global _ref
    if _ref is None and os.path.exists(_REF):
        R = C.CDLL(_REF)
        R.ref_xy_norm.restype = _d
        R.ref_xy_norm.argtypes = [_d, _d]
        R.ref_calc_radian.argtypes = [_d, _d, _d, _d, _pd]
        R.ref_inner_product.restype = _d
        R.ref_inner_product.argtypes = [_pd, _pd, _i]
        R.ref_convolution1d.argtypes = [_pd, _i, _pd, _i, _pd]
        R.ref_padded_index_range.argtypes = [_i, _i, _i, _pi]
        _ref = R
    return _ref


def ptr(a, t):
    return None if a is None else a.ctypes.data_as(t)


# ----------------------------------------------------------------------------- whole scan
POINT_DTYPE = np.dtype({"names": ["x", "y", "z", "pad", "intensity", "ring"],
                        "formats": ["<f4", "<f4", "<f4", "<f4", "<f4", "<u2"],
                        "offsets": [0, 4, 8, 12, 16, 20], "itemsize": 32})
"""PointXYZIR wire/AoS layout: lib/include/lidar_feature_library/point_type.hpp:62-86,
point_type_converter/point_type_converter/convert.py:134-145"""


def extract(points, params=None, canonical_ties=True, max_rings=65536):
    """Run the oracle over one scan given as a POINT_DTYPE structured array."""
    L = lib()
    params = params or default_params()
    pts = np.ascontiguousarray(points)
    assert pts.dtype == POINT_DTYPE
    n = len(pts)
    out = {
        "labels": np.zeros(n, np.uint8), "curvature": np.zeros(n, np.float64),
        "sorted_index": np.zeros(n, np.int32),
        "ring_id": np.zeros(max_rings, np.int32), "ring_count": np.zeros(max_rings, np.int32),
        "ring_status": np.zeros(max_rings, np.int32),
        "edge_index": np.zeros(n, np.int32), "surface_index": np.zeros(n, np.int32),
        "edge_points": np.zeros((n, 4), np.float32), "surface_points": np.zeros((n, 4), np.float32),
    }
    n_rings, n_edge, n_surf = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    ties = np.zeros(2, np.int64)

    def vp(a):
        return a.ctypes.data_as(C.c_void_p)

    rc = L.orc_extract(vp(pts), n, 32, 0, 4, 8, 20, C.byref(params), int(bool(canonical_ties)),
                       vp(out["labels"]), vp(out["curvature"]), vp(out["sorted_index"]),
                       vp(out["ring_id"]), vp(out["ring_count"]), vp(out["ring_status"]), max_rings,
                       C.cast(C.byref(n_rings), C.c_void_p), vp(out["edge_index"]),
                       C.cast(C.byref(n_edge), C.c_void_p), vp(out["surface_index"]),
                       C.cast(C.byref(n_surf), C.c_void_p), vp(out["edge_points"]),
                       vp(out["surface_points"]), vp(ties))
    if rc != 0:
        raise ValueError("orc_extract: bad arguments")
    nr, ne, ns = n_rings.value, n_edge.value, n_surf.value
    for k in ("ring_id", "ring_count", "ring_status"):
        out[k] = out[k][:nr].copy()
    out["edge_index"] = out["edge_index"][:ne].copy()
    out["surface_index"] = out["surface_index"][:ns].copy()
    out["edge_points"] = out["edge_points"][:ne].copy()
    out["surface_points"] = out["surface_points"][:ns].copy()
    out["angle_ties"], out["curvature_ties"] = int(ties[0]), int(ties[1])
    return out
