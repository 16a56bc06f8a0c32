"""ctypes binding of the C ABI in include/lfx.h (liblfx.so: HIP kernels + host side).

There is no fallback: if the library has not been built, or no MI355X is present when a
context is created, this raises.  Build with `python __graft_entry__.py` (or
`make -C lidar_feature_extraction_amd/csrc`).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LFX_LIB_PATH") or os.path.join(_HERE, "_lib", "liblfx.so")   # override: A/B builds only

LFX_N_KERNELS = 12
MAX_RINGS = 256

STAGE_LABEL, STAGE_OCCLUSION, STAGE_OUT_OF_RANGE, STAGE_PARALLEL_BEAM = 1, 2, 4, 8
STAGE_SINGLE_BLOCK, STAGE_CURVATURE, STAGE_ALL = 16, 32, 47


class LfxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("lfx error %d: %s" % (code, msg))
        self.code = code


class Params(C.Structure):
    """lfx_params == HyperParameters, extraction/include/lidar_feature_extraction/hyper_parameter.hpp:32-65"""
    _fields_ = [("padding", C.c_int32), ("neighbor_degree_threshold", C.c_double),
                ("distance_diff_threshold", C.c_double), ("parallel_beam_min_range_ratio", C.c_double),
                ("edge_threshold", C.c_double), ("surface_threshold", C.c_double),
                ("min_range", C.c_double), ("max_range", C.c_double), ("n_blocks", C.c_int32)]


class Layout(C.Structure):
    _fields_ = [("point_step", C.c_uint32), ("off_x", C.c_uint32), ("off_y", C.c_uint32),
                ("off_z", C.c_uint32), ("off_ring", C.c_uint32), ("ring_datatype", C.c_uint32),
                ("big_endian", C.c_uint32)]


class PointField(C.Structure):
    _fields_ = [("name", C.c_char_p), ("offset", C.c_uint32), ("datatype", C.c_uint8), ("count", C.c_uint32)]


# sensor_msgs/msg/PointField datatype codes
INT8, UINT8, INT16, UINT16, INT32, UINT32, FLOAT32, FLOAT64 = range(1, 9)


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("max_points_per_scan", C.c_uint32), ("max_batch", C.c_uint32),
                ("max_points_per_ring", C.c_uint32), ("max_rings", C.c_uint32), ("drop_zero_points", C.c_uint32),
                ("layout", Layout), ("outputs", C.c_uint32), ("stream_hint", C.c_uint32),
                ("ring_ids", C.POINTER(C.c_uint16)), ("n_ring_ids", C.c_uint32)]


class GatherStep(C.Structure):
    """lfx_gather_step (include/lfx.h): one step of a grouped exchange, lfx_gather_payload2."""
    _fields_ = [("dst", C.c_int), ("slot", C.c_uint32), ("d_edge", C.c_void_p), ("d_surface", C.c_void_p), ("d_offsets", C.c_void_p),
                ("d_edge_all", C.c_void_p), ("d_surface_all", C.c_void_p), ("d_offsets_all", C.c_void_p), ("counts_out", C.c_void_p)]


OUT_FEATURES, OUT_LABELS, OUT_CURVATURE, OUT_SORTED_INDEX, OUT_ALL = 1, 2, 4, 8, 15
STREAM_UNKNOWN, STREAM_TURNED_RINGS, STREAM_NO_GRID, STREAM_GRID_WITH_HOLES = 0, 1, 2, 3


class ScanResult(C.Structure):
    _fields_ = [("n_points", C.c_uint32), ("labels", C.POINTER(C.c_uint8)), ("curvature", C.POINTER(C.c_double)),
                ("sorted_index", C.POINTER(C.c_uint32)), ("n_sorted", C.c_uint32), ("n_rings", C.c_uint32),
                ("ring_id", C.POINTER(C.c_uint16)), ("ring_count", C.POINTER(C.c_uint32)),
                ("ring_offset", C.POINTER(C.c_uint32)), ("ring_status", C.POINTER(C.c_uint8)),
                ("n_edge", C.c_uint32), ("edge_points", C.POINTER(C.c_float)), ("edge_index", C.POINTER(C.c_uint32)),
                ("n_surface", C.c_uint32), ("surface_points", C.POINTER(C.c_float)),
                ("surface_index", C.POINTER(C.c_uint32))]


class AlignResult(C.Structure):
    """lfx_align_result (OptimizationResult, optimization_result.hpp:35-43)."""
    _fields_ = [("pose", C.c_double * 12), ("error", C.c_double), ("error_scale", C.c_double), ("iteration", C.c_int32),
                ("code", C.c_int32)]


class DeviceView(C.Structure):
    _fields_ = [("batch", C.c_uint32), ("max_rings", C.c_uint32), ("ring_capacity", C.c_uint32)] + \
        [(n, C.c_void_p) for n in (
            "scan_begin", "labels_sorted", "curvature_sorted", "sorted_index", "scan_info",
            "ring_count", "ring_status", "edge_points", "edge_index", "surface_points", "surface_index")]


EXPORTS = [
    "lfx_default_params", "lfx_launch_params", "lfx_create", "lfx_destroy", "lfx_last_error",
    "lfx_status_string", "lfx_ring_message", "lfx_range_message", "lfx_extract", "lfx_extract_submit", "lfx_extract_wait", "lfx_extract_batch", "lfx_extract_batch_device",
    "lfx_device_results", "lfx_batch_status", "lfx_scan_routes", "lfx_host_alloc", "lfx_host_free", "lfx_comm_unique_id", "lfx_comm_create",
    "lfx_comm_destroy", "lfx_comm_stats", "lfx_gather_counts", "lfx_gather_payload", "lfx_gather", "lfx_voxel_downsample", "lfx_downsample_surface",
    "lfx_map_create", "lfx_map_create_host", "lfx_map_destroy", "lfx_map_info", "lfx_map_nearest",
    "lfx_scan_to_map_residuals", "lfx_edge_residuals", "lfx_align_message", "lfx_scan_to_map_align", "lfx_align_point_pairs",
    "lfx_localize_batch", "lfx_localize_host",
    "lfx_layout_from_fields", "lfx_pack_xyz", "lfx_pack_xyz12", "lfx_pack_colored", "lfx_pack_features", "lfx_download_scan", "lfx_stage_ring", "lfx_stage_convolution1d",
    "lfx_stage_ring_projection", "lfx_label_to_color", "lfx_color_points_by_label", "lfx_set_profiling", "lfx_set_profiling_interval", "lfx_kernel_times", "lfx_kernel_name",
    "lfx_route_choice", "lfx_set_log_callback", "lfx_box_calibration", "lfx_gather_counts_slot", "lfx_gather_payload2", "lfx_set_ring_ids",
]
"""Every symbol include/lfx.h declares (tests/test_abi.py checks the library exports each)."""

HOOKS_LIB_PATH = os.path.join(_HERE, "_lib", "liblfx_testhooks.so")
"""Test infrastructure: the same library built -DLFX_TEST_HOOKS -- the only build whose lfx_create reads the LFX_DEBUG_*
switches (route pins, span variants, ablation flags) and whose gather takes another RCCL (LFX_RCCL_LIB)."""

_libs = {}


def load(test_hooks=False):
    """The library (test_hooks: its test-hooks build; with LFX_LIB_PATH set, that file either way)."""
    path = LIB_PATH if (not test_hooks or os.environ.get("LFX_LIB_PATH")) else HOOKS_LIB_PATH
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise ImportError(
            "liblfx.so is not built (%s): run `python __graft_entry__.py` -- there is no CPU fallback" % path)
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 (SONAME libamdhip64.so.7,
    # the name liblfx.so needs).  Loading torch first makes liblfx.so bind to that copy; the other
    # order would put a second runtime into the process, and the later one sees no GPU.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int
    L.lfx_default_params.argtypes = [C.POINTER(Params)]
    L.lfx_default_params.restype = None
    L.lfx_launch_params.argtypes = [C.POINTER(Params)]
    L.lfx_launch_params.restype = None
    L.lfx_create.argtypes = [C.POINTER(vp), i32, C.POINTER(Params), C.POINTER(Config)]
    L.lfx_destroy.argtypes = [vp]
    L.lfx_destroy.restype = None
    L.lfx_last_error.argtypes = [vp]
    L.lfx_last_error.restype = C.c_char_p
    L.lfx_status_string.argtypes = [i32]
    L.lfx_status_string.restype = C.c_char_p
    L.lfx_ring_message.argtypes = [i32, u32, C.POINTER(Params), C.c_char_p, C.c_size_t]
    L.lfx_range_message.argtypes = [i32, C.c_char_p, C.c_char_p, C.c_longlong, C.c_longlong, C.c_char_p, C.c_size_t]
    L.lfx_extract.argtypes = [vp, vp, C.c_size_t, C.POINTER(ScanResult)]
    L.lfx_extract_submit.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_uint64)]
    L.lfx_extract_wait.argtypes = [vp, C.c_uint64, C.POINTER(ScanResult)]
    L.lfx_extract_batch.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t), u32, C.POINTER(ScanResult)]
    L.lfx_extract_batch_device.argtypes = [vp, vp, C.POINTER(u32), u32, vp]
    L.lfx_device_results.argtypes = [vp, C.POINTER(DeviceView)]
    L.lfx_batch_status.argtypes = [vp, vp, C.POINTER(u32)]
    L.lfx_scan_routes.argtypes = [vp, vp, vp]
    L.lfx_host_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.lfx_host_free.argtypes = [vp, vp]
    L.lfx_host_free.restype = None
    L.lfx_comm_unique_id.argtypes = [vp]
    L.lfx_comm_create.argtypes = [vp, vp, i32, i32, C.POINTER(vp)]
    L.lfx_comm_destroy.argtypes = [vp]
    L.lfx_comm_destroy.restype = None
    L.lfx_comm_stats.argtypes = [vp, vp]
    L.lfx_gather_counts.argtypes = [vp, vp, vp, u32, vp]
    L.lfx_gather_payload.argtypes = [vp, vp, i32, vp, vp, vp, u32, u32, vp, vp, vp, C.c_size_t, vp, vp]
    L.lfx_gather_counts_slot.argtypes = [vp, vp, u32, vp, u32, vp]
    L.lfx_gather_payload2.argtypes = [vp, vp, C.POINTER(GatherStep), u32, u32, u32, C.c_size_t, vp]
    L.lfx_voxel_downsample.argtypes = [vp, vp, vp, vp, u32, u32, C.c_size_t, C.c_float, vp, vp, vp, vp]
    L.lfx_map_create.argtypes = [vp, vp, u32, C.c_float, C.POINTER(vp), vp]
    L.lfx_map_create_host.argtypes = [vp, vp, u32, C.c_float, C.POINTER(vp), vp]
    L.lfx_map_destroy.argtypes = [vp]
    L.lfx_map_destroy.restype = None
    L.lfx_map_info.argtypes = [vp, C.POINTER(u32), C.POINTER(C.c_float), C.POINTER(i32)]
    L.lfx_map_nearest.argtypes = [vp, vp, vp, u32, u32, vp, vp, vp, vp]
    L.lfx_scan_to_map_residuals.argtypes = [vp, i32, vp, C.POINTER(C.c_double), u32, vp, vp, vp, u32, u32, u32, vp, vp, vp]
    L.lfx_edge_residuals.argtypes = [vp, vp, C.POINTER(C.c_double), u32, vp, vp, vp]
    L.lfx_align_message.argtypes = [i32]
    L.lfx_align_message.restype = C.c_char_p
    pd, pres = C.POINTER(C.c_double), C.POINTER(AlignResult)
    L.lfx_scan_to_map_align.argtypes = [vp, vp, vp, u32, i32, vp, vp, vp, u32, u32, C.c_size_t, vp, vp, vp, u32, u32,
                                        C.c_size_t, u32, pd, pres, vp]
    L.lfx_align_point_pairs.argtypes = [vp, vp, vp, vp, vp, u32, C.c_size_t, u32, i32, pd, pres, vp]
    L.lfx_localize_batch.argtypes = [vp, vp, vp, u32, i32, C.c_float, u32, pd, pres, vp]
    L.lfx_localize_host.argtypes = [vp, vp, vp, u32, i32, C.c_float, vp, u32, vp, u32, pd, pres, vp]
    L.lfx_downsample_surface.argtypes = [vp, C.c_float, vp, vp, vp, vp]
    L.lfx_gather.argtypes = [vp, vp, i32, vp, vp, vp, u32, u32, vp, vp, vp, C.c_size_t, vp, vp]
    L.lfx_layout_from_fields.argtypes = [C.POINTER(PointField), C.c_uint32, C.c_uint32, C.c_int, C.POINTER(Layout)]
    L.lfx_pack_xyz.argtypes = [vp, vp, vp, vp, C.c_size_t, vp]
    L.lfx_pack_xyz12.argtypes = [vp, vp, vp, vp, C.c_size_t, vp]
    L.lfx_pack_colored.argtypes = [vp, vp, vp, C.c_size_t, vp]
    L.lfx_pack_features.argtypes = [vp, vp, vp, vp, C.c_size_t, vp]
    L.lfx_download_scan.argtypes = [vp, u32, vp, C.POINTER(ScanResult)]
    L.lfx_stage_ring.argtypes = [vp, C.POINTER(Params), u32, u32] + [vp] * 10
    L.lfx_stage_convolution1d.argtypes = [vp, vp, u32, vp, u32, vp]
    L.lfx_stage_ring_projection.argtypes = [vp, vp, C.c_size_t, vp, C.POINTER(u32), vp, vp]
    L.lfx_label_to_color.argtypes = [C.c_uint8, C.POINTER(C.c_uint8)]
    L.lfx_color_points_by_label.argtypes = [vp, vp, C.c_size_t, vp, vp]
    L.lfx_set_profiling.argtypes = [vp, i32]
    L.lfx_set_profiling_interval.argtypes = [vp, C.c_uint32]
    L.lfx_kernel_times.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.lfx_kernel_name.argtypes = [i32]
    L.lfx_kernel_name.restype = C.c_char_p
    L.lfx_box_calibration.argtypes = [vp, C.c_size_t, vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    _libs[path] = L
    return L


def check(ctx, rc, lib=None):
    if rc != 0:
        raise LfxError(rc, ((lib or load()).lfx_last_error(ctx) or b"").decode())
