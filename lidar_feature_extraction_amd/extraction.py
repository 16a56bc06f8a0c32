"""Host-side mirror of the reference's extraction node for tests, the bench and Python users.

The reference operator is the body of FeatureExtraction::Callback
(/root/reference/extraction/app/feature_extraction.cpp:114-157); its construction reads the
nine HyperParameters (hyper_parameter.hpp:32-65).  `FeatureExtraction` here keeps those names:
construct with HyperParameters, call ExtractFeatures(cloud) per scan.  Everything runs in the
HIP library through the C ABI (binding.py); nothing is computed in Python.
"""
import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

from . import binding as B
from .synth import POINT_DTYPE

LABEL_NAMES = ["Default", "Edge", "EdgeNeighbor", "Surface", "SurfaceNeighbor", "OutOfRange",
               "Occluded", "ParallelBeam"]          # point_label.hpp:32-42
RING_STATUS_NAMES = {0: "ok", 1: "sparse", 2: "too_few_for_convolution", 3: "too_few_for_blocks",
                     4: "block_too_small", 5: "zero_norm_pair", 7: "too_large"}


@dataclass
class HyperParameters:
    """hyper_parameter.hpp:35-43 (code defaults)."""
    padding: int = 5
    neighbor_degree_threshold: float = 2.0
    distance_diff_threshold: float = 0.3
    parallel_beam_min_range_ratio: float = 0.02
    edge_threshold: float = 0.05
    surface_threshold: float = 0.05
    min_range: float = 0.1
    max_range: float = 100.0
    n_blocks: int = 6

    @staticmethod
    def launch_yaml():
        """lidar_feature_launch/config/lidar_feature_extraction.param.yaml:3-10"""
        return HyperParameters(padding=2, neighbor_degree_threshold=3.0, edge_threshold=50.0, max_range=1000.0)

    def to_c(self):
        return B.Params(self.padding, self.neighbor_degree_threshold, self.distance_diff_threshold,
                        self.parallel_beam_min_range_ratio, self.edge_threshold, self.surface_threshold,
                        self.min_range, self.max_range, self.n_blocks)


@dataclass
class ScanFeatures:
    labels: np.ndarray          # u8 [n], original point order
    curvature: np.ndarray       # f64 [n], original point order
    sorted_index: np.ndarray    # u32 [n], rings ascending / angle ascending
    ring_id: np.ndarray
    ring_count: np.ndarray
    ring_offset: np.ndarray
    ring_status: np.ndarray
    edge_points: np.ndarray     # f32 [n_edge,4]: x y z (float)curvature
    edge_index: np.ndarray      # u32 [n_edge] original indices
    surface_points: np.ndarray
    surface_index: np.ndarray

    @property
    def edge_xyz(self):
        """what the node publishes on scan_edge (ToPointXYZ, feature_extraction.cpp:163)"""
        return self.edge_points[:, :3]

    @property
    def surface_xyz(self):
        return self.surface_points[:, :3]


def _np(ptr, n, dtype, shape=None):
    if n == 0 or not ptr:                  # (a NULL pointer: that output was not asked for, lfx_config.outputs)
        return np.zeros((0,) + tuple(shape[1:]) if shape else 0, dtype)
    a = np.ctypeslib.as_array(ptr, shape=(n,) if shape is None else shape).astype(dtype, copy=True)
    return a


def _result(r):
    ne, ns, n, nr = r.n_edge, r.n_surface, r.n_points, r.n_rings
    return ScanFeatures(
        labels=_np(r.labels, n, np.uint8), curvature=_np(r.curvature, n, np.float64),
        sorted_index=_np(r.sorted_index, r.n_sorted, np.uint32),
        ring_id=_np(r.ring_id, nr, np.uint16), ring_count=_np(r.ring_count, nr, np.uint32),
        ring_offset=_np(r.ring_offset, nr, np.uint32), ring_status=_np(r.ring_status, nr, np.uint8),
        edge_points=_np(r.edge_points, ne, np.float32, (ne, 4)) if ne else np.zeros((0, 4), np.float32),
        edge_index=_np(r.edge_index, ne, np.uint32),
        surface_points=_np(r.surface_points, ns, np.float32, (ns, 4)) if ns else np.zeros((0, 4), np.float32),
        surface_index=_np(r.surface_index, ns, np.uint32))


class FeatureExtraction:
    """One context = one GPU = one calling thread (feature_extraction.cpp:65-87,185)."""

    def __init__(self, params=None, device=0, max_points_per_scan=262144, max_batch=1,
                 max_points_per_ring=0, max_rings=0, drop_zero_points=False, layout=None, outputs=0, stream_hint=0, test_hooks=None,
                 ring_ids=None):
        # test_hooks: the context comes from the test-hooks build of the library (B.HOOKS_LIB_PATH), the only one whose
        # lfx_create reads the LFX_DEBUG_* switches (tests and tools/ only).  None: that build exactly when such a switch is
        # set in the environment -- the shipped library would not see it
        if test_hooks is None:
            test_hooks = any(k.startswith("LFX_DEBUG_") for k in os.environ)
        self._L = B.load(bool(test_hooks))
        self.params = params or HyperParameters()
        self._ctx = C.c_void_p()
        # layout: (point_step, off_x, off_y, off_z, off_ring[, ring_datatype, big_endian]) of the records or a
        # B.Layout (layout_from_fields), None = PointXYZIR
        if layout is None:
            lay = B.Layout(0, 0, 0, 0, 0, 0, 0)
        elif isinstance(layout, B.Layout):
            lay = layout
        else:
            lay = B.Layout(*(tuple(layout) + (0, 0))[:7])
        self._step = lay.point_step or 32
        # outputs: B.OUT_* mask of what ExtractFeatures / extract_batch bring back (0 = everything); the two clouds always do
        # stream_hint: B.STREAM_* (what the caller knows about the order its driver publishes in; spares the first batch a slower route)
        # ring_ids: the sensor's ring ids where they are not 0 .. max_rings-1 (lfx_config.ring_ids); None: the host entry points
        # look them up in a scan that carries another id
        ids = None if ring_ids is None else (C.c_uint16 * len(ring_ids))(*[int(r) for r in ring_ids])
        cfg = B.Config(C.sizeof(B.Config), max_points_per_scan, max_batch, max_points_per_ring, max_rings, int(bool(drop_zero_points)), lay,
                       int(outputs), int(stream_hint), ids, 0 if ids is None else len(ring_ids))
        self._pinned = []
        cp = self.params.to_c()
        rc = self._L.lfx_create(C.byref(self._ctx), device, C.byref(cp), C.byref(cfg))
        if rc != 0:
            self._ctx = C.c_void_p()
            raise B.LfxError(rc, (self._L.lfx_last_error(None) or b"").decode())
        self.max_batch = max_batch

    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            for ptr in getattr(self, "_pinned", []):
                self._L.lfx_host_free(self._ctx, ptr)
            self._pinned = []
            self._L.lfx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- the operator ------------------------------------------------------------------
    def ExtractFeatures(self, cloud):
        """cloud: POINT_DTYPE array (PointXYZIR records).  Returns ScanFeatures."""
        return self.extract_batch([cloud])[0]

    def extract_batch(self, clouds):
        clouds = [np.ascontiguousarray(c) for c in clouds]
        for c in clouds:
            if c.dtype.itemsize != self._step:
                raise TypeError("clouds must be arrays of %d-byte records (POINT_DTYPE for PointXYZIR)" % self._step)
        nb = len(clouds)
        ptrs = (C.c_void_p * nb)(*[c.ctypes.data for c in clouds])
        ns = (C.c_size_t * nb)(*[len(c) for c in clouds])
        res = (B.ScanResult * nb)()
        B.check(self._ctx, self._L.lfx_extract_batch(self._ctx, ptrs, ns, nb, res), self._L)
        return [_result(res[i]) for i in range(nb)]

    def submit(self, cloud):
        """lfx_extract_submit: queue one scan (upload, kernels, download) and return its ticket at once; at most two
        tickets may be outstanding.  A pinned `cloud` (pinned_like) must stay untouched until wait(ticket) returns."""
        cloud = np.ascontiguousarray(cloud)
        if cloud.dtype.itemsize != self._step:
            raise TypeError("clouds must be arrays of %d-byte records (POINT_DTYPE for PointXYZIR)" % self._step)
        t = C.c_uint64(0)
        B.check(self._ctx, self._L.lfx_extract_submit(self._ctx, C.c_void_p(cloud.ctypes.data), len(cloud), C.byref(t)), self._L)
        return int(t.value)

    def wait(self, ticket, raw=False):
        """lfx_extract_wait: the ScanFeatures of that ticket (raw=True: the ctypes result, nothing copied)."""
        res = B.ScanResult()
        B.check(self._ctx, self._L.lfx_extract_wait(self._ctx, C.c_uint64(int(ticket)), C.byref(res)), self._L)
        return res if raw else _result(res)

    def pinned_like(self, cloud):
        """A copy of `cloud` in pinned host memory (lfx_host_alloc): lfx_extract reads such a buffer by DMA instead of
        staging it.  Owned by this object (freed by close())."""
        cloud = np.ascontiguousarray(cloud)
        ptr = C.c_void_p()
        B.check(self._ctx, self._L.lfx_host_alloc(self._ctx, max(cloud.nbytes, 1), C.byref(ptr)), self._L)
        self._pinned.append(ptr)
        buf = (C.c_uint8 * cloud.nbytes).from_address(ptr.value)
        out = np.frombuffer(buf, dtype=cloud.dtype, count=len(cloud))
        out[...] = cloud
        return out

    def voxel_downsample(self, d_points, d_begin, d_count, count_stride, n_clouds, total_points, leaf, d_out, d_out_count,
                         d_status, stream=0):
        """lfx_voxel_downsample: Downsample (pcl::VoxelGrid, downsample.hpp:37-51) of device clouds; all pointers are device addresses."""
        B.check(self._ctx, self._L.lfx_voxel_downsample(
            self._ctx, C.c_void_p(int(d_points)), C.c_void_p(int(d_begin)), C.c_void_p(int(d_count)), int(count_stride),
            int(n_clouds), int(total_points), float(leaf), C.c_void_p(int(d_out)), C.c_void_p(int(d_out_count)),
            C.c_void_p(int(d_status)), C.c_void_p(int(stream))))

    def downsample_surface(self, leaf, d_out, d_out_count, d_status, stream=0):
        """lfx_downsample_surface: the surface clouds of the last device batch, as the localizer downsamples them (surface.hpp:111)."""
        B.check(self._ctx, self._L.lfx_downsample_surface(
            self._ctx, float(leaf), C.c_void_p(int(d_out)), C.c_void_p(int(d_out_count)), C.c_void_p(int(d_status)),
            C.c_void_p(int(stream))))

    def make_map(self, d_points, n_points, cell_size=1.0, stream=0):
        """lfx_map_create: the map a scan is matched against (KDTreeEigen, kdtree.hpp:50-71); cell_size 0 = no grid."""
        return ScanMap(self, d_points, n_points, cell_size, stream)

    def make_map_from_host(self, points, cell_size=1.0, stream=0):
        """lfx_map_create_host: points [n][4] float32 on the host."""
        return ScanMap(self, 0, 0, cell_size, stream, host_points=points)

    def scan_to_map_residuals(self, kind, scan_map, pose, n_neighbors, d_points, d_begin, d_count, count_stride, n_clouds,
                              max_points_per_cloud, d_residual, d_jacobian, stream=0):
        """lfx_scan_to_map_residuals: kind 0 = edge rows (edge.hpp:86-124), 1 = surface rows (surface.hpp:116-139);
        pose: 3 x 4 [R | t] (point_to_map); device addresses otherwise."""
        pm = np.ascontiguousarray(pose, np.float64).reshape(12)
        B.check(self._ctx, self._L.lfx_scan_to_map_residuals(
            self._ctx, int(kind), scan_map.handle, pm.ctypes.data_as(C.POINTER(C.c_double)), int(n_neighbors),
            C.c_void_p(int(d_points)), C.c_void_p(int(d_begin)), C.c_void_p(int(d_count)), int(count_stride), int(n_clouds),
            int(max_points_per_cloud), C.c_void_p(int(d_residual)), C.c_void_p(int(d_jacobian)), C.c_void_p(int(stream))))

    def edge_residuals(self, scan_map, pose, n_neighbors, d_residual, d_jacobian, stream=0):
        """lfx_edge_residuals: the edge clouds of the last device batch against an edge map."""
        pm = np.ascontiguousarray(pose, np.float64).reshape(12)
        B.check(self._ctx, self._L.lfx_edge_residuals(
            self._ctx, scan_map.handle, pm.ctypes.data_as(C.POINTER(C.c_double)), int(n_neighbors),
            C.c_void_p(int(d_residual)), C.c_void_p(int(d_jacobian)), C.c_void_p(int(stream))))

    def _align_results(self, res):
        out = []
        for r in res:
            out.append(dict(pose=np.array(r.pose[:], np.float64).reshape(3, 4), error=r.error, error_scale=r.error_scale,
                            iteration=r.iteration, code=r.code, success=r.code <= 2,
                            message=self._L.lfx_align_message(r.code).decode()))
        return out

    def scan_to_map_align(self, edge_map, surface_map, n_neighbors, max_iter, d_edge_points,
                          d_edge_begin, d_edge_count, edge_count_stride, max_edge, total_edge, d_surface_points, d_surface_begin,
                          d_surface_count, surface_count_stride, max_surface, total_surface, initial_poses, stream=0):
        """lfx_scan_to_map_align: Optimizer<LOAMOptimizationProblem>::Run (optimizer.hpp:79-123) per scan; initial_poses
        [n][3][4]; returns one dict per scan (pose, error, error_scale, iteration, code, success, message)."""
        poses = np.ascontiguousarray(initial_poses, np.float64).reshape(-1, 12)
        res = (B.AlignResult * len(poses))()
        v = lambda a: C.c_void_p(int(a))   # noqa: E731
        B.check(self._ctx, self._L.lfx_scan_to_map_align(
            self._ctx, edge_map.handle, surface_map.handle, int(n_neighbors), int(max_iter),
            v(d_edge_points), v(d_edge_begin), v(d_edge_count), int(edge_count_stride), int(max_edge), int(total_edge),
            v(d_surface_points), v(d_surface_begin), v(d_surface_count), int(surface_count_stride), int(max_surface),
            int(total_surface), len(poses), poses.ctypes.data_as(C.POINTER(C.c_double)), res, v(stream)))
        return self._align_results(res)

    def align_point_pairs(self, d_source, d_target, d_begin, d_count, max_points, total_points, max_iter, initial_poses, stream=0):
        """lfx_align_point_pairs: the same optimizer on AlignmentProblem (alignment.cpp:33-78)."""
        poses = np.ascontiguousarray(initial_poses, np.float64).reshape(-1, 12)
        res = (B.AlignResult * len(poses))()
        v = lambda a: C.c_void_p(int(a))   # noqa: E731
        B.check(self._ctx, self._L.lfx_align_point_pairs(
            self._ctx, v(d_source), v(d_target), v(d_begin), v(d_count), int(max_points), int(total_points), len(poses),
            int(max_iter), poses.ctypes.data_as(C.POINTER(C.c_double)), res, v(stream)))
        return self._align_results(res)

    def localize_batch(self, edge_map, surface_map, initial_poses, n_neighbors=15, max_iter=20, surface_leaf=1.0, stream=0):
        """lfx_localize_batch: Localizer::Update (localizer.hpp:71-80) for every scan of the last device batch."""
        poses = np.ascontiguousarray(initial_poses, np.float64).reshape(-1, 12)
        res = (B.AlignResult * len(poses))()
        B.check(self._ctx, self._L.lfx_localize_batch(
            self._ctx, edge_map.handle, surface_map.handle, int(n_neighbors), int(max_iter), float(surface_leaf), len(poses),
            poses.ctypes.data_as(C.POINTER(C.c_double)), res, C.c_void_p(int(stream))))
        return self._align_results(res)

    def localize_host(self, edge_map, surface_map, edge_points, surface_points, initial_pose, n_neighbors=15, max_iter=20,
                      surface_leaf=1.0, stream=0):
        """lfx_localize_host: Localizer::Update for one scan whose clouds ([n][4] float32) are on the host."""
        e = np.ascontiguousarray(edge_points, np.float32).reshape(-1, 4)
        sf = np.ascontiguousarray(surface_points, np.float32).reshape(-1, 4)
        pose = np.ascontiguousarray(initial_pose, np.float64).reshape(12)
        res = (B.AlignResult * 1)()
        B.check(self._ctx, self._L.lfx_localize_host(
            self._ctx, edge_map.handle, surface_map.handle, int(n_neighbors), int(max_iter), float(surface_leaf),
            C.c_void_p(e.ctypes.data), len(e), C.c_void_p(sf.ctypes.data), len(sf), pose.ctypes.data_as(C.POINTER(C.c_double)), res,
            C.c_void_p(int(stream))))
        return self._align_results(res)[0]

    def set_ring_ids(self, ring_ids):
        """lfx_set_ring_ids: the sensor's ring ids for the device path (None: back to 0 .. max_rings-1)."""
        if ring_ids is None:
            B.check(self._ctx, self._L.lfx_set_ring_ids(self._ctx, None, 0), self._L)
            return
        ids = (C.c_uint16 * len(ring_ids))(*[int(r) for r in ring_ids])
        B.check(self._ctx, self._L.lfx_set_ring_ids(self._ctx, ids, len(ring_ids)), self._L)

    def scan_routes(self, n_scans, stream=0):
        """lfx_scan_routes: per scan of the last batch 1 = read in place, 2 = in place through ring transforms, 3 = in place as a
        grid with (0, 0, 0) records the zero filter dropped, 0 = bucketed."""
        out = np.zeros(n_scans, np.uint8)
        B.check(self._ctx, self._L.lfx_scan_routes(self._ctx, C.c_void_p(int(stream)), C.c_void_p(out.ctypes.data)), self._L)
        return out

    def batch_status(self, stream=0):
        """lfx_batch_status: raises LfxError if a scan of the last device batch carries an error bit."""
        bad = C.c_uint32(0)
        B.check(self._ctx, self._L.lfx_batch_status(self._ctx, C.c_void_p(int(stream)), C.byref(bad)), self._L)

    def extract_batch_device(self, d_points, n_points, stream=0):
        """d_points: device address of the scans' records back to back; asynchronous on `stream`."""
        n = np.ascontiguousarray(n_points, dtype=np.uint32)
        B.check(self._ctx, self._L.lfx_extract_batch_device(
            self._ctx, C.c_void_p(int(d_points)), n.ctypes.data_as(C.POINTER(C.c_uint32)), len(n),
            C.c_void_p(int(stream))))

    def device_view(self):
        v = B.DeviceView()
        B.check(self._ctx, self._L.lfx_device_results(self._ctx, C.byref(v)), self._L)
        return v

    def pack_features(self, d_edge_out, d_surface_out, d_offsets_out, capacity_points, stream=0):
        """Pack the last device batch's clouds into caller-owned device buffers (see lfx.h)."""
        B.check(self._ctx, self._L.lfx_pack_features(
            self._ctx, C.c_void_p(int(d_edge_out)), C.c_void_p(int(d_surface_out)), C.c_void_p(int(d_offsets_out)),
            int(capacity_points), C.c_void_p(int(stream))))

    def pack_xyz(self, d_edge_out, d_surface_out, d_offsets_out, capacity_points, stream=0):
        """As pack_features, but pcl::PointXYZ wire records (x, y, z, 1.0f): scan_edge / scan_surface payloads."""
        B.check(self._ctx, self._L.lfx_pack_xyz(
            self._ctx, C.c_void_p(int(d_edge_out)), C.c_void_p(int(d_surface_out)), C.c_void_p(int(d_offsets_out)),
            int(capacity_points), C.c_void_p(int(stream))))

    def pack_xyz12(self, d_edge_out, d_surface_out, d_offsets_out, capacity_points, stream=0):
        """As pack_features, but tight x, y, z triples ([capacity][3] floats): the gather payload."""
        B.check(self._ctx, self._L.lfx_pack_xyz12(
            self._ctx, C.c_void_p(int(d_edge_out)), C.c_void_p(int(d_surface_out)), C.c_void_p(int(d_offsets_out)),
            int(capacity_points), C.c_void_p(int(stream))))

    def pack_colored(self, d_colored_out, d_offsets_out, capacity_points, stream=0):
        """colored_scan of the last device batch as 32-byte pcl::PointXYZRGB wire records (see lfx.h)."""
        B.check(self._ctx, self._L.lfx_pack_colored(
            self._ctx, C.c_void_p(int(d_colored_out)), C.c_void_p(int(d_offsets_out)), int(capacity_points),
            C.c_void_p(int(stream))))

    def download(self, scan, stream=0):
        r = B.ScanResult()
        B.check(self._ctx, self._L.lfx_download_scan(self._ctx, scan, C.c_void_p(int(stream)), C.byref(r)), self._L)
        return _result(r)

    # --- per-stage entry points ------------------------------------------------------------
    def stage_ring(self, x, y, flags, params=None, groups=None, curvature_in=None, range_in=None):
        """Run selected stages of the ring kernel on one angle-sorted ring.
        Returns dict(range, curvature, link, labels, status)."""
        x = np.ascontiguousarray(x, np.float32)
        y = np.ascontiguousarray(y, np.float32)
        n = len(x)
        g = None if groups is None else np.ascontiguousarray(groups, np.int32)
        ci = None if curvature_in is None else np.ascontiguousarray(curvature_in, np.float64)
        ri = None if range_in is None else np.ascontiguousarray(range_in, np.float64)
        out_r, out_c = np.zeros(n), np.zeros(n)
        out_l, out_lab = np.zeros(max(n - 1, 0), np.uint8), np.zeros(n, np.uint8)
        st = C.c_int32(0)
        cp = (params or self.params).to_c()

        def p(a):
            return None if a is None else C.c_void_p(a.ctypes.data)

        B.check(self._ctx, self._L.lfx_stage_ring(
            self._ctx, C.byref(cp), flags, n, p(x), p(y), p(g), p(ci), p(ri), p(out_r), p(out_c),
            p(out_l) if n > 1 else None, p(out_lab), C.cast(C.byref(st), C.c_void_p)))
        return {"range": out_r, "curvature": out_c, "link": out_l.astype(bool), "labels": out_lab,
                "status": st.value}

    def convolution1d(self, values, weight):
        """Convolution1D, convolution.cpp:35-66 (raises LfxError where the reference throws)."""
        v = np.ascontiguousarray(values, np.float64)
        w = np.ascontiguousarray(weight, np.float64)
        out = np.zeros(len(v))
        B.check(self._ctx, self._L.lfx_stage_convolution1d(
            self._ctx, C.c_void_p(v.ctypes.data), len(v), C.c_void_p(w.ctypes.data), len(w),
            C.c_void_p(out.ctypes.data)))
        return out

    def ring_projection(self, cloud):
        """ExtractAngleSortedRings, ring.hpp:141-147 -> {ring id: sorted original indices}."""
        cloud = np.ascontiguousarray(cloud)
        n = len(cloud)
        idx = np.zeros(n, np.uint32)
        nr = C.c_uint32(0)
        rid = np.zeros(B.MAX_RINGS, np.uint16)
        cnt = np.zeros(B.MAX_RINGS, np.uint32)
        B.check(self._ctx, self._L.lfx_stage_ring_projection(
            self._ctx, C.c_void_p(cloud.ctypes.data), n, C.c_void_p(idx.ctypes.data), C.byref(nr),
            C.c_void_p(rid.ctypes.data), C.c_void_p(cnt.ctypes.data)))
        out, off = {}, 0
        for k in range(nr.value):
            out[int(rid[k])] = idx[off:off + cnt[k]].copy()
            off += int(cnt[k])
        return out

    def ColorPointsByLabel(self, cloud, labels):
        """colored_scan (color_points.hpp:60-74): [n,4] f32 = x, y, z, packed rgb (as PCL's PointXYZRGB)."""
        cloud = np.ascontiguousarray(cloud)
        labels = np.ascontiguousarray(labels, np.uint8)
        out = np.zeros((len(cloud), 4), np.float32)
        B.check(self._ctx, self._L.lfx_color_points_by_label(
            self._ctx, C.c_void_p(cloud.ctypes.data), len(cloud), C.c_void_p(labels.ctypes.data),
            C.c_void_p(out.ctypes.data)))
        return out

    # --- measurement ---------------------------------------------------------------------
    def set_profiling(self, on, every=1):
        """HIP events around the kernels of every `every`-th batch (lfx_set_profiling_interval)."""
        B.check(self._ctx, self._L.lfx_set_profiling_interval(self._ctx, int(every)), self._L)
        B.check(self._ctx, self._L.lfx_set_profiling(self._ctx, int(bool(on))), self._L)

    def box_calibration(self, nbytes=0, stream=0):
        """(copy GB/s, shader clock MHz) of this device now: lfx_box_calibration."""
        gbs, mhz = C.c_double(0), C.c_double(0)
        B.check(self._ctx, self._L.lfx_box_calibration(self._ctx, int(nbytes), C.c_void_p(int(stream)), C.byref(gbs), C.byref(mhz)), self._L)
        return gbs.value, mhz.value

    def kernel_times(self):
        ms = (C.c_double * B.LFX_N_KERNELS)()
        cnt = (C.c_uint64 * B.LFX_N_KERNELS)()
        B.check(self._ctx, self._L.lfx_kernel_times(self._ctx, ms, cnt), self._L)
        return {self._L.lfx_kernel_name(k).decode(): (ms[k], int(cnt[k])) for k in range(B.LFX_N_KERNELS)}


def layout_from_fields(fields, point_step, is_bigendian=False):
    """fields: iterable of (name, offset, datatype, count) as in PointCloud2.fields -> binding.Layout for
    FeatureExtraction(layout=...).  Raises LfxError where the node would refuse the cloud (no ring channel)
    or pcl::fromROSMsg could not map x / y / z."""
    arr = (B.PointField * len(fields))(*[B.PointField(n.encode(), o, t, c) for (n, o, t, c) in fields])
    out = B.Layout()
    rc = B.load().lfx_layout_from_fields(arr, len(fields), point_step, int(bool(is_bigendian)), C.byref(out))
    if rc != 0:
        raise B.LfxError(rc, {-7: "the cloud has no ring field", -8: "x / y / z must be FLOAT32 and ring an integer field inside point_step"}.get(rc, "invalid field list"))
    return out



class ScanMap:
    """lfx_map: the index a scan is matched against -- the place of the reference's KDTreeEigen (kdtree.hpp:50-71)."""

    def __init__(self, fx, d_points, n_points, cell_size=1.0, stream=0, host_points=None):
        self._fx = fx
        self._L = fx._L
        h = C.c_void_p()
        if host_points is not None:
            pts = np.ascontiguousarray(host_points, np.float32).reshape(-1, 4)
            B.check(fx._ctx, self._L.lfx_map_create_host(fx._ctx, C.c_void_p(pts.ctypes.data), len(pts), float(cell_size), C.byref(h),
                                                         C.c_void_p(int(stream))))
        else:
            B.check(fx._ctx, self._L.lfx_map_create(fx._ctx, C.c_void_p(int(d_points)), int(n_points), float(cell_size), C.byref(h),
                                                    C.c_void_p(int(stream))))
        self.handle = h

    def info(self):
        n, cell, dims = C.c_uint32(), C.c_float(), (C.c_int32 * 3)()
        self._L.lfx_map_info(self.handle, C.byref(n), C.byref(cell), dims)
        return dict(n_points=n.value, cell_size=cell.value, dims=tuple(dims))

    def nearest(self, d_queries, n_queries, k, d_neighbours=0, d_squared_distances=0, d_indices=0, stream=0):
        """lfx_map_nearest: KDTreeEigen::NearestKSearch (src/kdtree.cpp:44-68) for a batch of queries on the device."""
        B.check(self._fx._ctx, self._L.lfx_map_nearest(
            self._fx._ctx, self.handle, C.c_void_p(int(d_queries)), int(n_queries), int(k), C.c_void_p(int(d_neighbours)),
            C.c_void_p(int(d_squared_distances)), C.c_void_p(int(d_indices)), C.c_void_p(int(stream))))

    def close(self):
        if self.handle:
            self._L.lfx_map_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
