"""CPU-side checks of the boundary: the C-ABI library loads, exports every symbol include/lfx.h
declares, and refuses to run without a device (no CPU fallback).  No compute calls."""
import ctypes as C
import os
import re

import pytest

from lidar_feature_extraction_amd import binding as LB

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LB.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return LB.load()


def declared_functions():
    text = open(os.path.join(ROOT, "include", "lfx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lfx_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    names = declared_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), "liblfx.so does not export " + n
    assert sorted(LB.EXPORTS) == names


def test_parameter_sets(lib):
    p = LB.Params()
    lib.lfx_default_params(C.byref(p))      # hyper_parameter.hpp:35-43
    assert (p.padding, p.neighbor_degree_threshold, p.distance_diff_threshold, p.parallel_beam_min_range_ratio,
            p.edge_threshold, p.surface_threshold, p.min_range, p.max_range, p.n_blocks) == \
        (5, 2.0, 0.3, 0.02, 0.05, 0.05, 0.1, 100.0, 6)
    lib.lfx_launch_params(C.byref(p))       # lidar_feature_extraction.param.yaml:3-10
    assert (p.padding, p.neighbor_degree_threshold, p.edge_threshold, p.surface_threshold, p.max_range, p.n_blocks) == \
        (2, 3.0, 50.0, 0.05, 1000.0, 6)


def test_invalid_arguments_are_rejected_before_any_device_work(lib):
    ctx = C.c_void_p()
    p = LB.Params()
    lib.lfx_default_params(C.byref(p))
    cfg = LB.Config(C.sizeof(LB.Config), 1000, 1, 0, 0, 0, LB.Layout(0, 0, 0, 0, 0))
    p.padding = 0                            # hyper_parameter.hpp:45 asserts padding > 0
    assert lib.lfx_create(C.byref(ctx), 0, C.byref(p), C.byref(cfg)) == -1
    p.padding = 99
    assert lib.lfx_create(C.byref(ctx), 0, C.byref(p), C.byref(cfg)) == -1
    lib.lfx_default_params(C.byref(p))
    cfg0 = LB.Config(C.sizeof(LB.Config), 0, 1, 0, 0, 0, LB.Layout(0, 0, 0, 0, 0))
    assert lib.lfx_create(C.byref(ctx), 0, C.byref(p), C.byref(cfg0)) == -1
    # struct_size: an uninitialised struct (0) and one too short to hold the two capacities are refused by name; a caller
    # built against an OLDER header (a shorter struct: here without outputs and stream_hint) is taken -- what it does not
    # know reads as zero -- and gets as far as the device check
    for size in (0, 8):
        bad = LB.Config(size, 1000, 1, 0, 0, 0, LB.Layout(0, 0, 0, 0, 0))
        assert lib.lfx_create(C.byref(ctx), 0, C.byref(p), C.byref(bad)) == -1
        assert b"struct_size" in lib.lfx_last_error(None)
    older = LB.Config(C.sizeof(LB.Config) - 8, 1000, 1, 0, 0, 0, LB.Layout(0, 0, 0, 0, 0), 0xFFFFFFFF, 0xFFFFFFFF)      # garbage behind its end
    rc = lib.lfx_create(C.byref(ctx), 0, C.byref(p), C.byref(older))
    assert rc in (0, -2) and b"stream_hint" not in lib.lfx_last_error(None), (rc, lib.lfx_last_error(None))
    if rc == 0:
        lib.lfx_destroy(ctx)
    assert lib.lfx_status_string(5).decode().startswith("two adjacent points")
    assert lib.lfx_kernel_name(1) == b"ring_unit_kernel"


def test_label_to_color_table(lib, refvec):
    """LabelToColor vectors: test_color_points.cpp:40-78 + color_points.cpp:39-68 (host-side helper, no device)."""
    names = ["Default", "Edge", "EdgeNeighbor", "Surface", "SurfaceNeighbor", "OutOfRange", "Occluded", "ParallelBeam"]
    for k, name in enumerate(names):
        rgb = (C.c_uint8 * 3)()
        assert lib.lfx_label_to_color(k, rgb) == 0
        assert list(rgb) == refvec["label_to_color"]["expect_rgb"][name]
    assert lib.lfx_label_to_color(8, (C.c_uint8 * 3)()) == -1      # ThrowIfInvalidLabelDetected


def _layout(lib, fields, point_step, big=False):
    arr = (LB.PointField * len(fields))(*[LB.PointField(n.encode(), o, t, c) for (n, o, t, c) in fields])
    out = LB.Layout()
    rc = lib.lfx_layout_from_fields(arr, len(fields), point_step, int(big), C.byref(out))
    return rc, out


def test_ring_channel_is_required(lib, refvec):
    """RingIsAvailable (test_ring.cpp:129-144): a cloud without a "ring" field is refused."""
    for case in refvec["ring_is_available"]["cases"]:
        fields = [tuple(f) for f in case["fields"]] + [("x", 0, 7, 1), ("y", 4, 7, 1), ("z", 8, 7, 1)]
        rc, _ = _layout(lib, fields, 32)
        assert (rc != -7) == case["available"]
        assert rc == (0 if case["available"] else -7)


def test_layout_from_point_cloud2_fields(lib, refvec):
    """The field lists the upstream converter's tests use (test_convert.py) map to record layouts."""
    for case in refvec["point_cloud2_layouts"]["cases"]:
        rc, lay = _layout(lib, [tuple(f) for f in case["fields"]], case["point_step"])
        assert rc == 0, case["name"]
        got = [lay.point_step, lay.off_x, lay.off_y, lay.off_z, lay.off_ring, lay.ring_datatype]
        assert got == case["layout"], case["name"]
        assert lay.big_endian == 0
    rc, lay = _layout(lib, [("x", 0, 7, 1), ("y", 4, 7, 1), ("z", 8, 7, 1), ("ring", 12, 4, 1)], 16, big=True)
    assert rc == 0 and lay.big_endian == 1
    # what pcl::fromROSMsg<PointXYZIR> could not map, or what lies outside the record
    assert _layout(lib, [("x", 0, 8, 1), ("y", 8, 7, 1), ("z", 12, 7, 1), ("ring", 16, 4, 1)], 24)[0] == -8   # x is FLOAT64
    assert _layout(lib, [("x", 0, 7, 1), ("y", 4, 7, 1), ("ring", 12, 4, 1)], 16)[0] == -8                    # no z
    assert _layout(lib, [("x", 0, 7, 1), ("y", 4, 7, 1), ("z", 8, 7, 1), ("ring", 12, 7, 1)], 16)[0] == -8    # ring FLOAT32
    assert _layout(lib, [("x", 0, 7, 1), ("y", 4, 7, 1), ("z", 8, 7, 1), ("ring", 15, 4, 1)], 16)[0] == -8    # ring past the record
    assert lib.lfx_layout_from_fields(None, 0, 16, 0, None) == -1


def test_no_cpu_fallback_without_a_device(lib):
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present")
    ctx = C.c_void_p()
    p = LB.Params()
    lib.lfx_default_params(C.byref(p))
    cfg = LB.Config(C.sizeof(LB.Config), 1000, 1, 0, 0, 0, LB.Layout(0, 0, 0, 0, 0))
    assert lib.lfx_create(C.byref(ctx), 0, C.byref(p), C.byref(cfg)) == -2          # LFX_ERR_NO_DEVICE
    assert b"no CPU path" in lib.lfx_last_error(None)
    from lidar_feature_extraction_amd import FeatureExtraction
    with pytest.raises(LB.LfxError):
        FeatureExtraction(max_points_per_scan=1000)


def test_product_does_not_touch_the_oracle():
    """The oracle is test infrastructure: nothing under the package or include/ may reference it."""
    for base in ("lidar_feature_extraction_amd", "include"):
        for dirpath, _d, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", "Makefile")):
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    assert "oracle" not in text.lower().replace("no cpu", ""), os.path.join(dirpath, f)


def test_reference_message_texts(lib):
    """RangeMessage* (range_message.hpp:37-83, vectors of test_range_message.cpp:35-61) and the exception texts of
    abandoned rings (convolution.cpp:40-41 pinned by test_convolution.cpp:61-70, index_range.cpp:36-38,
    neighbor.hpp:72-73, math.cpp:41) as the library reports them: host-side string functions, no device."""
    import json
    vec = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_unit_vectors.json")))
    kinds = {"LargerThanOrEqualTo": 0, "SmallerThanOrEqualTo": 1, "LargerThan": 2, "SmallerThan": 3}
    buf = C.create_string_buffer(256)
    for c in vec["range_message"]["cases"]:
        n = lib.lfx_range_message(kinds[c["kind"]], c["value_name"].encode(), c["range_name"].encode(), c["value"], c["range"], buf, 256)
        assert buf.value.decode() == c["expect"] and n == len(c["expect"])
    assert lib.lfx_range_message(7, b"i", b"max", 1, 2, buf, 256) == -1
    t = vec["throw_texts"]
    p = LB.Params()
    lib.lfx_default_params(C.byref(p))

    def msg(status, n):
        lib.lfx_ring_message(status, n, C.byref(p), buf, 256)
        return buf.value.decode()

    assert msg(2, 9) == t["too_few_for_convolution"].format(n=9, m=11)
    assert msg(3, 14) == t["too_few_for_blocks"].format(d=4, b=6)
    assert msg(4, 17) == t["block_too_small"].format(n=1)
    assert msg(5, 400) == t["zero_norm_pair"]
    assert msg(0, 400) == "" and msg(1, 3) == "" and msg(7, 5000) == ""
    p.padding = 1
    conv = [c for c in vec["convolution1d"]["cases"] if c.get("throws")][0]
    assert msg(2, len(conv["input"])) == conv["message"]
    # the library and the oracle agree on every cause over a sweep of ring lengths
    from oracle import binding as OB
    ob = C.create_string_buffer(256)
    for P, Bn in ((5, 6), (2, 6), (3, 17)):
        p.padding, p.n_blocks = P, Bn
        op = OB.Params(P, 2.0, 0.3, 0.02, 0.05, 0.05, 0.1, 100.0, Bn)
        for n in range(0, 80):
            for status in (2, 3, 4, 5):
                if status == 4 and n - 2 * P < Bn:
                    continue
                lib.lfx_ring_message(status, n, C.byref(p), buf, 256)
                OB.lib().orc_ring_message(status, n, C.byref(op), ob, 256)
                assert buf.value == ob.value


def test_alignment_texts_and_null_arguments(lib):
    """The result texts of optimization_result.hpp:43-79 and the success rule (:46-79: empty input and maximum iteration
    fail), without a device; entry points refuse null arguments before any device work."""
    lib.lfx_align_message.restype = C.c_char_p
    texts = {0: "Optimization successfully converged", 1: "The error is larger than previous iteration",
             2: "The scale is larger than previous iteration", 3: "The iteration reached the maximum value",
             4: "The input data is empty", 5: "No surface neighbourhood spans a plane"}
    for code, text in texts.items():
        assert lib.lfx_align_message(code).decode() == text
    assert lib.lfx_align_message(99) == b"unknown"
    null = C.c_void_p(0)
    assert lib.lfx_map_create(null, null, 10, C.c_float(1.0), null, null) == -1
    assert lib.lfx_map_create_host(null, null, 10, C.c_float(1.0), null, null) == -1
    assert lib.lfx_map_info(null, None, None, None) == -1
    assert lib.lfx_map_nearest(null, null, null, 1, 1, null, null, null, null) == -1
    assert lib.lfx_localize_batch(null, null, null, 15, 20, C.c_float(1.0), 1, None, None, null) == -1
    assert lib.lfx_localize_host(null, null, null, 15, 20, C.c_float(1.0), null, 0, null, 0, None, None, null) == -1
    lib.lfx_map_destroy(null)                 # a no-op


def test_the_shipped_library_reads_no_debug_switch(lib):
    """The LFX_DEBUG_* switches (route pins, span variants, ablation flags) and the RCCL override exist in the test-hooks
    build only (liblfx_testhooks.so, -DLFX_TEST_HOOKS): the shipped library does not even hold their names."""
    data = open(LB.LIB_PATH, "rb").read()
    assert b"LFX_DEBUG_" not in data and b"LFX_RCCL_LIB" not in data
    hooks = open(LB.HOOKS_LIB_PATH, "rb").read()
    assert b"LFX_DEBUG_FUSED" in hooks and b"LFX_RCCL_LIB" in hooks
