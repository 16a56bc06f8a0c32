"""Shared checkers: HIP path (ScanFeatures) against the CPU oracle (dict from oracle.binding.extract)."""
import numpy as np


def assert_scan_equal(got, want, ctx=""):
    """Bit-exact for every integer output; curvature compared by bit pattern (the north_star allows
    1e-5 relative, but label parity needs equal ordering, so equal bits is what is enforced)."""
    assert want["angle_ties"] == 0 or ctx.endswith("[ties]"), ctx + ": input has angle ties; compare in canonical mode only"
    assert got.ring_id.tolist() == want["ring_id"].tolist(), ctx + ": ring ids"
    assert got.ring_count.tolist() == want["ring_count"].tolist(), ctx + ": ring counts"
    assert np.array_equal(got.sorted_index, want["sorted_index"].astype(np.uint32)), ctx + ": ring projection (angle-sorted indices)"
    gs, ws = got.ring_status != 0, want["ring_status"] != 0
    assert np.array_equal(gs, ws), ctx + ": skipped rings differ: %s vs %s" % (got.ring_status.tolist(), want["ring_status"].tolist())
    bad = np.nonzero(got.labels != want["labels"])[0]
    assert bad.size == 0, ctx + ": %d labels differ, first at point %d: got %d want %d" % (
        bad.size, bad[0], got.labels[bad[0]], want["labels"][bad[0]])
    gc, wc = got.curvature.view(np.uint64), want["curvature"].view(np.uint64)
    badc = np.nonzero(gc != wc)[0]
    assert badc.size == 0, ctx + ": %d curvature values differ in bits, first at %d: %r vs %r" % (
        badc.size, badc[0], got.curvature[badc[0]], want["curvature"][badc[0]])
    assert np.array_equal(got.edge_index, want["edge_index"].astype(np.uint32)), ctx + ": edge index set"
    assert np.array_equal(got.surface_index, want["surface_index"].astype(np.uint32)), ctx + ": surface index set"
    assert got.edge_points.tobytes() == want["edge_points"].tobytes(), ctx + ": edge cloud"
    assert got.surface_points.tobytes() == want["surface_points"].tobytes(), ctx + ": surface cloud"


def status_codes_equal_where_single_cause(got, want):
    """Exact status code where only one skip cause can apply (the oracle reports the first site
    that throws in the reference's lazy evaluation order; the HIP path reports by precedence)."""
    for g, w in zip(got.ring_status.tolist(), want["ring_status"].tolist()):
        if w in (1, 2, 3):
            assert g == w


def assert_filtered_equal(got, want, keep, zero, ctx=""):
    """The zero filter (lfx_config.drop_zero_points; convert.py:162-163,192): `got` is the HIP path on a cloud whose (0, 0, 0)
    records (mask `zero`) it dropped itself, `want` the oracle on the cloud WITHOUT them (`keep` = their indices in the full
    cloud).  Same bar as assert_scan_equal, indices mapped through `keep`."""
    assert want["angle_ties"] == 0 or ctx.endswith("[ties]"), ctx + ": input has angle ties; compare in canonical mode only"
    assert got.ring_id.tolist() == want["ring_id"].tolist(), ctx + ": ring ids"
    assert got.ring_count.tolist() == want["ring_count"].tolist(), ctx + ": ring counts"
    assert np.array_equal(got.sorted_index, keep[want["sorted_index"]].astype(np.uint32)), ctx + ": ring projection (angle-sorted indices)"
    assert np.array_equal(got.ring_status != 0, want["ring_status"] != 0), ctx + ": skipped rings differ"
    bad = np.nonzero(got.labels[keep] != want["labels"])[0]
    assert bad.size == 0, ctx + ": %d labels differ, first at kept point %d: got %d want %d" % (
        bad.size, bad[0], got.labels[keep][bad[0]], want["labels"][bad[0]])
    assert not got.labels[zero].any(), ctx + ": a dropped record carries a label"
    assert got.curvature[keep].tobytes() == want["curvature"].tobytes(), ctx + ": curvature bits"
    assert np.array_equal(got.edge_index, keep[want["edge_index"]].astype(np.uint32)), ctx + ": edge index set"
    assert np.array_equal(got.surface_index, keep[want["surface_index"]].astype(np.uint32)), ctx + ": surface index set"
    assert got.edge_points.tobytes() == want["edge_points"].tobytes(), ctx + ": edge cloud"
    assert got.surface_points.tobytes() == want["surface_points"].tobytes(), ctx + ": surface cloud"
