"""lfx_extract_submit / lfx_extract_wait (the pipelined host API: the next scan's upload beside this scan's kernels):
every pipelined result equals the synchronous one, pinned and pageable input, scans of different shapes and routes back
to back, and the misuse cases are refused."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _same(a, b):
    assert np.array_equal(a.labels, b.labels) and a.curvature.tobytes() == b.curvature.tobytes()
    assert np.array_equal(a.sorted_index, b.sorted_index)
    assert np.array_equal(a.edge_index, b.edge_index) and np.array_equal(a.surface_index, b.surface_index)
    assert a.edge_points.tobytes() == b.edge_points.tobytes() and a.surface_points.tobytes() == b.surface_points.tobytes()
    assert np.array_equal(a.ring_id, b.ring_id) and np.array_equal(a.ring_count, b.ring_count) and np.array_equal(a.ring_status, b.ring_status)


@pytest.mark.parametrize("pinned", [False, True])
def test_every_pipelined_result_equals_the_synchronous_one(pinned):
    from lidar_feature_extraction_amd import FeatureExtraction, make_scan
    from lidar_feature_extraction_amd.binding import LfxError
    from oracle import binding as OB
    rings, cols = 32, 1024
    # organised, rotated (falls back at first), ragged (bucketing route), a short scan: the routes change under the pipeline
    kinds = [{}, {}, {"start_col": 200}, {"drop_fraction": 0.04}, {}, {"shuffle": True}, {}, {}]
    clouds = [make_scan(rings, cols, seed=4100 + i, **kw) for i, kw in enumerate(kinds)]
    clouds.append(make_scan(rings, 300, seed=4199))
    ref = FeatureExtraction(device=0, max_points_per_scan=rings * cols, max_batch=1, max_points_per_ring=cols, max_rings=rings)
    want = [ref.ExtractFeatures(c) for c in clouds]
    ref.close()
    w0 = OB.extract(clouds[0], canonical_ties=False)
    assert np.array_equal(want[0].labels, w0["labels"]) and np.array_equal(want[0].edge_index, w0["edge_index"].astype(np.uint32))
    fx = FeatureExtraction(device=0, max_points_per_scan=rings * cols, max_batch=1, max_points_per_ring=cols, max_rings=rings)
    src = [fx.pinned_like(c) for c in clouds] if pinned else [c.copy() for c in clouds]
    got = [None] * len(clouds)
    tickets = []
    for i, c in enumerate(src):
        tickets.append(fx.submit(c))
        if not pinned:
            c["x"][:] = np.nan          # pageable input was copied at once: the caller may reuse it
        if i >= 1:
            got[i - 1] = fx.wait(tickets[i - 1])
    got[-1] = fx.wait(tickets[-1])
    for g, w in zip(got, want):
        _same(g, w)
    # misuse: a third submit with two in flight; waiting out of order; an unknown ticket; the synchronous call meanwhile
    t1, t2 = fx.submit(clouds[0]), fx.submit(clouds[1])
    with pytest.raises(LfxError):
        fx.submit(clouds[2])
    with pytest.raises(LfxError):
        fx.wait(t2)
    with pytest.raises(LfxError):
        fx.wait(t2 + 7)
    with pytest.raises(LfxError):
        fx.ExtractFeatures(clouds[0])
    _same(fx.wait(t1), want[0])
    _same(fx.wait(t2), want[1])
    _same(fx.ExtractFeatures(clouds[3]), want[3])          # and the synchronous path works again afterwards
    fx.close()
