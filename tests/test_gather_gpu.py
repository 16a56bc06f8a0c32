"""lfx_gather over RCCL through the C ABI, rehearsed with one rank on the GPU box (two ranks cannot share one GPU in an
RCCL communicator): extraction -> lfx_pack_xyz12 -> lfx_gather_counts / lfx_gather_payload one step behind on a side
stream (CloudGather) -> reassembly into stream order -> every scan's clouds against the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("lanes", [1, 2, "pairs"])
def test_extract_pack_gather_reassemble_one_rank(lanes):
    """lanes = 2: two communicators, two side streams, step k's exchange on lane k mod 2, three sets of send buffers.
    "pairs" (what bench.py runs with a rotating destination): ONE communicator, steps 2m and 2m + 1 as one grouped exchange
    (lfx_gather_payload2) through a real RCCL, four sets of send buffers, an odd number of steps (the last one travels
    alone).  Every step's clouds must arrive whole and in order."""
    pairs = lanes == "pairs"
    lanes = 1 if pairs else lanes
    import torch
    from lidar_feature_extraction_amd import FeatureExtraction, make_scan, concat
    from lidar_feature_extraction_amd.gather import CloudGather, RcclGather, reassemble
    from oracle import binding as OB
    rings, cols, batch, steps = 16, 900, 3, 5
    dev = torch.device("cuda", 0)
    fx = FeatureExtraction(device=0, max_points_per_scan=rings * cols, max_batch=batch, max_points_per_ring=cols, max_rings=rings)
    uid = [RcclGather.unique_id() for _ in range(lanes)]
    cap = batch * rings * cols
    g = CloudGather(fx, 0, 1, uid if lanes > 1 else uid[0], dst="rotate" if (lanes > 1 or pairs) else 0, device=dev, capacity_points=cap, batch=batch,
                    pairs=pairs)
    stream = torch.cuda.current_stream().cuda_stream
    bufs = [(torch.zeros((cap, 3), dtype=torch.float32, device=dev), torch.zeros((cap, 3), dtype=torch.float32, device=dev),
             torch.zeros(2 * (batch + 1), dtype=torch.int32, device=dev)) for _ in range(4 if pairs else lanes + 1)]
    scans, outs, keep = [], [], []
    for step in range(steps):
        clouds = [make_scan(rings, cols, seed=5000 + step * batch + k, start_col=(17 * k if step == 1 else 0)) for k in range(batch)]
        scans.append(clouds)
        d = torch.from_numpy(concat(clouds).view(np.uint8).copy()).to(dev)
        keep.append(d)
        fx.extract_batch_device(d.data_ptr(), [len(c) for c in clouds], stream)
        e, s, o = bufs[step % len(bufs)]
        g.wait_buffer(e)
        fx.pack_xyz12(e.data_ptr(), s.data_ptr(), o.data_ptr(), cap, stream)
        out = g.submit(e, s, o, batch)
        if out is not None:
            g.done.synchronize()
            for o1 in (out if pairs else [out]):             # (pairs: a list of two steps' results)
                outs.append([{k: v.cpu().numpy().copy() for k, v in r.items()} for r in o1])
    out = g.flush()
    for o1 in (out if pairs else [out]):
        outs.append([{k: v.cpu().numpy().copy() for k, v in r.items()} for r in o1])
    fx.batch_status(stream)
    assert len(outs) == steps
    for step in range(steps):
        for k, (ge, gs) in enumerate(reassemble(outs[step], batch, 1, batch)):
            w = OB.extract(scans[step][k], canonical_ties=False)
            assert np.array_equal(ge, w["edge_points"][:, :3]), "step %d scan %d edge cloud" % (step, k)
            assert np.array_equal(gs, w["surface_points"][:, :3]), "step %d scan %d surface cloud" % (step, k)
            assert len(ge) > 0 and len(gs) > 0
    # capacity too small on the destination: every rank reports it, nothing hangs
    small = CloudGather(fx, 0, 1, RcclGather.unique_id(), dst=0, device=dev, capacity_points=8, batch=batch)
    e, s, o = bufs[0]
    small.submit(e, s, o, batch)
    from lidar_feature_extraction_amd.binding import LfxError
    with pytest.raises(LfxError) as err:
        small.flush()
    assert err.value.code == -4
    small.close()
    g.close()
    fx.close()
