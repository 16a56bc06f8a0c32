"""tests/shim/gather_worker.py RANK WORLD OUTDIR -- one rank of tests/test_gather_shim_gpu.py (a process of its own, GPU 0).

The stream: STEPS steps of BATCH scans per rank, scan i of a step -> rank i mod WORLD (gather.shard_scans).  Per step:
extraction, lfx_pack_xyz12, lfx_gather_counts, lfx_gather_payload to that step's destination.  What the destination
receives goes to OUTDIR/step<k>_rank<dst>.npz; every rank writes its lfx_comm_stats to OUTDIR/stats_rank<r>.json.
LFX_RCCL_LIB (set by the test) names the shim that lets two processes share the GPU."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lidar_feature_extraction_amd import FeatureExtraction, make_scan, concat      # noqa: E402
from lidar_feature_extraction_amd.binding import LfxError                          # noqa: E402
from lidar_feature_extraction_amd.gather import RcclGather, shard_scans            # noqa: E402

RINGS, COLS, BATCH = 16, 900, 2
# (destination, what is special about the step)
STEPS = [(0, "plain"), (1, "plain"), (0, "rank 1 has no features"), (1, "ragged"), (0, "capacity")]
# then pairs of steps, each pair ONE grouped exchange on the same communicator (lfx_gather_payload2): step numbers go on
PAIRS = [((0, "plain"), (1, "plain")), ((1, "rank 1 has no features"), (0, "ragged")), ((0, "plain"), (1, "capacity"))]


def stream_scan(step, i, kind, rank_of_scan):
    """Scan i of step `step` (the same on every rank: each rank builds only its own)."""
    if kind == "rank 1 has no features" and rank_of_scan == 1:
        return make_scan(RINGS, 8, seed=9000 + 10 * step + i, spikes=False)          # rings of 8 points: every ring is skipped
    if kind == "ragged":
        return make_scan(RINGS, COLS, seed=9000 + 10 * step + i, drop_fraction=0.03 * (1 + i))
    return make_scan(RINGS, COLS, seed=9000 + 10 * step + i)


def main():
    rank, world, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    fx = FeatureExtraction(device=0, max_points_per_scan=RINGS * COLS, max_batch=BATCH, max_points_per_ring=COLS, max_rings=RINGS)
    uid_path = os.path.join(out, "uid.bin")
    if rank == 0:
        uid = RcclGather.unique_id()
        with open(uid_path + ".part", "wb") as f:
            f.write(uid)
        os.rename(uid_path + ".part", uid_path)
    else:
        t0 = time.time()
        while not os.path.exists(uid_path):
            if time.time() - t0 > 60:
                raise SystemExit("no communicator id from rank 0")
            time.sleep(0.01)
        uid = open(uid_path, "rb").read()
    g = RcclGather(fx, rank, world, uid)
    stream = torch.cuda.current_stream().cuda_stream
    cap = BATCH * RINGS * COLS
    edge = torch.zeros((cap, 3), dtype=torch.float32, device=dev)
    surf = torch.zeros((cap, 3), dtype=torch.float32, device=dev)
    offs = torch.zeros(2 * (BATCH + 1), dtype=torch.int32, device=dev)
    edge_all = torch.zeros((cap * world, 3), dtype=torch.float32, device=dev)
    surf_all = torch.zeros((cap * world, 3), dtype=torch.float32, device=dev)
    offs_all = torch.zeros((world, 2 * (BATCH + 1)), dtype=torch.int32, device=dev)
    for step, (dst, kind) in enumerate(STEPS):
        mine = shard_scans(BATCH * world, rank, world)
        clouds = [stream_scan(step, i, kind, rank) for i in mine]
        d = torch.from_numpy(concat(clouds).view(np.uint8).copy()).to(dev)
        fx.extract_batch_device(d.data_ptr(), [len(c) for c in clouds], stream)
        fx.pack_xyz12(edge.data_ptr(), surf.data_ptr(), offs.data_ptr(), cap, stream)
        g.counts(offs.data_ptr(), BATCH, stream)
        capacity = 8 if kind == "capacity" else cap * world
        try:
            counts = g.payload(dst, edge.data_ptr(), surf.data_ptr(), offs.data_ptr(), BATCH, 3,
                               edge_all.data_ptr(), surf_all.data_ptr(), offs_all.data_ptr(), capacity, stream)
        except LfxError as e:
            with open(os.path.join(out, "step%d_error_rank%d.json" % (step, rank)), "w") as f:
                json.dump({"code": e.code, "text": str(e)}, f)
            continue
        torch.cuda.synchronize()
        if rank == dst:
            np.savez(os.path.join(out, "step%d_rank%d.npz" % (step, dst)), counts=counts, edge=edge_all.cpu().numpy(),
                     surface=surf_all.cpu().numpy(), offsets=offs_all.cpu().numpy())
    with open(os.path.join(out, "stats_rank%d.json" % rank), "w") as f:
        json.dump(g.stats(), f)
    # ---- two steps per grouped exchange
    sets = [(torch.zeros((cap, 3), dtype=torch.float32, device=dev), torch.zeros((cap, 3), dtype=torch.float32, device=dev),
             torch.zeros(2 * (BATCH + 1), dtype=torch.int32, device=dev)) for _ in range(2)]
    recv = [(torch.zeros((cap * world, 3), dtype=torch.float32, device=dev), torch.zeros((cap * world, 3), dtype=torch.float32, device=dev),
             torch.zeros((world, 2 * (BATCH + 1)), dtype=torch.int32, device=dev)) for _ in range(2)]
    step = len(STEPS)
    for pair in PAIRS:
        items, capacity = [], cap * world
        for slot, (dst, kind) in enumerate(pair):
            mine = shard_scans(BATCH * world, rank, world)
            clouds = [stream_scan(step + slot, i, kind, rank) for i in mine]
            d = torch.from_numpy(concat(clouds).view(np.uint8).copy()).to(dev)
            fx.extract_batch_device(d.data_ptr(), [len(c) for c in clouds], stream)
            e, sf, o = sets[slot]
            fx.pack_xyz12(e.data_ptr(), sf.data_ptr(), o.data_ptr(), cap, stream)
            g.counts(o.data_ptr(), BATCH, stream, slot=slot)
            ea, sa, oa = recv[slot]
            items.append({"dst": dst, "slot": slot, "edge": e.data_ptr(), "surface": sf.data_ptr(), "offsets": o.data_ptr(),
                          "edge_all": ea.data_ptr(), "surface_all": sa.data_ptr(), "offsets_all": oa.data_ptr()})
            if kind == "capacity":
                capacity = 8
        try:
            counts = g.payload_group(items, BATCH, 3, capacity, stream)
        except LfxError as e:
            with open(os.path.join(out, "step%d_error_rank%d.json" % (step, rank)), "w") as f:
                json.dump({"code": e.code, "text": str(e)}, f)
            step += 2
            continue
        torch.cuda.synchronize()
        for slot, (dst, kind) in enumerate(pair):
            if rank == dst:
                ea, sa, oa = recv[slot]
                np.savez(os.path.join(out, "step%d_rank%d.npz" % (step + slot, dst)), counts=counts[slot], edge=ea.cpu().numpy(),
                         surface=sa.cpu().numpy(), offsets=oa.cpu().numpy())
        step += 2
    with open(os.path.join(out, "stats_pairs_rank%d.json" % rank), "w") as f:
        json.dump(g.stats(), f)
    g.close()
    fx.close()


if __name__ == "__main__":
    main()
