"""The multi-rank path on CPU: world_size 2 over gloo.  `gather_clouds` is the rehearsal of lfx_gather's protocol
(totals first, then the variable-length clouds and the offsets tables to one rank); the flow around it -- scans sharded
scan i -> rank i mod N, each rank's clouds packed back to back, gather, reassembly into stream order -- is the one
bench.py runs on the GPU node, here with the CPU oracle standing in for the device on every rank."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lidar_feature_extraction_amd.gather import gather_clouds, gather_clouds_pair, reassemble, shard_scans


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_rank_data(rank, batch):
    rng = np.random.default_rng(100 + rank)
    ne = rng.integers(0, 40, batch)
    ns = rng.integers(0, 90, batch)
    if rank == 1:
        ne[0] = 0
    off = np.zeros(2 * (batch + 1), np.int32)
    off[1:batch + 1] = np.cumsum(ne)
    off[batch + 2:] = np.cumsum(ns)
    cap = 4096
    edge = np.zeros((cap, 4), np.float32)
    surf = np.zeros((cap, 4), np.float32)
    edge[:ne.sum()] = rng.standard_normal((ne.sum(), 4))
    surf[:ns.sum()] = rng.standard_normal((ns.sum(), 4))
    return edge, surf, off, int(ne.sum()), int(ns.sum())


def _worker(rank, world, port, batch, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    edge, surf, off, ne, ns = _make_rank_data(rank, batch)
    out = gather_clouds(torch.from_numpy(edge), torch.from_numpy(surf), torch.from_numpy(off), batch, dst=0)
    ok = True
    if rank == 0:
        ok = out is not None and len(out) == world
        for r in range(world):
            e, s, o, n_e, n_s = _make_rank_data(r, batch)
            ok = ok and out[r]["edge"].shape[0] == n_e and out[r]["surface"].shape[0] == n_s
            ok = ok and np.array_equal(out[r]["edge"].numpy(), e[:n_e]) and np.array_equal(out[r]["surface"].numpy(), s[:n_s])
            ok = ok and np.array_equal(out[r]["offsets"].numpy(), o)
    else:
        ok = out is None
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_gather_two_ranks_gloo():
    world, batch = 2, 7
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), batch, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def _stream_worker(rank, world, port, n_scans, ret):
    """Scan i of a 7-scan stream goes to rank i mod 2; every rank extracts its scans (oracle), packs their clouds as
    tight x, y, z triples with the offsets table lfx_pack_xyz12 writes, the clouds are gathered to rank 0 and put back
    into stream order; rank 0 compares every scan's clouds with the oracle's for that scan."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lidar_feature_extraction_amd import make_scan
    from oracle import binding as OB
    mine = shard_scans(n_scans, rank, world)
    batch = (n_scans + world - 1) // world                     # every rank packs `batch` scans (the last may be empty)
    edges, surfs, off = [], [], np.zeros(2 * (batch + 1), np.int32)
    for k in range(batch):
        if k < len(mine):
            w = OB.extract(make_scan(8, 300, seed=2000 + mine[k]), canonical_ties=False)
            edges.append(w["edge_points"][:, :3])
            surfs.append(w["surface_points"][:, :3])
        else:
            edges.append(np.zeros((0, 3), np.float32))
            surfs.append(np.zeros((0, 3), np.float32))
        off[k + 1] = off[k] + len(edges[-1])
        off[batch + 1 + k + 1] = off[batch + 1 + k] + len(surfs[-1])
    cap = 8 * 300 * batch
    e = np.zeros((cap, 3), np.float32)
    s = np.zeros((cap, 3), np.float32)
    e[:off[batch]] = np.concatenate(edges)
    s[:off[2 * batch + 1]] = np.concatenate(surfs)
    out = gather_clouds(torch.from_numpy(e), torch.from_numpy(s), torch.from_numpy(off), batch, dst=0)
    ok = True
    if rank == 0:
        per_rank = [{k: v.numpy() for k, v in o.items()} for o in out]
        for i, (ge, gs) in enumerate(reassemble(per_rank, n_scans, world, batch)):
            w = OB.extract(make_scan(8, 300, seed=2000 + i), canonical_ties=False)
            ok = ok and np.array_equal(ge, w["edge_points"][:, :3]) and np.array_equal(gs, w["surface_points"][:, :3])
            ok = ok and len(ge) > 0 and len(gs) > 0
    else:
        ok = out is None
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_stream_sharded_over_two_ranks_reassembles_to_the_oracle_clouds():
    world, n_scans = 2, 7
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_stream_worker, args=(world, _free_port(), n_scans, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def _pair_worker(rank, world, port, batch, ret):
    """Two steps as ONE exchange (the protocol of lfx_gather_payload2): step A's clouds to rank 0, step B's to rank 1, every
    send and receive of both posted together; in step B rank 1 has no edge point in its first scan and rank 0 none at all."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def data(step, r):
        e, s, o, n_e, n_s = _make_rank_data(10 * step + r, batch)
        if step == 1 and r == 0:                                   # a rank without a single feature in this step
            o = np.zeros_like(o)
            n_e = n_s = 0
        return e, s, o, n_e, n_s

    steps = []
    for step, dst in ((0, 0), (1, 1)):
        e, s, o, _, _ = data(step, rank)
        steps.append((torch.from_numpy(e), torch.from_numpy(s), torch.from_numpy(o), dst))
    out = gather_clouds_pair(steps, batch)
    ok = len(out) == 2
    for step, dst in ((0, 0), (1, 1)):
        if rank != dst:
            ok = ok and out[step] is None
            continue
        ok = ok and out[step] is not None and len(out[step]) == world
        for r in range(world):
            e, s, o, n_e, n_s = data(step, r)
            got = out[step][r]
            ok = ok and got["edge"].shape[0] == n_e and got["surface"].shape[0] == n_s
            ok = ok and np.array_equal(got["edge"].numpy(), e[:n_e]) and np.array_equal(got["surface"].numpy(), s[:n_s])
            ok = ok and np.array_equal(got["offsets"].numpy(), o)
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_two_steps_as_one_exchange_two_ranks_gloo():
    world, batch = 2, 5
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_pair_worker, args=(world, _free_port(), batch, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_shard_scans_round_robin():
    assert shard_scans(10, 1, 4) == [1, 5, 9]
    got = sorted(i for r in range(8) for i in shard_scans(37, r, 8))
    assert got == list(range(37))
