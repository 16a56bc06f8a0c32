"""The multi-rank path on CPU: world_size 2 over gloo.  Covers the gather of variable-length
clouds (what RCCL carries over xGMI on the GPU node) and the scan sharding."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lidar_feature_extraction_amd.gather import CloudGather, gather_clouds, shard_scans


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


def _check(out, world, batch, seed_shift=0):
    ok = out is not None and len(out) == world
    for r in range(world):
        e, s, o, n_e, n_s = _make_rank_data(r + seed_shift, batch)
        ok = ok and out[r]["edge"].shape[0] == n_e and out[r]["surface"].shape[0] == n_s
        ok = ok and np.array_equal(out[r]["edge"].numpy(), e[:n_e]) and np.array_equal(out[r]["surface"].numpy(), s[:n_s])
        ok = ok and np.array_equal(out[r]["offsets"].numpy(), o)
    return bool(ok)


def _pipelined_worker(rank, world, port, batch, ret):
    """CloudGather: the result of step k arrives with submit(k+1) / flush(); three steps."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = CloudGather(dst=0)
    outs = []
    for step in range(3):
        edge, surf, off, _ne, _ns = _make_rank_data(rank + 10 * step, batch)
        outs.append(g.submit(torch.from_numpy(edge), torch.from_numpy(surf), torch.from_numpy(off), batch))
    outs.append(g.flush())
    ok = outs[0] is None
    if rank == 0:
        for step in range(3):
            ok = ok and _check(outs[step + 1], world, batch, seed_shift=10 * step)
    else:
        ok = ok and all(o is None for o in outs)
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_pipelined_gather_two_ranks_gloo():
    world, batch = 2, 5
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_pipelined_worker, args=(world, _free_port(), batch, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


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


def test_shard_scans_round_robin():
    assert shard_scans(10, 1, 4) == [1, 5, 9]
    got = sorted(i for r in range(8) for i in shard_scans(37, r, 8))
    assert got == list(range(37))
