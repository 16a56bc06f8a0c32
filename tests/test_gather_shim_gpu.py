"""The N > 1 exchange of lfx_gather_payload, EXECUTED: two processes share GPU 0 through tests/shim/rccl_shim.cpp
(LFX_RCCL_LIB), so that the branch a one-GPU box cannot reach with a real RCCL communicator -- grouped ncclSend on the
sending rank, per-rank ncclRecv with its offset arithmetic on the destination (lfx_gather.hip) -- runs with world = 2:
destination 0 and destination 1, ragged totals, a rank without a single feature, and the capacity error on every rank.
The destination's buffers are reassembled into stream order and compared with the CPU oracle scan by scan."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM_DIR = os.path.join(ROOT, "tests", "shim")


def _sweep():
    """What the shim's communicators leave under /dev/shm (leave markers, files of a rank that was ended)."""
    import glob
    import shutil
    for d in glob.glob("/dev/shm/lfxshim_*"):
        shutil.rmtree(d, ignore_errors=True)


def _shim():
    so = os.path.join(SHIM_DIR, "_build", "librccl_shim.so")
    src = os.path.join(SHIM_DIR, "rccl_shim.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", SHIM_DIR])
    return so


def test_two_ranks_exchange_through_the_library(tmp_path):
    sys.path.insert(0, SHIM_DIR)
    import gather_worker as W
    from lidar_feature_extraction_amd.gather import reassemble, split_gathered
    from oracle import binding as OB
    world = 2
    # (the door for another nccl* implementation exists in the test-hooks build of the library only)
    env = dict(os.environ, LFX_RCCL_LIB=_shim(), LFX_LIB_PATH=os.path.join(ROOT, "lidar_feature_extraction_amd", "_lib", "liblfx_testhooks.so"))
    procs = [subprocess.Popen([sys.executable, os.path.join(SHIM_DIR, "gather_worker.py"), str(r), str(world), str(tmp_path)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = [p.communicate(timeout=240)[0].decode(errors="replace") for p in procs]
    _sweep()
    assert all(p.returncode == 0 for p in procs), "\n".join("rank %d (exit %s):\n%s" % (r, procs[r].returncode, outs[r][-2500:]) for r in range(world))
    stats = [json.load(open(tmp_path / ("stats_rank%d.json" % r))) for r in range(world)]
    n_plain = 0
    for step, (dst, kind) in enumerate(W.STEPS):
        if kind == "capacity":
            for r in range(world):          # every rank refuses, nobody is left in a send or a receive
                err = json.load(open(tmp_path / ("step%d_error_rank%d.json" % (step, r))))
                assert err["code"] == -4, err
            continue
        n_plain += 1
        z = np.load(tmp_path / ("step%d_rank%d.npz" % (step, dst)))
        per_rank = split_gathered(z["edge"], z["surface"], z["offsets"], z["counts"], W.BATCH)
        clouds = reassemble(per_rank, W.BATCH * world, world, W.BATCH)
        for i, (ge, gs) in enumerate(clouds):
            w = OB.extract(W.stream_scan(step, i, kind, i % world), canonical_ties=False)
            assert np.array_equal(ge, w["edge_points"][:, :3]), "step %d (%s) scan %d edge cloud" % (step, kind, i)
            assert np.array_equal(gs, w["surface_points"][:, :3]), "step %d (%s) scan %d surface cloud" % (step, kind, i)
            if kind == "rank 1 has no features" and i % world == 1:
                assert len(ge) == 0 and len(gs) == 0
            else:
                assert len(ge) > 0 and len(gs) > 0
        if kind == "rank 1 has no features":
            assert int(z["counts"][1][0]) == 0 and int(z["counts"][1][1]) == 0 and int(z["counts"][0][0]) > 0
        if kind == "ragged":
            assert int(z["counts"][0][0]) != int(z["counts"][1][0])
    # the send / receive branch ran on both ranks: each was the destination of two steps and the sender of two
    for r in range(world):
        assert stats[r]["sends"] == 6 and stats[r]["receives"] == 6, stats
        assert stats[r]["bytes_sent"] > 0 and stats[r]["bytes_received"] > 0
        assert stats[r]["all_gathers"] == len(W.STEPS)
    # ---- two steps as ONE grouped exchange on the same communicator (lfx_gather_payload2): both destinations of a pair
    #      receive what the single form delivers; a pair of which ONE step does not fit is refused whole on every rank
    step = len(W.STEPS)
    for pair in W.PAIRS:
        if any(kind == "capacity" for _, kind in pair):
            for r in range(world):
                err = json.load(open(tmp_path / ("step%d_error_rank%d.json" % (step, r))))
                assert err["code"] == -4, err
            assert not any(os.path.exists(tmp_path / ("step%d_rank%d.npz" % (step + k, d))) for k, (d, _) in enumerate(pair))
            step += 2
            continue
        for k, (dst, kind) in enumerate(pair):
            z = np.load(tmp_path / ("step%d_rank%d.npz" % (step + k, dst)))
            per_rank = split_gathered(z["edge"], z["surface"], z["offsets"], z["counts"], W.BATCH)
            clouds = reassemble(per_rank, W.BATCH * world, world, W.BATCH)
            for i, (ge, gs) in enumerate(clouds):
                w = OB.extract(W.stream_scan(step + k, i, kind, i % world), canonical_ties=False)
                assert np.array_equal(ge, w["edge_points"][:, :3]), "pair step %d (%s) scan %d edge cloud" % (step + k, kind, i)
                assert np.array_equal(gs, w["surface_points"][:, :3]), "pair step %d (%s) scan %d surface cloud" % (step + k, kind, i)
            if kind == "rank 1 has no features":
                assert int(z["counts"][1][0]) == 0 and int(z["counts"][1][1]) == 0 and int(z["counts"][0][0]) > 0
        step += 2
    stats2 = [json.load(open(tmp_path / ("stats_pairs_rank%d.json" % r))) for r in range(world)]
    for r in range(world):
        # two complete pairs: in each a rank sends one step (3 messages) and receives the other (3 messages)
        assert stats2[r]["sends"] == 6 + 6 and stats2[r]["receives"] == 6 + 6, stats2
        assert stats2[r]["all_gathers"] == len(W.STEPS) + 2 * len(W.PAIRS)
