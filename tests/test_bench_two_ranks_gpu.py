"""bench.py's own N > 1 flow, EXECUTED before the driver's one shot at it on an 8-GPU node: two fresh processes, ranks 0 and
1 of a world of 2, both on GPU 0 (a box of this pool has one) -- process group, communicator id from rank 0, the two
agreements, `CloudGather` one step behind on its side stream, the fence with its flush and barrier, the maximum over
the ranks, the one JSON line.  Two things differ from the driver's launch, both forced by the shared GPU: the rendezvous
traffic goes over gloo (`--dist-backend gloo`; the driver's default is nccl = RCCL) and the library's nccl* entry points
come from tests/shim (the test-hooks build of the library: a real RCCL communicator refuses two ranks on one device)."""
import json
import os
import socket
import subprocess
import sys

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


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("dst", ["0", "rotate", "rotate-lanes"])
def test_bench_runs_with_two_ranks(dst):
    """dst 0: every step's clouds to rank 0; rotate (the default): step k's to rank k mod N, two steps per grouped exchange
    on ONE communicator (lfx_gather_payload2); rotate-lanes: the older form, two communicators on two side streams."""
    world = 2
    lanes = dst == "rotate-lanes"
    dst = "rotate" if lanes else dst
    base = dict(os.environ, WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
                LFX_RCCL_LIB=_shim(), LFX_LIB_PATH=os.path.join(ROOT, "lidar_feature_extraction_amd", "_lib", "liblfx_testhooks.so"))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--batch", "8", "--steps", "3", "--warmup", "1",
           "--repeats", "2", "--no-cpu-baseline", "--dist-backend", "gloo", "--gather-dst", dst] + (["--gather-lanes", "2"] if lanes else [])
    procs = [subprocess.Popen(cmd, env=dict(base, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT)
             for r in range(world)]
    outs = [p.communicate(timeout=420) for p in procs]
    _sweep()
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d exit %s:\n%s" % (r, p.returncode, outs[r][1].decode(errors="replace")[-3000:])
    lines = [l for l in outs[0][0].decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line: %r" % (outs[0][0][-500:],)
    assert not [l for l in outs[1][0].decode().splitlines() if l.startswith("{")], "only rank 0 prints the line"
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["steps"] == 3 and d["scaling"] == "weak"
    assert "RCCL gather of clouds to rank" in d["config"]["sharding"] and "unavailable" not in d["config"]["sharding"], d["config"]["sharding"]
    assert ("k mod N" in d["config"]["sharding"]) == (dst == "rotate")
    assert ("2 in flight" if lanes else ("two steps per grouped exchange on one communicator" if dst == "rotate" else "1 in flight")) in d["config"]["sharding"]
    assert d["parity_spot_check"] is True
    assert d["value"] > 0 and d["roofline"]["frac"] > 0 and d["gather_ms_per_step"] > 0
    assert d["cpu_baseline"] is None and "N=1" in d["cpu_baseline_from"]
    # whole-job aggregate: both ranks' scans over the slowest rank's time
    assert abs(d["value"] - world * 8 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-3
    # the line says what RCCL was asked to move, rank by rank: both ranks took part, and the bytes of their sends and
    # receives are those of the gathered clouds (12 B per feature point + the offsets tables)
    def check_stats(stats, fixed_dst):
        assert [st["rank"] for st in stats] == list(range(world))
        for st in stats:
            assert st["all_gathers"] > 0 and st["exchanges"] > 0 and st["bytes_match"], st
            assert st["bytes_sent"] == 12 * st["points_sent"] + 8 * (8 + 1) * (st["sends"] // 3), st
            assert st["bytes_received"] == 12 * st["points_received"] + 8 * (8 + 1) * (st["receives"] // 3), st
        assert sum(st["bytes_sent"] for st in stats) == sum(st["bytes_received"] for st in stats) > 0
        if fixed_dst:
            assert stats[0]["sends"] == 0 and stats[0]["points_received"] > 0 and stats[1]["receives"] == 0 and stats[1]["points_sent"] > 0, stats
        else:
            assert all(st["points_sent"] > 0 and st["points_received"] > 0 for st in stats), stats
    check_stats(d["comm_stats"], dst == "0")
    assert len(d["repeat_values"]) == 2
    if dst == "rotate":
        # ... and the same run also holds the destination the metric names: a second timed region, every step to rank 0
        g0 = d["gather_dst0"]
        assert g0["value"] > 0 and g0["gather_ms_per_step"] > 0 and g0["steps"] >= 4
        check_stats(g0["comm_stats"], True)
    else:
        assert "gather_dst0" not in d


def test_a_rank_that_stops_is_reported_not_waited_for():
    """bench.py's watchdog (N > 1): rank 1 stops making progress in its third step; rank 0 then sits in the fence behind it.
    Both processes end by themselves with exit code 4 and say where they were -- the driver gets a reason, not a hang."""
    world = 2
    base = dict(os.environ, WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
                LFX_RCCL_LIB=_shim(), LFX_LIB_PATH=os.path.join(ROOT, "lidar_feature_extraction_amd", "_lib", "liblfx_testhooks.so"))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--batch", "8", "--steps", "3", "--warmup", "1",
           "--repeats", "1", "--no-cpu-baseline", "--dist-backend", "gloo", "--watchdog-seconds", "12", "--test-stall-rank", "1"]      # (under the 30 s after which the shim itself gives a transfer up)
    procs = [subprocess.Popen(cmd, env=dict(base, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT)
             for r in range(world)]
    outs = [p.communicate(timeout=300) for p in procs]
    _sweep()
    for r, p in enumerate(procs):
        err = outs[r][1].decode(errors="replace")
        assert p.returncode == 4, "rank %d exit %s:\n%s" % (r, p.returncode, err[-2000:])
        assert "bench.py watchdog: rank %d of 2 has made no progress" % r in err, err[-2000:]
        assert not [l for l in outs[r][0].decode().splitlines() if l.startswith("{")], "no line from a run that hung"


def test_a_second_region_that_hangs_costs_only_itself():
    """The fixed-destination region runs AFTER the headline's numbers are complete (and stashed): rank 1 stops in its second
    step there; after a minute without progress every rank ends with exit code 0 and rank 0 has printed the headline's line,
    `gather_dst0` saying what happened."""
    world = 2
    base = dict(os.environ, WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
                LFX_RCCL_LIB=_shim(), LFX_LIB_PATH=os.path.join(ROOT, "lidar_feature_extraction_amd", "_lib", "liblfx_testhooks.so"))
    # steps before the second region: 1 warm-up + 2 x 3 timed = 7 (no warm-up by the clock); the region's own second step is step 8
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--batch", "8", "--steps", "3", "--warmup", "1",
           "--repeats", "2", "--no-cpu-baseline", "--dist-backend", "gloo", "--warm-seconds", "0", "--watchdog-seconds", "20",
           "--test-stall-rank", "1", "--test-stall-step", "8"]
    procs = [subprocess.Popen(cmd, env=dict(base, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT)
             for r in range(world)]
    outs = [p.communicate(timeout=300) for p in procs]
    _sweep()
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d exit %s:\n%s" % (r, p.returncode, outs[r][1].decode(errors="replace")[-3000:])
    lines = [l for l in outs[0][0].decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] > 0 and d["parity_spot_check"] is True and d["comm_stats"][1]["bytes_match"]
    assert "abandoned" in d["gather_dst0"]["error"], d["gather_dst0"]
