#!/usr/bin/env python3
"""bench.py -- scans/s of the per-scan extraction hot path on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (ring projection, range, curvature, labelling, masks,
compaction: lfx_extract_batch_device) over one batch of synthetic scans whose point records are
already resident in HBM.  Workload at every N: HDL-64E-shaped 64-ring x 1800-column scans
(BASELINE.json configs[2], the shape the metric is quoted on), `--batch` scans per step per GPU.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU; scan i of the stream goes to rank i mod N (weak scaling: `--batch`
scans per step on every rank) and every step ends with the RCCL gather of the step's labelled clouds
to ONE rank (lidar_feature_extraction_amd/gather.py), inside the timed region.  Which rank rotates
(step k -> rank k mod N, two gathers in flight) by default: a sender reaches a destination over a
single xGMI link (~77 GB/s each way) and produces 117 GB/s of clouds, so a fixed destination
(`--gather-dst 0`) binds every sender at 66 % of the kernel rate; consecutive steps towards
different ranks use different links.

Rank 0 prints ONE JSON line; "roofline" prices the dominant kernel against the HBM roof with the
path's ALGORITHMIC bytes (25*N_pts + 16*(N_edge+N_surface) per scan), its duration
measured live with HIP events on the launch stream; "cpu_baseline" is the CPU oracle (a port of
the reference algorithm, oracle/) timed on this host on a bounded sample of the same scans.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="scans per step per GPU (118 M points, ~15 GB of device memory with all scratch)")
    ap.add_argument("--rings", type=int, default=64)
    ap.add_argument("--cols", type=int, default=1800)
    ap.add_argument("--params", default="code defaults", choices=["code defaults", "launch_yaml"],
                    help="hyper-parameters: the reference's code defaults (hyper_parameter.hpp:35-43, the headline) or its launch file's set (lidar_feature_extraction.param.yaml:3-10)")
    ap.add_argument("--unique", type=int, default=16, help="distinct synthetic scans per GPU (tiled to --batch)")
    ap.add_argument("--start-col", type=int, default=0, help="first column of every scan (a driver that does not cut its scans at -pi: every ring arrives rotated)")
    ap.add_argument("--reverse", action="store_true", help="clockwise sensor: every ring arrives in descending angle order")
    ap.add_argument("--drop-fraction", type=float, default=0.0,
                    help="a driver that omits invalid returns: this share of the records is missing (ragged rings: the bucketing route)")
    ap.add_argument("--drop-zero", action="store_true",
                    help="a driver that keeps the grid and writes invalid returns as (0, 0, 0): --drop-fraction of the records are zeroed "
                         "instead of removed and the context filters them (lfx_config.drop_zero_points, convert.py:162-163)")
    ap.add_argument("--shuffle", action="store_true", help="records in arbitrary order (the reference's documented input contract)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the timed region (--steps steps between two fences) is run this many times; value = the median")
    ap.add_argument("--gather-dst", default="rotate",
                    help="destination rank of the per-step gather: 'rotate' (step k's clouds land whole on rank k mod N: consecutive "
                         "steps use different xGMI links) or a fixed rank (every sender is then bound by its ONE link to that rank: "
                         "~77 GB/s each way = 512 k scans/s per GPU, 66 %% of the kernel rate, whatever N)")
    ap.add_argument("--gather-pairs", type=int, default=-1,
                    help="1: two consecutive steps' clouds travel as ONE grouped exchange on ONE communicator (lfx_gather_payload2: two links of "
                         "every sender busy at once); default: 1 with a rotating destination unless --gather-lanes asks for lanes")
    ap.add_argument("--watchdog-seconds", type=float, default=120.0,
                    help="N > 1 (or --force-gather): a step or a fence that makes no progress for this long ends the process with exit code 4 and says where")
    ap.add_argument("--test-stall-rank", type=int, default=-1, help=argparse.SUPPRESS)      # tests only: this rank stops making progress in its third step (the watchdog's test)
    ap.add_argument("--test-stall-step", type=int, default=2, help=argparse.SUPPRESS)       # ... or in this step (0-based, warm-up included)
    ap.add_argument("--gather-lanes", type=int, default=0,
                    help="gathers in flight (a communicator and a side stream each; step k on lane k mod L): 0 = 2 with a rotating "
                         "destination (the steps' exchanges overlap on their different links), 1 with a fixed one")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the rendezvous traffic (communicator id, agreements, barriers, the maximum "
                         "over ranks); gloo where the ranks share a GPU (rehearsal: a real RCCL communicator refuses that)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-configs", action="store_true", help="skip BASELINE.json's other configurations (the \"configs\" list of the line)")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the per-step RCCL gather")
    ap.add_argument("--force-gather", action="store_true", help="run the pack + gather step even with one rank (rehearsal)")
    ap.add_argument("--warm-seconds", type=float, default=0.25, help="warm-up by the clock after the --warmup steps (0: steps only)")
    ap.add_argument("--no-second-region", action="store_true", help="N>1 with a rotating destination: skip the short second region that gathers to rank 0")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams (each with its own context and scratch) the steps alternate over")
    return ap.parse_args()


def kernels_sha256():
    """Hash of the extraction kernels' sources: ties profiles/pmc_traffic.json to the code it was measured on."""
    h = hashlib.sha256()
    for name in ("lfx_kernels_common.hpp", "lfx_kernels_unit.hpp", "lfx_kernels_extract.hpp"):
        with open(os.path.join(ROOT, "lidar_feature_extraction_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


WORKLOADS = {(64, 1800): "hdl64-64x1800 (BASELINE.json configs[2])", (16, 1800): "vlp16-16x1800 (BASELINE.json configs[1])",
             (16, 900): "plumbing-16x900 (BASELINE.json configs[0])", (128, 2048): "os1-128x2048 (BASELINE.json configs[3])",
             (64, 3600): "64x3600 (a 0.1-degree sensor: units of 612 positions, the 12-chunk form of the unit kernels; not a BASELINE.json configuration)"}


def side_config(dev, rings, cols, batch, steps, warmup, drop_zero_fraction=0.0, repeats=3, curvature=True, params="code defaults"):
    """One of BASELINE.json's other configurations, measured inside the same run as the headline (a few steps, the same
    fences, HIP-event kernel durations, one scan checked against the oracle): so that every number DESIGN.md quotes for them
    has a driver-observed line behind it.  Single GPU, one stream, inputs resident in HBM."""
    from lidar_feature_extraction_amd import FeatureExtraction, HyperParameters, make_scan, concat
    from oracle import binding as oracle
    n_unique = min(8, batch)
    clouds = [make_scan(rings, cols, seed=1234 + j, vfov_deg=22.5 if rings >= 128 else 15.0) for j in range(n_unique)]
    if drop_zero_fraction > 0.0:
        for j, c in enumerate(clouds):
            gone = np.random.Generator(np.random.PCG64(99 + j)).uniform(0.0, 1.0, len(c)) < drop_zero_fraction
            for f in ("x", "y", "z"):
                c[f][gone] = 0.0
    valid = [((c["x"] != 0) | (c["y"] != 0) | (c["z"] != 0)) if drop_zero_fraction > 0.0 else np.ones(len(c), bool) for c in clouds]
    tiled = [clouds[j % n_unique] for j in range(batch)]
    d_points = torch.from_numpy(concat(tiled).view(np.uint8)).to(dev)
    n_list = np.array([len(c) for c in tiled], np.uint32)
    # curvature = False: a context created without LFX_OUT_CURVATURE (the per-point curvature array is not produced; the
    # clouds, the labels and the index sets are): 17 instead of 25 algorithmic bytes per point
    from lidar_feature_extraction_amd import binding as LB
    # params: "code defaults" (hyper_parameter.hpp:35-43) or "launch_yaml" (lidar_feature_extraction.param.yaml:3-10, what the
    # reference's launch file starts the node with: padding 2, 3 degrees, edge threshold 50, max_range 1000)
    hp = HyperParameters.launch_yaml() if params == "launch_yaml" else HyperParameters()
    op = oracle.Params(hp.padding, hp.neighbor_degree_threshold, hp.distance_diff_threshold, hp.parallel_beam_min_range_ratio,
                       hp.edge_threshold, hp.surface_threshold, hp.min_range, hp.max_range, hp.n_blocks)
    fx = FeatureExtraction(hp, device=dev.index, max_points_per_scan=max(len(c) for c in clouds), max_batch=batch,
                           max_points_per_ring=max(cols, 64), max_rings=rings, drop_zero_points=drop_zero_fraction > 0.0,
                           outputs=0 if curvature else (LB.OUT_FEATURES | LB.OUT_LABELS | LB.OUT_SORTED_INDEX),
                           # (what the caller knows about its stream: spares the first batch the plain form's refusal)
                           stream_hint=LB.STREAM_GRID_WITH_HOLES if drop_zero_fraction > 0.0 else 0)
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(warmup):
        fx.extract_batch_device(d_points.data_ptr(), n_list, stream)
    torch.cuda.synchronize()
    # (the clocks of a device that has just sat idle through the CPU baseline are still rising: warm up by the clock too)
    t_warm = time.perf_counter()
    while time.perf_counter() - t_warm < 0.25:
        for _ in range(max(1, steps // 2)):
            fx.extract_batch_device(d_points.data_ptr(), n_list, stream)
        torch.cuda.synchronize()
    dts = []
    for _ in range(repeats):
        t0 = time.perf_counter()
        for _ in range(steps):
            fx.extract_batch_device(d_points.data_ptr(), n_list, stream)
        torch.cuda.synchronize()
        dts.append(time.perf_counter() - t0)
    dt = float(np.median(dts))
    # the kernels' own durations from a pass of their own (the event pairs cost a small step a fifth of its time)
    fx.set_profiling(True, every=1)
    for _ in range(max(2, steps // 4)):
        fx.extract_batch_device(d_points.data_ptr(), n_list, stream)
    torch.cuda.synchronize()
    per_launch_us = {k: 1e3 * ms / max(cnt, 1) for k, (ms, cnt) in fx.kernel_times().items()}
    fx.set_profiling(False)
    dominant = max(per_launch_us, key=per_launch_us.get)
    feats, parity = [], None
    for j in range(n_unique):
        g = fx.download(j, stream)
        feats.append(len(g.edge_index) + len(g.surface_index))
        if j == 0:
            keep = np.nonzero(valid[0])[0]
            w = oracle.extract(np.ascontiguousarray(clouds[0][keep]), op, canonical_ties=False)
            parity = bool(np.array_equal(g.labels[keep], w["labels"]) and (not curvature or g.curvature[keep].tobytes() == w["curvature"].tobytes())
                          and np.array_equal(g.edge_index, keep[w["edge_index"]].astype(np.uint32))
                          and np.array_equal(g.surface_index, keep[w["surface_index"]].astype(np.uint32))
                          and g.edge_points.tobytes() == w["edge_points"].tobytes() and g.surface_points.tobytes() == w["surface_points"].tobytes())
    algo = (25 if curvature else 17) * sum(int(valid[j % n_unique].sum()) for j in range(batch)) + 16 * sum(feats[j % n_unique] for j in range(batch))
    fx.close()
    del d_points
    torch.cuda.empty_cache()
    # the dominant kernel's real HBM bytes, where the committed PMC profile has a section for this workload measured on these sources
    traffic = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        sec = tj.get("sections", {}).get("%dx%dx%d%s" % (rings, cols, batch, "+zeros" if drop_zero_fraction > 0.0 else ""))
        if sec and sec.get("kernels_sha256") == kernels_sha256() and curvature and params == "code defaults":
            traffic = sec.get("hbm_bytes_per_launch", {}).get(dominant)
    except Exception:
        traffic = None
    name = WORKLOADS.get((rings, cols), "%dx%d" % (rings, cols))
    if drop_zero_fraction > 0.0:
        name += ", %.0f %% of the returns written as (0, 0, 0) and filtered (convert.py:162-163)" % (100 * drop_zero_fraction)
    if params != "code defaults":
        name += ", parameters of the reference's launch file (lidar_feature_extraction.param.yaml:3-10)"
    if not curvature:
        name += ", a context without LFX_OUT_CURVATURE (clouds, labels and index sets only: what the node publishes; 17 algorithmic bytes per point)"
    return {"workload": name, "params": params, "scans_per_step": batch, "steps": steps, "value": round(batch * steps / dt, 2), "unit": "scans/s",
            "ms_per_step": round(1e3 * dt / steps, 4), "dominant_kernel": dominant,
            "frac": round(algo / (max(per_launch_us[dominant], 1e-9) * 1e-6) / 1e9 / HBM_PEAK_GBS, 5),
            "whole_path_frac": round(algo / (dt / steps) / 1e9 / HBM_PEAK_GBS, 5),
            "traffic": traffic, "traffic_gbs": (round(traffic / (max(per_launch_us[dominant], 1e-9) * 1e-6) / 1e9, 1) if traffic else None),
            "kernel_us_per_launch": {k: round(v, 2) for k, v in per_launch_us.items() if v > 0}, "parity_spot_check": parity}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d; launch with torch.distributed.run" % (a.gpus, world), file=sys.stderr)
        a.gpus = world
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or a.force_gather:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if a.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    # where the tensors of the rendezvous traffic live: on the device for nccl (= RCCL), on the host for gloo
    ddev = dev if a.dist_backend == "nccl" else torch.device("cpu")

    if not os.path.exists(os.path.join(ROOT, "lidar_feature_extraction_amd", "_lib", "liblfx.so")):
        import __graft_entry__
        __graft_entry__.build()
    from lidar_feature_extraction_amd import FeatureExtraction, HyperParameters, make_scan, concat
    from lidar_feature_extraction_amd.gather import CloudGather, RcclGather

    # ---- synthetic stream: scan i -> rank i mod world; seeds 1234 + scan id
    n_unique = max(1, min(a.unique, a.batch))
    clouds = [make_scan(a.rings, a.cols, seed=1234 + (j * world + rank), vfov_deg=22.5 if a.rings >= 128 else 15.0,
                        start_col=a.start_col, reverse=a.reverse, shuffle=a.shuffle,
                        drop_fraction=0.0 if a.drop_zero else a.drop_fraction) for j in range(n_unique)]
    if a.drop_zero and a.drop_fraction > 0.0:
        for j, c in enumerate(clouds):            # the grid stays, the invalid returns are (0, 0, 0) records
            gone = np.random.Generator(np.random.PCG64(99 + j)).uniform(0.0, 1.0, len(c)) < a.drop_fraction
            for f in ("x", "y", "z"):
                c[f][gone] = 0.0
    n_pts = max(len(c) for c in clouds)
    valid = [((c["x"] != 0) | (c["y"] != 0) | (c["z"] != 0)) if a.drop_zero else np.ones(len(c), bool) for c in clouds]
    tiled = [clouds[j % n_unique] for j in range(a.batch)]
    host = concat(tiled).view(np.uint8)
    d_points = torch.from_numpy(host).to(dev)
    n_list = np.array([len(c) for c in tiled], np.uint32)
    cap = max(a.cols, 64)          # the sensor's column count: no ring is longer
    # one context (= one set of device scratch) per stream.  --streams > 1 alternates consecutive
    # steps over streams, so the HBM-bound ring bucketing of one batch overlaps the latency-bound ring
    # kernel of the previous one (+8 % scans/s at 3 streams); the default is 1 so that the per-kernel
    # durations in "roofline" are those of undisturbed kernels
    n_streams = max(1, a.streams)
    # (with the gather: two contexts taking turns on ONE stream, so that the packing of step k's clouds -- on a stream of
    # its own, behind step k's last kernel -- runs under the kernels of step k + 1, which write another context's clouds)
    will_gather = (world > 1 and not a.no_gather) or a.force_gather
    if will_gather:
        n_streams = 2
    hp_main = HyperParameters.launch_yaml() if a.params == "launch_yaml" else HyperParameters()
    fxs = [FeatureExtraction(hp_main, device=local_rank, max_points_per_scan=n_pts, max_batch=a.batch,
                             max_points_per_ring=cap, max_rings=a.rings, drop_zero_points=a.drop_zero,
                             stream_hint=3 if (a.drop_zero and a.drop_fraction > 0) else 0) for _ in range(n_streams)]      # (3 = LFX_STREAM_GRID_WITH_HOLES)
    fx = fxs[0]
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=dev) for _ in range(n_streams - 1)]
    if will_gather:
        streams = [torch.cuda.current_stream()] * n_streams
    pack_stream = torch.cuda.Stream(device=dev) if will_gather else None
    pack_done = [None] * n_streams
    stream = streams[0].cuda_stream
    step_no = [0]

    use_gather = (world > 1 and not a.no_gather) or a.force_gather
    gather_error = None
    if use_gather:
        # two sets of packed-cloud buffers: while the clouds of step k-1 travel to rank 0 on a side
        # stream, step k extracts and packs into the other set (gather.py: CloudGather)
        feat_cap = int(a.batch * n_pts * 0.35) + 1024
        # x, y, z only (12 bytes per point): what the node publishes (pcl::PointXYZ clouds,
        # feature_extraction.cpp:163-166) and a quarter less to push through rank 0's links
        # rotating destination: two steps as one grouped exchange on one communicator (pairs) by default; --gather-lanes 2
        # is the older form with two communicators on two side streams
        use_pairs = (a.gather_pairs == 1) or (a.gather_pairs < 0 and a.gather_dst == "rotate" and a.gather_lanes <= 0)
        n_lanes = 1 if use_pairs else (a.gather_lanes if a.gather_lanes > 0 else 1)
        # one set more than there are gathers in flight: while those read their sets, the next step packs into a free one
        # (pairs: the pair on its way and the pair being packed)
        bufs = [(torch.zeros((feat_cap, 3), dtype=torch.float32, device=dev),
                 torch.zeros((feat_cap, 3), dtype=torch.float32, device=dev),
                 torch.zeros(2 * (a.batch + 1), dtype=torch.int32, device=dev)) for _ in range(4 if use_pairs else n_lanes + 1)]
        # RCCL through the library's own entry points (lfx_comm_*, lfx_gather_*); torch.distributed only carries the
        # 128-byte communicator id from rank 0 to the others
        # If the communicator cannot be made on some rank (the RCCL library does not open, the id does not arrive), every
        # rank drops the gather together and the line says so ("sharding"): a measurement of the sharded extraction without
        # its exchange is worth more than none.
        gather, gather0, gather_error, id0 = None, None, None, None
        # (one id more than the lanes: the communicator of the second, fixed-destination region -- see "gather_dst0" below)
        second_region = a.gather_dst == "rotate" and not a.no_second_region
        n_ids = n_lanes + (1 if second_region else 0)
        idt = torch.zeros(128 * n_ids, dtype=torch.uint8, device=ddev)
        # every rank first checks that it CAN enter the collective initialisation (the RCCL library opens and has the
        # entry points: making an id proves both) and the ranks agree on that before anyone calls ncclCommInitRank --
        # a rank that failed earlier would otherwise leave the others waiting inside it
        try:
            my_ids = b"".join(RcclGather.unique_id(fx) for _ in range(n_ids))
            if rank == 0:
                idt.copy_(torch.frombuffer(bytearray(my_ids), dtype=torch.uint8))
        except Exception as e:             # noqa: BLE001
            gather_error = "rank %d: %s" % (rank, e)
        if world > 1:
            cannot = torch.tensor([0 if gather_error is None else 1], dtype=torch.int32, device=ddev)
            dist.all_reduce(cannot, op=dist.ReduceOp.MAX)
            if int(cannot.item()) and gather_error is None:
                gather_error = "another rank cannot open its RCCL library"
            dist.broadcast(idt, 0)
        if gather_error is None and not bool(idt.any().item()):
            gather_error = "no communicator id from rank 0"
        if gather_error is None:
            try:
                all_ids = bytes(idt.cpu().numpy().tobytes())
                gather = CloudGather(fx, rank, world, [all_ids[128 * k:128 * (k + 1)] for k in range(n_lanes)],
                                     dst="rotate" if a.gather_dst == "rotate" else int(a.gather_dst), device=dev,
                                     capacity_points=feat_cap * world, batch=a.batch, pairs=use_pairs, profile=True)
                # (the second region's communicator is made when its turn comes, behind the headline's numbers: see below)
                id0 = all_ids[128 * n_lanes:128 * (n_lanes + 1)] if second_region else None
            except Exception as e:         # noqa: BLE001
                gather_error = "rank %d: %s" % (rank, e)
        if world > 1:
            failed = torch.tensor([0 if gather_error is None else 1], dtype=torch.int32, device=ddev)
            dist.all_reduce(failed, op=dist.ReduceOp.MAX)
            if int(failed.item()) and gather_error is None:
                gather_error = "another rank could not create its communicator"
        if gather_error is not None:
            print("bench.py: gather disabled: %s" % gather_error, file=sys.stderr)
            if gather is not None:
                gather.close()
            gather = None
            use_gather = False

    # Watchdog (N > 1): the first real multi-rank run of a flow is the one the driver times, and a hang there would cost
    # the record instead of telling what hung.  A thread looks at where the main thread last reported to be; no progress for
    # --watchdog-seconds ends the PROCESS with a message and exit code 4 (a plain exit: never a re-exec, never a retry).
    progress = {"where": "start", "step": 0, "t": time.monotonic()}
    cur = {"gather": gather if use_gather else None}       # the CloudGather the steps submit to (the second region swaps it)

    def mark(where):
        progress["where"], progress["step"], progress["t"] = where, step_no[0], time.monotonic()

    if (world > 1 or a.force_gather) and a.watchdog_seconds > 0:
        import threading

        def watch():
            while True:
                time.sleep(min(5.0, a.watchdog_seconds / 4))
                idle = time.monotonic() - progress["t"]
                if progress["where"] == "done":
                    return
                if progress["where"] == "host-side measurements":        # (no rank waits for another there: nothing to watch)
                    continue
                if progress.get("phase") == "second region" and idle > min(a.watchdog_seconds, 60.0):
                    # the headline's numbers are complete and stashed: a second region that hangs costs only itself
                    print("bench.py: rank %d: the second (fixed-destination) region made no progress for %.0f s in '%s': abandoned"
                          % (rank, idle, progress["where"]), file=sys.stderr)
                    if progress.get("stash") is not None:
                        line = dict(progress["stash"], gather_dst0={"error": "no progress for %.0f s in '%s': abandoned (the numbers above are complete)" % (idle, progress["where"])})
                        print(json.dumps(line))
                        sys.stdout.flush()
                    sys.stderr.flush()
                    os._exit(0)
                if idle > a.watchdog_seconds and progress.get("printed"):
                    os._exit(0)
                if idle > a.watchdog_seconds:
                    g = cur["gather"]
                    lane = ("pair of steps %d, %d" % ((progress["step"] - 1) & ~1, ((progress["step"] - 1) & ~1) + 1)) if (use_gather and g is not None and g.pairs) \
                        else ("lane %d" % ((progress["step"] - 1) % max(1, len(g.lanes))) if use_gather and g is not None else "no gather")
                    print("bench.py watchdog: rank %d of %d has made no progress for %.0f s in '%s' at step %d (%s, destination %s): giving up"
                          % (rank, world, idle, progress["where"], progress["step"], lane, a.gather_dst), file=sys.stderr)
                    sys.stderr.flush()
                    os._exit(4)
        threading.Thread(target=watch, daemon=True).start()

    def step():
        mark("step")
        if a.test_stall_rank == rank and step_no[0] == a.test_stall_step:
            time.sleep(1e6)
        k = step_no[0] % n_streams
        step_no[0] += 1
        if will_gather and not use_gather:
            k = 0                      # (the gather was given up: one context, as without it)
        if use_gather and pack_done[k] is not None:
            streams[k].wait_event(pack_done[k])        # this context's clouds of two steps ago have been packed
        fxs[k].extract_batch_device(d_points.data_ptr(), n_list, streams[k].cuda_stream)
        if use_gather:
            extracted = torch.cuda.Event()
            extracted.record(streams[k])
            edge_buf, surf_buf, offs = bufs[step_no[0] % len(bufs)]
            with torch.cuda.stream(pack_stream):
                pack_stream.wait_event(extracted)
                cur["gather"].wait_buffer(edge_buf)       # the gather that last read this set must be done
                fxs[k].pack_xyz12(edge_buf.data_ptr(), surf_buf.data_ptr(), offs.data_ptr(), feat_cap, pack_stream.cuda_stream)
                pack_done[k] = torch.cuda.Event()
                pack_done[k].record(pack_stream)
                cur["gather"].submit(edge_buf, surf_buf, offs, a.batch)

    def fence():
        mark("fence: flush of the last gathers")
        if use_gather:
            cur["gather"].flush()      # the last step's clouds
        mark("fence: device synchronise")
        torch.cuda.synchronize()
        if world > 1 or a.force_gather:
            mark("fence: barrier")
            dist.barrier()
        torch.cuda.synchronize()
        mark("fence passed")

    for _ in range(a.warmup):
        step()
    fence()
    # (the clocks of a device that has just been handed over are still rising after a handful of steps: warm up by the
    # clock too, as the side configurations are -- every rank the same number of steps, the gathers need that)
    warm_extra = 0
    if a.warm_seconds > 0:
        t_w = time.perf_counter()
        step()
        fence()
        per = max(time.perf_counter() - t_w, 1e-5)
        warm_extra = int(min(4000, a.warm_seconds / per))
        if world > 1:
            t = torch.tensor([warm_extra], dtype=torch.int64, device=ddev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            warm_extra = int(t.item())
        for _ in range(warm_extra):
            step()
        fence()
    # per-kernel durations: HIP events recorded around the launches inside the timed region, on the
    # stream the kernel is launched on (lfx_set_profiling); with one stream they are what
    # rocprofv3 --kernel-trace --stats reports for the same command
    # The event pairs are recorded around the launches of every 4th step: around every step they cost
    # ~7 % of the throughput being measured (0.815 vs 0.761 ms/step), which the sampled form avoids.
    for f in fxs:
        f.set_profiling(True, every=int(os.environ.get("LFX_BENCH_EVENT_EVERY", max(1, min(4, a.steps // 2)))))
    # The timed region -- exactly --steps steps between two fences (barrier + synchronize) -- is run --repeats times back
    # to back; `value` is the median of the repeats (one 30 ms sample moved by +-4 % from box to box and run to run), the
    # spread is reported beside it.  Every repeat's time is the maximum over the ranks.
    dts = []
    for _ in range(max(1, a.repeats)):
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=ddev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        dts.append(dt)
    dt = float(np.median(dts))
    scans_total = a.batch * a.steps * world
    value = scans_total / dt
    # which side set the rate of an N > 1 step: the time this rank's side stream spent inside the gathers (the exchange and
    # its waits for the peers), per step, maximum over the ranks -- beside the kernels' own durations in "roofline"
    gather_ms_per_step = None
    if use_gather:
        g_ms, g_n = gather.gather_ms()
        gather_ms_per_step = g_ms / max(g_n, 1)
        if world > 1:
            t = torch.tensor([gather_ms_per_step], dtype=torch.float64, device=ddev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            gather_ms_per_step = float(t.item())

    # ---- N > 1: what RCCL was asked to move, rank by rank, beside what the exchanges' own totals say it should have been
    #      (proof in the line itself that N ranks took part and the clouds travelled); then a SECOND, short timed region with
    #      the destination the metric names -- every step's clouds to rank 0 -- on a communicator of its own, so that one
    #      run of the driver's command holds both numbers
    def all_ranks(report):
        keys = ("rank", "sends", "receives", "bytes_sent", "bytes_received", "all_gathers", "points_sent", "points_received",
                "exchanges", "expected_bytes_sent", "expected_bytes_received", "bytes_match")
        mine = torch.tensor([int(report[k]) for k in keys], dtype=torch.int64, device=ddev)
        if world > 1:
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
        else:
            every = [mine]
        out_stats = [dict(zip(keys, [int(v) for v in t.cpu().tolist()])) for t in every]
        for st in out_stats:
            st["bytes_match"] = bool(st["bytes_match"])
        return out_stats

    comm_stats, dst0 = None, None
    if use_gather:
        mark("comm stats")
        comm_stats = all_ranks(gather.comm_report())
    mark("host-side measurements")

    # what THIS box gives right now (outside the timed region, ~50 ms): a plain float4 copy of 1 GiB and the shader clock
    # with every SIMD busy -- so that a slower box and a slower kernel can be told apart in one line of the driver's record
    box = None
    try:
        copy_gbs, clock_mhz = fx.box_calibration(0, stream)
        box = {"copy_gbs": round(copy_gbs, 1), "clock_mhz": round(clock_mhz, 1),
               "note": "lfx_box_calibration right after the timed region: read + written bytes per second of a float4 copy of 1 GiB (best of 3); "
                       "shader clock from a chain of dependent 32-bit adds, four waves per SIMD on every CU"}
    except Exception as e:                 # noqa: BLE001
        box = {"error": "%s: %s" % (type(e).__name__, e)}

    kt = {}
    for f in fxs:
        for k, (ms, cnt) in f.kernel_times().items():
            kt[k] = (kt.get(k, (0.0, 0))[0] + ms, kt.get(k, (0.0, 0))[1] + cnt)
        f.set_profiling(False)
    per_launch_us = {k: (1e3 * ms / max(cnt, 1)) for k, (ms, cnt) in kt.items()}
    dominant = max(per_launch_us, key=per_launch_us.get)
    sum_us = sum(per_launch_us.values())

    # ---- algorithmic bytes of one launch (= one batch): 25 B/point + 16 B/feature point
    feats = []
    parity = None
    for j in range(n_unique):
        g = fx.download(j, stream)
        feats.append(len(g.edge_index) + len(g.surface_index))
        if j == 0 and rank == 0:
            from oracle import binding as oracle          # checker only (never timed as the product)
            # (with the zero filter on, the reference sees the cloud without its (0, 0, 0) records: convert.py:162-163)
            keep = np.nonzero(valid[0])[0]
            op = oracle.Params(hp_main.padding, hp_main.neighbor_degree_threshold, hp_main.distance_diff_threshold, hp_main.parallel_beam_min_range_ratio,
                               hp_main.edge_threshold, hp_main.surface_threshold, hp_main.min_range, hp_main.max_range, hp_main.n_blocks)
            w = oracle.extract(np.ascontiguousarray(clouds[0][keep]), op, canonical_ties=False)
            parity = bool(np.array_equal(g.labels[keep], w["labels"]) and g.curvature[keep].tobytes() == w["curvature"].tobytes()
                          and np.array_equal(g.edge_index, keep[w["edge_index"]].astype(np.uint32))
                          and np.array_equal(g.surface_index, keep[w["surface_index"]].astype(np.uint32)))
    feat_batch = sum(feats[j % n_unique] for j in range(a.batch))
    # (points = the records the path labels: with the zero filter on, the (0, 0, 0) records are not among them)
    pts_batch = sum(int(valid[j % n_unique].sum()) for j in range(a.batch))
    algo_bytes = 25 * pts_batch + 16 * feat_batch
    achieved = algo_bytes / (max(per_launch_us[dominant], 1e-9) * 1e-6) / 1e9
    traffic, traffic_stale = None, False              # not measured in this run: read from the committed PMC profile of this workload
    tj = {}
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            # (one section per workload: "<rings>x<cols>x<batch>", "+zeros" for the grid with (0, 0, 0) records -- tools/merge_pmc.py)
            key = "%dx%dx%d%s" % (a.rings, a.cols, a.batch, "+zeros" if a.drop_zero and a.drop_fraction > 0 else "")
            tj = tj.get("sections", {}).get(key, tj if "sections" not in tj else {})
            if tj.get("batch") == a.batch and tj.get("rings") == a.rings and tj.get("cols") == a.cols:
                # the counters were collected on kernels whose source the file names by hash; after an edit they say nothing
                if tj.get("kernels_sha256") == kernels_sha256():
                    traffic = tj.get("hbm_bytes_per_launch", {}).get(dominant)
                else:
                    traffic_stale = True
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                # the kernel's real HBM rate: PMC bytes (profiles/pmc_traffic.json) over the live duration
                "traffic_gbs": (round(traffic / (max(per_launch_us[dominant], 1e-9) * 1e-6) / 1e9, 1) if traffic else None),
                "traffic_source": ("profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/profile.sh, FETCH doubled), not this run"
                                   if traffic else ("stale: profiles/pmc_traffic.json was measured on other kernel sources" if traffic_stale else None)),
                "algorithmic_bytes_per_launch": int(algo_bytes),
                "kernel_us_per_launch": {k: round(v, 2) for k, v in per_launch_us.items()},
                # the whole path by the wall clock of the timed region (as the side configs; N > 1: this rank's share of it) and
                # by the sum of the kernels' own HIP-event spans
                "whole_path_frac": round(algo_bytes / (dt / a.steps) / 1e9 / HBM_PEAK_GBS, 5),
                "whole_path_frac_kernel_sum": round(algo_bytes / (max(sum_us, 1e-9) * 1e-6) / 1e9 / HBM_PEAK_GBS, 5)}
    if box and box.get("copy_gbs"):
        # against what a copy reaches on this box at this moment: the dominant kernel's REAL traffic where the committed PMC
        # profile is of these sources (else its algorithmic bytes), and the same for the whole path
        roofline["frac_of_box_copy"] = round((traffic if traffic else algo_bytes) / (max(per_launch_us[dominant], 1e-9) * 1e-6) / 1e9 / box["copy_gbs"], 5)
        roofline["frac_of_box_copy_bytes"] = "hbm traffic (PMC)" if traffic else "algorithmic"
        roofline["algorithmic_frac_of_box_copy"] = round(achieved / box["copy_gbs"], 5)
    # the ceiling of this path with 32-byte records: the bytes that have to cross HBM (PMC: 32 B per point read, 9 B written,
    # the feature records written, read and written again) at the rate a copy reaches, as a fraction of the spec peak in
    # ALGORITHMIC bytes -- what `frac` could be at best
    if traffic and box and box.get("copy_gbs"):
        # (the path's own kernels only: the file also holds the calibration copy, the runtime's fills and the download kernels
        # of the parity check, none of which is part of a step)
        tj_all = sum(v for k, v in tj.get("hbm_bytes_per_launch", {}).items()
                     if isinstance(v, (int, float)) and k.startswith(("ring_", "feature_", "batch_", "fallback_", "grid_", "scan_count")))
        roofline["ceiling"] = {"whole_path_frac_at_box_copy_rate": round(algo_bytes / (tj_all / (box["copy_gbs"] * 1e9)) / 1e9 / HBM_PEAK_GBS, 5),
                               "hbm_bytes_per_step": int(tj_all)}

    # ---- CPU baseline: the oracle (port of the reference algorithm), 1 thread, bounded sample
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        from oracle import binding as oracle
        seen = [np.ascontiguousarray(c[v]) for c, v in zip(clouds, valid)]      # what the reference node would be handed
        oracle.extract(seen[0], canonical_ties=False)   # warm
        done, t1 = 0, time.perf_counter()
        while True:
            oracle.extract(seen[done % n_unique], canonical_ties=False)
            done += 1
            el = time.perf_counter() - t1
            if el >= a.cpu_seconds or done >= 2000:
                break
        cpu = {"value": round(done / el, 3), "unit": "scans/s", "cores": 1, "kind": "port",
               "ms_per_scan": round(1e3 * el / done, 3),
               "sample": "%d scans of %dx%d (the bench's own inputs, %d distinct), oracle/lfx_oracle.cpp, 1 thread, %.1f s"
                         % (done, a.rings, a.cols, n_unique, el)}
        # the same port with one scan per host thread (the reference itself is single-threaded,
        # feature_extraction.cpp:185; this is the "all host cores" figure of SURVEY.md 8d)
        from concurrent.futures import ThreadPoolExecutor
        n_thr = max(1, min(len(os.sched_getaffinity(0)), 16))      # a 1-GPU box's CPU share is 16 cores
        t_end = time.perf_counter() + min(a.cpu_seconds, 8.0)

        def work(t):
            k = 0
            while time.perf_counter() < t_end:
                oracle.extract(seen[(t + k) % n_unique], canonical_ties=False)
                k += 1
            return k
        t2 = time.perf_counter()
        with ThreadPoolExecutor(n_thr) as ex:
            total = sum(ex.map(work, range(n_thr)))
        el2 = time.perf_counter() - t2
        cpu["all_cores"] = {"value": round(total / el2, 3), "unit": "scans/s", "cores": n_thr,
                            "sample": "%d scans, one scan per thread, %.1f s" % (total, el2)}

    # ---- end to end through the synchronous host API (pageable host buffers in, host results out: H2D, the
    #      kernels, densify, D2H, un-permute to the caller's point order) -- the latency a ROS callback would see.
    #      Reported beside `value`, never as it (device-resident and PCIe-inclusive separately).
    end_to_end = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        from lidar_feature_extraction_amd import binding as LB

        piped = []

        def timed(fe, scans, reps):
            fe.ExtractFeatures(scans[0])
            t3 = time.perf_counter()
            for j in range(reps):
                fe.ExtractFeatures(scans[j % len(scans)])
            one = (time.perf_counter() - t3) / reps
            # pipelined: submit scan k + 1, then wait for scan k (two in flight: the upload of one beside the kernels of the other)
            reps_p = 4 * reps
            tk = fe.submit(scans[0])
            fe.wait(tk, raw=True)
            t3 = time.perf_counter()
            prev = fe.submit(scans[0])
            for j in range(1, reps_p):
                cur = fe.submit(scans[j % len(scans)])
                fe.wait(prev, raw=True)
                prev = cur
            fe.wait(prev, raw=True)
            piped.append(round(1e3 * (time.perf_counter() - t3) / reps_p, 4))
            some = [scans[j % len(scans)] for j in range(16)]
            fe.extract_batch(some)
            t3 = time.perf_counter()
            for _ in range(3):
                fe.extract_batch(some)
            return round(1e3 * one, 3), round(1e3 * (time.perf_counter() - t3) / 48, 3)

        # what the node consumes (feature_extraction.cpp:161-170: the two clouds), point buffer in pinned memory
        fe = FeatureExtraction(HyperParameters(), device=local_rank, max_points_per_scan=n_pts, max_batch=16,
                               max_points_per_ring=cap, max_rings=a.rings, outputs=LB.OUT_FEATURES)
        pinned = [fe.pinned_like(clouds[j]) for j in range(min(n_unique, 16))]
        one, many = timed(fe, pinned, 32)
        one_pg, many_pg = timed(fe, clouds, 16)
        fe.close()
        # every output (labels, curvature, sorted index: 13 more bytes per point over PCIe, and as many again copied by the
        # Python binding out of the pinned block into arrays of the caller's own), pageable input
        fe = FeatureExtraction(HyperParameters(), device=local_rank, max_points_per_scan=n_pts, max_batch=16,
                               max_points_per_ring=cap, max_rings=a.rings)
        one_all, many_all = timed(fe, clouds, 16)
        fe.close()
        end_to_end = {"ms_per_scan_one_at_a_time": one, "ms_per_scan_batches_of_16": many, "pipelined_ms_per_scan": piped[0],
                      "note": "lfx_extract / lfx_extract_batch through the Python binding: H2D of the 32-byte records, every kernel, the two "
                              "clouds written into pinned host memory, one synchronise; input in pinned memory (lfx_host_alloc), outputs = the "
                              "two clouds (what the node publishes)",
                      "pipelined_note": "lfx_extract_submit / lfx_extract_wait, one scan at a time, two in flight (nothing is copied out of the pinned result block)",
                      "pageable_input": {"ms_per_scan_one_at_a_time": one_pg, "ms_per_scan_batches_of_16": many_pg, "pipelined_ms_per_scan": piped[1]},
                      "pageable_input_all_outputs": {"ms_per_scan_one_at_a_time": one_all, "ms_per_scan_batches_of_16": many_all,
                                                     "pipelined_ms_per_scan": piped[2]}}

    # ---- the consumer of the two clouds: Localizer::Update on the device, one scan and a batch
    consumer = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and a.rings * a.cols <= 64 * 1800:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        try:
            import localize_bench
            one = localize_bench.run(rings=a.rings, cols=a.cols, batch=1, map_scans=16, steps=9, device=local_rank, kd_scans=1, cpu_scans=1)
            many = localize_bench.run(rings=a.rings, cols=a.cols, batch=32, map_scans=16, steps=3, device=local_rank)
        except Exception as e:             # noqa: BLE001  (a side measurement must not cost the line its headline)
            one = many = None
            consumer = {"error": "%s: %s" % (type(e).__name__, e)}
        if one is not None:
            consumer = {"localize_ms_one_scan": one["localize_ms_per_scan"], "localize_ms_per_scan_batches_of_32": many["localize_ms_per_scan"],
                        "iterations_mean": many["iterations_mean"], "edge_map_points": one["edge_map_points"],
                        "surface_map_points": one["surface_map_points"],
                        # the consumer's own baseline and roof: a KD-tree on this host (search only: a lower bound of the reference's
                        # Update) and the bytes its neighbour search has to read over the time the whole Update takes
                        "cpu_baseline": dict(one["kdtree_host"], unit="ms/scan", kind="port",
                                             # the oracle's own Update of the same scan (oracle/lfx_oracle_loc.cpp: Optimizer::Run with an
                                             # exhaustive neighbour search where the reference walks a KD-tree), one core
                                             oracle_update_ms_per_scan=one.get("cpu_oracle_ms_per_scan"),
                                             oracle_update_note=one.get("cpu_oracle_note"),
                                             max_pose_difference_to_oracle=one.get("max_pose_difference_to_oracle")),
                        "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                                     "achieved": round(one["search_bytes_per_scan"] / (1e-3 * one["localize_ms_per_scan"]) / 1e9, 2),
                                     "frac": round(one["search_bytes_per_scan"] / (1e-3 * one["localize_ms_per_scan"]) / 1e9 / HBM_PEAK_GBS, 5),
                                     "bytes_per_scan": one["search_bytes_per_scan"],
                                     "note": "16 B x the map points of the 27 grid cells around every query x iterations, over the time of one scan's "
                                             "Update: latency-bound (four dependent kernels and ~105 us per iteration, each a chain of round trips to memory), nowhere near a roof"},
                        "note": "lfx_localize_batch after extraction (Downsample + Optimizer::Run of the reference localizer, localizer.hpp:71-80; "
                                "clouds never leave the device); maps = the features of 16 scans along a track, grid cells of 1 m; "
                                "parity with Eigen / nanoflann / PCL arithmetic unpinned (DESIGN.md 7)"}

    # ---- BASELINE.json's other single-GPU configurations and the stream the reference's own pipeline produces (invalid
    #      returns as (0, 0, 0) records, filtered), a few steps each, in the same run
    configs = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and not a.no_side_configs and (a.rings, a.cols, a.batch) == (64, 1800, 1024):
        for f in fxs:
            f.close()
        fxs = []
        del d_points
        torch.cuda.empty_cache()
        configs = []
        for args in ((16, 900, 1024, 20, 3), (16, 1800, 1024, 20, 3), (128, 2048, 32, 40, 5), (64, 1800, 1024, 8, 2, 0.05),
                     (64, 3600, 256, 8, 2),
                     # (the two that are read against the headline: its steps, warm-up and repeats)
                     (64, 1800, 1024, 20, 3, 0.0, 5, False), (64, 1800, 1024, 20, 3, 0.0, 5, True, "launch_yaml")):
            try:
                configs.append(side_config(dev, *args))
            except Exception as e:         # noqa: BLE001  (a side measurement must not cost the line its headline)
                configs.append({"workload": "%dx%d" % args[:2], "error": "%s: %s" % (type(e).__name__, e)})

    if rank == 0:
        workload = WORKLOADS.get((a.rings, a.cols), "%dx%d" % (a.rings, a.cols)) + (", batch = 32" if (a.rings, a.batch) == (128, 32) else "")
        out = {
            "metric": "scans/sec (%d-ring x %d synthetic scans, extraction hot path, inputs resident in HBM)" % (a.rings, a.cols),
            "value": round(value, 2), "unit": "scans/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "repeats": len(dts), "value_min": round(scans_total / max(dts), 2), "value_max": round(scans_total / min(dts), 2),
            "repeat_values": [round(scans_total / d, 1) for d in dts], "warm_steps_by_clock": warm_extra,
            "ms_per_step": round(1e3 * dt / a.steps, 4), "ms_per_scan": round(1e3 * dt / (a.batch * a.steps), 6),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload, "rings": a.rings, "cols": a.cols,
                       "points_per_scan": n_pts, "scans_per_step_per_gpu": a.batch, "params": a.params,
                       "input_order": ("shuffled" if a.shuffle else ("rings in angle order" if not (a.start_col or a.reverse) else
                                       "rings %s%s" % ("reversed " if a.reverse else "", "rotated by %d columns" % a.start_col if a.start_col else ""))) +
                                      ((", %.0f %% of the returns %s" % (100 * a.drop_fraction, "written as (0, 0, 0) and filtered" if a.drop_zero else "missing"))
                                       if a.drop_fraction > 0 else ""),
                       "streams": 1 if use_gather else n_streams,
                       "sharding": "scan i -> gpu i mod N" + ((", RCCL gather of clouds to rank %s per step, %s" % (
                           "k mod N of step k" if a.gather_dst == "rotate" else a.gather_dst,
                           "two steps per grouped exchange on one communicator" if use_pairs else "%d in flight" % n_lanes)) if use_gather else "") +
                                   (" (gather unavailable: %s)" % gather_error if gather_error else "")},
            "roofline": roofline, "box": box, "cpu_baseline": cpu, "end_to_end": end_to_end, "consumer": consumer, "parity_spot_check": parity,
            "configs": configs,
        }
        if world > 1:
            out["cpu_baseline_from"] = "the N=1 line of the same bench (the host baseline is timed on rank 0 at N=1 only)"
            out["roofline"]["note"] = "rank 0's kernels (every rank runs the same launches on its own scans)"
        if gather_ms_per_step is not None:
            out["gather_ms_per_step"] = round(gather_ms_per_step, 4)
        if comm_stats is not None:
            out["comm_stats"] = comm_stats
            out["comm_stats_note"] = ("lfx_comm_stats of every rank after the headline's region (warm-up included): ncclSend / ncclRecv calls, the bytes "
                                      "they carried, counts all-gathers; expected_* = 12 B x the feature points + one offsets table per cloud pair, from "
                                      "the totals the exchanges themselves returned")
    # ---- the second region: every step's clouds gathered to rank 0 (the destination BASELINE.json's metric names), on a
    #      communicator of its own, made only now -- the headline's numbers are complete and stashed, so that whatever this
    #      region does (an exception on one rank, a hang: the watchdog then prints the stashed line and ends the process with
    #      exit code 0) costs only itself
    if use_gather and id0 is not None and fxs:
        progress["stash"] = out if rank == 0 else None
        progress["phase"] = "second region"
        mark("second region: communicator")
        try:
            gather0 = CloudGather(fx, rank, world, [id0], dst=0, device=dev, capacity_points=feat_cap * world, batch=a.batch, pairs=False, profile=True)
            cur["gather"] = gather0
            steps2 = max(4, a.steps // 2)
            for _ in range(2):
                step()
            fence()
            gather0.gather_ms()                  # (drop the warm-up's spans)
            t0 = time.perf_counter()
            for _ in range(steps2):
                step()
            fence()
            dt2 = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([dt2], dtype=torch.float64, device=ddev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt2 = float(t.item())
            g_ms, g_n = gather0.gather_ms()
            g2 = g_ms / max(g_n, 1)
            if world > 1:
                t = torch.tensor([g2], dtype=torch.float64, device=ddev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                g2 = float(t.item())
            dst0 = {"value": round(a.batch * steps2 * world / dt2, 2), "unit": "scans/s", "steps": steps2, "ms_per_step": round(1e3 * dt2 / steps2, 4),
                    "gather_ms_per_step": round(g2, 4), "comm_stats": all_ranks(gather0.comm_report()),
                    "note": "the same steps with every step's clouds gathered to rank 0 (the destination BASELINE.json's metric names), one "
                            "exchange in flight on a communicator of its own, timed between the same fences after the headline's region"}
        except Exception as e:             # noqa: BLE001
            dst0 = {"error": "%s: %s" % (type(e).__name__, e)}
        cur["gather"] = gather
        progress["phase"] = None
        progress["stash"] = None
    if rank == 0:
        if dst0 is not None:
            out["gather_dst0"] = dst0
        print(json.dumps(out))
        sys.stdout.flush()
    progress["printed"] = True             # (from here on a rank that waits in vain ends quietly: the line is out)
    mark("closing")
    if use_gather:
        gather.close()
        if gather0 is not None:
            gather0.close()
    for f in fxs:
        f.close()
    if world > 1 or a.force_gather:
        dist.barrier()
        dist.destroy_process_group()
    mark("done")


if __name__ == "__main__":
    main()
