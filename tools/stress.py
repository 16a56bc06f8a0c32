"""Randomised parity sweep (not part of pytest: minutes of oracle time).  For N random (sensor shape, input order,
hyper-parameter) draws: HIP path vs the CPU oracle, everything assert_scan_equal checks.  Run on the GPU box:
    python tools/stress.py [N] [seed]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from anywhere: the repo root holds the packages
import sys
import time
import numpy as np
from lidar_feature_extraction_amd import FeatureExtraction, HyperParameters, make_scan
from oracle import binding as OB
from tests.parity import assert_scan_equal

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
t0 = time.time()
n_ok = 0
for case in range(n_cases):
    rings = int(rng.choice([4, 8, 16, 32, 64]))
    cols = int(rng.integers(150, 2600))
    P = int(rng.choice([1, 2, 3, 5, 5, 5, 8, 15]))
    B = int(rng.choice([1, 2, 3, 6, 6, 6, 9, 17, 40]))
    hp = HyperParameters(padding=P, n_blocks=B,
                         neighbor_degree_threshold=float(rng.uniform(0.5, 6.0)),
                         distance_diff_threshold=float(rng.uniform(0.05, 1.0)),
                         parallel_beam_min_range_ratio=float(rng.uniform(0.005, 0.2)),
                         edge_threshold=float(rng.choice([0.01, 0.05, 0.1, 0.5])),
                         surface_threshold=float(rng.choice([0.001, 0.01, 0.1])),
                         min_range=float(rng.uniform(0.05, 1.0)), max_range=float(rng.choice([50.0, 100.0, 1000.0])))
    kw = {}
    order = rng.choice(["sorted", "rotated", "reversed", "revrot", "shuffled", "ragged", "ragrot"])
    if order in ("rotated", "revrot", "ragrot"):
        kw["start_col"] = int(rng.integers(1, cols))
    if order in ("reversed", "revrot"):
        kw["reverse"] = True
    if order == "shuffled":
        kw["shuffle"] = True
    if order in ("ragged", "ragrot"):
        kw["drop_fraction"] = float(rng.uniform(0.01, 0.4))
    seed = int(rng.integers(1, 1 << 30))
    clouds = [make_scan(rings, cols, seed=seed + i, sigma=float(rng.choice([0.01, 0.002, 0.03])), **kw) for i in range(2)]
    exact_cap = bool(rng.integers(0, 2))
    f = FeatureExtraction(hp, device=0, max_points_per_scan=rings * cols, max_batch=2,
                          max_points_per_ring=cols if exact_cap else 0, max_rings=rings if exact_cap else 0)
    op = OB.Params(hp.padding, hp.neighbor_degree_threshold, hp.distance_diff_threshold, hp.parallel_beam_min_range_ratio,
                   hp.edge_threshold, hp.surface_threshold, hp.min_range, hp.max_range, hp.n_blocks)
    for rep in range(2):                       # the second call may take the pre-pass order repair
        got = f.extract_batch(clouds)
        for i, c in enumerate(clouds):
            want = OB.extract(c, op, canonical_ties=False)
            if want["angle_ties"] or want["curvature_ties"]:
                want = OB.extract(c, op, canonical_ties=True)
            assert_scan_equal(got[i], want, "case %d: %dx%d P%d B%d %s rep%d scan%d seed%d" % (case, rings, cols, P, B, order, rep, i, seed))
    f.close()
    n_ok += 1
    if case % 20 == 19:
        print("%d cases ok, %.0f s" % (n_ok, time.time() - t0), flush=True)
print("all %d cases ok in %.0f s" % (n_ok, time.time() - t0))
