"""Randomised sweep of the device optimizer (lfx_align_point_pairs: Optimizer<AlignmentProblem>::Run of the reference) against
the CPU restatement: random point-pair problems (sizes 0..5000, noise, outliers, start poses near and far, iteration
limits 1..20) in ragged batches; the stopping reason, iteration, pose, error and scale of every problem.  Where two successive
errors (or scales) are equal to rounding the stopping test may fall either way: such a problem may differ by one iteration
with the pose equal to 1e-6; they are counted and reported.  On the GPU box:
    python tools/stress_align.py [N batches] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_align_gpu import _oracle_pairs, _pose, _run_pairs  # noqa: E402


def main():
    from lidar_feature_extraction_amd import FeatureExtraction
    n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    rng = np.random.default_rng(seed)
    fx = FeatureExtraction(device=0, max_points_per_scan=1024, max_batch=1)
    t0 = time.time()
    total, close_calls, codes = 0, 0, {}
    for b in range(n_batches):
        problems = []
        for _ in range(int(rng.integers(1, 40))):
            n = int(rng.choice([0, 1, 2, 3, 5, 17, 100, 1000, 5000], p=[0.03, 0.03, 0.03, 0.05, 0.06, 0.2, 0.3, 0.25, 0.05]))
            X = rng.uniform(-1, 1, (n, 3)) * float(10.0 ** rng.uniform(-0.5, 2))
            true = _pose(rng.normal(0, 0.4, 3), rng.normal(0, 3, 3))
            Y = X @ true[:, :3].T + true[:, 3] + rng.normal(0, float(10.0 ** rng.uniform(-4, -0.5)), (n, 3))
            if n > 10 and rng.random() < 0.6:
                out = rng.choice(n, max(1, n // int(rng.integers(3, 30))), replace=False)
                Y[out] += rng.normal(0, 5.0, (len(out), 3))
            far = rng.random() < 0.3
            start = _pose(rng.normal(0, 0.8 if far else 0.1, 3), rng.normal(0, 5 if far else 0.5, 3))
            problems.append((X, Y, start))
        max_iter = int(rng.choice([1, 2, 5, 10, 20]))
        got = _run_pairs(fx, problems, max_iter)
        for i, (r, pr) in enumerate(zip(got, problems)):
            w = _oracle_pairs(pr[0], pr[1], pr[2], max_iter)
            total += 1
            codes[r["code"]] = codes.get(r["code"], 0) + 1
            same = (r["code"], r["iteration"]) == (w["code"], w["iteration"])
            if np.isnan(w["pose"]).any() or np.isnan(r["pose"]).any():
                assert np.isnan(w["pose"]).any() == np.isnan(r["pose"]).any() and same, (b, i, r, w)
                continue
            tol = 1e-7 if same else 1e-5
            err = np.abs(r["pose"] - w["pose"]).max() / (1 + np.abs(w["pose"]).max())
            if not same:
                close_calls += 1
                assert abs(r["iteration"] - w["iteration"]) <= 1 and r["success"] == w["success"], (b, i, max_iter, len(pr[0]), r, w)
            assert err <= tol, (b, i, max_iter, len(pr[0]), err, r, w)
        if b % 10 == 9:
            print("%d batches, %d problems ok (%d close calls), %.0f s" % (b + 1, total, close_calls, time.time() - t0), flush=True)
    print("all %d problems in %d batches ok in %.0f s (seed %d; stopping reasons %s; %d differed by one iteration at a tie)" % (
        total, n_batches, time.time() - t0, seed, dict(sorted(codes.items())), close_calls))
    fx.close()


if __name__ == "__main__":
    main()
