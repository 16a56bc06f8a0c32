"""Randomised parity sweep (the long form; a 200-draw seeded slice runs under pytest -m gpu as
test_stress_slice).  For N random (sensor shape, input order, hyper-parameter) draws: HIP path vs the CPU oracle,
everything assert_scan_equal checks.  Run on the GPU box:
    python tools/stress.py [N] [seed]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from anywhere: the repo root holds the packages
from tests.stress_cases import run_cases  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
t0 = time.time()
seen = run_cases(n_cases, seed, report_every=20)
print("all %d cases ok in %.0f s (seed %d; input orders drawn: %s)" % (n_cases, time.time() - t0, seed, seen))
