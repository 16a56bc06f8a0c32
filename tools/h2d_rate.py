"""tools/h2d_rate.py -- what the box's PCIe link gives an upload of one 64 x 1800 scan (3.69 MB, pinned): the floor of any
host-in path per scan."""
import time

import torch

n = 64 * 1800 * 32
h = torch.empty(n, dtype=torch.uint8).pin_memory()
d = torch.empty(n, dtype=torch.uint8, device="cuda:0")
s = torch.cuda.Stream()
for depth in (1, 4):
    with torch.cuda.stream(s):
        for _ in range(20):
            d.copy_(h, non_blocking=True)
        s.synchronize()
        reps = 400
        t0 = time.perf_counter()
        for i in range(reps):
            d.copy_(h, non_blocking=True)
            if depth == 1:
                s.synchronize()
        s.synchronize()
        dt = (time.perf_counter() - t0) / reps
    print("pinned H2D of %.2f MB, %s: %.1f us (%.1f GB/s)" % (n / 1e6, "one at a time" if depth == 1 else "queued back to back", 1e6 * dt, n / dt / 1e9))
