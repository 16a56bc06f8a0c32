"""Synthetic spinning-lidar scans in the reference's input layout.

One scan = R rings x C columns of 32-byte PointXYZIR records
(/root/reference/lib/include/lidar_feature_library/point_type.hpp:62-86; wire offsets
x0 y4 z8 pad12 intensity16 ring20, /root/reference/point_type_converter/point_type_converter/convert.py:134-145),
emitted column-major (all rings of one firing, then the next azimuth), which is the order a
driver publishes and the order the reference node receives.

Scene: the sensor stands in an axis-aligned 20 m x 12 m room with thin pillars at 3-5.5 m
(edges + occlusions), a ground plane for the downward beams, an "open door" sector whose
returns lie beyond max_range, a sector of returns closer than min_range, and isolated
single-column range spikes (parallel-beam hits).  Every range carries additive Gaussian noise
(sigma = 1 cm): that removes exact curvature ties, whose order the reference leaves to an
unstable std::sort.  No point is (0,0,0) (the upstream converter drops those, convert.py:162-163).
"""
import numpy as np

POINT_DTYPE = np.dtype({"names": ["x", "y", "z", "pad", "intensity", "ring"],
                        "formats": ["<f4", "<f4", "<f4", "<f4", "<f4", "<u2"],
                        "offsets": [0, 4, 8, 12, 16, 20], "itemsize": 32})

SENSORS = {
    # name: (rings, columns, vertical field of view in degrees)
    "plumbing-16x900": (16, 900, 15.0),
    "vlp16-16x1800": (16, 1800, 15.0),
    "hdl64-64x1800": (64, 1800, 15.0),
    "os1-128x2048": (128, 2048, 22.5),
}


def _ray_room(cx, cy, dx, dy, x0, x1, y0, y1):
    with np.errstate(divide="ignore", invalid="ignore"):
        tx = np.where(dx > 0, (x1 - cx) / dx, np.where(dx < 0, (x0 - cx) / dx, np.inf))
        ty = np.where(dy > 0, (y1 - cy) / dy, np.where(dy < 0, (y0 - cy) / dy, np.inf))
    return np.minimum(tx, ty)


def _ray_circle(cx, cy, dx, dy, px, py, rad):
    ox, oy = cx - px, cy - py
    b = ox * dx + oy * dy
    c = ox * ox + oy * oy - rad * rad
    disc = b * b - c
    t = -b - np.sqrt(np.where(disc > 0, disc, np.nan))
    return np.where((disc > 0) & (t > 0), t, np.inf)


def make_scan(rings=64, cols=1800, seed=1234, vfov_deg=15.0, sigma=0.01, n_pillars=14,
              drop_fraction=0.0, shuffle=False, start_col=0, reverse=False,
              out_of_range=True, spikes=True):
    """Return one scan as a POINT_DTYPE array (rings*cols points, fewer with drop_fraction).

    drop_fraction  drop this share of points at random (ragged rings, as after the zero filter)
    shuffle        permute the points (forces the general ring projection, not the presorted one)
    start_col      rotate the firing sequence (scan starts at another azimuth)
    reverse        clockwise sensors: azimuth decreases with time
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    az = -np.pi + 2.0 * np.pi * (np.arange(cols) + 0.5) / cols
    elev = np.deg2rad(np.linspace(-vfov_deg, vfov_deg, rings))
    dx, dy = np.cos(az), np.sin(az)
    cx, cy, h = 1.3, -0.7, 1.8                      # sensor position in the room, height over ground
    r = _ray_room(cx, cy, dx, dy, -10.0, 10.0, -6.0, 6.0)
    pr = np.random.Generator(np.random.PCG64(4242))  # the scene is the same for every seed
    for k in range(n_pillars):
        ang = 2.0 * np.pi * (k + 0.37) / n_pillars + pr.uniform(-0.1, 0.1)
        dist = pr.uniform(3.0, 5.5)
        r = np.minimum(r, _ray_circle(cx, cy, dx, dy, cx + dist * np.cos(ang), cy + dist * np.sin(ang),
                                      pr.uniform(0.08, 0.2)))
    r2 = np.broadcast_to(r, (rings, cols)).copy()
    # downward beams hit the ground before the wall
    with np.errstate(divide="ignore"):
        ground = np.where(elev < -1e-6, h / np.tan(-elev), np.inf)
    r2 = np.minimum(r2, ground[:, None])
    if out_of_range:
        far = (az > 2.2) & (az < 2.45)               # open door: returns beyond max_range (100 m)
        r2[:, far] = 150.0 + 5.0 * np.sin(40.0 * az[far])[None, :]
        near = (az > -0.6) & (az < -0.52)            # something on the sensor housing: < min_range
        r2[:, near] = 0.05
    r2 = r2 + sigma * rng.standard_normal((rings, cols))
    if spikes:
        n_spk = max(1, rings * cols // 700)
        rr = rng.integers(0, rings, n_spk)
        cc = rng.integers(8, cols - 8, n_spk)
        r2[rr, cc] *= rng.uniform(0.55, 0.8, n_spk)
    r2 = np.maximum(r2, 0.02)

    order = np.arange(cols)
    if reverse:
        order = order[::-1]
    order = np.roll(order, -start_col)
    pts = np.zeros(rings * cols, POINT_DTYPE)
    grid = pts.reshape(cols, rings)                  # column-major emission: [column][ring]
    rs = r2[:, order].T                              # [col][ring]
    grid["x"] = (rs * dx[order][:, None]).astype(np.float32)
    grid["y"] = (rs * dy[order][:, None]).astype(np.float32)
    grid["z"] = (rs * np.tan(elev)[None, :]).astype(np.float32)
    grid["pad"] = 1.0
    grid["intensity"] = rng.uniform(0.0, 255.0, (cols, rings)).astype(np.float32)
    grid["ring"] = np.arange(rings, dtype=np.uint16)[None, :]
    if drop_fraction > 0.0:
        keep = rng.uniform(0.0, 1.0, pts.shape[0]) >= drop_fraction
        pts = pts[keep]
    if shuffle:
        pts = pts[rng.permutation(pts.shape[0])]
    return np.ascontiguousarray(pts)


def make_batch(n_scans, rings=64, cols=1800, seed=1234, **kw):
    """n_scans scans with seeds seed, seed+1, ... (seeds 1234+scan_id)."""
    return [make_scan(rings, cols, seed + i, **kw) for i in range(n_scans)]


def concat(clouds):
    """Back-to-back copy of several scans, keeping the 32-byte record layout (np.concatenate would
    repack the fields)."""
    out = np.zeros(sum(len(c) for c in clouds), POINT_DTYPE)
    at = 0
    for c in clouds:
        out[at:at + len(c)] = c
        at += len(c)
    return out
