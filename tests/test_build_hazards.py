"""A toolchain hazard found on the way: hipcc 7.2 can materialise a wave-uniform 64-bit constant in scalar
registers as `s_mov_b64 s[a:b], <64-bit literal>`; gfx950 has no 64-bit literals, the low word is what arrives (an infinity
became 0.0 in the grid search).  The compiler's assembly for the library must hold no such instruction.  CPU only (hipcc
cross-compiles; `make asmfile` rebuilds only when the sources changed)."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lidar_feature_extraction_amd", "csrc")
BUILD = os.path.join(ROOT, "lidar_feature_extraction_amd", "_build")
UNITS = ["lfx_api", "lfx_unit_v0", "lfx_unit_v1", "lfx_unit_v2", "lfx_unit_v3", "lfx_wire", "lfx_downsample", "lfx_localize"]          # (lfx_gather.hip holds no device code)


def test_no_scalar_move_with_a_64_bit_literal():
    subprocess.check_call(["make", "-s", "-j4", "-C", CSRC, "asmfile"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    bad = re.compile(r"\bs_mov_b64\s+s\[\d+:\d+\],\s*(0x[0-9a-fA-F]{9,}|-?\d{10,})")
    hits = []
    for unit in UNITS:
        with open(os.path.join(BUILD, unit + "_gfx950.s")) as f:
            for no, line in enumerate(f, 1):
                if bad.search(line):
                    hits.append("%s %d: %s" % (unit, no, line.strip()))
    assert not hits, "scalar 64-bit literals (truncated on gfx950):\n" + "\n".join(hits[:10])

