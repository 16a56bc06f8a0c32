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
UNITS = ["lfx_api", "lfx_wire", "lfx_downsample", "lfx_localize"]          # (lfx_gather.hip holds no device code)


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


def _regs(text):
    """Vector registers an operand list names: v7, v[4:7]."""
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", text):
        out.add(int(m.group(1)))
    return out


def test_look_back_sweep_registers_are_left_alone_until_its_wait():
    """The organised-scan kernel issues the first sweep of its look-back (three `global_load_dwordx4 ... sc1` in one asm
    statement, unit_sweep_issue) a stage before it waits for it (`s_waitcnt vmcnt(1)` in an asm statement of its own,
    unit_look_back).  The compiler does not know the destinations are in flight in between: if it ever copied, spilled or
    reused them there, the wave would read registers whose data has not landed -- and a path that never waits (the
    scan's first unit has no predecessor to sum) would leave a load in flight into registers that are someone else's by
    then.  Every sweep of every instantiation is followed along every path of the listing's control flow to an asm wait."""
    subprocess.check_call(["make", "-s", "-j4", "-C", CSRC, "asmfile"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    with open(os.path.join(BUILD, "lfx_api_gfx950.s")) as f:
        lines = f.read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN3lfx20ring_unit_org_kernel.*:\s*(;.*)?$", l)]
    assert len(starts) >= 16, "organised-scan kernel instantiations not found in the listing"
    checked = 0
    for st in starts:
        end = next(i for i in range(st, len(lines)) if "s_endpgm" in lines[i])
        body = lines[st:end + 1]
        heads = [i for i, l in enumerate(body) if re.search(r"global_load_dwordx4 v\[\d+:\d+\], v\[\d+:\d+\], off sc1", l)]
        assert heads, "sweep loads not found"
        asm_waits = {i for i in range(len(body)) if re.search(r"s_waitcnt vmcnt\([01]\)", body[i]) and any("ASMSTART" in body[k] for k in range(i - 2, i))}
        labels = {b.split(":")[0]: k for k, b in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", b)}
        for head in heads:
            group = [i for i in range(head, head + 9) if re.search(r"global_load_dwordx4 v\[\d+:\d+\], v\[\d+:\d+\], off( offset:\d+)? sc1", body[i])]
            assert len(group) == 3, "a sweep is three loads in one asm statement"
            dest = set()
            for i in group:
                dest |= _regs(body[i].split(",")[0])
            # every path from the sweep on (the blocks are not laid out in program order) must come to an asm wait, and no
            # instruction before it may name a destination of the sweep
            todo, seen = [group[-1] + 1], set()
            while todo:
                i = todo.pop()
                while i not in seen:
                    seen.add(i)
                    if i in asm_waits:
                        break
                    l = body[i].split(";")[0]
                    i += 1
                    if not l.strip() or l.strip().endswith(":") or l.lstrip().startswith("."):
                        continue
                    assert "s_endpgm" not in l, "%s: a path leaves the kernel with a sweep in flight" % lines[st][:60]
                    assert not (_regs(l) & dest), "%s: line %d touches a sweep destination before the wait: %s" % (lines[st][:60], i, l.strip())
                    m = re.search(r"s_(c?)branch\w*\s+(\.LBB\d+_\d+)", l)
                    if m:
                        todo.append(labels[m.group(2)])
                        if not m.group(1):
                            break
        checked += 1
    assert checked >= 16
