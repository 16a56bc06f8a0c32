"""The route selection of run_batch as a table (lfx_route_choice = choose_route of lfx_api.hip with nothing around it: no
context, no device).  Each row: the report the previous batch left (counters), the state the choices before it left, and
what must be chosen for the next batch -- the organised-scan kernel or bucketing for every scan, ring transforms, the
length the fall-back launches are sized for, the two-launch tail, order repair first, the second unit pass's size."""
import ctypes as C

import pytest

from lidar_feature_extraction_amd import binding as LB

DEFER, REDO, SLOW, PREFIXED, FALLBACK, FUSED_RAN, BATCH, ORDER_FELL, TURNED, CUT_RAN, ZERO_FELL, HOLES_RAN, ZERO_GROUPS = range(13)


@pytest.fixture(scope="module")
def lib():
    L = LB.load()
    L.lfx_route_choice.restype = C.c_int
    L.lfx_route_choice.argtypes = [C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_uint32), C.c_int, C.c_uint32, C.c_uint32,
                                   C.POINTER(C.c_uint32)]
    return L


def choose(lib, report=None, rings=0, state=(0, 0, 0, 0), possible=1, batch=64, max_rings=64):
    rep = (C.c_uint32 * 14)()                                  # LFX_ROUTE_REPORT_WORDS
    for k, v in (report or {}).items():
        rep[k] = v
    # LFX_ROUTE_STATE_WORDS = 5: (use_xform, bucket_all, retry_in, pre_order, use_holes); rows written before the holes form
    # existed give four and get four back
    st = (C.c_uint32 * 5)(*(tuple(state) + (0,) * (5 - len(state))))
    ch = (C.c_uint32 * 7)()                                    # LFX_ROUTE_CHOICE_WORDS
    assert lib.lfx_route_choice(rep, rings, st, possible, batch, max_rings, ch) == 0
    out = dict(fused=ch[0], xform=ch[1], fb_grid=ch[2], short_tail=ch[3], pre_order=ch[4], redo_cap=ch[5], holes=ch[6])
    if len(state) == 4:
        assert st[4] == 0, "nothing in these rows asks for the holes form"
    return out, tuple(st)[:len(state)]


CLEAN = {FUSED_RAN: 1, BATCH: 64, FALLBACK: 0}
ROWS = [
    # name, kwargs, expected choice (subset), expected state (use_xform, bucket_all, retry_in, pre_order)
    ("first batch of a context: organised route, fall-back launches for 8 scans, the full tail (nothing known yet)",
     dict(), dict(fused=1, xform=0, fb_grid=8, short_tail=0, pre_order=0, redo_cap=64 * 64), (0, 0, 0, 0)),
    ("a clean organised stream: the two-launch tail, second pass sized for 256 rings",
     dict(report=CLEAN, rings=4096), dict(fused=1, xform=0, fb_grid=8, short_tail=1, redo_cap=256), (0, 0, 0, 0)),
    ("three scans fell back: launches for 2 x 3 + 8 entries, the full tail again",
     dict(report={FUSED_RAN: 1, BATCH: 64, FALLBACK: 3}, rings=4096), dict(fused=1, fb_grid=14, short_tail=0), (0, 0, 0, 0)),
    ("the guess never exceeds the batch",
     dict(report={FUSED_RAN: 1, BATCH: 64, FALLBACK: 15}, rings=4096, batch=16), dict(fused=1, fb_grid=16), (0, 0, 0, 0)),
    ("most scans fell back for other reasons than order: every scan is bucketed, the organised route is retried in 16 batches",
     dict(report={FUSED_RAN: 1, BATCH: 64, FALLBACK: 40, ORDER_FELL: 2}, rings=4096), dict(fused=0, xform=0, fb_grid=64, short_tail=0), (0, 1, 15, 0)),
    ("... still bucketing while the countdown runs (the report of a bucketed batch says nothing about the organised route)",
     dict(report={FUSED_RAN: 0, BATCH: 64, FALLBACK: 64}, rings=4096, state=(0, 1, 5, 0)), dict(fused=0, fb_grid=64), (0, 1, 4, 0)),
    ("... the retry: organised route again, launches sized for the whole batch",
     dict(report={FUSED_RAN: 0, BATCH: 64, FALLBACK: 64}, rings=4096, state=(0, 1, 1, 0)), dict(fused=1, fb_grid=64, short_tail=0), (0, 1, 16, 0)),
    ("... and a retry that went well ends the bucketing",
     dict(report={FUSED_RAN: 1, BATCH: 64, FALLBACK: 1}, rings=4096, state=(0, 1, 16, 0)), dict(fused=1, fb_grid=10), (0, 0, 16, 0)),
    ("most scans fell back for the angle order of their rings alone: ring transforms from now on, organised route kept",
     dict(report={FUSED_RAN: 1, BATCH: 64, FALLBACK: 60, ORDER_FELL: 58}, rings=4096), dict(fused=1, xform=1, fb_grid=64), (1, 0, 0, 0)),
    ("with the transforms the stream is clean: they stay while rings keep needing them",
     dict(report={FUSED_RAN: 1, BATCH: 64, FALLBACK: 0, CUT_RAN: 1, TURNED: 4096}, rings=4096, state=(1, 0, 0, 0)),
     dict(fused=1, xform=1, short_tail=1), (1, 0, 0, 0)),
    ("(almost) no ring needs a transform any more: back to plain loads",
     dict(report={FUSED_RAN: 1, BATCH: 64, FALLBACK: 0, CUT_RAN: 1, TURNED: 10}, rings=4096, state=(1, 0, 0, 0)),
     dict(fused=1, xform=0, short_tail=1), (0, 0, 0, 0)),
    ("most scans fell back for a (0, 0, 0) record alone (zero filter on): the holes form from now on, organised route kept",
     dict(report={FUSED_RAN: 1, BATCH: 64, FALLBACK: 60, ZERO_FELL: 58}, rings=4096, state=(0, 0, 0, 0, 0)), dict(fused=1, xform=0, holes=1, fb_grid=64),
     (0, 0, 0, 0, 1)),
    ("with the count pass the stream is clean: the holes form stays while ring groups keep holding such records (one-launch tail)",
     dict(report={FUSED_RAN: 1, BATCH: 64, FALLBACK: 0, HOLES_RAN: 1, ZERO_GROUPS: 1024}, rings=4096, state=(0, 0, 0, 0, 1)),
     dict(fused=1, holes=1, short_tail=1), (0, 0, 0, 0, 1)),
    ("(almost) no ring group holds a zero record any more: back to the plain form",
     dict(report={FUSED_RAN: 1, BATCH: 64, FALLBACK: 0, HOLES_RAN: 1, ZERO_GROUPS: 1}, rings=4096, state=(0, 0, 0, 0, 1)),
     dict(fused=1, holes=0, short_tail=1), (0, 0, 0, 0, 0)),
    ("a stream hint (grid with holes) before any report: the count pass from the first batch on",
     dict(state=(0, 0, 0, 0, 1)), dict(fused=1, holes=1, fb_grid=8), (0, 0, 0, 0, 1)),
    ("rings turned AND holes: the transforms win (the holes form takes rings as they stand)",
     dict(state=(1, 0, 0, 0, 1)), dict(fused=1, xform=1, holes=0), (1, 0, 0, 0, 1)),
    ("a stream hint (turned rings) before any report: transforms from the first batch on",
     dict(state=(1, 0, 0, 0)), dict(fused=1, xform=1, fb_grid=8), (1, 0, 0, 0)),
    ("a stream hint (no grid): bucketing from the first batch on, retried later",
     dict(state=(0, 1, 16, 0)), dict(fused=0, fb_grid=64), (0, 1, 15, 0)),
    ("the organised-scan kernel is not possible (ring count unknown, another record layout): bucketing, no state touched",
     dict(report=CLEAN, rings=4096, possible=0), dict(fused=0, xform=0, fb_grid=64, short_tail=0), (0, 0, 0, 0)),
    ("more than a twentieth of the rings had their order repaired: repair before the first unit pass, second pass sized 2 x + 256",
     dict(report={FUSED_RAN: 0, BATCH: 64, FALLBACK: 64, REDO: 300}, rings=4096, possible=0), dict(pre_order=1, redo_cap=856), (0, 0, 0, 1)),
    ("... and off again once the repairs (before or after the pass) stop",
     dict(report={FUSED_RAN: 0, BATCH: 64, FALLBACK: 64, REDO: 3, PREFIXED: 100}, rings=4096, possible=0, state=(0, 0, 0, 1)),
     dict(pre_order=0, redo_cap=262), (0, 0, 0, 0)),
    ("repairs counted before the pass keep it on",
     dict(report={FUSED_RAN: 0, BATCH: 64, FALLBACK: 64, REDO: 0, PREFIXED: 4000}, rings=4096, possible=0, state=(0, 0, 0, 1)),
     dict(pre_order=1, redo_cap=256), (0, 0, 0, 1)),
]


@pytest.mark.parametrize("name,kw,want,want_state", ROWS, ids=[r[0][:60] for r in ROWS])
def test_route_choice_table(lib, name, kw, want, want_state):
    got, state = choose(lib, **kw)
    for k, v in want.items():
        assert got[k] == v, "%s: %s = %s, expected %s (all: %s)" % (name, k, got[k], v, got)
    assert state == want_state, "%s: state %s, expected %s" % (name, state, want_state)


def test_a_stream_that_turns_bad_and_recovers(lib):
    """The state carried through a sequence: clean -> 50 of 64 scans ragged for 20 batches -> clean again."""
    state = (0, 0, 0, 0)
    report, rings = None, 0
    routes = []
    for k in range(40):
        ch, state = choose(lib, report=report, rings=rings, state=state)
        routes.append(ch["fused"])
        bad = 3 <= k < 23
        if ch["fused"]:
            report = {FUSED_RAN: 1, BATCH: 64, FALLBACK: 50 if bad else 0}
        else:
            report = {FUSED_RAN: 0, BATCH: 64, FALLBACK: 64}
        rings = 4096
    assert routes[:4] == [1, 1, 1, 1]              # the batch after the first bad report is the first to be bucketed whole
    assert routes[4:19] == [0] * 15 and routes[19] == 1      # retried after 16 batches, still bad
    assert routes[20:35] == [0] * 15 and routes[35] == 1     # the second retry finds the stream clean ...
    assert routes[36:] == [1, 1, 1, 1]                       # ... and the organised route stays


def test_a_stream_that_grows_holes_and_loses_them(lib):
    """The state carried through a sequence (zero filter on): a clean grid -> zero records in every scan for 20 batches -> clean
    again.  The plain form refuses such scans (they fall back one by one), the report says why, the holes form takes over with
    the next batch and stays while ring groups keep holding zero records; then back to the plain form -- never the bucketing
    route for every scan."""
    state = (0, 0, 0, 0, 0)
    report, rings = None, 0
    seen = []
    for k in range(40):
        ch, state = choose(lib, report=report, rings=rings, state=state)
        seen.append((ch["fused"], ch["holes"]))
        holes_now = 3 <= k < 23
        if ch["holes"]:
            report = {FUSED_RAN: 1, BATCH: 64, FALLBACK: 0, HOLES_RAN: 1, ZERO_GROUPS: 1024 if holes_now else 0}
        else:
            report = {FUSED_RAN: 1, BATCH: 64, FALLBACK: 64 if holes_now else 0, ZERO_FELL: 64 if holes_now else 0}
        rings = 4096
    assert all(f == 1 for f, _ in seen), "the organised route throughout: %s" % seen
    assert [h for _, h in seen[:4]] == [0, 0, 0, 0]            # the batch after the first report of zero records is the first with the count pass
    assert all(h == 1 for _, h in seen[4:24]), seen[4:24]
    assert all(h == 0 for _, h in seen[25:]), seen[24:]         # one batch of holes form on clean scans reports "no zero records": back
