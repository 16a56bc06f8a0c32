"""Gather of the labelled clouds to one rank (the only exchange step of the path).

Scans are independent units (the reference node is stateless per message,
/root/reference/extraction/app/feature_extraction.cpp:173-179): each rank extracts its own scans
and rank `dst` collects the variable-length edge / surface clouds.  One process per GPU.

The exchange itself is the library's: lfx_comm_* / lfx_gather_counts / lfx_gather_payload (include/lfx.h) talk to
RCCL directly -- the totals first (all-gather of two numbers per rank), then one grouped send / recv over the direct
xGMI links; no ring, no reduction.  This module is the binding of those entry points (`RcclGather`) plus
`CloudGather`, which runs them one step behind the extraction: while the device extracts batch k, the clouds of
batch k-1 travel on a side stream, so rank dst's ingest (N-1 clouds per step) hides behind compute instead of adding
to it.  Every rank issues the same sequence of calls whatever its data, so the ranks cannot get out of step.

`gather_clouds` is the same protocol over `torch.distributed` (gloo on CPU): a rehearsal for the tests that run without
a GPU, never the product path.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import binding as B


class RcclGather:
    """One RCCL communicator (lfx_comm) tied to a FeatureExtraction context."""

    def __init__(self, fx, rank, world, unique_id):
        # the library the context was made by (the test-hooks build where an LFX_DEBUG_* switch asked for it): one copy of
        # the library, and with it one RCCL handle, per context
        self._L = fx._L
        self._fx = fx
        self.rank, self.world = rank, world
        self._comm = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        B.check(fx._ctx, self._L.lfx_comm_create(fx._ctx, C.cast(buf, C.c_void_p), rank, world, C.byref(self._comm)), self._L)

    @staticmethod
    def unique_id(fx=None):
        """Rank 0 makes it; hand the 128 bytes to every rank.  (fx: through the library of that context.)"""
        buf = (C.c_uint8 * 128)()
        L = fx._L if fx is not None else B.load()
        rc = L.lfx_comm_unique_id(C.cast(buf, C.c_void_p))
        if rc != 0:
            raise B.LfxError(rc, (L.lfx_last_error(None) or b"").decode())
        return bytes(buf)

    def close(self):
        if self._comm.value:
            self._L.lfx_comm_destroy(self._comm)
            self._comm = C.c_void_p()

    def stats(self):
        """lfx_comm_stats: {sends, receives, bytes_sent, bytes_received, all_gathers} posted by this rank so far."""
        out = (C.c_uint64 * 5)()
        self._L.lfx_comm_stats(self._comm, C.cast(out, C.c_void_p))
        return dict(zip(("sends", "receives", "bytes_sent", "bytes_received", "all_gathers"), [int(v) for v in out]))

    def counts(self, d_offsets, batch, stream=0, slot=0):
        B.check(self._fx._ctx, self._L.lfx_gather_counts_slot(self._fx._ctx, self._comm, int(slot), C.c_void_p(int(d_offsets)), batch,
                                                              C.c_void_p(int(stream))), self._L)

    def payload_group(self, steps, batch, floats_per_point, capacity_points, stream=0):
        """lfx_gather_payload2: one or two steps as ONE grouped exchange.  steps: dicts with dst, slot, edge, surface, offsets
        (device pointers of this rank's packed clouds) and, where this rank is the destination, edge_all, surface_all,
        offsets_all.  Returns every step's totals of every rank, a list of [world, 2] arrays."""
        arr = (B.GatherStep * len(steps))()
        counts = [np.zeros((self.world, 2), np.uint64) for _ in steps]
        for i, st in enumerate(steps):
            arr[i] = B.GatherStep(int(st["dst"]), int(st["slot"]), int(st["edge"]), int(st["surface"]), int(st["offsets"]),
                                  int(st.get("edge_all") or 0) or None, int(st.get("surface_all") or 0) or None,
                                  int(st.get("offsets_all") or 0) or None, counts[i].ctypes.data)
        B.check(self._fx._ctx, self._L.lfx_gather_payload2(self._fx._ctx, self._comm, arr, len(steps), batch, floats_per_point,
                                                           int(capacity_points), C.c_void_p(int(stream))), self._L)
        return counts

    def payload(self, dst, d_edge, d_surface, d_offsets, batch, floats_per_point, d_edge_all, d_surface_all, d_offsets_all,
                capacity_points, stream=0):
        """Returns the totals of every rank, [world, 2] (edge, surface)."""
        counts = np.zeros((self.world, 2), np.uint64)
        B.check(self._fx._ctx, self._L.lfx_gather_payload(
            self._fx._ctx, self._comm, dst, C.c_void_p(int(d_edge)), C.c_void_p(int(d_surface)), C.c_void_p(int(d_offsets)),
            batch, floats_per_point, C.c_void_p(int(d_edge_all or 0)), C.c_void_p(int(d_surface_all or 0)),
            C.c_void_p(int(d_offsets_all or 0)), int(capacity_points), C.c_void_p(counts.ctypes.data), C.c_void_p(int(stream))), self._L)
        return counts


def split_gathered(edge_all, surface_all, offsets_all, counts, batch):
    """Rank dst: the gathered buffers back into one dict per rank {edge, surface, offsets} (views)."""
    out, at_e, at_s = [], 0, 0
    for r in range(len(counts)):
        ne, ns = int(counts[r][0]), int(counts[r][1])
        out.append({"edge": edge_all[at_e:at_e + ne], "surface": surface_all[at_s:at_s + ns], "offsets": offsets_all[r]})
        at_e += ne
        at_s += ns
    return out


class CloudGather:
    """One-step-deep pipeline of the gather on side streams (GPU tensors, RCCL through the C ABI).

    submit(edge, surface, offsets, batch) registers this step's packed clouds (they must stay untouched until their gather
    is done: use `lanes` + 1 sets of buffers; `wait_buffer` orders their reuse), queues the all-gather of their
    totals, and completes the previous step's gather, returning on its destination rank the list split_gathered builds
    (views of the receive buffers, valid once `done` has passed; None elsewhere).  flush() completes the last one.

    Lanes.  A sender reaches a destination over ONE xGMI link (≈ 77 GB/s each way), and a GPU that extracts 780 k scans/s
    produces 117 GB/s of clouds: towards a fixed destination every sender is bound by that one link (66 % of the kernel
    rate, whatever the number of GPUs), and the destination's seven links together take in 0.54 TB/s.  With a rotating
    destination (step k -> rank k mod N) consecutive steps travel over DIFFERENT links -- if they are allowed to overlap:
    `unique_id` may be a list of communicator ids, one lane (communicator + side stream) each; step k's exchange runs on
    lane k mod len(ids), so that step k - 1's payload is still on its way while step k's leaves.

    Return shapes.  Without `pairs`, submit() and flush() return ONE step's result: on its destination rank the list
    split_gathered builds, None elsewhere (and None while nothing has completed).  With `pairs=True` steps complete two at a
    time: submit() returns None, or a LIST with one such result per step of the pair just completed (None at the places whose
    destination is another rank); flush() returns the list over every step it completed (possibly empty)."""

    def __init__(self, fx, rank, world, unique_id, dst=0, device=None, capacity_points=0, batch=1, floats_per_point=3, pairs=False,
                 profile=False):
        # dst = "rotate": step k's clouds go to rank k mod world (every rank then needs receive buffers): no single
        # rank's links carry all of the ingest
        self.rotate = dst == "rotate"
        self.dst, self.rank, self.world = (0 if self.rotate else int(dst)), rank, world
        self.fpp, self.batch, self.cap = floats_per_point, batch, int(capacity_points)
        ids = [unique_id] if isinstance(unique_id, (bytes, bytearray)) else list(unique_id)
        self.lanes = [(RcclGather(fx, rank, world, uid), torch.cuda.Stream(device=device)) for uid in ids]
        self.rccl, self.side = self.lanes[0]       # (the first lane under its old names: single-lane callers, stats)
        self.pending = None
        self.buffer_free = {}          # data_ptr of a send buffer -> event recorded after the gather that read it
        self.done = None               # event after the last completed gather's receives
        if self.rotate or rank == self.dst:
            # two receive sets, so that a consumer may still read step k-2's clouds while step k-1's arrive
            self.recv = [(torch.zeros((self.cap, self.fpp), dtype=torch.float32, device=device),
                          torch.zeros((self.cap, self.fpp), dtype=torch.float32, device=device),
                          torch.zeros((world, 2 * (batch + 1)), dtype=torch.int32, device=device)) for _ in range(2)]
        else:
            self.recv = [(None, None, None)] * 2
        # pairs: steps 2m and 2m + 1 travel as ONE grouped exchange on the one communicator (lfx_gather_payload2): with a
        # rotating destination their clouds go to two different ranks, i.e. over two of every sender's links at once -- what
        # two lanes buy, without a second communicator whose kernels could start in another order on another rank.  The pair
        # is posted one step after its second member was submitted (its totals have landed by then: the host does not wait),
        # so a caller needs FOUR sets of send buffers.
        self.pairs = bool(pairs)
        if self.pairs:
            assert len(self.lanes) == 1, "pairs: one communicator"
            # four receive sets: a pair's two destinations may be this rank twice in a row (world = 1 rehearsal), and a consumer
            # may still read the pair before
            if self.rotate or rank == self.dst:
                self.recv = self.recv + [(torch.zeros_like(self.recv[0][0]), torch.zeros_like(self.recv[0][1]), torch.zeros_like(self.recv[0][2]))
                                         for _ in range(2)]
        self.queue = []                # pairs: submitted steps whose payload has not been posted yet
        self.submitted = 0
        self.phase = 0
        self.step = 0
        self.received = 0              # gathers this rank has been the destination of: alternates the receive sets
        self.profile = bool(profile)   # record (start, end) timing events around every gather (gather_ms); off: nothing is kept
        self.spans = []
        # what this rank should have moved, from the totals every exchange returns: the other side of lfx_comm_stats
        self.acct = {"points_sent": 0, "tables_sent": 0, "points_received": 0, "tables_received": 0, "exchanges": 0}

    def _account(self, dst, counts, batch):
        self.acct["exchanges"] += 1
        if self.rank != dst:
            self.acct["points_sent"] += int(counts[self.rank][0]) + int(counts[self.rank][1])
            self.acct["tables_sent"] += 1
        else:
            for r in range(self.world):
                if r != dst:
                    self.acct["points_received"] += int(counts[r][0]) + int(counts[r][1])
                    self.acct["tables_received"] += 1

    def comm_report(self):
        """lfx_comm_stats of this rank's communicator(s) beside what the exchanges' own totals say it should have moved:
        bytes = 4 * floats_per_point per feature point + one offsets table (8 * (batch + 1) bytes) per cloud pair."""
        st = {"sends": 0, "receives": 0, "bytes_sent": 0, "bytes_received": 0, "all_gathers": 0}
        for rccl, _ in self.lanes:
            for k, v in rccl.stats().items():
                st[k] += v
        tab = 8 * (self.batch + 1)
        exp_s = self.acct["points_sent"] * 4 * self.fpp + self.acct["tables_sent"] * tab
        exp_r = self.acct["points_received"] * 4 * self.fpp + self.acct["tables_received"] * tab
        return dict(st, rank=self.rank, points_sent=self.acct["points_sent"], points_received=self.acct["points_received"],
                    exchanges=self.acct["exchanges"], expected_bytes_sent=exp_s, expected_bytes_received=exp_r,
                    bytes_match=bool(st["bytes_sent"] == exp_s and st["bytes_received"] == exp_r))

    def close(self):
        for rccl, _ in self.lanes:
            rccl.close()

    def wait_buffer(self, tensor):
        """Make the caller's current stream wait until the last gather reading `tensor` is done."""
        ev = self.buffer_free.pop(tensor.data_ptr(), None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def submit(self, edge, surface, offsets, batch):
        if self.pairs:
            return self._submit_paired(edge, surface, offsets, batch)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())
        rccl, side = self.lanes[(self.step + (1 if self.pending is not None else 0)) % len(self.lanes)]     # this step's lane
        prev, self.pending = self.pending, (edge, surface, offsets, batch)
        # the previous step's payload goes first: its totals landed long ago, and its side stream then does not sit on
        # this step's `ready` before moving data that has been waiting since the step before
        out = self._finish(prev) if prev is not None else None
        side.wait_event(ready)
        rccl.counts(offsets.data_ptr(), batch, side.cuda_stream)
        return out

    def _finish(self, p):
        edge, surface, offsets, batch = p
        rccl, side = self.lanes[self.step % len(self.lanes)]          # the lane its totals were queued on
        dst = self.step % self.world if self.rotate else self.dst
        self.last_dst = dst
        self.step += 1
        # (the receive set alternates with the gathers this rank RECEIVES: with an even world size and a rotating
        # destination a rank is the destination of steps of one parity only)
        ea, sa, oa = self.recv[self.received % 2] if self.rank == dst else (None, None, None)
        if self.rank == dst:
            self.received += 1
        t0 = None
        if self.profile:
            t0 = torch.cuda.Event(enable_timing=True)
            t0.record(side)
        counts = rccl.payload(dst, edge.data_ptr(), surface.data_ptr(), offsets.data_ptr(), batch, self.fpp,
                              ea.data_ptr() if ea is not None else 0, sa.data_ptr() if sa is not None else 0,
                              oa.data_ptr() if oa is not None else 0, self.cap, side.cuda_stream)
        self._account(dst, counts, batch)
        ev = torch.cuda.Event(enable_timing=self.profile)
        ev.record(side)
        if self.profile:
            self.spans.append((t0, ev))
        for t in (edge, surface, offsets):
            self.buffer_free[t.data_ptr()] = ev
        self.done = ev
        if self.rank != dst:
            return None
        return split_gathered(ea, sa, oa, counts, batch)

    # ---- pairs (lfx_gather_payload2)
    def _submit_paired(self, edge, surface, offsets, batch):
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())
        rccl, side = self.lanes[0]
        k = self.submitted                 # the step's number in the stream: its destination where that rotates
        slot = self.phase % 2              # its place in its pair = the slot of its totals (pairs start afresh after a flush)
        self.submitted += 1
        self.phase += 1
        out = None
        # a complete pair whose second member was submitted a step ago: its totals are on the host by now
        if len(self.queue) >= 2 and slot == 0:
            out = self._finish_group(self.queue[:2])
            self.queue = self.queue[2:]
        side.wait_event(ready)
        rccl.counts(offsets.data_ptr(), batch, side.cuda_stream, slot=slot)
        self.queue.append((edge, surface, offsets, batch, k, slot))
        return out

    def _finish_group(self, items):
        rccl, side = self.lanes[0]
        steps, recv_sets = [], []
        for edge, surface, offsets, batch, k, slot in items:
            dst = k % self.world if self.rotate else self.dst
            self.last_dst = dst
            st = {"dst": dst, "slot": slot, "edge": edge.data_ptr(), "surface": surface.data_ptr(), "offsets": offsets.data_ptr()}
            if self.rank == dst:
                ea, sa, oa = self.recv[self.received % len(self.recv)]
                self.received += 1
                st.update(edge_all=ea.data_ptr(), surface_all=sa.data_ptr(), offsets_all=oa.data_ptr())
                recv_sets.append((ea, sa, oa))
            else:
                recv_sets.append(None)
            steps.append(st)
        t0 = None
        if self.profile:
            t0 = torch.cuda.Event(enable_timing=True)
            t0.record(side)
        counts = rccl.payload_group(steps, items[0][3], self.fpp, self.cap, side.cuda_stream)
        for st, cn in zip(steps, counts):
            self._account(st["dst"], cn, items[0][3])
        ev = torch.cuda.Event(enable_timing=self.profile)
        ev.record(side)
        if self.profile:
            self.spans.append((t0, ev))
        for edge, surface, offsets, _, _, _ in items:
            for t in (edge, surface, offsets):
                self.buffer_free[t.data_ptr()] = ev
        self.done = ev
        self.step += len(items)
        outs = [split_gathered(r[0], r[1], r[2], c, items[0][3]) if r is not None else None for r, c in zip(recv_sets, counts)]
        return outs

    def gather_ms(self, reset=True):
        """(sum, count) of the side streams' time inside the gathers completed so far (payload exchange incl. its waits for
        the peers), in milliseconds; the side streams must be idle (flush())."""
        total = sum(a.elapsed_time(b) for a, b in self.spans)
        n = len(self.spans)
        if reset:
            self.spans = []
        return total, n

    def flush(self):
        if self.pairs:
            out = []
            while self.queue:
                n = 2 if len(self.queue) >= 2 and self.queue[0][5] == 0 else 1
                out += self._finish_group(self.queue[:n])
                self.queue = self.queue[n:]
            self.phase = 0
            for _, side in self.lanes:
                side.synchronize()
            return out
        prev, self.pending = self.pending, None
        out = self._finish(prev) if prev is not None else None
        for _, side in self.lanes:
            side.synchronize()
        return out


# --------------------------------------------------------------------------- CPU rehearsal (tests, gloo)
def _totals(offsets, batch):
    return torch.stack([offsets[batch], offsets[2 * batch + 1]]).to(torch.int64)


def gather_clouds(edge, surface, offsets, batch, dst=0, group=None):
    """The protocol of lfx_gather over torch.distributed (gloo): totals all-gathered first, then the clouds and the
    offsets tables to rank dst.  edge, surface: [capacity, k] f32 packed clouds of this rank; offsets: int32
    [2*(batch+1)].  Returns on rank dst what split_gathered returns, None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    totals = _totals(offsets, batch)
    all_totals = [torch.zeros_like(totals) for _ in range(world)]
    dist.all_gather(all_totals, totals, group=group)
    tot = torch.stack(all_totals).cpu()
    me, ms = int(tot[:, 0].max()), int(tot[:, 1].max())
    if me > edge.shape[0] or ms > surface.shape[0]:
        raise RuntimeError("packed feature buffers are smaller than the gathered clouds")
    e_send, s_send = edge[:me].contiguous(), surface[:ms].contiguous()
    if rank == dst:
        e_recv = [torch.empty_like(e_send) for _ in range(world)]
        s_recv = [torch.empty_like(s_send) for _ in range(world)]
        o_recv = [torch.empty_like(offsets) for _ in range(world)]
    else:
        e_recv = s_recv = o_recv = None
    dist.gather(e_send, e_recv, dst=dst, group=group)
    dist.gather(s_send, s_recv, dst=dst, group=group)
    dist.gather(offsets, o_recv, dst=dst, group=group)
    if rank != dst:
        return None
    return [{"edge": e_recv[r][:int(tot[r, 0])], "surface": s_recv[r][:int(tot[r, 1])], "offsets": o_recv[r]}
            for r in range(world)]


def gather_clouds_pair(steps, batch, group=None):
    """The protocol of lfx_gather_payload2 over torch.distributed (gloo): TWO steps' clouds, each to its own destination, with
    every send and receive of both steps posted before any is waited for (one group).  steps: [(edge, surface, offsets,
    dst), (edge, surface, offsets, dst)] with this rank's packed clouds of each step.  Returns a list of two: what
    gather_clouds returns for the step on its destination rank, None elsewhere."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    totals = []
    for edge, surface, offsets, _ in steps:                 # the totals of both steps first (lfx_gather_counts_slot, slots 0 and 1)
        t = _totals(offsets, batch)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t, group=group)
        totals.append(torch.stack(every).cpu())
    ops, recv = [], []
    for (edge, surface, offsets, dst), tot in zip(steps, totals):
        ne, ns = int(tot[rank, 0]), int(tot[rank, 1])
        if rank != dst:
            for t in (edge[:ne].contiguous(), surface[:ns].contiguous(), offsets.contiguous()):
                ops.append(dist.P2POp(dist.isend, t, dst, group))
            recv.append(None)
            continue
        parts = []
        for r in range(world):
            if r == dst:
                parts.append({"edge": edge[:ne].clone(), "surface": surface[:ns].clone(), "offsets": offsets.clone()})
                continue
            p = {"edge": torch.empty((int(tot[r, 0]),) + tuple(edge.shape[1:]), dtype=edge.dtype),
                 "surface": torch.empty((int(tot[r, 1]),) + tuple(surface.shape[1:]), dtype=surface.dtype),
                 "offsets": torch.empty_like(offsets)}
            for k in ("edge", "surface", "offsets"):
                ops.append(dist.P2POp(dist.irecv, p[k], r, group))
            parts.append(p)
        recv.append(parts)
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return recv


def shard_scans(n_scans, rank, world):
    """scan i -> rank i mod world."""
    return list(range(rank, n_scans, world))


def reassemble(per_rank, n_scans, world, batch):
    """Rank dst: the gathered per-rank clouds back into stream order.  per_rank[r] = {edge, surface, offsets} with
    offsets = [2][batch+1] exclusive prefixes of rank r's per-scan counts; scan i of the stream is rank i mod world's
    local scan i // world.  Returns [(edge_i, surface_i)] for i in range(n_scans)."""
    out = []
    for i in range(n_scans):
        r, k = i % world, i // world
        offs = per_rank[r]["offsets"]
        e0, e1 = int(offs[k]), int(offs[k + 1])
        s0, s1 = int(offs[batch + 1 + k]), int(offs[batch + 1 + k + 1])
        out.append((per_rank[r]["edge"][e0:e1], per_rank[r]["surface"][s0:s1]))
    return out
