"""Gather of the labelled clouds to one rank (the only exchange step of the path).

Scans are independent units (the reference node is stateless per message,
/root/reference/extraction/app/feature_extraction.cpp:173-179): each rank extracts its own scans
and rank `dst` collects the variable-length edge / surface clouds.  One process per GPU;
`torch.distributed` backend "nccl" is RCCL over xGMI on the MI355X node, "gloo" on CPU (tests).

Message shape: the totals first (2 ints per rank, all-gather), then one padded gather per cloud
kind -- rank dst receives over the direct links of all peers at once; no ring, no reduction.

`gather_clouds` does that synchronously.  `CloudGather` pipelines it one step deep: while the
device extracts batch k, the clouds of batch k-1 travel (on a side stream for GPU tensors), so
rank dst's ingest (N-1 clouds per step) hides behind compute instead of adding to it.  Every rank
issues the same sequence of collectives whatever its data, so the ranks cannot get out of step.
"""
import torch
import torch.distributed as dist


def _totals(offsets, batch):
    return torch.stack([offsets[batch], offsets[2 * batch + 1]]).to(torch.int64)


def _gather_payload(edge, surface, offsets, tot, dst, group):
    """tot: [world, 2] int64 on the host.  Padded gather of both clouds and the offsets table."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
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


def gather_clouds(edge, surface, offsets, batch, dst=0, group=None):
    """edge, surface: [capacity, 4] f32 packed clouds of this rank (lfx_pack_features);
    offsets: int32 [2*(batch+1)] exclusive prefixes of the per-scan counts (entry [batch] and
    [2*batch+1] are the totals).  Returns on rank dst a list with one dict per rank
    {edge [n_e,4], surface [n_s,4], offsets}, None elsewhere.  Synchronises the host once
    (the totals decide the padded message length)."""
    world = dist.get_world_size(group)
    totals = _totals(offsets, batch)
    all_totals = [torch.zeros_like(totals) for _ in range(world)]
    dist.all_gather(all_totals, totals, group=group)
    tot = torch.stack(all_totals).cpu()
    return _gather_payload(edge, surface, offsets, tot, dst, group)


class CloudGather:
    """One-step-deep pipeline of gather_clouds.

    submit(edge, surface, offsets, batch) registers this step's packed clouds (they must stay
    untouched until the NEXT submit returns: use two sets of buffers) and completes the previous
    step's gather, returning its result (rank dst) or None.  flush() completes the last one.
    For GPU tensors everything is issued on a side stream that first waits for the work queued on
    the caller's stream at submit time; `done_event` of the returned step lets the caller order the
    reuse of a buffer after its gather."""

    def __init__(self, dst=0, group=None, device=None):
        self.dst, self.group = dst, group
        self.cuda = device is not None and torch.device(device).type == "cuda"
        self.side = torch.cuda.Stream(device=device) if self.cuda else None
        self.pending = None
        self.buffer_free = {}          # id(buffer) -> event recorded after the gather that read it

    def wait_buffer(self, tensor):
        """Make the caller's current stream wait until the last gather reading `tensor` is done."""
        ev = self.buffer_free.pop(tensor.data_ptr(), None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def _issue_totals(self, offsets, batch):
        world = dist.get_world_size(self.group)
        totals = _totals(offsets, batch)
        all_totals = [torch.zeros_like(totals) for _ in range(world)]
        dist.all_gather(all_totals, totals, group=self.group)
        return all_totals

    def submit(self, edge, surface, offsets, batch):
        if self.cuda:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.side):
                self.side.wait_event(ready)
                all_totals = self._issue_totals(offsets, batch)
                # the totals go to pinned host memory right behind their all-gather, with an event of their
                # own: finishing this step later waits for THAT event only, not for whatever the side stream
                # has been given since (the next step's collectives wait for the next step's extraction)
                host_tot = torch.empty((len(all_totals), 2), dtype=torch.int64, pin_memory=True)
                host_tot.copy_(torch.stack(all_totals), non_blocking=True)
                landed = torch.cuda.Event()
                landed.record(self.side)
            totals = (host_tot, landed)
        else:
            totals = (self._issue_totals(offsets, batch), None)
        prev, self.pending = self.pending, (edge, surface, offsets, totals)
        return self._finish(prev) if prev is not None else None

    def _finish(self, p):
        edge, surface, offsets, (tot, landed) = p
        if self.cuda:
            landed.synchronize()
            with torch.cuda.stream(self.side):
                out = _gather_payload(edge, surface, offsets, tot, self.dst, self.group)
                ev = torch.cuda.Event()
                ev.record(self.side)
            for t in (edge, surface, offsets):
                self.buffer_free[t.data_ptr()] = ev
            return out
        tot = torch.stack(tot).cpu()
        return _gather_payload(edge, surface, offsets, tot, self.dst, self.group)

    def flush(self):
        prev, self.pending = self.pending, None
        out = self._finish(prev) if prev is not None else None
        if self.cuda:
            self.side.synchronize()
        return out


def shard_scans(n_scans, rank, world):
    """scan i -> rank i mod world."""
    return list(range(rank, n_scans, world))
