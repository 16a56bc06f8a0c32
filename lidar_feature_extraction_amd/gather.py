"""Gather of the labelled clouds to one rank (the only exchange step of the path).

Scans are independent units (the reference node is stateless per message,
/root/reference/extraction/app/feature_extraction.cpp:173-179): each rank extracts its own scans
and rank `dst` collects the variable-length edge / surface clouds.  One process per GPU;
`torch.distributed` backend "nccl" is RCCL over xGMI on the MI355X node, "gloo" on CPU (tests).

Message shape: the totals first (2 ints per rank, all-gather), then one padded gather per cloud
kind -- rank dst receives over the direct links of all peers at once; no ring, no reduction.
"""
import torch
import torch.distributed as dist


def gather_clouds(edge, surface, offsets, batch, dst=0, group=None):
    """edge, surface: [capacity, 4] f32 packed clouds of this rank (lfx_pack_features);
    offsets: int32 [2*(batch+1)] exclusive prefixes of the per-scan counts (entry [batch] and
    [2*batch+1] are the totals).  Returns on rank dst a list with one dict per rank
    {edge [n_e,4], surface [n_s,4], offsets}, None elsewhere.  Synchronises the host once
    (the totals decide the padded message length)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    totals = torch.stack([offsets[batch], offsets[2 * batch + 1]]).to(torch.int64)
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


def shard_scans(n_scans, rank, world):
    """scan i -> rank i mod world."""
    return list(range(rank, n_scans, world))
