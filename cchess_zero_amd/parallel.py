"""Multi-GPU: one process per GPU, games sharded by rank, (s, pi, z) records exchanged with one collective.

The reference's only parallelism is single-process tower data parallelism through host memory
(policy_value_network_gpus.py:66-95,216-250) and it plays ONE game at a time (main.py:1228-1230).
Here game g belongs to rank g % world; a search never communicates; when a batch of games has finished
every rank contributes its packed records to an all-gather (RCCL over xGMI on the GPU box, gloo in
the CPU tests) — the multi-GPU form of `self.data_buffer.extend(...)`, main.py:1240.  Records are
fixed-size (selfplay.REC_BYTES), ranks pad to the longest shard so that a single all_gather moves
everything; on the 8-GPU full mesh that is one hop per peer (per-link bound, 7 x ~153 GB/s per GPU).
"""
import numpy as np
import torch
import torch.distributed as dist

from ._lib import REC_BYTES


def shard_games(n_games, rank, world):
    """Indices of the games rank `rank` owns (g % world == rank)."""
    return np.arange(rank, n_games, world)


def gather_records_device(rec, group=None):
    """All-gather packed records from every rank WITHOUT leaving the device: rec is a uint8 [n_r, REC_BYTES] tensor on
    the collective's device (the GPU for RCCL, the CPU for gloo); returns (all [world, m, REC_BYTES] padded to the longest
    shard m, counts int64 [world]) on that device.  Two collectives: the per-rank counts (8 bytes each), then one padded
    uint8 all_gather_into_tensor — on the 8-GPU xGMI mesh one hop per peer.  Only the count exchange synchronises with
    the host (the padded size has to be known to allocate)."""
    world = dist.get_world_size(group)
    rec = rec.reshape(-1, REC_BYTES)
    dev = rec.device
    n = torch.tensor([rec.shape[0]], dtype=torch.int64, device=dev)
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, n, group=group)
    m = int(counts.max().item())
    out = torch.empty((world, m, REC_BYTES), dtype=torch.uint8, device=dev)
    if m == 0:
        return out, counts
    if rec.shape[0] == m:
        pad = rec.contiguous()
    else:
        pad = torch.zeros((m, REC_BYTES), dtype=torch.uint8, device=dev)
        pad[:rec.shape[0]] = rec
    dist.all_gather_into_tensor(out.view(world * m, REC_BYTES), pad, group=group)
    return out, counts


def gather_records(rec, device=None, group=None):
    """All-gather packed records [n_r, REC_BYTES] from every rank -> host array [sum n_r, REC_BYTES] (same on all ranks,
    ordered by rank): the multi-GPU `data_buffer.extend` (main.py:1240).  rec: a device tensor (stays on the device until
    the gathered result is copied out once) or a host array."""
    if not (dist.is_available() and dist.is_initialized()):
        if torch.is_tensor(rec):
            rec = rec.cpu().numpy()
        return np.ascontiguousarray(rec, np.uint8).reshape(-1, REC_BYTES)
    backend = dist.get_backend(group)
    dev = torch.device(device) if device is not None else (torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu"))
    if not torch.is_tensor(rec):
        rec = torch.from_numpy(np.ascontiguousarray(rec, np.uint8).reshape(-1, REC_BYTES))
    out, counts = gather_records_device(rec.to(dev), group)
    counts = counts.cpu().numpy()
    out = out.cpu().numpy()
    if out.shape[1] == 0:
        return np.zeros((0, REC_BYTES), np.uint8)
    return np.concatenate([out[r, :counts[r]] for r in range(len(counts))], axis=0)


def broadcast_weights(module, src=0, group=None):
    """After a policy update on rank `src`, every rank's replica gets the new parameters (RCCL broadcast of
    ~2.5 M fp32 values for the 7-block net); replaces the parameter-server variables on /cpu:0 of
    policy_value_network_gpus.py:206-214."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)


def allreduce_gradients(module, group=None):
    """Data-parallel training: average gradients over ranks in one flattened bucket; replaces
    average_gradients, policy_value_network_gpus.py:216-250."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    grads = [p.grad for p in module.parameters() if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat /= dist.get_world_size(group)
    o = 0
    for g in grads:
        g.copy_(flat[o:o + g.numel()].view_as(g))
        o += g.numel()
