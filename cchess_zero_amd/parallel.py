"""Multi-GPU: one process per GPU, games sharded by rank, (s, pi, z) records exchanged with one collective.

The reference's only parallelism is single-process tower data parallelism through host memory
(policy_value_network_gpus.py:66-95,216-250) and it plays ONE game at a time (main.py:1228-1230).
Here game g belongs to rank g % world; a search never communicates; when a batch of games has finished
every rank contributes its packed records to an all-gather (RCCL over xGMI on the GPU box, gloo in
the CPU tests) — the multi-GPU form of `self.data_buffer.extend(...)`, main.py:1240.  Records are
fixed-size (selfplay.REC_BYTES), ranks pad to the longest shard so that a single all_gather moves
everything; on the 8-GPU full mesh that is one hop per peer (per-link bound, 7 x ~153 GB/s per GPU).
"""
import os

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


class RecordExchange:
    """The per-ply record exchange of a long self-play run (configs[3]): every rank contributes what it drained since
    the last call, WITHOUT any host synchronisation on the collective — the payload has a fixed capacity, so nothing has
    to be agreed on before the buffers are allocated.  gather_records_device needs the largest shard size on the host
    (a count all-gather + .item()), i.e. a host wait on the slowest rank in every exchange: with 8 ranks under a small CPU
    quota that wait is scheduling jitter inside the timed region.  Here: one all_gather_into_tensor of [capacity + 1, REC_BYTES]
    per rank (row 0 carries the count); rows beyond the capacity are carried over to the next exchange, so the number of
    collectives is the same on every rank whatever the shard sizes.  The gathered tensor and the counts stay on the
    collective's device; the consumer looks at them whenever it wants (results())."""

    def __init__(self, capacity, device, group=None):
        self.capacity, self.device, self.group = int(capacity), torch.device(device), group
        self.world = dist.get_world_size(group)
        self.carry = torch.zeros((0, REC_BYTES), dtype=torch.uint8, device=self.device)
        self.exchanges = 0
        self._last = None
        # buffers are allocated once: the send block, and TWO receive blocks used alternately, so that the tensor handed out
        # by one exchange stays valid until the exchange after the next (the consumer looks at it whenever it wants)
        self._send = torch.zeros((self.capacity + 1, REC_BYTES), dtype=torch.uint8, device=self.device)
        self._out = [torch.empty((self.world, self.capacity + 1, REC_BYTES), dtype=torch.uint8, device=self.device) for _ in range(2)]

    def exchange(self, rec):
        """rec: uint8 [n, REC_BYTES] on the collective's device (n is known on the host: it is a tensor shape).  Returns the
        gathered [world, capacity + 1, REC_BYTES] block (row 0 of each rank: its count); valid until the next-but-one call."""
        rec = rec.reshape(-1, REC_BYTES)
        if self.carry.shape[0]:
            rec = torch.cat([self.carry, rec], 0)
        n = min(rec.shape[0], self.capacity)
        send = self._send
        send[0, :8].view(torch.int64).fill_(n)          # a fill kernel with a scalar argument: no host-to-device copy
        if n:
            send[1:1 + n] = rec[:n]
        self.carry = rec[n:].clone() if rec.shape[0] > n else rec[:0]
        out = self._out[self.exchanges & 1]
        dist.all_gather_into_tensor(out.view(self.world * (self.capacity + 1), REC_BYTES), send, group=self.group)
        self.exchanges += 1
        self._last = out
        return out

    def flush(self):
        """Drain what the capacity held back: exchanges of empty shards until no rank has rows pending (the ranks agree on that
        with one all-reduce per round — this is for the END of a run, outside any timed region).  Returns the gathered blocks
        (clones).  A consumer that stops calling exchange() must call this, or the carried rows are lost."""
        outs = []
        while True:
            t = torch.tensor([self.pending()], dtype=torch.int64, device=self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            if int(t.item()) == 0:
                return outs
            outs.append(self.exchange(self.carry[:0]).clone())

    def pending(self):
        """Records held back because the last shard exceeded the capacity (sent by the next exchange)."""
        return int(self.carry.shape[0])

    @staticmethod
    def unpack(out):
        """[world, capacity + 1, REC_BYTES] -> (records [sum n_r, REC_BYTES] host array in rank order, counts int64 [world])."""
        o = out.cpu()
        counts = o[:, 0, :8].contiguous().view(torch.int64).reshape(-1)
        parts = [o[r, 1:1 + int(counts[r])] for r in range(o.shape[0])]
        rec = torch.cat(parts, 0).numpy() if parts else np.zeros((0, REC_BYTES), np.uint8)
        return rec.reshape(-1, REC_BYTES), counts.numpy()


# ---- bench.py's multi-rank plumbing (kept here so that the gloo CPU tests run the very same code with 8 ranks) ----------
def max_over_ranks(seconds, device):
    """The slowest rank's time for a barrier-bracketed region (the bench contract: MAX over ranks)."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(seconds)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(values, device):
    """Element-wise sum of a short list of numbers over the ranks -> list of floats (float64 on the collective's device)."""
    if not (dist.is_available() and dist.is_initialized()):
        return [float(v) for v in values]
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    dist.all_reduce(t)
    return t.tolist()


def per_rank(value, device):
    """[value of rank 0, value of rank 1, ...] on every rank."""
    if not (dist.is_available() and dist.is_initialized()):
        return [float(value)]
    world, rank = dist.get_world_size(), dist.get_rank()
    t = torch.zeros(world, dtype=torch.float64, device=device)
    t[rank] = float(value)
    dist.all_reduce(t)
    return t.tolist()


def gather_selfcheck(device):
    """A ragged token batch through gather_records_device and through RecordExchange: True when every rank sees every
    rank's rows, else a string saying what failed (the bench line must survive a broken exchange)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    try:
        tok = torch.zeros((4 + rank, REC_BYTES), dtype=torch.uint8, device=device)
        tok[:, 0] = rank + 1
        allrec, counts = gather_records_device(tok)
        ok = counts.tolist() == [4 + r for r in range(world)] and int(allrec[world - 1, 0, 0]) == world
        ex = RecordExchange(5, device)                     # capacity 5 < the larger shards: exercises the carry-over
        rounds = (4 + world - 1 + 4) // 5                   # enough rounds for the largest shard (4 + world - 1 rows)
        seen = [0] * world
        for i in range(rounds):
            rec_i, c_i = RecordExchange.unpack(ex.exchange(tok if i == 0 else tok[:0]))
            o = 0
            for r in range(world):
                n = int(c_i[r])
                ok = ok and n <= 5 and all(int(x) == r + 1 for x in rec_i[o:o + n, 0])
                seen[r] += n
                o += n
        ok = ok and seen == [4 + r for r in range(world)] and ex.pending() == 0
        return bool(ok)
    except Exception as e:
        return "failed: %r" % (e,)


def pin_rank_to_cpus(local_rank, local_world, max_per_rank=8):
    """One process per GPU: give every rank its own CPUs so that 8 launcher threads do not migrate over (and contend for)
    the same cores — under a cgroup CPU quota smaller than the visible CPU count the scheduler otherwise spreads runnable
    threads over all visible CPUs and throttles them together.  Returns the CPU list, or None when affinity cannot be
    set.  The CPUs of a rank are contiguous: rank r gets allowed[r * k : (r + 1) * k], k = min(max_per_rank,
    allowed // local_world)."""
    global _affinity_before_pin
    try:
        allowed = sorted(os.sched_getaffinity(0))
        k = max(1, min(int(max_per_rank), len(allowed) // max(1, int(local_world))))
        mine = allowed[int(local_rank) * k:(int(local_rank) + 1) * k] or allowed
        os.sched_setaffinity(0, mine)
        _affinity_before_pin = allowed
        return mine
    except Exception:
        return None


_affinity_before_pin = None


def unpin_cpus():
    """Back to the CPU set the process had before pin_rank_to_cpus (bench.py's CPU baseline runs on all of the box's cores
    after the ranks are done)."""
    global _affinity_before_pin
    if _affinity_before_pin:
        try:
            os.sched_setaffinity(0, _affinity_before_pin)
        except Exception:
            pass
        _affinity_before_pin = None


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
