"""Candidate sharding across the GPUs of one node and the one exchange step of the path.

Not in the reference (it has no parallelism in the GP step).  Candidates are independent given
(X, y, ls, L, alpha), so rank r owns the contiguous block [lo, hi) of the candidate rows and
recomputes the N x N factorisation locally (deterministic kernels -> identical factors, no broadcast).
The only collective is an all-gather of one 24-byte record per rank - (best value, lowest global
index, NaN count) - reduced identically on every rank: max value, ties to the LOWEST global index
(the reference's np.argwhere(...)[0] rule, /root/reference/point_selector.py:207).
RCCL has no MAXLOC; an all-gather + local lexicographic reduce is deterministic and 24 B/rank.
"""
from __future__ import annotations

import struct
from typing import Tuple


def shard_bounds(M: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block partition: the first M % world ranks get one extra candidate."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, extra = divmod(M, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def reduce_records(records):
    """records: iterable of (best_val, best_idx, nan_count).  Pure-Python lexicographic reduce."""
    best_val, best_idx, nan_total = float("-inf"), 2 ** 63 - 1, 0
    for v, i, n in records:
        nan_total += int(n)
        if v != v:  # a NaN best value never wins; it is accounted for through nan_count
            continue
        if v > best_val or (v == best_val and i < best_idx):
            best_val, best_idx = v, int(i)
    return best_val, best_idx, nan_total


def allreduce_argmax(best_val: float, best_idx: int, nan_count: int, group=None, device=None,
                     force_collective: bool = False):
    """All ranks obtain the global (max value, lowest index, total NaN count).

    One all_gather of 3 x int64 per rank (the fp64 value travels as its bit pattern, so no rounding
    and no NaN canonicalisation can happen in transit).  Works on nccl (= RCCL, device tensors)
    and gloo (CPU tensors).  force_collective: run the all-gather even in a one-rank group (used to exercise
    the RCCL path on a single GPU)."""
    import torch
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized() or (dist.get_world_size(group) == 1
                                                                and not force_collective):
        return reduce_records([(best_val, best_idx, nan_count)])
    world = dist.get_world_size(group)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    bits = struct.unpack("<q", struct.pack("<d", float(best_val)))[0]
    mine = torch.tensor([bits, int(best_idx), int(nan_count)], dtype=torch.int64, device=device)
    out = torch.empty(world * 3, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(out, mine, group=group)
    rows = out.cpu().view(world, 3).tolist()
    recs = [(struct.unpack("<d", struct.pack("<q", b))[0], i, n) for b, i, n in rows]
    return reduce_records(recs)


def allreduce_status(status, group=None, force_collective: bool = False):
    """The exchange step straight from the device record: `status` is DeviceGP's int64[5] buffer (gpbo_result:
    value bits, index, NaN count, reserved; then the factorisation's info word).  On nccl (= RCCL) the record is
    gathered from device memory as it is - no host round trip before the collective - and read back once.
    Returns (best value, lowest index, NaN total, first non-zero info over the ranks or 0)."""
    import torch
    import torch.distributed as dist

    def parse(rows):
        recs, info = [], 0
        for b, i, n, _, w in rows:
            recs.append((struct.unpack("<d", struct.pack("<q", b))[0], i, n))
            w32 = w & 0xFFFFFFFF
            if info == 0 and w32:
                info = w32 - (1 << 32) if w32 >= (1 << 31) else w32
        return reduce_records(recs) + (info,)

    if not dist.is_available() or not dist.is_initialized() or (dist.get_world_size(group) == 1
                                                                and not force_collective):
        return parse([status.cpu().tolist()])
    world = dist.get_world_size(group)
    src = status if dist.get_backend(group) == "nccl" else status.cpu()
    out = torch.empty(world * 5, dtype=torch.int64, device=src.device)
    dist.all_gather_into_tensor(out, src.contiguous(), group=group)
    return parse(out.cpu().view(world, 5).tolist())


def _collective_device(group=None):
    """Where a tensor must live to travel through `group`: the current GPU on nccl (= RCCL), the host on gloo."""
    import torch
    import torch.distributed as dist

    if dist.get_backend(group) == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def gather_concat(local, total: int, group=None):
    """Concatenation, in rank order, of the contiguous shards produced under shard_bounds (the ARD likelihood grid: float32
    cells, sharded along axis 0).  Every rank gets the full host array; bytes travel unchanged, as ONE tensor collective
    (all_gather_into_tensor of a padded block per rank: device memory on RCCL, no pickling, no second collective for sizes -
    every rank knows every shard's length from shard_bounds and checks it).  A no-op outside a process group."""
    import numpy as np
    import torch
    import torch.distributed as dist

    local = np.ascontiguousarray(local)
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        if local.shape[0] != total:
            raise ValueError("gather_concat: shard does not cover the whole array")
        return local
    world = dist.get_world_size(group)
    row = int(np.prod(local.shape[1:], dtype=np.int64)) * local.itemsize   # bytes per row
    pad = -(-total // world) * row                                        # the longest shard, in bytes
    # The block of a rank: its row count (8 bytes), then its rows as raw bytes - any dtype travels unchanged, and a rank
    # whose shard is not the one shard_bounds assigns is seen by EVERY rank after the collective (all raise together; a
    # rank that raised before it would leave the others waiting inside it).
    mine = np.zeros(8 + pad, dtype=np.uint8)
    mine[:8] = np.frombuffer(np.int64(local.shape[0]).tobytes(), dtype=np.uint8)
    raw = local.reshape(-1).view(np.uint8)[:pad]
    mine[8: 8 + raw.size] = raw
    dev = _collective_device(group)
    flat = torch.empty(world * (8 + pad), dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(flat, torch.from_numpy(mine).to(dev), group=group)
    flat = flat.cpu().numpy().reshape(world, 8 + pad)
    parts = []
    for r in range(world):
        a, b = shard_bounds(total, world, r)
        rows = int(flat[r, :8].copy().view(np.int64)[0])
        if rows != b - a:
            raise ValueError(f"gather_concat: rank {r} holds {rows} rows, shard_bounds gives {b - a} of {total}")
        parts.append(flat[r, 8: 8 + (b - a) * row])
    return np.concatenate(parts).view(local.dtype).reshape((total,) + local.shape[1:])


def gather_concat_tensors(locals_, total: int, group=None):
    """The dense outputs of a sharded call (mean_func / cov_func / acq_func_eval, which
    /root/reference/select_parameters.py:167-169,303-305 reads) gathered WITHOUT leaving the device: `locals_` is a list of
    k one-dimensional tensors of this rank's shard (same length, same dtype, the shard shard_bounds gives this rank);
    they travel as one [k x padded] block per rank through all_gather_into_tensor (RCCL on nccl: device memory to
    device memory; gloo: CPU tensors) and every rank gets k tensors of `total` rows, bytes unchanged.  Round 2 pickled
    the host copies through the object collective: 3 x 134 MB per call at BASELINE config 3's M = 2^24."""
    import torch
    import torch.distributed as dist

    locals_ = [t.contiguous() for t in locals_]
    n = int(locals_[0].shape[0])
    if any(t.dim() != 1 or int(t.shape[0]) != n or t.dtype != locals_[0].dtype for t in locals_):
        raise ValueError("gather_concat_tensors: the shards must be one-dimensional, of one length and one dtype")
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        if n != total:
            raise ValueError("gather_concat_tensors: shard does not cover the whole array")
        return locals_
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds(total, world, rank)
    if hi - lo != n:
        raise ValueError(f"gather_concat_tensors: rank {rank} holds {n} rows, shard_bounds gives {hi - lo}")
    pad = -(-total // world)  # the longest shard
    on_device = dist.get_backend(group) == "nccl"
    k = len(locals_)
    dev = locals_[0].device if on_device else torch.device("cpu")
    mine = torch.zeros((k, pad), dtype=locals_[0].dtype, device=dev)
    for j, t in enumerate(locals_):
        mine[j, :n] = t if on_device else t.cpu()
    flat = torch.empty(world * k * pad, dtype=mine.dtype, device=dev)   # (flat in, flat out: what every backend accepts)
    dist.all_gather_into_tensor(flat, mine.view(-1), group=group)
    out = flat.view(world, k, pad)
    full = []
    for j in range(k):
        parts = []
        for r in range(world):
            a, b = shard_bounds(total, world, r)
            parts.append(out[r, j, : b - a])
        full.append(torch.cat(parts))
    return full


def all_agree(flag: bool, group=None) -> bool:
    """True iff `flag` is true on EVERY rank (a collective decision: e.g. append-or-refactorise must be taken the same
    way everywhere, or the ranks' factors differ at rounding level and the lowest-index tie rule no longer holds).
    One int32 all_reduce(MIN).  Outside a process group: the flag itself."""
    import torch
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return bool(flag)
    vote = torch.tensor([1 if flag else 0], dtype=torch.int32, device=_collective_device(group))
    dist.all_reduce(vote, op=dist.ReduceOp.MIN, group=group)
    return bool(int(vote.item()))
