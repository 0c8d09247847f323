"""The N>1 path on CPU: contiguous candidate sharding + the (value, lowest-index, NaN-count)
exchange, exercised with a real 2-process gloo group (the GPU path uses the same code over RCCL)."""
import os
import socket

import numpy as np
import pytest

from bayesian_optimisation_amd import distributed as D


def test_shard_bounds_cover_exactly_once():
    for M in [1, 7, 8, 1000, (1 << 24) + 3]:
        for world in [1, 2, 3, 8]:
            spans = [D.shard_bounds(M, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == M
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_reduce_records_tie_and_nan_rules():
    inf = float("inf")
    assert D.reduce_records([(1.0, 5, 0), (2.0, 9, 0), (2.0, 3, 0)]) == (2.0, 3, 0)
    assert D.reduce_records([(2.0, 9, 1), (-inf, 2 ** 63 - 1, 4)]) == (2.0, 9, 5)
    assert D.reduce_records([(float("nan"), 1, 2), (0.5, 7, 0)]) == (0.5, 7, 2)
    assert D.reduce_records([(-0.0, 4, 0), (0.0, 2, 0)]) == (0.0, 2, 0)  # -0.0 == 0.0: lowest index wins


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # a synthetic acquisition surface with an exact tie across the two shards
        M = 1001
        acq = np.cos(np.arange(M) * 0.37)
        acq[[100, 900]] = 5.0  # same max on rank 0 and rank 1 -> index 100 must win everywhere
        lo, hi = D.shard_bounds(M, world, rank)
        loc = acq[lo:hi]
        j = int(np.flatnonzero(loc == loc.max())[0])
        out = D.allreduce_argmax(float(loc[j]), lo + j, 1 if rank == 1 else 0)
        # the bit pattern of the value must survive the exchange
        out2 = D.allreduce_argmax(float(np.nextafter(1.0, 2.0)) if rank == 0 else 1.0, rank, 0)
        # dense arrays / the ARD grid: contiguous shards back into one array, bytes unchanged, dtype kept
        full = (np.arange(1001, dtype=np.float32) * np.float32(0.1)) ** 2
        got = D.gather_concat(full[lo:hi], M)
        ok_gather = got.dtype == np.float32 and np.array_equal(got, full)
        try:
            D.gather_concat(full[lo:hi], M + 1)
            ok_gather = False
        except ValueError:
            pass
        # the device-record form of the exchange (CPU tensors on gloo): rank 1 carries a failed factorisation
        import struct

        import torch

        # the tensor form of the dense gather (what the sharded drop-in class uses: one collective for mu / sigma / acq,
        # uneven shards - 1001 rows over 2 ranks - padded in transit and cut back, bit patterns unchanged)
        t3 = [torch.from_numpy(np.sin(np.arange(M) * c)[lo:hi].copy()) for c in (0.11, 0.23, 0.31)]
        g3 = D.gather_concat_tensors(t3, M)
        ok_gather = ok_gather and all(torch.equal(g, torch.from_numpy(np.sin(np.arange(M) * c)))
                                      for g, c in zip(g3, (0.11, 0.23, 0.31)))
        try:
            D.gather_concat_tensors([t3[0][:-1]], M)
            ok_gather = False
        except ValueError:
            pass

        bits = struct.unpack("<q", struct.pack("<d", float(loc[j])))[0]
        status = torch.tensor([bits, lo + j, 0, 0, 0 if rank == 0 else 17], dtype=torch.int64)
        ok_gather = ok_gather and D.allreduce_status(status) == (5.0, 100, 0, 17)
        q.put((rank, out, out2, ok_gather))
    finally:
        dist.destroy_process_group()


def test_two_process_gloo_exchange():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, out, out2, ok_gather in res:
        assert ok_gather
        assert out == (5.0, 100, 1)
        assert out2 == (float(np.nextafter(1.0, 2.0)), 0, 0)


def test_gather_concat_tensors_without_a_process_group():
    import torch

    a, b = torch.arange(7.0, dtype=torch.float64), torch.ones(7, dtype=torch.float64)
    out = D.gather_concat_tensors([a, b], 7)
    assert torch.equal(out[0], a) and torch.equal(out[1], b)
    with pytest.raises(ValueError):
        D.gather_concat_tensors([a, b], 8)
    with pytest.raises(ValueError):
        D.gather_concat_tensors([a, b[:5]], 7)


def test_gather_concat_without_a_process_group():
    a = np.arange(7.0)
    assert D.gather_concat(a, 7) is not None and np.array_equal(D.gather_concat(a, 7), a)
    with pytest.raises(ValueError):
        D.gather_concat(a, 8)


def test_allreduce_status_without_a_process_group():
    import struct

    import torch

    bits = struct.unpack("<q", struct.pack("<d", -2.5))[0]
    assert D.allreduce_status(torch.tensor([bits, 42, 3, 0, 0], dtype=torch.int64)) == (-2.5, 42, 3, 0)
    neg = (-7) & 0xFFFFFFFF  # an int32 info word of -7 in the low half of the slot
    assert D.allreduce_status(torch.tensor([bits, 42, 0, 0, neg], dtype=torch.int64))[3] == -7


def test_bench_starts_its_own_ranks_when_asked_for_more_than_one_gpu():
    """`python bench.py --gpus 2` with no rendezvous in the environment (how the driver calls it) must run TWO ranks:
    the launcher check joins a gloo group and reports the ranks it saw; a WORLD_SIZE that contradicts --gpus is an
    error, never a silently smaller run."""
    import json
    import subprocess
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--rendezvous-only"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2
    env["WORLD_SIZE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--rendezvous-only"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "refusing" in r.stderr
