"""CPU, gloo, world_size 2: the window-sharded cascade (partition -> all-gather -> identical gate on every rank ->
re-partition -> all-gather) gives exactly the single-process result.  The per-window compute is a deterministic
stand-in (a 1-layer oracle would only add minutes); the GPU suite covers the real kernels."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp

from zkast import dist as zdist


def _fake_logits(stage, win_idx):
    w = np.asarray(win_idx, dtype=np.float64)
    a = np.sin(0.37 * w + stage) * 2.0
    b = np.cos(0.11 * w * (stage + 1)) * 2.0
    return np.stack([a, b], 1).astype(np.float32)


def _worker(rank, world, port, n, thr, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []

    def stage_logits(stage, idx):
        calls.append((stage, np.asarray(idx).copy()))
        return _fake_logits(stage, idx)

    s1, idx, s2 = zdist.sharded_cascade(n, stage_logits, rank, world, thr, None)
    q.put((rank, s1, idx, s2, [(s, i.tolist()) for s, i in calls]))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(n, thr, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, thr, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return sorted(res, key=lambda r: r[0])


def test_two_rank_cascade_equals_single_process():
    n, thr = 37, 0.5
    ref1, refi, ref2 = zdist.sharded_cascade(n, _fake_logits, 0, 1, thr, None)
    assert 0 < len(refi) < n
    res = _run(n, thr)
    for rank, s1, idx, s2, calls in res:
        assert np.array_equal(s1, ref1) and np.array_equal(idx, refi) and np.array_equal(s2, ref2)
    # every window was computed exactly once per stage across the ranks, in contiguous shards
    st1 = sorted(i for _, _, _, _, calls in res for s, ids in calls if s == 0 for i in ids)
    st2 = sorted(i for _, _, _, _, calls in res for s, ids in calls if s == 1 for i in ids)
    assert st1 == list(range(n)) and st2 == refi.tolist()
    sizes = [len(ids) for _, _, _, _, calls in res for s, ids in calls if s == 1]
    assert max(sizes) - min(sizes) <= 1


def test_ragged_and_empty_cases():
    for n, thr in [(1, 0.5), (3, 0.999999)]:      # fewer windows than ranks; nothing passes the gate
        ref1, refi, ref2 = zdist.sharded_cascade(n, _fake_logits, 0, 1, thr, None)
        for rank, s1, idx, s2, _ in _run(n, thr):
            assert np.array_equal(s1, ref1) and np.array_equal(idx, refi) and s2.shape == ref2.shape
