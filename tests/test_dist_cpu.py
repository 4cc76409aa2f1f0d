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

    stats = {}
    s1, idx, s2 = zdist.sharded_cascade(n, stage_logits, rank, world, thr, None, stats=stats)
    # two gathers when something passed the gate, one otherwise; their wall time is what tools/run_configs.py reports per rank
    assert stats["gathers"] == (2 if len(idx) else 1) and stats["gather_s"] > 0.0
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


# ---- patient-level sharding of the batch driver (config 5: run_batch_simple_2stage.py equivalent on N ranks) ----
def _batch_worker(rank, world, port, root, out_dir, patients, q):
    import json
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from zkast import batch, pipeline as pl

    def gather_bytes(b):                       # stand-in for Context.allgather_bytes (RCCL) on the CPU
        box = [None] * world
        dist.all_gather_object(box, b)
        return box

    def fake_run_patient(files, *_a):          # the per-patient cascade itself is covered by the GPU suite
        n = sum(os.path.getsize(f) for f in files)
        return {"aggregate": {"files_used": files, "total_windows": n, "overall_zenker_ratio_over_swallow": (n % 7) / 7.0}}

    pl_run, pl.run_patient = pl.run_patient, fake_run_patient
    try:
        summ = {}
        st = batch.run_batch(patients, root, None, None, None, None, out_dir, rank=rank, world=world,
                             gather_bytes=gather_bytes, summaries=summ, log=lambda *_: None)
    finally:
        pl.run_patient = pl_run
    q.put((rank, st, summ))
    dist.barrier()
    dist.destroy_process_group()


def test_patient_sharded_batch_two_ranks(tmp_path):
    import json
    from zkast import batch, pipeline as pl
    patients = [f"{i:03d}" for i in range(7)] + ["missing"]
    for i, pid in enumerate(patients[:-1]):
        d = tmp_path / "Long" / ("Zenker" if i % 2 else "Healthy") / pid
        d.mkdir(parents=True)
        for k in range(2):
            pl.write_wav_pcm16(str(d / f"r{k}.wav"), np.zeros(100 + 10 * i + k, np.float32), 16000)
    assert batch.shard_patients(patients, 0, 2) == patients[0::2] and batch.shard_patients(patients, 1, 2) == patients[1::2]
    out = tmp_path / "out"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_batch_worker, args=(r, 2, port, str(tmp_path), str(out), patients, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    exp_status = {pid: "ok" for pid in patients[:-1]}
    exp_status["missing"] = "error"
    for rank, st, summ in res:                 # every rank ends with the full table, in list order
        assert st == exp_status and list(st) == patients
        assert set(summ) == set(patients[:-1])
    # every patient JSON was written exactly once, by the rank that owns the patient
    files = sorted(os.listdir(out))
    assert files == sorted(f"{pid}_2stage.json" for pid in patients[:-1])
    # the gathered aggregate blocks are what the patient-level aggregation reads from the files
    from zkast import aggregate as agg
    summary, rows = agg.aggregate(str(out), 0.5)
    assert summary["num_patient_results"] == 7
    assert {r["patient_id"]: r["ratio"] for r in rows} == {pid: res[0][2][pid]["overall_zenker_ratio_over_swallow"] for pid in patients[:-1]}


# ---- world-size-8 rehearsal of the exact configs[3] / configs[4] partitions (no 8-GPU node was ever available to the
#      builder: this is what stands in for the hardware run of zkast.dist / zkast.batch at N = 8) ----
def _clustered_logits(stage, win_idx):
    """stage 1: swallows cluster in time (three bursts of the 3 599-window recording; K = 517, not divisible by 8, and five
    of the eight stage-1 shards contain no swallow at all); stage 2: the deterministic stand-in."""
    w = np.asarray(win_idx, dtype=np.int64)
    if stage == 0:
        sw = ((w >= 100) & (w < 331)) | ((w >= 1700) & (w < 1903)) | ((w >= 3500) & (w < 3583))
        margin = np.where(sw, 2.0 + 0.001 * (w % 97), -2.0 - 0.001 * (w % 89))
        return np.stack([-margin / 2, margin / 2], 1).astype(np.float32)
    return _fake_logits(stage, win_idx)


def _worker8(rank, world, port, n, thr, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []

    def stage_logits(stage, idx):
        calls.append((stage, np.asarray(idx).copy()))
        return _clustered_logits(stage, idx)

    s1, idx, s2 = zdist.sharded_cascade(n, stage_logits, rank, world, thr, None)
    q.put((rank, s1, idx, s2, [(s, i.tolist()) for s, i in calls]))
    dist.barrier()
    dist.destroy_process_group()


def test_world8_cascade_of_the_30_minute_recording_with_a_clustered_gate():
    """configs[3]: N = 3 599 windows over 8 ranks (shards of 450 / 449), gate clustered in time."""
    n, thr, world = 3599, 0.5, 8
    ref1, refi, ref2 = zdist.sharded_cascade(n, _clustered_logits, 0, 1, thr, None)
    K = len(refi)
    assert K == 517 and K % world != 0
    shards = [zdist.shard_range(n, r, world) for r in range(world)]
    assert [hi - lo for lo, hi in shards] == [450] * 7 + [449] and shards[0][0] == 0 and shards[-1][1] == n
    empty = [r for r, (lo, hi) in enumerate(shards) if not ((refi >= lo) & (refi < hi)).any()]
    assert len(empty) >= 4                                    # re-using the stage-1 partition would idle these ranks
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, n, thr, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, s1, idx, s2, calls in res:                      # every rank ends with the single-process result
        assert np.array_equal(s1, ref1) and np.array_equal(idx, refi) and np.array_equal(s2, ref2)
    st1 = [ids for _, _, _, _, calls in res for s, ids in calls if s == 0]
    st2 = [ids for _, _, _, _, calls in res for s, ids in calls if s == 1]
    assert sorted(i for ids in st1 for i in ids) == list(range(n))
    assert [len(ids) for ids in st1] == [450] * 7 + [449]
    assert sorted(i for ids in st2 for i in ids) == refi.tolist()
    sizes = [len(ids) for ids in st2]
    assert len(sizes) == world and max(sizes) - min(sizes) <= 1 and sum(sizes) == K      # 65 x 5 + 64 x 3: balanced stage 2


def test_world8_patient_sharded_batch_of_64_patients(tmp_path):
    """configs[4]: 64 patients x 2 files over 8 ranks (8 patients each), every rank ends with the full status table and
    the gathered aggregate blocks, every patient JSON is written exactly once."""
    from zkast import aggregate as agg
    from zkast import batch, pipeline as pl
    world = 8
    patients = [f"{i:03d}" for i in range(64)]
    for i, pid in enumerate(patients):
        d = tmp_path / "Long" / ("Zenker" if i % 3 == 0 else "Healthy") / pid
        d.mkdir(parents=True)
        for k in range(2):
            pl.write_wav_pcm16(str(d / f"r{k}.wav"), np.zeros(50 + 3 * i + k, np.float32), 16000)
    owned = [batch.shard_patients(patients, r, world) for r in range(world)]
    assert all(len(o) == 8 for o in owned) and sorted(p for o in owned for p in o) == patients
    out = tmp_path / "out"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_batch_worker, args=(r, world, port, str(tmp_path), str(out), patients, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    # the single-process run of the same driver is the reference
    ref_out = tmp_path / "ref"

    def fake_run_patient(files, *_a):
        n = sum(os.path.getsize(f) for f in files)
        return {"aggregate": {"files_used": files, "total_windows": n, "overall_zenker_ratio_over_swallow": (n % 7) / 7.0}}

    pl_run, pl.run_patient = pl.run_patient, fake_run_patient
    try:
        ref_summ = {}
        ref_st = batch.run_batch(patients, str(tmp_path), None, None, None, None, str(ref_out), summaries=ref_summ,
                                 log=lambda *_: None)
    finally:
        pl.run_patient = pl_run
    for rank, st, summ in res:
        assert st == ref_st and list(st) == patients
        assert summ == ref_summ
    assert sorted(os.listdir(out)) == sorted(os.listdir(ref_out)) == sorted(f"{pid}_2stage.json" for pid in patients)
    s8, rows8 = agg.aggregate(str(out), 0.5)
    s1, rows1 = agg.aggregate(str(ref_out), 0.5)
    s8.pop("outputs_dir"), s1.pop("outputs_dir")
    for r in rows8 + rows1:
        r.pop("json_path")
    assert s8 == s1 and rows8 == rows1 and s8["num_patient_results"] == 64
