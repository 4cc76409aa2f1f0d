"""Two ranks (gloo rendezvous, both on the one GPU of the test box) run the window-sharded cascade with the real HIP
kernels; every rank must end with exactly the single-process result.  On the 8-GPU node the same code runs with
backend nccl (= RCCL over xGMI) and one GPU per rank (bench.py --gpus N)."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _models():
    from zkast import ZkASTConfig, ZkASTFeatureExtractor, ZkASTForAudioClassification, synth
    cas = np.load(os.path.join(os.path.dirname(__file__), "golden", "cascade.npz"))
    out = []
    for stage, (seed, shift, mean, std) in enumerate([
            (int(cas["s1_seed"]), float(cas["s1_bias_shift"]), float(cas["s1_mean"]), float(cas["s1_std"])),
            (int(cas["s2_seed"]), float(cas["s2_bias_shift"]), float(cas["s2_mean"]), float(cas["s2_std"]))]):
        sd = synth.make_ast_weights(seed, "wide")
        sd["classifier.dense.bias"][1] += np.float32(shift)
        m = ZkASTForAudioClassification(ZkASTConfig(num_labels=2), sd, stage=stage, fx_mean=mean, fx_std=std)
        out.append((m, ZkASTFeatureExtractor(mean=mean, std=std)))
    return out


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from zkast import dist as zdist
    from zkast import synth
    (m1, fx1), (m2, fx2) = _models()
    rec = synth.synth_recording(3, 16000 + 15 * 8000)
    s1, idx, s2 = zdist.ZkShardedCascade(m1, fx1, m2, fx2, rank, world)(rec, 1.0, 0.5, 0.5)
    q.put((rank, s1, idx, s2))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_cascade_matches_single_process():
    from zkast import dist as zdist
    from zkast import synth
    (m1, fx1), (m2, fx2) = _models()
    rec = synth.synth_recording(3, 16000 + 15 * 8000)
    ref1, refi, ref2 = zdist.ZkShardedCascade(m1, fx1, m2, fx2, 0, 1)(rec, 1.0, 0.5, 0.5)
    assert 0 < len(refi) < 16
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(2)]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, s1, idx, s2 in res:
        # per-window arithmetic does not depend on which windows share a micro-batch -> bit-identical
        assert np.array_equal(s1, ref1) and np.array_equal(idx, refi) and np.array_equal(s2, ref2)
