"""world_size-2 gloo test of the candidate sharding + all-gather + global first-minimum argmin
(the N > 1 path of bench.py / distributed.py).  CPU only: the scorer is the oracle."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, seed, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from morphablegraphs_amd import synthetic
    import framework_collective_helpers as distributed      # the torch.distributed carriers of the exchange live with the tests
    from oracle import c_oracle
    data = synthetic.make_primitive(seed=4, n_components=8, n_frames=30, n_dim=11, n_gmm=2)
    cp = c_oracle.COraclePrimitive(data)
    rng = np.random.default_rng(seed)
    S = rng.standard_normal((n, 8))
    S[n // 2] = S[3]                      # a tie: the FIRST of the two must win
    nan = np.nan
    cons = np.array([[0, 29.0, 1.0, 10.0, nan, -20.0, 0, 0], [1, 29.0, 2.0, 0.3, -1.0, 0.0, 0.0, 1.0]])
    idx, val, scores = distributed.sharded_best_candidate(S, lambda blk: cp.keyframe_errors_f64(blk, cons))
    np.save(os.path.join(out_dir, "r%d.npy" % rank), np.array([idx, val]))
    idx2, val2, none = distributed.sharded_best_candidate(S, lambda blk: cp.keyframe_errors_f64(blk, cons), exchange="minloc")
    assert none is None
    np.save(os.path.join(out_dir, "m%d.npy" % rank), np.array([idx2, val2]))
    if rank == 0:
        np.save(os.path.join(out_dir, "scores.npy"), scores.numpy())
        ref = cp.keyframe_errors_f64(S, cons)
        np.save(os.path.join(out_dir, "ref.npy"), ref)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_argmin_equals_single_process(tmp_path):
    for n in (37, 64):                    # ragged and even splits
        d = tmp_path / ("n%d" % n)
        d.mkdir()
        port = _free_port()
        mp.spawn(_worker, args=(2, port, n, 7, str(d)), nprocs=2, join=True)
        r0, r1 = np.load(d / "r0.npy"), np.load(d / "r1.npy")
        ref, scores = np.load(d / "ref.npy"), np.load(d / "scores.npy")
        np.testing.assert_array_equal(r0, r1)                       # every rank agrees
        np.testing.assert_allclose(scores, ref, rtol=0, atol=0)      # gathered in global order
        best = 0
        for i, e in enumerate(ref):                                  # the reference's loop
            if ref[best] > e:
                best = i
        assert int(r0[0]) == best and r0[1] == ref[best]
        m0, m1 = np.load(d / "m0.npy"), np.load(d / "m1.npy")       # the 16-byte (index, value) exchange: same winner
        np.testing.assert_array_equal(m0, r0)
        np.testing.assert_array_equal(m1, r0)
