"""world_size-2 (and 3) gloo tests of the PRODUCT's sharded seam (morphablegraphs_amd/distributed.py): the commands rank 0 broadcasts,
the ranks' blocks, the one all-gather of {error, global index, winning latent} and the first-minimum rule run through
distributed.run_command / distributed.COMMANDS -- the code a node of MI355X runs over RCCL -- with the two calls a communicator
consists of (broadcast_bytes, all_gather_rows) carried by torch.distributed's gloo backend, injected from here (the package itself
carries no tensor framework: its carriers are MgCommunicator = RCCL through the C-ABI, and files).  CPU only: the scorers are the
oracle's (tests/test_sharded_seam.py's stand-ins for the three HIP legs)."""
import os
import pickle
import socket
import sys

import numpy as np
import pytest
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


class GlooCommunicator(object):
    """The communicator interface of morphablegraphs_amd.distributed (rank, world, broadcast_bytes, all_gather_rows) over an
    initialised torch.distributed process group."""

    def __init__(self):
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def broadcast_bytes(self, payload, root=0):
        n = torch.tensor([len(payload) if self.rank == root else 0], dtype=torch.int64)
        dist.broadcast(n, src=root)
        buf = torch.frombuffer(bytearray(payload), dtype=torch.uint8) if self.rank == root else torch.empty((int(n.item()),), dtype=torch.uint8)
        if int(n.item()):
            dist.broadcast(buf, src=root)
        return payload if self.rank == root else buf.numpy().tobytes()

    def all_gather_rows(self, row):
        t = torch.as_tensor(np.ascontiguousarray(row, dtype=np.float64))
        out = torch.empty((self.world * t.numel(),), dtype=torch.float64)
        dist.all_gather_into_tensor(out, t)
        return out.numpy().reshape(self.world, t.numel())


def _rank(rank, world, port, n, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from morphablegraphs_amd import distributed
    import test_sharded_seam as seam
    comm = GlooCommunicator()
    nodes = {"p": seam._model()}
    hooks = {"evaluate_samples": {"scorer": seam.cpu_scorer}, "sample_and_evaluate": {"sampler": seam.cpu_sampler}, "options_step": {"stepper": seam.cpu_stepper}}
    if rank == 0:
        res = [distributed.run_command(comm, nodes, cmd, **hooks[cmd["op"]]) for cmd in seam._commands(n)]
        distributed.stop_workers(comm)
        with open(os.path.join(out_dir, "r0.pkl"), "wb") as f:
            pickle.dump(res, f)
    else:
        served = 0
        while True:
            cmd = distributed.decode_command(comm.broadcast_bytes(b"", 0))
            if cmd["op"] == "stop":
                break
            distributed.COMMANDS[cmd["op"]](comm, nodes, cmd, **hooks[cmd["op"]])
            served += 1
        with open(os.path.join(out_dir, "served%d" % rank), "w") as f:
            f.write(str(served))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 64), (2, 37), (3, 50)])
def test_product_seam_over_gloo_equals_the_single_process(tmp_path, world, n):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from morphablegraphs_amd import distributed
    import test_sharded_seam as seam
    mp.spawn(_rank, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    with open(tmp_path / "r0.pkl", "rb") as f:
        sharded = pickle.load(f)
    for r in range(1, world):
        assert (tmp_path / ("served%d" % r)).read_text() == "3"
    local = distributed.LocalCommunicator()
    nodes = {"p": seam._model()}
    cmds = seam._commands(n)
    one = [distributed._cmd_evaluate_samples(local, nodes, cmds[0], scorer=seam.cpu_scorer),
           distributed._cmd_sample_and_evaluate(local, nodes, cmds[1], sampler=seam.cpu_sampler),
           distributed._cmd_options_step(local, nodes, cmds[2], stepper=seam.cpu_stepper)]
    for k in (0, 1):
        assert sharded[k][0] == one[k][0] and sharded[k][1] == one[k][1]
        np.testing.assert_array_equal(sharded[k][2], one[k][2])
    for o in "abc":
        assert sharded[2][o][1] == one[2][o][1] and sharded[2][o][2] == one[2][o][2]
        np.testing.assert_array_equal(sharded[2][o][0], one[2][o][0])
    # the tie the commands plant (the minimum twice, in different ranks' blocks): the FIRST of the two won
    S = cmds[0]["samples"]
    err = seam._oracle_errors(seam._model(), S, seam.CONS)
    assert sharded[0][0] == int(np.flatnonzero(err == err.min())[0])
