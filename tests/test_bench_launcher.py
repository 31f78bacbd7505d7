"""bench.py's N > 1 plumbing without a GPU (--dry-run): typed as `python bench.py --gpus 2` it must launch its own
ranks and print exactly one JSON line; launched the driver's way (torch.distributed.run) it must read the ranks from
the environment.  bench.py carries no tensor framework (a test keeps it so): the dry run's exchange goes through the file rendezvous."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


def _run(cmd, env=None, timeout=240):
    e = dict(os.environ)
    e.update(env or {})
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    p = subprocess.run(cmd, cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    return p.stdout.decode()


def test_bench_carries_no_tensor_framework():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "import torch" not in src and "torch.cuda" not in src and "torch.distributed." not in src.replace("torch.distributed.run", "")


def test_gpus_n_typed_as_is_launches_its_own_ranks():
    out = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "5", "--warmup", "1", "--dry-run", "--batch", "64"])
    lines = _json_lines(out)
    assert len(lines) == 1, out
    r = lines[0]
    assert r["n_gpus"] == 2 and r["steps"] == 5 and r["warmup"] == 1 and r["dry_run"] is True
    assert r["config"]["global_candidates"] == 128 and r["scaling"] == "weak" and r["unit"] == "samples/s"
    assert r["value"] > 0 and abs(r["ms_per_step"] * r["value"] / 1e3 - 128) < 1e-6


def test_the_drivers_launch_form_reads_the_ranks_from_the_environment():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "4", "--warmup", "1", "--dry-run", "--batch", "32"])
    lines = _json_lines(out)
    assert len(lines) == 1, out
    assert lines[0]["n_gpus"] == 2 and lines[0]["config"]["global_candidates"] == 64


def test_single_rank_dry_run_and_world_size_mismatch():
    lines = _json_lines(_run([sys.executable, "bench.py", "--steps", "3", "--warmup", "0", "--dry-run"]))
    assert len(lines) == 1 and lines[0]["n_gpus"] == 1
    e = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "4", "--dry-run"], cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode != 0 and b"does not match WORLD_SIZE" in p.stderr


def test_a_dead_rank_ends_the_launch_instead_of_hanging_it():
    """rank 1 dies before the rendezvous; the launcher ends rank 0 (which would wait 300 s for it), removes the
    rendezvous files and returns non-zero within seconds."""
    import time
    e = dict(os.environ, MG_BENCH_DRY_RUN_DIES="1")
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    t0 = time.time()
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "0", "--dry-run", "--batch", "32"], cwd=ROOT, env=e,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode != 0 and time.time() - t0 < 60
    assert b"rank 1 exited with 3" in p.stderr


@pytest.mark.gpu
def test_collective_set_up_failure_ends_the_run_with_the_reason():
    """Two ranks on ONE GPU: RCCL refuses the second rank of a device, so mg_dist_init fails -- on both ranks; they tell each other
    through the file rendezvous and the run ends with the reason (there is no other carrier of the collective in bench.py).  What an
    8-GPU node would do if the library's own communicator could not be set up there."""
    base = [sys.executable, "bench.py", "--gpus", "2", "--steps", "30", "--warmup", "3", "--ramp-steps", "30", "--no-cpu-baseline",
            "--no-extra-configs", "--no-placement-compare"]
    env = dict(os.environ, MG_BENCH_OVERSUBSCRIBE="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    p = subprocess.run(base, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=400)
    assert p.returncode != 0 and b"could not be set up" in p.stderr and not _json_lines(p.stdout.decode())


def test_ranks_that_never_join_the_communicator_end_the_launch(tmp_path):
    """the launcher bounds the communicator's set-up on its own -- ranks that have not reported 'joined' within
    MG_BENCH_RDV_TIMEOUT seconds are ended and the launch returns non-zero (here: a stand-in rank that sleeps in its set-up)."""
    import time
    e = dict(os.environ, MG_BENCH_RDV_TIMEOUT="3", MG_BENCH_STALL_IN_SETUP="1")
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    t0 = time.time()
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "0", "--batch", "32", "--no-cpu-baseline"], cwd=ROOT, env=e,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode != 0 and time.time() - t0 < 60, p.stderr.decode()[-500:]
    assert b"had not joined the communicator" in p.stderr
