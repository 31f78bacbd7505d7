"""What exactly costs ~11 us per step when the scores are all-gathered (world of one rank, one GPU)?"""
import os, sys, time
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphablegraphs_amd import _capi, synthetic
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29578")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
side = torch.cuda.Stream(device=dev)
ctx = _capi.Context(0, stream=stream.cuda_stream)
prim = _capi.Primitive(ctx, synthetic.make_walk_primitive(seed=0))
B, L = 8192, 40
S = torch.randn(B, L, device=dev)
frames = torch.empty(B, 156, 79, device=dev)
lps = [torch.empty(B, device=dev) for _ in range(2)]
gs = [torch.empty(B, device=dev) for _ in range(2)]
evs = [torch.cuda.Event() for _ in range(4)]
works = [None, None]

def run(name, body, n=2000):
    for i in range(200): body(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n): body(i)
    th = time.perf_counter() - t0
    for w in works:
        if w is not None: w.wait()
    torch.cuda.synchronize()
    print("%-58s %.1f us/step   host enqueue %.1f us/step" % (name, 1e6 * (time.perf_counter() - t0) / n, 1e6 * th / n))

def kernel(i):
    prim.step_frames_and_logp_dev(S.data_ptr(), np.float32, B, L, frames.data_ptr(), lps[i & 1].data_ptr())
def k_only(i): kernel(i)
def k_event(i):
    kernel(i); evs[i & 3].record(stream)
def k_event_sidewait(i):
    kernel(i); evs[i & 3].record(stream); side.wait_event(evs[i & 3])
def k_side_copy(i):
    kernel(i); evs[i & 3].record(stream); side.wait_event(evs[i & 3])
    with torch.cuda.stream(side): gs[i & 1].copy_(lps[i & 1], non_blocking=True)
def k_side_copy_waitback(i):
    if i >= 2: stream.wait_event(evs[(i - 2) & 3 | 0])   # placeholder dependency back into our stream
    kernel(i); evs[i & 3].record(stream); side.wait_event(evs[i & 3])
    with torch.cuda.stream(side): gs[i & 1].copy_(lps[i & 1], non_blocking=True)
def k_gather_async(i):
    b = i & 1
    if works[b] is not None: works[b].wait(); works[b] = None
    kernel(i); works[b] = dist.all_gather_into_tensor(gs[b], lps[b], async_op=True)
def k_gather_async_nowait(i):
    kernel(i); dist.all_gather_into_tensor(gs[i & 1], lps[i & 1], async_op=True)
def k_gather_sync(i):
    kernel(i); dist.all_gather_into_tensor(gs[i & 1], lps[i & 1])
def k_same_stream_copy(i):
    kernel(i); gs[i & 1].copy_(lps[i & 1], non_blocking=True)

run("kernel only", k_only)
run("kernel + event record", k_event)
run("kernel + event record + side stream waits on it", k_event_sidewait)
run("kernel + 32 KB copy on a side stream", k_side_copy)
run("kernel + 32 KB copy on the same stream", k_same_stream_copy)
run("kernel + async all_gather, wait before buffer reuse", k_gather_async)
run("kernel + async all_gather, no wait", k_gather_async_nowait)
run("kernel + blocking all_gather", k_gather_sync)
run("kernel only (again)", k_only)
dist.destroy_process_group()
