"""All-gather cost per step against RCCL's channel count and CUs left free (world of one rank, one GPU).
usage: NCCL_MAX_NCHANNELS=.. python3 tools/probes/dist_overhead3.py"""
import os, sys, time
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from morphablegraphs_amd import _capi, synthetic
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29579")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
ctx = _capi.Context(0, stream=stream.cuda_stream)
prim = _capi.Primitive(ctx, synthetic.make_walk_primitive(seed=0))
B, L = 8192, 40
S = torch.randn(B, L, device=dev)
frames = torch.empty(B, 156, 79, device=dev)
lps = [torch.empty(B, device=dev) for _ in range(2)]
gs = [torch.empty(B, device=dev) for _ in range(2)]
works = [None, None]
def body(i, gather):
    b = i & 1
    if works[b] is not None: works[b].wait(); works[b] = None
    prim.step_frames_and_logp_dev(S.data_ptr(), np.float32, B, L, frames.data_ptr(), lps[b].data_ptr())
    if gather: works[b] = dist.all_gather_into_tensor(gs[b], lps[b], async_op=True)
def run(gather, n=2000):
    for i in range(200): body(i, gather)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): body(i, gather)
    for b in range(2):
        if works[b] is not None: works[b].wait(); works[b] = None
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / n
tag = "channels=%s proto=%s" % (os.environ.get("NCCL_MAX_NCHANNELS", "default"), os.environ.get("NCCL_PROTO", "default"))
for r in (0, 8):
    ctx.set_reserved_cus(r)
    print("%s reserved=%d  no gather %.1f  gather %.1f us/step" % (tag, r, run(False), run(True)), flush=True)
dist.destroy_process_group()
