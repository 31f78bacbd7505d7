"""Per-step cost of the RCCL all-gather next to the persistent kernel, on one GPU with a world of one rank:\nwith and without CUs reserved for it (mg_context_set_reserved_cus)."""
import os, sys, time
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from morphablegraphs_amd import _capi, synthetic
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
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
def step(i, gather):
    b = i & 1
    if works[b] is not None: works[b].wait(); works[b] = None
    prim.step_frames_and_logp_dev(S.data_ptr(), np.float32, B, L, frames.data_ptr(), lps[b].data_ptr())
    if gather: works[b] = dist.all_gather_into_tensor(gs[b], lps[b], async_op=True)
for gather, reserved in ((False, 0), (True, 0), (True, 4), (True, 8), (True, 16), (False, 8)):
    ctx.set_reserved_cus(reserved)
    for i in range(200): step(i, gather)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(2000): step(i, gather)
    t_issue = time.perf_counter() - t0
    for w in works:
        if w is not None: w.wait()
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    print("gather=%s reserved_cus=%d: %.1f us/step (host issue time %.1f us/step)" % (gather, reserved, 1e6*t/2000, 1e6*t_issue/2000))
dist.destroy_process_group()
