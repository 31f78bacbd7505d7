#!/usr/bin/env python3
"""bench.py -- motion-primitive samples scored + back-projected per second on MI355X.

A "step" is one pass of the hot path over one batch of synthetic latent candidates already
resident in HBM: back-projection to full frames (156 x 79 float32 per candidate) plus the GMM
log-likelihood of the same candidates ('walk' primitive: L=40, F=156, K=8; BASELINE.json
configs[1] per GPU).  With N > 1 ranks the candidate batch is sharded 8192 per GPU (configs[3])
and every step ends with the RCCL all-gather of the scores.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config walk|graph|optimizer]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (the driver's form)

`--gpus N` with N > 1 and no WORLD_SIZE in the environment launches the N ranks itself (child
processes, before anything touches a GPU) and forwards rank 0's JSON line.  One JSON line on rank 0.
Everything on the device goes through libmg_hip.so (ctypes C-ABI): buffers, the stream, the kernels
and the collective (mg_dist_*: RCCL loaded by the library, the unique id handed over through a file).  No tensor
framework anywhere in this file: under `python -m torch.distributed.run` only the environment variables it sets are read.
`--config graph` (BASELINE configs[2]: 16 primitives x 4096 candidates per planner step) and
`--config optimizer` (configs[4] per iteration on one GPU: 131072 candidates, score only) are the
other two single-GPU workloads, each with its own roofline.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0    # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
F64_MFMA_PEAK_TFLOPS = 78.6   # AMD's published MI355X float64 matrix figure (the guide has no float64 row)
PMC_SUMMARIES = [os.path.join(ROOT, "profiles", n) for n in ("r05_summary.json", "r05_frame_constraints_summary.json", "r04a_summary.json")]   # tools/prof_all.sh (r04a: its counter passes ran the tile-major kernel, the one slow-class boxes get)
METRIC = "motion-primitive samples scored+back-projected/sec; fraction of HBM roofline"
L, F, D, NB, K = 40, 156, 79, 31, 8


def pmc_traffic_bytes(kernel_substr):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same command
    (FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs, KiB units; FETCH_SIZE doubled as the gfx950
    correction for wide coalesced reads prescribes).  None when no summary is present."""
    for path in PMC_SUMMARIES:
        try:
            with open(path) as f:
                summ = json.load(f)
            for name, c in summ.get("pmc", {}).items():
                if kernel_substr in name and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                    return (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0, os.path.join("profiles", os.path.basename(path))
        except (OSError, ValueError):
            pass
    return None


def algorithmic_bytes(B):
    """per candidate 4L (latent) + 4FD (frames) + 4 (log p); per launch the constants E, mean, basis, GMM."""
    per_cand = 4 * L + 4 * F * D + 4
    consts = 4 * (NB * D * L + NB * D + 4 * F + K * (L * L + L + 1))
    return per_cand, consts, per_cand * B + consts


# ---------------------------------------------------------------------------------------------------------
# launching N ranks from one command line
# ---------------------------------------------------------------------------------------------------------
def self_launch(args, argv):
    """`python bench.py --gpus N` typed as is: N child processes (one rank per GPU), started before this process has
    touched a GPU and never exec'ed into; rank 0's stdout is forwarded, the exit code is the worst child's."""
    import shutil
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    rdv_dir = tempfile.mkdtemp(prefix="mg_bench_rdv_")      # 0700, a name nobody can guess: the ranks' rendezvous files
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus), "LOCAL_WORLD_SIZE": str(args.gpus), "MG_RDV_DIR": rdv_dir,
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's stdout is collected by a thread; the launcher polls ALL children: when one dies the others are ended (a rank
    # alone in a rendezvous or inside an RCCL collective never returns), and the whole run is bounded
    import glob
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + float(os.environ.get("MG_BENCH_LAUNCH_TIMEOUT", "1500"))
    # the communicator's set-up is bounded on its own: every rank drops a file once it is through mg_dist_init (a collective: a
    # rank whose peer never arrives would sit in it until the run's deadline); ranks missing after MG_BENCH_RDV_TIMEOUT seconds end
    # the launch -- fresh children or an exit, never a re-exec of a process that touched the GPU
    rdv_deadline = time.time() + float(os.environ.get("MG_BENCH_RDV_TIMEOUT", "240"))
    joined = args.gpus == 1 or args.dry_run
    failed = None
    try:
        while any(p.poll() is None for p in procs):
            bad = [r for r, p in enumerate(procs) if p.poll() not in (None, 0)]
            if not joined:
                joined = all(os.path.exists(os.path.join(rdv_dir, "joined.%d" % r)) for r in range(args.gpus))
            if bad or time.time() > deadline or (not joined and time.time() > rdv_deadline):
                missing = [r for r in range(args.gpus) if not os.path.exists(os.path.join(rdv_dir, "joined.%d" % r))]
                failed = ("rank %d exited with %d" % (bad[0], procs[bad[0]].returncode)) if bad else \
                    ("timed out" if joined else "ranks %s had not joined the communicator after %s s" % (missing, os.environ.get("MG_BENCH_RDV_TIMEOUT", "240")))
                for p in procs:
                    if p.poll() is None:
                        p.terminate()
                for p in procs:
                    try:
                        p.wait(10)
                    except subprocess.TimeoutExpired:
                        p.kill()
                break
            time.sleep(0.05)
    finally:
        shutil.rmtree(rdv_dir, ignore_errors=True)   # the rendezvous files, whatever a dead rank left
    reader.join(10)
    rcs = [p.wait() for p in procs]
    sys.stdout.write(b"".join(c for c in chunks if c).decode("utf-8", "replace"))
    sys.stdout.flush()
    if failed:
        sys.stderr.write("bench.py launcher: %s; the other ranks were ended\n" % failed)
        return 1
    return max(abs(rc) for rc in rcs)


def FileRendezvous(rank, world):
    """How the 128-byte RCCL unique id (and, in a dry run, everything else) travels between the ranks of one node without
    torch: morphablegraphs_amd.distributed.FileRendezvous under a name made of the launcher's pid and the master port."""
    from morphablegraphs_amd.distributed import FileRendezvous as _Rdv
    return _Rdv(rank, world)     # $MG_RDV_DIR (this launcher's mkdtemp) or, under torch.distributed.run, a private directory named after its pid and port


def ensure_built():
    if os.path.exists(os.path.join(ROOT, "morphablegraphs_amd", "csrc", "libmg_hip.so")):
        return
    # git-ignored build product missing (fresh checkout): local rank 0 builds it, the others wait for the file
    if int(os.environ.get("LOCAL_RANK", "0")) == 0:
        import __graft_entry__
        __graft_entry__.build()
    else:
        deadline = time.time() + 900.0
        while not os.path.exists(os.path.join(ROOT, "oracle", "libmg_oracle.so")):   # built last
            if time.time() > deadline:
                raise SystemExit("libmg_hip.so was not built by local rank 0 within 15 minutes")
            time.sleep(1.0)
        time.sleep(2.0)


# ---------------------------------------------------------------------------------------------------------
# the headline workload: frames + log p of 8192 candidates per GPU
# ---------------------------------------------------------------------------------------------------------
def run_walk(args, rank, local_rank, world):
    dry = args.dry_run
    if dry and os.environ.get("MG_BENCH_DRY_RUN_DIES") == str(rank):   # launcher test: a rank that dies before the rendezvous
        raise SystemExit(3)
    if world > 1 and os.environ.get("MG_BENCH_STALL_IN_SETUP"):        # launcher test: ranks that never get through the set-up
        time.sleep(600)
    rdv = FileRendezvous(rank, world) if world > 1 else None
    B = int(args.batch)
    per_cand, consts, bytes_launch = algorithmic_bytes(B)
    if dry:
        # no GPU in this process: the launcher, the rendezvous, the timing protocol and the JSON contract are
        # exercised with a stand-in step; `value` means nothing and the line says so
        scores = np.full((B,), float(rank), dtype=np.float32)

        def step():
            time.sleep(2e-4)

        def gather():
            return np.concatenate([np.frombuffer(b, dtype=np.float32) for b in rdv.all_gather(scores.tobytes())])

        def barrier():
            if world > 1:
                rdv.all_gather(b"b")

        def max_over_ranks(x):
            if world == 1:
                return x
            return max(float(np.frombuffer(b, dtype=np.float64)[0]) for b in rdv.all_gather(np.array([x]).tobytes()))
        for _ in range(args.warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        gathered = gather() if world > 1 else scores
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        if world > 1:
            assert gathered.shape == (world * B,) and all(gathered[r * B] == r for r in range(world))
        if rank == 0:
            print(json.dumps({"metric": METRIC, "value": world * B * args.steps / elapsed, "unit": "samples/s", "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
                              "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                              "dry_run": True,
                              "config": {"workload": "DRY RUN (no GPU): launcher, rendezvous and timing protocol only", "candidates_per_gpu": B,
                                         "global_candidates": world * B, "collective": "file rendezvous"}}))
        return 0

    ensure_built()
    from morphablegraphs_amd import _capi, synthetic
    from morphablegraphs_amd.distributed import first_min_argmin
    from morphablegraphs_amd.gaussian_mixture import sample_like_sklearn
    data = synthetic.make_walk_primitive(seed=0)
    dev_index = local_rank   # one rank per GPU of the node
    if os.environ.get("MG_BENCH_OVERSUBSCRIBE"):   # rehearsal on a box with fewer GPUs than ranks: the ranks share that many devices
        dev_index = local_rank % max(1, int(os.environ["MG_BENCH_OVERSUBSCRIBE"]))
    rccl_info = None
    try:
        ctx = _capi.Context(dev_index)
    except _capi.MGError as e:
        raise SystemExit("bench.py needs an MI355X per rank; there is no CPU fallback (%s)" % e)
    if world > 1:
        # The library's own RCCL communicator (librccl loaded by libmg_hip, the unique id handed over through the file
        # rendezvous).  Preflight first (what can fail on one rank alone), then the collective set-up; the ranks tell each other
        # how it went.  If it cannot be set up on ANY rank the run ends with the reason on every rank.
        uid, err = b"", b""
        try:
            ctx.dist_preflight()
            if rank == 0:
                uid = ctx.dist_unique_id()
        except Exception as e:   # noqa: BLE001 -- whatever went wrong, the other ranks must hear of it
            err = ("rank %d: %s" % (rank, e)).encode()[:300]
        errs = [e for e in rdv.all_gather(err) if e]
        uid, err0 = rdv.all_gather(uid)[0], (errs[0] if errs else b"")
        err = b""
        if not err0:
            try:
                ctx.dist_init(rank, world, uid)
            except Exception as e:   # noqa: BLE001
                err = ("rank %d: mg_dist_init: %s" % (rank, e)).encode()[:300]
            errs = [e for e in rdv.all_gather(err) if e]
            err0 = errs[0] if errs else b""
            if err0:
                try:
                    ctx.dist_finalize()
                except Exception:   # noqa: BLE001
                    pass
        if err0:
            if rank == 0:
                print("bench.py: mg_dist_* could not be set up (%s); the run ends here (there is no other carrier of the collective)" % err0.decode(errors="replace"),
                      file=sys.stderr)
            raise SystemExit(4)
        rccl_info = ctx.dist_info()
    if world > 1 and os.environ.get("MG_RDV_DIR"):      # tell the launcher's watchdog: this rank is through the set-up
        open(os.path.join(os.environ["MG_RDV_DIR"], "joined.%d" % rank), "w").close()
    prim = _capi.Primitive(ctx, data)
    if args.frames_kernel:
        ctx.set_option(_capi.MG_OPT_FRAMES_KERNEL, args.frames_kernel)
    kernel_name = prim.step_plan(B)["kernel"]   # (refined below, once the output buffer and its placement class are known)

    # synthetic latents: sklearn-style GMM draw on the host (RandomState(rank)), cast to f32, resident in HBM
    rs = np.random.RandomState(rank)
    S_host = sample_like_sklearn(B, np.array(data["gmm_weights"]), np.array(data["gmm_means"]),
                                 np.array(data["gmm_covars"]), rs)[0].astype(np.float32)
    S = ctx.upload(S_host)
    nbytes = B * F * D * 4
    # the output comes from the library's allocator for large outputs: where 404 MB land in HBM decides between two
    # speed classes of the kernel's store stream (DESIGN.md "Placement"); mg_device_malloc_placed probes for the fast one
    t_alloc = time.perf_counter()
    if args.output_alloc == "plain":      # memory that does not come through the library's placed regions: one hipMalloc
        ctx.set_option(_capi.MG_OPT_PLAIN_MALLOC, 1)
        frames = ctx.malloc(nbytes)
        ctx.set_option(_capi.MG_OPT_PLAIN_MALLOC, 0)
    else:
        frames = ctx.malloc_placed(nbytes)
    alloc_s = time.perf_counter() - t_alloc
    kernel_name = prim.step_plan(B, frames)["kernel"]   # a slow-class piece of the output arena gets the tile-major kernel
    probe = ctx.probe_placement(frames)   # pattern and fill time on THIS buffer: the in-run achievable ceiling
    fill_us = probe["pattern_us"] / probe["ratio"] if probe["ratio"] > 0 else None
    logps = [ctx.malloc(B * 4) for _ in range(2)]
    gathereds = [ctx.malloc(world * B * 4) for _ in range(2)] if world > 1 else None
    scalar_dev = ctx.malloc(8)
    scalars_dev = ctx.malloc(8 * max(world, 1))
    step_no = [0]

    def step():
        b = step_no[0] & 1
        step_no[0] += 1
        lp_ptr = logps[b].ptr.value
        if args.two_launch:
            prim.back_project_frames_dev(S, np.float32, B, L, frames, path=_capi.MG_PATH_MFMA)
            prim.gmm_log_prob_dev(S, np.float32, B, L, lp_ptr, np.float32)
        else:
            # mg_step_frames_and_logp: one launch, the mixture is scored inside the frames kernel
            prim.step_frames_and_logp_dev(S, np.float32, B, L, frames, lp_ptr)
        if world > 1:
            ctx.dist_all_gather(logps[b], gathereds[b], B, np.float32)   # RCCL on the kernels' stream, stream ordered

    def sync():
        ctx.synchronize()

    def barrier():
        sync()
        if world > 1:
            ctx.dist_all_gather(scalar_dev, scalars_dev, 1, np.float64)
        sync()

    def max_over_ranks(x):
        if world == 1:
            return x
        ctx.lib.mg_memcpy_h2d(ctx.handle, scalar_dev.ptr, np.array([x], dtype=np.float64).ctypes.data, 8)
        ctx.dist_all_gather(scalar_dev, scalars_dev, 1, np.float64)
        return float(ctx.download(scalars_dev, (world,), np.float64).max())

    # the chip raises its clock over the first ~0.1 s of load (the same kernel takes 98 us in a cold 25-launch run and 85 us
    # in steady state): an untimed ramp before the W warm-up steps, so that a short run measures the steady state too
    for _ in range(args.ramp_steps):
        step()
    for _ in range(args.warmup):
        step()
    barrier()
    # THE TIMED REGION: exactly K steps between barrier + synchronise, nothing else in it -- no timing events, no profiling
    # (round 4 bracketed every second launch with dispatch events inside this region: ~4 us each, 11.5 us of step - kernel on the
    # driver's box).  Five such windows back to back; the headline is the MEDIAN window (one hiccup of a shared box cannot move it),
    # all five are in the line (`ms_per_step_windows`).  The kernel's own duration comes from a SECOND pass of K steps with a
    # dispatch-attached event pair on every launch.
    windows = []
    for _ in range(5 if not args.single_window else 1):
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        windows.append(max_over_ranks(time.perf_counter() - t0))
    elapsed = float(np.median(windows))
    interval = max(1, args.event_interval) if args.event_interval else 1
    if not args.no_profile_events:
        ctx.profile_reset()
        ctx.profile_enable(interval)
        for _ in range(args.steps):
            step()
        barrier()
        ctx.profile_enable(False)
    last = (step_no[0] - 1) & 1
    gather_us = None
    if world > 1:
        # the collective alone: 200 all-gathers of the step's size back to back on the kernels' stream, nothing else running
        for _ in range(20):
            ctx.dist_all_gather(logps[0], gathereds[0], B, np.float32)
        barrier()
        tg = time.perf_counter()
        for _ in range(200):
            ctx.dist_all_gather(logps[0], gathereds[0], B, np.float32)
        sync()
        gather_us = max_over_ranks(1e6 * (time.perf_counter() - tg) / 200)
        # every rank holds the same global score vector: the graph-walk argmin needs no further exchange
        g_host = ctx.download(gathereds[last], (world * B,), np.float32)
        best_idx, best_val = first_min_argmin(-g_host)   # most likely candidate
    frames_ms, frames_n = ctx.profile_get("frames")
    gmm_ms, gmm_n = ctx.profile_get("gmm_log_prob")

    # sanity: the timed work produced real output
    row = ctx.download(frames.ptr.value + (B // 2) * F * D * 4 + (F - 1) * D * 4, (4,), np.float32)
    lp_host = ctx.download(logps[last], (8,), np.float32)
    if not np.isfinite(float(row.sum()) + float(lp_host.sum())):
        raise SystemExit("non-finite output")

    rc = 0
    if rank == 0:
        value = world * B * args.steps / elapsed
        placement = {"allocator": "one hipMalloc (MG_OPT_PLAIN_MALLOC)" if args.output_alloc == "plain" else "the library's placed output regions (mg_device_malloc / mg_device_malloc_placed: every buffer of 64 MiB and more, "
                                  "the *_host entry points' scratch included, is a piece of a region that went through the placement probe; scan of 32 candidates, up to 400 when those were all slow -- six tenths of the free memory held while that runs)",
                     "candidates_probed": frames.placement["probed"] if frames.placement else 0, "alloc_seconds": round(alloc_s, 4),
                     "pattern_over_fill": round(probe["ratio"], 4)}
        # the class is the arena's: decided once from the scan's own measurement and kept (mg_device_placement_info); the probe
        # above only supplies the in-run fill / pattern times of the roofline's 'achievable' figures
        arena = ctx.placement_info(frames)
        placement["fast_class"] = arena["fast"] if arena["region"] else probe["fast"]
        placement["pattern_TBps"] = round(arena["pattern_TBps"], 3) if arena["region"] else round(B * F * D * 4 / probe["pattern_us"] * 1e-6, 3)
        result = {
            "metric": METRIC, "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "untimed_steps": args.warmup + args.ramp_steps,
            "ms_per_step_windows": {"n": len(windows), "steps_each": args.steps, "min": 1e3 * min(windows) / args.steps, "median": 1e3 * elapsed / args.steps,
                                    "max": 1e3 * max(windows) / args.steps, "all": [round(1e3 * w / args.steps, 6) for w in windows],
                                    "headline": "the median window; no timing events inside any window"},
            "config": {"workload": "walk primitive L=40 F=156 D=79 NB=31 GMM k=8, batch=%d candidates/GPU "
                                   "(BASELINE.json configs[%d])" % (B, 1 if world == 1 else 3),
                       "candidates_per_gpu": B, "global_candidates": world * B,
                       "collective": "all_gather(logp) every step, RCCL through mg_dist_all_gather on the kernels' stream" if world > 1 else "none",
                       "rccl": ({"ranks_seen_by_rccl": rccl_info["ranks"], "rank0_device": rccl_info["device"],
                                 "all_gather_alone_us": round(gather_us, 2) if gather_us is not None else None,
                                 "all_gather_bytes_per_rank": 4 * B} if rccl_info else None),
                       "sharding": "contiguous candidate blocks, constants replicated",
                       "clock_ramp_steps": args.ramp_steps,
                       "output_placement": placement},
        }
        if frames_n > 0:
            avg_ms = frames_ms / frames_n
            # algorithmic bytes of the dominant kernel.  Fused step kernel: the whole step (latents, frames, log p,
            # E, mean, basis, mixture constants); stand-alone frames kernel: latents + frames (+ E, mean, basis)
            k_bytes = (B * (4 * L + 4 * F * D) + 4 * (NB * D * L + NB * D + 4 * F)) if args.two_launch else bytes_launch
            achieved = k_bytes / (avg_ms * 1e-3) / 1e9
            traffic = pmc_traffic_bytes(kernel_name) if B == 8192 else None       # (bytes, the file they were read from)
            result["roofline"] = {
                "bound": "hbm", "kernel": kernel_name, "achieved": achieved, "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic[0] if traffic else None,
                "traffic_source": ("committed rocprofv3 --pmc passes of this command (%s), not measured in this run" % traffic[1]) if traffic else None,
                "avg_kernel_ms": avg_ms, "launches_timed": frames_n, "event_interval": interval,
                "kernel_events": "a second pass of %d steps after the timed windows, a dispatch-attached event pair on every %s launch" % (args.steps, "" if interval == 1 else "%d-th" % interval),
                "step_minus_kernel_us": 1e3 * (1e3 * elapsed / args.steps - avg_ms),
                "algorithmic_bytes_per_launch": k_bytes, "launches_per_step": 2 if args.two_launch else 1,
                "gmm_kernel_avg_ms": (gmm_ms / gmm_n) if gmm_n else None,
                "step_algorithmic_bytes": bytes_launch, "step_achieved_GBps": bytes_launch / (elapsed / args.steps) / 1e9,
            }
            if fill_us:   # both denominators -- the spec peak and what a plain fill of the same buffer reaches in this run
                ach_peak = nbytes / (fill_us * 1e-6) / 1e9
                result["roofline"].update({"achievable_peak": ach_peak, "frac_of_achievable": achieved / ach_peak,
                                           "achievable_peak_how": "plain float4 fill of the same %d-byte output buffer, %.1f us" % (nbytes, fill_us),
                                           "store_stream_alone_us": probe["pattern_us"]})
            smp = np.sort(ctx.profile_samples("frames"))
            if len(smp) >= 10:   # spread of the event-bracketed launches (median, p10 / p90)
                result["roofline"].update({"kernel_ms_p10": float(smp[int(0.10 * (len(smp) - 1))]),
                                           "kernel_ms_p50": float(smp[int(0.50 * (len(smp) - 1))]),
                                           "kernel_ms_p90": float(smp[int(0.90 * (len(smp) - 1))])})
        if world == 1:
            def wall_us(n, out_buf):
                for _ in range(20):
                    prim.step_frames_and_logp_dev(S, np.float32, B, L, out_buf, logps[0])
                ctx.synchronize()
                t = time.perf_counter()
                for _ in range(n):
                    prim.step_frames_and_logp_dev(S, np.float32, B, L, out_buf, logps[0])
                ctx.synchronize()
                return 1e6 * (time.perf_counter() - t) / n
            if args.output_alloc != "plain" and not args.no_placement_compare:
                # beside the headline: (1) a SECOND buffer from the allocation call every device-pointer caller uses (mg_device_malloc: the
                # headline's region is in use, so this one is a region of its own, scanned for like the first); (2) memory that does not
                # come from the library at all -- one hipMalloc, wherever it landed -- which is what a caller's own tensor would be
                second = ctx.malloc(nbytes)
                ps = ctx.placement_info(second)      # the arena's own record of the region this piece came from
                us_second = wall_us(200, second)
                result["config"]["output_placement"]["default_malloc"] = {"allocator": "mg_device_malloc (a second buffer of the same size beside the headline's)", "step_us": round(us_second, 2),
                                                                          "pattern_over_fill": round(ps["ratio"], 4), "fast_class": ps["fast"], "pattern_TBps": round(ps["pattern_TBps"], 3)}
                second.free()
                ctx.set_option(_capi.MG_OPT_PLAIN_MALLOC, 1)
                plain = ctx.malloc(nbytes)
                ctx.set_option(_capi.MG_OPT_PLAIN_MALLOC, 0)
                pp = ctx.probe_placement(plain)
                us_plain = wall_us(200, plain)
                result["config"]["output_placement"]["foreign_hipMalloc"] = {"allocator": "one hipMalloc outside the library's regions (MG_OPT_PLAIN_MALLOC)", "step_us": round(us_plain, 2),
                                                                             "pattern_over_fill": round(pp["ratio"], 4), "fast_class": pp["fast"],
                                                                             "pattern_TBps": round(B * F * D * 4 / pp["pattern_us"] * 1e-6, 3) if pp["pattern_us"] > 0 else None}
                plain.free()
                result["value_default_malloc"] = B / (us_second * 1e-6)
                result["value_foreign_hipMalloc"] = B / (us_plain * 1e-6)
                res, used, nreg, nfast = ctx.output_bytes()
                result["config"]["output_placement"]["regions"] = {"reserved_bytes": res, "in_use_bytes": used, "regions": nreg, "fast": nfast}
            # what the boundary costs when it hands over host buffers (never part of `value`): median of 20 copies
            def med_us(fn):
                ts = []
                for _ in range(20):
                    ctx.synchronize()
                    t = time.perf_counter()
                    fn()
                    ts.append(time.perf_counter() - t)
                return 1e6 * float(np.median(ts))
            lp_h = np.empty((B,), dtype=np.float32)
            fr_h = np.empty((F, D), dtype=np.float32)
            C = __import__("ctypes")
            result["transfers_us"] = {
                "h2d_latents": med_us(lambda: ctx.lib.mg_memcpy_h2d(ctx.handle, S.ptr, S_host.ctypes.data_as(C.c_void_p), S_host.nbytes)), "h2d_latents_bytes": int(S_host.nbytes),
                "d2h_scores": med_us(lambda: ctx.lib.mg_memcpy_d2h(ctx.handle, lp_h.ctypes.data_as(C.c_void_p), logps[0].ptr, lp_h.nbytes)), "d2h_scores_bytes": 4 * B,
                "d2h_winner_frames": med_us(lambda: ctx.lib.mg_memcpy_d2h(ctx.handle, fr_h.ctypes.data_as(C.c_void_p), C.c_void_p(frames.ptr.value + (B // 2) * F * D * 4), fr_h.nbytes)),
                "d2h_winner_frames_bytes": 4 * F * D, "note": "pageable host memory, synchronous copies through the C-ABI",
            }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline_leg(data, S_host)
    for buf in [S, frames, scalar_dev, scalars_dev] + logps + (gathereds or []):
        buf.free()
    prim.close()
    if world > 1:
        ctx.dist_finalize()
    ctx.close()
    if rank == 0:
        if world == 1 and not args.no_extra_configs and B == 8192 and args.dry_run is False:
            # BASELINE configs[2] and configs[4] on this GPU, a few hundred steps each, under keys of their own (the headline keys
            # above are configs[1] and unchanged); `python bench.py --config graph | optimizer` prints each as a line of its own
            import copy
            for key, fn in (("graph", run_graph), ("optimizer", run_optimizer), ("frame_constraints", run_frame_constraints)):
                a2 = copy.copy(args)
                a2.steps, a2.warmup, a2.batch = (400, 100, 8192) if key != "frame_constraints" else (200, 20, 8192)
                try:
                    r = fn(a2, emit=False)
                    result[key] = {k: r[k] for k in ("value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline")}
                except Exception as e:   # the headline line is never lost to a secondary configuration
                    result[key] = {"error": "%s: %s" % (type(e).__name__, e)}
        print(json.dumps(result))
    return rc


def _ref_worker(job):
    data, S, budget = job
    from oracle import cpu_baseline
    return cpu_baseline.reference_shaped_rate(data, S, budget_s=budget, max_candidates=1 << 30)


def cpu_baseline_leg(data, S_host):
    """The reference-shaped per-candidate loop (numpy dot + 79 x scipy splev + sklearn score_samples) on one core and on
    N processes over disjoint slices, the plain-C oracle, the vectorised form: ~35 s of CPU in all."""
    from oracle import cpu_baseline
    ref = cpu_baseline.reference_shaped_rate(data, S_host, budget_s=10.0, max_candidates=1 << 30)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    # the cores this process may actually USE: the affinity mask says which CPUs it may run on, the cgroup's CPU quota how
    # much CPU time it gets (a one-GPU box of the pool shows 256 CPUs and grants 16 cores' worth: thirty-two processes then run
    # at half speed each, which is what the 15.5 x of round 2's 32-process leg was); one process per granted core
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                      # cgroup v2
            q, per = f.read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:      # cgroup v1
                q = float(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = float(f.read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    usable = avail if quota is None else max(1, min(avail, int(quota + 0.5)))
    nproc = max(1, min(usable, 32 if quota is None else 64))
    multi = None
    if nproc > 1:
        import multiprocessing as mp
        os.environ.setdefault("OMP_NUM_THREADS", "1")
        saved = {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")}
        for k in saved:
            os.environ[k] = "1"      # one core per process, as the reference's parallel server runs (one process per core)
        try:
            with mp.get_context("spawn").Pool(nproc) as pool:
                parts = np.array_split(S_host, nproc)
                t0 = time.perf_counter()
                res = pool.map(_ref_worker, [(data, p, 8.0) for p in parts])
                wall = time.perf_counter() - t0
            multi = {"rate": sum(r["n"] for r in res) / max(r["seconds"] for r in res), "n": sum(r["n"] for r in res),
                     "seconds": max(r["seconds"] for r in res), "wall_with_startup": wall}
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    cport = cpu_baseline.c_port_rate(data, S_host, budget_s=5.0)
    vec = cpu_baseline.vectorised_rate(data, S_host, budget_s=5.0)
    out = {
        "value": ref["rate"], "unit": "samples/s", "cores": 1, "kind": "port",
        "sample": "%d candidates drawn cyclically from the same batch, reference-shaped per-candidate loop "
                  "(numpy dot + 79x scipy splev + sklearn score_samples), %.1f s" % (ref["n"], ref["seconds"]),
        "c_port_value": cport["rate"],
        "c_port_sample": "%d candidates, plain-C float64 oracle (oracle/mg_oracle.c), 1 core, %.1f s" % (cport["n"], cport["seconds"]),
        "vectorised_value": vec["rate"], "vectorised_cores": vec["threads"],
        "vectorised_sample": "%d candidates, float32 GEMM (coefficients) + batched basis GEMM (frames materialised) + "
                             "batched sklearn score_samples, %d BLAS threads, %.1f s" % (vec["n"], vec["threads"], vec["seconds"]),
        "host_cores_available": os.cpu_count(), "affinity_cores": avail, "cgroup_cpu_quota_cores": quota,
    }
    if multi:
        out.update({"multi_process_value": multi["rate"], "multi_process_cores": nproc,
                    "multi_process_efficiency": multi["rate"] / (nproc * ref["rate"]),   # 1.0 = every process ran as fast as the single one; less: the
                                                                                        # processes shared cores (CPU quota below the process count) or memory bandwidth
                    "multi_process_sample": "the same reference-shaped loop in %d processes (one core each, disjoint candidate slices), %d candidates, %.1f s"
                                            % (nproc, multi["n"], multi["seconds"])})
    return out


# ---------------------------------------------------------------------------------------------------------
# BASELINE configs[2]: one planner step over the 16 options of a graph, 4096 candidates each
# ---------------------------------------------------------------------------------------------------------
def run_graph(args, emit=True):
    ensure_built()
    from morphablegraphs_amd import synthetic
    from morphablegraphs_amd.motion_state_graph import HipPrimitiveSet
    n = int(args.batch) if args.batch != 8192 else 4096
    prims = synthetic.make_graph_primitives(16)
    names = [p["name"] for p in prims]
    cons = {nm: [{"type": "position", "t": float(p["n_canonical_frames"] - 1), "weight": 1.0, "target": [10.0, None, 5.0]},
                 {"type": "direction", "t": float(p["n_canonical_frames"] - 1), "weight": 1.0, "target": [0.5, 1.0]}] for nm, p in zip(names, prims)}
    pset = HipPrimitiveSet(prims, separate_streams=False)
    ctxs = list({id(pset.nodes[nm]._prim.ctx): pset.nodes[nm]._prim.ctx for nm in names}.values())
    dev_counts = not getattr(args, "host_counts", False)
    for i in range(args.warmup):
        pset.evaluate_options_on_device(names, cons, n, seed=i, device_counts=dev_counts)
    for c in ctxs:
        c.profile_reset()
        c.profile_enable(1 if args.steps < 200 else 8)   # every 8th launch of a kind: event pairs around 4-us kernels are not free
    t0 = time.perf_counter()
    for i in range(args.steps):
        best, results = pset.evaluate_options_on_device(names, cons, n, seed=i, device_counts=dev_counts)
    elapsed = time.perf_counter() - t0
    slots = {}
    for c in ctxs:
        c.profile_enable(False)
        for slot in ("gmm_sample", "score_constraints", "argmin", "options_step"):
            ms, cnt = c.profile_get(slot)
            a = slots.setdefault(slot, [0.0, 0])
            a[0] += ms
            a[1] += cnt
    # flops of a step (float64 matrix pipe): the scorer's X . W^T with 14 keyframe channel rows per candidate, per option its
    # own L, and the sampler's x = mu + z L^T (triangular: L^2 per candidate)
    Ls = [int(np.shape(p["eigen_vectors_spatial"])[0]) for p in prims]
    score_flop = sum(2 * 14 * l for l in Ls) * n
    sample_flop = sum(l * l for l in Ls) * n
    fused = slots["options_step"][1] > 0
    if fused:   # the whole step is one launch of mg_options_fused_kernel
        k_ms = slots["options_step"][0] / slots["options_step"][1]
        k_flop, k_name, k_n, launches = score_flop + sample_flop, "mg_options_fused_kernel (one launch per step)", slots["options_step"][1], 1
        note = ("one launch per planner step: workgroups dealt over the options, a wave draws a 16-candidate tile (Philox + Box-Muller, x = mu + z L^T on the f64 "
                "matrix pipe), scores it from LDS and keeps the first minimum; the last workgroup of an option reduces and copies the winner; one read-back. "
                "The kernel is bound by its start-up (table and fragment loads per workgroup), the float64 vector work of the residuals and the last workgroup's "
                "reduction, not by the matrix pipe; no fences (write-through partials), the next step's component counts drawn ahead")
    else:
        k_ms = slots["score_constraints"][0] / max(1, slots["score_constraints"][1])
        k_flop, k_name, k_n, launches = score_flop / len(names), "mg_score_mfma_kernel (one per option)", slots["score_constraints"][1], 3 * len(names)
        note = "3 launch-latency-bound kernels per option (sampler, scorer, argmin + winner copy), all options enqueued by one C call and read back by one copy"
    result = {
        "metric": METRIC, "value": len(names) * n * args.steps / elapsed, "unit": "samples/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "graph-walk planner step: 16 synthetic primitives (L 12..40, F 40..160, K 1..8) x %d device-sampled candidates each, "
                               "2 root keyframe constraints, winner per option read back (BASELINE.json configs[2]); score only, no frames written" % n,
                   "options": len(names), "candidates_per_option": n, "launches_per_step": launches, "host_calls_per_step": 1,
                   "read_backs_per_step": 1,
                   "component_counts": ("drawn on the device (mg_options_step_device_counts: Philox-keyed multinomial per option, distributed like "
                                        "numpy.random.multinomial's counts, not NumPy's stream; --host-counts keeps the host draw); a step's kernel draws the counts "
                                        "of seed + 1 as well, so that this loop (seed = step number) needs the counts kernel in front only once") if dev_counts else
                                       "numpy.random.multinomial on the host, one call per option"},
        "roofline": {"bound": "mfma", "kernel": k_name, "achieved": k_flop / (k_ms * 1e-3) / 1e12 if k_ms else None,
                     "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": (k_flop / (k_ms * 1e-3) / 1e12 / F64_MFMA_PEAK_TFLOPS) if k_ms else None, "traffic": None,
                     "avg_kernel_ms": k_ms, "launches_timed": k_n,
                     "note": note + "; step flops %.3g (scorer) + %.3g (sampler)" % (score_flop, sample_flop),
                     "per_kernel_avg_us": {k: (1e3 * v[0] / v[1] if v[1] else None) for k, v in slots.items()}},
    }
    if not emit:
        return result
    print(json.dumps(result))
    return 0


# ---------------------------------------------------------------------------------------------------------
# BASELINE configs[4] per iteration on one GPU: 131072 candidates, objective only (no frames)
# ---------------------------------------------------------------------------------------------------------
def run_optimizer(args, emit=True):
    ensure_built()
    from morphablegraphs_amd import _capi, synthetic
    B = int(args.batch) if args.batch != 8192 else 131072
    ctx = _capi.Context(0)
    data = synthetic.make_walk_primitive(seed=0)
    prim = _capi.Primitive(ctx, data)
    cons = [{"type": "position", "t": 155.0, "weight": 1.0, "target": [40.0, None, -30.0]},
            {"type": "direction", "t": 155.0, "weight": 1.0, "target": [0.5, 1.0]}]
    cset = _capi.ConstraintSet(prim, cons)
    rows = 14
    flop = K * (2 * L * L + 2 * L) + 2 * rows * L          # mixture (dense x P_k, as sklearn does) + the keyframe channel rows
    S = ctx.upload(np.random.default_rng(0).standard_normal((B, L)).astype(np.float32))
    lp, err = ctx.malloc(B * 4), ctx.malloc(B * 8)
    obj = ctx.malloc(B * 8)
    fused = not getattr(args, "two_launch", False)
    if fused:
        try:   # the iteration in ONE launch (mg_objective_error_and_naturalness: the mixture kernel scores the constraints on the tile it holds)
            prim.objective_dev(cset, S, np.float32, B, L, 1.0, 1.0, obj_dev=obj)
        except _capi.MGError:
            fused = False

    def step():
        if fused:
            prim.objective_dev(cset, S, np.float32, B, L, 1.0, 1.0, obj_dev=obj)
        else:
            prim.gmm_log_prob_dev(S, np.float32, B, L, lp, np.float32)
            prim.score_constraints_dev(cset, S, np.float32, B, L, err, np.float64)
    for _ in range(args.warmup):
        step()
    ctx.synchronize()
    ctx.profile_reset()
    ctx.profile_enable(1 if args.steps < 200 else 8)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.synchronize()
    elapsed = time.perf_counter() - t0
    ctx.profile_enable(False)
    g_ms, g_n = ctx.profile_get("gmm_log_prob")
    s_ms, s_n = ctx.profile_get("score_constraints")
    gmm_flop = B * K * (2 * L * L + 2 * L) + (B * 2 * rows * L if fused else 0)   # fused: the one kernel does the keyframe channel rows as well
    g_avg = g_ms / max(1, g_n)
    result = {
        "metric": METRIC, "value": B * args.steps / elapsed, "unit": "samples/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "optimizer inner loop, score only: log p(x) + 2 root keyframe constraints of %d candidates per iteration on one GPU "
                               "(BASELINE.json configs[4] is this per iteration over 8 GPUs); no frames written, %d bytes per candidate" % (B, 4 * L + 12),
                   "candidates_per_iteration": B, "launches_per_step": 1 if fused else 2, "flop_per_candidate": flop,
                   "objective": ("error_scale * constraint error + quality_scale * (-log p) written by one launch (mg_objective_error_and_naturalness), %d bytes per candidate"
                                 % (4 * L + 8)) if fused else "log p (float32) and constraint errors (float64) by two launches"},
        "roofline": {"bound": "mfma", "kernel": "mg_gmm_logp_lds_kernel<.., SCORE> (mixture + keyframe constraints on the same latent tile)" if fused else "mg_gmm_logp_lds_kernel", "achieved": gmm_flop / (g_avg * 1e-3) / 1e12 if g_avg else None,
                     "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": (gmm_flop / (g_avg * 1e-3) / 1e12 / F64_MFMA_PEAK_TFLOPS) if g_avg else None,
                     "traffic": None, "avg_kernel_ms": g_avg, "launches_timed": g_n,
                     "score_kernel_avg_ms": (s_ms / max(1, s_n)) if not fused else None, "step_achieved_TFLOPs": B * flop / (elapsed / args.steps) / 1e12,
                     "step_frac": B * flop / (elapsed / args.steps) / 1e12 / F64_MFMA_PEAK_TFLOPS,
                     "peak_note": "float64 matrix peak AMD publishes for MI355X; the microarchitecture guide lists no float64 MFMA row"},
    }
    if emit:
        print(json.dumps(result))
    for b in (S, lp, err, obj):
        b.free()
    cset.close()
    prim.close()
    ctx.close()
    return 0 if emit else result


# ---------------------------------------------------------------------------------------------------------
# Candidates against constraints that walk EVERY frame: a root trajectory + a collision-avoidance position of the hand
# ---------------------------------------------------------------------------------------------------------
def run_frame_constraints(args, emit=True):
    """4096 resident candidates of the 'walk' shape scored against a TrajectoryConstraint on the root path and a
    GlobalTransformCAConstraint on the left hand (the per-frame constraint classes of SURVEY 8 row 'next'): three launches, no frames
    in memory -- mg_score_trajectory (root rows in LDS, closest-point walk), mg_joint_tracks (the hand's track from the control points
    of its chain's channels, in LDS), mg_score_frame_constraints.  The chain these replace (float64 frames -> forward kinematics ->
    scorer) is timed beside it on the same candidates, after the timed region."""
    ensure_built()
    from morphablegraphs_amd import _capi, synthetic
    from morphablegraphs_amd import frame_constraints as fc
    from morphablegraphs_amd.candidate_scoring import cached_trajectory
    B = int(args.batch) if args.batch != 8192 else 4096
    ctx = _capi.Context(0)
    prim = _capi.Primitive(ctx, synthetic.make_path_following_primitive(seed=0))
    joints, animated = synthetic.make_skeleton()
    sk = _capi.Skeleton(joints, animated)
    S_host = np.random.default_rng(0).standard_normal((B, L)).astype(np.float32)
    frames0 = prim.back_project_frames_f64(S_host[:1])[0]
    hand0 = prim.joint_tracks(sk, ["LeftHand"], S_host[:1])[0, :, 0]
    root_traj = {"type": "trajectory", "control_points": (frames0[::26, :3] + 0.25).tolist(), "min_u": 0.0, "weight": 1.0, "granularity": 1000}
    ca = {"type": "frame_ca_position", "joint": "LeftHand", "target": [float(hand0[60, 0]) + 3.0, None, float(hand0[60, 2]) - 2.0], "n_frames": F, "weight": 2.0}
    traj = cached_trajectory(prim, root_traj)
    scorer = fc.TrackScorer(prim, [ca], sk, None)
    S, err = ctx.upload(S_host), ctx.malloc(B * 8)

    def step():
        prim.score_trajectory_dev(traj, S, np.float32, B, L, err)
        scorer.score_dev(S, np.float32, B, L, err, accumulate=True)
    for _ in range(args.warmup):
        step()
    ctx.synchronize()
    ctx.profile_reset()
    ctx.profile_enable(1 if args.steps < 200 else 4)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.synchronize()
    elapsed = time.perf_counter() - t0
    ctx.profile_enable(False)
    per_kernel = {}
    for slot in ("trajectory", "joint_tracks", "frame_constraints"):
        ms, cnt = ctx.profile_get(slot)
        per_kernel[slot] = (ms / cnt) if cnt else None
    got = ctx.download(err, (B,), np.float64)
    # the chain the two new launches replace, on the same candidates: float64 frames in memory, forward kinematics over them, one scorer launch
    fc.FUSED = False
    chain_err = ctx.malloc(B * 8)
    n_chain = max(3, min(20, args.steps // 10))
    for timed in (False, True):
        if timed:
            ctx.synchronize()
            ctx.profile_reset()
            ctx.profile_enable(1)
            tc = time.perf_counter()
        for _ in range(n_chain if timed else 2):
            prim.score_trajectory_dev(traj, S, np.float32, B, L, chain_err)
            fc.add_frame_constraints_dev(prim, S_host, [ca], sk, None, chain_err, accumulate=True)
    ctx.synchronize()
    chain_ms = 1e3 * (time.perf_counter() - tc) / n_chain
    ctx.profile_enable(False)
    fms, fcnt = ctx.profile_get("frames")
    fc.FUSED = True
    same = bool(np.array_equal(got, ctx.download(chain_err, (B,), np.float64)))
    # mg_joint_tracks per candidate: the latent in, the hand's track out; what it keeps OUT of memory: F x D float64 frames written, read by the
    # forward kinematics, the track written and read again
    T, J = F, 1
    alg = B * (4 * L + T * J * 24 + 8)
    chain_bytes = B * (4 * L + 2 * 8 * F * D + 2 * T * J * 24 + 8)
    k_ms = per_kernel["joint_tracks"]
    # The roofline entry describes the step's DOMINANT kernel by time (round 4 named mg_joint_tracks against HBM while 76 % of the step
    # was the closest-point kernel: VERDICT r4).  That kernel moves ~1 MB: neither HBM nor the matrix pipe bounds it.  It is one
    # scalar float64 search per candidate and frame (the reference's L-BFGS-B, csrc/mg_traj_device.h), one lane per candidate: a wave
    # issues the union of its 64 lanes' paths, trip after trip -- bound by the float64 VALU issue rate of the ONE wave a SIMD holds
    # (a wave64 float64 instruction occupies the SIMD's 16 lanes for 4 cycles), on as many SIMDs as the batch has waves.
    dom = max((k for k in per_kernel if per_kernel[k]), key=lambda k: per_kernel[k])
    kernel_names = {"trajectory": "mg_trajectory_stream_kernel", "joint_tracks": "mg_joint_tracks_kernel", "frame_constraints": "mg_frame_constraint_list_kernel"}
    roofline = {"kernel": kernel_names[dom], "avg_kernel_ms": per_kernel[dom], "per_kernel_avg_us": {k: (1e3 * v if v is not None else None) for k, v in per_kernel.items()},
                "traffic": (pmc_traffic_bytes(kernel_names[dom]) or (None,))[0]}
    if dom == "trajectory":
        CLOCK_GHZ, SIMDS, ISSUE_CYCLES = 2.4, 1024, 4.0
        INSTR_PER_TRIP = 1262        # the search's loop body in this build's code object (llvm-objdump: 1262 instructions, 592 of them *_f64; tools/kernel_resources.sh)
        root = prim.joint_tracks(sk, [joints[0][0]], S_host)[:, :, 0]                       # the root paths the searches follow
        _, _, evals = prim.trajectory_closest_points(traj, root, 0.0, evaluations=True)     # (f, g) evaluations of every frame's search
        waves = (B + 63) // 64
        per_wave = np.array([evals[w * 64:(w + 1) * 64].max(axis=0).sum() for w in range(waves)], dtype=np.float64)   # a wave runs its slowest lane's trips, frame by frame
        trips_per_frame = float(per_wave.max()) / F
        cycles_per_frame = per_kernel[dom] * 1e-3 * CLOCK_GHZ * 1e9 / F
        issued = float(per_wave.sum()) * INSTR_PER_TRIP                                      # an upper estimate: every trip issues the whole loop body
        peak = SIMDS * CLOCK_GHZ * 1e9 / ISSUE_CYCLES
        roofline.update({
            "bound": "valu-f64-issue", "unit": "wave-instructions/s", "achieved": issued / (per_kernel[dom] * 1e-3), "peak": peak,
            "frac": issued / (per_kernel[dom] * 1e-3) / peak,
            "waves": waves, "waves_per_simd": waves / float(SIMDS), "frames": F,
            "evaluations_per_frame": {"mean_per_candidate": float(evals.mean()), "slowest_lane_of_the_slowest_wave": trips_per_frame},
            "cycles_per_frame": cycles_per_frame, "cycles_per_trip": cycles_per_frame / trips_per_frame,
            "issue_limited_cycles_per_trip": INSTR_PER_TRIP * ISSUE_CYCLES,
            "note": "the reference's closest-point search (scipy L-BFGS-B restated, one lane per candidate): per frame a wave runs the trips of its slowest lane; a trip is "
                    "one (f, g) evaluation + the line search's bookkeeping, ~%d instructions issued at %d cycles each by the ONE wave its SIMD holds.  The wave is issue "
                    "bound (cycles_per_trip vs issue_limited_cycles_per_trip); the chip is %d waves on %d SIMDs.  HBM and MFMA rooflines do not apply (%.0f MB of HBM traffic per launch, "
                    "most of it the candidates' root paths staged for the search: 8 ms at the kernel's pace would move 60 GB).  "
                    "MG_OPT_TRAJECTORY_SEARCH 1 (the monotone walk, eight lanes per candidate) takes ~0.4 ms for the same step and differs from the reference where "
                    "the distance has several basins" % (INSTR_PER_TRIP, int(ISSUE_CYCLES), waves, SIMDS, (roofline["traffic"] or 3.2e7) / 1e6)})
    else:
        roofline.update({"bound": "hbm", "achieved": alg / (k_ms * 1e-3) / 1e9 if k_ms else None, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": (alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if k_ms else None, "algorithmic_bytes": alg,
                         "note": "mg_joint_tracks moves %d bytes per candidate where the chain moved %d, and is bound by the L2 reads of the eigenvector rows its control points "
                                 "are made from, not by HBM" % (alg // B, chain_bytes // B)})
    result = {
        "metric": METRIC, "value": B * args.steps / elapsed, "unit": "samples/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%d resident candidates of the 'walk' shape against a root TrajectoryConstraint (6 control points, granularity 1000) and a "
                               "collision-avoidance position of the left hand over all %d frames (the reference's per-frame constraint classes, "
                               "constraints/spatial_constraints/); score only, no frames in memory" % (B, F),
                   "candidates": B, "launches_per_step": 3,
                   "chain_replaced": {"ms_per_step": chain_ms, "frames_kernel_avg_ms": (fms / fcnt) if fcnt else None, "steps": n_chain,
                                      "what": "mg_back_project_frames_f64 -> mg_joint_positions -> mg_score_frame_constraint with %d-byte float64 frames per candidate in "
                                              "memory (round 3's route, kept as the fallback); host-side latents uploaded per call" % (8 * F * D),
                                      "same_bits": same, "algorithmic_bytes": chain_bytes}},
        "roofline": roofline,
    }
    if emit:
        print(json.dumps(result))
    scorer.close()
    for b in (S, err, chain_err):
        b.free()
    prim.close()
    ctx.close()
    return 0 if emit else result


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--batch", type=int, default=8192, help="candidates per GPU")
    ap.add_argument("--config", choices=("walk", "graph", "optimizer", "frame_constraints"), default="walk",
                    help="walk = BASELINE configs[1] / [3] (the headline); graph = configs[2]; optimizer = configs[4] per iteration on one GPU; "
                         "frame_constraints = candidates against constraints that walk every frame (a root trajectory + a hand position)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--single-window", action="store_true", help="time one window of K steps instead of five (the headline is then that window)")
    ap.add_argument("--host-counts", action="store_true", help="--config graph: component counts from numpy.random.multinomial on the host (one call per option) instead of the device draw")
    ap.add_argument("--no-placement-compare", action="store_true", help="skip the steps on a second library buffer and on a foreign hipMalloc after the timed region (profiling runs)")
    ap.add_argument("--no-extra-configs", action="store_true", help="do not attach the graph / optimizer configurations to the default line")
    ap.add_argument("--two-launch", action="store_true",
                    help="frames kernel and log-likelihood kernel as two launches instead of the fused step kernel")
    ap.add_argument("--frames-kernel", type=int, choices=(0, 1, 2), default=0,
                    help="MG_OPT_FRAMES_KERNEL: 0 = the library's choice, 1 = tile-major, 2 = chunk-stationary (A/B runs)")
    ap.add_argument("--no-profile-events", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--event-interval", type=int, default=0,
                    help="every n-th launch of the events pass (after the timed windows) carries HIP start/stop events attached to the dispatch (0 = every launch)")
    ap.add_argument("--ramp-steps", type=int, default=1500,
                    help="untimed steps before the warm-up that let the chip reach its steady clock (~0.13 s; reported in config)")
    ap.add_argument("--output-alloc", choices=("placed", "plain"), default="placed",
                    help="placed: the library's allocator for large outputs (mg_device_malloc_placed); plain: one hipMalloc")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: exercise the launcher, the rendezvous and the JSON contract with a stand-in step (CPU tests)")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args, sys.argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if args.config != "walk":
        if world > 1:
            raise SystemExit("--config %s is a single-GPU workload" % args.config)
        return {"graph": run_graph, "optimizer": run_optimizer, "frame_constraints": run_frame_constraints}[args.config](args)
    return run_walk(args, rank, local_rank, world)


if __name__ == "__main__":
    sys.exit(main())
