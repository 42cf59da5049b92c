#!/usr/bin/env python3
"""Headline benchmark: clips/sec of the ViT-B masked-video-autoencoder PRETRAINING step (16x3x224x224 clips, tube mask
90 %, decoder depth 4) on N MI355X GPUs of one node -- BASELINE.json's metric on BASELINE.json configs[1] (N=1) / [2] (N=8).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One step = what the reference's train_one_epoch does per batch (engine_for_pretraining.py:29-69,168-179) minus its PNG
dump: lr/wd schedule write, target build + forward + MSE (fused), zero_grad, backward, gradient all-reduce (N>1,
overlapped), global grad norm, AdamW, loss read-back + finite check (issued while the backward runs, as the drop-in
engine does; both of the reference's device syncs are kept), device sync.  Inputs are synthetic, generated straight
into the model's device input buffers before the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

# the host driver supports dmabuf IPC only: RCCL's peer mappings need this before HIP initialises (already exported on the boxes)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# HIP spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues (4 by default).  With torch.distributed's streams in the
# process the weight-gradient side stream landed on the SAME hardware queue as the compute stream (one queue id in the kernel
# trace) and nothing overlapped: 8 queues gave the data-parallel step 0.24 ms back (12.58 -> 12.35 ms at one rank; no change without
# a process group).  Read when the HIP runtime starts, hence set before torch is imported.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

STEP_FLOP_PER_CLIP = 202.3e9   # algorithmic training-step FLOPs per clip (ViT-B, dec 4, mask 0.9)
ENC_STEP_FLOP_PER_CLIP = 84.37e9  # ... of which the encoder's forward + backward
PEAK_BF16 = 2.5e15             # MI355X dense bf16 MFMA peak (guides/MI355X_MICROARCH.md)
PEAK_HBM = 8.0e12              # spec HBM3E bandwidth


class _Args:
    opt = "adamw"
    lr = 1.5e-4
    weight_decay = 0.05
    opt_eps = 1e-8
    opt_betas = (0.9, 0.95)


def cpu_baseline(seconds_budget: float):
    """The oracle (fp32 CPU restatement of the reference step, pinned to the reference's fixtures) timed on this box's
    host cores on a bounded sample: ViT-B, batch 2 (BASELINE config[0] shapes), one warm-up step + timed steps."""
    from oracle import pretrain_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    print(f"[bench] cpu baseline on {cores} host threads ...", file=sys.stderr, flush=True)
    cfg = O.VIT_B
    P = O.keyed_params(cfg, "xavier")
    x = O.keyed_clips(2, cfg)
    np.random.seed(0)
    mask = torch.from_numpy(np.stack([O.tube_mask(cfg.grid, 0.9) for _ in range(2)])).bool()
    st = O.AdamWState()
    O.train_step(x, mask, P, cfg, st)          # warm-up (allocations, thread pool)
    t0 = time.time()
    n = 0
    while True:
        O.train_step(x, mask, P, cfg, st)
        n += 1
        print(f"[bench] cpu baseline step {n}: {time.time() - t0:.1f} s", file=sys.stderr, flush=True)
        if time.time() - t0 > seconds_budget or n >= 8:
            break
    dt = time.time() - t0
    return {"value": round(2 * n / dt, 4), "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": f"oracle fp32 torch-CPU train step, ViT-B dec4, batch 2, {n} timed steps after 1 warm-up ({dt:.1f} s)"}


def calibration(dev):
    """Three fixed probes timed in THIS process before the model's first step, so that the lines of two rounds (or two boxes of the
    pool: they differ by up to 10 %) can be normalised: (a) the 256 x 256 counted-vmcnt GEMM at 8192^3 (matrix pipe + clocks under
    load), (b) a pure streaming kernel over a 1 GiB f32 buffer (HBM), (c) the persistent 128 x 128 NT GEMM at the encoder's qkv
    shape (5120 x 2304 x 768: L2 -> LDS fill path and launch ramp).  Each: >= 20 ms of the same kernel first, then 10 timed launches
    between two events on the launch stream."""
    from mofo_amd import ops
    bf, f32 = torch.bfloat16, torch.float32

    def timed(f, iters=10, warm_ms=20.0):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        f(); torch.cuda.synchronize()
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        one = max(e0.elapsed_time(e1), 1e-3)
        for _ in range(int(warm_ms / one) + 1):
            f()
        e0.record()
        for _ in range(iters):
            f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e-3

    g = torch.Generator(device=dev).manual_seed(4242)
    out = {}
    keep = {k: os.environ.get(k) for k in ("MOFO_GEMM8", "MOFO_GEMM_K2")}
    try:
        A = torch.randn(8192, 8192, device=dev, generator=g).mul_(0.1).to(bf)
        Bm = torch.randn(8192, 8192, device=dev, generator=g).mul_(0.1).to(bf)
        Cm = torch.empty(8192, 8192, dtype=bf, device=dev)
        os.environ["MOFO_GEMM8"] = "1"
        t = timed(lambda: ops.gemm(ops.GEMM_NT, ops.EPI_BF16, A, Bm, Cm))
        out["gemm8_8192_tflops"] = round(2.0 * 8192 ** 3 / t / 1e12, 1)
        del A, Bm, Cm
        os.environ["MOFO_GEMM8"] = "0"
        os.environ["MOFO_GEMM_K2"] = "0"
        X = torch.randn(5120, 768, device=dev, generator=g).mul_(0.1).to(bf)
        W = torch.randn(2304, 768, device=dev, generator=g).mul_(0.1).to(bf)
        Y = torch.empty(5120, 2304, dtype=bf, device=dev)
        t = timed(lambda: ops.gemm(ops.GEMM_NT, ops.EPI_BF16, X, W, Y))
        out["gemm128_enc_qkv_tflops"] = round(2.0 * 5120 * 2304 * 768 / t / 1e12, 1)
        del X, W, Y
        src = torch.empty(1 << 28, dtype=f32, device=dev).normal_(generator=g)
        dst = torch.empty(1 << 28, dtype=bf, device=dev)
        t = timed(lambda: ops.cast_bf16(src, dst), iters=10)
        out["stream_1gib_gbps"] = round(6.0 * (1 << 28) / t / 1e9, 1)
        del src, dst
    finally:
        for k, v in keep.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
        torch.cuda.empty_cache()
    return out


def self_launch(n: int) -> int:
    """run `python -m torch.distributed.run --nproc-per-node n bench.py <same arguments>` as a child process and relay its output"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] no launcher in the environment: starting", " ".join(cmd), file=sys.stderr, flush=True)
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)      # stderr passes through
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    return proc.returncode if proc.returncode else (0 if line is not None else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)      # >= 50 timed steps after >= 10 warm-up
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU (BASELINE configs[1]/[2]: 32)")
    ap.add_argument("--mask", choices=["tube", "bb"], default="tube",
                    help="tube: TubeMaskingGenerator 0.9 (configs 1/2); bb: MOFO motion-bounding-box masks, 75%% in-box (config 3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-kernel-events", action="store_true", help="skip the per-kernel HIP-event timing (roofline block)")
    ap.add_argument("--breakdown", action="store_true", help="print the per-kernel-class table to stderr")
    ap.add_argument("--no-encoder-step", action="store_true", help="skip the encoder-only (fwd+bwd) timing reported in config")
    ap.add_argument("--input", choices=["f32", "uint8"], default="f32",
                    help="f32: the model's input contract (normalised clips in HBM); uint8: the loader's frame stack [B,H,W,T*3], "
                         "normalised inside the gather / target kernels (side measurement)")
    ap.add_argument("--fp8", action="store_true", help="side measurement (BASELINE configs[4]): the four forward Linears of every block (qkv, proj, fc1, "
                                                       "fc2) on OCP e4m3 operands with the block-scaled MFMA; attention and the backward bf16")
    ap.add_argument("--no-calibration", action="store_true", help="skip the three fixed probes of config.calibration (~1 s)")
    ap.add_argument("--model", choices=["vitb16", "vitl32"], default="vitb16",
                    help="vitb16: the headline workload (BASELINE configs[1]/[2]); vitl32: ViT-L, 32 frames (configs[4] in bf16; side measurement)")
    args = ap.parse_args()
    if args.fp8:
        os.environ["MOFO_FP8"] = "1"

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
            # `python bench.py --gpus N` without a launcher: start the N ranks as FRESH child processes, one per GPU, the way
            # the reference is launched (PRETRAIN.md:13-16; utils.py:277-296 reads RANK / WORLD_SIZE / LOCAL_RANK).  This parent
            # has made no GPU call (nothing above touches HIP) and never exec()s: it relays the children's one JSON line and
            # their return code.
            return self_launch(args.gpus)
        raise SystemExit(f"bench.py --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (torch.distributed.run --nproc-per-node {args.gpus})")
    # one rank per GPU; MOFO_DIST_BACKEND=gloo rehearses the N > 1 path with several ranks on ONE GPU (RCCL refuses that)
    backend = os.environ.get("MOFO_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local %= max(1, torch.cuda.device_count())
    elif torch.cuda.device_count() <= local or (world > 1 and torch.cuda.device_count() < int(os.environ.get("LOCAL_WORLD_SIZE", world))):
        raise SystemExit(f"bench.py --gpus {args.gpus}: rank {rank} (local {local}) needs its own GPU over RCCL but this node shows "
                         f"{torch.cuda.device_count()} device(s); launch one rank per GPU (MOFO_DIST_BACKEND=gloo rehearses on one GPU)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    force_dp = os.environ.get("MOFO_FORCE_DP") == "1" and "MASTER_ADDR" in os.environ
    if world > 1 or force_dp:
        # RCCL prints a version banner on STDOUT when its first communicator comes up: keep stdout for the one JSON line
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        rccl_log = None
        if backend == "nccl" and world > 1 and "NCCL_DEBUG" not in os.environ:
            # what RCCL built for this job (channels, rings / trees, transport): its INIT log goes to a per-rank file, rank 0 relays a
            # digest to stderr below.  The first 8-GPU record then says how many CUs the exchange occupies beside the backward.
            rccl_log = f"/tmp/mofo_rccl_{os.getpid()}.log"
            os.environ.update(NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,GRAPH", NCCL_DEBUG_FILE=rccl_log)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", init_method="env://", world_size=world, rank=rank, device_id=dev)
            else:
                dist.init_process_group(backend, init_method="env://", world_size=world, rank=rank)
            # the communicator really spans `world` ranks: a SUM all-reduce of ones (run_mae_pretraining.py:225-227's group)
            ones = torch.ones(1, dtype=torch.float32, device=dev)
            dist.all_reduce(ones)
            rccl_ranks = int(round(float(ones.item())))
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
        if rccl_ranks != dist.get_world_size() or rccl_ranks != world:
            raise SystemExit(f"collective backend '{backend}' sees {rccl_ranks} ranks, expected {world}")
        if rccl_log and rank == 0:
            try:
                keep = [ln.strip() for ln in open(rccl_log, errors="replace")
                        if any(k in ln for k in ("Channel", "channels", "Connected", "Trees", "Ring", "nranks", "comm 0x", "Using"))]
                for ln in keep[:16]:
                    print("[bench] rccl: " + ln[-220:], file=sys.stderr, flush=True)
                print(f"[bench] rccl: {sum('Channel' in ln for ln in keep)} channel lines in the INIT log", file=sys.stderr, flush=True)
            except OSError as exc:
                print(f"[bench] rccl: no INIT log ({exc})", file=sys.stderr, flush=True)
    else:
        rccl_ranks = 1

    from mofo_amd import _lib, optim_factory, utils
    from mofo_amd import modeling_pretrain as mp
    from mofo_amd.dist import DataParallel
    from mofo_amd.masking_generator import TubeMaskingGenerator, TubeMaskingGenerator_BB

    calib = None
    if not args.no_calibration:
        try:
            calib = calibration(dev)
        except Exception as exc:      # a probe must never take the measurement down with it
            calib = {"error": f"{type(exc).__name__}: {exc}"}
        if rank == 0:
            print(f"[bench] calibration: {calib}", file=sys.stderr, flush=True)
    torch.manual_seed(0)           # identical random-init replica on every rank (DDP would broadcast rank 0's)
    if args.model == "vitl32":     # BASELINE configs[4] shapes (--fp8: its e4m3 forward Linears): 3136 tokens, 320 visible
        model = mp.pretrain_videomae_large_patch16_224(decoder_depth=4, num_frames=32).to(dev)
        B, N, n_vis, grid, step_flop, enc_flop, label = args.batch, 3136, 320, (16, 14, 14), 1104.8e9, 610.0e9, "ViT-L (enc 24x1024, dec 4x512) 32x224x224"
        args.no_cpu_baseline = True
    else:
        model = mp.pretrain_videomae_base_patch16_224(decoder_depth=4).to(dev)
        B, N, n_vis, grid, step_flop, enc_flop, label = args.batch, 1568, 160, (8, 14, 14), STEP_FLOP_PER_CLIP, ENC_STEP_FLOP_PER_CLIP, "ViT-B (enc 12x768, dec 4x384) 16x224x224"
    clips, mask_u8 = model.input_buffers(B, n_vis)
    gen = torch.Generator(device=dev).manual_seed(1000 + rank)      # seed = base + rank, run_mae_pretraining.py:166
    clips.normal_(generator=gen)
    if args.input == "uint8":
        clips, _ = model.input_buffers(B, n_vis, uint8=True)
        clips.copy_(torch.randint(0, 256, clips.shape, dtype=torch.uint8, device=dev, generator=gen))
    np.random.seed(rank)
    if args.mask == "tube":
        mgen = TubeMaskingGenerator(grid, 0.9)
        masks = [mgen() for _ in range(B)]
    else:   # per-clip boxes x1,y1 ~ U{0..160}, w,h ~ U{32..160} clipped to 224, replicated over the 16 frames
        bgen = TubeMaskingGenerator_BB(grid, 0.9, 0.75)
        masks = []
        for _ in range(B):
            x1, y1 = np.random.randint(0, 161, 2)
            w_, h_ = np.random.randint(32, 161, 2)
            masks.append(bgen(np.tile(np.array([x1, y1, min(224, x1 + w_), min(224, y1 + h_)]), (16, 1))))
    mask_u8.copy_(torch.from_numpy(np.stack(masks).astype(np.uint8)))
    mask_dev = mask_u8            # resident in the model's input buffer, like the clips (no per-step staging copy)
    _Args.lr = 1.5e-4 * (B * world) / 256                          # run_mae_pretraining.py:217
    opt = optim_factory.create_optimizer(_Args, model)
    wrapped = DataParallel(model) if (world > 1 or force_dp) else model
    scaler = utils.NativeScalerWithGradNormCount()
    total_steps = args.steps + args.warmup
    lr_sched = utils_quiet(utils.cosine_scheduler, _Args.lr, 1e-5, 1, total_steps + 1, warmup_epochs=0)
    wd_sched = utils_quiet(utils.cosine_scheduler, 0.05, 0.05, 1, total_steps + 1)

    def step(it):
        for g in opt.param_groups:
            g["lr"] = lr_sched[it] * g["lr_scale"]
            if g["weight_decay"] > 0:
                g["weight_decay"] = wd_sched[it]
        loss = wrapped.forward_loss(clips, mask_dev, True)
        opt.zero_grad()
        scaler(loss, opt, clip_grad=None)                           # backward + grad-norm + AdamW enqueued ...
        lv = loss.item()                                            # ... then the loss read (engine_for_pretraining.py:69, sync #1)
        if not math.isfinite(lv):
            raise SystemExit(f"loss is {lv}")
        torch.cuda.synchronize()                                    # engine_for_pretraining.py:179 (sync #2)
        return lv

    # Per-kernel HIP events cost ~5 us of host time each (two per launch, ~500 launches per step), so bracketing EVERY
    # launch would make the timed region host-bound.  The warm-up steps are therefore fully instrumented (per-class
    # table, picks the dominant kernel class); in the timed region only that dominant class is bracketed.
    full = None
    n_instr = 0
    t_gpu0 = time.perf_counter()
    for it in range(args.warmup):
        if not args.no_kernel_events and (it >= min(2, args.warmup - 1)) and full is None:
            full = _lib.EventProfiler()          # step 0 is cold (module load, first touch, list recording) and step 1 is
                                                 # the first replay: neither is profiled when there are >= 3 warm-up steps
            _lib.PROFILER = full
        n_instr += full is not None
        step(it)
    _lib.PROFILER = None
    model.check_status()
    prof = None
    if full is not None and full.records:
        fs = full.summary()
        dom_key = max(fs, key=lambda k: fs[k]["ms"])
        prof = _lib.EventProfiler(only=dom_key)      # switched on right before the timed region (not during the checks / A/Bs below)
    # Self-check of the overlapped gradient exchange (first N > 1 RCCL run: the only place the side-stream hand-off of
    # runtime._seg_now can be value-checked): the all-reduced gradients of the production path against a fully synchronised
    # all-reduce of a saved copy, same inputs, no optimizer step in between (dist.GradSync.value_check).
    ar_check = None
    if world > 1 or force_dp:
        def _bwd_only():
            loss_ = wrapped.forward_loss(clips, mask_dev, True)
            opt.zero_grad()
            loss_.backward()
        try:
            # --fp8: the check freezes the delayed activation scales (dist.GradSync.value_check), so its two forwards quantise alike;
            # what remains is e4m3 rounding that reacts to last-bit differences of the inputs: 2e-2 (a range exchanged too early is O(1))
            ar_tol = 2e-2 if (args.fp8 or os.environ.get("MOFO_GRAD_BF16") == "1") else 1e-4
            ar_check = wrapped.sync.value_check(_bwd_only, tol=ar_tol)
        except Exception as exc:      # the check must never take the scaling measurement down with it: report, go on timing
            ar_check = {"ok": False, "max_rel": float("nan"), "ranges": 0, "worst_range": None, "error": f"{type(exc).__name__}: {exc}"}
        opt.zero_grad()
        if rank == 0:
            print(f"[bench] all-reduce value check: {'ok' if ar_check['ok'] else 'FAIL'} (max relative error {ar_check['max_rel']:.2e} over "
                  f"{ar_check['ranges']} ranges, worst {ar_check['worst_range']}{', ' + ar_check['error'] if 'error' in ar_check else ''})",
                  file=sys.stderr, flush=True)
    # Route A/B under the LIVE exchange (first multi-GPU runs: nothing below was ever measured beside an N-rank RCCL all-reduce, whose
    # channel workgroups stay resident on tens of CUs for most of the backward).  Two choices assume they own every CU: the
    # one-128x128-tile-per-CU GEMM (gemm_k2: 240 blocks x 128 KiB of LDS; MOFO_GEMM_K2=0 routes its shapes back to the co-resident
    # forms) and the weight-gradient launches on the main stream (MOFO_WGRAD_STREAM=side moves them beside the chain).  Five untimed
    # steps of each of the four combinations, max over ranks, the fastest is kept for the timed region; all four are reported.
    dp_ab = None
    if (world > 1 or force_dp) and os.environ.get("MOFO_DP_ROUTE_AB", "1") == "1":
        user = {k: os.environ.get(k) for k in ("MOFO_GEMM_K2", "MOFO_WGRAD_STREAM")}
        # (default streams: every weight-gradient launch on the main stream)
        routes = [("k2 on, default streams", {}), ("k2 off, default streams", {"MOFO_GEMM_K2": "0"}),
                  ("k2 on, all weight gradients on the side stream", {"MOFO_WGRAD_STREAM": "side"}),
                  ("k2 off, all weight gradients on the side stream", {"MOFO_GEMM_K2": "0", "MOFO_WGRAD_STREAM": "side"})]
        if any(v is not None for v in user.values()):
            routes = [("the caller's routes", {k: v for k, v in user.items() if v is not None})]   # the caller fixed them: only the plans are timed
        # ... and the encoder's BUCKET PLAN (round-5 review, item 6): shrinking buckets 6, 3, 2, 1 with three-block weight-gradient groups
        # (the smallest all-reduce is the one left exposed) against two buckets 7, 5 whose groups fill whole rounds of the ring kernel
        # (fewer, larger launches; 138 MB exposed instead of 33).  Which wins depends on what the live exchange costs the backward.
        rt_ = model.runtime()
        plans = [("buckets 6,3,2,1 / groups of 3", None, min(3, rt_._enc_group_cap))]
        if args.model == "vitb16" and rt_._enc_group_cap >= 7 and not os.environ.get("MOFO_ENC_BUCKETS") and not os.environ.get("MOFO_WGRAD_BLOCKS"):
            plans.append(("buckets 7,5 / groups of 7 and 5", [7, 5], 7))
        combos = [(f"{rn}; {pn}", env, (pb, pg)) for pn, pb, pg in plans for rn, env in routes]
        ms, exposed = {}, {}

        def _use(env, plan):
            for k in user:
                os.environ.pop(k, None)
            os.environ.update(env)
            rt_.set_enc_plan(plan[0], plan[1])   # re-plans the segments and drops the recorded lists (MOFO_WGRAD_STREAM is read while recording)

        opt.measure_exposed = True
        for name, env, plan in combos:
            _use(env, plan)
            step(args.warmup - 1 if args.warmup else 0)      # records the lists again (untimed)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            opt.exposed_events.clear()
            ta = time.perf_counter()
            for _ in range(5):
                step(args.warmup - 1 if args.warmup else 0)
            torch.cuda.synchronize()
            ex = [sum(a.elapsed_time(b) for a, b in ev) for ev in opt.exposed_events]
            tb = torch.tensor([(time.perf_counter() - ta) / 5 * 1e3, float(np.median(ex)) if ex else 0.0], dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(tb, op=dist.ReduceOp.MAX)   # every rank sees the same numbers -> the same choice
            ms[name], exposed[name] = round(float(tb[0].item()), 3), round(float(tb[1].item()), 3)
        opt.measure_exposed = False
        opt.exposed_events.clear()
        if ms:
            best = min(ms, key=ms.get)
            # every rank must have chosen the same plan (the ranges of the exchange follow from it): checked, not assumed
            pick = torch.tensor([float(list(ms).index(best))], dtype=torch.float64, device=dev)
            lo_, hi_ = pick.clone(), pick.clone()
            if world > 1:
                dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
                dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
            if float(lo_.item()) != float(hi_.item()):
                raise SystemExit("data-parallel route A/B: the ranks disagree on the fastest plan")
            sel = next(c for c in combos if c[0] == best)
            _use(sel[1], sel[2])
            step(args.warmup - 1 if args.warmup else 0)
            dp_ab = {"ms_per_step": ms, "exposed_allreduce_ms": exposed, "chosen": best, "plan_agreed": True,
                     "enc_buckets": rt_.enc_buckets(), "enc_group": rt_.wgrad_blocks}
            if rank == 0:
                print("[bench] data-parallel route / plan A/B (ms per step | exposed all-reduce ms, 5 steps each, max over ranks): " +
                      "; ".join(f"{k}: {v} | {exposed[k]}" for k, v in ms.items()) + f" -> {best}", file=sys.stderr, flush=True)
    # Route A/B at N = 1 (round-5 review, item 2): the routes this round added -- the decoder's weight gradients as ONE sliced launch,
    # 384-row ring tiles -- against the round-5 routes (a grouped launch per decoder block with split reductions, 256-row ring tiles),
    # 5 untimed steps each in THIS process on THIS box; the faster one is kept for the timed region, both are reported.
    route_ab = None
    if world == 1 and not force_dp and os.environ.get("MOFO_ROUTE_AB", "1") == "1" and args.model == "vitb16":
        keys = ("MOFO_WGRAD_SLICED", "MOFO_GEMM_R4")
        if all(os.environ.get(k) is None for k in keys):
            combos = [("round-6 routes (sliced decoder weight gradients, 384 x 128 ring)", {}),
                      ("round-5 routes (MOFO_WGRAD_SLICED=0 MOFO_GEMM_R4=0)", {"MOFO_WGRAD_SLICED": "0", "MOFO_GEMM_R4": "0"})]
            ms = {}

            def _use1(env):
                for k in keys:
                    os.environ.pop(k, None)
                os.environ.update(env)
                model.runtime().invalidate_lists()

            for rnd in range(2):                     # interleaved: A B A B
                for name, env in combos:
                    _use1(env)
                    step(args.warmup - 1 if args.warmup else 0)
                    torch.cuda.synchronize()
                    ta = time.perf_counter()
                    for _ in range(5):
                        step(args.warmup - 1 if args.warmup else 0)
                    torch.cuda.synchronize()
                    ms.setdefault(name, []).append((time.perf_counter() - ta) / 5 * 1e3)
            ms = {k: round(min(v), 3) for k, v in ms.items()}
            best = min(ms, key=ms.get)
            _use1(dict(combos)[best])
            step(args.warmup - 1 if args.warmup else 0)
            route_ab = {"ms_per_step": ms, "chosen": best}
            print("[bench] route A/B (ms per step, best of 2 x 5 steps): " + "; ".join(f"{k}: {v}" for k, v in ms.items()) + f" -> {best}",
                  file=sys.stderr, flush=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    if world > 1 or force_dp:
        opt.measure_exposed = True           # event pairs around each wait for a gradient range's all-reduce (FusedAdamW.step)
    _lib.PROFILER = prof                     # the dominant class only, in the timed region only
    t0 = time.perf_counter()
    last = None
    step_ends = []
    # every bracketed launch costs ~10 us of host time (two events): a class of 50 launches per step (the dgrad GEMMs) bracketed on every
    # step made the timed region 2 % slower than the same route unbracketed (11.08 vs 10.86 ms).  The dominant class is therefore
    # bracketed on every `stride`-th timed step only, so that the brackets cost <= ~8 launches' worth per step on average; its average
    # launch duration still comes from the timed region (>= 1 step in `stride`, hundreds of launches).
    stride = 1
    if prof is not None and full is not None and n_instr:
        per_step = fs[dom_key]["launches"] / n_instr
        stride = max(1, int(math.ceil(per_step / 8.0)))
    for it in range(args.steps):
        if prof is not None:
            prof.enabled = (it % stride == 0)
        last = step(args.warmup + it)        # ends with the reference's device synchronize: the host clock sees whole steps
        step_ends.append(time.perf_counter())
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    _lib.PROFILER = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    clips_per_s = B * world * args.steps / dt
    gpu_phase_s = time.perf_counter() - t_gpu0     # warm-up + instrumented + timed steps: what a GPU-busy sampler can see of this run
    per_step_ms = np.diff(np.array([t0] + step_ends)) * 1e3
    exposed_ms, exposed_per_rank = None, None
    if getattr(opt, "measure_exposed", False) and opt.exposed_events:
        # time the compute stream spent waiting for gradient exchanges that had not finished when the optimizer reached them
        ex = [sum(a.elapsed_time(b) for a, b in ev) for ev in opt.exposed_events]
        exposed_ms = float(np.median(ex))
        exposed_per_rank = [round(exposed_ms, 3)]
        if world > 1:
            t = torch.zeros(world, dtype=torch.float64, device=dev)
            t[rank] = exposed_ms
            dist.all_reduce(t)                       # every rank's median, rank by rank
            exposed_per_rank = [round(float(x), 3) for x in t.tolist()]
            exposed_ms = max(exposed_per_rank)
            if rank == 0:
                print("[bench] exposed all-reduce wait per rank (ms, median over steps): " + " ".join(f"{x:.3f}" for x in exposed_per_rank),
                      file=sys.stderr, flush=True)

    out = {"metric": "clips/sec (16x3x224x224, mask 90%) ViT-B pretrain step" if args.model == "vitb16" else "clips/sec (32x3x224x224, mask 90%) ViT-L pretrain step", "value": round(clips_per_s, 2), "unit": "clips/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "bf16" if not args.fp8 else "fp8 e4m3 (forward qkv / proj / fc1 / fc2 GEMMs) + bf16", "data": "synthetic",
           "config": {"workload": label + " " + ("tube" if args.mask == "tube" else "motion-BB") + " mask 0.9, per-GPU batch %d, bf16 MFMA + fp32 accumulate / "
                                  "encoder residual stream / optimizer (decoder residual stream bf16), full train step (target+fwd+loss+bwd+grad-norm+AdamW)" % B,
                      "global_batch": B * world, "per_gpu_batch": B, "parallelism": f"dp{world}", "mask": args.mask, "input": args.input, "final_loss": round(last, 5),
                      "step_mfma_frac": round(clips_per_s / world * step_flop / PEAK_BF16, 4),
                      "ms_per_step_median": round(float(np.median(per_step_ms)), 3), "rccl_ranks": rccl_ranks, "backend": backend if (world > 1 or force_dp) else None,
                      "exposed_allreduce_ms": None if exposed_ms is None else round(exposed_ms, 3),
                      "exposed_allreduce_ms_per_rank": exposed_per_rank,
                      "allreduce_value_check": None if ar_check is None else ("ok" if ar_check["ok"] else ("error: " + ar_check["error"] if "error" in ar_check else "fail")),
                      "allreduce_value_check_max_rel": None if ar_check is None or ar_check["max_rel"] != ar_check["max_rel"] else float("%.3e" % ar_check["max_rel"]),
                      "dp_route_ab": dp_ab, "route_ab": route_ab, "calibration": calib, "grad_transport": ("bf16" if os.environ.get("MOFO_GRAD_BF16") == "1" else "f32") if (world > 1 or force_dp) else None,
                      "hbm_peak_gb": round(torch.cuda.max_memory_allocated() / 1e9, 2),
                      "gpu_phase_s": round(gpu_phase_s, 2), "timed_s": round(dt, 3)}}

    if prof is not None:
        summ = prof.summary()
        summ = {k: v for k, v in summ.items()}
        warm = full.summary()
        names = {("gemm", 0, 0): "gemm_nt_bf16", ("gemm", 0, 1): "gemm_nt_bias_gelu", ("gemm", 0, 2): "gemm_nt_resid_f32",
                 ("gemm", 0, 3): "gemm_nt_pos_f32", ("gemm", 1, 0): "gemm_nn_bf16", ("gemm", 1, 4): "gemm_nn_dgelu",
                 ("gemm", 2, 5): "gemm_tn_wgrad_f32", ("gemm", 0, 6): "gemm_nt_resid_bf16", ("gemm", 0, 7): "gemm_nt_pos_bf16"}
        mfma_keys = [k for k in summ if k[0] in ("gemm", "attn_fwd", "attn_bwd", "attn_bwd_dq", "attn_bwd_dkv", "attn_bwd_1p")]
        total_ms = sum(v["ms"] for v in warm.values()) / max(1, n_instr) * args.steps
        dom = max(summ, key=lambda k: summ[k]["ms"])
        d = summ[dom]
        is_mfma = dom in mfma_keys
        ach = d["work"] / (d["ms"] * 1e-3)
        out["roofline"] = {"kernel": names.get(dom, "_".join(str(x) for x in dom)), "bound": "mfma" if is_mfma else "hbm",
                           "achieved": round(ach / (1e12 if is_mfma else 1e9), 2), "peak": (PEAK_BF16 / 1e12) if is_mfma else (PEAK_HBM / 1e9),
                           "unit": "TFLOP/s" if is_mfma else "GB/s", "frac": round(ach / (PEAK_BF16 if is_mfma else PEAK_HBM), 4),
                           "traffic": None, "launches": d["launches"], "bracketed_every_nth_step": stride, "avg_us": round(1e3 * d["ms"] / d["launches"], 2),
                           "share_of_kernel_time": round(d["ms"] * args.steps / max(1, len(range(0, args.steps, stride))) / total_ms, 3)}
        out["roofline"]["dropped_launches"] = d["dropped"]
        # algorithmic bytes of one launch, from the launches' own shapes (every operand read once, every output written
        # once; ops._gemm_args) -- not a constant
        if d.get("bytes"):
            out["roofline"]["algorithmic_bytes"] = round(d["bytes"] / d["launches"])
        # HBM-side traffic of the dominant class: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over THIS command
        # (tools/pmc_step.sh; FETCH_SIZE doubled as the guide prescribes for gfx950).  Counters cannot be read from inside the
        # process, so this is a RECORDED figure: it is reported only while the kernel sources are byte-identical to the ones
        # the profile was taken with (sha256 over mofo_amd/csrc), null otherwise.
        import glob
        recs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_step_traffic.json")))
        tpath = recs[-1] if recs else ""               # the newest round's record
        tname = "profiles/" + os.path.basename(tpath)
        out["roofline"]["traffic_source"] = None
        if args.model == "vitb16" and B == 32 and os.path.exists(tpath):   # recorded for the headline workload only
            try:
                prof_rec = json.load(open(tpath))
                rec = prof_rec.get("classes", {}).get(out["roofline"]["kernel"])
                if prof_rec.get("_meta", {}).get("csrc_sha256") != csrc_sha():
                    out["roofline"]["traffic_source"] = "stale: %s was recorded for other kernel sources" % tname
                elif rec:
                    out["roofline"]["traffic"] = round((rec["fetch_MB_per_launch"] + rec["write_MB_per_launch"]) * 1e6)
                    out["roofline"]["traffic_source"] = "recorded: %s (csrc sha256 %s)" % (tname, csrc_sha()[:12])
            except Exception:
                pass
        # the five largest kernel classes of the instrumented warm-up steps (ms per step, rate as a fraction of the class's roofline):
        # `roofline` above follows whichever class is the largest on THIS box, this list lets two lines be compared class by class
        top = sorted(warm.items(), key=lambda kv: -kv[1]["ms"])[:5]
        out["config"]["kernel_classes"] = [
            {"class": names.get(k, "_".join(str(x) for x in k)), "launches_per_step": v["launches"] // max(1, n_instr),
             "ms_per_step": round(v["ms"] / max(1, n_instr), 3),
             "frac": round(v["work"] / (v["ms"] * 1e-3) / (PEAK_BF16 if k in mfma_keys or k[0] in ("gemm", "attn_fwd", "attn_bwd", "attn_bwd_dq", "attn_bwd_dkv") else PEAK_HBM), 4)}
            for k, v in top]
        gem = [warm[k] for k in warm if k[0] == "gemm"]
        if gem:
            out["roofline"]["all_gemm_tflops"] = round(sum(g["work"] for g in gem) / sum(g["ms"] for g in gem) / 1e9, 2)
        if args.breakdown and rank == 0:
            print(f"{'kernel class':28s} {'launches':>8s} {'ms/step':>9s} {'share':>6s} {'rate':>12s}", file=sys.stderr)
            wtot = sum(v["ms"] for v in warm.values())
            print("(per-class table from the fully instrumented warm-up steps)", file=sys.stderr)
            for k, v in sorted(warm.items(), key=lambda kv: -kv[1]["ms"]):
                rate = v["work"] / (v["ms"] * 1e-3)
                mf = k[0] in ("gemm", "attn_fwd", "attn_bwd", "attn_bwd_dq", "attn_bwd_dkv", "attn_bwd_1p")
                print(f"{names.get(k, '_'.join(str(x) for x in k)):28s} {v['launches'] // max(1, n_instr):8d} {v['ms'] / max(1, n_instr):9.3f} "
                      f"{v['ms'] / wtot:6.1%} {rate / (1e12 if mf else 1e9):9.1f} {'TF/s' if mf else 'GB/s'}  (max {v['max_ms']:.3f} ms)", file=sys.stderr)
            print(f"sum of kernel time {wtot / max(1, n_instr):.3f} ms/step (warm-up) vs timed wall {1e3 * dt / args.steps:.3f} ms/step", file=sys.stderr)

    # Encoder-only step (SURVEY.md 8d / north star: "roofline fraction of the ViT-B encoder step"): patch gather + tubelet
    # embed + 12 blocks + norm forward, and the backward of exactly that (dgrad chain + grouped weight gradients incl. the
    # patch-embed wgrad) from the d(encoder output) buffer the last full step left behind.  84.37 GFLOP per clip.
    if world == 1 and not force_dp and not args.no_encoder_step:
        rt = model.runtime()
        w = next(iter(rt._ws.values()))

        def enc_step():
            rt.cached(w, "bench_enc_fwd", lambda: rt.encoder_forward(w))
            rt._accumulate = False
            rt.cached(w, "bench_enc_bwd", lambda: rt.encoder_backward(w, w.d_encout))
        for _ in range(3):
            enc_step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n_enc = 20
        for _ in range(n_enc):
            enc_step()
        torch.cuda.synchronize()
        enc_ms = 1e3 * (time.perf_counter() - t1) / n_enc
        if args.breakdown and rank == 0:
            ep = _lib.EventProfiler()
            _lib.PROFILER = ep
            for _ in range(3):
                enc_step()
            _lib.PROFILER = None
            es = ep.summary()
            print(f"encoder-only step {enc_ms:.3f} ms; per class (one stream, 3 instrumented steps):", file=sys.stderr)
            for k, v in sorted(es.items(), key=lambda kv: -kv[1]["ms"]):
                print(f"  {'_'.join(str(x) for x in k):24s} {v['launches'] // 3:5d} {v['ms'] / 3:8.3f} ms", file=sys.stderr)
            print(f"  sum {sum(v['ms'] for v in es.values()) / 3:.3f} ms", file=sys.stderr)
        out["config"]["encoder_step"] = {"ms": round(enc_ms, 3), "clips_per_s": round(B / enc_ms * 1e3, 1),
                                         "mfma_frac": round(B / (enc_ms * 1e-3) * enc_flop / PEAK_BF16, 4)}

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1 or force_dp:
        dist.destroy_process_group()


def csrc_sha() -> str:
    """sha256 over the kernel sources (mofo_amd/csrc, sorted by name): ties a recorded PMC profile to the code it measured"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "mofo_amd", "csrc")
    for f in sorted(os.listdir(d)):
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def host_cores() -> int:
    """threads this process may really use: affinity mask, cgroup cpu quota, and the GPU box's 16-core share per GPU"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 16))


def utils_quiet(fn, *a, **k):
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


if __name__ == "__main__":
    sys.exit(main() or 0)
