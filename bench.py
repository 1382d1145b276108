#!/usr/bin/env python3
"""Headline benchmark: poses/sec of one train_1.py step (zero_grad, forward, MSE, backward,
[gradient all-reduce], AdamW) of the 17-joint lifter at batch 4096 per GPU.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        --master-port P bench.py --gpus N --steps K --warmup W

GEMM arithmetic (--dtype): "f16x3" (default: every operand tensor written by its producer as two fp16 planes,
S x = h + l/2048, three fp16 MFMAs per product on two fp32 accumulators -- fp32-grade results, meets the 1e-3 mm
parity gate), "bf16x6" (fp32 operands split into three bf16 pieces inside the GEMM, six MFMAs -- also fp32-grade),
"fp32" (exact v_mfma_f32_32x32x2_f32) or "bf16" (operands rounded to bf16, ~1 mm: never the parity-gated number).
The modes not chosen are measured beside the headline (other_modes).

One JSON line on rank 0 (contract in the task statement).  Also in that line:
  roofline      the dominant kernel by time, the backward dual launch (dX = dz W and dW = dz^T a of one
                layer, 2 x 8.59 GFLOP): ALGORITHMIC FLOPs per launch / mean launch duration, measured
                with HIP events recorded on the launch stream around every such launch INSIDE the
                timed region (pl_prof_enable), against the gfx950 dense MFMA peak of the arithmetic
                type (fp32 matrix 157.3 TF, bf16 / fp16 2,516 TF); roofline_forward_gemm: the same for the
                forward kernel.  f16x3 issues 3 and bf16x6 6 MFMA FLOPs per algorithmic FLOP (mfma_issue_frac).
                traffic = PMC HBM-side bytes per launch (profiles/traffic.json).
  cpu_baseline  the restated reference step (oracle/torch_twin.py: stock PyTorch CPU eager, the
                ATen kernels the reference dispatches to) timed on this node's host cores as BASELINE.md
                section 4 plans it: 20 warm-up steps, then the MEDIAN of >= 50 timed steps, at B = 4096 (the
                value) and B = 64 (batch_64) -- rank 0, N=1 only.  A reported baseline, not the target.
  batch_64      the HIP step's latency at the reference's own batch size (config 0), same protocol, issued as one hipGraph
                replay and as an eager train_step (the lower of the two; both are in the entry).
  config.launch how the timed steps were issued at N = 1: "eager" (one call per kernel) or "hipgraph" (one replay of the
                captured step, train.GraphedTrainStep) -- --launch auto times a few untimed steps of each during warm-up
                and takes the faster (a busy host core makes the eager loop launch-bound: 0.95 instead of 0.61 ms per
                step on one box of round 3).  Under hipgraph the roofline's HIP events bracket eager steps right behind
                the timed region (events cannot be read out of a replayed graph); N > 1 is always eager.
  parity        eval-forward MPJPE (mm) of the HIP path against the numpy oracle on the bench
                batch, in the same run (gate 1e-3 mm, BASELINE.json).
  other_modes   this library's other arithmetic modes and stock PyTorch-ROCm eager of the same module
                (fp32 and bf16 autocast) on the same GPU, same batch, same step.
Inputs are synthetic H3.6M-shaped batches resident in HBM before the timed region starts.
"""
import argparse
import ctypes
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the pool's host driver only supports dmabuf IPC; RCCL across processes fails without this (set before torch loads)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

FLOP_PER_POSE = 25_618_432          # 2 x 12,809,216 MAC, GEMMs only, fwd+bwd
PEAK_F32_MATRIX_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MATRIX_TFLOPS = 2516.0    # dense bf16 MFMA peak (the 2:1-sparsity figure is NOT used)
BATCH = 4096
PROF_EVERY = 7    # HIP events around every 7th GEMM launch (coprime with the launches per step; see poselift.h)
PROF_SPARSE = 31  # ... every 31st once the timed region is long enough for >= 40 samples: an event pair is two barrier packets
                  # on the stream, and at every 7th launch the sampling itself cost the step 1 % (0.634 vs 0.627 ms, same box)


PROF_NOW = [PROF_EVERY]


def prof_every(steps):
    PROF_NOW[0] = PROF_SPARSE if steps * 8 >= 40 * PROF_SPARSE else PROF_EVERY
    return PROF_NOW[0]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=BATCH, help="per-GPU batch (default: BASELINE config 4096)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="do not bracket GEMM launches with HIP events")
    ap.add_argument("--dtype", choices=["fp32", "bf16", "bf16x6", "f16x3"], default="f16x3",
                    help="GEMM arithmetic of the headline run.  f16x3 (default), bf16x6 and fp32 meet the 1e-3 mm "
                         "parity gate; bf16 does not (about 1 mm)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N>1 only: one whole-arena all-reduce after backward instead of buckets overlapped with it")
    ap.add_argument("--sync-bn", action="store_true",
                    help="N>1 only: BatchNorm statistics over the global batch (10 small all-gathers per step); "
                         "off by default = statistics per shard, the DDP convention")
    ap.add_argument("--no-extras", action="store_true", help="skip the bf16-mode and PyTorch-eager side measurements")
    ap.add_argument("--launch", choices=["auto", "eager", "hipgraph"], default="auto",
                    help="how the timed steps are issued at N=1: one Python call per launch (eager), one hipGraph replay per "
                         "step (GraphedTrainStep), or whichever a short calibration during warm-up finds faster (auto: a slow "
                         "or busy host core makes the eager loop launch-bound)")
    ap.add_argument("--workload", choices=["lifter", "cycle"], default="lifter",
                    help="lifter (default): BASELINE configs[1]/[2], the headline.  cycle: BASELINE configs[4], one phase5 "
                         "cycle step (Model_2D + Model_3D on 256x256 frames, lifter x2, projector, TriangleLoss, Adam x4) at "
                         "128 frames per GPU (1024 over 8 GPUs); its own JSON line (frames/s)")
    ap.add_argument("--flip", action="store_true", help="cycle workload: the training-mode Flip branch (train_5 copy.py:174-199)")
    a = ap.parse_args()
    if a.workload == "cycle" and a.batch == BATCH:
        a.batch = 128                                       # configs[4]: batch 1024 over 8 GPUs
    if a.workload == "cycle" and a.steps == 200 and a.warmup == 20:
        a.steps, a.warmup = 10, 3                           # a cycle step is ~0.1 s: defaults that finish in minutes
    return a


def self_launch(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks here, one process per GPU, and relay rank 0's
    JSON line.  The parent never touches the GPU (torch.cuda.device_count() does not initialise it on this image);
    any rank that fails takes the job down with a non-zero exit -- an `n_gpus: 1` line for a `--gpus 8` request
    can not happen."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    backend = os.environ.get("POSELIFT_DIST_BACKEND")
    if have < a.gpus and backend != "gloo":
        raise SystemExit(f"bench.py --gpus {a.gpus}: only {have} GPU(s) visible (POSELIFT_DIST_BACKEND=gloo rehearses "
                         f"more ranks than devices, all folded onto the visible ones)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    import tempfile
    procs = []
    with tempfile.TemporaryFile("w+") as out0:
        for r in range(a.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=out0 if r == 0 else subprocess.DEVNULL))
        failed = False
        while any(p.poll() is None for p in procs):
            if any(p.poll() not in (None, 0) for p in procs):         # one rank died: the others would wait in a collective
                failed = True
                for p in procs:
                    if p.poll() is None:
                        p.kill()
                break
            time.sleep(0.2)
        rcs = [p.wait() for p in procs]
        out0.seek(0)
        line = out0.read()
    if failed or any(rcs):
        raise SystemExit(f"bench.py --gpus {a.gpus}: a rank failed (exit codes {rcs}); no result line")
    # exactly the result line (the gloo rehearsal backend prints a connection banner on stdout)
    sys.stdout.write("".join(ln + "\n" for ln in line.splitlines() if ln.startswith("{")))
    sys.stdout.flush()


def host_cores():
    """CPU share of this process: min(affinity, cgroup quota, 16).  A GPU box gives one GPU's
    share of the host (16 cores) through a quota, not through affinity; oversubscribing it
    (256 threads on a 16-core quota) made the baseline 100x slower than it is."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, min(n, 16))


def _median(v):
    v = sorted(v)
    return v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])


def cpu_baseline(batch, warm=20, steps=50):
    """The restated reference step on the host cores, BASELINE.md section 4's protocol: `warm` warm-up steps, the
    median of `steps` timed steps, at B = batch (the value) and at the reference's own B = 64."""
    import platform
    import torch
    from oracle.torch_twin import TwinLifter, twin_train_step
    cores = host_cores()
    torch.set_num_threads(cores)
    pkg = importlib.import_module("3d_poseestimation_amd")

    def measure(b):
        torch.manual_seed(0)
        model = TwinLifter(34, 51, linear_size=1024, num_stage=2, p_dropout=0.5, BN=True).train()
        opt = torch.optim.AdamW(model.parameters(), lr=1e-4)
        x, y = pkg.synth.synthetic_batch(b, 1234)
        full, fb = [], []
        for i in range(warm + steps):
            t0 = time.perf_counter()
            twin_train_step(model, opt, x, y)
            if i >= warm:
                full.append(time.perf_counter() - t0)
        for i in range(3 + max(5, steps // 5)):                 # forward + backward only, reported beside it
            t0 = time.perf_counter()
            opt.zero_grad()
            torch.nn.functional.mse_loss(model(x).reshape(y.shape), y).backward()
            if i >= 3:
                fb.append(time.perf_counter() - t0)
        return _median(full), _median(fb)

    t_full, t_fb = measure(batch)
    t64, t64_fb = measure(64)
    cpu = platform.processor() or platform.machine()
    try:
        cpu = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][0]
    except (OSError, IndexError):
        pass
    return {"value": round(batch / t_full, 1), "unit": "poses/s", "cores": cores, "kind": "port",
            "sample": f"median of {steps} train_1.py-style steps (zero_grad+fwd+MSE+bwd+AdamW) of the stock-PyTorch "
                      f"eager twin at batch {batch}, fp32, {cores} threads, after {warm} warm-up steps",
            "ms_per_step": round(1e3 * t_full, 2), "fwd_bwd_only_poses_per_s": round(batch / t_fb, 1),
            "batch_64": {"poses_per_s": round(64 / t64, 1), "ms_per_step": round(1e3 * t64, 3),
                         "fwd_bwd_only_poses_per_s": round(64 / t64_fb, 1)},
            "cpu_model": cpu, "torch": torch.__version__}


def read_rooflines(pkg, L, dtype, one, traffic):
    """(dual-launch roofline, forward-GEMM roofline) from the HIP-event records of pl_prof.
    achieved = ALGORITHMIC FLOPs per launch / mean launch duration; peak = dense MFMA peak of the
    arithmetic type (bf16x6 issues 6 bf16 MFMA FLOPs per algorithmic FLOP: `mfma_issue_frac`)."""
    peak = PEAK_F32_MATRIX_TFLOPS if dtype == "fp32" else PEAK_BF16_MATRIX_TFLOPS
    redundancy = {"bf16x6": 6, "f16x3": 3}.get(dtype, 1)

    def read(lo, hi, kernel, tkey):
        ms, n_l, fl = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
        pkg._lib.check(L.pl_prof_read(lo, hi, ctypes.byref(ms), ctypes.byref(n_l), ctypes.byref(fl)), "pl_prof_read")
        if not n_l.value:
            return None
        avg_ms = ms.value / n_l.value
        ach = fl.value / n_l.value / (avg_ms * 1e-3) / 1e12
        tr = None
        if traffic:
            tr = (traffic.get("per_dtype", {}).get(dtype, {}).get("dual" if "dual" in tkey else "forward", {})
                  .get("hbm_bytes_per_launch"))
            if tr is None and dtype == "bf16x6":
                tr = traffic.get(tkey)                         # round-1 figures of the bf16x6 kernels
        r = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
             "frac": round(ach / peak, 4), "traffic": tr,
             "kernel": kernel, "flop_per_launch": fl.value / n_l.value,
             "avg_launch_us": round(avg_ms * 1e3, 2), "launches_timed": n_l.value,
             "sampling": f"HIP events around every {PROF_NOW[0]}th GEMM launch of the timed region"}
        if redundancy > 1:
            r["mfma_issue_frac"] = round(redundancy * ach / peak, 4)
        return r
    # dominant kernel by time: the backward dual launch (dX = dz W and dW = dz^T a of one layer,
    # 2 x 8.59 GFLOP); the forward single-GEMM kernel is reported beside it
    kd = {"bf16x6": "gemm_x6_planes_dual_kernel", "f16x3": "planes_gemm_chain_kernel<f16x3>"}.get(
        dtype, "gemm_f32_dual_kernel<%s>" % dtype)
    ks = {"bf16x6": "gemm_x6_planes_kernel<NT>", "f16x3": "planes_gemm_wide_kernel<f16x3> (NT, k-tiles staged in pairs)"}.get(
        dtype, "gemm_f32_kernel<NT,%s>" % dtype)
    dual = read(1.9 * one, 2.1 * one, kd + " (4096x1024x1024 dX + 1024x1024x4096 dW in one launch -- f16x3 / bf16: one workgroup "
                "per CU runs its dX tile, then a dW item, the operand stream running on; 4 launches/step)",
                "gemm_f32_dual_hbm_bytes_per_launch")
    single = read(0.99 * one, 1.01 * one, ks + " (4096x1024x1024 forward, 4 launches/step)",
                  "gemm_f32_hbm_bytes_per_launch")
    return dual, single


def side_measurements(pkg, a, dev, x_eval, y_oracle):
    """Reported beside the headline: the other GEMM arithmetic mode of this
    library, and stock PyTorch-ROCm eager of the same module (oracle/torch_twin.py on the GPU) in
    fp32 and under bf16 autocast.  Same batch size, same step, 60 timed steps each."""
    import torch
    from oracle import lifter_oracle as orc
    from oracle.torch_twin import TwinLifter, twin_train_step
    res = {}

    def timed(step, n=60, warm=30):
        for _ in range(warm):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        return a.batch * n / (time.perf_counter() - t0)

    xb, yb = pkg.synth.synthetic_batch(a.batch, 99, dev)
    for other in ("f16x3", "bf16x6", "fp32", "bf16"):
        if other == a.dtype:
            continue
        torch.manual_seed(0)
        m = pkg.LinearModel(34, 51, compute_dtype=other).to(dev).train()
        opt = pkg.FlatAdamW(m, lr=1e-4)
        L = pkg.lib()
        PROF_NOW[0] = PROF_EVERY                            # (60 timed steps: the dense sampling)
        L.pl_prof_enable(PROF_EVERY)
        v = timed(lambda: pkg.train_step(m, opt, xb, yb))
        rl = read_rooflines(pkg, L, other, 2.0 * a.batch * 1024 * 1024, None)
        L.pl_prof_enable(0)
        m.eval()
        with torch.no_grad():
            ye = m(x_eval).cpu().numpy()
        st = {k: t.detach().cpu().numpy() for k, t in m.state_dict().items()}
        yo, _ = orc.forward(st, x_eval.cpu().numpy(), num_stage=2, train=False)
        res[f"this_library_{other}"] = {"poses_per_s": round(v, 1),
                                        "mpjpe_mm_eval_fwd_vs_oracle": float(f"{orc.mpjpe_mm(ye, yo):.3e}"),
                                        "roofline_dual": rl[0] and {k: rl[0][k] for k in ("achieved", "peak", "frac", "avg_launch_us")},
                                        "roofline_forward": rl[1] and {k: rl[1][k] for k in ("achieved", "peak", "frac", "avg_launch_us")}}
        del m, opt

    # the headline step captured once as a hipGraph and replayed (train.GraphedTrainStep; bitwise the eager step)
    torch.manual_seed(0)
    mg = pkg.LinearModel(34, 51, compute_dtype=a.dtype).to(dev).train()
    og = pkg.FlatAdamW(mg, lr=1e-4)
    gstep = pkg.GraphedTrainStep(mg, og, xb, yb)
    res["this_library_%s_hipgraph_replay" % a.dtype] = {"poses_per_s": round(timed(lambda: gstep(xb, yb)), 1)}
    x64, y64 = pkg.synth.synthetic_batch(64, 98, dev)
    torch.manual_seed(0)
    m64 = pkg.LinearModel(34, 51, compute_dtype=a.dtype).to(dev).train()
    g64 = pkg.GraphedTrainStep(m64, pkg.FlatAdamW(m64, lr=1e-4), x64, y64)
    for _ in range(20):
        g64(x64, y64)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        g64(x64, y64)
    torch.cuda.synchronize()
    res["batch_64_hipgraph_replay"] = {"ms_per_step": round(1e3 * (time.perf_counter() - t0) / 200, 4)}
    del mg, og, gstep, m64, g64

    for name, autocast in (("pytorch_rocm_eager_fp32", False), ("pytorch_rocm_eager_bf16_autocast", True)):
        torch.manual_seed(0)
        tw = TwinLifter(34, 51).to(dev).train()
        topt = torch.optim.AdamW(tw.parameters(), lr=1e-4)

        def step():
            if autocast:
                topt.zero_grad()
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    pred = tw(xb).reshape(a.batch, -1, 3)
                torch.nn.functional.mse_loss(pred.float(), yb).backward()
                topt.step()
            else:
                twin_train_step(tw, topt, xb, yb)
        res[name] = {"poses_per_s": round(timed(step), 1)}
        del tw, topt
    return res


def batch64_latency(pkg, a, dev, warm=20, steps=50):
    """BASELINE configs[0]'s batch on the GPU: one train_1.py step at B = 64 (launch-bound), median of `steps` synchronised
    steps -- issued as one hipGraph replay (train.GraphedTrainStep) and as an eager train_step; the lower one is the entry."""
    import torch
    res = {}
    for how in ("hipgraph", "eager"):
        torch.manual_seed(0)
        m = pkg.LinearModel(34, 51, compute_dtype=a.dtype).to(dev).train()
        opt = pkg.FlatAdamW(m, lr=1e-4)
        x, y = pkg.synth.synthetic_batch(64, 77, dev)
        step = pkg.GraphedTrainStep(m, opt, x, y) if how == "hipgraph" else (lambda p, q: pkg.train_step(m, opt, p, q))
        if how == "hipgraph":
            x, y = step.inputs               # (the batch sits in the graph's input buffers, as a feeder would leave it)
        ts = []
        for i in range(warm + steps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            step(x, y)
            torch.cuda.synchronize()
            if i >= warm:
                ts.append(time.perf_counter() - t0)
        res[how] = _median(ts)
    best = min(res, key=res.get)
    t = res[best]
    # serving at the same batch: model.eval(); model(x) (train_1.py:112-126), 200 pipelined calls
    m.eval()
    with torch.no_grad():
        for _ in range(20):
            m(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            m(x)
        torch.cuda.synchronize()
        t_eval = (time.perf_counter() - t0) / 200
    return {"ms_per_step": round(1e3 * t, 4), "poses_per_s": round(64 / t, 1), "launch": best,
            "eval_forward_us": round(1e6 * t_eval, 1),
            "ms_per_step_eager": round(1e3 * res["eager"], 4), "ms_per_step_hipgraph": round(1e3 * res["hipgraph"], 4),
            "how": f"median of {steps} synchronised steps after {warm} warm-ups (host launch latency included), issued as one "
                   "hipGraph replay (GraphedTrainStep) and as an eager train_step: the lower of the two"}


def conv_macs_per_frame(model, size=256):
    """Forward multiply-accumulates of one heat-map network (backbone + head) per frame, from the modules' shapes."""
    import torch.nn as nn
    macs, hw = 0, {}

    def walk(mod, h):
        nonlocal macs
        for m in mod.children():
            if isinstance(m, nn.Conv2d):
                h = (h + 2 * m.padding[0] - m.kernel_size[0]) // m.stride[0] + 1
                macs += h * h * m.out_channels * m.in_channels * m.kernel_size[0] * m.kernel_size[1]
            elif isinstance(m, nn.ConvTranspose2d):
                h = h * 2
                macs += h * h * m.out_channels * m.in_channels * 4          # 4x4 stride 2: four taps per output pixel
        return h
    r = model.preact
    h = walk(nn.Sequential(r.conv1), size)
    h = (h + 2 - 3) // 2 + 1                                               # max-pool 3x3 / 2
    for li in (1, 2, 3, 4):
        for blk in getattr(r, f"layer{li}"):
            hin = h
            h1 = walk(nn.Sequential(blk.conv1), hin)
            h2 = walk(nn.Sequential(blk.conv2), h1)
            h = walk(nn.Sequential(blk.conv3), h2)
            if blk.downsample is not None:
                walk(nn.Sequential(blk.downsample[0]), hin)
    h = walk(model.deconv_layers, h)
    walk(nn.Sequential(model.final_layer), h)
    return macs


def cycle_workload(pkg, a, rank, local, world):
    """BASELINE configs[4]: one phase5 cycle step (train_5 copy.py:147-236) per `step`, 128 frames of 256 x 256 per GPU,
    data parallel = one gradient all-reduce per model (dp.SyncedOptimizer).  Reference optimizers: Adam on all four models."""
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU path")
    dev = pkg.dp.local_device(local)
    torch.cuda.set_device(dev)
    torch.manual_seed(0)
    m2, m3 = pkg.Model_2D().train(), pkg.Model_3D().train()
    for m, seed in ((m2, 61), (m3, 62)):
        m.load_state_dict(pkg.synth.seeded_state(m.state_dict(), seed))
        with torch.no_grad():
            m.final_layer.weight.mul_(1e-3)
    macs = conv_macs_per_frame(m2) + conv_macs_per_frame(m3)
    m2, m3 = m2.to(dev), m3.to(dev)
    lift = pkg.LinearModel(34, 51, linear_size=1024, p_dropout=0.5, compute_dtype="f16x3").to(dev).train()   # train_5 copy.py:94
    proj = pkg.LinearModel(51, 34, linear_size=64, p_dropout=0.5, compute_dtype="f16x3").to(dev).train()     # :96
    lift.manual_seed(1234 + rank); proj.manual_seed(4321 + rank)
    lr = 1e-4
    opts = [pkg.FlatAdam(m2, lr=lr), pkg.FlatAdam(m3, lr=lr),                                                 # :105-109: Adam x4
            pkg.FlatAdamW(lift, lr=lr, weight_decay=0.0), pkg.FlatAdamW(proj, lr=lr, weight_decay=0.0)]
    if world > 1:
        opts = [pkg.dp.SyncedOptimizer(o, m) for o, m in zip(opts, (m2, m3, lift, proj))]
    B = a.batch
    pool = [(pkg.synth.seeded_frames(B, 70 + 10 * rank + i).to(dev),) + pkg.synth.synthetic_batch(B, 80 + 10 * rank + i, dev)
            for i in range(2)]
    crit = pkg.TriangleLoss(Project=True, era="lifter")
    torch.cuda.synchronize()

    def run(n, first=0):
        loss = None
        for i in range(first, first + n):
            fr, y1, y2 = pool[i % len(pool)]
            loss = pkg.cycle_step(m2, m3, lift, opts, fr, y1, y2, crit, model_proj=proj, Flip=a.flip)[0]
        return loss

    def barrier():
        if world > 1:
            dist.barrier()
    run(a.warmup)
    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss = run(a.steps, a.warmup)
    torch.cuda.synchronize(); barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        passes = 2 if a.flip else 1
        flop_frame = 2.0 * macs * 3 * passes                  # forward + data gradient + weight gradient (convolutions only)
        value = B * world * a.steps / dt
        tf = value * flop_frame / 1e12
        out = {"metric": "frames/sec, phase5 cycle step (Model_2D + Model_3D on 256x256 frames, lifter x2, projector, TriangleLoss, Adam x4)",
               "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": round(1e3 * dt / a.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f16x3 (two fp16 planes per fp32 operand, 3 fp16 MFMAs per product, fp32 accumulate)", "data": "synthetic",
               "config": {"workload": "BASELINE configs[4]: phase5_loop train_5 copy.py:147-236 cycle step, Triangle + Project"
                                      + (" + Flip" if a.flip else "") + (f", {dist.get_backend()} gradient all-reduce per model" if world > 1 else ""),
                          "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}",
                          "frame": "256x256x3 NHWC", "lifter": "LinearModel(34,51,1024)", "projector": "LinearModel(51,34,64)"},
               "final_loss": float(loss),
               "roofline": {"bound": "mfma", "achieved": round(tf / world, 2), "peak": PEAK_BF16_MATRIX_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(tf / world / PEAK_BF16_MATRIX_TFLOPS, 4), "traffic": None,
                            "kernel": "whole step (about 2,200 launches; no single dominant kernel), convolution FLOPs only",
                            "flop_per_frame": flop_frame, "mfma_issue_frac": round(3 * tf / world / PEAK_BF16_MATRIX_TFLOPS, 4)}}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cycle_cpu_baseline(pkg, a.flip)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cycle_cpu_baseline(pkg, flip, B=4, steps=2):
    """The same step on stock torch.nn modules (oracle/cycle_twin.py) on the host cores, fp32: a bounded sample."""
    import copy
    import torch
    from oracle import cycle_twin as twin
    from oracle.torch_twin import TwinLifter
    cores = host_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    m2, m3 = pkg.Model_2D().train(), pkg.Model_3D().train()
    for m, seed in ((m2, 61), (m3, 62)):
        m.load_state_dict(pkg.synth.seeded_state(m.state_dict(), seed))
    lift, proj = TwinLifter(34, 51, linear_size=1024).train(), TwinLifter(51, 34, linear_size=64).train()
    opts = [torch.optim.Adam(m.parameters(), lr=1e-4) for m in (m2, m3, lift, proj)]
    fr = pkg.synth.seeded_frames(B, 5).permute(0, 3, 1, 2).contiguous()
    y1, y2 = pkg.synth.synthetic_batch(B, 6)
    ts = []
    for i in range(1 + steps):
        t0 = time.perf_counter()
        twin.cycle_step(m2, m3, lift, proj, opts, fr, y1, y2, flip)
        if i:
            ts.append(time.perf_counter() - t0)
    t = _median(ts)
    return {"value": round(B / t, 3), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"median of {steps} cycle steps of the stock-PyTorch eager twin at batch {B} (256x256 frames), fp32, "
                      f"{cores} threads, after 1 warm-up step", "ms_per_step": round(1e3 * t, 1), "torch": torch.__version__}


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(a)                               # before anything in this process touches the GPU
    import torch
    import torch.distributed as dist
    pkg = importlib.import_module("3d_poseestimation_amd")
    rank, local, world = pkg.dp.init_from_env()
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.workload == "cycle":
        return cycle_workload(pkg, a, rank, local, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the lifter has no CPU path")
    dev = pkg.dp.local_device(local)
    torch.cuda.set_device(dev)
    L = pkg.lib()

    torch.manual_seed(0)                                    # same initial weights on every rank
    model = pkg.LinearModel(34, 51, linear_size=1024, num_stage=2, p_dropout=0.5, BN=True,
                            compute_dtype=a.dtype).to(dev).train()
    model.manual_seed(1234 + rank)                          # own dropout stream per rank
    if a.sync_bn and world > 1:
        model.set_sync_bn(True)
    opt = pkg.FlatAdamW(model, lr=1e-4)                     # train_1.py:39 (weight_decay 0.01)
    sync = pkg.dp.GradSync(overlap=not a.no_overlap) if world > 1 else None
    model.set_grad_sync(sync)                               # all-reduce overlapped with the backward tail
    pool = [pkg.synth.synthetic_batch(a.batch, 1234 + 1000 * rank + i, dev) for i in range(8)]
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()

    def run(n, first=0):
        for i in range(first, first + n):
            x, y = pool[i % len(pool)]
            pkg.train_step(model, opt, x, y, grad_sync=sync)

    if world > 1:                                           # RCCL communicator + the bucket sizes the step
        for _, _, lo, hi in model._bwd_ranges():            # uses are set up before anything is timed,
            dist.all_reduce(model.flat_grads[lo:hi])        # whatever --warmup says
        model.flat_grads.zero_()
    run(a.warmup)
    # N = 1: the same step as ONE hipGraph replay (train.GraphedTrainStep: zero_grad + forward + MSE + backward + AdamW, dropout
    # stream and Adam's t advanced on the device).  Which way of issuing it is used for the timed region is decided here, on
    # `cal` untimed steps of each -- on a healthy host the eager loop keeps ahead of the GPU at B = 4096 and is as fast.
    launch, gstep = "eager", None
    if world == 1 and a.launch != "eager":
        cal = max(10, min(a.warmup, 30))

        def run_graph(n, first=0):
            for i in range(first, first + n):
                gstep(*pool[i % len(pool)])
        try:
            gstep = pkg.GraphedTrainStep(model, opt, *pool[0])
            tc = []
            for fn in (run, run_graph):
                fn(3); torch.cuda.synchronize()
                tq = time.perf_counter(); fn(cal); torch.cuda.synchronize()
                tc.append((time.perf_counter() - tq) / cal)
            if a.launch == "hipgraph" or tc[1] < 0.98 * tc[0]:
                launch = "hipgraph"
        except Exception as e:                              # (capture refused: stay eager, say why)
            print(f"# hipgraph capture unavailable: {e}", file=sys.stderr)
            gstep = None
            if a.launch == "hipgraph":
                raise
    timed = run if launch == "eager" else run_graph
    if not a.no_prof and launch == "eager":
        L.pl_prof_enable(prof_every(a.steps))
    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    timed(a.steps, a.warmup)
    torch.cuda.synchronize(); barrier()
    dt = time.perf_counter() - t0
    roofline = roofline_single = None
    if not a.no_prof and launch == "hipgraph":
        # (events cannot be read out of a replayed graph: the same kernels are sampled on eager steps right behind the timed region)
        n_prof = min(a.steps, 64)
        L.pl_prof_enable(prof_every(n_prof))
        run(n_prof, a.warmup + a.steps)
        torch.cuda.synchronize()
    if not a.no_prof:
        one = 2.0 * a.batch * 1024 * 1024                  # one 1024-wide GEMM: 8.59 GFLOP at B=4096
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and a.batch == BATCH:
            traffic = json.load(open(tpath))

        roofline, roofline_single = read_rooflines(pkg, L, a.dtype, one, traffic)
        L.pl_prof_enable(0)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # N > 1: what the gradient all-reduce costs the step -- the same steps with the collective stubbed out (every
    # rank keeps its local gradients; results are discarded).  exposed = synchronised step - stubbed step.
    exposed_us = None
    if world > 1:
        class _NoSync(pkg.dp.GradSync):
            def launch_bucket(self, flat_slice):
                pass

            def __call__(self, m):
                return 1.0 / self.world()
        stub = _NoSync()
        model.set_grad_sync(stub)
        n_st = max(10, a.steps // 4)
        for i in range(5):
            pkg.train_step(model, opt, *pool[i % len(pool)], grad_sync=stub)
        barrier(); torch.cuda.synchronize()
        ts = time.perf_counter()
        for i in range(n_st):
            pkg.train_step(model, opt, *pool[i % len(pool)], grad_sync=stub)
        torch.cuda.synchronize(); barrier()
        t = torch.tensor([(time.perf_counter() - ts) / n_st], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        exposed_us = round(1e6 * (dt / a.steps - float(t.item())), 1)
        model.set_grad_sync(sync)
        pkg.dp.broadcast_model(model)                            # the stubbed steps let the replicas drift

    # forward+backward only (no optimizer / all-reduce), reported beside the headline number
    def run_fb(n):
        for i in range(n):
            x, y = pool[i % len(pool)]
            model.zero_grad(set_to_none=True)
            pkg.mse_loss(model(x).reshape(y.shape), y).backward()
    run_fb(5)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    nfb = max(10, a.steps // 4)
    run_fb(nfb)
    torch.cuda.synchronize()
    dt_fb = time.perf_counter() - t1

    out = None
    if rank == 0:
        # parity in the same run: eval forward vs the numpy oracle on a bench batch
        import numpy as np
        from oracle import lifter_oracle as orc
        model.eval()
        x, _ = pool[0]
        with torch.no_grad():
            y_gpu = model(x).cpu().numpy()
        st = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
        y_orc, _ = orc.forward(st, x.cpu().numpy(), num_stage=2, train=False)
        mpjpe = orc.mpjpe_mm(y_gpu, y_orc)
        model.train()
        poses = a.batch * world * a.steps
        value = poses / dt
        out = {
            "metric": "poses/sec fwd+bwd, 17-joint lifting batch 4096; MPJPE vs ref",
            "value": round(value, 1), "unit": "poses/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": {"fp32": "f32", "bf16": "bf16", "bf16x6": "bf16x6 (3-way bf16 split of fp32 operands, fp32 accumulate)",
                      "f16x3": "f16x3 (two fp16 planes per fp32 operand, 3 fp16 MFMAs per product, fp32 accumulate)"}[a.dtype],
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: phase1_lifting "
                                   "LinearModel 34-1024-2x(1024-1024)-51, BN+ReLU+Dropout(0.5), one train_1.py "
                                   "step = zero_grad+forward+MSE+backward+AdamW"
                                   + (f"+{dist.get_backend()} grad all-reduce" if world > 1 else ""),
                       "launch": launch + (" (one hipGraph replay per step)" if launch == "hipgraph" else " (one call per kernel)"),
                       "per_gpu_batch": a.batch, "global_batch": a.batch * world,
                       "parallelism": f"dp{world}" + ("+syncbn" if a.sync_bn and world > 1 else ""),
                       "gemm_arith": {
                           "fp32": "fp32 MFMA (v_mfma_f32_32x32x2_f32)",
                           "bf16": "bf16 MFMA (v_mfma_f32_32x32x16_bf16), operands rounded to bf16, fp32 accumulate",
                           "bf16x6": "bf16 MFMA, three-way operand split x 6 products (fp32-grade), fp32 accumulate and storage",
                           "f16x3": "fp16 MFMA (v_mfma_f32_16x16x32_f16), operands as two fp16 planes (22-23 bits) x 3 "
                                    "products on two fp32 accumulators (fp32-grade); planes staged by LDS-DMA"}[a.dtype]},
            "step_tflops": round(value * FLOP_PER_POSE / 1e12, 2),
            "step_frac_of_matrix_peak": round(value * FLOP_PER_POSE / 1e12 / (
                (PEAK_F32_MATRIX_TFLOPS if a.dtype == "fp32" else PEAK_BF16_MATRIX_TFLOPS) * world), 4),
            "fwd_bwd_only_poses_per_s_per_gpu": round(a.batch * nfb / dt_fb, 1),
            "mpjpe_mm_eval_fwd_vs_oracle": float(f"{mpjpe:.3e}"),
            "roofline": roofline,
            "roofline_forward_gemm": roofline_single,
        }
        if launch == "hipgraph":
            for rf in (roofline, roofline_single):
                if rf:
                    rf["sampling"] += " -- of eager steps right behind the timed region (the timed steps are graph replays)"
        if world > 1:
            out["allreduce"] = {"overlap": not a.no_overlap, "exposed_us_per_step": exposed_us,
                                "bytes": int(model.flat_grads.numel()) * 4,
                                "how": "step time with the collective minus the same steps with it stubbed out"}
        if world == 1:
            out["batch_64"] = batch64_latency(pkg, a, dev)
        if world == 1 and not a.no_extras:
            out["other_modes"] = side_measurements(pkg, a, dev, x, y_orc)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.batch)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
