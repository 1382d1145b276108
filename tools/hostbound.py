import importlib, sys, time, torch
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("3d_poseestimation_amd")
for dt in ("f16x3", "bf16"):
    torch.manual_seed(0)
    m = pkg.LinearModel(34, 51, compute_dtype=dt).cuda().train()
    opt = pkg.FlatAdamW(m, lr=1e-4)
    x, y = pkg.synth.synthetic_batch(4096, 1, "cuda")
    for _ in range(20): pkg.train_step(m, opt, x, y)
    torch.cuda.synchronize()
    N = 200
    t0 = time.perf_counter()
    for _ in range(N): pkg.train_step(m, opt, x, y)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{dt}: host enqueue {1e3*(t1-t0)/N:.3f} ms/step, total {1e3*(t2-t0)/N:.3f} ms/step, GPU tail after enqueue {1e3*(t2-t1):.1f} ms")
