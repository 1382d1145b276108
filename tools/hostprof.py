"""Where the host time of one lifter train step goes.  (1) enqueue time with an EMPTY queue (sync before every step):
the pure host cost; (2) the same with the queue left to run deep; (3) cProfile of 400 steps."""
import cProfile, importlib, pstats, sys, time, torch
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("3d_poseestimation_amd")
torch.manual_seed(0)
for B in (4096, 64):
    m = pkg.LinearModel(34, 51, compute_dtype="f16x3").cuda().train()
    opt = pkg.FlatAdamW(m, lr=1e-4)
    x, y = pkg.synth.synthetic_batch(B, 1, "cuda")
    for _ in range(20): pkg.train_step(m, opt, x, y)
    torch.cuda.synchronize()
    acc = 0.0
    for _ in range(300):
        t0 = time.perf_counter(); pkg.train_step(m, opt, x, y); acc += time.perf_counter() - t0
        torch.cuda.synchronize()
    print(f"B={B}: empty-queue enqueue {1e3*acc/300:.3f} ms/step")
    t0 = time.perf_counter()
    for _ in range(300): pkg.train_step(m, opt, x, y)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"B={B}: deep-queue enqueue {1e3*(t1-t0)/300:.3f} ms/step, total {1e3*(t2-t0)/300:.3f}")
if len(sys.argv) > 1:
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(400): pkg.train_step(m, opt, x, y)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)
