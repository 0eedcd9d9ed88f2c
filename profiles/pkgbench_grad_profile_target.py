"""rocprofv3 / cProfile target: the gradient benchmarks of the reference's PkgBenchmark suite (AD-1-order: loss + dloss/dps at 2^10
samples, ICNF(nvariables = 1) defaults) - kernel trace under rocprofv3, host profile with PROFILE=1."""
import cProfile, pstats, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
dev = torch.device("cuda:0")
r = torch.distributions.Beta(2.0, 4.0).sample((1, 1024)).float().to(dev)
icnf = pkg.ICNF(nvariables=1, device=dev)
ps, st = pkg.setup(torch.Generator().manual_seed(0), icnf)
ps = ps.to(dev)
mode = pkg.TrainMode(True)
for _ in range(10):
    pkg.loss_and_gradient(icnf, mode, r, ps, st)
torch.cuda.synchronize()
n = int(os.environ.get("N", "200"))
t0 = time.perf_counter()
for _ in range(n):
    pkg.loss_and_gradient(icnf, mode, r, ps, st)
torch.cuda.synchronize()
print("ms per call", 1e3 * (time.perf_counter() - t0) / n, icnf.last_solve_stats["naccept"], icnf.grad_path(mode), flush=True)
if os.environ.get("PROFILE"):
    pr = cProfile.Profile(); pr.enable()
    for _ in range(n):
        pkg.loss_and_gradient(icnf, mode, r, ps, st)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
