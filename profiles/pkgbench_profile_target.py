"""rocprofv3 target: the reference's PkgBenchmark scenario (benchmark/benchmarks.jl:11-19: ICNF(; nvariables = 1), defaults, 2^10 samples)
- `loss` in TrainMode{true} and TestMode, 200 calls each - so that the kernel trace says what a call's wall time is made of."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
dev = torch.device("cuda:0")
r = torch.distributions.Beta(2.0, 4.0).sample((1, 1024)).float().to(dev)
icnf = pkg.ICNF(nvariables=1, device=dev)
ps, st = pkg.setup(torch.Generator().manual_seed(0), icnf)
ps = ps.to(dev)
for name, mode in (("train", pkg.TrainMode(True)), ("test", pkg.TestMode())):
    for _ in range(5):
        pkg.loss(icnf, mode, r, ps, st)
    torch.cuda.synchronize()
    n = int(os.environ.get("N", "200"))
    t0 = time.perf_counter()
    for _ in range(n):
        pkg.loss(icnf, mode, r, ps, st)
    torch.cuda.synchronize()
    print(name, "ms per call", 1e3 * (time.perf_counter() - t0) / n, icnf.last_solve_stats, icnf.kernel_name(mode), flush=True)
