"""Where the host time of the PkgBenchmark `loss` call goes: cProfile over 2000 calls (benchmark/benchmarks.jl scenario, 2^10 samples)."""
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
for _ in range(20):
    pkg.loss(icnf, mode, r, ps, st)
torch.cuda.synchronize()
n = 2000
t0 = time.perf_counter()
for _ in range(n):
    pkg.loss(icnf, mode, r, ps, st)
torch.cuda.synchronize()
print("ms per call", 1e3 * (time.perf_counter() - t0) / n)
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    pkg.loss(icnf, mode, r, ps, st)
torch.cuda.synchronize()
pr.disable()
st_ = pstats.Stats(pr)
st_.sort_stats("tottime").print_stats(22)
