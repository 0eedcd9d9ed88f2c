"""rocprofv3 target: a few loss + gradient steps of a BASELINE configuration (CFG = cfg2 | cfg3 | cfg4), bench.py's workload."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, __graft_entry__ as entry
pkg = entry.load_package(); o64, oc = entry.load_oracle()
class A: pass
a = A(); a.batch = 0; a.path = 0; a.arith = "f32"; a.mode = "grad"
dev = torch.device("cuda:0")
w = bench.make_workload(pkg, o64, os.environ.get("CFG", "cfg2"), a, 0, dev, torch, grad=True)
for _ in range(int(os.environ.get("REPS", "6"))):
    pkg.loss_and_gradient(w["icnf"], w["mode"], *w["args"], eps=w["E"])
torch.cuda.synchronize()
