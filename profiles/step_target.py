"""rocprofv3 target: REPS steps of ONE bench.py workload, nothing else on the device - WL = cfg2 | cfg3 | cfg4 | cfg5 | nv20 with an
optional ':grad' (loss + gradient), e.g. WL=cfg4:grad.  profiles/traffic_all.sh sums FETCH_SIZE / WRITE_SIZE over every kernel of the
run and divides by REPS: HBM bytes per step of the WORKLOAD (forward, reverse sweep, weight-cotangent launches, reductions)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, __graft_entry__ as entry
pkg = entry.load_package(); o64, oc = entry.load_oracle()
class A: pass
wl = os.environ.get("WL", "cfg2")
name, _, md = wl.partition(":")
a = A(); a.batch = 0; a.path = 0; a.arith = "f32"; a.mode = "grad" if md == "grad" else "infer"
dev = torch.device("cuda:0")
w = bench.make_workload(pkg, o64, name, a, 0, dev, torch, grad=(md == "grad"), batch=bench.CONFIGS[name][2])
reps = int(os.environ.get("REPS", "3"))
for _ in range(reps):
    if md == "grad":
        pkg.loss_and_gradient(w["icnf"], w["mode"], *w["args"], eps=w["E"])
    else:
        pkg.loss(w["icnf"], w["mode"], *w["args"], eps=w["E"])
torch.cuda.synchronize()
