"""examples/usage.jl of the reference, line for line, on the HIP path: 1024 samples of Beta(2, 4), the ICNF of the example
(softplus 4 => 16 => 16 => 3 net, one variable + two augmented dimensions, lambda = 0.01, steer_rate = 0.1, the default
VCABM solver at 1e-4), ICNFModel fit for 300 epochs with WeightDecay + Adam, then ICNFDist: pdf of the data against the
true density and fresh samples.  Prints one JSON line with the timings and the example's three error measures.
(As in the reference, logp̂x is the density in the AUGMENTED space - basedist over nvariables + naugments dimensions,
src/core/base_icnf.jl:158-172 - so with naugments > 0 the "estimated pdf" is not comparable in scale with the data density;
the example's measures are printed for the record, the sample moments are the meaningful check.  The first fit call
includes the one-time library / kernel-module load.)"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
torch.manual_seed(0)

# ## Data
ndata, ndimensions = 1024, 1
data_dist = torch.distributions.Beta(2.0, 4.0)
r = data_dist.sample((ndimensions, ndata)).float()

# ## Parameters
nvariables = r.shape[0]
naugments = nvariables + 1
n_in = nvariables + naugments + 1   # add time concatenation
n_out = nvariables + naugments
n_hidden = n_in * 4

# ## Model
icnf = pkg.ICNF(
    nn=pkg.Chain(pkg.Dense(n_in, n_hidden, pkg.softplus), pkg.Dense(n_hidden, n_hidden, pkg.softplus), pkg.Dense(n_hidden, n_out)),
    nvariables=nvariables, naugments=naugments, nconditions=0,
    lambda1=0.01, lambda2=0.01, lambda3=0.01, steer_rate=0.1, tspan=(0.0, 1.0),
    device="cuda:0", autonomous=False, inplace=False, compute_mode=pkg.LuxVecJacMatrixMode(),
    sol_kwargs=dict(maxiters=2 ** 63 - 1, reltol=1e-4, abstol=1e-4, alg=pkg.VCABM()))

# ## Fit It
model = pkg.ICNFModel(icnf=icnf, batchsize=1024, epochs=300, callback=pkg.make_opt_callback(64),
                      weight_decay=1e-4, eta=0.001, beta=(0.9, 0.999), epsilon=1e-8,
                      init_rng=torch.Generator().manual_seed(1), shuffle_rng=torch.Generator().manual_seed(2))
df = r.t()                                        # DataFrame(permutedims(r), :auto): one row per sample
torch.cuda.synchronize()
t0 = time.perf_counter()
fitresult, _, report = model.fit(df)
torch.cuda.synchronize()
t_fit = time.perf_counter() - t0
warm = pkg.ICNFModel(icnf=icnf, batchsize=1024, epochs=100, callback=None, init_rng=torch.Generator().manual_seed(1))
t0 = time.perf_counter()
warm.fit(df)                                      # the same loop again, library and kernels already loaded
torch.cuda.synchronize()
t_warm = (time.perf_counter() - t0) / 100

icnf_mach_fn = os.path.join(os.environ.get("TMPDIR", "/tmp"), "icnf-machine.pt")
pkg.save_machine(icnf_mach_fn, model, fitresult)   # MLJBase.save(icnf_mach_fn, mach)  # save it
model, fitresult = pkg.load_machine(icnf_mach_fn)  # mach = machine(icnf_mach_fn)  # load it

# ## Use It
d = pkg.ICNFDist.from_fit(model, fitresult, pkg.TestMode())
actual_pdf = data_dist.log_prob(r[0].clamp(1e-6, 1 - 1e-6)).exp()
t0 = time.perf_counter()
estimated_pdf = d.pdf(r).cpu()
new_data = d.rand(ndata)
torch.cuda.synchronize()
t_use = time.perf_counter() - t0

# ## Evaluate It
diff = estimated_pdf - actual_pdf
res = {"mad": float(diff.abs().mean()), "msd": float((diff ** 2).mean()), "tv_dis": float(0.5 * diff.abs().sum() / ndata),
       "fit_s": t_fit, "fit_iterations": report["stats"]["iterations"], "ms_per_iteration": 1e3 * t_fit / report["stats"]["iterations"], "ms_per_iteration_warm": 1e3 * t_warm,
       "final_loss": report["stats"]["final_loss"], "pdf_and_rand_ms": 1e3 * t_use,
       "new_data_mean": float(new_data.mean()), "true_mean": 2.0 / 6.0, "new_data_std": float(new_data.std()),
       "true_std": (2.0 * 4.0 / (36.0 * 7.0)) ** 0.5}
print(json.dumps(res))
