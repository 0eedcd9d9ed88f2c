"""Host-side mirror of the reference's thin callers of the hot path:

  * `ICNFModel` / `CondICNFModel` — the MLJ `Unsupervised` models (src/exts/mlj_ext/core_icnf.jl:1-68,
    core_cond_icnf.jl, core.jl:8-43): `fit` = setup -> shuffled mini-batches -> WeightDecay + Adam on
    `loss(icnf, TrainMode{true}(), ...)` for `epochs` passes; `transform` = `exp.(logp̂x)` in TestMode.
  * `ICNFDist` / `CondICNFDist` — the Distributions.jl wrappers (src/exts/dist_ext/core_icnf.jl:1-75,
    core_cond_icnf.jl): `logpdf` -> `inference`, `rand` -> `generate`.

The reference differentiates the loss with Zygote through the ODE solve; here every optimiser step is one
`loss_and_gradient` call (the reverse-sweep HIP kernels), with the parameters and the Adam state resident
on the device.  With `torch.distributed` initialised each rank fits on its own rows of X and the gradient
is all-reduced inside `loss_and_gradient` (data parallel, identical parameters on every rank).

Tables are (n_samples, n_features) like MLJ's (`permutedims(MLJModelInterface.matrix(X))` makes them
feature-major, core_icnf.jl:33); anything `torch.as_tensor` accepts, or a pandas DataFrame.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Iterator, Optional, Tuple

import torch

from . import icnf as _icnf
from .icnf import ICNF, Mode, TestMode, TrainMode

__all__ = ["ICNFModel", "CondICNFModel", "ICNFDist", "CondICNFDist", "make_opt_callback", "save_machine", "load_machine"]


def make_opt_callback(n: int) -> Callable[[int, float], bool]:
    """Print every n-th iteration (src/exts/mlj_ext/core.jl:96-105).  Returning True stops the fit."""
    def opt_callback(it: int, l: float) -> bool:
        if it % n == 1:
            print(f"Iteration: {it} | Loss: {l}")
        return False
    return opt_callback


def _matrix(X, device) -> torch.Tensor:
    """permutedims(matrix(X)) as a Float32 device tensor: (n_features, n_samples)."""
    if hasattr(X, "to_numpy"):          # pandas DataFrame / Series
        X = X.to_numpy()
    t = torch.as_tensor(X, dtype=torch.float32)
    if t.dim() == 1:
        t = t[:, None]
    if t.dim() != 2:
        raise ValueError("DimensionMismatch: a table (n_samples, n_features) is expected")
    return t.to(device).t()


def epoch_batches(n: int, batchsize: int, gen: Optional[torch.Generator]) -> Iterator[torch.Tensor]:
    """MLUtils.DataLoader(data; batchsize, shuffle = true, partial = true) (core.jl:24-35): a fresh
    permutation of the n columns every epoch, consecutive batches, the last one possibly shorter;
    batchsize 0 = the whole data set in one batch (core.jl:37-43)."""
    bs = n if batchsize == 0 else batchsize
    if bs < 1:
        raise ValueError("batchsize must be >= 0")
    perm = torch.randperm(n, generator=gen)
    for lo in range(0, n, bs):
        yield perm[lo:lo + bs]


@dataclass
class _MLJICNF:
    icnf: ICNF
    loss: Callable = None                      # loss(icnf, mode, xs, [ys,] ps, st); None = the package's loss
    batchsize: int = 1024
    # sol_kwargs of the reference default (core_icnf.jl:14-28): epochs, callback, WeightDecay + Adam
    epochs: int = 300
    callback: Optional[Callable[[int, float], bool]] = field(default_factory=lambda: make_opt_callback(64))
    weight_decay: float = 1.0e-4
    eta: float = 0.001
    beta: Tuple[float, float] = (0.9, 0.999)
    epsilon: float = 1.0e-8
    shuffle_rng: Optional[torch.Generator] = None
    init_rng: Optional[torch.Generator] = None   # CPU generator for the initial parameters (LuxCore.setup(icnf.rng, icnf))

    _conditioned = False

    def _fit(self, x: torch.Tensor, y: Optional[torch.Tensor]):
        ic = self.icnf
        if ic.conditioned != self._conditioned:
            raise TypeError("MethodError: ICNFModel needs an unconditioned flow, CondICNFModel a conditioned one")
        if self.loss is not None and self.loss is not _icnf.loss:
            raise NotImplementedError("a custom loss has no gradient kernel; fit optimises the package's `loss`")
        ps, st = _icnf.setup(self.init_rng, ic)
        ps = ps.to(ic.device)
        # Optimisers.OptimiserChain(WeightDecay(lambda), Adam(eta, beta, epsilon)): the decay term lambda * p is
        # added to the gradient before Adam sees it = torch's (non-decoupled) Adam weight_decay
        opt = torch.optim.Adam([ps], lr=self.eta, betas=self.beta, eps=self.epsilon, weight_decay=self.weight_decay)
        mode = TrainMode(True)                 # TrainMode{true}() (core_icnf.jl:44)
        n = x.shape[1]
        it, last, stop = 0, float("nan"), False
        for _ in range(self.epochs):
            for idx in epoch_batches(n, self.batchsize, self.shuffle_rng):
                idx = idx.to(x.device)
                args = (x[:, idx],) + ((y[:, idx],) if y is not None else ()) + (ps, st)
                value, grad = _icnf.loss_and_gradient(ic, mode, *args)
                ps.grad = grad
                opt.step()
                it += 1
                if self.callback is not None:
                    last = float(value)
                    if self.callback(it, last):
                        stop = True
                        break
            if stop:
                break
        fitresult = (ps.detach(), st)
        report = {"stats": {"iterations": it, "final_loss": float(value) if it else last}}
        return fitresult, None, report

    @staticmethod
    def fitted_params(fitresult):
        """(learned_parameters = ps, states = st) (core.jl:3-6)."""
        ps, st = fitresult
        return {"learned_parameters": ps, "states": st}

    def _px(self, logp: torch.Tensor):
        px = torch.exp(logp).cpu().numpy()
        try:
            import pandas as pd
            return pd.DataFrame({"px": px})    # DataFrames.DataFrame(; px = exp.(logp̂x)) (core_icnf.jl:66-67)
        except Exception:  # pragma: no cover
            return {"px": px}


class ICNFModel(_MLJICNF):
    """MLJ model of an unconditioned flow (src/exts/mlj_ext/core_icnf.jl)."""

    def fit(self, X, verbosity: int = 0):
        return self._fit(_matrix(X, self.icnf.device), None)

    def transform(self, fitresult, Xnew):
        ps, st = fitresult
        logp = _icnf.inference(self.icnf, TestMode(), _matrix(Xnew, self.icnf.device), ps, st)[0]
        return self._px(logp)


class CondICNFModel(_MLJICNF):
    """MLJ model of a conditioned flow; data is the pair (X, Y) (src/exts/mlj_ext/core_cond_icnf.jl)."""
    _conditioned = True

    def fit(self, XY, verbosity: int = 0):
        X, Y = XY
        x, y = _matrix(X, self.icnf.device), _matrix(Y, self.icnf.device)
        if x.shape[1] != y.shape[1]:
            raise ValueError("DimensionMismatch: X and Y need the same number of rows")
        return self._fit(x, y)

    def transform(self, fitresult, XYnew):
        Xnew, Ynew = XYnew
        ps, st = fitresult
        dev = self.icnf.device
        logp = _icnf.inference(self.icnf, TestMode(), _matrix(Xnew, dev), _matrix(Ynew, dev), ps, st)[0]
        return self._px(logp)


class ICNFDist:
    """ICNFDist(icnf, mode, ps, st) <: ContinuousMultivariateDistribution (src/exts/dist_ext/core_icnf.jl).
    Arrays are feature-major like Distributions.jl's: logpdf of an (nvariables, n) matrix is an n-vector."""

    def __init__(self, icnf: ICNF, mode: Mode, ps: torch.Tensor, st: dict):
        self.icnf, self.mode, self.ps, self.st = icnf, mode, ps, st

    @classmethod
    def from_fit(cls, model: _MLJICNF, fitresult, mode: Mode, *extra):
        """ICNFDist(mach, mode) (core_icnf.jl:8-11)."""
        ps, st = fitresult
        return cls(model.icnf, mode, *extra, ps, st)

    def __len__(self) -> int:               # Base.length(d) (core.jl:6-8)
        return self.icnf.nvariables

    def _cond(self, n: int) -> tuple:
        return ()

    def logpdf(self, A) -> torch.Tensor:
        A = torch.as_tensor(A, dtype=torch.float32)
        vec = A.dim() == 1                   # a single point: computed as a one-column matrix (core_icnf.jl:21-27)
        if vec:
            A = A[:, None]
        out = _icnf.inference(self.icnf, self.mode, A.to(self.icnf.device), *self._cond(A.shape[1]), self.ps, self.st)[0]
        return out[0] if vec else out

    def pdf(self, A) -> torch.Tensor:
        return torch.exp(self.logpdf(A))

    def rand(self, n: Optional[int] = None) -> torch.Tensor:
        """rand(d) -> (nvariables,) ; rand(d, n) -> (nvariables, n) (core_icnf.jl:43-75)."""
        k = 1 if n is None else int(n)
        A = _icnf.generate(self.icnf, self.mode, *self._cond(k), self.ps, self.st, k)
        return A[:, 0] if n is None else A


class CondICNFDist(ICNFDist):
    """CondICNFDist(icnf, mode, ys, ps, st): the conditions are part of the distribution; a call on n
    columns uses ys[:, 1:n] (src/exts/dist_ext/core_cond_icnf.jl:45,79)."""

    def __init__(self, icnf: ICNF, mode: Mode, ys, ps: torch.Tensor, st: dict):
        super().__init__(icnf, mode, ps, st)
        ys = torch.as_tensor(ys, dtype=torch.float32)
        self.ys = (ys[:, None] if ys.dim() == 1 else ys).to(icnf.device)

    def _cond(self, n: int) -> tuple:
        if n > self.ys.shape[1]:
            raise IndexError(f"BoundsError: {n} columns requested, the distribution holds {self.ys.shape[1]} conditions")
        return (self.ys[:, :n],)


# ---------------------------------------------------------------------------------------------------------------
# MLJBase.save(file, mach) / machine(file) (examples/usage.jl:96-98): the fitted model as one file
# ---------------------------------------------------------------------------------------------------------------
_ALG_NAMES = {"VCABM": _icnf.VCABM, "Tsit5": _icnf.Tsit5, "RK4": _icnf.RK4}
_ACT_NAMES = {_icnf._lib.ACT_IDENTITY: "identity", _icnf._lib.ACT_TANH: "tanh", _icnf._lib.ACT_SOFTPLUS: "softplus"}


def save_machine(path: str, model: _MLJICNF, fitresult) -> None:
    """Everything needed to rebuild the machine: the flow's configuration (plain numbers and names), the model's
    hyper-parameters and the learned parameter vector.  Custom basedist / epsdist objects and callbacks are not stored."""
    ic = model.icnf
    if ic.basedist is not None or callable(ic.epsdist):
        raise NotImplementedError("save_machine: custom basedist / epsdist objects are not serialised")
    nn = ic.nn
    if nn.planar is not None:
        net = {"planar": [nn.planar.n_in, nn.planar.n_out, _ACT_NAMES[nn.layers[0].act_id], bool(nn.planar.use_bias)]}
    else:
        net = {"dense": [[l.n_in, l.n_out, _ACT_NAMES[l.act_id]] for l in nn.layers]}
    sol = {k: v for k, v in ic.sol_kwargs.items() if isinstance(v, (int, float, bool))}
    ic._solver()
    sol["alg"] = type(ic.sol_kwargs["alg"]).__name__
    ps, st = fitresult
    torch.save({
        "format": "cnf_amd.machine.v1", "conditioned": bool(model._conditioned),
        "icnf": {"nvariables": ic.nvariables, "naugments": ic.naugments, "nconditions": ic.nconditions,
                 "autonomous": ic.autonomous, "inplace": ic.inplace, "tspan": list(ic.tspan), "steer_rate": ic.steer_rate,
                 "lambda1": ic.lambda1, "lambda2": ic.lambda2, "lambda3": ic.lambda3, "nprobes": ic.nprobes,
                 "epsdist": ic.epsdist, "jacvec": bool(ic.compute_mode.jacvec), "net": net, "sol_kwargs": sol},
        "model": {"batchsize": model.batchsize, "epochs": model.epochs, "weight_decay": model.weight_decay, "eta": model.eta,
                  "beta": list(model.beta), "epsilon": model.epsilon},
        "ps": ps.detach().cpu(), "st": dict(st)}, path)


def load_machine(path: str, device="cuda:0"):
    """-> (model, fitresult) as `fit` returned them; the flow is rebuilt on `device`."""
    blob = torch.load(path, map_location="cpu", weights_only=True)
    if blob.get("format") != "cnf_amd.machine.v1":
        raise ValueError("load_machine: not a machine file of this package")
    c = dict(blob["icnf"])
    net = c.pop("net")
    if "planar" in net:
        n_in, n_out, act, use_bias = net["planar"]
        nn = _icnf.Chain(_icnf.PlanarLayer(n_in, n_out, act, use_bias=use_bias))
    else:
        nn = _icnf.Chain(*[_icnf.Dense(a, b, act) for a, b, act in net["dense"]])
    sol = dict(c.pop("sol_kwargs"))
    sol["alg"] = _ALG_NAMES[sol["alg"]]()
    cm = (_icnf.HIPJacVecMatrixMode if c.pop("jacvec") else _icnf.HIPVecJacMatrixMode)()
    c["tspan"] = tuple(c["tspan"])
    icnf = ICNF(nn=nn, compute_mode=cm, sol_kwargs=sol, device=device, **c)
    m = dict(blob["model"])
    m["beta"] = tuple(m["beta"])
    model = (CondICNFModel if blob["conditioned"] else ICNFModel)(icnf=icnf, **m)
    ps = blob["ps"]
    if torch.cuda.is_available():        # (a box without a GPU can still inspect the file; ps then stays on the host)
        ps = ps.to(icnf.device)
    return model, (ps, dict(blob["st"]))
