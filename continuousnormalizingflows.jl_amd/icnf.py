"""Host-side mirror of the reference's AbstractICNF interface for the batched (MatrixMode)
fixed-step path, sitting directly on the C ABI of libcnf_hip.so.

The reference is Julia (no Julia toolchain in this image), so this module plays the role the
`HIPMatrixMode` glue of INTEGRATION.md plays there: same names, argument order, array shapes
((rows, B), one column per sample) and error behaviour as the reference's `ICNF`, `inference`,
`generate`, `loss`, `TrainMode`/`TestMode` and ComputeMode types, so the parity tests read like
the reference's smoke tests (test/ci_tests/smoke_tests.jl).  torch is used only for device
memory, streams, RNG and torch.distributed.

Reference map (relative to the reference repo):
  ICNF(; ...)                      src/core/icnf.jl:53-141
  TrainMode / TestMode             src/core/types.jl:1-7
  *MatrixMode compute modes        src/core/types.jl:9-35
  inference / generate             src/core/base_icnf.jl:406-465
  loss                             src/core/icnf.jl:628-649
  (icnf)(xs, ps, st)               src/core/base_icnf.jl:509-523
"""
from __future__ import annotations

import ctypes as C
import math
import weakref
from dataclasses import dataclass
from typing import Callable, Dict, Optional, Sequence, Tuple, Union

import torch

from . import _lib
from ._lib import ptr as _ptr, stream_ptr as _stream_ptr
from .solvers import _adaptive_integrate, _solve_errors, _vcabm_integrate  # noqa: F401

__all__ = [
    "ICNF", "TrainMode", "TestMode", "Dense", "Chain", "PlanarLayer", "tanh", "softplus", "identity",
    "HIPVecJacMatrixMode", "HIPJacVecMatrixMode", "LuxVecJacMatrixMode", "LuxJacVecMatrixMode",
    "DIVecJacMatrixMode", "DIJacVecMatrixMode", "Tsit5", "RK4", "VCABM", "setup", "inference", "generate",
    "loss", "augmented_f", "loss_and_gradient",
]


# ---------------------------------------------------------------------------------------
# modes (src/core/types.jl)
# ---------------------------------------------------------------------------------------
class Mode:
    pass


class TestMode(Mode):
    __test__ = False  # not a pytest class

    def __repr__(self):
        return "TestMode()"


class TrainMode(Mode):
    """TrainMode{REG}; TrainMode() == TrainMode{true}() (src/core/types.jl:5-7)."""

    def __init__(self, reg: bool = True):
        self.reg = bool(reg)

    def __repr__(self):
        return f"TrainMode{{{str(self.reg).lower()}}}()"


class ComputeMode:
    jacvec = False


class MatrixMode(ComputeMode):
    """Whole batch in one solve.  `kernel_path` is PATH_AUTO / PATH_SIMT / PATH_MFMA."""

    def __init__(self, adback=None, kernel_path: int = _lib.PATH_AUTO, arith: int = _lib.ARITH_F32):
        self.adback = adback  # accepted for signature parity; AD is hand-written in the kernels
        self.kernel_path = kernel_path
        self.arith = arith    # ARITH_F32 (exact, default) or ARITH_BF16X6 (split-bf16 MFMA)


class HIPVecJacMatrixMode(MatrixMode):
    jacvec = False


class HIPJacVecMatrixMode(MatrixMode):
    jacvec = True


# The reference's MatrixMode names select the same estimators; they resolve to the HIP kernels.
LuxVecJacMatrixMode = DIVecJacMatrixMode = HIPVecJacMatrixMode
LuxJacVecMatrixMode = DIJacVecMatrixMode = HIPJacVecMatrixMode


# ---------------------------------------------------------------------------------------
# network description (Lux.Chain of Lux.Dense, src/core/icnf.jl:67-71)
# ---------------------------------------------------------------------------------------
def identity(x):
    return x


def tanh(x):
    return torch.tanh(x)


def softplus(x):
    return torch.nn.functional.softplus(x)


_ACT_IDS = {identity: _lib.ACT_IDENTITY, tanh: _lib.ACT_TANH, softplus: _lib.ACT_SOFTPLUS,
            None: _lib.ACT_IDENTITY, "identity": _lib.ACT_IDENTITY, "tanh": _lib.ACT_TANH,
            "softplus": _lib.ACT_SOFTPLUS}


@dataclass
class Dense:
    """Dense(in => out, activation)."""
    n_in: int
    n_out: int
    activation: Union[Callable, str, None] = identity

    @property
    def act_id(self) -> int:
        try:
            return _ACT_IDS[self.activation]
        except KeyError:
            raise TypeError(f"MethodError: no HIP kernel for activation {self.activation!r} "
                            "(supported: identity, tanh, softplus)") from None


@dataclass
class PlanarLayer:
    """PlanarLayer(in => out, activation): z -> u * activation(w' z + b)  (src/layers/planar_layer.jl:6-77).
    It is the Dense chain Dense(in => 1, activation) -> Dense(1 => out) with the second bias pinned to zero,
    so it runs on the same kernels; only the parameter order differs: ps = (u (out), w (in), b (1))
    (src/layers/planar_layer.jl:36-50).  Use it as `nn=Chain(PlanarLayer(...))` like the reference's tests
    (test/ci_tests/smoke_tests.jl:32-46)."""
    n_in: int
    n_out: int
    activation: Union[Callable, str, None] = identity
    use_bias: bool = True


class Chain:
    def __init__(self, *layers):
        if not layers:
            raise ValueError("Chain needs at least one Dense layer")
        self.planar = None
        if len(layers) == 1 and isinstance(layers[0], PlanarLayer):
            pl = layers[0]
            self.planar = pl
            layers = (Dense(pl.n_in, 1, pl.activation), Dense(1, pl.n_out, identity))
        elif any(isinstance(l, PlanarLayer) for l in layers):
            raise TypeError("MethodError: PlanarLayer is supported as the only layer of the Chain")
        for a, b in zip(layers[:-1], layers[1:]):
            if a.n_out != b.n_in:
                raise ValueError(f"DimensionMismatch: Dense({a.n_in}=>{a.n_out}) followed by "
                                 f"Dense({b.n_in}=>{b.n_out})")
        self.layers = tuple(layers)

    @property
    def widths(self):
        return [self.layers[0].n_in] + [l.n_out for l in self.layers]

    def param_offsets(self):
        """ComponentArray layout: layer_k.weight (out x in, column-major), layer_k.bias.
        For a PlanarLayer: (u, w, b); the ABI view appends the pinned zero bias (and a zero b when
        use_bias = false) behind the user's vector — see abi_params."""
        if self.planar is not None:
            pl = self.planar
            n_user = pl.n_out + pl.n_in + (1 if pl.use_bias else 0)
            b0 = pl.n_out + pl.n_in if pl.use_bias else n_user + pl.n_out   # b, or an appended zero
            return [pl.n_out, 0], [b0, n_user], n_user
        w_off, b_off, o = [], [], 0
        for l in self.layers:
            w_off.append(o)
            o += l.n_in * l.n_out
            b_off.append(o)
            o += l.n_out
        return w_off, b_off, o

    def abi_params(self, ps: torch.Tensor) -> torch.Tensor:
        """The vector handed to cnf_set_params: ps itself for Dense chains; for a PlanarLayer ps followed
        by the zeros that stand for the absent biases."""
        if self.planar is None:
            return ps
        pad = self.planar.n_out + (0 if self.planar.use_bias else 1)
        return torch.cat([ps, torch.zeros(pad, dtype=ps.dtype, device=ps.device)])


# ---------------------------------------------------------------------------------------
# solver selection (sol_kwargs.alg)
# ---------------------------------------------------------------------------------------
class Tsit5:
    alg_id = _lib.ALG_TSIT5


class RK4:
    alg_id = _lib.ALG_RK4


class VCABM:
    """The reference's default algorithm (src/core/icnf.jl:84-89): adaptive order (1..12), adaptive step Adams
    predictor-corrector (OrdinaryDiffEqAdamsBashforthMoulton.VCABM).  Always adaptive here."""
    alg_id = _lib.ALG_VCABM


# ---------------------------------------------------------------------------------------
# ICNF
# ---------------------------------------------------------------------------------------
class _Handle:
    """Owns one cnf_handle (one trace mode / regulariser combination)."""

    def __init__(self, cfg: _lib.CnfConfig):
        self.lib = _lib.load()
        self.ptr = C.c_void_p()
        _lib.check(self.lib.cnf_create(C.byref(self.ptr), C.byref(cfg)))
        self.params_key = None
        self.params_ref = None

    def __del__(self):
        try:
            if self.ptr:
                self.lib.cnf_destroy(self.ptr)
                self.ptr = None
        except Exception:
            pass


class ICNF:
    """Keyword constructor mirroring `ICNF(; ...)` (src/core/icnf.jl:53-103).

    Differences forced by the fixed-step HIP path (all raise instead of silently falling back):
      * `sol_kwargs.alg` is VCABM() (the reference's default when none is given), Tsit5() or RK4().  Tsit5 / RK4 with
        adaptive=False (and `dt` or `nsteps`): the whole solve is one fused launch.  VCABM() and Tsit5() with
        adaptive=True (OrdinaryDiffEq's default; reltol/abstol default 1e-4 as in the reference) step under the
        solver's controller on the host with one device attempt per step (`_vcabm_integrate`, `_adaptive_integrate`).
      * `nn` must be a Chain of Dense layers with identity/tanh/softplus activations.
      * data_type is Float32.
    """

    def __init__(self, *, data_type=torch.float32, compute_mode: Optional[ComputeMode] = None,
                 inplace: bool = False, autonomous: bool = False, device=None, rng=None,
                 tspan: Tuple[float, float] = (0.0, 1.0), nvariables: int = 1,
                 naugments: Optional[int] = None, nconditions: int = 0, n_in: Optional[int] = None,
                 n_out: Optional[int] = None, n_hidden: Optional[int] = None,
                 nn: Optional[Chain] = None, steer_rate: float = 0.1, lambda1: float = 0.01,
                 lambda2: float = 0.01, lambda3: float = 0.01, nprobes: int = 1,
                 basedist=None, epsdist=None, sol_kwargs: Optional[dict] = None, **aliases):
        # accept the reference's unicode keyword names too
        lambda1 = aliases.pop("λ₁", lambda1)
        lambda2 = aliases.pop("λ₂", lambda2)
        lambda3 = aliases.pop("λ₃", lambda3)
        if aliases:
            raise TypeError(f"MethodError: unknown keyword(s) {sorted(aliases)}")
        if data_type is not torch.float32:
            raise TypeError("MethodError: the HIP path computes in Float32 only")
        self.data_type = data_type
        self.compute_mode = compute_mode if compute_mode is not None else HIPVecJacMatrixMode()
        if not isinstance(self.compute_mode, MatrixMode):
            raise TypeError("MethodError: only MatrixMode compute modes are implemented "
                            "(VectorMode solves one sample at a time and is out of scope)")
        self.inplace = bool(inplace)
        self.autonomous = bool(autonomous)
        self.device = torch.device(device if device is not None else "cuda:0")
        if self.device.type != "cuda":
            raise TypeError("MethodError: HIPMatrixMode needs a cuda (ROCm) device; "
                            "there is no CPU fallback")
        self.tspan = (float(tspan[0]), float(tspan[1]))
        self.nvariables = int(nvariables)
        self.naugments = int(nvariables + 1 if naugments is None else naugments)  # icnf.jl:62
        self.nconditions = int(nconditions)
        D = self.nvariables + self.naugments
        n_in = D + (0 if autonomous else 1) + self.nconditions if n_in is None else n_in
        n_out = D if n_out is None else n_out
        n_hidden = 4 * n_in if n_hidden is None else n_hidden
        if nn is None:  # icnf.jl:67-71
            nn = Chain(Dense(n_in, n_hidden, softplus), Dense(n_hidden, n_hidden, softplus),
                       Dense(n_hidden, n_out))
        if not isinstance(nn, Chain):
            raise TypeError("MethodError: nn must be a Chain of Dense layers")
        self.nn = nn
        self.steer_rate = float(steer_rate)
        self.lambda1, self.lambda2, self.lambda3 = float(lambda1), float(lambda2), float(lambda3)
        self.nprobes = int(nprobes)
        # basedist / epsdist (src/core/icnf.jl:76-83): None = MvNormal(0, I), the case the kernels fuse
        # (log N(z) in the epilogue, z ~ N(0, I) in generate, Gaussian probes).  basedist may be any object with
        # `log_prob(z)` and `sample((n,))` on (n, D) tensors (torch.distributions style): logpdf(basedist, z) is
        # then evaluated on the host from the final state (base_icnf.jl:168), and the parameter gradient,
        # whose terminal costate assumes the Gaussian, is refused.  epsdist: None, "rademacher", or a callable
        # (generator, (B, K*D), device) -> tensor.
        self.basedist = basedist
        self.epsdist = epsdist
        if basedist is not None and not (hasattr(basedist, "log_prob") and hasattr(basedist, "sample")):
            raise TypeError("MethodError: basedist needs log_prob(z) and sample((n,))")
        if epsdist is not None and epsdist != "rademacher" and not callable(epsdist):
            raise TypeError("MethodError: epsdist is None, 'rademacher' or a callable")
        # icnf.rng draws the Hutchinson probes (base_icnf.jl:258-259).  When the columns are sharded over ranks each
        # rank must draw DIFFERENT probes for its columns, so the default generator is seeded with the rank; the STEER
        # end time (one draw per call for the whole batch, base_icnf.jl:23-43) comes from a second generator that is
        # seeded identically on every rank, so all shards integrate the same span (drawing it from the
        # per-rank stream desynchronises the ranks).  `sharded=False` makes every call of this ICNF rank-local (no
        # collectives even though a process group exists); None = follow torch.distributed / the installed Comm.
        self.rng = rng
        self.sharded: Optional[bool] = None
        rank = 0
        try:
            import torch.distributed as _d
            if _d.is_available() and _d.is_initialized():
                rank = _d.get_rank()
        except Exception:
            rank = 0
        if self.rng is None and torch.cuda.is_available():
            self.rng = torch.Generator(device=self.device)
            self.rng.manual_seed(rank)
        self.steer_rng = torch.Generator(device="cpu")
        self.steer_rng.manual_seed(0)
        self.sol_kwargs = dict(sol_kwargs or {})
        self._handles: Dict[tuple, _Handle] = {}
        w = nn.widths
        if w[0] != D + (0 if autonomous else 1) + self.nconditions or w[-1] != D:
            raise ValueError(
                f"DimensionMismatch: nn maps {w[0]} => {w[-1]} but the flow needs "
                f"{D + (0 if autonomous else 1) + self.nconditions} => {D}")
        if len(nn.layers) > _lib.MAX_LAYERS:
            raise ValueError(f"at most {_lib.MAX_LAYERS} Dense layers are supported")

    # -- type-parameter style flags (src/core/icnf.jl:105-125) --
    @property
    def D(self) -> int:
        return self.nvariables + self.naugments

    @property
    def S(self) -> int:
        return self.D + 3  # n_augments == 2 always (icnf.jl:143-145) plus dlogp

    @property
    def conditioned(self) -> bool:
        return self.nconditions != 0

    @property
    def augmented(self) -> bool:
        return self.naugments != 0

    def _solver(self):
        kw = self.sol_kwargs
        alg = kw.get("alg")
        if alg is None:   # the reference's default: sol_kwargs = (alg = VCABM(), reltol = abstol = 1e-4, ...) (src/core/icnf.jl:84-89)
            alg = VCABM()
            kw["alg"] = alg
        if not hasattr(alg, "alg_id"):
            raise NotImplementedError("sol_kwargs.alg must be VCABM(), Tsit5() or RK4() (other OrdinaryDiffEq algorithms "
                                      "are not implemented)")
        if kw.get("adaptive", True) and alg.alg_id == _lib.ALG_RK4:
            raise NotImplementedError("adaptive stepping is implemented for VCABM() and Tsit5(); use adaptive=False with RK4()")
        if not kw.get("adaptive", True) and alg.alg_id == _lib.ALG_VCABM:
            raise NotImplementedError("VCABM() is implemented as an adaptive solver only; use Tsit5() or RK4() with adaptive=False")
        return alg.alg_id

    @property
    def adaptive(self) -> bool:
        """OrdinaryDiffEq's default: an algorithm with an embedded pair steps adaptively unless adaptive=false."""
        self._solver()
        return bool(self.sol_kwargs.get("adaptive", True))

    def _nsteps(self, t0: float, t1: float) -> int:
        """`nsteps` equal steps (an extension of this host: sol_kwargs.nsteps); with `dt` the solve goes through the
        `*_fixed_dt` entries instead (see _fixed_dt)."""
        kw = self.sol_kwargs
        if "nsteps" in kw:
            return int(kw["nsteps"])
        if "dt" not in kw:
            raise NotImplementedError("sol_kwargs needs dt (or nsteps) with adaptive=False")
        n = abs(t1 - t0) / float(kw["dt"])
        return max(1, int(round(n)))

    def _fixed_dt(self) -> Optional[float]:
        """sol_kwargs = (alg, adaptive = false, dt): OrdinaryDiffEq's fixed-dt stepping — steps of dt and a shorter last
        step onto t1 (what a STEER-drawn t1 meets, base_icnf.jl:23-43) — served by cnf_inference_fixed_dt /
        cnf_integrate_fixed_dt.  None when the user gave `nsteps` (equal steps) instead."""
        kw = self.sol_kwargs
        if "nsteps" in kw or "dt" not in kw:
            return None
        return C.c_float(abs(float(kw["dt"]))).value       # the Float32 the ABI receives: one plan on both sides

    @staticmethod
    def fixed_dt_grid(t0: float, t1: float, dt: float):
        """The times of those steps, [t0, ..., t1]: the plan of cnf_integrate_fixed_dt (fixed_dt_plan, csrc/cnf_api.hip),
        computed like there in double from the Float32 values of t0, t1 and dt that cross the ABI."""
        t0, t1, dt = (C.c_float(float(v)).value for v in (t0, t1, dt))
        span, adt = abs(t1 - t0), abs(dt)
        tdir = 1.0 if t1 >= t0 else -1.0
        n = int(math.floor(span / adt + 1e-9))
        tol = 100.0 * 1.1920928955078125e-7 * max(abs(t0), abs(t1))
        if span - n * adt > tol and adt - (span - n * adt) <= tol:
            n += 1                       # a Float32 dt a hair above span / n: the last step is snapped onto t1, no tail
        if span - n * adt <= tol:
            return [t0] if n == 0 else [t0 + (t1 - t0) * i / n for i in range(n)] + [t1]
        return [t0 + tdir * adt * i for i in range(n + 1)] + [t1]

    def _steer_tspan(self, mode: Mode) -> Tuple[float, float]:
        """steer_tspan (src/core/base_icnf.jl:23-43)."""
        t0, t1 = self.tspan
        if self.steer_rate != 0.0 and isinstance(mode, TrainMode) and mode.reg:
            r = (torch.rand((), generator=self.steer_rng, dtype=torch.float32).item() * 2.0 - 1.0) * self.steer_rate
            t1 = t1 + abs(t1 - t0) * r
        return t0, t1

    def _group(self, group):
        """The `group` argument the reductions and adaptive loops receive: False (rank-local) when this ICNF opted out."""
        return False if self.sharded is False else group

    def invalidate_params(self) -> None:
        """Force the next call to repack `ps` (cnf_set_params).  The binding is skipped only for the SAME tensor object
        at an unchanged `ps._version`; writes that bypass autograd's version counter (through `ps.data`, a raw pointer,
        another library's kernel) must be followed by this call."""
        for h in self._handles.values():
            h.params_key = None
            h.params_ref = None

    def _handle(self, mode: Mode) -> _Handle:
        train = isinstance(mode, TrainMode)
        if not train and not isinstance(mode, TestMode):
            raise TypeError(f"MethodError: no method for mode {mode!r}")
        reg = train and mode.reg
        if not train:
            tmode = _lib.MODE_EXACT
        else:
            tmode = _lib.MODE_HUTCH_JVP if self.compute_mode.jacvec else _lib.MODE_HUTCH_VJP
        reg_z = int(reg and self.lambda1 != 0.0)               # NORM_Z   (icnf.jl:113)
        reg_j = int(reg and self.lambda2 != 0.0)               # NORM_J   (icnf.jl:114)
        reg_aug = int(reg and self.lambda3 != 0.0 and self.augmented)  # NORM_Z_AUG
        nprobes = self.nprobes if train else 1
        arith = getattr(self.compute_mode, "arith", _lib.ARITH_F32)
        # everything the library bakes into the handle (the reference's ICNF is immutable; here fields may be reassigned)
        key = (tmode, reg_z, reg_j, reg_aug, nprobes, self.compute_mode.kernel_path, arith)
        h = self._handles.get(key)
        if h is None:
            cfg = _lib.CnfConfig()
            cfg.nvars, cfg.naug, cfg.ncond = self.nvariables, self.naugments, self.nconditions
            cfg.autonomous = int(self.autonomous)
            cfg.n_layers = len(self.nn.layers)
            for i, wd in enumerate(self.nn.widths):
                cfg.widths[i] = wd
            for i, l in enumerate(self.nn.layers):
                cfg.acts[i] = l.act_id
            cfg.mode = tmode
            cfg.nprobes = nprobes
            cfg.reg_z, cfg.reg_j, cfg.reg_aug = reg_z, reg_j, reg_aug
            cfg.device_id = self.device.index or 0
            cfg.kernel_path = self.compute_mode.kernel_path
            cfg.arith = arith
            h = _Handle(cfg)
            self._handles[key] = h
        return h

    def _bind_params(self, h: _Handle, ps: torch.Tensor):
        w_off, b_off, n = self.nn.param_offsets()
        if ps.dtype != torch.float32 or ps.dim() != 1 or ps.numel() != n:
            raise ValueError(f"DimensionMismatch: ps must be a Float32 vector of length {n}")
        if ps.is_cuda and ps.device != self.device:
            raise ValueError(f"ps lives on {ps.device} but this ICNF is bound to {self.device}")
        # skip the repack only for the SAME tensor object at the same version: a data_ptr alone can be a freed
        # tensor's address handed out again by the caching allocator
        key = (ps.data_ptr(), ps._version, str(ps.device))
        same = h.params_ref is not None and h.params_ref() is ps
        if same and h.params_key == key:
            return
        ps_c = self.nn.abi_params(ps).contiguous()
        wo = (C.c_size_t * len(w_off))(*w_off)
        bo = (C.c_size_t * len(b_off))(*b_off)
        _lib.check(h.lib.cnf_set_params(h.ptr, _ptr(ps_c), ps_c.numel(), wo, bo, int(ps_c.is_cuda),
                                        _stream_ptr(self.device)))
        h.params_key = key
        h.params_ref = weakref.ref(ps)

    def kernel_path(self, mode: Mode) -> int:
        h = self._handle(mode)
        return int(h.lib.cnf_kernel_path(h.ptr))

    def grad_path(self, mode: Mode, B: Optional[int] = None, alg: int = _lib.ALG_TSIT5, on_grid: bool = False) -> int:
        """0: no gradient for this mode; 1: fused reverse-sweep kernel; 2: layer-wise path; 3: cooperative reverse sweep.
        Without `B` the handle's batch-independent hint (cnf_grad_path); with it, the implementation a call of B columns with
        `alg` on uniform steps / on a grid takes (cnf_grad_path_for)."""
        h = self._handle(mode)
        if B is None:
            return int(h.lib.cnf_grad_path(h.ptr))
        return int(h.lib.cnf_grad_path_for(h.ptr, int(B), int(alg), int(on_grid)))

    def grad_form(self, mode: Mode, B: int, alg: int = _lib.ALG_TSIT5, nsteps: int = 40, on_grid: bool = False) -> int:
        """Which form of the cooperative reverse sweep a call takes (cnf_grad_form_for): 0 none, 1 the sweeps that recompute both
        first-order chains, 2 the second form (the forward solve's stage store + the second-order sweep + products over tiles)."""
        h = self._handle(mode)
        return int(h.lib.cnf_grad_form_for(h.ptr, int(B), int(alg), int(nsteps), int(on_grid)))

    def kernel_family(self, mode: Mode, B: Optional[int] = None, whole_solve: bool = True) -> str:
        """Which kernel organisation serves this mode's handle ("per_wave", "coop", "coopx", "layered", "simt"), or - with `B` -
        a call of B columns ("tile_split" for small whole solves of per-wave shapes): cnf_kernel_family / cnf_kernel_family_for."""
        h = self._handle(mode)
        rc = int(h.lib.cnf_kernel_family(h.ptr)) if B is None else int(h.lib.cnf_kernel_family_for(h.ptr, int(B), int(whole_solve)))
        if rc < 0:
            _lib.check(rc)
        return _lib.FAMILY_NAMES[rc]

    def kernel_name(self, mode: Mode) -> str:
        h = self._handle(mode)
        return h.lib.cnf_kernel_name(h.ptr).decode()

    def repack_on_device(self, mode: Mode) -> bool:
        """True when the last parameter binding of this mode's handle was repacked by the device gather
        kernels (no host round trip) — cnf_repack_on_device."""
        h = self._handle(mode)
        rc = int(h.lib.cnf_repack_on_device(h.ptr))
        if rc < 0:
            _lib.check(rc)
        return bool(rc)

    # Lux-layer call (src/core/base_icnf.jl:509-523)
    def __call__(self, xs, ps, st):
        if self.conditioned:
            x, y = xs
            return inference(self, TrainMode(False), x, y, ps, st)[0], st
        return inference(self, TrainMode(False), xs, ps, st)[0], st


# ---------------------------------------------------------------------------------------
# helpers
# ---------------------------------------------------------------------------------------
def setup(rng: Optional[torch.Generator], icnf: ICNF):
    """LuxCore.setup + ComponentArray: flat Float32 parameter vector (glorot-uniform weights,
    zero biases — Lux.Dense defaults) and an empty state."""
    if icnf.nn.planar is not None:   # (u, w, b): glorot-uniform vectors, zero bias (planar_layer.jl:36-50)
        pl = icnf.nn.planar
        u = (torch.rand(pl.n_out, generator=rng, dtype=torch.float64) * 2 - 1) * math.sqrt(6.0 / (pl.n_out + 1))
        w = (torch.rand(pl.n_in, generator=rng, dtype=torch.float64) * 2 - 1) * math.sqrt(6.0 / (pl.n_in + 1))
        b = torch.zeros(1 if pl.use_bias else 0, dtype=torch.float64)
        return torch.cat([u, w, b]).to(torch.float32), {}
    parts = []
    for l in icnf.nn.layers:
        lim = math.sqrt(6.0 / (l.n_in + l.n_out))
        W = (torch.rand(l.n_out, l.n_in, generator=rng, dtype=torch.float64) * 2 - 1) * lim
        parts.append(W.t().reshape(-1))  # column-major (out x in)
        parts.append(torch.zeros(l.n_out, dtype=torch.float64))
    return torch.cat(parts).to(torch.float32), {}


def _colmajor(a: torch.Tensor, rows: int, name: str, device) -> torch.Tensor:
    """(rows, B) tensor -> Julia memory layout (B x rows contiguous), on the device."""
    if a.dim() != 2 or a.shape[0] != rows:
        raise ValueError(f"DimensionMismatch: {name} must be ({rows}, B), got {tuple(a.shape)}")
    if a.device != device:
        raise ValueError(f"{name} must live on {device} (got {a.device})")
    return a.to(torch.float32).t().contiguous()


def _draw_eps(icnf: ICNF, K: int, B: int) -> torch.Tensor:
    """rand!(rng, epsdist, eps) (src/core/base_icnf.jl:258-259): standard normal unless icnf.epsdist says otherwise."""
    shape = (B, K * icnf.D)
    if icnf.epsdist is None:
        return torch.randn(shape, generator=icnf.rng, device=icnf.device, dtype=torch.float32)
    if icnf.epsdist == "rademacher":
        r = torch.randint(0, 2, shape, generator=icnf.rng, device=icnf.device)
        return (2 * r - 1).to(torch.float32)
    e = icnf.epsdist(icnf.rng, shape, icnf.device)
    if tuple(e.shape) != shape:
        raise ValueError(f"DimensionMismatch: epsdist must return a {shape} tensor")
    return e.to(device=icnf.device, dtype=torch.float32).contiguous()


def _split_args(icnf: ICNF, args, what: str):
    """(xs, ps, st) or (xs, ys, ps, st) as in the reference's method table."""
    if icnf.conditioned:
        if len(args) != 4:
            raise TypeError(f"MethodError: {what}(icnf, mode, xs, ys, ps, st) expected for a "
                            "conditioned ICNF")
        return args
    if len(args) != 3:
        raise TypeError(f"MethodError: {what}(icnf, mode, xs, ps, st) expected")
    return args[0], None, args[1], args[2]


def inference(icnf: ICNF, mode: Mode, *args, eps: Optional[torch.Tensor] = None,
              return_state: bool = False, _raw: bool = False, group=None, _sp=None):
    """inference(icnf, mode, xs[, ys], ps, st) -> (logp̂x (B,), (Ė, ṅ, Ȧ)).

    `eps` ((K*D, B)) pins the Hutchinson probes; by default they are drawn from icnf.rng as
    the reference does.  xs is (nvariables, B), ys (nconditions, B).
    `group`: under an ADAPTIVE solver the error norm couples all columns, so when the batch is sharded over ranks the
    solve all-reduces its error sums over `group` (None = the default group / the installed Comm) and every rank must
    make this call; `group=False` (or `icnf.sharded = False`) runs a rank-local solve with no collectives.  Fixed-step
    solves never communicate."""
    xs, ys, ps, st = _split_args(icnf, args, "inference")
    group = icnf._group(group)
    h = icnf._handle(mode)
    icnf._bind_params(h, ps)
    dev = icnf.device
    x = _colmajor(xs, icnf.nvariables, "xs", dev)
    B = x.shape[0]
    y = _colmajor(ys, icnf.nconditions, "ys", dev) if icnf.conditioned else None
    if y is not None and y.shape[0] != B:
        raise ValueError("DimensionMismatch: xs and ys must have the same number of columns")
    train = isinstance(mode, TrainMode)
    K = icnf.nprobes if train else 1
    if eps is None:
        e = _draw_eps(icnf, K, B)  # drawn (and unused) in TestMode too, like base_icnf.jl:258
    else:
        e = _colmajor(eps, K * icnf.D, "eps", dev)
        if e.shape[0] != B:
            raise ValueError("DimensionMismatch: eps must have B columns")
    t0, t1 = icnf._steer_tspan(mode)
    alg = icnf._solver()
    logp = torch.empty(B, device=dev, dtype=torch.float32)
    regs = torch.empty(3, B, device=dev, dtype=torch.float32)
    want_state = return_state or icnf.basedist is not None
    sp = _sp if _sp is not None else _stream_ptr(dev)   # torch's current stream, looked up once per call (5 us a lookup; a small-batch solve is 200)
    if icnf.adaptive:
        u0 = torch.empty(B, icnf.S, device=dev, dtype=torch.float32)
        if B:
            _lib.check(h.lib.cnf_assemble_u0(h.ptr, _ptr(x), B, _ptr(u0), sp))
        uf = _adaptive_integrate(icnf, h, u0, t0, t1, e, y, group=group, _sp=sp)
        if B:
            _lib.check(h.lib.cnf_epilogue(h.ptr, _ptr(uf), B, _ptr(logp), _ptr(regs), sp))
    else:
        uf = torch.empty(B, icnf.S, device=dev, dtype=torch.float32) if want_state else None
        dt = icnf._fixed_dt()
        if dt is not None:
            _lib.check(h.lib.cnf_inference_fixed_dt(h.ptr, alg, dt, t0, t1, _ptr(x), _ptr(e), _ptr(y), B,
                                                    _ptr(logp), _ptr(regs), _ptr(uf), sp))
        else:
            _lib.check(h.lib.cnf_inference_fixed(h.ptr, alg, icnf._nsteps(t0, t1), t0, t1, _ptr(x), _ptr(e), _ptr(y), B,
                                                 _ptr(logp), _ptr(regs), _ptr(uf), sp))
    if icnf.basedist is not None:   # logp̂x = logpdf(basedist, z) - Δlogp (base_icnf.jl:168-169)
        logp = (icnf.basedist.log_prob(uf[:, :icnf.D]) - uf[:, icnf.D]).to(torch.float32)
    if _raw:   # internal: the (3, B) regulariser block as one tensor (no copies on the loss path)
        return logp, regs
    out = (logp, (regs[0], regs[1], regs[2]))
    if return_state:
        return out + (uf.t(),)
    return out


def generate(icnf: ICNF, mode: Mode, *args, z0: Optional[torch.Tensor] = None,
             eps: Optional[torch.Tensor] = None, group=None):
    """generate(icnf, mode, [ys,] ps, st, n) -> (nvariables, n) samples: integrate the base
    sample backwards over the reversed tspan (src/core/base_icnf.jl:351-404, 185-194)."""
    if icnf.conditioned:
        if len(args) != 4:
            raise TypeError("MethodError: generate(icnf, mode, ys, ps, st, n) expected")
        ys, ps, st, n = args
    else:
        if len(args) != 3:
            raise TypeError("MethodError: generate(icnf, mode, ps, st, n) expected")
        ps, st, n = args
        ys = None
    h = icnf._handle(mode)
    icnf._bind_params(h, ps)
    dev = icnf.device
    D, S = icnf.D, icnf.S
    if z0 is None and icnf.basedist is not None:
        z = icnf.basedist.sample((n,)).to(device=dev, dtype=torch.float32).reshape(n, D)
    elif z0 is None:
        z = torch.randn(n, D, generator=icnf.rng, device=dev, dtype=torch.float32)
    else:
        z = _colmajor(z0, D, "z0", dev)
    train = isinstance(mode, TrainMode)
    K = icnf.nprobes if train else 1
    e = _draw_eps(icnf, K, n) if eps is None else _colmajor(eps, K * D, "eps", dev)
    y = _colmajor(ys, icnf.nconditions, "ys", dev) if icnf.conditioned else None
    u0 = torch.zeros(n, S, device=dev, dtype=torch.float32)
    u0[:, :D] = z
    t0, t1 = icnf._steer_tspan(mode)
    alg = icnf._solver()
    if icnf.adaptive:
        u1 = _adaptive_integrate(icnf, h, u0, t1, t0, e, y, group=icnf._group(group))
    else:
        u1 = torch.empty_like(u0)
        dt = icnf._fixed_dt()
        if dt is not None:
            _lib.check(h.lib.cnf_integrate_fixed_dt(h.ptr, alg, dt, t1, t0, _ptr(u0), _ptr(e), _ptr(y), n,
                                                    _ptr(u1), _stream_ptr(dev)))
        else:
            _lib.check(h.lib.cnf_integrate_fixed(h.ptr, alg, icnf._nsteps(t0, t1), t1, t0, _ptr(u0), _ptr(e), _ptr(y), n,
                                                 _ptr(u1), _stream_ptr(dev)))
    return u1[:, :icnf.nvariables].t()


def augmented_f(icnf: ICNF, mode: Mode, u: torch.Tensor, ps: torch.Tensor, t: float,
                eps: Optional[torch.Tensor], ys: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One dynamics call du = f(u, p, t): the closure of make_ode_func
    (src/core/base_icnf.jl:62-78; src/core/icnf.jl:517-536).  u: (S, B)."""
    h = icnf._handle(mode)
    icnf._bind_params(h, ps)
    dev = icnf.device
    um = _colmajor(u, icnf.S, "u", dev)
    B = um.shape[0]
    train = isinstance(mode, TrainMode)
    K = icnf.nprobes if train else 1
    e = None if eps is None else _colmajor(eps, K * icnf.D, "eps", dev)
    y = _colmajor(ys, icnf.nconditions, "ys", dev) if icnf.conditioned else None
    du = torch.empty_like(um)
    _lib.check(h.lib.cnf_aug_f(h.ptr, _ptr(du), _ptr(um), float(t), _ptr(e), _ptr(y), B,
                               _stream_ptr(dev)))
    return du.t()


def loss_sums(icnf: ICNF, mode: Mode, logp: torch.Tensor, regs) -> torch.Tensor:
    """Device-side partial sums [Σ-logp, ΣĖ, Σṅ, ΣȦ] of this rank's columns."""
    h = icnf._handle(mode)
    B = logp.numel()
    r = torch.stack(list(regs)).contiguous() if not (
        isinstance(regs, torch.Tensor) and regs.is_contiguous()) else regs
    sums = torch.zeros(4, device=icnf.device, dtype=torch.float32)
    if B:
        _lib.check(h.lib.cnf_loss_sums(h.ptr, _ptr(logp.contiguous()), _ptr(r), B, _ptr(sums),
                                       _stream_ptr(icnf.device)))
    return sums


def loss_mean(icnf: ICNF, mode: Mode, logp: torch.Tensor, regs, _sp=None) -> torch.Tensor:
    """The scalar loss of an UNSHARDED batch from (logp, regs) in the library's two reduction kernels
    (`cnf_loss_mean`): mean(-logp + λ₁Ė + λ₂ṅ + λ₃Ȧ), combined in float64, returned as a 0-dim float32 tensor."""
    import ctypes as C
    h = icnf._handle(mode)
    B = logp.numel()
    if B == 0:
        return torch.full((), float("nan"), device=icnf.device, dtype=torch.float32)
    r = torch.stack(list(regs)).contiguous() if not (
        isinstance(regs, torch.Tensor) and regs.is_contiguous()) else regs
    out = torch.empty(1, device=icnf.device, dtype=torch.float32)
    lam = (C.c_double * 3)(float(icnf.lambda1), float(icnf.lambda2), float(icnf.lambda3))
    _lib.check(h.lib.cnf_loss_mean(h.ptr, _ptr(logp.contiguous()), _ptr(r), B, lam, None, _ptr(out),
                                   _sp if _sp is not None else _stream_ptr(icnf.device)))
    return out[0]


def _loss_adaptive_one_call(icnf: ICNF, mode: Mode, args, eps, sp):
    """`loss` of an unsharded batch under an adaptive solver as ONE library call (cnf_loss_adaptive: u0 assembly, the VCABM or
    adaptive Tsit5 solve, the epilogue, the mean - the same kernels `inference` + `loss_mean` enqueue, so the same bits).  What it
    saves is host time: at the reference's own batch size (2^10 samples, its PkgBenchmark suite) a solve is ~0.13 ms of kernel and
    the Python / ctypes side of four calls another ~0.03.  Returns None for an empty batch (the general path raises / returns NaN)."""
    xs, ys, ps, st = _split_args(icnf, args, "loss")
    h = icnf._handle(mode)
    icnf._bind_params(h, ps)
    dev = icnf.device
    x = _colmajor(xs, icnf.nvariables, "xs", dev)
    B = x.shape[0]
    if B == 0:
        return None
    y = _colmajor(ys, icnf.nconditions, "ys", dev) if icnf.conditioned else None
    if y is not None and y.shape[0] != B:
        raise ValueError("DimensionMismatch: xs and ys must have the same number of columns")
    K = icnf.nprobes if isinstance(mode, TrainMode) else 1
    if eps is None:
        e = _draw_eps(icnf, K, B)
    else:
        e = _colmajor(eps, K * icnf.D, "eps", dev)
        if e.shape[0] != B:
            raise ValueError("DimensionMismatch: eps must have B columns")
    t0, t1 = icnf._steer_tspan(mode)
    kw = icnf.sol_kwargs
    vcabm = icnf._solver() == _lib.ALG_VCABM
    cap = 4096
    rec = getattr(icnf, "_loss_records", None)
    if rec is None:
        rec = icnf._loss_records = (_lib.SolveStats(), (C.c_float * cap)(), (C.c_int32 * cap)(), (C.c_double * 3)())
    ss, dts, orders, lam = rec
    lam[0], lam[1], lam[2] = float(icnf.lambda1), float(icnf.lambda2), float(icnf.lambda3)
    out = torch.empty(1, device=dev, dtype=torch.float32)
    _solve_errors(lambda: h.lib.cnf_loss_adaptive(
        h.ptr, _lib.ALG_VCABM if vcabm else _lib.ALG_TSIT5, t0, t1, _ptr(x), _ptr(e), _ptr(y), B, float(kw.get("abstol", 1e-4)),
        float(kw.get("reltol", 1e-4)), float(kw["dt"]) if "dt" in kw else 0.0, _lib.clamp_maxiters(kw), lam, _ptr(out), None, None, None,
        C.byref(ss), dts, orders, cap, sp))
    m = min(ss.naccept, cap)
    stats = {"naccept": ss.naccept, "nreject": ss.nreject, "nf": ss.nf, "dts": [float(v) for v in dts[:m]],
             "alg_used": "VCABM" if vcabm else "Tsit5", "controller": "device" if h.lib.cnf_solve_controller(h.ptr) == 1 else "host"}
    if vcabm:
        stats["orders"] = [int(v) for v in orders[:m]]
    icnf.last_solve_stats = stats
    return out[0]


def loss(icnf: ICNF, mode: Mode, *args, eps: Optional[torch.Tensor] = None, group=None):
    """loss(icnf, mode, xs[, ys], ps, st) = mean(-logp̂x + λ₁Ė + λ₂ṅ + λ₃Ȧ)
    (src/core/icnf.jl:628-649).  When torch.distributed is initialised the batch columns are
    taken to be sharded over the ranks of `group`: the four partial sums and the column count
    are all-reduced (RCCL over xGMI on GPUs) and every rank returns the global mean."""
    from .sharding import reduce_loss
    group = icnf._group(group)
    from .sharding import is_sharded
    sp = _stream_ptr(icnf.device)
    if (icnf.adaptive and not is_sharded(group) and icnf.basedist is None and getattr(icnf, "adaptive_policy", "library") == "library"):
        out = _loss_adaptive_one_call(icnf, mode, args, eps, sp)
        if out is not None:
            return out
    logp, regs = inference(icnf, mode, *args, eps=eps, _raw=True, group=group, _sp=sp)
    if not is_sharded(group) and logp.numel():
        return loss_mean(icnf, mode, logp, regs, _sp=sp)       # one process: the mean comes out of the reduction kernels themselves
    sums = loss_sums(icnf, mode, logp, regs)
    return reduce_loss(sums, logp.numel(), (icnf.lambda1, icnf.lambda2, icnf.lambda3), group=group)


def loss_and_gradient(icnf: ICNF, mode: Mode, *args, eps: Optional[torch.Tensor] = None, group=None,
                      wrt_x: bool = False):
    """(loss, dloss/dps) for `loss(icnf, mode, xs, ps, st)` — what `Zygote.gradient` of the
    reference's training objective returns (src/exts/mlj_ext/core_icnf.jl:42-51), here the exact
    gradient of the discrete fixed-step loss, computed by the reverse-sweep HIP kernel.  With
    torch.distributed initialised the column shards' gradients (nparams floats) and loss sums are
    all-reduced (RCCL over xGMI) and every rank returns the global mean and its gradient.
    `wrt_x=True` also returns dloss/dxs, (nvariables, B) for this rank's columns (`DI.gradient` with respect
    to the data in test/ci_tests/smoke_tests.jl): the costate at t0, a by-product of the same sweep.

    Solver substitution (recorded in `icnf.last_solve_stats["alg_used"]` / `["gradient_of"]`): with a fixed-step
    solver the value and gradient are those of exactly the discretisation `loss()` evaluates.  With an ADAPTIVE solver
    the accepted steps of an adaptive *Tsit5* solve are frozen and the discrete solve on that grid is reversed — also
    when `sol_kwargs.alg` is VCABM (the reference's default), whose multistep recurrence has no one-step discrete
    adjoint here.  So under VCABM the returned value is the Tsit5-grid loss, which agrees with `loss()` (VCABM) only
    to the solver tolerance (tests/test_parity_gpu.py bounds the difference); the reference's own gradient
    (QuadratureAdjoint, icnf.jl:90-99) is likewise a separate solve that matches its forward pass to tolerance."""
    from .sharding import global_count, is_sharded, reduce_gradient, reduce_loss
    group = icnf._group(group)
    xs, ys, ps, st = _split_args(icnf, args, "loss_and_gradient")
    if icnf.basedist is not None:
        raise NotImplementedError("loss_and_gradient: the terminal costate assumes basedist = MvNormal(0, I)")
    h = icnf._handle(mode)
    icnf._bind_params(h, ps)
    dev = icnf.device
    x = _colmajor(xs, icnf.nvariables, "xs", dev)
    B = x.shape[0]
    y = _colmajor(ys, icnf.nconditions, "ys", dev) if icnf.conditioned else None
    K = icnf.nprobes if isinstance(mode, TrainMode) else 1      # as `inference`: TestMode ignores the probes
    e = _draw_eps(icnf, K, B) if eps is None else _colmajor(eps, K * icnf.D, "eps", dev)
    t0, t1 = icnf._steer_tspan(mode)
    # the library differentiates with respect to the vector it was given (Chain.abi_params): ps itself, or for a PlanarLayer ps
    # followed by the pinned zero biases, whose gradient entries are dropped below
    n_abi = ps.numel() if icnf.nn.planar is None else icnf.nn.abi_params(ps).numel()
    grad = torch.empty(n_abi, device=dev, dtype=torch.float32)
    gx = torch.zeros(B, icnf.nvariables, device=dev, dtype=torch.float32) if wrt_x else None
    sums = torch.empty(4, device=dev, dtype=torch.float32)
    lam = (C.c_float * 3)(icnf.lambda1, icnf.lambda2, icnf.lambda3)
    if icnf.adaptive and not is_sharded(group) and getattr(icnf, "adaptive_policy", "library") == "library":
        # single process: adaptive Tsit5 solve + frozen-grid gradient in one library call (cnf_loss_grad_adaptive)
        kw = icnf.sol_kwargs
        cap = 4096
        ss, tg = _lib.SolveStats(), (C.c_float * cap)()
        _solve_errors(lambda: h.lib.cnf_loss_grad_adaptive(
            h.ptr, t0, t1, _ptr(x), _ptr(e), _ptr(y), B, float(kw.get("abstol", 1e-4)), float(kw.get("reltol", 1e-4)),
            float(kw["dt"]) if "dt" in kw else 0.0, _lib.clamp_maxiters(kw), lam, _ptr(grad), _ptr(gx), _ptr(sums),
            C.byref(ss), tg, cap, _stream_ptr(dev)))
        ts = [float(v) for v in tg[:min(ss.naccept + 1, cap)]]
        icnf.last_solve_stats = {"naccept": ss.naccept, "nreject": ss.nreject, "nf": ss.nf, "tgrid": ts,
                                 "dts": [b - a for a, b in zip(ts, ts[1:])], "alg_used": "Tsit5",
                                 "gradient_of": "adaptive Tsit5 solve, accepted steps frozen (discrete adjoint on that grid)"}
    elif icnf.adaptive:
        # differentiate the discrete solve on the steps the adaptive solver accepted (frozen grid; the dependence
        # of the step sizes on ps is ignored - the discretise-then-optimise convention)
        u0 = torch.empty(B, icnf.S, device=dev, dtype=torch.float32)
        _lib.check(h.lib.cnf_assemble_u0(h.ptr, _ptr(x), B, _ptr(u0), _stream_ptr(dev)))
        # (also under VCABM: a multistep recurrence has no one-step discrete adjoint here, so training differentiates
        # the adaptive Tsit5 discretisation at the same tolerances - the reference's own QuadratureAdjoint gradient is
        # likewise a separate solve that matches the forward pass only to tolerance)
        _adaptive_integrate(icnf, h, u0, t0, t1, e, y, group=group, _tsit5=True)
        ts = [t0]
        for d in icnf.last_solve_stats["dts"]:
            ts.append(ts[-1] + d)
        ts[-1] = t1
        grid = (C.c_float * len(ts))(*ts)
        icnf.last_solve_stats["tgrid"] = ts
        icnf.last_solve_stats["gradient_of"] = "adaptive Tsit5 solve, accepted steps frozen (discrete adjoint on that grid)"
        _lib.check(h.lib.cnf_loss_grad_grid(h.ptr, _lib.ALG_TSIT5, len(ts) - 1, grid, _ptr(x), _ptr(e), _ptr(y), B, lam,
                                            _ptr(grad), _ptr(gx), _ptr(sums), _stream_ptr(dev)))
    else:
        icnf.last_solve_stats = {"alg_used": "Tsit5" if icnf._solver() == _lib.ALG_TSIT5 else "RK4",
                                 "gradient_of": "the fixed-step solve loss() evaluates (exact discrete adjoint)"}
        dt = icnf._fixed_dt()
        ts = icnf.fixed_dt_grid(t0, t1, dt) if dt is not None else None
        if ts is not None and len(ts) >= 2 and abs((ts[-1] - ts[-2]) - (ts[1] - ts[0])) > 1e-7 * abs(t1 - t0):
            # fixed dt with a shorter last step (a STEER-drawn t1): the same grid as cnf_inference_fixed_dt, reversed exactly
            grid = (C.c_float * len(ts))(*ts)
            icnf.last_solve_stats["tgrid"] = ts
            _lib.check(h.lib.cnf_loss_grad_grid(h.ptr, icnf._solver(), len(ts) - 1, grid, _ptr(x), _ptr(e), _ptr(y), B, lam,
                                                _ptr(grad), _ptr(gx), _ptr(sums), _stream_ptr(dev)))
        else:
            nsteps = (len(ts) - 1) if ts is not None and len(ts) >= 2 else icnf._nsteps(t0, t1)
            _lib.check(h.lib.cnf_loss_grad_fixed(h.ptr, icnf._solver(), nsteps, t0, t1, _ptr(x), _ptr(e),
                                                 _ptr(y), B, lam, _ptr(grad), _ptr(gx), _ptr(sums), _stream_ptr(dev)))
    value = reduce_loss(sums, B, (icnf.lambda1, icnf.lambda2, icnf.lambda3), group=group)
    gps = reduce_gradient(grad[:ps.numel()], B, group=group)
    if not wrt_x:
        return value, gps
    Bg = global_count(B, dev, group)
    return value, gps, (gx / Bg).t()
