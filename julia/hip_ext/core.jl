# src/exts/hip_ext/core.jl — MI355X (gfx950) implementation of the MatrixMode hot path behind a new ComputeMode.
#
# Drop this directory into the reference as `src/exts/hip_ext/` and add, after the existing extension includes of
# `src/ContinuousNormalizingFlows.jl` (the pattern of lines `include("exts/mlj_ext/core.jl")` ...):
#
#     include("exts/hip_ext/core.jl")
#     export HIPVecJacMatrixMode, HIPJacVecMatrixMode
#
# Everything here is additive: new types and new, strictly more specific methods of existing generic functions
# (`make_ode_func`, `augmented_f`, `base_sol`, `inference_sol`, an `rrule` for `loss`).  No reference method is
# replaced or made ambiguous — tests/test_julia_binding.py of the MI355X repository checks every signature below
# against the reference's method table textually.  Users opt in with
#
#     icnf = ICNF(; compute_mode = HIPVecJacMatrixMode(ADTypes.AutoZygote()), ...)
#
# and `inference`, `generate`, `loss`, `ICNFModel`, `ICNFDist` run unchanged on top.
#
# The shared library is libcnf_hip.so (C ABI: include/cnf.h).  ENV["CNF_HIP_LIB"] overrides its path.
#
# NOT EXECUTED in the build environment of the MI355X repository (no Julia toolchain there): reviewed against the
# reference sources and checked mechanically for ccall/ABI agreement and dispatch ambiguity only.

include("types.jl")
include("libcnf.jl")
include("handle.jl")
include("hot_path.jl")
include("rrule.jl")
include("comm.jl")
