# Resolve the environment of julia/Project.toml on a box with Julia >= 1.10 and a network:
#     julia --project=julia julia/setup_env.jl /path/to/ContinuousNormalizingFlows.jl
# then
#     julia --project=julia julia/make_reference_golden.jl        (writes tests/golden/ref_*.npz; see its header)
# The reference is developed from the given path (v0.31.0: the tree this repository was written against); the four packages the
# reference does not depend on are added by name - their compat bounds are the recipe's own (INTEGRATION.md).
import Pkg
length(ARGS) == 1 || error("usage: julia --project=julia julia/setup_env.jl <path to the reference repository>")
Pkg.develop(; path = ARGS[1])
Pkg.add(["OrdinaryDiffEqTsit5", "OrdinaryDiffEqLowOrderRK", "NPZ", "JSON"])
Pkg.compat("NPZ", "0.4")
Pkg.compat("JSON", "0.21")
Pkg.resolve()
Pkg.instantiate()
Pkg.status()
