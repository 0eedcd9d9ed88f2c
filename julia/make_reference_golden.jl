# make_reference_golden.jl — pin the MI355X path against the REFERENCE ITSELF.
#
# Runs impICNF/ContinuousNormalizingFlows.jl (the reference package, unmodified, on `cpu_device()`) on the committed inputs
# of every fixture in tests/golden/index.json and writes its outputs beside them as tests/golden/ref_<name>.npz, in the
# layout of the oracle-generated fixtures (tests/golden/make_golden.py): du (one dynamics call), logp, E, n, A, u1 (the
# fixed-step solve).  tests/test_reference_golden.py then checks the fp64 oracle, the C restatement and — on the GPU box —
# the HIP kernels against these files; until they exist every parity statement of the repository is "unpinned by the
# reference" (DESIGN.md section 2).
#
# THIS SCRIPT HAS NOT BEEN EXECUTED BY THE BUILDER: the build image has no Julia toolchain and no network.  It is the
# one-command recipe for anyone who has both:
#
#     julia --project=<env with the packages below> julia/make_reference_golden.jl [path/to/repo]
#
# Packages: ContinuousNormalizingFlows (v0.31.0, the reference), OrdinaryDiffEqTsit5, OrdinaryDiffEqLowOrderRK (the fixed-step
# algorithms of the BASELINE configurations; not dependencies of the reference), ADTypes, ComponentArrays, Distributions,
# Lux, LuxCore, NNlib, Random, Zygote, NPZ, JSON.
#
# NPZ.jl return types assumed (NPZ.npzread of a .npz written by numpy.savez): a Dict{String, Any} whose values are Arrays with the
# numpy shape and element type - `p` a Vector{Float32}, `xs` / `eps` / `u` / `ys` Matrix{Float32} (numpy C-order (rows, B) arrays
# arrive as (rows, B) Julia matrices: NPZ reverses the dimension order of the file and permutes back), and the 0-d array `t`
# a 0-dimensional Array{Float32, 0} (or, in some NPZ versions, a scalar): `first(data["t"])` reads both.  An absent key
# (`ys` of an unconditioned fixture) is tested with haskey.
#
# What is injected, and how (no reference code is changed):
#   p    the fixture's flat Float32 parameter vector is copied into the ComponentArray of LuxCore.setup — so the run also
#        TESTS the parameter order the C ABI assumes (layer_k.weight (out x in, column-major), layer_k.bias);
#   eps  `inference_prob` draws the probe with `Random.rand!(icnf.rng, icnf.epsdist, ϵ)` (src/core/base_icnf.jl:258-259);
#        `epsdist = FixedEps(eps)` is a Distributions.jl distribution whose `rand!` copies the fixture's probe matrix;
#   K    Hutchinson with K > 1 probes is not in the reference: by linearity of the augmented rows in the per-probe terms
#        (SURVEY.md section 8(a0)) the K-probe result is the mean over K single-probe runs of the reference;
#   shapes follow test/ci_tests/smoke_tests.jl:10-22 (Matrix inputs, one column per sample) and
#        benchmark/benchmarks.jl:11-19 (ICNF(; ...) + LuxCore.setup + ComponentArray).

import ADTypes, ComponentArrays, Distributions, JSON, Lux, LuxCore, NNlib, NPZ, Random, Zygote
import OrdinaryDiffEqTsit5, OrdinaryDiffEqLowOrderRK
import ContinuousNormalizingFlows
const CNF = ContinuousNormalizingFlows

const ROOT = length(ARGS) >= 1 ? ARGS[1] : normpath(joinpath(@__DIR__, ".."))
const GOLDEN = joinpath(ROOT, "tests", "golden")

"A 'distribution' whose draw is a given matrix: pins the probe vectors of one call."
struct FixedEps{T <: Real} <: Distributions.ContinuousMultivariateDistribution
    data::Matrix{T}
end
Base.length(d::FixedEps) = size(d.data, 1)
Base.eltype(::Type{FixedEps{T}}) where {T} = T
function Distributions._rand!(::Random.AbstractRNG, d::FixedEps, x::AbstractMatrix{<:Real})
    size(x) == size(d.data) || error("FixedEps: asked for $(size(x)), holding $(size(d.data))")
    return copyto!(x, d.data)
end
Distributions._rand!(::Random.AbstractRNG, d::FixedEps, x::AbstractVector{<:Real}) = copyto!(x, view(d.data, :, 1))

const ACTS = Dict(0 => identity, 1 => tanh, 2 => NNlib.softplus)

function build_icnf(meta, eps_k::Matrix{Float32})
    widths = Int.(meta["widths"])
    acts = Int.(meta["acts"])
    nn = Lux.Chain([Lux.Dense(widths[i] => widths[i + 1], ACTS[acts[i]]) for i in 1:length(acts)]...)
    mode_id = Int(meta["mode"])
    cm = mode_id == 1 ? CNF.LuxJacVecMatrixMode(ADTypes.AutoZygote()) : CNF.LuxVecJacMatrixMode(ADTypes.AutoZygote())
    alg = Int(meta["alg"]) == 1 ? OrdinaryDiffEqTsit5.Tsit5() : OrdinaryDiffEqLowOrderRK.RK4()
    nsteps = Int(meta["nsteps"])
    return CNF.ICNF(;
        nvariables = Int(meta["nvars"]),
        naugments = Int(meta["naug"]),
        nconditions = Int(meta["ncond"]),
        autonomous = Bool(meta["autonomous"]),
        nn,
        compute_mode = cm,
        steer_rate = 0.0f0,
        λ₁ = Bool(meta["reg_z"]) ? 1.0f-2 : 0.0f0,
        λ₂ = Bool(meta["reg_j"]) ? 1.0f-2 : 0.0f0,
        λ₃ = Bool(meta["reg_aug"]) ? 1.0f-2 : 0.0f0,
        epsdist = FixedEps(eps_k),
        sol_kwargs = (; alg, adaptive = false, dt = 1.0f0 / nsteps, save_everystep = false),
    )
end

function run_fixture(name, meta)
    data = NPZ.npzread(joinpath(GOLDEN, name * ".npz"))
    p = Float32.(data["p"])
    xs = Float32.(data["xs"])
    eps = Float32.(data["eps"])
    u = Float32.(data["u"])
    t = Float32(first(data["t"]))
    ys = haskey(data, "ys") ? Float32.(data["ys"]) : nothing
    D = Int(meta["nvars"]) + Int(meta["naug"])
    K = Int(meta["nprobes"])
    mode_id = Int(meta["mode"])
    reg = Bool(meta["reg_z"]) || Bool(meta["reg_j"]) || Bool(meta["reg_aug"])
    mode = mode_id == 2 ? CNF.TestMode() : CNF.TrainMode{reg}()
    acc = nothing
    for k in 1:K
        eps_k = eps[((k - 1) * D + 1):(k * D), :]
        icnf = build_icnf(meta, eps_k)
        ps0, st = LuxCore.setup(icnf.rng, icnf)
        ps = ComponentArrays.ComponentArray(ps0)
        length(ps) == length(p) || error("$name: the reference's parameter vector has $(length(ps)) entries, the fixture $(length(p))")
        copyto!(ComponentArrays.getdata(ps), p)
        # one dynamics call: the closure make_ode_func builds (src/core/base_icnf.jl:62-78)
        nn = ys === nothing ? CNF.add_conditions_nn(icnf) : CNF.add_conditions_nn(icnf, ys)
        du = CNF.augmented_f(u, ps, t, icnf, mode, nn, st, eps_k)
        # the fixed-step solve
        if ys === nothing
            logp, (E, n, A) = CNF.inference(icnf, mode, xs, ps, st)
            u1 = CNF.base_sol(icnf, CNF.inference_prob(icnf, mode, xs, ps, st))
        else
            logp, (E, n, A) = CNF.inference(icnf, mode, xs, ys, ps, st)
            u1 = CNF.base_sol(icnf, CNF.inference_prob(icnf, mode, xs, ys, ps, st))
        end
        cur = Dict("du" => Float64.(du), "logp" => Float64.(logp), "E" => Float64.(collect(E)), "n" => Float64.(collect(n)),
                   "A" => Float64.(collect(A)), "u1" => Float64.(u1))
        if acc === nothing
            acc = cur
        else
            for key in keys(acc)
                acc[key] .+= cur[key]
            end
        end
    end
    for key in keys(acc)
        acc[key] ./= K          # rows that do not depend on the probe (z, Ė, Ȧ) are identical in every run: their mean is themselves
    end
    NPZ.npzwrite(joinpath(GOLDEN, "ref_" * name * ".npz"), acc)
    return acc
end

function main()
    index = JSON.parsefile(joinpath(GOLDEN, "index.json"))
    for name in sort(collect(keys(index)))
        acc = run_fixture(name, index[name])
        println(name, ": reference logp[1:3] = ", acc["logp"][1:min(3, end)])
    end
    println("reference package version: ", pkgversion(CNF))
    println("wrote tests/golden/ref_*.npz - now run: python -m pytest tests/test_reference_golden.py (CPU) and -m gpu on an MI355X")
    return nothing
end

main()
