# The hot path on HIPMatrixMode: boundary A (one dynamics call) and boundary B (the whole solve).
#
# Every method below has the argument types of the reference method it specialises, with `<:HIPMatrixMode` in place of
# `<:MatrixMode` / `<:LuxVecJacMatrixMode` in the ICNF parameter and nothing else changed, so it is strictly more specific
# (never ambiguous):
#   make_ode_func   src/core/base_icnf.jl:62-78
#   augmented_f     src/core/icnf.jl:297-339 (TestMode), :517-559 (TrainMode, VecJac), :561-603 (TrainMode, JacVec)
#   base_sol        src/core/base_icnf.jl:134-140
#   inference_sol   src/core/base_icnf.jl:158-172

"""
The callable `make_ode_func` hands to `ODEFunction` (src/core/base_icnf.jl:62-78), as a struct instead of a closure so
that `base_sol` can reach `nn`, `st`, `ϵ` and `mode` when it runs the whole solve in one library call.
"""
struct HIPODEFunc{ICNFT <: ICNF, MODE <: Mode, NN <: LuxCore.AbstractLuxLayer, ST <: NamedTuple, EPS <: AbstractMatrix}
    icnf::ICNFT
    mode::MODE
    nn::NN
    st::ST
    ϵ::EPS
end

(f::HIPODEFunc)(u::Any, p::Any, t::Any) = augmented_f(u, p, t, f.icnf, f.mode, f.nn, f.st, f.ϵ)
(f::HIPODEFunc)(du::Any, u::Any, p::Any, t::Any) = augmented_f(du, u, p, t, f.icnf, f.mode, f.nn, f.st, f.ϵ)

function make_ode_func(
    icnf::ICNF{T, <:HIPMatrixMode},
    mode::Mode,
    nn::LuxCore.AbstractLuxLayer,
    st::NamedTuple,
    ϵ::AbstractMatrix{T},
) where {T <: AbstractFloat}
    return HIPODEFunc(icnf, mode, nn, st, ϵ)
end

"`ys` of a conditioned flow: `add_conditions_nn` wraps `icnf.nn` in a `CondLayer` (src/core/base_icnf.jl:49-60)."
conditions_of(nn::CondLayer) = nn.ys
conditions_of(::LuxCore.AbstractLuxLayer) = nothing

# ---- boundary A: du = f(u, p, t) ------------------------------------------------------------------------------

function hip_aug_f!(du::AbstractMatrix, u::AbstractMatrix, p::Any, t::Any, icnf::ICNF, mode::Mode, nn::LuxCore.AbstractLuxLayer, ϵ::AbstractMatrix)
    h = cached_handle(icnf, mode, p)
    d_du = DeviceArg(du; out = true)
    d_u = DeviceArg(u)
    d_e = DeviceArg(ϵ)
    d_y = DeviceArg(conditions_of(nn))
    GC.@preserve du u ϵ nn d_du d_u d_e d_y cnf_check(
        ccall(
            (:cnf_aug_f, libcnf),
            Cint,
            (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Cfloat, Ptr{Float32}, Ptr{Float32}, Int64, Ptr{Cvoid}),
            h.ptr, d_du.ptr, d_u.ptr, Float32(t), d_e.ptr, d_y.ptr, size(u, 2), current_stream(u),
        ),
    )
    finish!(d_du)
    return nothing
end

function augmented_f(
    u::Any,
    p::Any,
    t::Any,
    icnf::ICNF{T, <:HIPMatrixMode, false},
    mode::TestMode,
    nn::LuxCore.AbstractLuxLayer,
    st::NamedTuple,
    ϵ::AbstractMatrix{T},
) where {T <: AbstractFloat}
    du = similar(u)
    hip_aug_f!(du, u, p, t, icnf, mode, nn, ϵ)
    return du
end

function augmented_f(
    du::Any,
    u::Any,
    p::Any,
    t::Any,
    icnf::ICNF{T, <:HIPMatrixMode, true},
    mode::TestMode,
    nn::LuxCore.AbstractLuxLayer,
    st::NamedTuple,
    ϵ::AbstractMatrix{T},
) where {T <: AbstractFloat}
    hip_aug_f!(du, u, p, t, icnf, mode, nn, ϵ)
    return nothing
end

function augmented_f(
    u::Any,
    p::Any,
    t::Any,
    icnf::ICNF{T, <:HIPMatrixMode, false},
    mode::TrainMode,
    nn::LuxCore.AbstractLuxLayer,
    st::NamedTuple,
    ϵ::AbstractMatrix{T},
) where {T <: AbstractFloat}
    du = similar(u)
    hip_aug_f!(du, u, p, t, icnf, mode, nn, ϵ)
    return du
end

function augmented_f(
    du::Any,
    u::Any,
    p::Any,
    t::Any,
    icnf::ICNF{T, <:HIPMatrixMode, true},
    mode::TrainMode,
    nn::LuxCore.AbstractLuxLayer,
    st::NamedTuple,
    ϵ::AbstractMatrix{T},
) where {T <: AbstractFloat}
    hip_aug_f!(du, u, p, t, icnf, mode, nn, ϵ)
    return nothing
end

# ---- boundary B: the whole solve ----------------------------------------------------------------------------------

function base_sol(
    icnf::ICNF{T, <:HIPMatrixMode, INPLACE},
    prob::SciMLBase.AbstractODEProblem{<:AbstractMatrix{<:Real}, NTuple{2, T}, INPLACE},
) where {T <: AbstractFloat, INPLACE}
    f = prob.f.f
    fixed = fixed_step_args(icnf)
    if !(f isa HIPODEFunc) || (fixed === nothing && !is_default_vcabm(icnf) && !is_adaptive_tsit5(icnf))
        # any other OrdinaryDiffEq algorithm: SciML drives the steps, each right-hand side is one cnf_aug_f (boundary A)
        sol = SciMLBase.solve(prob; icnf.sol_kwargs...)
        return last(sol.u)
    end
    u0 = prob.u0
    t0, t1 = prob.tspan
    B = size(u0, 2)
    u1 = similar(u0)
    h = cached_handle(icnf, f.mode, prob.p)
    d_u0 = DeviceArg(u0)
    d_u1 = DeviceArg(u1; out = true)
    d_e = DeviceArg(f.ϵ)
    d_y = DeviceArg(conditions_of(f.nn))
    kw = icnf.sol_kwargs
    if fixed !== nothing
        alg, dt = fixed
        GC.@preserve u0 u1 f d_u0 d_u1 d_e d_y cnf_check(
            ccall(
                (:cnf_integrate_fixed_dt, libcnf),
                Cint,
                (Ptr{Cvoid}, Cint, Cfloat, Cfloat, Cfloat, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Int64, Ptr{Float32}, Ptr{Cvoid}),
                h.ptr, alg, dt, Float32(t0), Float32(t1), d_u0.ptr, d_e.ptr, d_y.ptr, B, d_u1.ptr, current_stream(u0),
            ),
        )
    elseif is_default_vcabm(icnf)
        # the reference's default sol_kwargs (alg = VCABM(), reltol = abstol = 1f-4; src/core/icnf.jl:84-89) in one call
        stats = Ref(CnfSolveStats(0, 0, 0, 0))
        GC.@preserve u0 u1 f d_u0 d_u1 d_e d_y cnf_check(
            ccall(
                (:cnf_solve_vcabm, libcnf),
                Cint,
                (Ptr{Cvoid}, Cfloat, Cfloat, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Int64, Cfloat, Cfloat, Cfloat, Cint, Ptr{Float32}, Ref{CnfSolveStats}, Ptr{Float32}, Ptr{Int32}, Int32, Ptr{Cvoid}),
                h.ptr, Float32(t0), Float32(t1), d_u0.ptr, d_e.ptr, d_y.ptr, B, Float32(get(kw, :abstol, 1.0f-4)), Float32(get(kw, :reltol, 1.0f-4)),
                Float32(get(kw, :dt, 0.0f0)), solver_maxiters(icnf), d_u1.ptr, stats, C_NULL, C_NULL, 0, current_stream(u0),
            ),
        )
    else
        stats = Ref(CnfSolveStats(0, 0, 0, 0))
        GC.@preserve u0 u1 f d_u0 d_u1 d_e d_y cnf_check(
            ccall(
                (:cnf_solve_tsit5, libcnf),
                Cint,
                (Ptr{Cvoid}, Cfloat, Cfloat, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Int64, Cfloat, Cfloat, Cfloat, Cint, Ptr{Float32}, Ref{CnfSolveStats}, Ptr{Float32}, Int32, Ptr{Cvoid}),
                h.ptr, Float32(t0), Float32(t1), d_u0.ptr, d_e.ptr, d_y.ptr, B, Float32(get(kw, :abstol, 1.0f-4)), Float32(get(kw, :reltol, 1.0f-4)),
                Float32(get(kw, :dt, 0.0f0)), solver_maxiters(icnf), d_u1.ptr, stats, C_NULL, 0, current_stream(u0),
            ),
        )
    end
    return finish!(d_u1)                                                # = last(sol.u); inference_sol / generate_sol slice it
end

"`basedist` is the standard normal the kernels' epilogue fuses (the constructor's default, src/core/icnf.jl:76-79)."
function is_std_normal(d::Distributions.Distribution)
    return d isa Distributions.MvNormal && iszero(Distributions.mean(d)) && Distributions.cov(d) == LinearAlgebra.I
end

function inference_sol(
    icnf::ICNF{T, <:HIPMatrixMode, INPLACE},
    mode::Mode,
    prob::SciMLBase.AbstractODEProblem{<:AbstractMatrix{<:Real}, NTuple{2, T}, INPLACE},
) where {T <: AbstractFloat, INPLACE}
    f = prob.f.f
    fixed = fixed_step_args(icnf)
    if f isa HIPODEFunc && fixed === nothing && is_std_normal(icnf.basedist) && (is_default_vcabm(icnf) || is_adaptive_tsit5(icnf)) &&
       size(prob.u0, 2) > 0
        # adaptive solver (the reference's default sol_kwargs): u0 assembly, the solve and the epilogue in ONE call - cnf_loss_adaptive
        # also leaves mean(-logp + λ₁Ė + λ₂ṅ + λ₃Ȧ) in `lossv`; the reference's `loss` recomputes it from these outputs
        t0, t1 = prob.tspan
        xs = prob.u0[1:(icnf.nvariables), :]
        B = size(xs, 2)
        logp = similar(xs, B)
        regs = similar(xs, B, 3)
        lossv = similar(xs, 1)
        h = cached_handle(icnf, mode, prob.p)
        d_x = DeviceArg(xs)
        d_e = DeviceArg(f.ϵ)
        d_y = DeviceArg(conditions_of(f.nn))
        d_l = DeviceArg(logp; out = true)
        d_r = DeviceArg(regs; out = true)
        d_v = DeviceArg(lossv; out = true)
        kw = icnf.sol_kwargs
        lambdas = Float64[icnf.λ₁, icnf.λ₂, icnf.λ₃]
        stats = Ref(CnfSolveStats(0, 0, 0, 0))
        GC.@preserve xs logp regs lossv lambdas f d_x d_e d_y d_l d_r d_v cnf_check(
            ccall(
                (:cnf_loss_adaptive, libcnf),
                Cint,
                (Ptr{Cvoid}, Cint, Cfloat, Cfloat, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Int64, Cfloat, Cfloat, Cfloat, Cint, Ptr{Float64}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ref{CnfSolveStats}, Ptr{Float32}, Ptr{Int32}, Int32, Ptr{Cvoid}),
                h.ptr, is_default_vcabm(icnf) ? Cint(2) : Cint(1), Float32(t0), Float32(t1), d_x.ptr, d_e.ptr, d_y.ptr, B,
                Float32(get(kw, :abstol, 1.0f-4)), Float32(get(kw, :reltol, 1.0f-4)), Float32(get(kw, :dt, 0.0f0)), solver_maxiters(icnf),
                pointer(lambdas), d_v.ptr, C_NULL, d_l.ptr, d_r.ptr, stats, C_NULL, C_NULL, 0, current_stream(xs),
            ),
        )
        finish!(d_l)
        finish!(d_r)
        return (logp, eachcol(regs))
    end
    if !(f isa HIPODEFunc) || fixed === nothing || !is_std_normal(icnf.basedist)
        # the reference's own epilogue on the final state of `base_sol` (src/core/base_icnf.jl:158-172)
        n_aug = n_augments(icnf, mode)
        fsol = base_sol(icnf, prob)
        z = fsol[begin:(end - n_aug - 1), :]
        Δlogp = fsol[(end - n_aug), :]
        augs = fsol[(end - n_aug + 1):end, :]
        logpz = oftype(Δlogp, Distributions.logpdf(icnf.basedist, z))
        Ȧ = permutedims(reg_z_aug(icnf, mode, z))
        return (logpz - Δlogp, eachrow(vcat(augs, Ȧ)))
    end
    # fused: u0 = [x; 0] assembled in registers, solve, logp̂x = log N(z₁) − Δlogp and Ȧ in the kernel epilogue
    alg, dt = fixed
    t0, t1 = prob.tspan
    xs = prob.u0[1:(icnf.nvariables), :]                                  # u0 = vcat(xs, zrs), base_icnf.jl:256-266
    B = size(xs, 2)
    logp = similar(xs, B)
    regs = similar(xs, B, 3)                                             # column-major: [Ė (B) | ṅ (B) | Ȧ (B)]
    h = cached_handle(icnf, mode, prob.p)
    d_x = DeviceArg(xs)
    d_e = DeviceArg(f.ϵ)
    d_y = DeviceArg(conditions_of(f.nn))
    d_l = DeviceArg(logp; out = true)
    d_r = DeviceArg(regs; out = true)
    GC.@preserve xs logp regs f d_x d_e d_y d_l d_r cnf_check(
        ccall(
            (:cnf_inference_fixed_dt, libcnf),
            Cint,
            (Ptr{Cvoid}, Cint, Cfloat, Cfloat, Cfloat, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Int64, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Cvoid}),
            h.ptr, alg, dt, Float32(t0), Float32(t1), d_x.ptr, d_e.ptr, d_y.ptr, B, d_l.ptr, d_r.ptr, C_NULL, current_stream(xs),
        ),
    )
    finish!(d_l)
    finish!(d_r)
    return (logp, eachcol(regs))                                         # (logp̂x, (Ė, ṅ, Ȧ))
end
