# Training: `Zygote.gradient(ps -> loss(icnf, mode, xs, ps, st), ps)` as MLJ's fit drives it
# (src/exts/mlj_ext/core_icnf.jl:42-51).  Zygote cannot differentiate through a `ccall`; these rules route the pullback to the
# reverse-sweep kernels (cnf_loss_grad_fixed / cnf_loss_grad_grid / cnf_loss_grad_adaptive): the exact gradient of the discrete
# loss, where the reference runs QuadratureAdjoint + ZygoteVJP (src/core/icnf.jl:90-99).  `sensealg` is not consulted.
#
# The rules cover `loss` for MatrixMode, unconditioned (src/core/icnf.jl:628-637) and conditioned (:639-649).

function hip_loss_and_gradient(icnf::ICNF{T, <:HIPMatrixMode}, mode::Mode, xs::AbstractMatrix{<:Real}, ys::Union{Nothing, AbstractMatrix{<:Real}}, ps::Any) where {T <: AbstractFloat}
    # the reverse-sweep kernels start from the terminal costate of the standard normal the forward epilogue fuses; any other
    # `basedist` would get the gradient (and value) of the wrong density - loud failure, as the Python host raises
    is_std_normal(icnf.basedist) || error(
        "HIPMatrixMode: the gradient kernels cover basedist = MvNormal(0, I) (the constructor's default, src/core/icnf.jl:76-79); " *
        "use a reference compute_mode to train with another base distribution",
    )
    h = cached_handle(icnf, mode, ps)
    B = size(xs, 2)
    n_aug_input = n_augments_input(icnf)
    ϵ = base_AT(icnf, icnf.nvariables + n_aug_input, B)                  # drawn where inference_prob draws it (base_icnf.jl:258-259)
    Random.rand!(icnf.rng, icnf.epsdist, ϵ)
    t0, t1 = steer_tspan(icnf, mode)
    p = ComponentArrays.getdata(ps)
    grad = similar(p)
    gx = similar(xs, Float32, icnf.nvariables, B)
    sums = similar(p, 4)
    λ = Float32[icnf.λ₁, icnf.λ₂, icnf.λ₃]
    d_x = DeviceArg(xs)
    d_e = DeviceArg(ϵ)
    d_y = DeviceArg(ys)
    d_g = DeviceArg(grad; out = true)
    d_gx = DeviceArg(gx; out = true)
    d_s = DeviceArg(sums; out = true)
    fixed = fixed_step_args(icnf)
    kw = icnf.sol_kwargs
    if fixed !== nothing
        alg, dt = fixed
        grid = fixed_dt_grid(Float32(t0), Float32(t1), dt)               # the steps cnf_inference_fixed_dt takes
        GC.@preserve xs ys ϵ grad gx sums λ grid d_x d_e d_y d_g d_gx d_s cnf_check(
            ccall(
                (:cnf_loss_grad_grid, libcnf),
                Cint,
                (Ptr{Cvoid}, Cint, Cint, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Int64, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Cvoid}),
                h.ptr, alg, length(grid) - 1, grid, d_x.ptr, d_e.ptr, d_y.ptr, B, λ, d_g.ptr, d_gx.ptr, d_s.ptr, current_stream(xs),
            ),
        )
    else
        # adaptive sol_kwargs (incl. the default VCABM): adaptive Tsit5 solve, accepted steps frozen, discrete adjoint on that grid
        stats = Ref(CnfSolveStats(0, 0, 0, 0))
        GC.@preserve xs ys ϵ grad gx sums λ d_x d_e d_y d_g d_gx d_s cnf_check(
            ccall(
                (:cnf_loss_grad_adaptive, libcnf),
                Cint,
                (Ptr{Cvoid}, Cfloat, Cfloat, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Int64, Cfloat, Cfloat, Cfloat, Cint, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ref{CnfSolveStats}, Ptr{Float32}, Int32, Ptr{Cvoid}),
                h.ptr, Float32(t0), Float32(t1), d_x.ptr, d_e.ptr, d_y.ptr, B, Float32(get(kw, :abstol, 1.0f-4)), Float32(get(kw, :reltol, 1.0f-4)),
                Float32(get(kw, :dt, 0.0f0)), solver_maxiters(icnf), λ, d_g.ptr, d_gx.ptr, d_s.ptr, stats, C_NULL, 0, current_stream(xs),
            ),
        )
    end
    finish!(d_g)
    finish!(d_gx)
    s = Array(finish!(d_s))
    value = (s[1] + icnf.λ₁ * s[2] + icnf.λ₂ * s[3] + icnf.λ₃ * s[4]) / B     # Statistics.mean(-logp̂x + λ₁Ė + λ₂ṅ + λ₃Ȧ)
    return T(value), grad ./ T(B), gx ./ T(B)
end

"Step times of OrdinaryDiffEq's fixed-dt stepping on (t0, t1): steps of dt, a shorter last step (see cnf_integrate_fixed_dt)."
function fixed_dt_grid(t0::Float32, t1::Float32, dt::Float32)
    span = abs(Float64(t1) - Float64(t0))
    adt = abs(Float64(dt))
    tdir = t1 >= t0 ? 1.0 : -1.0
    n = floor(Int, span / adt + 1.0e-9)
    tol = 100 * Float64(eps(Float32)) * max(abs(Float64(t0)), abs(Float64(t1)))
    if span - n * adt > tol && adt - (span - n * adt) <= tol
        n += 1          # a Float32 dt a hair above span / n: the last step is snapped onto t1 (fixed_dt_plan, csrc/cnf_api.hip)
    end
    if span - n * adt <= tol
        n == 0 && return Float32[t0, t1]
        return Float32[[Float64(t0) + (Float64(t1) - Float64(t0)) * i / n for i in 0:(n - 1)]; t1]
    end
    return Float32[[Float64(t0) + tdir * adt * i for i in 0:n]; t1]
end

function ChainRulesCore.rrule(
    ::typeof(loss),
    icnf::ICNF{T, <:HIPMatrixMode},
    mode::Mode,
    xs::AbstractMatrix{<:Real},
    ps::Any,
    st::NamedTuple,
) where {T <: AbstractFloat}
    value, g, gx = hip_loss_and_gradient(icnf, mode, xs, nothing, ps)
    ax = ComponentArrays.getaxes(ps)
    function hip_loss_pullback(ȳ)
        c = T(ChainRulesCore.unthunk(ȳ))
        return (
            ChainRulesCore.NoTangent(),
            ChainRulesCore.NoTangent(),
            ChainRulesCore.NoTangent(),
            c .* gx,
            ComponentArrays.ComponentArray(c .* g, ax),
            ChainRulesCore.NoTangent(),
        )
    end
    return value, hip_loss_pullback
end

function ChainRulesCore.rrule(
    ::typeof(loss),
    icnf::ICNF{T, <:HIPMatrixMode},
    mode::Mode,
    xs::AbstractMatrix{<:Real},
    ys::AbstractMatrix{<:Real},
    ps::Any,
    st::NamedTuple,
) where {T <: AbstractFloat}
    value, g, gx = hip_loss_and_gradient(icnf, mode, xs, ys, ps)
    ax = ComponentArrays.getaxes(ps)
    function hip_cond_loss_pullback(ȳ)
        c = T(ChainRulesCore.unthunk(ȳ))
        return (
            ChainRulesCore.NoTangent(),
            ChainRulesCore.NoTangent(),
            ChainRulesCore.NoTangent(),
            c .* gx,
            ChainRulesCore.NoTangent(),                                   # ys enter through CondLayer closures and are not differentiated (cond_layer.jl)
            ComponentArrays.ComponentArray(c .* g, ax),
            ChainRulesCore.NoTangent(),
        )
    end
    return value, hip_cond_loss_pullback
end
