# cnf_handle lifetime, configuration from an ICNF, parameter binding.

act_id(::typeof(identity)) = Int32(0)
act_id(::typeof(tanh)) = Int32(1)
act_id(::typeof(NNlib.tanh_fast)) = Int32(1)         # Lux swaps tanh -> tanh_fast on CPU arrays; same function to 3e-7
act_id(::typeof(NNlib.softplus)) = Int32(2)
act_id(f::Any) = error("HIPMatrixMode: no kernel for activation $f (supported: identity, tanh, softplus)")

"The `Lux.Dense` layers of `icnf.nn` (a `Lux.Chain` of `Dense`, src/core/icnf.jl:67-71), in order."
function dense_layers(nn::LuxCore.AbstractLuxLayer)
    nn isa Lux.Chain || error("HIPMatrixMode: nn must be a Lux.Chain of Lux.Dense layers, got $(typeof(nn))")
    ls = collect(values(nn.layers))
    all(l -> l isa Lux.Dense, ls) || error("HIPMatrixMode: every layer of the Chain must be Lux.Dense")
    all(l -> LuxCore.parameterlength(l) == l.in_dims * l.out_dims + l.out_dims, ls) ||
        error("HIPMatrixMode: Dense layers need use_bias = true")
    1 <= length(ls) <= CNF_MAX_LAYERS || error("HIPMatrixMode: 1 to $CNF_MAX_LAYERS Dense layers are supported")
    return ls
end

mutable struct Handle
    ptr::Ptr{Cvoid}
    params_id::UInt                      # objectid of the parameter vector bound last
    params_hash::UInt                    # unused since parameters are bound on every call (kept for layout stability)
    w_off::Vector{Csize_t}
    b_off::Vector{Csize_t}
    function Handle(cfg::CnfConfig)
        r = Ref{Ptr{Cvoid}}(C_NULL)
        cnf_check(ccall((:cnf_create, libcnf), Cint, (Ref{Ptr{Cvoid}}, Ref{CnfConfig}), r, Ref(cfg)))
        h = new(r[], UInt(0), UInt(0), Csize_t[], Csize_t[])
        finalizer(h) do x
            x.ptr == C_NULL || ccall((:cnf_destroy, libcnf), Cint, (Ptr{Cvoid},), x.ptr)
            x.ptr = C_NULL
        end
        return h
    end
end

"HIP device ordinal the handle binds to (extended in amdgpu.jl to follow `AMDGPU.device()`)."
current_device_id() = Int32(parse(Int, get(ENV, "CNF_HIP_DEVICE", "0")))

function cnf_config(
    icnf::ICNF{
        T,
        <:HIPMatrixMode,
        INPLACE,
        CONDITIONED,
        AUTONOMOUS,
        AUGMENTED,
        STEER,
        NORM_Z,
        NORM_J,
        NORM_Z_AUG,
    },
    mode::Mode,
) where {T <: AbstractFloat, INPLACE, CONDITIONED, AUTONOMOUS, AUGMENTED, STEER, NORM_Z, NORM_J, NORM_Z_AUG}
    T === Float32 || error("HIPMatrixMode computes in Float32 (data_type = $T)")
    ls = dense_layers(icnf.nn)
    widths = Int32[ls[1].in_dims; [l.out_dims for l in ls]]
    D = icnf.nvariables + icnf.naugments
    ncond = widths[1] - D - (AUTONOMOUS ? 0 : 1)                         # n_in = D + !autonomous + nconditions (icnf.jl:64)
    ncond >= 0 && widths[end] == D || error("HIPMatrixMode: nn maps $(widths[1]) => $(widths[end]), the flow needs $(D + !AUTONOMOUS + max(ncond, 0)) => $D")
    (ncond > 0) == CONDITIONED || error("HIPMatrixMode: nn input width does not match nconditions")
    reg = mode isa TrainMode{true}
    tmode = mode isa TestMode ? CNF_MODE_EXACT : (icnf.compute_mode isa HIPJacVecMatrixMode ? CNF_MODE_HUTCH_JVP : CNF_MODE_HUTCH_VJP)
    return CnfConfig(
        icnf.nvariables,
        icnf.naugments,
        ncond,
        AUTONOMOUS,
        length(ls),
        ntuple(i -> i <= length(widths) ? widths[i] : Int32(0), 9),
        ntuple(i -> i <= length(ls) ? act_id(ls[i].activation) : Int32(0), 8),
        tmode,
        1,                                                                # the reference draws one probe (base_icnf.jl:258-259)
        reg && NORM_Z,                                                    # icnf.jl:184-199
        reg && NORM_J,                                                    # icnf.jl:229-245
        reg && NORM_Z_AUG && AUGMENTED,                                   # base_icnf.jl:106-122
        current_device_id(),
        0,
        0,
    )
end

# One handle per (icnf, mode type): the ICNF struct is immutable, so its objectid identifies the configuration.
const HANDLES = Dict{Tuple{UInt, DataType}, Handle}()
const HANDLES_LOCK = ReentrantLock()

function cached_handle(icnf::ICNF{T, <:HIPMatrixMode}, mode::Mode, ps::Any) where {T <: AbstractFloat}
    h = lock(HANDLES_LOCK) do
        get!(HANDLES, (objectid(icnf), typeof(mode))) do
            Handle(cnf_config(icnf, mode))
        end
    end
    bind_params!(h, icnf, ps)
    return h
end

"Offsets (0-based) of every layer's weight and bias inside the flat parameter vector, read off the ComponentArray axes."
function param_offsets(icnf::ICNF, ps::ComponentArrays.ComponentArray)
    ls = dense_layers(icnf.nn)
    idx = ComponentArrays.ComponentArray(collect(1:length(ps)), ComponentArrays.getaxes(ps))
    names = keys(icnf.nn.layers)
    w_off = Csize_t[]
    b_off = Csize_t[]
    for (name, l) in zip(names, ls)
        w = getproperty(getproperty(idx, name), :weight)
        b = getproperty(getproperty(idx, name), :bias)
        # Lux.Dense weight is (out x in), column-major: W(o, i) at w_off + (o - 1) + out * (i - 1)
        size(w) == (l.out_dims, l.in_dims) && vec(w) == first(w):(first(w) + length(w) - 1) ||
            error("HIPMatrixMode: unexpected weight layout for $name")
        length(b) == l.out_dims || error("HIPMatrixMode: unexpected bias layout for $name")
        push!(w_off, first(w) - 1)
        push!(b_off, first(b) - 1)
    end
    return w_off, b_off
end

function bind_params!(h::Handle, icnf::ICNF, ps::Any)
    ps isa ComponentArrays.ComponentArray || error("HIPMatrixMode: ps must be the ComponentArray of LuxCore.setup (got $(typeof(ps)))")
    p = ComponentArrays.getdata(ps)
    n = length(p)
    # Bound on EVERY call, host vectors too: an in-place update (an optimiser step, a finite-difference probe) keeps the
    # vector's identity, and no cheap signature of its contents is safe (Base.hash of a large array samples it).  The cost is
    # one copy of a few KiB to 580 KiB and the device-side gather into the operand images (DESIGN.md section 3).
    id = objectid(p)
    sig = UInt(0)
    if isempty(h.w_off)
        h.w_off, h.b_off = param_offsets(icnf, ps)
    end
    if is_device_array(p)       # a device vector is re-gathered every call: one small kernel on the caller's stream
        GC.@preserve p cnf_check(
            ccall(
                (:cnf_set_params, libcnf),
                Cint,
                (Ptr{Cvoid}, Ptr{Float32}, Csize_t, Ptr{Csize_t}, Ptr{Csize_t}, Cint, Ptr{Cvoid}),
                h.ptr, Ptr{Float32}(UInt(pointer(p))), n, h.w_off, h.b_off, 1, current_stream(p),
            ),
        )
    else
        ph = p isa Array{Float32} ? p : convert(Array{Float32}, p)
        GC.@preserve ph cnf_check(
            ccall(
                (:cnf_set_params, libcnf),
                Cint,
                (Ptr{Cvoid}, Ptr{Float32}, Csize_t, Ptr{Csize_t}, Ptr{Csize_t}, Cint, Ptr{Cvoid}),
                h.ptr, pointer(ph), n, h.w_off, h.b_off, 0, C_NULL,
            ),
        )
    end
    h.params_id = id
    h.params_hash = sig
    return nothing
end

"""
    fixed_step_args(icnf) -> (alg_id, dt) or nothing

`sol_kwargs = (alg = Tsit5() | RK4(), adaptive = false, dt = ...)` is the fixed-step case the fused whole-solve kernels
serve.  The algorithm is recognised by type name so that OrdinaryDiffEqTsit5 / OrdinaryDiffEqLowOrderRK need not be
dependencies of the package (they are not dependencies of the reference either, Project.toml).
"""
function fixed_step_args(icnf::ICNF)
    kw = icnf.sol_kwargs
    haskey(kw, :alg) || return nothing
    name = nameof(typeof(kw.alg))
    alg = name === :Tsit5 ? CNF_ALG_TSIT5 : name === :RK4 ? CNF_ALG_RK4 : nothing
    (alg === nothing || get(kw, :adaptive, true) || !haskey(kw, :dt)) && return nothing
    return (alg, Float32(kw.dt))
end

is_default_vcabm(icnf::ICNF) = haskey(icnf.sol_kwargs, :alg) && nameof(typeof(icnf.sol_kwargs.alg)) === :VCABM &&
    get(icnf.sol_kwargs, :adaptive, true)
is_adaptive_tsit5(icnf::ICNF) = haskey(icnf.sol_kwargs, :alg) && nameof(typeof(icnf.sol_kwargs.alg)) === :Tsit5 &&
    get(icnf.sol_kwargs, :adaptive, true)

solver_maxiters(icnf::ICNF) = Cint(min(get(icnf.sol_kwargs, :maxiters, 100_000), typemax(Cint)))


# ---- introspection (include/cnf.h: which kernel organisation serves a handle / a call, which gradient implementation) --------

"`kernel_family(h)`: CNF_FAMILY_* of the handle (simt 0, per_wave 1, coop 2, coopx 3, tile_split 4, layered 5, coopd 6)."
kernel_family(h::Handle) = Int(ccall((:cnf_kernel_family, libcnf), Cint, (Ptr{Cvoid},), h.ptr))

"`kernel_family(h, B; whole_solve = true)`: the family a call of `B` columns takes (a fused solve, or one dynamics call)."
kernel_family(h::Handle, B::Integer; whole_solve::Bool = true) =
    Int(ccall((:cnf_kernel_family_for, libcnf), Cint, (Ptr{Cvoid}, Int64, Cint), h.ptr, B, whole_solve ? 1 : 0))

"`kernel_name(h)`: the kernel instance serving the handle, e.g. `mfma_vjp<HT=4,L=3,...>`."
kernel_name(h::Handle) = unsafe_string(ccall((:cnf_kernel_name, libcnf), Cstring, (Ptr{Cvoid},), h.ptr))

"`grad_path(h, B, alg_id; on_grid = false)`: 1 fused register-accumulator sweep, 2 layer-wise, 3 cooperative sweep (0: none)."
grad_path(h::Handle, B::Integer, alg_id::Integer; on_grid::Bool = false) =
    Int(ccall((:cnf_grad_path_for, libcnf), Cint, (Ptr{Cvoid}, Int64, Cint, Cint), h.ptr, B, alg_id, on_grid ? 1 : 0))

"`grad_form(h, B, alg_id, nsteps; on_grid = false)`: the form of the cooperative sweep a call takes - 0 none, 1 recomputing sweeps, 2 stage store + second-order sweep."
grad_form(h::Handle, B::Integer, alg_id::Integer, nsteps::Integer; on_grid::Bool = false) =
    Int(ccall((:cnf_grad_form_for, libcnf), Cint, (Ptr{Cvoid}, Int64, Cint, Cint, Cint), h.ptr, B, alg_id, nsteps, on_grid ? 1 : 0))

"`build_info()`: the compiler and flags the loaded library was built with."
build_info() = unsafe_string(ccall((:cnf_build_info, libcnf), Cstring, ()))
