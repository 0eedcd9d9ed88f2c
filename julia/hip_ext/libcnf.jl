# Thin `ccall` layer over include/cnf.h, plus the little HIP runtime needed to stage host arrays.

const libcnf = get(ENV, "CNF_HIP_LIB", "libcnf_hip.so")
const libhip = get(ENV, "CNF_HIP_RUNTIME", "libamdhip64.so")

const CNF_MAX_LAYERS = 8
const CNF_MODE_HUTCH_VJP = Int32(0)
const CNF_MODE_HUTCH_JVP = Int32(1)
const CNF_MODE_EXACT = Int32(2)
const CNF_ALG_RK4 = Int32(0)
const CNF_ALG_TSIT5 = Int32(1)
const CNF_COMM_ID_BYTES = 128

"Mirror of `cnf_config` (include/cnf.h): 30 Int32 fields, isbits, passed by reference."
struct CnfConfig
    nvars::Int32
    naug::Int32
    ncond::Int32
    autonomous::Int32
    n_layers::Int32
    widths::NTuple{9, Int32}
    acts::NTuple{8, Int32}
    mode::Int32
    nprobes::Int32
    reg_z::Int32
    reg_j::Int32
    reg_aug::Int32
    device_id::Int32
    kernel_path::Int32
    arith::Int32
end

"Mirror of `cnf_solve_stats` (include/cnf.h)."
struct CnfSolveStats
    naccept::Int32
    nreject::Int32
    nf::Int32
    max_order::Int32
end

function cnf_check(rc::Integer)
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:cnf_last_error, libcnf), Cstring, ()))
    return error("libcnf_hip (status $rc): $msg")
end

function hip_check(rc::Integer, what::AbstractString)
    rc == 0 && return nothing
    return error("HIP runtime: $what failed with code $rc")
end

# ---- device memory for host-resident arrays -------------------------------------------------------------------
# `icnf.device` may be `cpu_device()` (the reference's default): `Array{Float32}` inputs are then staged through
# hipMalloc / hipMemcpy around each call.  With `MLDataDevices.AMDGPUDevice()` the arrays are `ROCArray`s and their
# device pointers are passed straight through (see `devptr` below and the AMDGPU method in amdgpu.jl).

mutable struct DevBuf
    ptr::Ptr{Cvoid}
    nbytes::Csize_t
    function DevBuf(nbytes::Integer)
        r = Ref{Ptr{Cvoid}}(C_NULL)
        hip_check(ccall((:hipMalloc, libhip), Cint, (Ref{Ptr{Cvoid}}, Csize_t), r, max(nbytes, 4)), "hipMalloc")
        b = new(r[], nbytes)
        finalizer(b) do x
            x.ptr == C_NULL || ccall((:hipFree, libhip), Cint, (Ptr{Cvoid},), x.ptr)
            x.ptr = C_NULL
        end
        return b
    end
end

const HIP_MEMCPY_H2D = Cint(1)
const HIP_MEMCPY_D2H = Cint(2)

function upload(a::Array{Float32})
    b = DevBuf(sizeof(a))
    GC.@preserve a hip_check(
        ccall((:hipMemcpy, libhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Cint), b.ptr, pointer(a), sizeof(a), HIP_MEMCPY_H2D),
        "hipMemcpy (host to device)",
    )
    return b
end

function download!(a::Array, b::DevBuf)
    GC.@preserve a hip_check(
        ccall((:hipMemcpy, libhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Cint), pointer(a), b.ptr, sizeof(a), HIP_MEMCPY_D2H),
        "hipMemcpy (device to host)",
    )
    return a
end

"`true` for arrays whose `pointer` is already a device pointer on the handle's GPU (extended for `ROCArray`)."
is_device_array(::Any) = false

"hipStream_t of the caller (NULL = the null stream; extended for AMDGPU.jl tasks)."
current_stream(::Any) = C_NULL

"""
    DeviceArg(a)

A Float32 array as the ABI wants it: `ptr` is a device pointer.  Host arrays are uploaded on construction and — if
created with `out = true` — downloaded again by `finish!`.  `nothing` becomes a NULL pointer.
"""
struct DeviceArg{O, A}
    orig::O                              # the caller's array
    host::A                              # the dense Float32 host copy that is staged (=== orig for Array{Float32})
    buf::Union{Nothing, DevBuf}
    ptr::Ptr{Float32}
    out::Bool
end

function DeviceArg(a::Nothing; out::Bool = false)
    return DeviceArg{Nothing, Nothing}(nothing, nothing, nothing, Ptr{Float32}(C_NULL), false)
end

function DeviceArg(a::AbstractArray{<:Real}; out::Bool = false)
    if is_device_array(a)
        eltype(a) === Float32 || error("HIPMatrixMode computes in Float32; got a device array of $(eltype(a))")
        return DeviceArg{typeof(a), typeof(a)}(a, a, nothing, Ptr{Float32}(UInt(pointer(a))), out)
    end
    h = a isa Array{Float32} ? a : convert(Array{Float32}, a)      # views (xs slices), Float64 data, ...
    b = out ? DevBuf(sizeof(h)) : upload(h)
    return DeviceArg{typeof(a), typeof(h)}(a, h, b, Ptr{Float32}(b.ptr), out)
end

"Copy an output back into the caller's array (no-op for device arrays); returns that array."
function finish!(d::DeviceArg)
    if d.out && d.buf !== nothing
        download!(d.host, d.buf)
        d.host === d.orig || copyto!(d.orig, d.host)
    end
    return d.orig
end
