# Column shards over the GPUs of one node: the one exchange step of the path is the mean in `loss`
# (src/core/icnf.jl:636).  The reference has no multi-device path; a sharded Julia host (one process per GPU under
# MPI.jl / Distributed.jl, or one process driving several devices) evaluates `inference` on its column block and
# all-reduces the four partial sums + the column count through the library's RCCL communicator.

mutable struct Comm
    ptr::Ptr{Cvoid}
    function Comm(rank::Integer, nranks::Integer, id::Vector{UInt8}; device::Integer = current_device_id())
        length(id) == CNF_COMM_ID_BYTES || error("the RCCL unique id is $CNF_COMM_ID_BYTES bytes")
        r = Ref{Ptr{Cvoid}}(C_NULL)
        GC.@preserve id cnf_check(
            ccall((:cnf_comm_init, libcnf), Cint, (Ref{Ptr{Cvoid}}, Cint, Cint, Ptr{Cvoid}, Cint), r, rank, nranks, pointer(id), device),
        )
        c = new(r[])
        finalizer(c) do x
            x.ptr == C_NULL || ccall((:cnf_comm_destroy, libcnf), Cint, (Ptr{Cvoid},), x.ptr)
            x.ptr = C_NULL
        end
        return c
    end
end

"Rank 0 calls this and ships the 128 bytes to the other ranks (MPI.Bcast!, a file, a socket)."
function comm_unique_id()
    id = zeros(UInt8, CNF_COMM_ID_BYTES)
    GC.@preserve id cnf_check(ccall((:cnf_comm_unique_id, libcnf), Cint, (Ptr{Cvoid},), pointer(id)))
    return id
end

"""
    sharded_loss(comm, icnf, mode, xs_local, ps, st) -> global mean loss on every rank

`loss` (src/core/icnf.jl:628-637) for a batch whose columns are sharded over the ranks of `comm`: each rank runs
`inference` on its block, the five scalars are all-reduced by `cnf_allreduce_loss` (RCCL, 40 bytes).
"""
function sharded_loss(comm::Comm, icnf::ICNF{T, <:HIPMatrixMode}, mode::Mode, xs::AbstractMatrix{<:Real}, ps::Any, st::NamedTuple) where {T <: AbstractFloat}
    logp̂x, (Ė, ṅ, Ȧ) = inference(icnf, mode, xs, ps, st)
    sums = Float32[-sum(logp̂x), sum(Ė), sum(ṅ), sum(Ȧ)]
    out5 = zeros(Float64, 5)
    d_s = DeviceArg(sums)
    b_o = DevBuf(sizeof(out5))
    GC.@preserve sums d_s b_o cnf_check(
        ccall(
            (:cnf_allreduce_loss, libcnf),
            Cint,
            (Ptr{Cvoid}, Ptr{Float32}, Int64, Ptr{Float64}, Ptr{Cvoid}),
            comm.ptr, d_s.ptr, size(xs, 2), Ptr{Float64}(b_o.ptr), C_NULL,
        ),
    )
    download!(out5, b_o)                                                  # hipMemcpy on the null stream: ordered behind the all-reduce
    return T((out5[1] + icnf.λ₁ * out5[2] + icnf.λ₂ * out5[3] + icnf.λ₃ * out5[4]) / out5[5])
end
