# PartitionedArraysHIP.jl -- Julia glue: PartitionedArrays.jl types on top of libpa_hip.so (MI355X / gfx950).
#
# STATUS: shipped as source, NOT executed in this repo's CI -- there is no Julia toolchain in the build image
# (see DESIGN.md "Host language").  It is deliberately thin and mechanical: every method below is a `ccall`
# of one entry point of include/pa_hip.h, and the same entry points are exercised, in the same order, by the
# Python host mirror and its tests (tests/test_gpu_parity.py).
#
# What it plugs into (reference file:line, PartitionedArrays.jl v0.5.7):
#   local vector type  V of PVector{V}        src/p_vector.jl:8-26 (allocate_local_values, own_values, ghost_values)
#   vector cache       p_vector_cache_impl    src/p_vector.jl:451-468
#   ghost exchange     assemble_impl!         src/p_vector.jl:587-612   (used by assemble! :695 and consistent! :747)
#   local SpMV         spmv!, mul!(…,α,β)     src/sparse_utils.jl:609-669, src/p_sparse_matrix.jl:2088
#   BLAS-1             dot, norm, broadcast   src/p_vector.jl:1189-1277 (the local arrays' broadcast is what PVector's calls)
# With these methods defined, the reference's own `mul!(c::PVector,a::PSparseMatrix,b::PVector)`
# (src/p_sparse_matrix.jl:2090-2103) runs unchanged: pack/exchange on the comm stream, own*own on the compute
# stream, wait(t), own*ghost.
module PartitionedArraysHIP

using PartitionedArrays
using LinearAlgebra
using SparseArrays
using SparseMatricesCSR
import MPI

const libpa = get(ENV, "LIBPA_HIP", "libpa_hip.so")

const PA_SEG_OWN, PA_SEG_GHOST, PA_SEG_LOCAL = Cint(0), Cint(1), Cint(2)
const PA_CONSISTENT, PA_ASSEMBLE = Cint(0), Cint(1)

function check(status::Cint)
    status == 0 && return nothing
    error("libpa_hip: " * unsafe_string(ccall((:pa_last_error, libpa), Cstring, ())))
end

# ---------------------------------------------------------------- context (one per process / GPU)
mutable struct HIPContext
    handle::Ptr{Cvoid}
    comm::Ptr{Cvoid}
end
const CTX = Ref{Union{Nothing,HIPContext}}(nothing)
function context(device::Integer=parse(Int, get(ENV, "LOCAL_RANK", "0")))
    if CTX[] === nothing
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:pa_ctx_create, libpa), Cint, (Cint, Ref{Ptr{Cvoid}}), device, h))
        CTX[] = HIPContext(h[], C_NULL)
    end
    CTX[]
end
synchronize() = check(ccall((:pa_ctx_sync, libpa), Cint, (Ptr{Cvoid},), context().handle))

"RCCL communicator: rank = MPI rank = part-1 (src/mpi_array.jl:51); the unique id travels over MPI.bcast."
function init_comm!(comm::MPI.Comm=MPI.COMM_WORLD)
    c = context()
    c.comm != C_NULL && return c.comm
    id = zeros(UInt8, 128)
    MPI.Comm_rank(comm) == 0 && check(ccall((:pa_comm_unique_id, libpa), Cint, (Ptr{UInt8},), id))
    MPI.Bcast!(id, 0, comm)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:pa_comm_create, libpa), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Cint, Cint, Ref{Ptr{Cvoid}}),
                c.handle, id, MPI.Comm_rank(comm), MPI.Comm_size(comm), h))
    c.comm = h[]
end

# ---------------------------------------------------------------- local vector type
mutable struct HIPVector <: AbstractVector{Float64}
    handle::Ptr{Cvoid}
    n_own::Int
    n_ghost::Int
    function HIPVector(n_own::Integer, n_ghost::Integer)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:pa_vec_create, libpa), Cint, (Ptr{Cvoid}, Int64, Int64, Ref{Ptr{Cvoid}}),
                    context().handle, n_own, n_ghost, h))
        v = new(h[], n_own, n_ghost)
        finalizer(x -> ccall((:pa_vec_destroy, libpa), Cint, (Ptr{Cvoid},), x.handle), v)
    end
end
Base.size(v::HIPVector) = (v.n_own + v.n_ghost,)
Base.getindex(::HIPVector, ::Int) = error("scalar indexing of a HIPVector is not allowed; use Array(v)")
function HIPVector(host::Vector{Float64}, n_own::Integer)
    v = HIPVector(n_own, length(host) - n_own)
    check(ccall((:pa_vec_upload, libpa), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64), v.handle, host, 0, length(host)))
    v
end
function Base.Array(v::HIPVector)
    host = Vector{Float64}(undef, length(v))
    check(ccall((:pa_vec_download, libpa), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64), v.handle, host, 0, length(host)))
    host
end

"own_values / ghost_values of a HIPVector: a segment tag instead of a SubArray (src/p_vector.jl:20-26)."
struct HIPSegment <: AbstractVector{Float64}
    parent::HIPVector
    seg::Cint
end
Base.size(s::HIPSegment) = (s.seg == PA_SEG_OWN ? s.parent.n_own : s.parent.n_ghost,)
PartitionedArrays.allocate_local_values(::Type{HIPVector}, indices) =
    HIPVector(own_length(indices), ghost_length(indices))
PartitionedArrays.allocate_local_values(v::HIPVector, ::Type{Float64}, indices) =
    HIPVector(own_length(indices), ghost_length(indices))
PartitionedArrays.own_values(v::HIPVector, indices) = HIPSegment(v, PA_SEG_OWN)
PartitionedArrays.ghost_values(v::HIPVector, indices) = HIPSegment(v, PA_SEG_GHOST)
Base.fill!(s::HIPSegment, x) =
    (check(ccall((:pa_vec_fill, libpa), Cint, (Ptr{Cvoid}, Cint, Float64), s.parent.handle, s.seg, x)); s)
function LinearAlgebra.dot(a::HIPSegment, b::HIPSegment)
    out = Ref{Float64}(0.0)
    check(ccall((:pa_vec_dot, libpa), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Float64}), a.parent.handle, b.parent.handle, out))
    out[]
end

# ---------------------------------------------------------------- BLAS-1 on segments: what a solver loop does to a PVector
# The reference's PVector broadcast (src/p_vector.jl:1216-1277) hands the expression to the LOCAL arrays: `x .+= alpha .* u`
# becomes, per part, materialize!(own_values(x), broadcasted(+, own_values(x), broadcasted(*, alpha, own_values(u)))).
# For HIPSegments that expression is flattened into a linear combination  dest = sum_i coef_i * v_i + constant  and run by
# pa_vec_axpby (y = a*x + b*y, unfused multiply and add: the same roundings as the reference's loop for the statements of a CG
# iteration, HPCG/src/ref_cg.jl:56,64-65: c .+ beta .* u, x .+ alpha .* u, r .- alpha .* c).  Anything that is not a linear
# combination of at most two vectors is refused (no scalar fallback: scalar indexing of device memory is an error).
struct HIPStyle <: Base.Broadcast.AbstractArrayStyle{1} end
HIPStyle(::Val{1}) = HIPStyle()
HIPStyle(::Val{N}) where N = Base.Broadcast.DefaultArrayStyle{N}()
Base.BroadcastStyle(::Type{HIPSegment}) = HIPStyle()

"dest-independent linear form of a broadcast tree: (constant, [(segment, coefficient), ...])"
_lin(x::Number) = (Float64(x), Tuple{HIPSegment,Float64}[])
_lin(x::Base.RefValue{<:Number}) = _lin(x[])
_lin(x::HIPSegment) = (0.0, [(x, 1.0)])
function _lin(bc::Base.Broadcast.Broadcasted)
    f, a = bc.f, map(_lin, bc.args)
    if f === (+) && length(a) == 2
        return (a[1][1] + a[2][1], vcat(a[1][2], a[2][2]))
    elseif f === (-) && length(a) == 2
        return (a[1][1] - a[2][1], vcat(a[1][2], [(v, -c) for (v, c) in a[2][2]]))
    elseif f === (-) && length(a) == 1
        return (-a[1][1], [(v, -c) for (v, c) in a[1][2]])
    elseif f === (*) && length(a) == 2 && (isempty(a[1][2]) || isempty(a[2][2]))
        k, t = isempty(a[1][2]) ? (a[1][1], a[2]) : (a[2][1], a[1])
        return (k * t[1], [(v, k * c) for (v, c) in t[2]])
    elseif f === (/) && length(a) == 2 && isempty(a[2][2])
        return (a[1][1] / a[2][1], [(v, c / a[2][1]) for (v, c) in a[1][2]])
    elseif f === identity && length(a) == 1
        return a[1]
    end
    error("PartitionedArraysHIP: only linear combinations of device vectors are broadcast on the device (got $(f))")
end
_axpby!(y::HIPSegment, a, x::HIPSegment, b) =
    check(ccall((:pa_vec_axpby, libpa), Cint, (Ptr{Cvoid}, Float64, Ptr{Cvoid}, Float64, Cint),
                y.parent.handle, Float64(a), x.parent.handle, Float64(b), y.seg))
function Base.copyto!(dest::HIPSegment, bc::Base.Broadcast.Broadcasted{HIPStyle})
    k, terms = _lin(bc)
    all(t -> t[1].seg == dest.seg, terms) || error("PartitionedArraysHIP: a broadcast mixes own and ghost segments")
    k == 0.0 || isempty(terms) || error("PartitionedArraysHIP: vector + constant is not broadcast on the device")
    merged = Tuple{HIPSegment,Float64}[]                        # equal vectors merge: x .+ x is 2x
    for (v, c) in terms
        i = findfirst(t -> t[1].parent === v.parent, merged)
        i === nothing ? push!(merged, (v, c)) : (merged[i] = (v, merged[i][2] + c))
    end
    mine = findfirst(t -> t[1].parent === dest.parent, merged)
    b = mine === nothing ? 0.0 : merged[mine][2]
    rest = [t for t in merged if t[1].parent !== dest.parent]
    if isempty(rest)
        isempty(merged) ? fill!(dest, k) : _axpby!(dest, 0.0, dest, b)
    elseif length(rest) == 1
        _axpby!(dest, rest[1][2], rest[1][1], b)                 # dest = a*v + b*dest
    elseif length(rest) == 2 && mine === nothing
        _axpby!(dest, rest[1][2], rest[1][1], 0.0)               # dest = a*v, then += c*w
        _axpby!(dest, rest[2][2], rest[2][1], 1.0)
    else
        error("PartitionedArraysHIP: a broadcast of more than two device vectors is not supported")
    end
    dest
end
Base.copyto!(dest::HIPSegment, bc::Base.Broadcast.Broadcasted{<:Base.Broadcast.AbstractArrayStyle{0}}) =
    fill!(dest, _lin(bc)[1])                                     # u .= zero(T)  (HPCG/src/ref_cg.jl:71)
Base.copyto!(dest::HIPSegment, src::HIPSegment) =
    (dest.parent === src.parent || check(ccall((:pa_vec_copy, libpa), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint),
                                               dest.parent.handle, src.parent.handle, dest.seg)); dest)
Base.copy!(dest::HIPVector, src::HIPVector) =
    (check(ccall((:pa_vec_copy, libpa), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint), dest.handle, src.handle, PA_SEG_LOCAL)); dest)
Base.copyto!(dest::HIPVector, src::HIPVector) = copy!(dest, src)
Base.similar(v::HIPVector) = HIPVector(v.n_own, v.n_ghost)
Base.similar(v::HIPVector, ::Type{Float64}) = HIPVector(v.n_own, v.n_ghost)
Base.fill!(v::HIPVector, x) =
    (check(ccall((:pa_vec_fill, libpa), Cint, (Ptr{Cvoid}, Cint, Float64), v.handle, PA_SEG_LOCAL, x)); v)
"norm(own_values(a),p) of LinearAlgebra.norm(a::PVector,p) (src/p_vector.jl:1201-1206); p = 2 on the device."
function LinearAlgebra.norm(s::HIPSegment, p::Real=2)
    p == 2 || error("PartitionedArraysHIP: only the 2-norm is computed on the device")
    sqrt(dot(s, s))
end

# ---------------------------------------------------------------- local matrix type
mutable struct HIPCSR <: AbstractSparseMatrix{Float64,Int32}
    handle::Ptr{Cvoid}
    m::Int
    n::Int
end
Base.size(A::HIPCSR) = (A.m, A.n)
function _adopt(h, m, n)
    A = HIPCSR(h, m, n)
    finalizer(x -> ccall((:pa_csr_destroy, libpa), Cint, (Ptr{Cvoid},), x.handle), A)
end
"Upload a SparseMatrixCSR{1} block as the reference stores it (1-based rowptr/colval): HPCG/src/sparse_matrix.jl:115."
function HIPCSR(A::SparseMatrixCSR{1,Float64,Ti}) where Ti<:Union{Int32,Int64}
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:pa_csr_create, libpa), Cint,
                (Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Ptr{Float64}, Ref{Ptr{Cvoid}}),
                context().handle, size(A, 1), size(A, 2), nnz(A), A.rowptr, A.colval, sizeof(Ti), 1, A.nzval, h))
    _adopt(h[], size(A)...)
end
"Upload the default SparseMatrixCSC storage; converted to CSR on the way (spmv_csc! == spmv_csr! bit for bit)."
function HIPCSR(A::SparseMatrixCSC{Float64,Ti}) where Ti<:Union{Int32,Int64}
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:pa_csr_create_from_csc, libpa), Cint,
                (Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Ptr{Float64}, Ref{Ptr{Cvoid}}),
                context().handle, size(A, 1), size(A, 2), nnz(A), A.colptr, A.rowval, sizeof(Ti), 1, A.nzval, h))
    _adopt(h[], size(A)...)
end

# spmv!(b,A,x) (src/sparse_utils.jl:617-623) and muladd!(b,A,x) = mul!(b,A,x,1,1) (src/p_sparse_matrix.jl:2088)
function PartitionedArrays.spmv!(b::HIPSegment, A::HIPCSR, x::HIPSegment)
    check(ccall((:pa_spmv, libpa), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Float64, Float64),
                A.handle, x.parent.handle, x.seg, b.parent.handle, b.seg, 1.0, 0.0))
    b
end
function LinearAlgebra.mul!(b::HIPSegment, A::HIPCSR, x::HIPSegment, α::Number, β::Number)
    check(ccall((:pa_spmv, libpa), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Float64, Float64),
                A.handle, x.parent.handle, x.seg, b.parent.handle, b.seg, Float64(α), Float64(β)))
    b
end

# ---------------------------------------------------------------- vector assembly cache (the exchange plan)
struct HIPAssemblyCache{A}
    plans::A            # one pa_plan* per part (same back-end array type as the partition)
    reversed::Bool      # reverse(cache) of consistent! (src/p_vector.jl:427-437,748)
end
Base.reverse(c::HIPAssemblyCache) = HIPAssemblyCache(c.plans, !c.reversed)

function PartitionedArrays.p_vector_cache_impl(::Type{HIPVector}, vector_partition, index_partition)
    neighbors_snd, neighbors_rcv = assembly_neighbors(index_partition)
    indices_snd, indices_rcv = assembly_local_indices(index_partition, neighbors_snd, neighbors_rcv)
    plans = map(index_partition, neighbors_snd, neighbors_rcv, indices_snd, indices_rcv) do ids, ns, nr, is, ir
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:pa_plan_create, libpa), Cint,
                    (Ptr{Cvoid}, Int32, Int64, Int32, Ptr{Int32}, Ptr{Int32}, Ptr{Int32},
                     Int32, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Cint, Ref{Ptr{Cvoid}}),
                    context().handle, part_id(ids), local_length(ids),
                    length(ns), ns, is.ptrs, is.data, length(nr), nr, ir.ptrs, ir.data, 1, h))
        h[]
    end
    HIPAssemblyCache(plans, false)
end

_transport!(plans::DebugArray, mode) =        # all parts in this process (src/debug_array.jl:250)
    check(ccall((:pa_exchange_local, libpa), Cint, (Ptr{Ptr{Cvoid}}, Int32, Cint), plans.items, length(plans.items), mode))
_transport!(plans::MPIArray, mode) =          # one part per rank (src/mpi_array.jl:575-614) -> RCCL p2p over xGMI
    check(ccall((:pa_exchange_rccl, libpa), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint), plans.item, init_comm!(plans.comm), mode))

function PartitionedArrays.assemble_impl!(f, vector_partition, cache::HIPAssemblyCache)
    mode = cache.reversed ? PA_CONSISTENT : PA_ASSEMBLE
    (mode == PA_CONSISTENT) == (f === PartitionedArrays.insert) || error("HIP path supports insert (consistent!) and + (assemble!)")
    foreach(vector_partition, cache.plans) do v, p       # pack: src/p_vector.jl:595-599
        check(ccall((:pa_exchange_pack, libpa), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint), p, v.handle, mode))
    end
    _transport!(cache.plans, mode)                        # exchange!: :601
    PartitionedArrays.@fake_async begin                   # wait(t) + unpack: :603-611 (+ ghost zeroing of assemble!: :703-705)
        foreach(vector_partition, cache.plans) do v, p
            check(ccall((:pa_exchange_finish, libpa), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint), p, v.handle, mode))
        end
        nothing
    end
end
# assemble!(o,a) zero-fills the ghosts after the task (src/p_vector.jl:703-705); pa_exchange_finish already did,
# and fill!(::HIPSegment,0) above keeps that line of the reference valid.

# ---------------------------------------------------------------- operator level (optional fast path)
# The methods above already make the reference's mul! body run on the device.  `mul_fused!` queues the same
# pipeline (src/p_sparse_matrix.jl:2090-2142, assembled branch) with ONE ccall per process instead of five:
# pa_mul_all for DebugArray back-ends, pa_mul5 + the RCCL communicator for MPIArray back-ends.
function _matrix_handle(a, plan)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:pa_matrix_create, libpa), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Ptr{Cvoid}}),
                context().handle, a.blocks.own_own.handle, a.blocks.own_ghost.handle, plan, h))
    h[]
end
function mul_fused!(c::PVector, A::PSparseMatrix, b::PVector, α::Real=1.0, β::Real=0.0)
    @assert A.assembled
    plans = b.cache.plans
    ms = map(_matrix_handle, partition(A), plans)
    _mul_fused!(ms, partition(c), partition(b), plans, Float64(α), Float64(β))
    foreach(m -> ccall((:pa_matrix_destroy, libpa), Cint, (Ptr{Cvoid},), m), ms)
    c
end
_mul_fused!(ms::DebugArray, cs, bs, plans, α, β) =
    check(ccall((:pa_mul_all, libpa), Cint, (Ptr{Ptr{Cvoid}}, Int32, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Float64, Float64),
                ms.items, length(ms.items), [v.handle for v in cs.items], [v.handle for v in bs.items], α, β))
_mul_fused!(ms::MPIArray, cs, bs, plans, α, β) =
    check(ccall((:pa_mul5, libpa), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Float64),
                ms.item, init_comm!(ms.comm), cs.item.handle, bs.item.handle, α, β))

# ---------------------------------------------------------------- conversions
"Device twin of a host PVector{Vector{Float64}} whose local ids are [own | ghost] (block partitions)."
function to_hip(v::PVector)
    vals = map(partition(v), partition(axes(v, 1))) do x, ids
        HIPVector(collect(Float64, x), own_length(ids))
    end
    PVector(vals, partition(axes(v, 1)))
end
"Device twin of an assembled, split-format PSparseMatrix (src/p_sparse_matrix.jl:588-627,670-681)."
function to_hip(A::PSparseMatrix)
    @assert A.assembled
    mats = map(partition(A)) do a
        blocks = PartitionedArrays.split_matrix_blocks(HIPCSR(a.blocks.own_own), HIPCSR(a.blocks.own_ghost),
                                                       HIPCSR(a.blocks.ghost_own), HIPCSR(a.blocks.ghost_ghost))
        PartitionedArrays.split_matrix(blocks, a.row_permutation, a.col_permutation)
    end
    PSparseMatrix(mats, partition(axes(A, 1)), partition(axes(A, 2)), true)
end

end # module
