# PartitionedArraysHIP.jl -- Julia glue: PartitionedArrays.jl types on top of libpa_hip.so (MI355X / gfx950).
#
# STATUS: shipped as source, NOT executed in this repo's CI -- there is no Julia toolchain in the build image
# (see DESIGN.md "Host language").  It is deliberately thin and mechanical: every method below is a `ccall`
# of one entry point of include/pa_hip.h, and the same entry points are exercised, in the same order, by the
# Python host mirror and its tests (tests/test_gpu_*.py).
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
# The device layout is ALWAYS [own | ghost] (what the kernels and the own-value reductions need).  Index partitions whose
# local order is something else -- PermutedLocalIndices (src/p_range.jl:1372: uniform_partition with ghost layers, the
# partitions of test/p_vector_tests.jl:93-124) and hand-made LocalIndices -- carry `l2d`: the 1-based device position of
# every local id.  Whole-vector upload / download then speak the LOCAL order and the exchange plan is built from device
# positions; `nothing` when the local order already is [own | ghost] (block partitions: no copy, no indirection).
mutable struct HIPVector <: AbstractVector{Float64}
    handle::Ptr{Cvoid}
    n_own::Int
    n_ghost::Int
    l2d::Union{Nothing,Vector{Int32}}
    function HIPVector(n_own::Integer, n_ghost::Integer, l2d::Union{Nothing,Vector{Int32}}=nothing)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:pa_vec_create, libpa), Cint, (Ptr{Cvoid}, Int64, Int64, Ref{Ptr{Cvoid}}),
                    context().handle, n_own, n_ghost, h))
        v = new(h[], n_own, n_ghost, l2d)
        finalizer(x -> ccall((:pa_vec_destroy, libpa), Cint, (Ptr{Cvoid},), x.handle), v)
    end
end
"1-based device position of every local id of `indices`, or `nothing` when local ids already are [own | ghost]."
function local_to_device(indices)
    o2l, g2l = own_to_local(indices), ghost_to_local(indices)
    no, ng = length(o2l), length(g2l)
    (o2l == 1:no && g2l == (no + 1):(no + ng)) && return nothing
    l2d = Vector{Int32}(undef, no + ng)
    for (k, l) in enumerate(o2l); l2d[l] = k; end
    for (k, l) in enumerate(g2l); l2d[l] = no + k; end
    l2d
end
Base.size(v::HIPVector) = (v.n_own + v.n_ghost,)
Base.getindex(::HIPVector, ::Int) = error("scalar indexing of a HIPVector is not allowed; use Array(v)")
"Upload local values given in LOCAL order (permuted into the device layout when the partition needs it)."
function upload!(v::HIPVector, host::AbstractVector{<:Real})
    length(host) == length(v) || error("PartitionedArraysHIP: $(length(host)) values for a local vector of $(length(v))")
    dev = Vector{Float64}(undef, length(host))
    if v.l2d === nothing
        copyto!(dev, host)
    else
        for l in eachindex(host); dev[v.l2d[l]] = host[l]; end
    end
    check(ccall((:pa_vec_upload, libpa), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64), v.handle, dev, 0, length(dev)))
    v
end
function HIPVector(host::Vector{Float64}, n_own::Integer, l2d::Union{Nothing,Vector{Int32}}=nothing)
    upload!(HIPVector(n_own, length(host) - n_own, l2d), host)
end
"Local values in LOCAL order."
function Base.Array(v::HIPVector)
    dev = Vector{Float64}(undef, length(v))
    check(ccall((:pa_vec_download, libpa), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64), v.handle, dev, 0, length(dev)))
    v.l2d === nothing ? dev : dev[v.l2d]
end

"own_values / ghost_values of a HIPVector: a segment tag instead of a SubArray (src/p_vector.jl:20-26)."
struct HIPSegment <: AbstractVector{Float64}
    parent::HIPVector
    seg::Cint
end
Base.size(s::HIPSegment) = (s.seg == PA_SEG_OWN ? s.parent.n_own : s.parent.n_ghost,)
PartitionedArrays.allocate_local_values(::Type{HIPVector}, indices) =
    HIPVector(own_length(indices), ghost_length(indices), local_to_device(indices))
PartitionedArrays.allocate_local_values(v::HIPVector, ::Type{Float64}, indices) =
    HIPVector(own_length(indices), ghost_length(indices), local_to_device(indices))
PartitionedArrays.own_values(v::HIPVector, indices) = HIPSegment(v, PA_SEG_OWN)
PartitionedArrays.ghost_values(v::HIPVector, indices) = HIPSegment(v, PA_SEG_GHOST)
Base.fill!(s::HIPSegment, x) =
    (check(ccall((:pa_vec_fill, libpa), Cint, (Ptr{Cvoid}, Cint, Float64), s.parent.handle, s.seg, x)); s)
# dot(a::PVector,b::PVector) reduces dot(own_values(a),own_values(b)) over the parts (src/p_vector.jl:1189-1199): the device
# reduction IS over own values (pa_vec_dot), so any other segment is refused rather than silently answered with the own
# segment's result.
function LinearAlgebra.dot(a::HIPSegment, b::HIPSegment)
    (a.seg == PA_SEG_OWN && b.seg == PA_SEG_OWN) ||
        error("PartitionedArraysHIP: dot / norm are computed on own values only (got a ghost segment)")
    out = Ref{Float64}(0.0)
    check(ccall((:pa_vec_dot, libpa), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Float64}), a.parent.handle, b.parent.handle, out))
    out[]
end

# ---------------------------------------------------------------- BLAS-1 on segments: what a solver loop does to a PVector
# The reference's PVector broadcast (src/p_vector.jl:1216-1277) hands the expression to the LOCAL arrays: `x .+= alpha .* u`
# becomes, per part, materialize!(own_values(x), broadcasted(+, own_values(x), broadcasted(*, alpha, own_values(u)))).
# For HIPSegments that expression is flattened into  dest = a * v (+ b * w)  and run by pa_vec_axpby (y = a*x + b*y, unfused
# multiply and add; b == 0 does not read y).  ONLY forms whose roundings are provably the element-wise loop's are accepted:
#     k .* v   v .* k   -v   v .+ w   v .- w   v .+ k .* w   v .- k .* w   k .* v .+ l .* w        (k, l scalars)
# i.e. every vector is multiplied by ONE scalar at most (k*(l*v), k .* (v .+ w), a .* v .+ b .* v would each round
# differently when flattened) and nothing is divided (v ./ k is not v .* (1/k)).  These are the statements of a CG
# iteration (HPCG/src/ref_cg.jl:56,64-65: c .+ beta .* u, x .+ alpha .* u, r .- alpha .* c).  Everything else is refused
# (no scalar fallback: scalar indexing of device memory is an error).
struct HIPStyle <: Base.Broadcast.AbstractArrayStyle{1} end
HIPStyle(::Val{1}) = HIPStyle()
HIPStyle(::Val{N}) where N = Base.Broadcast.DefaultArrayStyle{N}()
Base.BroadcastStyle(::Type{HIPSegment}) = HIPStyle()

"linear form of a broadcast tree: (constant, [(segment, coefficient, scaled::Bool), ...]); scaled = a scalar was applied"
_lin(x::Number) = (Float64(x), Tuple{HIPSegment,Float64,Bool}[])
_lin(x::Base.RefValue{<:Number}) = _lin(x[])
_lin(x::HIPSegment) = (0.0, [(x, 1.0, false)])
function _lin(bc::Base.Broadcast.Broadcasted)
    f, a = bc.f, map(_lin, bc.args)
    if f === (+) && length(a) == 2
        return (a[1][1] + a[2][1], vcat(a[1][2], a[2][2]))
    elseif f === (-) && length(a) == 2                       # negation is exact: (-k)*w + v == v - k*w bit for bit
        return (a[1][1] - a[2][1], vcat(a[1][2], [(v, -c, sc) for (v, c, sc) in a[2][2]]))
    elseif f === (-) && length(a) == 1
        return (-a[1][1], [(v, -c, sc) for (v, c, sc) in a[1][2]])
    elseif f === (*) && length(a) == 2 && (isempty(a[1][2]) || isempty(a[2][2]))
        k, t = isempty(a[1][2]) ? (a[1][1], a[2]) : (a[2][1], a[1])
        isempty(t[2]) && return (k * t[1], t[2])             # scalar * scalar
        (length(t[2]) == 1 && !t[2][1][3] && t[1] == 0.0) ||
            error("PartitionedArraysHIP: a scalar applied to a sum or to an already scaled vector would round differently " *
                  "from the element-wise loop; write it as k .* v .+ l .* w")
        v, c, _ = t[2][1]
        return (0.0, [(v, k * c, true)])                     # c is +-1 here: k*c is exact
    elseif f === identity && length(a) == 1
        return a[1]
    end
    error("PartitionedArraysHIP: only k .* v, v .+- w and v .+- k .* w are broadcast on the device (got $(f)); " *
          "division is not (v ./ k is not v .* (1/k))")
end
_axpby!(y::HIPSegment, a, x::HIPSegment, b) =
    check(ccall((:pa_vec_axpby, libpa), Cint, (Ptr{Cvoid}, Float64, Ptr{Cvoid}, Float64, Cint),
                y.parent.handle, Float64(a), x.parent.handle, Float64(b), y.seg))
function Base.copyto!(dest::HIPSegment, bc::Base.Broadcast.Broadcasted{HIPStyle})
    k, terms = _lin(bc)
    all(t -> t[1].seg == dest.seg, terms) || error("PartitionedArraysHIP: a broadcast mixes own and ghost segments")
    k == 0.0 || isempty(terms) || error("PartitionedArraysHIP: vector + constant is not broadcast on the device")
    isempty(terms) && return fill!(dest, k)
    length(terms) <= 2 || error("PartitionedArraysHIP: a broadcast of more than two device vectors is not supported")
    if length(terms) == 2 && terms[1][1].parent === terms[2][1].parent
        # the same vector twice: v .+ v and v .- v are exact as 2v and 0v; a .* v .+ b .* v is not (a+b) .* v
        (abs(terms[1][2]) == 1.0 && abs(terms[2][2]) == 1.0) ||
            error("PartitionedArraysHIP: a .* v .+ b .* v would round differently as (a+b) .* v")
        terms[1][2] + terms[2][2] == 0.0 &&
            error("PartitionedArraysHIP: v .- v is not broadcast on the device (0 .* v has the sign of v's zeros wrong); use fill!")
        terms = [(terms[1][1], terms[1][2] + terms[2][2], true)]
    end
    mine = findfirst(t -> t[1].parent === dest.parent, terms)
    if length(terms) == 1
        v, a, _ = terms[1]
        _axpby!(dest, a, v, 0.0)                                 # dest = a*v (b == 0: dest is not read; v may be dest itself)
    elseif mine !== nothing
        other = terms[3 - mine]
        _axpby!(dest, other[2], other[1], terms[mine][2])        # dest = a*v + b*dest
    else
        _axpby!(dest, terms[1][2], terms[1][1], 0.0)             # dest = a*v, then dest = c*w + 1*dest
        _axpby!(dest, terms[2][2], terms[2][1], 1.0)
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
Base.similar(v::HIPVector) = HIPVector(v.n_own, v.n_ghost, v.l2d)
Base.similar(v::HIPVector, ::Type{Float64}) = HIPVector(v.n_own, v.n_ghost, v.l2d)
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
    t::Union{Nothing,HIPCSR}     # transpose(A) as a block of its own, built on the device when first asked for
end
Base.size(A::HIPCSR) = (A.m, A.n)
function _adopt(h, m, n)
    A = HIPCSR(h, m, n, nothing)
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

# transpose(A) of a block resident in HBM (csrc/pa_transpose.hip): the reference's transposed product calls
# `mul!(ch,transpose(aoh),bo,α,1)` / `mul!(co,transpose(aoo),bo,α,1)` on the local blocks (src/p_sparse_matrix.jl:2150-2159)
# and `spmtv!(b,A,x)` (src/sparse_utils.jl:613-615,625-631); both land here.  A' is built once per block, on the device, with
# its rows' entries in ascending row of A -- the order the reference's scatter loop adds them in.
function _transposed(A::HIPCSR)
    if A.t === nothing
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:pa_csr_create_transpose, libpa), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}), A.handle, h))
        A.t = _adopt(h[], A.n, A.m)
    end
    A.t
end
function LinearAlgebra.mul!(b::HIPSegment, At::Transpose{Float64,HIPCSR}, x::HIPSegment, α::Number, β::Number)
    T = _transposed(parent(At))
    check(ccall((:pa_spmv, libpa), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Float64, Float64),
                T.handle, x.parent.handle, x.seg, b.parent.handle, b.seg, Float64(α), Float64(β)))
    b
end
PartitionedArrays.spmtv!(b::HIPSegment, A::HIPCSR, x::HIPSegment) = mul!(b, transpose(A), x, 1.0, 0.0)

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
        l2d = local_to_device(ids)               # the cache's local ids as positions of the device layout [own | ghost]
        dev(lids) = l2d === nothing ? convert(Vector{Int32}, lids) : Int32[l2d[l] for l in lids]
        check(ccall((:pa_plan_create, libpa), Cint,
                    (Ptr{Cvoid}, Int32, Int64, Int32, Ptr{Int32}, Ptr{Int32}, Ptr{Int32},
                     Int32, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Cint, Ref{Ptr{Cvoid}}),
                    context().handle, part_id(ids), local_length(ids),
                    length(ns), ns, is.ptrs, dev(is.data), length(nr), nr, ir.ptrs, dev(ir.data), 1, h))
        h[]
    end
    plans isa MPIArray && PA_TRANSPORT == "ipc" && connect_ipc!(plans)
    HIPAssemblyCache(plans, false)
end

_transport!(plans::DebugArray, mode) =        # all parts in this process (src/debug_array.jl:250)
    check(ccall((:pa_exchange_local, libpa), Cint, (Ptr{Ptr{Cvoid}}, Int32, Cint), plans.items, length(plans.items), mode))
_transport!(plans::MPIArray, mode) =          # one part per rank (src/mpi_array.jl:575-614) -> RCCL p2p over xGMI
    check(ccall((:pa_exchange_rccl, libpa), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint), plans.item, init_comm!(plans.comm), mode))

# PA_TRANSPORT=ipc (csrc/pa_push.hip): instead of pack + RCCL, the pack kernel stores every slice straight into the neighbours'
# receive buffers, mapped over hipIpc.  Once per cache: every rank publishes its plan's blob (ipc handles + slice tables), gathers
# everybody's (MPI.Allgatherv, as init_comm! broadcasts the RCCL id) and opens its neighbours'.
const PA_TRANSPORT = get(ENV, "PA_TRANSPORT", "rccl")
function connect_ipc!(plans::MPIArray)
    plan = plans.item
    n = Ref{Int64}(0)
    check(ccall((:pa_plan_ipc_blob_size, libpa), Cint, (Ptr{Cvoid}, Ref{Int64}), plan, n))
    blob = Vector{UInt8}(undef, n[])
    check(ccall((:pa_plan_ipc_blob, libpa), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Int64), plan, blob, n[]))
    sizes = MPI.Allgather(Int64[n[]], plans.comm)
    everything = MPI.Allgatherv(blob, MPI.VBuffer(Vector{UInt8}(undef, sum(sizes)), Int32.(sizes)), plans.comm)
    offs = cumsum(vcat(0, sizes[1:end-1]))
    ptrs = [pointer(everything, o + 1) for o in offs]
    GC.@preserve everything check(ccall((:pa_plan_ipc_connect, libpa), Cint, (Ptr{Cvoid}, Int32, Ptr{Ptr{UInt8}}, Ptr{Int64}),
                                        plan, length(sizes), ptrs, sizes))
    MPI.Barrier(plans.comm)                      # nobody pushes before everybody has mapped
    nothing
end
_push_ipc!(v, plan, mode) = check(ccall((:pa_exchange_push_ipc, libpa), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint), plan, v.handle, mode))

function PartitionedArrays.assemble_impl!(f, vector_partition, cache::HIPAssemblyCache)
    mode = cache.reversed ? PA_CONSISTENT : PA_ASSEMBLE
    (mode == PA_CONSISTENT) == (f === PartitionedArrays.insert) || error("HIP path supports insert (consistent!) and + (assemble!)")
    if cache.plans isa MPIArray && PA_TRANSPORT == "ipc"  # pack + exchange! in one kernel (the plans were connected when made)
        foreach((v, p) -> _push_ipc!(v, p, mode), vector_partition, cache.plans)
    else
        foreach(vector_partition, cache.plans) do v, p   # pack: src/p_vector.jl:595-599
            check(ccall((:pa_exchange_pack, libpa), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint), p, v.handle, mode))
        end
        _transport!(cache.plans, mode)                    # exchange!: :601
    end
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

# mul!(c,transpose(A),b,α,β) (src/p_sparse_matrix.jl:2144-2162) the same way: ghost(c) = α A_oh' own(b), assemble!(c) under
# own(c) = β own(c) + α A_oo' own(b), one ccall per process (pa_mul5_transpose / _all).  c lives on axes(A,2).
function _matrix_handle_t(a, plan)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:pa_matrix_create_transposed, libpa), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Ptr{Cvoid}}),
                context().handle, _transposed(a.blocks.own_own).handle, _transposed(a.blocks.own_ghost).handle, plan, h))
    h[]
end
function mul_fused!(c::PVector, At::Transpose{Float64,<:PSparseMatrix}, b::PVector, α::Real=1.0, β::Real=0.0)
    A = parent(At)
    @assert A.assembled
    plans = c.cache.plans
    ms = map(_matrix_handle_t, partition(A), plans)
    _mul_fused_t!(ms, partition(c), partition(b), Float64(α), Float64(β))
    foreach(m -> ccall((:pa_matrix_destroy, libpa), Cint, (Ptr{Cvoid},), m), ms)
    c
end
_mul_fused_t!(ms::DebugArray, cs, bs, α, β) =
    check(ccall((:pa_mul5_transpose_all, libpa), Cint, (Ptr{Ptr{Cvoid}}, Int32, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Float64, Float64),
                ms.items, length(ms.items), [v.handle for v in cs.items], [v.handle for v in bs.items], α, β))
_mul_fused_t!(ms::MPIArray, cs, bs, α, β) =
    check(ccall((:pa_mul5_transpose, libpa), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Float64),
                ms.item, init_comm!(ms.comm), cs.item.handle, bs.item.handle, α, β))

# ---------------------------------------------------------------- conversions
"Device twin of a host PVector{Vector{Float64}} on any kind of index partition (block, permuted, hand-made LocalIndices)."
function to_hip(v::PVector)
    vals = map(partition(v), partition(axes(v, 1))) do x, ids
        HIPVector(collect(Float64, x), own_length(ids), local_to_device(ids))
    end
    PVector(vals, partition(axes(v, 1)))
end
"""
    to_hip(v::PVector{<:PartitionedArrays.SplitVector})

Device twin of a host PVector in SPLIT format (`pzeros(...;split_format=true)`, `pvector(...;split_format=true)`:
src/p_vector.jl:132-187, the `SplitVector` method of assemble_impl! :620-656).  A SplitVector already keeps its own and ghost values
in two contiguous blocks -- the device layout -- so they are uploaded as they are, block by block, without the pass through the local
order that `to_hip(::PVector)` makes; its `permutation` (local id -> position in [own | ghost]) IS the `l2d` of the HIPVector.
"""
function to_hip(v::PVector{<:PartitionedArrays.SplitVector})
    vals = map(partition(v), partition(axes(v, 1))) do x, ids
        own, ghost = x.blocks.own, x.blocks.ghost
        no, ng = length(own), length(ghost)
        perm = convert(Vector{Int32}, x.permutation)
        d = HIPVector(no, ng, perm == 1:(no + ng) ? nothing : perm)
        no > 0 && check(ccall((:pa_vec_upload, libpa), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64), d.handle, convert(Vector{Float64}, own), 0, no))
        ng > 0 && check(ccall((:pa_vec_upload, libpa), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64), d.handle, convert(Vector{Float64}, ghost), no, ng))
        d
    end
    PVector(vals, partition(axes(v, 1)))
end
"Host SplitVector twin of a device PVector: the two blocks downloaded as they lie in HBM."
function to_split(v::PVector{HIPVector})
    vals = map(partition(v)) do d
        dev = Vector{Float64}(undef, length(d))
        check(ccall((:pa_vec_download, libpa), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64), d.handle, dev, 0, length(dev)))
        perm = d.l2d === nothing ? collect(Int32, 1:length(d)) : d.l2d
        PartitionedArrays.split_vector(dev[1:d.n_own], dev[d.n_own+1:end], perm)
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

# ---------------------------------------------------------------- HPCG set-up route (optional)
# HPCG/src/sparse_matrix.jl:105-122 builds every part's COO triplets on the host and psparse assembles them; for the 27-point
# operator the own_own block -- 26/27 of the entries -- is a pure function of the part's box, so it is GENERATED in HBM
# (pa_hpcg_own_block_create: the arrays the reference's chain ends with, byte for byte; tests/test_gpu_setup.py) and only the
# part's surface (own_ghost, the ghost ids in first-seen order) comes from the host matrix the reference built.  `A`: the host
# PSparseMatrix of HPCG.build_p_matrix (assembled, split format, Int32 CSR); nx,ny,nz: the part's box; gn: the global grid;
# g0[part]: the 1-based global coordinates of the part's first node.  Returns the device twin of A and of the right-hand side.
function hpcg_blocks_hip(A::PSparseMatrix, nx::Integer, ny::Integer, nz::Integer, gn::NTuple{3,<:Integer}, g0)
    @assert A.assembled
    bs = map(partition(axes(A, 1))) do rows
        HIPVector(zeros(Float64, local_length(rows)), own_length(rows), local_to_device(rows))
    end
    mats = map(partition(A), g0, bs) do a, first_node, b
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:pa_hpcg_own_block_create, libpa), Cint,
                    (Ptr{Cvoid}, Int64, Int64, Int64, Int64, Int64, Int64, Int64, Int64, Int64, Ref{Ptr{Cvoid}}, Ptr{Cvoid}),
                    context().handle, nx, ny, nz, gn[1], gn[2], gn[3], first_node[1], first_node[2], first_node[3], h, C_NULL))
        check(ccall((:pa_hpcg_rhs, libpa), Cint,
                    (Ptr{Cvoid}, Int64, Int64, Int64, Int64, Int64, Int64, Int64, Int64, Int64, Ptr{Cvoid}),
                    context().handle, nx, ny, nz, gn[1], gn[2], gn[3], first_node[1], first_node[2], first_node[3], b.handle))
        n = nx * ny * nz
        blocks = PartitionedArrays.split_matrix_blocks(_adopt(h[], n, n), HIPCSR(a.blocks.own_ghost),
                                                       HIPCSR(a.blocks.ghost_own), HIPCSR(a.blocks.ghost_ghost))
        PartitionedArrays.split_matrix(blocks, a.row_permutation, a.col_permutation)
    end
    PSparseMatrix(mats, partition(axes(A, 1)), partition(axes(A, 2)), true), PVector(bs, partition(axes(A, 1)))
end

# ---------------------------------------------------------------- psparse on the device (optional route)
const BlockIndices = Union{PartitionedArrays.LocalIndicesWithConstantBlockSize,PartitionedArrays.LocalIndicesWithVariableBlockSize}
_box(r::BlockIndices) = (length(r.n), collect(Int64, r.n), Int64[first(x) for x in r.ranges], Int64[last(x) for x in r.ranges])
_empty_csr(n) = HIPCSR(SparseMatrixCSR{1}(0, n, Int32[1], Int32[], Float64[]))
function _assembly_blocks(h::Ptr{Cvoid}, rows, cols)
    a, b = Ref{Ptr{Cvoid}}(C_NULL), Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:pa_coo_assembly_blocks, libpa), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Ref{Ptr{Cvoid}}), h, a, b))
    check(ccall((:pa_coo_assembly_destroy, libpa), Cint, (Ptr{Cvoid},), h))
    no, nc, ng = own_length(rows), own_length(cols), ghost_length(cols)
    blocks = PartitionedArrays.split_matrix_blocks(_adopt(a[], no, nc), _adopt(b[], no, ng), _empty_csr(nc), _empty_csr(ng))
    PartitionedArrays.split_matrix(blocks, PartitionedArrays.local_permutation(rows), PartitionedArrays.local_permutation(cols))
end
function _assembly_ghosts(h::Ptr{Cvoid})
    v = [Ref{Int64}(0) for _ in 1:5]
    check(ccall((:pa_coo_assembly_info, libpa), Cint,
                (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}, Ref{Int64}, Ref{Int64}, Ref{Int64}, Ptr{Float64}), h, v[1], v[2], v[3], v[4], v[5], C_NULL))
    g = zeros(Int64, v[3][])
    check(ccall((:pa_coo_assembly_ghosts, libpa), Cint, (Ptr{Cvoid}, Ptr{Int64}), h, g))
    g
end
"""
    psparse_hip(I,J,V,rows) -> PSparseMatrix whose blocks are HIPCSR

`psparse(I,J,V,rows,cols;assembled=true)` with `cols = union_ghost(rows,J,find_owner(rows,J))` (src/p_sparse_matrix.jl:1249-1270;
the route of HPCG.build_p_matrix, HPCG/src/sparse_matrix.jl:105-122, and test/gallery_tests.jl:33) with everything per triplet
on the device (csrc/pa_assemble.hip): one upload of (I,J,V) per part, ghost columns in first-seen order, a stable (row, column)
sort with duplicates added in input order, the own | ghost split.  Block row partitions without ghosts.
"""
function psparse_hip(I, J, V, rows)
    asm = map(I, J, V, rows) do i, j, v, r
        r isa BlockIndices && ghost_length(r) == 0 || error("psparse_hip: block row partition without ghosts expected")
        D, n, lo, hi = _box(r)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:pa_coo_assemble, libpa), Cint,
                    (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Int32, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64},
                     Ptr{Int64}, Ptr{Int64}, Int64, Ptr{Int64}, Cint, Ref{Ptr{Cvoid}}),
                    context().handle, length(i), convert(Vector{Int64}, i), convert(Vector{Int64}, j), convert(Vector{Float64}, v),
                    D, n, lo, hi, n, lo, hi, 0, C_NULL, 1, h))
        h[]
    end
    ghosts = map(_assembly_ghosts, asm)
    cols = map(union_ghost, rows, ghosts, find_owner(rows, ghosts))     # (already distinct and in first-seen order)
    mats = map(_assembly_blocks, asm, rows, cols)
    PSparseMatrix(mats, rows, cols, true)
end
"""
    psparse_disassembled_hip(I,J,V,rows,cols) -> assembled PSparseMatrix whose blocks are HIPCSR

`psparse(I,J,V,rows,cols) |> fetch` with the default flags, then `assemble` (src/p_sparse_matrix.jl:1150-1219,1590-1756; the
route of test/fem_example.jl): per part the sub-assembled matrix on the device (ghost rows and columns in first-seen order), its
ghost rows sent to their owners with the reference's own `exchange`, the own rows and what arrived through the assembled route.
"""
function psparse_disassembled_hip(I, J, V, rows, cols; reuse::Bool=false)
    subs = map(I, J, V, rows, cols) do i, j, v, r, c
        D, nr, lor, hir = _box(r)
        _, nc, loc, hic = _box(c)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        reuse && check(ccall((:pa_coo_keep_input_slots, libpa), Cint, (Ptr{Cvoid}, Cint), context().handle, 1))
        check(ccall((:pa_coo_subassemble, libpa), Cint,
                    (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Int32, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64},
                     Ptr{Int64}, Ptr{Int64}, Ref{Ptr{Cvoid}}),
                    context().handle, length(i), convert(Vector{Int64}, i), convert(Vector{Int64}, j), convert(Vector{Float64}, v),
                    D, nr, lor, hir, nc, loc, hic, h))
        reuse && check(ccall((:pa_coo_keep_input_slots, libpa), Cint, (Ptr{Cvoid}, Cint), context().handle, 0))
        h[]
    end
    surf = map(subs) do h                       # ghost rows: gids (first-seen order) and entries sorted by (row, column)
        v = [Ref{Int64}(0) for _ in 1:4]
        check(ccall((:pa_coo_subassembly_info, libpa), Cint, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}, Ref{Int64}, Ref{Int64}), h, v[1], v[2], v[3], v[4]))
        gids, gr, gc, gv = zeros(Int64, v[1][]), zeros(Int32, v[4][]), zeros(Int32, v[4][]), zeros(Float64, v[4][])
        check(ccall((:pa_coo_subassembly_ghost_rows, libpa), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int32}, Ptr{Int32}, Ptr{Float64}), h, gids, gr, gc, gv))
        (gids, gr, gc, gv, _assembly_ghosts(h))
    end
    row_ghosts = map(s -> s[1], surf)
    rows_sa = map(union_ghost, rows, row_ghosts, find_owner(rows, row_ghosts))
    parts_snd, parts_rcv = assembly_neighbors(rows_sa)
    # setup_cache_snd (:1598-1650): the ghost rows' entries as global triplets, grouped by the owner of their row
    snd = map(surf, rows_sa, cols, parts_snd) do s, r, c, ps
        gids, gr, gc, gv, cg = s
        gI = gids[gr .+ 1]
        gJ = [k < own_length(c) ? own_to_global(c)[k+1] : cg[k-own_length(c)+1] for k in gc]
        owner = ghost_to_owner(r)[gr .+ 1]
        halves = vcat(findall(k -> k < own_length(c), gc), findall(k -> k >= own_length(c), gc))    # ghost_own's entries, then ghost_ghost's
        sel = [[e for e in halves if owner[e] == p] for p in ps]
        (JaggedArray([gI[x] for x in sel]), JaggedArray([gJ[x] for x in sel]), JaggedArray([gv[x] for x in sel]), sel, halves)
    end
    graph = ExchangeGraph(parts_snd, parts_rcv)
    Ircv = exchange(map(x -> x[1], snd), graph) |> fetch
    Jrcv = exchange(map(x -> x[2], snd), graph) |> fetch
    Vrcv = exchange(map(x -> x[3], snd), graph) |> fetch
    fin = map(subs, Ircv, Jrcv, Vrcv) do h, i, j, v
        f = Ref{Ptr{Cvoid}}(C_NULL)
        reuse && check(ccall((:pa_coo_keep_input_slots, libpa), Cint, (Ptr{Cvoid}, Cint), context().handle, 1))
        check(ccall((:pa_coo_assemble_finish, libpa), Cint, (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ref{Ptr{Cvoid}}),
                    h, length(i.data), convert(Vector{Int64}, i.data), convert(Vector{Int64}, j.data), convert(Vector{Float64}, v.data), f))
        reuse && check(ccall((:pa_coo_keep_input_slots, libpa), Cint, (Ptr{Cvoid}, Cint), context().handle, 0))
        f[]
    end
    ghosts = map(_assembly_ghosts, fin)
    cols_fa = map(union_ghost, cols, ghosts, find_owner(cols, ghosts))
    cache = nothing
    if reuse
        # The cache of psparse(...;reuse=true) (src/p_sparse_matrix.jl:1183-1219,1598-1689; psparse! :1291-1305), built ON THE DEVICE.
        # One part's stored values are the vector W = [nonzeros(own_own) | nonzeros(own_ghost) || the ghost rows' entries]; then
        #   sparse_matrix!(A,V,K) + split_format_locally! + setup_snd  = ONE deterministic scatter-add W[dest[p]] += V[p]  (pa_scatter),
        #   psparse_assemble_impl! (:1762-1816)                        = assemble! of W over a plan: idx_snd = the ghost-row slots in the
        #                                                                order they are sent (k_snd), idx_rcv = where the received triplets
        #                                                                landed (k_rcv, returned by pa_coo_reuse_scatter),
        #   nonzeros(blocks) .= W[own part]                            = pa_csr_update_values_from.
        parts = map(subs, fin, snd, Ircv, I, rows, parts_snd, parts_rcv) do h, f, sn, ir, i, r, ps, pr
            info = [Ref{Int64}(0) for _ in 1:5]
            check(ccall((:pa_coo_assembly_info, libpa), Cint,
                        (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}, Ref{Int64}, Ref{Int64}, Ref{Int64}, Ptr{Float64}), f, info[1], info[2], info[3], info[4], info[5], C_NULL))
            nnz_oo, nnz_oh = info[4][], info[5][]
            sel, halves = sn[4], sn[5]
            gslot = Vector{Int32}(undef, length(halves))              # 0-based position of ghost-row entry e in [ghost_own | ghost_ghost]
            for (k, e) in enumerate(halves); gslot[e] = k - 1; end
            sc = Ref{Ptr{Cvoid}}(C_NULL)
            k_rcv = zeros(Int32, max(length(ir.data), 1))
            check(ccall((:pa_coo_reuse_scatter, libpa), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Int32}, Ref{Ptr{Cvoid}}, Int64, Ptr{Int32}),
                        h, f, isempty(gslot) ? C_NULL : gslot, sc, length(ir.data), k_rcv))
            n_own_vals, n_ghost_vals = nnz_oo + nnz_oh, length(gslot)
            idx_snd = Int32[n_own_vals + gslot[e] + 1 for x in sel for e in x]
            ptrs_snd = Int32[1; 1 .+ cumsum(Int32[length(x) for x in sel])]
            plan = Ref{Ptr{Cvoid}}(C_NULL)
            check(ccall((:pa_plan_create, libpa), Cint,
                        (Ptr{Cvoid}, Int32, Int64, Int32, Ptr{Int32}, Ptr{Int32}, Ptr{Int32},
                         Int32, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Cint, Ref{Ptr{Cvoid}}),
                        context().handle, part_id(r), n_own_vals + n_ghost_vals,
                        length(ps), convert(Vector{Int32}, ps), ptrs_snd, idx_snd,
                        length(pr), convert(Vector{Int32}, pr), convert(Vector{Int32}, ir.ptrs), k_rcv[1:length(ir.data)], 1, plan))
            (plan[], sc[], HIPVector(n_own_vals, n_ghost_vals), HIPVector(length(i), 0), nnz_oo)
        end
        plans = map(x -> x[1], parts)
        plans isa MPIArray && PA_TRANSPORT == "ipc" && connect_ipc!(plans)
        cache = HIPReassemblyCache(plans, map(x -> x[2], parts), map(x -> x[3], parts), map(x -> x[4], parts), map(x -> x[5], parts))
    end
    foreach(h -> check(ccall((:pa_coo_assembly_destroy, libpa), Cint, (Ptr{Cvoid},), h)), subs)
    C = PSparseMatrix(map(_assembly_blocks, fin, rows, cols_fa), rows, cols_fa, true)
    reuse ? (C, cache) : C
end

"cache of `psparse_disassembled_hip(...; reuse=true)`: per part the plan that assembles W, the scatter V -> W, W and V in HBM, nnz(own_own)"
struct HIPReassemblyCache{A,B,C,D,E}
    plans::A
    scatters::B
    W::C
    Vdev::D
    nnz_oo::E
end
"""
    psparse_hip!(C, V, cache) -> C

`psparse!(C,V,cache) |> wait` (src/p_sparse_matrix.jl:1291-1305): the same sparsity pattern, new COO values.  After the upload of `V`
nothing leaves HBM: the scatter-add into W (sparse_matrix! + split_format_locally!, src/sparse_utils.jl:454-466,
src/p_sparse_matrix.jl:901-935), assemble! of W over the cache's plan (psparse_assemble_impl!, :1762-1816: ghost-row values travel to
their owners and are added in ascending position), and the blocks' values taken from W in place (pa_csr_update_values_from -- what
derived storage a block keeps, value dictionary, pattern-ELL stream, the fused product's boundary block, follows at the update).
"""
function psparse_hip!(C::PSparseMatrix, V, cache::HIPReassemblyCache)
    foreach((vd, v) -> upload!(vd, v), cache.Vdev, V)
    foreach(cache.scatters, cache.W, cache.Vdev) do sc, w, vd
        check(ccall((:pa_scatter_add, libpa), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint), sc, w.handle, vd.handle, 1))
    end
    PartitionedArrays.assemble_impl!(+, cache.W, HIPAssemblyCache(cache.plans, false)) |> wait
    foreach(partition(C), cache.W, cache.nnz_oo) do a, w, k
        check(ccall((:pa_csr_update_values_from, libpa), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64), a.blocks.own_own.handle, w.handle, 0))
        check(ccall((:pa_csr_update_values_from, libpa), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64), a.blocks.own_ghost.handle, w.handle, k))
    end
    C
end

end # module
