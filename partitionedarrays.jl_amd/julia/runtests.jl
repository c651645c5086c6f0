# runtests.jl -- one command for a maintainer with a Julia and an MI355X:
#
#     LIBPA_HIP=/path/to/libpa_hip.so julia --project=<env with PartitionedArrays 0.5.7, SparseMatricesCSR, MPI> \
#         partitionedarrays.jl_amd/julia/runtests.jl
#
# NOT executed in this repository (no Julia in the build image; DESIGN.md "Host language").  It replays, through the device types
# of PartitionedArraysHIP.jl, the reference's OWN literal tests of the hot path -- the values below are the ones its test files
# hold -- plus the device-vs-host comparisons every block of the glue needs:
#   test/p_vector_tests.jl:93-142          consistent! / assemble! on the hand-made 4-part partition
#   test/p_sparse_matrix_tests.jl:207-248  mul! of the diagonal matrix 2 I on a uniform partition (own values, then consistent!)
#   HPCG/test/hpcg_benchmark_tests.jl      A * 1 == b for the 27-point operator (per part, exactly)
using Test
using LinearAlgebra
using PartitionedArrays
using SparseMatricesCSR
include(joinpath(@__DIR__, "PartitionedArraysHIP.jl"))
using .PartitionedArraysHIP: to_hip, mul_fused!, HIPVector, HIPCSR, synchronize

host(v::PVector) = map(x -> x isa HIPVector ? Array(x) : collect(x), partition(v))

@testset "PartitionedArraysHIP" begin
    rank = DebugArray(LinearIndices((4,)))

    @testset "consistent! / assemble!: test/p_vector_tests.jl:93-142" begin
        n = 10
        row_partition = map(rank) do part
            if part == 1
                LocalIndices(n, part, [1, 2, 3, 5, 7, 8], Int32[1, 1, 1, 2, 3, 3])
            elseif part == 2
                LocalIndices(n, part, [2, 4, 5, 10], Int32[1, 2, 2, 4])
            elseif part == 3
                LocalIndices(n, part, [6, 7, 8, 5, 4, 10], Int32[3, 3, 3, 2, 2, 4])
            else
                LocalIndices(n, part, [1, 3, 7, 9, 10], Int32[1, 1, 3, 4, 4])
            end
        end
        v = pzeros(row_partition)
        map(rank, partition(v), row_partition) do part, values, indices
            o = local_to_owner(indices)
            for lid in 1:length(o)
                o[lid] == part && (values[lid] = 10 * part)
            end
        end
        d = to_hip(v)
        consistent!(d) |> wait
        map(host(d), row_partition) do values, indices
            o = local_to_owner(indices)
            @test all(values[lid] == 10 * o[lid] for lid in 1:length(o))
        end
        d = to_hip(pfill(10.0, row_partition))
        assemble!(d) |> wait
        expected = ([20.0, 20.0, 20.0, 0.0, 0.0, 0.0], [0.0, 20.0, 30.0, 0.0], [10.0, 30.0, 20.0, 0.0, 0.0, 0.0], [0.0, 0.0, 0.0, 10.0, 30.0])
        map(rank, host(d)) do part, values
            @test values == expected[part]
        end
    end

    @testset "mul! of 2 I: test/p_sparse_matrix_tests.jl:207-248" begin
        n = 10
        row_partition = uniform_partition(rank, n)
        I, J, V = map(row_partition) do rows
            i = collect(own_to_global(rows))
            i, copy(i), fill(2.0, length(i))
        end |> tuple_of_arrays
        A = psparse(sparsecsr, I, J, V, row_partition, row_partition) |> fetch
        x = pfill(3.0, axes(A, 2); split_format=true)
        dA, dx = to_hip(A), to_hip(x)
        db = similar(dx, axes(dA, 1))
        mul!(db, dA, dx)                                  # the reference's own mul! body on the device types
        map(own_values(db)) do values
            @test all(Array(values.parent)[1:length(values)] .== 6)
        end
        consistent!(db) |> wait
        map(host(db)) do values
            @test all(values .== 6)
        end
        dc = similar(dx, axes(dA, 1))
        mul_fused!(dc, dA, dx)                            # one library call per process (pa_mul_all): the same bits
        @test host(dc) == host(db) || all(map((a, b, r) -> a[1:own_length(r)] == b[1:own_length(r)], host(dc), host(db), row_partition))
    end

    @testset "device product == host product, random x" begin
        nodes, parts = (8, 8, 8), (2, 2, 1)
        args = laplacian_fdm(nodes, parts, rank)
        A = psparse(sparsecsr, args...) |> fetch
        x = prand(partition(axes(A, 2)))
        consistent!(x) |> wait
        b = similar(x, axes(A, 1))
        mul!(b, A, x)
        dA, dx = to_hip(A), to_hip(x)
        db = similar(dx, axes(dA, 1))
        mul!(db, dA, dx)
        map(own_values(b), host(db), partition(axes(A, 1))) do want, got, rows
            @test got[1:own_length(rows)] == collect(want)       # bit for bit: same products, same order (csrc/pa_spmv_kernel.h)
        end
        c = copy(b)
        mul!(c, A, x, 0.3, -1.5)
        dc = to_hip(b)
        mul_fused!(dc, dA, dx, 0.3, -1.5)
        map(own_values(c), host(dc), partition(axes(A, 1))) do want, got, rows
            @test got[1:own_length(rows)] == collect(want)
        end
    end
    synchronize()
end
