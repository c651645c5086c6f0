# bench/julia_reference.jl -- the reference's own mul! / consistent! on the host cores of this box, for the line beside bench.py's
# (BASELINE.md section 4: optional; needs a Julia with PartitionedArrays 0.5.7 and the HPCG sub-package's build_p_matrix -- neither
# is in the build image, so this file has never been executed here).
#
#     julia --project=<env> -t <threads> bench/julia_reference.jl [n=64] [np=4] [steps=20]
#
# Prints one JSON line: GFLOP/s of mul!(c,A,b) on the HPCG 27-point matrix, n^3 rows per part, np parts as DebugArray parts run in
# sequence (BASELINE config 1: "DebugArray backend, sequential parts on CPU"), and the seconds of a consistent!.
using PartitionedArrays, LinearAlgebra, Printf
import HPCG                                            # /root/reference/HPCG: build_p_matrix, compute_optimal_shape_XYZ

n = length(ARGS) >= 1 ? parse(Int, ARGS[1]) : 64
np = length(ARGS) >= 2 ? parse(Int, ARGS[2]) : 4
steps = length(ARGS) >= 3 ? parse(Int, ARGS[3]) : 20
ranks = DebugArray(LinearIndices((np,)))
npx, npy, npz = HPCG.compute_optimal_shape_XYZ(np)
A, b = HPCG.build_p_matrix(ranks, n, n, n, npx * n, npy * n, npz * n, npx, npy, npz)
x = pones(partition(axes(A, 2)))
c = similar(x, axes(A, 1))
mul!(c, A, x)                                          # compile
nnz_total = sum(map(a -> length(a.blocks.own_own.nzval) + length(a.blocks.own_ghost.nzval), partition(A)))
t = @elapsed for _ in 1:steps
    mul!(c, A, x)
end
tc = @elapsed for _ in 1:steps
    consistent!(x) |> wait
end
@printf("{\"what\": \"PartitionedArrays.jl v0.5.7 mul! on the host, HPCG 27-pt %d^3 rows per part, %d DebugArray parts in sequence\", \"threads\": %d, \"gflops\": %.3f, \"ms_per_mul\": %.4f, \"ms_per_consistent\": %.4f}\n",
        n, np, Threads.nthreads(), 2.0 * nnz_total * steps / t / 1e9, 1e3 * t / steps, 1e3 * tc / steps)
