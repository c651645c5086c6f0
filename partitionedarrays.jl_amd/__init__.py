"""partitionedarrays.jl_amd -- MI355X-native device path for PartitionedArrays.jl's distributed SpMV.

Host-side mirror (Python over the C ABI of libpa_hip.so) of the reference's interface for ONE path:
`mul!(c::PVector, A::PSparseMatrix, b::PVector)` and the ghost exchange (`consistent!`, `assemble!`,
`exchange!`) it depends on.  Julia's `f!` is spelled `f_` here.  See DESIGN.md.
"""
from ._lib import PAError, LIB_PATH, SEG_OWN, SEG_GHOST, SEG_LOCAL, CONSISTENT, ASSEMBLE  # noqa: F401
from .primitives import (MAIN, DebugArray, TorchDistArray, ExchangeGraph, with_debug, with_torchdist,  # noqa: F401
                         linear_indices, pmap, pforeach, tuple_of_arrays, getany, local_items, map_main,
                         gather, scatter, multicast, reduction, preduce, scan, exchange, exchange_graph,
                         find_rcv_ids_gather_scatter, is_consistent)
from .p_range import (JaggedArray, LocalIndices, PRange, local_range, uniform_partition, variable_partition,  # noqa: F401
                      find_owner, filter_ghost, union_ghost, assembly_neighbors, assembly_local_indices)
from .p_vector import (Context, Event, Graph, context, init_comm, DeviceVector, DeviceAssemblyCache, Task, PVector,  # noqa: F401
                       pvector_from_function, pfill, pzeros, pones, similar, pvector, consistent_, assemble_, consistent32_, assemble32_, exchange32_, exchange_raw_,
                       dot, norm, axpby_, copy_, slots_supported, dot_slot, axpby_slot_, cg_update_, write_slot,
                       read_slots, on_partition, pvector_disassembled, pvector_, VectorReassemblyCache,
                       pvector_from_function_values)
from .p_sparse_matrix import (HostCSR, DeviceCSR, DeviceSELL, DeviceVector32, DeviceCSR32, spmv32_, SplitMatrixBlocks, PSparseMatrix, compresscoo, sparse_matrix,  # noqa: F401
                              split_format_locally, spmv_, psparse, psparse_from_coo, mul_, mul_c_, mul_no_lat_c_, mul5_, mul_no_overlap_,
                              psparse_disassembled, psparse_assemble_host, psparse_, MatrixReassemblyCache,
                              mul5_transpose_, transposed_blocks, renumber_for_locality, psystem, psystem_, tune_output_placement)
from .gallery import laplacian_fem, laplacian_fdm, build_matrix, build_p_matrix, build_split_blocks_fused, compute_optimal_shape_XYZ  # noqa: F401
from .hpcg import (CgTimer, ref_cg_, opt_cg_, hpcg_benchmark, cg_work, mul_no_lat_, mul_no_lat_unsplit_, restrict_operator, GaussSeidel, ColoredGaussSeidelSpMV, MgPreconditioner, pc_setup, pc_solve_,  # noqa: F401
                   ldiv_)
from . import fem_example  # noqa: F401,E402
