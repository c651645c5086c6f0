"""
Writes tests/golden/reference_literals.json.

There is no Julia in the build image, so the reference cannot be executed to generate vectors.
Every value below is a LITERAL taken from the reference's own tests / doctests (data, not code);
the `src` field cites where (/root/reference-relative file:line).  Run: python tests/golden/make_golden.py
"""
import json, os

G = {}

G["local_range"] = {"src": "test/p_range_tests.jl:7-15", "cases": [
    # [p, np, n, ghost, periodic, start, stop]
    [1, 3, 10, False, False, 1, 3], [2, 3, 10, False, False, 4, 6], [3, 3, 10, False, False, 7, 10],
    [1, 3, 10, True, False, 1, 4], [2, 3, 10, True, False, 3, 7], [3, 3, 10, True, False, 6, 10],
    [1, 3, 10, True, True, 0, 4], [2, 3, 10, True, True, 3, 7], [3, 3, 10, True, True, 6, 11]]}

G["uniform_partition"] = [
    {"src": "src/p_range.jl:569-575", "np": [4], "n": [10], "ghost": None, "periodic": None,
     "local_to_global": [[1, 2], [3, 4], [5, 6, 7], [8, 9, 10]]},
    {"src": "src/p_range.jl:577-582", "np": [2, 2], "n": [4, 4], "ghost": None, "periodic": None,
     "local_to_global": [[1, 2, 5, 6], [3, 4, 7, 8], [9, 10, 13, 14], [11, 12, 15, 16]]},
    {"src": "test/p_range_tests.jl:210-223", "np": [2, 2], "n": [5, 4], "ghost": None, "periodic": None,
     "local_to_global": [[1, 2, 6, 7], [3, 4, 5, 8, 9, 10], [11, 12, 16, 17], [13, 14, 15, 18, 19, 20]]},
    {"src": "test/p_range_tests.jl:226-237", "np": [2, 2], "n": [5, 4], "ghost": [True, True], "periodic": None,
     "local_to_global": [[1, 2, 3, 6, 7, 8, 11, 12, 13], [2, 3, 4, 5, 7, 8, 9, 10, 12, 13, 14, 15],
                         [6, 7, 8, 11, 12, 13, 16, 17, 18], [7, 8, 9, 10, 12, 13, 14, 15, 17, 18, 19, 20]]},
    {"src": "test/p_range_tests.jl:239-250", "np": [2, 2], "n": [4, 4], "ghost": [True, True], "periodic": [True, True],
     "local_to_global": [[16, 13, 14, 15, 4, 1, 2, 3, 8, 5, 6, 7, 12, 9, 10, 11],
                         [14, 15, 16, 13, 2, 3, 4, 1, 6, 7, 8, 5, 10, 11, 12, 9],
                         [8, 5, 6, 7, 12, 9, 10, 11, 16, 13, 14, 15, 4, 1, 2, 3],
                         [6, 7, 8, 5, 10, 11, 12, 9, 14, 15, 16, 13, 2, 3, 4, 1]]},
    {"src": "test/p_range_tests.jl:252-263", "np": [2, 2], "n": [4, 4], "ghost": [True, True], "periodic": [False, True],
     "local_to_global": [[13, 14, 15, 1, 2, 3, 5, 6, 7, 9, 10, 11], [14, 15, 16, 2, 3, 4, 6, 7, 8, 10, 11, 12],
                         [5, 6, 7, 9, 10, 11, 13, 14, 15, 1, 2, 3], [6, 7, 8, 10, 11, 12, 14, 15, 16, 2, 3, 4]]},
    {"src": "src/p_vector.jl:727-731 (uniform_partition(rank,6,true))", "np": [2], "n": [6], "ghost": [True], "periodic": None,
     "local_to_global": [[1, 2, 3, 4], [3, 4, 5, 6]]},
]

G["variable_partition"] = [
    {"src": "src/p_range.jl:694-703", "n_own": [3, 2, 2, 3], "local_to_global": [[1, 2, 3], [4, 5], [6, 7], [8, 9, 10]]},
    {"src": "test/p_range_tests.jl:185-208", "n_own": [4, 2, 6, 3],
     "local_to_global": [[1, 2, 3, 4], [5, 6], [7, 8, 9, 10, 11, 12], [13, 14, 15]]},
]

G["find_owner"] = {"src": "src/p_range.jl:322-344", "np": [4], "n": [10],
                   "gids": [[3], [4, 5], [7, 2], [9, 10, 1]], "owners": [[2], [2, 3], [3, 1], [4, 4, 1]]}

G["exchange"] = [
    {"src": "test/primitives_tests.jl:164-204", "snd_ids": [[3, 4], [1, 3], [1, 4], [2]],
     "rcv_ids": [[2, 3], [4], [1, 2], [1, 3]],
     "note": "the test sends snd = map(i->10*i,snd_ids), i.e. 10*(destination id), written out in snd_literal",
     "snd_literal": [[30, 40], [10, 30], [10, 40], [20]],
     "rcv": [[10, 10], [20], [30, 30], [40, 40]]},
    {"src": "src/primitives.jl:893-919 (doctest)", "snd_ids": [[3, 4], [1, 3], [1, 4], [2]], "rcv_ids": None,
     "snd_literal": [[10, 10], [20, 20], [30, 30], [40]], "rcv": [[20, 30], [40], [10, 20], [10, 30]]},
]
G["exchange_in_place"] = {"src": "test/primitives_tests.jl:245-288 (exchange! into map(similar,parts_rcv))",
                          "snd_ids": [[3, 4], [1, 3], [1, 4], [2]], "rcv_ids": [[2, 3], [4], [1, 2], [1, 3]],
                          "note": "data_snd = map(i->10*i,parts_snd)", "snd_literal": [[30, 40], [10, 30], [10, 40], [20]],
                          "rcv": [[10, 10], [20], [30, 30], [40, 40]]}
G["exchange_jagged"] = {"src": "test/primitives_tests.jl:220-234", "snd_ids": [[3, 4], [1, 3], [1, 4], [2]],
                        "rcv_ids": [[2, 3], [4], [1, 2], [1, 3]],
                        "snd": [[[1, 2, 3], [1, 2, 3, 4]], [[1], [1, 2, 3]], [[1], [1, 2, 3, 4]], [[1, 2]]],
                        "rcv": [[[1], [1]], [[1, 2]], [[1, 2, 3], [1, 2, 3]], [[1, 2, 3, 4], [1, 2, 3, 4]]]}
G["exchange_ring"] = {"src": "docs/examples.jl:58-91", "np": 3, "snd_ids": [[2], [3], [1]],
                      "data": [[10], [20], [30]], "after_3_exchanges": [[10], [20], [30]]}

G["p_vector_local_indices"] = {
    "src": "test/p_vector_tests.jl:93-142", "n": 10,
    "local_to_global": [[1, 2, 3, 5, 7, 8], [2, 4, 5, 10], [6, 7, 8, 5, 4, 10], [1, 3, 7, 9, 10]],
    "local_to_owner": [[1, 1, 1, 2, 3, 3], [1, 2, 2, 4], [3, 3, 3, 2, 2, 4], [1, 1, 3, 4, 4]],
    "consistent_rule": "own values = 10*part before; after consistent! every local value == 10*owner",
    "assemble_input": 10.0,
    "assemble_local_values": [[20.0, 20.0, 20.0, 0.0, 0.0, 0.0], [0.0, 20.0, 30.0, 0.0],
                              [10.0, 30.0, 20.0, 0.0, 0.0, 0.0], [0.0, 0.0, 0.0, 10.0, 30.0]],
    "assemble_collect": [20.0, 20.0, 20.0, 20.0, 30.0, 10.0, 30.0, 20.0, 10.0, 30.0]}

G["doc_consistent"] = {"src": "src/p_vector.jl:722-745", "np": [2], "n": [6], "ghost": [True],
                       "before": [[1, 1, 1, 1], [2, 2, 2, 2]], "after": [[1, 1, 1, 2], [1, 2, 2, 2]]}
G["doc_assemble"] = {"src": "src/p_vector.jl:670-693", "np": [2], "n": [6], "ghost": [True],
                     "before": [[1.0, 1.0, 1.0, 1.0], [1.0, 1.0, 1.0, 1.0]],
                     "after": [[1.0, 1.0, 2.0, 0.0], [0.0, 2.0, 1.0, 1.0]]}

G["mul_diag"] = {"src": "test/p_sparse_matrix_tests.jl:207-248,285-291", "np": [4], "n": [10],
                 "diag": 2.0, "x": 3.0, "y": 6.0, "fillstored": 1.0, "y_fillstored": 3.0}

G["sparse_utils_mat"] = {"src": "test/sparse_utils_tests.jl:14-45", "I": [1, 2, 5, 4, 1], "J": [3, 6, 1, 1, 3],
                         "V": [4, 5, 3, 2, 5], "m": 7, "n": 6,
                         "dense_note": "sparse(I,J,V,m,n) combines duplicates with +: A[1,3]=9",
                         "x": [1, 2, 3, 4, 5, 6], "Ax": [27.0, 30.0, 0.0, 2.0, 3.0, 0.0, 0.0],
                         "Ax_note": "NOT a literal of the reference (its test compares spmv! with the library mul!); "
                                    "worked by hand from the literal I,J,V,x above (all products are small integers)"}

G["hpcg"] = {"src": "HPCG/test/hpcg_benchmark_tests.jl:15-28", "seq_grid": [32, 32, 16],
             "parts": [2, 2, 1], "n_per_part": [16, 16, 16],
             "property": "b of the sequential build == collect(pb) of the 2x2x1 partitioned build"}

G["hpcg_known_answer"] = {"src": "HPCG/test/hpcg_benchmark_tests.jl:31-41", "np": 4, "parts": [2, 2, 1], "n": [32, 32, 32],
                          "levels": 4, "maxiter": 50, "expected_ref_tol": 2.877476184683206e-13, "assert_below": 1.0e-12}

G["ghost_first_seen"] = {"src": "SURVEY.md Appendix A (derived from HPCG/src/sparse_matrix.jl:41-57 + src/p_range.jl:226-239)",
                         "global": [8, 4, 4], "parts": [2, 1, 1], "part": 2, "ghost_gids_head": [4, 12, 36, 44, 20, 52]}

# G6 collectives on 4 parts (rank = 1..4): literal expectations of test/primitives_tests.jl
G["collectives"] = {"src": "test/primitives_tests.jl:40-150", "np": 4,
                    "gather_10rank": {"snd": [10, 20, 30, 40], "rcv": [10, 20, 30, 40], "destination": 2, "lines": "40-49,58-62"},
                    "gather_ragged": {"snd": [[1], [1, 2], [1, 2, 3], [1, 2, 3, 4]], "rcv_all": [[1], [1, 2], [1, 2, 3], [1, 2, 3, 4]], "lines": "64-78"},
                    "scatter_roundtrip": "scatter(gather(snd)) == snd (lines 51-55, 67-72)",
                    "multicast_rank_source2": 2, "multicast_ragged_source2": [1, 2], "multicast_lines": "103-112",
                    "scan": {"a": "3*mod(rank,3)", "a_values": [3, 6, 0, 3], "inclusive_init0": [3, 9, 9, 12], "exclusive_init1": [1, 4, 10, 10], "lines": "114-126"},
                    "reduction": {"sum_init0": 10, "sum_init10_all": 20, "reduce": 10, "reduce_init2": 12, "lines": "140-150"}}

# G10 JaggedArray construction / equality
G["jagged_array"] = {"src": "test/jagged_array_tests.jl:6-22", "a": [[1, 2], [3, 4, 5], [], [3, 4]],
                     "data": [1, 2, 3, 4, 5, 3, 4], "ptrs": [1, 3, 6, 6, 8],
                     "note": "data/ptrs follow from length_to_ptrs! (src/jagged_array.jl:11-18): ptrs[1] = 1, ptrs[i+1] = ptrs[i] + length(a[i])"}

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_literals.json")
with open(out, "w") as f:
    json.dump(G, f, indent=1)
print("wrote", out)
