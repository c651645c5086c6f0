"""Extracts, from /root/reference, the SIGNATURES (name, positional argument names, file:line) of the generic functions the
Julia glue overloads or calls -- interface facts, not source text -- into tests/golden/reference_signatures.json.
Run in the build container (the reference is not on the GPU box):  python tests/golden/make_signatures.py"""
import json
import os
import re

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_signatures.json")

# (function name, file, line of the method the glue's method mirrors)
WANT = [
    ("allocate_local_values", "src/p_vector.jl", 8), ("allocate_local_values", "src/p_vector.jl", 12),
    ("own_values", "src/p_vector.jl", 20), ("ghost_values", "src/p_vector.jl", 24),
    ("p_vector_cache_impl", "src/p_vector.jl", 451), ("assemble_impl!", "src/p_vector.jl", 587),
    ("spmv!", "src/sparse_utils.jl", 617), ("split_matrix_blocks", "src/p_sparse_matrix.jl", 594),
    ("split_matrix", "src/p_sparse_matrix.jl", 629), ("assembly_neighbors", "src/p_range.jl", 417),
    ("assembly_local_indices", "src/p_range.jl", 471), ("insert", "src/p_vector.jl", 755),
]


def split_args(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        depth += ch in "({["
        depth -= ch in ")}]"
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [a.strip() for a in out]


def main():
    sigs = []
    for name, path, line in WANT:
        txt = open(os.path.join(REF, path)).read().splitlines()[line - 1]
        m = re.search(r"(?:function\s+)?" + re.escape(name) + r"\((.*?)\)(?:\s*where.*)?(?:\s*=.*)?$", txt.strip())
        assert m, (name, path, line, txt)
        args = m.group(1).split(";")[0]
        pos = []
        for a in split_args(args):
            nm, _, ty = a.partition("::")
            pos.append({"name": nm.strip(), "type": ty.strip()})
        sigs.append({"function": name, "src": f"{path}:{line}", "positional": pos})
    json.dump({"reference": "fverdugo/PartitionedArrays.jl v0.5.7", "signatures": sigs}, open(OUT, "w"), indent=1)
    print(f"wrote {len(sigs)} signatures to {OUT}")


if __name__ == "__main__":
    main()
