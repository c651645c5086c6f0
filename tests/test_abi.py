"""The C-ABI library loads (no GPU needed) and exports every symbol include/pa_hip.h declares."""
import ctypes
import os
import re

from __graft_entry__ import ROOT, load_package


HEADERS = ("pa_hip.h", "pa_hip_experimental.h")       # the contract and the experimental tier (include/pa_hip.h, "TIERS")


def _header_text(names=HEADERS):
    return "\n".join(open(os.path.join(ROOT, "include", n)).read() for n in names)


def _declared(names=HEADERS):
    txt = re.sub(r"/\*.*?\*/", "", _header_text(names), flags=re.S)
    return sorted(set(re.findall(r"\b(pa_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    pa = load_package()
    lib = ctypes.CDLL(pa.LIB_PATH)
    names = _declared()
    assert len(names) >= 45
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/pa_hip.h but not exported"


def test_binding_covers_header():
    load_package()
    import pa_amd._lib as L
    assert sorted(L.EXPORTS) == _declared()


def test_errors_are_statuses_not_aborts():
    load_package()
    import pa_amd._lib as L
    assert L.lib.pa_version() >= 100
    st = L.lib.pa_ctx_sync(None)
    assert st == -2 and b"NULL" in L.lib.pa_last_error()
    n = ctypes.c_int(-1)
    assert L.lib.pa_device_count(ctypes.byref(n)) == 0 and n.value >= 0


# ---- the Julia glue (not runnable here: no julia in the image) is checked statically against the header ---------------
def _prototypes():
    txt = re.sub(r"/\*.*?\*/", "", _header_text(), flags=re.S)
    protos = {}
    for ret, name, args in re.findall(r"([A-Za-z_][\w \*]*?)\b(pa_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", txt):
        args = [a.strip() for a in args.split(",")] if args.strip() not in ("", "void") else []
        protos[name] = (ret.strip(), args)
    return protos


def _c_class(t):
    t = t.replace("const", " ")
    if "*" in t or "[" in t:
        return "ptr"
    t = t.split()
    t = t[0] if len(t) == 1 else " ".join(t[:-1])        # drop the parameter name
    return {"int": "i32", "int32_t": "i32", "int64_t": "i64", "double": "f64", "size_t": "u64"}[t]


def _jl_class(t):
    t = t.strip()
    if t.startswith(("Ptr{", "Ref{")) or t == "Cstring":
        return "ptr"
    return {"Cint": "i32", "Int32": "i32", "Int64": "i64", "Float64": "f64", "Csize_t": "u64"}[t]


def _split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        depth += ch in "({"
        depth -= ch in ")}"
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [x.strip() for x in out]


def test_julia_glue_ccalls_match_header():
    """Every `ccall((:pa_x, libpa), Ret, (Args...), ...)` of the glue names a declared entry point with the
    declared number and classes (pointer / 32-bit / 64-bit / double) of arguments."""
    protos = _prototypes()
    src = open(os.path.join(ROOT, "partitionedarrays.jl_amd", "julia", "PartitionedArraysHIP.jl")).read()
    src = "\n".join(l for l in src.splitlines() if not l.lstrip().startswith("#"))
    calls = re.findall(r"ccall\(\(:(pa_[a-z0-9_]+),\s*libpa\),\s*(\w+),\s*\(((?:[^()]|\([^()]*\))*?)\)\s*[,)]", src)
    assert len(calls) >= 20
    for name, ret, argt in calls:
        assert name in protos, f"glue calls {name}, which include/pa_hip.h does not declare"
        cret, cargs = protos[name]
        jl = [_jl_class(t) for t in _split_top(argt.rstrip(","))] if argt.strip() else []
        cc = [_c_class(a) for a in cargs]
        assert jl == cc, f"{name}: glue passes {jl}, header declares {cc}"
        assert _jl_class(ret) == _c_class(cret + " x"), f"{name}: return type"


def test_the_glue_and_the_c_example_need_only_the_contract_header():
    """a binding author must be able to tell contract from lab.  Every entry point the Julia glue ccalls and
    everything examples/c_abi_smoke.c uses is declared in include/pa_hip.h; pa_hip_experimental.h holds none of them, and the
    two headers declare disjoint sets that together are exactly what the library's binding table lists."""
    contract, lab = set(_declared(("pa_hip.h",))), set(_declared(("pa_hip_experimental.h",)))
    assert not (contract & lab)
    src = open(os.path.join(ROOT, "partitionedarrays.jl_amd", "julia", "PartitionedArraysHIP.jl")).read()
    used = set(re.findall(r"ccall\(\(:(pa_[a-z0-9_]+),", src))
    assert used and used <= contract, sorted(used - contract)
    csrc = open(os.path.join(ROOT, "examples", "c_abi_smoke.c")).read()
    assert '#include "pa_hip.h"' in csrc and "pa_hip_experimental.h" not in csrc
    assert set(re.findall(r"\b(pa_[a-z0-9_]+)\s*\(", csrc)) <= contract
    assert not [n for n in contract if n.startswith("pa_host_")] and len(contract) < 110


def test_committed_bench_line_follows_the_contract():
    """profiles/rNN_bench_n1.json is one line printed by bench.py on an MI355X: the keys the driver and the judge read."""
    import glob
    import json
    path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_n1.json")))[-1]
    d = json.load(open(path))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["unit"] == "GFLOP/s" and d["dtype"] == "f64" and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["traffic"] > 3e9
    if "frac_algorithmic_csr" in r:      # round 3 on: `frac` is the physical (moved-bytes) fraction, never above 1
        assert 0 < r["frac"] <= 1.0 and d["warmup_effective"] >= d["warmup"] and "cold" in d
        assert abs(r["achieved"] - r["moved_bytes_per_launch"] / r["avg_launch_ms"] / 1e6) < 1.0
        assert abs(r["achieved_algorithmic_csr"] - r["algorithmic_bytes_per_launch"] / r["avg_launch_ms"] / 1e6) < 1.0
        assert len(d["general_csr"]) == 2 and all(e["bit_identical_to_headline_product"] for e in d["general_csr"])
    else:                                 # rounds 1-2 quoted `frac` on the reference's CSR bytes
        assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / r["avg_launch_ms"] / 1e6) < 1.0
    assert abs(d["value"] - 2 * d["config"]["nnz_per_part"] * d["n_gpus"] / d["ms_per_step"] / 1e6) < 0.5
    if int(os.path.basename(path)[1:3]) >= 6:
        # the three accountings side by side in `roofline`, and no entry anywhere in the line may quote a GB/s figure above
        # the 8 TB/s peak (algorithmic bytes: columns regenerated from patterns, values from a dictionary) without the moved-bytes
        # fraction beside it
        assert r["frac_spec_8d"] == r["frac_algorithmic_csr"] and 0 < r["frac_counter"] <= 1.05 and abs(r["frac_counter"] - r["frac"]) < 0.08

        def walk(o, where):
            if isinstance(o, dict):
                over = [k for k, v in o.items() if k.endswith("gbps") and isinstance(v, (int, float)) and v > 8000.0]
                assert not over or ("frac_moved" in o and 0 < o["frac_moved"] <= 1.0), (where, over)
                for k, v in o.items():
                    walk(v, where + "/" + k)
            elif isinstance(o, list):
                for i, v in enumerate(o):
                    walk(v, f"{where}[{i}]")
        walk(d, "")


def test_header_is_valid_c99_and_the_c_example_links():
    """include/pa_hip.h is a C header (extern "C" guards, no C++ in the declarations): gcc -std=c99 -Werror accepts it,
    and examples/c_abi_smoke.c -- the ABI used from plain C -- compiles and links against libpa_hip.so."""
    import subprocess
    load_package()
    inc = os.path.join(ROOT, "include")
    for h in HEADERS:
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c", os.path.join(inc, h)])
    out = os.path.join(ROOT, "examples", "c_abi_smoke")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", inc,
                           os.path.join(ROOT, "examples", "c_abi_smoke.c"), "-L", os.path.join(ROOT, "partitionedarrays.jl_amd"),
                           "-lpa_hip", "-Wl,-rpath," + os.path.join(ROOT, "partitionedarrays.jl_amd"),
                           "-Wl,-rpath,/opt/rocm/lib", "-o", out])
    assert os.path.exists(out)


# ---- the glue's METHODS against the reference's generic functions -----------------------------------
def _glue_src():
    src = open(os.path.join(ROOT, "partitionedarrays.jl_amd", "julia", "PartitionedArraysHIP.jl")).read()
    return "\n".join(l for l in src.splitlines() if not l.lstrip().startswith("#"))


def _signatures():
    import json
    return json.load(open(os.path.join(ROOT, "tests", "golden", "reference_signatures.json")))["signatures"]


def _glue_methods(src):
    """(qualified name, [(arg name, arg type)]) of every method the glue adds to a function of another module."""
    out = []
    for m in re.finditer(r"(?:^|\n)\s*(?:function\s+)?((?:PartitionedArrays|LinearAlgebra|Base)\.[A-Za-z_!]+)\(((?:[^()]|\([^()]*\))*)\)\s*(?:=|\n)", src):
        args = []
        for a in _split_top(m.group(2).split(";")[0]):
            nm, _, ty = a.partition("::")
            args.append((nm.strip(), ty.strip()))
        out.append((m.group(1), args))
    return out


def test_julia_glue_methods_match_the_reference_generic_functions():
    """The glue overloads allocate_local_values / own_values / ghost_values (src/p_vector.jl:8-26), p_vector_cache_impl
    (:451), assemble_impl! (:587) and spmv! (src/sparse_utils.jl:617): each method must have the arity and the argument
    ORDER of the reference's generic function (tests/golden/reference_signatures.json, extracted by make_signatures.py) --
    a `::Type{...}` slot where the reference has one, and the reference's own names for the named slots."""
    sigs = _signatures()
    by_name = {}
    for s in sigs:
        by_name.setdefault(s["function"], []).append(s)
    methods = _glue_methods(_glue_src())
    seen = set()
    for qname, args in methods:
        mod, fn = qname.split(".")
        if mod != "PartitionedArrays" or fn not in by_name:
            continue
        cands = [s for s in by_name[fn] if len(s["positional"]) == len(args)]
        assert cands, f"{qname}: the glue defines {len(args)} positional arguments, the reference has {[len(s['positional']) for s in by_name[fn]]}"
        ok = False
        for s in cands:
            good = True
            for (gn, gt), ref in zip(args, s["positional"]):
                ref_is_type = ref["type"].startswith("Type")
                if ref_is_type != gt.startswith("Type"):
                    good = False
                if ref["name"] in ("indices", "f", "vector_partition", "index_partition", "cache") and gn and gn != ref["name"]:
                    good = False
            ok = ok or good
        assert ok, f"{qname}{args} does not line up with any reference method {[s['positional'] for s in cands]}"
        seen.add(fn)
    assert seen >= {"allocate_local_values", "own_values", "ghost_values", "p_vector_cache_impl", "assemble_impl!", "spmv!"}, seen
    # calls INTO the reference: arity of what the glue calls
    src = _glue_src()
    for fn, n in (("assembly_neighbors", 1), ("assembly_local_indices", 3), ("split_matrix_blocks", 4), ("split_matrix", 3)):
        m = re.search(r"(?<![\w.!])(?:PartitionedArrays\.)?" + re.escape(fn) + r"\(((?:[^()]|\([^()]*\))*)\)", src)
        assert m, fn
        assert len(_split_top(m.group(1))) == n, (fn, m.group(1))
        assert any(len(s["positional"]) == n for s in by_name[fn]), fn
    # insert is compared by identity in assemble_impl!: it must be the reference's two-argument insert(a,b) = b
    assert [len(s["positional"]) for s in by_name["insert"]] == [2]


def test_reference_signature_fixture_is_current():
    """Only where the reference tree exists (the build container): the committed fixture equals a fresh extraction."""
    import importlib.util
    import json
    import pytest
    if not os.path.isdir("/root/reference/src"):
        pytest.skip("no reference tree on this box")
    spec = importlib.util.spec_from_file_location("make_signatures", os.path.join(ROOT, "tests", "golden", "make_signatures.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want = json.load(open(mod.OUT))
    old = mod.OUT
    try:
        mod.OUT = os.path.join("/tmp", "reference_signatures_check.json")
        mod.main()
        assert json.load(open(mod.OUT)) == want
    finally:
        mod.OUT = old


# ---- the glue's ccall sequence of mul!, replayed through ctypes with the glue's own type tuples -------------------------
_JL2C = {"Cint": ctypes.c_int, "Int32": ctypes.c_int32, "Int64": ctypes.c_int64, "Float64": ctypes.c_double}


def _glue_ccalls(func_pattern):
    """[(entry point, [ctypes types])] of the ccalls inside the glue function whose header matches `func_pattern`, in
    source order; pointer-like Julia types (Ptr{...}, Ref{...}) become c_void_p."""
    src = _glue_src()
    m = re.search(func_pattern, src)
    assert m, func_pattern
    body = src[m.start():]
    end = re.search(r"\n(?:end|function |[A-Za-z_.!]+\([^\n]*\) =)", body[1:])
    body = body[:end.start() + 1 + (3 if body[end.start() + 1:].startswith("\nend") else 0)] if end else body
    out = []
    for name, ret, argt in re.findall(r"ccall\(\(:(pa_[a-z0-9_]+),\s*libpa\),\s*(\w+),\s*\(((?:[^()]|\([^()]*\))*?)\)\s*[,)]", body):
        types = [(_JL2C.get(t.strip(), ctypes.c_void_p)) for t in _split_top(argt.rstrip(","))]
        out.append((name, types))
    return out


import pytest  # noqa: E402


@pytest.mark.gpu
def test_glue_ccall_sequence_of_mul_replayed_through_ctypes(orc):
    """What the reference's mul!(c,a,b) (src/p_sparse_matrix.jl:2090-2103) executes once the glue's methods exist,
    replayed call by call with the argument type tuples WRITTEN IN THE GLUE (parsed from its ccalls, not from the header
    or from the Python binding): p_vector_cache_impl -> pa_plan_create; consistent!(b) -> assemble_impl!(insert, ...,
    reverse(cache)) -> pa_exchange_pack per part, pa_exchange_local, [own x own: spmv! -> pa_spmv], wait(t) ->
    pa_exchange_finish per part, [own x ghost: mul!(b,A,x,1,1) -> pa_spmv].  4 parts of an 8 x 8 x 4 HPCG grid, bit-exact
    against the oracle.  The glue itself cannot run here (no Julia); this pins its call sequence and its type tuples."""
    import numpy as np
    pa = load_package()
    lib = ctypes.CDLL(pa.LIB_PATH)
    seq = {
        "ctx": _glue_ccalls(r"function context\("),
        "vec": _glue_ccalls(r"function HIPVector\(n_own::Integer, n_ghost::Integer, l2d"),
        "upload": _glue_ccalls(r"function upload!\(v::HIPVector"),
        "download": _glue_ccalls(r"function Base\.Array\(v::HIPVector\)"),
        "csr": _glue_ccalls(r"function HIPCSR\(A::SparseMatrixCSR"),
        "cache": _glue_ccalls(r"function PartitionedArrays\.p_vector_cache_impl"),
        "local": _glue_ccalls(r"_transport!\(plans::DebugArray"),
        "impl": _glue_ccalls(r"function PartitionedArrays\.assemble_impl!"),
        "spmv": _glue_ccalls(r"function PartitionedArrays\.spmv!"),
        "mul5": _glue_ccalls(r"function LinearAlgebra\.mul!"),
    }
    assert [n for n, _ in seq["impl"]] == ["pa_exchange_pack", "pa_exchange_finish"]
    assert [n for n, _ in seq["cache"]] == ["pa_plan_create"] and [n for n, _ in seq["local"]] == ["pa_exchange_local"]

    def call(key, k, *vals):
        name, types = seq[key][k]
        assert len(types) == len(vals), (name, len(types), len(vals))
        f = getattr(lib, name)
        f.restype, f.argtypes = ctypes.c_int, types
        st = f(*vals)
        assert st == 0, (name, lib.pa_last_error())

    P = ctypes.c_void_p
    vp = lambda a: a.ctypes.data_as(P)
    # the host data the reference would hold (the oracle builds it the reference's way, 1-based)
    A, _b, _ = orc.hpcg_build_p_matrix(4, 4, 4, 2, 2, 1)
    cache = orc.p_vector_cache([np.zeros(c.n_local) for c in A.cols], A.cols)
    ctx = P()
    call("ctx", 0, 0, ctypes.byref(ctx))
    nparts = len(A.cols)
    xs, ys, oo, oh, plans, keep = [], [], [], [], [], []
    xh = [orc.hash_x(c.local_to_global) * (c.local_to_owner == c.part) for c in A.cols]
    for p in range(nparts):
        c, r = A.cols[p], A.rows[p]
        for lst, n_own, n_ghost, host in ((xs, c.n_own, c.n_ghost, xh[p]), (ys, r.n_own, 0, None)):
            h = P()
            call("vec", 0, ctx, n_own, n_ghost, ctypes.byref(h))
            if host is not None:
                hb = np.ascontiguousarray(host)
                call("upload", 0, h, vp(hb), 0, len(hb))
            lst.append(h)
        for lst, M in ((oo, A.blocks[p].own_own), (oh, A.blocks[p].own_ghost)):
            h = P()
            rp, cv, nz = (np.ascontiguousarray(M.rowptr, np.int32), np.ascontiguousarray(M.colval, np.int32), np.ascontiguousarray(M.nzval))
            keep += [rp, cv, nz]
            call("csr", 0, ctx, M.m, M.n, len(nz), vp(rp), vp(cv), 4, 1, vp(nz), ctypes.byref(h))
            lst.append(h)
        ns, nr = np.ascontiguousarray(cache.neighbors_snd[p], np.int32), np.ascontiguousarray(cache.neighbors_rcv[p], np.int32)
        ls, lr = cache.local_indices_snd[p], cache.local_indices_rcv[p]
        arrs = [ns, np.ascontiguousarray(ls.ptrs, np.int32), np.ascontiguousarray(ls.data, np.int32), nr,
                np.ascontiguousarray(lr.ptrs, np.int32), np.ascontiguousarray(lr.data, np.int32)]
        keep += arrs
        h = P()
        call("cache", 0, ctx, p + 1, c.n_local, len(ns), vp(arrs[0]), vp(arrs[1]), vp(arrs[2]), len(nr), vp(arrs[3]), vp(arrs[4]),
             vp(arrs[5]), 1, ctypes.byref(h))
        plans.append(h)
    CONSISTENT, OWN, GHOST = 0, 0, 1
    for p in range(nparts):                                   # t = consistent!(b): pack ...
        call("impl", 0, plans[p], xs[p], CONSISTENT)
    arr = (ctypes.c_void_p * nparts)(*[h.value for h in plans])
    call("local", 0, arr, nparts, CONSISTENT)                 # ... exchange!
    for p in range(nparts):                                   # foreach(spmv!, own_values(c), own_own, own_values(b))
        call("spmv", 0, oo[p], xs[p], OWN, ys[p], OWN, 1.0, 0.0)
    for p in range(nparts):                                   # wait(t)
        call("impl", 1, plans[p], xs[p], CONSISTENT)
    for p in range(nparts):                                   # foreach(muladd!, own_values(c), own_ghost, ghost_values(b))
        call("mul5", 0, oh[p], xs[p], GHOST, ys[p], OWN, 1.0, 1.0)
    want = [np.zeros(r.n_local) for r in A.rows]
    orc.mul(want, A, [v.copy() for v in xh])
    for p in range(nparts):
        got = np.zeros(A.rows[p].n_own)
        call("download", 0, ys[p], vp(got), 0, len(got))
        assert np.array_equal(got, want[p][:len(got)]), f"part {p + 1}"


@pytest.mark.gpu
def test_glue_blas1_ccalls_replay_the_reference_cg_statements():
    """The statements of the reference CG loop on vectors (HPCG/src/ref_cg.jl:56-71: u .= c .+ beta .* u, x .+= alpha .* u,
    r .-= alpha .* c, u .= 0, copyto!(r,b), norm(r)) as the glue lowers them -- `_lin` flattens the broadcast tree into
    (coefficient, vector) pairs and `copyto!` issues pa_vec_axpby(dest, a, v, b) -- replayed with the type tuples written in
    the glue, bit for bit against the same statements in numpy (unfused multiply, then add)."""
    import numpy as np
    pa = load_package()
    lib = ctypes.CDLL(pa.LIB_PATH)
    seq = {
        "ctx": _glue_ccalls(r"function context\("),
        "vec": _glue_ccalls(r"function HIPVector\(n_own::Integer, n_ghost::Integer, l2d"),
        "upload": _glue_ccalls(r"function upload!\(v::HIPVector"),
        "download": _glue_ccalls(r"function Base\.Array\(v::HIPVector\)"),
        "axpby": _glue_ccalls(r"_axpby!\(y::HIPSegment"),
        "fill": _glue_ccalls(r"Base\.fill!\(s::HIPSegment"),
        "copy": _glue_ccalls(r"Base\.copyto!\(dest::HIPSegment, src::HIPSegment\)"),
        "dot": _glue_ccalls(r"function LinearAlgebra\.dot\(a::HIPSegment"),
    }
    assert [n for n, _ in seq["axpby"]] == ["pa_vec_axpby"] and [n for n, _ in seq["copy"]] == ["pa_vec_copy"]
    assert [n for n, _ in seq["fill"]] == ["pa_vec_fill"] and [n for n, _ in seq["dot"]] == ["pa_vec_dot"]

    def call(key, *vals):
        name, types = seq[key][0]
        assert len(types) == len(vals), (name, len(types), len(vals))
        f = getattr(lib, name)
        f.restype, f.argtypes = ctypes.c_int, types
        assert f(*vals) == 0, (name, lib.pa_last_error())

    P = ctypes.c_void_p
    vp = lambda a: a.ctypes.data_as(P)
    ctx = P()
    call("ctx", 0, ctypes.byref(ctx))
    n_own, n_ghost, OWN = 70001, 13, 0
    rng = np.random.default_rng(5)
    host = {k: rng.standard_normal(n_own + n_ghost) * 10.0 ** rng.integers(-3, 4, n_own + n_ghost) for k in "xucrb"}
    dev = {}
    for k, h in host.items():
        dev[k] = P()
        call("vec", ctx, n_own, n_ghost, ctypes.byref(dev[k]))
        call("upload", dev[k], vp(h), 0, len(h))
    alpha, beta = 0.7310585786300049, 1.0 / 3.0
    o = slice(0, n_own)
    # u .= c .+ beta .* u      ->  terms [(c,1),(u,beta)], dest u:  axpby(u, 1, c, beta)
    call("axpby", dev["u"], 1.0, dev["c"], beta, OWN)
    host["u"][o] = host["c"][o] + beta * host["u"][o]
    # x .+= alpha .* u         ->  [(x,1),(u,alpha)], dest x:       axpby(x, alpha, u, 1)
    call("axpby", dev["x"], alpha, dev["u"], 1.0, OWN)
    host["x"][o] = host["x"][o] + alpha * host["u"][o]
    # r .-= alpha .* c         ->  [(r,1),(c,-alpha)], dest r:      axpby(r, -alpha, c, 1)
    call("axpby", dev["r"], -alpha, dev["c"], 1.0, OWN)
    host["r"][o] = host["r"][o] - alpha * host["c"][o]
    # c .= x .- u  (neither is dest)  ->  axpby(c, 1, x, 0); axpby(c, -1, u, 1)
    call("axpby", dev["c"], 1.0, dev["x"], 0.0, OWN)
    call("axpby", dev["c"], -1.0, dev["u"], 1.0, OWN)
    host["c"][o] = host["x"][o] - host["u"][o]
    # norm(r): sqrt(dot(r,r)) over the own values -- pairwise order is the library's, so compare to the documented tolerance
    out = ctypes.c_double(0.0)
    call("dot", dev["r"], dev["r"], ctypes.byref(out))
    assert abs(out.value - float(host["r"][o] @ host["r"][o])) <= 1e-12 * out.value
    # copyto!(own_values(b), own_values(r)); fill!(own_values(u), 0): ghosts untouched
    call("copy", dev["b"], dev["r"], OWN)
    host["b"][o] = host["r"][o]
    call("fill", dev["u"], OWN, 0.0)
    host["u"][o] = 0.0
    for k in "xucrb":
        got = np.empty(n_own + n_ghost)
        call("download", dev[k], vp(got), 0, len(got))
        assert np.array_equal(got, host[k]), k


def _glue_local_to_device(own_to_local, ghost_to_local):
    """`local_to_device(indices)` of the glue restated (it cannot run here): 1-based device position of every local id --
    own ids first in own_to_local order, then the ghosts -- or None when the local order already is [own | ghost]."""
    no, ng = len(own_to_local), len(ghost_to_local)
    if list(own_to_local) == list(range(1, no + 1)) and list(ghost_to_local) == list(range(no + 1, no + ng + 1)):
        return None
    l2d = [0] * (no + ng)
    for k, l in enumerate(own_to_local, 1):
        l2d[l - 1] = k
    for k, l in enumerate(ghost_to_local, 1):
        l2d[l - 1] = no + k
    return l2d


def test_glue_restatement_of_local_to_device_is_the_glue_s():
    """The two loops of the glue's local_to_device and the places that use it, checked in its source: the restatement above
    is what the replay tests run."""
    src = _glue_src()
    m = re.search(r"function local_to_device\(indices\)(.*?)\nend", src, re.S)
    assert m
    body = m.group(1)
    assert "own_to_local(indices), ghost_to_local(indices)" in body
    assert "for (k, l) in enumerate(o2l); l2d[l] = k; end" in body and "for (k, l) in enumerate(g2l); l2d[l] = no + k; end" in body
    assert "o2l == 1:no && g2l == (no + 1):(no + ng)) && return nothing" in body
    assert src.count("local_to_device(indices)") >= 3 and "l2d = local_to_device(ids)" in src         # allocate_local_values x2, the plan
    assert "dev(is.data)" in src and "dev(ir.data)" in src                                            # plan built from device positions
    assert "dev[v.l2d[l]] = host[l]" in src and "dev[v.l2d]" in src                                   # upload / Array speak the local order
    # dot / norm: own values only, anything else is refused
    m = re.search(r"function LinearAlgebra\.dot\(a::HIPSegment, b::HIPSegment\)(.*?)\nend", src, re.S)
    assert m and "a.seg == PA_SEG_OWN && b.seg == PA_SEG_OWN" in m.group(1) and "error(" in m.group(1)
    # broadcast: no division, no scalar applied to a sum or to a scaled vector
    assert "f === (/)" not in src and "would round differently" in src


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["hand_made_local_indices", "uniform_ghost_periodic"])
def test_glue_consistent_on_permuted_local_indices_replayed_through_ctypes(orc, which):
    """the glue's consistent! on partitions whose local order is NOT [own | ghost] -- the hand-made
    LocalIndices of test/p_vector_tests.jl:93-124 (tests/golden p_vector_local_indices) and
    uniform_partition(ranks,(2,2),(6,6),(true,true),(true,true)) = PermutedLocalIndices (src/p_range.jl:1372) -- replayed with the
    type tuples written in the glue: allocate_local_values -> pa_vec_create + the l2d map, upload in local order,
    p_vector_cache_impl -> pa_plan_create on DEVICE positions, assemble_impl!(insert, reverse(cache)) -> pack / exchange /
    finish, Array(v) back in local order.  Every local value must equal 10 * its owner (:116-124), bit for bit; then
    assemble! on the same plans against the oracle."""
    import json
    import numpy as np
    pa = load_package()
    lib = ctypes.CDLL(pa.LIB_PATH)
    seq = {
        "ctx": _glue_ccalls(r"function context\("),
        "vec": _glue_ccalls(r"function HIPVector\(n_own::Integer, n_ghost::Integer, l2d"),
        "upload": _glue_ccalls(r"function upload!\(v::HIPVector"),
        "download": _glue_ccalls(r"function Base\.Array\(v::HIPVector\)"),
        "cache": _glue_ccalls(r"function PartitionedArrays\.p_vector_cache_impl"),
        "local": _glue_ccalls(r"_transport!\(plans::DebugArray"),
        "impl": _glue_ccalls(r"function PartitionedArrays\.assemble_impl!"),
    }
    assert [n for n, _ in seq["upload"]] == ["pa_vec_upload"] and [n for n, _ in seq["download"]] == ["pa_vec_download"]

    def call(key, k, *vals):
        name, types = seq[key][k]
        assert len(types) == len(vals), (name, len(types), len(vals))
        f = getattr(lib, name)
        f.restype, f.argtypes = ctypes.c_int, types
        assert f(*vals) == 0, (name, lib.pa_last_error())

    P = ctypes.c_void_p
    vp = lambda a: a.ctypes.data_as(P)
    if which == "hand_made_local_indices":
        c = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_literals.json")))["p_vector_local_indices"]
        cols = [orc.Indices(c["n"], p + 1, np.array(g), np.array(o)) for p, (g, o) in enumerate(zip(c["local_to_global"], c["local_to_owner"]))]
    else:
        cols = orc.uniform_partition((2, 2), (6, 6), (1, 1), (True, True))
    assert any(_glue_local_to_device(ci.own_to_local, ci.ghost_to_local) is not None for ci in cols)    # really permuted
    cache = orc.p_vector_cache([np.zeros(ci.n_local) for ci in cols], cols)
    ctx = P()
    call("ctx", 0, 0, ctypes.byref(ctx))
    vecs, plans, l2ds, keep = [], [], [], []
    for p, ci in enumerate(cols):
        l2d = _glue_local_to_device(ci.own_to_local, ci.ghost_to_local)
        l2ds.append(l2d)
        h = P()
        call("vec", 0, ctx, ci.n_own, ci.n_ghost, ctypes.byref(h))
        vecs.append(h)
        dev = (lambda lids: np.ascontiguousarray(lids, np.int32)) if l2d is None else \
              (lambda lids, l2d=l2d: np.array([l2d[l - 1] for l in lids], np.int32))
        ns, nr = np.ascontiguousarray(cache.neighbors_snd[p], np.int32), np.ascontiguousarray(cache.neighbors_rcv[p], np.int32)
        ls, lr = cache.local_indices_snd[p], cache.local_indices_rcv[p]
        arrs = [ns, np.ascontiguousarray(ls.ptrs, np.int32), dev(ls.data), nr, np.ascontiguousarray(lr.ptrs, np.int32), dev(lr.data)]
        keep += arrs
        hp = P()
        call("cache", 0, ctx, p + 1, ci.n_local, len(ns), vp(arrs[0]), vp(arrs[1]), vp(arrs[2]), len(nr), vp(arrs[3]), vp(arrs[4]),
             vp(arrs[5]), 1, ctypes.byref(hp))
        plans.append(hp)

    def upload(p, host):                       # upload!(v, host): local order -> device layout
        l2d, dev = l2ds[p], np.array(host, float)
        if l2d is not None:
            dev = np.empty(len(host))
            for l, val in enumerate(host):
                dev[l2d[l] - 1] = val
        call("upload", 0, vecs[p], vp(dev), 0, len(dev))

    def array(p):                              # Array(v): device layout -> local order
        dev = np.empty(cols[p].n_local)
        call("download", 0, vecs[p], vp(dev), 0, len(dev))
        return dev if l2ds[p] is None else dev[np.array(l2ds[p]) - 1]

    def exchange(mode):
        for p in range(len(cols)):
            call("impl", 0, plans[p], vecs[p], mode)
        arr = (ctypes.c_void_p * len(cols))(*[h.value for h in plans])
        call("local", 0, arr, len(cols), mode)
        for p in range(len(cols)):
            call("impl", 1, plans[p], vecs[p], mode)
    CONSISTENT, ASSEMBLE = 0, 1
    for p, ci in enumerate(cols):              # v[l] = 10*part on own ids, 0 on ghosts (test/p_vector_tests.jl:105-113)
        upload(p, 10.0 * ci.part * (ci.local_to_owner == ci.part))
    exchange(CONSISTENT)
    for p, ci in enumerate(cols):
        assert array(p).tolist() == (10.0 * ci.local_to_owner).tolist(), f"part {p + 1}"
    # assemble! on the same plans: local values l + part/16, against the oracle's assemble!
    host = [np.arange(1, ci.n_local + 1) + ci.part / 16.0 for ci in cols]
    for p in range(len(cols)):
        upload(p, host[p])
    exchange(ASSEMBLE)
    want = [h.copy() for h in host]
    orc.assemble(want, cols, cache)
    for p in range(len(cols)):
        assert np.array_equal(array(p), want[p]), f"assemble! part {p + 1}"


def test_julia_runtests_and_reference_bench_scripts_are_consistent_with_the_glue():
    """a Julia maintainer gets one command (julia/runtests.jl) and the host-side reference timing BASELINE.md
    promises (bench/julia_reference.jl).  Neither can run here; statically: every name runtests.jl imports from the glue is
    defined there, its literal expectations are the reference test file's (test/p_vector_tests.jl:129-139), and the scripts
    cite the reference lines they replay."""
    glue = open(os.path.join(ROOT, "partitionedarrays.jl_amd", "julia", "PartitionedArraysHIP.jl")).read()
    rt = open(os.path.join(ROOT, "partitionedarrays.jl_amd", "julia", "runtests.jl")).read()
    names = re.search(r"using \.PartitionedArraysHIP: ([^\n]+)", rt).group(1)
    for n in [x.strip() for x in names.split(",")]:
        assert re.search(r"(?m)^(?:function |mutable struct |struct |const )?" + re.escape(n) + r"(?=[\s({<:])", glue) or \
            re.search(r"(?m)^" + re.escape(n) + r"\(", glue), n
    for lit in ("[20.0, 20.0, 20.0, 0.0, 0.0, 0.0]", "[0.0, 20.0, 30.0, 0.0]", "[10.0, 30.0, 20.0, 0.0, 0.0, 0.0]", "[0.0, 0.0, 0.0, 10.0, 30.0]"):
        assert lit in rt
    assert "test/p_vector_tests.jl:93-142" in rt and "test/p_sparse_matrix_tests.jl:207-248" in rt
    bj = open(os.path.join(ROOT, "bench", "julia_reference.jl")).read()
    assert "HPCG.build_p_matrix" in bj and "mul!(c, A, x)" in bj and "consistent!(x)" in bj
    assert "hpcg_blocks_hip" in glue and "pa_hpcg_own_block_create" in glue



@pytest.mark.gpu
def test_glue_split_vector_conversion_and_psparse_reuse_replayed_through_ctypes(orc):
    """VERDICT r05 "Next" #9, the two pieces the glue lacked, replayed with the type tuples WRITTEN IN THE GLUE:
    (i) to_hip(::PVector{<:SplitVector}) (src/p_vector.jl:132-187): pa_vec_create, then the own block and the ghost block uploaded as
        they are at offsets 0 and n_own; downloaded through Base.Array's ccall the vector reads [own | ghost];
    (ii) psparse_hip!(C,V,cache) (psparse!, src/p_sparse_matrix.jl:1291-1305): upload of V, pa_scatter_add into W, assemble! of W
        (pa_exchange_pack / pa_exchange_local / pa_exchange_finish over the cache's plans), pa_csr_update_values_from twice per part --
        on a cache the Python route built with the same entry points (pa_coo_reuse_scatter, pa_plan_create), against the oracle's
        re-assembly, bit for bit.  The call sequence of the glue's reuse=true branch itself is pinned by name below."""
    import numpy as np
    pa = load_package()
    import pa_amd._lib as L
    lib = ctypes.CDLL(pa.LIB_PATH)
    P = ctypes.c_void_p
    vp = lambda a: a.ctypes.data_as(P)       # noqa: E731
    seq = {
        "ctx": _glue_ccalls(r"function context\("),
        "vec": _glue_ccalls(r"function HIPVector\(n_own::Integer, n_ghost::Integer, l2d"),
        "split": _glue_ccalls(r"function to_hip\(v::PVector\{<:PartitionedArrays\.SplitVector\}\)"),
        "download": _glue_ccalls(r"function Base\.Array\(v::HIPVector\)"),
        "upload": _glue_ccalls(r"function upload!\(v::HIPVector"),
        "psparse!": _glue_ccalls(r"function psparse_hip!\(C::PSparseMatrix"),
        "assemble": _glue_ccalls(r"function PartitionedArrays\.assemble_impl!\(f, vector_partition, cache::HIPAssemblyCache\)"),
        "local": _glue_ccalls(r"_transport!\(plans::DebugArray"),
        "reuse": _glue_ccalls(r"function psparse_disassembled_hip\(I, J, V, rows, cols; reuse::Bool=false\)"),
    }
    assert [n for n, _ in seq["split"]] == ["pa_vec_upload", "pa_vec_upload"]
    assert [n for n, _ in seq["psparse!"]] == ["pa_scatter_add", "pa_csr_update_values_from", "pa_csr_update_values_from"]
    names = [n for n, _ in seq["reuse"]]
    for must in ("pa_coo_keep_input_slots", "pa_coo_subassemble", "pa_coo_assemble_finish", "pa_coo_assembly_info", "pa_coo_reuse_scatter",
                 "pa_plan_create", "pa_coo_assembly_destroy"):
        assert must in names, must
    assert names.index("pa_coo_subassemble") < names.index("pa_coo_assemble_finish") < names.index("pa_coo_reuse_scatter") < names.index("pa_plan_create")

    def call(entry, *vals):
        name, types = entry
        assert len(types) == len(vals), (name, len(types), len(vals))
        f = getattr(lib, name)
        f.restype, f.argtypes = ctypes.c_int, types
        assert f(*vals) == 0, (name, lib.pa_last_error())
    ctx = P()
    call(seq["ctx"][0], 0, ctypes.byref(ctx))
    # (i) a SplitVector's two blocks
    own, ghost = np.arange(1.0, 8.0), np.array([70.0, 80.0, 90.0])
    d = P()
    call(seq["vec"][0], ctx, len(own), len(ghost), ctypes.byref(d))
    call(seq["split"][0], d, vp(own), 0, len(own))
    call(seq["split"][1], d, vp(ghost), len(own), len(ghost))
    back = np.zeros(len(own) + len(ghost))
    call(seq["download"][0], d, vp(back), 0, len(back))
    assert back.tolist() == own.tolist() + ghost.tolist()
    # (ii) psparse! on a FEM matrix of four parts
    nodes, parts = (23, 17), (2, 2)
    ranks = pa.DebugArray([1, 2, 3, 4])
    I, J, V, rows, cols = pa.laplacian_fem(nodes, parts, ranks)
    A, cache = pa.psparse_disassembled(I, J, V, rows, cols, reuse=True)
    V2 = [v * 1.5 + orc.hash_x(np.arange(len(v)) + 7 * k) * 1e-3 for k, v in enumerate(V.items)]
    for vd, v in zip(cache.Vdev.items, V2):
        call(seq["upload"][0], vd.h, vp(np.ascontiguousarray(v)), 0, len(v))
    for sc, w, vd in zip(cache.scatters.items, cache.W.items, cache.Vdev.items):
        call(seq["psparse!"][0], sc, w.h, vd.h, 1)
    pack = [e for e in seq["assemble"] if e[0] == "pa_exchange_pack"][0]
    fin = [e for e in seq["assemble"] if e[0] == "pa_exchange_finish"][0]
    for w, p in zip(cache.W.items, cache.plans.items):
        call(pack, p, w.h, L.ASSEMBLE)
    arr = (P * 4)(*[p.value for p in cache.plans.items])
    call(seq["local"][0], arr, 4, L.ASSEMBLE)
    for w, p in zip(cache.W.items, cache.plans.items):
        call(fin, p, w.h, L.ASSEMBLE)
    for blk, w, k in zip(A.matrix_partition.items, cache.W.items, cache.nnz_oo.items):
        call(seq["psparse!"][1], blk.own_own.h, w.h, 0)
        call(seq["psparse!"][2], blk.own_ghost.h, w.h, int(k))
    Io, Jo, Vo, orows, ocols = orc.laplacian_fem(nodes, parts)
    Ao, _ = orc.psparse_disassembled(Io, Jo, [np.asarray(v).copy() for v in V2], orows, ocols)
    xo = [orc.hash_x(c.local_to_global) * (c.local_to_owner == c.part) for c in Ao.cols]
    x = pa.pvector_from_function(lambda ind: orc.hash_x(ind.get_local_to_global()) * (ind.get_local_to_owner() == ind.part), A.col_partition)
    y = pa.pzeros(A.row_partition)
    pa.mul_(y, A, x)
    yo = [np.zeros(r.n_local) for r in Ao.rows]
    orc.mul(yo, Ao, [v.copy() for v in xo])
    for got, e, r in zip(y.own_values().items, yo, Ao.rows):
        assert np.array_equal(got, e[:r.n_own])


def test_every_translation_unit_of_csrc_is_in_the_makefile():
    """A source file that the Makefile does not list is a kernel that never ships: every .hip / .cpp under csrc/ is in SRC, every
    header in HDR (a changed header must rebuild the objects that include it)."""
    import glob
    import re
    csrc = os.path.join(ROOT, "partitionedarrays.jl_amd", "csrc")
    mk = open(os.path.join(csrc, "Makefile")).read()
    src = set(re.search(r"^SRC\s*:=\s*(.*)$", mk, re.M).group(1).split())
    hdr = set(os.path.basename(h) for h in re.search(r"^HDR\s*:=\s*(.*)$", mk, re.M).group(1).split())
    have = set(os.path.basename(f) for f in glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.cpp")))
    assert have == src, (sorted(have - src), sorted(src - have))
    headers = set(os.path.basename(f) for f in glob.glob(os.path.join(csrc, "*.h")))
    assert headers <= hdr, sorted(headers - hdr)
