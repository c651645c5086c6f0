"""The C-ABI library loads (no GPU needed) and exports every symbol include/pa_hip.h declares."""
import ctypes
import os
import re

from __graft_entry__ import ROOT, load_package


def _declared():
    txt = open(os.path.join(ROOT, "include", "pa_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
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
    txt = open(os.path.join(ROOT, "include", "pa_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
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
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / r["avg_launch_ms"] / 1e6) < 1.0
    assert abs(d["value"] - 2 * d["config"]["nnz_per_part"] * d["n_gpus"] / d["ms_per_step"] / 1e6) < 0.5


def test_header_is_valid_c99_and_the_c_example_links():
    """include/pa_hip.h is a C header (extern "C" guards, no C++ in the declarations): gcc -std=c99 -Werror accepts it,
    and examples/c_abi_smoke.c -- the ABI used from plain C -- compiles and links against libpa_hip.so."""
    import subprocess
    load_package()
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c",
                           os.path.join(inc, "pa_hip.h")])
    out = os.path.join(ROOT, "examples", "c_abi_smoke")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", inc,
                           os.path.join(ROOT, "examples", "c_abi_smoke.c"), "-L", os.path.join(ROOT, "partitionedarrays.jl_amd"),
                           "-lpa_hip", "-Wl,-rpath," + os.path.join(ROOT, "partitionedarrays.jl_amd"),
                           "-Wl,-rpath,/opt/rocm/lib", "-o", out])
    assert os.path.exists(out)
