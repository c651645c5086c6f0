"""Corrupted inputs at the C ABI: a valid CSR block / exchange plan with ONE thing broken at random (a column out of range,
a row pointer that decreases or overshoots, a wrong entry count, a local id outside the vector, a neighbour id that is no
part, ptrs that do not start at 1, ...).  Every call must come back with a status < 0 and a message -- never a fault, never
a silently accepted object whose product then reads out of bounds.  python tests/fuzz/fuzz_bad_inputs.py [cases] [seed0]"""
import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from __graft_entry__ import load_package
pa = load_package()
import pa_amd._lib as L
ctx = pa.context()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
accepted, rejected = [], 0
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    m, n = int(rng.integers(1, 300)), int(rng.integers(1, 300))
    lens = rng.integers(0, 6, m)
    rp = np.concatenate([[1], 1 + np.cumsum(lens)]).astype(np.int32)
    nnz = int(rp[-1] - 1)
    col = np.sort(rng.integers(1, n + 1, nnz)).astype(np.int32) if nnz else np.zeros(0, np.int32)
    val = rng.standard_normal(nnz)
    kind = int(rng.integers(0, 9))
    what = ""
    try:
        if kind == 0 and nnz:
            col[int(rng.integers(0, nnz))] = n + int(rng.integers(1, 1000)); what = "column > n_cols"
        elif kind == 1 and nnz:
            col[int(rng.integers(0, nnz))] = int(rng.integers(-1000, 1)); what = "column < 1"
        elif kind == 2 and m > 1:
            i = int(rng.integers(1, m)); rp[i] = rp[i - 1] - 1 - int(rng.integers(0, 5)); what = "row pointer decreases"
        elif kind == 3:
            rp[-1] = rp[-1] + int(rng.integers(1, 50)); what = "last row pointer beyond nnz"
        elif kind == 4:
            rp[0] = int(rng.integers(2, 5)); what = "row pointers do not start at the index base"
        elif kind in (5, 6, 7, 8):
            # exchange plan: one part, n_local, one fake neighbour each way
            n_local = m + 3
            ns = np.array([2], np.int32); nr = np.array([2], np.int32)
            sp = np.array([1, 3], np.int32); sd = np.array([m + 1, m + 2], np.int32)
            rpn = np.array([1, 2], np.int32); rd = np.array([1], np.int32)
            if kind == 5: sd[0] = n_local + int(rng.integers(1, 100)); what = "plan: send local id > n_local"
            elif kind == 6: rd[0] = int(rng.integers(-50, 1)); what = "plan: receive local id < 1"
            elif kind == 7: sp[0] = 2; what = "plan: ptrs do not start at 1"
            else: ns[0] = int(rng.integers(-5, 1)); what = "plan: neighbour id < 1"
            h = C.c_void_p()
            L.call("pa_plan_create", ctx.h, 1, n_local, 1, L.ptr(ns), L.ptr(sp), L.ptr(sd), 1, L.ptr(nr), L.ptr(rpn), L.ptr(rd), 1, C.byref(h))
            accepted.append((seed0 + case, what)); continue
        else:
            continue
        H = pa.HostCSR(m, n, rp, col, val)
        A = pa.DeviceCSR(H)
        x = pa.DeviceVector(n, 0).upload(rng.standard_normal(n))
        y = pa.DeviceVector(m, 0)
        pa.spmv_(y, A, x); ctx.sync()
        accepted.append((seed0 + case, what))
    except L.PAError as e:
        rejected += 1
    except (AssertionError, ValueError) as e:       # the Python mirror's own checks
        rejected += 1
print(f"done: {n_cases} cases, {rejected} rejected with a status, {len(accepted)} accepted")
for a in accepted[:20]: print("  ACCEPTED:", a)
