"""Random blocks through every launch the library can choose (default, x windows forced, row split only, SELL) against the
oracle's spmv_csr! / mul! loops, bit for bit.  python tests/fuzz/fuzz_spmv.py [cases] [seed0]"""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from __graft_entry__ import load_package, load_oracle
pa = load_package()
orc = load_oracle()
import pa_amd._lib as L
args = [a for a in sys.argv[1:] if not a.startswith("--")]
n_cases = int(args[0]) if len(args) > 0 else 100
seed0 = int(args[1]) if len(args) > 1 else 0
t0 = time.time()
bad = 0
cover = {"default_windows": 0, "default_big_windows": 0, "forced_windows": 0, "rest_chunks": 0, "padded_slots_rows_of_8k": 0, "c32_chunks": 0}
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    m = int(rng.integers(20_000, 400_000))
    n = m if rng.random() < 0.6 else int(m * rng.uniform(0.5, 1.6)) + 1
    law = int(rng.integers(0, 8))
    if law >= 6:
        # round 6: rows that follow column PATTERNS (what pattern-ELL storage is for, csrc/pa_pell.h): a handful of offsets (with runs of
        # three consecutive ones half of the time: the kernel's one-gather-per-run form), every row keeps a random subset of them -- the
        # row masks --, offsets that leave the matrix are dropped as at a grid's faces, a few rows are empty
        m = n = int(rng.integers(1, 40000))
        if rng.random() < 0.5:
            starts = np.unique(rng.integers(-min(n, 3000), min(n, 3000), int(rng.integers(1, 10))))
            offs = np.unique(np.concatenate([starts, starts + 1, starts + 2]))
        else:
            offs = np.unique(rng.integers(-min(n, 3000), min(n, 3000), int(rng.integers(1, 28))))
        keep = rng.random((m, len(offs))) < rng.choice([1.0, 0.97, 0.6])
        keep[rng.random(m) < 0.01] = False
        cols2 = np.arange(m)[:, None] + offs[None, :]
        keep &= (cols2 >= 0) & (cols2 < n)
        lens = keep.sum(1)
        rows = np.repeat(np.arange(m), lens)
        col = cols2[keep]
        band = -1
    elif law == 0: lens = np.full(m, int(rng.integers(1, 33)))
    elif law == 1: lens = rng.integers(0, int(rng.integers(2, 70)), m)
    elif law == 2: lens = np.where(rng.random(m) < 0.01, rng.integers(500, 4000, m), rng.integers(0, 10, m))
    elif law == 3: lens = np.minimum((rng.pareto(1.5, m) * 3).astype(np.int64), 3000)
    elif law == 4: lens = np.where(rng.random(m) < 0.5, 0, rng.integers(1, 20, m))
    elif law == 5: lens = np.repeat(rng.integers(1, 40, (m + 63) // 64), 64)[:m]
    if law < 6: band = int(rng.choice([8, 60, 400, 1500, 2300, 3500, 6000, 9000, 10**9]))
    rp = np.concatenate([[1], 1 + np.cumsum(lens)]).astype(np.int64)
    if rp[-1] > 40_000_000:
        continue
    if law < 6:
        rows = np.repeat(np.arange(m), lens)
        centre = (rows * (n / m)).astype(np.int64)
        if band >= 10**9: col = rng.integers(0, n, len(rows))
        else: col = np.clip(centre + rng.integers(-band, band + 1, len(rows)), 0, n - 1)
    order = np.lexsort((col, rows))
    val = rng.standard_normal(len(rows))
    if "--few-values" in sys.argv:                    # with PA_SPMV_VALUE_DICT=1: the lossless value dictionary (<= 64 values)
        nv = int(rng.integers(1, 80))
        val = rng.standard_normal(nv)[rng.integers(0, nv, len(rows))]
    two = "--two-values" in sys.argv                 # ... at most two values: the select decode (VD = 2, round 5)
    if two:
        nv = int(rng.integers(1, 3))
        val = rng.standard_normal(nv)[rng.integers(0, nv, len(rows))]
    if not (two and nv == 2): val[rng.random(len(rows)) < 0.01] = -0.0
    H = pa.HostCSR(m, n, rp.astype(np.int32), (col[order] + 1).astype(np.int32), val)
    Ho = orc.CSR(m, n, H.rowptr, H.colval, H.nzval)
    xh = rng.standard_normal(n)
    want = np.zeros(m); orc.oracle_c().spmv_csr(want, xh, Ho)
    alpha, beta = float(rng.standard_normal()), float(rng.choice([0.0, 1.0, rng.standard_normal()]))
    y0 = rng.standard_normal(m); want5 = y0.copy(); orc.oracle_c().mul5_csr(want5, Ho, xh, alpha, beta)
    ghost = rng.random() < 0.3
    x = (pa.DeviceVector(3, n).upload(np.concatenate([np.zeros(3), xh])) if ghost else pa.DeviceVector(n, 0).upload(xh))
    seg = L.SEG_GHOST if ghost else L.SEG_OWN
    for sw in (None, "2", "0"):
        if sw is None: os.environ.pop("PA_SPMV_XWIN", None)
        else: os.environ["PA_SPMV_XWIN"] = sw
        A = pa.DeviceCSR(H)
        xw, enc = A.xwin(), A.encoding()
        if sw is None:
            pm = A.pell()
            cover["pattern_ell_" + ("none", "fp64", "one_bit", "one_byte")[pm["mode"]]] = cover.get("pattern_ell_" + ("none", "fp64", "one_bit", "one_byte")[pm["mode"]], 0) + (law >= 6)
            cover["pattern_ell_unroll_%d" % pm["unroll"]] = cover.get("pattern_ell_unroll_%d" % pm["unroll"], 0) + (pm["mode"] > 0)
            cover["default_windows"] += xw["groups"] > 0
            cover["default_big_windows"] += xw["big_groups"] > 0
            cover["default_ring"] = cover.get("default_ring", 0) + (xw.get("ring_groups", 0) > 0)
            cover["c32_chunks"] += enc["c32"] > 0
            cover["padded_slots_rows_of_8k"] += bool(law == 0 and lens[0] % 8 == 0)
        if sw == "2":
            cover["forced_windows"] += xw["groups"] > 0
            cover["rest_chunks"] += 0 < xw["chunks"] < A.info()["n_chunks"]
        y = pa.DeviceVector(m, 0)
        pa.spmv_(y, A, x, x_segment=seg)
        ok1 = np.array_equal(y.download(), want)
        y.upload(y0.copy())
        pa.spmv_(y, A, x, x_segment=seg, alpha=alpha, beta=beta)
        ok2 = np.array_equal(y.download(), want5)
        if not (ok1 and ok2):
            bad += 1
            print(f"MISMATCH case {seed0 + case} XWIN={sw}: m {m} n {n} law {law} band {band} nnz {H.nnz} ghost {ghost} spmv {ok1} mul5 {ok2} {A.encoding()} {A.xwin()}", flush=True)
        del A, y
    os.environ.pop("PA_SPMV_XWIN", None)
    # the same block in SELL-C-sigma storage, and its transpose uploaded through the CSC entry point (A' stored by rows =
    # A stored by columns: the product with the transposed block is A' * z, the oracle's mul5_csr_t)
    S = pa.DeviceSELL(H, sigma=int(rng.choice([1, 64, 1024])))
    y = pa.DeviceVector(m, 0)
    pa.spmv_(y, S, x, x_segment=seg)
    if not np.array_equal(y.download(), want):
        bad += 1
        print(f"MISMATCH case {seed0 + case} SELL: m {m} n {n} law {law} band {band}", flush=True)
    zt = rng.standard_normal(m)
    wt = np.zeros(n); orc.oracle_c().mul5_csr_t(wt, Ho, zt, 1.0, 0.0)
    T = pa.DeviceCSR.transposed(H)
    yt = pa.DeviceVector(n, 0)
    pa.spmv_(yt, T, pa.DeviceVector(m, 0).upload(zt))
    if not np.array_equal(yt.download(), wt):
        bad += 1
        print(f"MISMATCH case {seed0 + case} transposed (CSC upload): m {m} n {n} law {law} band {band}", flush=True)
    del S, T, y, yt
    if case % 10 == 9:
        print(f"{case + 1} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print(f"done: {n_cases} cases, {bad} mismatches; blocks by what ran: {cover}")
