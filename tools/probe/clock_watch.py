"""Is the product kernel's fast/slow mode a clock / power state?  Long K1 loops on several block re-creations while a
thread samples the sysfs clock levels and the socket power."""
import glob, sys, threading, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from __graft_entry__ import load_package
pa = load_package()
import pa_amd._lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = pa.context()
A, b = pa.build_p_matrix(pa.DebugArray([1]), n, n, n, n, n, n, 1, 1, 1, keep_host=True)
h = pa.local_items(A.host_blocks)[0][0]
rows = h.m
del A
dev = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
base = dev[0].rsplit("/", 1)[0] if dev else None
power = sorted(glob.glob(base + "/hwmon/hwmon*/power1_*")) if base else []
print("sysfs:", base, [p.rsplit('/', 1)[1] for p in power])


def cur(name):
    try:
        for line in open(f"{base}/{name}"):
            if line.strip().endswith("*"):
                return line.split(":")[1].strip().rstrip("*").strip()
    except OSError:
        return "?"
    return "?"


def watts():
    for p in power:
        if p.endswith("power1_average") or p.endswith("power1_input"):
            try:
                return int(open(p).read()) / 1e6
            except (OSError, ValueError):
                pass
    return -1.0


samples, stop = [], False


def sampler():
    while not stop:
        samples.append((time.perf_counter(), cur("pp_dpm_sclk"), cur("pp_dpm_fclk"), cur("pp_dpm_mclk"), cur("pp_dpm_socclk"), watts()))
        time.sleep(0.05)


th = threading.Thread(target=sampler)
th.start()
rng = np.random.default_rng(0)
keep = []
for t in range(6):
    dA = pa.DeviceCSR(h)
    x = pa.DeviceVector(rows, 0).upload(rng.random(rows))
    y = pa.DeviceVector(rows, 0)
    t0 = time.perf_counter()
    e0 = ctx.event().record(L.STREAM_COMPUTE)
    for _ in range(1500):
        L.call("pa_spmv", dA.h, x.h, L.SEG_OWN, y.h, L.SEG_OWN, 1.0, 0.0)
    e1 = ctx.event().record(L.STREAM_COMPUTE)
    ctx.sync()
    t1 = time.perf_counter()
    s = [q for q in samples if t0 + 0.3 < q[0] < t1]
    def mode(k):
        v = [q[k] for q in s]
        return max(set(v), key=v.count) if v else "?"
    print(f"block {t}: {e0.elapsed_ms(e1) / 1500:.4f} ms/launch over {t1 - t0:.2f} s | sclk {mode(1)} fclk {mode(2)} mclk {mode(3)} socclk {mode(4)} "
          f"power {np.mean([q[5] for q in s]) if s else -1:.0f} W (max {max([q[5] for q in s]) if s else -1:.0f})", flush=True)
    keep.append((dA, x, y))
stop = True
th.join()
