"""What the context's arena costs to build, by size (PA_ARENA_GIB)."""
import os, sys, time, subprocess
if len(sys.argv) == 2:
    import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from __graft_entry__ import load_package
    pa = load_package()
    ctx = pa.context()
    t = time.perf_counter()
    info = ctx.arena(build=True)
    ctx.sync()
    print(f"PA_ARENA_GIB={os.environ.get('PA_ARENA_GIB', 'default')}: build {time.perf_counter() - t:.2f} s  -> {info['gib']:.0f} GiB, map {info['map_ms']:.0f} ms, classes {info['class_gib']}", flush=True)
else:
    for g in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("8", "32", "96", "")):
        e = dict(os.environ)
        if g: e["PA_ARENA_GIB"] = g
        else: e.pop("PA_ARENA_GIB", None)
        subprocess.run([sys.executable, __file__, "child"], env=e)

