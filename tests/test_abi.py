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
