"""A short run of every fuzzer of tests/fuzz/ (random inputs against the oracle, bit for bit).  The long
runs are recorded under profiles/r02_fuzz_*.log; these keep the fuzzers themselves working and give every test run a few
hundred inputs nobody wrote by hand (fixed seeds: a failure reproduces with `python tests/fuzz/<script> <cases> <seed0>`)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _fuzz(script, cases, seed0, env=None, extra=()):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", script), str(cases), str(seed0), *extra],
                       cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0 and "MISMATCH" not in r.stdout and "done:" in r.stdout, tail
    return r.stdout


@pytest.mark.parametrize("script,cases,seed0", [("fuzz_spmv.py", 24, 424200), ("fuzz_mul.py", 40, 424300), ("fuzz_fem.py", 60, 424400),
                                                ("fuzz_exchange.py", 200, 424500), ("fuzz_partitions.py", 200, 424600),
                                                ("fuzz_hpcg.py", 30, 424700), ("fuzz_cg.py", 12, 424800)])
def test_fuzzer_short_run(script, cases, seed0):
    out = _fuzz(script, cases, seed0)
    assert " 0 mismatches" in out or " 0 with mismatches" in out, out[-500:]


@pytest.mark.gpu_extended
def test_fuzzers_under_guarded_poisoned_buffers():
    """The same with every device buffer in its own mapping, unmapped space behind it and 0xFF bytes in it (PA_DEBUG_GUARD=2): an
    overrun of more than 16 bytes faults, an uninitialised read shows as NaN."""
    for script, cases, seed0 in (("fuzz_spmv.py", 12, 424900), ("fuzz_mul.py", 30, 425000), ("fuzz_exchange.py", 100, 425100)):
        out = _fuzz(script, cases, seed0, env={"PA_DEBUG_GUARD": "2"})
        assert " 0 mismatches" in out or " 0 with mismatches" in out, out[-500:]


def test_random_call_orders_on_exchange_plans_are_refused_or_right():
    out = _fuzz("fuzz_call_order.py", 150, 425300)
    assert " 0 with mismatches" in out, out[-500:]


def test_corrupted_inputs_are_rejected_with_a_status():
    out = _fuzz("fuzz_bad_inputs.py", 300, 425200)
    assert ", 0 accepted" in out, out[-800:]


@pytest.mark.gpu_extended
@pytest.mark.parametrize("switches", [{"PA_SPMV_VALUE_DICT": "1"}, {"PA_CTX_PER_PART": "1"}, {"PA_PUSH": "0"}, {"PA_MUL_GHOST_FROM_BUFFER": "0"},
                                      {"PA_SPMV_COLSPLIT": "3"}, {"PA_SPMV_PELL": "0"}, {"PA_SPMV_PELL_RUNS3": "0", "PA_SPMV_VALUE_DICT": "1"}])
def test_fuzzers_under_the_round_4_switches(switches):
    """The routes round 4 added or made the default, each forced or switched off: the value dictionary on EVERY block that
    qualifies (the fuzzers' blocks are below the automatic threshold), one device context per part, the round-3 exchange
    (pack + copies), own x ghost behind the unpack, every block of 64 entries or more as a chain of three column pieces."""
    for script, cases, seed0 in (("fuzz_spmv.py", 12, 426000), ("fuzz_mul.py", 30, 426100), ("fuzz_fem.py", 30, 426200), ("fuzz_exchange.py", 100, 426300)):
        out = _fuzz(script, cases, seed0, env=switches)
        assert " 0 mismatches" in out or " 0 with mismatches" in out, (switches, out[-500:])


def test_spmv_fuzzer_on_two_value_dictionaries():
    """Round 5: blocks with at most two stored values take the select decode of the one-byte value stream (VD = 2); every launch the
    library can choose, against the oracle."""
    out = _fuzz("fuzz_spmv.py", 10, 427000, env={"PA_SPMV_VALUE_DICT": "1"}, extra=("--two-values",))
    assert " 0 mismatches" in out, out[-500:]

