"""The C ABI under callers that get it wrong: NULL pointers, degenerate sizes, one wrong field in an otherwise well-formed description.
Every such call must return a negative status with a message -- never crash, never throw across the boundary, never leave the context
unusable.  The calls run in a child process (tests/helpers/null_sweep_worker.py) so that a crash is a test failure with the name of
the call that died, not the end of the test session."""
import os
import subprocess
import sys

import pytest

WORKER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "null_sweep_worker.py")


def _run(mode):
    out = subprocess.run([sys.executable, WORKER, mode], capture_output=True, text=True, timeout=900)
    lines = [l for l in out.stdout.strip().splitlines() if l.strip()]
    assert out.returncode == 0 and lines and lines[-1].startswith("done"), (
        f"the child died (exit {out.returncode}) in: {lines[-1] if lines else '(nothing printed)'}\n{out.stderr[-1500:]}")
    return lines[:-1], int(lines[-1].split()[1])


def test_null_context_every_entry_returns_invalid():
    """(no GPU needed: without a context every entry refuses before it touches one)"""
    lines, n = _run("null-ctx")
    assert n > 120
    for l in lines:
        assert int(l.split()[-1]) < 0, l


@pytest.mark.gpu
def test_live_context_null_pointers_and_degenerate_sizes():
    lines, n = _run("live-ctx")
    assert n > 120
    accepted = [l for l in lines if int(l.split()[-1]) >= 0]
    # what may succeed with nothing but a context: the calls that take no array or treat an empty one as nothing to do
    harmless = ("bms_ctx_", "bms_host_", "bms_xor_timeseries", "bms_fletcher32", "bms_row_norm", "bms_rotate_", "bms_multishuffle")
    for l in accepted:
        assert l.startswith(harmless), l


@pytest.mark.gpu
def test_one_wrong_field_at_a_time():
    lines, n = _run("live-structs")
    assert n > 100
    for l in lines:
        parts = l.split()
        if parts[1] == "well-formed":
            assert int(parts[2]) == 0, l
        elif parts[1] == "well-formed-again":
            assert int(parts[2]) == 0 and int(parts[3]) == 1, l  # the context still gives the first answer, bit for bit
        else:
            assert int(parts[-1]) < 0, l


@pytest.mark.gpu
def test_building_blocks_with_one_wrong_scalar():
    lines, n = _run("live-blocks")
    assert n > 40
    for l in lines:
        label, want, rc = l.split()
        assert (int(rc) == 0) if want == "want0" else (int(rc) < 0), l
