"""`make SAN=1` (scri_amd/csrc/Makefile): the host side of the engine -- shard plans, output windows, knot ranges, column parts,
search bounds, rotor / harmonic / conformal tables, the frame integrator -- compiled with AddressSanitizer and
UndefinedBehaviorSanitizer and run over the five BASELINE shapes, 1..8 shards, 1..8 column parts, series of 2..9 samples and odd
grids (tools/san/host_san_driver.cpp).  No GPU sanitizer exists on this pool; this is the part that can be covered."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_planning_code_is_clean_under_asan_and_ubsan():
    env = dict(os.environ, HIPCC=os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"))
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "scri_amd", "csrc"), "san"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       timeout=900)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0, out[-3000:]
    assert "host sanitizer run:" in out and "clean" in out
    assert "AddressSanitizer" not in out and "runtime error:" not in out and "LeakSanitizer" not in out, out[-3000:]
    # the run really was instrumented
    nm = subprocess.run(["nm", os.path.join(ROOT, "tools", "san", "host_san_driver")], stdout=subprocess.PIPE).stdout.decode()
    assert "__asan_report" in nm and "__ubsan_handle" in nm
