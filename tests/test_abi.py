"""The C-ABI shared library loads and exports every symbol include/scri_amd.h declares (no GPU needed);
without a GPU the product path fails loudly instead of falling back to anything."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "scri_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bms_[a-z_A-Z0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from scri_amd import _lib

    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared_functions()
    assert len(names) >= 18
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/scri_amd.h but not exported"
    # and the Python binding knows each of them
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)
    assert _lib.load().bms_version() >= 1


def test_struct_layouts_match_header_sizes():
    from scri_amd import _lib

    # sizes implied by the header on LP64: guards against drift between the header and the ctypes mirror
    assert ctypes.sizeof(_lib.bms_shard) == 32 + 8
    assert ctypes.sizeof(_lib.bms_transformation) == 8 + 8 + 32 + 24 + 16
    assert ctypes.sizeof(_lib.bms_wm_input) == 8 + 8 + 8 + 8 + 4 * 8 + 32 + 32 + 16 + 16 + 16 + 32 + 16


# The route options of a context (scri_amd/csrc/env.h): A/B switches between routes with the same results.  A context reads them from
# the environment ONCE, in bms_ctx_create; afterwards only bms_ctx_set_option / bms_ctx_get_option touch them.
ROUTE_OPTIONS = {
    "AXIS_BOOST_MIN_WORK", "GEMM_EVAL_STEP", "GRID_MULTIPLY_FULL_GRID", "NO_AXIS_BOOST_SEPARABLE", "NO_BSPLINE", "NO_COLUMN_SORT",
    "NO_FUSED_ABD_MIX", "NO_FUSED_ANALYSIS", "NO_GEMM_EVAL", "NO_LARGE_ANALYSIS", "NO_LARGE_SYNTHESIS", "NO_PLAN_CACHE",
    "NO_SEPARABLE_SYNTHESIS", "NO_SMALL_DENSE", "NO_SPLIT_ANALYSIS", "NO_SPLIT_SYNTHESIS", "ROTATE_STAGED", "ROTATE_VALU", "TRACE",
    "TWO_SWEEPS", "WALK_FIRST", "SYNTHESIS_EVAL", "NO_SYNTHESIS_EVAL", "NO_ABD_SIGMA_EVAL", "NO_ROTATE_PIPELINE",
}


def test_route_options_are_per_context_and_the_environment_is_read_in_one_place():
    """no `getenv` on any call path.  The sources reach the environment in exactly two places, both in env.h:
    RouteOptions::read_environment (called by bms_ctx_create and nothing else) and the probe macro, which the default build compiles to
    a null pointer.  Knock-outs, host-blocking traces, disabled guards and unvalidated tuning knobs exist only with -DSCRI_AMD_PROBES
    (make PROBES=1 -> libscri_amd_probes.so): the default library does not even contain their names."""
    from scri_amd import _lib

    csrc = os.path.join(ROOT, "scri_amd", "csrc")
    env_h = open(os.path.join(csrc, "env.h")).read()
    assert set(re.findall(r'X\((\w+), "(\w+)"\)', env_h)) == {(n, n) for n in ROUTE_OPTIONS}
    assert len(re.findall(r"(?<![A-Za-z_])getenv\(", env_h)) == 2  # read_environment + the probe macro
    readers = 0
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h")) and f != "env.h":
            text = open(os.path.join(csrc, f)).read()
            assert not re.search(r"(?<![A-Za-z_:])getenv\(", text) and "std::getenv" not in text, f"{f} calls getenv directly"
            assert "route_env(" not in text, f"{f} still reads a route switch from the environment per call"
            readers += len(re.findall(r"read_environment\(\)", text))
            for m in re.finditer(r"read_environment\(\)", text):
                assert "bms_ctx_create" in text[max(0, m.start() - 3000) : m.start()], f"{f}: read_environment outside bms_ctx_create"
    assert readers == 1
    blob = open(os.path.join(ROOT, "scri_amd", "libscri_amd.so"), "rb").read()
    for probe in (b"GEMM_EVAL_DBG", b"GEMM_EVAL_TRACE", b"ASSUME_REGULAR_MESH", b"BS_XP", b"SE_KNOCK", b"ZGEMM_ST_ROWS_LOG2", b"DOWN_CUS"):
        assert probe not in blob, probe
    for name in ROUTE_OPTIONS:  # the options are known by name to bms_ctx_set_option
        assert name.encode() + b"\x00" in blob, name
    assert _lib.LIB_PATH.endswith("libscri_amd.so") or os.environ.get("SCRI_AMD_LIB_PATH")


def _have_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_have_gpu(), reason="only meaningful on a machine without a GPU")
def test_no_cpu_fallback():
    import scri_amd

    with pytest.raises(scri_amd.BMSError, match="no CPU fallback"):
        scri_amd.Context(0)
    w = scri_amd.WaveformModes(t=np.linspace(0, 1, 10), data=np.zeros((10, 21), dtype=complex), ell_min=2, ell_max=4,
                               dataType=scri_amd.h, frameType=scri_amd.Inertial)
    with pytest.raises(scri_amd.BMSError):
        w.transform(time_translation=0.1)
    with pytest.raises(scri_amd.BMSError):
        w.rotate_decomposition_basis([1.0, 0, 0, 0])


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "scri_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f"{f} imports the oracle"
                assert "/root/reference" not in text


def test_one_hip_runtime_whichever_is_loaded_first():
    """torch bundles its own libamdhip64; the library must end up on the same copy even when it is loaded before torch
    (two runtimes in one process: the second to initialise finds no GPU)."""
    import subprocess
    import sys

    code = (
        "import os, sys; sys.path.insert(0, %r)\n"
        "from scri_amd import _lib; _lib.load()\n"
        "import torch\n"
        "paths = {l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l}\n"
        "print(len(paths))\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip().splitlines()[-1] == "1"


def test_no_cpp_exception_crosses_the_boundary():
    """The callers are C / ctypes: a std::bad_alloc that unwound into them would end the process.  Every entry that returns a status is
    a function-try-block (engine.h, BMS_CATCH); here a host-only entry is asked for a 1 GiB table under an address-space limit that
    cannot hold it, in a child process: it must come back with BMS_ERR_NOMEM and a message, not abort."""
    import subprocess
    import sys

    from scri_amd import _lib

    code = (
        "import ctypes, resource, sys; sys.path.insert(0, %r)\n"
        "from scri_amd import _lib; lib = _lib.load()\n"
        "vm = int([l for l in open('/proc/self/status') if l.startswith('VmSize')][0].split()[1]) * 1024\n"
        "resource.setrlimit(resource.RLIMIT_AS, (vm + (256 << 20), resource.getrlimit(resource.RLIMIT_AS)[1]))\n"
        "fr = (ctypes.c_double * 4)(1, 0, 0, 0); v = (ctypes.c_double * 3)(0, 0, 0.1); out = (ctypes.c_double * 4)()\n"
        "lib.bms_ring_colatitudes.restype = ctypes.c_int\n"
        "print(lib.bms_ring_colatitudes(fr, v, 1 << 27, 4, out), lib.bms_last_error(None).decode())\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.returncode, out.stderr[-2000:])
    status, message = out.stdout.strip().splitlines()[-1].split(" ", 1)
    assert int(status) == _lib.BMS_ERR_NOMEM and "bad_alloc" in message
    # ... and every status-returning definition in the engine carries the guard
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scri_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.startswith("engine_") and f.endswith(".hip"):
            text = open(os.path.join(csrc, f)).read()
            for m in re.finditer(r'^extern "C" int (bms_\w+)\(([^{;]*)\)\s*(try\s*)?\{', text, flags=re.M):
                assert m.group(3) or m.group(1) == "bms_version", f"{f}: {m.group(1)} has no function-try-block"


def test_probe_build_carries_the_probe_switches_and_the_same_abi():
    """`make PROBES=1` -> libscri_amd_probes.so: the variant the scripts under tools/probes load through SCRI_AMD_LIB_PATH.  It must keep
    building, export what the header declares, and -- unlike the default library -- contain the knock-out / trace switches."""
    import shutil
    import subprocess

    if not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")):
        pytest.skip("no hipcc")
    csrc = os.path.join(ROOT, "scri_amd", "csrc")
    subprocess.check_call(["make", "-C", csrc, "PROBES=1", "-j8"], stdout=subprocess.DEVNULL)
    path = os.path.join(ROOT, "scri_amd", "libscri_amd_probes.so")
    blob = open(path, "rb").read()
    for name in (b"SCRI_AMD_GEMM_EVAL_DBG", b"SCRI_AMD_GEMM_EVAL_TRACE", b"SCRI_AMD_SE_KNOCK", b"SCRI_AMD_ASSUME_REGULAR_MESH"):
        assert name in blob, name
    lib = ctypes.CDLL(path)
    for name in _declared_functions():
        assert hasattr(lib, name), name
