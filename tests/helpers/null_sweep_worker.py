"""Child process of tests/test_gpu_abi_robustness.py and tests/test_abi.py: calls every status-returning entry of the C ABI with NULL
pointers and degenerate integers.  A library that dereferences one of them dies here, in the child, and the parent reports which call.

    python null_sweep_worker.py {null-ctx | live-ctx | live-structs}

Prints one line per call: `<entry> <integer fill> <status>`; the last line is `done <n calls>`."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from scri_amd import _lib  # noqa: E402

SKIP = {"bms_version", "bms_ctx_create", "bms_ctx_destroy", "bms_last_error", "bms_host_alloc", "bms_host_free"}


def main(mode):
    lib = _lib.load()
    handle = None
    if mode == "live-ctx":
        handle = _lib.Context(0).handle
    n = 0
    for name, (restype, argtypes) in sorted(_lib.SIGNATURES.items()):
        if name in SKIP or restype is not ctypes.c_int:
            continue
        for fill in (0, -1, 7):
            args = []
            for k, t in enumerate(argtypes):
                if k == 0 and t is ctypes.c_void_p and name not in ("bms_host_register", "bms_host_unregister"):
                    args.append(handle)
                elif k == 0 and t == ctypes.POINTER(ctypes.c_void_p):  # the *_multi entries: an array of contexts
                    args.append((ctypes.c_void_p * 1)(handle) if handle else None)
                elif t in (ctypes.c_int, ctypes.c_int64, ctypes.c_uint64):
                    args.append(1 if (k == 1 and "multi" in name) else (fill if t is not ctypes.c_uint64 else max(fill, 0)))
                elif t is ctypes.c_double:
                    args.append(float(fill))
                else:
                    args.append(None)
            print(name, fill, end=" ", flush=True)
            rc = getattr(lib, name)(*args)
            print(rc, flush=True)
            n += 1
    print("done", n, flush=True)


def structs(_mode):
    """Well-formed calls with ONE field of the input description made wrong at a time: every one must come back with a negative
    status (and the library must still give the first answer afterwards)."""
    import copy

    import numpy as np

    from scri_amd import engine

    lib = _lib.load()
    ctx = _lib.Context(0)
    h = ctx.handle
    n, lmin, lmax, s = 96, 2, 4, -2
    nm = (lmax + 1) ** 2 - lmin**2
    rng = np.random.default_rng(5)
    t = np.linspace(0.0, 12.0, n)
    data = rng.normal(size=(n, nm)) + 1j * rng.normal(size=(n, nm))
    aux = rng.normal(size=(n, nm)) + 1j * rng.normal(size=(n, nm))
    st = np.zeros(4, dtype=complex)
    st[0] = 0.3
    dp = ctypes.POINTER(ctypes.c_double)

    def wm():
        w = _lib.bms_wm_input()
        w.n_times, w.t, w.data, w.ld, w.mem = n, t.ctypes.data_as(dp), data.ctypes.data, nm, _lib.BMS_HOST
        w.ell_min, w.ell_max, w.spin_weight, w.conformal_weight, w.type_term, w.n_aux = lmin, lmax, s, -1, _lib.BMS_TERM_H, 0
        return w

    def tr():
        return engine.make_transformation(st, (1.0, 0.0, 0.0, 0.0), (0.0, 0.0, 0.1), 2 * lmax + 1, 2 * lmax + 1, lmax)

    t_out = np.zeros(n)
    out = np.zeros((n, 2 * (2 * lmax + 1) ** 2), dtype=complex)  # (wide enough for the grid flavour)
    got = ctypes.c_int64(0)
    first = ctypes.c_int64(0)
    args_out = (t_out.ctypes.data_as(dp), out.ctypes.data, ctypes.byref(got))

    def shard(**kw):
        sh = _lib.bms_shard()
        sh.data_row0, sh.data_rows, sh.out_i0, sh.out_i1, sh.col_part, sh.col_parts = 0, n, 0, n, 0, 1
        for k, v in kw.items():
            setattr(sh, k, v)
        return sh

    entries = {
        "bms_transform_modes": lambda w, T: lib.bms_transform_modes(h, ctypes.byref(w), ctypes.byref(T), *args_out),
        "bms_modes_to_grid": lambda w, T: lib.bms_modes_to_grid(h, ctypes.byref(w), ctypes.byref(T), *args_out),
        "bms_transform_modes_pipelined": lambda w, T: lib.bms_transform_modes_pipelined(h, ctypes.byref(w), ctypes.byref(T), 3, *args_out),
        "bms_transform_modes_series": lambda w, T: lib.bms_transform_modes_series(
            h, ctypes.byref(w), 1, ctypes.byref(T), args_out[0], args_out[1], None, args_out[2]),
        "bms_transform_modes_shard": lambda w, T: lib.bms_transform_modes_shard(
            h, ctypes.byref(w), ctypes.byref(T), ctypes.byref(shard()), args_out[0], args_out[1], args_out[2], ctypes.byref(first)),
    }
    nan = float("nan")
    wm_wrong = [("t", None), ("data", None), ("mem", 7), ("n_times", -1), ("n_times", 3), ("ld", nm - 1), ("ell_min", -1),
                ("ell_max", lmin - 1), ("n_aux", 5), ("n_aux", -1), ("n_aux", 1), ("type_term", _lib.BMS_TERM_PSI), ("ell_max", 60000)]
    tr_wrong = [("supertranslation", None), ("ell_max_supertranslation", 0), ("frame_rotation", (0.0, 0.0, 0.0, 0.0)),
                ("frame_rotation", (nan, 0.0, 0.0, 0.0)), ("boost_velocity", (0.0, 0.0, 1.0)), ("boost_velocity", (nan, 0.0, 0.0)),
                ("n_theta", 1), ("n_phi", 0), ("ell_max_out", 1), ("n_theta", 40000), ("ell_max_out", 60000)]
    count = 0
    for name, call in entries.items():
        out[:] = 0
        rc = call(wm(), tr())
        print(name, "well-formed", rc, flush=True)
        baseline = out.copy()
        for field, value in wm_wrong:
            w = wm()
            if field == "n_aux" and value == 1:
                w.n_aux = 1  # ... whose array pointer stays NULL
            else:
                setattr(w, field, value)
            print(name, f"wm.{field}={value}", end=" ", flush=True)
            print(call(w, tr()), flush=True)
            count += 1
        for field, value in tr_wrong:
            T = tr()
            if field == "n_theta" and value == 40000:
                T.n_theta = T.n_phi = 40000
            elif isinstance(value, tuple):
                getattr(T, field)[:] = value
            else:
                setattr(T, field, value)
            print(name, f"tr.{field}={value}", end=" ", flush=True)
            print(call(wm(), T), flush=True)
            count += 1
        out[:] = 0
        rc = call(wm(), tr())
        print(name, "well-formed-again", rc, int(np.array_equal(out, baseline)), flush=True)
    # shards that do not describe rows of the series
    for kw in (dict(data_row0=-1), dict(data_rows=n + 1), dict(out_i0=9, out_i1=3), dict(col_part=3, col_parts=2), dict(col_part=-1, col_parts=2)):
        w, T, sh = wm(), tr(), shard(**kw)
        print("bms_transform_modes_shard", f"shard{kw}".replace(" ", ""), end=" ", flush=True)
        rc = lib.bms_transform_modes_shard(h, ctypes.byref(w), ctypes.byref(T), ctypes.byref(sh), args_out[0], args_out[1], args_out[2], ctypes.byref(first))
        print(rc, flush=True)
        count += 1
    # the six-field flavour
    L = 3
    raw = rng.normal(size=(6, n, (L + 1) ** 2)) + 1j * rng.normal(size=(6, n, (L + 1) ** 2))
    raw_out = np.zeros_like(raw)

    def abd(u=t.ctypes.data_as(dp), r=raw.ctypes.data, mem=_lib.BMS_HOST, nn=n, ell=L, T=None):
        T = T or engine.make_transformation(st, (1.0, 0.0, 0.0, 0.0), (0.0, 0.0, 0.1), 2 * L + 1, 2 * L + 1, L)
        return lib.bms_transform_abd(h, u, r, mem, nn, ell, ctypes.byref(T), args_out[0], raw_out.ctypes.data, args_out[2])

    print("bms_transform_abd", "well-formed", abd(), flush=True)
    for label, kw in (("u=None", dict(u=None)), ("raw=None", dict(r=None)), ("mem=7", dict(mem=7)), ("n=1", dict(nn=1)), ("n=-5", dict(nn=-5)),
                      ("ell_max=-1", dict(ell=-1))):
        print("bms_transform_abd", label, end=" ", flush=True)
        print(abd(**kw), flush=True)
        count += 1
    for field, value in tr_wrong:
        if field == "ell_max_out" and value == 1:
            value = -1
        T = engine.make_transformation(st, (1.0, 0.0, 0.0, 0.0), (0.0, 0.0, 0.1), 2 * L + 1, 2 * L + 1, L)
        if field == "n_theta" and value == 40000:
            T.n_theta = T.n_phi = 40000
        elif isinstance(value, tuple):
            getattr(T, field)[:] = value
        else:
            setattr(T, field, value)
        print("bms_transform_abd", f"tr.{field}={value}", end=" ", flush=True)
        print(abd(T=T), flush=True)
        count += 1
    print("bms_transform_abd", "well-formed-again", abd(), 1, flush=True)
    print("done", count, flush=True)


if __name__ == "__main__":
    (structs if sys.argv[1] == "live-structs" else main)(sys.argv[1])
