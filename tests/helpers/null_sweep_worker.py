"""Child process of tests/test_gpu_abi_robustness.py and tests/test_abi.py: calls every status-returning entry of the C ABI with NULL
pointers and degenerate integers.  A library that dereferences one of them dies here, in the child, and the parent reports which call.

    python null_sweep_worker.py {null-ctx | live-ctx | live-structs | live-blocks}

Prints one line per call: `<entry> <integer fill> <status>`; the last line is `done <n calls>`."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from scri_amd import _lib  # noqa: E402

SKIP = {"bms_version", "bms_ctx_create", "bms_ctx_destroy", "bms_last_error", "bms_host_alloc", "bms_host_free"}


def main(mode):
    lib = _lib.load()
    handle = keep = None
    if mode == "live-ctx":
        keep = _lib.Context(0)  # (the object owns the bms_ctx: it must outlive the calls)
        handle = keep.handle
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


def blocks(_mode):
    """The building blocks and series operators with real buffers and ONE wrong scalar each."""
    import numpy as np

    lib = _lib.load()
    keep = _lib.Context(0)  # (the object owns the bms_ctx: it must outlive the calls)
    h = keep.handle
    dp = ctypes.POINTER(ctypes.c_double)
    vp = lambda a: ctypes.c_void_p(a.ctypes.data)  # noqa: E731
    dpp = lambda a: a.ctypes.data_as(dp)  # noqa: E731
    rng = np.random.default_rng(9)
    n, L = 40, 4
    nm = (L + 1) ** 2
    nt = nph = 2 * L + 1
    modes = rng.normal(size=(n, nm)) + 1j * rng.normal(size=(n, nm))
    grid = np.zeros((n, nt * nph), dtype=complex)
    modes2 = modes.copy()  # (a result array that outlives the call)
    t = np.linspace(0.0, 4.0, n)
    t_bad = t.copy()
    t_bad[7] = t_bad[6]
    x_new = np.linspace(0.5, 3.5, 11)
    out_s = np.zeros((11, nm), dtype=complex)
    q = (ctypes.c_double * 4)(1.0, 0.0, 0.0, 0.0)
    spinors = np.tile(np.array([1.0, 0.0, 0.0, 0.0]), (n, 1))
    D = np.zeros(sum((2 * l + 1) ** 2 for l in range(L + 1)), dtype=complex)
    rot = np.tile(np.array([1.0, 0.0, 0.0, 0.0]), (5, 1))
    ev_out = np.zeros((n, 5), dtype=complex)
    widths = (ctypes.c_int * 2)(8, 8)
    words = np.zeros(64, dtype=np.uint16)
    words_out = np.zeros(64, dtype=np.uint16)
    chk = ctypes.c_uint32(0)
    norm_out = np.zeros(n)
    H, bad = _lib.BMS_HOST, 7
    cases = [
        ("bms_salm2map well-formed", lambda: lib.bms_salm2map(h, vp(modes), H, n, 0, L, nt, nph, vp(grid)), 0),
        ("bms_salm2map mem", lambda: lib.bms_salm2map(h, vp(modes), bad, n, 0, L, nt, nph, vp(grid)), -1),
        ("bms_salm2map n_theta=1", lambda: lib.bms_salm2map(h, vp(modes), H, n, 0, L, 1, nph, vp(grid)), -1),
        ("bms_salm2map spin=9", lambda: lib.bms_salm2map(h, vp(modes), H, n, 9, L, nt, nph, vp(grid)), -1),
        ("bms_salm2map ell_max=-1", lambda: lib.bms_salm2map(h, vp(modes), H, n, 0, -1, nt, nph, vp(grid)), -1),
        ("bms_salm2map ell_max=60000", lambda: lib.bms_salm2map(h, vp(modes), H, n, 0, 60000, nt, nph, vp(grid)), -1),
        ("bms_map2salm well-formed", lambda: lib.bms_map2salm(h, vp(grid), H, n, nt, nph, 0, 0, L, vp(modes2)), 0),
        ("bms_map2salm mem", lambda: lib.bms_map2salm(h, vp(grid), bad, n, nt, nph, 0, 0, L, vp(modes2)), -1),
        ("bms_map2salm ell_min>ell_max", lambda: lib.bms_map2salm(h, vp(grid), H, n, nt, nph, 0, L + 1, L, vp(modes2)), -1),
        ("bms_map2salm n_phi=0", lambda: lib.bms_map2salm(h, vp(grid), H, n, nt, 0, 0, 0, L, vp(modes2)), -1),
        ("bms_cubic_spline well-formed", lambda: lib.bms_cubic_spline(h, dpp(t), n, vp(modes), nm, nm, H, dpp(x_new), 11, vp(out_s)), 0),
        ("bms_cubic_spline mem", lambda: lib.bms_cubic_spline(h, dpp(t), n, vp(modes), nm, nm, bad, dpp(x_new), 11, vp(out_s)), -1),
        ("bms_cubic_spline knots", lambda: lib.bms_cubic_spline(h, dpp(t_bad), n, vp(modes), nm, nm, H, dpp(x_new), 11, vp(out_s)), -1),
        ("bms_cubic_spline n=3", lambda: lib.bms_cubic_spline(h, dpp(t), 3, vp(modes), nm, nm, H, dpp(x_new), 11, vp(out_s)), -1),
        ("bms_cubic_spline ld<cols", lambda: lib.bms_cubic_spline(h, dpp(t), n, vp(modes), nm - 1, nm, H, dpp(x_new), 11, vp(out_s)), -1),
        ("bms_spline_derivative order=9", lambda: lib.bms_spline_derivative(h, dpp(t), n, vp(modes), nm, nm, H, dpp(x_new), 11, 9, vp(out_s)), -1),
        ("bms_spline_derivative mem", lambda: lib.bms_spline_derivative(h, dpp(t), n, vp(modes), nm, nm, bad, dpp(x_new), 11, 1, vp(out_s)), -1),
        ("bms_rotate_const well-formed", lambda: lib.bms_rotate_const(h, vp(modes), H, n, nm, 0, L, q), 0),
        ("bms_rotate_const mem", lambda: lib.bms_rotate_const(h, vp(modes), bad, n, nm, 0, L, q), -1),
        ("bms_rotate_const ld", lambda: lib.bms_rotate_const(h, vp(modes), H, n, nm - 1, 0, L, q), -1),
        ("bms_rotate_const ell", lambda: lib.bms_rotate_const(h, vp(modes), H, n, nm, 3, 2, q), -1),
        ("bms_rotate_const ell_max=60000", lambda: lib.bms_rotate_const(h, vp(modes), H, n, 1 << 40, 0, 60000, q), -1),
        ("bms_rotate_series mem", lambda: lib.bms_rotate_series(h, vp(modes), bad, n, nm, 0, L, vp(spinors)), -1),
        ("bms_rotate_series n=-1", lambda: lib.bms_rotate_series(h, vp(modes), H, -1, nm, 0, L, vp(spinors)), -1),
        ("bms_rotate_const_D mem", lambda: lib.bms_rotate_const_D(h, vp(modes), bad, n, nm, 0, L, vp(D)), -1),
        ("bms_wigner_D ell", lambda: lib.bms_wigner_D(h, q, 2, 1, vp(D)), -1),
        ("bms_wigner_D ell_max=60000", lambda: lib.bms_wigner_D(h, q, 0, 60000, vp(D)), -1),
        ("bms_evaluate_modes well-formed", lambda: lib.bms_evaluate_modes(h, vp(modes), H, n, nm, 0, 0, L, dpp(rot), 5, vp(ev_out)), 0),
        ("bms_evaluate_modes mem", lambda: lib.bms_evaluate_modes(h, vp(modes), bad, n, nm, 0, 0, L, dpp(rot), 5, vp(ev_out)), -1),
        ("bms_evaluate_modes ld", lambda: lib.bms_evaluate_modes(h, vp(modes), H, n, nm - 1, 0, 0, L, dpp(rot), 5, vp(ev_out)), -1),
        ("bms_evaluate_modes n_rot=-1", lambda: lib.bms_evaluate_modes(h, vp(modes), H, n, nm, 0, 0, L, dpp(rot), -1, vp(ev_out)), -1),
        ("bms_row_norm well-formed", lambda: lib.bms_row_norm(h, vp(modes), nm, n, nm, H, 0, dpp(norm_out)), 0),
        ("bms_row_norm mem", lambda: lib.bms_row_norm(h, vp(modes), nm, n, nm, bad, 0, dpp(norm_out)), -1),
        ("bms_row_norm ld", lambda: lib.bms_row_norm(h, vp(modes), nm - 1, n, nm, H, 0, dpp(norm_out)), -1),
        ("bms_grid_multiply mem", lambda: lib.bms_grid_multiply(h, vp(modes), 0, L, vp(modes), 0, L, bad, n, 2 * L, L, vp(modes2)), -1),
        ("bms_grid_multiply spins", lambda: lib.bms_grid_multiply(h, vp(modes), 4, L, vp(modes), 4, L, H, n, 2 * L, L, vp(modes2)), -1),
        ("bms_grid_multiply l", lambda: lib.bms_grid_multiply(h, vp(modes), 0, L, vp(modes), 0, L, H, n, 2, 3, vp(modes2)), -1),
        ("bms_xor_timeseries mem", lambda: lib.bms_xor_timeseries(h, vp(modes), bad, n, 2 * nm, 0), -1),
        ("bms_xor_timeseries words=-1", lambda: lib.bms_xor_timeseries(h, vp(modes), H, n, -1, 0), -1),
        ("bms_multishuffle widths", lambda: lib.bms_multishuffle(h, vp(words), vp(words_out), H, 64, (ctypes.c_int * 2)(8, 9), 2, 1), -1),
        ("bms_multishuffle mem", lambda: lib.bms_multishuffle(h, vp(words), vp(words_out), bad, 64, widths, 2, 1), -1),
        ("bms_multishuffle well-formed", lambda: lib.bms_multishuffle(h, vp(words), vp(words_out), H, 64, widths, 2, 1), 0),
        ("bms_fletcher32 odd", lambda: lib.bms_fletcher32(h, vp(words), H, 127, ctypes.byref(chk)), -1),
        ("bms_fletcher32 mem", lambda: lib.bms_fletcher32(h, vp(words), bad, 128, ctypes.byref(chk)), -1),
        ("bms_fletcher32 well-formed", lambda: lib.bms_fletcher32(h, vp(words), H, 128, ctypes.byref(chk)), 0),
        ("bms_ctx_set_option unknown", lambda: lib.bms_ctx_set_option(h, b"NO_SUCH_ROUTE", 1), -1),
        ("bms_ctx_set_option GEMM_EVAL_STEP=5", lambda: lib.bms_ctx_set_option(h, b"GEMM_EVAL_STEP", 5), -1),
    ]
    for label, call, want in cases:
        print(label.replace(" ", ":"), end=" ", flush=True)
        rc = call()
        print("want0" if want == 0 else "wantneg", rc, flush=True)
    print("done", len(cases), flush=True)


if __name__ == "__main__":
    {"live-structs": structs, "live-blocks": blocks}.get(sys.argv[1], main)(sys.argv[1])
