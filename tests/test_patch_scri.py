"""The graft of SURVEY 8(b) / INTEGRATION.md section 3 (`scri_amd.patch_scri`), exercised on a minimal stand-in for the
`scri` package: a module with a `rotations` submodule (dispatcher + the two kernels looked up at call time, as
scri/rotations.py:284-392 does), a `WaveformModes` and an `AsymptoticBondiData` class carrying the attributes the adapters
read.  CPU: the mechanics (what is replaced, `_reference` aliases, uninstall).  GPU: the patched entry points against the
oracle."""
import types

import numpy as np
import pytest

from oracle import abd_ref, quat, rotations_ref, wigner
from oracle import waveform_grid_ref as grid_ref
from oracle.containers import ABD, WM


def make_stub_scri():
    scri = types.ModuleType("scri")
    rot = types.ModuleType("scri.rotations")
    scri.rotations = rot
    scri.Inertial, scri.h = 1, 7

    class WaveformModes:
        def __init__(self, t, data, ell_min, ell_max, frameType=1, dataType=7, r_is_scaled_out=True, m_is_scaled_out=True,
                     history=None, constructor_statement=None, frame=None):
            self.t, self.data, self.ell_min, self.ell_max = np.asarray(t, dtype=float), np.asarray(data), ell_min, ell_max
            self.frameType, self.dataType = frameType, dataType
            self.r_is_scaled_out, self.m_is_scaled_out = r_is_scaled_out, m_is_scaled_out
            self.history = list(history or []) + ([constructor_statement] if constructor_statement else [])
            self.frame = np.zeros((0, 4)) if frame is None else frame

        @property
        def n_times(self):
            return self.t.size

        def transform(self, **kwargs):
            raise RuntimeError("the reference's CPU transform was called")

        def __repr__(self):
            return "w"

    class AsymptoticBondiData:
        def __init__(self, time, ell_max):
            self.u = np.array(time, dtype=float)
            self.ell_max = ell_max
            self._raw_data = np.zeros((6, self.u.size, (ell_max + 1) ** 2), dtype=complex)

        def transform(self, **kwargs):
            raise RuntimeError("the reference's CPU transform was called")

    for i, name in enumerate(("psi0", "psi1", "psi2", "psi3", "psi4", "sigma")):
        def _get(self, i=i):
            return self._raw_data[i]

        def _set(self, value, i=i):
            self._raw_data[i][:] = value

        setattr(AsymptoticBondiData, name, property(_get, _set))

    # the two kernels: CPU loops (here: the oracle's restatement), with the reference's signatures
    def _rotate_decomposition_basis_by_constant(data, ell_min, ell_max, D, tmp):
        data[:] = rotations_ref.rotate_by_constant(data, ell_min, ell_max, D)

    def _rotate_decomposition_basis_by_series(data, R_basis, ell_min, ell_max, D):
        data[:] = rotations_ref.rotate_by_series(data, R_basis, ell_min, ell_max)

    rot._rotate_decomposition_basis_by_constant = _rotate_decomposition_basis_by_constant
    rot._rotate_decomposition_basis_by_series = _rotate_decomposition_basis_by_series

    def rotate_decomposition_basis(W, R_basis):
        """Dispatch as scri/rotations.py:284-343 does: precompute D for a constant rotor, hand spinors for a series, and
        look the kernels up in the module at call time."""
        R = np.asarray(R_basis, dtype=float)
        D = np.empty(wigner.total_size_D_matrices(W.ell_min, W.ell_max), dtype=complex)
        if R.ndim == 2:
            rot._rotate_decomposition_basis_by_series(W.data, quat.as_spinor_array(R), W.ell_min, W.ell_max, D)
        else:
            Ra, Rb = quat.as_spinor_array(R)
            D[:] = wigner.wigner_D_matrices(Ra, Rb, W.ell_min, W.ell_max)
            rot._rotate_decomposition_basis_by_constant(W.data, W.ell_min, W.ell_max, D, np.empty(2 * W.ell_max + 1, dtype=complex))
        return W

    rot.rotate_decomposition_basis = rotate_decomposition_basis
    WaveformModes.rotate_decomposition_basis = rotate_decomposition_basis
    scri.WaveformModes, scri.AsymptoticBondiData = WaveformModes, AsymptoticBondiData
    return scri


def test_install_and_uninstall_mechanics():
    import scri_amd
    from scri_amd import adapters

    scri = make_stub_scri()
    ref = dict(c=scri.rotations._rotate_decomposition_basis_by_constant, s=scri.rotations._rotate_decomposition_basis_by_series,
               wt=scri.WaveformModes.transform, at=scri.AsymptoticBondiData.transform)
    patched = scri_amd.patch_scri(scri)
    for name in ("rotations._rotate_decomposition_basis_by_constant", "rotations._rotate_decomposition_basis_by_series",
                 "WaveformModes.rotate_decomposition_basis", "WaveformModes.transform", "AsymptoticBondiData.transform"):
        assert name in patched
    assert scri.rotations._rotate_decomposition_basis_by_constant is not ref["c"]
    assert scri.rotations._rotate_decomposition_basis_by_constant_reference is ref["c"]
    assert scri.rotations._rotate_decomposition_basis_by_series_reference is ref["s"]
    assert scri.WaveformModes.transform_reference is ref["wt"] and scri.WaveformModes.transform is not ref["wt"]
    assert scri.AsymptoticBondiData.transform_reference is ref["at"] and scri.AsymptoticBondiData.transform is not ref["at"]
    scri_amd.patch_scri(scri)  # idempotent: the aliases keep pointing at the reference
    assert scri.rotations._rotate_decomposition_basis_by_constant_reference is ref["c"]
    assert scri.WaveformModes.transform_reference is ref["wt"]
    adapters.uninstall(scri)
    assert scri.rotations._rotate_decomposition_basis_by_constant is ref["c"]
    assert scri.rotations._rotate_decomposition_basis_by_series is ref["s"]
    assert scri.WaveformModes.transform is ref["wt"] and scri.AsymptoticBondiData.transform is ref["at"]
    assert not hasattr(scri.WaveformModes, "transform_reference")


@pytest.mark.gpu
def test_patched_entry_points_match_oracle(ctx):
    import scri_amd
    from scri_amd import adapters, synthetic

    scri = make_stub_scri()
    scri_amd.patch_scri(scri, ctx=ctx)
    try:
        # rotations: the dispatcher is the stub's own, the kernels are the GPU's
        t, data, rot = synthetic.cfg1()
        w = scri.WaveformModes(t, data.copy(), 2, 4)
        w.rotate_decomposition_basis(rot["constant"])
        Ra, Rb = quat.as_spinor_array(rot["constant"])
        expect = rotations_ref.rotate_by_constant(data, 2, 4, wigner.wigner_D_matrices(Ra, Rb, 2, 4))
        assert np.abs(w.data - expect).max() < 4e-13
        w.rotate_decomposition_basis(rot["series"])
        expect = rotations_ref.rotate_by_series(expect, quat.as_spinor_array(rot["series"]), 2, 4)
        assert np.abs(w.data - expect).max() < 4e-13
        view = np.zeros((2000, 30), dtype=complex)  # a sliced view (row stride > n_modes) is rotated in place
        view[:, 3:24] = data
        w = scri.WaveformModes(t, view[:, 3:24], 2, 4)
        w.rotate_decomposition_basis(rot["series"])
        assert np.abs(view[:, 3:24] - rotations_ref.rotate_by_series(data, quat.as_spinor_array(rot["series"]), 2, 4)).max() < 4e-13
        assert np.all(view[:, :3] == 0) and np.all(view[:, 24:] == 0)
        # WaveformModes.transform
        t, data, spec = synthetic.workload("cfg3", n_times=300)
        data = data[:, : 7 * 7 - 4]
        out = scri.WaveformModes(t, data, 2, 6).transform(**spec["kwargs"])
        assert isinstance(out, scri.WaveformModes)
        e = grid_ref.transform(WM(t=t, data=data, ell_min=2, ell_max=6, dataType=7), **spec["kwargs"])
        assert out.t.shape == e.t.shape and np.abs(out.data - e.data).max() < 1e-12
        # AsymptoticBondiData.transform
        u, raw, spec = synthetic.abd_workload("cfg5", n_times=200, ell_max=3)
        a = scri.AsymptoticBondiData(u, 3)
        a.psi0, a.psi1, a.psi2, a.psi3, a.psi4, a.sigma = raw
        out = a.transform(**spec["kwargs"])
        assert isinstance(out, scri.AsymptoticBondiData) and out.ell_max == 3
        e = abd_ref.transform(ABD(u, raw, 3), **spec["kwargs"])
        assert out.u.shape == e.u.shape and np.abs(out._raw_data - e.raw).max() < 1e-12 * max(1.0, np.abs(e.raw).max())
    finally:
        adapters.uninstall(scri)
    with pytest.raises(RuntimeError, match="reference's CPU transform"):
        scri.WaveformModes(t, data, 2, 6).transform()


def test_install_on_the_reference_package_itself():
    """Build container only (skipped where /root/reference is absent): the graft finds its three seams on the real `scri`
    package (imported on the stand-ins of tests/golden/reference_standins.py, in a subprocess) and the patched rotation
    dispatcher reaches the GPU stub with the reference's own arguments (recorded, not executed: no GPU here)."""
    import os
    import subprocess
    import sys

    if not os.path.isdir("/root/reference/scri"):
        pytest.skip("reference checkout not present")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import reference_standins as standins
scri = standins.install()
import scri_amd
from scri_amd import engine
calls = []
engine.rotate_const_D = lambda d, lmin, lmax, D, ctx=None: calls.append(("const_D", d.shape, lmin, lmax, np.asarray(D).shape))
engine.rotate_series = lambda d, lmin, lmax, sp, ctx=None: calls.append(("series", d.shape, lmin, lmax, np.asarray(sp).shape))
patched = scri_amd.patch_scri(scri)
assert "AsymptoticBondiData.transform" in patched and "WaveformModes.transform" in patched
assert scri.rotations._rotate_decomposition_basis_by_constant_reference.__module__ == "scri.rotations"
t = np.linspace(0.0, 1.0, 9)
w = scri.WaveformModes(t=t, data=np.ones((9, 21), dtype=complex), ell_min=2, ell_max=4, frameType=scri.Inertial, dataType=scri.h,
                       r_is_scaled_out=True, m_is_scaled_out=True)
w.rotate_decomposition_basis(np.quaternion(0.5, 0.5, 0.5, 0.5))
w.rotate_physical_system(standins.as_quat_array(np.tile([0.5, -0.5, 0.5, 0.5], (9, 1))))
assert calls == [("const_D", (9, 21), 2, 4, (155,)), ("series", (9, 21), 2, 4, (9, 2))], calls
assert len(w.frame) == 9
print("ok")
""" % (root, os.path.join(root, "tests", "golden"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]
