"""GPU parity of AsymptoticBondiData.transform (bms_transform_abd) against the CPU oracle."""
import numpy as np
import pytest

from oracle import abd_ref, wigner
from oracle.containers import ABD

pytestmark = pytest.mark.gpu


def smooth_abd(n, ell_max, seed, t0=-15.0, t1=25.0):
    rng = np.random.default_rng(seed)
    u = np.linspace(t0, t1, n)
    nm = (ell_max + 1) ** 2
    LM = wigner.LM_range(0, ell_max)
    raw = np.zeros((6, n, nm), dtype=complex)
    phase = 0.07 * u + 3e-4 * u**2
    for i, s in enumerate(ABD.spins):
        a = (rng.normal(size=nm) + 1j * rng.normal(size=nm)) * 10.0 ** (-LM[:, 0] / 4.0)
        a[: s * s] = 0
        raw[i] = a[None, :] * np.exp(1j * LM[None, :, 1] * phase[:, None]) * (1 + 0.01 * u[:, None])
    return ABD(u, raw, ell_max)


def real_st(ell_max, seed, scale):
    rng = np.random.default_rng(seed)
    a = scale * (rng.normal(size=(ell_max + 1) ** 2) + 1j * rng.normal(size=(ell_max + 1) ** 2))
    return a  # the ABD flavour imposes reality itself


CASES = [
    dict(),
    dict(time_translation=0.6),
    dict(space_translation=[0.2, 0.1, -0.3]),
    dict(boost_velocity=[0.02, -0.01, 0.03]),
    dict(supertranslation="st", frame_rotation=[0.4, 1, -2, 0.3], boost_velocity=[3e-3, 1e-3, -2e-3]),
    dict(supertranslation="st", working_ell_max=7, output_ell_max=3),
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_abd_transform_matches_oracle(ctx, case):
    import scri_amd

    kw = dict(CASES[case])
    if kw.get("supertranslation") == "st":
        kw["supertranslation"] = real_st(2, 33, 0.05)
    o = smooth_abd(300, 4, 50 + case)
    expect = abd_ref.transform(o, **{k: (np.array(v) if isinstance(v, list) else v) for k, v in kw.items()})
    g = scri_amd.AsymptoticBondiData(o.u, o.ell_max, ctx=ctx)
    g._raw_data[:] = o.raw
    got = g.transform(**kw)
    assert got.n_times == expect.n_times and got.ell_max == expect.ell_max
    assert np.abs(got.u - expect.u).max() < 1e-13
    scale = max(1.0, np.abs(expect.raw).max())
    for i, name in enumerate(("psi0", "psi1", "psi2", "psi3", "psi4", "sigma")):
        err = np.abs(getattr(got, name) - expect.raw[i]).max()
        assert err < 1e-12 * scale, (name, err)


def test_abd_interpolate_matches_oracle(ctx):
    import scri_amd

    o = smooth_abd(200, 3, 7)
    g = scri_amd.AsymptoticBondiData(o.u, o.ell_max, ctx=ctx)
    g._raw_data[:] = o.raw
    tn = np.linspace(o.u[3], o.u[-5], 333)
    assert np.abs(g.interpolate(tn)._raw_data - o.interpolate(tn).raw).max() < 1e-12
