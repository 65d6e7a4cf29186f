"""BMSTransformation algebra with transform_supertranslation on the GPU: mirrors
tests/test_bms_transformations.py:297-338 (reorder / inverse consistency) and checks
transform_supertranslation against the oracle (scri/bms_transformations.py:151-180)."""
import numpy as np
import pytest

from oracle import abd_ref

pytestmark = pytest.mark.gpu

S = np.array([1, 2 + 4j, 3, -2 + 4j, 7 - 5j, -3 - 2j, 4, 3 - 2j, 7 + 5j]) * 1e-3
Q = np.array([1.0, 2, 3, 4]) / np.sqrt(30)
V = np.array([1, 2, 3]) * 1e-4


def test_transform_supertranslation_matches_oracle(ctx):
    from scri_amd import bms_transformations as bt, synthetic

    Sr = synthetic.real_supertranslation(S)
    L = bt.LorentzTransformation(frame_rotation=Q, boost_velocity=np.array([0.02, -0.01, 0.03]), ell_max=6)
    got = bt.transform_supertranslation(Sr, L, ctx=ctx)
    Li = L.inverse(output_order=["frame_rotation", "boost_velocity"])
    expect = abd_ref.transform_supertranslation(Sr, Li.frame_rotation.components, Li.boost_velocity, 6)
    assert np.abs(got - expect).max() < 1e-15


def test_BMS_reorder_consistency(ctx):
    from scri_amd import bms_transformations as bt

    B = bt.BMSTransformation(supertranslation=S, frame_rotation=Q, boost_velocity=V, ctx=ctx)
    orders = [
        ["supertranslation", "frame_rotation", "boost_velocity"],
        ["frame_rotation", "supertranslation", "boost_velocity"],
        ["frame_rotation", "boost_velocity", "supertranslation"],
        ["supertranslation", "boost_velocity", "frame_rotation"],
        ["boost_velocity", "supertranslation", "frame_rotation"],
        ["boost_velocity", "frame_rotation", "supertranslation"],
    ]
    for o in orders:
        Bo = B.reorder(o)
        assert Bo.order == o
        assert B.is_close_to(Bo.reorder(B.order)), o
    chain = B
    for o in (orders[1], orders[3], orders[4], orders[2], orders[5], B.order):
        chain = chain.reorder(o)
    assert B.is_close_to(chain)


def test_BMS_inverse_composition_consistency(ctx):
    from scri_amd import bms_transformations as bt

    B = bt.BMSTransformation(supertranslation=S, frame_rotation=Q, boost_velocity=V, ctx=ctx)
    Bi = B.inverse()
    ident = bt.BMSTransformation()
    assert (B * Bi).is_close_to(ident)
    assert (Bi * B).is_close_to(ident)


# ---- the reference's ABD-level tests of the group algebra (tests/test_bms_transformations.py:192-294, 341-460): the
# transformations are applied to Kerr data with the GPU engine, their inverses / compositions come from the algebra
def _kerr_abd(ctx):
    import scri_amd

    mass, spin, ell_max = 2.0, 0.456, 8
    u = np.linspace(-100, 100, num=500)
    nm = (ell_max + 1) ** 2
    psi2, psi1 = np.zeros(nm, dtype=complex), np.zeros(nm, dtype=complex)
    psi2[0] = -mass * np.sqrt(4 * np.pi)
    psi1[2] = -np.sqrt(2) * (3j * spin / 2) * (np.sqrt((8 / 3) * np.pi))
    return scri_amd.AsymptoticBondiData.from_initial_values(u, ell_max=ell_max, psi2=psi2, psi1=psi1, ctx=ctx)


def _fields(a):
    return np.array([np.asarray(getattr(a, f)) for f in ("sigma", "psi4", "psi3", "psi2", "psi1", "psi0")])


def _apply(a, T):
    kw = dict(frame_rotation=T.frame_rotation.components, boost_velocity=T.boost_velocity)
    if hasattr(T, "supertranslation"):
        kw["supertranslation"] = T.supertranslation
    return a.transform(**kw)


def _common_window(abd, a, b):
    lo = np.argmin(abs(abd.t - max(a.t[0], b.t[0])))
    hi = np.argmin(abs(abd.t - min(a.t[-1], b.t[-1]))) + 1
    return a.interpolate(abd.t[lo:hi]), b.interpolate(abd.t[lo:hi])


Q2 = np.array([5.0, -6, 7, -8]) / np.sqrt(174)
V2 = np.array([-4, 5, -6]) * 1e-4
S2 = np.array([-3, 1 - 2j, 5, -1 - 2j, -6 - 4j, 0 + 1j, 3, 0 + 1j, -6 + 4j]) * 1e-3


@pytest.mark.parametrize("kind", ["Lorentz", "BMS"])
def test_abd_inverse(ctx, kind):
    from scri_amd import bms_transformations as bt

    abd = _kerr_abd(ctx)
    if kind == "Lorentz":
        T = bt.LorentzTransformation(frame_rotation=Q, boost_velocity=V, order=["frame_rotation", "boost_velocity"])
        Ti = T.inverse(output_order=["frame_rotation", "boost_velocity"])
    else:
        order = ["supertranslation", "frame_rotation", "boost_velocity"]
        T = bt.BMSTransformation(supertranslation=S, frame_rotation=Q, boost_velocity=V, order=order, ctx=ctx)
        Ti = T.inverse(output_order=order)
    check = _apply(_apply(abd, T), Ti)
    assert np.allclose(_fields(abd.interpolate(check.t)), _fields(check))


@pytest.mark.parametrize("kind", ["Lorentz", "BMS"])
def test_abd_composition(ctx, kind):
    from scri_amd import bms_transformations as bt

    abd = _kerr_abd(ctx)
    if kind == "Lorentz":
        order = ["frame_rotation", "boost_velocity"]
        T1 = bt.LorentzTransformation(frame_rotation=Q, boost_velocity=V, order=order)
        T2 = bt.LorentzTransformation(frame_rotation=Q2, boost_velocity=V2, order=order)
    else:
        order = ["supertranslation", "frame_rotation", "boost_velocity"]
        T1 = bt.BMSTransformation(supertranslation=S, frame_rotation=Q, boost_velocity=V, order=order, ctx=ctx)
        T2 = bt.BMSTransformation(supertranslation=S2, frame_rotation=Q2, boost_velocity=V2, order=order, ctx=ctx)
    composed = T2 * T1
    two_steps = _apply(_apply(abd, T1), T2)
    one_step = _apply(abd, composed)
    a, b = _common_window(abd, two_steps, one_step)
    assert np.allclose(_fields(a), _fields(b))
