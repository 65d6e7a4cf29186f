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
