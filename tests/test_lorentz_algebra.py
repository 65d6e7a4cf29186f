"""LorentzTransformation algebra (host-only): mirrors tests/test_bms_transformations.py:53-190 of the reference
(reorder / inverse / compose against 4x4 Lorentz matrices and the Wigner-rotation formula)."""
import numpy as np

from oracle import quat
from scri_amd import bms_transformations as bt

Q1 = np.array([1.0, 2, 3, 4]) / np.sqrt(30)
Q2 = np.array([5.0, -6, 7, -8]) / np.sqrt(174)
V1 = np.array([1, 2, 3]) * 1e-2
V2 = np.array([-4, 5, -6]) * 1e-2


def rotation_matrix(q):
    s, x, y, z = np.asarray(q)
    return np.array(
        [
            [1, 0, 0, 0],
            [0, 1 - 2 * y**2 - 2 * z**2, 2 * x * y - 2 * s * z, 2 * x * z + 2 * s * y],
            [0, 2 * x * y + 2 * s * z, 1 - 2 * x**2 - 2 * z**2, 2 * y * z - 2 * s * x],
            [0, 2 * x * z - 2 * s * y, 2 * y * z + 2 * s * x, 1 - 2 * x**2 - 2 * y**2],
        ]
    )


def boost_matrix(v):
    n = np.linalg.norm(v)
    g = 1 / np.sqrt(1 - n**2)
    B = np.eye(4)
    B[0, 0] = g
    B[0, 1:] = B[1:, 0] = -g * v
    B[1:, 1:] += (g - 1) * np.outer(v, v) / n**2
    return B


def test_Lorentz_reorder():
    L = bt.LorentzTransformation(frame_rotation=Q1, boost_velocity=V1, order=["frame_rotation", "boost_velocity"])
    Lr = L.reorder(output_order=L.order[::-1])
    assert Lr.order == ["boost_velocity", "frame_rotation"]
    assert L.is_close_to(Lr.reorder(output_order=L.order))
    assert np.allclose(boost_matrix(V1) @ rotation_matrix(Q1), rotation_matrix(Lr.frame_rotation) @ boost_matrix(Lr.boost_velocity))


def test_pure_inverses():
    assert np.allclose(quat.qconj(Q1), bt.LorentzTransformation(frame_rotation=Q1).inverse().frame_rotation.components)
    assert np.allclose(-V1, bt.LorentzTransformation(boost_velocity=V1).inverse().boost_velocity)


def test_Lorentz_inverse_and_composition_consistency():
    L = bt.LorentzTransformation(frame_rotation=Q1, boost_velocity=V1, order=["frame_rotation", "boost_velocity"])
    Li = L.inverse()
    assert (L * Li).is_close_to(bt.LorentzTransformation())
    assert (Li * L).is_close_to(bt.LorentzTransformation())
    expect = bt.LorentzTransformation(frame_rotation=quat.qconj(Q1), boost_velocity=-V1, order=["boost_velocity", "frame_rotation"])
    assert expect.is_close_to(L.inverse(output_order=["boost_velocity", "frame_rotation"]))


def test_frame_rotation_composition():
    L1, L2 = bt.LorentzTransformation(frame_rotation=Q1), bt.LorentzTransformation(frame_rotation=Q2)
    assert np.allclose(quat.qmul(Q2, Q1), (L1 * L2).frame_rotation.components)
    assert np.allclose(quat.qmul(Q1, Q2), (L2 * L1).frame_rotation.components)


def test_boost_velocity_composition_wigner_rotation():
    """Eqs. (65)-(70) of arXiv:1102.2001, as in the reference test."""
    L1, L2 = bt.LorentzTransformation(boost_velocity=V1), bt.LorentzTransformation(boost_velocity=V2)
    g1, g2 = 1 / np.sqrt(1 - V1 @ V1), 1 / np.sqrt(1 - V2 @ V2)
    g12 = g1 * g2 * (1 + V1 @ V2)
    v1v2 = (V1 + g2 * V2 + (g2 - 1) * (V1 @ V2) * V2 / (V2 @ V2)) / (g2 * (1 + V1 @ V2))
    v2v1 = (V2 + g1 * V1 + (g1 - 1) * (V2 @ V1) * V1 / (V1 @ V1)) / (g1 * (1 + V2 @ V1))
    theta = np.arccos((1 + g1 + g2 + g12) ** 2 / ((1 + g1) * (1 + g2) * (1 + g12)) - 1)
    c = np.cross(V1, V2)
    q12 = np.array([np.cos(theta / 2), *(c / np.linalg.norm(c) * np.sin(theta / 2))])
    q21 = np.array([np.cos(theta / 2), *(-c / np.linalg.norm(c) * np.sin(theta / 2))])
    L12, L21 = L1 * L2, L2 * L1
    assert np.allclose(v1v2, L12.boost_velocity) and np.allclose(q12, L12.frame_rotation.components)
    assert np.allclose(v2v1, L21.boost_velocity) and np.allclose(q21, L21.frame_rotation.components)


def test_Lorentz_composition_against_4x4_matrices():
    L1 = bt.LorentzTransformation(frame_rotation=Q1, boost_velocity=V1)
    L2 = bt.LorentzTransformation(frame_rotation=Q2, boost_velocity=V2)
    for A, B_, (qa, va, qb, vb) in ((L1, L2, (Q1, V1, Q2, V2)), (L2, L1, (Q2, V2, Q1, V1))):
        C = A * B_
        expect = boost_matrix(vb) @ rotation_matrix(qb) @ boost_matrix(va) @ rotation_matrix(qa)
        assert np.allclose(expect, boost_matrix(C.boost_velocity) @ rotation_matrix(C.frame_rotation))
