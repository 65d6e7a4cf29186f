"""Host restatements of the numpy-quaternion / scri utility pieces the frame functions need (scri_amd/quaternions.py,
scri_amd/utilities.py::transition_function): algebraic identities and known answers."""
import numpy as np


def _random_rotors(n, seed):
    q = np.random.default_rng(seed).normal(size=(n, 4))
    return q / np.linalg.norm(q, axis=-1)[:, None]


def test_log_exp_sqrt_slerp():
    from scri_amd import quaternions as Q

    R = _random_rotors(200, 1)
    assert np.abs(Q.exp(Q.log(R)) - R).max() < 1e-14
    assert np.abs(Q.log(R)[:, 0]).max() < 1e-14  # unit quaternions: pure-vector logarithm
    s = Q.sqrt(R)
    assert np.abs(Q.multiply(s, s) - R).max() < 1e-14
    one = np.array([1.0, 0, 0, 0])
    assert np.array_equal(Q.log(one), np.zeros(4)) and np.array_equal(Q.exp(np.zeros(4)), one)
    A, B = _random_rotors(50, 2), _random_rotors(50, 3)
    assert np.abs(Q.slerp(A, B, np.zeros(50)) - A).max() < 1e-14
    end = Q.slerp(A, B, np.ones(50))
    assert np.minimum(np.abs(end - B).max(axis=-1), np.abs(end + B).max(axis=-1)).max() < 1e-13
    half = Q.slerp(A, B, np.full(50, 0.5))
    # half way: the same angle to both ends
    da = np.abs(np.sum(half * A, axis=-1))
    db = np.abs(np.sum(half * B, axis=-1))
    assert np.abs(da - db).max() < 1e-13


def test_minimal_rotation_removes_the_spin_about_the_axis():
    from scri_amd import quaternions as Q

    t = np.linspace(0, 100, 2000)
    th, ph = 0.4, 0.05 * t
    v = np.stack([np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th) * np.ones_like(t)], axis=-1)
    R = Q.sqrt(np.concatenate([v[:, 2:3], -np.cross(v, np.array([0.0, 0.0, 1.0]))], axis=-1))  # sqrt(-v z): z -> v
    z = np.array([0.0, 0, 0, 1])
    assert np.abs(Q.multiply(Q.multiply(R, z), Q.conjugate(R))[:, 1:] - v).max() < 1e-15
    junk = np.zeros((t.size, 4))
    junk[:, 0], junk[:, 3] = np.cos(0.35 * np.sin(0.3 * t)), np.sin(0.35 * np.sin(0.3 * t))
    R = Q.multiply(R, junk)
    Rm = Q.minimal_rotation(R, t, iterations=3)
    axis = Q.multiply(Q.multiply(Rm, z), Q.conjugate(Rm))[:, 1:]
    assert np.abs(axis - v).max() < 1e-14  # the axis is untouched
    along = np.sum(Q.angular_velocity(Rm, t) * axis, axis=-1)
    before = np.sum(Q.angular_velocity(R, t) * axis, axis=-1)
    assert np.abs(along[50:-50]).max() < 1e-12 and np.abs(before).max() > 0.1


def test_optimal_alignment():
    from scri_amd import quaternions as Q

    rng = np.random.default_rng(5)
    R = _random_rotors(1, 6)[0]
    R = R if R[0] >= 0 else -R
    a = rng.normal(size=(40, 3))
    b = Q.multiply(Q.multiply(R, np.concatenate([np.zeros((40, 1)), a], axis=-1)), Q.conjugate(R))[:, 1:]
    assert np.abs(Q.optimal_alignment_in_Euclidean_metric(a, b) - R).max() < 1e-14
    # time-weighted (cubic-spline integral of a_j b_k): smooth vector functions of time
    t = np.linspace(0, 10, 200) + 0.01 * rng.uniform(-1, 1, 200)
    a = np.stack([np.sin(t), np.cos(0.7 * t), 0.3 * t - 1.0], axis=-1)
    b = Q.multiply(Q.multiply(R, np.concatenate([np.zeros((200, 1)), a], axis=-1)), Q.conjugate(R))[:, 1:]
    assert np.abs(Q.optimal_alignment_in_Euclidean_metric(a, b, t) - R).max() < 1e-13


def test_transition_function():
    from scri_amd.utilities import transition_function

    x = np.linspace(0, 1, 101)
    f, i0, i1 = transition_function(x, 0.2, 0.8, return_indices=True)
    assert np.all(f[:i0] == 0.0) and np.all(f[i1:] == 1.0) and x[i0 - 1] <= 0.2 < x[i0] and x[i1 - 1] < 0.8 <= x[i1]
    assert np.all(np.diff(f) >= 0) and abs(f[50] - 0.5) < 1e-15
    assert np.allclose(f + transition_function(x, 0.2, 0.8, y0=1.0, y1=0.0), 1.0, atol=1e-15)
    g = transition_function(x, 0.2, 0.8, y0=3.0, y1=-1.0)
    assert g[0] == 3.0 and g[-1] == -1.0


def test_from_spherical_coords_takes_z_to_the_direction():
    from oracle import quat
    from scri_amd import quaternions

    rng = np.random.default_rng(2)
    th, ph = rng.uniform(0, np.pi, (4, 6)), rng.uniform(0, 2 * np.pi, (4, 6))
    R = quaternions.from_spherical_coords(th, ph)
    assert R.shape == (4, 6, 4) and np.abs(np.linalg.norm(R, axis=-1) - 1).max() < 1e-15
    assert np.abs(R - quat.from_spherical_coords(th, ph)).max() == 0.0
    z = np.zeros((4, 6, 4))
    z[..., 3] = 1.0
    v = quaternions.multiply(quaternions.multiply(R, z), quaternions.conjugate(R))[..., 1:]
    n = np.stack([np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th)], axis=-1)
    assert np.abs(v - n).max() < 1e-15
