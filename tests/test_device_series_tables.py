"""CPU: the per-column tables scri_amd/device_series.py hands to bms_mode_map (eth / ethbar factors, the bar permutation,
embeddings of one l range in another) reproduce the host ModesTimeSeries operators when the kernel's map
    out[t][j] = ca_j op(A[t][ia_j]) + cb_j op(B[t][ib_j])
is evaluated with numpy (the kernel itself is compared with the host class on the GPU: tests/test_gpu_device_resident.py)."""
import numpy as np
import pytest

from scri_amd import device_series as ds
from scri_amd.mode_algebra import LM_total_size


def _map(A, ia, ca, conj_a=False, B=None, ib=None, cb=None, conj_b=False):
    def side(X, idx, coef, conj):
        src = np.where(idx[None, :] >= 0, X[:, np.maximum(idx, 0)], 0.0)
        return coef[None, :] * (np.conj(src) if conj else src)

    out = side(A, np.asarray(ia), np.asarray(ca), conj_a)
    if B is not None:
        out = out + side(B, np.asarray(ib), np.asarray(cb), conj_b)
    return out


def _series(ell_min, ell_max, s, seed):
    from scri_amd.modes_time_series import ModesTimeSeries

    rng = np.random.default_rng(seed)
    n = LM_total_size(ell_min, ell_max)
    a = rng.normal(size=(7, n)) + 1j * rng.normal(size=(7, n))
    return ModesTimeSeries(a, np.arange(7.0), spin_weight=s, ell_min=ell_min, ell_max=ell_max)


@pytest.mark.parametrize("s", [-2, -1, 0, 1, 2])
@pytest.mark.parametrize("ell_min,ell_max", [(0, 5), (2, 6)])
def test_operator_tables(s, ell_min, ell_max):
    h = _series(ell_min, ell_max, s, 100 + 10 * s + ell_max)
    A = h.ndarray
    n = A.shape[1]
    ident = ds._identity(n)
    assert np.allclose(_map(A, ident, ds._eth_factor(ell_min, ell_max, s, True)), h.eth.ndarray, atol=1e-14)
    assert np.allclose(_map(A, ident, ds._eth_factor(ell_min, ell_max, s, False)), h.ethbar.ndarray, atol=1e-14)
    perm, sign = ds._bar_tables(ell_min, ell_max, s)
    assert np.array_equal(_map(A, perm, sign, conj_a=True), h.bar.ndarray)
    if s == 0:
        assert np.allclose(_map(A, ident, np.full(n, 0.5 + 0j), B=A, ib=perm, cb=0.5 * sign, conj_b=True), h.real.ndarray, atol=1e-15)
        assert np.allclose(_map(A, ident, np.full(n, -0.5j), B=A, ib=perm, cb=0.5j * sign, conj_b=True), h.imag.ndarray, atol=1e-15)


def test_embedding_tables_add_series_of_different_ranges():
    a, b = _series(2, 4, -1, 1), _series(0, 6, -1, 2)
    lo, hi = 0, 6
    n = LM_total_size(lo, hi)
    got = _map(a.ndarray, ds._embed(2, 4, lo, hi), np.ones(n), B=b.ndarray, ib=ds._embed(0, 6, lo, hi), cb=-np.ones(n))
    assert np.array_equal(got, (a - b).ndarray)
    # truncation = an embedding into a shorter range
    n3 = LM_total_size(0, 3)
    assert np.array_equal(_map(b.ndarray, ds._embed(0, 6, 0, 3), np.ones(n3)), b.truncate_ell(3).ndarray)
    assert np.all(ds._embed(3, 4, 0, 2) == -1)
