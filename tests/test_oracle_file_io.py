"""Host logic of scri_amd.file_io against the loop restatement (oracle/file_io_ref.py): monotonic-time selection."""
import numpy as np


def test_index_is_monotonic_matches_the_loop():
    from oracle import file_io_ref
    from scri_amd import file_io

    rng = np.random.default_rng(3)
    for trial in range(50):
        n = int(rng.integers(2, 60))
        y = np.cumsum(rng.normal(0.3, 1.0, size=n))
        if trial % 2:
            y = -y
        if trial % 5 == 0:
            y[rng.integers(0, n, size=3)] = y[0]  # ties are dropped too
        assert np.array_equal(file_io.index_is_monotonic(y), file_io_ref.index_is_monotonic(y))
        assert np.array_equal(file_io.monotonize(y), y[file_io_ref.index_is_monotonic(y)])
    assert file_io.index_is_monotonic(np.array([1.0])).tolist() == [True]
    assert file_io.monotonic_indices(np.array([0.0, 1.0, 0.5, 2.0])).tolist() == [0, 1, 3]
