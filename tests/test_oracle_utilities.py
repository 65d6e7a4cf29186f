"""Pins of oracle/utilities_ref.py (bit transforms of scri/utilities.py:194-406)."""
import numpy as np
import pytest

from oracle import utilities_ref as ur


def _widths(rng, bit_width):
    possible = 2 ** np.arange(0, int(np.log2(bit_width)))
    w = []
    while sum(w) < bit_width:
        nxt = int(rng.choice(possible))
        if sum(w) + nxt <= bit_width:
            w.append(nxt)
    return tuple(w)


@pytest.mark.parametrize("bit_width", [8, 16, 32, 64])
def test_multishuffle_reversibility_and_hdf5_equivalence(bit_width):
    # tests/test_utilities.py:20-53
    rng = np.random.default_rng(123 + bit_width)
    dt = np.dtype(f"u{bit_width // 8}")
    data = rng.integers(0, 2**bit_width, size=400, dtype=dt)
    for widths in [(1,) * bit_width, (8,) * (bit_width // 8)] + [_widths(rng, bit_width) for _ in range(6)]:
        sh = ur.multishuffle(data, widths)
        assert np.array_equal(data, ur.multishuffle(sh, widths, forward=False)), widths
    # byte-wise multishuffle is the HDF5 shuffle filter: byte k of every element stored together
    hdf5 = data.view(np.uint8).reshape(data.size, bit_width // 8).T.copy().reshape(-1).view(dt)
    assert np.array_equal(ur.multishuffle(data, (8,) * (bit_width // 8)), hdf5)


def test_fletcher32_published_vectors():
    for text, expect in ((b"abcde", 0xF04FC729), (b"abcdef", 0x56502D2A), (b"abcdefgh", 0xEBE19591)):
        padded = text + bytes(len(text) % 2)
        assert ur.fletcher32(np.frombuffer(padded, dtype=np.uint8)) == expect


def test_xor_timeseries_round_trip_and_definition():
    rng = np.random.default_rng(4)
    c = rng.normal(size=(50, 7)) + 1j * rng.normal(size=(50, 7))
    x = ur.xor_timeseries(c)
    assert np.array_equal(x[0], c[0])
    assert np.array_equal(x.view(np.uint64)[5], c.view(np.uint64)[4] ^ c.view(np.uint64)[5])
    assert np.array_equal(ur.xor_timeseries_reverse(x).view(np.uint64), c.view(np.uint64))
