"""GPU parity (bit-exact) of the storage-format bit transforms against the oracle, plus the reference's own tests
(tests/test_utilities.py) and larger round trips."""
import numpy as np
import pytest

from oracle import utilities_ref as ur

pytestmark = pytest.mark.gpu


def _widths(rng, bit_width):
    possible = 2 ** np.arange(0, int(np.log2(bit_width)))
    w = []
    while sum(w) < bit_width:
        nxt = int(rng.choice(possible))
        if sum(w) + nxt <= bit_width:
            w.append(nxt)
    return tuple(w)


@pytest.mark.parametrize("bit_width", [8, 16, 32, 64])
def test_multishuffle_matches_oracle_bit_for_bit(ctx, bit_width):
    from scri_amd import utilities

    rng = np.random.default_rng(123 + bit_width)
    dt = np.dtype(f"u{bit_width // 8}")
    data = rng.integers(0, 2**bit_width, size=777, dtype=dt)
    odd = {8: (3, 5), 16: (5, 3, 7, 1), 32: (11, 9, 12), 64: (13, 17, 3, 31)}[bit_width]  # widths that straddle words
    for widths in [(1,) * bit_width, (8,) * (bit_width // 8), (bit_width,), odd] + [_widths(rng, bit_width) for _ in range(5)]:
        sh = utilities.multishuffle(widths)(data)
        assert sh.dtype == dt and np.array_equal(sh, ur.multishuffle(data, widths)), widths
        assert np.array_equal(utilities.multishuffle(widths, forward=False)(sh), data), widths
    hdf5 = data.view(np.uint8).reshape(data.size, bit_width // 8).T.copy().reshape(-1).view(dt)
    assert np.array_equal(utilities.multishuffle((8,) * (bit_width // 8))(data), hdf5)  # == the HDF5 shuffle filter
    with pytest.raises(ValueError, match="Total bit width"):
        utilities.multishuffle((8, 4))


def test_multishuffle_large_round_trip(ctx):
    from scri_amd import utilities

    rng = np.random.default_rng(9)
    data = rng.integers(0, 2**63, size=2_000_003, dtype=np.uint64)
    widths = (8, 8, 4, 4, 4, 4, 2, 2, 2, 2, 1, 1, 1, 1, 4, 16)
    sh = utilities.multishuffle(widths)(data)
    assert not np.array_equal(sh, data)
    assert np.array_equal(utilities.multishuffle(widths, forward=False)(sh), data)
    # the least significant piece (16 bits) of every element leads the stream
    assert np.array_equal(sh[: data.size // 4].view(np.uint16)[: 1000], (data[:1000] & 0xFFFF).astype(np.uint16))


def test_xor_timeseries_matches_oracle(ctx):
    from scri_amd import utilities

    rng = np.random.default_rng(4)
    for shape in ((5000, 77), (3, 2), (1, 8), (1025, 1)):
        c = rng.normal(size=shape) + 1j * rng.normal(size=shape)
        x = utilities.xor_timeseries(c.copy())
        assert np.array_equal(x.view(np.uint64), ur.xor_timeseries(c).view(np.uint64)), shape
        back = utilities.xor_timeseries_reverse(x.copy())
        assert np.array_equal(back.view(np.uint64), c.view(np.uint64)), shape
    f = rng.normal(size=(300, 4))
    assert np.array_equal(utilities.xor_timeseries_reverse(utilities.xor_timeseries(f.copy())), f)


def test_fletcher32(ctx):
    from scri_amd import utilities

    for text, expect in ((b"abcde", 0xF04FC729), (b"abcdef", 0x56502D2A), (b"abcdefgh", 0xEBE19591)):
        padded = text + bytes(len(text) % 2)
        assert utilities.fletcher32(np.frombuffer(padded, dtype=np.uint8)) == expect
    rng = np.random.default_rng(1)
    for n in (1, 359, 360, 361, 100_001):
        d = rng.integers(0, 2**16, size=n, dtype=np.uint16)
        assert utilities.fletcher32(d) == ur.fletcher32(d), n
    big = rng.normal(size=1_000_000)
    d = big.view(np.uint16).astype(np.uint64)
    n = d.size
    expect = (int(((n - np.arange(n)) % 65535 * d % 65535).sum() % 65535) << 16) | int(d.sum() % 65535)
    assert utilities.fletcher32(big) == expect
