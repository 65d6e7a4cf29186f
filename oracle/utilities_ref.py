"""ORACLE (test infrastructure only -- never imported by the product): CPU restatement of the storage-format bit
transforms of scri/utilities.py:194-406 -- xor_timeseries (:195-217), xor_timeseries_reverse (:220-232), fletcher32
(:235-268) and multishuffle (:271-406).  Integer work: parity is bit-exact.  Pinned by the reference's own tests
(tests/test_utilities.py: reversibility; byte-wise multishuffle == the HDF5 shuffle filter = byte transpose) and the
published Fletcher-32 test vectors ("abcde" -> 0xF04FC729, "abcdef" -> 0x56502D2A, "abcdefgh" -> 0xEBE19591)."""
import numpy as np


def xor_timeseries(c):
    """utilities.py:195-217 (returns a new array): row i >= 1 becomes row[i-1] ^ row[i], rows viewed as uint64"""
    u = np.array(c, copy=True).view(np.uint64)
    out = u.copy()
    out[1:] = np.bitwise_xor(u[:-1], u[1:])
    return out.view(np.asarray(c).dtype)


def xor_timeseries_reverse(c):
    """utilities.py:220-232: running XOR along the first axis"""
    u = np.array(c, copy=True).view(np.uint64)
    return np.bitwise_xor.accumulate(u, axis=0).view(np.asarray(c).dtype)


def fletcher32(data):
    """utilities.py:235-268, literal loop (16-bit words, blocks of 360, modulus 65535)"""
    d = np.ascontiguousarray(data).reshape(-1).view(np.uint16)
    c0 = 0
    c1 = 0
    j = 0
    while j < d.size:
        n = min(360, d.size - j)
        for i in range(n):
            c0 = (c0 + int(d[j])) & 0xFFFFFFFF
            c1 = (c1 + c0) & 0xFFFFFFFF
            j += 1
        c0 %= 65535
        c1 %= 65535
    return (c1 << 16 | c0) & 0xFFFFFFFF


def multishuffle(a, shuffle_widths, forward=True):
    """utilities.py:271-406, literal loops in Python integers (small inputs only)"""
    bit_width = int(np.sum(shuffle_widths))
    if bit_width not in (8, 16, 32, 64):
        raise ValueError(f"Total bit width must be one of [8, 16, 32, 64], not {bit_width}")
    dtype = np.dtype(f"u{bit_width // 8}")
    x = [int(v) for v in np.ascontiguousarray(a).view(dtype)]
    full = (1 << bit_width) - 1
    rev = list(reversed([int(w) for w in shuffle_widths]))
    y = [0] * len(x)
    bit = 0
    for i, w in enumerate(rev):
        shift = sum(rev[:i])
        mask = (1 << w) - 1
        for e in range(len(x)):
            bi, be = divmod(bit, bit_width)
            if forward:
                masked = (x[e] >> shift) & mask
                y[bi] = (y[bi] + (masked << be)) & full
                if be + w > bit_width:
                    y[bi + 1] = (y[bi + 1] + (masked >> (bit_width - be))) & full
            else:
                masked = (x[bi] >> be) & mask
                y[e] = (y[e] + (masked << shift)) & full
                if be + w > bit_width:
                    y[e] = (y[e] + ((((x[bi + 1] << (bit_width - be)) & full) & mask) << shift)) & full
            bit += w
    return np.array(y, dtype=dtype)
