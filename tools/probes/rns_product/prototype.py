"""Prototype (numpy, CPU) of the complex fp64 product emulated in residue arithmetic, as the int8 matrix pipe would run it:
every step below is the arithmetic the device kernels perform, in the same number formats, checked against exact
integer arithmetic (Python ints) and against a double-double reference of the product.

  A (rows x K, complex)  -> row scale 2^sA_i, integers X = rint(a 2^sA_i), |X| <= 2^53, L1 of the row <= 2^61
  B (K x cols, complex)  -> column scale 2^sB_j, |X| <= 2^53
  16 moduli m = 1 mod 4 (or powers of such primes): a complex integer x + i y has the two images x + iota y, x - iota y
  (iota^2 = -1 mod m), each a ring homomorphism, so the complex product is TWO real products per modulus: 32 int8 planes
  int32 sums t+ , t- per modulus; fraction of the Chinese remainder reconstruction
     frac_x = sum_l (t+ + t-) theta_l  mod 1,  frac_y = sum_l (t+ - t-) eta_l  mod 1,   X_C = P frac (centred)
Usage: python tools/probes/rns_product_prototype.py [rows] [K] [cols]"""
import sys
from fractions import Fraction

import numpy as np

MODULI = [241, 233, 229, 197, 193, 181, 173, 169, 157, 149, 137, 125, 113, 109, 101, 97]


def constants():
    P = 1
    for m in MODULI:
        P *= m
    out = []
    for m in MODULI:
        iota = next(i for i in range(2, m) if (i * i + 1) % m == 0)
        c = pow(P // m, -1, m)
        gx = (pow(2, -1, m) * c) % m
        gy = (pow(2 * iota, -1, m) * c) % m
        out.append((m, iota, gx, gy))
    return P, out


def split_weight(g, m, bits_hi):
    """g / m as hi (bits_hi bits below the binary point, exact) + lo (double)"""
    hi_int = (g << bits_hi) // m
    hi = hi_int / 2.0**bits_hi
    lo = float(Fraction((g << bits_hi) - hi_int * m, m) / 2**bits_hi)
    return hi, lo


def scales(M, axis):
    """power-of-two scale per row (axis=1) of a complex matrix: largest part < 2^53 and L1 norm of the parts <= 2^61"""
    parts = np.maximum(np.abs(M.real), np.abs(M.imag)).max(axis=axis)
    l1 = (np.abs(M.real) + np.abs(M.imag)).sum(axis=axis)
    _, e_max = np.frexp(parts)
    _, e_l1 = np.frexp(l1)
    s = np.minimum(53 - e_max, 61 - e_l1)
    s[parts == 0] = 0
    return s


def residues(X, m, iota):
    """the two images of the complex integers X (float64 arrays holding integers < 2^53 in magnitude), as the device forms
    them: fp64 remainder of each part (quotient off by one at most when the fraction is within 2^-6 of a half: the
    remainder then stays within m/2 + 4), then the images and a second, exact remainder"""
    inv = 1.0 / m
    rx = X.real - m * np.rint(X.real * inv)
    ry = X.imag - m * np.rint(X.imag * inv)
    assert np.abs(rx).max() <= m / 2 + 4 and np.abs(ry).max() <= m / 2 + 4
    u = rx + iota * ry
    v = rx - iota * ry
    u = u - m * np.rint(u * np.float32(inv))
    v = v - m * np.rint(v * np.float32(inv))
    assert np.abs(u).max() <= 127 and np.abs(v).max() <= 127
    return u.astype(np.int8), v.astype(np.int8)


def rns_product(A, B, fold="reduce"):
    P, cons = constants()
    sA = scales(A, 1)
    sB = scales(B, 0)
    XA = np.rint(np.ldexp(A.real, sA[:, None])) + 1j * np.rint(np.ldexp(A.imag, sA[:, None]))
    XB = np.rint(np.ldexp(B.real, sB[None, :])) + 1j * np.rint(np.ldexp(B.imag, sB[None, :]))
    assert max(np.abs(XA.real).max(), np.abs(XA.imag).max(), np.abs(XB.real).max(), np.abs(XB.imag).max()) <= 2.0**53
    rows, cols = A.shape[0], B.shape[1]
    if fold == "reduce":
        S = np.zeros((4, rows, cols))
    else:
        S = np.zeros((6, rows, cols))
    for m, iota, gx, gy in cons:
        ua, va = residues(XA, m, iota)
        ub, vb = residues(XB, m, iota)
        tp = ua.astype(np.int32) @ ub.astype(np.int32)  # int8 x int8 -> int32, exact
        tm = va.astype(np.int32) @ vb.astype(np.int32)
        assert max(np.abs(tp).max(), np.abs(tm).max()) < 2**31
        ts, td = tp + tm, tp - tm
        if fold == "reduce":
            xh, xl = split_weight(gx, m, 39)
            yh, yl = split_weight(gy, m, 39)
            for t, (h, l), k in ((ts, (xh, xl), 0), (td, (yh, yl), 2)):
                tf = t.astype(np.float32)  # |t| < 2^24.3: cvt exact up to 2^24, beyond: see the assertion
                assert np.abs(t).max() < 2**24
                q = np.rint(tf * np.float32(1.0 / m))
                r = (tf - np.float32(m) * q).astype(np.float64)
                assert np.all((t - r.astype(np.int64)) % m == 0) and np.abs(r).max() <= m
                S[k] += r * h
                S[k + 1] += r * l
        else:
            for t, g, k in ((ts, gx, 0), (td, gy, 3)):
                h1, rest = split_weight(g, m, 23)
                hi_int = (g << 46) // m
                h2 = (hi_int - (((g << 23) // m) << 23)) / 2.0**46
                h3 = float(Fraction((g << 46) - hi_int * m, m) / 2**46)
                tdbl = t.astype(np.float64)
                S[k] += tdbl * h1
                S[k + 1] += tdbl * h2
                S[k + 2] += tdbl * h3
    Ph = float(P)
    Pl = float(P - int(Ph))
    out = []
    for k in (0, S.shape[0] // 2):
        if fold == "reduce":
            fh = S[k] - np.rint(S[k])
            fl = S[k + 1]
        else:
            fh = (S[k] - np.rint(S[k])) + (S[k + 1] - np.rint(S[k + 1]))
            fl = S[k + 2]
        wrap = np.rint(fh + fl)
        fh = fh - wrap
        h = Ph * fh
        # error of the product Ph * fh through the fused multiply-add identity, emulated here with exact rationals on a sample only;
        # the device uses fma(Ph, fh, -h).  numpy has no fma: split Ph
        Ph_hi = np.float64(np.float32(Ph))
        Ph_lo = Ph - Ph_hi
        fh_hi = fh.astype(np.float32).astype(np.float64)
        fh_lo = fh - fh_hi
        e = ((Ph_hi * fh_hi - h) + Ph_hi * fh_lo + Ph_lo * fh_hi) + Ph_lo * fh_lo
        out.append(h + (e + Ph * fl + Pl * fh))
    C = out[0] + 1j * out[1]
    C = np.ldexp(np.ldexp(C.real, -sA[:, None]), -sB[None, :]) + 1j * np.ldexp(np.ldexp(C.imag, -sA[:, None]), -sB[None, :])
    return C, (XA, XB, sA, sB, P)


def exact_product(XA, XB, sA, sB, rows, cols):
    """integer product with Python ints on a sample of entries"""
    K = XA.shape[1]
    out = {}
    for i, j in zip(rows, cols):
        re = im = 0
        for k in range(K):
            ar, ai, br, bi = int(XA[i, k].real), int(XA[i, k].imag), int(XB[k, j].real), int(XB[k, j].imag)
            re += ar * br - ai * bi
            im += ar * bi + ai * br
        out[(i, j)] = (re, im)
    return out


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 285
    cols = int(sys.argv[3]) if len(sys.argv) > 3 else 80
    rng = np.random.default_rng(5)
    for name, decay in (("flat spectrum", 0.0), ("modes falling 10^-6 over K", 6.0), ("10^-12 over K", 12.0)):
        w = 10.0 ** (-decay * np.arange(K) / K)
        A = (rng.normal(size=(rows, K)) + 1j * rng.normal(size=(rows, K))) * w[None, :] * 10.0 ** rng.uniform(-3, 3, size=(rows, 1))
        B = rng.normal(size=(K, cols)) + 1j * rng.normal(size=(K, cols))
        ref = (A.astype(np.clongdouble) @ B.astype(np.clongdouble))  # 64-bit mantissa reference
        plain = A @ B
        scale = (np.abs(A).sum(axis=1)[:, None] * np.abs(B).max(axis=0)[None, :]) + 1e-300
        for fold in ("reduce", "three"):
            C, (XA, XB, sA, sB, P) = rns_product(A, B, fold)
            err = np.abs(C - ref).astype(np.float64)
            # exactness of the integer part on a sample
            ii = rng.integers(0, rows, 12)
            jj = rng.integers(0, cols, 12)
            ex = exact_product(XA, XB, sA, sB, ii, jj)
            worst = 0.0
            for (i, j), (re, im) in ex.items():
                got_re = Fraction(float(C[i, j].real)) * Fraction(2) ** int(sA[i] + sB[j])
                got_im = Fraction(float(C[i, j].imag)) * Fraction(2) ** int(sA[i] + sB[j])
                for got, want in ((got_re, re), (got_im, im)):
                    if want != 0:
                        worst = max(worst, abs(float((got - want) / want)))
                    assert abs(want) < P // 2
            print(f"{name:28s} fold={fold:6s}: max |C - ref| / (|A|_1 |B|_max) = {np.max(err / scale):.2e}   plain fp64 A @ B: "
                  f"{np.max(np.abs(plain - ref).astype(np.float64) / scale):.2e}   integer product vs exact, relative: {worst:.1e}")


if __name__ == "__main__":
    main()
