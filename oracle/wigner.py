"""Wigner-D matrices, spin-weighted spherical harmonics and mode index algebra.

Restates the ``spherical_functions`` (>=2022.4, un-vendored) pieces the hot path calls
(SURVEY Appendix A.1, A.3, A.5).  Call sites in the reference:
``sf._Wigner_D_matrices`` scri/rotations.py:327,381; ``sf._linear_matrix_offset`` :359,384;
``sf.SWSH_grid`` scri/waveform_grid.py:470,471,509,514,533; ``sf.Modes.evaluate``
scri/asymptotic_bondi_data/transformations.py:187,312-334; ``sf.eth_GHP/ethbar_GHP``
scri/waveform_grid.py:488,497,508; ``sf.LM_index`` etc. scri/waveform_grid.py:49-86.

Published definition (spherical_functions documentation, "Wigner D matrices"):

  D^l_{m',m}(R) = sqrt[(l+m)!(l-m)!/((l+m')!(l-m')!)]
                  sum_rho C(l+m',rho) C(l-m',l-rho-m) (-1)^rho
                          Ra^(l+m'-rho) conj(Ra)^(l-rho-m) Rb^(rho-m'+m) conj(Rb)^rho
  sYlm(R)       = (-1)^s sqrt((2l+1)/4pi) D^l_{m,-s}(R)

Two evaluators are provided:
  * ``wigner_D_exact``  -- the literal sum in 60-digit mpmath arithmetic (ground truth, slow);
  * ``wigner_d_column`` / ``wigner_D_matrices`` / ``swsh_grid`` -- vectorised, via the standard
    three-term recurrence in l (exactly the same function; checked against the exact evaluator in
    tests/test_oracle_wigner.py).  The recurrence is carried in numpy's extended precision
    (``np.longdouble``: 64-bit mantissa on x86) and rounded to float64 at the end: in plain float64
    it loses ~l ulp, which is what kept two of the reference's hand-tuned tolerances
    (tests/test_waveform_grid.py: 4e-14, 5e-14) out of reach of the restatement; with the rounded
    extended values the oracle passes them as they stand.  ``EXTENDED = False`` restores float64.
"""
import math
import numpy as np

# ----------------------------------------------------------------------------- index algebra


def LM_index(ell, m, ell_min):
    return ell * (ell + 1) - ell_min**2 + m


def LM_total_size(ell_min, ell_max):
    return (ell_max + 1) ** 2 - ell_min**2


def LM_range(ell_min, ell_max):
    return np.array([[ell, m] for ell in range(ell_min, ell_max + 1) for m in range(-ell, ell + 1)], dtype=int)


def linear_matrix_offset(ell, ell_min):
    """Offset of the (2l+1)^2 block of D^l in the packed D array (sf._linear_matrix_offset)."""
    return (4 * ell**3 - ell) // 3 - (4 * ell_min**3 - ell_min) // 3


def total_size_D_matrices(ell_min, ell_max):
    """sf.WignerD._total_size_D_matrices  (scri/rotations.py:299)."""
    return linear_matrix_offset(ell_max + 1, ell_min)


def LMpM_index(ell, mp, m, ell_min):
    return linear_matrix_offset(ell, ell_min) + (2 * ell + 1) * (mp + ell) + (m + ell)


# ----------------------------------------------------------------------------- small helpers


def constant_as_ell_0_mode(c):
    return c * math.sqrt(4 * math.pi)


def constant_from_ell_0_mode(m):
    return m / math.sqrt(4 * math.pi)


def vector_as_ell_1_modes(v):
    v = np.asarray(v, dtype=float)
    return np.array(
        [
            (v[0] + 1j * v[1]) * math.sqrt(2 * math.pi / 3.0),
            v[2] * math.sqrt(4 * math.pi / 3.0) + 0j,
            (-v[0] + 1j * v[1]) * math.sqrt(2 * math.pi / 3.0),
        ]
    )


def vector_from_ell_1_modes(modes):
    modes = np.asarray(modes, dtype=complex)
    return np.array(
        [
            (modes[0] - modes[2]) / (2 * math.sqrt(2 * math.pi / 3.0)),
            (modes[0] + modes[2]) / (2j * math.sqrt(2 * math.pi / 3.0)),
            modes[1] / math.sqrt(4 * math.pi / 3.0),
        ]
    )


def eth_NP(modes, s, ell_min=0):
    """Newman-Penrose eth on mode weights: x sqrt((l-s)(l+s+1)); spin s -> s+1."""
    modes = np.array(modes, dtype=complex)
    n = modes.shape[-1]
    ell_max = int(round(math.sqrt(n + ell_min**2))) - 1
    out = np.zeros_like(modes)
    for ell in range(ell_min, ell_max + 1):
        f = math.sqrt((ell - s) * (ell + s + 1)) if ell >= abs(s) and ell >= abs(s + 1) else 0.0
        i0 = LM_index(ell, -ell, ell_min)
        out[..., i0 : i0 + 2 * ell + 1] = f * modes[..., i0 : i0 + 2 * ell + 1]
    return out


def ethbar_NP(modes, s, ell_min=0):
    """Newman-Penrose ethbar: x -sqrt((l+s)(l-s+1)); spin s -> s-1."""
    modes = np.array(modes, dtype=complex)
    n = modes.shape[-1]
    ell_max = int(round(math.sqrt(n + ell_min**2))) - 1
    out = np.zeros_like(modes)
    for ell in range(ell_min, ell_max + 1):
        f = -math.sqrt((ell + s) * (ell - s + 1)) if ell >= abs(s) and ell >= abs(s - 1) else 0.0
        i0 = LM_index(ell, -ell, ell_min)
        out[..., i0 : i0 + 2 * ell + 1] = f * modes[..., i0 : i0 + 2 * ell + 1]
    return out


def eth_GHP(modes, s, ell_min=0):
    return eth_NP(modes, s, ell_min) / math.sqrt(2)


def ethbar_GHP(modes, s, ell_min=0):
    return ethbar_NP(modes, s, ell_min) / math.sqrt(2)


# ----------------------------------------------------------------------------- exact D


def wigner_D_exact(Ra, Rb, ell, mp, m, dps=60):
    """Literal published sum, in mpmath arithmetic with `dps` digits.  Returns mpc."""
    import mpmath

    with mpmath.workdps(dps):
        Ra = mpmath.mpc(complex(Ra).real, complex(Ra).imag)
        Rb = mpmath.mpc(complex(Rb).real, complex(Rb).imag)
        Rac, Rbc = mpmath.conj(Ra), mpmath.conj(Rb)
        f = mpmath.factorial
        pref = mpmath.sqrt(f(ell + m) * f(ell - m) / (f(ell + mp) * f(ell - mp)))
        total = mpmath.mpc(0)
        for rho in range(max(0, mp - m), min(ell + mp, ell - m) + 1):
            term = (
                mpmath.binomial(ell + mp, rho)
                * mpmath.binomial(ell - mp, ell - rho - m)
                * (-1) ** rho
                * Ra ** (ell + mp - rho)
                * Rac ** (ell - rho - m)
                * Rb ** (rho - mp + m)
                * Rbc**rho
            )
            total += term
        return pref * total


def wigner_D_matrices_exact(Ra, Rb, ell_min, ell_max, dps=60):
    out = np.empty(total_size_D_matrices(ell_min, ell_max), dtype=complex)
    for ell in range(ell_min, ell_max + 1):
        for mp in range(-ell, ell + 1):
            for m in range(-ell, ell + 1):
                out[LMpM_index(ell, mp, m, ell_min)] = complex(wigner_D_exact(Ra, Rb, ell, mp, m, dps))
    return out


# ----------------------------------------------------------------------------- D via l-recurrence

EXTENDED = True  # carry the recurrences in np.longdouble, round to float64 at the end


def _real_t():
    return np.longdouble if EXTENDED else np.float64


def _cplx_t():
    return np.clongdouble if EXTENDED else np.complex128



def _sqrt_int(n):
    """sqrt of a non-negative integer in the working precision."""
    return np.sqrt(_real_t()(n))


def _d_start(ell0, mp, m, ra, rb):
    """d^{l0}_{mp,m}(ra=cos(b/2), rb=sin(b/2)) for l0 = max(|mp|,|m|): single-term closed forms."""
    if ell0 == mp:
        return (-1.0) ** (ell0 - m) * _sqrt_int(math.comb(2 * ell0, ell0 - m)) * ra ** (ell0 + m) * rb ** (ell0 - m)
    if ell0 == -mp:
        return _sqrt_int(math.comb(2 * ell0, ell0 + m)) * ra ** (ell0 - m) * rb ** (ell0 + m)
    if ell0 == m:
        return _sqrt_int(math.comb(2 * ell0, ell0 - mp)) * ra ** (ell0 + mp) * rb ** (ell0 - mp)
    # ell0 == -m
    return (-1.0) ** (ell0 + mp) * _sqrt_int(math.comb(2 * ell0, ell0 + mp)) * ra ** (ell0 - mp) * rb ** (ell0 + mp)


def wigner_d_chain(mp, m, ra, rb, ell_max):
    """Real Wigner small-d  d^l_{mp,m}  for l = 0..ell_max (zeros below max(|mp|,|m|)).

    ra, rb are arrays (|Ra|, |Rb|) of any common shape; result has shape ra.shape + (ell_max+1,).
    Three-term recurrence in l (Varshalovich 4.8.2 (16)):
      l sqrt((l+1)^2-mp^2) sqrt((l+1)^2-m^2) d^{l+1}
         = (2l+1) [l(l+1) cos(b) - mp m] d^l - (l+1) sqrt(l^2-mp^2) sqrt(l^2-m^2) d^{l-1}
    """
    ra = np.asarray(ra, dtype=_real_t())
    rb = np.asarray(rb, dtype=_real_t())
    out = np.zeros(ra.shape + (ell_max + 1,), dtype=_real_t())
    ell0 = max(abs(mp), abs(m))
    if ell0 > ell_max:
        return out
    # Well-conditioned form of cos(b): cos(b) = sigma (1 - 2 t), t = min(ra, rb)^2, so that
    # l(l+1) cos(b) - mp m = sigma [(l(l+1) - sigma mp m) - 2 l(l+1) t] keeps full relative accuracy
    # near the poles (forming ra^2 - rb^2 loses l^2 eps / 2 there).
    sig = np.where(ra >= rb, 1.0, -1.0).astype(_real_t())
    t = np.where(ra >= rb, rb * rb, ra * ra)
    dm1 = np.zeros(ra.shape, dtype=_real_t())
    d0 = _d_start(ell0, mp, m, ra, rb)
    out[..., ell0] = d0
    for ell in range(ell0, ell_max):
        if ell == 0:
            d1 = sig * (1.0 - 2.0 * t) * d0
        else:
            c1 = (2 * ell + 1) * sig * ((ell * (ell + 1) - sig * (mp * m)) - (2 * ell * (ell + 1)) * t)
            c2 = (ell + 1) * _sqrt_int((ell * ell - mp * mp) * (ell * ell - m * m))
            den = ell * _sqrt_int(((ell + 1) ** 2 - mp * mp) * ((ell + 1) ** 2 - m * m))
            d1 = (c1 * d0 - c2 * dm1) / den
        out[..., ell + 1] = d1
        dm1, d0 = d0, d1
    return out


def _polar(Ra, Rb):
    Ra = np.asarray(Ra, dtype=complex).astype(_cplx_t())
    Rb = np.asarray(Rb, dtype=complex).astype(_cplx_t())
    ra, rb = np.abs(Ra), np.abs(Rb)
    n = np.sqrt(ra * ra + rb * rb)
    # unit phases, with the convention phase(0) = 1
    ea = np.where(ra > 0, Ra / np.where(ra > 0, ra, 1.0), 1.0)
    eb = np.where(rb > 0, Rb / np.where(rb > 0, rb, 1.0), 1.0)
    return ra / n, rb / n, ea, eb


def wigner_D_matrices(Ra, Rb, ell_min, ell_max):
    """Packed D matrices, layout of sf._Wigner_D_matrices: for l, row-major (m', m).

    Ra, Rb: arrays of common shape S.  Returns complex array S + (total_size,).
    D^l_{m',m} = ea^(m'+m) eb^(m-m') d^l_{m',m}(ra, rb).
    """
    ra, rb, ea, eb = _polar(Ra, Rb)
    out = np.zeros(ra.shape + (total_size_D_matrices(ell_min, ell_max),), dtype=complex)
    for mp in range(-ell_max, ell_max + 1):
        for m in range(-ell_max, ell_max + 1):
            d = wigner_d_chain(mp, m, ra, rb, ell_max)
            phase = ea ** (mp + m) * eb ** (m - mp)
            for ell in range(max(ell_min, abs(mp), abs(m)), ell_max + 1):
                out[..., LMpM_index(ell, mp, m, ell_min)] = (phase * d[..., ell]).astype(complex)
    return out


_PI = np.longdouble("3.14159265358979323846264338327950288")


def swsh_grid(R, s, ell_max):
    """sf.SWSH_grid(R, s, ell_max): sYlm at each rotor, shape R.shape[:-1] + ((ell_max+1)^2,),
    modes from l=0 with zeros for l<|s|.   R: float array [..., 4] (w,x,y,z)."""
    R = np.asarray(R, dtype=float)
    Ra = R[..., 0] + 1j * R[..., 3]
    Rb = R[..., 2] + 1j * R[..., 1]
    ra, rb, ea, eb = _polar(Ra, Rb)
    out = np.zeros(ra.shape + ((ell_max + 1) ** 2,), dtype=complex)
    sign = (-1.0) ** s
    for m in range(-ell_max, ell_max + 1):
        d = wigner_d_chain(m, -s, ra, rb, ell_max)  # D^l_{m,-s}
        phase = ea ** (m - s) * eb ** (-s - m)
        for ell in range(max(abs(m), abs(s)), ell_max + 1):
            norm = np.sqrt(_real_t()(2 * ell + 1) / (4 * _PI))
            out[..., LM_index(ell, m, 0)] = (sign * norm * phase * d[..., ell]).astype(complex)
    return out


def modes_evaluate(modes, R, s):
    """sf.Modes(modes, spin_weight=s).evaluate(R): sum_lm f_lm sYlm(R); modes[..., (L+1)^2] from l=0.
    Result shape modes.shape[:-1] + R.shape[:-1]."""
    modes = np.asarray(modes, dtype=complex)
    ell_max = int(round(math.sqrt(modes.shape[-1]))) - 1
    Y = swsh_grid(R, s, ell_max)
    return np.tensordot(modes, Y, axes=([-1], [-1]))
