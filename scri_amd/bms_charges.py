"""BMS charges of AsymptoticBondiData (scri/asymptotic_bondi_data/bms_charges.py:14-286), the step after the
transformation path.  Products of fields run on the GPU (ModesTimeSeries.multiply ->
bms_grid_multiply on a grid that is exact for l_a + l_b), time derivatives through bms_spline_derivative; what is left
here is the bookkeeping of the formulas.  The functions are grafted onto AsymptoticBondiData as methods.
"""
from math import sqrt

import numpy as np


def mass_aspect(self, truncate_ell=max):
    """Bondi mass aspect M = -Re{psi2 + sigma d_t sigma-bar}  (bms_charges.py:14-47).

    truncate_ell: an int truncates every term to that ell_max (terms that are not needed are not computed); a callable is
    used as the truncator of the product (default `max`: the larger ell_max of the two factors); a false value leaves the
    choice to the series' own multiplication_truncator (the reference's plain `sigma * sigma.bar.dot`: `max` for every
    AsymptoticBondiData built by file_io, from_initial_values and map_to_superrest_frame, `sum` for a bare one)."""
    # one formula, three ways of choosing the band limit of its two terms
    if callable(truncate_ell):
        product_ell, psi2 = truncate_ell, self.psi2
    elif truncate_ell:
        product_ell, psi2 = (lambda ells: truncate_ell), self.psi2.truncate_ell(truncate_ell)
    else:
        product_ell, psi2 = None, self.psi2  # (None: the truncator the series carries)
    news_term = self.sigma.multiply(self.sigma.bar.dot, truncator=product_ell)
    return -((psi2 + news_term).real)


# Rows: (t, x, y, z); columns: (Re a_00, Re a_1-1, Im a_1-1, Re a_10, Re a_11, Im a_11).  The real part of an l <= 1 function against
# (1, sin th cos ph, sin th sin ph, cos th) over the sphere, divided by 4 pi: Y_00 = 1/sqrt(4 pi), Y_10 = sqrt(3/4pi) cos th,
# Y_1+-1 = -+ sqrt(3/8pi) sin th e^{+-i ph}.
_ASPECT_TO_VECTOR = np.array(
    [
        [1.0, 0.0, 0.0, 0.0, 0.0, 0.0],
        [0.0, 1.0 / sqrt(6), 0.0, 0.0, -1.0 / sqrt(6), 0.0],
        [0.0, 0.0, 1.0 / sqrt(6), 0.0, 0.0, 1.0 / sqrt(6)],
        [0.0, 0.0, 0.0, 1.0 / sqrt(3), 0.0, 0.0],
    ]
) / sqrt(4 * np.pi)


def charge_vector_from_aspect(charge):
    """l <= 1 modes of a charge aspect as a four-vector: v = (1/4pi) int Re{a} (1, sin th cos ph, sin th sin ph, cos th)
    (bms_charges.py:50-69), as ONE real 4 x 6 matrix applied to (Re a_00, Re a_1-1, Im a_1-1, Re a_10, Re a_11, Im a_11)."""
    a = np.asarray(charge)[..., :4]
    parts = np.stack([a[..., 0].real, a[..., 1].real, a[..., 1].imag, a[..., 2].real, a[..., 3].real, a[..., 3].imag], axis=-1)
    return parts @ _ASPECT_TO_VECTOR.T


def _minkowski_norm_squared(P):
    """E^2 - |p|^2 of four-vectors along the last axis"""
    return P[..., 0] ** 2 - np.einsum("...i,...i->...", P[..., 1:], P[..., 1:])


def bondi_rest_mass(self):
    """Rest mass of the Bondi four-momentum (bms_charges.py:72-76)"""
    return np.sqrt(_minkowski_norm_squared(self.bondi_four_momentum()))


_DIPOLE = 1  # the charges below are the l <= 1 part of their aspects


def _four_vector(aspect):
    return charge_vector_from_aspect(aspect.ndarray)


def _three_vector(aspect):
    return _four_vector(aspect)[:, 1:]


def _dipole_product(a, b):
    """a b truncated to l <= 1 (a grid product on a grid that is exact for l_a + l_b; only the l <= 1 modes are kept)"""
    return a.multiply(b, truncator=lambda ells: _DIPOLE)


def bondi_four_momentum(self):
    """l < 2 part of the mass aspect as a four-vector (bms_charges.py:79-88)"""
    return _four_vector(self.mass_aspect(_DIPOLE))


def _psi1_sigma_term(self, ell_max):
    """psi1 + sigma eth sigma-bar, truncated (the common part of the angular-momentum, boost and CoM aspects)"""
    return self.psi1.truncate_ell(ell_max) + self.sigma.multiply(self.sigma.bar.eth_GHP, truncator=lambda ells: ell_max)


def bondi_angular_momentum(self):
    """Total Bondi angular momentum vector from i (psi1 + sigma eth sigma-bar)  (bms_charges.py:91-106)"""
    return _three_vector(1j * _psi1_sigma_term(self, _DIPOLE))


def _com_aspect(self):
    """- [psi1 + sigma eth sigma-bar + (1/2) eth(sigma sigma-bar)], l <= 1: the aspect of G = N + t P"""
    shear_norm = _dipole_product(self.sigma, self.sigma.bar)
    return -(_psi1_sigma_term(self, _DIPOLE) + 0.5 * shear_norm.eth_GHP)


def bondi_boost_charge(self):
    """N = G - t P as an aspect: - [psi1 + sigma eth sigma-bar + (1/2) eth(sigma sigma-bar) - t eth Re{psi2 + sigma d_t sigma-bar}]
    (bms_charges.py:163-182)"""
    energy_density = (self.psi2.truncate_ell(_DIPOLE) + _dipole_product(self.sigma, self.sigma.bar.dot)).real
    return _three_vector(_com_aspect(self) + self.t[:, np.newaxis] * energy_density.eth_GHP)


def bondi_CoM_charge(self):
    """G = N + t P = - [psi1 + sigma eth sigma-bar + (1/2) eth(sigma sigma-bar)]  (bms_charges.py:185-200)"""
    return _three_vector(_com_aspect(self))


def bondi_dimensionless_spin(self):
    """Dimensionless Bondi spin vector chi = [gamma (J + v x N) - (gamma - 1) (J . v^) v^] / M^2 with v = p / E
    (bms_charges.py:139-160), as one expression on (P, J, N)."""
    P, J, N = self.bondi_four_momentum(), self.bondi_angular_momentum(), self.bondi_boost_charge()
    velocity = P[:, 1:] / P[:, :1]
    speed = np.sqrt(np.einsum("ti,ti->t", velocity, velocity))[:, np.newaxis]
    direction = np.divide(velocity, speed, out=velocity.copy(), where=speed != 0)  # (a vanishing velocity has no direction: kept)
    lorentz = 1.0 / np.sqrt(1.0 - speed**2)
    along = np.einsum("ti,ti->t", J, direction)[:, np.newaxis] * direction
    return (lorentz * (J + np.cross(velocity, N)) - (lorentz - 1.0) * along) / _minkowski_norm_squared(P)[:, np.newaxis]


def CWWY_angular_momentum(self):
    """Chen/Wang/Wang/Yau angular momentum vector, Eq. (5) of arXiv:2102.03235 (bms_charges.py:109-136): the Bondi
    expression plus the supertranslation potential D^-1 (ethbar^2 sigma + eth^2 sigma-bar) times eth of the mass aspect."""
    from .map_to_superrest_frame import D_inverse
    from .modes_time_series import ModesTimeSeries

    potential = self.sigma.ethbar_GHP.ethbar_GHP + self.sigma.bar.eth_GHP.eth_GHP
    if hasattr(potential, "scale_by_ell"):  # device-resident series: the diagonal D^-1 as one more mode map
        potential = potential.scale_by_ell(lambda l: 0.0 if l < 2 else 4.0 / ((l + 2) * (l + 1) * l * (l - 1)))
    else:
        potential = ModesTimeSeries(D_inverse(potential.ndarray, self.ell_max), self.t, spin_weight=0, ell_min=0, ell_max=self.ell_max)
    correction = _dipole_product(potential, self.mass_aspect().eth_GHP)
    return _three_vector(1j * (_psi1_sigma_term(self, _DIPOLE) + correction))


def supermomentum(self, supermomentum_def, **kwargs):
    """Supermomentum Psi = psi2 + sigma d_t sigma-bar + f  (bms_charges.py:203-286);
    f = 0 ('Bondi-Sachs'/'BS'), eth^2 sigma-bar ('Moreschi'/'M'), (eth^2 sigma-bar - ethbar^2 sigma)/2 ('Geroch'/'G'),
    -ethbar^2 sigma ('Geroch-Winicour'/'GW').  integrated=True returns -Psi-bar / (2 sqrt(pi)).
    working_ell_max / output_ell_max are passed to grid_multiply."""
    return_integrated = kwargs.pop("integrated", False)
    name = supermomentum_def.lower()
    if name not in ("bondi-sachs", "bs", "moreschi", "m", "geroch", "g", "geroch-winicour", "gw"):
        raise ValueError(
            f"Supermomentum defintion '{supermomentum_def}' not recognized. Please choose one of "
            "the following options:\n"
            "  * 'Bondi-Sachs' or 'BS'\n"
            "  * 'Moreschi' or 'M'\n"
            "  * 'Geroch' or 'G'\n"
            "  * 'Geroch-Winicour' or 'GW'"
        )
    base = self.psi2 + self.sigma.grid_multiply(self.sigma.bar.dot, **kwargs)
    if name in ("bondi-sachs", "bs"):
        result = base
    elif name in ("moreschi", "m"):
        result = base + self.sigma.bar.eth_GHP.eth_GHP
    elif name in ("geroch", "g"):
        result = base + 0.5 * (self.sigma.bar.eth_GHP.eth_GHP - self.sigma.ethbar_GHP.ethbar_GHP)
    else:
        result = base - self.sigma.ethbar_GHP.ethbar_GHP
    if return_integrated:
        return -0.5 * result.bar / np.sqrt(np.pi)
    return result


METHODS = (
    mass_aspect, bondi_rest_mass, bondi_four_momentum, bondi_angular_momentum, bondi_boost_charge, bondi_CoM_charge,
    bondi_dimensionless_spin, CWWY_angular_momentum, supermomentum,
)
