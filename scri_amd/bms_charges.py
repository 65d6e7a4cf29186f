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
    used as the truncator of the product (default `max`: the larger ell_max of the two factors); a false value keeps
    the full product."""
    if callable(truncate_ell):
        return -(self.psi2 + self.sigma.multiply(self.sigma.bar.dot, truncator=truncate_ell)).real
    elif truncate_ell:
        return -(
            self.psi2.truncate_ell(truncate_ell) + self.sigma.multiply(self.sigma.bar.dot, truncator=lambda tup: truncate_ell)
        ).real
    else:
        return -(self.psi2 + self.sigma * self.sigma.bar.dot).real


def charge_vector_from_aspect(charge):
    """l <= 1 modes of a charge aspect as a four-vector: v = (1/4pi) int Re{a} (1, sin th cos ph, sin th sin ph, cos th)
    (bms_charges.py:50-69)."""
    charge = np.asarray(charge)
    four_vector = np.empty(charge.shape[:-1] + (4,), dtype=float)
    four_vector[..., 0] = charge[..., 0].real
    four_vector[..., 1] = (charge[..., 1] - charge[..., 3]).real / sqrt(6)
    four_vector[..., 2] = (charge[..., 1] + charge[..., 3]).imag / sqrt(6)
    four_vector[..., 3] = charge[..., 2].real / sqrt(3)
    return four_vector / np.sqrt(4 * np.pi)


def bondi_rest_mass(self):
    """Rest mass of the Bondi four-momentum (bms_charges.py:72-76)"""
    four_momentum = self.bondi_four_momentum()
    return np.sqrt(four_momentum[:, 0] ** 2 - np.sum(four_momentum[:, 1:] ** 2, axis=1))


def bondi_four_momentum(self):
    """l < 2 part of the mass aspect as a four-vector (bms_charges.py:79-88)"""
    ell_max = 1
    charge_aspect = self.mass_aspect(ell_max).ndarray
    return charge_vector_from_aspect(charge_aspect)


def _psi1_sigma_term(self, ell_max):
    """psi1 + sigma eth sigma-bar, truncated (the common part of the angular-momentum, boost and CoM aspects)"""
    return self.psi1.truncate_ell(ell_max) + self.sigma.multiply(self.sigma.bar.eth_GHP, truncator=lambda tup: ell_max)


def bondi_angular_momentum(self):
    """Total Bondi angular momentum vector from i (psi1 + sigma eth sigma-bar)  (bms_charges.py:91-106)"""
    ell_max = 1
    charge_aspect = (1j * _psi1_sigma_term(self, ell_max)).ndarray
    return charge_vector_from_aspect(charge_aspect)[:, 1:]


def bondi_boost_charge(self):
    """- [psi1 + sigma eth sigma-bar + (1/2) eth(sigma sigma-bar) - t eth Re{psi2 + sigma d_t sigma-bar}]
    (bms_charges.py:163-182)"""
    ell_max = 1
    mass_term = (
        self.psi2.truncate_ell(ell_max) + self.sigma.multiply(self.sigma.bar.dot, truncator=lambda tup: ell_max)
    ).real.eth_GHP
    charge_aspect = -(
        _psi1_sigma_term(self, ell_max)
        + 0.5 * self.sigma.multiply(self.sigma.bar, truncator=lambda tup: ell_max).eth_GHP
        - self.t[:, np.newaxis] * mass_term
    ).ndarray
    return charge_vector_from_aspect(charge_aspect)[:, 1:]


def bondi_CoM_charge(self):
    """G = N + t P = - [psi1 + sigma eth sigma-bar + (1/2) eth(sigma sigma-bar)]  (bms_charges.py:185-200)"""
    ell_max = 1
    charge_aspect = -(
        _psi1_sigma_term(self, ell_max) + 0.5 * self.sigma.multiply(self.sigma.bar, truncator=lambda tup: ell_max).eth_GHP
    ).ndarray
    return charge_vector_from_aspect(charge_aspect)[:, 1:]


def bondi_dimensionless_spin(self):
    """Dimensionless Bondi spin vector (bms_charges.py:139-160)"""
    N = self.bondi_boost_charge()
    J = self.bondi_angular_momentum()
    P = self.bondi_four_momentum()
    M_sqr = (P[:, 0] ** 2 - np.sum(P[:, 1:] ** 2, axis=1))[:, np.newaxis]
    v = P[:, 1:] / (P[:, 0])[:, np.newaxis]
    v_norm = np.linalg.norm(v, axis=1)
    vhat = v.copy()
    t_idx = v_norm != 0  # normalise only where the velocity does not vanish
    vhat[t_idx] = v[t_idx] / v_norm[t_idx, np.newaxis]
    gamma = (1 / np.sqrt(1 - v_norm**2))[:, np.newaxis]
    J_dot_vhat = np.einsum("ij,ij->i", J, vhat)[:, np.newaxis]
    return (gamma * (J + np.cross(v, N)) - (gamma - 1) * J_dot_vhat * vhat) / M_sqr


def CWWY_angular_momentum(self):
    """Chen/Wang/Wang/Yau angular momentum vector, Eq. (5) of arXiv:2102.03235 (bms_charges.py:109-136): the Bondi
    expression plus the supertranslation potential D^-1 (ethbar^2 sigma + eth^2 sigma-bar) times eth of the mass aspect."""
    from .map_to_superrest_frame import D_inverse
    from .modes_time_series import ModesTimeSeries

    ell_max = 1
    potential = self.sigma.ethbar_GHP.ethbar_GHP + self.sigma.bar.eth_GHP.eth_GHP
    if hasattr(potential, "scale_by_ell"):  # device-resident series: the diagonal D^-1 as one more mode map
        potential = potential.scale_by_ell(lambda l: 0.0 if l < 2 else 4.0 / ((l + 2) * (l + 1) * l * (l - 1)))
    else:
        potential = ModesTimeSeries(D_inverse(potential.ndarray, self.ell_max), self.t, spin_weight=0, ell_min=0, ell_max=self.ell_max)
    charge_aspect = (
        1j * (_psi1_sigma_term(self, ell_max) + potential.multiply(self.mass_aspect().eth_GHP, truncator=lambda tup: ell_max))
    ).ndarray
    return charge_vector_from_aspect(charge_aspect)[:, 1:]


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
            "the following options:\\n"
            "  * 'Bondi-Sachs' or 'BS'\\n"
            "  * 'Moreschi' or 'M'\\n"
            "  * 'Geroch' or 'G'\\n"
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
