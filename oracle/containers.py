"""Minimal containers + type tables for the oracle (test infrastructure only).

Follows scri/__init__.py:78-86 (type tables), scri/waveform_base.py:222-231,440-446 (fields,
spin/conformal weight), scri/waveform_base.py:950-967 (interpolate),
scri/asymptotic_bondi_data/__init__.py:36-75 (ABD storage [6, N, (lmax+1)^2], order
psi0, psi1, psi2, psi3, psi4, sigma).
"""
import sys
from dataclasses import dataclass, field, replace
import numpy as np
from scipy.interpolate import CubicSpline

FrameType = [UnknownFrameType, Inertial, Coprecessing, Coorbital, Corotating] = range(5)
DataType = [UnknownDataType, psi0, psi1, psi2, psi3, psi4, sigma, h, hdot, news, psin, psim] = range(12)
DataNames = ["UnknownDataType", "Psi0", "Psi1", "Psi2", "Psi3", "Psi4", "sigma", "h", "hdot", "news", "psin", "PsiM"]
SpinWeights = [sys.maxsize, 2, 1, 0, -1, -2, 2, -2, -2, -2, sys.maxsize, 0]
ConformalWeights = [sys.maxsize, 2, 1, 0, -1, -2, 1, 0, -1, -1, -3, 0]
RScaling = [sys.maxsize, 5, 4, 3, 2, 1, 2, 1, 1, 1, 0, 0]
MScaling = [sys.maxsize, 2, 2, 2, 2, 2, 0, 0, 1, 1, 2, 1]


@dataclass
class WM:
    """Stand-in for scri.WaveformModes (only what the hot path reads/writes)."""

    t: np.ndarray
    data: np.ndarray
    ell_min: int
    ell_max: int
    dataType: int = h
    frameType: int = Inertial
    r_is_scaled_out: bool = True
    m_is_scaled_out: bool = True
    frame: np.ndarray = field(default_factory=lambda: np.zeros((0, 4)))

    @property
    def n_times(self):
        return self.t.shape[0]

    @property
    def spin_weight(self):
        return SpinWeights[self.dataType]

    @property
    def conformal_weight(self):  # waveform_base.py:445-446
        return ConformalWeights[self.dataType] + (-RScaling[self.dataType] if self.r_is_scaled_out else 0)

    def copy(self):
        return replace(self, t=self.t.copy(), data=self.data.copy(), frame=np.array(self.frame, copy=True))

    def interpolate(self, tprime):  # waveform_base.py:950-967 (frame squad not restated: Inertial, empty frame)
        return replace(self, t=np.array(tprime, copy=True), data=CubicSpline(self.t, self.data)(tprime))


@dataclass
class ABD:
    """Stand-in for scri.AsymptoticBondiData: raw[6, N, (lmax+1)^2] = psi0..psi4, sigma."""

    u: np.ndarray
    raw: np.ndarray
    ell_max: int

    spins = (2, 1, 0, -1, -2, 2)

    @property
    def t(self):
        return self.u

    @property
    def n_times(self):
        return self.u.shape[0]

    def interpolate(self, tprime):  # asymptotic_bondi_data/__init__.py:218-233
        return ABD(np.array(tprime, copy=True), CubicSpline(self.u, self.raw, axis=1)(tprime), self.ell_max)
