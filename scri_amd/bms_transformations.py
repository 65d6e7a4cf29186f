"""Algebra of BMS transformations: ``LorentzTransformation`` and ``BMSTransformation`` with composition,
inversion and re-ordering, and ``transform_supertranslation`` (scri/bms_transformations.py:10-592).

The SL(2,C) bookkeeping is scalar work and stays on the host.  ``transform_supertranslation`` (the
one-slice grid transform S' = L^-1 S L, scri/bms_transformations.py:151-180) runs through the same engine
entry points as the waveform transforms: ``bms_rotor_grid`` (boosted grid), ``bms_swsh_grid`` (sYlm at the
grid rotors, HIP) and ``bms_map2salm`` (quadrature GEMM, HIP).

Re-ordering follows one rule instead of the reference's case table (same results): with the elements applied
left to right, a supertranslation that comes after Lorentz elements `Lam` equals the supertranslation
``transform_supertranslation(S, Lam)`` placed before them, and conversely with ``Lam.inverse()``.
"""
import copy

import numpy as np
import scipy.linalg

from . import engine
from .asymptotic_bondi_data import _process_transformation_kwargs

NORMAL_ORDER = ["supertranslation", "frame_rotation", "boost_velocity"]
_LORENTZ = ("frame_rotation", "boost_velocity")


class Rotor(np.ndarray):
    """Unit quaternion (w, x, y, z) as a float array with the `.components` accessor of np.quaternion."""

    def __new__(cls, q):
        return np.asarray(q, dtype=float).reshape(4).view(cls)

    @property
    def components(self):
        return np.asarray(self)


# ------------------------------------------------------------------------------------------------ SL(2,C)


_PAULI = np.array([[[0, 1], [1, 0]], [[0, -1j], [1j, 0]], [[1, 0], [0, -1]]], dtype=complex)
_SIGMA_Y_FLIP = np.array([1.0, -1.0, 1.0])


def fourvec_to_spin_matrix(fourvec):
    """Inner product of a four vector with the Pauli matrices (Penrose & Rindler vol. 1, eq. 1.2.39);
    scri/bms_transformations.py:10-24."""
    half = fourvec[0] / 2
    c, s = np.cos(half), np.sin(half)
    return np.array(
        [
            [c + 1j * fourvec[3] * s, (-fourvec[2] + 1j * fourvec[1]) * s],
            [(fourvec[2] + 1j * fourvec[1]) * s, c - 1j * fourvec[3] * s],
        ]
    )


def Lorentz_to_spin_matrix(lorentz):
    """scri/bms_transformations.py:27-56."""
    q = np.asarray(lorentz.frame_rotation.components, dtype=float)
    vec_norm = np.linalg.norm(q[1:])
    psi = 2 * np.arctan2(vec_norm, q[0])
    axis = q[1:] / vec_norm if psi != 0 else np.zeros(3)
    beta = np.linalg.norm(lorentz.boost_velocity)
    chi = 1j * np.arctanh(beta)
    vhat = lorentz.boost_velocity / beta if chi != 0 else np.zeros(3)
    rot = fourvec_to_spin_matrix([psi, *axis])
    boost = fourvec_to_spin_matrix([chi, *vhat])
    if lorentz.order.index("frame_rotation") < lorentz.order.index("boost_velocity"):
        return boost @ rot
    return rot @ boost


def pure_spin_matrix_to_Lorentz(A, is_rotation=None, tol=1e-14):
    """A unitary (rotation) or Hermitian (boost) spin matrix -> rotor / velocity; scri/bms_transformations.py:59-111."""
    # log A = sum_k c_k sigma_k with c_k = tr(log A sigma_k) / 2.  The spin matrices of fourvec_to_spin_matrix are
    # exp(i psi/2 (n_x sigma_x - n_y sigma_y + n_z sigma_z)) for a rotation and the same with psi -> i artanh(beta) for a boost, so the
    # rotation generator sits in the imaginary parts of c and the boost generator in minus the real parts, each with the sign of
    # the sigma_y component flipped.
    c = np.einsum("ij,kji->k", scipy.linalg.logm(A), _PAULI) / 2
    nvec_re = _SIGMA_Y_FLIP * c.imag
    nvec_im = -_SIGMA_Y_FLIP * c.real
    if is_rotation is None:
        if np.linalg.norm(nvec_im) < tol:
            is_rotation = True
        elif np.linalg.norm(nvec_re) < tol:
            is_rotation = False
        else:
            raise ValueError("spin matrix is neither a pure rotation nor a pure boost")
    nvec = nvec_re if is_rotation else nvec_im
    psi = 2 * np.linalg.norm(nvec)
    nhat = nvec / np.linalg.norm(nvec) if psi != 0 else np.array([0.0, 0.0, 1.0])
    if is_rotation:
        return Rotor([np.cos(psi / 2), *(nhat * np.sin(psi / 2))])
    return np.tanh(psi) * nhat


def spin_matrix_to_Lorentz(A, output_order=("frame_rotation", "boost_velocity"), ell_max=12):
    """Polar decomposition by SVD into a unitary (rotation) and a Hermitian (boost) factor;
    scri/bms_transformations.py:114-148."""
    output_order = list(output_order)
    if np.allclose(A, np.zeros_like(A)):
        return LorentzTransformation(ell_max=ell_max)
    u, s, vh = np.linalg.svd(A)
    i_rot = output_order.index("frame_rotation") if "frame_rotation" in output_order else np.inf
    i_boost = output_order.index("boost_velocity") if "boost_velocity" in output_order else np.inf
    rot = u @ vh
    boost = u @ np.diag(s) @ u.conj().T if i_rot < i_boost else vh.conj().T @ np.diag(s) @ vh
    frame_rotation = pure_spin_matrix_to_Lorentz(rot, is_rotation=True)
    boost_velocity = pure_spin_matrix_to_Lorentz(boost, is_rotation=False)
    return LorentzTransformation(
        frame_rotation=frame_rotation.components, boost_velocity=boost_velocity, order=output_order, ell_max=ell_max
    )


# ------------------------------------------------------------------------------------------------ supertranslations


def transform_supertranslation(S, lorentz, ell_max=None, ctx=None):
    """S' = L^-1 S L times the conformal factor: the supertranslation that appears when S is commuted through
    the Lorentz transformation `lorentz` (scri/bms_transformations.py:151-180)."""
    if ell_max is None:
        ell_max = lorentz.ell_max
    n_theta = 2 * ell_max + 1
    S = np.asarray(S, dtype=complex)
    linv = lorentz.inverse(output_order=["frame_rotation", "boost_velocity"])
    rotors = engine.rotor_grid(linv.frame_rotation.components, linv.boost_velocity, n_theta, n_theta, ctx=ctx)
    # k = 1 / (gamma (1 - v.r)), r = R z R^-1  (conformal_factors, transformations.py:151-196)
    w, x, y, z = np.moveaxis(rotors, -1, 0)
    rhat = np.stack([2 * (x * z + w * y), 2 * (y * z - w * x), w * w - x * x - y * y + z * z], axis=-1)
    v = np.asarray(linv.boost_velocity, dtype=float)
    gamma = 1 / np.sqrt(1 - np.dot(v, v))
    k = 1.0 / (gamma * (1 - rhat @ v))
    lS = int(round(np.sqrt(S.size))) - 1
    Y = engine.swsh_grid(rotors, 0, 0, lS, ctx=ctx)  # [n_theta, n_phi, (lS+1)^2]
    values = (k * (Y @ S)).real
    return engine.map2salm(values.astype(complex), 0, ell_max, ctx=ctx)


# ------------------------------------------------------------------------------------------------ Lorentz


class LorentzTransformation:
    """frame_rotation (unit quaternion), boost_velocity and the order in which they are applied
    (scri/bms_transformations.py:183-266)."""

    def __init__(self, **kwargs):
        self.ell_max = copy.deepcopy(kwargs.pop("ell_max", 12))
        frame_rotation, boost_velocity, _, _, _ = _process_transformation_kwargs(self.ell_max, **kwargs)
        self.frame_rotation = Rotor(frame_rotation)
        self.boost_velocity = np.array(boost_velocity, dtype=float)
        self.order = [x for x in copy.deepcopy(kwargs.pop("order", list(_LORENTZ))) if x != "supertranslation"]
        for name in _LORENTZ:
            if name not in self.order:
                self.order.append(name)

    def __repr__(self):
        vals = {"frame_rotation": self.frame_rotation.components, "boost_velocity": self.boost_velocity}
        return "LorentzTransformation(\n" + "".join(f"\t{k}={vals[k]}\n" for k in self.order) + ")"

    def copy(self):
        return LorentzTransformation(
            frame_rotation=self.frame_rotation.components, boost_velocity=self.boost_velocity, order=self.order, ell_max=self.ell_max
        )

    def reorder(self, output_order):
        if not ("frame_rotation" in output_order and "boost_velocity" in output_order):
            raise ValueError("Not enough transformations")
        lorentz_order = [x for x in output_order if x in _LORENTZ]
        if self.order == lorentz_order:
            return self.copy()
        # (the reference drops ell_max here and falls back to 12; it is carried along instead)
        return spin_matrix_to_Lorentz(Lorentz_to_spin_matrix(self), output_order=lorentz_order, ell_max=self.ell_max)

    def inverse(self, output_order=None):
        if output_order is None:
            output_order = self.order[::-1]
        return spin_matrix_to_Lorentz(
            np.linalg.inv(Lorentz_to_spin_matrix(self)), output_order=[x for x in output_order if x in _LORENTZ], ell_max=self.ell_max
        )

    def is_close_to(self, other):
        return np.allclose(self.frame_rotation.components, other.frame_rotation.components) and np.allclose(
            self.boost_velocity, other.boost_velocity
        )

    def __mul__(self, other):
        """`other` applied after `self` (passive transformations), output order rotation then boost."""
        return spin_matrix_to_Lorentz(
            Lorentz_to_spin_matrix(other) @ Lorentz_to_spin_matrix(self),
            output_order=["frame_rotation", "boost_velocity"],
            ell_max=max(self.ell_max, other.ell_max),
        )


# ------------------------------------------------------------------------------------------------ BMS


class BMSTransformation:
    """supertranslation modes (l <= ell_max), frame_rotation, boost_velocity and their order of application
    (scri/bms_transformations.py:269-592)."""

    def __init__(self, **kwargs):
        self.ell_max = copy.deepcopy(kwargs.pop("ell_max", 12))
        self._ctx = kwargs.pop("ctx", None)
        frame_rotation, boost_velocity, supertranslation, _, _ = _process_transformation_kwargs(self.ell_max, **kwargs)
        self.frame_rotation = Rotor(frame_rotation)
        self.boost_velocity = np.array(boost_velocity, dtype=float)
        self.supertranslation = np.pad(supertranslation, (0, (self.ell_max + 1) ** 2 - supertranslation.size))
        self.order = copy.deepcopy(kwargs.pop("order", list(NORMAL_ORDER)))
        for name in NORMAL_ORDER:
            if name not in self.order:
                self.order.append(name)

    def __repr__(self):
        vals = {
            "frame_rotation": self.frame_rotation.components,
            "boost_velocity": self.boost_velocity,
            "supertranslation": self.supertranslation[:9],
        }
        return "BMSTransformation(\n" + "".join(f"\t{k}={vals[k]}\n" for k in self.order) + ")"

    def copy(self):
        return BMSTransformation(
            frame_rotation=self.frame_rotation.components,
            boost_velocity=self.boost_velocity,
            supertranslation=self.supertranslation,
            order=self.order,
            ell_max=self.ell_max,
            ctx=self._ctx,
        )

    def _lorentz(self, names=_LORENTZ, order=None, values=None):
        src = self if values is None else values
        kw = dict(ell_max=self.ell_max, order=[x for x in (order or self.order) if x in names])
        if "frame_rotation" in names:
            kw["frame_rotation"] = src.frame_rotation.components
        if "boost_velocity" in names:
            kw["boost_velocity"] = src.boost_velocity
        return LorentzTransformation(**kw)

    def reorder(self, output_order):
        if not all(name in output_order for name in NORMAL_ORDER):
            raise ValueError("Not enough transformations")
        output_order = list(output_order)
        # ---- to normal order (supertranslation, rotation, boost)
        before = self.order[: self.order.index("supertranslation")]  # Lorentz elements applied before S
        L_normal = self._lorentz().reorder(NORMAL_ORDER)
        S_normal = self.supertranslation
        if before:
            lam = self._lorentz(names=tuple(before), order=before)
            S_normal = transform_supertranslation(self.supertranslation, lam, ctx=self._ctx)
        # ---- to the requested order
        L_out = L_normal.reorder(output_order)
        S_out = S_normal
        before_out = output_order[: output_order.index("supertranslation")]
        if before_out:
            lam = LorentzTransformation(
                ell_max=self.ell_max,
                order=before_out,
                **({"frame_rotation": L_out.frame_rotation.components} if "frame_rotation" in before_out else {}),
                **({"boost_velocity": L_out.boost_velocity} if "boost_velocity" in before_out else {}),
            )
            S_out = transform_supertranslation(S_normal, lam.inverse(), ctx=self._ctx)
        return BMSTransformation(
            frame_rotation=L_out.frame_rotation.components,
            boost_velocity=L_out.boost_velocity,
            supertranslation=S_out,
            ell_max=self.ell_max,
            order=output_order,
            ctx=self._ctx,
        )

    def inverse(self, output_order=None):
        if output_order is None:
            output_order = self.order[::-1]
        normal = self.reorder(NORMAL_ORDER)
        L_inv = normal._lorentz().inverse(output_order=["boost_velocity", "frame_rotation"])
        inv = BMSTransformation(
            frame_rotation=L_inv.frame_rotation.components,
            boost_velocity=L_inv.boost_velocity,
            supertranslation=-normal.supertranslation,
            ell_max=normal.ell_max,
            order=NORMAL_ORDER[::-1],
            ctx=self._ctx,
        )
        return inv if inv.order == list(output_order) else inv.reorder(output_order)

    def is_close_to(self, other):
        return (
            np.allclose(self.frame_rotation.components, other.frame_rotation.components)
            and np.allclose(self.boost_velocity, other.boost_velocity)
            and np.allclose(self.supertranslation, other.supertranslation)
        )

    def __mul__(self, other):
        """`other` applied after `self`; result in normal order (scri/bms_transformations.py:539-592)."""
        ell_max = max(self.ell_max, other.ell_max)
        b1, b2 = self.reorder(NORMAL_ORDER), other.reorder(NORMAL_ORDER)
        L1, L2 = b1._lorentz(), b2._lorentz()
        L1.ell_max = L2.ell_max = ell_max
        Lc = L1 * L2
        S1 = np.pad(b1.supertranslation, (0, (ell_max + 1) ** 2 - b1.supertranslation.shape[0]))
        S2 = np.pad(b2.supertranslation, (0, (ell_max + 1) ** 2 - b2.supertranslation.shape[0]))
        return BMSTransformation(
            frame_rotation=Lc.frame_rotation.components,
            boost_velocity=Lc.boost_velocity,
            supertranslation=transform_supertranslation(S1, L2, ctx=self._ctx) + S2,
            ell_max=ell_max,
            order=list(NORMAL_ORDER),
            ctx=self._ctx,
        )

    # HDF5 persistence (scri/bms_transformations.py:594-631) needs h5py, which is an optional dependency here
    def to_file(self, filename, file_write_mode="w", group=None):
        import h5py

        with h5py.File(filename, file_write_mode) as hf:
            g = hf.create_group(group) if group is not None else hf
            g.create_dataset("supertranslation", data=self.supertranslation)
            g.create_dataset("frame_rotation", data=self.frame_rotation.components)
            g.create_dataset("boost_velocity", data=self.boost_velocity)
            g.create_dataset("order", data=self.order)
            g.create_dataset("ell_max", data=self.ell_max)

    def from_file(self, filename, group=None):
        import h5py

        with h5py.File(filename, "r") as hf:
            g = hf[group] if group is not None else hf
            new = BMSTransformation(
                frame_rotation=np.array(g.get("frame_rotation")),
                boost_velocity=np.array(g.get("boost_velocity")),
                supertranslation=np.array(g.get("supertranslation")),
                order=[x.decode("utf-8") for x in np.array(g.get("order"))],
                ell_max=int(np.array(g.get("ell_max"))),
            )
        self.__dict__.update(new.__dict__)
