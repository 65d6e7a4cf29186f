"""Mode-space operators of ``scri.WaveformModes`` either side of the transformation path, as a mixin of
`scri_amd.WaveformModes`: spin raising / lowering (scri/waveform_modes.py:478-572), the parity conjugates with their symmetric and
antisymmetric parts and violation measures (:724-943), the conjugate-pair storage form (:659-703), precision truncation (:458-476)
and the all-angles inner product (:574-656).

Every one of them except the truncation is a map along the mode axis with per-column tables -- a sign, a factor, a partner column,
a conjugation -- i.e. one launch of `bms_mode_map` (kernels_modes.hip), the kernel that keeps the charge loops resident in HBM; the
tables are derived here from (l, m, s).  Signs and halves are exact in floating point, so the results equal the reference's to the
bit (tests/test_gpu_mode_operators.py mirrors tests/test_parity.py's np.array_equal checks)."""
import numpy as np

from . import engine
from . import UnknownDataType, RScaling, MScaling, DataNamesLaTeX
from .mode_algebra import LM_range


def time_intersection(t1, t2, min_step=None, min_time=None, max_time=None):
    """Common time axis of two series (scri/extrapolation.py:47-122): starts at the later of the two starts, advances by the
    smaller of the two local steps (at least min_step), ends before the earlier of the two ends."""
    t1, t2 = np.asarray(t1, dtype=float), np.asarray(t2, dtype=float)
    if t1.size == 0:
        raise ValueError("t1 is empty.  Assuming this is not desired.")
    if t2.size == 0:
        raise ValueError("t2 is empty.  Assuming this is not desired.")
    lo = max(t1[0], t2[0]) if min_time is None else max(t1[0], t2[0], min_time)
    hi = min(t1[-1], t2[-1]) if max_time is None else min(t1[-1], t2[-1], max_time)
    if lo > t1[-1] or lo > t2[-1]:
        raise ValueError(f"Empty intersection in t1=[{t1[0]}, ..., {t1[-1]}], t2=[{t2[0]}, ..., {t2[-1]}] with min_time={min_time}")
    if hi < t1[0] or hi < t2[0]:
        raise ValueError(f"Empty intersection in t1=[{t1[0]}, ..., {t1[-1]}], t2=[{t2[0]}, ..., {t2[-1]}] with max_time={max_time}")
    if min_step is None:
        min_step = min(np.min(np.diff(t1)), np.min(np.diff(t2)))
    out = [lo]
    i1 = i2 = 0
    while out[-1] < hi:
        now = out[-1]
        # the interval (t[i - 1], t[i]] of either series that holds `now` (index 0, i.e. the wrap-around step, outside its range)
        if now < t1[0] or now > t1[-1]:
            i1 = 0
        else:
            i1 = max(i1, 1)
            while now > t1[i1] and i1 < t1.size:
                i1 += 1
        if now < t2[0] or now > t2[-1]:
            i2 = 0
        else:
            i2 = max(i2, 1)
            while now > t2[i2] and i2 < t2.size:
                i2 += 1
        out.append(now + max(min(t1[i1] - t1[i1 - 1], t2[i2] - t2[i2 - 1]), min_step))
    return np.array(out[:-1])


_OPS = {"ð": +1, "ð̅": -1, "+": +1, "-": -1, +1: +1, -1: -1}
_CONVENTIONS = {"NP": 1.0, "GHP": 0.5}
# what a reflection does to a rotor (w, x, y, z): numpy-quaternion's {x,y,z,}_parity_{conjugate,symmetric_part,antisymmetric_part}
_FRAME = {
    "conjugate": {"x_": (1, 1, -1, -1), "y_": (1, -1, 1, -1), "z_": (1, -1, -1, 1), "": (1, 1, 1, 1)},
    "symmetric_part": {"x_": (1, 1, 0, 0), "y_": (1, 0, 1, 0), "z_": (1, 0, 0, 1), "": (1, 1, 1, 1)},
    "antisymmetric_part": {"x_": (0, 0, 1, 1), "y_": (0, 1, 0, 1), "z_": (0, 1, 1, 0), "": (0, 0, 0, 0)},
}


class ModeOperators:
    """Mixin of WaveformModes; uses its fields (data / _dev, ell_min, ell_max, spin_weight, frame, history)."""

    # ------------------------------------------------------------------ the one kernel
    def _columns(self):
        LM = LM_range(self.ell_min, self.ell_max)
        return LM[:, 0], LM[:, 1]

    def _partner(self):
        """column of (l, -m) for every column (l, m)"""
        ell, m = self._columns()
        return (ell * (ell + 1) - self.ell_min**2 - m).astype(np.int32)

    def _mode_map(self, idx_a, coef_a, conj_a=False, idx_b=None, coef_b=None, conj_b=False, stay=False):
        """out[t, j] = coef_a[j] op(data[t, idx_a[j]]) + coef_b[j] op(data[t, idx_b[j]]): a host array -- or, for a device-resident
        object with stay=True, the device tensor (the object is read where it lives either way)"""
        if self.is_device_resident:
            from .device_series import DeviceModesTimeSeries

            d = DeviceModesTimeSeries(self._dev, self.t, self.spin_weight if abs(self.spin_weight) < 100 else 0, self.ell_min, self.ell_max, ctx=self._ctx)
            out = d._map(len(idx_a), idx_a, coef_a, conj_a, other=d if idx_b is not None else None, idx_b=idx_b, coef_b=coef_b, conj_b=conj_b)
            return out if stay else out.cpu().numpy()
        b = self.data if idx_b is not None else None
        return engine.mode_map(self.data, idx_a, coef_a, conj_a, b, idx_b, coef_b, conj_b, ctx=self._ctx)

    # ------------------------------------------------------------------ eth, ethbar (waveform_modes.py:478-572)
    def ladder_factor(self, operations, s, ell, eth_convention="NP"):
        """Factor a sequence of spin raising (+1, '+', 'ð') / lowering (-1, '-', 'ð̅') operators, applied right to left,
        puts on a harmonic of spin weight s and degree ell; 'NP' or 'GHP' (a factor 1/sqrt2 per operator) convention."""
        if isinstance(operations, str):
            operations = operations.replace("ð̅", "-").replace("ð", "+")
        keys = _OPS.keys()
        if not set(operations).issubset(keys):
            key_strings = {k for k in keys if isinstance(k, str)}
            raise ValueError(
                "operations must be a string composed of {} or a list with elements coming from the set {}".format(key_strings, set(keys))
            )
        if eth_convention not in _CONVENTIONS:
            raise ValueError("eth_convention must be one of {}".format(set(_CONVENTIONS.keys())))
        product, sign = 1.0, 1.0
        for op in reversed(operations):
            step = _OPS[op]
            sign *= step
            product *= ((ell - s * step) * (ell + s * step + 1.0) if ell >= abs(s) else 0.0) * _CONVENTIONS[eth_convention]
            s += step
        return sign * np.sqrt(product)

    def apply_eth(self, operations, eth_convention="NP"):
        """The mode data with the operators applied (an array shaped like `data`, same (l, m) layout; the object is not
        modified and the spin weight of the result is the caller's business, as in the reference)."""
        ell, _ = self._columns()
        s = self.spin_weight
        per_ell = {int(l): self.ladder_factor(operations, s, int(l), eth_convention=eth_convention) for l in np.unique(ell)}
        coef = np.array([per_ell[int(l)] for l in ell], dtype=complex)
        return self._mode_map(np.arange(ell.size, dtype=np.int32), coef)

    @property
    def eth(self):
        """the spin-raised mode data"""
        return self.apply_eth(operations="+")

    @property
    def ethbar(self):
        """the spin-lowered mode data"""
        return self.apply_eth(operations="-")

    # ------------------------------------------------------------------ parity (waveform_modes.py:724-943)
    def _parity_tables(self, direction):
        """(partner column, sign) of the conjugate: W[l, m] = sign conj(f[partner])"""
        ell, m = self._columns()
        s = self.spin_weight
        own = np.arange(ell.size, dtype=np.int32)
        if direction == "x_":
            return own, np.where(m % 2 == 0, 1.0, -1.0)
        if direction == "y_":
            return own, np.ones(ell.size)
        if direction == "z_":
            return self._partner(), np.where((ell + s) % 2 == 0, 1.0, -1.0)
        return self._partner(), np.where((ell + s + m) % 2 == 0, 1.0, -1.0)

    def _parity(self, direction, part):
        if self.dataType == UnknownDataType:
            raise ValueError(f"Cannot compute parity type for {self.data_type_string}.")
        idx, sign = self._parity_tables(direction)
        n = idx.size
        if part == "conjugate":
            data = self._mode_map(idx, sign.astype(complex), conj_a=True, stay=True)
        else:
            half = 0.5 if part == "symmetric_part" else -0.5
            data = self._mode_map(np.arange(n, dtype=np.int32), np.full(n, 0.5 + 0j), False, idx, (half * sign).astype(complex), True, stay=True)
        W = self.copy_without_data()
        W.t = self.t.copy()
        if isinstance(data, np.ndarray):
            W.data = data
        else:  # the result of a device-resident object stays in HBM
            W._host, W._dev = None, data
        W.frame = self.frame * np.asarray(_FRAME[part][direction], dtype=float)[None, :]
        W._append_history(f"{W} = {self}.{direction}parity_{part}")
        return W

    def _violation(self, direction, normalized):
        anti = self._parity(direction, "antisymmetric_part").norm()
        return np.sqrt(anti / self.norm()) if normalized else anti

    x_parity_conjugate = property(lambda self: self._parity("x_", "conjugate"), doc="Reflect modes across the y-z plane (along the x axis)")
    x_parity_symmetric_part = property(lambda self: self._parity("x_", "symmetric_part"), doc="Component invariant under x_parity_conjugate")
    x_parity_antisymmetric_part = property(lambda self: self._parity("x_", "antisymmetric_part"), doc="Component that changes sign under x_parity_conjugate")
    x_parity_violation_squared = property(lambda self: self._violation("x_", False), doc="(Squared) norm of the x-parity-antisymmetric component")
    x_parity_violation_normalized = property(lambda self: self._violation("x_", True), doc="Norm of the x-parity-antisymmetric component over the norm")
    y_parity_conjugate = property(lambda self: self._parity("y_", "conjugate"), doc="Reflect modes across the x-z plane (along the y axis)")
    y_parity_symmetric_part = property(lambda self: self._parity("y_", "symmetric_part"), doc="Component invariant under y_parity_conjugate")
    y_parity_antisymmetric_part = property(lambda self: self._parity("y_", "antisymmetric_part"), doc="Component that changes sign under y_parity_conjugate")
    y_parity_violation_squared = property(lambda self: self._violation("y_", False), doc="(Squared) norm of the y-parity-antisymmetric component")
    y_parity_violation_normalized = property(lambda self: self._violation("y_", True), doc="Norm of the y-parity-antisymmetric component over the norm")
    z_parity_conjugate = property(lambda self: self._parity("z_", "conjugate"), doc="Reflect modes across the x-y plane (along the z axis)")
    z_parity_symmetric_part = property(lambda self: self._parity("z_", "symmetric_part"), doc="Component invariant under z_parity_conjugate")
    z_parity_antisymmetric_part = property(lambda self: self._parity("z_", "antisymmetric_part"), doc="Component that changes sign under z_parity_conjugate")
    z_parity_violation_squared = property(lambda self: self._violation("z_", False), doc="(Squared) norm of the z-parity-antisymmetric component")
    z_parity_violation_normalized = property(lambda self: self._violation("z_", True), doc="Norm of the z-parity-antisymmetric component over the norm")
    parity_conjugate = property(lambda self: self._parity("", "conjugate"), doc="Reflect modes along all axes")
    parity_symmetric_part = property(lambda self: self._parity("", "symmetric_part"), doc="Component invariant under parity_conjugate")
    parity_antisymmetric_part = property(lambda self: self._parity("", "antisymmetric_part"), doc="Component that changes sign under parity_conjugate")
    parity_violation_squared = property(lambda self: self._violation("", False), doc="(Squared) norm of the parity-antisymmetric component")
    parity_violation_normalized = property(lambda self: self._violation("", True), doc="Norm of the parity-antisymmetric component over the norm")

    # ------------------------------------------------------------------ conjugate pairs (waveform_modes.py:659-703), in place
    def convert_to_conjugate_pairs(self):
        """s[l, m] = (f[l, m] + conj f[l, -m]) / sqrt2 goes to column m > 0, d[l, m] = (f[l, m] - conj f[l, -m]) / sqrt2 to column
        -m; m = 0 stays.  The sqrt2 keeps the sum of squared magnitudes at each time."""
        _, m = self._columns()
        n = m.size
        own, partner = np.arange(n, dtype=np.int32), self._partner()
        r = 1.0 / np.sqrt(2.0)
        idx_a = np.where(m >= 0, own, partner).astype(np.int32)  # always the m > 0 member of the pair
        coef_a = np.where(m == 0, 1.0, r).astype(complex)
        idx_b = np.where(m > 0, partner, np.where(m < 0, own, -1)).astype(np.int32)  # the m < 0 member, conjugated
        coef_b = np.where(m > 0, r, np.where(m < 0, -r, 0.0)).astype(complex)
        out = self._mode_map(idx_a, coef_a, False, idx_b, coef_b, True, stay=True)
        if isinstance(out, np.ndarray):
            self.data = out
        else:
            self._host, self._dev = None, out
        self._append_history(f"{self}.convert_to_conjugate_pairs()")

    def convert_from_conjugate_pairs(self):
        """the inverse of convert_to_conjugate_pairs: f[l, m] = (s + d) / sqrt2, f[l, -m] = conj(s - d) / sqrt2"""
        _, m = self._columns()
        n = m.size
        own, partner = np.arange(n, dtype=np.int32), self._partner()
        r = 1.0 / np.sqrt(2.0)
        plus = np.where(m >= 0, own, partner).astype(np.int32)   # column holding s (m > 0)
        minus = np.where(m >= 0, partner, own).astype(np.int32)  # column holding d (m < 0)
        data = self.data
        out = np.empty_like(data)
        pos, neg, zero = m > 0, m < 0, m == 0
        up = engine.mode_map(data, plus, np.where(zero, 1.0, r).astype(complex), False, data, np.where(zero, -1, minus).astype(np.int32),
                             np.where(pos, r, np.where(neg, -r, 0.0)).astype(complex), False, ctx=self._ctx)
        out[:, pos | zero] = up[:, pos | zero]
        out[:, neg] = np.conjugate(up[:, neg])
        self.data = out
        self._append_history(f"{self}.convert_from_conjugate_pairs()")

    # ------------------------------------------------------------------ truncation (waveform_modes.py:458-476), in place
    def truncate(self, tol=1e-10):
        """Set to zero the bits of `data` that typically contribute less than `tol` times the norm at that instant: every
        mode is rounded to a multiple of the power of two below tol / sqrt(n_modes) times the norm."""
        if tol != 0.0:
            data = self.data
            quantum = np.linalg.norm(data, axis=1) * (tol / np.sqrt(self.n_modes))
            scale = (2.0 ** np.floor(-np.log2(quantum)))[:, np.newaxis]
            data *= scale
            np.round(data, out=data)
            data /= scale
        self._append_history(f"{self}.truncate(tol={tol})")

    # ------------------------------------------------------------------ inner product (waveform_modes.py:574-656)
    def inner_product(self, b, t1=None, t2=None, allow_LM_differ=False, allow_times_differ=False):
        """The all-angles inner product <self, b>: integral over [t1, t2] of sum_lm conj(self_lm) b_lm (cubic-spline quadrature
        of the samples, as quaternion.calculus.spline_definite_integral does)."""
        if self.spin_weight != b.spin_weight:
            raise ValueError("Spin weights must match in inner_product")
        A, B = self, b
        if (self.ell_min != b.ell_min) or (self.ell_max != b.ell_max):
            if not allow_LM_differ:
                raise ValueError("ell_min and ell_max must match in inner_product (use allow_LM_differ=True to override)")
            clip = slice(max(self.ell_min, b.ell_min), min(self.ell_max, b.ell_max) + 1)
            if clip.start >= clip.stop:
                raise ValueError("Intersection of (ell,m) modes is empty.  Assuming this is not desired.")
            A, B = A[:, clip], B[:, clip]
        times = self.t
        if not np.array_equal(self.t, b.t):
            if not allow_times_differ:
                raise ValueError("Time samples must match in inner_product (use allow_times_differ=True to override)")
            times = time_intersection(self.t, b.t)
            A, B = A.interpolate(times), B.interpolate(times)
        t1 = times[0] if t1 is None else t1
        t2 = times[-1] if t2 is None else t2
        integrand = np.sum(np.conjugate(A.data) * B.data, axis=1)
        ends = engine.spline_derivative(times, integrand[:, np.newaxis], np.array([t1, t2], dtype=float), order=-1, ctx=self._ctx)
        return complex(ends[1, 0] - ends[0, 0])

    # ------------------------------------------------------------------ what the object says about itself (waveform_base.py:425-516)
    @property
    def is_valid(self):
        return self.ensure_validity(alter=False, assertions=False)

    @property
    def gamma_weight(self):
        """Non-conformal effect of a boost: the power of gamma a mass that was scaled out picks up (a monopole cannot depend on
        the direction the way r and the field do)"""
        return (MScaling[self.dataType] if self.m_is_scaled_out else 0) + (
            -RScaling[self.dataType] if (self.r_is_scaled_out and self.m_is_scaled_out) else 0
        )

    @property
    def r_scaling(self):
        return RScaling[self.dataType]

    @property
    def m_scaling(self):
        return MScaling[self.dataType]

    @property
    def data_type_latex(self):
        return DataNamesLaTeX[self.dataType]

    @property
    def descriptor_string(self):
        """A name for the content suitable for file names, e.g. 'rMpsi4' or 'rhOverM', from what the object knows of itself"""
        if self.dataType == UnknownDataType:
            return self.data_type_string
        r_power, name = RScaling[self.dataType], self.data_type_string
        text = ""
        if self.r_is_scaled_out and r_power >= 1:
            text = "r" if r_power == 1 else "r" + str(r_power)
        if not self.m_is_scaled_out:
            return text + name
        m_power = MScaling[self.dataType] - (r_power if self.r_is_scaled_out else 0)
        if m_power < -1:
            return text + name + "OverM" + str(-m_power)
        if m_power == -1:
            return text + name + "OverM"
        if m_power == 0:
            return text + name
        return text + ("M" if m_power == 1 else "M" + str(m_power)) + name

    def deepcopy(self):
        """an alias of copy(), which is deep anyway"""
        W = self.copy()
        W._append_history(f"{W} = {self}.deepcopy()")
        return W

    def SI_units(self, current_unit_mass_in_solar_masses, distance_from_source_in_megaparsecs=100):
        """The waveform a source of the given total mass shows at the given distance, in SI units (times in s; curvature
        scalars in s^-2), assuming the data are in geometric units as `dataType`, `r_is_scaled_out` and `m_is_scaled_out` say
        (waveform_base.py:970-1045)."""
        import warnings

        from . import m_sun_in_meters, speed_of_light, parsec_in_meters

        if not self.r_is_scaled_out:
            warnings.warn("\nTrying to convert to SI units, the radius is supposedly not scaled out.\n"
                          "This seems to suggest that the data may already be in some units...")
        if not self.m_is_scaled_out:
            warnings.warn("\nTrying to convert to SI units, the mass is supposedly not scaled out.\n"
                          "This seems to suggest that the data may already be in some units...")
        M_in_meters = current_unit_mass_in_solar_masses * m_sun_in_meters
        M_in_seconds = M_in_meters / speed_of_light
        R_over_M = distance_from_source_in_megaparsecs * (1e6 * parsec_in_meters) / M_in_meters
        scale = 1.0
        if self.r_is_scaled_out:
            scale = R_over_M ** -self.r_scaling
        if self.m_is_scaled_out:
            scale = scale * M_in_meters ** -self.m_scaling if self.r_is_scaled_out else M_in_meters ** -self.m_scaling
        W = self.copy_without_data()
        W.t = M_in_seconds * self.t if self.m_is_scaled_out else np.copy(self.t)
        W.frame = np.copy(self.frame)
        W.data = scale * self.data
        W.m_is_scaled_out = False
        W.r_is_scaled_out = False
        W._append_history(
            "{} = {}.SI_units(current_unit_mass_in_solar_masses={}, distance_from_source_in_megaparsecs={})".format(
                W, self, current_unit_mass_in_solar_masses, distance_from_source_in_megaparsecs))
        return W

    # ------------------------------------------------------------------ pickling, printing (waveform_base.py:735-796)
    def __getstate__(self):
        """State for copy / pickle: the weights as a host array (a device-resident object is read back), no engine context (the
        unpickled object takes the process-wide one on first use)"""
        import copy

        state = copy.deepcopy({k: v for k, v in self.__dict__.items() if k not in ("_ctx", "_dev", "_host")})  # (deep, as in the reference)
        state["_host"] = self._dev.cpu().numpy() if self._dev is not None else np.array(self._host, copy=True)  # (the source stays where it is)
        return state

    def __setstate__(self, state):
        from .waveform_modes import _next_num

        old_num = state.get("num")
        self.__dict__.update(state)
        self._ctx, self._dev = None, None
        _next_num[0] += 1
        self.num = _next_num[0]  # every object keeps a number of its own
        self.history = list(self.history)
        self._append_history(f"copied, deepcopied, or unpickled as {self}")
        self._append_history("{} = {}".format(self, f"{self}".replace(str(self.num), str(old_num))))

    def __repr__(self):
        from textwrap import dedent

        opts = np.get_printoptions()
        np.set_printoptions(threshold=6, linewidth=150, precision=6)
        try:
            rep = """
             {0}(
                 t={1},
                 frame={2},
                 data={3},
                 frameType={4}, dataType={5},
                 r_is_scaled_out={6}, m_is_scaled_out={7})  # num = {8}""".format(
                type(self).__name__, str(self.t).replace("\n", "\n" + " " * 19), str(self.frame).replace("\n", "\n" + " " * 23),
                str(self.data).replace("\n", "\n" + " " * 22), self.frameType, self.dataType, self.r_is_scaled_out, self.m_is_scaled_out, self.num)
        finally:
            np.set_printoptions(**opts)
        return dedent(rep)

    # ------------------------------------------------------------------ difference of two waveforms (waveform_base.py:577-687)
    def compare(self, w_a, min_time_step=0.005, min_time=-3.0e300):
        """A waveform holding A - B on the common time axis of the two (B = this object): the mode data of both are interpolated
        there (not-a-knot cubic splines, on the GPU) and subtracted mode by mode, and `frame` is the rotation taking this frame
        into A's, with the sign that keeps its mean rotor nearer to +1 than to -1."""
        import warnings

        from . import quaternions
        from .waveform_modes import WaveformModes

        if self.frameType != w_a.frameType:
            warnings.warn("\nWarning:\n    This Waveform is in the " + self.frame_type_string + " frame,"
                          "\n    The Waveform in the argument is in the " + w_a.frame_type_string + " frame."
                          "\n    Comparing them probably does not make sense.\n")
        if self.n_modes != w_a.n_modes:
            raise Exception("Trying to compare waveforms with mismatched LM data.\nA.n_modes=" + str(w_a.n_modes) + "\tB.n_modes()=" + str(self.n_modes))
        times = time_intersection(self.t, w_a.t)
        # A's modes may come in another order: column of (l, m) in this object for every column of A
        order = np.array([self.index(ell, m) for ell, m in w_a.LM], dtype=int)
        data = engine.cubic_spline(w_a.t, w_a.data, times, ctx=self._ctx) - engine.cubic_spline(self.t, self.data[:, order], times, ctx=self._ctx)
        fa, fb = np.asarray(w_a.frame, dtype=float).reshape(-1, 4), np.asarray(self.frame, dtype=float).reshape(-1, 4)
        at = lambda f, t: quaternions.squad(f, t, times) if f.shape[0] > 1 else f  # noqa: E731
        if fa.shape[0] >= 1 and fb.shape[0] >= 1:
            frame = quaternions.multiply(at(fa, w_a.t), quaternions.conjugate(at(fb, self.t)) / np.sum(at(fb, self.t) ** 2, axis=-1, keepdims=True))
        elif fb.shape[0] >= 1:
            B = at(fb, self.t)
            frame = quaternions.conjugate(B) / np.sum(B**2, axis=-1, keepdims=True)
        elif fa.shape[0] >= 1:
            frame = at(fa, w_a.t)
        else:
            frame = np.zeros((0, 4))
        frame = np.atleast_2d(frame)
        if frame.shape[0] == times.shape[0] and frame.shape[0] > 1:
            # mean rotor in the chordal metric: the normalised time integral of the rotor series
            mean = engine.spline_derivative(times, frame.astype(complex), times[-1:], order=-1, ctx=self._ctx).real[0]
            if mean[0] < 0:  # nearer to -1 than to +1: |R + 1| < |R - 1|
                frame = -frame
        elif frame.shape[0] == 1 and frame[0, 0] < 0:
            frame = -frame
        W = WaveformModes(t=times, data=data, history=[], frame=frame, frameType=self.frameType, dataType=self.dataType,
                          r_is_scaled_out=self.r_is_scaled_out, m_is_scaled_out=self.m_is_scaled_out, ell_min=w_a.ell_min, ell_max=w_a.ell_max,
                          ctx=self._ctx)
        W.history += ["B.compare(A)\n", "### A.history.str():\n" + "".join(w_a.history), "### B.history.str():\n" + "".join(self.history),
                      "### End of old histories from `compare`"]
        return W
