"""The reference's own tutorials (docs/tutorial_abd.rst, docs/tutorial_waveformmodes.rst, docs/tutorial_bms.rst), statement by
statement, with `scri_amd` in the place of `scri`: what a user of the reference types must work unchanged on this package.  Only the
statements of the path are here (construction, access, calculus, products, transformations, BMS algebra); file readers and plotting
are not."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_tutorial_abd_statement_by_statement():
    import scri_amd as scri
    from scri_amd.mode_algebra import LM_index  # (the tutorial's `from spherical_functions import LM_index as lm`)

    rng = np.random.default_rng(11)
    # docs/tutorial_abd.rst:33-37
    abd = scri.asymptotic_bondi_data.AsymptoticBondiData(
        time=np.linspace(0, 10, 11),
        ell_max=8,
        multiplication_truncator=max,
    )
    psi4_mode_data = np.zeros((11, 81))
    abd.psi4 = psi4_mode_data  # :69
    for name, s in (("psi4", -2), ("psi3", -1), ("psi2", 0), ("psi1", 1), ("psi0", 2), ("sigma", 2)):
        d = rng.normal(size=(11, 81)) + 1j * rng.normal(size=(11, 81))
        d[:, : s * s] = 0.0
        setattr(abd, name, d)
    assert np.array_equal(abd.t, np.linspace(0, 10, 11)) and np.array_equal(abd.u, abd.t)  # :115
    l, m = 2, 1
    assert abd.psi4[:, abd.psi4.index(l, m)].shape == (11,)  # :138
    assert np.array_equal(abd.psi4[:, LM_index(l, m, 0)], abd.psi4[:, abd.psi4.index(l, m)])  # :142 (sf.LM_index)
    assert type(abd.sigma).__name__ == "ModesTimeSeries"  # :148-149
    for attr in ("dot", "ddot", "int", "iint", "eth_GHP", "ethbar_GHP", "eth", "ethbar"):  # :158-172
        assert getattr(abd.psi4, attr).shape == (11, 81), attr
    assert abd.sigma.s == 2 and abd.sigma.bar.s == -2 and abd.sigma.dot.eth_GHP.eth_GHP.s == 4  # :182-187
    with pytest.raises(ValueError):  # :195 "This will throw an error"
        abd.psi4 + abd.psi3
    assert (abd.psi4.eth_GHP + abd.psi3).s == -1  # :198
    by_3j = abd.sigma * abd.psi4  # :207
    by_grid = abd.sigma.grid_multiply(abd.psi4)  # :219
    assert by_3j.s == 0 and by_grid.s == 0 and by_3j.ell_max == 8 and by_grid.ell_max == 8
    # (ell_max = 8 cannot hold the product of two l <= 8 fields: the two agree where no aliasing reaches, l <= 0 .. not beyond)
    full = abd.sigma.grid_multiply(abd.psi4, working_ell_max=16, output_ell_max=8)
    assert np.abs(np.asarray(full) - np.asarray(by_3j)).max() < 1e-11 * np.abs(np.asarray(by_3j)).max()
    assert np.iscomplexobj(np.asarray(abd.psi2.real)) and abd.psi2.ndarray.real.dtype == np.float64  # :245-249
    assert abd.sigma.bar.s == -2 and np.conjugate(abd.sigma.ndarray).shape == (11, 81)  # :253-257
    abd_prime = abd.transform(  # :271-274
        space_translation=[-1.0, 4.0, 0.2],
        boost_velocity=[0.0, 0.0, 1e-2],
    )
    assert type(abd_prime) is type(abd) and abd_prime.n_times <= abd.n_times and abd_prime.sigma.shape[1] == 81
    h = scri.asymptotic_bondi_data.map_to_superrest_frame.MT_to_WM(2.0 * abd.sigma.bar)  # :340
    assert isinstance(h, scri.WaveformModes) and h.data.shape[0] == 11 and abd.h.data.shape == h.data.shape  # :357


def test_tutorial_waveformmodes_statement_by_statement():
    import scri_amd as scri
    from scri_amd.mode_algebra import LM_index

    rng = np.random.default_rng(12)
    t = np.linspace(0, 10, 100)
    my_strain_data = (rng.normal(size=(100, 77)) + 1j * rng.normal(size=(100, 77))) * np.exp(-0.05 * t)[:, None]
    h = scri.WaveformModes(  # docs/tutorial_waveformmodes.rst:42-51
        dataType=scri.h,
        t=t,
        data=my_strain_data,
        ell_min=2,
        ell_max=8,
        frameType=scri.Inertial,
        r_is_scaled_out=True,
        m_is_scaled_out=True,
    )
    l, m = 2, 1
    # :163-167 (the tutorial writes h.index(l,m,h.ell_min); the method takes (ell, m) -- scri/waveform_modes.py:423 -- there as here)
    assert np.array_equal(h.data[:, h.index(l, m)], h.data[:, LM_index(l, m, h.ell_min)])
    with pytest.raises(TypeError):
        h.index(l, m, h.ell_min)
    h_grid = h.to_grid()  # :174
    # (the default grid resolves ell_max + 1: the identity supertranslation has l <= 1, scri/waveform_grid.py:96-99)
    assert isinstance(h_grid, scri.WaveformGrid) and (h_grid.n_theta, h_grid.n_phi) == (19, 19) and np.array_equal(h_grid.t, h.t)
    h_modes = h_grid.to_modes()  # :177
    assert h_modes.ell_max == 9 and h_modes.ell_min == 2
    assert np.abs(h_modes.data[:, :77] - h.data).max() < 1e-12 and np.abs(h_modes.data[:, 77:]).max() < 1e-12
    assert h_grid.to_modes(5).ell_max == 5  # :180-181
    new_t = np.linspace(1.0, 9.0, 37)
    assert h.interpolate(new_t).data.shape == (37, 77)  # :190
    for attr in ("data_dot", "data_ddot", "data_int", "data_iint"):  # :193-202
        assert getattr(h, attr).shape == (100, 77)
    # :206-208 (grafted functions, scri/__init__.py:146-148: methods to call, there as here)
    assert h.energy_flux().shape == (100,) and h.angular_momentum_flux().shape == (100, 3) and h.momentum_flux().shape == (100, 3)
    assert h.apply_eth("++--", eth_convention="GHP").data.shape == (100, 77)  # :211
    h_prime = h.transform(space_translation=[-1.0, 4.0, 0.2], boost_velocity=[0.0, 0.0, 1e-2])  # :228
    assert h_prime.data.shape[1] == 77 and h_prime.n_times <= 100
    R = np.array([np.cos(0.3), 0.0, np.sin(0.3), 0.0])
    rotated = scri.WaveformModes(h).rotate_decomposition_basis(R)  # :245
    assert np.abs(rotated.norm() - h.norm()).max() < 1e-12 * h.norm().max()
    assert scri.WaveformModes(h).rotate_physical_system(R).data.shape == h.data.shape  # :248
    co = scri.WaveformModes(h).to_corotating_frame()  # :258
    assert co.frameType == scri.Corotating and co.frame.shape[0] == 100  # :266
    assert scri.WaveformModes(h).to_coprecessing_frame().frameType == scri.Coprecessing  # :259
    assert np.abs(co.to_inertial_frame().data - h.data).max() < 1e-10 * np.abs(h.data).max()  # :260


def test_tutorial_bms_statement_by_statement():
    import scri_amd as scri
    from scri_amd import bms_transformations

    S = np.array([1, 2 + 4j, 3, -2 + 4j, 7 - 5j, -3 - 2j, 4, 3 - 2j, 7 + 5j]) * 1e-3  # docs/tutorial_bms.rst:219-221
    q = np.array([1.0, 2.0, 3.0, 4.0]) / np.sqrt(30.0)  # np.quaternion(1, 2, 3, 4).normalized().components
    v = np.array([1, 2, 3]) * 1e-4
    BMS1 = bms_transformations.BMSTransformation(supertranslation=S, frame_rotation=q, boost_velocity=v)  # :223-225
    again = BMS1.reorder(["boost_velocity", "supertranslation", "frame_rotation"])  # :229
    assert again.order == ["boost_velocity", "supertranslation", "frame_rotation"]
    BMS1_inv = BMS1.inverse()  # :234
    BMS2 = bms_transformations.BMSTransformation(supertranslation=S[::-1].conj() * 0.5, frame_rotation=[0.8, 0.0, 0.6, 0.0], boost_velocity=-2 * v)
    prod = BMS2 * BMS1  # :238
    assert isinstance(prod, bms_transformations.BMSTransformation)
    one = BMS1_inv * BMS1
    assert np.abs(np.asarray(one.supertranslation)).max() < 1e-12 and np.abs(np.asarray(one.boost_velocity)).max() < 1e-13
