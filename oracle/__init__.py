"""CPU oracle for the BMS-transformation hot path of moble/scri.

TEST INFRASTRUCTURE ONLY.  This package is a plain numpy/scipy/mpmath restatement of the
reference algorithm (scri 2024.0.13, files cited per function as ``scri/...:line`` relative to
the reference checkout) and of the published algorithms of its un-vendored third-party
dependencies (``spherical_functions``, ``spinsfast``, ``numpy-quaternion``; only ``>=`` floors
are pinned upstream, ``pyproject.toml:20-30``).  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it -- as the checker, never as the
thing measured or shipped.  The product (``scri_amd``) never imports this package and fails
loudly when its HIP library is missing.

Pinning status: the reference cannot be imported in the build container (numba, quaternion,
spherical_functions, spinsfast are absent) and holds no numeric golden outputs for this path.
The oracle is therefore pinned against every *analytic* known-answer test the reference's own
test-suite holds for the path (tests/test_oracle_known_answers.py mirrors
``tests/test_waveform_grid.py``, ``tests/test_rotations.py``,
``tests/test_asymptoticbondidata.py``, ``tests/test_bms_transformations.py`` of the
reference) and, for the spline step, against the live ``scipy.interpolate`` the reference
itself calls.  Bit-level parity with the reference's numba/sf output is UNPINNED (no golden
vectors exist upstream); see DESIGN.md "Oracle".
"""
