"""Opt-in drop-in: graft the GPU implementations onto an installed `scri`, the same way scri wires itself
(scri/__init__.py:125-150: ``WaveformModes.rotate_decomposition_basis = rotate_decomposition_basis`` ...)."""


def patch_scri():
    import scri  # noqa: F401  (only available where the reference is installed)
    from . import adapters

    return adapters.install(scri)
