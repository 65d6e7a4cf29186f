"""Opt-in drop-in: graft the GPU implementations onto an installed `scri`, the same way scri wires itself
(scri/__init__.py:125-150: ``WaveformModes.rotate_decomposition_basis = rotate_decomposition_basis`` ...)."""


def patch_scri(scri=None, ctx=None):
    """Graft onto `scri` (default: the installed package).  Returns the patched attribute names."""
    if scri is None:
        import scri  # only available where the reference is installed
    from . import adapters

    return adapters.install(scri, ctx=ctx)


def unpatch_scri(scri=None):
    if scri is None:
        import scri
    from . import adapters

    adapters.uninstall(scri)
