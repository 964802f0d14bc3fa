"""Adapters from "whatever the pipeline handed to ``setup``" to provider / telescope.

Mirrors ``draco/core/io.py:251-276`` (``get_telescope`` / ``get_beamtransfer``): accept a
``ProductManager``-like (has ``.beamtransfer``), a beam-transfer provider (has
``.beam_m`` and ``.telescope``) or, for ``get_telescope``, a bare telescope; anything
else raises ``RuntimeError`` with the reference's message.
"""

from __future__ import annotations

from .products import BeamTransferProvider, ForeignProvider, TransitTelescope


def get_beamtransfer(obj):
    """Return a provider out of the input (``io.py:265-276``)."""
    if isinstance(obj, BeamTransferProvider):
        return obj
    if hasattr(obj, "beamtransfer"):  # ProductManager-like
        return get_beamtransfer(obj.beamtransfer)
    if hasattr(obj, "beam_m") and hasattr(obj, "telescope"):  # e.g. a real driftscan BeamTransfer
        return ForeignProvider(obj)
    raise RuntimeError(f"Could not get BeamTransfer instance out of {obj!r}")


def get_telescope(obj):
    """Return a telescope out of the input (``io.py:251-262``)."""
    try:
        return get_beamtransfer(obj).telescope
    except RuntimeError:
        if isinstance(obj, TransitTelescope) or all(hasattr(obj, a) for a in ("lmax", "mmax", "frequencies")):
            return obj
    raise RuntimeError(f"Could not get telescope instance out of {obj!r}")
