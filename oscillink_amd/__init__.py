"""oscillink_amd -- MI355X (gfx950) implementation of the Oscillink lattice settle path.

Drop-in for the reference's public names on that path (oscillink/__init__.py:4-21):
    from oscillink_amd import Oscillink, OscillinkLattice, compute_diffusion_gates, verify_receipt
"""
from .lattice import OscillinkLattice, __version__, json_line_logger  # noqa: F401
from .receipts import verify_receipt, verify_receipt_mode  # noqa: F401
from .diffusion import compute_diffusion_gates  # noqa: F401

Oscillink = OscillinkLattice

__all__ = [
    "Oscillink",
    "OscillinkLattice",
    "verify_receipt",
    "verify_receipt_mode",
    "compute_diffusion_gates",
    "json_line_logger",
]
