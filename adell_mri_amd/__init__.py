"""adell_mri_amd -- MI355X-native (gfx950) U-Net / UNETR forward-backward path of
CCIG-Champalimaud/adell-mri behind the reference's own nn.Module surface.

Python here is host plumbing (device memory, streams, torch.distributed); the
arithmetic lives in ``libadellhip.so`` (``adell_mri_amd/csrc``, C ABI in
``include/adell_hip.h``). There is no CPU / eager-torch fallback.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
