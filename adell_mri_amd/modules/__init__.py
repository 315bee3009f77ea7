"""Host-side mirror of ``adell_mri.modules`` for the U-Net / UNETR hot path: same
class names, constructor signatures, module tree and ``state_dict`` keys; the
arithmetic runs in ``libadellhip.so``."""
