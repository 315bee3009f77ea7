"""TEST INFRASTRUCTURE ONLY -- the parity oracle of adell_mri_amd.

Two independent CPU restatements of the reference's hot path:

* ``oracle.cops``      ctypes front-end of ``oracle/c/adell_oracle.c`` (plain C
  loops, fp64 accumulation, torch's NCDHW layout): op-level checker.
* ``oracle.torch_ref`` functional stock-torch interpreter of the reference's
  U-Net ``state_dict`` (model-level checker, CPU baseline).

Both are pinned against fixtures generated from the real reference
(``oracle/make_golden.py`` -> ``tests/golden``). Nothing under ``adell_mri_amd``
may import this package: only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg do.
"""
