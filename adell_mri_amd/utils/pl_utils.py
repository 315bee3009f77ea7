"""``get_devices`` (adell_mri/utils/pl_utils.py:424-458): the ``--dev`` string of the
entrypoints -> (accelerator, device list, parallelisation strategy) for ``Trainer``.

``"cuda:0,1,2,3,4,5,6,7"`` -> ``("gpu", [0..7], "ddp")``: one process per GPU, gradients
all-reduced over RCCL (``adell_mri_amd.parallel``). The strategy for more than one device
defaults to the environment variable ``ADELL_PARALLEL_STRATEGY`` (pl_utils.py:21), else "ddp".
"""
import os
from typing import List, Tuple, Union

ADELL_PARALLEL_STRATEGY = os.environ.get("ADELL_PARALLEL_STRATEGY", "ddp")


def get_devices(device_str: str, strategy: str = None) -> Tuple[str, Union[List[int], str], str]:
    accelerator = "gpu" if "cuda" in device_str else "cpu"
    if ":" not in device_str:
        return accelerator, [0], "auto"
    try:
        devices = [int(i) for i in device_str.split(":")[-1].split(",")]
    except Exception:  # noqa: BLE001  ("cuda:all" and the like)
        devices = "auto"
    strategy_out = "auto"
    if len(devices) > 1:
        strategy_out = ADELL_PARALLEL_STRATEGY if strategy is None else strategy
    return accelerator, devices, strategy_out
