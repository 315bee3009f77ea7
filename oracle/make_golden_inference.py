"""tests/golden/inference_ops.npz from the REAL reference operators
(/root/reference/adell_mri/utils/inference.py:262-990: SlidingWindowSegmentation, FlippedInference,
SegmentationInference), run in the build container only.

The module's single MONAI use is ``isinstance(x, MetaTensor)`` (inference.py:11, 64-90): a stub
``monai.data.meta_tensor`` whose MetaTensor is a bare ``torch.Tensor`` subclass lets it import; no
plain tensor is an instance of it, so every code path the fixtures take is the reference's own. The
'network' is the closed form of tests/cases.py::inference_net (the same function the parity tests
call), the inputs are seeded normals stored in the file.

    python oracle/make_golden_inference.py
"""
import os
import sys
import types

import numpy as np

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
REF = os.environ.get("ADELL_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

for name, path in [("adell_mri", "adell_mri"), ("adell_mri.utils", "adell_mri/utils")]:
    m = types.ModuleType(name)
    m.__path__ = [os.path.join(REF, path)]
    sys.modules[name] = m
for name in ("monai", "monai.data", "monai.data.meta_tensor"):
    sys.modules[name] = types.ModuleType(name)


class MetaTensor(torch.Tensor):
    """Stand-in for the isinstance checks of make_meta (no plain tensor is one)."""


sys.modules["monai.data.meta_tensor"].MetaTensor = MetaTensor
sys.modules["monai"].data = sys.modules["monai.data"]
sys.modules["monai.data"].meta_tensor = sys.modules["monai.data.meta_tensor"]

from adell_mri.utils.inference import (FlippedInference, SegmentationInference,  # noqa: E402
                                       SlidingWindowSegmentation)
from cases import INFERENCE_CASES, inference_net  # noqa: E402


def main():
    out = {}
    for i, (name, (shape, kind, kw)) in enumerate(INFERENCE_CASES.items()):
        x = torch.randn(shape, generator=torch.Generator().manual_seed(100 + i))
        X = {"image": x, "meta": "not a tensor"} if kw.get("as_dict") else x
        if kind == "sliding":
            op = SlidingWindowSegmentation(sliding_window_size=kw["window"],
                                           inference_function=inference_net(kw["n_classes"]),
                                           n_classes=kw["n_classes"], stride=kw["stride"],
                                           inference_batch_size=kw["batch"])
        elif kind == "flip":
            op = FlippedInference(inference_net(kw["n_out"]), flips=kw["flips"],
                                  flip_keys=kw.get("flip_keys"))
        else:
            # n_classes = 2 -> one output channel (inference.py:874: n_classes if > 2 else 1)
            op = SegmentationInference(base_inference_function=inference_net(1),
                                       sliding_window_size=kw["window"], stride=kw["stride"],
                                       inference_batch_size=kw["batch"],
                                       n_classes=kw["n_classes"], flip=kw["flip"])
        keep = x.clone()
        y = op(X)
        assert torch.equal(x, keep), name
        out[name + "/x"] = x.numpy()
        out[name + "/y"] = y.numpy().astype(np.float32)
        print(name, tuple(y.shape), float(y.mean()))
    path = os.path.join(ROOT, "tests", "golden", "inference_ops.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
