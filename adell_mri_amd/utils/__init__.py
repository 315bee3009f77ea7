from .utils import ExponentialMovingAverage  # noqa: F401
from .inference import (FlippedInference, SegmentationInference,  # noqa: F401
                        SlidingWindowSegmentation, TensorListReduction)
