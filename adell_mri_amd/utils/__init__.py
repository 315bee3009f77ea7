from .utils import ExponentialMovingAverage  # noqa: F401
