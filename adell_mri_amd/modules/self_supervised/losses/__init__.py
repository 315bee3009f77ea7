from .vicreg import VICRegLoss  # noqa: F401
