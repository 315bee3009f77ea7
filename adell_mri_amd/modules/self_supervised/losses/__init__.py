from .functional import byol_loss, simsiam_loss  # noqa: F401
from .ntxent import NTXentLoss  # noqa: F401
from .vicreg import VICRegLoss  # noqa: F401
