from ._models import Model  # noqa: F401
from .bsc import BSC  # noqa: F401
from .sssc import SSSC  # noqa: F401
