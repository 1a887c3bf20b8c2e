from .eas import (cross, cross_randflip, cross_sparseflip, evolve_states, fitparents, randflip,  # noqa: F401
                  randparents, sparseflip)
from .utils import init_states, vary_Kn  # noqa: F401
