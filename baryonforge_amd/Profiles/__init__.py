from .BaryonCorrection import *  # noqa: F401,F403
from . import BaryonCorrection  # noqa: F401
