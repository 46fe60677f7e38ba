from .io import *  # noqa: F401,F403
from .Tabulate import *  # noqa: F401,F403
from .Parallelize import *  # noqa: F401,F403
