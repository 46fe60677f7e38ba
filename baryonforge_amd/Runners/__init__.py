from .HealpixRunner import *  # noqa: F401,F403
from .SnapshotRunner import *  # noqa: F401,F403
from .Map2DRunner import *  # noqa: F401,F403
