from .HealpixRunner import *  # noqa: F401,F403
