from .raymarching import *  # noqa: F401,F403
from .raymarching import MarchArena, march_rays_train_arena  # noqa: F401
