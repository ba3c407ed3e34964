from .io import load_ply, save_ply  # noqa: F401
from .sample import *  # noqa: F401,F403
