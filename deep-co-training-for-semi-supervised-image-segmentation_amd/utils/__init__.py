from .utils import *  # noqa: F401,F403
from .utils import iterator_, map_, dict_merge, fix_all_seed  # noqa: F401
