from .nonlin import *      # noqa: F401,F403
from .mlp import *         # noqa: F401,F403
from .deltaconv import *   # noqa: F401,F403
