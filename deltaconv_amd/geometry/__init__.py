from .operators import *          # noqa: F401,F403
from .grad_div_mls import *       # noqa: F401,F403
from .graph import Graph, knn_graph, as_graph  # noqa: F401
from .utils import batch_dot      # noqa: F401
from .fps import geodesic_fps    # noqa: F401
