"""MI355X-native hot path of LTS4/graph-learning: LogModel / LGMRF behind a C ABI.

Host side is NumPy + ctypes only (no PyTorch); all compute is in HIP kernels for
gfx950 in ``libgraphlearn_amd.so``.  There is no CPU fallback.
"""

from ._lib import Context, GraphLearnError, LogDegreeSession, default_context, device_count
from .evaluation import relative_error
from .lgmrf import LGMRF, nonnegative_qp_solver
from .smooth_learning import LogModel, get_theta, gsp_learn_graph_log_degrees

__all__ = [
    "Context",
    "GraphLearnError",
    "LGMRF",
    "LogDegreeSession",
    "LogModel",
    "default_context",
    "device_count",
    "get_theta",
    "gsp_learn_graph_log_degrees",
    "nonnegative_qp_solver",
    "relative_error",
]
__version__ = "0.1.0"
