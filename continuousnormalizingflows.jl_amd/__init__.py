"""MI355X-native hot path of impICNF/ContinuousNormalizingFlows.jl: hand-written HIP kernels
behind a C ABI (include/cnf.h, csrc/), plus the host-side mirror of the reference's
ICNF / inference / generate / loss interface (icnf.py) and the column-sharding helper.

The directory name follows the repository contract (`continuousnormalizingflows.jl_amd`) and is
not a valid Python identifier; `__graft_entry__.load_package()` registers it as `cnf_amd`.
"""
from . import _lib
from ._lib import get_tuning, reload_tuning, set_tuning  # noqa: F401
from .icnf import *  # noqa: F401,F403
from .icnf import loss_sums, loss_mean  # noqa: F401
from .icnf import loss_and_gradient  # noqa: F401
from .sharding import Comm, get_comm, reduce_gradient, reduce_loss, set_comm, shard_columns  # noqa: F401
from .adapters import *  # noqa: F401,F403
from .adapters import epoch_batches  # noqa: F401
