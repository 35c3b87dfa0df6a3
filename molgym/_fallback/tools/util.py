from molgym_amd.tools.util import *  # noqa: F401,F403
from molgym_amd.tools.util import (InfoSaver, RolloutSaver, compute_gradient_norm, count_vars, discount_cumsum,  # noqa: F401
                                   get_optimizer, init_device, set_seeds, to_numpy)
