"""molgym.ppo on the HIP hot path (molgym_amd/ppo.py mirrors /root/reference/molgym/ppo.py name for name)."""
from molgym_amd.ppo import *  # noqa: F401,F403
from molgym_amd.ppo import (batch_ppo, batch_rollout, collect_data_batch, compute_buffer_stats, compute_loss,  # noqa: F401
                            compute_mean_dict, gather_rollout, get_batch_generator, train)
