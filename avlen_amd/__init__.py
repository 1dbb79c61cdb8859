"""avlen_amd -- MI355X (gfx950) implementation of the AVLEN / SAVi PPO rollout-and-update hot path.

Drop-in mirror of ss_baselines/savi/ppo/policy.py (Policy.act* / evaluate_actions*), ppo.py (PPO.update) and
models/rollout_storage.py; all arithmetic runs in hand-written HIP kernels behind the C ABI of
include/avlen_hip.h (libavlen_hip.so).  There is no CPU or PyTorch fallback.
"""
__version__ = "0.1.0"

from . import config as _config
_config.check_hw_queues()
