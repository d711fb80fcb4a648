"""xpt_mde_2021_amd -- MI355X-native hot path of goodgodgd/xpt-mde-2021.

Self-supervised depth/pose training step (DepthNet / PoseNet, differentiable view synthesis,
multi-scale photometric + SSIM + smoothness loss) behind the reference's own entry points
(``config.py`` / ``model/model_main.py`` / ``model/train_val.py``).  The synthesis and loss
arithmetic runs in hand-written gfx950 HIP kernels (``csrc/``) behind the C ABI declared in
``include/xpt_hip.h``; there is no CPU fallback for those ops.
"""
__version__ = "0.1.0"

import os as _os

# MIOpen's FAST find mode for the few dense convolutions that still go through the library (train_val.configure_backend);
# set at import so that it precedes MIOpen's initialisation.
_os.environ.setdefault("MIOPEN_FIND_MODE", "2")
