"""xpt_mde_2021_amd -- MI355X-native hot path of goodgodgd/xpt-mde-2021.

Self-supervised depth/pose training step (DepthNet / PoseNet, differentiable view synthesis,
multi-scale photometric + SSIM + smoothness loss) behind the reference's own entry points
(``config.py`` / ``model/model_main.py`` / ``model/train_val.py``).  The synthesis and loss
arithmetic runs in hand-written gfx950 HIP kernels (``csrc/``) behind the C ABI declared in
``include/xpt_hip.h``; there is no CPU fallback for those ops.
"""
__version__ = "0.1.0"
