"""MI355X-native 2D->3D pose-lifting train path (the hot path of RHnejad/3D_PoseEstimation).

Import as `importlib.import_module("3d_poseestimation_amd")` or through the `poselift`
alias module at the repository root.
"""
from ._lib import PoseliftError, lib  # noqa: F401
from .model import Linear, LinearModel, weight_init  # noqa: F401
from .optim import FlatAdamW  # noqa: F401
from .arena import FlatAdam, ModuleArena  # noqa: F401
from .train import (cycle_step, epoch_mpjpe_mm, eval_step, flip_average, flip_frames_nhwc, flip_pose, loss_MPJPE, mse_loss,  # noqa: F401
                    predict_flip_tta, train_step, GraphedTrainStep, GraphedModuleStep)
from .heads import soft_argmax_2d, soft_argmax_3d, soft_argmax_3d_nhwc  # noqa: F401
from .losses import TriangleLoss, l1_loss, l1_terms  # noqa: F401
from .backbone import Model_2D, Model_3D, ResNet  # noqa: F401
from .data import PoseFeeder, epoch_indices  # noqa: F401
from . import arena, backbone, conv, data, dp, layout, synth  # noqa: F401
