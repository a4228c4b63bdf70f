"""MI355X-native train step of relgukxilef/GAN-Class-Transfer2's denoising U-Net (train.py).

The directory name carries a hyphen (project convention); import it as `gan_class_transfer2_amd`
(the alias package next to it extends its __path__ here).
"""
from . import _lib, build, data, engine, model, sampler # noqa: F401
from .data import ImageDataset                    # noqa: F401
from .sampler import log_sample, make_log_sample  # noqa: F401
from ._lib import BF16, F16, F32, Gct2Error       # noqa: F401
from .engine import Topology, UNetEngine          # noqa: F401
from .model import (Adam, Block, Dense, Denoiser, DownShuffle, LambdaCallback, LossScaleOptimizer,  # noqa: F401
                    Residual, Sequential, Trainer, UpShuffle, WarmUp, alpha_dash, configure, identity)
