"""mrdis -- MI355X-native (gfx950) hot path of the multi-modal MR representation-
disentanglement training step (reference: ouyangjiahong/representation-disentanglement,
src/model.py + src/main_missing.py).

The directory name carries a hyphen (`representation-disentanglement_amd`), so
import it through the `mrdis` alias at the repository root:

    import mrdis
    conv = mrdis.Conv2d(is_cond=True)(7, 32, 4, 2, padding=1)      # reference's factory

Layout: csrc/ (hand-written HIP kernels + C ABI, include/mrdis.h), hip.py (ctypes
binding), ops.py (autograd pairing + torch.ops.mrdis.*), model.py (mirror of the
reference's module interface), trainer.py (train step, flat-arena Adam, gradient
all-reduce), train.py (the runnable main_missing.py-equivalent: epoch loop, scheduler, stat.csv, checkpoints).  No CPU fallback exists for the hot path.
"""
from . import hip, ops, model, model3d, trainer               # noqa: F401
from .hip import MrdisError, MrdisLibraryError, LIB_PATH      # noqa: F401
from .model import (CondConv2d, Conv2d, HipConv2d, BatchNorm2d, Conv_BN_Act_New,          # noqa: F401
                    Act_Deconv_BN_Concat_New, AnatomyEncoderEncNew, AnatomyEncoderDecNew,
                    ModalityEncoderNew, SPADEBlockNew, SPADENewShared, SPADENewNotShared,
                    Discriminator, MultimodalModel, expand_type)
from .trainer import (TrainStep, GraphedTrainStep, make_train_step, regular_mask, EvalStep, ArenaAdam, GradAllReduce, DEFAULT_CONFIG, load_config_yaml,   # noqa: F401
                      derive_config, build_model, synthetic_batch, fit_to_model, forward_losses,
                      save_checkpoint, load_checkpoint_model, LOSS_KEYS)

from .model3d import BasicBlock, VAEBranch, UNet3D, NVNet3D, HipConv3d, nvnet_loss   # noqa: F401
from .data import VolumeStore, SliceDataset, BatchLoader, load_idx_list   # noqa: F401
from . import train   # noqa: F401,E402

__version__ = '0.1.0'
