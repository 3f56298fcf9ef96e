"""mixgan-tts_amd -- MI355X-native (gfx950) diffusion hot path of MixGAN-TTS.

Host-side mirror of the reference's module surface (same class names, constructor and
forward signatures, state_dict keys) over hand-written HIP kernels reached through the C ABI
in include/mixgan_hip.h.  Import as `mixgan_tts_amd` (see ../mixgan_tts_amd.py).

There is no CPU or eager-PyTorch fallback: a forward on a machine without the built
`libmixgan_hip.so` / without a GPU raises.
"""
from ._lib import lib, library_path, MixganHipError  # noqa: F401
from .schedule import beta_schedule, diffusion_buffers  # noqa: F401
from .blocks import ConvNorm, LinearNorm, DiffusionEmbedding, Mish, ResidualBlock  # noqa: F401
from .denoiser import Denoiser  # noqa: F401
from .diffusion import GaussianDiffusion  # noqa: F401
from .discriminator import JCUDiscriminator  # noqa: F401
from . import ops, autograd, losses, distributed  # noqa: F401
from .train_step import HotPathTrainer, AuxTrainer  # noqa: F401
from . import lingops, vocoder, data  # noqa: F401
from . import optimizer  # noqa: F401
from .optimizer import ScheduledOptim, FlatAdam  # noqa: F401
from .distributed import GradBucket  # noqa: F401
from .mixgantts import MixGANTTS, get_mask_from_lengths  # noqa: F401
from .transformer import Decoder, FFTBlock, PostNet, MultiHeadAttention, PositionwiseFeedForward  # noqa: F401
from .model_io import get_model, save_checkpoint, get_param_num  # noqa: F401

__version__ = "0.1.0"
