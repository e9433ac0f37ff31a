"""MI355X-native ESRGAN x4 generator hot path (HIP kernels behind the reference's module API)."""
from ._lib import DTYPE_BF16, DTYPE_F32, LIB_PATH, Sr355Error, build  # noqa: F401
from . import ops  # noqa: F401  (registers the sr355:: torch.library ops)
from .esrgan_model import GeneratorRRDB, RRDB, RRDBNet, ResidualDenseBlock  # noqa: F401
from .esrgan_aux import Discriminator, VGGFeatureExtractor, adversarial_step, pixel_step  # noqa: F401
from .sharded import sharded_forward, tiled_forward  # noqa: F401
from .optim import FlatAdam  # noqa: F401
from . import cvae  # noqa: F401  (CVAE of the conditional-VAE script)
from . import cyclegan  # noqa: F401  (GeneratorSR / ResidualBlock of the CycleGAN scripts; not re-exported: the
#                                       ESRGAN file has a different class under the name Discriminator)

__all__ = ["GeneratorRRDB", "RRDBNet", "RRDB", "ResidualDenseBlock", "Discriminator", "VGGFeatureExtractor",
           "pixel_step", "adversarial_step", "FlatAdam", "cyclegan", "cvae", "sharded_forward", "tiled_forward", "ops", "build",
           "DTYPE_BF16", "DTYPE_F32", "LIB_PATH", "Sr355Error"]
