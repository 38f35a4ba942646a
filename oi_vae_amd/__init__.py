"""oi_vae_amd -- MI355X-native oi-VAE training-step engine.

Drop-in for the model API of samuela/oi-vae's ``lib`` package (NormalNet,
BayesianGroupLassoGenerator, Normal, NormalPriorTheta, OIVAE, Lambda,
KL_Normals): same names, arguments and return values, with the training step
executed by hand-written gfx950 HIP kernels behind a C ABI
(include/oivae_hip.h).
"""
from ._lib import OivaeError  # noqa: F401
from .engine import Engine, ModelSpec  # noqa: F401
from .distributions import DistributionCat, Normal  # noqa: F401
from .models import BayesianGroupLassoGenerator, FirstLayerSparseDecoder, NormalNet  # noqa: F401
from .oivae import OIVAE, NormalPriorTheta  # noqa: F401
from .utils import KL_Normals, Lambda  # noqa: F401
from . import optim, parallel  # noqa: F401
