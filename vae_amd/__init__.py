"""vae_amd: the MI355X-native reparameterizer sample + KL + NLL + ELBO hot path.

Layout: csrc/ (HIP kernels + C ABI -> libvae_amd.so), _lib.py (ctypes binding),
functional.py (autograd entry points), reparameterizers/ (the reference's module classes),
elbo.py (AbstractVAE.loss_function / nll / kld and the VRNN aggregation), dist.py (batch
sharding + the packed scalar all-reduce).
"""
from . import _lib  # noqa: F401
from ._lib import nan_flags  # noqa: F401
from . import functional  # noqa: F401
from . import graph  # noqa: F401
from .reparameterizers import get_reparameterizer, is_module_a_reparameterizer, reparam_dict  # noqa: F401
from .elbo import ELBOMixin, MSGMixin, AutoencoderMixin, VAENoKLMixin, loss_function, nll, vrnn_loss_function  # noqa: F401

EPS_TOL = 1e-6
