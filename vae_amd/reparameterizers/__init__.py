"""Name -> class lookup of the HIP-backed reparameterizers.

Same three public names and string keys as the reference's registry
(reparameterizers/__init__.py:12-46) so `get_reparameterizer('discrete')(config=cfg)` and
`is_module_a_reparameterizer(layer)` keep working in SimpleVAE / VRNN / the Parallel and Sequential
VAEs; every class behind a key launches kernels from libvae_amd.so (no eager fallback).
"""
from . import bernoulli, beta, concat_reparameterizer, flow, gumbel, isotropic_gaussian, mixture
from . import sequential_reparameterizer

IsotropicGaussian = isotropic_gaussian.IsotropicGaussian
GumbelSoftmax = gumbel.GumbelSoftmax
Bernoulli = bernoulli.Bernoulli
Beta = beta.Beta
Mixture = mixture.Mixture
FlowReparameterizer = flow.FlowReparameterizer
ConcatReparameterizer = concat_reparameterizer.ConcatReparameterizer
SequentialReparameterizer = sequential_reparameterizer.SequentialReparameterizer

reparam_dict = dict(
    isotropic_gaussian=IsotropicGaussian,
    discrete=GumbelSoftmax,
    bernoulli=Bernoulli,
    beta=Beta,
    mixture=Mixture,
    flow=FlowReparameterizer,
    concat=ConcatReparameterizer,
    sequential=SequentialReparameterizer,
)
_ALL_TYPES = tuple(reparam_dict.values())


def get_reparameterizer(reparam_type_str):
    """The class registered under `reparam_type_str` (AssertionError for an unknown key, as the reference)."""
    cls = reparam_dict.get(reparam_type_str)
    assert cls is not None, "Unknown reparameterizer requested: {}".format(reparam_type_str)
    return cls


def is_module_a_reparameterizer(module):
    """True for an instance of any registered class (used by the Sequential chain to tell the
    dense projections from the reparameterizers)."""
    return isinstance(module, _ALL_TYPES)
