"""Registry with the reference's names (reparameterizers/__init__.py:12-46)."""
from .beta import Beta
from .mixture import Mixture
from .bernoulli import Bernoulli
from .gumbel import GumbelSoftmax
from .isotropic_gaussian import IsotropicGaussian
from .concat_reparameterizer import ConcatReparameterizer
from .sequential_reparameterizer import SequentialReparameterizer
from .flow import FlowReparameterizer

reparam_dict = {
    'flow': FlowReparameterizer,
    'beta': Beta,
    'bernoulli': Bernoulli,
    'discrete': GumbelSoftmax,
    'isotropic_gaussian': IsotropicGaussian,
    'mixture': Mixture,
    'concat': ConcatReparameterizer,
    'sequential': SequentialReparameterizer,
}


def get_reparameterizer(reparam_type_str):
    assert reparam_type_str in reparam_dict, "Unknown reparameterizer requested: {}".format(reparam_type_str)
    return reparam_dict[reparam_type_str]


def is_module_a_reparameterizer(module):
    return isinstance(module, tuple(reparam_dict.values()))
