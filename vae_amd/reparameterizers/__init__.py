"""Registry with the reference's names (reparameterizers/__init__.py:12-46)."""
from .isotropic_gaussian import IsotropicGaussian

reparam_dict = {
    'isotropic_gaussian': IsotropicGaussian,
}


def get_reparameterizer(reparam_type_str):
    assert reparam_type_str in reparam_dict, "Unknown reparameterizer requested: {}".format(reparam_type_str)
    return reparam_dict[reparam_type_str]


def is_module_a_reparameterizer(module):
    return isinstance(module, tuple(reparam_dict.values()))
