"""ClassConditionedPriorVAE's latent half with the reference's semantics (conditioned.py:44-66
_clamp_variance, :89-110 _process_labels, :112-135 reparameterize, :137-147 kld).

The label projector (`self.prior_mlp`), encoder and decoder are the host model's stock PyTorch.  What
is here is the pair "posterior from the encoder logits, prior from the projected labels, KL between
them [+ KL(prior || N(0,1))]" -- the same pattern as a VRNN step, so for an IsotropicGaussian it is the
same single fused launch (`functional.gauss_pair`: both draws, the sigmoid variance clamp, both KLs),
with the KLs riding on the returned logvar tensors so that kld() costs nothing.
"""
import torch
import torch.nn.functional as F

from . import vrnn
from .elbo import ELBOMixin
from .reparameterizers.isotropic_gaussian import IsotropicGaussian


class ConditionedPriorMixin(ELBOMixin):
    """Inherit next to nn.Module; expects self.config, self.reparameterizer, self.prior_mlp."""

    def _clamp_variance(self, logits):
        return vrnn.clamp_variance(logits, self.config, self.reparameterizer)

    def _project_labels(self, labels, dtype):
        """conditioned.py:99-108 without the clamp: one-hot int labels, cast, run the projector."""
        device = labels.device
        assert labels.dim() <= 2, "unknown labels with dimension {} received".format(labels.dim())
        if labels.dim() == 1:
            labels = F.one_hot(labels, self.config['output_size'])
        return self.prior_mlp(labels.type(dtype).to(device))

    def _process_labels(self, labels, dtype):
        return self._clamp_variance(self._project_labels(labels, dtype))

    def reparameterize(self, logits, labels, force=False, noise=None):
        """`noise` = (posterior noise, prior noise) to inject the two draws (posterior is drawn first)."""
        assert labels is not None, "labels needed for class-conditioned vae."
        rep = self.reparameterizer
        if isinstance(rep, IsotropicGaussian) and self.config['reparam_type'] == 'isotropic_gaussian' \
                and logits.dim() == 2:
            lm = {'encoder_logits': logits, 'prior_logits': self._project_labels(labels, logits.dtype),
                  'x_features': None}
            z, params = vrnn._reparameterize_fused(rep, lm, self.config, (None, None) if noise is None else noise,
                                                   force=force)
            z_posterior, params_posterior, params_prior = z['posterior'], params['posterior'], params['prior']
        else:
            kw = [{} if noise is None or n is None else {'noise': n} for n in (noise or (None, None))]
            z_posterior, params_posterior = rep(logits, force=force, **kw[0])
            _, params_prior = rep(self._process_labels(labels, logits.dtype), force=force, **kw[1])
        return z_posterior, {'labels': labels, 'logits': logits, 'prior': params_prior, 'posterior': params_posterior}

    def kld(self, dist):
        return vrnn.kld(self.reparameterizer, dist, self.config)
