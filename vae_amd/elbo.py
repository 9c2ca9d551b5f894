"""AbstractVAE's ELBO methods with the reference's signatures (abstract_vae.py:319-388, :400-410,
:451-471, :506-520) and the VRNN aggregation (vrnn.py:723-733, :767-829; additive_vrnn.py:78-104),
routed through the fused NLL+ELBO kernels.

`ELBOMixin` is what a host model class inherits (it supplies loss_function / nll / kld /
reparameterize / mut_info / compute_kl_beta and expects `self.config`, `self.reparameterizer`,
optionally `self.kl_annealer`, `self.decode`).  The free functions are the same code for callers
that hold tensors only.
"""
import torch

from . import functional as F_
from . import rng

MEAN_KEYS = ('loss_mean', 'elbo_mean', 'nll_mean', 'kld_mean', 'additional_loss_mean', 'mut_info_mean')


def nll(x, recon_x, nll_type):
    """[B] negative log-likelihood (abstract_vae.py:461-471)."""
    return F_.nll(x, recon_x, nll_type)


def _pack(loss, elbo, means, kl_beta, dtype):
    means = means.to(dtype)
    return {
        'loss': loss,
        'elbo': elbo,
        'loss_mean': means[0],
        'elbo_mean': means[1],
        'nll_mean': means[2],
        'kld_mean': means[3],
        'additional_loss_mean': means[4],
        'kl_beta_scalar': kl_beta,
        'mut_info_mean': means[5],
    }


def reparameterize_multi(reparameterizer, logits, num_samples, force=False, noise=None):
    """The reference's "call the reparameterizer N times on the same logits" loops (msg.py:63-69,
    abstract_vae.py:348-350) -> (z_list, params_list).  One launch when the reparameterizer has a
    multi-sample kernel (IsotropicGaussian), the plain loop otherwise."""
    if hasattr(reparameterizer, 'forward_multi'):
        return reparameterizer.forward_multi(logits, num_samples, force=force, noise=noise)
    pairs = [reparameterizer(logits, force=force) for _ in range(num_samples)]
    return [p[0] for p in pairs], [p[1] for p in pairs]


def loss_function(recon_x, x, kld, nll_type, kl_beta=1.0, mut_info=None, extra_loss_terms=None, inv_count=None,
                  nll_given=None, fuse_backward=True):
    """The loss dict of abstract_vae.py:377-388 from tensors.  One fused launch (+ a 1-block finalize)
    when the NLL comes from (x, recon_x); a LIST of decoder outputs in `recon_x` is the K > 1 case
    (:347-354: nll averaged over the list, x read once); `nll_given` [B] uses the combine kernel.
    `fuse_backward` (K == 1, recon_x requiring grad): d(loss_mean)/d(recon_x) is formed by the forward launch
    itself, so a training step reads x and recon_x once, not twice; any other backward use falls back to the
    general backward kernel with identical results."""
    add = None
    if extra_loss_terms:
        vals = list(extra_loss_terms.values())
        add = vals[0] if len(vals) == 1 else torch.sum(torch.stack(vals, 0), 0)
    if nll_given is None and isinstance(recon_x, (list, tuple)):
        _, loss, elbo, means = F_.nll_multi_elbo(x, list(recon_x), nll_type, kld=kld, add=add, mi=mut_info,
                                                 kl_beta=kl_beta, inv_count=inv_count)
    elif nll_given is None:
        _, loss, elbo, means = F_.nll_elbo(x, recon_x, nll_type, kld=kld, add=add, mi=mut_info, kl_beta=kl_beta,
                                           inv_count=inv_count, fuse_backward=fuse_backward)
    else:
        loss, elbo, means = F_.elbo_combine(nll_given, kld=kld, add=add, mi=mut_info, kl_beta=kl_beta,
                                            inv_count=inv_count)
    return _pack(loss, elbo, means, kl_beta, loss.dtype)


def clamp_mut_info(config, mut_info):
    """abstract_vae.py:473-488."""
    strategy = config['mut_clamp_strategy'].strip().lower()
    if strategy == 'norm':
        return mut_info / torch.norm(mut_info, p=2)
    if strategy == 'clamp':
        return torch.clamp(mut_info, min=-config['mut_clamp_value'], max=config['mut_clamp_value'])
    if strategy != 'none':
        raise KeyError(strategy)
    return mut_info


def vrnn_loss_function(step_losses):
    """vrnn.py:767-829: per-step loss dicts summed, then every entry divided by the step count."""
    agg = None
    for lt in step_losses:
        if agg is None:
            agg = {**lt, 'count': 1}
            continue
        for k, v in lt.items():
            agg[k] = agg[k] + v
        agg['count'] += 1
    for k in agg.keys():
        if k != 'count':
            agg[k] = agg[k] / agg['count']
    return agg


class ELBOMixin(object):
    """Drop-in for the ELBO half of AbstractVAE (the encoder/decoder half stays the host's)."""

    kl_annealer = None
    inv_count = None  # 1/B_global when the batch is sharded (vae_amd.dist)
    _nan_check_calls = 0

    def nan_check(self, force=False):
        """The reference checks nll and kld for NaN on the host at every step (abstract_vae.py:366-369: one stream
        synchronisation per step).  Here the closing wave of the kernels raises a sticky device bit when a batch mean
        comes out NaN, at no cost; config['vae_amd_nan_check'] = N reads it every N-th loss_function call (one
        synchronisation per N steps) and raises FloatingPointError naming what went NaN.  0 / absent: never read
        (call model.nan_check(force=True) or vae_amd.nan_flags() at the end of an epoch)."""
        every = int(self.config.get('vae_amd_nan_check', 0) or 0)
        if not force:
            if every <= 0 or rng.capturing():   # reading the word synchronises the stream: never inside a graph capture
                return
            self._nan_check_calls += 1
            if self._nan_check_calls % every:
                return
        from ._lib import describe_nan_flags, nan_flags
        flags = nan_flags(clear=True)
        if flags:
            raise FloatingPointError("vae_amd: NaN in " + describe_nan_flags(flags))

    def compute_kl_beta(self, kl_beta_list):
        if self.kl_annealer is not None:
            kl_beta_list = self.kl_annealer(kl_beta_list)
        return kl_beta_list

    def reparameterize(self, logits, labels=None, force=False):
        return self.reparameterizer(logits, force=force)

    def kld(self, dist_a):
        return self.reparameterizer.kl(dist_a)

    def reparameterize_aggregate_posterior(self):
        """abstract_vae.py:159-172: a training-mode draw from the EMA of the encoder logits that the host model keeps
        in `self.aggregate_posterior` (helpers.layers.EMA: stock, un-vendored)."""
        training_tmp = self.reparameterizer.training
        self.reparameterizer.train(True)
        ema = self.aggregate_posterior.ema_val
        labels = torch.arange(self.config['output_size'], device=ema.device)
        z_samples, _ = self.reparameterize(ema, labels=labels)
        self.reparameterizer.train(training_tmp)
        return z_samples

    def nll(self, x, recon_x, nll_type):
        return nll(x, recon_x, nll_type)

    def mut_info(self, dist_params, batch_size):
        if self.config.get('continuous_mut_info', 0) > 0 or self.config.get('discrete_mut_info', 0) > 0:
            return self._clamp_mut_info(self.reparameterizer.mutual_info(dist_params))
        return None  # the reference's zeros[B]; the kernel treats a missing term as 0

    def _clamp_mut_info(self, mut_info):
        return clamp_mut_info(self.config, mut_info)

    def _own_nll(self):
        """True when the host class (or the instance) brings its own `nll`: the reference's loss_function calls
        self.nll (abstract_vae.py:344, :352), so an override must be honoured, not bypassed by the fused launch."""
        return getattr(self.nll, '__func__', None) is not ELBOMixin.nll

    def loss_function(self, recon_x, x, params, K=1, **extra_loss_terms):
        nll_type = self.config['nll_type']
        if self._own_nll():
            # the host's nll, called as the reference calls it (abstract_vae.py:344-354); the combine and the six means
            # are one launch on its [B] result
            nll_ = self.nll(x, recon_x, nll_type)
            if self.training:
                for _ in range(1, K):
                    z_k, _ = self.reparameterize(logits=params['logits'], labels=params.get('labels', None))
                    nll_ = nll_ + self.nll(x, self.decode(z_k), nll_type)
                nll_ = nll_ / K
            kl_beta = self.compute_kl_beta([self.config['kl_beta']])[0]
            out = loss_function(None, x, self.kld(params), nll_type, kl_beta=kl_beta,
                                mut_info=self.mut_info(params, x.size(0)), extra_loss_terms=extra_loss_terms,
                                inv_count=self.inv_count, nll_given=nll_)
            self.nan_check()
            return out
        if self.training and K > 1:
            # abstract_vae.py:347-354.  The K-1 further draws come from one launch when reparameterize() is
            # the stock one; each draw is still decoded by its own decode() call (batch-norm decoders would
            # differ otherwise); the K decoder outputs then go through ONE NLL+ELBO launch.
            stock = type(self).reparameterize is ELBOMixin.reparameterize
            if stock and hasattr(self.reparameterizer, 'forward_multi'):
                z_more, _ = reparameterize_multi(self.reparameterizer, params['logits'], K - 1)
            else:
                z_more = [self.reparameterize(logits=params['logits'], labels=params.get('labels', None))[0]
                          for _ in range(1, K)]
            recon_x = [recon_x] + [self.decode(z_k) for z_k in z_more]
        kld = self.kld(params)
        mi = self.mut_info(params, x.size(0))
        kl_beta = self.compute_kl_beta([self.config['kl_beta']])[0]
        out = loss_function(recon_x, x, kld, nll_type, kl_beta=kl_beta, mut_info=mi,
                            extra_loss_terms=extra_loss_terms, inv_count=self.inv_count)
        self.nan_check()
        return out

    def likelihood(self, loader, K=1000, noise=None):
        """abstract_vae.py:281-317: per sample, the log-sum-exp of the elbo over (K // B) * B draws of its posterior,
        lowered by log(K // B) (the reference's normalisation, not the log of the number of values: kept), averaged
        over every sample the loader yields.

        With the model in eval mode, the stock reparameterize() / kld() and an IsotropicGaussian, a minibatch costs
        a handful of launches instead of B * (K // B) Python iterations of ~40 launches on expand().contiguous()
        copies: all draws of a group of samples come from ONE read of their logits (csrc/multi.hip, z [S, n, L]), go
        through ONE decoder call, and are scored by the step-table NLL + ELBO launch (csrc/elbo.hip: up to 128 draws
        per launch read the targets where they are) whose closing launch folds the log-sum-exp over the draws.
        config['vae_amd_likelihood_rows'] (default 65536) bounds the decoder batch.  `noise`: optional iterable of
        one [S, B, L] tensor per minibatch (S = (K // B) * B, draw-major) in place of the in-kernel Philox draws.
        Anything else -- a model in training mode (batch-norm decoders see other batches), an overridden
        reparameterize / kld, another reparameterizer -- takes the reference's loop."""
        import math
        from .reparameterizers.isotropic_gaussian import IsotropicGaussian
        batched = (not self.training and type(self).reparameterize is ELBOMixin.reparameterize
                   and type(self).kld is ELBOMixin.kld and type(self.reparameterizer) is IsotropicGaussian)
        noise = iter(noise) if noise is not None else None
        with torch.no_grad():
            likelihood = []
            for minibatch, labels in loader:
                dev = 'cuda' if self.config['cuda'] else None
                minibatch = minibatch.to(dev) if dev else minibatch
                labels = labels.to(dev) if dev else minibatch
                z_logits = self.encode(minibatch)
                batch_size = z_logits.shape[0]
                passes = K // batch_size
                if batched and z_logits.dim() == 2 and passes >= 1:
                    lse = self._likelihood_minibatch(minibatch, z_logits, passes,
                                                     None if noise is None else next(noise))
                    likelihood.append(lse - math.log(passes))
                    continue
                for idx in range(batch_size):
                    z_logits_i = z_logits[idx].expand_as(z_logits).contiguous()
                    sample_i = minibatch[idx].expand_as(minibatch).contiguous()
                    label_i = labels[idx].expand_as(labels).contiguous()
                    elbo = []
                    for count in range(passes):
                        z, params = self.reparameterize(z_logits_i, labels=label_i)
                        loss_t = self.loss_function(self.decode(z), sample_i, params=params)
                        elbo.append(loss_t['elbo'])
                    flat = torch.cat([e.unsqueeze(0) for e in elbo], 0).view([-1])
                    likelihood.append((torch.logsumexp(flat, dim=0) - math.log(count + 1)).reshape(1))
            return torch.mean(torch.cat(likelihood, 0))

    def _likelihood_minibatch(self, x, z_logits, passes, noise=None):
        """[B] log-sum-exp over S = passes * B draws per sample of elbo = nll + kld (abstract_vae.py:301-315, :357)."""
        import math
        rep = self.reparameterizer
        B, L = z_logits.shape[0], rep.output_size
        S = passes * B
        if not rep.training:
            # eval-mode reparameterizer: every "draw" is the posterior mean (isotropic_gaussian.py:100-112), the S
            # values are one value: decode it once
            z, params = self.reparameterize(z_logits)
            return self.loss_function(self.decode(z), x, params=params)['elbo'] + math.log(S)
        rows = max(int(self.config.get('vae_amd_likelihood_rows', 1 << 16)), 1)
        group = max(min(rows // S, B), 1)                      # samples per decoder call
        draws = min(S, rows) if group == 1 else S              # draws per decoder call (a lone sample may need several)
        out = []
        for i0 in range(0, B, group):
            lg, xi = z_logits[i0:i0 + group], x[i0:i0 + group]
            n = lg.shape[0]
            parts = []
            for s0 in range(0, S, draws):
                Sc = min(draws, S - s0)
                nz = None if noise is None else noise[s0:s0 + Sc, i0:i0 + n]
                seed, offset = (0, 0) if nz is not None else rng.next_calls(Sc)
                z, _, kld, _ = F_.gauss_reparam_multi(lg, Sc, noise=nz, eps_tol=rep.eps_tol, seed=seed, offset=offset,
                                                      row0=rep.row_offset + i0, want_logvar=False)
                recon = self.decode(z.view(Sc * n, L))
                recon = recon.view((Sc, n) + tuple(recon.shape[1:]))
                for t0 in range(0, Sc, F_.MAX_STEPS):
                    T = min(F_.MAX_STEPS, Sc - t0)
                    parts.append(F_.nll_steps_elbo([xi] * T, [recon[t] for t in range(t0, t0 + T)],
                                                   self.config['nll_type'], kld=kld.repeat(T), with_lse=True)[6])
            out.append(parts[0] if len(parts) == 1 else torch.logsumexp(torch.stack(parts, 0), 0))
        return torch.cat(out, 0)


class MSGMixin(ELBOMixin):
    """The reparameterize / kld pair of MSGVAE (msg.py:43-69): max_time_steps draws of the same logits, KL
    from the last parameter set.  The draws are one launch for an IsotropicGaussian."""

    def kld(self, dist_list):
        return self.reparameterizer.kl(dist_list[-1])

    def reparameterize(self, logits, force=False):
        return reparameterize_multi(self.reparameterizer, logits, self.config['max_time_steps'], force=force)


class AutoencoderMixin(ELBOMixin):
    """Autoencoder's latent/loss half (autoencoder.py:37-46, :92-115): no reparameterization, the loss is the
    NLL alone -- one launch for nll, its mean and (in training) d(loss_mean)/d(recon_x)."""

    def reparameterize(self, logits, labels=None, force=False):
        return logits, {}

    def loss_function(self, recon_x, x, **unused_kwargs):
        nll_, _, _, means = F_.nll_elbo(x, recon_x, self.config['nll_type'], inv_count=self.inv_count,
                                        fuse_backward=True)
        m = means.to(nll_.dtype)
        return {'loss': nll_, 'elbo': torch.zeros_like(nll_), 'loss_mean': m[2], 'elbo_mean': 0, 'nll_mean': m[2],
                'kld_mean': 0, 'kl_beta_scalar': 0, 'proxy_mean': 0, 'mut_info_mean': 0}


class VAENoKLMixin(ELBOMixin):
    """VAENoKL.loss_function (autoencoder.py:12-32): reparameterizes as usual, the objective drops the KL."""

    def loss_function(self, recon_x, x, reparam_map):
        nll_, _, _, means = F_.nll_elbo(x, recon_x, self.config['nll_type'], inv_count=self.inv_count,
                                        fuse_backward=True)
        m = means.to(nll_.dtype)
        zero = m[3] * 0     # mean of zeros_like(nll): a device scalar, as in the reference
        return {'loss': nll_, 'loss_mean': m[2], 'elbo_mean': zero, 'nll_mean': m[2], 'kld_mean': zero,
                'proxy_mean': zero, 'mut_info_mean': zero}
