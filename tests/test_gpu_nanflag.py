"""The sticky saw-a-NaN word (include/vae_amd.h: vae_nan_flag, VAE_NAN_*): raised by the closing waves when a batch
mean comes out NaN, read without a per-step synchronisation.  Stands in for the reference's per-step host check
nan_check_and_break (abstract_vae.py:366-369, isotropic_gaussian.py:89-90)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture
def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda', 0)


def _model(dev, every):
    import torch.nn as nn
    import vae_amd
    from vae_amd.elbo import ELBOMixin

    class Tiny(ELBOMixin, nn.Module):
        def __init__(self, config):
            nn.Module.__init__(self)
            self.config = config
            self.reparameterizer = vae_amd.get_reparameterizer('isotropic_gaussian')(config)

    config = {'reparam_type': 'isotropic_gaussian', 'continuous_size': 32, 'nll_type': 'bernoulli', 'kl_beta': 1.0,
              'cuda': True, 'half': False, 'continuous_mut_info': 0, 'discrete_mut_info': 0,
              'vae_amd_nan_check': every}
    return Tiny(config).to(dev).train()


@pytest.mark.parametrize('B', [7, 5000])      # one closing block / several
def test_clean_batches_leave_the_flag_down_and_a_nan_row_raises_it(dev, B):
    import vae_amd
    from vae_amd import functional as F_
    vae_amd.nan_flags(clear=True)
    g = torch.Generator(device=dev).manual_seed(3)
    x = (torch.rand(B, 96, generator=g, device=dev) < 0.5).float()
    recon = torch.randn(B, 96, generator=g, device=dev)
    kld = torch.rand(B, generator=g, device=dev)
    F_.nll_elbo(x, recon, 'bernoulli', kld=kld)
    assert vae_amd.nan_flags(clear=False) == 0
    bad = recon.clone()
    bad[B // 2, 5] = float('nan')
    F_.nll_elbo(x, bad, 'bernoulli', kld=kld)
    assert vae_amd.nan_flags(clear=False) == 1                    # VAE_NAN_NLL: loss, elbo and nll means
    F_.nll_elbo(x, recon, 'bernoulli', kld=kld)                   # sticky: a clean step does not lower it
    assert vae_amd.nan_flags(clear=True) == 1
    assert vae_amd.nan_flags(clear=True) == 0
    kbad = kld.clone()
    kbad[0] = float('nan')
    F_.nll_elbo(x, recon, 'bernoulli', kld=kbad)
    assert vae_amd.nan_flags(clear=True) == 1 | 2                 # the kld mean, and the loss / elbo means with it


def test_nan_in_the_encoder_logits_raises_the_latent_bit(dev):
    import vae_amd
    model = _model(dev, every=0)
    vae_amd.nan_flags(clear=True)
    logits = torch.randn(300, 32, device=dev)
    model.reparameterize(logits)
    assert vae_amd.nan_flags(clear=False) == 0
    logits[17, 3] = float('nan')
    model.reparameterize(logits)
    assert vae_amd.nan_flags(clear=True) == 4


def test_the_model_checks_every_nth_step(dev):
    import vae_amd
    model = _model(dev, every=3)
    vae_amd.nan_flags(clear=True)
    x = (torch.rand(64, 40, device=dev) < 0.5).float()
    logits = torch.randn(64, 32, device=dev)
    recon = torch.randn(64, 40, device=dev)
    bad = recon.clone()
    bad[1, 1] = float('nan')
    _, params = model.reparameterize(logits)
    model.loss_function(bad, x, params)             # call 1: raised on the device, nobody looks
    model.loss_function(recon, x, params)           # call 2
    with pytest.raises(FloatingPointError, match='nll'):
        model.loss_function(recon, x, params)       # call 3 reads the word
    model.loss_function(recon, x, params)
    model.loss_function(recon, x, params)
    model.loss_function(recon, x, params)           # call 6: cleared by the read before, stays quiet
    model.nan_check(force=True)


def test_static_entry_keeps_its_own_word(dev):
    import ctypes
    from vae_amd._lib import call, load, ptr, stream_ptr
    load()
    B, D = 33, 64
    x = (torch.rand(B, D, device=dev) < 0.5).float()
    recon = torch.randn(B, D, device=dev)
    recon[4, 4] = float('nan')
    nll, loss, elbo = (torch.empty(B, device=dev) for _ in range(3))
    means = torch.empty(8, device=dev)
    grecon = torch.empty_like(recon)
    out = ctypes.c_uint32(0)
    call('vae_nan_flag', None, 1, ctypes.byref(out), stream_ptr(dev))
    for r in (recon, torch.randn(B, D, device=dev)):
        call('vae_nll_elbo_fused_static', ptr(x), ptr(r), 0, None, None, None, 1.0, 1.0 / B, 1.0, ptr(nll), ptr(loss),
             ptr(elbo), ptr(means), ptr(grecon), B, D, 0, stream_ptr(dev))
    call('vae_nan_flag', None, 1, ctypes.byref(out), stream_ptr(dev))
    assert out.value == 1                           # survived the second call's counter reset
    call('vae_nan_flag', None, 0, ctypes.byref(out), stream_ptr(dev))
    assert out.value == 0
