"""GPU parity: mutual_info() of the leaf reparameterizers against the reference's own outputs
(tests/golden/mi.pt): value and the gradient w.r.t. the raw logits of x and of the re-encoded reconstruction."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def close(a, b, rtol=1e-5, atol=1e-6):
    torch.testing.assert_close(a.detach().float().cpu(), b.detach().float().cpu(), rtol=rtol, atol=atol)


@pytest.mark.parametrize('name', ['gauss', 'gumbel', 'beta'])
def test_mutual_info_golden(golden, name):
    from vae_amd.reparameterizers import IsotropicGaussian, GumbelSoftmax, Beta
    g = golden('mi')['mi_' + name]
    width = g['logits'].shape[1]
    cfg = {'continuous_size': width, 'discrete_size': width, 'cuda': True, 'half': False,
           'continuous_mut_info': g['scale'], 'discrete_mut_info': g['scale']}
    m = {'gauss': IsotropicGaussian, 'gumbel': GumbelSoftmax, 'beta': Beta}[name](cfg).train()
    la = g['logits'].to(DEV).requires_grad_(True)
    lb = g['logits_xhat'].to(DEV).requires_grad_(True)
    _, pa = m(la)
    _, pb = m(lb)
    mi = m.mutual_info({**pa, 'q_z_given_xhat': pb})
    assert mi.shape == g['mi'].shape
    tol = dict(rtol=1e-4, atol=1e-4) if name == 'beta' else dict(rtol=1e-5, atol=1e-5)   # beta: see test_gpu_beta_flow
    close(mi, g['mi'], **tol)
    ga, gb = torch.autograd.grad((mi * g['w'].to(DEV)).sum(), [la, lb])
    close(ga, g['grad_logits'], **tol)
    close(gb, g['grad_logits_xhat'], **tol)


def test_gumbel_mutual_info_large_vs_formula():
    """K = 1024 rows: the categorical-KL route against the reference's six-op formula evaluated in torch."""
    from vae_amd.reparameterizers import GumbelSoftmax
    import torch.nn.functional as F
    torch.manual_seed(0)
    B, K = 512, 1024
    m = GumbelSoftmax({'discrete_size': K, 'cuda': True, 'half': False, 'discrete_mut_info': 1.0}).train()
    la, lb = torch.randn(B, K, device=DEV) * 3, torch.randn(B, K, device=DEV) * 3
    _, pa = m(la)
    _, pb = m(lb)
    mi = m.mutual_info({**pa, 'q_z_given_xhat': pb})
    p_z = F.softmax(la.double(), -1)
    ref = -(p_z * F.log_softmax(la.double(), -1)).sum(-1) + (p_z * F.log_softmax(lb.double(), -1)).sum(-1)
    close(mi, ref.float(), rtol=1e-5, atol=1e-5)
