"""Build container only (needs /root/reference): time the REAL reference's reparameterize + kl + loss_function
(forward + backward) next to oracle/cpu_ref.elbo_step on the same inputs and threads, to show that the oracle --
which is what bench.py's cpu_baseline times on the GPU box, where the reference cannot travel -- costs what the
reference costs.  The NLL is the oracle's in both (helpers.distributions is un-vendored).

    python scripts/time_reference_here.py
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
from make_golden import install_reference, base_config  # noqa: E402
from oracle import cpu_ref as O  # noqa: E402


def main():
    vae = install_reference()
    AV = sys.modules['models.vae.abstract_vae'].AbstractVAE
    B, L = 4096, 512
    g = torch.Generator().manual_seed(1234)
    logits = torch.randn(B, 2 * L, generator=g)
    noise = torch.randn(B, L, generator=g)
    x = (torch.rand(B, 1, 28, 28, generator=g) < 0.13).float()
    recon = torch.randn(B, 1, 28, 28, generator=g)
    cfg = base_config(continuous_size=2 * L, nll_type='bernoulli')
    rep = vae.reparameterizers.IsotropicGaussian(cfg)
    rep.train(True)

    class Stub(object):
        training = True
        config = cfg
        kl_annealer = None
        reparameterizer = rep
        compute_kl_beta = AV.compute_kl_beta

        def nll(self, x_, r_, t_):
            return O.nll(x_, r_, t_)

        def kld(self, p):
            return self.reparameterizer.kl(p)

        def mut_info(self, p, b):
            return torch.zeros(b)

    st = Stub()

    def ref_step():
        lg = logits.detach().requires_grad_(True)
        rc = recon.detach().requires_grad_(True)
        z, params = rep(lg)
        out = AV.loss_function(st, rc, x, params)
        (out['loss_mean'] + z.mean()).backward()

    def oracle_step():
        O.elbo_step(logits, noise, x, recon)

    for name, fn in (('reference', ref_step), ('oracle', oracle_step)):
        fn()
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 8.0:
            fn()
            n += 1
        dt = time.perf_counter() - t0
        print('%-9s %d threads: %.1f ms / step of B=%d, L=%d, 1x28x28 Bernoulli -> %.3f M samples/s'
              % (name, torch.get_num_threads(), dt / n * 1e3, B, L, B * n / dt / 1e6))


if __name__ == '__main__':
    main()
