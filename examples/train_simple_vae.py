#!/usr/bin/env python3
"""What a host model looks like once the hot path is swapped in (simple_vae.py + abstract_vae.py of the reference,
restated with nn.Linear networks and synthetic data): the encoder / decoder are stock PyTorch, the latent and loss
half comes from vae_amd (ELBOMixin + a reparameterizer class from the registry).

    python examples/train_simple_vae.py --steps 50                       # one GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 \
        examples/train_simple_vae.py --steps 50                         # one process per GPU, RCCL

Multi-GPU: every rank owns B rows of the global batch.  vae_amd.dist.configure() points the reparameterizer at its
shard (global row offset for the in-kernel Philox draws, 1/B_global scaling of the batch means); globalize_loss() is
the one packed all-reduce of the loss scalars; the network gradients live in ONE flat buffer (vae_amd.dist.
FlatGradients) that is summed over ranks as one message after backward -- the kernels already divide by B_global, so
the sum IS the global-batch gradient.  The code path is the same under RCCL (the default, one GPU per rank) and under
EXAMPLE_BACKEND=gloo (several ranks rehearsing on one GPU): only the collective underneath differs.  --ddp swaps the
flat buffer for torch's DistributedDataParallel (bucketed, overlapped with backward; RCCL only -- it averages, so the
loss is multiplied by world_size)."""
import argparse
import os
import sys
import time

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vae_amd  # noqa: E402
from vae_amd import dist as vdist  # noqa: E402
from vae_amd.elbo import ELBOMixin  # noqa: E402


class SimpleVAE(ELBOMixin, nn.Module):
    def __init__(self, config):
        nn.Module.__init__(self)
        self.config = config
        self.reparameterizer = vae_amd.get_reparameterizer(config['reparam_type'])(config)
        d_in = 28 * 28
        self.encoder = nn.Sequential(nn.Linear(d_in, 512), nn.ReLU(), nn.Linear(512, self.reparameterizer.input_size))
        self.decoder = nn.Sequential(nn.Linear(self.reparameterizer.output_size, 512), nn.ReLU(), nn.Linear(512, d_in))

    def forward(self, x):
        z, params = self.reparameterize(self.encoder(x.flatten(1)))
        return self.decoder(z).view_as(x), params


def main():
    if os.environ.get('EXAMPLE_DEBUG'):      # where is every thread if a run stalls?
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ['EXAMPLE_DEBUG']), exit=True)
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--batch', type=int, default=4096, help='rows per GPU')
    ap.add_argument('--latent', type=int, default=64)
    ap.add_argument('--buckets', type=int, default=1,
                    help='cut the flat gradient buffer into this many pieces, each all-reduced as soon as backward has filled it')
    ap.add_argument('--ddp', action='store_true', help="torch DDP instead of vae_amd.dist.FlatGradients (RCCL only)")
    args = ap.parse_args()
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    # EXAMPLE_BACKEND=gloo lets several ranks share one GPU (a rehearsal of the wiring on a one-GPU box)
    backend = os.environ.get('EXAMPLE_BACKEND', 'nccl')
    dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', 0)) % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    if world > 1:
        if backend == 'nccl':
            os.environ.setdefault('TORCH_NCCL_HIGH_PRIORITY', '1')   # small collectives beside full-chip launches
            torch.distributed.init_process_group('nccl', device_id=dev)
        else:
            torch.distributed.init_process_group(backend)
    config = {'reparam_type': 'isotropic_gaussian', 'continuous_size': 2 * args.latent, 'nll_type': 'bernoulli',
              'kl_beta': 1.0, 'cuda': True, 'half': False, 'continuous_mut_info': 0, 'discrete_mut_info': 0}
    torch.manual_seed(0)
    model = SimpleVAE(config).to(dev).train()
    net = model
    ddp = world > 1 and args.ddp
    if ddp and backend != 'nccl':
        raise SystemExit('--ddp needs one GPU per rank over RCCL (EXAMPLE_BACKEND=nccl)')
    if world > 1:
        vdist.configure(model, args.batch * world)
    if ddp:
        net = nn.parallel.DistributedDataParallel(model, device_ids=[dev.index])
    flat = None if ddp else vdist.FlatGradients(model.parameters(), buckets=args.buckets)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    gen = torch.Generator(device=dev).manual_seed(1 + rank)
    proto = (torch.rand(16, 1, 28, 28, generator=gen, device=dev) < 0.15).float()      # 16 "digit" prototypes
    t0 = time.perf_counter()
    for step in range(args.steps):
        idx = torch.randint(0, 16, (args.batch,), generator=gen, device=dev)
        flip = (torch.rand(args.batch, 1, 28, 28, generator=gen, device=dev) < 0.02).float()
        x = (proto[idx] + flip).remainder(2)
        recon, params = net(x)
        loss = model.loss_function(recon, x, params)
        if world > 1:
            loss, _ = vdist.globalize_loss(loss)
        if ddp:      # DDP averages the per-rank gradients; they have to SUM to the global-batch gradient
            opt.zero_grad(set_to_none=True)
            (loss['loss_mean'] * world).backward()
        else:
            flat.zero()
            loss['loss_mean'].backward()
            flat.all_reduce()       # identity on one rank
        opt.step()
        if rank == 0 and (step % 10 == 0 or step == args.steps - 1):
            print('step %4d  loss %.3f  nll %.3f  kld %.3f' % (step, float(loss['loss_mean']), float(loss['nll_mean']),
                                                              float(loss['kld_mean'])), flush=True)
    torch.cuda.synchronize()
    if rank == 0:
        dt = time.perf_counter() - t0
        print('%d steps, %.0f samples/s (whole model, %d GPU%s)' % (args.steps, args.steps * args.batch * world / dt,
                                                                    world, 's' if world > 1 else ''))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
