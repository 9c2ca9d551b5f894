"""Autograd-aware entry points over the C ABI (one `torch.autograd.Function` per fused kernel pair).

Every function takes ROCm tensors, allocates its outputs with torch, launches on torch's
current HIP stream through `_lib.call` and never synchronises.  Shapes are flattened to rows
(`[rows, features]`) here; the module classes in `vae_amd.reparameterizers` restore the
reference's shapes.
"""
import torch

from . import _lib
from ._lib import call, check_device, dtype_code, ptr, stream_ptr, workspace

_MASK64 = (1 << 64) - 1


def _rows(t):
    """A 2-D row view with unit inner stride (no copy when the layout already allows it)."""
    if t.dim() != 2:
        t = t.reshape(-1, t.shape[-1])
    if t.stride(1) != 1 or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
        t = t.contiguous()
    return t


def _dense(t):
    return None if t is None else t.contiguous()


class _Guard(object):
    """Make the tensors' device current for the launch when it is not already."""

    def __init__(self, dev):
        self.ctx = None if dev.index == torch.cuda.current_device() else torch.cuda.device(dev)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.ctx is not None:
            self.ctx.__exit__(*a)


# --------------------------------------------------------------------------------------
# IsotropicGaussian: split + reparameterize + KL vs N(0,1) + the two mean diagnostics
# --------------------------------------------------------------------------------------
class _GaussReparamKL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, noise, eps_tol, training, seed, offset, row0, mean_scale):
        ctx.set_materialize_grads(False)
        dev = check_device(logits, noise)
        B, L = logits.shape[0], logits.shape[1] // 2
        z = torch.empty((B, L), dtype=logits.dtype, device=dev) if training else None
        logvar = torch.empty((B, L), dtype=logits.dtype, device=dev)
        kl = torch.empty((B,), dtype=logits.dtype, device=dev)
        means = torch.zeros((2,), dtype=torch.float32, device=dev)
        if mean_scale is None:
            mean_scale = 1.0 / max(B * L, 1)
        with _Guard(dev):
            call('vae_gauss_reparam_kl_fwd', ptr(logits), logits.stride(0), ptr(noise), ptr(z), ptr(logvar), ptr(kl),
                 ptr(means), ptr(workspace(dev)), B, L, eps_tol, mean_scale, int(training), seed & _MASK64,
                 offset & _MASK64, row0, dtype_code(logits), stream_ptr(dev))
        ctx.save_for_backward(logits, noise)
        ctx.cfg = (eps_tol, training, seed, offset, row0, mean_scale)
        if z is None:
            z = logits.new_empty((0,))
            ctx.mark_non_differentiable(z)
        return z, logvar, kl, means

    @staticmethod
    def backward(ctx, gz, glogvar, gkl, gmeans):
        logits, noise = ctx.saved_tensors
        eps_tol, training, seed, offset, row0, mean_scale = ctx.cfg
        dev = logits.device
        B, L = logits.shape[0], logits.shape[1] // 2
        if not training:
            gz = None
        gz, glogvar, gkl = _dense(gz), _dense(glogvar), _dense(gkl)
        gmeans = None if gmeans is None else gmeans.contiguous().float()
        glogits = torch.empty((B, 2 * L), dtype=logits.dtype, device=dev)
        with _Guard(dev):
            call('vae_gauss_reparam_kl_bwd', ptr(logits), logits.stride(0), ptr(noise), ptr(gz), ptr(glogvar),
                 ptr(gkl), ptr(gmeans), ptr(glogits), B, L, eps_tol, mean_scale, int(training), seed & _MASK64,
                 offset & _MASK64, row0, dtype_code(logits), stream_ptr(dev))
        return glogits, None, None, None, None, None, None, None


def gauss_reparam_kl(logits, noise=None, eps_tol=1e-6, training=True, seed=0, offset=0, row0=0, mean_scale=None):
    """logits [B, 2L] -> (z [B, L] or None in eval, logvar [B, L], kl [B], means float32[2]).

    Reference: reparameterizers/isotropic_gaussian.py:77-131, :158-170, :204-215.
    `noise` None in training draws Philox N(0,1) in the kernel keyed by (seed, offset, row0+row).
    """
    logits = _rows(logits)
    if noise is not None:
        noise = noise.reshape(logits.shape[0], logits.shape[1] // 2).contiguous()
    z, logvar, kl, means = _GaussReparamKL.apply(logits, noise, float(eps_tol), bool(training), int(seed),
                                                 int(offset), int(row0), mean_scale)
    return (z if training else None), logvar, kl, means


class _GaussKL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu_q, s_q, mu_p, s_p):
        dev = check_device(mu_q, s_q, mu_p, s_p)
        B, L = mu_q.shape
        kl = torch.empty((B,), dtype=mu_q.dtype, device=dev)
        with _Guard(dev):
            call('vae_gauss_kl_fwd', ptr(mu_q), mu_q.stride(0), ptr(s_q), s_q.stride(0), ptr(mu_p),
                 0 if mu_p is None else mu_p.stride(0), ptr(s_p), 0 if s_p is None else s_p.stride(0), ptr(kl), B, L,
                 dtype_code(mu_q), stream_ptr(dev))
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(mu_q, s_q, mu_p, s_p)
        return kl

    @staticmethod
    def backward(ctx, gkl):
        if gkl is None:
            return None, None, None, None
        mu_q, s_q, mu_p, s_p = ctx.saved_tensors
        dev = mu_q.device
        B, L = mu_q.shape
        gkl = gkl.contiguous()
        g_mu_q = torch.empty((B, L), dtype=mu_q.dtype, device=dev)
        g_s_q = torch.empty_like(g_mu_q)
        need_p = mu_p is not None and (ctx.needs_input_grad[2] or ctx.needs_input_grad[3])
        g_mu_p = torch.empty_like(g_mu_q) if need_p else None
        g_s_p = torch.empty_like(g_mu_q) if need_p else None
        with _Guard(dev):
            call('vae_gauss_kl_bwd', ptr(mu_q), mu_q.stride(0), ptr(s_q), s_q.stride(0), ptr(mu_p),
                 0 if mu_p is None else mu_p.stride(0), ptr(s_p), 0 if s_p is None else s_p.stride(0), ptr(gkl),
                 ptr(g_mu_q), ptr(g_s_q), ptr(g_mu_p), ptr(g_s_p), B, L, dtype_code(mu_q), stream_ptr(dev))
        return g_mu_q, g_s_q, g_mu_p, g_s_p


def gauss_kl(mu_q, s_q, mu_p=None, s_p=None):
    """KL(N(mu_q, s_q) || N(mu_p, s_p)) summed over the last dim (scale used as given, no exp).

    Reference: reparameterizers/isotropic_gaussian.py:158-190; prior None = N(0,1).
    """
    shape = mu_q.shape[:-1]
    args = [_rows(t) if t is not None else None for t in (mu_q, s_q, mu_p, s_p)]
    if args[2] is not None:
        # priors built by prior_params() may broadcast; the kernel wants full rows
        args[2] = _rows(args[2].expand_as(mu_q))
        args[3] = _rows(args[3].expand_as(mu_q))
    return _GaussKL.apply(*args).reshape(shape)


# --------------------------------------------------------------------------------------
# NLL + ELBO
# --------------------------------------------------------------------------------------
class _NllElbo(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, recon, kld, add, mi, nll_code, kl_beta, inv_count, with_elbo):
        ctx.set_materialize_grads(False)
        dev = check_device(x, recon, kld, add, mi)
        B, D = x.shape
        nll = torch.empty((B,), dtype=x.dtype, device=dev)
        if with_elbo:
            loss = torch.empty_like(nll)
            elbo = torch.empty_like(nll)
            means = torch.zeros((6,), dtype=torch.float32, device=dev)
            ws = workspace(dev)
        else:
            loss = elbo = means = ws = None
        with _Guard(dev):
            call('vae_nll_elbo_fwd', ptr(x), ptr(recon), nll_code, ptr(kld), ptr(add), ptr(mi), kl_beta, inv_count,
                 ptr(nll), ptr(loss), ptr(elbo), ptr(means), ptr(ws), B, D, dtype_code(x), stream_ptr(dev))
        ctx.save_for_backward(x, recon)
        ctx.cfg = (nll_code, kl_beta, inv_count, kld is not None, add is not None, mi is not None)
        if not with_elbo:
            loss = x.new_empty((0,))
            elbo = x.new_empty((0,))
            means = x.new_empty((0,), dtype=torch.float32)
            ctx.mark_non_differentiable(loss, elbo, means)
        return nll, loss, elbo, means

    @staticmethod
    def backward(ctx, gnll, gloss, gelbo, gmeans):
        x, recon = ctx.saved_tensors
        nll_code, kl_beta, inv_count, has_kld, has_add, has_mi = ctx.cfg
        dev = x.device
        B, D = x.shape
        gnll, gloss, gelbo = _dense(gnll), _dense(gloss), _dense(gelbo)
        if gloss is not None and gloss.numel() == 0:
            gloss = gelbo = gmeans = None
        gmeans = None if gmeans is None else gmeans.contiguous().float()
        grecon = torch.empty_like(recon)
        gkld = torch.empty((B,), dtype=x.dtype, device=dev) if has_kld and ctx.needs_input_grad[2] else None
        gadd = torch.empty((B,), dtype=x.dtype, device=dev) if has_add and ctx.needs_input_grad[3] else None
        gmi = torch.empty((B,), dtype=x.dtype, device=dev) if has_mi and ctx.needs_input_grad[4] else None
        with _Guard(dev):
            call('vae_nll_elbo_bwd', ptr(x), ptr(recon), nll_code, ptr(gnll), ptr(gloss), ptr(gelbo), ptr(gmeans),
                 kl_beta, inv_count, ptr(grecon), ptr(gkld), ptr(gadd), ptr(gmi), B, D, dtype_code(x),
                 stream_ptr(dev))
        return None, grecon, gkld, gadd, gmi, None, None, None, None


def _flat_pixels(x, recon, nll_type):
    code = _lib.NLL_TYPES[nll_type.strip().lower()]
    B = x.shape[0]
    xf = x.reshape(B, -1).contiguous()
    rf = recon.reshape(B, -1).contiguous()
    want = xf.shape[1] * (1 if code == 0 else 2)
    if rf.shape[1] != want:
        raise ValueError("nll '%s': recon has %d values per sample, expected %d" % (nll_type, rf.shape[1], want))
    if rf.dtype != xf.dtype:
        xf = xf.to(rf.dtype)
    return code, xf, rf


def nll(x, recon_x, nll_type):
    """Per-sample negative log-likelihood [B] (abstract_vae.py:461-471; definitions: oracle/cpu_ref.py)."""
    code, xf, rf = _flat_pixels(x, recon_x, nll_type)
    return _NllElbo.apply(xf, rf, None, None, None, code, 0.0, 0.0, False)[0]


def nll_elbo(x, recon_x, nll_type, kld=None, add=None, mi=None, kl_beta=1.0, inv_count=None):
    """Fused nll + ELBO combine: returns (nll, loss, elbo) [B] and means float32[6]
    ordered (loss, elbo, nll, kld, additional, mut_info).  abstract_vae.py:344-388 with K == 1."""
    code, xf, rf = _flat_pixels(x, recon_x, nll_type)
    if inv_count is None:
        inv_count = 1.0 / max(xf.shape[0], 1)
    return _NllElbo.apply(xf, rf, _dense(kld), _dense(add), _dense(mi), code, float(kl_beta), float(inv_count), True)


class _ElboCombine(torch.autograd.Function):
    @staticmethod
    def forward(ctx, nll_, kld, add, mi, kl_beta, inv_count):
        ctx.set_materialize_grads(False)
        dev = check_device(nll_, kld, add, mi)
        B = nll_.shape[0]
        loss = torch.empty_like(nll_)
        elbo = torch.empty_like(nll_)
        means = torch.zeros((6,), dtype=torch.float32, device=dev)
        with _Guard(dev):
            call('vae_elbo_combine_fwd', ptr(nll_), ptr(kld), ptr(add), ptr(mi), kl_beta, inv_count, ptr(loss),
                 ptr(elbo), ptr(means), ptr(workspace(dev)), B, dtype_code(nll_), stream_ptr(dev))
        ctx.cfg = (kl_beta, inv_count, B, nll_.dtype, dev, kld is not None, add is not None, mi is not None)
        return loss, elbo, means

    @staticmethod
    def backward(ctx, gloss, gelbo, gmeans):
        kl_beta, inv_count, B, dtype, dev, has_kld, has_add, has_mi = ctx.cfg
        gloss, gelbo = _dense(gloss), _dense(gelbo)
        gmeans = None if gmeans is None else gmeans.contiguous().float()
        gnll = torch.empty((B,), dtype=dtype, device=dev) if ctx.needs_input_grad[0] else None
        gkld = torch.empty((B,), dtype=dtype, device=dev) if has_kld and ctx.needs_input_grad[1] else None
        gadd = torch.empty((B,), dtype=dtype, device=dev) if has_add and ctx.needs_input_grad[2] else None
        gmi = torch.empty((B,), dtype=dtype, device=dev) if has_mi and ctx.needs_input_grad[3] else None
        with _Guard(dev):
            call('vae_elbo_combine_bwd', ptr(gloss), ptr(gelbo), ptr(gmeans), kl_beta, inv_count, ptr(gnll),
                 ptr(gkld), ptr(gadd), ptr(gmi), B, _lib.VAE_F32 if dtype == torch.float32 else _lib.VAE_BF16,
                 stream_ptr(dev))
        return gnll, gkld, gadd, gmi, None, None


def elbo_combine(nll_, kld=None, add=None, mi=None, kl_beta=1.0, inv_count=None):
    """(loss, elbo) [B] + means float32[6] from a given nll [B] (abstract_vae.py:356-388)."""
    nll_ = nll_.contiguous()
    if inv_count is None:
        inv_count = 1.0 / max(nll_.shape[0], 1)
    return _ElboCombine.apply(nll_, _dense(kld), _dense(add), _dense(mi), float(kl_beta), float(inv_count))
