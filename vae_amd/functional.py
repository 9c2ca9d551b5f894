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


def _scalars(n, dev, B):
    """fp32[n] for a kernel's batch scalars.  Every launch with B > 0 writes all n of them (the closing block of the
    grid-wide sum), so no fill kernel is spent on them; a B == 0 call is a no-op and gets zeros."""
    if B > 0:
        return torch.empty((n,), dtype=torch.float32, device=dev)
    return torch.zeros((n,), dtype=torch.float32, device=dev)


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
    def forward(ctx, logits, noise, eps_tol, training, seed, offset, row0, mean_scale, want_logvar=True):
        ctx.set_materialize_grads(False)
        dev = check_device(logits, noise)
        B, L = logits.shape[0], logits.shape[1] // 2
        z = torch.empty((B, L), dtype=logits.dtype, device=dev) if training else None
        logvar = torch.empty((B, L), dtype=logits.dtype, device=dev) if want_logvar else None
        kl = torch.empty((B,), dtype=logits.dtype, device=dev)
        means = _scalars(2, dev, B)
        if mean_scale is None:
            mean_scale = 1.0 / max(B * L, 1)
        with _Guard(dev):
            call('vae_gauss_reparam_kl_fwd', ptr(logits), logits.stride(0), ptr(noise), ptr(z), 0, ptr(logvar), ptr(kl),
                 ptr(means), ptr(workspace(dev)), B, L, eps_tol, mean_scale, int(training), seed & _MASK64,
                 offset & _MASK64, row0, dtype_code(logits), stream_ptr(dev))
        ctx.save_for_backward(logits, noise)
        ctx.cfg = (eps_tol, training, seed, offset, row0, mean_scale)
        nd = []
        if z is None:
            z = logits.new_empty((0,))
            nd.append(z)
        if logvar is None:
            logvar = logits.new_empty((0,))
            nd.append(logvar)
        if nd:
            ctx.mark_non_differentiable(*nd)
        return z, logvar, kl, means

    @staticmethod
    def backward(ctx, gz, glogvar, gkl, gmeans):
        logits, noise = ctx.saved_tensors
        eps_tol, training, seed, offset, row0, mean_scale = ctx.cfg
        if glogvar is not None and glogvar.numel() == 0:
            glogvar = None
        dev = logits.device
        B, L = logits.shape[0], logits.shape[1] // 2
        if not training:
            gz = None
        gz, glogvar, gkl = _dense(gz), _dense(glogvar), _dense(gkl)
        gmeans = None if gmeans is None else gmeans.contiguous().float()
        glogits = torch.empty((B, 2 * L), dtype=logits.dtype, device=dev)
        with _Guard(dev):
            call('vae_gauss_reparam_kl_bwd', ptr(logits), logits.stride(0), ptr(noise), ptr(gz), 0, ptr(glogvar),
                 ptr(gkl), ptr(gmeans), ptr(glogits), 0, B, L, eps_tol, mean_scale, int(training), seed & _MASK64,
                 offset & _MASK64, row0, dtype_code(logits), stream_ptr(dev))
        return glogits, None, None, None, None, None, None, None, None


def gauss_reparam_kl(logits, noise=None, eps_tol=1e-6, training=True, seed=0, offset=0, row0=0, mean_scale=None,
                     want_logvar=True):
    """logits [B, 2L] -> (z [B, L] or None in eval, logvar [B, L] (None unless `want_logvar`), kl [B],
    means float32[2]).

    Reference: reparameterizers/isotropic_gaussian.py:77-131, :158-170, :204-215.
    `noise` None in training draws Philox N(0,1) in the kernel keyed by (seed, offset, row0+row).
    """
    logits = _rows(logits)
    if noise is not None:
        noise = noise.reshape(logits.shape[0], logits.shape[1] // 2).contiguous()
    z, logvar, kl, means = _GaussReparamKL.apply(logits, noise, float(eps_tol), bool(training), int(seed),
                                                 int(offset), int(row0), mean_scale, bool(want_logvar))
    return (z if training else None), (logvar if want_logvar else None), kl, means


class _GaussReparamKLNchw(torch.autograd.Function):
    """4-D logits [B, 2C, H, W] split on the channel dimension, read where they are (no [mu(W) | s(W)] repack)."""

    @staticmethod
    def forward(ctx, logits, noise, eps_tol, training, seed, offset, row0, mean_scale):
        ctx.set_materialize_grads(False)
        dev = check_device(logits, noise)
        B, C2, H, W = logits.shape
        C = C2 // 2
        z = torch.empty((B, C, H, W), dtype=logits.dtype, device=dev) if training else None
        kl = torch.empty((B, C, H), dtype=logits.dtype, device=dev)
        means = _scalars(2, dev, B)
        if mean_scale is None:
            mean_scale = 1.0 / max(B * C * H * W, 1)
        with _Guard(dev):
            call('vae_gauss_reparam_kl_fwd_nchw', ptr(logits), ptr(noise), ptr(z), None, ptr(kl), ptr(means),
                 ptr(workspace(dev)), B, C, H, W, eps_tol, mean_scale, int(training), seed & _MASK64, offset & _MASK64,
                 row0, dtype_code(logits), stream_ptr(dev))
        ctx.save_for_backward(logits, noise)
        ctx.cfg = (eps_tol, training, seed, offset, row0, mean_scale)
        if z is None:
            z = logits.new_empty((0,))
            ctx.mark_non_differentiable(z)
        return z, kl, means

    @staticmethod
    def backward(ctx, gz, gkl, gmeans):
        logits, noise = ctx.saved_tensors
        eps_tol, training, seed, offset, row0, mean_scale = ctx.cfg
        dev = logits.device
        B, C2, H, W = logits.shape
        if not training:
            gz = None
        gz, gkl = _dense(gz), _dense(gkl)
        gmeans = None if gmeans is None else gmeans.contiguous().float()
        glogits = torch.empty_like(logits)
        with _Guard(dev):
            call('vae_gauss_reparam_kl_bwd_nchw', ptr(logits), ptr(noise), ptr(gz), None, ptr(gkl), ptr(gmeans),
                 ptr(glogits), B, C2 // 2, H, W, eps_tol, mean_scale, int(training), seed & _MASK64, offset & _MASK64,
                 row0, dtype_code(logits), stream_ptr(dev))
        return glogits, None, None, None, None, None, None, None


def gauss_reparam_kl_nchw(logits, noise=None, eps_tol=1e-6, training=True, seed=0, offset=0, row0=0, mean_scale=None):
    """logits [B, 2C, H, W] -> (z [B, C, H, W] or None in eval, kl [B, C, H], means float32[2]): the reference's
    channel split for 4-D logits (isotropic_gaussian.py:108-114) with the KL over the last dimension only (:170)."""
    logits = logits.contiguous()
    B, C2, H, W = logits.shape
    if noise is not None:
        noise = noise.reshape(B, C2 // 2, H, W).to(logits.dtype).contiguous()
    z, kl, means = _GaussReparamKLNchw.apply(logits, noise, float(eps_tol), bool(training), int(seed), int(offset),
                                             int(row0), mean_scale)
    return (z if training else None), kl, means


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
            means = _scalars(6, dev, B)
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


class _NllElboFused(torch.autograd.Function):
    """nll + ELBO combine with d(loss_mean)/d(recon) produced by the SAME pass over x and recon.  backward():
    only mean gradients arriving (loss_mean.backward(), the training loop's case) -> the stored gradient is
    handed out after a device-side rescale that is a no-op for a unit upstream gradient; any per-sample
    gradient (nll / loss / elbo [B]) -> the general backward kernel recomputes from x and recon."""

    @staticmethod
    def forward(ctx, x, recon, kld, add, mi, nll_code, kl_beta, inv_count):
        ctx.set_materialize_grads(False)
        dev = check_device(x, recon, kld, add, mi)
        B, D = x.shape
        nll = torch.empty((B,), dtype=x.dtype, device=dev)
        loss, elbo = torch.empty_like(nll), torch.empty_like(nll)
        means = _scalars(6, dev, B)
        grecon = torch.empty_like(recon)
        with _Guard(dev):
            call('vae_nll_elbo_fused', ptr(x), ptr(recon), nll_code, ptr(kld), ptr(add), ptr(mi), kl_beta, inv_count,
                 1.0, ptr(nll), ptr(loss), ptr(elbo), ptr(means), ptr(workspace(dev)), ptr(grecon), B, D,
                 dtype_code(x), stream_ptr(dev))
        ctx.save_for_backward(x, recon)
        ctx.grecon = grecon
        ctx.cfg = (nll_code, kl_beta, inv_count, kld is not None, add is not None, mi is not None)
        return nll, loss, elbo, means

    @staticmethod
    def backward(ctx, gnll, gloss, gelbo, gmeans):
        x, recon = ctx.saved_tensors
        nll_code, kl_beta, inv_count, has_kld, has_add, has_mi = ctx.cfg
        dev = x.device
        B, D = x.shape
        gnll, gloss, gelbo = _dense(gnll), _dense(gloss), _dense(gelbo)
        gmeans = None if gmeans is None else gmeans.contiguous().float()
        new = lambda: torch.empty((B,), dtype=x.dtype, device=dev)
        gkld = new() if has_kld and ctx.needs_input_grad[2] else None
        gadd = new() if has_add and ctx.needs_input_grad[3] else None
        gmi = new() if has_mi and ctx.needs_input_grad[4] else None
        grecon = ctx.grecon
        ctx.grecon = None
        with _Guard(dev):
            if gnll is None and gloss is None and gelbo is None and gmeans is not None and grecon is not None:
                call('vae_nll_fused_backward', ptr(grecon), grecon.numel(), ptr(gmeans), kl_beta, inv_count,
                     ptr(gkld), ptr(gadd), ptr(gmi), B, dtype_code(x), stream_ptr(dev))
            else:
                grecon = torch.empty_like(recon)
                call('vae_nll_elbo_bwd', ptr(x), ptr(recon), nll_code, ptr(gnll), ptr(gloss), ptr(gelbo), ptr(gmeans),
                     kl_beta, inv_count, ptr(grecon), ptr(gkld), ptr(gadd), ptr(gmi), B, D, dtype_code(x),
                     stream_ptr(dev))
        return None, grecon, gkld, gadd, gmi, None, None, None


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


def nll_elbo(x, recon_x, nll_type, kld=None, add=None, mi=None, kl_beta=1.0, inv_count=None, fuse_backward=False):
    """Fused nll + ELBO combine: returns (nll, loss, elbo) [B] and means float32[6]
    ordered (loss, elbo, nll, kld, additional, mut_info).  abstract_vae.py:344-388 with K == 1.
    `fuse_backward`: also form d(loss_mean)/d(recon) in the same pass (one read of x and recon per training
    step instead of two); results are identical, the gradient buffer is simply allocated a step earlier."""
    code, xf, rf = _flat_pixels(x, recon_x, nll_type)
    if inv_count is None:
        inv_count = 1.0 / max(xf.shape[0], 1)
    # fp16 keeps the two-pass form: the one-pass kernel stores d(loss_mean)/d(recon) already scaled by 1/B in the
    # tensor dtype, and with the loss scaling that fp16 training runs under (upstream gradient != 1) that value -- an
    # fp16 subnormal from B ~ 4096 up -- would be rounded BEFORE it is multiplied back up; the general backward forms
    # c/B in fp32 and rounds once.  (bf16 / fp32: same exponent range as fp32, the rescale costs at most one more
    # rounding at the last place.)
    if fuse_backward and rf.dtype != torch.float16 and rf.requires_grad and torch.is_grad_enabled() and xf.shape[0] > 0:
        out = _NllElboFused.apply(xf, rf, _dense(kld), _dense(add), _dense(mi), code, float(kl_beta), float(inv_count))
        return out
    return _NllElbo.apply(xf, rf, _dense(kld), _dense(add), _dense(mi), code, float(kl_beta), float(inv_count), True)


class _ElboCombine(torch.autograd.Function):
    @staticmethod
    def forward(ctx, nll_, kld, add, mi, kl_beta, inv_count):
        ctx.set_materialize_grads(False)
        dev = check_device(nll_, kld, add, mi)
        B = nll_.shape[0]
        loss = torch.empty_like(nll_)
        elbo = torch.empty_like(nll_)
        means = _scalars(6, dev, B)
        with _Guard(dev):
            call('vae_elbo_combine_fwd', ptr(nll_), ptr(kld), ptr(add), ptr(mi), kl_beta, inv_count, ptr(loss),
                 ptr(elbo), ptr(means), ptr(workspace(dev)), B, dtype_code(nll_), stream_ptr(dev))
        ctx.cfg = (kl_beta, inv_count, B, nll_.dtype, dtype_code(nll_), dev, kld is not None, add is not None,
                   mi is not None)
        return loss, elbo, means

    @staticmethod
    def backward(ctx, gloss, gelbo, gmeans):
        kl_beta, inv_count, B, dtype, code, dev, has_kld, has_add, has_mi = ctx.cfg
        gloss, gelbo = _dense(gloss), _dense(gelbo)
        gmeans = None if gmeans is None else gmeans.contiguous().float()
        gnll = torch.empty((B,), dtype=dtype, device=dev) if ctx.needs_input_grad[0] else None
        gkld = torch.empty((B,), dtype=dtype, device=dev) if has_kld and ctx.needs_input_grad[1] else None
        gadd = torch.empty((B,), dtype=dtype, device=dev) if has_add and ctx.needs_input_grad[2] else None
        gmi = torch.empty((B,), dtype=dtype, device=dev) if has_mi and ctx.needs_input_grad[3] else None
        with _Guard(dev):
            call('vae_elbo_combine_bwd', ptr(gloss), ptr(gelbo), ptr(gmeans), kl_beta, inv_count, ptr(gnll),
                 ptr(gkld), ptr(gadd), ptr(gmi), B, code, stream_ptr(dev))
        return gnll, gkld, gadd, gmi, None, None


def elbo_combine(nll_, kld=None, add=None, mi=None, kl_beta=1.0, inv_count=None):
    """(loss, elbo) [B] + means float32[6] from a given nll [B] (abstract_vae.py:356-388)."""
    nll_ = nll_.contiguous()
    if inv_count is None:
        inv_count = 1.0 / max(nll_.shape[0], 1)
    return _ElboCombine.apply(nll_, _dense(kld), _dense(add), _dense(mi), float(kl_beta), float(inv_count))


# --------------------------------------------------------------------------------------
# GumbelSoftmax
# --------------------------------------------------------------------------------------
class _Gumbel(torch.autograd.Function):
    """want = (y, z_hard, log_q_z) flags: outputs that are not wanted are not written (empty placeholders)."""

    @staticmethod
    def forward(ctx, logits, unif, tau, seed, offset, row0, want):
        ctx.set_materialize_grads(False)
        dev = check_device(logits, unif)
        B, K = logits.shape
        new = lambda w: torch.empty((B, K), dtype=logits.dtype, device=dev) if w else None
        y, zh, lq = new(want[0]), new(want[1]), new(want[2])
        with _Guard(dev):
            call('vae_gumbel_fwd', ptr(logits), logits.stride(0), ptr(unif), ptr(y), 0, ptr(zh), 0, ptr(lq), B, K, tau,
                 seed & _MASK64, offset & _MASK64, row0, dtype_code(logits), stream_ptr(dev))
        ctx.save_for_backward(y, lq)
        ctx.tau = tau
        outs, nd = [], []
        for t in (y, zh, lq):
            if t is None:
                t = logits.new_empty((0,))
                nd.append(t)
            outs.append(t)
        if nd:
            ctx.mark_non_differentiable(*nd)
        return tuple(outs)

    @staticmethod
    def backward(ctx, gy, gzh, glq):
        y, lq = ctx.saved_tensors
        if y is None:
            gy = gzh = None
        if lq is None:
            glq = None
        if gy is None and gzh is None and glq is None:
            return None, None, None, None, None, None, None
        ref = y if y is not None else lq
        dev = ref.device
        B, K = ref.shape
        gy, gzh, glq = _dense(gy), _dense(gzh), _dense(glq)
        gl = torch.empty_like(ref)
        with _Guard(dev):
            call('vae_gumbel_bwd', ptr(y), 0, ptr(lq), ptr(gy), 0, ptr(gzh), 0, ptr(glq), ptr(gl), 0, B, K, ctx.tau,
                 dtype_code(ref), stream_ptr(dev))
        return gl, None, None, None, None, None, None


def gumbel_softmax(logits, unif=None, tau=1.0, seed=0, offset=0, row0=0, want=(True, True, True)):
    """logits [..., K] -> (y soft sample, z_hard straight-through, log_q_z), all [..., K]; an entry of `want`
    set False leaves that output unwritten (returned as None).

    Reference: reparameterizers/gumbel.py:110-124, :257-304.  `unif` is the torch.rand draw of :269
    (None: Philox in the kernel)."""
    shape = logits.shape
    rows = _rows(logits)
    if unif is not None:
        unif = unif.reshape(rows.shape).to(rows.dtype).contiguous()
    want = tuple(bool(w) for w in want)
    outs = _Gumbel.apply(rows, unif, float(tau), int(seed), int(offset), int(row0), want)
    return tuple(o.reshape(shape) if w else None for o, w in zip(outs, want))


def log_softmax_rows(logits):
    """log_q_z = log_softmax(logits, -1) (gumbel.py:119) alone: the Gumbel kernel with only that output."""
    return gumbel_softmax(logits, None, 1.0, want=(False, False, True))[2]


class _CategoricalKL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lq, lp):
        ctx.set_materialize_grads(False)
        dev = check_device(lq, lp)
        B, K = lq.shape
        kl = torch.empty((B,), dtype=lq.dtype, device=dev)
        with _Guard(dev):
            call('vae_categorical_kl_fwd', ptr(lq), lq.stride(0), ptr(lp), lp.stride(0), ptr(kl), B, K,
                 dtype_code(lq), stream_ptr(dev))
        ctx.save_for_backward(lq, lp)
        return kl

    @staticmethod
    def backward(ctx, gkl):
        if gkl is None:
            return None, None
        lq, lp = ctx.saved_tensors
        dev = lq.device
        B, K = lq.shape
        gq = torch.empty((B, K), dtype=lq.dtype, device=dev)
        gp = torch.empty_like(gq) if ctx.needs_input_grad[1] else None
        with _Guard(dev):
            call('vae_categorical_kl_bwd', ptr(lq), lq.stride(0), ptr(lp), lp.stride(0), ptr(gkl.contiguous()),
                 ptr(gq), ptr(gp), B, K, dtype_code(lq), stream_ptr(dev))
        return gq, gp


def categorical_kl(log_q, log_p):
    """sum softmax(q) * (log_softmax(q) - log_softmax(p)) over the last dim (gumbel.py:222-254)."""
    shape = log_q.shape[:-1]
    return _CategoricalKL.apply(_rows(log_q), _rows(log_p.expand_as(log_q))).reshape(shape)


# --------------------------------------------------------------------------------------
# Bernoulli (relaxed)
# --------------------------------------------------------------------------------------
class _Bernoulli(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, unif, tau, seed, offset, row0, want_hard=True):
        ctx.set_materialize_grads(False)
        dev = check_device(logits, unif)
        B, K = logits.shape
        y = torch.empty((B, K), dtype=logits.dtype, device=dev)
        zh = torch.empty_like(y) if want_hard else None
        with _Guard(dev):
            call('vae_bernoulli_fwd', ptr(logits), logits.stride(0), ptr(unif), ptr(y), 0, ptr(zh), 0, B, K, tau,
                 seed & _MASK64, offset & _MASK64, row0, dtype_code(logits), stream_ptr(dev))
        ctx.save_for_backward(logits, y)
        ctx.tau = tau
        if zh is None:
            zh = logits.new_empty((0,))
            ctx.mark_non_differentiable(zh)
        return y, zh

    @staticmethod
    def backward(ctx, gy, gzh):
        if gzh is not None and gzh.numel() == 0:
            gzh = None
        if gy is None and gzh is None:
            return None, None, None, None, None, None, None
        logits, y = ctx.saved_tensors
        dev = y.device
        B, K = y.shape
        gl = torch.empty_like(y)
        with _Guard(dev):
            call('vae_bernoulli_bwd', ptr(logits), logits.stride(0), ptr(y), 0, ptr(_dense(gy)), 0, ptr(_dense(gzh)), 0,
                 ptr(gl), 0, B, K, ctx.tau, dtype_code(y), stream_ptr(dev))
        return gl, None, None, None, None, None, None


def relaxed_bernoulli(logits, unif=None, tau=1.0, seed=0, offset=0, row0=0, want_hard=True):
    """logits [..., K] -> (y relaxed sample, z_hard straight-through or None unless `want_hard`).
    bernoulli.py:111-136."""
    shape = logits.shape
    rows = _rows(logits)
    if unif is not None:
        unif = unif.reshape(rows.shape).to(rows.dtype).contiguous()
    y, zh = _Bernoulli.apply(rows, unif, float(tau), int(seed), int(offset), int(row0), bool(want_hard))
    return y.reshape(shape), (zh.reshape(shape) if want_hard else None)


class _BernoulliKL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lq, lp):
        ctx.set_materialize_grads(False)
        dev = check_device(lq, lp)
        B, K = lq.shape
        kl = torch.empty((B,), dtype=lq.dtype, device=dev)
        with _Guard(dev):
            call('vae_bernoulli_kl_fwd', ptr(lq), lq.stride(0), ptr(lp), 0 if lp is None else lp.stride(0), ptr(kl),
                 B, K, dtype_code(lq), stream_ptr(dev))
        ctx.save_for_backward(lq, lp)
        return kl

    @staticmethod
    def backward(ctx, gkl):
        if gkl is None:
            return None, None
        lq, lp = ctx.saved_tensors
        dev = lq.device
        B, K = lq.shape
        gq = torch.empty((B, K), dtype=lq.dtype, device=dev)
        gp = torch.empty_like(gq) if (lp is not None and ctx.needs_input_grad[1]) else None
        with _Guard(dev):
            call('vae_bernoulli_kl_bwd', ptr(lq), lq.stride(0), ptr(lp), 0 if lp is None else lp.stride(0),
                 ptr(gkl.contiguous()), ptr(gq), ptr(gp), B, K, dtype_code(lq), stream_ptr(dev))
        return gq, gp


def bernoulli_kl(logits_q, logits_p=None):
    """bernoulli.py:147-191 summed over the last dim; logits_p None = uniform prior (p = 0.5)."""
    shape = logits_q.shape[:-1]
    lp = None if logits_p is None else _rows(logits_p.expand_as(logits_q))
    return _BernoulliKL.apply(_rows(logits_q), lp).reshape(shape)


# --------------------------------------------------------------------------------------
# Beta
# --------------------------------------------------------------------------------------
class _BetaSplitKL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, mean_scale):
        ctx.set_materialize_grads(False)
        dev = check_device(logits)
        B, L = logits.shape[0], logits.shape[1] // 2
        c1 = torch.empty((B, L), dtype=logits.dtype, device=dev)
        c0 = torch.empty_like(c1)
        kl = torch.empty((B,), dtype=logits.dtype, device=dev)
        means = _scalars(2, dev, B)
        if mean_scale is None:
            mean_scale = 1.0 / max(B * L, 1)
        with _Guard(dev):
            call('vae_beta_split_kl_fwd', ptr(logits), logits.stride(0), ptr(c1), ptr(c0), ptr(kl), ptr(means),
                 ptr(workspace(dev)), B, L, mean_scale, dtype_code(logits), stream_ptr(dev))
        ctx.save_for_backward(logits)
        ctx.mean_scale = mean_scale
        return c1, c0, kl, means

    @staticmethod
    def backward(ctx, g1, g0, gkl, gmeans):
        (logits,) = ctx.saved_tensors
        dev = logits.device
        B, L = logits.shape[0], logits.shape[1] // 2
        gmeans = None if gmeans is None else gmeans.contiguous().float()
        gl = torch.empty((B, 2 * L), dtype=logits.dtype, device=dev)
        with _Guard(dev):
            call('vae_beta_split_kl_bwd', ptr(logits), logits.stride(0), ptr(_dense(g1)), ptr(_dense(g0)),
                 ptr(_dense(gkl)), ptr(gmeans), ptr(gl), B, L, ctx.mean_scale, dtype_code(logits), stream_ptr(dev))
        return gl, None


def beta_split_kl(logits, mean_scale=None):
    """logits [B, 2L] -> (conc1, conc2 [B, L], kl vs Beta(1.1,1.1) [B], means float32[2]).  beta.py:105-124."""
    return _BetaSplitKL.apply(_rows(logits), mean_scale)


class _BetaKL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a1, a0, b1, b0):
        ctx.set_materialize_grads(False)
        dev = check_device(a1, a0, b1, b0)
        B, L = a1.shape
        kl = torch.empty((B,), dtype=a1.dtype, device=dev)
        sb1 = 0 if b1 is None else b1.stride(0)
        sb0 = 0 if b0 is None else b0.stride(0)
        with _Guard(dev):
            call('vae_beta_kl_fwd', ptr(a1), a1.stride(0), ptr(a0), a0.stride(0), ptr(b1), sb1, ptr(b0), sb0,
                 ptr(kl), B, L, dtype_code(a1), stream_ptr(dev))
        ctx.save_for_backward(a1, a0, b1, b0)
        return kl

    @staticmethod
    def backward(ctx, gkl):
        if gkl is None:
            return None, None, None, None
        a1, a0, b1, b0 = ctx.saved_tensors
        dev = a1.device
        B, L = a1.shape
        ga1 = torch.empty((B, L), dtype=a1.dtype, device=dev)
        ga0 = torch.empty_like(ga1)
        need_p = b1 is not None and (ctx.needs_input_grad[2] or ctx.needs_input_grad[3])
        gb1 = torch.empty_like(ga1) if need_p else None
        gb0 = torch.empty_like(ga1) if need_p else None
        sb1 = 0 if b1 is None else b1.stride(0)
        sb0 = 0 if b0 is None else b0.stride(0)
        with _Guard(dev):
            call('vae_beta_kl_bwd', ptr(a1), a1.stride(0), ptr(a0), a0.stride(0), ptr(b1), sb1, ptr(b0), sb0,
                 ptr(gkl.contiguous()), ptr(ga1), ptr(ga0), ptr(gb1), ptr(gb0), B, L, dtype_code(a1), stream_ptr(dev))
        return ga1, ga0, gb1, gb0


def beta_kl(a1, a0, b1=None, b0=None):
    """KL(Beta(a1, a0) || Beta(b1, b0) or Beta(1.1, 1.1)) over the last dim.  beta.py:110-136."""
    shape = a1.shape[:-1]
    args = [_rows(a1), _rows(a0), None, None]
    if b1 is not None:
        args[2], args[3] = _rows(b1.expand_as(a1)), _rows(b0.expand_as(a0))
    return _BetaKL.apply(*args).reshape(shape)


# --------------------------------------------------------------------------------------
# Flow "KL"
# --------------------------------------------------------------------------------------
class _FlowKL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, logdet, inv_count, reduce_fn):
        ctx.set_materialize_grads(False)
        dev = check_device(u, logdet)
        cross = u.dim() == 2
        if cross:
            B, F = u.shape
            R = 1
        else:
            B, R, F = u.shape
        kl = torch.empty((B,), dtype=u.dtype, device=dev)
        mean = torch.zeros((1,), dtype=torch.float32, device=dev)
        if inv_count is None:
            inv_count = 1.0 / max(B, 1)
        args = (ptr(u), ptr(logdet), R, ptr(kl), ptr(mean), ptr(workspace(dev)), B, F, inv_count, int(cross))
        with _Guard(dev):
            if cross and reduce_fn is not None:
                call('vae_flow_kl_fwd', *args, 1, dtype_code(u), stream_ptr(dev))
                reduce_fn(mean)                      # sharded batch: all-reduce(sum) of the partial mean
                call('vae_flow_kl_fwd', *args, 2, dtype_code(u), stream_ptr(dev))
            else:
                call('vae_flow_kl_fwd', *args, 0, dtype_code(u), stream_ptr(dev))
        ctx.save_for_backward(u)
        ctx.cfg = (cross, B, R, F, inv_count, reduce_fn, logdet.shape)
        return kl

    @staticmethod
    def backward(ctx, gkl):
        if gkl is None:
            return None, None, None, None
        (u,) = ctx.saved_tensors
        cross, B, R, F, inv_count, reduce_fn, ld_shape = ctx.cfg
        dev = u.device
        gkl = gkl.contiguous()
        gu = torch.empty_like(u) if ctx.needs_input_grad[0] else None
        if gu is not None:
            with _Guard(dev):
                if cross:
                    gsum = gkl.float().sum().reshape(1)
                    if reduce_fn is not None:
                        reduce_fn(gsum)
                    call('vae_flow_kl_bwd', ptr(u), None, ptr(gsum), ptr(gu), B, F, inv_count, dtype_code(u),
                         stream_ptr(dev))
                else:
                    call('vae_flow_kl_bwd', ptr(u), ptr(gkl), None, ptr(gu), B, R * F, 1.0 / R, dtype_code(u),
                         stream_ptr(dev))
        gld = None
        if ctx.needs_input_grad[1]:
            gld = (-gkl).reshape(B, 1).expand(B, R).reshape(ld_shape) if cross else \
                (-gkl / R).reshape(B, 1).expand(B, R).reshape(ld_shape)
        return gu, gld, None, None


def flow_kl(u, logdet, inv_count=None, reduce_fn=None):
    """flow.py:67-78.  u [B,F] + logdet [B,1]: kl_i = mean_j(nlp_j) - logdet_i (the reference's [B,B]
    broadcast, SURVEY Q6, in O(B*F)).  u [B,R,F] + logdet [B,R]: kl_i = mean_r(nlp_ir - logdet_ir)."""
    if u.dim() == 2 and logdet.dim() == 2 and logdet.shape[1] != 1:
        raise ValueError("flow_kl: 2-D recon_x needs logdet [B,1]")
    return _FlowKL.apply(u.contiguous(), logdet.contiguous(), inv_count, reduce_fn)


# --------------------------------------------------------------------------------------
# VRNN per-step pair: posterior + clamped prior reparameterization and both KLs
# --------------------------------------------------------------------------------------
class _GaussPair(torch.autograd.Function):
    @staticmethod
    def forward(ctx, enc, pri, noise_q, noise_p, eps_tol, training, with_prior_kl, seed, offset, row0, mean_scale,
                want_logvars=True, want_means=True):
        ctx.set_materialize_grads(False)
        dev = check_device(enc, pri, noise_q, noise_p)
        B, L = enc.shape[0], enc.shape[1] // 2
        new = lambda *s: torch.empty(s, dtype=enc.dtype, device=dev)
        z_q = new(B, L) if training else None
        z_p = new(B, L) if training else None
        lv_q, lv_p = (new(B, L), new(B, L)) if want_logvars else (None, None)
        clamped, kl_pair = new(B, 2 * L), new(B)
        kl_prior = new(B) if with_prior_kl else None
        # without the four mean diagnostics the launch is a pure row kernel (no grid-wide sum, no ticket chain): at the
        # per-step batches of a VRNN (a few thousand rows, 5-8 us launches) that chain is a third of the launch
        means = _scalars(4, dev, B) if want_means else None
        if mean_scale is None:
            mean_scale = 1.0 / max(B * L, 1)
        with _Guard(dev):
            call('vae_gauss_pair_fwd', ptr(enc), enc.stride(0), ptr(pri), pri.stride(0), ptr(noise_q), ptr(noise_p),
                 ptr(z_q), ptr(z_p), ptr(lv_q), ptr(lv_p), ptr(clamped), ptr(kl_pair), ptr(kl_prior), ptr(means),
                 ptr(workspace(dev)) if want_means else None, B, L, eps_tol, mean_scale, int(training), seed & _MASK64,
                 offset & _MASK64, row0, dtype_code(enc), stream_ptr(dev))
        ctx.save_for_backward(enc, pri, noise_q, noise_p)
        ctx.cfg = (eps_tol, training, seed, offset, row0, mean_scale)
        nd = []
        if means is None:
            means = torch.empty((0,), dtype=torch.float32, device=dev)
            nd.append(means)
        if not training:
            z_q, z_p = enc.new_empty((0,)), enc.new_empty((0,))
            nd += [z_q, z_p]
        if kl_prior is None:
            kl_prior = enc.new_empty((0,))
            nd.append(kl_prior)
        if not want_logvars:        # produced on first read by the caller (vae_amd.vrnn) instead
            lv_q, lv_p = enc.new_empty((0,)), enc.new_empty((0,))
            nd += [lv_q, lv_p]
        if nd:
            ctx.mark_non_differentiable(*nd)
        return z_q, z_p, lv_q, lv_p, clamped, kl_pair, kl_prior, means

    @staticmethod
    def backward(ctx, gz_q, gz_p, glv_q, glv_p, gcl, gkl_pair, gkl_prior, gmeans):
        enc, pri, noise_q, noise_p = ctx.saved_tensors
        eps_tol, training, seed, offset, row0, mean_scale = ctx.cfg
        dev = enc.device
        B, L = enc.shape[0], enc.shape[1] // 2
        if not training:
            gz_q = gz_p = None
        gmeans = None if (gmeans is None or gmeans.numel() == 0) else gmeans.contiguous().float()
        g_enc = torch.empty((B, 2 * L), dtype=enc.dtype, device=dev)
        g_pri = torch.empty_like(g_enc)
        with _Guard(dev):
            call('vae_gauss_pair_bwd', ptr(enc), enc.stride(0), ptr(pri), pri.stride(0), ptr(noise_q), ptr(noise_p),
                 ptr(_dense(gz_q)), ptr(_dense(gz_p)), ptr(_dense(glv_q)), ptr(_dense(glv_p)), ptr(_dense(gcl)),
                 ptr(_dense(gkl_pair)), ptr(_dense(gkl_prior)), ptr(gmeans), ptr(g_enc), ptr(g_pri), B, L, eps_tol,
                 mean_scale, int(training), seed & _MASK64, offset & _MASK64, row0, dtype_code(enc), stream_ptr(dev))
        return g_enc, g_pri, None, None, None, None, None, None, None, None, None, None, None


def gauss_pair(enc_logits, prior_logits, noise_post=None, noise_prior=None, eps_tol=1e-6, training=True,
               with_prior_kl=True, seed=0, offset=0, row0=0, mean_scale=None, want_logvars=True, want_means=True):
    """One VRNN step (vrnn.py:419-447 + :723-733) in one launch.  Returns (z_post, z_prior, logvar_post,
    logvar_prior, clamped_prior_logits, kl_pair, kl_prior, means[4]); z_* are None in eval (the caller
    uses the mu views), kl_prior is None unless `with_prior_kl`, the logvars are None (not written) unless
    `want_logvars`."""
    enc, pri = _rows(enc_logits), _rows(prior_logits)
    B, L = enc.shape[0], enc.shape[1] // 2
    nq = None if noise_post is None else noise_post.reshape(B, L).contiguous()
    np_ = None if noise_prior is None else noise_prior.reshape(B, L).contiguous()
    out = _GaussPair.apply(enc, pri, nq, np_, float(eps_tol), bool(training), bool(with_prior_kl), int(seed),
                           int(offset), int(row0), mean_scale, bool(want_logvars), bool(want_means))
    z_q, z_p, lv_q, lv_p, clamped, kl_pair, kl_prior, means = out
    if not want_means:
        means = None
    if not want_logvars:
        lv_q = lv_p = None
    return ((z_q if training else None), (z_p if training else None), lv_q, lv_p, clamped, kl_pair,
            (kl_prior if with_prior_kl else None), means)


# --------------------------------------------------------------------------------------
# ConcatReparameterizer over kernel-backed leaves: children read strided slices of the shared
# logits buffer, write their samples straight into the concatenated output and their gradients
# straight into ONE d(logits) buffer -- no torch.cat, no narrow-backward zero-fills and adds.
# --------------------------------------------------------------------------------------
ROW_KERNEL = True   # tests flip it to compare the one-launch row kernel with the per-child launches


def _poff(t, elems):
    import ctypes
    return ctypes.c_void_p(t.data_ptr() + elems * t.element_size())


def _row_kernel_ok(logits, specs, noises=(None, None), training=True):
    """Concat(Gaussian, Gumbel) whose whole row fits the one-launch row kernel (csrc/concat.hip); the two children's
    noise must come from one source (both injected or both drawn in the kernel)."""
    if len(specs) != 2 or specs[0][0] != 'gauss' or specs[1][0] != 'gumbel':
        return False
    if (noises[1] is None) != (noises[0] is None) and not (noises[0] is None and not training):
        return False
    vn = 16 // logits.element_size()
    if logits.data_ptr() % 16 or logits.stride(0) % vn:
        return False
    return bool(_lib.load().vae_concat_gauss_gumbel_supported(specs[0][1], specs[1][1], dtype_code(logits)))


class _ConcatLeaves(torch.autograd.Function):
    """specs: tuple of ('gauss', L, eps_tol, mean_scale) | ('gumbel', K, tau) | ('bern', K, tau);
    noises / seeds: one entry per child (tensor or None; (seed, offset))."""

    @staticmethod
    def forward(ctx, logits, specs, training, noises, seeds, row0):
        ctx.set_materialize_grads(False)
        dev = check_device(logits, *[n for n in noises if n is not None])
        B, W_in = logits.shape
        rs = logits.stride(0)
        W_out = sum(s[1] for s in specs)
        need = sum(2 * s[1] if s[0] == 'gauss' else s[1] for s in specs)
        if W_in != need:      # the kernels index the row by the specs: a narrower row would be read past its end
            raise ValueError("concat: logits have %d columns, the children consume %d" % (W_in, need))
        dt = dtype_code(logits)
        new = lambda *s: torch.empty(s, dtype=logits.dtype, device=dev)
        zcat = new(B, W_out)
        ctx.row_kernel = ROW_KERNEL and B > 0 and _row_kernel_ok(logits, specs, noises, training)
        if ctx.row_kernel:
            # the pair of BASELINE config 3: the whole [2L + K] row in ONE launch
            (_, L, eps_tol, mean_scale), (_, K, tau) = specs
            empty = logits.new_empty((0,))
            kl = new(B)
            means = torch.empty((0,), dtype=torch.float32, device=dev)   # produced on first read by the module class
            other, lq = (empty, empty) if training else (new(B, K), new(B, K))
            with _Guard(dev):
                call('vae_concat_gauss_gumbel_fwd', ptr(logits), rs, ptr(noises[0]), ptr(noises[1]), ptr(zcat), W_out,
                     None if training else ptr(other), None if training else ptr(lq), ptr(kl), B, L, K, eps_tol, tau,
                     int(training), seeds[0][0] & _MASK64, seeds[0][1] & _MASK64, seeds[1][0] & _MASK64,
                     seeds[1][1] & _MASK64, row0, dt, stream_ptr(dev))
            outs = [zcat, empty, kl, means, other, lq]
            ctx.specs, ctx.training, ctx.seeds, ctx.row0 = specs, training, seeds, row0
            ctx.n_noise = len(noises)
            ctx.save_for_backward(logits, *noises, *outs)
            return tuple(outs)
        outs, saved = [zcat], []
        i_off = o_off = 0
        st = stream_ptr(dev)
        with _Guard(dev):
            for spec, noise, (seed, offset) in zip(specs, noises, seeds):
                kind, n = spec[0], spec[1]
                if kind == 'gauss':
                    _, L, eps_tol, mean_scale = spec
                    logvar, kl = logits.new_empty((0,)), new(B)   # logvar is produced lazily by the module class
                    means = _scalars(2, dev, B)
                    ms = 1.0 / max(B * L, 1) if mean_scale is None else mean_scale
                    call('vae_gauss_reparam_kl_fwd', _poff(logits, i_off), rs, ptr(noise), _poff(zcat, o_off), W_out,
                         None, ptr(kl), ptr(means), ptr(workspace(dev)), B, L, eps_tol, ms, int(training),
                         seed & _MASK64, offset & _MASK64, row0, dt, st)
                    outs += [logvar, kl, means]
                    i_off, o_off = i_off + 2 * L, o_off + L
                else:
                    K, tau = spec[1], spec[2]
                    # the output that does NOT go into the concatenated sample: y in eval; in training it would be
                    # z_hard, which -- like log_q_z -- the module class produces on first read instead
                    other = logits.new_empty((0,)) if training else new(B, K)
                    y_ptr, y_s, zh_ptr, zh_s = (_poff(zcat, o_off), W_out, None, K) if training else \
                        (ptr(other), K, _poff(zcat, o_off), W_out)
                    if kind == 'gumbel':
                        lq = logits.new_empty((0,)) if training else new(B, K)
                        call('vae_gumbel_fwd', _poff(logits, i_off), rs, ptr(noise), y_ptr, y_s, zh_ptr, zh_s,
                             None if training else ptr(lq), B, K, tau, seed & _MASK64, offset & _MASK64, row0, dt, st)
                        outs += [other, lq]
                    else:
                        call('vae_bernoulli_fwd', _poff(logits, i_off), rs, ptr(noise), y_ptr, y_s, zh_ptr, zh_s, B, K,
                             tau, seed & _MASK64, offset & _MASK64, row0, dt, st)
                        outs += [other]
                    i_off, o_off = i_off + K, o_off + K
        ctx.specs, ctx.training, ctx.seeds, ctx.row0 = specs, training, seeds, row0
        ctx.n_noise = len(noises)
        ctx.save_for_backward(logits, *noises, *outs)
        return tuple(outs)

    @staticmethod
    def backward(ctx, gzcat, *gex):
        specs, training = ctx.specs, ctx.training
        sv = ctx.saved_tensors
        logits, noises, outs = sv[0], sv[1:1 + ctx.n_noise], sv[1 + ctx.n_noise:]
        zcat = outs[0]
        dev = logits.device
        B, W_in = logits.shape
        rs = logits.stride(0)
        W_out = zcat.shape[1]
        dt = dtype_code(logits)
        gzcat = _dense(gzcat)
        glogits = torch.empty((B, W_in), dtype=logits.dtype, device=dev)
        if ctx.row_kernel and training and all(g is None for g in (gex[0], gex[2], gex[3], gex[4])):
            (_, L, eps_tol, _), (_, K, tau) = specs
            gkl = _dense(gex[1])
            with _Guard(dev):
                call('vae_concat_gauss_gumbel_bwd', ptr(logits), rs, ptr(noises[0]), ptr(zcat), W_out, ptr(gzcat),
                     W_out, ptr(gkl), ptr(glogits), W_in, B, L, K, eps_tol, tau, ctx.seeds[0][0] & _MASK64,
                     ctx.seeds[0][1] & _MASK64, ctx.row0, dt, stream_ptr(dev))
            return glogits, None, None, None, None, None
        i_off = o_off = 0
        k = 0          # index into outs[1:] / gex
        st = stream_ptr(dev)
        gslice = lambda off: (None, 0) if gzcat is None else (_poff(gzcat, off), W_out)
        with _Guard(dev):
            for spec, noise, (seed, offset) in zip(specs, noises, ctx.seeds):
                kind = spec[0]
                if kind == 'gauss':
                    _, L, eps_tol, mean_scale = spec
                    glv, gkl, gm = None, _dense(gex[k + 1]), gex[k + 2]
                    gm = None if gm is None else gm.contiguous().float()
                    ms = 1.0 / max(B * L, 1) if mean_scale is None else mean_scale
                    gz_p, gz_s = gslice(o_off)
                    call('vae_gauss_reparam_kl_bwd', _poff(logits, i_off), rs, ptr(noise), gz_p, gz_s, ptr(glv),
                         ptr(gkl), ptr(gm), _poff(glogits, i_off), W_in, B, L, eps_tol, ms, int(training),
                         seed & _MASK64, offset & _MASK64, ctx.row0, dt, st)
                    k += 3
                    i_off, o_off = i_off + 2 * L, o_off + L
                else:
                    K, tau = spec[1], spec[2]
                    other, g_other = outs[1 + k], _dense(gex[k])
                    gc_p, gc_s = gslice(o_off)
                    if training:   # y lives in zcat; z_hard was not produced here
                        y_p, y_s, gy_p, gy_s, gzh_p, gzh_s = _poff(zcat, o_off), W_out, gc_p, gc_s, None, K
                    else:          # z_hard lives in zcat, y is `other`
                        y_p, y_s, gy_p, gy_s, gzh_p, gzh_s = ptr(other), K, ptr(g_other), K, gc_p, gc_s
                    if kind == 'gumbel':
                        lq, glq = outs[2 + k], _dense(gex[k + 1])
                        if training:
                            lq = glq = None
                        call('vae_gumbel_bwd', y_p, y_s, ptr(lq), gy_p, gy_s, gzh_p, gzh_s, ptr(glq),
                             _poff(glogits, i_off), W_in, B, K, tau, dt, st)
                        k += 2
                    else:
                        call('vae_bernoulli_bwd', _poff(logits, i_off), rs, y_p, y_s, gy_p, gy_s, gzh_p, gzh_s,
                             _poff(glogits, i_off), W_in, B, K, tau, dt, st)
                        k += 1
                    i_off, o_off = i_off + K, o_off + K
        return glogits, None, None, None, None, None


def concat_leaves(logits, specs, training, noises, seeds, row0=0):
    """Run kernel-backed Concat children over one [B, sum(in)] logits buffer.  Returns the flat tuple
    (zcat, then per child: gauss -> (empty placeholder), kl, means[2]; gumbel -> other, log_q_z; bern -> other), where
    `other` is y in eval (z_hard went into zcat) and an empty placeholder in training (y went into zcat; z_hard and
    log_q_z are then produced on first read by the module classes)."""
    logits = _rows(logits)
    B = logits.shape[0]
    fixed = []
    for spec, n in zip(specs, noises):
        if n is not None:
            n = n.reshape(B, spec[1]).to(logits.dtype).contiguous()
        fixed.append(n)
    return _ConcatLeaves.apply(logits, tuple(specs), bool(training), tuple(fixed), tuple(seeds), int(row0))


# --------------------------------------------------------------------------------------
# Multi-sample forms: S Gaussian draws from one logits read; S-sample NLL + ELBO reading x once
# (abstract_vae.py:347-354 loss_function K > 1, msg.py:55-69, abstract_vae.py:301-311)
# --------------------------------------------------------------------------------------
class _GaussMulti(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, noise, S, eps_tol, seed, offset, row0, mean_scale, want_logvar=True):
        ctx.set_materialize_grads(False)
        dev = check_device(logits, noise)
        B, L = logits.shape[0], logits.shape[1] // 2
        z = torch.empty((S, B, L), dtype=logits.dtype, device=dev)
        logvar = torch.empty((B, L), dtype=logits.dtype, device=dev) if want_logvar else None
        kl = torch.empty((B,), dtype=logits.dtype, device=dev)
        means = _scalars(2, dev, B)
        if mean_scale is None:
            mean_scale = 1.0 / max(B * L, 1)
        with _Guard(dev):
            call('vae_gauss_reparam_multi_fwd', ptr(logits), logits.stride(0), ptr(noise), ptr(z), ptr(logvar), ptr(kl),
                 ptr(means), ptr(workspace(dev)), B, L, S, eps_tol, mean_scale, seed & _MASK64, offset & _MASK64, row0,
                 dtype_code(logits), stream_ptr(dev))
        ctx.save_for_backward(logits, noise)
        ctx.cfg = (S, eps_tol, seed, offset, row0, mean_scale)
        if logvar is None:
            logvar = logits.new_empty((0,))
            ctx.mark_non_differentiable(logvar)
        return z, logvar, kl, means

    @staticmethod
    def backward(ctx, gz, glogvar, gkl, gmeans):
        logits, noise = ctx.saved_tensors
        S, eps_tol, seed, offset, row0, mean_scale = ctx.cfg
        dev = logits.device
        B, L = logits.shape[0], logits.shape[1] // 2
        if glogvar is not None and glogvar.numel() == 0:
            glogvar = None
        gz, glogvar, gkl = _dense(gz), _dense(glogvar), _dense(gkl)
        gmeans = None if gmeans is None else gmeans.contiguous().float()
        glogits = torch.empty((B, 2 * L), dtype=logits.dtype, device=dev)
        with _Guard(dev):
            call('vae_gauss_reparam_multi_bwd', ptr(logits), logits.stride(0), ptr(noise), ptr(gz), ptr(glogvar),
                 ptr(gkl), ptr(gmeans), ptr(glogits), B, L, S, eps_tol, mean_scale, seed & _MASK64, offset & _MASK64,
                 row0, dtype_code(logits), stream_ptr(dev))
        return glogits, None, None, None, None, None, None, None, None


def gauss_reparam_multi(logits, num_samples, noise=None, eps_tol=1e-6, seed=0, offset=0, row0=0, mean_scale=None,
                        want_logvar=True):
    """logits [B, 2L] -> (z [S, B, L], logvar [B, L], kl [B], means float32[2]): S training-mode draws
    from one read of the logits.  `noise` [S, B, L] (or a list of S [B, L] tensors) injects the draws;
    None: Philox, draw s on call offset `offset + s`."""
    logits = _rows(logits)
    B, L = logits.shape[0], logits.shape[1] // 2
    S = int(num_samples)
    if noise is not None:
        if isinstance(noise, (list, tuple)):
            noise = torch.stack([n.reshape(B, L) for n in noise], 0)
        noise = noise.reshape(S, B, L).to(logits.dtype).contiguous()
    z, logvar, kl, means = _GaussMulti.apply(logits, noise, S, float(eps_tol), int(seed), int(offset), int(row0),
                                             mean_scale, bool(want_logvar))
    return z, (logvar if want_logvar else None), kl, means


class _NllMultiElbo(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, kld, add, mi, nll_code, kl_beta, inv_count, with_elbo, *recons):
        ctx.set_materialize_grads(False)
        dev = check_device(x, kld, add, mi, *recons)
        B, D = x.shape
        S = len(recons)
        nll = torch.empty((B,), dtype=x.dtype, device=dev)
        if with_elbo:
            loss, elbo = torch.empty_like(nll), torch.empty_like(nll)
            means = _scalars(6, dev, B)
            ws = workspace(dev)
        else:
            loss = elbo = means = ws = None
        with _Guard(dev):
            call('vae_nll_multi_elbo_fwd', ptr(x), _lib.ptr_array(recons), S, nll_code, ptr(kld), ptr(add), ptr(mi),
                 kl_beta, inv_count, ptr(nll), ptr(loss), ptr(elbo), ptr(means), ptr(ws), B, D, dtype_code(x),
                 stream_ptr(dev))
        ctx.save_for_backward(x, *recons)
        ctx.cfg = (nll_code, kl_beta, inv_count, kld is not None, add is not None, mi is not None)
        if not with_elbo:
            loss, elbo = x.new_empty((0,)), x.new_empty((0,))
            means = x.new_empty((0,), dtype=torch.float32)
            ctx.mark_non_differentiable(loss, elbo, means)
        return nll, loss, elbo, means

    @staticmethod
    def backward(ctx, gnll, gloss, gelbo, gmeans):
        x, recons = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        nll_code, kl_beta, inv_count, has_kld, has_add, has_mi = ctx.cfg
        dev = x.device
        B, D = x.shape
        gnll, gloss, gelbo = _dense(gnll), _dense(gloss), _dense(gelbo)
        if gloss is not None and gloss.numel() == 0:
            gloss = gelbo = gmeans = None
        gmeans = None if gmeans is None else gmeans.contiguous().float()
        grecons = [torch.empty_like(r) for r in recons]
        gkld = torch.empty((B,), dtype=x.dtype, device=dev) if has_kld and ctx.needs_input_grad[1] else None
        gadd = torch.empty((B,), dtype=x.dtype, device=dev) if has_add and ctx.needs_input_grad[2] else None
        gmi = torch.empty((B,), dtype=x.dtype, device=dev) if has_mi and ctx.needs_input_grad[3] else None
        with _Guard(dev):
            call('vae_nll_multi_elbo_bwd', ptr(x), _lib.ptr_array(recons), _lib.ptr_array(grecons), len(recons),
                 nll_code, ptr(gnll), ptr(gloss), ptr(gelbo), ptr(gmeans), kl_beta, inv_count, ptr(gkld), ptr(gadd),
                 ptr(gmi), B, D, dtype_code(x), stream_ptr(dev))
        return (None, gkld, gadd, gmi, None, None, None, None) + tuple(grecons)


MAX_SAMPLES = 8  # VAE_MAX_SAMPLES in include/vae_amd.h


def nll_multi_elbo(x, recons, nll_type, kld=None, add=None, mi=None, kl_beta=1.0, inv_count=None, with_elbo=True):
    """nll [B] = mean over the S decoder outputs in `recons` of nll(x, recon_s) (abstract_vae.py:344-354), fused
    with the ELBO combine (:356-388).  Returns (nll, loss, elbo, means[6]); x is read once per launch.
    More than MAX_SAMPLES outputs: one launch per group, the group NLLs averaged, then the combine kernel."""
    S = len(recons)
    flat = [_flat_pixels(x, r, nll_type) for r in recons]
    code, xf = flat[0][0], flat[0][1]
    rfs = [f[2] for f in flat]
    if inv_count is None:
        inv_count = 1.0 / max(xf.shape[0], 1)
    if S <= MAX_SAMPLES:
        return _NllMultiElbo.apply(xf, _dense(kld), _dense(add), _dense(mi), code, float(kl_beta), float(inv_count),
                                   bool(with_elbo), *rfs)
    total = None
    for i in range(0, S, MAX_SAMPLES):
        grp = rfs[i:i + MAX_SAMPLES]
        part = _NllMultiElbo.apply(xf, None, None, None, code, 0.0, 0.0, False, *grp)[0] * (len(grp) / S)
        total = part if total is None else total + part
    if not with_elbo:
        return total, None, None, None
    loss, elbo, means = elbo_combine(total, kld=kld, add=add, mi=mi, kl_beta=kl_beta, inv_count=inv_count)
    return total, loss, elbo, means


MAX_STEPS = 128  # VAE_MAX_STEPS in include/vae_amd.h


class _NllStepsElbo(torch.autograd.Function):
    """T per-step (target, reconstruction) pairs scored as one [T*B] launch, the tensors read where they are."""

    @staticmethod
    def forward(ctx, kld, add, mi, nll_code, kl_beta, kl_betas, inv_count, T, with_lse, *tensors):
        ctx.set_materialize_grads(False)
        xs, recons = tensors[:T], tensors[T:]
        dev = check_device(kld, add, mi, *tensors)
        B, D = xs[0].shape
        dt = recons[0].dtype
        new = lambda n: torch.empty((n,), dtype=dt, device=dev)
        nll, loss, elbo, loss_avg, elbo_avg = new(T * B), new(T * B), new(T * B), new(B), new(B)
        lse = new(B) if with_lse else None
        means = _scalars(6, dev, B)
        with _Guard(dev):
            call('vae_nll_steps_elbo_fwd', _lib.ptr_array(xs), _lib.ptr_array(recons), T, nll_code, ptr(kld), ptr(add),
                 ptr(mi), _lib.float_array(kl_betas), kl_beta, inv_count, ptr(nll), ptr(loss), ptr(elbo), ptr(loss_avg),
                 ptr(elbo_avg), ptr(lse), ptr(means), ptr(workspace(dev)), B, D, dtype_code(recons[0]),
                 stream_ptr(dev))
        ctx.save_for_backward(*tensors)
        ctx.cfg = (nll_code, kl_beta, kl_betas, inv_count, T, kld is not None, add is not None, mi is not None)
        if lse is None:
            lse = new(0)
        ctx.mark_non_differentiable(lse)
        return nll, loss, elbo, means, loss_avg, elbo_avg, lse

    @staticmethod
    def backward(ctx, gnll, gloss, gelbo, gmeans, gloss_avg, gelbo_avg, glse):
        nll_code, kl_beta, kl_betas, inv_count, T, has_kld, has_add, has_mi = ctx.cfg
        xs, recons = ctx.saved_tensors[:T], ctx.saved_tensors[T:]
        dev = xs[0].device
        B, D = xs[0].shape
        gnll, gloss, gelbo = _dense(gnll), _dense(gloss), _dense(gelbo)
        # a gradient on a per-sample average is 1/T of it on each step's entry ([T*B] vectors: tiny next to the pixels)
        if gloss_avg is not None:
            spread = (gloss_avg / T).repeat(T)
            gloss = spread if gloss is None else gloss + spread
        if gelbo_avg is not None:
            spread = (gelbo_avg / T).repeat(T)
            gelbo = spread if gelbo is None else gelbo + spread
        gmeans = None if gmeans is None else gmeans.contiguous().float()
        grecons = [torch.empty_like(r) for r in recons]
        new = lambda: torch.empty((T * B,), dtype=recons[0].dtype, device=dev)
        gkld = new() if has_kld and ctx.needs_input_grad[0] else None
        gadd = new() if has_add and ctx.needs_input_grad[1] else None
        gmi = new() if has_mi and ctx.needs_input_grad[2] else None
        with _Guard(dev):
            call('vae_nll_steps_elbo_bwd', _lib.ptr_array(xs), _lib.ptr_array(recons), _lib.ptr_array(grecons), T,
                 nll_code, ptr(gnll), ptr(gloss), ptr(gelbo), ptr(gmeans), _lib.float_array(kl_betas), kl_beta,
                 inv_count, ptr(gkld), ptr(gadd), ptr(gmi), B, D, dtype_code(recons[0]), stream_ptr(dev))
        return (gkld, gadd, gmi, None, None, None, None, None, None) + (None,) * T + tuple(grecons)


def nll_steps_elbo(xs, recons, nll_type, kld=None, add=None, mi=None, kl_beta=1.0, inv_count=None, with_lse=False):
    """The VRNN loss (vrnn.py:805-829) over T = len(recons) time steps in one launch pair: xs / recons are the per-step
    targets and decoder outputs (a tensor may repeat); kld / add / mi the per-step KL, additional-loss and
    mutual-information vectors concatenated [T*B] (abstract_vae.py:356-377 runs once per step); kl_beta one float, or a
    sequence of T floats when the annealer hands every step its own weight (abstract_vae.py:363 is inside the step loop).
    Returns (nll, loss, elbo [T*B], means float32[6], loss_avg, elbo_avg [B]); the means are over all T*B entries
    (= the mean over steps of the per-step means), the averages are per sample over the steps.
    with_lse=True appends logsumexp over the steps of elbo, [B] (not differentiable): the steps are then draws of one
    posterior per sample, AbstractVAE.likelihood's estimate (abstract_vae.py:301-315)."""
    T = len(recons)
    assert 1 <= T <= MAX_STEPS and len(xs) == T
    flat = [_flat_pixels(x, r, nll_type) for x, r in zip(xs, recons)]
    code = flat[0][0]
    xf, rf = [f[1] for f in flat], [f[2] for f in flat]
    B = xf[0].shape[0]
    for a, b in zip(xf, rf):
        if a.shape != xf[0].shape or b.shape != rf[0].shape or b.dtype != rf[0].dtype:
            raise ValueError("nll_steps_elbo: every step must have the same shapes and dtype")
    for name, v in (('kld', kld), ('add', add), ('mi', mi)):
        if v is not None and v.numel() != T * B:
            raise ValueError("nll_steps_elbo: %s must hold T*B = %d values, got %d" % (name, T * B, v.numel()))
    if inv_count is None:
        inv_count = 1.0 / max(T * B, 1)
    kl_betas = None
    if isinstance(kl_beta, (list, tuple)):
        if len(kl_beta) != T:
            raise ValueError("nll_steps_elbo: %d KL weights for %d steps" % (len(kl_beta), T))
        kl_betas = tuple(float(b) for b in kl_beta)
        kl_beta = kl_betas[0]
    out = _NllStepsElbo.apply(_dense(kld), _dense(add), _dense(mi), code, float(kl_beta), kl_betas, float(inv_count), T,
                              bool(with_lse), *xf, *rf)
    return out if with_lse else out[:6]


# --------------------------------------------------------------------------------------
# Flow-layer tails (flows/__init__.py): MADE / MADESplit, coupling, BatchNormFlow (+ Reverse).
# The running logdet of FlowSequential rides along as a [B] tensor (logdet_in -> logdet_out).
# --------------------------------------------------------------------------------------
class _MadeTail(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, m, a, logdet_in, pre_tanh, bn_stats=None):
        """a None: `m` is the packed [B, 2F] trunk output [m | a] (MADE); else two [B, F] tensors (MADESplit).
        bn_stats: the stats buffer of the BatchNormFlow that consumes u next -- the launch then leaves u's column
        moments there (vae_made_tail_fwd_moments)."""
        ctx.set_materialize_grads(False)
        dev = check_device(x, m, a, logdet_in)
        B, F = x.shape
        packed = a is None
        u = torch.empty((B, F), dtype=x.dtype, device=dev)
        ld = torch.empty((B,), dtype=x.dtype, device=dev)
        import ctypes
        a_ptr = ctypes.c_void_p(m.data_ptr() + F * m.element_size()) if packed else ptr(a)
        with _Guard(dev):
            if bn_stats is not None:
                call('vae_made_tail_fwd_moments', ptr(x), x.stride(0), ptr(m), m.stride(0), a_ptr, m.stride(0) if packed
                     else a.stride(0), ptr(u), ptr(logdet_in), ptr(ld), ptr(bn_stats), B, F, int(pre_tanh), dtype_code(x),
                     stream_ptr(dev))
            else:
                call('vae_made_tail_fwd', ptr(x), x.stride(0), ptr(m), m.stride(0), a_ptr, m.stride(0) if packed else
                     a.stride(0), ptr(u), ptr(logdet_in), ptr(ld), B, F, int(pre_tanh), dtype_code(x), stream_ptr(dev))
        ctx.save_for_backward(u, m, a)
        ctx.cfg = (packed, pre_tanh, logdet_in is not None)
        return u, ld

    @staticmethod
    def backward(ctx, gu, gld):
        u, m, a = ctx.saved_tensors
        packed, pre_tanh, has_ld = ctx.cfg
        dev = u.device
        B, F = u.shape
        import ctypes
        gu, gld = _dense(gu), _dense(gld)
        gx = torch.empty_like(u) if ctx.needs_input_grad[0] else None
        if packed:
            gma = torch.empty((B, 2 * F), dtype=u.dtype, device=dev)
            gm_p, ga_p, sg = ptr(gma), ctypes.c_void_p(gma.data_ptr() + F * gma.element_size()), 2 * F
            a_p, sa = ctypes.c_void_p(m.data_ptr() + F * m.element_size()), m.stride(0)
        else:
            gm, ga = torch.empty_like(u), torch.empty_like(u)
            gm_p, ga_p, sg = ptr(gm), ptr(ga), F
            a_p, sa = ptr(a), a.stride(0)
        with _Guard(dev):
            call('vae_made_tail_bwd', ptr(u), a_p, sa, ptr(gu), ptr(gld), ptr(gx), gm_p, sg, ga_p, sg, B, F,
                 int(pre_tanh), dtype_code(u), stream_ptr(dev))
        gin = gld if has_ld else None
        if packed:
            return gx, gma, None, gin, None, None
        return gx, gm, ga, gin, None, None


def bn_stats_buffer(x):
    """The fp32 stats buffer of a BatchNormFlow over x's columns, when the MADE tail in front of it can leave the
    column moments there (None otherwise)."""
    F = x.shape[-1]
    if x.shape[0] == 0 or not _lib.load().vae_made_tail_moments_supported(F, dtype_code(x)):
        return None
    return torch.empty((_lib.load().vae_bnflow_stats_floats(F),), dtype=torch.float32, device=x.device)


def made_tail(x, m, a=None, logdet_in=None, pre_exp_tanh=False, bn_stats=None):
    """u = (x - m) * exp(-a), logdet_out = logdet_in - sum_f a  (flows/__init__.py:121-125, :176-178).
    `a` None: `m` is MADE's packed trunk output [B, 2F] = [m | a].  Returns (u [B, F], logdet_out [B]).
    bn_stats (from bn_stats_buffer): also leave u's column moments there for bnflow(..., moments=bn_stats)."""
    x = _rows(x)
    m = _rows(m)
    a = None if a is None else _rows(a)
    ld = None if logdet_in is None else logdet_in.reshape(-1).contiguous()
    if bn_stats is not None:
        ok = all(t is None or (t.data_ptr() % 16 == 0 and t.stride(0) % (16 // t.element_size()) == 0) for t in (x, m, a))
        if not ok:
            raise ValueError("made_tail(bn_stats=...): inputs must be 16-byte aligned rows")
    return _MadeTail.apply(x, m, a, ld, bool(pre_exp_tanh), bn_stats)


class _CouplingTail(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, s_raw, t_raw, mask, logdet_in):
        ctx.set_materialize_grads(False)
        dev = check_device(x, s_raw, t_raw, mask, logdet_in)
        B, F = x.shape
        y = torch.empty_like(x)
        ld = torch.empty((B,), dtype=x.dtype, device=dev)
        with _Guard(dev):
            call('vae_coupling_tail_fwd', ptr(x), ptr(s_raw), ptr(t_raw), ptr(mask), ptr(y), ptr(logdet_in), ptr(ld),
                 B, F, dtype_code(x), stream_ptr(dev))
        ctx.save_for_backward(x, s_raw, mask)
        ctx.has_ld = logdet_in is not None
        return y, ld

    @staticmethod
    def backward(ctx, gy, gld):
        x, s_raw, mask = ctx.saved_tensors
        dev = x.device
        B, F = x.shape
        gy, gld = _dense(gy), _dense(gld)
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gs = torch.empty_like(x) if s_raw is not None else None
        gt = torch.empty_like(x)
        with _Guard(dev):
            call('vae_coupling_tail_bwd', ptr(x), ptr(s_raw), ptr(mask), ptr(gy), ptr(gld), ptr(gx), ptr(gs), ptr(gt),
                 B, F, dtype_code(x), stream_ptr(dev))
        return gx, gs, gt, None, (gld if ctx.has_ld else None)


def coupling_tail(x, s_raw, t_raw, mask, logdet_in=None):
    """y = x * exp(s_raw*(1-mask)) + t_raw*(1-mask), logdet_out = logdet_in + sum_f s_raw*(1-mask)
    (flows/__init__.py:524-528); s_raw None = the additive layer (:458-460).  mask: [F] floats."""
    x = x.contiguous()
    mask = mask.to(device=x.device, dtype=torch.float32).contiguous()
    ld = None if logdet_in is None else logdet_in.reshape(-1).contiguous()
    return _CouplingTail.apply(x, _dense(s_raw), t_raw.contiguous(), mask, ld)


class _BNFlow(torch.autograd.Function):
    """reduce_fn / count: statistics over the rows of several ranks (`reduce_fn(t)` sums a small tensor over ranks in
    place, `count` = rows over all ranks): local moments -> one reduction of 2F+1 doubles -> the apply; the backward
    likewise reduces its 2F+1 column sums between its two kernels."""

    @staticmethod
    def forward(ctx, x, log_gamma, beta, mean_in, var_in, logdet_in, eps, training, reverse, reduce_fn, count,
                moments=None):
        ctx.set_materialize_grads(False)
        dev = check_device(x, log_gamma, beta, mean_in, var_in, logdet_in)
        B, F = x.shape
        y = torch.empty_like(x)
        ld = torch.empty((B,), dtype=x.dtype, device=dev)
        # `moments`: a stats buffer in which the MADE tail that produced x has already left x's column moments
        stats = moments if moments is not None else \
            torch.empty((_lib.load().vae_bnflow_stats_floats(F),), dtype=torch.float32, device=dev)
        lg32, b32 = log_gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        sync = bool(training) and reduce_fn is not None and count is not None
        with _Guard(dev):
            if sync:
                # this rank's moments (y == NULL: no apply), summed over ranks as n*mean, n*E[x^2], n in double
                pack = torch.zeros((2 * F + 1,), dtype=torch.float64, device=dev)
                if B > 0:
                    call('vae_bnflow_fwd', ptr(x), ptr(lg32), ptr(b32), None, None, ptr(stats), None, None, None, B, F,
                         eps, 1, int(reverse), dtype_code(x), stream_ptr(dev))
                    m, v = stats[:F].double(), stats[F:2 * F].double() - eps
                    pack[:F] = m * B
                    pack[F:2 * F] = (v + m * m) * B
                    pack[2 * F] = B
                reduce_fn(pack)
                gmean = pack[:F] / count
                gvar = pack[F:2 * F] / count - gmean * gmean + eps
                mean_in, var_in = gmean.float().contiguous(), gvar.float().contiguous()
            call('vae_bnflow_fwd', ptr(x), ptr(lg32), ptr(b32), ptr(mean_in), ptr(var_in), ptr(stats), ptr(y),
                 ptr(logdet_in), ptr(ld), B, F, eps, (2 if moments is not None else 1) if (training and not sync) else 0,
                 int(reverse), dtype_code(x), stream_ptr(dev))
        ctx.save_for_backward(x, lg32, stats)
        ctx.cfg = (training, reverse, logdet_in is not None, log_gamma.dtype, reduce_fn if sync else None, count)
        ctx.mark_non_differentiable(stats)
        return y, ld, stats

    @staticmethod
    def backward(ctx, gy, gld, _gstats):
        x, lg32, stats = ctx.saved_tensors
        training, reverse, has_ld, pdt, reduce_fn, count = ctx.cfg
        dev = x.device
        B, F = x.shape
        gld = _dense(gld)
        gy = torch.zeros_like(x) if gy is None else gy.contiguous()
        gx = torch.empty_like(x)
        with _Guard(dev):
            if reduce_fn is None:
                call('vae_bnflow_bwd', ptr(x), ptr(gy), ptr(gld), ptr(stats), ptr(gx), B, F, int(training),
                     int(reverse), dtype_code(x), stream_ptr(dev))
                sums = stats[6 * F:8 * F + 1]
            else:
                if B > 0:
                    call('vae_bnflow_bwd_sums', ptr(x), ptr(gy), ptr(gld), ptr(stats), B, F, int(reverse),
                         dtype_code(x), stream_ptr(dev))
                else:
                    stats[6 * F:8 * F + 1].zero_()
                sums = stats[6 * F:8 * F + 1].clone()      # this rank's rows: the parameter gradients (DDP sums them)
                reduce_fn(stats[6 * F:8 * F + 1])
                call('vae_bnflow_bwd_apply', ptr(x), ptr(gy), ptr(stats), ptr(gx), B, F, int(count), int(reverse),
                     dtype_code(x), stream_ptr(dev))
        g_beta = sums[:F].to(pdt)
        g_lg = (torch.exp(lg32) * sums[F:2 * F] + sums[2 * F]).to(pdt)
        return gx, g_lg, g_beta, None, None, (gld if has_ld else None), None, None, None, None, None, None


class _ActNorm(torch.autograd.Function):
    """ActNorm 'direct' (flows/__init__.py:292-295): y = (x - bias) * exp(weight), logdet += sum(weight).  The same
    per-column affine map as BatchNormFlow in eval mode with mean = bias, var = 1, log_gamma = weight, beta = 0, so it
    runs on those kernels (csrc/flowtail.hip); the backward's column sums give the parameter gradients:
    d bias = -exp(weight) * sum_b gy,  d weight = exp(weight) * sum_b gy * (x - bias) + sum_b glogdet."""

    @staticmethod
    def forward(ctx, x, weight, bias, logdet_in):
        ctx.set_materialize_grads(False)
        dev = check_device(x, weight, bias, logdet_in)
        B, F = x.shape
        y = torch.empty_like(x)
        ld = torch.empty((B,), dtype=x.dtype, device=dev)
        stats = torch.empty((_lib.load().vae_bnflow_stats_floats(F),), dtype=torch.float32, device=dev)
        w32, b32 = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        zeros, ones = torch.zeros_like(w32), torch.ones_like(w32)
        with _Guard(dev):
            call('vae_bnflow_fwd', ptr(x), ptr(w32), ptr(zeros), ptr(b32), ptr(ones), ptr(stats), ptr(y),
                 ptr(logdet_in), ptr(ld), B, F, 0.0, 0, 0, dtype_code(x), stream_ptr(dev))
        ctx.save_for_backward(x, w32, stats)
        ctx.cfg = (logdet_in is not None, weight.dtype)
        return y, ld

    @staticmethod
    def backward(ctx, gy, gld):
        x, w32, stats = ctx.saved_tensors
        has_ld, pdt = ctx.cfg
        dev = x.device
        B, F = x.shape
        gld = _dense(gld)
        gy = torch.zeros_like(x) if gy is None else gy.contiguous()
        gx = torch.empty_like(x)
        with _Guard(dev):
            call('vae_bnflow_bwd', ptr(x), ptr(gy), ptr(gld), ptr(stats), ptr(gx), B, F, 0, 0, dtype_code(x),
                 stream_ptr(dev))
        sums = stats[6 * F:8 * F + 1]
        scale = torch.exp(w32)
        g_weight = (scale * sums[F:2 * F] + sums[2 * F]).to(pdt)
        g_bias = (-scale * sums[:F]).to(pdt)
        return gx, g_weight, g_bias, (gld if has_ld else None)


def actnorm(x, weight, bias, logdet_in=None):
    """-> (y [B, F], logdet_out [B]) of an ActNorm layer with the given parameters (no data-dependent init here)."""
    x = x.contiguous()
    F = x.shape[1]
    if weight.numel() != F or bias.numel() != F:
        raise ValueError("actnorm: the parameter vectors must have x.shape[1] = %d entries" % F)
    ld = None if logdet_in is None else logdet_in.reshape(-1).contiguous()
    if x.shape[0] == 0:
        return torch.empty_like(x), x.new_empty((0,))
    return _ActNorm.apply(x, weight, bias, ld)


def bnflow(x, log_gamma, beta, running_mean=None, running_var=None, logdet_in=None, eps=1e-5, training=True,
           reverse=False, reduce_fn=None, count=None, moments=None):
    """BatchNormFlow 'direct' (flows/__init__.py:233-257), optionally with the Reverse that follows it in a
    MAF block folded into the store.  Returns (y [B, F], logdet_out [B], batch_mean [F], batch_var [F]) --
    the last two are what the caller folds into its running statistics (:240-246).  With `reduce_fn` and `count`
    (rows over all ranks) the batch statistics are those of the whole sharded batch."""
    x = x.contiguous()
    ld = None if logdet_in is None else logdet_in.reshape(-1).contiguous()
    F = x.shape[1]
    if log_gamma.numel() != F or beta.numel() != F or (not training and (running_mean.numel() != F or
                                                                         running_var.numel() != F)):
        raise ValueError("bnflow: the parameter vectors must have x.shape[1] = %d entries" % F)
    mi = None if training else running_mean.float().contiguous()
    vi = None if training else running_var.float().contiguous()
    if moments is not None and (not training or reduce_fn is not None):
        moments = None      # only the plain training forward takes its statistics from the producer's launch
    y, ldo, stats = _BNFlow.apply(x, log_gamma, beta, mi, vi, ld, float(eps), bool(training), bool(reverse),
                                  reduce_fn, count, moments)
    return y, ldo, stats[:F], stats[F:2 * F]
