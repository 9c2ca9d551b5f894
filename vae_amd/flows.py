"""Flow layers with the reference's interface (flows/__init__.py) whose elementwise tails run in the HIP
kernels of csrc/flowtail.hip.  The masked linears and coupling nets are stock PyTorch
GEMMs; what follows them -- (x - m)*exp(-a) + logdet row sum, x*exp(s)+t, batch-norm statistics +
normalise + logdet, the Reverse permutation and FlowSequential's running `logdets += logdet` -- is fused.

Class names, constructor arguments, parameter / buffer names (state_dict keys) and `forward(inputs,
cond_inputs=None, mode='direct')` follow the reference, so a reference checkpoint of a MAF / MAF-split /
RealNVP flow loads.  `mode='inverse'` (sampling; not on the training hot path) is plain PyTorch.

The glow-side layers are here too, so that every `flow_type` of reparameterizers/flow.py:20-49 builds: ActNorm runs
its per-column affine map on the BatchNormFlow kernels; InvertibleMM / LUInvertibleMM are one GEMM and a log-determinant
(stock torch ops), Shuffle an index permutation.  Not here: Sigmoid / Logit.  `helpers.layers.BasicDenseBlock` /
`str_to_activ_module` are un-vendored: the coupling nets default to Linear -> act -> Linear -> act -> Linear and can be
injected.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as F_

_ACTS = {'relu': nn.ReLU, 'tanh': nn.Tanh, 'elu': nn.ELU, 'selu': nn.SELU, 'sigmoid': nn.Sigmoid,
         'softplus': nn.Softplus, 'leaky_relu': nn.LeakyReLU, 'identity': nn.Identity}


def str_to_activ_module(name):
    """Stand-in for helpers.layers.str_to_activ_module (un-vendored): activation name -> nn.Module class."""
    key = name.strip().lower()
    assert key in _ACTS, "unknown activation {}".format(name)
    return _ACTS[key]


def get_mask(in_features, out_features, in_flow_features, mask_type=None):
    """MADE connectivity (flows/__init__.py:16-33): unit degrees cycle over the flow dimension; an output may
    see an input of lower-or-equal degree (strictly lower for the output layer)."""
    period_in = in_flow_features if mask_type == 'input' else in_flow_features - 1
    deg_in = torch.arange(in_features) % period_in
    if mask_type == 'output':
        deg_out = torch.arange(out_features) % in_flow_features - 1
    else:
        deg_out = torch.arange(out_features) % (in_flow_features - 1)
    return (deg_out[:, None] >= deg_in[None, :]).float()


class MaskedLinear(nn.Module):
    """flows/__init__.py:48-68 (a stock GEMM with a masked weight)."""

    def __init__(self, in_features, out_features, mask, cond_in_features=None, bias=True):
        super(MaskedLinear, self).__init__()
        self.linear = nn.Linear(in_features, out_features)
        if cond_in_features is not None:
            self.cond_linear = nn.Linear(cond_in_features, out_features, bias=False)
        self.register_buffer('mask', mask)

    def forward(self, inputs, cond_inputs=None):
        out = F.linear(inputs, self.linear.weight * self.mask, self.linear.bias)
        if cond_inputs is not None:
            out = out + self.cond_linear(cond_inputs)
        return out


def _as_col(ld):
    return ld.unsqueeze(-1)


class _FlowLayer(nn.Module):
    """forward() keeps the reference's return (output, this layer's logdet); FlowSequential calls `_step`,
    which threads the running logdet [B] through the fused tail instead."""

    def forward(self, inputs, cond_inputs=None, mode='direct'):
        if mode == 'direct':
            out, ld = self._step(inputs, cond_inputs, None)
            return out, _as_col(ld)
        return self._inverse(inputs, cond_inputs)


class MADE(_FlowLayer):
    """flows/__init__.py:144-187."""

    def __init__(self, num_inputs, num_hidden, num_cond_inputs=None, act='relu', pre_exp_tanh=False):
        super(MADE, self).__init__()
        act_func = str_to_activ_module(act)
        input_mask = get_mask(num_inputs, num_hidden, num_inputs, mask_type='input')
        hidden_mask = get_mask(num_hidden, num_hidden, num_inputs)
        output_mask = get_mask(num_hidden, num_inputs * 2, num_inputs, mask_type='output')
        self.joiner = MaskedLinear(num_inputs, num_hidden, input_mask, num_cond_inputs)
        self.trunk = nn.Sequential(act_func(), MaskedLinear(num_hidden, num_hidden, hidden_mask), act_func(),
                                   MaskedLinear(num_hidden, num_inputs * 2, output_mask))

    def _step(self, inputs, cond_inputs, logdet_acc, bn_stats=None):
        ma = self.trunk(self.joiner(inputs, cond_inputs))       # [B, 2F] = [m | a]; never chunked
        return F_.made_tail(inputs, ma, None, logdet_in=logdet_acc, bn_stats=bn_stats)

    def _inverse(self, inputs, cond_inputs):
        x = torch.zeros_like(inputs)
        for i_col in range(inputs.shape[1]):
            m, a = self.trunk(self.joiner(x, cond_inputs)).chunk(2, 1)
            x[:, i_col] = inputs[:, i_col] * torch.exp(a[:, i_col]) + m[:, i_col]
        return x, -a.sum(-1, keepdim=True)


class MADESplit(_FlowLayer):
    """flows/__init__.py:71-141."""

    def __init__(self, num_inputs, num_hidden, num_cond_inputs=None, s_act='tanh', t_act='relu', pre_exp_tanh=False):
        super(MADESplit, self).__init__()
        self.pre_exp_tanh = pre_exp_tanh
        input_mask = get_mask(num_inputs, num_hidden, num_inputs, mask_type='input')
        hidden_mask = get_mask(num_hidden, num_hidden, num_inputs)
        output_mask = get_mask(num_hidden, num_inputs, num_inputs, mask_type='output')
        act_func = str_to_activ_module(s_act)
        self.s_joiner = MaskedLinear(num_inputs, num_hidden, input_mask, num_cond_inputs)
        self.s_trunk = nn.Sequential(act_func(), MaskedLinear(num_hidden, num_hidden, hidden_mask), act_func(),
                                     MaskedLinear(num_hidden, num_inputs, output_mask))
        act_func = str_to_activ_module(t_act)
        self.t_joiner = MaskedLinear(num_inputs, num_hidden, input_mask, num_cond_inputs)
        self.t_trunk = nn.Sequential(act_func(), MaskedLinear(num_hidden, num_hidden, hidden_mask), act_func(),
                                     MaskedLinear(num_hidden, num_inputs, output_mask))

    def _nets(self, x, cond_inputs):
        return self.s_trunk(self.s_joiner(x, cond_inputs)), self.t_trunk(self.t_joiner(x, cond_inputs))

    def _step(self, inputs, cond_inputs, logdet_acc, bn_stats=None):
        m, a = self._nets(inputs, cond_inputs)
        return F_.made_tail(inputs, m, a, logdet_in=logdet_acc, pre_exp_tanh=self.pre_exp_tanh, bn_stats=bn_stats)

    def _inverse(self, inputs, cond_inputs):
        x = torch.zeros_like(inputs)
        for i_col in range(inputs.shape[1]):
            m, a = self._nets(x, cond_inputs)
            if self.pre_exp_tanh:
                a = torch.tanh(a)
            x[:, i_col] = inputs[:, i_col] * torch.exp(a[:, i_col]) + m[:, i_col]
        return x, -a.sum(-1, keepdim=True)


class BatchNormFlow(_FlowLayer):
    """flows/__init__.py:216-271.  On one device the batch statistics are those of the call's rows, as in the
    reference.  When the batch is sharded, `vae_amd.dist.configure` sets `reduce_fn` and `inv_count` (= 1 / rows over
    all ranks) and the statistics -- forward moments and the backward's column sums -- are those of the whole batch, so
    an N-rank run equals the single-device run; leave them None for per-rank statistics."""

    def __init__(self, num_inputs, momentum=0.0, eps=1e-5):
        super(BatchNormFlow, self).__init__()
        self.log_gamma = nn.Parameter(torch.zeros(num_inputs))
        self.beta = nn.Parameter(torch.zeros(num_inputs))
        self.momentum = momentum
        self.eps = eps
        self.register_buffer('running_mean', torch.zeros(num_inputs))
        self.register_buffer('running_var', torch.ones(num_inputs))
        self.reduce_fn = None      # set by vae_amd.dist.configure: in-place sum of a small tensor over ranks
        self.inv_count = None      # 1 / (rows over all ranks)

    def takes_moments(self):
        """True when the MADE tail in front may leave this layer's batch moments (plain single-device training)."""
        return self.training and not (self.reduce_fn is not None and self.inv_count is not None)

    def _step(self, inputs, cond_inputs, logdet_acc, reverse=False, moments=None):
        sync = self.training and self.reduce_fn is not None and self.inv_count is not None
        y, ld, mean, var = F_.bnflow(inputs, self.log_gamma, self.beta, self.running_mean, self.running_var,
                                     logdet_in=logdet_acc, eps=self.eps, training=self.training, reverse=reverse,
                                     reduce_fn=self.reduce_fn if sync else None,
                                     count=int(round(1.0 / self.inv_count)) if sync else None, moments=moments)
        if self.training:
            self.batch_mean, self.batch_var = mean, var
            with torch.no_grad():
                self.running_mean.mul_(self.momentum).add_(mean.to(self.running_mean.dtype) * (1 - self.momentum))
                self.running_var.mul_(self.momentum).add_(var.to(self.running_var.dtype) * (1 - self.momentum))
        return y, ld

    def forward(self, inputs, cond_inputs=None, mode='direct'):
        if mode == 'direct':
            out, ld = self._step(inputs, cond_inputs, None)
            return out, ld[:1]                     # the reference's logdet here is one value, broadcast by the caller
        return self._inverse(inputs, cond_inputs)

    def _inverse(self, inputs, cond_inputs):
        mean, var = (self.batch_mean, self.batch_var) if self.training else (self.running_mean, self.running_var)
        x_hat = (inputs - self.beta) / torch.exp(self.log_gamma)
        return x_hat * var.sqrt() + mean, (-self.log_gamma + 0.5 * torch.log(var)).sum(-1, keepdim=True)


class Reverse(nn.Module):
    """flows/__init__.py:392-409.  FlowSequential folds it into a preceding BatchNormFlow's store."""

    def __init__(self, num_inputs):
        super(Reverse, self).__init__()
        self.perm = np.array(np.arange(0, num_inputs)[::-1])
        self.inv_perm = np.argsort(self.perm)

    def forward(self, inputs, cond_inputs=None, mode='direct'):
        perm = self.perm if mode == 'direct' else self.inv_perm
        return inputs[:, torch.as_tensor(perm.copy(), device=inputs.device)], \
            torch.zeros(inputs.size(0), 1, device=inputs.device)


def _default_net(total_inputs, num_hidden, num_outputs, act):
    act_func = str_to_activ_module(act)
    return nn.Sequential(nn.Linear(total_inputs, num_hidden), act_func(), nn.Linear(num_hidden, num_hidden),
                         act_func(), nn.Linear(num_hidden, num_outputs))


class CouplingLayer(_FlowLayer):
    """flows/__init__.py:466-533.  scale_net / translate_net: stock modules (the reference builds them from the
    un-vendored helpers.layers.BasicDenseBlock); injectable, default Linear-act-Linear-act-Linear."""

    def __init__(self, num_inputs, num_hidden, mask, num_cond_inputs=None, s_act='tanh', t_act='relu',
                 normalization_str='batchnorm', layer_modifier='none', scale_net=None, translate_net=None):
        super(CouplingLayer, self).__init__()
        self.num_inputs = num_inputs
        self.mask = mask
        total_inputs = num_inputs + (num_cond_inputs or 0)
        self.scale_net = scale_net if scale_net is not None else _default_net(total_inputs, num_hidden, num_inputs, s_act)
        self.translate_net = translate_net if translate_net is not None else \
            _default_net(total_inputs, num_hidden, num_inputs, t_act)

    def _nets(self, inputs, cond_inputs):
        masked = inputs * self.mask
        if cond_inputs is not None:
            masked = torch.cat([masked, cond_inputs], -1)
        return self.scale_net(masked), self.translate_net(masked)

    def _step(self, inputs, cond_inputs, logdet_acc):
        s_raw, t_raw = self._nets(inputs, cond_inputs)
        return F_.coupling_tail(inputs, s_raw, t_raw, self.mask, logdet_in=logdet_acc)

    def _inverse(self, inputs, cond_inputs):
        s_raw, t_raw = self._nets(inputs, cond_inputs)
        log_s, t = s_raw * (1 - self.mask), t_raw * (1 - self.mask)
        return (inputs - t) * torch.exp(-log_s), -log_s.sum(-1, keepdim=True)


class AdditiveCouplingLayer(_FlowLayer):
    """flows/__init__.py:412-463."""

    def __init__(self, num_inputs, num_hidden, mask, num_cond_inputs=None, act='relu', normalization_str='batchnorm',
                 layer_modifier='none', net=None):
        super(AdditiveCouplingLayer, self).__init__()
        self.num_inputs = num_inputs
        self.mask = mask
        total_inputs = num_inputs + (num_cond_inputs or 0)
        self.net = net if net is not None else _default_net(total_inputs, num_hidden, num_inputs, act)

    def _t(self, inputs, cond_inputs):
        masked = inputs * self.mask
        if cond_inputs is not None:
            masked = torch.cat([masked, cond_inputs], -1)
        return self.net(masked)

    def _step(self, inputs, cond_inputs, logdet_acc):
        return F_.coupling_tail(inputs, None, self._t(inputs, cond_inputs), self.mask, logdet_in=logdet_acc)

    def _inverse(self, inputs, cond_inputs):
        t = self._t(inputs, cond_inputs) * (1 - self.mask)
        return inputs - t, torch.zeros_like(t).sum(-1, keepdim=True)


class ActNorm(_FlowLayer):
    """flows/__init__.py:274-299: y = (x - bias) * exp(weight); the first batch seen sets weight = -log(std + 1e-12)
    and bias = mean per feature.  Same parameter names; `initialized` is plain Python state, as there."""

    def __init__(self, num_inputs):
        super(ActNorm, self).__init__()
        self.weight = nn.Parameter(torch.ones(num_inputs))
        self.bias = nn.Parameter(torch.zeros(num_inputs))
        self.initialized = False

    def _init_from(self, inputs):
        with torch.no_grad():
            self.weight.copy_(torch.log(1.0 / (inputs.std(0) + 1e-12)))
            self.bias.copy_(inputs.mean(0))
        self.initialized = True

    def _step(self, inputs, cond_inputs, logdet_acc):
        if not self.initialized:
            self._init_from(inputs)
        return F_.actnorm(inputs, self.weight, self.bias, logdet_in=logdet_acc)

    def _inverse(self, inputs, cond_inputs):
        if not self.initialized:
            self._init_from(inputs)
        total = -self.weight.sum().reshape(1, 1).expand(inputs.size(0), 1)
        return inputs * torch.exp(-self.weight) + self.bias, total


class InvertibleMM(nn.Module):
    """flows/__init__.py:302-321: y = x @ W, logdet = log|det W| for every row.  One GEMM + one slogdet: stock ops."""

    def __init__(self, num_inputs):
        super(InvertibleMM, self).__init__()
        self.W = nn.Parameter(torch.empty(num_inputs, num_inputs))
        nn.init.orthogonal_(self.W)

    def forward(self, inputs, cond_inputs=None, mode='direct'):
        logabsdet = torch.slogdet(self.W)[1].reshape(1, 1).expand(inputs.size(0), 1)
        if mode == 'direct':
            return inputs @ self.W, logabsdet
        return inputs @ torch.inverse(self.W), -logabsdet


class LUInvertibleMM(nn.Module):
    """flows/__init__.py:324-369: W = P (L_strict + I) (U_strict + diag(sign_S * exp(log_S))) from a fixed LU
    factorisation of a random orthogonal matrix; logdet = sum(log_S).  Parameters L, U, log_S as there; P, sign_S and
    the masks are non-persistent buffers (they follow .to(device) and, as in the reference, are not checkpointed)."""

    def __init__(self, num_inputs):
        super(LUInvertibleMM, self).__init__()
        import scipy.linalg
        w0 = torch.empty(num_inputs, num_inputs)
        nn.init.orthogonal_(w0)
        perm, lower, upper = scipy.linalg.lu(w0.numpy())
        diag = np.diag(upper)
        strict = torch.tril(torch.ones(num_inputs, num_inputs), -1)
        self.L = nn.Parameter(torch.from_numpy(lower))
        self.U = nn.Parameter(torch.from_numpy(upper))
        self.log_S = nn.Parameter(torch.from_numpy(np.log(np.abs(diag))))
        self.register_buffer('P', torch.from_numpy(perm), persistent=False)
        self.register_buffer('sign_S', torch.from_numpy(np.sign(diag)), persistent=False)
        self.register_buffer('L_mask', strict, persistent=False)
        self.register_buffer('U_mask', strict.t().clone(), persistent=False)
        self.register_buffer('eye', torch.eye(num_inputs), persistent=False)

    def matrix(self):
        lower = self.L * self.L_mask + self.eye
        upper = self.U * self.U_mask + torch.diag(self.sign_S * torch.exp(self.log_S))
        return self.P @ lower @ upper

    def forward(self, inputs, cond_inputs=None, mode='direct'):
        total = self.log_S.sum().reshape(1, 1).expand(inputs.size(0), 1)
        if mode == 'direct':
            return inputs @ self.matrix(), total
        return inputs @ torch.inverse(self.matrix()), -total


class Shuffle(nn.Module):
    """flows/__init__.py:372-389: a fixed random permutation of the features (numpy's generator, as there)."""

    def __init__(self, num_inputs):
        super(Shuffle, self).__init__()
        perm = np.random.permutation(num_inputs)
        self.register_buffer('perm', torch.from_numpy(perm).long(), persistent=False)
        self.register_buffer('inv_perm', torch.from_numpy(np.argsort(perm)).long(), persistent=False)

    def forward(self, inputs, cond_inputs=None, mode='direct'):
        index = self.perm if mode == 'direct' else self.inv_perm
        return inputs[:, index], torch.zeros(inputs.size(0), 1, device=inputs.device, dtype=inputs.dtype)


class FlowSequential(nn.Sequential):
    """flows/__init__.py:536-579.  'direct': every kernel-backed layer adds its logdet to the running [B]
    total inside its own launch; BatchNormFlow followed by Reverse is one launch; a MADE / MADESplit tail in front of
    a training BatchNormFlow also leaves that layer's column moments, so the layer reads its input once."""

    def forward(self, inputs, cond_inputs=None, mode='direct', logdets=None):
        self.num_inputs = inputs.size(-1)
        assert mode in ['direct', 'inverse']
        mods = list(self._modules.values())
        if mode == 'inverse':
            if logdets is None:
                logdets = torch.zeros(inputs.size(0), 1, device=inputs.device)
            for module in reversed(mods):
                inputs, logdet = module(inputs, cond_inputs, mode)
                logdets = logdets + logdet
            return inputs, logdets
        acc = None if logdets is None else logdets.reshape(-1)
        i = 0
        moments = None
        while i < len(mods):
            module = mods[i]
            if isinstance(module, BatchNormFlow):
                rev = i + 1 < len(mods) and isinstance(mods[i + 1], Reverse)
                inputs, acc = module._step(inputs, cond_inputs, acc, reverse=rev, moments=moments)
                moments = None
                i += 2 if rev else 1
                continue
            if isinstance(module, (MADE, MADESplit)) and i + 1 < len(mods) and isinstance(mods[i + 1], BatchNormFlow) \
                    and mods[i + 1].takes_moments():
                moments = F_.bn_stats_buffer(inputs.reshape(-1, inputs.size(-1)))
                inputs, acc = module._step(inputs, cond_inputs, acc, bn_stats=moments)
                i += 1
                continue
            if isinstance(module, _FlowLayer):
                inputs, acc = module._step(inputs, cond_inputs, acc)
            else:
                inputs, logdet = module(inputs, cond_inputs, mode)
                if not isinstance(module, (Reverse, Shuffle)):    # a permutation adds nothing
                    acc = logdet.reshape(-1).expand(inputs.size(0)) if acc is None else acc + logdet.reshape(-1)
            i += 1
        if acc is None:
            acc = torch.zeros(inputs.size(0), device=inputs.device, dtype=inputs.dtype)
        return inputs, acc.unsqueeze(-1)

    def log_probs(self, inputs, cond_inputs=None):
        u, log_jacob = self(inputs, cond_inputs)
        log_probs = (-0.5 * u.pow(2) - 0.5 * math.log(2 * math.pi)).sum(-1, keepdim=True)
        return (log_probs + log_jacob).sum(-1, keepdim=True)

    def sample(self, num_samples=None, noise=None, cond_inputs=None):
        device = next(self.parameters()).device
        if noise is None:
            noise = torch.empty(num_samples, self.num_inputs).normal_()
        noise = noise.to(device)
        if cond_inputs is not None:
            cond_inputs = cond_inputs.to(device)
        return self.forward(noise, cond_inputs, mode='inverse')[0]


def build_maf_flow(num_inputs, num_hidden, num_cond_inputs=None, num_blocks=5, activation_str='relu'):
    """flows/__init__.py:582-593."""
    modules = []
    for _ in range(num_blocks):
        modules += [MADE(num_inputs, num_hidden, num_cond_inputs, act=activation_str), BatchNormFlow(num_inputs),
                    Reverse(num_inputs)]
    return FlowSequential(*modules)


def build_maf_split_flow(num_inputs, num_hidden, num_cond_inputs=None, num_blocks=5, s_activation_str='tanh',
                         t_activation_str='relu'):
    """flows/__init__.py:596-609."""
    modules = []
    for _ in range(num_blocks):
        modules += [MADESplit(num_inputs, num_hidden, num_cond_inputs, s_act=s_activation_str, t_act=t_activation_str),
                    BatchNormFlow(num_inputs), Reverse(num_inputs)]
    return FlowSequential(*modules)


def build_realnvp_flow(num_inputs, num_hidden, num_cond_inputs=None, num_blocks=5, s_activation_str='tanh',
                       t_activation_str='relu', normalization_str='batchnorm', cuda=True):
    """flows/__init__.py:628-648."""
    modules = []
    device = torch.device('cuda' if cuda else 'cpu')
    mask = (torch.arange(0, num_inputs) % 2).to(device).float()
    for _ in range(num_blocks):
        modules += [CouplingLayer(num_inputs, num_hidden, mask, num_cond_inputs, s_act=s_activation_str,
                                  t_act=t_activation_str, normalization_str=normalization_str),
                    BatchNormFlow(num_inputs)]
        mask = 1 - mask
    return FlowSequential(*modules)


def build_maf_split_glow_flow(num_inputs, num_hidden, num_cond_inputs=None, num_blocks=5, s_activation_str='tanh',
                              t_activation_str='relu'):
    """flows/__init__.py:612-625: MADESplit -> BatchNormFlow -> InvertibleMM per block."""
    modules = []
    for _ in range(num_blocks):
        modules += [MADESplit(num_inputs, num_hidden, num_cond_inputs, s_act=s_activation_str, t_act=t_activation_str),
                    BatchNormFlow(num_inputs), InvertibleMM(num_inputs)]
    return FlowSequential(*modules)


def build_glow_flow(num_inputs, num_hidden, num_cond_inputs=None, num_blocks=5, activation_str='relu',
                    normalization_str='batchnorm', layer_modifier='none', cuda=True):
    """flows/__init__.py:651-673: ActNorm -> InvertibleMM -> AdditiveCouplingLayer per block, alternating masks."""
    modules = []
    device = torch.device('cuda' if cuda else 'cpu')
    mask = (torch.arange(0, num_inputs) % 2).to(device).float()
    for _ in range(num_blocks):
        modules += [ActNorm(num_inputs), InvertibleMM(num_inputs),
                    AdditiveCouplingLayer(num_inputs, num_hidden, mask, act=activation_str,
                                          num_cond_inputs=num_cond_inputs, normalization_str=normalization_str,
                                          layer_modifier=layer_modifier)]
        mask = 1 - mask
    return FlowSequential(*modules)
