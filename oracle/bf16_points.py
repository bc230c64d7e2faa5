"""Oracle for the plain-bf16 plans: the reference network in fp64 arithmetic with operands ROUNDED TO bf16 exactly where
libsimq's bf16 plan rounds them.  TEST INFRASTRUCTURE (see oracle/__init__.py).

Why: /root/reference/train.py:108-141 in bf16 is only defined through torch.autocast, whose own gradient is 0.4-0.6 off the fp64
gradient on these nets -- a bar calibrated on it cannot see a wrong bf16 kernel.  This model removes the part of that distance
that is DESIGN (where a value is stored as bf16) and leaves what is the kernels' own: accumulation order and fp32 (instead of fp64)
arithmetic between two rounding points.  With `points=False` every rounding is the identity and the functions below ARE
oracle.fcn.fcn_forward / oracle.learner.train_step in fp64 (tests/test_oracle_golden.py holds them to that, and to landing inside
the reference's autocast calibration with the points on).

The rounding points (spatial-intention-maps_amd/csrc/forward.hip forward_impl, backward.hip backward_impl), reference layer by layer
(networks.py:16-26, resnet.py:31-47,93-104):
  * every convolution except the last 1x1 (conv3) multiplies bf16(operand) x bf16(weight), products and sums exact (fp32 accumulate
    in the kernels, fp64 here); the network input is rounded inside the first convolution;
  * the PRE-BatchNorm output of those convolutions is stored as bf16, except head conv2's (fp32); the batch statistics come from the
    UNROUNDED accumulators, the normalisation is applied to the stored (rounded) value;
  * post-BatchNorm activations inside the residual blocks exist as bf16 only (operand, residual and ReLU mask alike); the pooled stem
    output and the head's first activation exist in fp32 too: block 1's identity shortcut adds the fp32 pooled map, conv2 of the head
    multiplies bf16(a1);
  * eval mode (target net): BatchNorm is folded into the convolution epilogues -- conv -> scale/shift -> (+ bf16 residual) -> ReLU ->
    ONE rounding to the bf16 plane; head conv2 stays fp32;
  * backward: the BatchNorm input gradients (= the dy operands of every weight gradient / dgrad) and the activation gradients that travel
    between the residual blocks' kernels (gradient w.r.t. a block output / a block's inner activation / the pooled map) are stored as
    bf16; weight gradients accumulate in fp32.
Arithmetic BETWEEN two points is fp64 here and fp32 in the kernels (BatchNorm affine map, bilinear weights, Huber): that difference,
1e-7 relative, is what the parity bars of tests/test_gpu_sized.py measure after the network amplified it.
"""
import torch
import torch.nn.functional as F

from . import fcn
from .learner import apply_transform, clip_grad_norm, grad_keys, sgd_step

PREFIX = fcn.PREFIX
BN_EPS, BN_MOMENTUM = fcn.BN_EPS, fcn.BN_MOMENTUM


def rnd(t):
    """Round-to-nearest-even to bf16, back in the tensor's own dtype."""
    return t.to(torch.bfloat16).to(t.dtype)


class _RoundFwd(torch.autograd.Function):
    """value stored as bf16 (straight-through for the gradient)"""
    @staticmethod
    def forward(ctx, x):
        return rnd(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundBwd(torch.autograd.Function):
    """gradient w.r.t. this value is stored as bf16"""
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return rnd(g)


class Points:
    """Where the plan rounds.  on=False: every point is the identity (the plain fp64 oracle)."""

    def __init__(self, on=True):
        self.on = on

    def fwd(self, x):                       # stored as bf16
        return _RoundFwd.apply(x) if self.on else x

    def bwd(self, x):                       # its gradient is stored as bf16
        return _RoundBwd.apply(x) if (self.on and x.requires_grad) else x

    def both(self, x):
        return self.fwd(self.bwd(x))

    def w(self, w):                         # weight plane (bf16 copy of the fp32 master weight; the gradient stays fp32)
        return _RoundFwd.apply(w) if self.on else w


def _bn_train(y_stat, y_used, gamma, beta, state, name, update):
    """Train-mode BatchNorm2d whose statistics come from `y_stat` (the unrounded accumulators) and whose affine map is applied to
    `y_used` (the stored value).  Running statistics as nn.BatchNorm2d (momentum 0.1, unbiased variance)."""
    n = y_stat.numel() // y_stat.shape[1]
    mean = y_stat.mean(dim=(0, 2, 3))
    var = y_stat.var(dim=(0, 2, 3), unbiased=False)
    if update:
        with torch.no_grad():
            state[name + '.running_mean'].mul_(1 - BN_MOMENTUM).add_(mean.detach(), alpha=BN_MOMENTUM)
            state[name + '.running_var'].mul_(1 - BN_MOMENTUM).add_(var.detach() * (n / max(n - 1, 1)), alpha=BN_MOMENTUM)
            state[name + '.num_batches_tracked'] += 1
    invstd = torch.rsqrt(var + BN_EPS)
    scale = gamma * invstd
    shift = beta - mean * scale
    return y_used * scale[None, :, None, None] + shift[None, :, None, None]


def _bn_eval_coeff(state, name):
    invstd = torch.rsqrt(state[name + '.running_var'] + BN_EPS)
    scale = state[name + '.weight'] * invstd
    return scale, state[name + '.bias'] - state[name + '.running_mean'] * scale


def fcn_forward(state, x, training, points=True, update_buffers=True, taps=None):
    """FCN.forward (networks.py:16-26) in the dtype of `state` (use fp64) with the bf16 plan's rounding points.  `training`: batch
    statistics (train / train-no-grad modes of simq_forward), else the folded eval form of the target net."""
    P = Points(points)
    r = PREFIX + 'resnet18.'
    p = PREFIX
    conv = F.conv2d

    def tap(name, t):           # optional named intermediates (bisecting a disagreement with the HIP plan)
        if taps is not None:
            taps[name] = t.detach()
        return t

    def bn_t(y, name):          # train: statistics from the accumulators, affine map on the stored pre-BN value
        y = P.bwd(y)            # (its gradient = the dy operand of this convolution's weight gradient / dgrad: a bf16 plane)
        return _bn_train(y, P.fwd(y), state[name + '.weight'], state[name + '.bias'], state, name, update_buffers)

    if training:
        y0 = conv(P.fwd(x), P.w(state[r + 'conv1.weight']), None, stride=2, padding=3)
        pooled = F.max_pool2d(F.relu(bn_t(y0, r + 'bn1')), kernel_size=3, stride=2, padding=1)
        tap('stem.conv', y0)
        tap('stem.pool', pooled)
        pooled = P.bwd(pooled)                      # gradient w.r.t. the pooled map travels as bf16
        cur_op, cur_id = P.fwd(pooled), pooled      # operand plane | the identity shortcut of block 1 reads the fp32 map
        for li in range(1, 5):
            for bi in range(2):
                b = '%slayer%d.%d.' % (r, li, bi)
                y1 = conv(cur_op, P.w(state[b + 'conv1.weight']), None, stride=1, padding=1)
                a1 = P.both(F.relu(bn_t(y1, b + 'bn1')))
                y2 = conv(a1, P.w(state[b + 'conv2.weight']), None, stride=1, padding=1)
                o = bn_t(y2, b + 'bn2')
                if (b + 'downsample.0.weight') in state:
                    yd = conv(cur_op, P.w(state[b + 'downsample.0.weight']), None, stride=1)
                    identity = bn_t(yd, b + 'downsample.1')
                else:
                    identity = cur_id
                out = P.both(F.relu(o + identity))
                tap('layer%d.%d' % (li, bi), out)
                cur_op = cur_id = out
        yh1 = conv(cur_op, P.w(state[p + 'conv1.weight']), state[p + 'conv1.bias'])
        a1 = tap('head.a1', F.relu(bn_t(yh1, p + 'bn1')))           # fp32 activation; conv2 multiplies its bf16 plane
        z2 = conv(P.fwd(a1), P.w(state[p + 'conv2.weight']), state[p + 'conv2.bias'])
        z2 = P.bwd(z2)                              # (the gradient w.r.t. conv2's output is the bf16 operand of its weight gradient / dgrad)
        # conv2 is 1x1: conv2(upsample(a)) == upsample(conv2(a)) (networks.py:21-22 commuted, docs/history.md 4); BatchNorm 2 sees the 48x48 map
        yh2 = F.interpolate(z2, scale_factor=2, mode='bilinear', align_corners=True)
        a2 = F.relu(_bn_train(yh2, yh2, state[p + 'bn2.weight'], state[p + 'bn2.bias'], state, p + 'bn2', update_buffers))
        q = F.interpolate(a2, scale_factor=2, mode='bilinear', align_corners=True)
        return conv(q, state[p + 'conv3.weight'], state[p + 'conv3.bias'])

    # eval: BatchNorm folded into the convolution epilogues, one rounding per layer output
    def folded(xop, wname, bname, bnname, identity=None, relu=True):
        sc, sh = _bn_eval_coeff(state, bnname)
        y = conv(xop, P.w(state[wname]), state[bname] if bname else None, stride=1, padding=state[wname].shape[-1] // 2)
        y = y * sc[None, :, None, None] + sh[None, :, None, None]
        if identity is not None:
            y = y + identity
        return P.fwd(F.relu(y) if relu else y)

    y0 = P.fwd(conv(P.fwd(x), P.w(state[r + 'conv1.weight']), None, stride=2, padding=3))
    sc, sh = _bn_eval_coeff(state, r + 'bn1')
    pooled = F.max_pool2d(F.relu(y0 * sc[None, :, None, None] + sh[None, :, None, None]), kernel_size=3, stride=2, padding=1)
    tap('stem.pool', pooled)
    cur = P.fwd(pooled)
    for li in range(1, 5):
        for bi in range(2):
            b = '%slayer%d.%d.' % (r, li, bi)
            a1 = folded(cur, b + 'conv1.weight', None, b + 'bn1')
            identity = cur
            if (b + 'downsample.0.weight') in state:
                identity = folded(cur, b + 'downsample.0.weight', None, b + 'downsample.1', relu=False)
            cur = tap('layer%d.%d' % (li, bi), folded(a1, b + 'conv2.weight', None, b + 'bn2', identity=identity))
    a1 = folded(cur, p + 'conv1.weight', p + 'conv1.bias', p + 'bn1')
    sc, sh = _bn_eval_coeff(state, p + 'bn2')
    z2 = conv(a1, P.w(state[p + 'conv2.weight']), state[p + 'conv2.bias']) * sc[None, :, None, None] + sh[None, :, None, None]
    a2 = F.relu(F.interpolate(z2, scale_factor=2, mode='bilinear', align_corners=True))
    q = F.interpolate(a2, scale_factor=2, mode='bilinear', align_corners=True)
    return conv(q, state[p + 'conv3.weight'], state[p + 'conv3.bias'])


def train_step(cfg, state, target_state, spec, momentum_bufs, batch, discount_factor, lr, momentum, weight_decay, points=True,
               dtype=torch.float64, extras=None):
    """oracle.learner.train_step (train.py:108-141) over fcn_forward above: same order of the three forwards (policy on the states,
    policy train-mode no-grad on the non-final next states, target eval on them), Huber loss, clip, SGD.  `state` / `target_state`
    must already be in `dtype`."""
    B = cfg.batch_size
    state_batch = torch.cat([apply_transform(s) for s in batch.state]).to(dtype)
    action_batch = torch.tensor(batch.action, dtype=torch.long)
    reward_batch = torch.tensor(batch.reward, dtype=torch.float32).to(dtype)
    nf = torch.cat([apply_transform(s) for s in batch.next_state if s is not None]).to(dtype)
    mask = torch.tensor([s is not None for s in batch.next_state], dtype=torch.bool)
    keys = grad_keys(spec)
    params = [state[k] for k in keys]
    for t in params:
        t.requires_grad_(True)
        t.grad = None
    q = fcn_forward(state, state_batch, True, points)
    q_sa = q.view(B, -1).gather(1, action_batch.view(B, 1)).view(-1)
    nsv = torch.zeros(B, dtype=dtype)
    with torch.no_grad():
        n = nf.size(0)
        if cfg.use_double_dqn:
            best = fcn_forward(state, nf, True, points).view(n, -1).max(1)[1].view(n, 1)
            nsv[mask] = fcn_forward(target_state, nf, False, points).view(n, -1).gather(1, best).view(-1)
        else:
            nsv[mask] = fcn_forward(target_state, nf, False, points).view(n, -1).max(1)[0]
    y = reward_batch + discount_factor * nsv
    td_error = torch.abs(q_sa - y).detach()
    loss = F.smooth_l1_loss(q_sa, y)
    grads = torch.autograd.grad(loss, params)
    for t in params:
        t.requires_grad_(False)
    grads = [g.clone() for g in grads]
    if extras is not None:                                   # (the keys oracle.learner.train_step hands out)
        from collections import OrderedDict
        extras['grads'] = OrderedDict((k, g.clone()) for k, g in zip(keys, grads))
        extras['q'] = q_sa.detach().clone()
        extras['y'] = y.detach().clone()
        extras['output'] = q.detach()
    total = clip_grad_norm(grads, cfg.grad_norm_clipping) if cfg.grad_norm_clipping is not None else None
    if extras is not None:
        extras['total_norm'] = None if total is None else float(total)
    with torch.no_grad():
        sgd_step(params, grads, momentum_bufs, lr, momentum, weight_decay)
    return {'td_error': td_error.mean().item(), 'loss': loss.detach().item()}


def dense_gradient(state, spec, x, R, points=True):
    """Gradient of sum(Q * R) (a dense upstream gradient, the autograd path of simq_backward) w.r.t. every parameter; returns (Q, grads)."""
    keys = grad_keys(spec)
    params = [state[k] for k in keys]
    for t in params:
        t.requires_grad_(True)
    q = fcn_forward(state, x, True, points)
    grads = torch.autograd.grad((q * R).sum(), params)
    for t in params:
        t.requires_grad_(False)
    return q.detach(), [g.clone() for g in grads]
