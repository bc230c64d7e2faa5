"""Oracle: functional CPU restatement of the reference FCN Q-network.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows
  /root/reference/networks.py:6-26   (FCN.__init__ / FCN.forward)
  /root/reference/resnet.py:19-47    (BasicBlock)
  /root/reference/resnet.py:52-104   (ResNet.__init__, _make_layer, features)

The network is expressed over a flat ``state`` dict keyed exactly like the
reference's ``DataParallel(FCN).state_dict()`` (138 entries, ``module.`` prefix,
OIHW conv weights), so fixtures and checkpoints are interchangeable.  All
arithmetic is the same sequence of ATen ops the reference executes, which makes
the fp32 result bit-identical to the imported reference on the same host.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

BN_EPS = 1e-5        # nn.BatchNorm2d default, resnet.py:24 / networks.py:11
BN_MOMENTUM = 0.1    # nn.BatchNorm2d default
PREFIX = 'module.'   # policies.py:39 wraps every net in DataParallel


def _bn_entries(name, c):
    return [(name + '.weight', (c,), 'bn_weight'), (name + '.bias', (c,), 'bn_bias'),
            (name + '.running_mean', (c,), 'bn_mean'), (name + '.running_var', (c,), 'bn_var'),
            (name + '.num_batches_tracked', (), 'bn_count')]


def state_spec(num_input_channels, num_output_channels):
    """Ordered (key, shape, kind) list == reference state_dict() order.

    resnet.py:55-68 (stem, layer1-4, fc), networks.py:10-14 (head).
    """
    spec = []
    r = PREFIX + 'resnet18.'
    spec.append((r + 'conv1.weight', (64, num_input_channels, 7, 7), 'conv_w'))
    spec += _bn_entries(r + 'bn1', 64)
    inplanes = 64
    for li, planes in enumerate((64, 128, 256, 512), start=1):
        for bi in range(2):
            b = '%slayer%d.%d.' % (r, li, bi)
            cin = inplanes if bi == 0 else planes
            spec.append((b + 'conv1.weight', (planes, cin, 3, 3), 'conv_w'))
            spec += _bn_entries(b + 'bn1', planes)
            spec.append((b + 'conv2.weight', (planes, planes, 3, 3), 'conv_w'))
            spec += _bn_entries(b + 'bn2', planes)
            if bi == 0 and cin != planes:          # resnet.py:79-83
                spec.append((b + 'downsample.0.weight', (planes, cin, 1, 1), 'conv_w'))
                spec += _bn_entries(b + 'downsample.1', planes)
        inplanes = planes
    spec.append((r + 'fc.weight', (1000, 512), 'fc_w'))   # resnet.py:68, never used by features()
    spec.append((r + 'fc.bias', (1000,), 'fc_b'))
    p = PREFIX
    spec.append((p + 'conv1.weight', (128, 512, 1, 1), 'conv_w'))
    spec.append((p + 'conv1.bias', (128,), 'conv_b'))
    spec += _bn_entries(p + 'bn1', 128)
    spec.append((p + 'conv2.weight', (32, 128, 1, 1), 'conv_w'))
    spec.append((p + 'conv2.bias', (32,), 'conv_b'))
    spec += _bn_entries(p + 'bn2', 32)
    spec.append((p + 'conv3.weight', (num_output_channels, 32, 1, 1), 'conv_w'))
    spec.append((p + 'conv3.bias', (num_output_channels,), 'conv_b'))
    return spec


def is_parameter(kind):
    return kind in ('conv_w', 'conv_b', 'bn_weight', 'bn_bias', 'fc_w', 'fc_b')


def has_gradient(kind):
    """fc.* are parameters but features() never touches them (resnet.py:93-104)."""
    return kind in ('conv_w', 'conv_b', 'bn_weight', 'bn_bias')


class _BN:
    """BatchNorm2d forward, functional.  train mode: batch statistics (biased var
    for normalisation, unbiased for the running estimate, momentum 0.1) and
    num_batches_tracked += 1, exactly what nn.BatchNorm2d does."""

    def __init__(self, state, training, update_buffers=True):
        self.s = state
        self.training = training
        self.update = update_buffers

    def __call__(self, x, name):
        s = self.s
        if self.training:
            rm, rv = s[name + '.running_mean'], s[name + '.running_var']
            if not self.update:
                rm, rv = rm.clone(), rv.clone()
            else:
                s[name + '.num_batches_tracked'] += 1
            return F.batch_norm(x, rm, rv, s[name + '.weight'], s[name + '.bias'],
                                True, BN_MOMENTUM, BN_EPS)
        return F.batch_norm(x, s[name + '.running_mean'], s[name + '.running_var'],
                            s[name + '.weight'], s[name + '.bias'], False, BN_MOMENTUM, BN_EPS)


def fcn_forward(state, x, training, taps=None, update_buffers=True):
    """FCN.forward (networks.py:16-26) on NCHW ``x``.

    ``state``: dict of tensors keyed as state_spec(); BN buffers are updated IN
    PLACE when ``training`` (as the reference modules do) unless ``update_buffers`` is False
    (a DataParallel replica on device != 0, whose buffer updates are discarded).  ``taps``: optional
    OrderedDict that receives named intermediate activations (for bisecting).
    """
    bn = _BN(state, training, update_buffers)
    r = PREFIX + 'resnet18.'

    def tap(name, t):
        if taps is not None:
            taps[name] = t.detach()
        return t

    # resnet.py:94-97
    x = F.conv2d(x, state[r + 'conv1.weight'], None, stride=2, padding=3)
    tap('stem.conv', x)
    x = F.relu(bn(x, r + 'bn1'))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    tap('stem.pool', x)
    # resnet.py:99-102, BasicBlock.forward resnet.py:31-47
    for li in range(1, 5):
        for bi in range(2):
            b = '%slayer%d.%d.' % (r, li, bi)
            identity = x
            out = F.conv2d(x, state[b + 'conv1.weight'], None, stride=1, padding=1)
            out = F.relu(bn(out, b + 'bn1'))
            out = F.conv2d(out, state[b + 'conv2.weight'], None, stride=1, padding=1)
            out = bn(out, b + 'bn2')
            if (b + 'downsample.0.weight') in state:
                identity = F.conv2d(x, state[b + 'downsample.0.weight'], None, stride=1)
                identity = bn(identity, b + 'downsample.1')
            x = F.relu(out + identity)
            tap('layer%d.%d' % (li, bi), x)
    # networks.py:18-26
    p = PREFIX
    x = F.conv2d(x, state[p + 'conv1.weight'], state[p + 'conv1.bias'])
    x = F.relu(bn(x, p + 'bn1'))
    tap('head.a1', x)
    x = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True)
    x = F.conv2d(x, state[p + 'conv2.weight'], state[p + 'conv2.bias'])
    x = F.relu(bn(x, p + 'bn2'))
    tap('head.a2', x)
    x = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True)
    x = F.conv2d(x, state[p + 'conv3.weight'], state[p + 'conv3.bias'])
    return x


def state_from_numpy(np_state, dtype=torch.float32):
    """numpy dict (reference key order) -> torch CPU state, floats cast to ``dtype``."""
    out = OrderedDict()
    for k, v in np_state.items():
        t = torch.from_numpy(v.copy()) if v.shape != () else torch.tensor(v.item())
        if t.dtype.is_floating_point:
            t = t.to(dtype)
        else:
            t = t.to(torch.int64)
        out[k] = t
    return out
