"""Diagnostic (GPU box): is the B = 128 outlier of the gradient study (gs_b128_09: 2.0e-2 against the reference's 2.9e-3, insensitive to every
arithmetic option and to the backward path taken) a ReLU-mask flip at a pixel the one-hot TD gradient enters the network through?  Compares the
sign of the head's last activation a2 = relu(bn2(.)) (networks.py:24) between the HIP forward and the fp64 oracle at the <= 4 pixels x 32
channels each transition's action pixel interpolates from, and reports the oracle's pre-activation where they differ."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'spatial-intention-maps_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import simq  # noqa: E402
from oracle import cases, fcn as ofcn  # noqa: E402
from oracle import learner as olearner  # noqa: E402
from simq._lib import MODE_TRAIN  # noqa: E402
import test_gpu_fcn as T  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def lerp(o, n_in, n_out):
    real = (n_in - 1) / (n_out - 1) * o
    i0 = min(int(real), n_in - 1)
    return i0, min(i0 + 1, n_in - 1)


for cname in sys.argv[1:] or ['gs_b128_09', 'gs_b128_03']:
    name, cin, cout, B, wseed, dseed = [c for c in cases.GRAD_STUDY_B128_CASES if c[0] == cname][0]
    batch = cases.make_batch(cin, cout, B, dseed)
    x = torch.cat([olearner.apply_transform(s) for s in batch.state])
    policy = T.make_net(simq, cin, cout, wseed, True)
    policy._forward_raw(x.permute(0, 2, 3, 1).contiguous().cuda(), MODE_TRAIN)
    a2_hip = policy.saved_activation('head.a2', B, 'train').cpu()          # [B,48,48,32] post-ReLU
    st = cases.oracle_state(cin, cout, wseed, torch.float64)
    # fp64 oracle up to the pre-activation of the head's second BatchNorm
    pre = {}
    orig_relu = F.relu
    taps = {}
    ofcn.fcn_forward(st, x.double(), True, taps=taps, update_buffers=False)
    a2_64 = taps['head.a2']                                                   # post-ReLU [B,32,48,48]
    # pre-activation: recompute the last BatchNorm input from the oracle's head.a1
    p = ofcn.PREFIX
    h = F.interpolate(taps['head.a1'], scale_factor=2, mode='bilinear', align_corners=True)
    h = F.conv2d(h, st[p + 'conv2.weight'], st[p + 'conv2.bias'])
    mean, var = h.mean(dim=(0, 2, 3)), h.var(dim=(0, 2, 3), unbiased=False)
    pre64 = (h - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + 1e-5) * st[p + 'bn2.weight'][None, :, None, None] + st[p + 'bn2.bias'][None, :, None, None]
    flips, looked = [], 0
    for b, a in enumerate(batch.action):
        co, pix = divmod(int(a), 96 * 96)
        oy, ox = divmod(pix, 96)
        for yy in set(lerp(oy, 48, 96)):
            for xx in set(lerp(ox, 48, 96)):
                m_h = a2_hip[b, yy, xx, :] > 0
                m_o = a2_64[b, :, yy, xx] > 0
                looked += 32
                for ci in torch.nonzero(m_h != m_o).flatten().tolist():
                    flips.append((b, yy, xx, ci, float(pre64[b, ci, yy, xx]), float(a2_hip[b, yy, xx, ci])))
    print('%s: %d mask elements under the one-hot gradient looked at, %d differ between HIP fp32 and the fp64 oracle' % (name, looked, len(flips)))
    for f in flips:
        print('   transition %d pixel (%d,%d) channel %d: oracle pre-activation %.3g, HIP activation %.3g' % f)
    allm = ((a2_hip.permute(0, 3, 1, 2) > 0) != (a2_64 > 0))
    print('   (whole 48x48x32 map: %d of %d mask elements differ)' % (int(allm.sum()), allm.numel()))
