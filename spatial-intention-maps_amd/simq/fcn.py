"""simq.FCN -- drop-in for the reference ``networks.FCN`` (networks.py:6-26) whose
arithmetic runs in libsimq's HIP kernels.

Interface kept from the reference (what train.py / policies.py touch):
  FCN(num_input_channels, num_output_channels)
  net(x[B,Cin,96,96]) -> Tensor[B,Cout,96,96]   with autograd (loss.backward() works)
  .parameters()        72 tensors in reference order (70 receive .grad; fc.* never do)
  .state_dict() / .load_state_dict()   reference keys + OIHW shapes ("module." prefix as
                       produced by the DataParallel wrapper of policies.py:39)
  .train() / .eval()   BatchNorm batch statistics vs running statistics

Device layout: ONE flat fp32 parameter buffer (OHWI conv weights) + identically laid-out
gradient buffer; every nn.Parameter is a view into the flat buffer with the reference's logical
shape (conv weights: [O,I,H,W] as a permuted view of the OHWI storage), so torch.optim.SGD /
clip_grad_norm_ / optimizer.state_dict() operate on it unchanged and interchangeably with the
reference, and the fused learner (simq.learner) can update all parameters with one kernel and
all-reduce one message.
"""
import math
from collections import OrderedDict

import numpy as np
import torch

from . import arch
from ._lib import (KIND_CONV_W, MODE_EVAL, MODE_TRAIN, MODE_TRAIN_NOGRAD, Plan, SimqError, lib, ptr, stream_ptr)

W = arch.STATE_WIDTH


class _FCNFunction(torch.autograd.Function):
    """FCN.forward / its backward as one autograd node over the flat buffers."""

    @staticmethod
    def forward(ctx, net, x_nhwc, *params):
        q = net._forward_raw(x_nhwc, MODE_TRAIN)
        ctx.net = net
        ctx.generation = net._train_generation
        ctx.batch = x_nhwc.shape[0]
        return q

    @staticmethod
    def backward(ctx, dq):
        net = ctx.net
        if ctx.generation != net._train_generation:
            raise SimqError('simq.FCN: backward() after a newer grad-mode forward of the same net -- the saved '
                            'activations were overwritten (one grad-mode forward per backward)')
        g = net._backward_raw(dq.contiguous(), ctx.batch)
        return (None, None) + tuple(net.reference_views(g))


def _reference_view(flat, shape):
    """View of one tensor of a flat (parameter-layout) buffer under the reference's LOGICAL shape: convolution weights are stored
    OHWI and presented as [O, I, H, W] through a permuted (channels-last strided) view, so that everything torch sees -- parameter
    shapes, .grad, torch.optim state, optimizer.state_dict() in a checkpoint -- indexes exactly like the reference's OIHW tensors
    while the kernels keep their K-contiguous rows."""
    v = flat.view(shape)
    return v.permute(0, 3, 1, 2) if len(shape) == 4 else v


class FCN(torch.nn.Module):
    def __init__(self, num_input_channels=3, num_output_channels=1, device=None, dataparallel_keys=True, precision='fp32', options=None):
        super().__init__()
        if device is None:
            device = torch.device('cuda')
        self.device_ = torch.device(device)
        if self.device_.type != 'cuda':
            raise SimqError('simq.FCN needs a GPU device (got %s); there is no CPU path' % device)
        self.num_input_channels, self.num_output_channels = int(num_input_channels), int(num_output_channels)
        self.key_prefix = arch.PREFIX if dataparallel_keys else ''
        self.precision = precision   # 'fp32' (exact fp32 MFMA) | 'bf16x3' (split-bf16, fp32-class) | 'bf16'
        self.plan = Plan(num_input_channels, num_output_channels, precision, options)   # options: simq_plan_options overrides (A/B, diagnostics)
        P = self.plan.param_count
        self.flat_params = torch.zeros(P, dtype=torch.float32, device=self.device_)
        self.flat_grads = torch.zeros(P, dtype=torch.float32, device=self.device_)
        self.bn_buffers = torch.zeros(self.plan.bnbuf_count, dtype=torch.float32, device=self.device_)
        self.num_batches_tracked = OrderedDict((name, 0) for name, _, _ in self.plan.bn_layers)
        self._grad_views = []
        self._param_names = []
        by_name = {name: (off, shape, kind) for name, off, shape, kind in self.plan.tensors}
        # register in the reference's parameter order (72 tensors; resnet18.fc.* exist in the reference,
        # resnet.py:68, but features() never calls them, so they never receive a gradient)
        for key, _, skind in arch.state_spec(self.num_input_channels, self.num_output_channels):
            k = key[len(arch.PREFIX):]
            if skind in arch.TRAINABLE_KINDS:
                off, shape, kind = by_name[k]
                n = int(math.prod(shape))
                pname = k.replace('.', '__')
                self.register_parameter(pname, torch.nn.Parameter(_reference_view(self.flat_params[off:off + n], shape)))
                self._param_names.append((k, pname, kind))
                self._grad_views.append((off, n, tuple(shape)))
            elif skind == 'fc_w':
                self.fc_weight = torch.nn.Parameter(torch.zeros(1000, 512, device=self.device_))
            elif skind == 'fc_b':
                self.fc_bias = torch.nn.Parameter(torch.zeros(1000, device=self.device_))
        self._ws = {}
        self._train_generation = 0
        self._train_input_inplace = False   # the last grad-mode forward was simq_train_step's (minibatch convolved in place, no copy in the workspace)
        self._last_step_event = self._last_step_stream = None   # the learner's last step on this net (as policy OR target), see _order_behind_last_step
        self.step_options = None            # simq.learner.StepOptions of the learner this net is the policy of (None: the defaults)
        self.wcache = torch.empty(max(int(lib.c.simq_wcache_bytes(self.plan.handle)), 16), dtype=torch.uint8, device=self.device_)
        self.weights_dirty = True      # set whenever flat_params changes; the next forward refreshes the weight cache
        self._weights_stamp = 0
        self.reset_parameters()

    # ------------------------------------------------------------------ init / (de)serialisation
    def reset_parameters(self):
        """Same distributions as the reference: kaiming_normal(fan_out) for ResNet convs, BN 1/0
        (resnet.py:70-75); PyTorch defaults for the three head convs and fc (networks.py:10-14)."""
        sd = OrderedDict()
        for key, shape, kind in arch.state_spec(self.num_input_channels, self.num_output_channels):
            k = key[len(arch.PREFIX):]
            in_resnet = k.startswith('resnet18.')
            if kind == 'conv_w':
                if in_resnet:
                    std = math.sqrt(2.0 / (shape[0] * shape[2] * shape[3]))
                    t = torch.randn(shape) * std
                else:
                    bound = 1.0 / math.sqrt(shape[1] * shape[2] * shape[3])
                    t = (torch.rand(shape) * 2 - 1) * bound
            elif kind == 'conv_b':
                cw = [s for kk, s, _ in arch.state_spec(self.num_input_channels, self.num_output_channels)
                      if kk == key.replace('.bias', '.weight')][0]
                bound = 1.0 / math.sqrt(cw[1] * cw[2] * cw[3])
                t = (torch.rand(shape) * 2 - 1) * bound
            elif kind in ('bn_weight', 'bn_var'):
                t = torch.ones(shape)
            elif kind in ('bn_bias', 'bn_mean'):
                t = torch.zeros(shape)
            elif kind == 'bn_count':
                t = torch.tensor(0, dtype=torch.int64)
            elif kind == 'fc_w':
                t = (torch.rand(shape) * 2 - 1) / math.sqrt(512.0)
            elif kind == 'fc_b':
                t = (torch.rand(shape) * 2 - 1) / math.sqrt(512.0)
            sd[key] = t
        self.load_state_dict(sd)

    # ------------------------------------------------------------------ ordering against the learner's own stream
    def _mark_step(self, stream):
        """A learner step that reads or writes this net's buffers was just enqueued on `stream`."""
        ev = torch.cuda.Event()
        ev.record(stream)
        self._last_step_event, self._last_step_stream = ev, stream

    def _order_behind_last_step(self):
        """A learner may issue its steps on a launch stream of its own (simq.learner.LearnerStreams: concurrent robot groups).  Whatever
        touches this net next on ANOTHER stream -- a forward of DQNPolicy.step, state_dict() for a checkpoint, the target sync of
        train.py:267-269 -- is ordered behind that step here, so that the caller keeps the reference's single-stream semantics without a join
        at the end of every loop pass.  Same stream: stream order already holds, nothing is enqueued."""
        ev = self._last_step_event
        if ev is not None:
            cur = torch.cuda.current_stream(self.device_)
            if cur != self._last_step_stream:
                cur.wait_event(ev)

    def state_dict(self, *args, destination=None, prefix='', keep_vars=False):
        """Reference-format state dict (138 keys, OIHW conv weights)."""
        self._order_behind_last_step()
        out = OrderedDict() if destination is None else destination
        pre = prefix + self.key_prefix
        by_name = {name: (off, shape, kind) for name, off, shape, kind in self.plan.tensors}
        bn_by_name = {name: (off, ch) for name, off, ch in self.plan.bn_layers}
        for key, shape, kind in arch.state_spec(self.num_input_channels, self.num_output_channels):
            k = key[len(arch.PREFIX):]
            if kind in arch.TRAINABLE_KINDS:
                off, dshape, dkind = by_name[k]
                n = int(math.prod(dshape))
                t = self.flat_params[off:off + n].view(dshape)
                if dkind == KIND_CONV_W:
                    t = t.permute(0, 3, 1, 2).contiguous()       # OHWI -> OIHW
                else:
                    t = t.clone()
            elif kind in ('bn_mean', 'bn_var'):
                bn = k.rsplit('.', 1)[0]
                off, ch = bn_by_name[bn]
                o = off + (ch if kind == 'bn_var' else 0)
                t = self.bn_buffers[o:o + ch].clone()
            elif kind == 'bn_count':
                t = torch.tensor(self.num_batches_tracked[k.rsplit('.', 1)[0]], dtype=torch.int64, device=self.device_)
            elif kind == 'fc_w':
                t = self.fc_weight.detach().clone()
            else:
                t = self.fc_bias.detach().clone()
            out[pre + k] = t.detach()
        return out

    def load_state_dict(self, state_dict, strict=True):
        """Accepts reference-format dicts with or without the DataParallel 'module.' prefix."""
        self._order_behind_last_step()
        sd = {}
        for k, v in state_dict.items():
            sd[k[len(arch.PREFIX):] if k.startswith(arch.PREFIX) else k] = v
        by_name = {name: (off, shape, kind) for name, off, shape, kind in self.plan.tensors}
        bn_by_name = {name: (off, ch) for name, off, ch in self.plan.bn_layers}
        missing = []
        with torch.no_grad():
            for key, shape, kind in arch.state_spec(self.num_input_channels, self.num_output_channels):
                k = key[len(arch.PREFIX):]
                if k not in sd:
                    missing.append(k)
                    continue
                v = torch.as_tensor(sd[k])
                if kind != 'bn_count' and tuple(v.shape) != tuple(shape):
                    raise SimqError('load_state_dict: %s has shape %s, expected %s' % (k, tuple(v.shape), tuple(shape)))
                if kind in arch.TRAINABLE_KINDS:
                    off, dshape, dkind = by_name[k]
                    n = int(math.prod(dshape))
                    v = v.to(torch.float32)
                    if dkind == KIND_CONV_W:
                        v = v.permute(0, 2, 3, 1)                  # OIHW -> OHWI
                    self.flat_params[off:off + n].copy_(v.reshape(-1).to(self.device_))
                elif kind in ('bn_mean', 'bn_var'):
                    off, ch = bn_by_name[k.rsplit('.', 1)[0]]
                    o = off + (ch if kind == 'bn_var' else 0)
                    self.bn_buffers[o:o + ch].copy_(v.to(torch.float32).to(self.device_))
                elif kind == 'bn_count':
                    self.num_batches_tracked[k.rsplit('.', 1)[0]] = int(v)
                elif kind == 'fc_w':
                    self.fc_weight.copy_(v.to(self.device_))
                else:
                    self.fc_bias.copy_(v.to(self.device_))
        self.weights_dirty = True
        if strict and missing:
            raise SimqError('load_state_dict: missing keys %s' % missing[:5])
        return torch.nn.modules.module._IncompatibleKeys(missing, [])

    def copy_state_from(self, other):
        """target.load_state_dict(policy.state_dict()) (train.py:214,269) without the OIHW round trip."""
        self._order_behind_last_step()
        other._order_behind_last_step()
        with torch.no_grad():
            self.flat_params.copy_(other.flat_params)
            self.bn_buffers.copy_(other.bn_buffers)
            self.fc_weight.copy_(other.fc_weight)
            self.fc_bias.copy_(other.fc_bias)
        self.num_batches_tracked = OrderedDict(other.num_batches_tracked)
        self.weights_dirty = True

    def reference_views(self, flat):
        """Per-parameter views (reference order and logical OIHW shapes) of a flat buffer laid out like flat_params."""
        return [_reference_view(flat[off:off + n], shape) for (off, n, shape) in self._grad_views]

    # ------------------------------------------------------------------ raw kernels over flat buffers
    def _workspace(self, slot, batch):
        # 'train' is the only slot a backward pass reads; every other slot (no-grad / eval forwards) does without the weight-gradient slabs
        need = self.plan.workspace_bytes(batch, forward_only=(slot != 'train'))
        ws = self._ws.get(slot)
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.uint8, device=self.device_)
            self._ws[slot] = ws
        return ws

    def _ensure_weights(self):
        if self.weights_dirty:
            self._order_behind_last_step()
            lib.call('simq_weights_prepare', self.plan.handle, ptr(self.flat_params), ptr(self.wcache), stream_ptr(self.device_))
            self.weights_dirty = False
            self._weights_stamp += 1
            # (a reader on another stream -- the early target-net forward of simq.learner -- orders itself behind the parameters' last change)
            self._weights_event = torch.cuda.Event()
            self._weights_event.record(torch.cuda.current_stream(self.device_))

    def _forward_raw(self, x_nhwc, mode, sync=None):
        """x_nhwc [B,96,96,Cin] fp32 contiguous on device -> q [B,Cout,96,96].  sync: a simq.dist.SyncBN (global-minibatch
        BatchNorm statistics in the train modes)."""
        if x_nhwc.dtype != torch.float32 or not x_nhwc.is_contiguous() or x_nhwc.device != self.flat_params.device:
            raise SimqError('simq.FCN: input must be a contiguous fp32 tensor on %s' % self.device_)
        B = x_nhwc.shape[0]
        if tuple(x_nhwc.shape[1:]) != (W, W, self.num_input_channels):
            raise SimqError('simq.FCN: expected input [B,%d,%d,%d] (NHWC), got %s'
                            % (W, W, self.num_input_channels, tuple(x_nhwc.shape)))
        if B < 1:
            raise SimqError('simq.FCN: empty batch')
        self._order_behind_last_step()
        if mode == MODE_TRAIN:
            self._train_generation += 1
            self._train_input_inplace = False      # this forward keeps its copy of the input in the workspace
        ws = self._workspace('train' if mode == MODE_TRAIN else 'tmp', B)
        q = torch.empty((B, self.num_output_channels, W, W), dtype=torch.float32, device=self.device_)
        self._ensure_weights()
        if sync is not None:
            lib.call('simq_forward_sync', self.plan.handle, mode, B, ptr(self.flat_params), ptr(self.wcache), ptr(self.bn_buffers),
                     ptr(x_nhwc), ptr(q), ptr(ws), stream_ptr(self.device_), sync.bind(ws))
        else:
            lib.call('simq_forward', self.plan.handle, mode, B, ptr(self.flat_params), ptr(self.wcache), ptr(self.bn_buffers), ptr(x_nhwc),
                     ptr(q), ptr(ws), stream_ptr(self.device_))
        if mode != MODE_EVAL:
            for k in self.num_batches_tracked:
                self.num_batches_tracked[k] += 1
        return q

    def _train_workspace_for_backward(self):
        ws = self._ws.get('train')
        if ws is None:
            raise SimqError('simq.FCN: backward without a grad-mode forward')
        if self._train_input_inplace:
            raise SimqError('simq.FCN: the last grad-mode forward ran inside simq.train (simq_train_step convolves the minibatch in place and keeps no '
                            'copy in the workspace): a backward pass called on its own would differentiate the first convolution against a stale '
                            'input -- run a grad-mode forward first')
        self._order_behind_last_step()
        self._adopt_side_stream()
        return ws

    def _adopt_side_stream(self):
        """The side stream of a backward pass called on its own: one the host has TESTED not to share the launch stream's hardware queue
        (simq.learner.LearnerStreams), handed to the plan (simq_plan_adopt_side_stream) -- once per (plan, launch stream)."""
        from . import learner
        cur = torch.cuda.current_stream(self.device_)
        key = (self.plan.handle.value if hasattr(self.plan.handle, 'value') else int(self.plan.handle), cur.cuda_stream)
        if getattr(self, '_adopted_side', None) != key:
            ls = learner.learner_streams(self)
            if ls.launch is None or ls.launch.cuda_stream == cur.cuda_stream:
                ls.bind(cur)
                lib.call('simq_plan_adopt_side_stream', self.plan.handle, ls.side.cuda_stream)
            self._adopted_side = key

    def _backward_raw(self, dq, batch, phase=0):
        """dq [B,Cout,96,96] -> flat gradient buffer (overwritten).  phase 1 / 2: the two halves of the walk
        (head + layer4, then the rest) for callers that overlap the gradient all-reduce with phase 2."""
        ws = self._train_workspace_for_backward()
        lib.call('simq_backward_phase', self.plan.handle, batch, ptr(self.flat_params), ptr(self.wcache), ptr(dq),
                 ptr(self.flat_grads), ptr(ws), phase, stream_ptr(self.device_))
        return self.flat_grads

    def _backward_onehot(self, action, q_sa, y, grad_scale, batch, phase=0, sync=None):
        """Backward of the TD loss from its one-hot upstream gradient dQ[b][action[b]] = clamp(q_sa - y, -1, 1) * grad_scale
        (no dense dQ map); same phases as _backward_raw.  sync: as in _forward_raw."""
        ws = self._train_workspace_for_backward()
        if sync is not None:
            lib.call('simq_backward_sync', self.plan.handle, batch, ptr(self.flat_params), ptr(self.wcache), None, ptr(action), ptr(q_sa),
                     ptr(y), float(grad_scale), ptr(self.flat_grads), ptr(ws), phase, stream_ptr(self.device_), sync.bind(ws))
            return self.flat_grads
        lib.call('simq_backward_onehot', self.plan.handle, batch, ptr(self.flat_params), ptr(self.wcache), ptr(action), ptr(q_sa),
                 ptr(y), float(grad_scale), ptr(self.flat_grads), ptr(ws), phase, stream_ptr(self.device_))
        return self.flat_grads

    @property
    def grad_bucket_split(self):
        """flat_grads[split:] (head + layer4, 75 % of the bytes) is final after backward phase 1."""
        return int(lib.c.simq_grad_bucket_split(self.plan.handle))

    def to_nhwc(self, x_nchw):
        x = x_nchw.to(self.device_, torch.float32).contiguous()
        out = torch.empty((x.shape[0], W, W, x.shape[1]), dtype=torch.float32, device=self.device_)
        lib.call('simq_nchw_to_nhwc', ptr(x), ptr(out), x.shape[0], x.shape[1], W * W, stream_ptr(self.device_))
        return out

    # ------------------------------------------------------------------ nn.Module surface
    def forward(self, x):
        """x: [B,Cin,96,96] (reference layout, networks.py:16)."""
        if x.dim() != 4 or x.shape[1] != self.num_input_channels or x.shape[2] != W or x.shape[3] != W:
            raise SimqError('simq.FCN: expected input [B,%d,%d,%d], got %s' % (self.num_input_channels, W, W, tuple(x.shape)))
        return self.forward_nhwc(self.to_nhwc(x))

    def forward_nhwc(self, x_nhwc):
        """Same as forward() for inputs already in the replay/HWC layout [B,96,96,Cin]."""
        # public entry point: an external optimiser (torch.optim.SGD over .parameters()) may have stepped the flat
        # buffer since the last call, so the derived weight cache is refreshed every time (~0.1 ms); the fused
        # learner (simq.train) calls _forward_raw and tracks the dirty flag itself
        self.weights_dirty = True
        if not self.training:
            return self._forward_raw(x_nhwc, MODE_EVAL)
        if not torch.is_grad_enabled():
            return self._forward_raw(x_nhwc, MODE_TRAIN_NOGRAD)
        params = [getattr(self, pname) for _, pname, _ in self._param_names]
        return _FCNFunction.apply(self, x_nhwc, *params)

    def argmax(self, q):
        """Flat first-index argmax of one Q-map [Cout,96,96] (policies.py:64: o.view(1,-1).max(1)[1].item())."""
        q = q.contiguous()
        idx = torch.empty(1, dtype=torch.int64, device=self.device_)
        lib.call('simq_q_argmax', ptr(q), 1, q.numel(), ptr(idx), None, stream_ptr(self.device_))
        return int(idx.item())

    def saved_activation(self, name, batch, slot='tmp'):
        """NHWC view [B,H,W,C] of an activation the last forward left in its workspace
        ('stem.pool', 'layer<1-4>.<0-1>', 'head.a1', 'head.a2') -- parity bisecting aid."""
        import ctypes
        if name == 'head.a2' and slot != 'train':
            raise SimqError("saved_activation: 'head.a2' is written by grad-mode forwards only (slot 'train'); the no-grad and the folded "
                            "eval forward never store it (include/simq.h simq_workspace_tensor)")
        off, n, ch = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int()
        lib.call('simq_workspace_tensor', self.plan.handle, batch, name.encode(), ctypes.byref(off), ctypes.byref(n), ctypes.byref(ch))
        ws = self._ws[slot]
        hw = int(round((n.value // (batch * ch.value)) ** 0.5))
        return ws[off.value:off.value + 4 * n.value].view(torch.float32).view(batch, hw, hw, ch.value)

    @staticmethod
    def _typed_view(buf, off, n, ch, st, batch):
        """(storage code of simq_workspace_tensor_ex: 0 fp32, 1 bf16, 2 fp64, 3 uint8) -> the tensor in its natural shape."""
        if st == 2:
            return buf[off:off + 8 * n].view(torch.float64).view(2, ch)
        if st == 3:
            t = buf[off:off + n]
        elif st:
            t = buf[off:off + 2 * n].view(torch.bfloat16)
        else:
            t = buf[off:off + 4 * n].view(torch.float32)
        if n == 4 * ch:
            return t.view(4, ch)
        hw = int(round((n // (batch * ch)) ** 0.5))
        return t.view(batch, hw, hw, ch)

    def stored_tensor(self, name, batch, slot='train'):
        """A tensor of the last forward in workspace `slot` (simq_workspace_tensor_ex: 'layer<l>.<b>.<y1|a1|y2|yd|out>' as [B,24,24,C] fp32 or
        bf16, 'layer<l>.<b>.<bn1|bn2|bnd>' as [4,C] = scale | shift | mean | invstd, 'layer<l>.<b>.<red1|red2|redd>' as [2,C] fp64,
        'stem.pool.plane', and -- round 6 -- the stem's and the head's: 'stem.y0', 'stem.bn', 'stem.idx', 'head.y1', 'head.bn1', 'head.a1.plane',
        'head.z2', 'head.y2', 'head.bn2', 'head.z3', '<stem|head>.red*') -- teacher-forced tests."""
        import ctypes
        off, n, ch, st = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int(), ctypes.c_int()
        lib.call('simq_workspace_tensor_ex', self.plan.handle, batch, name.encode(), ctypes.byref(off), ctypes.byref(n), ctypes.byref(ch), ctypes.byref(st))
        return self._typed_view(self._ws[slot], off.value, n.value, ch.value, st.value, batch)

    def backward_traced(self, dq, batch):
        """simq_backward_traced: the backward pass of the last grad-mode forward with every gradient tensor of the walk kept (teacher-forced
        tests).  Returns (flat gradient buffer, lookup) with lookup('layer<l>.<b>.<g_out|dy2|dz|dyd|da1|dy1|g_in>' | 'head.<da2|dy2|dz2|da1|dy1>'
        | 'stem.<dz|dy0>') -> NHWC tensor in the plan's storage type."""
        import ctypes
        ws = self._train_workspace_for_backward()
        trace = torch.empty(int(lib.c.simq_backward_trace_bytes(self.plan.handle, batch)), dtype=torch.uint8, device=self.device_)
        lib.call('simq_backward_traced', self.plan.handle, batch, ptr(self.flat_params), ptr(self.wcache), ptr(dq.contiguous()), ptr(self.flat_grads),
                 ptr(ws), ptr(trace), stream_ptr(self.device_))

        def lookup(name):
            off, n, ch, st = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int(), ctypes.c_int()
            lib.call('simq_backward_trace_tensor', self.plan.handle, batch, name.encode(), ctypes.byref(off), ctypes.byref(n), ctypes.byref(ch), ctypes.byref(st))
            return self._typed_view(trace, off.value, n.value, ch.value, st.value, batch)
        return self.flat_grads, lookup

    def infer_argmax_batch(self, states, need_q=False):
        """Eval-mode forward of several HWC states (numpy [96,96,C] or device tensors [1,96,96,C]) in ONE batch + one
        argmax launch: the multi-robot form of infer_argmax (batched multi-env inference)."""
        if self.training:
            raise SimqError('infer_argmax_batch: the net must be in eval mode (policies.py:56)')
        if len(states) == 1:
            a, q = self.infer_argmax(states[0], need_q)
            return [a], [q]
        if all(not torch.is_tensor(s) for s in states):
            x = torch.from_numpy(np.stack([np.ascontiguousarray(s, dtype=np.float32) for s in states])).to(self.device_)
        else:
            x = torch.cat([s if torch.is_tensor(s) else torch.from_numpy(np.ascontiguousarray(s, dtype=np.float32)).unsqueeze(0).to(self.device_)
                           for s in states]).contiguous()
        q = self.forward_nhwc(x)
        n = q[0].numel()
        idx = torch.empty(len(states), dtype=torch.int64, device=self.device_)
        lib.call('simq_q_argmax', ptr(q), len(states), n, ptr(idx), None, stream_ptr(self.device_))
        acts = idx.tolist()
        qs = list(q.cpu().numpy()) if need_q else [None] * len(states)
        return acts, qs

    # ------------------------------------------------------------------ batch-1 inference (DQNPolicy.step hot loop)
    def infer_argmax(self, state_hwc, need_q=False):
        """Eval-mode forward of ONE HWC state + flat first-index argmax (policies.py:59-64: apply_transform -> net ->
        view(1,-1).max(1)[1].item()).  Only the 8-byte index comes back; the Q-map is copied to the host only when
        `need_q` (the reference's debug output, policies.py:66).  Measured alternatives that were SLOWER on this
        runtime and are therefore not used: a hipGraph capture of the ~75 nodes (5.7 ms per replay vs 0.6 ms of eager
        launches) and staging the state through a pinned host buffer (8 ms; CPU writes to pinned memory are slow)."""
        if self.training:
            raise SimqError('infer_argmax: the net must be in eval mode (policies.py:56)')
        # a device tensor [1,96,96,C] is taken as is (DQNIntentionPolicy hands over state + predicted map in HBM)
        x = state_hwc if torch.is_tensor(state_hwc) else \
            torch.from_numpy(np.ascontiguousarray(state_hwc, dtype=np.float32)).unsqueeze(0).to(self.device_)
        q = self.forward_nhwc(x)
        a = self.argmax(q[0])
        return (a, q[0].cpu().numpy()) if need_q else (a, None)
