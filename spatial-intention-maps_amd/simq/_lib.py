"""ctypes binding of include/simq.h (libsimq.so).

No fallback: if the shared library is missing or does not load, importing this
module raises -- the product path never routes around the HIP kernels.
torch is imported first so that libsimq.so binds to the HIP runtime
(libamdhip64.so.7) already loaded by PyTorch-ROCm and both share one device
context / stream table.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int64, c_void_p

import torch  # noqa: F401  (must precede the CDLL: shares the HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
# The product library.  tools/ may name the ablation build (libsimq_ablate.so, `make -C csrc ablate`: kernel-selection switches from
# SIMQ_* environment variables, timing-ablation kernels) through SIMQ_LIBRARY; `build_flags` says which one is loaded and bench.py /
# the parity tests refuse anything but the product.  Neither build reads its ARITHMETIC from the environment (simq_plan_options).
LIB_PATH = os.environ.get('SIMQ_LIBRARY') or os.path.join(_HERE, 'libsimq.so')

MODE_EVAL, MODE_TRAIN, MODE_TRAIN_NOGRAD = 0, 1, 2
KIND_CONV_W, KIND_CONV_B, KIND_BN_W, KIND_BN_B = 0, 1, 2, 3
PRECISIONS = {'fp32': 0, 'bf16x3': 1, 'bf16': 2}
COMM_ID_BYTES, COMM_F32, COMM_F64 = 128, 0, 1


class SimqError(RuntimeError):
    pass


if not os.path.exists(LIB_PATH):
    raise SimqError('libsimq.so not found at %s -- build it with `python -c "import __graft_entry__ as g; g.build()"` '
                    '(or `make -C spatial-intention-maps_amd/csrc`); there is no CPU fallback' % LIB_PATH)

_c = ctypes.CDLL(LIB_PATH)

REDUCE_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_int64, c_void_p)     # simq_reduce_fn


class SyncArgs(ctypes.Structure):
    """simq_sync of include/simq.h (cross-rank BatchNorm statistics)."""
    _fields_ = [('reduce', REDUCE_FN), ('user', c_void_p), ('global_batch', c_int), ('world_size', c_int)]


class PlanOptions(ctypes.Structure):
    """simq_plan_options of include/simq.h: the algebraic forms / storage precisions / fusions of a plan (fixed at creation)."""
    _fields_ = [(n, c_int) for n in (
        'struct_bytes', 'winograd', 'winograd_min_cc', 'winograd_f4_forward', 'winograd_f4_min_tiles', 'winograd_f4_grad',
        'winograd_f4_fwd_grad_min_cc', 'winograd_wgrad', 'winograd_wgrad_f4', 'stem_bf16', 'bf16_act_grads', 'keep_fp32_activations', 'fold_eval_bn_bf16',
        'fuse_bn_backward_sums', 'fuse_stem_backward_sums', 'fuse_bn1_apply', 'deterministic', 'bn1_mask_from_preact',
        'wgrad_ksplit', 'fwd_overlap', 'wgrad_overlap', 'plane_xcd', 'wgrad_xcd_group', 'tail_split', 'early_target_after_block', 'gemm_split')]


class LaunchOpts(ctypes.Structure):
    """simq_launch_opts of include/simq.h: kernel selection / block scheduling of ONE standalone operator call (per-kernel tests, tools/)."""
    _fields_ = [(n, c_int) for n in ('struct_bytes', 'force_bm', 'force_bn', 'tail_split', 'plane_xcd', 'wgrad_xcd_group', 'wgrad_ksplit', 'gemm_split')]


class TrainArgs(ctypes.Structure):
    """simq_train_args of include/simq.h (the whole TD step in one library call)."""
    _fields_ = [('plan', c_void_p),
                ('batch', c_int), ('num_nonfinal', c_int), ('global_batch', c_int), ('use_double_dqn', c_int), ('first_step', c_int),
                ('sync_bn', c_int),
                ('gamma', c_float), ('lr', c_float), ('momentum', c_float), ('weight_decay', c_float), ('max_norm', c_float),
                ('reserved2_', c_float)] + [(n, c_void_p) for n in (
                    'params', 'wcache', 'bnbuf', 'grads', 'momentum_buf', 'ws_train', 'ws_tmp',
                    't_params', 't_wcache', 't_bnbuf', 't_ws',
                    'state', 'next_state', 'action', 'reward', 'nonfinal_pos',
                    'q', 'q_next', 'q_tgt', 'dq', 'nsv', 'vals', 'best', 'q_sa', 'y', 'td', 'out4',
                    'opt_scratch', 'total_norm', 'stream', 'side_stream')] + [('global_nonfinal', c_int), ('struct_bytes', c_int), ('comm', c_void_p), ('loss_host', c_void_p), ('target_stream', c_void_p), ('third_stream', c_void_p)]


_SIGS = {
    'simq_version': (c_int, []),
    'simq_build_flags': (c_int, []),
    'simq_last_error': (c_char_p, []),
    'simq_plan_create': (c_int, [c_int, c_int, POINTER(c_void_p)]),
    'simq_plan_create_ex': (c_int, [c_int, c_int, c_int, POINTER(c_void_p)]),
    'simq_plan_options_default': (None, [c_void_p]),
    'simq_launch_opts_default': (None, [c_void_p]),
    'simq_plan_create_opts': (c_int, [c_int, c_int, c_int, c_void_p, POINTER(c_void_p)]),
    'simq_plan_get_options': (c_int, [c_void_p, c_void_p]),
    'simq_plan_precision': (c_int, [c_void_p]),
    'simq_plan_destroy': (None, [c_void_p]),
    'simq_param_count': (c_int64, [c_void_p]),
    'simq_param_num_tensors': (c_int, [c_void_p]),
    'simq_param_tensor_info': (c_int, [c_void_p, c_int, c_char_p, c_int, POINTER(c_int64), POINTER(c_int64 * 4), POINTER(c_int)]),
    'simq_bnbuf_count': (c_int64, [c_void_p]),
    'simq_bn_num_layers': (c_int, [c_void_p]),
    'simq_bn_layer_info': (c_int, [c_void_p, c_int, c_char_p, c_int, POINTER(c_int64), POINTER(c_int)]),
    'simq_workspace_bytes': (c_int64, [c_void_p, c_int]),
    'simq_workspace_bytes_forward': (c_int64, [c_void_p, c_int]),
    'simq_workspace_tensor': (c_int, [c_void_p, c_int, c_char_p, POINTER(c_int64), POINTER(c_int64), POINTER(c_int)]),
    'simq_workspace_tensor_ex': (c_int, [c_void_p, c_int, c_char_p, POINTER(c_int64), POINTER(c_int64), POINTER(c_int), POINTER(c_int)]),
    'simq_wcache_bytes': (c_int64, [c_void_p]),
    'simq_weights_prepare': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    'simq_forward': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'simq_backward': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'simq_backward_phase': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    'simq_grad_bucket_split': (c_int64, [c_void_p]),
    'simq_backward_trace_bytes': (c_int64, [c_void_p, c_int]),
    'simq_backward_trace_tensor': (c_int, [c_void_p, c_int, c_char_p, POINTER(c_int64), POINTER(c_int64), POINTER(c_int), POINTER(c_int)]),
    'simq_backward_traced': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'simq_train_step': (c_int, [c_void_p]),
    'simq_train_loss_wait': (c_int, [c_void_p]),
    'simq_plan_adopt_side_stream': (c_int, [c_void_p, c_void_p]),
    'simq_backward_onehot': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_void_p]),
    'simq_q_argmax': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'simq_q_gather': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'simq_scatter_next_values': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    'simq_td_huber': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_float, c_float,
                              c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'simq_clip_sgd_step': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_int,
                                   c_void_p, c_void_p, c_void_p]),
    'simq_bce_with_logits': (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    'simq_split_last_channel': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    'simq_sigmoid_concat': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    'simq_replay_gather': (c_int, [c_void_p, c_int64, c_void_p, c_int, c_void_p, c_void_p]),
    'simq_nchw_to_nhwc': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'simq_nhwc_to_nchw': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'simq_conv2d_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 9 + [c_void_p, c_void_p] + [c_void_p]),
    'simq_gemm_f32_batched': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'simq_conv2d_dgrad': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 8 + [c_void_p] + [c_void_p]),
    'simq_conv2d_wgrad': (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 9 + [c_void_p] + [c_void_p]),
    'simq_conv2d_fwd_bf16': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 10 + [c_void_p, c_void_p, c_void_p] + [c_void_p]),
    'simq_conv2d_wgrad_bf16': (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 10 + [c_void_p, c_void_p] + [c_void_p]),
    'simq_conv2d_wgrad_bf16_slab_bytes': (c_int64, []),
    'simq_conv2d_wgrad_bf16_slab': (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 10 + [c_void_p, c_void_p, c_void_p] + [c_void_p]),
    'simq_upsample2x_fwd': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'simq_upsample2x_bwd': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'simq_profile_start': (c_int, []),
    'simq_profile_stop': (c_int, [c_void_p, c_int]),
    'simq_launch_counts_reset': (c_int, []),
    'simq_launch_count': (c_int64, [c_char_p]),
    'simq_launch_counts': (c_int, [c_char_p, c_int]),
    'simq_conv2d_wgrad_winograd': (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 5 + [c_void_p, c_void_p] + [c_void_p]),
    'simq_conv2d_fwd_winograd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 5 + [c_void_p, c_void_p, c_void_p] + [c_void_p]),
    'simq_conv2d_fwd_winograd4': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 5 + [c_void_p, c_void_p, c_void_p] + [c_void_p]),
    'simq_bn_relu_apply': (c_int, [c_void_p] * 5 + [c_int, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'simq_bn_relu_backward': (c_int, [c_void_p, c_void_p, c_int] + [c_void_p] * 8 + [c_int64, c_int, c_int, c_void_p]),
    'simq_conv2d_fwd_bnrelu_in': (c_int, [c_void_p] * 6 + [c_int] * 10 + [c_void_p, c_void_p] + [c_void_p]),
    'simq_conv2d_wgrad_bnrelu_in': (c_int, [c_void_p] * 5 + [c_int] * 10 + [c_void_p, c_void_p] + [c_void_p]),
    'simq_conv2d_fwd_stem_f32': (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 4 + [c_void_p, c_void_p]),
    'simq_conv2d_fwd_stem_bf16': (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 4 + [c_void_p, c_void_p, c_void_p]),
    'simq_conv2d_wgrad_stem_bf16': (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 4 + [c_void_p, c_void_p]),
    'simq_forward_sync': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'simq_backward_sync': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                   c_int, c_void_p, c_void_p]),
    'simq_comm_reduce_f64': (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    'simq_forward_sync_null': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'simq_comm_unique_id': (c_int, [c_void_p]),
    'simq_comm_init': (c_int, [c_void_p, c_int, c_int, POINTER(c_void_p)]),
    'simq_comm_adopt_stream': (c_int, [c_void_p, c_void_p]),
    'simq_comm_world_size': (c_int, [c_void_p]),
    'simq_comm_rank': (c_int, [c_void_p]),
    'simq_comm_allreduce': (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    'simq_comm_broadcast': (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    'simq_comm_wait': (c_int, [c_void_p, c_void_p]),
    'simq_comm_progress': (c_int, [c_void_p, c_void_p]),
    'simq_comm_time_waits': (c_int, [c_void_p, c_int]),
    'simq_comm_last_wait_ms': (c_int, [c_void_p, POINTER(c_float)]),
    'simq_comm_destroy': (c_int, [c_void_p]),
}

EXPORTS = tuple(_SIGS)
# operators whose last argument is `const simq_launch_opts*`
OPTS_FUNCS = frozenset(('simq_gemm_f32_batched', 'simq_conv2d_fwd', 'simq_conv2d_dgrad', 'simq_conv2d_wgrad', 'simq_conv2d_fwd_bf16', 'simq_conv2d_wgrad_bf16',
                        'simq_conv2d_wgrad_bf16_slab', 'simq_conv2d_wgrad_winograd', 'simq_conv2d_fwd_winograd', 'simq_conv2d_fwd_winograd4',
                        'simq_conv2d_fwd_bnrelu_in', 'simq_conv2d_wgrad_bnrelu_in'))


def launch_counts():
    """{kernel family: launches since the last simq_launch_counts_reset} (include/simq.h: the launch log)."""
    buf = ctypes.create_string_buffer(4096)
    check(_c.simq_launch_counts(buf, 4096), 'simq_launch_counts')
    return {kv.split('=')[0]: int(kv.split('=')[1]) for kv in buf.value.decode().split(';') if kv}


def launch_opts(tile=None, **fields):
    """A simq_launch_opts with the library's defaults; tile=(bm, bn) forces a block tile, other fields by name (plane_xcd=0, ...)."""
    o = LaunchOpts()
    _c.simq_launch_opts_default(ctypes.byref(o))
    if tile is not None and tile[0] > 0:
        o.force_bm, o.force_bn = int(tile[0]), int(tile[1])
    for k, v in fields.items():
        if k == 'struct_bytes' or not hasattr(o, k):
            raise SimqError('unknown launch option %r (fields of simq_launch_opts: %s)' % (k, ', '.join(n for n, _ in LaunchOpts._fields_[1:])))
        setattr(o, k, int(v))
    return o

for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(_c, _name)      # AttributeError here == header / library mismatch
    _fn.restype = _res
    _fn.argtypes = _args


def last_error():
    return (_c.simq_last_error() or b'').decode()


def check(rc, what):
    if rc != 0:
        raise SimqError('%s failed (%d): %s' % (what, rc, last_error()))


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else c_void_p(t.data_ptr())


def stream_ptr(device=None):
    """hipStream_t of torch's current stream on `device`."""
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


class Lib:
    """Thin checked wrappers; all pointers come from torch tensors."""
    c = _c
    version = _c.simq_version()
    build_flags = _c.simq_build_flags()          # 0: the product; 1: the ablation build (tools only)
    path = LIB_PATH


    @staticmethod
    def call(name, *args, opts=None):
        """Checked call.  The convolution operators of include/simq.h end in `const simq_launch_opts* opts`: pass opts=launch_opts(...) to
        force a tile / a scheduling variant for THAT call; omitted = NULL = what a default plan launches."""
        if name in OPTS_FUNCS:
            args = args + (ctypes.byref(opts) if opts is not None else None,)
        elif opts is not None:
            raise SimqError('%s takes no simq_launch_opts' % name)
        check(getattr(_c, name)(*args), name)


lib = Lib()

if Lib.build_flags != 0:
    # the ablation build's KERNEL SELECTION follows SIMQ_* environment switches: fine for tools/, never silently for a training run
    import warnings
    warnings.warn('simq: SIMQ_LIBRARY selected the ablation build %s (simq_build_flags = %d): kernel selection follows SIMQ_* environment '
                  'variables; bench.py and the parity tests refuse it -- unset SIMQ_LIBRARY for anything but tools/' % (LIB_PATH, Lib.build_flags),
                  RuntimeWarning, stacklevel=2)


# in-process default overrides for every Plan created afterwards (tools/ and diagnostics set entries here; never read from the
# environment: a plan's arithmetic must not depend on who exported what)
DEFAULT_PLAN_OPTIONS = {}


class Plan:
    """Owns a simq_plan* and exposes its buffer layout."""

    def __init__(self, num_input_channels, num_output_channels, precision='fp32', options=None):
        """options: {simq_plan_options field: int} overriding the defaults (include/simq.h) -- A/B measurements and diagnostics."""
        self.cin, self.cout = int(num_input_channels), int(num_output_channels)
        if precision not in PRECISIONS:
            raise SimqError('unknown precision %r (choose from %s)' % (precision, sorted(PRECISIONS)))
        self.precision = precision
        o = PlanOptions()
        _c.simq_plan_options_default(ctypes.byref(o))
        for k, v in dict(DEFAULT_PLAN_OPTIONS, **(options or {})).items():
            if k == 'struct_bytes' or not hasattr(o, k):
                raise SimqError('unknown plan option %r (fields of simq_plan_options: %s)' % (k, ', '.join(n for n, _ in PlanOptions._fields_[1:])))
            setattr(o, k, int(v))
        h = c_void_p()
        check(_c.simq_plan_create_opts(self.cin, self.cout, PRECISIONS[precision], ctypes.byref(o), ctypes.byref(h)), 'simq_plan_create_opts')
        self.handle = h
        got = PlanOptions()
        check(_c.simq_plan_get_options(h, ctypes.byref(got)), 'simq_plan_get_options')
        self.options = {n: getattr(got, n) for n, _ in PlanOptions._fields_[1:]}
        self.param_count = _c.simq_param_count(h)
        self.bnbuf_count = _c.simq_bnbuf_count(h)
        self.tensors = []     # (name, offset, shape(list), kind)
        buf = ctypes.create_string_buffer(128)
        for i in range(_c.simq_param_num_tensors(h)):
            off, shape, kind = c_int64(), (c_int64 * 4)(), c_int()
            check(_c.simq_param_tensor_info(h, i, buf, 128, ctypes.byref(off), ctypes.byref(shape), ctypes.byref(kind)),
                  'simq_param_tensor_info')
            shp = list(shape) if kind.value == KIND_CONV_W else [shape[0]]
            self.tensors.append((buf.value.decode(), off.value, shp, kind.value))
        self.bn_layers = []   # (name, offset, channels)
        for i in range(_c.simq_bn_num_layers(h)):
            off, ch = c_int64(), c_int()
            check(_c.simq_bn_layer_info(h, i, buf, 128, ctypes.byref(off), ctypes.byref(ch)), 'simq_bn_layer_info')
            self.bn_layers.append((buf.value.decode(), off.value, ch.value))

    def workspace_bytes(self, batch, forward_only=False):
        """Bytes of a workspace for `batch` samples; forward_only: one that never serves a backward pass (no weight-gradient slabs)."""
        n = (_c.simq_workspace_bytes_forward if forward_only else _c.simq_workspace_bytes)(self.handle, int(batch))
        if n < 0:
            raise SimqError('simq_workspace_bytes(%d) failed' % batch)
        return n

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                _c.simq_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass
