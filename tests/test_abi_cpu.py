"""CPU: libsimq.so loads and exports every symbol include/simq.h declares; the plan layout agrees
with the reference state_dict (no kernel is launched here -- there is no GPU in this container)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def L():
    import __graft_entry__ as ge
    ge.build()                       # idempotent (make): the extension must exist, there is no fallback
    from simq import _lib
    return _lib


def header_symbols():
    text = open(os.path.join(ROOT, 'include', 'simq.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(simq_[a-z0-9_]+)\s*\(', text)))


def test_every_declared_symbol_is_exported(L):
    syms = header_symbols()
    assert len(syms) >= 25
    raw = ctypes.CDLL(L.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), 'libsimq.so does not export %s' % s
    assert sorted(L.EXPORTS) == syms, 'ctypes binding and header disagree'
    assert L.lib.version == 500


def test_no_process_global_behaviour_switch_is_exported(L):
    """Round 5: what a plan computes and how it schedules its launches is a property of the plan (simq_plan_options); a standalone
    operator call carries its own simq_launch_opts.  The product library exports no simq_tune_* (or any other setter of process state)."""
    import subprocess
    out = subprocess.run(['nm', '-D', '--defined-only', L.LIB_PATH], stdout=subprocess.PIPE, text=True, check=True).stdout
    exported = [ln.split()[-1] for ln in out.splitlines() if len(ln.split()) == 3 and ln.split()[1] in 'TDBR']
    assert not [s for s in exported if 'tune' in s], [s for s in exported if 'tune' in s]
    assert sorted(exported) == sorted(L.EXPORTS)                          # (csrc/libsimq.map: the C-ABI of include/simq.h and nothing else)
    # the scheduling options live in the plan, validated at creation, reported back
    plan = L.Plan(4, 2, options={'fwd_overlap': 0, 'wgrad_overlap': 1, 'plane_xcd': 0, 'wgrad_ksplit': 2, 'wgrad_xcd_group': 2, 'tail_split': 1})
    for k, v in (('fwd_overlap', 0), ('wgrad_overlap', 1), ('plane_xcd', 0), ('wgrad_ksplit', 2), ('wgrad_xcd_group', 2), ('tail_split', 1)):
        assert plan.options[k] == v
    d = L.Plan(4, 2).options
    assert (d['fwd_overlap'], d['wgrad_overlap'], d['plane_xcd'], d['wgrad_ksplit'], d['wgrad_xcd_group'], d['tail_split']) == (2, 4, 1, 0, 1, 0)
    for bad in ({'wgrad_ksplit': 3}, {'fwd_overlap': 3}, {'wgrad_overlap': 5}, {'wgrad_xcd_group': -1}, {'early_target_after_block': 8}):
        with pytest.raises(L.SimqError):
            L.Plan(4, 2, options=bad)
    for bad in ({'gemm_split': 2}, {'gemm_split': -1}):
        with pytest.raises(L.SimqError):
            L.Plan(4, 2, options=bad)
    assert L.Plan(4, 2, options={'gemm_split': 0}).options['gemm_split'] == 0 and d['gemm_split'] == 1
    o = L.launch_opts(tile=(96, 64), plane_xcd=0)
    assert (o.struct_bytes, o.force_bm, o.force_bn, o.plane_xcd, o.wgrad_xcd_group, o.gemm_split) == (ctypes.sizeof(L.LaunchOpts), 96, 64, 0, 1, 0)
    with pytest.raises(L.SimqError):
        L.launch_opts(nonsense=1)
    # a launch-opts struct of another version is refused before anything is launched
    o.struct_bytes = 4
    assert L.lib.c.simq_conv2d_fwd(None, None, None, None, 1, 24, 24, 64, 64, 3, 3, 1, 1, None, None, ctypes.byref(o)) != 0
    assert 'struct_bytes' in L.last_error()
    assert L.lib.c.simq_train_loss_wait(None) != 0 and 'NULL' in L.last_error()


def test_plan_layout_matches_reference_state_dict(L):
    from simq import arch
    for cin, cout in ((4, 2), (5, 1), (10, 2), (3, 1)):
        plan = L.Plan(cin, cout)
        spec = [(k[len(arch.PREFIX):], s) for k, s, kind in arch.state_spec(cin, cout) if kind in arch.TRAINABLE_KINDS]
        assert [t[0] for t in plan.tensors] == [k for k, _ in spec]          # reference order, 70 tensors
        off = 0
        for (name, toff, shape, kind), (_, rshape) in zip(plan.tensors, spec):
            assert toff == off
            if len(rshape) == 4:
                assert shape == [rshape[0], rshape[2], rshape[3], rshape[1]]  # OIHW -> OHWI
            else:
                assert shape == list(rshape)
            off += int(np.prod(rshape))
        assert plan.param_count == off
        bn_names = [k[len(arch.PREFIX):-len('.running_mean')] for k, _, kind in arch.state_spec(cin, cout) if kind == 'bn_mean']
        assert [b[0] for b in plan.bn_layers] == bn_names and len(bn_names) == 22
        assert plan.bnbuf_count == 2 * sum(b[2] for b in plan.bn_layers)
        assert plan.workspace_bytes(1) > 0 and plan.workspace_bytes(32) > 16 * plan.workspace_bytes(1) // 2


def test_argument_errors_are_reported_not_thrown(L):
    h = ctypes.c_void_p()
    assert L.lib.c.simq_plan_create(4, 9, ctypes.byref(h)) != 0
    assert 'num_output_channels' in L.last_error()
    with pytest.raises(L.SimqError):
        L.Plan(0, 1)
    plan = L.Plan(4, 2)
    assert L.lib.c.simq_forward(plan.handle, 1, 2, None, None, None, None, None, None, None) != 0
    assert 'NULL' in L.last_error()
    # every entry point validates before it launches anything (no GPU in this tier): status codes + messages, no exceptions
    assert L.lib.c.simq_backward(plan.handle, 2, None, None, None, None, None, None) != 0 and 'NULL' in L.last_error()
    assert L.lib.c.simq_train_step(None) != 0 and 'NULL' in L.last_error()
    args = L.TrainArgs()
    args.plan = plan.handle
    args.batch, args.num_nonfinal, args.global_batch = 4, 4, 4
    assert L.lib.c.simq_train_step(ctypes.byref(args)) != 0 and 'struct_bytes' in L.last_error()      # a struct of another version is refused
    args.struct_bytes = ctypes.sizeof(L.TrainArgs)
    assert L.lib.c.simq_train_step(ctypes.byref(args)) != 0 and 'NULL buffer' in L.last_error()
    assert L.lib.c.simq_bce_with_logits(None, None, 10, None, None, None) != 0 and 'bad argument' in L.last_error()
    assert L.lib.c.simq_split_last_channel(None, None, None, 10, 1, None) != 0
    assert L.lib.c.simq_clip_sgd_step(None, None, None, 0, 1.0, 0.1, 0.9, 0.0, 1, None, None, None) != 0
    assert L.lib.c.simq_weights_prepare(plan.handle, None, None, None) != 0
    h2 = ctypes.c_void_p()
    assert L.lib.c.simq_plan_create_ex(4, 2, 7, ctypes.byref(h2)) != 0 and 'precision' in L.last_error()
    assert L.lib.c.simq_workspace_bytes(plan.handle, 32) > L.lib.c.simq_workspace_bytes(plan.handle, 1) > 0
    # data-parallel gradient exchange (simq_comm_*): librccl.so.1 is bound at run time; arguments are validated before any device call
    ident = (ctypes.c_ubyte * L.COMM_ID_BYTES)()
    assert L.lib.c.simq_comm_unique_id(None) != 0 and 'NULL' in L.last_error()
    hc = ctypes.c_void_p()
    assert L.lib.c.simq_comm_init(None, 1, 0, ctypes.byref(hc)) != 0 and 'NULL' in L.last_error()
    assert L.lib.c.simq_comm_init(ident, 2, 5, ctypes.byref(hc)) != 0 and 'rank 5 of 2' in L.last_error()
    assert L.lib.c.simq_comm_allreduce(None, None, 0, 0, None) != 0 and 'bad argument' in L.last_error()
    assert L.lib.c.simq_comm_broadcast(None, None, 0, 0, None) != 0
    assert L.lib.c.simq_comm_wait(None, None) != 0 and 'NULL' in L.last_error()
    assert L.lib.c.simq_comm_world_size(None) == -1 and L.lib.c.simq_comm_rank(None) == -1
    assert L.lib.c.simq_comm_destroy(None) == 0


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under the package (or csrc) may reference it."""
    pkg = os.path.join(ROOT, 'spatial-intention-maps_amd')
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                src = open(os.path.join(d, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), os.path.join(d, f)


def test_transition_tracker_matches_reference_fixture(golden_dir):
    """simq.TransitionTracker (host logic) against tests/golden/tracker.npz, written by running the reference's
    train.TransitionTracker (train.py:47-68) on the same scripted episode (oracle/gen_golden.py gen_tracker)."""
    import numpy as np
    from oracle import cases
    from simq.learner import TransitionTracker
    g = np.load('%s/tracker.npz' % golden_dir)
    rows = cases.run_tracker(TransitionTracker)
    assert len(rows) == 2
    for i, r in enumerate(rows):
        assert r.shape == g['buffer%d' % i].shape and np.array_equal(r, g['buffer%d' % i])
    # the array handed out as next_state is the very object handed out as the next state (what the aliased ring keys on)
    initial, script = cases.tracker_script()
    tr = TransitionTracker(initial)
    last_next = {}
    for action, reward, state, done in script:
        tr.update_action(action)
        for i, lst in enumerate(tr.update_step_completed(reward, state, done)):
            for (s, a, r, ns) in lst:
                key = (i, float(s[0, 0, 0]))
                if key in last_next:
                    assert last_next.pop(key) is s
                if ns is not None:
                    last_next[(i, float(ns[0, 0, 0]))] = ns


def test_reference_written_checkpoint_loads_without_the_reference(golden_dir):
    """tests/golden/ref_checkpoint.pth.tar was pickled by the reference's own train.ReplayBuffer / train.Transition
    (oracle/gen_golden.py checkpoint; layout of train.py:324-334).  simq.load_checkpoint resolves those class names to
    this package's: ring geometry, record order, values and the state/next_state object aliasing all survive."""
    import sys
    import simq
    from simq import learner
    from oracle import cases
    assert 'train' not in sys.modules
    ck = simq.load_checkpoint(os.path.join(golden_dir, 'ref_checkpoint.pth.tar'))
    assert ck['timestep'] == 7 and ck['episode'] == 2
    assert len(ck['optimizers'][0]['param_groups'][0]['params']) == 72 and ck['optimizers'][0]['state'] == {}
    buf = ck['replay_buffers'][0]
    assert type(buf) is learner.ReplayBuffer and all(type(t) is learner.Transition for t in buf.buffer)
    tr = cases.checkpoint_transitions()
    assert buf.capacity == cases.CKPT_CAPACITY == len(buf) and buf.position == 1
    for got, want in zip(buf.buffer, [tr[3], tr[1], tr[2]]):                  # 4 pushes into 3 slots: slot 0 overwritten
        assert np.array_equal(got.state, want[0]) and got.action == want[1] and got.reward == want[2]
        assert (got.next_state is None) == (want[3] is None)
        if want[3] is not None:
            assert np.array_equal(got.next_state, want[3])
    assert buf.buffer[1].next_state is buf.buffer[2].state                     # one ndarray, pickled once
    buf.push(tr[0][0], 1, 0.0, None)                                           # and it is a working ring
    assert buf.position == 2 and len(buf) == 3


def test_checkpoint_written_here_loads_in_a_reference_style_main_script(tmp_path):
    """save_checkpoint names the ring classes `__main__.ReplayBuffer` / `__main__.Transition` -- what `python train.py` itself
    writes (train.py:26-45 define them in the script, train.py:324-334 pickle them) -- so a process that defines those two classes
    in its main script and calls plain torch.load (train.py:200) gets ITS classes back, without this package on the path.  The
    child script below stands in for the reference's: a ring class with capacity / buffer / position and the 4-field Transition.
    Our own loader reads the same file back into simq's classes."""
    import subprocess, sys, textwrap
    import torch
    import simq
    from simq import learner
    from oracle import cases
    ring = learner.ReplayBuffer(3)
    for (s, a, r, ns) in cases.checkpoint_transitions():
        ring.push(s, a, r, ns)
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(3))], lr=0.1, momentum=0.9)
    path = simq.save_checkpoint(tmp_path, 7, 2, [opt], [ring])
    import zipfile
    with zipfile.ZipFile(path) as z:                                          # torch.save's zip container: the pickle is data.pkl
        raw = b''.join(z.read(n) for n in z.namelist() if n.endswith('data.pkl'))
    assert b'__main__' in raw and b'simq' not in raw
    child = textwrap.dedent("""
        import sys
        from collections import namedtuple
        import numpy as np, torch
        assert not any(m == 'simq' or m.startswith('simq.') for m in sys.modules)
        Transition = namedtuple('Transition', ('state', 'action', 'reward', 'next_state'))
        class ReplayBuffer:
            def __init__(self, capacity):
                self.capacity, self.buffer, self.position = capacity, [], 0
        ck = torch.load(sys.argv[1], weights_only=False)
        buf = ck['replay_buffers'][0]
        assert type(buf) is ReplayBuffer and all(type(t) is Transition for t in buf.buffer), type(buf)
        assert (ck['timestep'], ck['episode'], buf.capacity, buf.position, len(buf.buffer)) == (7, 2, 3, 1, 3)
        assert buf.buffer[1].next_state is buf.buffer[2].state and buf.buffer[0].state.dtype == np.float32
        print('OK', float(sum(t.state.sum() for t in buf.buffer)))
    """)
    r = subprocess.run([sys.executable, '-c', child, path], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600,
                       env={k: v for k, v in os.environ.items() if k != 'PYTHONPATH'}, cwd=str(tmp_path))
    assert r.returncode == 0 and r.stdout.startswith('OK'), r.stdout[-2000:]
    back = simq.load_checkpoint(path)['replay_buffers'][0]
    assert type(back) is learner.ReplayBuffer and all(type(t) is learner.Transition for t in back.buffer)
    assert abs(float(r.stdout.split()[1]) - float(sum(t.state.sum() for t in back.buffer))) < 1e-3
    # reference_names=False keeps this package's names (only simq.load_checkpoint / an importable simq can read those)
    path2 = simq.save_checkpoint(tmp_path / 'own', 8, 2, [opt], [ring], reference_names=False)
    assert type(simq.load_checkpoint(path2)['replay_buffers'][0]) is learner.ReplayBuffer


def test_sampler_arrays_pack_and_unpack_unchanged():
    """learner._upload_packed: the per-batch host arrays of DeviceReplayBuffer.gather (ring slots, next-state slots, actions, rewards,
    positions) come back as tensors of the same dtype and content -- here through the non-CUDA path, which the gloo data-parallel
    tests use as well; empty arrays (an all-terminal shard) stay empty."""
    import numpy as np
    import torch
    from simq import learner
    arrays = (np.arange(7, dtype=np.int64) * 3, np.asarray([], np.int64), np.asarray([5, 0, 9], np.int64),
              np.asarray([0.5, -1.25, 3.0], np.float32), np.asarray([0, 2], np.int32))
    out = learner._upload_packed(torch.device('cpu'), arrays)
    assert [t.dtype for t in out] == [torch.int64, torch.int64, torch.int64, torch.float32, torch.int32]
    for a, t in zip(arrays, out):
        assert t.numel() == a.size and np.array_equal(t.numpy(), a)


def test_winograd_transform_matrices_are_an_exact_restatement_of_the_convolution():
    """The matrices conv_winograd.hip hard-codes (F(2x2,3x3): B^T, G, A^T and the transposes its weight gradient uses), in
    fp64 numpy: Y = A^T[(G g G^T) . (B^T d B)]A is the 3x3 correlation of a 4x4 patch, and dMt = A dY A^T, dU = dMt . V,
    dg = G^T dU G is its exact weight gradient."""
    BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
    G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)
    AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)
    rng = np.random.RandomState(0)
    d, g, dY = rng.randn(4, 4), rng.randn(3, 3), rng.randn(2, 2)
    direct = np.array([[(d[i:i + 3, j:j + 3] * g).sum() for j in range(2)] for i in range(2)])
    V, U = BT @ d @ BT.T, G @ g @ G.T
    assert np.allclose(AT @ (U * V) @ AT.T, direct, atol=1e-13)
    dg_direct = np.array([[(dY * d[ky:ky + 2, kx:kx + 2]).sum() for kx in range(3)] for ky in range(3)])
    dU = (AT.T @ dY @ AT) * V
    assert np.allclose(G.T @ dU @ G, dg_direct, atol=1e-13)
    # data gradient = the same forward transform applied to dY with the flipped filter (what the plan's dgrad launches do)
    dd_direct = np.zeros((4, 4))
    for i in range(2):
        for j in range(2):
            dd_direct[i:i + 3, j:j + 3] += dY[i, j] * g
    dyp = np.zeros((6, 6)); dyp[2:4, 2:4] = dY                       # zero-padded dY, 'full' correlation with the flipped filter
    gf = g[::-1, ::-1]
    full = np.array([[(dyp[i:i + 3, j:j + 3] * gf).sum() for j in range(4)] for i in range(4)])
    assert np.allclose(full, dd_direct, atol=1e-13)


def test_plan_options_are_a_property_of_the_plan_and_not_of_the_environment(L, monkeypatch):
    """include/simq.h simq_plan_options: defaults, round trip through simq_plan_get_options, validation -- and the product library
    neither reads SIMQ_* environment switches (it is built without SIMQ_ABLATIONS) nor is the ablation build."""
    assert L.lib.build_flags == 0
    for name in ('SIMQ_WINOGRAD', 'SIMQ_WINOGRAD_F4_GRAD', 'SIMQ_FP32_ACT_GRADS', 'SIMQ_NO_STEM16'):
        monkeypatch.setenv(name, '0')                       # would have changed the arithmetic in round 2
    d = L.Plan(4, 2).options
    assert d == {'winograd': 1, 'winograd_min_cc': 128 * 128, 'winograd_f4_forward': 1, 'winograd_f4_min_tiles': 256, 'winograd_f4_grad': 2,
                 'winograd_f4_fwd_grad_min_cc': 512 * 512, 'winograd_wgrad': 1, 'winograd_wgrad_f4': 1, 'stem_bf16': 1, 'bf16_act_grads': 1, 'keep_fp32_activations': 0,
                 'fold_eval_bn_bf16': 1, 'fuse_bn_backward_sums': 1, 'fuse_stem_backward_sums': 1, 'fuse_bn1_apply': 1, 'deterministic': 0, 'bn1_mask_from_preact': 1,
                 'wgrad_ksplit': 0, 'fwd_overlap': 2, 'wgrad_overlap': 4, 'plane_xcd': 1, 'wgrad_xcd_group': 1, 'tail_split': 0,
                 'early_target_after_block': 4, 'gemm_split': 1}
    p = L.Plan(5, 1, 'bf16', options={'stem_bf16': 0, 'keep_fp32_activations': 1})
    assert p.options['stem_bf16'] == 0 and p.options['keep_fp32_activations'] == 1 and p.options['winograd'] == 1
    # the Winograd weight cache exists only when the option is on: the option changes the plan, not a global
    assert L._c.simq_wcache_bytes(L.Plan(4, 2, options={'winograd': 0}).handle) < L._c.simq_wcache_bytes(L.Plan(4, 2).handle)
    with pytest.raises(L.SimqError):
        L.Plan(4, 2, options={'winograd_f4_grad': 3})
    with pytest.raises(L.SimqError):
        L.Plan(4, 2, options={'not_an_option': 1})
    o = L.PlanOptions()
    L._c.simq_plan_options_default(ctypes.byref(o))
    o.struct_bytes = 8                                      # a caller compiled against another struct
    h = ctypes.c_void_p()
    assert L._c.simq_plan_create_opts(4, 2, 0, ctypes.byref(o), ctypes.byref(h)) != 0 and b'struct_bytes' in L._c.simq_last_error()
    # the library binary carries no SIMQ_* switch names
    blob = open(L.LIB_PATH, 'rb').read()
    assert blob.count(b'SIMQ_WINOGRAD') == 0 and blob.count(b'SIMQ_BF16_') == 0 and blob.count(b'_DBG') == 0


def test_the_python_host_keeps_no_process_wide_behaviour_switch_either():
    """Round 6: how a learner issues its step is a StepOptions value of the call / of the learner, where a ring runs its copies an option of the
    ring -- simq.learner has no module-level A/B switch left (the five of rounds 3-5 are gone), and the option types validate their input."""
    import simq.learner as sl
    for gone in ('EARLY_TARGET_FORWARD', 'GATHER_ON_UPLOAD_STREAM', 'UPLOAD_STREAM', 'OVERLAP_TARGET_FORWARD', 'FUSED_LIBRARY_STEP', '_SIDE_STREAMS', '_EARLY_STREAMS'):
        assert not hasattr(sl, gone), gone
    upper = [n for n in vars(sl) if n.isupper() and isinstance(getattr(sl, n), bool)]
    assert not upper, 'module-level boolean switches in simq.learner: %s' % upper
    o = sl.StepOptions()
    assert o == sl.DEFAULT_STEP_OPTIONS == (True, True, True) and sl.StepOptions(fused=0, early_target_forward=0) == (False, True, False)
    with pytest.raises(AttributeError):
        o.fused = False                                     # a value, not a mutable switch
    with pytest.raises(sl.SimqError):
        sl.DeviceReplayBuffer(4, 4, device='cpu', upload_stream='sometimes')
    ring = sl.DeviceReplayBuffer(4, 4, device='cpu', upload_stream='index')
    assert ring.upload_stream == 'index' and ring.ring_on_upload_stream is False
    ring.sync_ring()                                        # (no device: nothing to wait for)


def test_streams_of_a_step_are_sorted_by_hardware_queue(monkeypatch):
    """Round 6, the logic of simq.learner._HardwareQueues on a stand-in for the device: streams land on 4 hardware queues round-robin (as the
    HIP runtime deals them), "share a queue" is what the two-spin-kernel test would measure.  A learner's side / third / early streams must
    avoid the launch stream's queue and each other's, the early stream must sit on the upload stream's queue when that is not the launch
    stream's, and streams handed back are reused without a new test."""
    import simq.learner as sl

    class FakeStream:
        made = 0

        def __init__(self, device=None, priority=0):
            FakeStream.made += 1
            self.cuda_stream = 1000 + FakeStream.made
            self.queue = FakeStream.made % 4

    main = FakeStream()
    main.queue = 0
    tests = []
    q = sl._HardwareQueues('fake')
    q.ok = True
    monkeypatch.setattr(q, '_shared', lambda a, b: tests.append(1) or a.queue == b.queue)
    monkeypatch.setattr(sl.torch.cuda, 'Stream', FakeStream)
    c_main = q.classify(main)
    got = q.acquire({c_main}, 3)
    assert len({s.queue for s in got}) == 3 and main.queue not in {s.queue for s in got}
    assert q.classify(got[0]) != c_main and len(q.reps) <= 4
    up = q.acquire({c_main}, 1)[0]
    same = q.acquire_in(q.classify(up), {c_main})
    assert same.queue == up.queue and same is not up
    n_tests, n_made = len(tests), FakeStream.made
    q.release(got)
    again = q.acquire({c_main}, 3)
    assert len(tests) == n_tests and FakeStream.made == n_made          # (from the pool: no new stream, no new test)
    assert len({s.queue for s in again}) == 3 and main.queue not in {s.queue for s in again}
    # fewer queues than roles: still never the launch stream's queue
    FakeStream.made = 0
    q2 = sl._HardwareQueues('fake2')
    q2.ok = True
    monkeypatch.setattr(q2, '_shared', lambda a, b: (a.queue % 2) == (b.queue % 2))
    m2 = FakeStream()
    got2 = q2.acquire({q2.classify(m2)}, 3)
    assert len(got2) == 3 and all((s.queue % 2) != (m2.queue % 2) for s in got2)
