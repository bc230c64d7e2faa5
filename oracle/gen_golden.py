#!/usr/bin/env python3
"""Pin the oracle against the imported reference and write tests/golden/*.npz.

TEST INFRASTRUCTURE.  Run ONLY in the build container (where /root/reference is
mounted):   python -m oracle.gen_golden

For every case the reference's own code (networks.FCN wrapped in DataParallel as
policies.py:39 does, train.train, train.ReplayBuffer) is executed on seeded
inputs and compared BIT-EXACTLY with the oracle restatement; only then are the
results stored.  The reference source never leaves /root/reference -- fixtures
hold inputs' seeds and outputs only.
"""
import os
import random
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
REF = '/root/reference'
if not os.path.isdir(REF):
    sys.exit('gen_golden: %s not mounted -- fixtures can only be generated in the build container' % REF)
sys.path.insert(0, REF)

import torch  # noqa: E402

sys.modules['utils'] = types.ModuleType('utils')                 # train.py:20, used only inside main()
_tb = types.ModuleType('torch.utils.tensorboard')
_tb.SummaryWriter = object
sys.modules['torch.utils.tensorboard'] = _tb                     # train.py:17

import networks as ref_networks  # noqa: E402
import train as ref_train  # noqa: E402


def import_ref_policies():
    """The reference's own policies.py, imported with the two modules this image lacks replaced by what policies.py uses of them:
    torchvision.transforms.ToTensor on a float32 HWC ndarray (policies.py:19,45: CHW tensor, no rescaling for float input) and the
    three VectorEnv statics (envs.py:366-376 -> the constants at envs.py:810,1090,2010).  Nothing else of either module is touched
    by policies.py:11-146."""
    if 'policies' in sys.modules:
        return sys.modules['policies']
    tv, tvt = types.ModuleType('torchvision'), types.ModuleType('torchvision.transforms')

    class ToTensor:
        def __call__(self, pic):
            assert isinstance(pic, np.ndarray) and pic.dtype == np.float32 and pic.ndim == 3
            return torch.from_numpy(np.ascontiguousarray(pic.transpose(2, 0, 1)))
    tvt.ToTensor = ToTensor
    tv.transforms = tvt
    envs = types.ModuleType('envs')

    class VectorEnv:
        _channels = {'pushing_robot': 1, 'lifting_robot': 2, 'throwing_robot': 2, 'rescue_robot': 2}    # envs.py:810,1090 (+ subclasses)

        @staticmethod
        def get_state_width():
            return 96                                                                                # envs.py:2010

        @staticmethod
        def get_num_output_channels(robot_type):
            if robot_type not in VectorEnv._channels:
                raise Exception(robot_type)                                                          # envs.py:1052
            return VectorEnv._channels[robot_type]

        @staticmethod
        def get_action_space(robot_type):
            return VectorEnv.get_num_output_channels(robot_type) * 96 * 96                           # envs.py:376
    envs.VectorEnv = VectorEnv
    saved = {k: sys.modules.get(k) for k in ('torchvision', 'torchvision.transforms', 'envs')}
    sys.modules.update({'torchvision': tv, 'torchvision.transforms': tvt, 'envs': envs})
    try:
        import policies as ref_policies
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return ref_policies


def ref_policy(cls_name, cfg, seeds, random_seed):
    """policies.DQNPolicy / DQNIntentionPolicy of the reference itself with seeded weights loaded into the nets it built
    (build_policy_nets draws torch-RNG initial weights; the python `random` stream policies.py:17 seeds is untouched by that)."""
    pol = getattr(import_ref_policies(), cls_name)(cfg, train=False, random_seed=random_seed)
    nets = list(pol.policy_nets) + list(getattr(pol, 'intention_nets', []))
    for net, (cin, cout, seed) in zip(nets, seeds):
        sd = fcn.state_from_numpy(synth.make_state_dict(cin, cout, seed))
        assert list(sd.keys()) == list(net.state_dict().keys())
        net.load_state_dict(sd)
        net.eval()                                  # (policies.py:31-32 does this for a loaded checkpoint when train=False)
    return pol

from oracle import cases, fcn, learner  # noqa: E402
from oracle import policy as opolicy  # noqa: E402
from simq import synth  # noqa: E402


def ref_net(cin, cout, seed):
    net = torch.nn.DataParallel(ref_networks.FCN(num_input_channels=cin, num_output_channels=cout))
    sd = fcn.state_from_numpy(synth.make_state_dict(cin, cout, seed))
    assert list(sd.keys()) == list(net.state_dict().keys()), 'state_dict key order differs from reference'
    for k, v in net.state_dict().items():
        assert tuple(v.shape) == tuple(sd[k].shape), k
    net.load_state_dict(sd)
    return net


def assert_same(a, b, what):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    if not torch.equal(a, b):
        raise AssertionError('oracle != reference for %s (max abs diff %g)' % (what, (a.double() - b.double()).abs().max()))


def gen_forward():
    for name, cin, cout, B, wseed, dseed in cases.FORWARD_CASES:
        x_hwc = synth.make_states(B, cin, dseed)
        x = torch.cat([learner.apply_transform(s) for s in x_hwc])
        out = {}
        # eval mode
        net = ref_net(cin, cout, wseed)
        net.eval()
        with torch.no_grad():
            q_ref = net(x)
        st = cases.oracle_state(cin, cout, wseed)
        taps = {}
        with torch.no_grad():
            q_or = fcn.fcn_forward(st, x, False, taps)
        assert_same(q_or, q_ref, name + ' eval')
        out['q_eval'] = q_ref.numpy()
        for k, t in taps.items():
            out['tap_eval.' + k] = np.array([float(t.double().mean()), float(t.double().abs().mean()),
                                             float(t.double().pow(2).mean().sqrt())])
        # train mode (batch statistics, running-stat update)
        net = ref_net(cin, cout, wseed)
        net.train()
        with torch.no_grad():
            q_ref = net(x)
        st = cases.oracle_state(cin, cout, wseed)
        with torch.no_grad():
            q_or = fcn.fcn_forward(st, x, True)
        assert_same(q_or, q_ref, name + ' train')
        for k, v in net.state_dict().items():
            assert_same(st[k], v, name + ' train buffer ' + k)
        out['q_train'] = q_ref.numpy()
        out['bn_buffers_after'] = cases.bn_buffer_vector(st).astype(np.float32)
        np.savez(os.path.join(cases.GOLDEN_DIR, name + '.npz'), **out)
        print('forward case', name, 'oracle == reference (bit-exact); saved')


def gen_train(case_list=None, keep_output=True):
    for name, cin, cout, B, wseed, dseed in (case_list or cases.TRAIN_CASES):
        cfg = cases.make_cfg(B)
        batch = cases.make_batch(cin, cout, B, dseed)
        spec = fcn.state_spec(cin, cout)
        # --- reference: two consecutive train() calls (2nd exercises momentum) ---
        policy, target = ref_net(cin, cout, wseed), ref_net(cin, cout, wseed + 1000)
        policy.train()
        target.eval()
        opt = torch.optim.SGD(policy.parameters(), lr=cases.LR, momentum=cases.MOMENTUM,
                              weight_decay=cases.WEIGHT_DECAY)                    # train.py:186
        tf = learner.apply_transform
        info_ref = [ref_train.train(cfg, policy, target, opt, batch, tf, cases.GAMMA) for _ in range(2)]
        # --- oracle fp32 ---
        st, tg = cases.oracle_state(cin, cout, wseed), cases.oracle_state(cin, cout, wseed + 1000)
        mom = [None] * len(learner.grad_keys(spec))
        extras = [{}, {}]
        info_or = [learner.train_step(cfg, st, tg, spec, mom, batch, cases.GAMMA, cases.LR, cases.MOMENTUM,
                                      cases.WEIGHT_DECAY, extras=extras[i]) for i in range(2)]
        for i in range(2):
            assert info_or[i] == info_ref[i], (name, i, info_or[i], info_ref[i])
        for k, v in policy.state_dict().items():
            assert_same(st[k], v, name + ' post-step ' + k)
        for (k, p), m in zip([(k, p) for k, p in policy.named_parameters() if p.grad is not None], mom):
            assert_same(m, opt.state[p]['momentum_buffer'], name + ' momentum ' + k)
        # --- oracle fp64 (the accuracy yardstick for gradients) ---
        st64, tg64 = cases.oracle_state(cin, cout, wseed, torch.float64), cases.oracle_state(cin, cout, wseed + 1000, torch.float64)
        ex64 = {}
        info64 = learner.train_step(cfg, st64, tg64, spec, [None] * len(mom), batch, cases.GAMMA, cases.LR,
                                    cases.MOMENTUM, cases.WEIGHT_DECAY, dtype=torch.float64, extras=ex64)
        out = {
            'loss': np.array([i['loss'] for i in info_ref]), 'td_error': np.array([i['td_error'] for i in info_ref]),
            'total_norm': np.array([e['total_norm'] for e in extras]),
            'q_sa': extras[0]['q'].numpy(), 'y': extras[0]['y'].numpy(),
            'output_step1': extras[0]['output'].numpy() if (B <= 4 and keep_output) else np.zeros(0, np.float32),
            'param_summary_after2': cases.param_summary(st, spec),
            'bn_buffers_after2': cases.bn_buffer_vector(st).astype(np.float32),
            'num_batches_tracked': np.array([int(st[k]) for k in st if k.endswith('num_batches_tracked')]),
            'loss64': np.array(info64['loss']), 'td_error64': np.array(info64['td_error']),
            'total_norm64': np.array(ex64['total_norm']),
        }
        g32, g64 = cases.grad_summary(extras[0]['grads']), cases.grad_summary(ex64['grads'])
        out['grad_keys'] = np.array(list(g32.keys()))
        out['grad32'] = np.stack(list(g32.values()))
        out['grad64'] = np.stack(list(g64.values()))
        # global relative L2 error of the reference's own fp32 gradient vs fp64
        num = sum(float((extras[0]['grads'][k].double() - ex64['grads'][k]).pow(2).sum()) for k in g32)
        den = sum(float(ex64['grads'][k].pow(2).sum()) for k in g32)
        out['ref_fp32_grad_relerr'] = np.array((num / den) ** 0.5)
        np.savez(os.path.join(cases.GOLDEN_DIR, name + '.npz'), **out)
        print('train case', name, 'oracle == reference (bit-exact, 2 steps); ref fp32 grad rel err vs fp64 = %.3g'
              % out['ref_fp32_grad_relerr'])


def gen_train_sized(case_list=None):
    """BASELINE configs[2..4]'s per-GPU shapes (cases.TRAIN_CASES_SIZED): the reference's own train.train (imported), two consecutive
    calls in fp32, bit-exact against the oracle; the fp64 oracle as the yardstick; and the error the REFERENCE makes on the same
    batch (i) in its own fp32 and (ii) with its own modules under torch.autocast('cpu', bfloat16) -- the calibration the HIP bf16
    path is held to at these sizes (train-mode BatchNorm over 64-128 samples is well conditioned, unlike the 4-8 sample fixtures).
    Only summaries are stored: scalars, per-transition q / y, per-tensor gradient norms and 16 sampled elements per tensor of the
    gradient and of the first parameter update."""
    from torch.nn.functional import smooth_l1_loss
    import time
    for name, cin, cout, B, wseed, dseed in (case_list or cases.TRAIN_CASES_SIZED):
        t_start = time.time()
        cfg, batch, spec = cases.make_cfg(B), cases.make_batch(cin, cout, B, dseed), fcn.state_spec(cin, cout)
        gkeys = learner.grad_keys(spec)
        # --- the reference itself, fp32, two steps ---
        policy, target = ref_net(cin, cout, wseed), ref_net(cin, cout, wseed + 1000)
        policy.train()
        target.eval()
        opt = torch.optim.SGD(policy.parameters(), lr=cases.LR, momentum=cases.MOMENTUM, weight_decay=cases.WEIGHT_DECAY)   # train.py:186
        p0 = {k: v.detach().clone() for k, v in policy.state_dict().items()}
        info_ref = [ref_train.train(cfg, policy, target, opt, batch, learner.apply_transform, cases.GAMMA)]
        p1_ref = {k: v.detach().clone() for k, v in policy.state_dict().items()}
        info_ref.append(ref_train.train(cfg, policy, target, opt, batch, learner.apply_transform, cases.GAMMA))
        # --- oracle fp32, bit-exact (also hands out the pre-clip gradient) ---
        st, tg = cases.oracle_state(cin, cout, wseed), cases.oracle_state(cin, cout, wseed + 1000)
        mom, extras = [None] * len(gkeys), [{}, {}]
        info_or = [learner.train_step(cfg, st, tg, spec, mom, batch, cases.GAMMA, cases.LR, cases.MOMENTUM, cases.WEIGHT_DECAY,
                                      extras=extras[i]) for i in range(2)]
        assert info_or == info_ref, (name, info_or, info_ref)
        for k, v in policy.state_dict().items():
            assert_same(st[k], v, name + ' post-step ' + k)
        for (k, p), m in zip([(k, p) for k, p in policy.named_parameters() if p.grad is not None], mom):
            assert_same(m, opt.state[p]['momentum_buffer'], name + ' momentum ' + k)
        print('  %s: reference + oracle fp32 done (%.0f s)' % (name, time.time() - t_start), flush=True)
        # --- oracle fp64, two steps ---
        st64, tg64 = cases.oracle_state(cin, cout, wseed, torch.float64), cases.oracle_state(cin, cout, wseed + 1000, torch.float64)
        mom64, ex64 = [None] * len(gkeys), {}
        i64 = [learner.train_step(cfg, st64, tg64, spec, mom64, batch, cases.GAMMA, cases.LR, cases.MOMENTUM, cases.WEIGHT_DECAY,
                                  dtype=torch.float64, extras=ex64)]
        p1_64 = {k: st64[k].detach().clone() for k in gkeys}
        bn1_64 = cases.bn_buffer_vector(st64)
        i64.append(learner.train_step(cfg, st64, tg64, spec, mom64, batch, cases.GAMMA, cases.LR, cases.MOMENTUM, cases.WEIGHT_DECAY,
                                      dtype=torch.float64))
        print('  %s: oracle fp64 done (%.0f s)' % (name, time.time() - t_start), flush=True)
        # --- the reference's modules under bf16 autocast (train.py:109-135 restated so that autocast wraps the three forwards) ---
        pol16, tgt16 = ref_net(cin, cout, wseed), ref_net(cin, cout, wseed + 1000)
        pol16.train()
        tgt16.eval()
        state_b = torch.cat([learner.apply_transform(s) for s in batch.state])
        act = torch.tensor(batch.action, dtype=torch.long)
        rew = torch.tensor(batch.reward, dtype=torch.float32)
        nf = torch.cat([learner.apply_transform(s) for s in batch.next_state if s is not None])
        mask = torch.tensor([s is not None for s in batch.next_state], dtype=torch.bool)
        with torch.autocast('cpu', dtype=torch.bfloat16):
            q16 = pol16(state_b).float().view(B, -1).gather(1, act.unsqueeze(1)).squeeze(1)
            nsv = torch.zeros(B)
            with torch.no_grad():
                best = pol16(nf).float().view(nf.size(0), -1).max(1)[1].view(-1, 1)
                nsv[mask] = tgt16(nf).float().view(nf.size(0), -1).gather(1, best).view(-1)
        y16 = rew + cases.GAMMA * nsv
        loss16 = smooth_l1_loss(q16, y16)
        loss16.backward()
        named16 = {('module.' + k if not k.startswith('module.') else k): p for k, p in pol16.module.named_parameters()}
        g16 = {k: named16[k].grad.detach().clone() for k in gkeys}
        opt16 = torch.optim.SGD(pol16.parameters(), lr=cases.LR, momentum=cases.MOMENTUM, weight_decay=cases.WEIGHT_DECAY)
        torch.nn.utils.clip_grad_norm_(pol16.parameters(), cases.CLIP)
        opt16.step()
        p1_16 = {k: v.detach().clone() for k, v in pol16.state_dict().items()}
        print('  %s: reference under bf16 autocast done (%.0f s)' % (name, time.time() - t_start), flush=True)

        def sampled(d, cast=lambda t: t.double()):
            return np.stack([cast(d[k]).reshape(-1)[torch.tensor(cases.sample_indices(d[k].numel()))].numpy() for k in gkeys])
        rl2 = lambda a, b: float(np.sqrt(((np.asarray(a, np.float64) - b) ** 2).sum() / (b ** 2).sum()))
        relmax = lambda a, b: float(np.abs(np.asarray(a, np.float64) - b).max() / np.abs(b).max())
        g64s, g32s, g16s = sampled(ex64['grads']), sampled(extras[0]['grads']), sampled(g16)
        d64s = sampled({k: p1_64[k] - p0[k].double() for k in gkeys})
        d32s = sampled({k: p1_ref[k].double() - p0[k].double() for k in gkeys})
        d16s = sampled({k: p1_16[k].double() - p0[k].double() for k in gkeys})
        fullrel = lambda g: (sum(float((g[k].double() - ex64['grads'][k]).pow(2).sum()) for k in gkeys)
                             / sum(float(ex64['grads'][k].pow(2).sum()) for k in gkeys)) ** 0.5
        out = {
            'loss': np.array([i['loss'] for i in info_ref]), 'td_error': np.array([i['td_error'] for i in info_ref]),
            'total_norm': np.array([e['total_norm'] for e in extras]),
            'q_sa': extras[0]['q'].numpy(), 'y': extras[0]['y'].numpy(),
            'param_summary_after2': cases.param_summary(st, spec),
            'bn_buffers_after2': cases.bn_buffer_vector(st).astype(np.float32),
            'num_batches_tracked': np.array([int(st[k]) for k in st if k.endswith('num_batches_tracked')]),
            # fp64 yardstick
            'loss64': np.array([i['loss'] for i in i64]), 'td_error64': np.array([i['td_error'] for i in i64]),
            'total_norm64': np.array(ex64['total_norm']), 'q_sa64': ex64['q'].numpy(), 'y64': ex64['y'].numpy(),
            'grad_keys': np.array(gkeys),
            'grad_norm64': np.array([float(ex64['grads'][k].norm()) for k in gkeys]),
            'grad64': g64s, 'dparam64': d64s, 'bn_buffers_after1_64': bn1_64,
            # the reference's own fp32 against fp64 (what "as accurate as the reference" means on this batch)
            'ref_fp32_grad_relerr': np.array(fullrel(extras[0]['grads'])), 'ref_grad_err': np.array(rl2(g32s, g64s)),
            'ref_dparam_err': np.array(rl2(d32s, d64s)),
            'ref_loss_err': np.array([abs(info_ref[j]['loss'] - i64[j]['loss']) / abs(i64[j]['loss']) for j in range(2)]),
            'ref_q_err': np.array(relmax(extras[0]['q'].numpy(), ex64['q'].numpy())),
            # the reference under bf16 autocast against fp64, on this batch
            'bf16cal_loss': np.array(abs(float(loss16.detach()) - i64[0]['loss']) / abs(i64[0]['loss'])),
            'bf16cal_td_error': np.array(abs(float(torch.abs(q16 - y16).mean()) - i64[0]['td_error']) / abs(i64[0]['td_error'])),
            'bf16cal_q_sa': np.array(relmax(q16.detach().numpy(), ex64['q'].numpy())),
            'bf16cal_y': np.array(relmax(y16.detach().numpy(), ex64['y'].numpy())),
            'bf16cal_grad': np.array(fullrel(g16)), 'bf16cal_grad_sampled': np.array(rl2(g16s, g64s)),
            'bf16cal_dparam_sampled': np.array(rl2(d16s, d64s)),
        }
        np.savez_compressed(os.path.join(cases.GOLDEN_DIR, name + '.npz'), **out)
        print('sized train case %s: oracle == reference (bit-exact, 2 steps).  vs fp64 -- reference fp32: grad %.3g (sampled %.3g), '
              'update %.3g, loss %.2g / %.2g, q %.2g;  reference bf16-autocast: loss %.3g, td %.3g, q_sa %.3g, y %.3g, grad %.3g '
              '(sampled %.3g), update %.3g   [%.0f s]'
              % (name, out['ref_fp32_grad_relerr'], out['ref_grad_err'], out['ref_dparam_err'], out['ref_loss_err'][0],
                 out['ref_loss_err'][1], out['ref_q_err'], out['bf16cal_loss'], out['bf16cal_td_error'], out['bf16cal_q_sa'],
                 out['bf16cal_y'], out['bf16cal_grad'], out['bf16cal_grad_sampled'], out['bf16cal_dparam_sampled'],
                 time.time() - t_start), flush=True)


def gen_dense_grad():
    """The reference's own networks.FCN (train mode, wrapped as policies.py:39 does) differentiated through a DENSE upstream gradient
    at BASELINE configs[2..4]'s per-GPU sizes: loss = sum(Q * R), R seeded (cases.dense_upstream).  Unlike the TD loss's one-hot
    gradient, a dense gradient does not cancel catastrophically in the train-mode BatchNorm backward, so this is the
    network-level gradient fixture with a TIGHT bar: the fp64 gradient (per-tensor norms + 16 sampled elements per tensor) beside the
    error of the reference's own fp32 and of the reference under bf16 autocast on the same inputs.  The oracle's forward_backward
    restatement is asserted bit-exact against the reference first."""
    import time
    for name, cin, cout, B, wseed, dseed in cases.DENSE_GRAD_CASES:
        t0 = time.time()
        spec = fcn.state_spec(cin, cout)
        gkeys = learner.grad_keys(spec)
        x = torch.cat([learner.apply_transform(s) for s in synth.make_states(B, cin, dseed)])
        R = torch.from_numpy(cases.dense_upstream(cout, B, dseed))

        def ref_grads(autocast):
            net = ref_net(cin, cout, wseed)
            net.train()
            if autocast:
                with torch.autocast('cpu', dtype=torch.bfloat16):
                    q = net(x).float()
            else:
                q = net(x)
            (q * R).sum().backward()
            named = {'module.' + k: p for k, p in net.module.named_parameters()}
            return q.detach(), {k: named[k].grad.detach().clone() for k in gkeys}
        q32, g32 = ref_grads(False)
        # oracle fp32 == reference (bit-exact), then fp64
        def oracle_grads(dtype):
            st = cases.oracle_state(cin, cout, wseed, dtype)
            params = [st[k] for k in gkeys]
            for p_ in params:
                p_.requires_grad_(True)
            q = fcn.fcn_forward(st, x.to(dtype), True)
            grads = torch.autograd.grad((q * R.to(dtype)).sum(), params)
            return q.detach(), dict(zip(gkeys, grads))
        qo, go = oracle_grads(torch.float32)
        assert_same(qo, q32, name + ' forward')
        for k in gkeys:
            assert_same(go[k], g32[k], name + ' gradient ' + k)
        q64, g64 = oracle_grads(torch.float64)
        q16, g16 = ref_grads(True)

        def sampled(d):
            return np.stack([d[k].double().reshape(-1)[torch.tensor(cases.sample_indices(d[k].numel()))].numpy() for k in gkeys])
        rl2 = lambda a, b: float(np.sqrt(((np.asarray(a, np.float64) - b) ** 2).sum() / (b ** 2).sum()))
        fullrel = lambda g: (sum(float((g[k].double() - g64[k]).pow(2).sum()) for k in gkeys) / sum(float(g64[k].pow(2).sum()) for k in gkeys)) ** 0.5
        worst = lambda g: max(float((g[k].double() - g64[k]).norm() / g64[k].norm()) for k in gkeys if float(g64[k].norm()) > 1e-3 * max(float(g64[j].norm()) for j in gkeys))
        relmax = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())
        out = dict(grad_keys=np.array(gkeys), grad_norm64=np.array([float(g64[k].norm()) for k in gkeys]), grad64=sampled(g64),
                   q_checksum64=np.array([float(q64.sum()), float(q64.abs().sum()), float((q64 * R.double()).sum())]),
                   ref_fp32_grad=np.array(fullrel(g32)), ref_fp32_grad_sampled=np.array(rl2(sampled(g32), sampled(g64))),
                   ref_fp32_worst_tensor=np.array(worst(g32)), ref_fp32_q=np.array(relmax(q32, q64)),
                   bf16cal_grad=np.array(fullrel(g16)), bf16cal_grad_sampled=np.array(rl2(sampled(g16), sampled(g64))),
                   bf16cal_worst_tensor=np.array(worst(g16)), bf16cal_q=np.array(relmax(q16, q64)))
        np.savez_compressed(os.path.join(cases.GOLDEN_DIR, name + '.npz'), **out)
        print('dense-gradient case %s: oracle == reference (bit-exact).  vs fp64 -- reference fp32: Q %.2g, gradient %.3g (sampled %.3g, worst '
              'tensor %.3g);  reference bf16-autocast: Q %.3g, gradient %.3g (sampled %.3g, worst tensor %.3g)   [%.0f s]'
              % (name, out['ref_fp32_q'], out['ref_fp32_grad'], out['ref_fp32_grad_sampled'], out['ref_fp32_worst_tensor'], out['bf16cal_q'],
                 out['bf16cal_grad'], out['bf16cal_grad_sampled'], out['bf16cal_worst_tensor'], time.time() - t0), flush=True)


def gen_bf16_points():
    """Fixtures of the bf16 plan's MODEL (oracle/bf16_points.py: fp64 arithmetic, operands rounded to bf16 where libsimq's plain-bf16 plan
    rounds them) at the sizes the bf16 configs run: the TD step of cases.TRAIN_CASES_SIZED[:2] and the dense-upstream gradient of
    cases.DENSE_GRAD_CASES[0].  Validated here before anything is written: with the rounding points OFF the model is the fp64 oracle
    (1e-12; the oracle itself is pinned bit-exact to the imported reference by the generators above), with them ON it lands inside the
    reference's own bf16-autocast calibration of the same batch (train_sized / dense_grad fixtures).  Summaries only."""
    import time
    from . import bf16_points as bp
    rl2 = lambda a, b: float(np.sqrt(((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2).sum() / (np.asarray(b, np.float64) ** 2).sum()))

    def sampled(d, keys):
        return np.stack([d[k].double().reshape(-1)[torch.tensor(cases.sample_indices(d[k].numel()))].numpy() for k in keys])
    for name, cin, cout, B, wseed, dseed in cases.TRAIN_CASES_SIZED[:2]:
        t0 = time.time()
        cfg, batch, spec = cases.make_cfg(B), cases.make_batch(cin, cout, B, dseed), fcn.state_spec(cin, cout)
        gkeys = learner.grad_keys(spec)

        def run(fn, **kw):
            st, tg = cases.oracle_state(cin, cout, wseed, torch.float64), cases.oracle_state(cin, cout, wseed + 1000, torch.float64)
            ex = {}
            info = fn(cfg, st, tg, spec, [None] * len(gkeys), batch, cases.GAMMA, cases.LR, cases.MOMENTUM, cases.WEIGHT_DECAY, extras=ex, **kw)
            return info, ex, st
        i64, e64, s64 = run(learner.train_step, dtype=torch.float64)
        ioff, eoff, soff = run(bp.train_step, points=False)
        assert abs(ioff['loss'] - i64['loss']) <= 1e-12 * abs(i64['loss']), (name, ioff, i64)
        assert rl2(sampled(eoff['grads'], gkeys), sampled(e64['grads'], gkeys)) < 1e-10, name
        ion, eon, son = run(bp.train_step, points=True)
        cal = np.load(os.path.join(cases.GOLDEN_DIR, name + '.npz'))
        g_on, g_64 = sampled(eon['grads'], gkeys), sampled(e64['grads'], gkeys)
        dist = dict(q_sa=float((eon['q'] - e64['q']).abs().max() / e64['q'].abs().max()), loss=abs(ion['loss'] - i64['loss']) / abs(i64['loss']),
                    grad_sampled=rl2(g_on, g_64), y_maxabs=float((eon['y'] - e64['y']).abs().max()))
        print('  %s: model vs fp64 -- q_sa %.3g, loss %.3g, sampled gradient %.3g, max |dy| %.3g;  reference under autocast on this batch: q_sa %.3g, '
              'loss %.3g, gradient %.3g' % (name, dist['q_sa'], dist['loss'], dist['grad_sampled'], dist['y_maxabs'], float(cal['bf16cal_q_sa']),
                                            float(cal['bf16cal_loss']), float(cal['bf16cal_grad'])), flush=True)
        assert dist['q_sa'] <= 2.0 * float(cal['bf16cal_q_sa']) and dist['grad_sampled'] <= 2.0 * float(cal['bf16cal_grad']), (name, dist)
        out = dict(loss=np.array(ion['loss']), td_error=np.array(ion['td_error']), q_sa=eon['q'].numpy(), y=eon['y'].numpy(),
                   total_norm=np.array(eon['total_norm']), grad_keys=np.array(gkeys),
                   grad_norm=np.array([float(eon['grads'][k].norm()) for k in gkeys]), grad=g_on,
                   q_checksum=np.array([float(eon['output'].sum()), float(eon['output'].abs().sum())]),
                   bn_buffers=cases.bn_buffer_vector(son), vs_fp64=np.array([dist['q_sa'], dist['loss'], dist['grad_sampled'], dist['y_maxabs']]))
        np.savez_compressed(os.path.join(cases.GOLDEN_DIR, 'bf16pts_' + name + '.npz'), **out)
        print('bf16-points case %s saved [%.0f s]' % (name, time.time() - t0), flush=True)
    for name, cin, cout, B, wseed, dseed in cases.DENSE_GRAD_CASES[:1]:
        t0 = time.time()
        spec = fcn.state_spec(cin, cout)
        gkeys = learner.grad_keys(spec)
        x = torch.cat([learner.apply_transform(s_) for s_ in synth.make_states(B, cin, dseed)]).double()
        R = torch.from_numpy(cases.dense_upstream(cout, B, dseed)).double()
        q_off, g_off = bp.dense_gradient(cases.oracle_state(cin, cout, wseed, torch.float64), spec, x, R, points=False)
        cal = np.load(os.path.join(cases.GOLDEN_DIR, name + '.npz'))
        assert rl2(sampled(dict(zip(gkeys, g_off)), gkeys), cal['grad64']) < 1e-10, name
        q_on, g_on = bp.dense_gradient(cases.oracle_state(cin, cout, wseed, torch.float64), spec, x, R, points=True)
        gs = sampled(dict(zip(gkeys, g_on)), gkeys)
        d_grad, d_q = rl2(gs, cal['grad64']), float((q_on - q_off).abs().max() / q_off.abs().max())
        print('  %s: model vs fp64 -- Q %.3g, sampled gradient %.3g;  reference under autocast: Q %.3g, gradient (sampled) %.3g'
              % (name, d_q, d_grad, float(cal['bf16cal_q']), float(cal['bf16cal_grad_sampled'])), flush=True)
        assert d_q <= 2.0 * float(cal['bf16cal_q']) and d_grad <= 2.0 * float(cal['bf16cal_grad_sampled']), (name, d_q, d_grad)
        out = dict(grad_keys=np.array(gkeys), grad_norm=np.array([float(g.norm()) for g in g_on]), grad=gs,
                   q_checksum=np.array([float(q_on.sum()), float(q_on.abs().sum()), float((q_on * R).sum())]),
                   q_sample=q_on.reshape(-1)[torch.tensor(cases.sample_indices(q_on.numel(), 4096))].numpy(), vs_fp64=np.array([d_q, d_grad]))
        np.savez_compressed(os.path.join(cases.GOLDEN_DIR, 'bf16pts_' + name + '.npz'), **out)
        print('bf16-points case %s saved [%.0f s]' % (name, time.time() - t0), flush=True)


def gen_dp():
    """Fixture G7: the reference's multi-GPU form is nn.DataParallel (policies.py:39) -- the minibatch is cut
    into contiguous chunks, every replica runs the reference's own FCN on its chunk with ITS OWN train-mode BatchNorm statistics,
    the replicas' gradients are summed and only replica 0's running statistics persist.  Emulated here on the CPU with the
    reference classes themselves: replica r > 0 is a deepcopy of the reference module (what DataParallel.replicate produces),
    each runs train.py:114-129 on its chunk with the Huber SUM divided by the GLOBAL batch, gradients are added in rank order.
    The oracle's dp_emulation must agree bit for bit; the fp64 oracle is the yardstick stored beside it."""
    import copy
    from torch.nn.functional import smooth_l1_loss
    for name, cin, cout, gB, world, wseed, dseed in cases.DP_CASES:
        cfg = cases.make_cfg(gB)
        batch = cases.make_batch(cin, cout, gB, dseed)
        spec = fcn.state_spec(cin, cout)
        policy, target = ref_net(cin, cout, wseed), ref_net(cin, cout, wseed + 1000)
        policy.train()
        target.eval()
        chunk = -(-gB // world)
        total, sums, q_all, y_all, empty_shards = None, torch.zeros(2), [], [], 0
        for r in range(world):
            lo, hi = min(r * chunk, gB), min((r + 1) * chunk, gB)
            if lo == hi:
                continue
            replica = policy if r == 0 else copy.deepcopy(policy)
            replica.zero_grad()
            st_b = torch.cat([learner.apply_transform(s) for s in batch.state[lo:hi]])
            act = torch.tensor(batch.action[lo:hi], dtype=torch.long)
            rew = torch.tensor(batch.reward[lo:hi], dtype=torch.float32)
            nf = [learner.apply_transform(s) for s in batch.next_state[lo:hi] if s is not None]
            mask = torch.tensor([s is not None for s in batch.next_state[lo:hi]], dtype=torch.bool)
            out = replica(st_b)                                                                  # train.py:114
            q = out.view(hi - lo, -1).gather(1, act.unsqueeze(1)).squeeze(1)                     # train.py:115
            nsv = torch.zeros(hi - lo)
            if nf:
                nfns = torch.cat(nf)
                with torch.no_grad():                                                            # train.py:118-122
                    best = replica(nfns).view(len(nf), -1).max(1)[1].view(len(nf), 1)
                    nsv[mask] = target(nfns).view(len(nf), -1).gather(1, best).view(-1)
            else:
                empty_shards += 1
            y = rew + cases.GAMMA * nsv                                                          # train.py:126
            huber = smooth_l1_loss(q, y, reduction='sum')
            (huber / gB).backward()
            flat = torch.cat([p.grad.reshape(-1) for p in replica.parameters() if p.grad is not None])
            total = flat if total is None else total + flat
            sums = sums + torch.stack([huber.detach(), torch.abs(q - y).detach().sum()])
            q_all.append(q.detach()); y_all.append(y.detach())
        # oracle restatement, fp32: bit-exact with the emulation above
        st, tg = cases.oracle_state(cin, cout, wseed), cases.oracle_state(cin, cout, wseed + 1000)
        o_total, o_loss, o_td = learner.dp_emulation(cfg, st, tg, spec, batch, world, cases.GAMMA)
        assert_same(o_total, total, name + ' gradient sum')
        assert o_loss == float(sums[0]) / gB and o_td == float(sums[1]) / gB, (name, o_loss, o_td, sums)
        for k, v in policy.state_dict().items():
            assert_same(st[k], v, name + ' replica-0 buffer ' + k)
        # fp64 yardstick
        st64, tg64 = cases.oracle_state(cin, cout, wseed, torch.float64), cases.oracle_state(cin, cout, wseed + 1000, torch.float64)
        t64, loss64, td64 = learner.dp_emulation(cfg, st64, tg64, spec, batch, world, cases.GAMMA, dtype=torch.float64)
        gkeys = learner.grad_keys(spec)

        def split(flat):
            out, off = {}, 0
            for k in gkeys:
                n = st[k].numel()
                out[k] = flat[off:off + n].view(st[k].shape)
                off += n
            return out
        g32, g64 = cases.grad_summary(split(total)), cases.grad_summary(split(t64))
        relerr = float((total.double() - t64).norm() / t64.norm())
        # the same minibatch as ONE replica (what a single device computes): how far per-shard BN statistics move the gradient
        st1, tg1 = cases.oracle_state(cin, cout, wseed, torch.float64), cases.oracle_state(cin, cout, wseed + 1000, torch.float64)
        one, _, _ = learner.dp_emulation(cfg, st1, tg1, spec, batch, 1, cases.GAMMA, dtype=torch.float64)
        np.savez(os.path.join(cases.GOLDEN_DIR, name + '.npz'),
                 loss=np.array(o_loss), td_error=np.array(o_td), loss64=np.array(loss64), td_error64=np.array(td64),
                 total_norm=np.array(float(total.norm())), total_norm64=np.array(float(t64.norm())),
                 q_sa=torch.cat(q_all).numpy(), y=torch.cat(y_all).numpy(), grad_keys=np.array(gkeys),
                 grad32=np.stack(list(g32.values())), grad64=np.stack(list(g64.values())), ref_fp32_grad_relerr=np.array(relerr),
                 bn_buffers_after=cases.bn_buffer_vector(st).astype(np.float32), all_terminal_shards=np.array(empty_shards),
                 shard_vs_single_replica_relerr=np.array(float((t64 - one).norm() / one.norm())))
        print('dp case', name, 'oracle.dp_emulation == reference replicas (bit-exact); fp32 grad rel err vs fp64 = %.3g; '
              'all-terminal shards: %d; per-shard-BN vs single replica: %.3g' % (relerr, empty_shards, float((t64 - one).norm() / one.norm())))


def gen_dp_literal():
    """nn.DataParallel's literal scatter (policies.py:39): the double-DQN forward policy_net(non_final_next_states) (train.py:121) hands
    DataParallel the COMPACTED non-final tensor, which torch.chunk cuts into ceil(N'/world) rows per replica -- so replica r picks the greedy
    actions of compacted chunk r, not of the next states of ITS slice of the minibatch (the default sharding of this package: gen_dp).
    Emulated with the reference's own modules: the first forward replica by replica over torch.chunk(state_batch), the no-grad forward
    replica by replica over torch.chunk(non_final_next_states) itself (replica r > 0 = a deepcopy, what DataParallel.replicate produces;
    only replica 0's running statistics persist), then train.py:122-132 on the gathered outputs exactly as the reference's device 0 does.
    The oracle's dp_emulation_literal must agree bit for bit."""
    import copy
    from torch.nn.functional import smooth_l1_loss
    for name, cin, cout, gB, world, wseed, dseed in cases.DP_LITERAL_CASES:
        cfg = cases.make_cfg(gB)
        batch = cases.make_batch(cin, cout, gB, dseed)
        spec = fcn.state_spec(cin, cout)
        policy, target = ref_net(cin, cout, wseed), ref_net(cin, cout, wseed + 1000)
        policy.train()
        target.eval()
        state_b = torch.cat([learner.apply_transform(s) for s in batch.state])
        act = torch.tensor(batch.action, dtype=torch.long)
        rew = torch.tensor(batch.reward, dtype=torch.float32)
        nfns = torch.cat([learner.apply_transform(s) for s in batch.next_state if s is not None])
        mask = torch.tensor([s is not None for s in batch.next_state], dtype=torch.bool)
        policy.zero_grad()
        # train.py:114 through DataParallel: scatter = torch.chunk(x, world), replicas forward their chunk, outputs are gathered
        outs = []
        for r, xs in enumerate(torch.chunk(state_b, world)):
            replica = policy if r == 0 else copy.deepcopy(policy)
            outs.append((replica, replica(xs)))
        # gradients flow back into each replica's own parameters; DataParallel reduce-adds them onto device 0 = the rank-order sum
        output = torch.cat([o for _, o in outs])
        q = output.view(gB, -1).gather(1, act.unsqueeze(1)).squeeze(1)                                     # train.py:115
        nsv = torch.zeros(gB)
        with torch.no_grad():                                                                              # train.py:118-122
            bests = []
            for r, xs in enumerate(torch.chunk(nfns, world)):
                replica = policy if r == 0 else copy.deepcopy(policy)
                bests.append(replica(xs).view(xs.size(0), -1).max(1)[1])
            best = torch.cat(bests)
            nsv[mask] = target(nfns).view(nfns.size(0), -1).gather(1, best.view(-1, 1)).view(-1)
        y = rew + cases.GAMMA * nsv                                                                        # train.py:126
        loss = smooth_l1_loss(q, y)                                                                        # train.py:129
        loss.backward()
        total = None
        for replica, _ in outs:
            flat = torch.cat([p.grad.reshape(-1) for p in replica.parameters() if p.grad is not None])
            total = flat if total is None else total + flat
        td = torch.abs(q - y).detach().mean()
        # oracle restatement, fp32: bit-exact
        st, tg = cases.oracle_state(cin, cout, wseed), cases.oracle_state(cin, cout, wseed + 1000)
        o_total, o_loss, o_td, o_best, o_q, o_y = learner.dp_emulation_literal(cfg, st, tg, spec, batch, world, cases.GAMMA)
        assert torch.equal(o_best, best), (name, o_best, best)
        assert_same(o_q, q.detach(), name + ' q')
        assert_same(o_y, y, name + ' y')
        assert_same(o_total, total, name + ' gradient sum')
        for k, v in policy.state_dict().items():
            assert_same(st[k], v, name + ' replica-0 buffer ' + k)
        # how the default sharding of this package (next states follow their transitions' slice) differs on this batch
        st2, tg2 = cases.oracle_state(cin, cout, wseed), cases.oracle_state(cin, cout, wseed + 1000)
        s_total, s_loss, s_td = learner.dp_emulation(cfg, st2, tg2, spec, batch, world, cases.GAMMA)
        # fp64 yardstick
        st64, tg64 = cases.oracle_state(cin, cout, wseed, torch.float64), cases.oracle_state(cin, cout, wseed + 1000, torch.float64)
        t64, loss64, td64, best64, _, y64 = learner.dp_emulation_literal(cfg, st64, tg64, spec, batch, world, cases.GAMMA, dtype=torch.float64)
        gkeys = learner.grad_keys(spec)

        def split(flat):
            out, off = {}, 0
            for k in gkeys:
                n = st[k].numel()
                out[k] = flat[off:off + n].view(st[k].shape)
                off += n
            return out
        g32, g64 = cases.grad_summary(split(total)), cases.grad_summary(split(t64))
        np.savez(os.path.join(cases.GOLDEN_DIR, name + '.npz'),
                 loss=np.array(float(loss)), td_error=np.array(float(td)), loss64=np.array(loss64), td_error64=np.array(td64),
                 total_norm=np.array(float(total.norm())), total_norm64=np.array(float(t64.norm())),
                 q_sa=q.detach().numpy(), y=y.numpy(), best=best.numpy(), best64=best64.numpy(), grad_keys=np.array(gkeys),
                 grad32=np.stack(list(g32.values())), grad64=np.stack(list(g64.values())),
                 ref_fp32_grad_relerr=np.array(float((total.double() - t64).norm() / t64.norm())),
                 bn_buffers_after=cases.bn_buffer_vector(st).astype(np.float32),
                 slice_sharding_loss=np.array(s_loss), slice_vs_literal_grad=np.array(float((s_total - total).norm() / total.norm())),
                 nonfinal=np.array(int(mask.sum())))
        print('dp-literal case %s: oracle == reference replicas (bit-exact); %d non-final next states in chunks of %d; greedy actions '
              'fp32 == fp64: %s; slice-sharding differs by loss %.3g vs %.3g, gradient %.3g' % (
                  name, int(mask.sum()), -(-int(mask.sum()) // world), bool(torch.equal(best, best64)), s_loss, float(loss),
                  float((s_total - total).norm() / total.norm())))


def gen_grad_study(case_list=None, fname='grad_study.npz'):
    """Gradient-parity study: fp32 gradients of these small train-mode-BN batches are only 1e-4 .. 1e-2
    accurate -- for the reference as much as for any other fp32 implementation -- and WHICH implementation is luckier changes from
    batch to batch.  So the bar is a distribution: 10 seeded B=8 and 3 seeded B=32 batches; per case the reference's own fp32
    train.train (imported, two consecutive calls) and the fp64 oracle.  Stored per case: the fp64 gradient / parameter-update
    summaries (16 sampled elements per tensor) and the REFERENCE-fp32 error on exactly those samples -- the yardstick the GPU test
    (tests/test_gpu_fcn.py::test_gradient_parity_distribution) holds the HIP path to (median <= 2 x, no case > 10 x)."""
    case_list = case_list or cases.GRAD_STUDY_CASES
    out = {'names': np.array([c[0] for c in case_list])}
    for name, cin, cout, B, wseed, dseed in case_list:
        cfg, batch, spec = cases.make_cfg(B), cases.make_batch(cin, cout, B, dseed), fcn.state_spec(cin, cout)
        gkeys = learner.grad_keys(spec)
        # the reference itself, fp32, two steps
        policy, target = ref_net(cin, cout, wseed), ref_net(cin, cout, wseed + 1000)
        policy.train()
        target.eval()
        opt = torch.optim.SGD(policy.parameters(), lr=cases.LR, momentum=cases.MOMENTUM, weight_decay=cases.WEIGHT_DECAY)
        p0 = {k: v.detach().clone() for k, v in policy.state_dict().items()}
        info_ref = [ref_train.train(cfg, policy, target, opt, batch, learner.apply_transform, cases.GAMMA)]
        p1_ref = {k: v.detach().clone() for k, v in policy.state_dict().items()}
        info_ref.append(ref_train.train(cfg, policy, target, opt, batch, learner.apply_transform, cases.GAMMA))
        # the oracle fp32 (bit-exact with the reference: gives access to the pre-clip gradient) and fp64
        st, tg = cases.oracle_state(cin, cout, wseed), cases.oracle_state(cin, cout, wseed + 1000)
        ex32 = {}
        i32 = learner.train_step(cfg, st, tg, spec, [None] * len(gkeys), batch, cases.GAMMA, cases.LR, cases.MOMENTUM, cases.WEIGHT_DECAY, extras=ex32)
        assert i32 == info_ref[0], (name, i32, info_ref[0])
        st64, tg64 = cases.oracle_state(cin, cout, wseed, torch.float64), cases.oracle_state(cin, cout, wseed + 1000, torch.float64)
        mom64, ex64 = [None] * len(gkeys), {}
        i64 = [learner.train_step(cfg, st64, tg64, spec, mom64, batch, cases.GAMMA, cases.LR, cases.MOMENTUM, cases.WEIGHT_DECAY,
                                  dtype=torch.float64, extras=ex64)]
        p1_64 = {k: st64[k].detach().clone() for k in gkeys}
        i64.append(learner.train_step(cfg, st64, tg64, spec, mom64, batch, cases.GAMMA, cases.LR, cases.MOMENTUM, cases.WEIGHT_DECAY,
                                      dtype=torch.float64))
        g64s, g32s, d64s, d32s = [], [], [], []
        for k in gkeys:
            idx = torch.tensor(cases.sample_indices(ex64['grads'][k].numel()))
            g64s.append(ex64['grads'][k].reshape(-1)[idx].numpy())
            g32s.append(ex32['grads'][k].double().reshape(-1)[idx].numpy())
            d64s.append((p1_64[k] - p0[k].double()).reshape(-1)[idx].numpy())
            d32s.append((p1_ref[k].double() - p0[k].double()).reshape(-1)[idx].numpy())
        g64s, g32s, d64s, d32s = (np.stack(a) for a in (g64s, g32s, d64s, d32s))
        rl2 = lambda a, b: float(np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum()))
        out[name + '.grad64'] = g64s
        out[name + '.dparam64'] = d64s
        out[name + '.total_norm64'] = np.array(ex64['total_norm'])
        out[name + '.loss64'] = np.array([i['loss'] for i in i64])
        out[name + '.td64'] = np.array([i['td_error'] for i in i64])
        out[name + '.ref_grad_err'] = np.array(rl2(g32s, g64s))
        out[name + '.ref_dparam_err'] = np.array(rl2(d32s, d64s))
        out[name + '.ref_loss_err'] = np.array([abs(info_ref[j]['loss'] - i64[j]['loss']) / abs(i64[j]['loss']) for j in range(2)])
        print('grad study %-10s ref fp32 vs fp64: sampled grad rel-L2 %.3g, sampled update rel-L2 %.3g, loss err %.2g / %.2g (step 1 / 2)'
              % (name, out[name + '.ref_grad_err'], out[name + '.ref_dparam_err'], out[name + '.ref_loss_err'][0], out[name + '.ref_loss_err'][1]), flush=True)
    np.savez_compressed(os.path.join(cases.GOLDEN_DIR, fname), **out)


def gen_bf16_calibration(train_cases=None, fname='bf16_calibration.npz'):
    """Fixture G8: how far the REFERENCE moves when its own modules run in bf16 -- torch.autocast('cpu', bfloat16) over
    the imported networks.FCN / train.train arithmetic (bf16 convolution operands and outputs, fp32 BatchNorm statistics and
    parameters: the mixed-precision recipe the opt-in `precision='bf16'` plans follow) -- measured against the fp64 oracle on the
    forward and train fixtures' own inputs.  The GPU tests hold the HIP bf16 path to a small multiple of THESE errors instead of an
    asserted constant."""
    from torch.nn.functional import smooth_l1_loss
    out = {}

    def rel(a, b):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        return float(np.abs(a - b).max() / np.abs(b).max())
    for name, cin, cout, B, wseed, dseed in (cases.FORWARD_CASES if train_cases is None else []):
        x = torch.cat([learner.apply_transform(s) for s in synth.make_states(B, cin, dseed)])
        for training in (False, True):
            st64 = cases.oracle_state(cin, cout, wseed, torch.float64)
            with torch.no_grad():
                q64 = fcn.fcn_forward(st64, x.double(), training)
            net = ref_net(cin, cout, wseed)
            net.train(training)
            with torch.no_grad(), torch.autocast('cpu', dtype=torch.bfloat16):
                q16 = net(x)
            out['%s.%s' % (name, 'train' if training else 'eval')] = rel(q16.float().numpy(), q64.numpy())
    for name, cin, cout, B, wseed, dseed in (train_cases or cases.TRAIN_CASES):
        cfg, batch, spec = cases.make_cfg(B), cases.make_batch(cin, cout, B, dseed), fcn.state_spec(cin, cout)
        st64, tg64 = cases.oracle_state(cin, cout, wseed, torch.float64), cases.oracle_state(cin, cout, wseed + 1000, torch.float64)
        ex64 = {}
        i64 = learner.train_step(cfg, st64, tg64, spec, [None] * len(learner.grad_keys(spec)), batch, cases.GAMMA, cases.LR,
                                 cases.MOMENTUM, cases.WEIGHT_DECAY, dtype=torch.float64, extras=ex64)
        policy, target = ref_net(cin, cout, wseed), ref_net(cin, cout, wseed + 1000)
        policy.train()
        target.eval()
        # train.py:109-132 under autocast (the reference's train() moves tensors itself; restated here so that autocast wraps the forwards)
        state_b = torch.cat([learner.apply_transform(s) for s in batch.state])
        act = torch.tensor(batch.action, dtype=torch.long)
        rew = torch.tensor(batch.reward, dtype=torch.float32)
        nf = torch.cat([learner.apply_transform(s) for s in batch.next_state if s is not None])
        mask = torch.tensor([s is not None for s in batch.next_state], dtype=torch.bool)
        with torch.autocast('cpu', dtype=torch.bfloat16):
            q = policy(state_b).float().view(B, -1).gather(1, act.unsqueeze(1)).squeeze(1)
            nsv = torch.zeros(B)
            with torch.no_grad():
                best = policy(nf).float().view(nf.size(0), -1).max(1)[1].view(-1, 1)
                nsv[mask] = target(nf).float().view(nf.size(0), -1).gather(1, best).view(-1)
        y = rew + cases.GAMMA * nsv
        loss = smooth_l1_loss(q, y)
        loss.backward()
        grads = {k: p.grad for k, p in policy.named_parameters() if p.grad is not None}
        num = sum(float((grads[k[len('module.'):] if k[len('module.'):] in grads else k].double() - ex64['grads'][k]).pow(2).sum())
                  for k in ex64['grads'] if (k in grads or k[len('module.'):] in grads))
        den = sum(float(ex64['grads'][k].pow(2).sum()) for k in ex64['grads'])
        out[name + '.loss'] = abs(float(loss) - i64['loss']) / abs(i64['loss'])
        out[name + '.td_error'] = abs(float(torch.abs(q - y).mean()) - i64['td_error']) / abs(i64['td_error'])
        out[name + '.grad'] = (num / den) ** 0.5
    np.savez(os.path.join(cases.GOLDEN_DIR, fname), **{k: np.array(v) for k, v in out.items()})
    for k, v in out.items():
        print('bf16 calibration (reference under torch.autocast bf16 vs fp64)  %-28s %.4g' % (k, v))


def gen_intention():
    """train.train_intention (train.py:143-158) run twice on the reference vs the oracle restatement, bit-exact."""
    for name, cin_full, B, wseed, dseed in cases.INTENTION_CASES:
        cin = cin_full - 1
        batch = cases.make_batch(cin_full, 1, B, dseed)
        spec = fcn.state_spec(cin, 1)
        net = ref_net(cin, 1, wseed)
        net.train()
        opt = torch.optim.SGD(net.parameters(), lr=cases.LR, momentum=cases.MOMENTUM, weight_decay=cases.WEIGHT_DECAY)   # train.py:190
        info_ref = [ref_train.train_intention(net, opt, batch, learner.apply_transform) for _ in range(2)]
        st = cases.oracle_state(cin, 1, wseed)
        mom = [None] * len(learner.grad_keys(spec))
        extras = [{}, {}]
        info_or = [learner.train_intention_step(st, spec, mom, batch, cases.LR, cases.MOMENTUM, cases.WEIGHT_DECAY,
                                                extras=extras[i]) for i in range(2)]
        assert info_or == info_ref, (name, info_or, info_ref)
        for k, v in net.state_dict().items():
            assert_same(st[k], v, name + ' post-step ' + k)
        st64 = cases.oracle_state(cin, 1, wseed, torch.float64)
        ex64 = {}
        info64 = learner.train_intention_step(st64, spec, [None] * len(mom), batch, cases.LR, cases.MOMENTUM,
                                              cases.WEIGHT_DECAY, dtype=torch.float64, extras=ex64)
        g32, g64 = cases.grad_summary(extras[0]['grads']), cases.grad_summary(ex64['grads'])
        num = sum(float((extras[0]['grads'][k].double() - ex64['grads'][k]).pow(2).sum()) for k in g32)
        den = sum(float(ex64['grads'][k].pow(2).sum()) for k in g32)
        np.savez(os.path.join(cases.GOLDEN_DIR, name + '.npz'),
                 loss_intention=np.array([i['loss_intention'] for i in info_ref]), loss64=np.array(info64['loss_intention']),
                 output_step1=extras[0]['output'].numpy(), grad_keys=np.array(list(g32.keys())),
                 grad32=np.stack(list(g32.values())), grad64=np.stack(list(g64.values())),
                 ref_fp32_grad_relerr=np.array((num / den) ** 0.5),
                 param_summary_after2=cases.param_summary(st, spec),
                 bn_buffers_after2=cases.bn_buffer_vector(st).astype(np.float32))
        print('intention case', name, 'oracle == reference (bit-exact, 2 steps); ref fp32 grad rel err vs fp64 = %.3g'
              % ((num / den) ** 0.5))


def gen_intention_step():
    """DQNIntentionPolicy.step (policies.py:119-146, through step_intention :97-117) of the REFERENCE's own class, compared bit-exactly
    with the oracle restatement, then stored."""
    cin = 5
    cfg = types.SimpleNamespace(robot_config=[{'lifting_robot': 1}, {'pushing_robot': 1}], num_input_channels=cin,
                                final_exploration=0.01, checkpoint_path=None)
    wseeds = [71, 72, 73, 74]
    ref = ref_policy('DQNIntentionPolicy', cfg, [(cin, 2, 71), (cin, 1, 72), (cin - 1, 1, 73), (cin - 1, 1, 74)], 9)
    seeds = iter(wseeds)
    pol = opolicy.DQNIntentionPolicy(cfg, lambda ci, co: cases.oracle_state(ci, co, next(seeds)), train=False, random_seed=9)
    s = synth.make_states(2, cin - 1, 81)
    a_ref, info_ref = ref.step([[s[0]], [s[1]]], exploration_eps=0.0, debug=True)
    a, info = pol.step([[s[0]], [s[1]]], exploration_eps=0.0, debug=True)
    assert a == a_ref, (a, a_ref)
    for key in ('output_intention', 'state_intention', 'output'):
        for i in range(2):
            assert_same(info[key][i][0], info_ref[key][i][0], 'intention step %s[%d]' % (key, i))
    # the non-debug call and a stochastic one consume the python RNG alike
    random.seed(123)
    r1 = [ref.step([[s[0]], [s[1]]], exploration_eps=0.5) for _ in range(3)]
    random.seed(123)
    o1 = [pol.step([[s[0]], [s[1]]], exploration_eps=0.5) for _ in range(3)]
    assert r1 == o1, (r1, o1)
    np.savez(os.path.join(cases.GOLDEN_DIR, 'intention_step.npz'), actions=np.array([a_ref[0][0], a_ref[1][0]], dtype=np.int64),
             output_intention=np.stack([info_ref['output_intention'][0][0], info_ref['output_intention'][1][0]]),
             state_intention=np.stack([info_ref['state_intention'][0][0], info_ref['state_intention'][1][0]]),
             q0=info_ref['output'][0][0], q1=info_ref['output'][1][0],
             eps_half_actions=np.array([[x[0][0], x[1][0]] for x in r1], dtype=np.int64))
    print('intention policy.step case: oracle == reference policies.DQNIntentionPolicy (bit-exact); saved')


def gen_tracker():
    """train.TransitionTracker (train.py:47-68) on a scripted episode; the fixture pins simq.TransitionTracker."""
    rows = cases.run_tracker(ref_train.TransitionTracker)
    assert sum(len(r) for r in rows) > 20
    np.savez(os.path.join(cases.GOLDEN_DIR, 'tracker.npz'), **{'buffer%d' % i: r for i, r in enumerate(rows)})
    print('tracker case saved: %s transitions per buffer' % [len(r) for r in rows])


def gen_checkpoint():
    """A checkpoint_*.pth.tar as train.py:324-334 writes it, produced with the reference's own ReplayBuffer / Transition
    classes (pickled under the module name `train`) and a fresh torch SGD over the reference FCN (train.py:186)."""
    buf = ref_train.ReplayBuffer(cases.CKPT_CAPACITY)
    for t in cases.checkpoint_transitions():
        buf.push(*t)
    assert type(buf).__module__ == 'train' and type(buf.buffer[0]).__module__ == 'train'
    net = ref_net(cases.CKPT_CIN, 2, cases.CKPT_SEED)
    opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    path = os.path.join(cases.GOLDEN_DIR, 'ref_checkpoint.pth.tar')
    torch.save({'timestep': 7, 'episode': 2, 'optimizers': [opt.state_dict()], 'replay_buffers': [buf]}, path)
    print('checkpoint fixture saved: %d bytes' % os.path.getsize(path))


def gen_sampler():
    out = {}
    for n, B, seed in cases.SAMPLER_CASES:
        ref_buf, or_buf = ref_train.ReplayBuffer(n), learner.ReplayBuffer(n)
        for i in range(n + 3):      # wrap the ring by 3
            ref_buf.push(i, i, float(i), None)
            or_buf.push(i, i, float(i), None)
        assert [t.state for t in ref_buf.buffer] == [t.state for t in or_buf.buffer]
        assert ref_buf.position == or_buf.position
        random.seed(seed)
        a = ref_buf.sample(B)
        random.seed(seed)
        b = or_buf.sample(B)
        assert a == b
        random.seed(seed)
        idx = random.sample(range(n), B)   # index-driven form used by the device ring
        assert [ref_buf.buffer[i].state for i in idx] == list(a.state)
        out['n%d_b%d_s%d' % (n, B, seed)] = np.array(a.state, dtype=np.int64)
        out['idx_n%d_b%d_s%d' % (n, B, seed)] = np.array(idx, dtype=np.int64)
    np.savez(os.path.join(cases.GOLDEN_DIR, 'sampler.npz'), **out)
    print('sampler cases: oracle == reference; random.sample(range(n),B) reproduces the picks')


def gen_step():
    """DQNPolicy.step (policies.py:47-74) of the REFERENCE's own class (import_ref_policies) on seeded weights and states, compared
    bit-exactly with the oracle restatement -- actions under exploration_eps 0 / 0.5 / 1 (the random.random() / randrange draw
    order of policies.py:61-62), None entries, the debug outputs -- then stored."""
    cin, wseed = 4, 51
    cfg = types.SimpleNamespace(robot_config=[{'lifting_robot': 2}, {'pushing_robot': 1}], num_input_channels=cin,
                                final_exploration=0.01, checkpoint_path=None)
    ref = ref_policy('DQNPolicy', cfg, [(cin, 2, wseed), (cin, 1, wseed + 1)], 5)
    s = synth.make_states(3, cin, 61)
    state = [[s[0], None], [s[1]]]
    acts_ref = [ref.step(state, exploration_eps=eps) for eps in (0.0, 0.5, 1.0, 0.5)]
    a_ref, info_ref = ref.step([[None, s[2]], [None]], exploration_eps=0.0, debug=True)
    seeds = iter([wseed, wseed + 1])
    pol = opolicy.DQNPolicy(cfg, lambda ci, co: cases.oracle_state(ci, co, next(seeds)), train=False, random_seed=5)
    acts = [pol.step(state, exploration_eps=eps) for eps in (0.0, 0.5, 1.0, 0.5)]
    a, info = pol.step([[None, s[2]], [None]], exploration_eps=0.0, debug=True)
    assert acts == acts_ref and a == a_ref, (acts, acts_ref, a, a_ref)
    assert_same(info['output'][0][1], info_ref['output'][0][1], 'policy step debug output')
    assert info_ref['output'][0][0] is None and info_ref['output'][1][0] is None
    np.savez(os.path.join(cases.GOLDEN_DIR, 'policy_step.npz'),
             actions=np.array([[x[0][0], x[1][0]] for x in acts_ref], dtype=np.int64),
             debug_action=np.array([a_ref[0][1]], dtype=np.int64), debug_output=info_ref['output'][0][1])
    print('policy.step case: oracle == reference policies.DQNPolicy (bit-exact); saved')


if __name__ == '__main__':
    torch.manual_seed(0)
    os.makedirs(cases.GOLDEN_DIR, exist_ok=True)
    gens = {'sampler': gen_sampler, 'forward': gen_forward, 'step': gen_step, 'train': gen_train,
            'intention': gen_intention, 'intention_step': gen_intention_step,
            'train_full': lambda: gen_train(cases.TRAIN_CASES_FULL),
            'train_cin': lambda: (gen_train(cases.TRAIN_CASES_CIN, keep_output=False), gen_bf16_calibration(cases.TRAIN_CASES_CIN, 'bf16_calibration_cin.npz')), 'train_sized': gen_train_sized, 'dense_grad': gen_dense_grad, 'tracker': gen_tracker, 'checkpoint': gen_checkpoint,
            'dp': gen_dp, 'dp_literal': gen_dp_literal, 'bf16_calibration': gen_bf16_calibration, 'bf16_points': gen_bf16_points,
            'grad_study': gen_grad_study,
            'grad_study_b64': lambda: gen_grad_study(cases.GRAD_STUDY_B64_CASES, 'grad_study_b64.npz'),
            'grad_study_b32': lambda: gen_grad_study(cases.GRAD_STUDY_B32_CASES, 'grad_study_b32.npz'),
            'grad_study_b128': lambda: gen_grad_study(cases.GRAD_STUDY_B128_CASES, 'grad_study_b128.npz')}
    for which in (sys.argv[1:] or list(gens)):     # e.g. `python -m oracle.gen_golden intention` regenerates one family
        gens[which]()
