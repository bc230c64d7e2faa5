#!/usr/bin/env python3
"""GPU check of the data-parallel TD step: run under
   python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tests/diag/dp_check.py [--backend gloo]
Every rank runs simq.learner.train_step on its shard (per-rank BN statistics, 1/global_batch loss scaling, one flat
all-reduce); rank 0 compares the all-reduced gradient, the loss and rank-0's running statistics with the single-process
sharded emulation of the oracle (oracle.learner.dp_emulation)."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'spatial-intention-maps_amd')]
import numpy as np
import torch
import torch.distributed as dist

ap = argparse.ArgumentParser(); ap.add_argument('--backend', default='nccl'); args = ap.parse_args()
rank, world, local = int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), int(os.environ.get('LOCAL_RANK', 0))
ndev = torch.cuda.device_count()
torch.cuda.set_device(local % ndev)
dev = torch.device('cuda', local % ndev)
if args.backend == 'nccl':
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
else:
    dist.init_process_group('gloo', rank=rank, world_size=world)
import simq
from simq import dist as sdist, synth
from simq.learner import train_step, assemble_batch
from oracle import cases, fcn as ofcn, learner as ol

CIN, COUT, GB, WSEED, DSEED = 4, 2, 8, 71, 72
cfg = cases.make_cfg(GB); spec = ofcn.state_spec(CIN, COUT)
batch = cases.make_batch(CIN, COUT, GB, DSEED)
lo, hi = sdist.shard_bounds(GB, world, rank)
shard = ol.Transition(*[f[lo:hi] for f in batch])
policy, target = simq.FCN(CIN, COUT, device=dev), simq.FCN(CIN, COUT, device=dev)
policy.load_state_dict(ofcn.state_from_numpy(synth.make_state_dict(CIN, COUT, WSEED))); policy.train()
target.load_state_dict(ofcn.state_from_numpy(synth.make_state_dict(CIN, COUT, WSEED + 1))); target.eval()
p_before = policy.flat_params.clone()
info = train_step(policy, target, assemble_batch(shard, dev), cases.GAMMA, hi - lo, cases.LR, cases.MOMENTUM, cases.WEIGHT_DECAY,
                  cases.CLIP, use_double_dqn=True, process_group=dist.group.WORLD, global_batch=GB)
# every rank must end with identical parameters
chk = torch.stack([policy.flat_params.double().sum(), policy.flat_params.double().pow(2).sum()])
all_chk = [torch.zeros_like(chk) for _ in range(world)]
dist.all_gather(all_chk, chk)
assert all(torch.equal(all_chk[0], c) for c in all_chk), 'ranks diverged: %r' % (all_chk,)
if rank == 0:
    st, tg = cases.oracle_state(CIN, COUT, WSEED), cases.oracle_state(CIN, COUT, WSEED + 1)
    total, loss, td = ol.dp_emulation(cfg, st, tg, spec, batch, world, cases.GAMMA)
    tn = float(policy._simq_opt_state.total_norm.item()); coef = min(1.0, cases.CLIP / (tn + 1e-6))
    g = policy.flat_grads.cpu() / coef
    # oracle flat gradient is in reference (OIHW) layout, tensor by tensor
    off = 0; num = den = 0.0
    for (name, _, kind), (o, n, shape) in zip(policy._param_names, policy._grad_views):
        t = g[o:o + n].view(shape)
        if len(shape) == 4: t = t.permute(0, 3, 1, 2)
        r = total[off:off + n].view(t.shape); off += n
        num += float((t.double() - r.double()).pow(2).sum()); den += float(r.double().pow(2).sum())
    err = (num / den) ** 0.5
    sd = policy.state_dict()
    bn = np.concatenate([sd[k].cpu().double().numpy().ravel() for k in sd if k.endswith('running_mean') or k.endswith('running_var')])
    bn_err = np.abs(bn - cases.bn_buffer_vector(st)).max() / np.abs(cases.bn_buffer_vector(st)).max()
    print('dp_check world=%d backend=%s: loss hip %.6f oracle %.6f | td hip %.6f oracle %.6f | grad rel-L2 err %.3g | total_norm %.4f vs %.4f | rank0 BN err %.3g'
          % (world, args.backend, info['loss'], loss, info['td_error'], td, err, tn, float(total.norm()), bn_err))
    assert abs(info['loss'] - loss) <= 1e-4 * abs(loss) and abs(info['td_error'] - td) <= 1e-4 * abs(td)
    assert err < 5e-2 and bn_err < 1e-4
    print('dp_check OK')
dist.destroy_process_group()
