#!/usr/bin/env python3
"""bench.py -- Q-map transitions/sec of the spatial-action-map DQN training step on MI355X.

A "step" is ONE pass of the reference's training loop body (train.py:252-258) over the robot groups of the workload: for
every group, one minibatch draw (train.py:256) and one full train() call (train.py:108-141) -- replay index sampling + HBM
gather, policy forward (train-mode BN), double-DQN next-state forwards (policy train-mode no-grad + target eval), TD target
+ Huber, backward, [gradient all-reduce], global-norm clip, momentum SGD and the two scalar read-backs the reference does
(.item()).  That is M2 of SURVEY 8d (65.0 GFLOP/transition); nothing is skipped.  M1 -- the literal "fwd+bwd" of
BASELINE.json's metric text: policy forward + gather + Huber + backward only -- is timed beside it (`value_fwd_bwd_only`).

Workloads = BASELINE.json's configs (`--workload`, default `auto` = by GPU count):
  configs1   lifting_1-small_empty:           Cin 4 -> Cout 2,                 global minibatch   32, fp32   (auto: N=1, N=2 sharded)
  configs2   lifting_4-small_divider:         Cin 5 -> Cout 2,                 global minibatch  128, bf16   (N=1: reported beside configs1)
  configs3   lifting_2_pushing_2-large_empty: two nets, Cin 5 -> Cout 2 and 1, global minibatch  256 PER NET, fp32  (auto: N=4)
  configs4   lifting_4-large_empty:           Cin 5 -> Cout 2,                 global minibatch 1024, bf16   (auto: N=8)
  weak32     configs1's net at 32 transitions PER GPU (weak scaling; auto for every other N, and timed as a second leg of
             every N>1 run so that a like-for-like 1 -> N curve exists next to the config-faithful `value`)
The global minibatch is sharded over the ranks (contiguous slices of the commonly drawn indices, DataParallel's scatter,
policies.py:39): per-rank BatchNorm statistics, gradient all-reduce in two buckets overlapped with the backward walk.

N GPUs: one process per GPU.  Started under torch.distributed.run (the driver's form) the ranks are taken from the
environment; started bare (`python bench.py --gpus N`) the script launches its own N ranks through torch.distributed.run
on 127.0.0.1 and exits with their status.

Prints ONE JSON line on rank 0 (driver contract) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import ctypes
import json
import os
import random
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, 'spatial-intention-maps_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

# multi-process GPU work on this ROCm stack needs dmabuf IPC (RCCL / cross-process tensor sharing); keep whatever the
# launcher exported, default to the supported mode otherwise -- must be in the environment before the HIP runtime starts
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

GAMMA, LR, MOMENTUM, WD, CLIP = 0.75, 0.01, 0.9, 1e-4, 100.0
REPLAY_ITEMS = 10000                     # replay_buffer_size of the reference's experiment configs (…-base.yml:29), resident in HBM per rank and net
PEAK_FP32_MFMA_TFLOPS = 157.3            # MI355X_MICROARCH.md: v_mfma_f32_*_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0           # dense bf16 MFMA (not the 2:1-sparse marketing figure)
PEAK_HBM_GBS = 8000.0

# per transition, SURVEY 8d [probe-derived]: (M1 fwd+bwd, M2 full train()) for Cin 4 / Cin 5 (the first convolution is the only
# layer that sees Cin; Cout only moves the 0.6-MMAC last layer)
FLOPS = {4: (38.963e9, 64.976e9), 5: (38.99e9, 65.03e9)}

WORKLOADS = {
    'configs1': dict(config='BASELINE configs[1] lifting_1-small_empty', nets=[(4, 2)], global_batch=32, precision='fp32'),
    'configs2': dict(config='BASELINE configs[2] lifting_4-small_divider', nets=[(5, 2)], global_batch=128, precision='bf16'),
    'configs3': dict(config='BASELINE configs[3] lifting_2_pushing_2-large_empty (lifting Cout=2 + pushing Cout=1, train.py:255-257)',
                     nets=[(5, 2), (5, 1)], global_batch=256, precision='fp32'),
    'configs4': dict(config='BASELINE configs[4] lifting_4-large_empty', nets=[(5, 2)], global_batch=1024, precision='bf16'),
    'weak32': dict(config="BASELINE configs[1]'s net at 32 transitions per GPU (weak scaling)", nets=[(4, 2)], per_gpu=32, precision='fp32'),
}
AUTO = {1: 'configs1', 2: 'configs1', 4: 'configs3', 8: 'configs4'}
DTYPE_NAMES = {'fp32': 'f32', 'bf16x3': 'bf16x3 (split-bf16 operands, 3 MFMA products, f32 accumulate)', 'bf16': 'bf16 (f32 accumulate / BN / optimiser)'}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', default='auto', choices=['auto'] + sorted(WORKLOADS),
                    help='BASELINE config to run (auto: N=1 configs1, N=2 configs1 sharded, N=4 configs3, N=8 configs4, otherwise weak32)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-m1', action='store_true', help='skip the forward+backward-only (M1) leg (profiling runs: every launch then belongs to a full step)')
    ap.add_argument('--no-extras', action='store_true', help='skip the second leg (N=1: configs[2] bf16 beside configs[1]; N>1: weak32 beside the config)')
    ap.add_argument('--precision', default=None, choices=['fp32', 'bf16x3', 'bf16'], help="override the workload's arithmetic (tools only)")
    ap.add_argument('--batch', type=int, default=None, help='override: transitions per GPU (and net) per step (tools only)')
    ap.add_argument('--cin', type=int, default=None, help='override: input channels of a single Cout=2 net (tools only)')
    ap.add_argument('--plan-option', action='append', default=[], metavar='NAME=INT', help='tools only: a simq_plan_options override for every plan of the run (A/B), e.g. deterministic=1')
    ap.add_argument('--no-overlap', action='store_true', help='tools only: the target-net forward on the main stream instead of the side stream (A/B of the two-stream overlap)')
    ap.add_argument('--wgrad-xcd-group', type=int, default=None, choices=[0, 1, 2], help='tools only: simq_plan_options.wgrad_xcd_group of every plan, A/B')
    ap.add_argument('--wgrad-overlap', type=int, default=None, choices=[0, 1, 2, 3, 4], help='tools only: simq_plan_options.wgrad_overlap of every plan, A/B')
    ap.add_argument('--no-upload-stream', action='store_true', help='tools only: the per-batch index upload on the consuming stream (A/B)')
    ap.add_argument('--fwd-overlap', type=int, default=None, choices=[0, 1, 2], help='tools only: simq_plan_options.fwd_overlap of every plan, A/B (2 = default: three forwards side by side)')
    ap.add_argument('--plane-xcd', type=int, default=None, choices=[0, 1], help='tools only: simq_plan_options.plane_xcd of every plan (batched GEMM planes per XCD), A/B')
    ap.add_argument('--early-target', type=int, default=None, choices=[0, 1], help='tools only: StepOptions.early_target_forward of every learner (the target-net forward of a step beside the previous step), A/B')
    ap.add_argument('--group-streams', type=int, default=0, choices=[0, 1], help='tools only: multi-net workloads (configs3): 1 = every robot group issues its step on a launch stream of its own (the groups run side by side), 0 = one after the other on one stream (the reference\'s order; measured level with 1 once a step\'s streams are chosen by test: profiles/r06_ab_concurrent_groups.txt), A/B')
    ap.add_argument('--group-issue', default='defer', choices=['defer', 'stagger'], help="tools only: with --group-streams 1: 'defer' = enqueue every group's step, then wait for the losses (simq.train_groups); 'stagger' = wait for each group's loss before enqueueing the next group's step")
    ap.add_argument('--third-leg-split', type=int, default=0, choices=[0, 1], help='tools only: simq_plan_options.gemm_split of the third leg (1: the headline form again -- is a THIRD leg slower as such?)')
    ap.add_argument('--stream-skew', type=int, default=0, help='tools only: take this many streams from torch\'s stream pool before the learners do (which hardware queue a stream shares follows the order of creation: A/B)')
    ap.add_argument('--single-rank-comm', action='store_true', help='tools only, N = 1: run the DATA-PARALLEL form of the step (backward phases, two gradient buckets all-reduced on a 1-rank RCCL communicator) on the one GPU a lease has')
    ap.add_argument('--replay', type=int, default=REPLAY_ITEMS, help='transitions resident in the HBM replay ring per net')
    ap.add_argument('--sustained-seconds', type=float, default=3.0, help='length of the sustained leg behind the timed window (0 = skip)')
    ap.add_argument('--watchdog-seconds', type=int, default=120, help='multi-rank runs: abort with a diagnosis when a phase makes no progress for this long')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help='torch.distributed backend for --gpus > 1 (nccl == RCCL; gloo only to debug the rank logic on one GPU)')
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, the form the reference's
    single-process nn.DataParallel, policies.py:39, takes here) and hand their exit status back."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('OMP_NUM_THREADS', '8')
    return subprocess.call(cmd, env=env)


class Watchdog:
    """A collective that never completes would hold the whole GPU lease: every phase of a multi-rank run is bracketed by a timer
    that, on expiry, prints WHICH rank is stuck WHERE (phase, step, libsimq's count of enqueued / completed collectives) and ends the
    process with a non-zero status, so the launcher tears the other ranks down."""

    def __init__(self, rank, seconds, enabled, progress=None):
        self.rank, self.seconds, self.enabled, self.progress = rank, seconds, enabled, progress
        self.label, self.deadline, self.thread = None, None, None

    def _fire(self):
        extra = ''
        if self.progress is not None:
            try:
                extra = ' | ' + self.progress()
            except Exception as ex:        # noqa: BLE001
                extra = ' | (no progress report: %r)' % (ex,)
        print('bench WATCHDOG: rank %d made no progress for %d s in [%s]%s -- exiting' % (self.rank, self.seconds, self.label, extra),
              file=sys.stderr, flush=True)
        os._exit(3)

    def _watch(self):
        while True:
            time.sleep(1.0)
            d = self.deadline
            if d is not None and time.monotonic() > d:
                self._fire()

    def arm(self, label):
        """(Re)start the countdown: two attribute stores on the timed path -- ONE watcher thread polls the deadline once a second."""
        if not self.enabled:
            return
        self.label = label
        self.deadline = time.monotonic() + self.seconds
        if self.thread is None:
            import threading
            self.thread = threading.Thread(target=self._watch, daemon=True)
            self.thread.start()

    def disarm(self):
        self.deadline = None


def cpu_baseline(cin, cout, batch):
    """The oracle (CPU restatement, pinned bit-exact to the reference in the build container) running the SAME step on this
    box's host cores.  Bounded sample (~20-30 s): a sweep over thread counts (1 warm-up + 1 timed train() call each, at the GPU
    workload's batch size capped at 32 -- the rate is flat in the batch), then THREE more timed calls at the winning
    count; `value` is the rate over those three.  A count whose warm-up call is already 3x slower than the best so far is not
    timed further (oversubscribing a 256-thread host made MKL-DNN 30x slower)."""
    import torch
    from oracle import cases, fcn as ofcn, learner as olearner
    from simq import synth
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    sweep = sorted({max(1, min(avail, n)) for n in (32, 64, 128)})
    cfg = cases.make_cfg(batch)
    spec = ofcn.state_spec(cin, cout)
    trs = synth.make_transitions(batch, cin, cout, 3, terminal_frac=0.1)
    b = olearner.Transition(*zip(*trs))

    def fresh():
        return (ofcn.state_from_numpy(synth.make_state_dict(cin, cout, 1)), ofcn.state_from_numpy(synth.make_state_dict(cin, cout, 2)),
                [None] * len(olearner.grad_keys(spec)))

    def call(st, tg, mom):
        t0 = time.perf_counter()
        olearner.train_step(cfg, st, tg, spec, mom, b, GAMMA, LR, MOMENTUM, WD)
        return time.perf_counter() - t0

    results, best_call = [], None
    t_all = time.perf_counter()
    for nthreads in sweep:
        if time.perf_counter() - t_all > 20:
            break
        torch.set_num_threads(nthreads)
        st, tg, mom = fresh()
        warm = call(st, tg, mom)
        if best_call is not None and warm > 3 * best_call:
            results.append((batch / warm, nthreads))
            continue
        dt = call(st, tg, mom)
        best_call = dt if best_call is None else min(best_call, dt)
        results.append((batch / dt, nthreads))
    rate_sweep, nthreads = max(results)
    torch.set_num_threads(nthreads)
    st, tg, mom = fresh()
    warm = call(st, tg, mom)
    calls = [call(st, tg, mom) for _ in range(3)]
    rate = 3 * batch / sum(calls)
    return {'value': round(rate, 3), 'unit': 'transitions/s', 'cores': nthreads, 'kind': 'port', 'timed_calls': 3,
            'sample': 'oracle train_step (== reference train.py:108-141 on torch-CPU/MKL-DNN fp32), Cin %d, batch %d: thread-count sweep (1 warm-up + 1 timed '
                      'call each: %s tr/s), then 1 warm-up (%.1f s) + 3 timed calls (%s s) at the best count = `cores`; host reports %d CPUs (%d usable)'
                      % (cin, batch, ', '.join('%d thr: %.1f' % (n, r) for r, n in results), warm, ' / '.join('%.2f' % c for c in calls),
                         os.cpu_count() or 0, avail)}


PMC_TRAFFIC = {   # precision -> (committed rocprofv3 PMC summaries newest first, kernel whose bytes per launch `roofline.traffic` quotes)
    'fp32': (('r06_pmc_traffic.json',), 'gemm_split3_kernel'),   # (name prefix; round 6: the transform-domain GEMMs on the bf16 matrix cores)
    'fp32_mfma': (('r06_pmc_traffic_fp32_mfma.json', 'r05_pmc_traffic.json', 'r04_pmc_traffic.json', 'r03_pmc_traffic.json', 'r02_pmc_traffic.json'), 'igemm_conv_kernel<64, 64, true, true'),
    'bf16': (('r06_pmc_traffic_bf16_b128.json', 'r05_pmc_traffic_bf16_b128.json', 'r04_pmc_traffic_bf16_b128.json', 'r03_pmc_traffic_bf16_b128.json', 'r02_pmc_traffic_bf16_b128.json'), None),     # None: the kernel named by DOMINANT_BF16 below
}
DOMINANT_BF16 = 'igemm_bf16_img_kernel'      # name prefix of the bf16 leg's dominant kernel in the rocprofv3 summaries


def pmc_traffic(precision):
    """(HBM-side bytes per launch of the dominant kernel, the file they were read from): FETCH_SIZE x2 + WRITE_SIZE, KiB -> bytes,
    from the committed rocprofv3 PMC passes over this same command (tools/pmc_traffic.py).  PMC counters cannot be read from inside
    the timed process, so this is a RECORDED number -- `traffic_source` in the line names the file -- or (None, None)."""
    if precision not in PMC_TRAFFIC:
        return None, None
    files, kernel = PMC_TRAFFIC[precision]
    for f in files:
        try:
            t = json.load(open(os.path.join(ROOT, 'profiles', f)))
        except (OSError, ValueError):
            continue
        hits = [v for k, v in t.items() if kernel is not None and k.startswith(kernel)]
        if hits:
            return round(max(hits, key=lambda v: v['launches'])['bytes_per_launch']), 'profiles/' + f
        cand = [(v['launches'] * v['bytes_per_launch'], v) for k, v in t.items() if k.startswith(DOMINANT_BF16) or k.startswith('igemm_bf16_dma_kernel')]
        if kernel is None and cand:
            return round(max(cand, key=lambda kv: kv[0])[1]['bytes_per_launch']), 'profiles/' + f
    return None, None


KERNEL_NAMES = {
    'fp32': 'gemm_split3_kernel (batched transform-domain GEMM of the Winograd layers -- 128->256, 256- and 512-channel 3x3 convolutions: 16 GEMMs per '
            'launch for the grad-mode forward in F(2x2,3x3), 36 for the no-grad forwards, dgrads and weight gradients in F(4x4,3x3) -- on the bf16 '
            'matrix cores: both fp32 operands split exactly into three bf16 pieces while staged, six v_mfma_f32_32x32x16_bf16 partial products per '
            'fp32 product, fp32 accumulate; achieved = EXECUTED bf16 flops (6 x the fp32 contraction\'s) / time, priced against the dense bf16 peak)',
    'fp32_mfma': 'igemm_conv_kernel<64,64,true,true> (the same batched transform-domain GEMMs on v_mfma_f32_16x16x4_f32, simq_plan_options.gemm_split = 0: '
                 'the form of rounds 1-5; achieved = EXECUTED fp32 flops / time)',
    'bf16x3': 'igemm_bf16_kernel<NP=2> (split-bf16 implicit GEMM, 3 x v_mfma_f32_16x16x32_bf16 per product; '
              'achieved counts ALGORITHMIC flops, matrix-core work is 3x that)',
    'bf16': 'igemm_bf16_img_kernel (image-tile 3x3 convolution: one 24x24 map x 128 channels per block, halo patch in LDS, ping-pong wave '
            'groups, v_mfma_f32_16x16x32_bf16): forward + dgrad of the 256- and 512-channel 3x3 convolutions; the 288-row LDS-DMA kernels of '
            'the other layers are counted under all_gemm_tiles)'}


def resolve_workload(args, world):
    name = args.workload if args.workload != 'auto' else AUTO.get(world, 'weak32')
    w = dict(WORKLOADS[name], name=name)
    if args.cin is not None:
        w['nets'] = [(args.cin, 2)]
        w['config'] += ' [--cin %d]' % args.cin
    if args.precision is not None:
        w['precision'] = args.precision
    if args.batch is not None:
        w.pop('global_batch', None)
        w['per_gpu'] = args.batch
        w['config'] += ' [--batch %d per GPU]' % args.batch
    if 'per_gpu' in w:
        w['global_batch'] = w['per_gpu'] * world
        w['scaling'] = 'weak'
    else:
        if w['global_batch'] % world:
            sys.exit('bench.py: %s has a global minibatch of %d, not divisible by %d ranks' % (name, w['global_batch'], world))
        w['per_gpu'] = w['global_batch'] // world
        w['scaling'] = 'strong'          # the config fixes the TOTAL minibatch; more GPUs shard it
    return w


def main():
    args = parse()
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))
    import numpy as np
    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    args.gpus = world
    if not torch.cuda.is_available():
        sys.exit('bench.py needs an MI355X (no GPU visible); there is no CPU product path')
    ndev = torch.cuda.device_count()
    if world > 1 and args.backend == 'nccl' and ndev < world:
        sys.exit('bench.py: %d ranks need %d GPUs, only %d visible (--backend gloo stacks ranks on one GPU to debug the rank logic)' % (world, world, ndev))
    local_dev = local_rank % ndev       # (gloo debugging may stack ranks on one GPU)
    torch.cuda.set_device(local_dev)
    dev = torch.device('cuda', local_dev)
    pg = None
    if world > 1 or args.single_rank_comm:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29541')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        pg = dist.group.WORLD

    _skew = [torch.cuda.Stream(dev) for _ in range(args.stream_skew)]
    import simq
    from simq import dist as sdist, synth
    from simq._lib import MODE_TRAIN, lib, ptr, stream_ptr
    from simq.learner import StepOptions, _opt_state, learner_streams, train_step
    # A/B switches are plan options (include/simq.h simq_plan_options): the library has no process-global state to flip
    plan_opts = {kv.split('=')[0]: int(kv.split('=')[1]) for kv in args.plan_option}
    for name in ('plane_xcd', 'fwd_overlap', 'wgrad_overlap', 'wgrad_xcd_group'):
        if getattr(args, name) is not None:
            plan_opts[name] = getattr(args, name)
    # ... and the host side has none either (round 6): how a learner issues its step is a StepOptions value handed to every train_step call,
    # where a ring runs its copies and gathers is an option of the ring
    step_opts = StepOptions(overlap_target_forward=not args.no_overlap,
                            early_target_forward=True if args.early_target is None else bool(args.early_target))
    ring_upload_stream = not args.no_upload_stream
    if lib.build_flags != 0:
        sys.exit('bench.py: %s is the ablation build (simq_build_flags = %d); the benchmark runs the product library only' % (lib.path, lib.build_flags))

    # gradient exchange: libsimq's own RCCL communicator (simq_comm_*: the data-parallel step is then ONE library call, buckets on
    # the communicator's stream); checked once against torch.distributed's all-reduce.  If RCCL cannot be initialised through
    # the library on this node the step uses torch.distributed's RCCL collectives around the backward phases -- still
    # RCCL over xGMI, never a CPU path.  Comm's construction fails on EVERY rank or on none (simq.dist.Comm), and the check below
    # is agreed with a MIN all-reduce, so the ranks cannot end up on different transports.
    comm, transport, comm_world = None, 'none (single GPU)', None
    if pg is not None:
        transport = 'torch.distributed (%s)' % args.backend
        if args.backend == 'nccl' and os.environ.get('SIMQ_BENCH_COMM', '1') != '0':
            boot = Watchdog(rank, args.watchdog_seconds, args.watchdog_seconds > 0)
            boot.arm('simq_comm construction (RCCL identifier broadcast + ncclCommInitRank)')
            try:
                comm = sdist.Comm(pg, dev)
            except Exception as ex:            # noqa: BLE001  (collective failure: every rank is here)
                if rank == 0:
                    print('bench: simq_comm unavailable (%r); using torch.distributed collectives' % (ex,), file=sys.stderr)
                comm = None
            boot.arm('simq_comm probe all-reduce against torch.distributed')
            if comm is not None:
                probe = torch.arange(1024, dtype=torch.float32, device=dev) * (rank + 1)
                want = probe.clone()
                torch.distributed.all_reduce(want, group=pg)
                # two RCCL communicators live in this process (torch.distributed's and libsimq's): never let collectives of both be
                # in flight at once -- ranks may schedule them in different orders and wait for each other forever
                torch.cuda.synchronize(dev)
                good = 1.0
                try:
                    comm.all_reduce(probe)
                    comm.wait()
                    torch.cuda.synchronize(dev)
                    good = 1.0 if torch.equal(probe, want) else 0.0
                except Exception:              # noqa: BLE001
                    good = 0.0
                ok = torch.tensor([good], device=dev)
                torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN, group=pg)
                if float(ok.item()) != 1.0:
                    if rank == 0:
                        print('bench: simq_comm all-reduce disagrees with torch.distributed; using torch.distributed collectives', file=sys.stderr)
                    comm.close()
                    comm = None
                else:
                    transport = 'libsimq simq_comm (RCCL, library-owned stream)'
                    comm_world = comm.world_size()
            boot.disarm()

    # multi-rank runs: a phase that makes no progress for --watchdog-seconds ends the process with a diagnosis (a hang would hold the lease)
    def comm_progress():
        if comm is None:
            return 'transport %s' % transport
        pr = comm.progress()
        return 'simq_comm: %d collectives enqueued, %d completed, last = %s' % (pr['enqueued'], pr['completed'], pr['last'])

    dog = Watchdog(rank, args.watchdog_seconds, world > 1 and args.watchdog_seconds > 0, comm_progress)

    def barrier(label='barrier'):
        dog.arm(label)
        if pg is not None:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)
        dog.disarm()

    def run_workload(nets, B, precision, steps, warmup, replay_items, leg_opts=None):
        """nets: [(Cin, Cout)] -- one policy/target pair, optimiser state and replay ring per robot group (train.py:180-195);
        B: this rank's transitions per net and step."""
        gB = B * world
        if replay_items < gB:
            sys.exit('bench.py: the replay ring (%d transitions) is smaller than the global minibatch (%d); raise --replay' % (replay_items, gB))
        groups = []
        for gi, (cin, cout) in enumerate(nets):
            # random-init weights of the reference architecture with the reference's own initialisers (resnet.py:70-75,
            # PyTorch defaults for the head): the same seed on every rank gives identical DataParallel replicas, and TD errors
            # stay O(1) so that many steps of synthetic training remain finite
            torch.manual_seed(20260928 + gi)
            popt = dict(plan_opts, **(leg_opts or {})) or None
            policy = simq.FCN(cin, cout, device=dev, precision=precision, options=popt)
            target = simq.FCN(cin, cout, device=dev, precision=precision, options=popt)
            target.copy_state_from(policy)
            policy.train()
            target.eval()
            # synthetic replay, resident in HBM before the timed region (same content on every rank), filled in chunks
            ring = simq.DeviceReplayBuffer(replay_items, cin, device=dev, upload_stream=ring_upload_stream)
            for c0 in range(0, replay_items, 1000):
                trs = synth.make_transitions(min(1000, replay_items - c0), cin, cout, 5 + 31 * gi + c0, terminal_frac=0.1)
                ring.push_many(np.stack([t[0] for t in trs]), [t[1] for t in trs], [t[2] for t in trs],
                               np.stack([t[3] if t[3] is not None else np.zeros_like(t[0]) for t in trs]),
                               [t[3] is None for t in trs])
            # several robot groups (train.py:255-257): independent networks -- every group issues its step on a launch stream of its own
            # (simq.learner.LearnerStreams) and the groups' steps run side by side on the device
            own = len(nets) > 1 and bool(args.group_streams) and pg is None
            ls = learner_streams(policy, own_launch_stream=own)
            groups.append(dict(policy=policy, target=target, ring=ring, opt=_opt_state(policy, None), drawn=None, cin=cin, cout=cout,
                               opts=step_opts, ls=ls))
        random.seed(1234)                   # every rank draws the same global minibatches, then takes its slice
        torch.cuda.synchronize(dev)         # (the rings were filled on the current stream; the groups' launch streams start behind that)

        def draw(g):
            idx = g['ring'].sample_indices(gB)
            return g['ring'].gather(sdist.shard_indices(idx, world, rank), allow_all_final=world > 1)

        exposed_log = []      # (multi-rank runs, measurement pass only: ms the main stream stood in the step's last simq_comm_wait)

        def step(log_exposed=False):
            # train.py:252-258: per robot group one minibatch draw (host-side picks + index upload + HBM gather) and one train().
            # train_step returns the loss as train.py:137-139 does (loss.item()), but waits only for the copy of the four sums the
            # library issues right behind the TD / Huber launch -- not for backward + SGD -- so the draw and the launches of the next
            # step are enqueued while this step still runs (a full stream synchronisation per step left the device idle for 70-90 us
            # at every step boundary: tools/idle_gaps.sh).  Picks, their order per group and the work per step are unchanged; the
            # timed region ends with a device synchronisation.
            # Several groups with launch streams of their own: every group's draw + step is enqueued on ITS stream; 'defer' waits for the
            # losses only after all groups' steps are enqueued (what simq.train_groups does), 'stagger' group by group.
            info, pending = None, []
            for g in groups:
                launch = g['ls'].launch
                with torch.cuda.stream(launch if launch is not None else torch.cuda.current_stream(dev)):
                    batch = g['drawn'] if g['drawn'] is not None else draw(g)
                    defer = launch is not None and args.group_issue == 'defer'
                    info = train_step(g['policy'], g['target'], batch, GAMMA, B, LR, MOMENTUM, WD, CLIP, use_double_dqn=True,
                                      opt_state=g['opt'], process_group=pg, global_batch=gB, sync='defer' if defer else True, comm=comm,
                                      options=g['opts'])
                    if log_exposed:
                        exposed_log.append(comm.last_wait_ms())
                    g['drawn'] = draw(g)
                if defer:
                    pending.append(info)
                elif not np.isfinite(info['loss']):
                    sys.exit('bench: non-finite loss %r' % (info,))
            for pnd in pending:
                info = pnd.result()
                if not np.isfinite(info['loss']):
                    sys.exit('bench: non-finite loss %r' % (info,))
            return info

        for i in range(warmup):
            dog.arm('%s warm-up step %d of %d' % (precision, i + 1, warmup))
            info = step()
        barrier('barrier behind the warm-up')
        t0 = time.perf_counter()
        for i in range(steps):
            dog.arm('%s timed step %d of %d' % (precision, i + 1, steps))
            info = step()
        barrier('barrier behind the timed steps')
        dt = time.perf_counter() - t0
        if pg is not None:
            dt = sdist.max_over_ranks(dt, dev, pg)
        value = gB * len(groups) * steps / dt

        # sustained leg (extra key, never `value`): the driver's 20 steps are ~0.2 s -- on a power-managed part that is a cold-clock
        # number.  The same step for >= --sustained-seconds more, in windows of ~0.5 s (a device synchronisation between windows only).
        sustained = None
        if args.sustained_seconds > 0:
            per = max(1, int(round(0.5 / (dt / steps))))
            # (every loop decision is taken on rank-agreed numbers -- the max-reduced window times -- so that all ranks run the same windows)
            rates, n_done, total = [], 0, 0.0
            while total < args.sustained_seconds or len(rates) < 2:
                t_w = time.perf_counter()
                for i in range(per):
                    dog.arm('%s sustained window %d step %d' % (precision, len(rates) + 1, i + 1))
                    info_s = step()
                barrier('barrier behind a sustained window')
                dw = time.perf_counter() - t_w
                if pg is not None:
                    dw = sdist.max_over_ranks(dw, dev, pg)
                rates.append(gB * len(groups) * per / dw)
                n_done += per
                total += dw
                if len(rates) >= 64:
                    break
            sustained = {'transitions_per_s': round(gB * len(groups) * n_done / total, 1), 'seconds': round(total, 2), 'steps': n_done,
                         'window_steps': per, 'windows': len(rates), 'window_min': round(min(rates), 1), 'window_max': round(max(rates), 1),
                         'vs_timed_window': round(gB * len(groups) * n_done / total / value, 4), 'last_loss': info_s['loss']}

        # Exposed communication (extra key, outside every timed window): per rank, how long the main stream stood waiting in the step's last
        # simq_comm_wait -- the un-overlapped part of gradient bucket 2 (bucket 1 travels beside backward phase 2, the loss scalars beside the whole backward pass; DESIGN 6
        # prices the exposed part at ~0.12 ms).  Timing events around the wait inside libsimq (simq_comm_time_waits); each query synchronises,
        # hence its own pass.
        exposed = None
        if comm is not None:
            comm.time_waits(True)
            for i in range(8):
                dog.arm('%s exposed-communication pass, step %d' % (precision, i + 1))
                step(log_exposed=True)
            barrier('barrier behind the exposed-communication pass')
            comm.time_waits(False)
            mine = torch.tensor([float(np.mean(exposed_log)), float(np.max(exposed_log))], dtype=torch.float64, device=dev)
            allr = [torch.zeros_like(mine) for _ in range(world)]
            torch.distributed.all_gather(allr, mine, group=pg)
            exposed = {'what': 'ms per train() call the main stream waited in simq_comm_wait (gradient bucket 2 not hidden behind backward phase 2)',
                       'calls_timed': len(exposed_log), 'per_rank_mean_ms': [round(float(t[0]), 4) for t in allr],
                       'per_rank_max_ms': [round(float(t[1]), 4) for t in allr]}

        # M1 of SURVEY 8d, the literal reading of the metric ("fwd+bwd"): policy forward (train-mode BN) + gather + Huber +
        # backward only -- no next-state forwards, all-reduce, clip or SGD.  Reported beside the full-step `value`, never instead of it.
        m1 = []
        for g in groups:
            idx = g['ring'].sample_indices(gB)
            fb = g['ring'].gather(sdist.shard_indices(idx, world, rank), allow_all_final=world > 1)
            m1.append(dict(fb=fb, nq=g['cout'] * 96 * 96, out=[torch.empty(B, dtype=torch.float32, device=dev) for _ in range(3)],
                           o4=torch.empty(4, dtype=torch.float32, device=dev), nsv=torch.zeros(B, dtype=torch.float32, device=dev)))

        def fwd_bwd():
            for g, m in zip(groups, m1):
                q = g['policy']._forward_raw(m['fb'].state, MODE_TRAIN)
                dq = torch.empty_like(q)
                lib.call('simq_td_huber', ptr(q), B, m['nq'], ptr(m['fb'].action), ptr(m['fb'].reward), ptr(m['nsv']), GAMMA, 1.0 / gB,
                         ptr(m['out'][0]), ptr(m['out'][1]), ptr(m['out'][2]), ptr(m['o4']), ptr(dq), stream_ptr(dev))
                g['policy']._backward_raw(dq, B)

        dt_m1 = float('nan')
        if not args.no_m1:
            for _ in range(2):
                fwd_bwd()
            barrier()
            t2 = time.perf_counter()
            for _ in range(steps):
                fwd_bwd()
            barrier()
            dt_m1 = time.perf_counter() - t2
            if pg is not None:
                dt_m1 = sdist.max_over_ranks(dt_m1, dev, pg)
        dog.disarm()
        return {'value': value, 'dt': dt, 'dt_m1': dt_m1, 'info': info, 'step': step, 'B': B, 'gB': gB, 'n_nets': len(groups), 'sustained': sustained,
                'm1': None if args.no_m1 else gB * len(groups) * steps / dt_m1, 'groups': groups, 'exposed_comm': exposed}

    def roofline_pass(step_fn, groups_of_step, steps, precision, ms_per_step, per_gpu_rate):
        """Live per-launch timing of the GEMM-class kernels (hipEventRecord pairs on the launch stream, simq_profile_*) over
        `steps` more steps with every stream overlap of the step off (target-net forward, the policy's no-grad forward and the weight
        gradients all on the launch stream), so that every bracket times one kernel ALONE on the device: a roofline fraction is a
        property of the kernel, and two kernels sharing the CUs each look slower than either is."""
        from simq._lib import Plan
        # (the learners' own step options for these steps: everything on ONE launch stream -- no side / early stream, no per-group stream)
        kept = [(g, g['opts'], g['ls'].launch) for g in groups_of_step]
        for g in groups_of_step:
            g['opts'] = StepOptions(overlap_target_forward=False, early_target_forward=False)
            g['ls'].launch = None
        # a SECOND plan per net -- same network, same buffer layout, fwd_overlap = wgrad_overlap = 0 -- serves these steps: what a plan
        # schedules is a property of the plan (simq_plan_options), nothing process-global is flipped
        swapped = []
        for g in groups_of_step:
            for net in (g['policy'], g['target']):
                solo = Plan(net.num_input_channels, net.num_output_channels, net.precision, dict(net.plan.options, fwd_overlap=0, wgrad_overlap=0))
                assert solo.param_count == net.plan.param_count and solo.workspace_bytes(8) == net.plan.workspace_bytes(8)
                swapped.append((net, net.plan))
                net.plan = solo
        lib.call('simq_profile_start')
        barrier()
        t1 = time.perf_counter()
        for _ in range(steps):
            step_fn()
        barrier()
        dt_inst = time.perf_counter() - t1
        out = (ctypes.c_double * 12)()
        lib.call('simq_profile_stop', out, 3)
        torch.cuda.synchronize(dev)
        for g, o, launch in kept:
            g['opts'], g['ls'].launch = o, launch
        for net, plan in swapped:
            net.plan = plan
        dom = {'launches': out[0], 'ms': out[1], 'flops': out[2], 'bytes': out[3]}      # kind 0: the dominant kernel of the precision
        wg = {'launches': out[4], 'ms': out[5], 'flops': out[6], 'bytes': out[7]}       # kind 1: direct weight-gradient launches
        oth = {'launches': out[8], 'ms': out[9], 'flops': out[10], 'bytes': out[11]}    # kind 2: every other implicit-GEMM tile
        allg = {k: dom[k] + oth[k] for k in dom}
        # fp32: the batched GEMM of the Winograd layers (EXECUTED flops: 16 x 2*T*Cout*Cin per launch, 2.25x fewer than the 3x3
        # convolution it implements); bf16: the image-tile kernel; when a precision has no kind-0 launches, all tiles
        ig = dom if dom['launches'] > 0 else allg
        tf = lambda d: d['flops'] / (d['ms'] * 1e-3) / 1e12 if d['ms'] > 0 else 0.0
        # fp32 plans, round 6: the dominant kernel runs on the bf16 matrix cores (simq_plan_options.gemm_split = 1, the default) and issues SIX
        # bf16 MFMA products per fp32 product: its executed flops are 6 x the contraction's and its roof is the dense bf16 peak; every other
        # fp32 kernel (direct convolutions, weight gradients, stem) stays on v_mfma_f32_16x16x4_f32 and is priced against the fp32 matrix peak
        split = precision == 'fp32' and groups_of_step[0]['policy'].plan.options.get('gemm_split', 0) == 1 and dom['launches'] > 0
        key = precision if (precision != 'fp32' or split) else 'fp32_mfma'
        PEAK = PEAK_FP32_MFMA_TFLOPS if precision == 'fp32' else PEAK_BF16_MFMA_TFLOPS
        fp32_equiv = None
        if split:
            fp32_equiv = {'fp32_contraction_tflops': round(tf(dom), 2), 'of_fp32_mfma_peak': round(tf(dom) / PEAK_FP32_MFMA_TFLOPS, 4)}
            dom = dict(dom, flops=6.0 * dom['flops'])
            ig = dom
            # the rest of the fp32 step's matrix work in bf16-pipe-equivalent flops (x peak ratio), so that ONE peak prices the sums below
            scale = PEAK_BF16_MFMA_TFLOPS / PEAK_FP32_MFMA_TFLOPS
            oth = dict(oth, flops=scale * oth['flops'])
            wg = dict(wg, flops=scale * wg['flops'])
            allg = {k: dom[k] + oth[k] for k in dom}
            PEAK = PEAK_BF16_MFMA_TFLOPS
        executed = (dom['flops'] + oth['flops'] + wg['flops']) / steps          # matrix-core flops actually issued per step
        traffic, traffic_source = pmc_traffic(key)
        return {
            'bound': 'mfma', 'kernel': KERNEL_NAMES[key], 'fp32_equivalent': fp32_equiv,
            'matrix_pipe': 'bf16 (v_mfma_f32_32x32x16_bf16; the other fp32 kernels of the step run v_mfma_f32_16x16x4_f32 and enter all_gemm_tiles / wgrad / '
                           'whole_step figures scaled by the peak ratio, i.e. as matrix-pipe TIME)' if split else ('fp32 (v_mfma_f32_16x16x4_f32)' if precision == 'fp32' else 'bf16'),
            'achieved': round(tf(ig), 2), 'peak': PEAK, 'unit': 'TFLOP/s', 'frac': round(tf(ig) / PEAK, 4),
            'traffic': traffic, 'traffic_source': traffic_source,
            'launches_per_step': ig['launches'] / steps, 'avg_launch_ms': round(ig['ms'] / max(ig['launches'], 1), 5),
            'algorithmic_flops_per_launch': ig['flops'] / max(ig['launches'], 1),
            'kernel_ms_per_step': round(ig['ms'] / steps, 4),
            # flat copies of the secondary figures (nested objects do not survive the driver's summary of the line)
            'all_gemm_tiles_launches_per_step': allg['launches'] / steps, 'all_gemm_tiles_ms_per_step': round(allg['ms'] / steps, 4),
            'all_gemm_tiles_achieved': round(tf(allg), 2), 'all_gemm_tiles_frac': round(tf(allg) / PEAK, 4),
            'wgrad_launches_per_step': wg['launches'] / steps, 'wgrad_ms_per_step': round(wg['ms'] / steps, 4),
            'wgrad_achieved': round(tf(wg), 2), 'wgrad_frac': round(tf(wg) / PEAK, 4),
            # the whole step against the matrix-core roof: EXECUTED matrix flops per step / un-instrumented step time / peak
            'whole_step_executed_gflop': round(executed / 1e9, 2),
            'whole_step_executed_frac': round(executed / (ms_per_step * 1e-3) / (PEAK * 1e12), 4),
            'whole_step_hbm_frac_activation_lower_bound': round(per_gpu_rate * (61.9e6 if precision == 'fp32' else 31.0e6) / (PEAK_HBM_GBS * 1e9), 5),
            'ms_per_step_instrumented': round(dt_inst / steps * 1e3, 3),
            'timed': 'HIP-event pairs around every launch of the kernel over %d steps behind the timed region, each kernel ALONE on the device '
                     '(the step\'s stream overlaps off for these steps only; `value` is the overlapped step); the matching rocprofv3 trace is '
                     'profiles/r06_bench_%s_kernel_trace_serial.txt, the overlapped step\'s profiles/r06_bench_%s_kernel_trace.txt' % (
                         steps, 'b32' if precision == 'fp32' else 'bf16_b128', 'b32' if precision == 'fp32' else 'bf16_b128'),
        }

    def per_rank(roof):
        """Every rank's own dominant-kernel figure (each rank times its own launches), gathered to rank 0."""
        if pg is None or roof is None:
            return None
        mine = {'rank': rank, 'achieved': roof['achieved'], 'frac': roof['frac'], 'kernel_ms_per_step': roof['kernel_ms_per_step']}
        every = [None] * world
        torch.distributed.all_gather_object(every, mine, group=pg)
        return every

    def release(w):
        for g in w['groups']:
            g.clear()
        w.clear()
        torch.cuda.empty_cache()

    wl = resolve_workload(args, world)
    w = run_workload(wl['nets'], wl['per_gpu'], wl['precision'], args.steps, args.warmup, args.replay)
    value, dt, info, gB = w['value'], w['dt'], w['info'], w['gB']
    roof = None
    if not args.no_roofline:
        roof = roofline_pass(w['step'], w['groups'], args.steps, wl['precision'], dt / args.steps * 1e3, value / world)
        ranks_roof = per_rank(roof)
        if ranks_roof is not None:
            roof['per_rank'] = ranks_roof
    m1, dt_m1, n_nets, sustained, exposed_comm = w['m1'], w['dt_m1'], w['n_nets'], w['sustained'], w['exposed_comm']
    release(w)

    # second leg, reported beside the headline, never instead of it.  N=1 (configs[1] headline): BASELINE configs[2] on the bf16
    # matrix-core path.  N>1: the weak-scaling form (configs[1]'s net at 32 transitions per GPU) next to the config-faithful value.
    extra_name = None
    if not args.no_extras and args.batch is None and args.cin is None and args.precision is None:
        if world == 1 and wl['name'] == 'configs1':
            extra_name = 'configs2'
        elif world > 1 and wl['name'] != 'weak32':
            extra_name = 'weak32'
    extras, roof_x = None, None
    if extra_name is not None:
        xl = resolve_workload(argparse.Namespace(workload=extra_name, cin=None, precision=None, batch=None), world)
        try:
            e = run_workload(xl['nets'], xl['per_gpu'], xl['precision'], args.steps, args.warmup, args.replay)
            extras = {'workload': '%s: Cin %d, global minibatch %d (%d per GPU), %s' % (xl['config'], xl['nets'][0][0], e['gB'], e['B'], DTYPE_NAMES[xl['precision']]),
                      'full_step_transitions_per_s': round(e['value'], 1), 'ms_per_step': round(e['dt'] / args.steps * 1e3, 3),
                      'steps': args.steps, 'warmup': args.warmup, 'scaling': xl['scaling'],
                      'fwd_bwd_only_transitions_per_s': None if e['m1'] is None else round(e['m1'], 1),
                      'fwd_bwd_only_ms_per_step': None if e['m1'] is None else round(e['dt_m1'] / args.steps * 1e3, 3), 'last_loss': e['info']['loss'],
                      'sustained': e['sustained'], 'exposed_comm': e['exposed_comm']}
            if not args.no_roofline:
                roof_x = roofline_pass(e['step'], e['groups'], args.steps, xl['precision'], e['dt'] / args.steps * 1e3, e['value'] / world)
            release(e)
        except Exception as ex:       # the headline line must not depend on the second leg
            if pg is not None:
                raise                 # (a rank that left the collectives cannot rejoin them)
            extras = {'error': repr(ex)}

    # third leg (N = 1, configs[1] only): the SAME fp32 workload with the transform-domain GEMMs on the fp32 matrix pipe
    # (simq_plan_options.gemm_split = 0, the form of rounds 1-5) -- so that the line carries both forms of the fp32 arithmetic side by side.
    # (Mid round 6 this leg read 10-13 % low as the third workload of the process -- its side stream had landed on the launch stream's hardware
    # queue; simq.learner now picks a step's streams from streams TESTED to run concurrently, and a workload reads the same whether it is the
    # first or the fourth of a process: profiles/r06_third_leg_order_effect.txt.)
    mfma_leg, roof_m = None, None
    if extra_name == 'configs2' and 'gemm_split' not in plan_opts and wl['precision'] == 'fp32':
        try:
            e = run_workload(wl['nets'], wl['per_gpu'], wl['precision'], args.steps, args.warmup, args.replay, leg_opts={'gemm_split': args.third_leg_split})
            mfma_leg = {'workload': '%s with simq_plan_options.gemm_split = %d: the transform-domain GEMMs on v_mfma_f32_16x16x4_f32 (rounds 1-5)' % (wl['config'], args.third_leg_split),
                        'full_step_transitions_per_s': round(e['value'], 1), 'ms_per_step': round(e['dt'] / args.steps * 1e3, 3),
                        'fwd_bwd_only_transitions_per_s': None if e['m1'] is None else round(e['m1'], 1), 'last_loss': e['info']['loss'],
                        'sustained': e['sustained']}
            if not args.no_roofline:
                roof_m = roofline_pass(e['step'], e['groups'], args.steps, wl['precision'], e['dt'] / args.steps * 1e3, e['value'] / world)
            release(e)
        except Exception as ex:       # noqa: BLE001
            mfma_leg = {'error': repr(ex)}

    # the CPU baseline LAST: its 32-128 OpenMP threads keep spinning for a while after the last oracle call and slowed the host thread that
    # enqueues the GPU legs behind it (round 6: the fp32-MFMA leg read 3586 tr/s behind it, 3990 on its own)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(wl['nets'][0][0], wl['nets'][0][1], min(wl['per_gpu'], 32))

    if rank == 0:
        cin0 = wl['nets'][0][0]
        flop_m1, flop_m2 = FLOPS.get(cin0, FLOPS[5])
        line = {
            'metric': 'Q-map transitions/sec, 96x96: `value` = full train() step (M2: policy fwd + bwd, double-DQN next-state forwards, clip, SGD); '
                      '`value_fwd_bwd_only` = policy fwd+bwd alone (M1, the literal reading of "fwd+bwd")',
            'value': round(value, 2), 'unit': 'transitions/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': wl['scaling'],
            'vs_baseline': None, 'dtype': DTYPE_NAMES[wl['precision']], 'data': 'synthetic',
            'value_fwd_bwd_only': None if m1 is None else round(m1, 2),
            'ms_per_step_fwd_bwd_only': None if m1 is None else round(dt_m1 / args.steps * 1e3, 3),
            'config': {'workload': '%s: %s, 96x96, global minibatch %d%s = %d per GPU%s, double DQN, device-resident replay of %d transitions per net'
                                   % (wl['config'], ' + '.join('Cin=%d->Cout=%d' % n for n in wl['nets']), gB, ' per net' if n_nets > 1 else '',
                                      wl['per_gpu'], ' and net' if n_nets > 1 else '', args.replay),
                       'workload_key': wl['name'], 'nets': n_nets, 'global_batch': gB, 'per_gpu_batch': wl['per_gpu'],
                       'transitions_per_step': gB * n_nets, 'parallelism': 'dp%d' % world,
                       'concurrent_groups': (args.group_issue if (n_nets > 1 and args.group_streams and world == 1) else None),
                       'gradient_transport': transport, 'simq_comm_world_size': comm_world, 'backend': args.backend if world > 1 else None,
                       'flop_per_transition': flop_m2, 'flop_per_transition_fwd_bwd_only': flop_m1,
                       'fp32_gemm_form': (None if wl['precision'] != 'fp32' else
                                          ('gemm_split = %d: %s' % ((plan_opts.get('gemm_split', 1)), 'transform-domain GEMMs of the Winograd layers on the bf16 matrix cores through an '
                                           'EXACT three-way split of both fp32 operands (6 partial products, fp32 accumulate; fp32 in, fp32 out, round-off <= the '
                                           'fp32-MFMA form\'s against fp64: tests/test_gpu_ops.py); the same workload on v_mfma_f32_16x16x4_f32 is the leg `fp32_mfma_configs1`'
                                           if plan_opts.get('gemm_split', 1) == 1 else 'transform-domain GEMMs on v_mfma_f32_16x16x4_f32'))),
                       'last_loss': info['loss'], 'last_td_error': info['td_error']},
            'roofline': roof, 'cpu_baseline': cpu, 'sustained': sustained, 'exposed_comm': exposed_comm,
        }
        # BASELINE.md publishes no number for this metric (BASELINE.json "published": {}), so `vs_baseline` stays null by the bench
        # contract; what exists is north_star's TARGET -- ">= 10k Q-map forward+backward transitions/sec ... at 1 GPU" -- and `vs_target`
        # is the ratio to it.  The target is a forward+backward (M1) rate; fp32 arithmetic cannot reach it on this part (the direct-
        # convolution fp32 MFMA ceiling is 4.04 k tr/s), the bf16 leg is the one that does.
        target = 10000.0
        vs_target = {'target': 'north_star: >= 10 000 Q-map forward+backward transitions/s on synthetic 96x96xC batches at 1 GPU', 'target_value': target,
                     'applies_to': 'per-GPU forward+backward rate (value_fwd_bwd_only / n_gpus)',
                     'this_workload_fwd_bwd': None if m1 is None else round(m1 / world / target, 4),
                     'this_workload_full_step': round(value / world / target, 4)}
        if extras is not None:
            key = 'bf16_configs2' if extra_name == 'configs2' else extra_name
            line[key] = extras
            line['roofline_' + key] = roof_x
            for k in ('full_step_transitions_per_s', 'fwd_bwd_only_transitions_per_s', 'ms_per_step'):
                if k in extras:
                    line['config'][key + '_' + k] = extras[k]
            if extras.get('fwd_bwd_only_transitions_per_s') is not None:
                vs_target[key + '_fwd_bwd'] = round(extras['fwd_bwd_only_transitions_per_s'] / world / target, 4)
                vs_target[key + '_full_step'] = round(extras['full_step_transitions_per_s'] / world / target, 4)
            if extras.get('sustained'):
                line['config'][key + '_sustained_transitions_per_s'] = extras['sustained']['transitions_per_s']
            if roof_x is not None and roof is not None:
                for k in ('achieved', 'frac', 'traffic', 'traffic_source', 'avg_launch_ms', 'launches_per_step', 'kernel_ms_per_step',
                          'whole_step_executed_frac', 'wgrad_frac', 'all_gemm_tiles_frac'):
                    line['roofline'][key + '_' + k] = roof_x[k]
        if mfma_leg is not None:
            line['fp32_mfma_configs1'] = mfma_leg
            line['roofline_fp32_mfma_configs1'] = roof_m
            if 'full_step_transitions_per_s' in mfma_leg:
                line['config']['fp32_mfma_configs1_full_step_transitions_per_s'] = mfma_leg['full_step_transitions_per_s']
                line['config']['fp32_mfma_configs1_fwd_bwd_only_transitions_per_s'] = mfma_leg['fwd_bwd_only_transitions_per_s']
            if roof_m is not None and roof is not None:
                for k in ('achieved', 'frac', 'avg_launch_ms', 'kernel_ms_per_step', 'whole_step_executed_frac'):
                    line['roofline']['fp32_mfma_configs1_' + k] = roof_m[k]
        line['vs_target'] = vs_target
        line['vs_baseline_note'] = 'null: BASELINE.md holds no published number for this metric; see vs_target for the ratio to the north_star target'
        print(json.dumps(line), flush=True)
    if comm is not None:
        comm.close()
    if pg is not None:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
