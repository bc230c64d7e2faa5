#!/usr/bin/env python3
"""bench.py -- Q-map transitions/sec of the spatial-action-map DQN training step on MI355X.

Workload (BASELINE.json configs[1], the config the metric is quoted on):
  lifting_1-small_empty: Cin=4 -> Cout=2, minibatch 32 per GPU, fp32, double DQN,
  lr 0.01 / momentum 0.9 / wd 1e-4 / clip 100, synthetic replay (seeded).
A "step" is ONE full reference train() call (train.py:108-141) on one sampled minibatch:
  replay index sampling + HBM gather, policy forward (train-mode BN), double-DQN next-state
  forwards (policy train-mode no-grad + target eval), TD target + Huber, backward, global-norm
  clip, momentum SGD, and the two scalar read-backs the reference does (.item()).
  That is metric definition M2 of SURVEY 8d (65.0 GFLOP/transition); nothing is skipped.
N GPUs: one process per GPU (torch.distributed, backend nccl == RCCL), weak scaling (32
transitions per GPU), per-rank BatchNorm statistics (the reference's DataParallel semantics),
ONE all-reduce of the flat 45 MB gradient buffer per step.

Prints ONE JSON line on rank 0 (driver contract) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import ctypes
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, 'spatial-intention-maps_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

# multi-process GPU work on this ROCm stack needs dmabuf IPC (RCCL / cross-process tensor sharing); keep whatever the
# launcher exported, default to the supported mode otherwise -- must be in the environment before the HIP runtime starts
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import numpy as np  # noqa: E402
import torch  # noqa: E402

CIN, COUT, BATCH_PER_GPU = 4, 2, 32
GAMMA, LR, MOMENTUM, WD, CLIP = 0.75, 0.01, 0.9, 1e-4, 100.0
REPLAY_ITEMS = 1024                      # synthetic transitions resident in the HBM ring per rank
FLOP_M1, FLOP_M2 = 38.963e9, 64.976e9    # per transition, SURVEY 8d [probe-derived] (fwd+bwd / full train())
PEAK_FP32_MFMA_TFLOPS = 157.3            # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0           # dense bf16 MFMA (not the 2:1-sparse marketing figure)
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-m1', action='store_true', help='skip the forward+backward-only (M1) leg (profiling runs: every launch then belongs to a full step)')
    ap.add_argument('--no-extras', action='store_true', help='skip the opt-in bf16 configs[2] leg of the default N=1 run')
    ap.add_argument('--precision', default='fp32', choices=['fp32', 'bf16x3', 'bf16'],
                    help="arithmetic of the 3x3/1x1 convolutions; the default 'fp32' (exact) is the BASELINE configs[1] workload")
    ap.add_argument('--batch', type=int, default=BATCH_PER_GPU, help='transitions per GPU per step (default: configs[1])')
    ap.add_argument('--cin', type=int, default=CIN)
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help='torch.distributed backend for --gpus > 1 (nccl == RCCL; gloo only to debug the rank logic on one GPU)')
    return ap.parse_args()


def cpu_baseline(batch):
    """The oracle (CPU restatement, pinned bit-exact to the reference in the build container)
    running the SAME step on this box's host cores.  Bounded sample (~10-30 s): 1 warm-up + up to 3
    timed train() calls at the GPU workload's batch size.  Threads: the CPUs this process may run
    on, capped at 32 (oversubscribing a 256-thread host made MKL-DNN 30x slower: 0.26 tr/s)."""
    from oracle import cases, fcn as ofcn, learner as olearner
    from simq import synth
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    nthreads = max(1, min(avail, 32))
    torch.set_num_threads(nthreads)
    cfg = cases.make_cfg(batch)
    spec = ofcn.state_spec(CIN, COUT)
    st = ofcn.state_from_numpy(synth.make_state_dict(CIN, COUT, 1))
    tg = ofcn.state_from_numpy(synth.make_state_dict(CIN, COUT, 2))
    mom = [None] * len(olearner.grad_keys(spec))
    trs = synth.make_transitions(batch, CIN, COUT, 3, terminal_frac=0.1)
    b = olearner.Transition(*zip(*trs))
    t0 = time.perf_counter()
    olearner.train_step(cfg, st, tg, spec, mom, b, GAMMA, LR, MOMENTUM, WD)
    warm = time.perf_counter() - t0
    steps = 3 if warm < 8 else 1
    t0 = time.perf_counter()
    for _ in range(steps):
        olearner.train_step(cfg, st, tg, spec, mom, b, GAMMA, LR, MOMENTUM, WD)
    dt = time.perf_counter() - t0
    return {'value': round(batch * steps / dt, 3), 'unit': 'transitions/s', 'cores': nthreads,
            'kind': 'port',
            'sample': 'oracle train_step (== reference train.py:108-141 on torch-CPU/MKL-DNN fp32), batch %d, '
                      '1 warm-up (%.1f s) + %d timed calls (%.1f s); host reports %d CPUs (%d usable)'
                      % (batch, warm, steps, dt, os.cpu_count() or 0, avail)}


PMC_TRAFFIC = {   # precision -> (committed rocprofv3 PMC summary, kernel whose bytes per launch `roofline.traffic` quotes)
    'fp32': ('r02_pmc_traffic.json', 'igemm_conv_kernel<64, 64, true, true>'),
    'bf16': ('r02_pmc_traffic_bf16_b128.json', None),          # None: the kernel named by DOMINANT_BF16 below
}
DOMINANT_BF16 = 'igemm_bf16_img_kernel'      # name prefix of the bf16 leg's dominant kernel in the rocprofv3 summaries


def pmc_traffic(precision):
    """HBM-side bytes per launch of the dominant kernel (FETCH_SIZE x2 + WRITE_SIZE, KiB -> bytes) from the committed
    rocprofv3 PMC passes over this same command (profiles/r02_pmc_traffic*.json, tools/pmc_traffic.py); PMC counters cannot
    be read from inside the timed process, so the value is the recorded one, or None when the file is absent."""
    if precision not in PMC_TRAFFIC:
        return None
    fname, kernel = PMC_TRAFFIC[precision]
    for f in (fname, fname.replace('r02_', 'r01_')):
        try:
            t = json.load(open(os.path.join(ROOT, 'profiles', f)))
        except (OSError, ValueError):
            continue
        if kernel is not None and kernel in t:
            return round(t[kernel]['bytes_per_launch'])
        cand = [(v['launches'] * v['bytes_per_launch'], v) for k, v in t.items() if k.startswith(DOMINANT_BF16) or k.startswith('igemm_bf16_dma_kernel')]
        if kernel is None and cand:
            return round(max(cand, key=lambda kv: kv[0])[1]['bytes_per_launch'])
    return None


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit('bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d' % (args.gpus, args.gpus))
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit('bench.py needs an MI355X (no GPU visible); there is no CPU product path')
    ndev = torch.cuda.device_count()
    if world > 1 and args.backend == 'nccl' and ndev < world:
        sys.exit('bench.py: %d ranks need %d GPUs, only %d visible' % (world, world, ndev))
    local_dev = local_rank % ndev       # (gloo debugging may stack ranks on one GPU)
    torch.cuda.set_device(local_dev)
    dev = torch.device('cuda', local_dev)
    pg = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        pg = dist.group.WORLD

    import simq
    from simq import dist as sdist, synth
    from simq._lib import lib
    from simq.learner import _opt_state, train_step

    # gradient exchange: libsimq's own RCCL communicator (simq_comm_*: the data-parallel step is then ONE library call, buckets on
    # the communicator's stream); checked once against torch.distributed's all-reduce.  If RCCL cannot be initialised through
    # the library on this node the step falls back to torch.distributed's RCCL collectives around the backward phases -- still
    # RCCL over xGMI, never a CPU path.  `transport` in the JSON line says which one ran.
    comm, transport = None, 'none (single GPU)'
    if pg is not None:
        transport = 'torch.distributed (%s)' % args.backend
        if args.backend == 'nccl' and os.environ.get('SIMQ_BENCH_COMM', '1') != '0':
            try:
                comm = sdist.Comm(pg, dev)
                probe = torch.arange(1024, dtype=torch.float32, device=dev) * (rank + 1)
                want = probe.clone()
                torch.distributed.all_reduce(want, group=pg)
                comm.all_reduce(probe)
                comm.wait()
                torch.cuda.synchronize(dev)
                ok = torch.tensor([1.0 if torch.equal(probe, want) else 0.0], device=dev)
                torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN, group=pg)
                if float(ok.item()) != 1.0:
                    raise RuntimeError('simq_comm all-reduce disagrees with torch.distributed')
                transport = 'libsimq simq_comm (RCCL, library-owned stream)'
            except Exception as ex:            # noqa: BLE001  (report and use the other RCCL transport)
                if rank == 0:
                    print('bench: simq_comm unavailable (%r); using torch.distributed collectives' % (ex,), file=sys.stderr)
                comm = None

    def run_workload(CIN, BATCH_PER_GPU, precision, steps, warmup, replay_items):
        # random-init weights of the reference architecture with the reference's own initialisers (resnet.py:70-75,
        # PyTorch defaults for the head): the same seed on every rank gives identical DataParallel replicas, and TD errors
        # stay O(1) so that many steps of synthetic training remain finite
        torch.manual_seed(20260928)
        policy = simq.FCN(CIN, COUT, device=dev, precision=precision)
        target = simq.FCN(CIN, COUT, device=dev, precision=precision)
        target.copy_state_from(policy)
        policy.train()
        target.eval()
        st_opt = _opt_state(policy, None)

        # synthetic replay, resident in HBM before the timed region (same content on every rank)
        trs = synth.make_transitions(replay_items, CIN, COUT, 5, terminal_frac=0.1)
        ring = simq.DeviceReplayBuffer(replay_items, CIN, device=dev)
        ring.push_many(np.stack([t[0] for t in trs]), [t[1] for t in trs], [t[2] for t in trs],
                       np.stack([t[3] if t[3] is not None else np.zeros_like(t[0]) for t in trs]),
                       [t[3] is None for t in trs])
        B, gB = BATCH_PER_GPU, BATCH_PER_GPU * world
        random.seed(1234)                   # every rank draws the same global minibatch, then takes its slice

        def draw():
            idx = ring.sample_indices(gB)
            return ring.gather(sdist.shard_indices(idx, world, rank), allow_all_final=world > 1)

        drawn = [None]

        def step():
            # One minibatch draw (host-side picks + index upload + HBM gather) and one train() per step, as in train.py:252-258.
            # The draw for step k+1 is issued while step k's kernels run and BEFORE step k's loss is read back (the
            # reference's .item() sync), so the host-side sampler is hidden behind the GPU instead of idling it; picks, their
            # order and the work per step are unchanged (the replay ring is static during the benchmark).
            batch = drawn[0] if drawn[0] is not None else draw()
            out4 = train_step(policy, target, batch, GAMMA, B, LR, MOMENTUM, WD, CLIP, use_double_dqn=True,
                              opt_state=st_opt, process_group=pg, global_batch=gB, sync=False, comm=comm)
            drawn[0] = draw()
            o = out4.tolist()
            return {'td_error': o[1] / gB, 'loss': o[0] / gB}

        def barrier():
            if pg is not None:
                torch.distributed.barrier()
            torch.cuda.synchronize(dev)

        for _ in range(warmup):
            info = step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            info = step()
        barrier()
        dt = time.perf_counter() - t0
        if pg is not None:
            dt = sdist.max_over_ranks(dt, dev, pg)
        if not np.isfinite(info['loss']):
            sys.exit('bench: non-finite loss %r' % (info,))
        value = gB * steps / dt

        # M1 of SURVEY 8d, the literal reading of the metric ("fwd+bwd"): policy forward (train-mode BN) + gather + Huber +
        # backward only -- no next-state forwards, clip or SGD.  Reported beside the full-step `value`, never instead of it.
        from simq._lib import MODE_TRAIN, ptr, stream_ptr
        idx = ring.sample_indices(gB)
        fb = ring.gather(sdist.shard_indices(idx, world, rank), allow_all_final=world > 1)
        nq = COUT * 96 * 96
        fb_out = [torch.empty(B, dtype=torch.float32, device=dev) for _ in range(3)]
        fb_o4 = torch.empty(4, dtype=torch.float32, device=dev)
        fb_nsv = torch.zeros(B, dtype=torch.float32, device=dev)

        def fwd_bwd():
            q = policy._forward_raw(fb.state, MODE_TRAIN)
            dq = torch.empty_like(q)
            lib.call('simq_td_huber', ptr(q), B, nq, ptr(fb.action), ptr(fb.reward), ptr(fb_nsv), GAMMA, 1.0 / gB, ptr(fb_out[0]),
                     ptr(fb_out[1]), ptr(fb_out[2]), ptr(fb_o4), ptr(dq), stream_ptr(dev))
            policy._backward_raw(dq, B)

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
        if pg is not None and not args.no_m1:
            dt_m1 = sdist.max_over_ranks(dt_m1, dev, pg)

        return {'value': value, 'dt': dt, 'dt_m1': dt_m1, 'info': info, 'step': step, 'barrier': barrier, 'B': B, 'gB': gB}

    global CIN, BATCH_PER_GPU
    CIN, BATCH_PER_GPU = args.cin, args.batch
    w = run_workload(CIN, BATCH_PER_GPU, args.precision, args.steps, args.warmup, REPLAY_ITEMS)
    value, dt, dt_m1, info, step, barrier, B, gB = (w[k] for k in ('value', 'dt', 'dt_m1', 'info', 'step', 'barrier', 'B', 'gB'))

    KERNEL_NAMES = {
        'fp32': 'igemm_conv_kernel<64,64,true,true> (batched transform-domain GEMM of the Winograd layers -- 128->256, 256- and 512-channel 3x3 '
                'convolutions: 16 GEMMs per launch for the grad-mode forward in F(2x2,3x3), 36 for the no-grad forwards, dgrads and weight '
                'gradients in F(4x4,3x3); v_mfma_f32_16x16x4_f32; achieved = EXECUTED flops / time)',
        'bf16x3': 'igemm_bf16_kernel<NP=2> (split-bf16 implicit GEMM, 3 x v_mfma_f32_16x16x32_bf16 per product; '
                  'achieved counts ALGORITHMIC flops, matrix-core work is 3x that)',
        'bf16': 'igemm_bf16_img_kernel (image-tile 3x3 convolution: one 24x24 map x 128 channels per block, halo patch in LDS, ping-pong wave '
                'groups, v_mfma_f32_16x16x32_bf16): forward + dgrad of the 256- and 512-channel 3x3 convolutions; the 288-row LDS-DMA kernels of '
                'the other layers are counted under all_gemm_tiles)'}

    def roofline_pass(step_fn, barrier_fn, steps, precision, ms_per_step, per_gpu_rate):
        """Live per-launch timing of the GEMM-class kernels (hipEventRecord pairs on the launch stream, simq_profile_*) over
        `steps` more steps with the two-stream overlap off, so that every bracket times one kernel alone."""
        import simq.learner as slearner
        keep = slearner.OVERLAP_TARGET_FORWARD
        slearner.OVERLAP_TARGET_FORWARD = False
        lib.call('simq_profile_start')
        barrier_fn()
        t1 = time.perf_counter()
        for _ in range(steps):
            step_fn()
        barrier_fn()
        dt_inst = time.perf_counter() - t1
        out = (ctypes.c_double * 12)()
        lib.call('simq_profile_stop', out, 3)
        slearner.OVERLAP_TARGET_FORWARD = keep
        dom = {'launches': out[0], 'ms': out[1], 'flops': out[2], 'bytes': out[3]}      # kind 0: the dominant kernel of the precision
        wg = {'launches': out[4], 'ms': out[5], 'flops': out[6], 'bytes': out[7]}       # kind 1: direct weight-gradient launches
        oth = {'launches': out[8], 'ms': out[9], 'flops': out[10], 'bytes': out[11]}    # kind 2: every other implicit-GEMM tile
        allg = {k: dom[k] + oth[k] for k in dom}
        # fp32: the batched GEMM of the Winograd layers (EXECUTED flops: 16 x 2*T*Cout*Cin per launch, 2.25x fewer than the 3x3
        # convolution it implements); bf16: the large-tile LDS-DMA kernel; when a precision has no kind-0 launches, all tiles
        ig = dom if dom['launches'] > 0 else allg
        tf = lambda d: d['flops'] / (d['ms'] * 1e-3) / 1e12 if d['ms'] > 0 else 0.0
        PEAK = PEAK_FP32_MFMA_TFLOPS if precision == 'fp32' else PEAK_BF16_MFMA_TFLOPS
        executed = (dom['flops'] + oth['flops'] + wg['flops']) / steps          # matrix-core flops actually issued per step
        return {
            'bound': 'mfma', 'kernel': KERNEL_NAMES[precision],
            'achieved': round(tf(ig), 2), 'peak': PEAK, 'unit': 'TFLOP/s', 'frac': round(tf(ig) / PEAK, 4),
            'traffic': pmc_traffic(precision),
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
        }

    roof = None
    if not args.no_roofline:
        roof = roofline_pass(step, barrier, args.steps, args.precision, dt / args.steps * 1e3, value / world)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(min(BATCH_PER_GPU, 32))

    # BASELINE configs[2] (lifting_4-small_divider: Cin=5, batch 128, bf16 operands) on the opt-in bf16 matrix-core path,
    # reported beside the fp32 headline, never instead of it (N=1 default run only; a few seconds)
    extras, roof_bf16 = None, None
    if world == 1 and args.precision == 'fp32' and not args.no_extras and args.cin == 4 and args.batch == 32:
        try:
            e = run_workload(5, 128, 'bf16', 10, 3, 256)
            extras = {'workload': 'BASELINE configs[2] lifting_4-small_divider (Cin=5), minibatch 128, bf16 operands (f32 accumulate / BN / optimiser)',
                      'full_step_transitions_per_s': round(e['value'], 1), 'ms_per_step': round(e['dt'] / 10 * 1e3, 3),
                      'fwd_bwd_only_transitions_per_s': round(e['gB'] * 10 / e['dt_m1'], 1),
                      'fwd_bwd_only_ms_per_step': round(e['dt_m1'] / 10 * 1e3, 3), 'last_loss': e['info']['loss']}
            if not args.no_roofline:
                roof_bf16 = roofline_pass(e['step'], e['barrier'], 10, 'bf16', e['dt'] / 10 * 1e3, e['value'])
            del e
            torch.cuda.empty_cache()
        except Exception as ex:       # the headline line must not depend on the opt-in leg
            extras = {'error': repr(ex)}

    if rank == 0:
        m1 = None if args.no_m1 else round(gB * args.steps / dt_m1, 2)
        line = {
            'metric': 'Q-map transitions/sec (full train() step: 3 fwd + bwd + clip + SGD, 96x96)',
            'value': round(value, 2), 'unit': 'transitions/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': {'fp32': 'f32', 'bf16x3': 'bf16x3 (split-bf16 operands, 3 MFMA products, f32 accumulate)', 'bf16': 'bf16 (f32 accumulate / BN / optimiser)'}[args.precision], 'data': 'synthetic',
            # BASELINE's metric text says "fwd+bwd": the literal reading (M1 of SURVEY 8d: policy forward + gather + Huber + backward only,
            # no next-state forwards / all-reduce / clip / SGD) beside the full-step `value` (M2), never instead of it
            'value_fwd_bwd_only': m1, 'ms_per_step_fwd_bwd_only': None if args.no_m1 else round(dt_m1 / args.steps * 1e3, 3),
            'config': {'workload': '%s (Cin=%d, Cout=2, 96x96), minibatch %d per GPU, double DQN, '
                                   'device-resident replay of %d transitions' % ('lifting_1-small_empty' if CIN == 4 else 'Cin=%d variant' % CIN, CIN, BATCH_PER_GPU, REPLAY_ITEMS),
                       'global_batch': gB, 'parallelism': 'dp%d' % world, 'gradient_transport': transport,
                       'flop_per_transition': FLOP_M2, 'flop_per_transition_fwd_bwd_only': FLOP_M1,
                       'last_loss': info['loss'], 'last_td_error': info['td_error']},
            'roofline': roof, 'cpu_baseline': cpu,
        }
        if extras is not None:      # BASELINE configs[2] on the opt-in bf16 path: flat `bf16_*` keys + its own roofline object
            line['bf16_configs2'] = extras
            line['roofline_bf16_configs2'] = roof_bf16
            for k in ('full_step_transitions_per_s', 'fwd_bwd_only_transitions_per_s', 'ms_per_step'):
                if k in extras:
                    line['config']['bf16_configs2_' + k] = extras[k]
            if roof_bf16 is not None:
                for k in ('achieved', 'frac', 'traffic', 'avg_launch_ms', 'launches_per_step', 'kernel_ms_per_step', 'whole_step_executed_frac',
                          'wgrad_frac', 'all_gemm_tiles_frac'):
                    line['roofline']['bf16_configs2_' + k] = roof_bf16[k]
        print(json.dumps(line))
    if comm is not None:
        comm.close()
    if pg is not None:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
