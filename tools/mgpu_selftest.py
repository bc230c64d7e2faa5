#!/usr/bin/env python3
"""First contact with more than one GPU: one PASS / FAIL table, so that a lease on a multi-GPU node yields a diagnosis even if the
benchmark itself fails.

Stages, in order (each in its own process group with a time limit; a stage that fails or hangs is reported and the next one still runs):
  1  visible devices, librccl.so.1 loadable, HSA_ENABLE_IPC_MODE_LEGACY
  2  N ranks (torch.distributed.run on 127.0.0.1): torch.distributed's own RCCL all-reduce -- is the node's RCCL / xGMI path alive at all?
  3  N ranks: libsimq's communicator (simq_comm_*: ncclCommInitRank through the hand-declared ABI, library-owned stream, event fork /
     join) -- construction, a probe all-reduce of 1 KB and of the 45 MB gradient buffer against torch.distributed's result, a
     broadcast, simq_comm_progress, and the bus bandwidth of the 45 MB all-reduce (what DESIGN 6's scaling estimate rests on)
  4  the data-parallel tests that skip on a one-GPU box (tests/test_gpu_dp.py -k two_gpus: the fused simq_train_step over simq_comm against
     the reference-replica fixtures, SyncBN through simq_comm_reduce_f64)
  5  bench.py --gpus N --steps 5 (the config BASELINE.json maps to N GPUs + the weak-scaling leg), its JSON line summarised
Replaces the reduce-add of torch.nn.DataParallel (reference policies.py:39) -- the reference has no counterpart of this tool.

usage: python tools/mgpu_selftest.py [--gpus N] [--timeout SECONDS_PER_STAGE]
"""
import argparse
import json
import os
import signal
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')


def worker():
    """One rank of stages 2 / 3 (started by torch.distributed.run)."""
    import ctypes
    sys.path.insert(0, os.path.join(ROOT, 'spatial-intention-maps_amd'))
    import torch
    import torch.distributed as dist
    rank, world, local = int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), int(os.environ['LOCAL_RANK'])
    stage = os.environ['SIMQ_SELFTEST_STAGE']
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    report = {}
    t = torch.full((1024,), float(rank + 1), device=dev)
    dist.all_reduce(t)
    torch.cuda.synchronize(dev)
    report['torch_allreduce_ok'] = bool(torch.all(t == world * (world + 1) / 2))
    if stage == '3':
        from simq import dist as sdist
        comm = sdist.Comm(dist.group.WORLD, dev)
        report['comm_world'] = comm.world_size()
        ok = True
        for n in (256, 11249826):
            a = (torch.arange(n, dtype=torch.float32, device=dev) % 977) * (rank + 1)
            want = a.clone()
            dist.all_reduce(want)
            torch.cuda.synchronize(dev)                 # (never both communicators' collectives in flight at once)
            comm.all_reduce(a)
            comm.wait()
            torch.cuda.synchronize(dev)
            ok = ok and bool(torch.equal(a, want))
        report['comm_allreduce_matches_torch'] = ok
        b = torch.full((4096,), float(rank), device=dev)
        comm.broadcast(b, src=world - 1)
        comm.wait()
        torch.cuda.synchronize(dev)
        report['comm_broadcast_ok'] = bool(torch.all(b == world - 1))
        pr = comm.progress()
        report['comm_progress'] = pr
        # bus bandwidth of the gradient all-reduce: 2 (N-1)/N x bytes / time (ring model)
        g = torch.zeros(11249826, dtype=torch.float32, device=dev)
        for _ in range(3):
            comm.all_reduce(g)
        comm.wait()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            comm.all_reduce(g)
        comm.wait()
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / reps
        report['grad_allreduce_ms'] = round(dt * 1e3, 3)
        report['grad_allreduce_busbw_GBs'] = round(2.0 * (world - 1) / world * g.numel() * 4 / dt / 1e9, 1)
        comm.close()
    every = [None] * world
    dist.all_gather_object(every, report)
    if rank == 0:
        print('SELFTEST_REPORT ' + json.dumps(every), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def run(cmd, timeout, env=None):
    """-> (status, seconds, output tail).  The whole process group is killed on a timeout (a hung collective keeps its ranks alive)."""
    t0 = time.time()
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, start_new_session=True, cwd=ROOT)
    try:
        out, _ = p.communicate(timeout=timeout)
        status = 'PASS' if p.returncode == 0 else 'FAIL (exit %d)' % p.returncode
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)            # the exact group this stage started, nothing else
        except ProcessLookupError:
            pass
        out, _ = p.communicate()
        status = 'HANG (killed after %d s)' % timeout
    return status, time.time() - t0, out or ''


def launch_ranks(n, stage, timeout):
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, SIMQ_SELFTEST_STAGE=stage, OMP_NUM_THREADS='4')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__), '--worker']
    status, secs, out = run(cmd, timeout, env)
    rep = None
    for line in out.splitlines():
        if line.startswith('SELFTEST_REPORT '):
            rep = json.loads(line[len('SELFTEST_REPORT '):])
    return status, secs, out, rep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=0, help='ranks to use (default: every visible GPU)')
    ap.add_argument('--timeout', type=int, default=600, help='seconds per stage')
    ap.add_argument('--worker', action='store_true')
    args = ap.parse_args()
    if args.worker:
        return worker()
    rows = []

    def row(name, status, secs, detail=''):
        rows.append((name, status, secs, detail))
        print('[%s] %-58s %6.1f s  %s' % (status.split()[0], name, secs, detail), flush=True)
    t0 = time.time()
    import torch
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    n = args.gpus or ndev
    try:
        import ctypes
        ctypes.CDLL('librccl.so.1')
        rccl = 'librccl.so.1 loads'
    except OSError as ex:
        rccl = 'librccl.so.1 does NOT load: %s' % ex
    row('1 environment', 'PASS' if (ndev >= 2 and n >= 2 and n <= ndev and 'NOT' not in rccl) else 'FAIL', time.time() - t0,
        '%d GPUs visible, %d ranks requested; %s; HSA_ENABLE_IPC_MODE_LEGACY=%s' % (ndev, n, rccl, os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')))
    if ndev < 2 or n < 2 or n > ndev:
        print('mgpu_selftest: needs at least two visible GPUs -- nothing else can run here')
        return 1
    status, secs, out, rep = launch_ranks(n, '2', args.timeout)
    ok = status == 'PASS' and rep is not None and all(r.get('torch_allreduce_ok') for r in rep)
    row('2 torch.distributed RCCL all-reduce, %d ranks' % n, 'PASS' if ok else (status if status != 'PASS' else 'FAIL'), secs, '' if ok else out[-600:].replace('\n', ' | '))
    status, secs, out, rep = launch_ranks(n, '3', args.timeout)
    ok = status == 'PASS' and rep is not None and all(r.get('comm_allreduce_matches_torch') and r.get('comm_broadcast_ok') and r.get('comm_world') == n for r in rep)
    detail = out[-800:].replace('\n', ' | ')
    if rep:
        detail = '45 MB gradient all-reduce %.2f ms = %.0f GB/s bus bandwidth (rank 0); progress %s' % (
            rep[0].get('grad_allreduce_ms', float('nan')), rep[0].get('grad_allreduce_busbw_GBs', float('nan')), rep[0].get('comm_progress'))
    row('3 libsimq simq_comm (init, all-reduce, broadcast), %d ranks' % n, 'PASS' if ok else (status if status != 'PASS' else 'FAIL'), secs, detail)
    status, secs, out = run([sys.executable, '-m', 'pytest', 'tests/test_gpu_dp.py', '-q', '-m', 'gpu', '-k', 'two_gpus', '-x'], args.timeout)
    tail = [ln for ln in out.splitlines() if 'passed' in ln or 'failed' in ln or 'skipped' in ln or 'error' in ln.lower()][-3:]
    if status == 'PASS' and any('skipped' in ln and 'passed' not in ln for ln in tail):
        status = 'FAIL (skipped)'
    row('4 two-GPU data-parallel tests (fused step over simq_comm, SyncBN)', status, secs, ' | '.join(tail))
    status, secs, out = run([sys.executable, 'bench.py', '--gpus', str(n), '--steps', '5', '--warmup', '2', '--sustained-seconds', '0', '--no-cpu-baseline'], args.timeout)
    detail = out[-600:].replace('\n', ' | ')
    for line in reversed(out.splitlines()):
        if line.startswith('{'):
            try:
                d = json.loads(line)
                detail = '%s: %.0f tr/s on %d GPUs (%s scaling), %.2f ms per step, transport: %s; weak32 leg: %s tr/s' % (
                    d['config']['workload_key'], d['value'], d['n_gpus'], d['scaling'], d['ms_per_step'], d['config']['gradient_transport'],
                    d['config'].get('weak32_full_step_transitions_per_s'))
            except (ValueError, KeyError) as ex:
                detail = 'unparsable bench line (%r)' % (ex,)
            break
    row('5 bench.py --gpus %d --steps 5' % n, status, secs, detail)
    print('\n%-62s %-24s %8s' % ('stage', 'result', 'seconds'))
    for name, status, secs, _ in rows:
        print('%-62s %-24s %8.1f' % (name, status, secs))
    return 0 if all(r[1] == 'PASS' for r in rows) else 1


if __name__ == '__main__':
    sys.exit(main())
