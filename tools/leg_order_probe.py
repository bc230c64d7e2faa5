#!/usr/bin/env python3
"""GPU: why does a workload read 10-13 % low when it is the third of one process (profiles/r06_third_leg_order_effect.txt)?  The configs[1]
step (two nets, a 10 000-transition HBM ring, 30 timed steps) three times in ONE process under
  release+empty   nets and ring dropped and torch.cuda.empty_cache() between the runs (what bench.py does between its legs)
  release         dropped, the caching allocator keeps its blocks
  keep            everything of the earlier runs stays alive
usage: leg_order_probe.py release+empty|release|keep [bf16-in-between]"""
import gc, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'spatial-intention-maps_amd')]
import numpy as np, torch, simq
from simq import synth
from simq.learner import _opt_state, train_step
mode = sys.argv[1] if len(sys.argv) > 1 else 'release+empty'
dev = torch.device('cuda:0')
keep = []


def run(cin, cout, B, precision, items=10000, steps=30):
    torch.manual_seed(1)
    policy, target = simq.FCN(cin, cout, device=dev, precision=precision), simq.FCN(cin, cout, device=dev, precision=precision)
    target.copy_state_from(policy); policy.train(); target.eval()
    ring = simq.DeviceReplayBuffer(items, cin, device=dev)
    for c0 in range(0, items, 1000):
        trs = synth.make_transitions(1000, cin, cout, 5 + c0, terminal_frac=0.1)
        ring.push_many(np.stack([t[0] for t in trs]), [t[1] for t in trs], [t[2] for t in trs],
                       np.stack([t[3] if t[3] is not None else np.zeros_like(t[0]) for t in trs]), [t[3] is None for t in trs])
    opt = _opt_state(policy, None)
    random.seed(3)
    drawn = ring.gather(ring.sample_indices(B))
    def step():
        nonlocal drawn
        info = train_step(policy, target, drawn, 0.75, B, 0.01, 0.9, 1e-4, 100.0, opt_state=opt)
        drawn = ring.gather(ring.sample_indices(B))
        return info
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    if mode == 'keep': keep.append((policy, target, ring, opt, drawn))
    # which of the learner's streams share a HARDWARE queue (kernels of one queue run in order: two spin kernels take twice as long)
    from simq.learner import learner_streams, _upload_stream
    ls = learner_streams(policy)
    names = {"main": torch.cuda.current_stream(dev), "upload": _upload_stream(dev), "side": ls.side, "third": ls.third, "early": ls.early}
    def pair(a, b, cycles=20_000_000):
        torch.cuda.synchronize(); t = time.perf_counter()
        with torch.cuda.stream(names[a]): torch.cuda._sleep(cycles)
        with torch.cuda.stream(names[b]): torch.cuda._sleep(cycles)
        torch.cuda.synchronize(); return time.perf_counter() - t
    torch.cuda.synchronize(); t = time.perf_counter(); torch.cuda._sleep(20_000_000); torch.cuda.synchronize(); one = time.perf_counter() - t
    shared = [a + '+' + b for a, b in (('main', 'upload'), ('main', 'side'), ('main', 'third'), ('main', 'early'), ('side', 'third'), ('side', 'early'), ('third', 'early'), ('upload', 'side'), ('upload', 'third'), ('upload', 'early')) if pair(a, b) > 1.6 * one]
    print('   streams sharing a hardware queue: %s' % (', '.join(shared) or 'none'), flush=True)
    return B * steps / dt


for i in range(4):
    if len(sys.argv) > 2 and i == 1:
        r = run(5, 2, 128, 'bf16'); tag = 'bf16 configs2'
    else:
        r = run(4, 2, 32, 'fp32'); tag = 'fp32 configs1'
    print('%s run %d (%s): %.1f tr/s   allocated %.2f GB reserved %.2f GB' % (mode, i + 1, tag, r, torch.cuda.memory_allocated() / 1e9, torch.cuda.memory_reserved() / 1e9), flush=True)
    gc.collect()
    if mode == 'release+empty': torch.cuda.empty_cache()
