"""Diagnostic (GPU box, not collected by pytest): how much does the SECOND step's loss of the sized fp32 fixtures move when only the
fp32 summation order of the convolutions changes?  Runs tests/test_gpu_sized.py's two-step case with the implicit-GEMM tile forced to
several menu entries (same exact-fp32 products, different order) and prints the step-2 loss against the fixture's fp64 value, and
how many of the step-2 TD targets moved by more than 1e-3 between variants (a flipped greedy next action of double DQN).
usage: python tests/diag_step2_sensitivity.py [case index]"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'spatial-intention-maps_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)

# The product library has no way to force a tile for a whole network (round 5: no process-global switches); the ablation build reads
# SIMQ_IGEMM_TILE once per process, so every variant runs in a child process on libsimq_ablate.so.
if len(sys.argv) > 2 and sys.argv[2] == '--child':
    import simq  # noqa: E402
    from oracle import cases  # noqa: E402
    import test_gpu_sized as T  # noqa: E402
    case = cases.TRAIN_CASES_SIZED[int(sys.argv[1])]
    g = np.load(os.path.join(ROOT, 'tests', 'golden', case[0] + '.npz'))
    r = T.run_two_steps(simq, case, 'fp32')
    l2 = r['info'][1]['loss']
    e2 = abs(l2 - float(g['loss64'][1])) / float(g['loss64'][1])
    dp = T.rl2(r['dparam'], g['dparam64'])
    print('%s tile %-6s step-2 loss %.6f (fp64 %.6f: rel %.4f; reference fp32 rel %.4f)  update err %.5f' % (
        case[0], os.environ.get('SIMQ_IGEMM_TILE', 'auto'), l2, float(g['loss64'][1]), e2, float(g['ref_loss_err'][1]), dp))
    sys.exit(0)

idx = sys.argv[1] if len(sys.argv) > 1 else '1'
for tile in (None, '32x32', '64x64', '96x64', '64x32'):
    env = dict(os.environ, SIMQ_LIBRARY=os.path.join(ROOT, 'spatial-intention-maps_amd', 'simq', 'libsimq_ablate.so'))
    env.pop('SIMQ_IGEMM_TILE', None)
    if tile:
        env['SIMQ_IGEMM_TILE'] = tile
    out = subprocess.run([sys.executable, os.path.abspath(__file__), idx, '--child'], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    print('\n'.join(ln for ln in out.stdout.splitlines() if ' tile ' in ln) or out.stdout[-800:])
