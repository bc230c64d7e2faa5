"""Would an F(3x3,3x3) Winograd form be an accuracy / cost middle ground for the GRAD-MODE forward?  CPU-only simulation.

The grad-mode forward of the fp32 plan keeps F(2x2,3x3) for all layers but layer4's 512->512 (conv_winograd.hip: the forward's round-off is the
gradient's error; F(4x4,3x3) there doubled the gradient study's median).  F(3x3,3x3) -- 25 products per 9 outputs, 24 = 8 x 3 so the maps tile --
would execute 1.44 x fewer matrix FLOPs than F(2x2,3x3).  This probe builds the Cook-Toom matrices for a list of interpolation-point sets
(exact rationals, verified against the direct correlation), runs the fp32 pipeline  U = G w G^T,  V = B^T d B,  M = sum_c V.U (a serial fp32
accumulation over the channels, like the MFMA's FMA chain),  Y = A^T M A  on ReLU'd normal inputs, and reports the error against the fp64 direct
convolution.  Result (profiles/r05_winograd_f3_points.txt): every F(3x3,3x3) point set has the error of the tuned F(4x4,3x3) form (1.3-1.4e-6
rms against 1.5e-6; F(2x2,3x3): 4.5e-7) -- it would cost the gradient what F(4x4) costs it for a third of F(4x4)'s saving.  Not built.

usage: python tools/probes/winograd_points.py [channels]
"""
import sys
from fractions import Fraction as F

import numpy as np


def cook_toom(points, m, r):
    """A^T (m x n), G (n x r), B^T (n x n) of F(m, r) on the finite `points` + infinity, n = m + r - 1 (Toom-Cook / Lavin's construction)"""
    n = m + r - 1
    a = [F(p) for p in points]
    assert len(a) == n - 1
    AT = [[(a[j] ** i if j < n - 1 else (F(1) if i == m - 1 else F(0))) for j in range(n)] for i in range(m)]
    G = []
    for j in range(n - 1):
        N = F(1)
        for l in range(n - 1):
            if l != j:
                N *= a[j] - a[l]
        G.append([a[j] ** k / N for k in range(r)])
    G.append([F(0)] * (r - 1) + [F(1)])
    # B^T from the bilinear identity: sum_j AT[i][j] G[j][k] BT[j][:] = e_{i+k} for every output i and tap k
    C = np.zeros((m * r, n))
    S = np.zeros((m * r, n))
    for i in range(m):
        for k in range(r):
            for j in range(n):
                C[i * r + k, j] = float(AT[i][j] * G[j][k])
            S[i * r + k, i + k] = 1
    BT = np.linalg.lstsq(C, S, rcond=None)[0]
    return AT, G, [[F(float(v)).limit_denominator(1000) for v in row] for row in BT]


def tof(M):
    return np.array([[float(v) for v in row] for row in M])


def exactness(AT, G, BT, m, r):
    rng = np.random.default_rng(0)
    d, g = rng.standard_normal(m + r - 1), rng.standard_normal(r)
    y = tof(AT) @ ((tof(G) @ g) * (tof(BT) @ d))
    return np.abs(y - np.array([sum(d[i + k] * g[k] for k in range(r)) for i in range(m)])).max()


def sim_err(AT, G, BT, m, r, C, tiles=64, seed=1):
    rng = np.random.default_rng(seed)
    n = m + r - 1
    d = np.maximum(rng.standard_normal((tiles, n, n, C)), 0)
    w = rng.standard_normal((r, r, C)) / np.sqrt(r * r * C)
    ref = np.zeros((tiles, m, m))
    for i in range(m):
        for j in range(m):
            ref[:, i, j] = np.einsum('tabc,abc->t', d[:, i:i + r, j:j + r, :], w)
    A32, G32, B32 = (tof(M).astype(np.float32) for M in (AT, G, BT))
    U = np.einsum('ia,abc,jb->ijc', G32, w.astype(np.float32), G32).astype(np.float32)
    V = np.einsum('ia,tabc,jb->tijc', B32, d.astype(np.float32), B32).astype(np.float32)
    M = np.zeros((tiles, n, n), np.float32)
    for c in range(C):
        M = (M + V[..., c] * U[None, :, :, c]).astype(np.float32)
    e = np.einsum('ia,tab,jb->tij', A32, M, A32).astype(np.float32).astype(np.float64) - ref
    return np.abs(e).max() / np.abs(ref).max(), np.sqrt((e ** 2).mean() / (ref ** 2).mean())


SETS = [
    ('F(2x2) {0,1,-1}            [grad-mode forward]', (0, 1, -1), 2),
    ('F(4x4) {0,1,-1,1/2,-2}     [no-grad forwards, dgrads]', (0, 1, -1, F(1, 2), -2), 4),
    ('F(4x4) {0,1,-1,2,-2}       [textbook]', (0, 1, -1, 2, -2), 4),
    ('F(3x3) {0,1,-1,2}', (0, 1, -1, 2), 3),
    ('F(3x3) {0,1,-1,-2}', (0, 1, -1, -2), 3),
    ('F(3x3) {0,1,-1,1/2}', (0, 1, -1, F(1, 2)), 3),
    ('F(3x3) {0,1,-1,-1/2}', (0, 1, -1, F(-1, 2)), 3),
    ('F(3x3) {0,1,-1,3/2}', (0, 1, -1, F(3, 2)), 3),
    ('F(3x3) {0,1,-1,2/3}', (0, 1, -1, F(2, 3)), 3),
    ('F(3x3) {0,1/2,-1/2,1}', (0, F(1, 2), F(-1, 2), 1), 3),
]

if __name__ == '__main__':
    chans = [int(a) for a in sys.argv[1:]] or [256, 512]
    print('# fp32 Winograd forms against the fp64 direct 3x3 convolution, ReLU(N(0,1)) inputs, N(0, 1/(9C)) weights, 64 tiles x 3 seeds')
    print('%-56s %9s  %s' % ('form {finite points} + inf', 'exactness', '  '.join('C=%d max-rel / rms-rel' % c for c in chans)))
    for name, pts, m in SETS:
        AT, G, BT = cook_toom(pts, m, 3)
        cols = []
        for c in chans:
            errs = [sim_err(AT, G, BT, m, 3, c, seed=s) for s in range(3)]
            cols.append('%.2e / %.2e      ' % (np.mean([e[0] for e in errs]), np.mean([e[1] for e in errs])))
        print('%-56s %9.1e  %s' % (name, exactness(AT, G, BT, m, 3), '  '.join(cols)))
