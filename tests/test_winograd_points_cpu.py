"""tools/probes/winograd_points.py (the Cook-Toom generator behind DESIGN 7's "F(3x3,3x3) is as noisy as F(4x4,3x3)") is checked against
the transform matrices the kernels use: it must reproduce the F(4x4,3x3) matrices of conv_winograd.hip exactly, and every form it builds must
satisfy the bilinear identity of the 3-tap correlation (reference: the 3x3 / stride-1 nn.Conv2d of resnet.py:19-20)."""
import os
import sys
from fractions import Fraction as F

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools', 'probes'))
import winograd_points as wp  # noqa: E402

H = F(1, 2)
# conv_winograd.hip, "F(4x4, 3x3) for the NO-GRAD forwards": interpolation points {0, 1, -1, 1/2, -2, inf}
AT4 = [[1, 1, 1, 1, 1, 0], [0, 1, -1, H, -2, 0], [0, 1, 1, F(1, 4), 4, 0], [0, 1, -1, F(1, 8), -8, 1]]
G4 = [[1, 0, 0], [F(1, 3)] * 3, [F(-1, 3), F(1, 3), F(-1, 3)], [F(-16, 15), F(-8, 15), F(-4, 15)], [F(1, 15), F(-2, 15), F(4, 15)], [0, 0, 1]]
BT4 = [[1, F(-3, 2), -2, F(3, 2), 1, 0], [0, -1, H, F(5, 2), 1, 0], [0, 1, F(-5, 2), H, 1, 0], [0, -2, -1, 2, 1, 0], [0, H, -1, -H, 1, 0],
       [0, 1, F(-3, 2), -2, F(3, 2), 1]]


def test_generator_reproduces_the_f4_matrices_of_the_kernels():
    AT, G, BT = wp.cook_toom((0, 1, -1, H, -2), 4, 3)
    for got, want in ((AT, AT4), (G, G4), (BT, BT4)):
        assert [[F(v) for v in row] for row in got] == [[F(v) for v in row] for row in want]


def test_the_f2_matrices_of_the_kernels_are_the_generated_ones_up_to_row_signs():
    # conv_winograd.hip: G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]], A^T = [[1,1,1,0],[0,1,-1,-1]]
    AT, G, BT = wp.cook_toom((0, 1, -1), 2, 3)
    Gk = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]])
    BTk = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]])
    ATk = np.array([[1, 1, 1, 0], [0, 1, -1, -1]])
    sg = np.array([-1, 1, 1, 1])        # a sign on element j of G and of B^T (or of A^T's column j) cancels in A^T [(G g) . (B^T d)]
    assert np.array_equal(wp.tof(G) * sg[:, None], Gk)
    sb = np.array([-1, 1, 1, -1])
    assert np.array_equal(wp.tof(BT) * sb[:, None], BTk)
    assert np.array_equal(wp.tof(AT) * (sg * sb)[None, :], ATk)


@pytest.mark.parametrize('name,points,m', wp.SETS)
def test_every_form_is_an_exact_correlation(name, points, m):
    AT, G, BT = wp.cook_toom(points, m, 3)
    assert wp.exactness(AT, G, BT, m, 3) < 1e-14
    # two dimensions: Y = A^T [(G g G^T) . (B^T d B)] A against the direct 3x3 correlation
    rng = np.random.default_rng(3)
    n = m + 2
    d, g = rng.standard_normal((n, n)), rng.standard_normal((3, 3))
    A, Gm, B = wp.tof(AT), wp.tof(G), wp.tof(BT)
    y = A @ ((Gm @ g @ Gm.T) * (B @ d @ B.T)) @ A.T
    ref = np.array([[(d[i:i + 3, j:j + 3] * g).sum() for j in range(m)] for i in range(m)])
    assert np.abs(y - ref).max() < 1e-12


def test_f3_is_no_quieter_than_the_tuned_f4():
    """the finding DESIGN 7 quotes: fp32 round-off of every F(3x3,3x3) point set >= 0.8 x the tuned F(4x4,3x3) form's, ~3 x F(2x2,3x3)'s"""
    def rms(points, m):
        AT, G, BT = wp.cook_toom(points, m, 3)
        return np.mean([wp.sim_err(AT, G, BT, m, 3, 128, tiles=32, seed=s)[1] for s in range(2)])
    f2, f4 = rms((0, 1, -1), 2), rms((0, 1, -1, H, -2), 4)
    f3 = min(rms(p, 3) for _, p, m in wp.SETS if m == 3)
    assert f3 > 0.8 * f4 and f3 > 2.0 * f2, (f2, f3, f4)
