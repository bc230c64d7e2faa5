"""CPU oracle for the spatial-action-map DQN hot path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (PyTorch-CPU fp32 / fp64 tensors, no GPU, no
HIP) of the reference algorithm on the hot path named by BASELINE.json:

  * ``oracle.fcn``      <- /root/reference/networks.py:6-26, resnet.py:19-104
  * ``oracle.learner``  <- /root/reference/train.py:26-45 (Transition, ReplayBuffer),
                           train.py:108-141 (train), torch.nn.utils.clip_grad_norm_,
                           torch.optim.SGD as configured at train.py:186
  * ``oracle.policy``   <- /root/reference/policies.py:35-74 (build_policy_nets,
                           apply_transform, step)

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker / the timed CPU baseline -- never
as a product code path.  The product (``simq``) fails loudly when its HIP
library is missing; it never falls back to this package.

Pinning: the reference ships no tests / golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned against the reference itself,
imported in the build container by ``oracle/gen_golden.py`` (bit-exact
comparison, then small fixtures are written to ``tests/golden/``).
"""
