"""Shared by the GPU parity tests: the SECOND train() call checked as what it is -- one evaluation of train.py:114-129 from the
state the first call left.

Comparing the second loss with the fp64 TRAJECTORY measures the chaos of the synthetic problem, not the implementation: with nothing
but the fp32 summation order of the convolutions changed (the tile menu forced to 32x32 / 64x64 / 96x64) the b64 fixture's second
loss moves by 2.2 .. 7.0 % of itself (tests/diag_step2_sensitivity.py), and the reference's own fp32 is 0.07 .. 0.9 % off the fp64
trajectory on the same fixtures.  So the fp64 oracle is run from the HIP path's OWN post-step-1 state dict (parameters + buffers)
on the same batch -- three forwards, no backward -- and q_sa / the TD targets of the second HIP step must match it per transition.
A TD target may differ only through a TIE of the double-DQN greedy action (train.py:121): then the HIP value must be the target
net's value at an action whose policy value is within `tol` of the maximum."""
import numpy as np
import torch

from oracle import cases
from oracle import fcn as ofcn
from oracle import learner as olearner


def snapshot(net):
    return {k: v.detach().clone().cpu() for k, v in net.state_dict().items()}


def second_step_against_the_oracle(sd1, sd_target, batch, q_sa2, y2, info2, tol=1e-4, max_ties=2, double_dqn=True):
    """sd1 / sd_target: snapshot() of the policy net after the first call / of the target net; q_sa2, y2: policy._last of the second
    call; info2: what it returned.  Asserts the bars; returns (q error, worst TD-target error over the untied transitions, ties)."""
    B = len(batch.state)
    f64 = lambda sd: {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
    st, tg = f64(sd1), f64(sd_target)
    state_b = torch.cat([olearner.apply_transform(s) for s in batch.state]).double()
    act = torch.tensor(batch.action, dtype=torch.long)
    rew = np.asarray(batch.reward, np.float64)
    nf = torch.cat([olearner.apply_transform(s) for s in batch.next_state if s is not None]).double()
    mask = np.array([s is not None for s in batch.next_state])
    # (the GPU box's host reports 256 CPUs; MKL-DNN is at its best around 32 threads there and many times slower with all of them)
    threads = torch.get_num_threads()
    torch.set_num_threads(min(32, threads))
    try:
        return _second_step(st, tg, state_b, act, rew, nf, mask, B, q_sa2, y2, info2, tol, max_ties, double_dqn)
    finally:
        torch.set_num_threads(threads)


def _second_step(st, tg, state_b, act, rew, nf, mask, B, q_sa2, y2, info2, tol, max_ties, double_dqn):
    with torch.no_grad():
        q = ofcn.fcn_forward(st, state_b, True).reshape(B, -1).gather(1, act.unsqueeze(1)).squeeze(1).numpy()
        qp = ofcn.fcn_forward(st, nf, True).reshape(nf.size(0), -1).numpy() if double_dqn else None
        qt = ofcn.fcn_forward(tg, nf, False).reshape(nf.size(0), -1).numpy()
    if qp is None:
        qp = qt                                                               # vanilla DQN: max over the target net's own map
    q_sa2, y2 = np.asarray(q_sa2, np.float64), np.asarray(y2, np.float64)
    e_q = float(np.abs(q_sa2 - q).max() / np.abs(q).max())
    v_hip = (y2 - rew) / cases.GAMMA
    assert np.abs(v_hip[~mask]).max(initial=0.0) < 1e-6, 'terminal transitions: y = reward'
    best = qp.argmax(1)
    v_or = qt[np.arange(len(best)), best]
    scale_t, scale_p = np.abs(qt).max(), np.abs(qp).max()
    err = np.abs(v_hip[mask] - v_or) / scale_t
    ties = 0
    for j in np.nonzero(err > tol)[0]:
        cand = np.nonzero(np.abs(qt[j] - v_hip[mask][j]) <= tol * scale_t)[0]
        assert len(cand) and (qp[j, cand] >= qp[j].max() - tol * scale_p).any(), \
            'transition %d: TD target off by %.3g and not a tie of the greedy action' % (j, err[j])
        ties += 1
    e_y = float(err[err <= tol].max(initial=0.0))
    d = q_sa2 - y2
    huber = float(np.where(np.abs(d) < 1.0, 0.5 * d * d, np.abs(d) - 0.5).mean())
    assert e_q < tol and ties <= max_ties, (e_q, e_y, ties)
    assert abs(info2['loss'] - huber) <= 1e-5 * huber and abs(info2['td_error'] - float(np.abs(d).mean())) <= 1e-5 * float(np.abs(d).mean())
    return e_q, e_y, ties
