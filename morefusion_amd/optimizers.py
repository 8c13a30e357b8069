"""Chainer-compatible Adam for torch parameters.

The ICC / ICP trajectories of the reference depend on ``chainer.optimizers.Adam``
(third party; call sites examples/ycb_video/pose_refinement/
check_iterative_collision_check_link.py:48-50, check_iterative_closest_point_link.py:40-43),
whose update differs from ``torch.optim.Adam`` in where eps enters:

    m += (1 - b1) (g - m);  v += (1 - b2) (g*g - v)
    p -= alpha_t * m / (sqrt(v) + eps),   alpha_t = alpha sqrt(1 - b2^t) / (1 - b1^t)

with alpha_t evaluated in double precision.  Restated from Chainer v7 -- parity
unpinned (no reference test pins it).  The fused on-device loop (k_icc_step in
csrc/icc.hip) implements the same rule; this class serves the step-by-step API
(``loss.backward(); optimizer.update(); link.zerograds()``), including the
reference's per-parameter ``param.update_rule.hyperparam.alpha *= 0.1`` idiom.
"""
import math

import torch


class _Hyperparam:
    def __init__(self, alpha, beta1, beta2, eps):
        self.alpha, self.beta1, self.beta2, self.eps = alpha, beta1, beta2, eps


class _UpdateRule:
    def __init__(self, hp):
        self.hyperparam = _Hyperparam(hp.alpha, hp.beta1, hp.beta2, hp.eps)
        self.t = 0
        self.m = None
        self.v = None


class Adam:
    def __init__(self, alpha=0.001, beta1=0.9, beta2=0.999, eps=1e-8):
        self.hyperparam = _Hyperparam(alpha, beta1, beta2, eps)
        self.target = None
        self.t = 0

    def setup(self, link):
        self.target = link
        for p in link.parameters():
            p.update_rule = _UpdateRule(self.hyperparam)
        return self

    @torch.no_grad()
    def update(self):
        self.t += 1
        for p in self.target.parameters():
            if p.grad is None:
                continue
            rule = p.update_rule
            hp = rule.hyperparam
            rule.t += 1
            if rule.m is None:
                rule.m = torch.zeros_like(p)
                rule.v = torch.zeros_like(p)
            g = p.grad
            fix1 = 1.0 - math.pow(hp.beta1, rule.t)
            fix2 = 1.0 - math.pow(hp.beta2, rule.t)
            alpha_t = hp.alpha * math.sqrt(fix2) / fix1
            rule.m += (1 - hp.beta1) * (g - rule.m)
            rule.v += (1 - hp.beta2) * (g * g - rule.v)
            p -= alpha_t * rule.m / (torch.sqrt(rule.v) + hp.eps)
