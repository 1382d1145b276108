"""Stock-PyTorch eager twin of the lifter.  TEST INFRASTRUCTURE ONLY.

An nn.Module assembled from stock torch.nn layers with the same architecture,
parameter names and registration order as LinearModel
(/root/reference/phase1_lifting/baselineModel.py:50-102) and a restated
train_1.py:75-100 step.  It dispatches to exactly the ATen CPU kernels the
reference dispatches to, so it serves as
  * the CPU baseline that bench.py times on the GPU node's host cores
    (cpu_baseline.kind == "port"), and
  * a second, independent cross-check of the numpy oracle.
The reference file itself cannot travel to the GPU box; this twin can.
"""
import torch
from torch import nn


class _ResidualPair(nn.Module):
    """Two Linear->BN->ReLU->Dropout groups with an identity skip
    (baselineModel.py:14-47)."""

    def __init__(self, width, p, bn):
        super().__init__()
        self.w1 = nn.Linear(width, width)
        self.batch_norm1 = nn.BatchNorm1d(width)
        self.w2 = nn.Linear(width, width)
        self.batch_norm2 = nn.BatchNorm1d(width)
        self._p, self._bn = p, bn

    def _group(self, lin, norm, t):
        t = lin(t)
        if self._bn:
            t = norm(t)
        return nn.functional.dropout(torch.relu(t), self._p, self.training)

    def forward(self, t):
        return t + self._group(self.w2, self.batch_norm2,
                               self._group(self.w1, self.batch_norm1, t))


class TwinLifter(nn.Module):
    def __init__(self, i_dim, o_dim, linear_size=1024, num_stage=2, p_dropout=0.5, BN=True):
        super().__init__()
        self.w1 = nn.Linear(i_dim, linear_size)
        self.batch_norm1 = nn.BatchNorm1d(linear_size)
        self.linear_stages = nn.ModuleList(
            _ResidualPair(linear_size, p_dropout, BN) for _ in range(num_stage))
        self.w2 = nn.Linear(linear_size, o_dim)
        self._p, self._bn = p_dropout, BN

    def forward(self, x):
        t = self.w1(x.flatten(1))
        if self._bn:
            t = self.batch_norm1(t)
        t = nn.functional.dropout(torch.relu(t), self._p, self.training)
        for blk in self.linear_stages:
            t = blk(t)
        return self.w2(t)


def twin_train_step(model, optimizer, y1, y2):
    """train_1.py:75-100 restated: zero_grad, forward, reshape, MSE(mean), backward, step."""
    optimizer.zero_grad()
    pred = model(y1).reshape(y1.shape[0], -1, 3)
    loss = nn.functional.mse_loss(pred, y2)
    loss.backward()
    optimizer.step()
    return loss, pred
