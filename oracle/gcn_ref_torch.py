"""
oracle/gcn_ref_torch.py  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

torch-CPU restatement of the reference's `regular` layer loop with the SAME library ops the reference issues
(dense `bmm` over the [B,T,T] adjacency, two `F.linear` per layer, `/ denom`, `relu`, autograd for the backward),
so that its timing on the GPU box's host cores stands for what the reference's own CPU path costs there
(the reference itself cannot travel to that box).  Used only by bench.py's `cpu_baseline` leg and by
tests/test_oracle_golden.py, which pins it to tests/golden/layers_c1_l2.npz (recorded from the live reference).

Reference lines followed (model/gcn.py):
  260-262  adj_matrix = where(adj != 0, 1, 0); denom = adj_matrix.sum(2) + 1; mask = (rowsum + colsum).eq(0)
  269-271  Ax = adj_matrix.bmm(h); AxW = W[l](Ax); AxW = AxW + W[l](h)
  390-393  AxW / denom; relu; dropout on every layer but the last (p = 0 here: parity / timing runs)
"""
import torch
import torch.nn.functional as F


def prep(adj):
    """gcn.py:260-262 on a float32 [B,T,T] tensor with deprel labels."""
    A = torch.where(adj != 0, torch.ones_like(adj), torch.zeros_like(adj)).type(torch.float32)
    denom = A.sum(2).unsqueeze(2) + 1
    mask = (A.sum(2) + A.sum(1)).eq(0).unsqueeze(2)
    return A, denom, mask


def forward(A, denom, x, weights, biases):
    """gcn.py:266-271, 390-393 (gcn_dropout = 0)."""
    h = x
    for W, b in zip(weights, biases):
        Ax = A.bmm(h)
        AxW = F.linear(Ax, W, b)
        AxW = AxW + F.linear(h, W, b)
        h = F.relu(AxW / denom)
    return h


def forward_backward(adj, x, weights, biases, gy):
    """One fwd+bwd of the stack the way the reference's training step runs it: returns (h, dx, [dW], [db]) as tensors."""
    A, denom, _ = prep(adj)
    x = x.detach().clone().requires_grad_(True)
    Ws = [w.detach().clone().requires_grad_(True) for w in weights]
    bs = [b.detach().clone().requires_grad_(True) for b in biases]
    h = forward(A, denom, x, Ws, bs)
    h.backward(gy)
    return h.detach(), x.grad, [w.grad for w in Ws], [b.grad for b in bs]
