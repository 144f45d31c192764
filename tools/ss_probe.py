#!/usr/bin/env python3
"""Scheduled-sampling draw chain, kernel by kernel (event-timed, eager): the gathered logits product of the fired rows, the list
multinomial, the step embedding, the gathered / scattered x -> gates product -- against the dense forms the in-line path launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sub-gc_amd"), ROOT]
import torch
from subgc import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
S, V1, R, E = 640, 9488, 1000, 1000
ops.ensure_workspace(dev)
H = torch.randn(S, R, device=dev); W = torch.randn(V1, R, device=dev) * 0.03; b = torch.zeros(V1, device=dev)
Wx = torch.randn(4 * R, E, device=dev) * 0.03
xt = torch.randn(S, E, device=dev); Gx = torch.empty(S, 4 * R, device=dev)
out = torch.empty(S, V1, device=dev)
sel = torch.rand(2, S, device=dev)
live = torch.tensor([S, S], dtype=torch.int32, device=dev)


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    c.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(c) / n


for prob in (0.05, 0.25):
    fired, cnt = ops.ss_plan(sel, live, prob)
    ft, ct = fired[1], cnt[1:2]
    k = int(ct.item())
    u = torch.rand(S, device=dev); tok = torch.zeros(S, dtype=torch.long, device=dev)
    print(f"p={prob}: {k} fired rows of {S}")
    print("  logits gathered (a_rows, m_dev) %7.1f us   dense M=%d %7.1f us   dense M=640 %7.1f us" % (
        timed(lambda: ops.gemm(H, W, out, tb=True, bias=b, a_rows=ft, m_dev=ct)), k,
        timed(lambda: ops.gemm(H[:k], W, out[:k], tb=True, bias=b)), timed(lambda: ops.gemm(H, W, out, tb=True, bias=b))))
    print("  multinomial list %7.1f us   all rows %7.1f us" % (
        timed(lambda: ops.multinomial_rows_list_(out, ft, ct, u, tok)), timed(lambda: ops.multinomial_rows_(out, u, sel[1], prob, tok))))
    print("  x->gates gathered+scattered %7.1f us   dense M=%d %7.1f us   dense M=640 %7.1f us" % (
        timed(lambda: ops.gemm(xt, Wx, Gx, tb=True, a_rows=ft, c_rows=ft, m_dev=ct)), k,
        timed(lambda: ops.gemm(xt[:k], Wx, Gx[:k], tb=True)), timed(lambda: ops.gemm(xt, Wx, Gx, tb=True))))
