#!/usr/bin/env python3
"""The fused decode LSTM step (subgc_lstm_step_skinny) in isolation: att-LSTM (K = 2R) and lang-LSTM (K = 3R) alternate
inside one captured graph, as in the decode loop (both weight matrices stay resident in the Infinity Cache).
    python tools/ubench/lstm_step_bench.py [S] [R]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "sub-gc_amd"))
import torch  # noqa: E402

from subgc import ops  # noqa: E402
from subgc.ops import _ptr, _stream, call, ld  # noqa: E402


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    dev = torch.device("cuda:0")
    ops.ensure_workspace(dev)
    torch.manual_seed(0)
    for b16 in (0, 1):
        dt = torch.bfloat16 if b16 else torch.float32
        Ws = {K: (torch.randn(4 * R, K, device=dev) * 0.02).to(dt) for K in (2 * R, 3 * R)}
        xs = {K: torch.randn(16, K, device=dev) for K in Ws}
        b0 = torch.randn(4 * R, device=dev)
        c0, c1 = torch.randn(16, R, device=dev), torch.empty(16, R, device=dev)
        h = torch.empty(16, R, device=dev)
        ref = None
        for variant in (0,):
            def go(K):
                x, W = xs[K], Ws[K]
                call("subgc_lstm_step_skinny", _ptr(x, torch.float32), ld(x), _ptr(W), ld(W), K, S, R, 0, 0, 0, 0, 0, 0, _ptr(b0, torch.float32), 0,
                     _ptr(c0, torch.float32), _ptr(c1, torch.float32), _ptr(h, torch.float32), ld(h), 0, 0, 0, 0, b16, _stream())
            go(2 * R)
            torch.cuda.synchronize()
            if ref is None:
                ref = h[:S].clone()
            err = float((h[:S] - ref).abs().max())
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for _ in range(2):
                    go(2 * R); go(3 * R)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s):
                    for _ in range(20):
                        go(2 * R); go(3 * R)
            torch.cuda.synchronize()
            for _ in range(3):
                g.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            print(f"bf16 weights={b16}  {1e3 * e0.elapsed_time(e1) / 200:.2f} us per (att + lang) pair   err {err:.2e}", flush=True)


main()
