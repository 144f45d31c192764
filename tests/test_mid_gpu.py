"""The fused middle of a train-decoder step (csrc/recurrent_mid.hip: subgc_mid_fwd / subgc_mid_bwd) against the three launches it
replaces -- subgc_lstm_fwd + the h2att product + subgc_attn_fwd (reference: AttModel.py:411-413, 453-466) -- on the same inputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _sets(S, nmax, gen, dev):
    lens = torch.from_numpy(gen.integers(0, nmax + 1, size=S).astype(np.int32))
    lens[0] = nmax                                                   # the longest set is present; some are empty
    off = torch.zeros(S, dtype=torch.int32)
    off[1:] = torch.cumsum(lens[:-1], 0)
    return lens.to(dev), off.to(dev), int(lens.sum())


@pytest.mark.parametrize("bf", [False, True])
@pytest.mark.parametrize("S,R,A,nmax,parts", [(7, 64, 32, 5, 1), (300, 1000, 512, 11, 3), (1000, 1000, 512, 9, 2), (129, 1000, 512, 37, 2), (33, 200, 96, 101, 1)])
def test_mid_fwd_matches_the_three_launches(S, R, A, nmax, parts, bf):
    from subgc import ops
    dev = torch.device("cuda:0")
    gen = np.random.default_rng(S * 7 + R)
    rnd = lambda *s, sc=1.0: torch.from_numpy((gen.standard_normal(s) * sc).astype(np.float32)).to(dev)
    lens, off, total = _sets(S, nmax, gen, dev)
    planes = rnd(parts, S, 4 * R, sc=0.5)
    g1, g2, b0, b1 = rnd(S, 4 * R, sc=0.3), rnd(S, 4 * R, sc=0.3), rnd(4 * R, sc=0.1), rnd(4 * R, sc=0.1)
    c_prev = rnd(S, R)
    wq, bq, w_a, b_a = rnd(A, R, sc=R ** -0.5), rnd(A, sc=0.1), rnd(A, sc=0.3), rnd(1)
    u, v = rnd(max(total, 1), A), rnd(max(total, 1), R)
    if bf:
        wq_, u_, v_ = ops.as_b16(wq), ops.as_b16(u), ops.as_b16(v)
    else:
        wq_, u_, v_ = wq, u, v
    rows_h2 = max(S - 3, 1)

    def buffers():
        H = ops.act_padded((S, 3 * R), dev, bf, zero_rows=S)
        Hn = ops.act_padded((S, 2 * R), dev, bf, zero_rows=S)
        return dict(H=H, Hn=Hn, c=torch.zeros(S, R, device=dev), G=torch.zeros(S, 4 * R, device=dev), q=torch.zeros(S, A, device=dev),
                    al=torch.zeros(S, nmax + 2, device=dev))

    a, b = buffers(), buffers()
    # the three launches
    ops.lstm_fwd(planes[0] if parts == 1 else planes.sum(0), g1, g2, b0, b1, c_prev, a["c"], a["H"][:, R:2 * R], a["Hn"][:, R:], None, 1.0, None, a["G"],
                 S, R, rows_h=S, rows_h2=rows_h2)
    QP = torch.empty(8 * S * A, device=dev)
    nq, sq = ops.gemm_planes(a["H"][:, R:2 * R], wq_, QP, tb=True)
    ops.attn_fwd(u_, v_, a["q"], w_a, b_a, off, lens, a["H"][:, :R], a["al"], S, A, R, q=(QP, nq, sq, bq))
    # one launch
    wq_mid = wq_ if bf else ops.transpose_f32(wq)                   # fp32 operands: the K-major copy
    ops.mid_fwd(planes[0], parts, S * 4 * R, g1, g2, b0, b1, c_prev, b["c"], b["H"][:, R:2 * R], b["Hn"][:, R:], b["G"], wq_mid, bq, b["q"], u_, v_, w_a, b_a,
                off, lens, b["H"][:, :R], b["al"], S, R, A, rows_h=S, rows_h2=rows_h2)
    torch.cuda.synchronize()
    f = lambda t: t.float()
    # the cell update is the same arithmetic on the same numbers (up to the compiler's choice of fused multiply-adds)
    torch.testing.assert_close(a["c"], b["c"], atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(a["G"], b["G"], atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(f(a["H"][:, R:2 * R]), f(b["H"][:, R:2 * R]), atol=1e-2 if bf else 1e-5, rtol=1e-2 if bf else 1e-5)
    torch.testing.assert_close(f(a["Hn"]), f(b["Hn"]), atol=1e-2 if bf else 1e-5, rtol=1e-2 if bf else 1e-5)
    tol = dict(atol=2e-3, rtol=2e-3) if bf else dict(atol=2e-5, rtol=2e-5)
    torch.testing.assert_close(a["q"], b["q"], **(dict(atol=1e-4, rtol=1e-4) if not bf else tol))
    torch.testing.assert_close(a["al"], b["al"], **(dict(atol=1e-4, rtol=1e-3) if not bf else tol))
    torch.testing.assert_close(f(a["H"][:, :R]), f(b["H"][:, :R]), **(dict(atol=1e-4, rtol=1e-3) if not bf else dict(atol=2e-2, rtol=2e-2)))
    assert torch.all(b["al"].sum(1)[lens > 0].sub(1).abs() < 1e-5)


def test_transpose_f32_and_refusals():
    """subgc_transpose_f32 (the K-major weight of the fp32 fused product) on shapes that are not multiples of its 64 x 64 tile; subgc_mid_fwd
    refuses what it does not cover with an error instead of a wrong answer (fp32 operands beyond 1024 rows, query widths above 512)."""
    from subgc import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(3)
    for rows, cols in ((512, 1000), (65, 3), (1, 130)):
        x = torch.randn(rows, cols, generator=g).to(dev)
        assert torch.equal(ops.transpose_f32(x), x.t().contiguous())
    S, R, A = 1100, 64, 32
    z = lambda *s: torch.zeros(*s, device=dev)
    lens, off = torch.ones(S, dtype=torch.int32, device=dev), torch.arange(S, dtype=torch.int32, device=dev)
    with pytest.raises(ops.SubgcError, match="not covered"):
        ops.mid_fwd(z(S, 4 * R), 1, 0, None, None, None, None, None, z(S, R), z(S, 3 * R)[:, R:2 * R], None, z(S, 4 * R), z(R, A), z(A), z(S, A), z(S, A), z(S, R),
                    z(A), z(1), off, lens, z(S, 3 * R)[:, :R], z(S, 2), S, R, A)


def test_cell_kernels_report_moved_bytes():
    """subgc_prof_last_moved: the LSTM cell launches report the bytes they really move (planes, gate terms, saved gates) beside the 12-float
    algorithmic count of the bench's hbm_bound_kernels block."""
    from subgc import _lib, ops
    dev = torch.device("cuda:0")
    S, R = 64, 128
    z = lambda *s: torch.zeros(*s, device=dev)
    _lib.prof_enable("lstm", True)
    ops.lstm_fwd(z(S, 4 * R), z(S, 4 * R), z(S, 4 * R), z(4 * R), z(4 * R), z(S, R), z(S, R), z(S, R), z(S, R), None, 1.0, None, z(S, 4 * R), S, R)
    torch.cuda.synchronize()
    _lib.prof_enable("lstm", False)
    n, ms, work = _lib.prof_collect("lstm")
    moved = _lib.prof_last_moved("lstm")
    assert n == 1 and work == 4.0 * S * R * 12
    assert moved == S * R * (4.0 * 4 * 3 + 4 + 4 + 4 * 2 + 16)          # three 4-gate inputs, c_prev, c, two fp32 h copies, the saved gates
