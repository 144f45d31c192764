"""subgc_gemm_f32_wgrad / subgc_gemm_bf16_wgrad: a linear layer's weight gradient dY^T X and bias gradient sum_k dY[k, :] in ONE launch
(the workgroups of dW's tile column 0 add up the dY tiles they stage anyway) against fp64 of the same operands, over every form the
dispatch picks: 128 x 128 and 64 x 64 tiles, whole and split K, 256 x 256 bf16 tiles, ragged K (device-side row count), scalar
(unaligned) operands, accumulation into both destinations.  Reference: autograd's Linear backward (grad_output.t().mm(input),
grad_output.sum(0)) under every nn.Linear / nn.LSTMCell of models/AttModel.py and models/lib/graph_conv_unit.py."""
import pytest
import torch

from subgc import ops
import subgc.functions as F_

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


def rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)).to(DEV)


# (rows K, out M, in N): logit layer, LSTM gates, GCN fc, h2att, small / odd shapes, one K-tile, K not a multiple of 8 / 32
SHAPES = [(7000, 9488, 1000), (7000, 4000, 1000), (16640, 1024, 512), (4736, 512, 1024), (640, 512, 1000), (333, 200, 72), (31, 64, 64),
          (1000, 132, 36), (333, 201, 70), (5, 8, 8), (2176, 4000, 2000), (12000, 1024, 1024)]


@pytest.mark.parametrize("K,M,N", SHAPES)
@pytest.mark.parametrize("store", ["f32", "bf16"])
def test_wgrad_matches_fp64(K, M, N, store):
    pad = lambda n: n + 8 + (-n) % 8                            # a leading dimension that is not the width; bf16 rows start 16-byte aligned
    dy, x = rnd(K, pad(M), seed=K + M), rnd(K, pad(N), seed=K + N + 1)
    if store == "bf16":
        dy, x = dy.to(BF), x.to(BF)
    dy, x = dy[:, :M], x[:, :N]
    wantW, wantb = dy.double().t() @ x.double(), dy.double().sum(0)
    dW, db = torch.full((M, N), float("nan"), device=DEV), torch.full((M,), float("nan"), device=DEV)
    ops.wgrad(dy, x, dW, db)
    sw, sb = float(wantW.abs().max()), float(wantb.abs().max())
    assert float((dW.double() - wantW).abs().max()) < 2e-5 * sw
    assert float((db.double() - wantb).abs().max()) < 2e-5 * max(sb, K ** 0.5)
    # accumulate into both
    dW2, db2 = rnd(M, N, seed=3), rnd(M, seed=4)
    w0, b0 = dW2.clone(), db2.clone()
    ops.wgrad(dy, x, dW2, db2, accum=True, db_accum=True)
    assert float((dW2.double() - w0.double() - wantW).abs().max()) < 2e-5 * sw
    assert float((db2.double() - b0.double() - wantb).abs().max()) < 2e-5 * max(sb, K ** 0.5)
    # dW written, db accumulated (the flags are independent)
    db3 = b0.clone()
    ops.wgrad(dy, x, dW2, db3, accum=False, db_accum=True)
    assert torch.equal(dW2, dW)
    assert torch.equal(db3, db2)


@pytest.mark.parametrize("store", ["f32", "bf16"])
@pytest.mark.parametrize("live", [0, 1, 700, 2999, 3000])
def test_wgrad_ragged_rows(store, live):
    """m_dev bounds the rows (the packed attention rows of prepared_backward): rows beyond it hold garbage and must not be read into either sum."""
    K, M, N = 3000, 512, 1024
    dy, x = rnd(K, M, seed=1), rnd(K, N, seed=2)
    dy[live:] = float("nan")
    x[live:] = float("nan")
    if store == "bf16":
        dy, x = dy.to(BF), x.to(BF)
    m_dev = torch.tensor([live], device=DEV, dtype=torch.int32)
    wantW, wantb = dy[:live].double().t() @ x[:live].double(), dy[:live].double().sum(0)
    dW, db = torch.full((M, N), float("nan"), device=DEV), torch.full((M,), float("nan"), device=DEV)
    ops.wgrad(dy, x, dW, db, m_dev=m_dev)
    assert float((dW.double() - wantW).abs().max()) < 2e-5 * max(float(wantW.abs().max()), 1.0)
    assert float((db.double() - wantb).abs().max()) < 2e-5 * max(float(wantb.abs().max()), 1.0)


def test_wgrad_is_reproducible_and_equals_the_two_pass_form():
    """Fixed summation order (no float atomics): two calls agree bit for bit; the weight gradient is the plain product's, bit for bit
    (same kernel, same plan), the bias gradient the column-sum kernel's within rounding."""
    for store in ("f32", "bf16"):
        K, M, N = 7000, 4000, 1000
        dy, x = rnd(K, M, seed=5), rnd(K, N, seed=6)
        if store == "bf16":
            dy, x = dy.to(BF), x.to(BF)
        a = ops.wgrad(dy, x, torch.empty(M, N, device=DEV), torch.empty(M, device=DEV))
        b = ops.wgrad(dy, x, torch.empty(M, N, device=DEV), torch.empty(M, device=DEV))
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        ref = ops.gemm(dy, x, torch.empty(M, N, device=DEV), ta=True)
        assert torch.equal(a[0], ref)
        col = ops.colsum(dy)
        assert float((a[1] - col).abs().max()) < 1e-5 * float(col.abs().max())


def test_wgrad_under_the_opt_in_arithmetics_runs_as_two_passes():
    K, M, N = 2000, 1024, 512
    dy, x = rnd(K, M, seed=7), rnd(K, N, seed=8)
    wantb = dy.double().sum(0)
    for mode in ("bf16x3", "bf16"):
        with ops.gemm_mode(mode):
            dW, db = ops.wgrad(dy, x, torch.empty(M, N, device=DEV), torch.empty(M, device=DEV))
            ref = ops.gemm(dy, x, torch.empty(M, N, device=DEV), ta=True)
        assert torch.equal(dW, ref)
        assert float((db.double() - wantb).abs().max()) < 2e-5 * float(wantb.abs().max())


def test_linear_backward_is_the_same_with_and_without_the_fold():
    for bf in (False, True):
        torch.manual_seed(0)
        x = rnd(4736, 1024, seed=1).requires_grad_()
        W, b = rnd(512, 1024, seed=2).mul_(0.03).requires_grad_(), rnd(512, seed=3).requires_grad_()
        W16 = ops.as_b16(W.detach()) if bf else None
        g = rnd(4736, 512, seed=4)
        out = {}
        for fold in (True, False):
            F_.FOLD_BIAS_SUMS = fold
            try:
                for t in (x, W, b):
                    t.grad = None
                y = F_.linear(x, W, b, relu=True, W16=W16)
                y.backward(g)
                out[fold] = (x.grad.clone(), W.grad.clone(), b.grad.clone())
            finally:
                F_.FOLD_BIAS_SUMS = True
        assert torch.equal(out[True][0], out[False][0]) and torch.equal(out[True][1], out[False][1])
        assert float((out[True][2] - out[False][2]).abs().max()) < 1e-5 * float(out[False][2].abs().max())


@pytest.mark.parametrize("store", ["f32", "bf16"])
def test_wgrad_degenerate_sizes(store):
    """No rows: both gradients are zero (written) or untouched (accumulated); no input columns: only the bias gradient exists."""
    dt = BF if store == "bf16" else torch.float32
    M, N = 64, 128
    dy, x = torch.empty(0, M, device=DEV, dtype=dt), torch.empty(0, N, device=DEV, dtype=dt)
    dW, db = torch.full((M, N), 3.0, device=DEV), torch.full((M,), 5.0, device=DEV)
    ops.wgrad(dy, x, dW, db, accum=True, db_accum=True)
    assert float(dW.min()) == 3.0 == float(dW.max()) and float(db.min()) == 5.0 == float(db.max())
    ops.wgrad(dy, x, dW, db)
    assert float(dW.abs().max()) == 0.0 and float(db.abs().max()) == 0.0
    dy = rnd(100, M, seed=1).to(dt)
    db = torch.full((M,), float("nan"), device=DEV)
    ops.wgrad(dy, torch.empty(100, 0, device=DEV, dtype=dt), torch.empty(M, 0, device=DEV), db)
    assert float((db.double() - dy.double().sum(0)).abs().max()) < 1e-4


def test_wgrad_rejects_mismatched_operands():
    dy, x = rnd(100, 64, seed=1), rnd(100, 32, seed=2)
    with pytest.raises(ops.SubgcError):
        ops.wgrad(dy, x.to(BF), torch.empty(64, 32, device=DEV), torch.empty(64, device=DEV))
    with pytest.raises(ops.SubgcError):
        ops.wgrad(dy, x[:50], torch.empty(64, 32, device=DEV), torch.empty(64, device=DEV))
    with pytest.raises(ops.SubgcError):
        ops.wgrad(dy, x, torch.empty(64, 32, device=DEV), torch.empty(32, device=DEV))


@pytest.mark.parametrize("K,M,N,splits", [(7000, 9488, 1000, 0), (7000, 4000, 1000, 3), (16640, 1024, 512, 8), (333, 200, 72, 0), (2176, 4000, 2000, 2),
                                            (12001, 1024, 1024, 4), (65, 264, 40, 0)])
def test_wgrad_eight_phase_form(K, M, N, splits):
    """The 256 x 256 x 64 eight-phase form of the bf16 weight gradient (csrc/gemm_bf16_p8.h), forced: column sums read from the landed K-major
    images of dY in the phases that read the same image, whole and split K (partial sums behind the planes), ragged rows."""
    pad = lambda n: n + 8 + (-n) % 8
    dy, x = rnd(K, pad(M), seed=K + M).to(BF)[:, :M], rnd(K, pad(N), seed=K + N + 1).to(BF)[:, :N]
    wantW, wantb = dy.double().t() @ x.double(), dy.double().sum(0)
    sw, sb = float(wantW.abs().max()), float(wantb.abs().max())
    with ops.gemm_tune(tile="p8", splits=splits):
        dW, db = torch.full((M, N), float("nan"), device=DEV), torch.full((M,), float("nan"), device=DEV)
        ops.wgrad(dy, x, dW, db)
        assert float((dW.double() - wantW).abs().max()) < 2e-5 * sw
        assert float((db.double() - wantb).abs().max()) < 2e-5 * max(sb, K ** 0.5)
        dW2, db2 = rnd(M, N, seed=3), rnd(M, seed=4)
        w0, b0 = dW2.clone(), db2.clone()
        ops.wgrad(dy, x, dW2, db2, accum=True, db_accum=True)
        assert float((dW2.double() - w0.double() - wantW).abs().max()) < 2e-5 * sw
        assert float((db2.double() - b0.double() - wantb).abs().max()) < 2e-5 * max(sb, K ** 0.5)
        again_W, again_b = torch.empty_like(dW), torch.empty_like(db)
        ops.wgrad(dy, x, again_W, again_b)
        assert torch.equal(again_W, dW) and torch.equal(again_b, db)         # fixed summation orders
        live = K // 3
        g = dy.clone(); g[live:] = float("nan")
        h = x.clone(); h[live:] = float("nan")
        m_dev = torch.tensor([live], device=DEV, dtype=torch.int32)
        ops.wgrad(g, h, dW, db, m_dev=m_dev)
        wW, wb = dy[:live].double().t() @ x[:live].double(), dy[:live].double().sum(0)
        assert float((dW.double() - wW).abs().max()) < 2e-5 * max(float(wW.abs().max()), 1.0)
        assert float((db.double() - wb).abs().max()) < 2e-5 * max(float(wb.abs().max()), K ** 0.5)
