"""BatchNorm fused into the GCN aggregation kernels (csrc/gcn_bn.hip; reference graph_conv_unit.py:28-36 with gcn_bn = 1,
graph_conv.py:26,33): one-pass statistics, normalise-on-load forward, BatchNorm backward in two launches -- against
nn.BatchNorm1d + the dense-incidence arithmetic of the reference on the same inputs, for fp32 and bf16-stored unit outputs."""
import numpy as np
import pytest
import torch

from subgc import functions as F_
from subgc import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _graph(B, N, K, seed):
    g = torch.Generator().manual_seed(seed)
    rel = torch.full((B, K, 2), N - 1, dtype=torch.long)
    rel[:, :K - 1] = torch.randint(0, N - 1, (B, K - 1, 2), generator=g)
    return rel


def _maps(rel, N):
    B, K, _ = rel.shape
    m = torch.zeros(B, N, K, 2, dtype=torch.float64)
    for b in range(B):
        for k in range(K):
            m[b, rel[b, k, 0], k, 0] = 1.0
            m[b, rel[b, k, 1], k, 1] = 1.0
    return m


def _collect(src_bn, A):
    """graph_conv_unit.py:34-36 on an already transformed (and normalised) source: relu((A @ src) / (rowsum + 1e-7))."""
    return torch.relu((A @ src_bn) / (A.sum(2, keepdim=True) + 1e-7))


@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("M,C", [(37, 64), (2405, 128), (16640, 1024)])
def test_bn_stats_and_backward_match_batchnorm1d(M, C, bf16):
    torch.manual_seed(M + C)
    x = (torch.randn(M, C) * torch.rand(1, C) * 3 + torch.randn(1, C) * 2).to(DEV)
    if bf16:
        x = x.to(torch.bfloat16)
    gamma, beta = (torch.rand(C, device=DEV) + 0.5), torch.randn(C, device=DEV)
    rm, rv = torch.randn(C, device=DEV) * 0.1, torch.rand(C, device=DEV) + 0.5
    ops.ensure_workspace(torch.device(DEV))
    bn = torch.nn.BatchNorm1d(C).to(DEV).double()
    with torch.no_grad():
        bn.weight.copy_(gamma); bn.bias.copy_(beta); bn.running_mean.copy_(rm); bn.running_var.copy_(rv)
    xd = x.double().requires_grad_(True)
    y = bn(xd)
    dy = torch.randn(M, C, device=DEV)
    y.backward(dy.double())
    rm2, rv2 = rm.clone(), rv.clone()
    aff, rstd = ops.bn_stats(x, gamma, beta, rm2, rv2, True)
    mean = x.double().mean(0)
    var = x.double().var(0, unbiased=False)
    np.testing.assert_allclose(aff[0].cpu().numpy(), mean.cpu().numpy(), atol=2e-6 * float(x.double().abs().max()), rtol=1e-5)
    np.testing.assert_allclose(rstd.cpu().numpy(), (1.0 / torch.sqrt(var + 1e-5)).cpu().numpy(), rtol=3e-5)
    np.testing.assert_allclose(aff[1].cpu().numpy(), (gamma.double() / torch.sqrt(var + 1e-5)).cpu().numpy(), rtol=3e-5)
    np.testing.assert_array_equal(aff[2].cpu().numpy(), beta.cpu().numpy())
    np.testing.assert_allclose(rm2.cpu().numpy(), bn.running_mean.cpu().numpy(), atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(rv2.cpu().numpy(), bn.running_var.cpu().numpy(), atol=1e-5, rtol=3e-5)
    dg, db = torch.full((C,), 7.0, device=DEV), torch.full((C,), -3.0, device=DEV)
    dx = ops.bn_bwd_fused(dy, x, gamma, aff, rstd, dg, db, True)
    assert dx.dtype == x.dtype
    sc = float(xd.grad.abs().max())
    np.testing.assert_allclose(dx.float().cpu().numpy(), xd.grad.cpu().numpy(), atol=(1e-2 if bf16 else 2e-5) * sc, rtol=(2e-2 if bf16 else 1e-4))
    np.testing.assert_allclose((dg - 7.0).cpu().numpy(), bn.weight.grad.cpu().numpy(), atol=3e-5 * float(bn.weight.grad.abs().max()) + 1e-4, rtol=1e-4)
    np.testing.assert_allclose((db + 3.0).cpu().numpy(), bn.bias.grad.cpu().numpy(), atol=3e-5 * float(bn.bias.grad.abs().max()) + 1e-4, rtol=1e-4)
    # eval mode: the triple comes from the running statistics
    aff_e, _ = ops.bn_stats(x, gamma, beta, rm, rv, False)
    np.testing.assert_allclose(aff_e[1].cpu().numpy(), (gamma / torch.sqrt(rv + 1e-5)).cpu().numpy(), rtol=1e-5)
    np.testing.assert_array_equal(aff_e[0].cpu().numpy(), rm.cpu().numpy())


@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("B,N,K,L", [(3, 37, 65, 64), (2, 101, 301, 128)])
def test_fused_bn_aggregation_fwd_bwd_matches_dense_reference(B, N, K, L, bf16):
    torch.manual_seed(B * N)
    rel = _graph(B, N, K, B)
    maps = _maps(rel, N).to(DEV)
    ops.ensure_workspace(torch.device(DEV))
    ptr, edges = ops.csr_build(rel.to(DEV), N)
    dt = torch.bfloat16 if bf16 else torch.float32
    tol = dict(atol=3e-2, rtol=3e-2) if bf16 else dict(atol=3e-5, rtol=2e-4)

    def params():
        return [(torch.rand(L, device=DEV) + 0.5).requires_grad_(True), torch.randn(L, device=DEV).requires_grad_(True)]

    def ref_bn(y, g, b):
        yd = y.double()
        mu, var = yd.mean((0, 1)), yd.var((0, 1), unbiased=False)
        return (yd - mu) / torch.sqrt(var + 1e-5) * g.double() + b.double()

    # ---- nodes <- relations
    y0 = (torch.randn(B, K, L, device=DEV) * 2 + 0.3).to(dt).requires_grad_(True)
    y1 = (torch.randn(B, K, L, device=DEV) - 0.2).to(dt).requires_grad_(True)
    skip = torch.randn(B, N, L, device=DEV, requires_grad=True)
    (g0, b0), (g1, b1) = params(), params()
    stats = tuple(torch.zeros(L, device=DEV) if i % 2 == 0 else torch.ones(L, device=DEV) for i in range(4))
    out, out16 = F_.GcnNodesBnFn.apply(y0, y1, skip, rel.to(DEV), ptr, edges, N, g0, b0, g1, b1, stats, True, True)
    want = (_collect(ref_bn(y0, g0, b0), maps[..., 0]) + _collect(ref_bn(y1, g1, b1), maps[..., 1])) / 2 + skip.double()
    np.testing.assert_allclose(out.detach().cpu().numpy(), want.detach().cpu().numpy(), **tol)
    np.testing.assert_allclose(out16.float().cpu().numpy(), out.detach().cpu().numpy(), atol=1e-2 * float(out.abs().max()), rtol=1e-2)
    w = torch.randn_like(out)
    (out * w).sum().backward()
    got = [t.grad.clone() for t in (y0, y1, skip, g0, b0, g1, b1)]
    for t in (y0, y1, skip, g0, b0, g1, b1):
        t.grad = None
    (want * w.double()).sum().backward()
    for a, t, name in zip(got, (y0, y1, skip, g0, b0, g1, b1), "y0 y1 skip g0 b0 g1 b1".split()):
        assert a.dtype == t.dtype
        sc = float(t.grad.abs().max()) + 1e-12
        np.testing.assert_allclose(a.float().cpu().numpy(), t.grad.float().cpu().numpy(), atol=tol["atol"] * sc, rtol=tol["rtol"] * 5, err_msg=name)
        t.grad = None
    assert float(stats[0].abs().max()) > 0                                 # running statistics were updated

    # ---- relations <- nodes
    y2 = (torch.randn(B, N, L, device=DEV) + 0.1).to(dt).requires_grad_(True)
    y3 = (torch.randn(B, N, L, device=DEV) * 0.5).to(dt).requires_grad_(True)
    skp = torch.randn(B, K, L, device=DEV, requires_grad=True)
    (g2, b2), (g3, b3) = params(), params()
    stats = tuple(torch.zeros(L, device=DEV) if i % 2 == 0 else torch.ones(L, device=DEV) for i in range(4))
    outp = F_.GcnEdgesBnFn.apply(y2, y3, skp, rel.to(DEV), ptr, edges, K, g2, b2, g3, b3, stats, True, False)
    wantp = (_collect(ref_bn(y2, g2, b2), maps[..., 0].transpose(1, 2)) + _collect(ref_bn(y3, g3, b3), maps[..., 1].transpose(1, 2))) / 2 + skp.double()
    np.testing.assert_allclose(outp.detach().cpu().numpy(), wantp.detach().cpu().numpy(), **tol)
    w = torch.randn_like(outp)
    (outp * w).sum().backward()
    got = [t.grad.clone() for t in (y2, y3, skp, g2, b2, g3, b3)]
    for t in (y2, y3, skp, g2, b2, g3, b3):
        t.grad = None
    (wantp * w.double()).sum().backward()
    for a, t, name in zip(got, (y2, y3, skp, g2, b2, g3, b3), "y2 y3 skip g2 b2 g3 b3".split()):
        sc = float(t.grad.abs().max()) + 1e-12
        np.testing.assert_allclose(a.float().cpu().numpy(), t.grad.float().cpu().numpy(), atol=tol["atol"] * sc, rtol=tol["rtol"] * 5, err_msg=name)


@pytest.mark.parametrize("B,N,K,L", [(3, 37, 65, 64), (2, 101, 301, 128)])
def test_bf16_stored_unit_outputs_aggregate_like_their_fp32_values(B, N, K, L):
    """compute_dtype = bf16 without BatchNorm (Sub-GC presets, Flickr shape): GcnNodesB16Fn / GcnEdgesB16Fn read the bf16 unit outputs
    directly and write bf16 gradients -- same results as GcnNodesFn / GcnEdgesFn on the fp32 values of the same bf16 numbers, gradients
    equal after rounding to bf16, the bf16 copy of the result equal to the rounded result."""
    torch.manual_seed(B * N + 1)
    rel = _graph(B, N, K, B).to(DEV)
    ops.ensure_workspace(torch.device(DEV))
    ptr, edges = ops.csr_build(rel, N)
    BFT = torch.bfloat16

    def pair(shape):
        a16 = (torch.randn(*shape, device=DEV) * 2).to(BFT)
        return a16.clone().requires_grad_(True), a16.float().requires_grad_(True)

    (y0, y0f), (y1, y1f) = pair((B, K, L)), pair((B, K, L))
    skip = torch.randn(B, N, L, device=DEV, requires_grad=True)
    skipf = skip.detach().clone().requires_grad_(True)
    out, out16 = F_.GcnNodesB16Fn.apply(y0, y1, skip, rel, ptr, edges, N, True)
    ref = F_.GcnNodesFn.apply(y0f, y1f, skipf, rel, ptr, edges, N)
    assert torch.equal(out, ref) and torch.equal(out16, ref.to(BFT))
    w = torch.randn_like(out)
    (out * w).sum().backward()
    (ref * w).sum().backward()
    assert y0.grad.dtype == BFT and torch.equal(y0.grad, y0f.grad.to(BFT)) and torch.equal(y1.grad, y1f.grad.to(BFT)) and torch.equal(skip.grad, skipf.grad)

    (y2, y2f), (y3, y3f) = pair((B, N, L)), pair((B, N, L))
    skp = torch.randn(B, K, L, device=DEV, requires_grad=True)
    skpf = skp.detach().clone().requires_grad_(True)
    outp = F_.GcnEdgesB16Fn.apply(y2, y3, skp, rel, ptr, edges, K, False)
    refp = F_.GcnEdgesFn.apply(y2f, y3f, skpf, rel, ptr, edges, K)
    assert torch.equal(outp, refp)
    w = torch.randn_like(outp)
    (outp * w).sum().backward()
    (refp * w).sum().backward()
    assert torch.equal(y2.grad, y2f.grad.to(BFT)) and torch.equal(y3.grad, y3f.grad.to(BFT)) and torch.equal(skp.grad, skpf.grad)


@pytest.mark.parametrize("M,C,bf16", [(16640, 1024, True), (9472, 1024, True), (4736, 1024, False), (37, 8, False)])
def test_bn_pair_launches_equal_the_single_calls(M, C, bf16):
    """subgc_bn_stats_pair / subgc_bn_bwd_fused_pair: the two units a fused aggregation consumes, two launches instead of four each -- the
    slab plans and summation orders of the single calls, so their results bit for bit (statistics, running statistics, d(x), affine
    gradients written or accumulated)."""
    g = torch.Generator().manual_seed(M + C)
    mk = lambda *s: torch.randn(*s, generator=g).to(DEV)
    xs = [(mk(M, C) * 2 + 1), (mk(M, C) * 0.5 - 2)]
    if bf16:
        xs = [x.to(torch.bfloat16) for x in xs]
    gam, bet = [mk(C), mk(C)], [mk(C), mk(C)]
    run = lambda: ([torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)], [torch.ones(C, device=DEV), torch.ones(C, device=DEV)])
    rm_s, rv_s = run()
    single = [ops.bn_stats(xs[i], gam[i], bet[i], rm_s[i], rv_s[i], True) for i in range(2)]
    rm_p, rv_p = run()
    pair = ops.bn_stats_pair(xs[0], xs[1], gam[0], bet[0], rm_p[0], rv_p[0], gam[1], bet[1], rm_p[1], rv_p[1], True)
    for i in range(2):
        assert torch.equal(single[i][0], pair[i][0]) and torch.equal(single[i][1], pair[i][1])
        assert torch.equal(rm_s[i], rm_p[i]) and torch.equal(rv_s[i], rv_p[i])
    dys = [mk(M, C), mk(M, C)]
    for acc in (False, True):
        base = [mk(C), mk(C), mk(C), mk(C)]
        dg_s, db_s = [base[0].clone(), base[1].clone()], [base[2].clone(), base[3].clone()]
        dx_s = [ops.bn_bwd_fused(dys[i], xs[i], gam[i], single[i][0], single[i][1], dg_s[i], db_s[i], acc) for i in range(2)]
        dg_p, db_p = [base[0].clone(), base[1].clone()], [base[2].clone(), base[3].clone()]
        dx_p = ops.bn_bwd_fused_pair(dys[0], dys[1], xs[0], xs[1], gam[0], gam[1], pair[0][0], pair[1][0], pair[0][1], pair[1][1], dg_p[0], dg_p[1],
                                     db_p[0], db_p[1], acc)
        for i in range(2):
            assert torch.equal(dx_s[i], dx_p[i]) and torch.equal(dg_s[i], dg_p[i]) and torch.equal(db_s[i], db_p[i])
