"""Per-kernel numerics of libsubgc_hip.so on the MI355X, each against a plain PyTorch fp32/fp64
reference of the same op (integer outputs bit-exact, floats atol/rtol stated per test)."""
import numpy as np
import pytest
import torch

from oracle import subgc_oracle as O
from subgc import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rnd(*s, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*s, generator=g) * scale).to(DEV)


def close(a, b, atol=1e-4, rtol=1e-4, msg=""):
    torch.testing.assert_close(a.float().cpu(), b.float().cpu(), atol=atol, rtol=rtol, msg=lambda m: f"{msg}: {m}")


# ----------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(640, 4000, 1000), (130, 96, 300), (64, 64, 32), (37, 51, 48), (1, 9488, 1000), (257, 129, 65)])
@pytest.mark.parametrize("mode", ["nt", "nn", "tn"])
def test_gemm_modes(M, N, K, mode):
    a = rnd(M, K, seed=1); b = rnd(K, N, seed=2)
    ref = (a.double() @ b.double()).float()
    out = torch.empty(M, N, device=DEV)
    if mode == "nt":
        ops.gemm(a, b.t().contiguous(), out, tb=True)
    elif mode == "nn":
        ops.gemm(a, b, out)
    else:
        ops.gemm(a.t().contiguous(), b, out, ta=True)
    close(out, ref, atol=2e-4 * K ** 0.5, rtol=1e-4, msg=f"{mode} {M}x{N}x{K}")


def test_gemm_epilogue_and_views():
    M, N, K = 200, 136, 72
    big_a = rnd(M, K + 24, seed=3); a = big_a[:, 8:8 + K]                # strided A view (ld != K)
    w = rnd(N, K, seed=4); bias = rnd(N, seed=5); add = rnd(M, N, seed=6)
    keep = (torch.rand(M, N, generator=torch.Generator().manual_seed(7)) > 0.5).to(torch.uint8).to(DEV)
    big_c = torch.zeros(M, N + 40, device=DEV); c = big_c[:, 16:16 + N]
    prev = rnd(M, N, seed=8); c.copy_(prev)
    ops.gemm(a, w, c, tb=True, bias=bias, add=add, keep=None, relu=True, accum=True)
    ref = torch.relu(a.double() @ w.double().t() + bias.double() + add.double()).float() + prev
    close(c, ref, atol=2e-3)
    assert float(big_c[:, :16].abs().max()) == 0 and float(big_c[:, 16 + N:].abs().max()) == 0
    c2 = torch.empty(M, N, device=DEV)
    ops.gemm(a, w, c2, tb=True, bias=bias, relu=True, keep=keep, keep_scale=2.0)
    close(c2, torch.relu(a.double() @ w.double().t() + bias.double()).float() * keep * 2.0, atol=2e-3)


def test_gemm_ragged_rows_gather_scatter():
    M, N, K = 300, 64, 48
    src = rnd(500, K, seed=1); w = rnd(N, K, seed=2)
    rows = torch.randint(0, 500, (M,), generator=torch.Generator().manual_seed(3)).int().to(DEV)
    rows[5] = -1
    m_dev = torch.tensor([211], dtype=torch.int32, device=DEV)
    out = torch.full((M, N), 7.0, device=DEV)
    ops.gemm(src, w, out, tb=True, a_rows=rows, m_dev=m_dev)
    g = src[rows.clamp(min=0).long()] * (rows >= 0).float().unsqueeze(1)
    ref = (g.double() @ w.double().t()).float()
    close(out[:211], ref[:211], atol=1e-3)
    assert float((out[211:] - 7.0).abs().max()) == 0                     # rows beyond *m_dev untouched
    # scatter rows of C
    perm = torch.randperm(M, generator=torch.Generator().manual_seed(4)).int().to(DEV)
    out2 = torch.zeros(M, N, device=DEV)
    a = rnd(M, K, seed=5)
    ops.gemm(a, w, out2, tb=True, c_rows=perm)
    close(out2[perm.long()], (a.double() @ w.double().t()).float(), atol=1e-3)
    # weight-gradient form over a ragged row set: K bounded on the device
    dy = rnd(M, N, seed=6); x = rnd(M, K, seed=7)
    dw = torch.empty(N, K, device=DEV)
    ops.gemm(dy, x, dw, ta=True, m_dev=m_dev)
    close(dw, (dy[:211].double().t() @ x[:211].double()).float(), atol=2e-3)
    close(ops.colsum(dy, m_dev=m_dev), dy[:211].sum(0), atol=1e-3)


# ----------------------------------------------------------------------------- index kernels
def test_row_argmax_first_max_and_skip():
    x = rnd(300, 1599, seed=1)
    x[3, 10] = x[3, 700] = 50.0          # tie -> first
    x[4, 0] = 99.0                       # skipped column
    x[5] = 0.0; x[5, 0] = 1.0            # one-hot(0) dummy row -> class 1 when skipping column 0
    idx, val = ops.row_argmax(x, skip=1, want_val=True)
    ref_v, ref_i = x[:, 1:].cpu().max(1)
    assert torch.equal(idx.cpu(), ref_i + 1)
    assert torch.equal(val.cpu(), ref_v)
    assert int(idx[3]) == 10 and int(idx[5]) == 1
    assert torch.equal(ops.row_argmax(x).cpu(), x.cpu().max(1)[1])


def test_csr_build():
    B, K, N = 5, 65, 37
    g = torch.Generator().manual_seed(0)
    rel = torch.randint(0, N, (B, K, 2), generator=g)
    rel[:, 50:] = N - 1
    ptr, edges = ops.csr_build(rel.to(DEV), N)
    ptr, edges = ptr.cpu().numpy(), edges.cpu().numpy()
    for role in range(2):
        for b in range(B):
            for n in range(N):
                want = np.nonzero(rel[b, :, role].numpy() == n)[0]
                got = edges[role, b, ptr[role, b, n]:ptr[role, b, n + 1]]
                np.testing.assert_array_equal(got, want)
            assert ptr[role, b, N] == K


# ----------------------------------------------------------------------------- GCN aggregation
def _dense_maps(rel, N):
    B, K, _ = rel.shape
    return O.make_map(B, N, K, rel, torch.zeros(1))


@pytest.mark.parametrize("L", [32, 1024])
def test_gcn_nodes_and_edges_fwd_bwd(L):
    B, K, N = 4, 65, 37
    g = torch.Generator().manual_seed(1)
    rel = torch.randint(0, N - 1, (B, K, 2), generator=g); rel[:, 60:] = N - 1
    F0, F1 = (torch.randn(B, K, L, generator=g, dtype=torch.float64).requires_grad_() for _ in range(2))
    F2, F3 = (torch.randn(B, N, L, generator=g, dtype=torch.float64).requires_grad_() for _ in range(2))
    skx = torch.randn(B, N, L, generator=g, dtype=torch.float64); skp = torch.randn(B, K, L, generator=g, dtype=torch.float64)
    ms, mo = (m.double() for m in _dense_maps(rel, N))
    unit = lambda adj, f: torch.relu(torch.bmm(adj, f) / (adj.sum(2, keepdim=True) + 1e-7))
    X = (unit(ms, F0) + unit(mo, F1)) / 2 + skx
    Pq = (unit(ms.transpose(1, 2), F2) + unit(mo.transpose(1, 2), F3)) / 2 + skp
    gx = torch.randn(B, N, L, generator=g, dtype=torch.float64); gp = torch.randn(B, K, L, generator=g, dtype=torch.float64)
    (X * gx).sum().backward(); (Pq * gp).sum().backward()

    d = lambda t: t.detach().float().to(DEV).contiguous()
    reld = rel.to(DEV)
    ptr, edges = ops.csr_build(reld, N)
    Xh, act = ops.gcn_nodes_fwd(d(F0), d(F1), ptr, edges, d(skx), B, N, K, L)
    close(Xh, X, atol=1e-5)
    dF0, dF1 = ops.gcn_nodes_bwd(d(gx), act, reld, ptr, B, N, K, L)
    close(dF0, F0.grad, atol=1e-5); close(dF1, F1.grad, atol=1e-5)
    Ph = ops.gcn_edges_fwd(d(F2), d(F3), reld, d(skp), B, N, K, L)
    close(Ph, Pq, atol=1e-5)
    dF2, dF3 = ops.gcn_edges_bwd(d(gp), d(F2), d(F3), ptr, edges, B, N, K, L)
    close(dF2, F2.grad, atol=1e-5); close(dF3, F3.grad, atol=1e-5)


def test_gcn_div_by_one_plus_eps_is_not_identity():
    """fp32(1 + 1e-7) = 1.00000012: x / c != x -- compare in fp32 bit-exactly."""
    B, K, N, L = 1, 3, 4, 4
    rel = torch.tensor([[[0, 1], [1, 2], [2, 0]]])
    F2 = torch.full((B, N, L), 3.0); F3 = torch.full((B, N, L), 5.0)
    out = ops.gcn_edges_fwd(F2.to(DEV), F3.to(DEV), rel.to(DEV), None, B, N, K, L).cpu()
    c = torch.tensor(1.0) + torch.tensor(1e-7)
    want = (torch.relu(torch.tensor(3.0) / c) + torch.relu(torch.tensor(5.0) / c)) / 2
    assert torch.equal(out, torch.full((B, K, L), float(want)))
    assert float(want) != 4.0


def test_batchnorm_fwd_bwd():
    M, C = 333, 96
    x = rnd(M, C, seed=1, scale=3.0) + 1.5
    bn = torch.nn.BatchNorm1d(C).double()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C) + 0.5); bn.bias.copy_(torch.randn(C))
    xr = x.double().cpu().requires_grad_()
    y = bn(xr)
    gy = torch.randn(M, C, dtype=torch.float64, generator=torch.Generator().manual_seed(2))
    (y * gy).sum().backward()
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    gam, bet = bn.weight.detach().float().to(DEV), bn.bias.detach().float().to(DEV)
    yh, sm, sr = ops.bn_fwd(x, gam, bet, rm, rv, True)
    close(yh, y, atol=2e-5); close(rm, bn.running_mean, atol=1e-5); close(rv, bn.running_var, atol=1e-5)
    dx, dg, db = ops.bn_bwd(gy.float().to(DEV), x, gam, sm, sr)
    close(dx, xr.grad, atol=2e-5); close(dg, bn.weight.grad, atol=2e-4); close(db, bn.bias.grad, atol=2e-4)
    bn.eval()
    ye, _, _ = ops.bn_fwd(x, gam, bet, rm, rv, False)
    close(ye, bn(x.double().cpu()), atol=2e-5)


# ----------------------------------------------------------------------------- sGPN
def test_subgraph_pool_fwd_bwd_matches_gather_bmm():
    Bimg, N, L, G = 3, 37, 64, 24
    g = torch.Generator().manual_seed(3)
    X = torch.rand(Bimg, N, L, generator=g, dtype=torch.float64).requires_grad_()
    idx = torch.full((G, N), N - 1, dtype=torch.long); w = torch.zeros(G, N, dtype=torch.float64)
    img = torch.randint(0, Bimg, (G,), generator=g)
    for q in range(G):
        n = int(torch.randint(2, 12, (1,), generator=g))
        idx[q, :n] = torch.sort(torch.randperm(N - 1, generator=g)[:n])[0]; w[q, :n] = 1
    denom = w.sum(1)
    gathered = X[img.unsqueeze(1), idx]                                        # [G,N,L]
    clean = gathered * w.unsqueeze(-1)
    ref = torch.cat((clean.max(1)[0], clean.sum(1) / denom.unsqueeze(1)), -1)
    gout = torch.randn(G, 2 * L, generator=g, dtype=torch.float64)
    (ref * gout).sum().backward()
    X2 = X.detach().float().reshape(Bimg * N, L).to(DEV)
    args = (idx.to(DEV), N, w.float().to(DEV), N, 1, denom.float().to(DEV), img.int().to(DEV))
    out, am = ops.pool_fwd(X2, *args, G, N, L)
    close(out, ref, atol=1e-6)
    dX = torch.zeros(Bimg * N, L, device=DEV)
    ops.pool_bwd(gout.float().to(DEV), *args, am, dX, G, N, L)
    close(dX.view(Bimg, N, L), X.grad, atol=1e-5)


def test_gpn_score_bce_fwd_bwd():
    G, H = 40, 24
    hid = torch.relu(rnd(G, H, seed=1)).double().cpu().requires_grad_()
    w2 = rnd(1, H, seed=2).double().cpu().requires_grad_(); b2 = torch.tensor([0.3], dtype=torch.float64, requires_grad=True)
    keep = (torch.rand(G, H, generator=torch.Generator().manual_seed(3)) > 0.5)
    s = torch.sigmoid((hid * keep * 2.0) @ w2.t() + b2)
    tgt = torch.cat((torch.ones(G // 2, 1), torch.zeros(G // 2, 1))).double()
    loss = torch.nn.functional.binary_cross_entropy(s, tgt)
    (loss * 1.7).backward()
    f = lambda t: t.detach().float().to(DEV)
    sc, ls = ops.gpn_score_fwd(f(hid), keep.to(torch.uint8).to(DEV), 2.0, f(w2), f(b2))
    close(sc, s, atol=1e-6); close(ls, loss, atol=1e-6)
    dh, dw, db = ops.gpn_score_bwd(f(hid), keep.to(torch.uint8).to(DEV), 2.0, f(w2), sc, torch.tensor(1.7, device=DEV))
    close(dh, hid.grad, atol=1e-6); close(dw, w2.grad, atol=1e-5); close(db, b2.grad, atol=1e-6)


@pytest.mark.parametrize("M,pool,thres,maxk", [(80, 10, 0.55, 1000), (48, 14, 0.75, 10), (600, 12, 0.55, 1000), (30, 8, 0.3, 1)])
def test_subgraph_nms_matches_python_sets(M, pool, thres, maxk):
    N = 37
    rng = np.random.default_rng(M)
    idx = np.full((M, N), N - 1, np.int64); mask = np.zeros((M, N), np.float32)
    for m in range(M):
        n = rng.integers(2, 9)
        idx[m, :n] = np.sort(rng.choice(pool, n, replace=False)); mask[m, :n] = 1
    score = rng.random(M).astype(np.float32)
    score[3] = score[11]                                            # a tie: larger index first
    idx[7] = idx[2]; mask[7] = mask[2]                              # duplicate node set
    want = O.subgraph_nms(score, idx, mask, thres, maxk, sort_kind="stable")
    keep, n = ops.subgraph_nms(torch.from_numpy(score).to(DEV), torch.from_numpy(idx).to(DEV),
                               torch.from_numpy(mask.sum(1)).int().to(DEV), thres, maxk)
    got = keep[: int(n.item())].cpu().numpy()
    np.testing.assert_array_equal(got, want)


# ----------------------------------------------------------------------------- decoder kernels
def test_pack_rows_and_gather_scatter():
    S, N = 1500, 37
    g = torch.Generator().manual_seed(0)
    lens = torch.randint(0, 12, (S,), generator=g).int()
    idx = torch.randint(0, N, (S, N), generator=g)
    img = torch.randint(0, 9, (S,), generator=g).int()
    off, total, src, sent = (t.cpu() for t in ops.pack_rows(lens.to(DEV), idx.to(DEV), img.to(DEV), S, N))
    ref_off = torch.cumsum(lens, 0) - lens
    assert torch.equal(off, ref_off.int()) and int(total) == int(lens.sum())
    for s in (0, 1, 77, S - 1):
        for i in range(int(lens[s])):
            m = int(ref_off[s]) + i
            assert int(src[m]) == int(img[s]) * N + int(idx[s, i]) and int(sent[m]) == s
    assert bool((src[int(total):] == -1).all())
    table = rnd(9 * N, 16, seed=1)
    dst = torch.zeros(S * N, 16, device=DEV)
    ops.gather_rows(table, src.to(DEV), dst, m_dev=total.to(DEV))
    close(dst[: int(total)], table[src[: int(total)].long().to(DEV)], atol=0)
    acc = torch.zeros(9 * N, 16, device=DEV)
    ops.scatter_add_rows(dst, src.to(DEV), acc, m_dev=total.to(DEV))
    ref = torch.zeros(9 * N, 16, device=DEV).index_add_(0, src[: int(total)].long().to(DEV), dst[: int(total)])
    close(acc, ref, atol=1e-4)


def test_embed_fwd_bwd():
    V, E, n = 51, 48, 200
    table = rnd(V, E, seed=1)
    tok = torch.randint(0, V, (n, 3), generator=torch.Generator().manual_seed(2)).to(DEV)
    keep = (torch.rand(n, E, generator=torch.Generator().manual_seed(3)) > 0.3).to(torch.uint8).to(DEV)
    out = torch.empty(n, E, device=DEV)
    ops.embed_fwd(table, tok[:, 1], 3, keep, 1.5, out)
    ref = torch.relu(table[tok[:, 1]]) * keep * 1.5
    close(out, ref, atol=0)
    gout = rnd(n, E, seed=4)
    dt = torch.zeros(V, E, device=DEV)
    ops.embed_bwd(table, tok[:, 1], 3, keep, 1.5, gout, dt)
    ref_dt = torch.zeros(V, E, device=DEV).index_add_(0, tok[:, 1], gout * keep * 1.5 * (table[tok[:, 1]] > 0))
    close(dt, ref_dt, atol=1e-4)


def test_lstm_gates_fwd_bwd():
    S, R = 70, 48
    cell = torch.nn.LSTMCell(R, R).double()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(S, R, generator=g, dtype=torch.float64); h0 = torch.randn(S, R, generator=g, dtype=torch.float64)
    c0 = torch.randn(S, R, generator=g, dtype=torch.float64, requires_grad=True)
    ga = (x @ cell.weight_ih.t()).detach().requires_grad_(); gb = (h0 @ cell.weight_hh.t()).detach().requires_grad_()
    pre = ga + gb + cell.bias_ih + cell.bias_hh
    i, f, gg, o = pre.chunk(4, 1)
    c1 = torch.sigmoid(f) * c0 + torch.sigmoid(i) * torch.tanh(gg); h1 = torch.sigmoid(o) * torch.tanh(c1)
    keep = (torch.rand(S, R, generator=g) > 0.5)
    dh, dhd, dc = (torch.randn(S, R, generator=g, dtype=torch.float64) for _ in range(3))
    ((h1 * dh).sum() + (h1 * keep * 2.0 * dhd).sum() + (c1 * dc).sum()).backward()
    f32 = lambda t: t.detach().float().to(DEV).contiguous()
    c = torch.empty(S, R, device=DEV); hbuf = torch.zeros(S, 3 * R, device=DEV); h2 = torch.empty(S, R, device=DEV)
    hd = torch.empty(S, R, device=DEV); gates = torch.empty(S, 4 * R, device=DEV)
    k8 = keep.to(torch.uint8).to(DEV)
    ops.lstm_fwd(f32(ga), f32(gb), None, f32(cell.bias_ih), f32(cell.bias_hh), f32(c0), c, hbuf[:, R:2 * R], h2, k8, 2.0, hd, gates, S, R)
    close(c, c1, atol=1e-6); close(hbuf[:, R:2 * R], h1, atol=1e-6); close(h2, h1, atol=1e-6); close(hd, h1 * keep * 2.0, atol=1e-6)
    assert float(hbuf[:, :R].abs().max()) == 0 and float(hbuf[:, 2 * R:].abs().max()) == 0
    dpre = torch.empty(S, 4 * R, device=DEV); dcp = torch.empty(S, R, device=DEV)
    ops.lstm_bwd(gates, f32(c0), c, f32(dh), None, f32(dhd), k8, 2.0, f32(dc), dpre, dcp, S, R)
    close(dpre, ga.grad, atol=1e-5); close(dcp, c0.grad, atol=1e-5)


def test_attention_step_fwd_bwd_equals_softmax_mask_renorm():
    S, N, A, R = 9, 37, 24, 48
    g = torch.Generator().manual_seed(7)
    lens = torch.tensor([1, 2, 5, 11, 7, 3, 36, 4, 9], dtype=torch.int32)
    off = (torch.cumsum(lens, 0) - lens).int(); rows = int(lens.sum())
    u = torch.randn(rows, A, generator=g, dtype=torch.float64, requires_grad=True)
    v = torch.randn(rows, R, generator=g, dtype=torch.float64, requires_grad=True)
    ah = torch.randn(S, A, generator=g, dtype=torch.float64, requires_grad=True)
    wa = torch.randn(1, A, generator=g, dtype=torch.float64, requires_grad=True); ba = torch.tensor([0.2], dtype=torch.float64, requires_grad=True)
    n_max = int(lens.max())
    # the reference formulation on the padded layout: softmax over n_max, mask, renormalise (AttModel.py:461-466)
    up = torch.zeros(S, n_max, A, dtype=torch.float64); vp = torch.zeros(S, n_max, R, dtype=torch.float64); mk = torch.zeros(S, n_max, dtype=torch.float64)
    for s in range(S):
        l, o = int(lens[s]), int(off[s])
        up[s, :l] = u[o:o + l]; vp[s, :l] = v[o:o + l]; mk[s, :l] = 1
    e = (torch.tanh(up + ah.unsqueeze(1)) @ wa.t()).squeeze(-1) + ba
    wgt = torch.softmax(e, 1) * mk; wgt = wgt / wgt.sum(1, keepdim=True)
    ctx = torch.bmm(wgt.unsqueeze(1), vp).squeeze(1)
    gctx = torch.randn(S, R, generator=g, dtype=torch.float64)
    (ctx * gctx).sum().backward()
    f = lambda t: t.detach().float().to(DEV).contiguous()
    ctx_h = torch.zeros(S, 3 * R, device=DEV); al = torch.empty(S, N, device=DEV)
    ops.attn_fwd(f(u), f(v), f(ah), f(wa), f(ba), off.to(DEV), lens.to(DEV), ctx_h[:, :R], al, S, A, R)
    close(ctx_h[:, :R], ctx, atol=1e-5); close(al[:, :n_max], wgt, atol=1e-6)
    assert float(al[:, n_max:].abs().max()) == 0
    dah = torch.empty(S, A, device=DEV); du = torch.ones(rows, A, device=DEV); dv = torch.ones(rows, R, device=DEV)
    dwa = torch.zeros(S, A, device=DEV); dba = torch.zeros(S, device=DEV)
    gbuf = torch.zeros(S, 3 * R, device=DEV); gbuf[:, :R] = f(gctx)
    ops.attn_bwd(f(u), f(v), f(ah), f(wa), off.to(DEV), lens.to(DEV), al, gbuf[:, :R], dah, du, dv, dwa, dba, S, A, R)
    close(dah, ah.grad, atol=1e-5); close(du - 1, u.grad, atol=1e-5); close(dv - 1, v.grad, atol=1e-5)
    close(dwa.sum(0, keepdim=True), wa.grad, atol=1e-4); close(dba.sum().view(1), ba.grad, atol=1e-5)


def test_log_softmax_nll_step_active():
    S, T, V = 12, 17, 9488
    x = rnd(S * T, V, seed=1, scale=3.0)
    labels = torch.zeros(S, T + 1, dtype=torch.long)
    g = torch.Generator().manual_seed(2)
    for s in range(S):
        n = int(torch.randint(3, 12, (1,), generator=g)); labels[s, 1:n + 1] = torch.randint(1, V, (n,), generator=g)
    mask = (torch.arange(T + 1).view(1, -1) < (labels > 0).sum(1, keepdim=True) + 2).float()
    active = ops.step_active(labels.to(DEV), T)
    ref_active = torch.ones(T, dtype=torch.int32)
    for t in range(1, T):
        if labels[:, t].sum() == 0:
            ref_active[t:] = 0
            break
    assert torch.equal(active.view(S, T).cpu(), ref_active.view(1, T).expand(S, T))
    lp = x.clone()
    ops.log_softmax_rows_(lp, active)
    ref = torch.log_softmax(x.double(), 1).float() * active.view(-1, 1)
    close(lp, ref, atol=2e-5)
    lab, msk = labels.to(DEV), mask.to(DEV)
    loss, scratch = ops.masked_nll_fwd(lp.view(S, T, V), lab[:, 1:], msk[:, 1:])
    ref_loss = O.lm_criterion(ref.view(S, T, V).cpu().double(), labels[:, 1:], mask[:, 1:].double())
    close(loss, ref_loss, atol=1e-5)
    dlp = ops.masked_nll_bwd(lab[:, 1:], msk[:, 1:], scratch, torch.tensor(1.0, device=DEV), S, T, V)
    lpr = ref.view(S, T, V).cpu().double().requires_grad_()
    O.lm_criterion(lpr, labels[:, 1:], mask[:, 1:].double()).backward()
    close(dlp, lpr.grad, atol=1e-7)
    xr = x.double().cpu().requires_grad_()
    (torch.log_softmax(xr, 1) * active.view(-1, 1).cpu() * lpr.grad.view(S * T, V)).sum().backward()
    dl = torch.empty_like(x)
    ops.log_softmax_rows_bwd(lp, dlp.view(S * T, V), dl, active)
    close(dl, xr.grad, atol=1e-6)


@pytest.mark.parametrize("V", [9488, 7001, 52])                       # four-columns-per-lane form, scalar form (V % 4 != 0), one short row
@pytest.mark.parametrize("b16", [False, True])
def test_criterion_on_raw_logits_with_row_lse_equals_the_log_softmax_path(V, b16):
    """Loss-only path (functions_packed): logits stay raw, subgc_row_lse_f32 gives the row log-sum-exp, the criterion and its fused
    backward subtract it on the fly -- same loss and d(logits) as log_softmax_rows_ + the plain calls, and as torch in fp64."""
    rows = 37
    x = rnd(rows, V, seed=V, scale=3.0)
    g = torch.Generator().manual_seed(V)
    tgt = torch.randint(0, V, (rows, 1), generator=g).to(DEV)
    msk = (torch.rand(rows, 1, generator=g) > 0.25).float().to(DEV)
    den = msk.sum().view(1) + 3.0                                     # the packed decoder's denominator covers rows this call does not see
    lse = ops.row_lse(x)
    close(lse, torch.logsumexp(x.double(), 1), atol=2e-5)
    loss_a, sc_a = ops.masked_nll_fwd(x.view(rows, 1, V), tgt, msk, den=den, lse=lse)
    lp = ops.log_softmax_rows_(x.clone())
    loss_b, sc_b = ops.masked_nll_fwd(lp.view(rows, 1, V), tgt, msk, den=den)
    close(loss_a, loss_b, atol=1e-6)
    ref = -(torch.log_softmax(x.double(), 1).gather(1, tgt) * msk.double()).sum() / den.double()
    close(loss_a, ref.view(()), atol=1e-5)
    dt = torch.bfloat16 if b16 else torch.float32
    ld = -(-V // 8) * 8
    one = torch.tensor(0.7, device=DEV)
    da = torch.zeros(rows, ld, device=DEV, dtype=dt)[:, :V]
    db = torch.zeros(rows, ld, device=DEV, dtype=dt)[:, :V]
    ops.nll_logsoftmax_bwd(x, tgt, msk, sc_a, one, da, None, rows, 1, V, lse=lse)
    ops.nll_logsoftmax_bwd(lp, tgt, msk, sc_b, one, db, None, rows, 1, V)
    assert torch.equal(da, db)                                        # exp(x - lse) IS exp(logp): bit-identical
    want = 0.7 * msk.double() / den.double() * (torch.softmax(x.double(), 1) - torch.zeros(rows, V, device=DEV, dtype=torch.double).scatter_(1, tgt, 1.0))
    close(da.float(), want, atol=2e-3 * float(want.abs().max()) if b16 else 1e-7, rtol=1e-2 if b16 else 1e-4)


def test_decode_pick_greedy_topk_and_finished_masking():
    n, V, T = 6, 9488, 20
    logp = torch.log_softmax(rnd(n, V, seed=3, scale=2.0), 1)
    logp[2, 0] = 5.0                                                  # row 2 emits EOS at t = 0
    seq = torch.zeros(n, T, dtype=torch.long, device=DEV); slp = torch.zeros(n, T, device=DEV)
    it = torch.zeros(n, dtype=torch.long, device=DEV); unf = torch.zeros(n, dtype=torch.int32, device=DEV)
    cnt = torch.zeros(T, dtype=torch.int32, device=DEV)
    ops.decode_pick(logp, 0, 1.0, None, 0, seq, slp, it, unf, cnt[0:1], None)
    v, i = logp.cpu().max(1)
    assert torch.equal(seq[:, 0].cpu(), i * (i > 0)) and torch.equal(slp[:, 0].cpu(), v)
    assert torch.equal(unf.cpu(), (i > 0).int()) and (int(cnt[0]) != 0) == bool((i > 0).any())      # a flag: non-zero iff a row is unfinished
    # step 1: a finished row stays finished even if it would pick a word; seqLogprobs still written (un-masked)
    ops.decode_pick(logp.roll(1, 1).contiguous(), 0, 1.0, None, 1, seq, slp, it, unf, cnt[1:2], cnt[0:1])
    assert int(seq[2, 1]) == 0 and float(slp[2, 1]) != 0.0
    # top-k with injected uniforms == the oracle's inverse-CDF rule
    u = torch.tensor([0.0, 0.3, 0.5, 0.7, 0.95, 0.999], device=DEV)
    seq.zero_(); slp.zero_()
    ops.decode_pick(logp, 3, 0.6, u, 0, seq, slp, it, unf, cnt[2:3], None)
    lp = torch.log_softmax(logp.cpu().double() / 0.6, 1)
    top, idx = torch.topk(lp, 3, 1)
    pr = torch.exp(top - torch.logsumexp(top, 1, keepdim=True)); cdf = pr.cumsum(1)
    pick = (u.cpu().double().view(-1, 1) >= cdf).sum(1).clamp(max=2)
    want = idx.gather(1, pick.view(-1, 1)).view(-1)
    assert torch.equal(seq[:, 0].cpu(), want * (want > 0))
    close(slp[:, 0], top.gather(1, pick.view(-1, 1)).view(-1), atol=1e-5)
    # device-side early break: once the live count is 0 nothing is written any more
    zero = torch.zeros(1, dtype=torch.int32, device=DEV)
    before = seq.clone()
    ops.decode_pick(logp, 0, 1.0, None, 5, seq, slp, it, unf, cnt[5:6], zero)
    assert torch.equal(seq, before)


def test_dropout_mask_rate_and_determinism():
    a = ops.dropout_mask((1000, 1000), 0.5, 123, 0, DEV)
    b = ops.dropout_mask((1000, 1000), 0.5, 123, 0, DEV)
    c = ops.dropout_mask((1000, 1000), 0.5, 124, 0, DEV)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert abs(float(a.float().mean()) - 0.5) < 5e-3
    assert abs(float(ops.dropout_mask((777, 333), 0.2, 9, 4, DEV).float().mean()) - 0.8) < 5e-3
    # consecutive 4-aligned segments of one stream == one launch over the concatenation (AttModel._masks draws all of a forward's masks at once)
    n1, n2, n3 = 1001, 640 * 37, 17 * 333
    o2, o3 = (n1 + 3) // 4 * 4, (n1 + 3) // 4 * 4 + (n2 + 3) // 4 * 4
    whole = ops.dropout_mask((o3 + n3,), 0.5, 77, 0, DEV)
    assert torch.equal(whole[:n1], ops.dropout_mask((n1,), 0.5, 77, 0, DEV))
    assert torch.equal(whole[o2:o2 + n2], ops.dropout_mask((n2,), 0.5, 77, o2, DEV))
    assert torch.equal(whole[o3:o3 + n3], ops.dropout_mask((n3,), 0.5, 77, o3, DEV))


def test_clip_adam_matches_torch():
    n = 10007
    p0, g0 = rnd(n, seed=1), rnd(n, seed=2, scale=5.0)
    pr = p0.clone().cpu().requires_grad_(); pr.grad = g0.clone().cpu()
    opt = torch.optim.Adam([pr], lr=5e-4, betas=(0.9, 0.999), eps=1e-8)
    tot = pr.grad.norm(2); pr.grad.mul_(10.0 / max(float(tot), 10.0))
    opt.step()
    p, g = p0.clone(), g0.clone(); m = torch.zeros(n, device=DEV); v = torch.zeros(n, device=DEV)
    ss = torch.zeros(1, device=DEV)
    ops.sumsq(g, ss)
    ops.clip_adam_step(p, g, m, v, ss, 10.0, 5e-4, 0.9, 0.999, 1e-8, 0.0, 1)
    close(g, pr.grad, atol=1e-6); close(p, pr.detach(), atol=1e-6)
    # the same sweep with the iteration's optimizer.zero_grad() folded in (subgc_clip_adam_step_zero): identical update, gradient zeroed
    for n2 in (n, 10008):                                     # scalar and float4 forms
        p2 = torch.cat([p0, p0[:n2 - n]]) if n2 > n else p0.clone()
        g2 = torch.cat([g0, g0[:n2 - n]]) if n2 > n else g0.clone()
        pa, ga = p2.clone(), g2.clone()
        ma, va, mb, vb = (torch.zeros(n2, device=DEV) for _ in range(4))
        s2 = torch.zeros(1, device=DEV); ops.sumsq(g2, s2)
        ops.clip_adam_step(pa, ga, ma, va, s2, 10.0, 5e-4, 0.9, 0.999, 1e-8, 0.0, 1)
        pb, gb = p2.clone(), g2.clone()
        ops.clip_adam_step(pb, gb, mb, vb, s2, 10.0, 5e-4, 0.9, 0.999, 1e-8, 0.0, 1, zero_grad=True)
        assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb) and float(gb.abs().max()) == 0.0 and float(ga.abs().max()) > 0


def test_flat_adam_zero_grad_skips_the_next_fill_only_while_it_is_safe():
    """FlatAdam.step(zero_grad=True) leaves the bucket zeroed; the next flatten_grads skips its fill pass -- but not after a forward
    (a backward may have written gradients without a prepare in between) and not after a torch op wrote to the buffer."""
    import argparse
    from subgc import parallel
    import subgc.models as models
    from test_packed_gpu import OPT
    torch.manual_seed(0)
    m = models.setup(argparse.Namespace(**OPT)).to(DEV).train()
    adam = parallel.FlatAdam(m)
    g = m.flatten_grads()
    g.fill_(1.0)
    adam.step(zero_grad=True)
    torch.cuda.synchronize()
    assert float(g.abs().max()) == 0.0 and m.__dict__.get("_grads_are_zero") is not None
    m.flatten_grads()
    assert m.__dict__.get("_grads_are_zero") is None            # consumed
    g.fill_(2.0); adam.step(zero_grad=True)
    g.add_(3.0)                                                # a torch write after the sweep: the flag must not be trusted
    m.flatten_grads(); torch.cuda.synchronize()
    assert float(g.abs().max()) == 0.0
    g.fill_(2.0); adam.step(zero_grad=False); torch.cuda.synchronize()
    assert float(g.abs().max()) > 0.0                          # torch semantics by default: gradients stay readable
    m.flatten_grads(); torch.cuda.synchronize()
    assert float(g.abs().max()) == 0.0


def test_gemm_splitk_plain_bias_and_accumulate():
    """M = 640 recurrent shapes take the split-K form (workspace partials + reduce) once a workspace is
    registered: same numbers as the tile form."""
    ops.ensure_workspace(DEV)
    M, N, K = 640, 3000, 4000
    a = rnd(M, K, seed=1); w = rnd(N, K, seed=2); bias = rnd(N, seed=3)
    ref = (a.double() @ w.double().t() + bias.double()).float()
    out = torch.full((M, N), 3.0, device=DEV)
    ops.gemm(a, w, out, tb=True, bias=bias)
    close(out, ref, atol=2e-2, rtol=1e-4)
    big = torch.ones(M, N + 8, device=DEV); view = big[:, 4:4 + N]
    ops.gemm(a, w, view, tb=True, accum=True)
    close(view, ref - bias + 1.0, atol=2e-2, rtol=1e-4)
    assert float((big[:, :4] - 1).abs().max()) == 0 and float((big[:, 4 + N:] - 1).abs().max()) == 0
    wn = rnd(K, 2000, seed=4)
    out2 = torch.empty(M, 2000, device=DEV)
    ops.gemm(a, wn, out2)
    close(out2, (a.double() @ wn.double()).float(), atol=2e-2, rtol=1e-4)


@pytest.mark.parametrize("M,N,K", [(1, 9488, 1000), (10, 4000, 2000), (10, 4000, 3000), (16, 512, 1000), (7, 2048, 2048), (3, 64, 48), (10, 9488, 1000)])
def test_gemm_skinny_weight_streaming_form(M, N, K):
    """decode shapes (M = captions of one image <= 16): the weight-streaming kernel, incl. strided A, bias, relu"""
    big = rnd(M, K + 8, seed=1); a = big[:, 4:4 + K] if (K % 4 == 0) else big[:, :K]
    w = rnd(N, K, seed=2); bias = rnd(N, seed=3)
    out = torch.empty(M, N, device=DEV)
    ops.gemm(a, w, out, tb=True, bias=bias, relu=True)
    close(out, torch.relu(a.double() @ w.double().t() + bias.double()).float(), atol=2e-4 * K ** 0.5, rtol=1e-4)
    ops.gemm(a, w, out, tb=True)
    close(out, (a.double() @ w.double().t()).float(), atol=2e-4 * K ** 0.5, rtol=1e-4)


@pytest.mark.parametrize("rows,cols,k", [(1, 50, 5), (7, 9488, 12), (33, 1000, 32), (4, 16384, 3), (5, 300, 8)])
def test_row_topk_matches_sorted_log_softmax(rows, cols, k):
    """subgc_row_topk_f32 == the leading k columns of torch.sort(log_softmax(x), descending) (CaptionModel.py:60);
    ties resolve to the smaller index; a strided (ld > cols) input is honoured."""
    torch.manual_seed(rows * 131 + cols)
    buf = torch.randn(rows, cols + 3, device=DEV) * 3
    x = buf[:, :cols]
    x[:, 5] = x[:, 2]                                            # exact ties
    vals = torch.empty(rows, k, device=DEV)
    idx = torch.empty(rows, k, device=DEV, dtype=torch.int32)
    ops.row_topk(x, k, vals, idx, log_softmax=True)
    lp = torch.log_softmax(x.double(), 1)
    order = torch.argsort(-lp, dim=1, stable=True)[:, :k]
    np.testing.assert_array_equal(idx.cpu().numpy(), order.cpu().numpy())
    np.testing.assert_allclose(vals.cpu().numpy(), lp.gather(1, order).float().cpu().numpy(), atol=2e-6, rtol=1e-6)
    ops.row_topk(x, k, vals, idx, log_softmax=False)
    np.testing.assert_array_equal(vals.cpu().numpy(), x.gather(1, order).cpu().numpy())


@pytest.mark.parametrize("mode,M,N,K", [("nt", 1280, 4000, 1000), ("nn", 640, 3000, 4000), ("tn", 4000, 1000, 2176), ("nt", 2560, 512, 2048)])
def test_gemm_bf16x3_split_mode_is_fp32_grade(mode, M, N, K):
    """SUBGC_GEMM_MODE_BF16X3: fp32 operands split exactly into 3 bf16 planes, 6 bf16-MFMA terms, fp32 accumulate
    (csrc/gemm_x3.h).  Its error against an fp64 product must stay within 2x of the fp32-MFMA kernel's."""
    from subgc import _lib
    ops.ensure_workspace(DEV)
    a = rnd(*((K, M) if mode == "tn" else (M, K)), seed=5)
    b = rnd(*((N, K) if mode == "nt" else (K, N)), seed=6)
    ad = a.double().t() if mode == "tn" else a.double()
    bd = b.double().t() if mode == "nt" else b.double()
    ref = ad @ bd
    errs = []
    for m in ("f32", "bf16x3"):
        with ops.gemm_mode(m):
            out = torch.empty(M, N, device=DEV)
            ops.gemm(a, b, out, ta=mode == "tn", tb=mode == "nt")
            errs.append(float((out.double() - ref).abs().max()))
    scale = float(ref.abs().max())
    assert errs[0] < 2e-5 * scale and errs[1] < 2e-5 * scale, errs
    assert errs[1] <= 2.0 * errs[0] + 1e-7 * scale, errs


def test_gemm_bf16_mode_rounds_operands_only():
    """mode "bf16": operands rounded to nearest-even bf16, fp32 accumulate -> equals an fp64 product of the ROUNDED operands
    to fp32 accumulation accuracy, and differs from the exact product at the bf16 level."""
    ops.ensure_workspace(DEV)
    M, N, K = 2560, 4000, 1024                                   # >= 384 tiles of 128x128: the form the modes apply to
    a, b = rnd(M, K, seed=8), rnd(N, K, seed=9)
    out = torch.empty(M, N, device=DEV)
    with ops.gemm_mode("bf16"):
        ops.gemm(a, b, out, tb=True)
    ar, br = a.bfloat16().double(), b.bfloat16().double()
    ref_r = ar @ br.t()
    ref = a.double() @ b.double().t()
    scale = float(ref.abs().max())
    assert float((out.double() - ref_r).abs().max()) < 2e-5 * scale
    err = float((out.double() - ref).abs().max())
    assert 1e-4 * scale < err < 2e-2 * scale
    assert ops.gemm_mode.current == "f32"


def test_batched_nms_equals_per_image_nms():
    torch.manual_seed(4)
    sizes = [40, 1, 0, 200, 7]
    N = 37
    tot = sum(sizes)
    score = torch.rand(tot, device=DEV)
    score[5] = score[9]                                              # a tie
    lens = torch.randint(1, 10, (tot,), device=DEV, dtype=torch.int32)
    idx = torch.stack([torch.randperm(14, device=DEV)[:N - 23].repeat(3)[:N] for _ in range(tot)]).contiguous()
    keep, n_keep, offs = ops.subgraph_nms_batched(score, idx, lens, sizes, 0.4, 8)
    for b, (g0, n) in enumerate(zip(offs[:-1], sizes)):
        k1, n1 = ops.subgraph_nms(score[g0:g0 + n], idx[g0:g0 + n], lens[g0:g0 + n], 0.4, 8)
        c = int(n1.item()) if n else 0
        assert int(n_keep[b]) == c
        np.testing.assert_array_equal(keep[g0:g0 + c].cpu().numpy(), k1[:c].cpu().numpy())


@pytest.mark.parametrize("S,R,T,N", [(9, 48, 5, 7), (40, 1000, 17, 37), (6, 2048, 20, 12)])
def test_deferred_dv_equals_the_per_step_accumulation(S, R, T, N):
    """subgc_attn_bwd with dv = NULL + ONE subgc_attn_dv_accum after the loop (packed layout: step t holds its live sentences
    as a prefix) vs the per-step read-modify-write of d(v), and vs the plain sum  dv_j = sum_t alpha_t[s, j] * dctx_t[s]."""
    g = torch.Generator().manual_seed(S + R)
    A = 32
    lens = torch.randint(1, N + 1, (S,), generator=g).to(torch.int32)
    off = (torch.cumsum(lens, 0) - lens).to(torch.int32)
    rows = int(lens.sum())
    live = sorted(torch.randint(1, S + 1, (T,), generator=g).tolist(), reverse=True)       # live sentences per step: non-increasing
    live[0] = S
    step_off = torch.tensor([0] + list(np.cumsum(live)), dtype=torch.int32)
    tot = int(step_off[-1])
    rnd = lambda *s: torch.randn(*s, generator=g).to(DEV)
    u, v, ah, w_a = rnd(rows, A), rnd(rows, R), rnd(tot, A), rnd(1, A)
    alpha = torch.rand(tot, N, generator=g).to(DEV)
    dctx = rnd(tot, R)
    lens_d, off_d, so_d = lens.to(DEV), off.to(DEV), step_off.to(DEV)
    new = lambda *s: torch.empty(*s, device=DEV)
    du1, dv1, du2 = torch.zeros(rows, A, device=DEV), torch.zeros(rows, R, device=DEV), torch.zeros(rows, A, device=DEV)
    keep, dv2 = new(tot, R), torch.full((rows, R), 7.0, device=DEV)                          # dv2 is overwritten, never read
    dah1, dah2, dwa1, dwa2, dba1, dba2 = new(tot, A), new(tot, A), new(tot, A), new(tot, A), new(tot), new(tot)
    for t in range(T - 1, -1, -1):
        o, m = int(step_off[t]), live[t]
        sl = slice(o, o + m)
        ops.attn_bwd(u, v, ah[sl], w_a, off_d, lens_d, alpha[sl], dctx[sl], dah1[sl], du1, dv1, dwa1[sl], dba1[sl], m, A, R)
        ops.attn_bwd(u, v, ah[sl], w_a, off_d, lens_d, alpha[sl], dctx[sl], dah2[sl], du2, None, dwa2[sl], dba2[sl], m, A, R, dctx_keep=keep[sl])
    ops.attn_dv_accum(alpha, keep, so_d, T, off_d, lens_d, dv2, S, R)
    assert torch.equal(keep, dctx) and torch.equal(du1, du2) and torch.equal(dah1, dah2) and torch.equal(dwa1, dwa2) and torch.equal(dba1, dba2)
    want = torch.zeros(rows, R, dtype=torch.float64)
    al, dc = alpha.double().cpu(), dctx.double().cpu()
    for t in range(T):
        for s in range(live[t]):
            f = int(step_off[t]) + s
            want[int(off[s]):int(off[s]) + int(lens[s])] += al[f, :int(lens[s]), None] * dc[f][None, :]
    torch.testing.assert_close(dv2.double().cpu(), want, atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(dv1, dv2, atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("M,N,bf", [(9000, 4000, False), (16640, 1024, True), (21760, 9488, False), (300, 512, False), (5000, 37, False)])
def test_column_sums_slab_partials_and_atomic_fallback(M, N, bf):
    """subgc_colsum_*: tall matrices take the slab-partials form (workspace + finishing launch, fixed summation order), short
    or unaligned ones the atomic form; accumulate, a device-side row count, and bit-reproducibility of the partials form."""
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, N, generator=g).to(DEV)
    if bf:
        x = x.to(torch.bfloat16)
    ref = x.double().sum(0)
    a = ops.colsum(x)
    torch.testing.assert_close(a.double(), ref, atol=2e-3 * (M ** 0.5) / 30, rtol=1e-5)
    if N % 4 == 0 and M > 4096:                                                       # the partials form: same bits every time
        assert torch.equal(a, ops.colsum(x))
    base = torch.full((N,), 3.0, device=DEV)
    ops.colsum(x, out=base, accumulate=True)
    torch.testing.assert_close(base.double(), ref + 3.0, atol=2e-3 * (M ** 0.5) / 30, rtol=1e-5)
    rows = M // 3
    m_dev = torch.tensor([rows], dtype=torch.int32, device=DEV)
    torch.testing.assert_close(ops.colsum(x, m_dev=m_dev).double(), x[:rows].double().sum(0), atol=2e-3 * (M ** 0.5) / 30, rtol=1e-5)


@pytest.mark.parametrize("mode,M,N,K", [("nt", 8320, 1024, 512), ("nn", 8320, 1024, 512), ("nt", 8200, 512, 96), ("nt", 16520, 2048, 64)])
def test_gemm_f32_row_cut_shapes(mode, M, N, K):
    """Tile counts a few tiles above whole rounds of the 512 workgroup slots (Sub_GC_Kar's 8320 relation rows x 1024 columns: 520 tiles of
    128 x 128) run as two launches -- the whole rounds and the remaining rows (gemm_f32.hip, subgc_gemm_f32); every epilogue operand is
    row-offset with them."""
    g = torch.Generator().manual_seed(M + K)
    a = torch.randn(M, K + 4, generator=g).to(DEV)[:, :K]
    b = (torch.randn(N, K + 8, generator=g).to(DEV)[:, :K]) if mode == "nt" else (torch.randn(K, N + 8, generator=g).to(DEV)[:, :N])
    bias, add = torch.randn(N, generator=g).to(DEV), torch.randn(M, N, generator=g).to(DEV)
    keep = (torch.rand(M, N, generator=g) < 0.5).to(torch.uint8).to(DEV)
    ref = a.double() @ (b.double().t() if mode == "nt" else b.double())
    want = torch.relu(ref + bias.double() + add.double()) * keep.double() * 2.0
    out = torch.full((M, N), float("nan"), device=DEV)
    ops.gemm(a, b, out, tb=mode == "nt", bias=bias, add=add, relu=True, keep=keep, keep_scale=2.0)
    assert float((out.double() - want).abs().max()) < 2e-5 * float(want.abs().max())
    plain = torch.full((M, N), float("nan"), device=DEV)
    ops.gemm(a, b, plain, tb=mode == "nt")
    assert float((plain.double() - ref).abs().max()) < 2e-5 * float(ref.abs().max())
    acc = add.clone()
    ops.gemm(a, b, acc, tb=mode == "nt", accum=True)
    assert float((acc.double() - ref - add.double()).abs().max()) < 2e-5 * float(ref.abs().max())


@pytest.mark.parametrize("mode,M,N,K", [("nn", 8320, 512, 1024), ("nt", 8320, 1024, 512), ("nn", 4736, 512, 1024), ("tn", 1024, 512, 8320),
                                        ("nt", 640, 1000, 264), ("nn", 300, 72, 40), ("nt", 8320, 512, 1024)])
def test_gemm_f32_pair_equals_two_single_launches(mode, M, N, K):
    """subgc_gemm_f32_pair: two fp32 products of one shape in one launch (halves of the grid; tile, K parts and row cut chosen for both
    together) against fp64 and against two subgc_gemm_f32 calls."""
    g = torch.Generator().manual_seed(M + N + K)
    mk_a = lambda: torch.randn(*((K, M + 4) if mode == "tn" else (M, K + 4)), generator=g).to(DEV)[:, :(M if mode == "tn" else K)]
    mk_b = lambda: torch.randn(*((N, K + 8) if mode == "nt" else (K, N + 8)), generator=g).to(DEV)[:, :(K if mode == "nt" else N)]
    a1, a2, b1, b2 = mk_a(), mk_a(), mk_b(), mk_b()
    bias1, bias2 = torch.randn(N, generator=g).to(DEV), torch.randn(N, generator=g).to(DEV)
    ref = lambda a, b: (a.double().t() if mode == "tn" else a.double()) @ (b.double().t() if mode == "nt" else b.double())
    r1, r2 = ref(a1, b1) + bias1.double(), ref(a2, b2) + bias2.double()
    o1, o2 = torch.full((M, N), float("nan"), device=DEV), torch.full((M, N), float("nan"), device=DEV)
    ops.gemm_pair(a1, a2, b1, b2, o1, o2, ta=mode == "tn", tb=mode == "nt", bias1=bias1, bias2=bias2)
    s1, s2 = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
    ops.gemm(a1, b1, s1, ta=mode == "tn", tb=mode == "nt", bias=bias1)
    ops.gemm(a2, b2, s2, ta=mode == "tn", tb=mode == "nt", bias=bias2)
    for o, s_, r in ((o1, s1, r1), (o2, s2, r2)):
        tol = 2e-5 * float(r.abs().max())
        assert float((o.double() - r).abs().max()) < tol and float((o - s_).abs().max()) <= tol
    p1, p2 = o1.clone(), o2.clone()
    ops.gemm_pair(a1, a2, b1, b2, p1, p2, ta=mode == "tn", tb=mode == "nt", accum=True)
    assert float((p1.double() - o1.double() - r1 + bias1.double()).abs().max()) < 4e-5 * float(r1.abs().max())
    assert float((p2.double() - o2.double() - r2 + bias2.double()).abs().max()) < 4e-5 * float(r2.abs().max())


def test_wave_reductions_on_the_dpp_path():
    """wave_sum / wave_max (common.h: quad_perm, row mirrors, row_bcast, readlane 63 instead of six ds_bpermute exchanges) through the kernels
    that are nothing but such reductions: row sums of 64-column blocks (subgc_colsum's scalar form is block-based, so use row log-sum-exp: a
    wave_max and a wave_sum per row) against fp64, on rows whose maximum sits in every lane position and with negative values only."""
    V = 9488
    x = torch.randn(130, V, device=DEV) * 3 - 20.0                          # all negative: a max seeded with 0 would be wrong
    for r in range(128):
        x[r, (r * 73) % V] = -0.5 + r * 1e-3                               # the row maximum visits many lane / wave positions
    lse = ops.row_lse(x)
    want = torch.logsumexp(x.double(), dim=1)
    assert float((lse.double() - want).abs().max()) < 1e-5
