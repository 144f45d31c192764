"""Decode-time x->gates table (AttModel.xt_gates_table, subgc_token_rows_f32): with frozen weights the word-input term
of the attention LSTM (AttModel.py:332 -> :409-411) is a function of the token alone.  The lookup must equal embed + GEMM,
follow weight updates, and leave the decoded tokens of the golden cases unchanged."""
import numpy as np
import pytest
import torch

from subgc import functions as F_
from subgc import ops, synthetic
from test_parity_gpu import DEV, build

pytestmark = pytest.mark.gpu


def test_token_rows_kernel_is_a_plain_lookup():
    g = torch.Generator().manual_seed(3)
    for rows, C, ld in ((50, 4000, 4000), (9488, 192, 200), (7, 37, 37)):
        table = torch.randn(rows, ld, generator=g).to(DEV)[:, :C]
        tok = torch.randint(-2, rows + 3, (33,), generator=g).to(DEV)           # out-of-range tokens are clamped like the embedding kernel
        out = torch.full((33, C), 7.0, device=DEV)
        ops.token_rows(table, tok, out)
        assert torch.equal(out, table[tok.clamp(0, rows - 1)])
    assert ops.token_rows(table, tok[:0], out[:0]).shape == (0, C)


def test_table_equals_embed_plus_gemm_and_tracks_the_weights(golden):
    g = golden("subgc_greedy")
    m = build(g, golden("subgc_train").group("weights"), False)
    R = m.rnn_size
    tab = m.xt_gates_table()
    assert tab.shape == (m.vocab_size + 1, 4 * R) and m.xt_gates_table() is tab    # cached while the weights stand
    want = torch.relu(m.P("embed.0.weight").double()) @ m.P("core.att_lstm.weight_ih")[:, 2 * R:].double().t()
    torch.testing.assert_close(tab.double(), want, atol=1e-5, rtol=1e-5)
    with torch.no_grad():
        m.P("embed.0.weight").mul_(0.5)                                            # e.g. an optimizer step between two evaluations
    tab2 = m.xt_gates_table()
    assert tab2 is not tab
    torch.testing.assert_close(tab2, 0.5 * tab, atol=1e-6, rtol=1e-5)


@pytest.mark.parametrize("name", ["subgc_greedy", "subgc_greedy_nms55"])
def test_decode_with_and_without_the_table(golden, name, monkeypatch):
    g = golden(name)
    m = build(g, golden("subgc_train").group("weights"), False)
    b = {k: v.to(DEV) for k, v in g.tensors("inputs").items()}
    opt = g.meta["sample_opt"]
    with_tab = m(*synthetic.sample_args(b), opt=opt, mode="sample")
    seen = []
    orig = F_.DecodeState.__init__

    def no_table(self, pr, P, N, want_att, xt_table=None, **kw):
        seen.append(xt_table is not None)
        orig(self, pr, P, N, want_att, None, **kw)

    monkeypatch.setattr(F_.DecodeState, "__init__", no_table)
    m.__dict__.pop("_graph_cache", None)                                          # the cached hipGraph holds a table-backed state
    without = m(*synthetic.sample_args(b), opt=opt, mode="sample")
    assert seen and all(seen)                                                      # the product path did pass a table
    assert torch.equal(with_tab[0], without[0])
    torch.testing.assert_close(with_tab[1], without[1], atol=1e-4, rtol=1e-4)
    np.testing.assert_array_equal(with_tab[0].cpu().numpy(), g.group("out")["seq"])


def test_cached_decode_state_follows_load_state_dict(golden):
    """The one-image decode replays a captured hipGraph holding weight snapshots: new weights must retire it."""
    g = golden("subgc_greedy")
    w = golden("subgc_train").group("weights")
    m = build(g, w, False)
    b = {k: v.to(DEV) for k, v in g.tensors("inputs").items()}
    opt = g.meta["sample_opt"]
    first = m(*synthetic.sample_args(b), opt=opt, mode="sample")
    w2 = {k: (v * (0.5 if k.startswith(("core.", "embed.")) else 1.0)).astype(v.dtype) for k, v in w.items()}
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w2.items()})
    again = m(*synthetic.sample_args(b), opt=opt, mode="sample")
    fresh = build(g, w2, False)(*synthetic.sample_args(b), opt=opt, mode="sample")
    assert torch.equal(again[0], fresh[0])
    torch.testing.assert_close(again[1], fresh[1], atol=1e-5, rtol=1e-5)
    assert not torch.allclose(first[1], again[1])


@pytest.mark.parametrize("S,R,K", [(1, 48, 96), (10, 1000, 2000), (16, 1000, 3000), (7, 52, 1000), (17, 48, 144), (20, 1000, 2000), (30, 1000, 3000), (32, 52, 1000)])
def test_fused_lstm_step_equals_gemm_plus_cell(S, R, K):
    """subgc_lstm_step_skinny (row-permuted weights, cell update in the GEMM epilogue) vs fp64 LSTMCell arithmetic."""
    g = torch.Generator().manual_seed(S * 1000 + R)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(DEV)
    x, W = rnd(S, K), rnd(4 * R, K, sc=K ** -0.5)
    b0, b1, add2, cp = rnd(4 * R, sc=0.3), rnd(4 * R, sc=0.3), rnd(S, 4 * R, sc=0.5), rnd(S, R)
    table = rnd(23, 4 * R, sc=0.5)
    tok = torch.randint(-1, 25, (S,), generator=g).to(DEV)
    Wp = W[ops.lstm_gate_perm(R, DEV)].contiguous()
    for use_tok in (True, False):
        c = torch.empty(S, R, device=DEV)
        wide = torch.full((S, 3 * R), 9.0, device=DEV)                              # h lands in column slices of wider buffers
        h2 = torch.empty(S, R, device=DEV)
        add1 = table if use_tok else table[tok.clamp(0, 22)].contiguous()
        ops.lstm_step_skinny(x, Wp, cp, c, [wide[:, R:2 * R], h2], b0, b1, add1, tok if use_tok else None, add2)
        pre = x.double() @ W.double().t() + b0.double() + b1.double() + table[tok.clamp(0, 22)].double() + add2.double()
        i, f, gg, o = pre[:, :R].sigmoid(), pre[:, R:2 * R].sigmoid(), pre[:, 2 * R:3 * R].tanh(), pre[:, 3 * R:].sigmoid()
        cn = f * cp.double() + i * gg
        hn = o * cn.tanh()
        torch.testing.assert_close(c.double(), cn, atol=2e-5, rtol=1e-5)
        torch.testing.assert_close(h2.double(), hn, atol=2e-5, rtol=1e-5)
        assert torch.equal(wide[:, R:2 * R], h2) and float(wide[:, :R].min()) == 9.0 and float(wide[:, 2 * R:].min()) == 9.0
    c2 = torch.empty(S, R, device=DEV)                                               # no additive terms, no c_prev
    ops.lstm_step_skinny(x, Wp, None, c2, [h2])
    pre = x.double() @ W.double().t()
    want = pre[:, :R].sigmoid() * pre[:, 2 * R:3 * R].tanh()
    torch.testing.assert_close(c2.double(), want, atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("name", ["subgc_greedy", "subgc_greedy_nms55", "subgc_sct"])
def test_decode_with_fused_and_unfused_lstm_steps(golden, name, monkeypatch):
    g = golden(name)
    m = build(g, golden("subgc_train").group("weights"), False)
    b = {k: v.to(DEV) for k, v in g.tensors("inputs").items()}
    opt = g.meta["sample_opt"]
    fused = m(*synthetic.sample_args(b), opt=opt, mode="sample")
    used = []
    orig = F_.DecodeState.__init__

    def unfused(self, *a, fuse_lstm=False, **kw):
        used.append(fuse_lstm)
        orig(self, *a, fuse_lstm=False, **kw)

    monkeypatch.setattr(F_.DecodeState, "__init__", unfused)
    m.__dict__.pop("_graph_cache", None)
    plain = m(*synthetic.sample_args(b), opt=opt, mode="sample")
    assert used and all(used)                                                        # the product path asked for the fused steps
    assert torch.equal(fused[0], plain[0])
    torch.testing.assert_close(fused[1], plain[1], atol=1e-4, rtol=1e-4)
    np.testing.assert_array_equal(fused[0].cpu().numpy(), g.group("out")["seq"])
    if opt.get("return_att"):
        torch.testing.assert_close(fused[4], plain[4], atol=1e-5, rtol=1e-4)


@pytest.mark.parametrize("M,N,K", [(17, 1024, 2048), (37, 512, 1024), (65, 1024, 512), (80, 9488, 1000), (33, 72, 300), (10, 4000, 3000)])
def test_weight_streaming_gemm_up_to_80_rows(M, N, K):
    """The matrix-pipe weight-streaming form (one-image encoder GEMMs, 37 node / 65 relation rows): bias, residual, ReLU."""
    g = torch.Generator().manual_seed(M * 7 + N)
    x, W = torch.randn(M, K, generator=g).to(DEV), (torch.randn(N, K, generator=g) * K ** -0.5).to(DEV)
    bias, add = torch.randn(N, generator=g).to(DEV), torch.randn(M, N + 8, generator=g).to(DEV)[:, :N]
    ref = x.double() @ W.double().t()
    out = torch.full((M + 1, N), 5.0, device=DEV)
    ops.gemm(x, W, out[:M], tb=True)
    torch.testing.assert_close(out[:M].double(), ref, atol=2e-5, rtol=1e-5)
    assert float(out[M].min()) == 5.0                                                # rows past M untouched
    ops.gemm(x, W, out[:M], tb=True, bias=bias, add=add, relu=True)
    torch.testing.assert_close(out[:M].double(), (ref + bias.double() + add.double()).clamp_min(0), atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("S,R,K", [(640, 1000, 2000), (384, 1000, 3000), (100, 1000, 2000), (640, 52, 104)])
def test_lstm_fwd_gemm_equals_gemm_then_lstm_fwd(S, R, K):
    """subgc_lstm_fwd_gemm: split-K partial planes summed inside the cell kernel (training steps) vs the two separate entry points."""
    ops.ensure_workspace(DEV)
    g = torch.Generator().manual_seed(S + R)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(DEV)
    x, W = rnd(S, K), rnd(4 * R, K, sc=K ** -0.5)
    g1, g2, b0, b1, cp = rnd(S, 4 * R, sc=0.5), rnd(S, 4 * R, sc=0.5), rnd(4 * R, sc=0.3), rnd(4 * R, sc=0.3), rnd(S, R)
    keep = (torch.rand(S, R, generator=g) > 0.5).to(torch.uint8).to(DEV)
    new = lambda *s: torch.empty(*s, device=DEV)
    outs = []
    for fused in (True, False):
        pre, c, h, h2, hd, gates = new(S, 4 * R), new(S, R), new(S, 2 * R), new(S, R), new(S, R), new(S, 4 * R)
        if fused:
            ops.lstm_fwd_gemm(x, W, pre, g1, g2, b0, b1, cp, c, h[:, R:], h2, keep, 2.0, hd, gates, S, R)
        else:
            ops.gemm(x, W, pre, tb=True)
            ops.lstm_fwd(pre, g1, g2, b0, b1, cp, c, h[:, R:], h2, keep, 2.0, hd, gates, S, R)
        outs.append((c, h[:, R:].clone(), h2, hd, gates))
    for a, b in zip(*outs):
        torch.testing.assert_close(a, b, atol=2e-5, rtol=1e-5)


def test_one_image_graph_path_over_varied_candidate_and_survivor_counts(golden):
    """Soak of the replayed one-image path: candidate counts G, survivor counts n (incl. 0 via an empty image) and the decode
    mode change from call to call; every call must equal the eager path (same kernels, no graph, no static buffers)."""
    g = golden("subgc_greedy")
    w = golden("subgc_beam").group("weights")
    opt = g.meta["opt"]
    mk = lambda M, seed: {k: v.to(DEV) for k, v in synthetic.make_test_batch(M, seed=seed, D=opt["att_feat_size"], N=g.tensors("inputs")["att_feats"].size(1),
                                                                           K=g.tensors("inputs")["rel_ind"].size(1)).items()}
    cases = [(3, 11), (8, 12), (2, 13), (20, 14), (1, 16), (2, 17), (8, 15)]           # 2M candidates -> min(2M, 6) survivors: n in {2, 4, 6}
    mg, me = build(g, w, False, gpn_nms_thres=0.55, gpn_max_subg=6), build(g, w, False, gpn_nms_thres=0.55, gpn_max_subg=6)
    me.decode_hipgraph = False
    seen_n = set()
    for rnd in range(2):
        for M, seed in cases:
            b = mk(M, seed)
            # plain greedy takes the SPECULATIVE replay (loop queued for min(max_subg, candidates) rows before the survivor count is read;
            # fewer survivors -> surplus rows cut, log-probs after the real rows' early break zeroed); return_att and beam 2 wait for the count
            for sopt in (dict(sample_max=1, beam_size=1), dict(sample_max=1, beam_size=1, return_att=1), dict(sample_max=1, beam_size=2)):
                a = mg(*synthetic.sample_args({k: v.clone() for k, v in b.items()}), opt=sopt, mode="sample")
                e = me(*synthetic.sample_args({k: v.clone() for k, v in b.items()}), opt=sopt, mode="sample")
                seen_n.add(a[0].size(0))
                assert len(a) == len(e)
                for x, y in zip(a, e):
                    assert x.shape == y.shape
                    if x.dtype in (torch.int64, torch.int32):
                        assert torch.equal(x.cpu(), y.cpu())
                    else:
                        torch.testing.assert_close(x.cpu(), y.cpu(), atol=1e-4, rtol=1e-4)
                if sopt["beam_size"] > 1:
                    for db, eb in zip(mg.done_beams, me.done_beams):
                        for d, h in zip(db, eb):
                            assert torch.equal(d["seq"], h["seq"]) and abs(d["p"] - h["p"]) < 1e-3
    assert len(seen_n) >= 3 and not me.__dict__.get("_graph_cache") and mg.__dict__.get("_graph_cache")
    # candidate counts 2..40 share ONE capacity class of static buffers, so graphs are keyed by survivors and mode only
    assert len(mg._front_cache) == 1 and len(mg._graph_cache) <= 3 * len(seen_n)


def test_empty_batches_are_no_ops():
    """S = 0 / n = 0 / M = 0 (an image without surviving sub-graphs): every new entry point returns without launching."""
    R, K = 48, 96
    e = lambda *s, dt=torch.float32: torch.empty(*s, device=DEV, dtype=dt)
    ops.lstm_step_skinny(e(0, K), e(4 * R, K), None, e(0, R), [e(0, R)])
    ops.lstm_fwd_gemm(e(0, K), e(4 * R, K), e(0, 4 * R), None, None, None, None, None, e(0, R), e(0, R), None, None, 1.0, None, None, 0, R)
    ops.gather_rows_multi([(e(5, 8), e(0, 8))], e(0, dt=torch.int32))
    assert ops.token_rows(e(7, 16), e(0, dt=torch.long), e(0, 16)).shape == (0, 16)
    from subgc import beam
    tb = beam.DeviceTables(0, 1, 20, 2, DEV)
    ops.beam_step(e(0, 4), e(0, 4, dt=torch.int32), tb, e(0, dt=torch.long), e(0, dt=torch.int32), 0, 20, 1, 2, 4, 50, 0, 0.5)
    torch.cuda.synchronize()


@pytest.mark.parametrize("n", [10007, 40000])                                    # scalar and float4 forms
def test_clip_adam_grad_scale_equals_scaling_first(n):
    """grad_scale = 1/world inside the fused sweep == averaging the summed gradient in a separate pass, then the plain step."""
    g = torch.Generator().manual_seed(n)
    p0, g0 = torch.randn(n, generator=g).to(DEV), (torch.randn(n, generator=g) * 9.0).to(DEV)
    outs = []
    for folded in (True, False):
        p, gr, m, v, ss = p0.clone(), g0.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV), torch.zeros(1, device=DEV)
        if not folded:
            gr.mul_(0.25)
        ops.sumsq(gr, ss)
        ops.clip_adam_step(p, gr, m, v, ss, 10.0, 5e-4, 0.9, 0.999, 1e-8, 0.01, 3, 0.25 if folded else 1.0)
        outs.append((p, gr, m, v))
    for a, b in zip(*outs):
        torch.testing.assert_close(a, b, atol=1e-6, rtol=1e-5)


@pytest.mark.parametrize("opt_over", [dict(sample_max=1, beam_size=1), dict(sample_max=1, beam_size=2)])
def test_decode_after_the_fused_optimizer_step_uses_the_new_weights(golden, opt_over):
    """train -> eval -> train -> eval: `parallel.FlatAdam.step` writes the weights through raw device pointers (no torch
    version counter moves), so it has to retire the decode-time snapshots itself (x->gates table, K-concatenated LSTM
    matrices, captured hipGraphs); a second decode must equal a freshly built model holding the updated weights."""
    from subgc import parallel
    g = golden("subgc_greedy")
    m = build(g, golden("subgc_train").group("weights"), False)
    b = {k: v.to(DEV) for k, v in g.tensors("inputs").items()}
    opt = dict(g.meta["sample_opt"], **opt_over)
    first = m(*synthetic.sample_args(b), opt=opt, mode="sample")
    adam = parallel.FlatAdam(m, lr=5e-2)                                            # a large step so that every caption changes
    grads = m.flatten_grads()
    grads.copy_(torch.randn(grads.numel(), generator=torch.Generator().manual_seed(5)).to(DEV))
    adam.step()
    again = m(*synthetic.sample_args(b), opt=opt, mode="sample")
    w2 = {k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items()}
    fresh = build(g, w2, False)(*synthetic.sample_args(b), opt=opt, mode="sample")
    assert torch.equal(again[0], fresh[0])
    torch.testing.assert_close(again[1], fresh[1], atol=1e-5, rtol=1e-5)
    assert not torch.allclose(first[1], again[1])


@pytest.mark.parametrize("return_att", [0, 1])
def test_greedy_pick_folded_into_the_step_launches_equals_the_separate_pick(golden, return_att):
    """One-image greedy decode with the pick in the logits / attention-LSTM launches (DecodeState.greedy_loop: packed 64-bit arg-max,
    lazy log-sum-exp) against the same replayed loop with the separate subgc_decode_pick launch: tokens, kept sub-graphs and attention
    weights identical, log-probs to fp32 rounding; includes rows that finish early and images whose loop breaks before T."""
    g = golden("subgc_greedy")
    w = golden("subgc_beam").group("weights")
    w["logit.bias"][0] += 1.5                                      # <eos> wins early for some rows: finished-row masking and the early break
    opt = g.meta["opt"]
    N, K = g.tensors("inputs")["att_feats"].size(1), g.tensors("inputs")["rel_ind"].size(1)
    ma, mb = build(g, w, False, gpn_nms_thres=0.55, gpn_max_subg=6), build(g, w, False, gpn_nms_thres=0.55, gpn_max_subg=6)
    mb.decode_fused_pick = False
    sopt = dict(sample_max=1, beam_size=1, return_att=return_att)
    lens = set()
    for seed in range(30, 42):
        b = {k: v.to(DEV) for k, v in synthetic.make_test_batch(4 + seed % 5, seed=seed, D=opt["att_feat_size"], N=N, K=K).items()}
        x = ma(*synthetic.sample_args({k: v.clone() for k, v in b.items()}), opt=sopt, mode="sample")
        y = mb(*synthetic.sample_args({k: v.clone() for k, v in b.items()}), opt=sopt, mode="sample")
        assert len(x) == len(y)
        assert torch.equal(x[0], y[0]) and torch.equal(x[3], y[3])
        torch.testing.assert_close(x[1], y[1], atol=2e-5, rtol=1e-5)
        torch.testing.assert_close(x[2], y[2], atol=1e-6, rtol=1e-6)
        if return_att:
            assert x[4].shape == y[4].shape
            torch.testing.assert_close(x[4], y[4], atol=1e-6, rtol=1e-5)
        lens.update((x[0] > 0).sum(1).tolist())
    assert len(lens) >= 3 and min(lens) < ma.seq_length              # rows of different lengths were decoded


def test_no_garbage_collection_inside_a_graph_capture(golden):
    """A dead reference cycle that owns an older hipGraph must not be collected while another capture is in progress (freeing the graph's
    pool during a capture aborts the process): ops.graph_capture collects before it starts and keeps the collector off until it ends."""
    import gc
    from subgc.models import sampling
    g = golden("subgc_greedy")
    b = {k: v.to(DEV) for k, v in g.tensors("inputs").items()}
    sopt = g.meta["sample_opt"]
    w = golden("subgc_train").group("weights")
    old = build(g, w, False)
    old(*synthetic.sample_args({k: v.clone() for k, v in b.items()}), opt=sopt, mode="sample")     # captures a graph
    old.self_cycle = old                                                                             # dead cycle once dropped
    del old
    seen = []
    real = sampling._GraphedLoop._loop

    def spy(self):
        if torch.cuda.is_current_stream_capturing():
            seen.append(gc.isenabled())
        return real(self)

    sampling._GraphedLoop._loop = spy
    thr = gc.get_threshold()
    gc.set_threshold(1)                                                                              # collect at every opportunity
    try:
        m = build(g, w, False)
        out = m(*synthetic.sample_args({k: v.clone() for k, v in b.items()}), opt=sopt, mode="sample")
    finally:
        gc.set_threshold(*thr)
        sampling._GraphedLoop._loop = real
    assert seen == [False] and gc.isenabled()
    assert torch.equal(out[0].cpu(), torch.from_numpy(g.group("out")["seq"]))


@pytest.mark.parametrize("S,R,V,K1,K2,wb16", [(10, 1000, 9488, 1000, 2000, False), (16, 48, 150, 96, 144, False), (1, 52, 1003, 100, 104, False),
                                              (10, 1000, 9488, 1000, 2000, True), (7, 48, 77, 96, 48, True)])
def test_dual_weight_stream_and_cell_pick_kernels(S, R, V, K1, K2, wb16):
    """Round 6's greedy step, kernel by kernel against torch: subgc_skinny_dual = [logits with the arg-max / log-sum-exp epilogue | a
    gate product in the permuted row order, written gate-major], then subgc_lstm_cell_pick = the attention LSTM's cell whose word is
    that arg-max, with the bookkeeping of AttModel.py:295-319 (finished rows feed 0, seq / unfinished / live count)."""
    g = torch.Generator().manual_seed(S * 131 + R)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(DEV)
    h, Wl, bl = rnd(S, K1), rnd(V, K1, sc=K1 ** -0.5), rnd(V, sc=0.2)
    Wl[5] = Wl[3]; bl[5] = bl[3]                                                    # an exact tie: the smaller column must win
    x2, Wg = rnd(S, K2), rnd(4 * R, K2, sc=K2 ** -0.5)
    Wgp = Wg[ops.lstm_gate_perm(R, DEV)].contiguous()
    if wb16:
        Wl, Wgp, Wg = Wl.bfloat16(), Wgp.bfloat16(), Wg.bfloat16()
    logits_ref = h.double() @ Wl.double().t() + bl.double()
    gates_ref = x2.double() @ Wg.double().t()
    # (a) plain dual: both results written
    out1, out2 = torch.empty(S, V, device=DEV), torch.full((S, 4 * R + 8), 7.0, device=DEV)
    ops.skinny_dual(h, Wl, out1, x2, Wgp, out2[:, :4 * R], bias1=bl, unperm2_R=R)
    torch.testing.assert_close(out1.double(), logits_ref, atol=3e-5, rtol=1e-5)
    torch.testing.assert_close(out2[:, :4 * R].double(), gates_ref, atol=3e-5, rtol=1e-5)
    assert float(out2[:, 4 * R:].min()) == 7.0                                      # nothing past the gate columns
    # (b) with the pick epilogue: no logits in memory, arg-max of out1's own bits, log-sum-exp partials
    wgs = (V + 15) // 16
    best = torch.zeros(2, 16 * 8 * 16, dtype=torch.int64, device=DEV)
    lse = torch.zeros(1, wgs * 16 * 2, device=DEV)
    pre = torch.empty(S, 4 * R, device=DEV)
    ops.skinny_dual(h, Wl, None, x2, Wgp, pre, bias1=bl, unperm2_R=R, best=best[0], lse_part=lse[0])
    assert torch.equal(pre, out2[:, :4 * R])
    word = out1.argmax(1)                                                           # torch: the first maximum
    # the cell: table row of the picked word (finished rows feed word 0), fc term, two biases
    T, t_prev = 5, 2
    table, add2, b0, b1, cp = rnd(V, 4 * R, sc=0.3), rnd(S, 4 * R, sc=0.3), rnd(4 * R, sc=0.2), rnd(4 * R, sc=0.2), rnd(S, R)
    unf_in = (torch.arange(S) % 3 != 1).int().to(DEV)
    unf_out, seq = torch.full((S,), -5, dtype=torch.int32, device=DEV), torch.full((S, T), -1, dtype=torch.int64, device=DEV)
    count, prev = torch.zeros(1, dtype=torch.int32, device=DEV), torch.ones(1, dtype=torch.int32, device=DEV)
    best[1].fill_(3)
    c, h0, hw = torch.empty(S, R, device=DEV), torch.empty(S, R, device=DEV), torch.full((S, 3 * R), 9.0, device=DEV)
    ops.lstm_cell_pick(pre, cp, c, [h0, hw[:, R:2 * R]], b0, b1, table, add2, pick=(best[0], unf_in, unf_out, seq, t_prev, count, prev), best_reset=best[1])
    fed = word * unf_in.long()
    assert torch.equal(seq[:, t_prev], fed) and int(seq[:, :t_prev].max()) == -1 and int(seq[:, t_prev + 1:].max()) == -1
    assert torch.equal(unf_out, (unf_in.bool() & (fed > 0)).int()) and int(count) == int(unf_out.sum())
    assert int(best[1].view(128, 16)[:, 0].abs().max()) == 0                        # the other buffer's 128 slot words (one per 128-byte line) are ready for this step's logits
    p = pre.double() + table[fed].double() + add2.double() + b0.double() + b1.double()
    i, f, gg, o = p[:, :R].sigmoid(), p[:, R:2 * R].sigmoid(), p[:, 2 * R:3 * R].tanh(), p[:, 3 * R:].sigmoid()
    cn = f * cp.double() + i * gg
    torch.testing.assert_close(c.double(), cn, atol=2e-5, rtol=1e-5)
    torch.testing.assert_close(h0.double(), o * cn.tanh(), atol=2e-5, rtol=1e-5)
    assert torch.equal(hw[:, R:2 * R], h0) and float(hw[:, :R].min()) == 9.0 and float(hw[:, 2 * R:].min()) == 9.0
    # a loop the reference has already left (*prev_count == 0) files nothing
    seq2, cnt2 = seq.clone(), torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.lstm_cell_pick(pre, cp, c, [h0], b0, b1, table, add2, pick=(best[0], unf_in, unf_out.clone(), seq2, t_prev + 1, cnt2, cnt2.clone()), best_reset=None)
    assert torch.equal(seq2, seq) and int(cnt2) == 0
    # the word given directly (step 0)
    tok = torch.randint(0, V, (S,), generator=g).to(DEV)
    ops.lstm_cell_pick(pre, cp, c, [h0], b0, b1, table, add2, tok=tok)
    p = pre.double() + table[tok].double() + add2.double() + b0.double() + b1.double()
    cn = p[:, R:2 * R].sigmoid() * cp.double() + p[:, :R].sigmoid() * p[:, 2 * R:3 * R].tanh()
    torch.testing.assert_close(c.double(), cn, atol=2e-5, rtol=1e-5)
    # the log-probability of the picked word from the per-workgroup (max, sum exp) partials
    counts = torch.full((1,), S, dtype=torch.int32, device=DEV)
    seqlp = torch.zeros(S, 1, device=DEV)
    ops.pick_lse_finish(lse, V, counts, seqlp)
    want = out1.double().max(1).values - torch.logsumexp(out1.double(), 1)
    torch.testing.assert_close(seqlp[:, 0].double(), want, atol=2e-5, rtol=1e-5)
