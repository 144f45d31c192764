"""Shared attention sets of the Full-GC model (functions.PreparedShared + csrc/attention_group.hip): every sentence of an image attends
over the same node rows (reference AttModel.py:140-149 on the x5 replicated features of gcn_backbone.py:50-51), so att_embed /
ctx2att run once per image and one workgroup serves the image's sentences.  With dropout off the result must equal the replicated
(per-sentence) path: loss, log-probabilities and every gradient; the golden Full-GC cases run through it as well (test_parity_gpu /
test_bf16_storage_gpu run with dropout off, where the default "auto" mode shares).  With dropout ON the reference draws five INDEPENDENT
att_embed keep-masks per image (gcn_backbone.py:50-51 -> AttModel.py:113-119): the default then runs the replicated path, pinned here
against the oracle with injected per-sentence masks; share_attention_sets = 1 (tied masks) is an explicit opt-in."""
import numpy as np
import pytest
import torch

from subgc import synthetic
import subgc.models as models
from test_parity_gpu import DEV, build, close, run_train

pytestmark = pytest.mark.gpu


def _batch(g, B, seed, ragged):
    o = g.meta["opt"]
    b = synthetic.make_train_batch(B, D=o["att_feat_size"], vocab=o["vocab_size"], n_obj_cls=o.get("sg_obj_cnt", 1599), seed=seed,
                                   fc_size=o["fc_feat_size"], min_len=1 if ragged else 5, max_len=16)
    if ragged:
        b["att_masks"][::3, 0, 0, 36] = 1.0                  # some sentences attend over 37 rows, the others over 36
        b["labels"][5:8] = 0                                 # sentences with no words at all: dead after step 0 in the packed decoder
        b["masks"][5:8, 2:] = 0
    return b


@pytest.mark.parametrize("packed", [True, False])
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("ragged", [False, True])
def test_shared_sets_equal_replicated_sets(golden, packed, dtype, ragged):
    g = golden("fullgc_train")
    res = {}
    for share in (1, 0):
        m = build(g, g.group("weights"), True, compute_dtype=dtype, share_attention_sets=share, drop_prob_lm=0.0)
        m.packed_decoder = packed
        batch = _batch(g, 4, 21, ragged)
        out, loss = run_train(m, batch)
        with torch.no_grad():
            outputs = m(*synthetic.forward_args({k: v.to(DEV) for k, v in batch.items()}))[0]
        res[share] = (float(out["lang_loss"]), outputs.cpu(), {k: p.grad.clone().cpu() for k, p in m.named_parameters()})
    l1, o1, g1 = res[1]
    l0, o0, g0 = res[0]
    tol = dict(atol=3e-2, rtol=3e-2) if dtype == "bf16" else dict(atol=2e-5, rtol=2e-4)
    assert abs(l1 - l0) < (2e-2 if dtype == "bf16" else 2e-5) * max(1.0, abs(l0))
    close(o1, o0, "outputs", **tol)
    top = max(float(v.abs().max()) for v in g0.values())
    for k in g0:
        sc = float(g0[k].abs().max())
        if sc == 0.0:
            assert float(g1[k].abs().max()) == 0.0, k
            continue
        if k == "pred_emb_prj.bias" or k.endswith(("fc_rgt.bias", "fc_lft.bias")) or sc < 1e-5 * top:
            # a constant shift in front of a BatchNorm: the true gradient is zero, what is there is rounding noise
            assert float(g1[k].abs().max()) < (2e-2 if dtype == "bf16" else 1e-4) * top, k
            continue
        if dtype == "bf16":
            a, b = g1[k].double().flatten(), g0[k].double().flatten()
            cos = float((a @ b) / (a.norm() * b.norm() + 1e-30))
            assert cos > 0.99, (k, cos)
        else:
            np.testing.assert_allclose(g1[k].numpy(), g0[k].numpy(), atol=3e-5 * sc + 1e-9, rtol=2e-3, err_msg=k)


def test_auto_mode_shares_only_where_it_is_the_reference_arithmetic(golden):
    """default (-1): PreparedShared when att_embed's dropout is inactive (p = 0 here; eval mode likewise), the reference's replicated
    rows with independent masks when it is active; 1 forces sharing (tied masks), 0 forbids it; injected masks are per sentence."""
    from subgc import functions as F_
    g = golden("fullgc_train")
    seen = []
    orig = F_.make_prepared

    def spy(meta, *a, **k):
        pr = orig(meta, *a, **k)
        seen.append(type(pr).__name__)
        return pr

    F_.make_prepared = spy
    try:
        for p_drop, share in ((0.0, -1), (0.5, -1), (0.5, 1), (0.0, 0), (0.5, 0)):
            m = build(g, g.group("weights"), True, drop_prob_lm=p_drop, share_attention_sets=share)
            assert m.share_attention_sets == share
            run_train(m, _batch(g, 2, 3, False))
        m = build(g, g.group("weights"), True, drop_prob_lm=0.5)                     # the default of a model built without the option
        assert m.share_attention_sets == -1
        run_train(m, _batch(g, 2, 3, False))
        m.eval()
        with torch.no_grad():
            m(*synthetic.forward_args({k: v.to(DEV) for k, v in _batch(g, 2, 3, False).items()}))
    finally:
        F_.make_prepared = orig
    assert seen == ["PreparedShared", "Prepared", "PreparedShared", "Prepared", "Prepared", "Prepared", "PreparedShared"], seen


def test_fullgc_train_with_independent_per_sentence_masks_matches_oracle(golden):
    """The default Full-GC training path under dropout: every sentence of an image has its OWN att_embed keep-mask (injected here, so
    the oracle can apply the same ones), as the reference draws them on the x5 replicated node rows.  Loss and every gradient."""
    from oracle import subgc_oracle as O
    g = golden("fullgc_train")
    w = g.group("weights")
    p = 0.5
    m = build(g, w, True, drop_prob_lm=p)
    batch = g.tensors("inputs")
    S, T, N = batch["labels"].size(0), batch["labels"].size(1) - 1, 37
    o = g.meta["opt"]
    R, E = o["rnn_size"], o["input_encoding_size"]
    gen = torch.Generator().manual_seed(11)
    mk = lambda *s: (torch.rand(*s, generator=gen) >= p).to(torch.uint8)
    masks = {"fc": mk(S, R), "att": mk(S * N, R), "xt": mk(T, S, E), "out": mk(T, S, R)}
    # five sentences of one image must really differ in their att mask (that is the point of the test)
    a = masks["att"].view(S, N, R)
    assert not torch.equal(a[0], a[1])
    m.injected_masks = {k: v.to(DEV) for k, v in masks.items()}
    out, loss = run_train(m, {k: v.clone() for k, v in batch.items()})
    lens = torch.full((S,), 36, dtype=torch.long)                                     # AttModel.py:148-149: masks forced to ones on [:, :36]
    off = torch.cumsum(lens, 0) - lens
    att = torch.zeros(S, N, R, dtype=torch.uint8)
    for s_ in range(S):
        att[s_, :36] = masks["att"][int(off[s_]): int(off[s_]) + 36]
    om = {"fc": masks["fc"].float(), "att": att.float(), "xt": masks["xt"].permute(1, 0, 2).float(), "out": masks["out"].permute(1, 0, 2).float()}
    orc = O.Oracle(g.opt(drop_prob_lm=p), w, requires_grad=True); orc.training = True
    ref = O.loss_wrapper(orc, {k: v.clone() for k, v in batch.items()}, masks=om)
    ref["lang_loss"].backward()
    close(out["lang_loss"], ref["lang_loss"], "lang_loss (independent masks)")
    top = max(float(pp.grad.abs().max()) for pp in orc.P.values() if pp.grad is not None)
    n = 0
    for k, pp in m.named_parameters():
        gk = orc.P[k].grad
        if gk is None or float(gk.abs().max()) < 1e-5 * top:
            continue
        close(pp.grad, gk, "grad " + k, atol=2e-4 * max(1.0, float(gk.abs().max())), rtol=2e-3)
        n += 1
    assert n >= 30, n


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("packed", [True, False])
@pytest.mark.parametrize("p_drop", [0.5, 0.3])
def test_dedup_att_embed_equals_the_replicated_product(golden, dtype, packed, p_drop):
    """Full-GC under dropout (independent per-sentence masks): relu(att_embed(x)) once per NODE row + a masked gather per copy
    (functions.Prepared, dedup) against the product over the replicated rows themselves -- same Philox masks, same loss, same gradients.
    The forward is the SAME arithmetic in both forms, also under bf16 storage and a keep scale that is not a power of two (p = 0.3:
    1 / 0.7): the masked copy is rounded once from the fp32 product, like the replicated product's epilogue -- equal loss."""
    g = golden("fullgc_train")
    res = {}
    for dd in (1, 0):
        m = build(g, g.group("weights"), True, compute_dtype=dtype, drop_prob_lm=p_drop, dedup_att_embed=dd)
        assert m.dedup_att_embed == bool(dd)
        m.packed_decoder = packed
        m._dropout_calls = 0
        batch = _batch(g, 4, 21, True)
        out, loss = run_train(m, batch)
        res[dd] = (float(out["lang_loss"]), {k: p.grad.clone().cpu() for k, p in m.named_parameters()})
    (l1, g1), (l0, g0) = res[1], res[0]
    assert abs(l1 - l0) < 2e-5 * max(1.0, abs(l0))
    top = max(float(v.abs().max()) for v in g0.values())
    n = 0
    for k in g0:
        sc = float(g0[k].abs().max())
        if sc < 1e-5 * top or k == "pred_emb_prj.bias" or k.endswith(("fc_rgt.bias", "fc_lft.bias")):
            continue                                              # a constant shift in front of a BatchNorm: true gradient zero, rounding noise
        if dtype == "bf16":
            a, b = g1[k].double().flatten(), g0[k].double().flatten()
            assert float((a @ b) / (a.norm() * b.norm() + 1e-30)) > 0.99, k
        else:
            np.testing.assert_allclose(g1[k].numpy(), g0[k].numpy(), atol=3e-5 * sc + 1e-9, rtol=2e-3, err_msg=k)
        n += 1
    assert n >= 30
