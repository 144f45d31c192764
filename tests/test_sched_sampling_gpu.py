"""Scheduled sampling (reference AttModel.py:157-167; switched on by train.sh's `--scheduled_sampling_start 0`): the
Philox uniform kernel, the inverse-CDF multinomial kernel, and the model path against the oracle with the same injected
uniforms (the reference draws from torch's RNG stream, which cannot be reproduced elsewhere; the distribution is equal)."""
import numpy as np
import pytest
import torch

from oracle import subgc_oracle as O
from subgc import ops
import subgc.models as models
from test_parity_gpu import DEV, build, close, run_train

pytestmark = pytest.mark.gpu


def test_uniform_kernel_is_a_counter_based_stream():
    a = ops.uniform((1000003,), 1234, 0, DEV)
    b = ops.uniform((1000003,), 1234, 0, DEV)
    c = ops.uniform((1000,), 1234, 1000, DEV)
    d = ops.uniform((1000,), 1235, 0, DEV)
    assert torch.equal(a, b) and torch.equal(a[1000:2000], c) and not torch.equal(a[:1000], d)
    assert float(a.min()) >= 0.0 and float(a.max()) < 1.0
    assert abs(float(a.mean()) - 0.5) < 2e-3 and abs(float(a.var()) - 1 / 12) < 2e-3
    hist = torch.histc(a, bins=16, min=0, max=1) / a.numel()
    assert float((hist - 1 / 16).abs().max()) < 2e-3


@pytest.mark.parametrize("rows,V,stride", [(640, 9488, 17), (33, 50, 1), (7, 1000, 3)])
def test_multinomial_rows_is_the_inverse_cdf_in_index_order(rows, V, stride):
    g = torch.Generator().manual_seed(rows + V)
    logits = (torch.randn(rows, V + 5, generator=g) * 3).to(DEV)[:, :V]               # strided rows
    u = torch.rand(rows, generator=g).to(DEV)
    u[0], u[1] = 0.0, 0.999999
    sel = torch.rand(rows, generator=g).to(DEV)
    base = torch.randint(0, V, (rows, stride), generator=g).to(DEV)
    tok = base.clone()
    ops.multinomial_rows_(logits, u, sel, 0.6, tok[:, stride - 1])
    p = torch.softmax(logits.double(), 1)
    cdf = p.cumsum(1)
    want = (cdf <= (u.double() * cdf[:, -1]).unsqueeze(1)).sum(1).clamp(max=V - 1)
    chosen = sel < 0.6
    got = tok[:, stride - 1]
    assert torch.equal(got[~chosen], base[:, stride - 1][~chosen]) and int(chosen.sum()) > 0
    if stride > 1:
        assert torch.equal(tok[:, : stride - 1], base[:, : stride - 1])
    # a draw may legitimately differ by one position where u*Z falls within fp32 rounding of a cdf step
    diff = (got[chosen] - want[chosen]).abs()
    assert int((diff > 1).sum()) == 0 and float((diff == 0).float().mean()) > 0.98
    # and it is a sample of the right distribution: empirical frequencies over many uniforms
    one = logits[:1].expand(20000, V).contiguous()
    uu = torch.rand(20000, generator=g).to(DEV)
    t2 = torch.zeros(20000, dtype=torch.long, device=DEV)
    ops.multinomial_rows_(one, uu, torch.zeros(20000, device=DEV), 1.0, t2)
    top = torch.topk(p[0], 3).indices
    freq = torch.bincount(t2, minlength=V).double() / 20000
    assert float((freq[top] - p[0][top]).abs().max()) < 0.02


def test_model_with_scheduled_sampling_matches_oracle(golden):
    g = golden("subgc_train")
    w = g.group("weights")
    m = build(g, w, True, sampling_prob=0.5)
    assert m.ss_prob == 0.5
    batch = g.tensors("inputs")
    S, T = batch["labels"].shape[0], batch["labels"].shape[1] - 1
    gen = torch.Generator().manual_seed(77)
    sel_u, u = torch.rand(T, S, generator=gen), torch.rand(T, S, generator=gen)
    m.injected_ss = (sel_u.to(DEV), u.to(DEV))
    out, loss = run_train(m, batch)
    orc = O.Oracle(g.opt(gpn_drop_prob=0.0, sampling_prob=0.5), w, requires_grad=True)
    orc.training = True
    ref = O.loss_wrapper(orc, batch, ss=(sel_u, u))
    (ref["lang_loss"] + ref["gpn_loss"]).backward()
    changed = sum(int((tok != batch["labels"][:, i]).sum()) for i, tok in orc.ss_tokens.items())
    assert changed > 10, "the draws should replace a good number of ground-truth words"
    close(out["lang_loss"], ref["lang_loss"], "lang_loss")
    plain = O.loss_wrapper(O.Oracle(g.opt(gpn_drop_prob=0.0), w), batch)
    assert abs(float(plain["lang_loss"]) - float(ref["lang_loss"])) > 1e-3, "scheduled sampling must change the loss"
    m.injected_ss = (sel_u.to(DEV), u.to(DEV))
    outputs, _, _ = m(*__import__("subgc").synthetic.forward_args({k: v.to(DEV) for k, v in batch.items()}))
    close(outputs, ref["outputs"], "outputs", atol=2e-4)
    for k in ("logit.weight", "embed.0.weight", "core.att_lstm.weight_ih", "core.lang_lstm.weight_hh", "ctx2att.weight", "obj_v_proj.weight"):
        gr = orc.P[k].grad
        close(m.P(k).grad, gr, "grad " + k, atol=2e-5 + 2e-3 * float(gr.abs().max()), rtol=5e-3)
    # without injection the model draws its own uniforms and still trains
    m.injected_ss = None
    out2, loss2 = run_train(m, batch)
    assert torch.isfinite(loss2)


@pytest.mark.parametrize("packed", [False, True])
def test_scheduled_sampling_matches_the_reference_golden(golden, packed):
    """The reference's own run of AttModel.py:157-167 with injected selector / draw numbers (golden `subgc_ss_train`):
    outputs, loss and every live gradient of the HIP path, through both decoder Functions."""
    g = golden("subgc_ss_train")
    ref = g.group("out")
    w = golden("subgc_train").group("weights")
    batch = golden("subgc_train").tensors("inputs")
    m = build(g, w, True)
    assert m.ss_prob == 0.25
    m.packed_decoder = packed
    inj = (torch.from_numpy(ref["sel_u"]).to(DEV), torch.from_numpy(ref["u"]).to(DEV))
    m.injected_ss = inj
    out, loss = run_train(m, batch)
    close(out["lang_loss"], ref["lang_loss"], "lang_loss")
    close(out["gpn_loss"], ref["gpn_loss"], "gpn_loss")
    grads, dead = g.group("grads"), set(g.meta["dead_params"])
    for k, p in m.named_parameters():
        if k in dead:
            assert float(p.grad.abs().max()) == 0.0, k
        else:
            close(p.grad, grads[k], "grad " + k, atol=2e-4, rtol=2e-3)
    if not packed:
        m.injected_ss = inj
        outputs, _, _ = m(*__import__("subgc").synthetic.forward_args({k: v.to(DEV) for k, v in batch.items()}))
        close(outputs, ref["outputs"], "outputs")


def test_ss_plan_and_list_multinomial_kernels():
    """subgc_ss_plan: fired[t] = ascending rows r < live[t] with sel[t][r] < prob, nothing at t = 0; subgc_multinomial_rows_list draws for
    exactly those rows from COMPACT logits rows and equals the all-rows kernel (subgc_multinomial_rows) on the same logits."""
    g = torch.Generator().manual_seed(7)
    T, S, V, prob = 9, 700, 333, 0.3
    sel = torch.rand(T, S, generator=g).to(DEV)
    live = torch.tensor([700, 700, 650, 400, 257, 256, 64, 1, 0], dtype=torch.int32).to(DEV)
    fired, cnt = ops.ss_plan(sel, live, prob)
    cnt_h, fired_h, sel_h, live_h = cnt.cpu(), fired.cpu(), sel.cpu(), live.cpu()
    assert int(cnt_h[0]) == 0 and int(cnt_h[8]) == 0
    for t in range(1, T):
        want = [r for r in range(int(live_h[t])) if float(sel_h[t, r]) < prob]
        assert fired_h[t, :int(cnt_h[t])].tolist() == want, t
    t = 2
    m = int(live_h[t])
    logits = (torch.randn(m, V, generator=g) * 3).to(DEV)
    u = torch.rand(S, generator=g).to(DEV)
    base = torch.randint(0, V, (m,), generator=g).to(DEV)
    a, b = base.clone(), base.clone()
    ops.multinomial_rows_(logits, u[:m].contiguous(), sel[t][:m].contiguous(), prob, a)
    compact = logits[fired[t][:int(cnt_h[t])].long()].contiguous()
    pad = torch.zeros(m - compact.size(0), V, device=DEV)
    ops.multinomial_rows_list_(torch.cat([compact, pad]), fired[t], cnt[t:t + 1], u, b)
    assert torch.equal(a, b) and not torch.equal(a, base)


@pytest.mark.parametrize("drop", [0.0])          # (with dropout the two forms assign the random x_t keep-mask rows differently: both valid, not comparable)
def test_fired_rows_side_stream_form_equals_the_in_line_form(drop):
    """The packed decoder's two scheduled-sampling forms on the same Philox stream: draw chain on the fired rows only, on a side stream
    beside the recurrent product (default) vs. every live row in line (functions_packed.SS_FIRED_ROWS_ONLY = False).  Same words fed,
    same loss and gradients up to the rounding of the differently tiled logit products."""
    import argparse
    from subgc import functions_packed as FP, synthetic
    from test_packed_gpu import OPT
    torch.manual_seed(1)
    m = models.setup(argparse.Namespace(**dict(OPT, sampling_prob=0.35, drop_prob_lm=drop))).to(DEV).train()
    batch = synthetic.make_train_batch(24, D=256, vocab=300, n_obj_cls=60, seed=3, fc_size=256, min_len=3, max_len=16)
    res = {}
    for fast in (False, True):
        FP.SS_FIRED_ROWS_ONLY = fast
        try:
            m._dropout_calls = 0
            lw = models.LossWrapper(m, None)
            b = {k: v.to(DEV) for k, v in batch.items()}
            m.flatten_grads()
            out = lw(b["fc_feats"], b["att_feats"], b["labels"], b["masks"], b["att_masks"], None, None, None, b["obj_dist"], None, b["rel_ind"],
                     None, b["pred_dist"], b["gpn_obj_ind"], b["gpn_pred_ind"], b["gpn_nrel_ind"], b["gpn_pool_mtx"])
            models.total_loss(out).backward()
            torch.cuda.synchronize()
            res[fast] = (float(out["lang_loss"]), m.flat_grads.clone())
        finally:
            FP.SS_FIRED_ROWS_ONLY = False
    (l0, g0), (l1, g1) = res[False], res[True]
    assert abs(l0 - l1) < 1e-5 * max(1.0, abs(l0)), (l0, l1)
    scale = float(g0.abs().max())
    np.testing.assert_allclose(g1.cpu().numpy(), g0.cpu().numpy(), atol=2e-5 * scale + 1e-8, rtol=2e-4)
