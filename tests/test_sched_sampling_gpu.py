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
