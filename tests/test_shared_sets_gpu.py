"""Shared attention sets of the Full-GC model (functions.PreparedShared + csrc/attention_group.hip): every sentence of an image attends
over the same node rows (reference AttModel.py:140-149 on the x5 replicated features of gcn_backbone.py:50-51), so att_embed /
ctx2att run once per image and one workgroup serves the image's sentences.  With dropout off the result must equal the replicated
(per-sentence) path: loss, log-probabilities and every gradient; the golden Full-GC cases run through it as well (test_parity_gpu /
test_bf16_storage_gpu build their models with the default share_attention_sets = 1)."""
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


def test_shared_mode_is_what_the_full_gc_train_path_runs(golden):
    """The default Full-GC training forward builds PreparedShared (and injected keep-masks switch it off: those are per sentence)."""
    from subgc import functions as F_
    g = golden("fullgc_train")
    m = build(g, g.group("weights"), True, drop_prob_lm=0.0)
    seen = []
    orig = F_.make_prepared

    def spy(meta, *a, **k):
        pr = orig(meta, *a, **k)
        seen.append(type(pr).__name__)
        return pr

    F_.make_prepared = spy
    try:
        run_train(m, _batch(g, 2, 3, False))
        m.share_attention_sets = False
        run_train(m, _batch(g, 2, 3, False))
    finally:
        F_.make_prepared = orig
    assert seen == ["PreparedShared", "Prepared"]
