"""The loss-only fast path (length-sorted packed decoder, functions_packed.py) must give the SAME loss and
the SAME gradients as the full path that also produces `outputs` (functions.DecoderFn), including when
the reference's early break triggers and when a batch contains very short and very long sentences."""
import argparse

import numpy as np
import pytest
import torch

from subgc import synthetic
import subgc.models as models
from subgc.functions_packed import live_plan

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

OPT = dict(caption_model="topdown", vocab_size=300, input_encoding_size=128, rnn_size=128, num_layers=1, drop_prob_lm=0.0,
           max_length=20, seq_length=16, fc_feat_size=96, att_feat_size=256, att_hid_size=64, use_bn=0, sampling_prob=0.0,
           use_gpn=1, embed_dim=20, gcn_dim=128, noun_fuse=1, pred_emb_type=1, gcn_layers=2, gcn_residual=2, gcn_bn=0,
           gpn_drop_prob=0.0, obj_name_path=None, rel_name_path=None, sg_obj_cnt=60, sg_pred_cnt=21)


def grads_of(model, batch, packed):
    model.packed_decoder = packed
    lw = models.LossWrapper(model, None)
    b = {k: v.to(DEV) for k, v in batch.items()}
    model.flatten_grads()
    out = lw(b["fc_feats"], b["att_feats"], b["labels"], b["masks"], b["att_masks"], None, None, None, b["obj_dist"], None, b["rel_ind"],
             None, b["pred_dist"], b["gpn_obj_ind"], b["gpn_pred_ind"], b["gpn_nrel_ind"], b["gpn_pool_mtx"])
    (out["lang_loss"] + out["gpn_loss"]).backward()
    torch.cuda.synchronize()
    return float(out["lang_loss"]), model.flat_grads.clone()


@pytest.mark.parametrize("min_len,max_len", [(5, 16), (1, 16), (2, 6)])
def test_packed_equals_unpacked(min_len, max_len):
    torch.manual_seed(0)
    m = models.setup(argparse.Namespace(**OPT))
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "gcn_collect" in n and "weight" in n:
                p.mul_(30.0)
    m = m.to(DEV).train()
    batch = synthetic.make_train_batch(6, D=256, vocab=300, n_obj_cls=60, seed=5, fc_size=256, min_len=min_len, max_len=max_len)
    l0, g0 = grads_of(m, batch, packed=False)
    l1, g1 = grads_of(m, batch, packed=True)
    assert abs(l0 - l1) < 2e-5 * max(1.0, abs(l0))
    scale = float(g0.abs().max())
    np.testing.assert_allclose(g1.cpu().numpy(), g0.cpu().numpy(), atol=2e-5 * scale + 1e-7, rtol=2e-4)


def test_live_plan_counts_and_early_break():
    labels = torch.zeros(5, 8, dtype=torch.long)
    for s, n in enumerate([3, 1, 5, 2, 5]):
        labels[s, 1:n + 1] = 7
    mask = (torch.arange(8).view(1, -1) < torch.tensor([5, 3, 7, 4, 7]).view(-1, 1)).float()   # len + 2 ones
    perm, M, den = live_plan(labels.to(DEV), mask[:, 1:].to(DEV))
    # step t is live for sentence s while t <= len_s; the longest sentences have 5 tokens -> steps 0..5, then all labels are
    # zero from t = 6 on (early break at t = 6)
    assert M == [5, 5, 4, 3, 2, 2, 0]
    assert float(den) == float(mask[:, 1:].sum())
    assert sorted(perm.tolist()[:2]) == [2, 4]


@pytest.mark.parametrize("kind", ["subgc_f32", "subgc_bf16", "fullgc_bf16_shared", "fullgc_f32_dropout"])
@pytest.mark.parametrize("packed", [True, False])
def test_recurrence_issued_from_c_is_the_step_by_step_loop(kind, packed):
    """subgc_recurrence_fwd / subgc_recurrence_bwd (one library crossing per direction) against the same T steps issued one entry
    point at a time from Python (ops.RECURRENCE_IN_C = False): same kernels, same launch order, same arguments -- the loss and every
    gradient of the flat bucket are BIT-identical wherever the kernels are (the split-K planes are summed in a fixed order; the only
    atomics of the step, embed_bwd / pool_bwd / scatter_add, sit outside the loop and give the usual last-bit noise)."""
    from subgc import ops
    torch.manual_seed(0)
    opt = dict(OPT)
    if kind.startswith("fullgc"):
        opt.update(use_gpn=0, noun_fuse=0, pred_emb_type=2, gcn_layers=4, gcn_residual=1, gcn_bn=1)
    if "bf16" in kind:
        opt.update(compute_dtype="bf16")
    if "dropout" in kind:
        opt.update(drop_prob_lm=0.5)
    m = models.setup(argparse.Namespace(**opt)).to(DEV).train()
    batch = synthetic.make_train_batch(6, D=256, vocab=300, n_obj_cls=60, seed=9, fc_size=256, min_len=2, max_len=16)
    res = {}
    for in_c in (False, True):
        ops.RECURRENCE_IN_C = in_c
        try:
            m._dropout_calls = 0                                   # the same Philox stream for both runs
            m.packed_decoder = packed
            lw = models.LossWrapper(m, None)
            b = {k: v.to(DEV) for k, v in batch.items()}
            m.flatten_grads()
            out = lw(b["fc_feats"], b["att_feats"], b["labels"], b["masks"], b["att_masks"], None, None, None, b["obj_dist"], None, b["rel_ind"],
                     None, b["pred_dist"], b["gpn_obj_ind"], b["gpn_pred_ind"], b["gpn_nrel_ind"], b["gpn_pool_mtx"])
            models.total_loss(out).backward()
            torch.cuda.synchronize()
            res[in_c] = (float(out["lang_loss"]), m.flat_grads.clone())
        finally:
            ops.RECURRENCE_IN_C = True
    (l0, g0), (l1, g1) = res[False], res[True]
    assert l0 == l1
    lo, hi = next((lo, hi) for st, lo, hi in m.grad_buckets() if st == "recurrent")
    emb_o, emb_n, _ = m._slots["embed.0.weight"]
    same = torch.ones_like(g0, dtype=torch.bool)
    same[emb_o:emb_o + emb_n] = False                              # embed_bwd accumulates with fp32 atomics
    assert torch.equal(g0[lo:hi][same[lo:hi]], g1[lo:hi][same[lo:hi]]), "the recurrent slice must be bit-identical"
    scale = float(g0.abs().max())
    np.testing.assert_allclose(g1.cpu().numpy(), g0.cpu().numpy(), atol=1e-6 * scale + 1e-9, rtol=1e-5)


@pytest.mark.parametrize("kind", ["subgc_f32", "fullgc_f32_bn", "subgc_bf16", "fullgc_bf16_bn"])
def test_paired_gcn_units_equal_the_unit_by_unit_path(kind):
    """functions.UnitPairFn (concatenated fc_lft: one N = 2 * 512 product, one K = 2 * 512 data-gradient product, one weight-gradient
    product per pair) against the two units run one by one (`pair_gcn_units = 0`: graph_conv_unit.py:28-36 as four Linear launches per
    pair): same loss and the same gradient for every parameter -- fp32 up to the summation order of the wider products, bf16 up to the
    rounding of the differently accumulated bf16 intermediates."""
    torch.manual_seed(0)
    opt = dict(OPT)
    if kind.startswith("fullgc"):
        opt.update(use_gpn=0, noun_fuse=0, pred_emb_type=2, gcn_layers=4, gcn_residual=1, gcn_bn=1)
    bf = "bf16" in kind
    if bf:
        opt.update(compute_dtype="bf16")
    batch = synthetic.make_train_batch(6, D=256, vocab=300, n_obj_cls=60, seed=4, fc_size=256, min_len=2, max_len=16)
    res, sd = {}, None
    for pair in (0, 1):
        torch.manual_seed(0)
        m = models.setup(argparse.Namespace(**dict(opt, pair_gcn_units=pair)))
        if sd is None:
            sd = {k: v.clone() for k, v in m.state_dict().items()}
            for k, v in sd.items():
                if "gcn_collect" in k and "weight" in k and "bn" not in k:
                    v.mul_(30.0)
        m.load_state_dict(sd)
        m = m.to(DEV).train()
        assert bool(m.pair_gcn_units) == bool(pair)
        m.packed_decoder = True
        lw = models.LossWrapper(m, None)
        b = {k: v.to(DEV) for k, v in batch.items()}
        m.flatten_grads()
        out = lw(b["fc_feats"], b["att_feats"], b["labels"], b["masks"], b["att_masks"], None, None, None, b["obj_dist"], None, b["rel_ind"],
                 None, b["pred_dist"], b["gpn_obj_ind"], b["gpn_pred_ind"], b["gpn_nrel_ind"], b["gpn_pool_mtx"])
        models.total_loss(out).backward()
        torch.cuda.synchronize()
        loss = float(out["lang_loss"])
        res[pair] = (loss, {k: p.grad.clone() for k, p in m.named_parameters()})
    (l0, g0), (l1, g1) = res[0], res[1]
    assert abs(l0 - l1) < (2e-2 if bf else 2e-5) * max(1.0, abs(l0))
    for k in g0:
        a, b = g0[k].double().flatten(), g1[k].double().flatten()
        scale = float(a.abs().max())
        if scale == 0.0:
            assert float(b.abs().max()) == 0.0, k                        # dead parameters stay dead
            continue
        if bf:
            cos = float((a @ b) / (a.norm() * b.norm() + 1e-30))
            assert cos > 0.99, (k, cos)
        else:
            np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), atol=3e-5 * scale + 1e-9, rtol=3e-4, err_msg=k)


def _with_lengths(batch, lengths, vocab=300):
    """The batch with caption j cut / extended to lengths[j] words (labels [.., 0, w1..wn, 0..], masks = n + 2 ones, as the loader builds them)."""
    b = {k: v.clone() for k, v in batch.items()}
    g = torch.Generator().manual_seed(11)
    T2 = b["labels"].size(1)
    b["labels"].zero_()
    b["masks"].zero_()
    for j, n in enumerate(lengths):
        b["labels"][j, 1:n + 1] = torch.randint(1, vocab + 1, (n,), generator=g)
        b["masks"][j, :n + 2] = 1
    assert T2 >= max(lengths) + 2
    return b


@pytest.mark.parametrize("case", ["one_image", "one_empty_caption", "all_empty", "all_full_length", "one_word_each"])
@pytest.mark.parametrize("packed", [True, False])
def test_degenerate_caption_lengths_match_the_oracle(case, packed):
    """Edge cases of the teacher-forced loop (models/AttModel.py:157-175: the early break on an all-zero label column; the criterion's mask
    of len + 2 ones): a single image, a caption without words next to full ones, a batch of empty captions (the loop breaks at step 1),
    captions that use every step, one word each -- loss and gradients against the CPU oracle, packed and unpacked decoder."""
    from oracle import subgc_oracle as O
    torch.manual_seed(0)
    opt = argparse.Namespace(**OPT)
    m = models.setup(opt)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.to(DEV).train()
    B = 1 if case == "one_image" else 3
    batch = synthetic.make_train_batch(B, D=256, vocab=300, n_obj_cls=60, seed=7, fc_size=256, min_len=3, max_len=12)
    batch = {k: torch.as_tensor(v) for k, v in batch.items()}
    S = batch["labels"].size(0)
    lengths = {"one_image": [4, 9, 1, 16, 7][:S], "one_empty_caption": [0] + [5 + (j % 7) for j in range(S - 1)], "all_empty": [0] * S,
               "all_full_length": [16] * S, "one_word_each": [1] * S}[case]
    batch = _with_lengths(batch, lengths)
    loss, grads = grads_of(m, batch, packed)
    orc = O.Oracle(opt, sd, requires_grad=True)
    orc.training = True
    ref = O.loss_wrapper(orc, batch)
    (ref["lang_loss"] + ref["gpn_loss"]).backward()
    assert abs(loss - float(ref["lang_loss"])) < 1e-4 * max(1.0, abs(float(ref["lang_loss"])))
    for k, p in m.named_parameters():
        want = orc.P[k].grad
        if want is None:
            assert float(p.grad.abs().max()) == 0.0, k
            continue
        scale = float(want.abs().max())
        np.testing.assert_allclose(p.grad.cpu().numpy(), want.numpy(), atol=2e-4 * max(scale, 1e-3) + 2e-6, rtol=5e-3, err_msg=f"{case} {k}")


@pytest.mark.parametrize("kind", ["subgc_f32", "subgc_bf16", "fullgc_f32_dropout", "fullgc_bf16_dropout"])
@pytest.mark.parametrize("packed", [True, False])
@pytest.mark.parametrize("in_c", [True, False])
def test_deferred_du_equals_the_per_step_accumulation(kind, packed, in_c):
    """functions.DEFER_DU (subgc_attn_bwd_planes_de + one subgc_attn_du_accum after the BPTT loop) against d(u) read-modify-written at
    every step: the same sums in a different order -- loss identical, every gradient equal to fp32 rounding (ctx2att / att_embed and
    everything upstream of them are what d(u) feeds).  Per-sentence sets only (Sub-GC, and Full-GC under dropout = replicated rows)."""
    from subgc import ops
    from subgc import functions as F_
    torch.manual_seed(0)
    opt = dict(OPT)
    if kind.startswith("fullgc"):
        opt.update(use_gpn=0, noun_fuse=0, pred_emb_type=2, gcn_layers=4, gcn_residual=1, gcn_bn=1)
    if "bf16" in kind:
        opt.update(compute_dtype="bf16")
    if "dropout" in kind:
        opt.update(drop_prob_lm=0.5)
    m = models.setup(argparse.Namespace(**opt)).to(DEV).train()
    batch = synthetic.make_train_batch(6, D=256, vocab=300, n_obj_cls=60, seed=9, fc_size=256, min_len=1, max_len=16)
    res = {}
    for defer in (False, True):
        F_.DEFER_DU, ops.RECURRENCE_IN_C = defer, in_c
        try:
            m._dropout_calls = 0
            res[defer] = grads_of(m, batch, packed) if m.gpn else _grads_fullgc(m, batch, packed)
        finally:
            F_.DEFER_DU, ops.RECURRENCE_IN_C = True, True
    (l0, g0), (l1, g1) = res[False], res[True]
    assert l0 == l1
    scale = float(g0.abs().max())
    if "bf16" in kind:
        a, b = g1.double(), g0.double()
        assert float((a @ b) / (a.norm() * b.norm())) > 0.9999
        np.testing.assert_allclose(g1.cpu().numpy(), g0.cpu().numpy(), atol=2e-3 * scale)
    else:
        np.testing.assert_allclose(g1.cpu().numpy(), g0.cpu().numpy(), atol=2e-6 * scale + 1e-9, rtol=1e-4)
    lo, hi = next((lo, hi) for st, lo, hi in m.grad_buckets() if st == "prepare")
    assert float(g1[lo:hi].abs().max()) > 0                                # ctx2att / att_embed gradients are there at all


def _grads_fullgc(model, batch, packed):
    model.packed_decoder = packed
    lw = models.LossWrapper(model, None)
    b = {k: v.to(DEV) for k, v in batch.items()}
    model.flatten_grads()
    out = lw(b["fc_feats"], b["att_feats"], b["labels"], b["masks"], b["att_masks"], None, None, None, b["obj_dist"], None, b["rel_ind"],
             None, b["pred_dist"], b["gpn_obj_ind"], b["gpn_pred_ind"], b["gpn_nrel_ind"], b["gpn_pool_mtx"])
    out["lang_loss"].backward()
    torch.cuda.synchronize()
    return float(out["lang_loss"]), model.flat_grads.clone()
