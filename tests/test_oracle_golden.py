"""Pin the CPU oracle (oracle/subgc_oracle.py) against vectors produced by the reference itself.

The golden files were written by tests/golden/make_golden.py, which imports and runs
/root/reference (CPU fp32).  Tolerances: floats atol=rtol=2e-5 here (same library, same op
order up to the restatement), indices exact.
"""
import numpy as np
import pytest
import torch

from oracle import subgc_oracle as O
from subgc import synthetic

ATOL = RTOL = 2e-5


def close(a, b, name, atol=ATOL, rtol=RTOL):
    a = a.detach().numpy() if torch.is_tensor(a) else np.asarray(a)
    np.testing.assert_allclose(a, b, atol=atol, rtol=rtol, err_msg=name)


def run_train(G, name, weights_from=None):
    g = G(name)
    w = G(weights_from or name).group("weights")
    batch = g.tensors("inputs")
    orc = O.Oracle(g.opt(gpn_drop_prob=0.0), w, requires_grad=True)
    orc.training = True
    tap = {}
    out = O.loss_wrapper(orc, batch, tap=tap)
    loss = out["lang_loss"] + (out["gpn_loss"] if out["gpn_loss"] is not None else 0.0)
    loss.backward()
    return g, orc, out, tap, loss


@pytest.mark.parametrize("name", ["subgc_train", "subgc_gtsubg_train", "fullgc_train"])
def test_train_forward_intermediates_and_grads(golden, name):
    g, orc, out, tap, loss = run_train(golden, name)
    ref = g.group("out")
    close(out["outputs"], ref["outputs"], "outputs")
    close(out["lang_loss"], ref["lang_loss"], "lang_loss")
    close(loss, ref["loss"], "loss")
    if "gpn_loss" in ref:
        close(out["gpn_loss"], ref["gpn_loss"], "gpn_loss")
        close(out["subgraph_score"], ref["subgraph_score"], "score")
    for k in ("fusion_x", "fusion_p", "x_obj_out", "read_out", "att_sel", "fc_sel", "mask_sel", "p_fc", "p_att", "pp_att"):
        if k in ref and k in tap:
            close(tap[k], ref[k], k)
    for l in range(g.meta["opt"]["gcn_layers"]):
        close(tap[f"gcn_x_layer{l}"], ref[f"gcn_x_layer{l}"], f"gcn_x_layer{l}")
        close(tap[f"gcn_p_layer{l}"], ref[f"gcn_p_layer{l}"], f"gcn_p_layer{l}")
    for k in ("h_att", "c_att", "h_lang", "c_lang", "alpha", "ctx", "logp"):
        close(torch.stack(tap["step_" + k], 0), ref["step_" + k], "step_" + k)
    grads = g.group("grads")
    dead = set(g.meta["dead_params"])
    for k, p in orc.P.items():
        if k in dead:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
        else:
            close(p.grad, grads[k], "grad " + k, atol=5e-5, rtol=1e-4)
    bn_after = g.group("bn_after")
    if bn_after:
        for k, v in bn_after.items():
            if "num_batches" not in k:
                close(orc.buffers[k], v, k)


def sample_case(G, name, weights_from, **kw):
    g = G(name)
    batch = g.tensors("inputs")
    orc = O.Oracle(g.opt(), G(weights_from).group("weights"))
    args = synthetic.sample_args(batch)
    return g, orc.sample(*args, opt=g.meta["sample_opt"], **kw), g.group("out")


@pytest.mark.parametrize("name,wf", [("subgc_greedy", "subgc_train"), ("subgc_greedy_nms55", "subgc_train"),
                                      ("subgc_sct", "subgc_train"), ("fullgc_greedy", "fullgc_train")])
def test_greedy_decode_token_identical(golden, name, wf):
    g, ret, ref = sample_case(golden, name, wf)
    np.testing.assert_array_equal(ret[3].numpy(), ref["keep_ind"])
    np.testing.assert_array_equal(ret[0].numpy(), ref["seq"])
    close(ret[1], ref["seqLogprobs"], "seqLogprobs")
    close(ret[2], ref["subgraph_score"], "score")
    if "att2_weights" in ref:
        close(ret[4], ref["att2_weights"], "att2_weights")
        np.testing.assert_array_equal(ret[4].numpy().argmax(-1), ref["att2_weights"].argmax(-1))


def test_topk_path_pinned(golden):
    """Follow the reference's sampled path; its every token must be in our top-k set with the same
    renormalised log-prob, and the per-step distributions must match."""
    ref = golden("subgc_topk").group("out")
    tap = {}
    g, ret, _ = sample_case(golden, "subgc_topk", "subgc_train", forced=torch.from_numpy(ref["seq"]), tap=tap)
    np.testing.assert_array_equal(ret[3].numpy(), ref["keep_ind"])
    steps = ref["step_logp"].shape[0]
    close(torch.stack(tap["step_logp"][:steps], 0), ref["step_logp"], "step_logp", atol=1e-4)
    seq, lps = ref["seq"], ref["seqLogprobs"]
    alive = np.ones(seq.shape[0], bool)
    for t in range(min(steps, seq.shape[1])):
        idx = tap["topk_idx"][t].numpy(); top = tap["topk_lp"][t].numpy()
        for r in range(seq.shape[0]):
            if alive[r] and seq[r, t] > 0:
                j = np.where(idx[r] == seq[r, t])[0]
                assert len(j) == 1, (r, t)
                assert abs(top[r, j[0]] - lps[r, t]) < 1e-4
        alive &= seq[:, t] > 0


def test_topk_inverse_cdf_sampling_is_within_topk(golden):
    g = golden("subgc_topk")
    orc = O.Oracle(g.opt(), golden("subgc_train").group("weights"))
    tap = {}
    u = torch.rand(10, 20, generator=torch.Generator().manual_seed(0))
    ret = orc.sample(*synthetic.sample_args(g.tensors("inputs")), opt=g.meta["sample_opt"], uniforms=u, tap=tap)
    seq = ret[0].numpy()
    alive = np.ones(seq.shape[0], bool)
    for t in range(len(tap["topk_idx"])):
        idx = tap["topk_idx"][t].numpy()
        for r in range(seq.shape[0]):
            if alive[r] and seq[r, t] > 0:
                assert seq[r, t] in idx[r]
        alive &= seq[:, t] > 0


def test_nms_tie_rule_and_duplicates():
    score = np.array([0.9, 0.5, 0.9, 0.1], np.float32)
    obj = np.full((4, 37), 36); m = np.zeros((4, 37), np.float32)
    for i, nodes in enumerate([[1, 2, 3], [1, 2, 3], [1, 2, 3], [7, 8]]):
        obj[i, :len(nodes)] = nodes; m[i, :len(nodes)] = 1
    keep = O.subgraph_nms(score, obj, m, 0.75, 10, sort_kind="stable")
    np.testing.assert_array_equal(keep, [2, 3])          # tie 0.9/0.9 -> larger index first; dups suppressed
    keep1 = O.subgraph_nms(score, obj, m, 0.75, 1, sort_kind="stable")
    np.testing.assert_array_equal(keep1, [2])


def test_scheduled_sampling_path_matches_the_reference(golden):
    """AttModel.py:157-167 run BY THE REFERENCE with its two random streams replaced by injected numbers
    (tests/golden/make_golden.py ss_case): the oracle, fed the same numbers, must feed the same words at every step and
    reproduce outputs, losses and every gradient."""
    g = golden("subgc_ss_train")
    ref = g.group("out")
    w = golden("subgc_train").group("weights")
    batch = golden("subgc_train").tensors("inputs")
    orc = O.Oracle(g.opt(gpn_drop_prob=0.0), w, requires_grad=True)
    orc.training = True
    assert orc.cfg.ss_prob == 0.25
    out = O.loss_wrapper(orc, batch, ss=(torch.from_numpy(ref["sel_u"]), torch.from_numpy(ref["u"])))
    (out["lang_loss"] + out["gpn_loss"]).backward()
    fed = ref["fed_tokens"]                                       # [steps run, S]: the words the reference actually fed
    labels = batch["labels"].numpy()
    assert (fed != labels[:, :fed.shape[0]].T).sum() == g.meta["changed_words"] > 10
    for i in range(1, fed.shape[0]):
        want = fed[i]
        got = orc.ss_tokens[i].numpy() if i in orc.ss_tokens else labels[:, i]
        np.testing.assert_array_equal(got, want, err_msg=f"words fed at step {i}")
    close(out["outputs"], ref["outputs"], "outputs")
    close(out["lang_loss"], ref["lang_loss"], "lang_loss")
    close(out["gpn_loss"], ref["gpn_loss"], "gpn_loss")
    grads = g.group("grads")
    dead = set(g.meta["dead_params"])
    for k, p in orc.P.items():
        if k in dead:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
        else:
            close(p.grad, grads[k], "grad " + k, atol=5e-5, rtol=1e-4)


def test_loader_restatement_matches_the_reference_getitem(golden):
    """dataloaders/dataloader.py:139-157,225-367 run by the reference on fabricated dataset entries (golden `loader`): the
    oracle's sub-graph sampling (same np.random stream), caption picking (same python `random` stream) and tensor building
    reproduce all 12 model-facing arrays of every image, in the sampled branch and in the `use_gt_subg` branch."""
    import random as pyrandom
    from loader_golden import LoaderCase
    c = LoaderCase(golden)
    m = c.meta
    assert set(m["branches"]) == {"pos_pad", "pos_draw", "neg_plain", "neg_all", "neg_le_thres", "neg_few"}
    for gt, tag in ((0, "smp"), (1, "gt")):
        np.random.seed(m["np_seed"][gt])
        pyrandom.seed(m["py_seed"][gt])
        for b in range(m["n_images"]):
            im = c.image(b)
            if gt:
                ids = c.gt_ids()
            else:
                ids = O.choose_subgraphs(im["iou"], m["thres"], c.hb, c.S)
                np.testing.assert_array_equal(ids, c.out[f"smp{b}_mask_idx"], err_msg=f"image {b}: drawn sub-graph ids")
            nm, pm, nrel = c.chosen(im, ids)
            caps = O.pick_captions(c.raw["label"], c.raw["label_start_ix"], c.raw["label_end_ix"], b, c.S, c.Lq)
            got = O.assemble_image(im["object_fmap"], im["object_dist"], im["rel_ind"], im["pred_dist"], nm, pm, caps, c.obj_num, c.rel_num, nrel)
            for k, want in c.expect(tag, b).items():
                assert got[k].shape == want.shape and got[k].dtype == want.dtype, (tag, b, k, got[k].shape, want.shape, got[k].dtype, want.dtype)
                np.testing.assert_array_equal(got[k], want, err_msg=f"{tag}{b} {k}")
