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
