"""The HIP path pinned on the reference's INTERMEDIATES: fusion output, every live GCN layer output, the
sGPN read-out, the selected node rows / read-out projection, the prepared features (p_fc, p_att, pp_att) and every decoder step's
h / c / alpha / ctx / log-probs, against tests/golden/*_out.npz (written by running the reference, make_golden.py:83-139).
A kernel regression then shows up as "gcn_x_layer1 wrong", not as "gradient of X off by 3e-3".

`model.tap = {}` is the debug hook (AttModel.__init__): no cost when None; with it the train forward runs the unpacked decoder and
the decode the eager token loop -- the same kernels the packed / graph-replayed paths launch (tests/test_packed_gpu.py,
tests/test_decode_plumbing_gpu.py pin those against these)."""
import numpy as np
import pytest
import torch

from subgc import synthetic

pytestmark = pytest.mark.gpu
ATOL = RTOL = 1e-4


def _model(g, weights, **over):
    import subgc.models as models
    m = models.setup(g.opt(caption_model="topdown", gpn_drop_prob=0.0, **over))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})
    return m.to("cuda:0")


def _cmp(tap, ref, key, steps=None, cols=None):
    got = tap[key].float().cpu().numpy()
    want = ref[key]
    if steps is not None:
        got = got[:steps]
    if cols is not None:
        got = got[..., :cols]
    assert got.shape == want.shape, (key, got.shape, want.shape)
    np.testing.assert_allclose(got, want, atol=ATOL, rtol=RTOL, err_msg=key)


def _post_residual(ref, layers, residual):
    """The golden files hold each GCN layer's output as the layer RETURNED it (forward hook); the residual of gcn_backbone.py:41-46 is added
    after the hook, and the HIP aggregation kernels add it in the same launch -- the product's layer outputs are the post-residual ones."""
    ref = dict(ref)
    skip_x, skip_p = ref["fusion_x"], ref["fusion_p"]
    for l in range(layers):
        x, p = ref[f"gcn_x_layer{l}"], ref[f"gcn_p_layer{l}"]
        if (l + 1) % residual == 0:
            x, p = x + skip_x, p + skip_p
            skip_x, skip_p = x, p
        ref[f"gcn_x_layer{l}"], ref[f"gcn_p_layer{l}"] = x, p
    np.testing.assert_allclose(ref[f"gcn_x_layer{layers - 1}"], ref["x_obj_out"], atol=1e-6)      # the convention, checked on the goldens themselves
    return ref


def _compare_all(tap, ref, layers, expect, residual):
    ref = _post_residual(ref, layers, residual)
    seen = []
    for k in ("fusion_x", "fusion_p", "x_obj_out", "read_out", "att_sel", "fc_sel", "p_fc", "p_att", "pp_att"):
        if k in tap and k in ref:
            _cmp(tap, ref, k); seen.append(k)
    for l in range(layers):
        for k in (f"gcn_x_layer{l}", f"gcn_p_layer{l}"):
            if k in tap:                                   # dead outputs are not computed (SURVEY 8a note): absent from the tap
                _cmp(tap, ref, k); seen.append(k)
    steps = ref["step_h_att"].shape[0]
    n_max = ref["step_alpha"].shape[2]
    for k in ("h_att", "c_att", "h_lang", "c_lang", "ctx", "logp"):
        if "step_" + k in tap:
            _cmp(tap, ref, "step_" + k, steps=steps); seen.append("step_" + k)
    _cmp(tap, ref, "step_alpha", steps=steps, cols=n_max); seen.append("step_alpha")
    # the attention arg-max feeds the grounding output: exact
    np.testing.assert_array_equal(tap["step_alpha"][:steps, :, :n_max].argmax(2).cpu().numpy(), ref["step_alpha"].argmax(2))
    missing = [k for k in expect if k not in seen]
    assert not missing, missing
    return seen


STEP_KEYS = ["step_h_att", "step_c_att", "step_h_lang", "step_c_lang", "step_alpha", "step_logp"]


@pytest.mark.parametrize("name", ["subgc_train", "fullgc_train", "subgc_gtsubg_train"])
def test_train_forward_intermediates_match_reference(golden, name):
    g = golden(name)
    m = _model(g, g.group("weights")).train()
    b = {k: v.to("cuda:0") for k, v in g.tensors("inputs").items()}
    m.tap = {}
    outputs, gpn_loss, score = m(*synthetic.forward_args(b))
    tap, m.tap = m.tap, None
    ref = g.group("out")
    np.testing.assert_allclose(outputs.detach().cpu().numpy(), ref["outputs"], atol=ATOL, rtol=RTOL)
    expect = ["fusion_x", "fusion_p", "x_obj_out", "p_fc", "p_att", "pp_att", "step_ctx"] + STEP_KEYS
    layers = g.meta["opt"]["gcn_layers"]
    if name.startswith("subgc"):
        expect += ["read_out", "att_sel", "fc_sel", "gcn_p_layer0", "gcn_x_layer1"]         # the live half of the two-layer GCN
        sel_mask = (torch.arange(ref["mask_sel"].shape[1]).view(1, -1) < tap["sel_lens"].cpu().view(-1, 1)).float().numpy()
        np.testing.assert_array_equal(sel_mask, ref["mask_sel"])
    else:
        expect += [f"gcn_x_layer{l}" for l in range(layers)] + [f"gcn_p_layer{l}" for l in range(layers - 1)]
    _compare_all(tap, ref, layers, expect, g.meta["opt"]["gcn_residual"])
    # a second call without the tap takes the product path again and files nothing
    m(*synthetic.forward_args(b))
    assert m.tap is None


@pytest.mark.parametrize("name,weights", [("subgc_greedy", "subgc_train"), ("fullgc_greedy", "fullgc_train")])
def test_greedy_decode_intermediates_match_reference(golden, name, weights):
    g = golden(name)
    m = _model(g, golden(weights).group("weights")).eval()
    b = {k: v.to("cuda:0") for k, v in g.tensors("inputs").items()}
    m.tap = {}
    ret = m(*synthetic.sample_args(b), opt=dict(g.meta["sample_opt"]), mode="sample")
    tap, m.tap = m.tap, None
    ref = g.group("out")
    np.testing.assert_array_equal(ret[0].cpu().numpy(), ref["seq"])
    np.testing.assert_array_equal(ret[3].cpu().numpy(), ref["keep_ind"])
    expect = ["fusion_x", "x_obj_out", "p_fc", "p_att", "pp_att"] + STEP_KEYS
    if name.startswith("subgc"):
        expect += ["read_out", "att_sel", "fc_sel"]
        np.testing.assert_array_equal(tap["keep_ind"].cpu().numpy(), ref["keep_ind"])
        np.testing.assert_allclose(tap["subgraph_score_raw"].cpu().numpy(), ref["subgraph_score_raw"], atol=1e-5)
        # the reference pools all 5 identical counterparts ([2, 5, M, 2L], gpn.py:86-96 then reads counterpart 0); the product pools that one
        M = g.meta["M"]
        ref["read_out"] = ref["read_out"].reshape(2, 5, M, -1)[:, 0].reshape(2 * M, -1)
    _compare_all(tap, ref, g.meta["opt"]["gcn_layers"], expect, g.meta["opt"]["gcn_residual"])
    # the graph-replayed product path (no tap) returns the same tokens / log-probs as the tapped eager loop just did
    again = m(*synthetic.sample_args(b), opt=dict(g.meta["sample_opt"]), mode="sample")
    np.testing.assert_array_equal(again[0].cpu().numpy(), ret[0].cpu().numpy())
    np.testing.assert_allclose(again[1].cpu().numpy(), ret[1].cpu().numpy(), atol=1e-5)
