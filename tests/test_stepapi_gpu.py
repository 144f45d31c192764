"""The reference's step-level methods (`_prepare_feature`, `get_logprobs_state`, `beam_search`, AttModel.py:328-368,
CaptionModel.py:28-176) on the HIP path: driven exactly like the reference's `_sample_sentences` loop (:203-231), one
sub-graph at a time with a caller-owned state, they must reproduce the reference's own beams (tests/golden)."""
import numpy as np
import pytest
import torch

from subgc import synthetic
from subgc.models import sampling
from test_parity_gpu import DEV, build, close

pytestmark = pytest.mark.gpu


def dense_subgraphs(m, b):
    """(fc [n, 2L], att [n, N, L], masks [n, N], keep) of the kept sub-graphs, in the reference's dense layout."""
    args = synthetic.sample_args(b)
    att_feats, obj_dist, rel_ind, pred_dist = args[1], args[4], args[6], args[8]
    B, N, _ = att_feats.shape
    X2 = m._encode(att_feats, obj_dist, pred_dist, rel_ind).reshape(B * N, m.GCN_dim).contiguous()
    sel = sampling.select_subgraphs(m, X2, N, [(0, b["gpn_obj_ind"], b["att_masks"], b["gpn_pool_mtx"])])[0]
    att = X2[sel["idx"].long()]                                       # image 0: node index == row of X2
    masks = (torch.arange(N, device=DEV).view(1, N) < sel["lens"].view(-1, 1)).float()
    return sel["fc"], att * masks.unsqueeze(-1), masks, sel["keep"]


@pytest.mark.parametrize("name", ["subgc_beam3", "subgc_beam4_div"])
def test_reference_style_beam_loop_through_step_api(golden, name):
    g = golden(name)
    m = build(g, golden("subgc_beam").group("weights"), False)
    b = {k: v.to(DEV) for k, v in g.tensors("inputs").items()}
    ref = g.group("out")
    opt = g.meta["sample_opt"]
    beam = opt["beam_size"]
    fc, att, masks, keep = dense_subgraphs(m, b)
    np.testing.assert_array_equal(keep.cpu().numpy(), ref["keep_ind"])
    p_fc, p_att, pp_att, p_masks = m._prepare_feature(fc, att, masks)
    assert p_att.shape[1] == int(masks.sum(1).max()) and float(p_att[p_masks == 0].abs().max()) == 0.0
    for k in range(fc.size(0)):                                       # AttModel.py:216-231
        state = m.init_hidden(beam)
        t_fc = p_fc[k:k + 1].expand(beam, -1)
        t_att = p_att[k:k + 1].expand(beam, -1, -1).contiguous()
        t_pp = pp_att[k:k + 1].expand(beam, -1, -1).contiguous()
        t_m = p_masks[k:k + 1].expand(beam, -1).contiguous()
        it = torch.zeros(beam, dtype=torch.long, device=DEV)
        logprobs, state = m.get_logprobs_state(it, t_fc, t_att, t_pp, t_m, state)
        assert float(torch.logsumexp(logprobs, 1).abs().max()) < 1e-4 and state[0].shape == (2, beam, m.rnn_size)
        done = m.beam_search(state, logprobs, t_fc, t_att, t_pp, t_m, None, None, opt=opt)
        np.testing.assert_array_equal(np.stack([d["seq"].numpy() for d in done]), ref["done_seq"][k])
        np.testing.assert_allclose(np.stack([d["logps"].numpy() for d in done]), ref["done_logps"][k], atol=1e-4)
        np.testing.assert_allclose([d["p"] for d in done], ref["done_p"][k], atol=2e-3, rtol=1e-5)


def test_get_logprobs_state_equals_oracle_step(golden):
    from oracle import subgc_oracle as O
    g = golden("subgc_greedy")
    w = golden("subgc_train").group("weights")
    m = build(g, w, False)
    b = {k: v.to(DEV) for k, v in g.tensors("inputs").items()}
    fc, att, masks, _ = dense_subgraphs(m, b)
    p_fc, p_att, pp_att, p_masks = m._prepare_feature(fc, att, masks)
    orc = O.Oracle(g.opt(), w)
    f, v, u, mk = O.prepare_feature(orc.P, orc.cfg, fc.cpu(), att.cpu(), masks.cpu(), False)
    close(p_fc, f, "p_fc"); close(p_att, v, "p_att"); close(pp_att, u, "pp_att")
    n, R = fc.size(0), m.rnn_size
    gen = torch.Generator().manual_seed(3)
    h = torch.randn(2, n, R, generator=gen) * 0.3
    c = torch.randn(2, n, R, generator=gen) * 0.3
    it = torch.randint(0, 50, (n,), generator=gen)
    want_lp, ((h1, h2), (c1, c2)), want_w = O.core_step(orc.P, orc.cfg, it, f, v, u, mk, ((h[0], h[1]), (c[0], c[1])), False)
    lp, (nh, nc), alpha = m.get_logprobs_state(it.to(DEV), p_fc, p_att, pp_att, p_masks, (h.to(DEV), c.to(DEV)), return_att=True)
    close(lp, want_lp, "logprobs"); close(alpha, want_w, "att weights", atol=1e-5)
    close(nh[0], h1, "h_att"); close(nh[1], h2, "h_lang"); close(nc[0], c1, "c_att"); close(nc[1], c2, "c_lang")
    with pytest.raises(NotImplementedError):
        m.train().get_logprobs_state(it.to(DEV), p_fc, p_att, pp_att, p_masks, (h.to(DEV), c.to(DEV)))
