"""Debug bounds mode (subgc_debug_bounds / ops.debug_bounds): a bad index in a loader tensor fails LOUDLY at the entry point that consumes
it, with the tensor named -- one case per kernel family (scene-graph CSR / aggregation, sGPN candidate lists incl. the reference's
node-list-vs-mask assert gpn.py:117-118, word ids, criterion targets, class ids, decode-time candidate tables) and through the model
API for a train step and a decode call.  With the mode off (default) nothing is checked and valid inputs behave as always."""
import argparse

import pytest
import torch

from subgc import ops, synthetic
import subgc.models as models

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
OPT = dict(caption_model="topdown", vocab_size=60, input_encoding_size=64, rnn_size=64, num_layers=1, drop_prob_lm=0.0, max_length=20, seq_length=16,
           fc_feat_size=48, att_feat_size=128, att_hid_size=32, use_bn=0, sampling_prob=0.0, use_gpn=1, embed_dim=20, gcn_dim=64, noun_fuse=1,
           pred_emb_type=1, gcn_layers=2, gcn_residual=2, gcn_bn=0, gpn_drop_prob=0.0, obj_name_path=None, rel_name_path=None, sg_obj_cnt=40,
           sg_pred_cnt=21)


def test_mode_is_off_by_default_and_restored():
    from subgc._lib import lib
    assert lib().subgc_debug_bounds(0) == 0
    with ops.debug_bounds():
        assert lib().subgc_debug_bounds(1) == 1
        with ops.debug_bounds(False):
            assert lib().subgc_debug_bounds(0) == 0
        assert lib().subgc_debug_bounds(1) == 1
    assert lib().subgc_debug_bounds(0) == 0


def test_scene_graph_family_rejects_a_bad_node_id():
    N, K, B = 9, 12, 3
    rel = torch.randint(0, N, (B, K, 2), generator=torch.Generator().manual_seed(0)).to(DEV)
    with ops.debug_bounds():
        ops.csr_build(rel, N)                                     # valid: passes
        bad = rel.clone(); bad[1, 7, 1] = N
        with pytest.raises(ops.SubgcError, match=r"csr_build: rel_ind.*outside \[0, 8\].*row 19, column 1: 9"):
            ops.csr_build(bad, N)
        neg = rel.clone(); neg[2, 0, 0] = -3
        with pytest.raises(ops.SubgcError, match="rel_ind"):
            ops.csr_build(neg, N)
    ops.csr_build(bad, N)                                         # mode off: the documented clamp, no error
    torch.cuda.synchronize()


def test_word_ids_and_criterion_targets():
    V1, E, n = 30, 16, 40
    table = torch.randn(V1, E, device=DEV)
    tok = torch.randint(0, V1, (n,), device=DEV)
    out = torch.empty(n, E, device=DEV)
    logp = torch.log_softmax(torch.randn(4, 5, V1, device=DEV), -1)
    tgt = torch.randint(0, V1, (4, 6), device=DEV)
    msk = torch.ones(4, 6, device=DEV)
    with ops.debug_bounds():
        ops.embed_fwd(table, tok, 1, None, 1.0, out)
        ops.masked_nll_fwd(logp, tgt[:, 1:], msk[:, 1:])
        bad = tok.clone(); bad[17] = V1
        with pytest.raises(ops.SubgcError, match=r"embed_fwd: tok \(word ids\).*row 17, column 0: 30"):
            ops.embed_fwd(table, bad, 1, None, 1.0, out)
        with pytest.raises(ops.SubgcError, match="embed_bwd: tok"):
            ops.embed_bwd(table, bad, 1, None, 1.0, torch.zeros(n, E, device=DEV), torch.zeros_like(table))
        badt = tgt.clone(); badt[2, 3] = V1 + 4
        with pytest.raises(ops.SubgcError, match=r"masked_nll_fwd: target.*row 2, column 2: 34"):
            ops.masked_nll_fwd(logp, badt[:, 1:], msk[:, 1:])


def _train_batch():
    return {k: v.to(DEV) for k, v in synthetic.make_train_batch(2, D=128, vocab=60, n_obj_cls=40, seed=1, fc_size=128).items()}


def _loss(m, b):
    lw = models.LossWrapper(m, None)
    m.flatten_grads()
    out = lw(b["fc_feats"], b["att_feats"], b["labels"], b["masks"], b["att_masks"], None, None, None, b["obj_dist"], None, b["rel_ind"], None,
             b["pred_dist"], b["gpn_obj_ind"], b["gpn_pred_ind"], b["gpn_nrel_ind"], b["gpn_pool_mtx"])
    return models.total_loss(out)


def test_a_train_step_fails_at_the_consuming_entry_point():
    torch.manual_seed(0)
    m = models.setup(argparse.Namespace(**OPT)).to(DEV).train()
    good = _train_batch()
    with ops.debug_bounds():
        ref = float(_loss(m, good))                               # a valid batch is untouched by the mode
    assert abs(ref - float(_loss(m, good))) == 0.0
    N = good["att_feats"].size(2) if good["att_feats"].dim() == 4 else good["att_masks"].size(-1)
    cases = {
        "rel_ind": ("rel_ind", lambda t: t.index_put_((torch.tensor(0), torch.tensor(0), torch.tensor(0), torch.tensor(1)) if t.dim() == 4 else (torch.tensor(0), torch.tensor(0), torch.tensor(1)), torch.tensor(10 ** 6, device=DEV))),
        "gpn_obj_ind": ("gpn_obj_ind", lambda t: t.view(-1).index_put_((torch.tensor(5),), torch.tensor(N + 3, device=DEV))),
        "labels": ("tok|target", lambda t: t.view(-1).index_put_((torch.tensor(2),), torch.tensor(10 ** 5, device=DEV))),
    }
    for key, (pat, poke) in cases.items():
        b = {k: v.clone() for k, v in good.items()}
        poke(b[key])
        with ops.debug_bounds():
            with pytest.raises(ops.SubgcError, match=pat):
                _loss(m, b)
                torch.cuda.synchronize()
    # the reference's consistency assert (gpn.py:117-118): a node list that names a real node where the attention mask says "padding"
    b = {k: v.clone() for k, v in good.items()}
    flat_idx, flat_m = b["gpn_obj_ind"].view(-1), b["att_masks"].view(-1)
    pos = int((flat_m == 0).nonzero()[0])
    flat_idx[pos] = 0
    with ops.debug_bounds():
        with pytest.raises(ops.SubgcError, match="gpn_obj_ind vs att_masks.*gpn.py:117-118"):
            _loss(m, b)


def test_a_decode_call_checks_the_candidate_tables():
    torch.manual_seed(0)
    topt = dict(OPT, test_LSTM=1, gpn_nms_thres=0.55, gpn_max_subg=5)
    m = models.setup(argparse.Namespace(**topt)).to(DEV).eval()
    tb = {k: v.to(DEV) for k, v in synthetic.make_test_batch(12, D=128, n_obj_cls=40, seed=2, fc_size=128, node_pool=12).items()}
    with ops.debug_bounds():
        ret = m(*synthetic.sample_args(tb), opt=dict(sample_max=1, beam_size=1), mode="sample")
        assert ret[0].numel() > 0
        bad = {k: v.clone() for k, v in tb.items()}
        bad["gpn_obj_ind"].view(-1)[3] = 999
        with pytest.raises(ops.SubgcError, match="gpn_obj_ind"):
            m(*synthetic.sample_args(bad), opt=dict(sample_max=1, beam_size=1), mode="sample")
