"""Eval glue (misc/eval_utils.py:105-141, misc/utils.py:59-81): sentence decoding is pinned against strings produced by
the reference's own decode_sequence (tests/golden/make_golden.py eval_cases); the GPU test runs the whole
sample -> rank -> sentences pipeline for several images at once against the oracle run image by image."""
import numpy as np
import pytest
import torch

from oracle import subgc_oracle as O
from subgc import eval_glue, synthetic


@pytest.mark.parametrize("flag", [0, 1])
def test_decode_sequence_matches_reference_strings(golden, flag):
    g = golden("eval_glue")
    out = g.group("out")
    seq = torch.from_numpy(out["seq"])
    want = [str(s) for s in out[f"sents_{flag}"]]
    assert O.decode_sequence(g.meta["vocab"], seq, flag) == want
    assert eval_glue.decode_sequence(g.meta["vocab"], seq, flag) == want
    assert eval_glue.decode_sequence(g.meta["vocab"], seq.tolist(), flag) == want
    assert any(a != b for a, b in zip(out["sents_0"], out["sents_1"]))


@pytest.mark.gpu
@pytest.mark.parametrize("sct", [0, 1])
def test_caption_images_matches_oracle_pipeline(golden, sct):
    import subgc.models as models
    g = golden("subgc_greedy")
    w = golden("subgc_beam").group("weights")
    w["logit.bias"][0] += 2.0
    opt = g.opt(caption_model="topdown", gpn_drop_prob=0.0, sct=sct)
    m = models.setup(opt)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    m = m.to("cuda:0").eval()
    D = g.meta["opt"]["att_feat_size"]
    cpu = [synthetic.make_test_batch(M, D=D, seed=400 + i, fc_size=D, node_pool=pool) for i, (M, pool) in enumerate([(24, 14), (5, None), (30, 10)])]
    vocab = {str(i): f"w{i}" for i in range(1, 60)}
    vocab["7"], vocab["9"] = "the", "of"
    infos = [{"id": 1000 + i} for i in range(len(cpu))]
    kw = dict(sample_max=1, beam_size=1, sct=sct, remove_bad_endings=1)
    preds = eval_glue.caption_images(m, [{k: v.to("cuda:0") for k, v in b.items()} for b in cpu], infos, vocab, kw)
    orc = O.Oracle(opt, w)
    for b, info, p in zip(cpu, infos, preds):
        r = orc.sample(*synthetic.sample_args(b), opt=kw, nms_sort_kind="stable")
        seq, score, ind, _ = O.rank_subgraphs(True, r[0], r[2], r[3], bool(sct))
        assert p["image_id"] == info["id"]
        np.testing.assert_array_equal(p["sorted_subgraph_ind"], ind.numpy())
        np.testing.assert_allclose(p["subgraph_score"], score.numpy(), atol=1e-5)
        assert p["caption"] == O.decode_sequence(vocab, seq, 1)
    assert not m.training


@pytest.mark.gpu
def test_rank_desc_is_a_stable_descending_sort():
    from subgc import ops
    torch.manual_seed(0)
    for n in (1, 2, 37, 1000, 8192):
        s = torch.randn(n, device="cuda:0")
        s[::7] = float(s[0])                                      # ties
        srt, order = ops.rank_desc(s)
        ws, wo = torch.sort(s.cpu(), descending=True, stable=True)
        np.testing.assert_array_equal(order.cpu().numpy(), wo.numpy())
        np.testing.assert_array_equal(srt.cpu().numpy(), ws.numpy())
