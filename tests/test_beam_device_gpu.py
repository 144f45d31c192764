"""Beam search with the per-step bookkeeping on the device (subgc_beam_step) against the host bookkeeping of
subgc/beam.py::search (the restatement of CaptionModel.py:28-176 that tests/test_beam_oracle.py pins on the reference's
goldens): same engine, same decoder steps, every table compared bit for bit."""
import numpy as np
import pytest
import torch

from subgc import beam, synthetic
from subgc import functions as F_
from subgc.models import sampling
from test_parity_gpu import DEV, build

pytestmark = pytest.mark.gpu

OPTS = [dict(beam_size=2), dict(beam_size=3, length_penalty="wu_0.7"), dict(beam_size=5, decoding_constraint=1),
        dict(beam_size=4, group_size=2, diversity_lambda=0.5, decoding_constraint=1, length_penalty="avg_1.0"),
        dict(beam_size=6, group_size=3, diversity_lambda=0.3), dict(beam_size=10)]


def prepared(m, ims):
    att = torch.cat([im["att_feats"][:1] for im in ims])
    I, N, _ = att.shape
    X2 = m._encode(att, torch.cat([im["obj_dist"][:1] for im in ims]), torch.cat([im["pred_dist"][:1] for im in ims]),
                   torch.cat([im["rel_ind"][:1] for im in ims])).reshape(I * N, m.GCN_dim).contiguous()
    sel = sampling.select_subgraphs(m, X2, N, [(i, im["gpn_obj_ind"], im["att_masks"], im["gpn_pool_mtx"]) for i, im in enumerate(ims)])
    cat = lambda k: torch.cat([s[k] for s in sel]).contiguous()
    P = m._decoder_params()
    return F_.Prepared(cat("fc"), X2, cat("lens"), cat("idx"), cat("img"), N, P, None, None, 1.0), P, N


@pytest.mark.parametrize("opt", OPTS, ids=lambda o: "-".join(f"{k[:4]}{v}" for k, v in o.items()))
def test_device_beam_step_equals_host_bookkeeping(golden, opt):
    g = golden("subgc_beam3")
    m = build(g, golden("subgc_beam").group("weights"), False)
    ims = [{k: v.to(DEV) for k, v in g.tensors("inputs").items()}]
    pr, P, N = prepared(m, ims)
    T = m.seq_length
    with torch.no_grad():
        dev_out = beam.beam_decode(pr, P, N, T, dict(opt), xt_table=m.xt_gates_table(), on_device=True)
        host_out = beam.beam_decode(pr, P, N, T, dict(opt), xt_table=m.xt_gates_table(), on_device=False)
    assert torch.equal(dev_out[0], host_out[0]) and torch.equal(dev_out[1], host_out[1])
    assert len(dev_out[2]) == len(host_out[2]) == pr.S
    ended_early = 0
    for db, hb in zip(dev_out[2], host_out[2]):
        assert len(db) == len(hb) == opt["beam_size"]
        for d, h in zip(db, hb):
            assert torch.equal(d["seq"], h["seq"]) and torch.equal(d["logps"], h["logps"])
            assert d["p"] == h["p"] and abs(d["unaug_p"] - h["unaug_p"]) <= 1e-5 * max(1.0, abs(h["unaug_p"]))
            ended_early += int((d["seq"] == 0).any())
    assert ended_early > 0                                                       # the weights of this fixture do reach <eos>


def test_product_path_never_uses_the_host_bookkeeping(golden, monkeypatch):
    """One image: the search is captured once (hipGraph) and replayed; several images: launched eagerly.  Either way on the device."""
    g = golden("subgc_beam3")
    m = build(g, golden("subgc_beam").group("weights"), False)
    b = {k: v.to(DEV) for k, v in g.tensors("inputs").items()}
    monkeypatch.setattr(beam, "search", lambda *a, **k: (_ for _ in ()).throw(AssertionError("host bookkeeping on the product path")))
    from test_beam_oracle import check_beams
    for _ in range(3):                                                            # capture, then two replays
        ret = m(*synthetic.sample_args({k: v.clone() for k, v in b.items()}), opt=g.meta["sample_opt"], mode="sample")
        check_beams(ret, m.done_beams, g.group("out"), atol=1e-4)
    assert any(k[0] == "beam" for k in m._graph_cache)
    rs = m.sample_images([b, b], opt=g.meta["sample_opt"])
    for r, per in zip(rs, m.done_beams):
        check_beams(r, per, g.group("out"), atol=1e-4)
    m.decode_hipgraph = False                                                     # eager one-image search: same result
    m.__dict__.pop("_graph_cache")
    ret = m(*synthetic.sample_args(b), opt=g.meta["sample_opt"], mode="sample")
    check_beams(ret, m.done_beams, g.group("out"), atol=1e-4)
    assert not m.__dict__.get("_graph_cache")


def test_full_gc_beam3_as_in_test_sh_matches_the_oracle(golden):
    """test.sh decodes Full_GC_Kar with --beam_size 3: one row per image (no sub-graphs), 3 beams, replayed as a hipGraph."""
    from oracle import subgc_oracle as O
    g = golden("fullgc_greedy")
    w = golden("fullgc_train").group("weights")
    m = build(g, w, False)
    orc = O.Oracle(g.opt(), w)
    opt = dict(sample_max=1, beam_size=3)
    b = g.tensors("inputs")
    want = orc.sample_beam(*synthetic.sample_args({k: v.clone() for k, v in b.items()}), opt=opt)
    for _ in range(2):                                                            # capture, replay
        got = m(*synthetic.sample_args({k: v.clone().to(DEV) for k, v in b.items()}), opt=opt, mode="sample")
        assert torch.equal(got[0].cpu(), want[0])
        torch.testing.assert_close(got[1].cpu(), want[1], atol=1e-4, rtol=1e-4)
        assert len(m.done_beams) == 1 and len(m.done_beams[0]) == 3
        for d, h in zip(m.done_beams[0], want[4][0]):
            assert torch.equal(d["seq"], h["seq"])
            assert abs(d["p"] - h["p"]) < 2e-3


def test_sct_mode_with_beam2_as_in_test_sh(golden):
    """test.sh's controllability runs: --sct 1 --beam_size 2 -- every candidate sub-graph is decoded (no NMS), two beams each."""
    g = golden("subgc_sct")
    m = build(g, golden("subgc_beam").group("weights"), False)
    assert m.sct
    b = {k: v.to(DEV) for k, v in g.tensors("inputs").items()}
    opt = dict(sample_max=1, beam_size=2)
    got = m(*synthetic.sample_args(b), opt=opt, mode="sample")
    n = got[0].size(0)
    assert n == b["gpn_obj_ind"].size(2) * 2 and len(m.done_beams) == n          # pos + neg slots, all kept
    pr, P, N = prepared(m, [b])
    with torch.no_grad():
        host = beam.beam_decode(pr, P, N, m.seq_length, opt, xt_table=m.xt_gates_table(), on_device=False)
    assert torch.equal(got[0].cpu(), host[0]) and torch.equal(got[1].cpu(), host[1])
    for db, hb in zip(m.done_beams, host[2]):
        for d, h in zip(db, hb):
            assert torch.equal(d["seq"], h["seq"]) and d["p"] == h["p"]


def test_gather_rows_multi_is_four_gathers():
    from subgc import ops
    g = torch.Generator().manual_seed(5)
    S = 30
    src = [torch.randn(S, c, generator=g).to(DEV) for c in (2000, 3000, 1000, 1000)]
    views = [src[0], src[1][:, 2000:], src[2], src[3]]                            # one operand is a column slice, like the lang-LSTM slot
    rows = torch.randint(0, S, (S,), generator=g).to(torch.int32).to(DEV)
    rows[3] = -1                                                                  # negative row -> zeros, like subgc_gather_rows
    dst = [torch.full_like(v, 3.0) for v in views]
    dstv = [dst[0], dst[1], dst[2], dst[3]]
    ops.gather_rows_multi(list(zip(views, dstv)), rows)
    for v, d in zip(views, dstv):
        want = v[rows.long().clamp_min(0)].clone()
        want[3] = 0
        assert torch.equal(d, want)
    one = torch.empty_like(views[2])
    ops.gather_rows_multi([(views[2], one)], rows)
    assert torch.equal(one, dstv[2])
