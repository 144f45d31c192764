"""Grounding material (misc/grd_utils.py:36-58).  tests/golden/grd_out.npz holds what the reference's OWN
get_grounding_material appended to grd_output for fabricated detector boxes / sub-graph masks (make_golden.py grd_cases): the boxes of
the arg-max nodes of every grounded word, for the best-ranked caption and for a consensus re-ranker's pick.

CPU: the oracle's restatement reproduces it.  GPU: `eval_glue.caption_images(..., return_att=1)` -- all images of a case in ONE decode
batch, one ranking launch, one grounding launch, one host copy -- reproduces it through `eval_glue.grounding_material`; and the kernel
equals the oracle on the Flickr stress shape (101 nodes), where the reference's own test branch cannot run."""
import numpy as np
import pytest
import torch

from oracle import subgc_oracle as O
from subgc import synthetic


def _case(golden, name):
    g = golden("grd")
    meta = g.meta
    opt = dict(meta["opt"][name])
    opt.setdefault("obj_name_path", None); opt.setdefault("rel_name_path", None)
    import argparse
    w = golden("subgc_beam").group("weights") if name == "subgc" else golden("fullgc_train").group("weights")
    if name == "subgc":
        w["logit.bias"][0] += 0.5
    det_wd = {int(k): v for k, v in meta["det_id_to_det_wd"].items()}
    return g.group("out"), meta, argparse.Namespace(**opt), w, det_wd


def _canonical(b, M):
    """candidate id -> the smallest candidate id with the same node set."""
    oi, am = b["gpn_obj_ind"][0].numpy().reshape(2 * M, -1), b["att_masks"][0].numpy().reshape(2 * M, -1)
    sets = [tuple(oi[c][am[c] > 0].tolist()) for c in range(2 * M)]
    first = {}
    for c, st in enumerate(sets):
        first.setdefault(st, c)
    rep = np.array([first[st] for st in sets])
    return lambda a: rep[np.asarray(a, dtype=np.int64)]


def _check(out, name, img, pick_consensus, got):
    for consensus in (0, 1):
        tag = f"{name}_{img['id']}_{consensus}"
        r = got[consensus]
        np.testing.assert_array_equal(np.array(r["bbox"], np.float64).reshape(-1, 4), out[tag + "_bbox"], err_msg=tag)
        np.testing.assert_array_equal(np.array(r["idx_in_sent"], np.int64), out[tag + "_idx_in_sent"], err_msg=tag)
        assert list(r["clss"]) == [str(c) for c in out[tag + "_clss"]], tag


@pytest.mark.parametrize("name", ["subgc", "fullgc"])
def test_oracle_grounding_matches_reference_function(golden, name):
    out, meta, opt, w, det_wd = _case(golden, name)
    orc = O.Oracle(opt, w)
    gpn = name == "subgc"
    for img in meta["cases"][name]:
        b = synthetic.make_test_batch(img["M"], D=opt.att_feat_size, seed=img["seed"], fc_size=opt.att_feat_size, node_pool=img["pool"])
        seqq, _, score, keep, att = orc.sample(*synthetic.sample_args(b), opt=dict(sample_max=1, beam_size=1, return_att=1))
        seq, _, sorted_ind, sort_ind = O.rank_subgraphs(gpn, seqq, score, keep)
        sents = O.decode_sequence(meta["vocab"], seq)
        assert sents == [str(s) for s in out[f"{name}_{img['id']}_sents"]]
        np.testing.assert_array_equal(sorted_ind.numpy(), out[f"{name}_{img['id']}_sorted_ind"])
        boxes = out[f"{name}_{img['id']}_boxes"] * max(img["wh"]) / 592
        got = {}
        for consensus, pick in ((0, 0), (1, img["pick"])):
            if gpn:
                cand = int(sorted_ind[pick])                                          # candidate id in the original order (pos half | neg half)
                M = img["M"]
                nodes = b["gpn_obj_ind"][0, cand // M, cand % M][b["att_masks"][0, cand // M, cand % M] > 0].numpy()
            else:
                nodes = np.arange(36)
            _, node = O.grounding_argmax(att, sort_ind if gpn else None, pick, len(sents[pick].split()), nodes)
            got[consensus] = O.grounding_material(sents, pick, node, boxes, meta["wd_to_lemma"], meta["lemma_det_id_dict"], det_wd)
        _check(out, name, img, img["pick"], got)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["subgc", "fullgc"])
def test_caption_images_grounding_matches_reference_function(golden, name):
    import subgc.models as models
    from subgc import eval_glue
    out, meta, opt, w, det_wd = _case(golden, name)
    opt.caption_model = "topdown"
    m = models.setup(opt)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    m = m.to("cuda:0").eval()
    imgs = meta["cases"][name]
    cpu = [synthetic.make_test_batch(i["M"], D=opt.att_feat_size, seed=i["seed"], fc_size=opt.att_feat_size, node_pool=i["pool"]) for i in imgs]
    dev = [{k: v.to("cuda:0") for k, v in b.items()} for b in cpu]
    infos = [{"id": i["id"]} for i in imgs]
    kw = dict(sample_max=1, beam_size=1, return_att=1)
    res = {}
    for consensus in (0, 1):
        preds = eval_glue.caption_images(m, dev, infos, meta["vocab"], kw, grd_pick=[i["pick"] for i in imgs] if consensus else None)
        res[consensus] = preds
    for j, img in enumerate(imgs):
        got = {}
        for consensus in (0, 1):
            p = res[consensus][j]
            assert p["image_id"] == img["id"]
            assert p["caption"] == [str(s) for s in out[f"{name}_{img['id']}_sents"]]
            # candidates with IDENTICAL node sets tie in score; which of them survives the NMS depends on the reference's unstable
            # np.argsort (gpn.py:112) -- compare the kept candidates up to that choice (same node set => same caption, same boxes)
            canon = _canonical(cpu[j], img["M"]) if name == "subgc" else (lambda a: a)
            np.testing.assert_array_equal(canon(p["sorted_subgraph_ind"]), canon(out[f"{name}_{img['id']}_sorted_ind"]))
            assert p["grounding"]["subg_index"] == (img["pick"] if consensus else 0)
            got[consensus] = eval_glue.grounding_material(p, out[f"{name}_{img['id']}_boxes"], meta["wd_to_lemma"], meta["lemma_det_id_dict"],
                                                          det_wd, img_wh=img["wh"])
        _check(out, name, img, img["pick"], got)
    # the one-image, reference-shaped call returns the same attention rows the batch buffer holds
    one = m(*synthetic.sample_args(dev[0]), opt=kw, mode="sample")
    orc = O.Oracle(opt, w)
    want = orc.sample(*synthetic.sample_args(cpu[0]), opt=kw, nms_sort_kind="stable")
    np.testing.assert_array_equal(one[4].argmax(2).cpu().numpy(), want[4].argmax(2).numpy())


@pytest.mark.gpu
@pytest.mark.parametrize("N,I", [(101, 7), (37, 1), (12, 64)])
def test_grounding_argmax_kernel_equals_oracle_on_other_shapes(N, I):
    """Flickr stress shape (101 nodes) and batch extremes: random attention rows with exact ties, random token rows incl. empty and
    full-length captions, random ranking; indices must be exact."""
    from subgc import ops
    rng = np.random.default_rng(N * 100 + I)
    T, T1 = 20, 21
    sizes = rng.integers(0, 11, size=I)
    sizes[0] = 10
    bounds = [0] + np.cumsum(sizes).tolist()
    rows = bounds[-1]
    AL = rng.random((T1, rows, N)).astype(np.float32)
    AL[:, :, N // 2:] = 0.0                                               # padded columns
    AL[3, :, 1] = AL[3, :, 4] = 2.0                                       # an exact tie: the first column wins
    AL[5] = 0.0                                                           # an all-zero step (after an early break): column 0
    seq = rng.integers(1, 50, size=(rows, T))
    for r in range(rows):
        seq[r, rng.integers(0, T + 1):] = 0
    if rows:
        seq[0, :] = 7                                                     # no <eos> at all: 20 words
    score = rng.random(rows).astype(np.float32)
    if rows > 3:
        score[2] = score[1]                                               # tie in the ranking: input order
    keep = rng.integers(0, 1000, size=rows)
    idx = np.stack([np.sort(rng.permutation(N)) for _ in range(max(rows, 1))])[:rows]
    pick = [int(rng.integers(0, max(s, 1))) for s in sizes]
    d = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to("cuda:0")
    for identity in (False, True):
        h = ops.eval_collect(d(score, torch.float32), d(keep, torch.int64), d(seq, torch.int64), bounds, identity=identity,
                             AL=d(AL, torch.float32), idx=d(idx, torch.int64), pick=pick)
        for i, (a, b) in enumerate(zip(bounds, bounds[1:])):
            sc, sq, kp = torch.from_numpy(score[a:b]), torch.from_numpy(seq[a:b]), torch.from_numpy(keep[a:b])
            s_seq, s_score, s_keep, sort_ind = O.rank_subgraphs(not identity, sq, sc, kp)
            np.testing.assert_array_equal(h["order"][a:b], sort_ind.numpy())
            np.testing.assert_array_equal(h["seq"][a:b], s_seq.numpy())
            np.testing.assert_array_equal(h["keep"][a:b], s_keep.numpy())
            np.testing.assert_array_equal(h["score"][a:b], s_score.numpy())
            if b == a:
                assert h["n_words"][i] == 0 and (h["att2"][i] == -1).all() and (h["node"][i] == -1).all()
                continue
            row = int(sort_ind[pick[i]])
            w = int((np.cumprod(seq[a + row] > 0)).sum())
            att = torch.from_numpy(AL[:, a:b]).permute(1, 0, 2)
            att2, node = O.grounding_argmax(att, sort_ind, pick[i], w, idx[a + row])
            assert h["n_words"][i] == w
            np.testing.assert_array_equal(h["att2"][i, :w], att2)
            np.testing.assert_array_equal(h["node"][i, :w], node)
            assert (h["att2"][i, w:] == -1).all() and (h["node"][i, w:] == -1).all()


@pytest.mark.gpu
def test_eval_rank_rows_nan_scores_and_row_limit():
    """subgc_eval_rank_rows at its advertised limit (8192 rows of one image: 65 544 bytes of dynamic LDS, above the 64 KiB default) and
    with NaN scores: NaN ranks as -inf with the index as tie-break, so the order is a permutation (no rank taken twice, none left out)."""
    from subgc import ops
    rng = np.random.default_rng(5)
    rows, T = 8192 + 300, 4
    bounds = [0, 8192, rows]
    score = rng.random(rows).astype(np.float32)
    score[[3, 100, 8191, 8200, 8201]] = np.nan
    score[50] = score[49]
    seq = rng.integers(1, 50, size=(rows, T))
    keep = np.arange(rows)
    dev = "cuda:0"
    out = ops.eval_collect(torch.from_numpy(score).to(dev), torch.from_numpy(keep).to(dev), torch.from_numpy(seq).to(dev), bounds)
    for a, b in zip(bounds, bounds[1:]):
        key = np.where(np.isnan(score[a:b]), -np.inf, score[a:b])
        want = np.argsort(-key, kind="stable")
        np.testing.assert_array_equal(out["order"][a:b], want)
        np.testing.assert_array_equal(out["keep"][a:b], keep[a:b][want])
        np.testing.assert_array_equal(out["seq"][a:b], seq[a:b][want])
    with pytest.raises(ops.SubgcError):
        ops.eval_collect(torch.zeros(8193, device=dev), torch.zeros(8193, device=dev, dtype=torch.int64), torch.zeros(8193, T, device=dev, dtype=torch.int64), [0, 8193])
