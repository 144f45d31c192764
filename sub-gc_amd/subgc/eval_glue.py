"""The model-facing part of the reference's evaluation loop (misc/eval_utils.py:98-146, misc/utils.py:59-81,
misc/grd_utils.py:36-58).

What `eval_split` does between the model call and the metric scripts: rank an image's captions by sGPN score,
map the kept sub-graphs back to their original indices, turn token rows into sentences, collect one
`predictions` entry per image and -- for the grounding experiments (`return_att_weight`) -- find for every word of
the chosen caption the graph node with the largest attention weight.  Here the model call is `sample_images`
(many images per decode batch); ranking, reordering and the grounding arg-max are ONE launch each over the whole
batch (`subgc_eval_rank_rows`, `subgc_grounding_argmax`) and everything the host needs arrives in ONE copy per
decode batch (`ops.eval_collect`).  The text side of the grounding protocol (lemma -> detection class, box files:
`grounding_material`) is dictionary look-ups on the host; the COCO / GVD metric scripts are out of scope.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import ops

# misc/utils.py:16-17
BAD_ENDINGS = ("with", "in", "on", "of", "a", "at", "to", "for", "an", "this", "his", "her", "that", "the")


def decode_sequence(ix_to_word, seq, remove_bad_endings=None):
    """Token rows -> sentences (misc/utils.py:59-81): words up to the first 0; optionally strip dangling
    function words (`REMOVE_BAD_ENDINGS`, set by eval_utils.py:38-39)."""
    if remove_bad_endings is None:
        remove_bad_endings = int(os.getenv("REMOVE_BAD_ENDINGS", "0"))
    rows = seq.detach().cpu().tolist() if torch.is_tensor(seq) else [list(r) for r in seq]
    out = []
    for row in rows:
        words = []
        for ix in row:
            if ix <= 0:
                break
            words.append(ix_to_word[str(int(ix))])
        txt = " ".join(words)
        if remove_bad_endings:
            parts = txt.split(" ")
            cut = 0
            for j in range(len(parts)):
                if parts[-j - 1] not in BAD_ENDINGS:
                    cut = -j
                    break
            txt = " ".join(parts[0:len(parts) + cut])
        out.append(txt)
    return out


def rank_subgraphs(model, seqq, subgraph_score, keep_nms_ind, sct_mode=False):
    """eval_utils.py:105-121 -> (seq, subgraph_score, sorted_subgraph_ind, sort_ind)."""
    if sct_mode:                                                          # controllability: input order, first half only
        valid = subgraph_score.size(0) // 2
        return seqq[:valid], subgraph_score[:valid], keep_nms_ind[:valid], keep_nms_ind[:valid].long()
    if model.gpn:
        sorted_score, sort_ind = ops.rank_desc(subgraph_score.float())
        return seqq[sort_ind.to(seqq.device)], sorted_score, keep_nms_ind[sort_ind], sort_ind
    sort_ind = torch.arange(subgraph_score.size(0), device=keep_nms_ind.device).type_as(keep_nms_ind)
    return seqq, subgraph_score, keep_nms_ind, sort_ind


@torch.no_grad()
def caption_images(model, images, infos, ix_to_word, eval_kwargs=None, group=256, shard=False, grd_pick=None):
    """The testing branch of eval_split for a list of loader items: returns the `predictions` list
    (eval_utils.py:132-141): {'image_id', 'caption': [...], 'subgraph_score', 'sorted_subgraph_ind'} per image.

    Single-process by default, like the reference's eval (one process, eval_utils.py:98-104): safe to call on rank 0 only
    while the other ranks of a data-parallel job wait elsewhere.
    `shard=True` makes the call COLLECTIVE: EVERY rank of the default process group must call it with the SAME
    `images` / `infos` lists; images go round-robin across the ranks, every rank captions its share, the predictions are
    gathered once at the end and every rank returns the full list.  The image ids are exchanged first and a mismatch (a caller
    that already split its list per rank, or ranks evaluating different splits) raises on every rank instead of merging
    results of different lists by index.

    `eval_kwargs["return_att"] = 1` (the grounding experiments, eval_utils.py:98-101,143-146): every entry also gets
    `"grounding"`: {'subg_index', 'sort_ind', 'att2_ind', 'node_ind'} -- for the caption ranked `subg_index` (0 = best by sGPN score;
    `grd_pick[i]` = the consensus re-ranker's choice for image i, grd_utils.py:31-35) the arg-max attention column of every word
    position and the full-graph node id (= box row) it stands for (grd_utils.py:36-47; `grounding_material` finishes the entry)."""
    import torch.distributed as dist
    from . import parallel
    eval_kwargs = dict(eval_kwargs or {})
    world = dist.get_world_size() if dist.is_initialized() else 1
    if shard and world > 1:
        ids = [info["id"] for info in infos]
        if len(ids) != len(images):
            raise ValueError("caption_images: one `infos` entry per image")
        seen = [None] * world
        dist.all_gather_object(seen, ids)
        if any(s != ids for s in seen):
            raise ValueError("caption_images(shard=True) is collective: every rank must pass the same image list "
                             f"(rank {dist.get_rank()} has {len(ids)} images, the ranks hold {[len(s) for s in seen]})")
        mine, idx = parallel.shard_images(images, dist.get_rank(), world)
        local = caption_images(model, mine, [infos[i] for i in idx], ix_to_word, eval_kwargs, group, shard=False,
                               grd_pick=None if grd_pick is None else [grd_pick[i] for i in idx])
        return parallel.gather_by_index(local, idx, len(images))
    sct_mode = eval_kwargs.get("sct", 0) == 1
    rbe = eval_kwargs.get("remove_bad_endings", 0)
    return_att = eval_kwargs.get("return_att", 0) == 1
    if grd_pick is not None and len(grd_pick) != len(images):
        raise ValueError("caption_images: one grd_pick entry per image")
    was_training = model.training
    model.eval()
    predictions = []
    # a decode batch holds up to group x gpn_max_subg sub-graph rows (x beam): keep it near 8 k rows -- 256 images at the
    # Karpathy setting (10 sub-graphs), 8 at the MRNN setting (up to 1000), where one image already fills the chip
    group = max(1, min(group, 8192 // max(1, int(getattr(model, "gpn_max_subg", 1)) * max(1, int(eval_kwargs.get("beam_size", 1))))))
    try:
        for i in range(0, len(images), group):
            chunk, chunk_infos = images[i:i + group], infos[i:i + group]
            hold = {"skip_att": True}
            results = model.sample_images(chunk, opt=eval_kwargs) if sct_mode else _sample_batch(model, chunk, eval_kwargs, hold)
            if "bounds" not in hold:
                # per image: controllability mode (input order, first half, no ranking; rare) and models whose sample_images does not
                # expose the batch tensors
                for info, r in zip(chunk_infos, results):
                    seq, score, sorted_ind, _ = rank_subgraphs(model, r[0], r[2], r[3], sct_mode)
                    predictions.append({"image_id": info["id"], "caption": decode_sequence(ix_to_word, seq, rbe),
                                        "subgraph_score": score.cpu().numpy(), "sorted_subgraph_ind": sorted_ind.cpu().numpy()})
                continue
            bounds = hold["bounds"]
            if hold["rows"] == 0:
                for info in chunk_infos:
                    predictions.append({"image_id": info["id"], "caption": [], "subgraph_score": np.zeros(0, np.float32),
                                        "sorted_subgraph_ind": np.zeros(0, np.int64)})
                continue
            ground = return_att and hold.get("AL") is not None
            pick = None if grd_pick is None else grd_pick[i:i + group]
            # eval_utils.py:105-115 for every image of the batch + grd_utils.py:36-47: two launches, one host copy
            h = ops.eval_collect(hold["score"], hold["keep"], hold["seq"], bounds, identity=not model.gpn,
                                 AL=hold["AL"] if ground else None, idx=hold["idx"] if ground else None, pick=pick if ground else None)
            for j, (info, a, b) in enumerate(zip(chunk_infos, bounds, bounds[1:])):
                entry = {"image_id": info["id"], "caption": decode_sequence(ix_to_word, h["seq"][a:b], rbe),
                         "subgraph_score": h["score"][a:b], "sorted_subgraph_ind": h["keep"][a:b]}
                if ground:
                    w = int(h["n_words"][j])
                    entry["grounding"] = {"subg_index": 0 if pick is None else int(pick[j]), "sort_ind": h["order"][a:b],
                                          "att2_ind": h["att2"][j, :w].astype(np.int64), "node_ind": h["node"][j, :w].astype(np.int64)}
                predictions.append(entry)
    finally:
        model.train(was_training)
    return predictions


def _sample_batch(model, chunk, eval_kwargs, hold):
    try:
        return model.sample_images(chunk, opt=eval_kwargs, batch_out=hold)
    except TypeError as e:                                              # a stand-in model without the batch view (tests)
        if "batch_out" not in str(e):
            raise
        hold.clear()
        return model.sample_images(chunk, opt=eval_kwargs)


def grounding_material(entry, boxes, wd_to_lemma, lemma_det_id_dict, det_id_to_det_wd, img_wh=None):
    """misc/grd_utils.py:36-58 for one `predictions` entry of `caption_images(..., return_att=1)`: the sentence ranked
    `entry["grounding"]["subg_index"]`, its words -> lemma -> detection class; for every word that names one, the box of the node
    that word attended to most.  `boxes` [N_nodes, 4]: the scene-graph detector's boxes of the image (`img_wh` = (w, h): rescaled by
    max(w, h) / 592 like :27-28; None: used as given).  -> {'clss', 'idx_in_sent', 'bbox'} (what the reference appends to
    grd_output[image_id]).  Words are taken from the sentence (after remove_bad_endings, if that was on): arg-max positions beyond
    the sentence are simply not used, like the `[:len(grd_wd)]` slice."""
    g = entry["grounding"]
    boxes = np.asarray(boxes)
    if img_wh is not None:
        boxes = boxes * max(img_wh) / 592
    words = entry["caption"][g["subg_index"]].split()
    out = {"clss": [], "idx_in_sent": [], "bbox": []}
    for j, wd in enumerate(words[:len(g["node_ind"])]):
        if wd not in wd_to_lemma:
            continue
        lemma = wd_to_lemma[wd]
        if lemma in lemma_det_id_dict:
            out["bbox"].append(boxes[int(g["node_ind"][j])].tolist())
            out["clss"].append(det_id_to_det_wd[lemma_det_id_dict[lemma]])
            out["idx_in_sent"].append(j)
    return out
