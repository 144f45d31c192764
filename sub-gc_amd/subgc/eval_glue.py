"""The model-facing part of the reference's evaluation loop (misc/eval_utils.py:98-141, misc/utils.py:59-81).

What `eval_split` does between the model call and the metric scripts: rank an image's captions by sGPN score,
map the kept sub-graphs back to their original indices, turn token rows into sentences and collect one
`predictions` entry per image.  Here the model call is `sample_images` (many images per decode batch), the
ranking is a device kernel (`subgc_rank_desc_f32`) and only the final token rows cross to the host.
Grounding material (`get_grounding_material`, GVD dictionaries) and the COCO metric scripts are outside
the hot path (out of scope).
"""
from __future__ import annotations

import os

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
def caption_images(model, images, infos, ix_to_word, eval_kwargs=None, group=256, shard=False):
    """The testing branch of eval_split for a list of loader items: returns the `predictions` list
    (eval_utils.py:132-141): {'image_id', 'caption': [...], 'subgraph_score', 'sorted_subgraph_ind'} per image.

    Single-process by default, like the reference's eval (one process, eval_utils.py:98-104): safe to call on rank 0 only
    while the other ranks of a data-parallel job wait elsewhere.
    `shard=True` makes the call COLLECTIVE: EVERY rank of the default process group must call it with the SAME
    `images` / `infos` lists; images go round-robin across the ranks, every rank captions its share, the predictions are
    gathered once at the end and every rank returns the full list.  The image ids are exchanged first and a mismatch (a caller
    that already split its list per rank, or ranks evaluating different splits) raises on every rank instead of merging
    results of different lists by index."""
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
        local = caption_images(model, mine, [infos[i] for i in idx], ix_to_word, eval_kwargs, group, shard=False)
        return parallel.gather_by_index(local, idx, len(images))
    sct_mode = eval_kwargs.get("sct", 0) == 1
    rbe = eval_kwargs.get("remove_bad_endings", 0)
    was_training = model.training
    model.eval()
    predictions = []
    # a decode batch holds up to group x gpn_max_subg sub-graph rows (x beam): keep it near 8 k rows -- 256 images at the
    # Karpathy setting (10 sub-graphs), 8 at the MRNN setting (up to 1000), where one image already fills the chip
    group = max(1, min(group, 8192 // max(1, int(getattr(model, "gpn_max_subg", 1)) * max(1, int(eval_kwargs.get("beam_size", 1))))))
    try:
        for i in range(0, len(images), group):
            results = model.sample_images(images[i:i + group], opt=eval_kwargs)
            for info, r in zip(infos[i:i + group], results):
                seq, score, sorted_ind, _ = rank_subgraphs(model, r[0], r[2], r[3], sct_mode)
                predictions.append({"image_id": info["id"], "caption": decode_sequence(ix_to_word, seq, rbe),
                                    "subgraph_score": score.cpu().numpy(), "sorted_subgraph_ind": sorted_ind.cpu().numpy()})
    finally:
        model.train(was_training)
    return predictions
