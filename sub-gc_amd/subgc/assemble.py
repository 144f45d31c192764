"""On-device batch assembly: the tensor-building half of the reference loader (dataloaders/dataloader.py:269-367).

The reference builds every training batch with numpy loops inside `__getitem__` on 6 worker processes:
index compaction of the chosen sub-graph masks, a dense [5,2,hb,N,N] diagonal pooling matrix per image,
dummy-node padding of the scene graph, label/mask rows.  Here the loader only has to hand over the RAW
per-image arrays (already on the device) and the chosen sub-graph ids; three HBM-bound kernels build the
model's 14 arguments.  The random choice of sub-graphs (:229-270) and of captions (:139-157) is host-side integer work on
a handful of numbers per image and stays on the host (`choose_subgraphs`, `pick_captions`: same random streams as the
reference, so a seeded run draws the same mini-batches).  Pinned on what the reference's own `__getitem__` returns
(tests/golden/loader_*.npz).
"""
from __future__ import annotations

import random as _pyrandom

import numpy as np
import torch

from ._lib import call
from .ops import _ptr, _stream


def choose_subgraphs(node_iou_mtx, thres, gpn_batch, rng=np.random):
    """Ids (into the image's `subgraph_mask_list`) of the positive and negative sub-graphs of every sentence's mini-batch
    -> int [S, gpn_batch, 2] (dataloader.py:229-270).  Column j + 5 of `node_iou_mtx [S, 5 + M]` is the node IoU of sampled
    candidate j with each sentence's nouns; the first 5 entries of the list are the sentences' own sub-graphs.  `rng` is
    consumed call for call like the reference consumes np.random (randint, then choice)."""
    iou = np.asarray(node_iou_mtx)[:, 5:]
    S = iou.shape[0]
    pos = iou >= thres
    neg = (iou < thres) & ~pos.any(0)[None, :]               # a candidate that is positive for ANY sentence is nobody's negative
    share = pos / (pos.sum(0) + 1e-7)                         # a candidate shared by several sentences is drawn less often by each
    share = (share.T / (share.sum(1) + 1e-7)).T
    out = np.empty((S, gpn_batch, 2), dtype=np.int64)
    for i in range(S):
        cand = np.flatnonzero(pos[i])
        if cand.size < gpn_batch:                             # too few positives: the sentence's own sub-graph (entry i) fills the front
            out[i, :gpn_batch - cand.size, 0] = i
            out[i, gpn_batch - cand.size:, 0] = cand + 5
        else:
            p = share[i][cand]
            j = int(rng.randint(p.shape[0], size=1)[0])       # one random entry absorbs the rounding so that p sums to 1
            p[j] = 1.0 - (p.sum() - p[j])
            out[i, :, 0] = rng.choice(cand, size=gpn_batch, replace=True, p=p) + 5
        cand = np.flatnonzero(neg[i])
        if cand.size >= gpn_batch:
            out[i, :, 1] = rng.choice(cand, size=gpn_batch, replace=False) + 5
            continue
        loose = np.flatnonzero(iou[i] <= thres)               # fall back: at or below the threshold, else anything
        pool = np.flatnonzero(iou[i] <= 1.0) if loose.size == 0 else (loose if cand.size == 0 else cand)
        out[i, :, 1] = rng.choice(pool, size=gpn_batch, replace=True) + 5
    return out


def pick_captions(label, label_start_ix, label_end_ix, ix, seq_per_img, seq_length, rng=_pyrandom):
    """The `seq_per_img` caption rows of image `ix` (dataloader.py:139-157): the first ones, or draws with replacement when
    the image has fewer (python `random.randint`, as the reference)."""
    first, last = int(label_start_ix[ix]) - 1, int(label_end_ix[ix]) - 1          # the h5 pointers are 1-based
    if last < first:
        raise ValueError(f"image {ix} has no caption")
    if last - first + 1 >= seq_per_img:
        return np.asarray(label[first:first + seq_per_img, :seq_length])
    rows = [rng.randint(first, last) for _ in range(seq_per_img)]
    return np.asarray(label)[rows, :seq_length]


def pad_segments(src, start, count, R, pad):
    """dst[g, r] = src[start[g] + r] for r < min(count[g], R), else `pad` (int64 rows; subgc_pad_segments_i64)."""
    G, C = start.numel(), src.size(1)
    dst = torch.empty(G, R, C, device=start.device, dtype=torch.int64)
    call("subgc_pad_segments_i64", _ptr(src.contiguous(), torch.int64) if src.numel() else None, _ptr(start.contiguous(), torch.int64),
         _ptr(count.contiguous(), torch.int64), G, R, C, int(pad), _ptr(dst), _stream())
    return dst


def subgraph_indices(node_mask, N, pad=None, want_att_mask=True, want_pool_mtx=False):
    """node_mask [..., W] (bool/uint8) -> (ind [..., N] int64, att_mask [..., N] | None, pool_mtx [..., N, N] | None)
    (dataloader.py:276-291: `nonzero()` positions, dummy index padding, prefix mask, diagonal scatter matrix)."""
    lead, W = node_mask.shape[:-1], node_mask.size(-1)
    m = node_mask.reshape(-1, W).to(torch.uint8).contiguous()
    G, dev = m.size(0), m.device
    ind = torch.empty(G, N, device=dev, dtype=torch.int64)
    att = torch.empty(G, N, device=dev, dtype=torch.float32) if want_att_mask else None
    pool = torch.empty(G, N, N, device=dev, dtype=torch.float32) if want_pool_mtx else None
    call("subgc_mask_compact", _ptr(m), m.stride(0), G, W, N, int(N - 1 if pad is None else pad), _ptr(ind), _ptr(att), _ptr(pool), _stream())
    return (ind.view(*lead, N), None if att is None else att.view(*lead, N), None if pool is None else pool.view(*lead, N, N))


def pad_rows(src, off, R, limit, *, onehot0=False, pad=0):
    """Packed ragged rows -> [B, R, C] with the loader's padding rows (dataloader.py:336-354)."""
    B, C = off.numel() - 1, src.size(1)
    off = off.to(torch.int64).contiguous()
    dst = torch.empty(B, R, C, device=src.device, dtype=src.dtype)
    if src.dtype == torch.float32:
        call("subgc_pad_rows_f32", _ptr(src.contiguous()), _ptr(off), B, R, C, int(limit), int(onehot0), _ptr(dst), _stream())
    elif src.dtype == torch.int64:
        call("subgc_pad_rows_i64", _ptr(src.contiguous()), _ptr(off), B, R, C, int(limit), int(pad), _ptr(dst), _stream())
    else:
        raise TypeError(f"pad_rows: float32 or int64 rows, got {src.dtype}")
    return dst


def caption_labels(captions):
    """captions [S, seq_length] int64 -> (labels [S, seq_length+2], masks [S, seq_length+2]) (dataloader.py:356-363)."""
    S, Lq = captions.shape
    captions = captions.contiguous()
    labels = torch.empty(S, Lq + 2, device=captions.device, dtype=torch.int64)
    masks = torch.empty(S, Lq + 2, device=captions.device, dtype=torch.float32)
    call("subgc_caption_labels", _ptr(captions, torch.int64), captions.stride(0), S, Lq, _ptr(labels), _ptr(masks), _stream())
    return labels, masks


def assemble_train_batch(raw, obj_num, rel_num, want_pool_mtx=True):
    """raw (all on the device):
         object_fmap [B, obj_num-1, D] f32, object_dist [B, obj_num-1, C] f32           (sg_output, :331-333)
         rel_ind [sum_k, 2] i64, pred_dist [sum_k, P] f32, rel_off [B+1] i64            (ragged relations, :344-354)
         node_mask [B*S, 2, hb, obj_num-1] u8, pred_mask [B*S, 2, hb, rel_num-1] u8     (the chosen pos/neg sub-graphs, :276-301)
         captions [B*S, seq_length] i64                                                 (:356)
         optional: nrel [sum, 2] i64 + nrel_start, nrel_count [B*S, 2, hb] i64: the re-indexed relation endpoints of the
         chosen sub-graphs as segments of one packed array (:303-308; the model never reads gpn_nrel_ind, the dict has it)
       -> the dict `get_batch` returns (dataloader.py:190-205), ready for LossWrapper."""
    fmap, dist = raw["object_fmap"], raw["object_dist"]
    B, n_obj, D = fmap.shape
    if n_obj != obj_num - 1:
        raise ValueError(f"the loader pads exactly one dummy node: expected {obj_num - 1} objects per image, got {n_obj}")
    dev = fmap.device
    per_img = torch.arange(B + 1, device=dev, dtype=torch.int64) * n_obj
    out = {
        "fc_feats": torch.zeros(B, D, device=dev),                                                       # :343
        "att_feats": pad_rows(fmap.reshape(B * n_obj, D), per_img, obj_num, n_obj),                      # dummy node = zeros (:340)
        "obj_dist": pad_rows(dist.reshape(B * n_obj, -1), per_img, obj_num, n_obj, onehot0=True),        # dummy = class 0 (:341)
        "rel_ind": pad_rows(raw["rel_ind"], raw["rel_off"], rel_num, rel_num - 1, pad=obj_num - 1),      # :349,353-354
        "pred_dist": pad_rows(raw["pred_dist"], raw["rel_off"], rel_num, rel_num - 1, onehot0=True),     # :350,353
    }
    out["gpn_obj_ind"], out["att_masks"], out["gpn_pool_mtx"] = subgraph_indices(raw["node_mask"], obj_num, want_pool_mtx=want_pool_mtx)
    out["gpn_pred_ind"], _, _ = subgraph_indices(raw["pred_mask"], rel_num, want_att_mask=False)
    if "nrel" in raw:
        lead = raw["nrel_start"].shape
        out["gpn_nrel_ind"] = pad_segments(raw["nrel"], raw["nrel_start"].reshape(-1), raw["nrel_count"].reshape(-1), rel_num,
                                           obj_num - 1).view(*lead, rel_num, 2)
    out["labels"], out["masks"] = caption_labels(raw["captions"])
    return out
