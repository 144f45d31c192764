"""On-device batch assembly: the tensor-building half of the reference loader (dataloaders/dataloader.py:269-367).

The reference builds every training batch with numpy loops inside `__getitem__` on 6 worker processes:
index compaction of the chosen sub-graph masks, a dense [5,2,hb,N,N] diagonal pooling matrix per image,
dummy-node padding of the scene graph, label/mask rows.  Here the loader only has to hand over the RAW
per-image arrays (already on the device) and the chosen sub-graph ids; three HBM-bound kernels build the
model's 14 arguments.  The random choice of sub-graphs (:232-267) is host-side integer work and stays there.
"""
from __future__ import annotations

import torch

from ._lib import call
from .ops import _ptr, _stream


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
    out["labels"], out["masks"] = caption_labels(raw["captions"])
    return out
