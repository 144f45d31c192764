"""Synthetic scene-graph batches with the tensor contract of the reference loader.

No dataset ships with the reference (`/root/reference/.MISSING_LARGE_BLOBS`), so
tests, golden-vector generation and `bench.py` all draw batches from here.  The
shapes/dtypes/padding conventions follow `dataloaders/dataloader.py:194-205`
(dict layout) and `:269-356` (dummy node N-1, dummy relation K-1, prefix masks,
diagonal pooling matrix); the distributions are the ones SURVEY.md §8(d) fixes.

Pure numpy + torch CPU: the caller moves tensors to the device.
"""
from __future__ import annotations

import numpy as np
import torch

TRAIN_KEYS = (
    "fc_feats", "att_feats", "labels", "masks", "att_masks", "obj_dist", "rel_ind",
    "pred_dist", "gpn_obj_ind", "gpn_pred_ind", "gpn_nrel_ind", "gpn_pool_mtx",
)


def _row_normalised(rng, shape):
    d = rng.random(shape, dtype=np.float32)
    d /= d.sum(-1, keepdims=True)
    return d


def _subgraphs(rng, lead, N, min_nodes, max_nodes, node_pool=None):
    """Index lists / prefix masks / diagonal pool matrices for `lead`-shaped sub-graph slots."""
    obj_ind = np.full(lead + (N,), N - 1, dtype=np.int64)
    masks = np.zeros(lead + (N,), dtype=np.float32)
    pool = np.zeros(lead + (N, N), dtype=np.float32)
    for pos in np.ndindex(*lead):
        n = int(rng.integers(min_nodes, max_nodes + 1))
        nodes = np.sort(rng.choice(node_pool or (N - 1), size=min(n, node_pool or n), replace=False))
        n = len(nodes)
        obj_ind[pos][:n] = nodes
        masks[pos][:n] = 1.0
        pool[pos][np.arange(n), np.arange(n)] = 1.0
    return obj_ind, masks, pool


def make_graph(rng, B, N, K, D, n_obj_cls, n_pred_cls, n_edges=None, fc_size=None):
    """Per-image scene graph: region features, class distributions, relation endpoints."""
    n_edges = K - 1 if n_edges is None else n_edges
    att = np.abs(rng.standard_normal((B, N, D), dtype=np.float32))
    att[:, N - 1] = 0.0  # dummy node (dataloader.py:343-346)
    obj_dist = _row_normalised(rng, (B, N, n_obj_cls))
    obj_dist[:, N - 1] = 0.0
    obj_dist[:, N - 1, 0] = 1.0
    pred_dist = _row_normalised(rng, (B, K, n_pred_cls))
    pred_dist[:, n_edges:] = 0.0
    pred_dist[:, n_edges:, 0] = 1.0
    rel_ind = np.full((B, K, 2), N - 1, dtype=np.int64)
    rel_ind[:, :n_edges] = rng.integers(0, N - 1, size=(B, n_edges, 2))
    fc = np.zeros((B, fc_size or D), dtype=np.float32)  # dataloader.py:343 (all zeros)
    return dict(fc_feats=fc, att_feats=att, obj_dist=obj_dist, pred_dist=pred_dist, rel_ind=rel_ind)


def make_train_batch(B, *, N=37, K=65, D=2048, vocab=9487, seq_length=16, S=5, hb=2,
                     n_obj_cls=1599, n_pred_cls=21, n_edges=None, min_nodes=2, max_nodes=11,
                     min_len=5, max_len=None, seed=0, fc_size=None):
    """A training batch of B images (S sentences each, hb pos + hb neg sub-graphs per sentence)."""
    rng = np.random.default_rng(seed)
    max_len = seq_length if max_len is None else max_len
    out = make_graph(rng, B, N, K, D, n_obj_cls, n_pred_cls, n_edges, fc_size)
    obj_ind, masks, pool = _subgraphs(rng, (B * S, 2, hb), N, min_nodes, min(max_nodes, N - 1))
    out["gpn_obj_ind"], out["att_masks"], out["gpn_pool_mtx"] = obj_ind, masks, pool
    out["gpn_pred_ind"] = np.full((B * S, 2, hb, K), K - 1, dtype=np.int64)
    out["gpn_nrel_ind"] = np.full((B * S, 2, hb, K, 2), N - 1, dtype=np.int64)
    labels = np.zeros((B * S, seq_length + 2), dtype=np.int64)
    lmask = np.zeros((B * S, seq_length + 2), dtype=np.float32)
    for j in range(B * S):
        n = int(rng.integers(min_len, max_len + 1))
        labels[j, 1:n + 1] = rng.integers(1, vocab + 1, size=n)
        lmask[j, :n + 2] = 1.0
    out["labels"], out["masks"] = labels, lmask
    return {k: torch.from_numpy(v) for k, v in out.items()}


def make_test_batch(M, *, N=37, K=65, D=2048, S=5, n_obj_cls=1599, n_pred_cls=21, n_edges=None,
                    min_nodes=2, max_nodes=11, seed=0, fc_size=None, node_pool=None):
    """One test image with M candidate (pos, neg) sub-graph pairs, replicated over the S
    counterparts exactly as `dataloader_test.py` does (the model only reads counterpart 0)."""
    rng = np.random.default_rng(seed)
    out = make_graph(rng, 1, N, K, D, n_obj_cls, n_pred_cls, n_edges, fc_size)
    obj_ind, masks, pool = _subgraphs(rng, (1, 2, M), N, min_nodes, min(max_nodes, N - 1), node_pool)
    rep = lambda a: np.ascontiguousarray(np.broadcast_to(a, (S,) + a.shape[1:]))
    out["gpn_obj_ind"], out["att_masks"], out["gpn_pool_mtx"] = rep(obj_ind), rep(masks), rep(pool)
    out["gpn_pred_ind"] = np.full((S, 2, M, K), K - 1, dtype=np.int64)
    out["gpn_nrel_ind"] = np.full((S, 2, M, K, 2), N - 1, dtype=np.int64)
    return {k: torch.from_numpy(v) for k, v in out.items()}


def forward_args(batch, device=None):
    """Positional args of `model(...)` in train mode (`loss_wrapper.py:18-19`)."""
    g = (lambda k: batch[k].to(device) if device is not None else batch[k])
    return (g("fc_feats"), g("att_feats"), g("labels"), g("att_masks"), None, g("obj_dist"), None,
            g("rel_ind"), None, g("pred_dist"), g("gpn_obj_ind"), g("gpn_pred_ind"),
            g("gpn_nrel_ind"), g("gpn_pool_mtx"))


def sample_args(batch, device=None):
    """Positional args of `model(..., mode='sample')` (`eval_utils.py:98-104`): no labels."""
    a = forward_args({**batch, "labels": batch.get("labels", batch["rel_ind"])}, device)
    return a[:2] + a[3:]
