"""The decode-side index kernels of round 4 against their torch formulations (what models/sampling.py did with ATen ops before):
subgc_gpn_test_prep (cat / diagonal / sum over counterpart 0 of every image, read through ONE address table), subgc_gather_blocks
(torch.cat of block 0 of every image's loader tensor), subgc_nms_compact (survivors in image order), subgc_gather_rows_multi_i64 (index
with int64 rows, any 4- / 8-byte dtype), subgc_decode_batch_finish (per-image early break of a batched decode), the word-level
zero_ / copy_ helpers and the pinned upload ring.  All integer / index work: bit-exact."""
import numpy as np
import pytest
import torch

from subgc import ops, synthetic

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _images(sizes, N=37, seed=0):
    out = []
    for i, M in enumerate(sizes):
        b = synthetic.make_test_batch(M, seed=seed + i, D=64, fc_size=64, n_obj_cls=30) if M else None
        if b is None:                                               # an image without candidates: zero-size loader tensors
            b = synthetic.make_test_batch(1, seed=seed + i, D=64, fc_size=64, n_obj_cls=30)
            b = {k: (v[:, :, :0].contiguous() if k in ("gpn_obj_ind", "att_masks", "gpn_pool_mtx") else v) for k, v in b.items()}
        out.append({k: v.to(DEV) for k, v in b.items()})
    return out


def test_gpn_test_prep_equals_cat_diagonal_sum():
    ims = _images([7, 0, 3, 12])
    N = ims[0]["att_feats"].size(1)
    items = [(10 + i, im["gpn_obj_ind"], im["att_masks"], im["gpn_pool_mtx"]) for i, im in enumerate(ims)]
    idx, w, denom, lens, img, o32, sizes, alive = ops.gpn_test_prep(items, N, DEV)
    want_idx = torch.cat([g[0].reshape(-1, N) for _, g, _, _ in items])
    want_w = torch.cat([p_[0].diagonal(dim1=-2, dim2=-1).reshape(-1, N) for _, _, _, p_ in items])
    want_len = torch.cat([a[0].reshape(-1, N) for _, _, a, _ in items]).sum(1)
    assert sizes == [14, 0, 6, 24] and o32.cpu().tolist() == [0, 14, 14, 20, 44]
    assert torch.equal(idx, want_idx) and torch.equal(w, want_w) and torch.equal(denom, want_len) and torch.equal(lens, want_len.int())
    assert img.cpu().tolist() == [10] * 14 + [12] * 6 + [13] * 24


def test_gather_blocks_stacks_counterpart_zero():
    ims = _images([5, 2, 9])
    keys = ("att_feats", "obj_dist", "pred_dist", "rel_ind")
    outs = ops.stack_first([[im[k] for im in ims] for k in keys])
    for k, o in zip(keys, outs):
        assert torch.equal(o, torch.cat([im[k][:1] for im in ims])) and o.dtype == ims[0][k].dtype


def test_nms_compact_and_take_rows():
    g = torch.Generator().manual_seed(3)
    sizes = [6, 0, 9, 4]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    n_keep = torch.tensor([3, 0, 9, 1], dtype=torch.int32, device=DEV)
    keep_all = torch.full((int(offs[-1]),), -7, dtype=torch.int64, device=DEV)
    want_keep, want_glob = [], []
    for b, (o, nk) in enumerate(zip(offs[:-1], n_keep.cpu().tolist())):
        ks = sorted(torch.randperm(sizes[b], generator=g)[:nk].tolist())
        keep_all[o:o + nk] = torch.tensor(ks, dtype=torch.int64, device=DEV)
        want_keep += ks; want_glob += [k + int(o) for k in ks]
    o32 = torch.tensor(offs, dtype=torch.int32, device=DEV)
    keep, glob = ops.nms_compact(keep_all, n_keep, o32, len(sizes), 13)
    assert keep.cpu().tolist() == want_keep and glob.cpu().tolist() == want_glob
    # gather with int64 rows over mixed dtypes in one launch
    G = int(offs[-1])
    a = torch.randn(G, 10, generator=g).to(DEV); b = torch.randint(0, 99, (G, 5), generator=g).to(DEV)
    c = torch.randint(0, 99, (G,), generator=g, dtype=torch.int32).to(DEV); d = torch.randn(G, generator=g).to(DEV)
    oa, ob = torch.empty(13, 10, device=DEV), torch.empty(13, 5, device=DEV, dtype=torch.int64)
    oc, od = torch.empty(13, device=DEV, dtype=torch.int32), torch.empty(13, device=DEV)
    ops.take_rows([(a, oa), (b, ob), (c, oc), (d, od)], glob)
    assert torch.equal(oa, a[glob]) and torch.equal(ob, b[glob]) and torch.equal(oc, c[glob]) and torch.equal(od, d[glob])


def test_decode_batch_finish_equals_the_torch_formula():
    g = torch.Generator().manual_seed(5)
    T, sizes = 20, [4, 1, 7, 3, 2]
    n = sum(sizes)
    seq = torch.randint(1, 50, (n, T), generator=g)
    ends = torch.randint(0, T + 3, (n,), generator=g)
    for r in range(n):
        seq[r, int(ends[r]):] = 0                                   # a finished row emits 0 from there on
    seq[4] = 0                                                      # the one-row image stops at step 0
    seq[5:12, :] = torch.randint(1, 50, (7, T), generator=g)        # an image that never stops
    seqlp = -torch.rand(n, T, generator=g)
    bounds = [0]
    for s_ in sizes:
        bounds.append(bounds[-1] + s_)
    # the former torch post-processing of models/sampling.decode
    alive = (seq > 0).int().cumprod(1)
    row_img = torch.from_numpy(np.repeat(np.arange(len(sizes)), sizes))
    per = torch.zeros(len(sizes), T, dtype=alive.dtype).index_add_(0, row_img, alive)
    stopped = (per == 0).int()
    brk = torch.where(stopped.any(1), stopped.argmax(1), torch.full_like(stopped[:, 0], T - 1).long())
    want = seqlp * (torch.arange(T).view(1, T) <= brk[row_img].view(-1, 1))
    ds, dl = seq.to(DEV), seqlp.to(DEV).clone()
    out = ops.decode_batch_finish(ds, dl, bounds).cpu()
    assert out[:, 0].tolist() == brk.tolist() and out[:, 1].bool().tolist() == stopped.any(1).tolist()
    assert torch.equal(dl.cpu().abs(), want.abs()) and torch.equal(dl.cpu() != 0, want != 0)


def test_word_helpers_and_upload_ring():
    for dt in (torch.float32, torch.int32, torch.int64):
        t = torch.randint(1, 9, (7, 13)).to(dt).to(DEV)
        c = ops.copy_(torch.empty_like(t), t)
        assert torch.equal(c, t)
        assert float(ops.zero_(c).abs().sum()) == 0
    v = torch.arange(5, device=DEV, dtype=torch.int64)
    assert torch.equal(ops.zero_(v), torch.zeros(5, dtype=torch.int64, device=DEV))
    ups = [ops.upload(list(range(i, i + 300)), torch.int64, DEV) for i in range(6)]          # more uploads than staging buffers, no sync between
    torch.cuda.synchronize()
    for i, u_ in enumerate(ups):
        assert u_.cpu().tolist() == list(range(i, i + 300))
    big = ops.upload(list(range(5000)), torch.int32, DEV)                                   # grows the ring
    assert big.cpu().tolist() == list(range(5000))
