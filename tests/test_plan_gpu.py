"""Index plumbing of a training step on the device (csrc/plan.hip) against the torch ops it replaced: the packed decoder's row
plan (stable sort by live steps, early break of AttModel.py:171-172, the criterion's denominator), the packed per-step prefixes,
the sGPN input views (gpn.py:43-52), the per-sentence selection (gpn.py:63-78), and the small helpers.  All integer results exact."""
import numpy as np
import pytest
import torch

from subgc import ops, synthetic
from subgc.functions_packed import Plan

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _torch_plan(labels, mask_t):
    S, T = mask_t.shape
    steps = torch.arange(1, T + 1).view(1, T)
    live = ((mask_t > 0) * steps).amax(1)
    any_tok = (labels[:, :T] != 0).any(0)
    any_tok[0] = True
    live = torch.minimum(live, torch.cumprod(any_tok.to(torch.int64), 0).sum())
    order = torch.sort(live, descending=True, stable=True)
    counts = (order.values.view(1, S) > torch.arange(T).view(T, 1)).sum(1)
    return order.indices, counts, mask_t.sum()


@pytest.mark.parametrize("S,T,seed", [(5, 7, 0), (64, 17, 1), (640, 17, 2), (1280, 17, 3), (1000, 31, 4), (3, 1, 5)])
def test_live_plan_matches_the_torch_restatement(S, T, seed):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(0, T, (S,), generator=g)
    if seed % 2:
        lens = torch.clamp(lens, max=max(T - 4, 0))                     # early break: every label is zero from some step on
    labels = torch.zeros(S, T + 1, dtype=torch.long)
    mask = torch.zeros(S, T + 1)
    for s in range(S):
        n = int(lens[s])
        labels[s, 1:n + 1] = torch.randint(1, 50, (n,), generator=g)
        mask[s, :min(n + 2, T + 1)] = 1.0
    if seed == 4:
        mask[::7, -1] = 1.0                                              # live mask entries past the early break still count in the denominator
    perm, counts, den = _torch_plan(labels, mask[:, 1:])
    pl = Plan(labels.to(DEV), mask[:, 1:].to(DEV))
    assert pl.wait() == counts.tolist()
    np.testing.assert_array_equal(pl.perm.cpu().numpy(), perm.numpy())
    np.testing.assert_array_equal(pl.perm32.cpu().numpy(), perm.numpy())
    np.testing.assert_array_equal(pl.inv32.cpu().numpy()[perm.numpy()], np.arange(S))
    offs = np.concatenate([[0], np.cumsum(counts.numpy())])
    np.testing.assert_array_equal(pl.offs.cpu().numpy(), offs)
    assert float(pl.den) == float(den)
    # packed rows
    N = 9
    idx = torch.randint(0, 30, (S, N), generator=g)
    lens_n = torch.randint(1, N, (S,), generator=g).int()
    img = torch.randint(0, 10, (S,), generator=g).int()
    target = labels[:, 1:]
    lp, tok, tgt, msk, lens_p, idx_p, img_p = ops.packed_rows(labels.to(DEV), target.to(DEV), mask[:, 1:].to(DEV), pl.perm32, pl.offs, lens_n.to(DEV),
                                                            idx.to(DEV), img.to(DEV))
    rows = int(offs[-1])
    M = counts.tolist()
    lab_s = labels[perm]
    want_tok = torch.cat([lab_s[:M[t], t] for t in range(T)])
    want_tgt = torch.cat([target[perm][:M[t], t] for t in range(T)])
    want_msk = torch.cat([mask[:, 1:][perm][:M[t], t] for t in range(T)])
    np.testing.assert_array_equal(tok[:rows].cpu().numpy(), want_tok.numpy())
    np.testing.assert_array_equal(tgt[:rows, 0].cpu().numpy(), want_tgt.numpy())
    np.testing.assert_array_equal(msk[:rows, 0].cpu().numpy(), want_msk.numpy())
    np.testing.assert_array_equal(lp.cpu().numpy(), lab_s.numpy())
    np.testing.assert_array_equal(idx_p.cpu().numpy(), idx[perm].numpy())
    np.testing.assert_array_equal(lens_p.cpu().numpy(), lens_n[perm].numpy())
    np.testing.assert_array_equal(img_p.cpu().numpy(), img[perm].numpy())


@pytest.mark.parametrize("B,hb,N", [(2, 2, 37), (3, 1, 12), (4, 3, 101)])
def test_gpn_prep_and_select_match_the_view_arithmetic(B, hb, N):
    b = synthetic.make_train_batch(B, N=N, K=8, D=8, vocab=20, n_obj_cls=5, n_pred_cls=3, hb=hb, seed=B, max_nodes=min(11, N - 1))
    oi, pm, am = b["gpn_obj_ind"].to(DEV), b["gpn_pool_mtx"].to(DEV), b["att_masks"].to(DEV)
    b5 = oi.size(0)
    G = 2 * b5 * hb
    idx, w, denom, img = ops.gpn_prep(oi, pm, am, b5 // B)
    np.testing.assert_array_equal(idx.cpu().numpy(), oi.permute(1, 0, 2, 3).reshape(G, N).cpu().numpy())
    np.testing.assert_array_equal(w.cpu().numpy(), pm.diagonal(dim1=-2, dim2=-1).permute(1, 0, 2, 3).reshape(G, N).cpu().numpy())
    np.testing.assert_array_equal(denom.cpu().numpy(), am.permute(1, 0, 2, 3).reshape(G, N).sum(1).cpu().numpy())
    want_img = torch.div(torch.arange(b5), b5 // B, rounding_mode="floor").repeat_interleave(hb).repeat(2)
    np.testing.assert_array_equal(img.cpu().numpy(), want_img.numpy())
    torch.manual_seed(B)
    score = torch.randn(G, 1, device=DEV)
    score[0:hb] = 0.25                                                    # a tie: the first max wins
    ro = torch.randn(G, 14, device=DEV)
    sel_idx, lens, ro_sel, img_s = ops.gpn_select(score, oi, am, ro, b5 // B)
    sel = score.view(2, b5, hb)[0].argmax(1) if hb > 1 else torch.zeros(b5, dtype=torch.long, device=DEV)
    sel[0] = 0
    ar = torch.arange(b5, device=DEV)
    np.testing.assert_array_equal(sel_idx.cpu().numpy(), oi[:, 0][ar, sel].cpu().numpy())
    np.testing.assert_array_equal(lens.cpu().numpy(), am[:, 0][ar, sel].sum(1).int().cpu().numpy())
    np.testing.assert_array_equal(ro_sel.cpu().numpy(), ro.view(2, b5, hb, 14)[0][ar, sel].cpu().numpy())
    np.testing.assert_array_equal(img_s.cpu().numpy(), want_img[:b5 * hb:hb].numpy())


def test_add_n_fill2d_row_count_and_argmax_i32():
    torch.manual_seed(0)
    for n in (1, 7, 4096, 100003):
        ts = [torch.randn(n, device=DEV) for _ in range(4)]
        for k in (2, 3, 4):
            want = ts[0].clone()
            for t in ts[1:k]:
                want = want + t
            got = ops.add_n(ts[:k])
            np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-6, atol=1e-6)
    x = torch.zeros(6, 2, 3, 10, device=DEV)
    win = x[:, 0, 0]
    ops.fill2d_(win[:, :7], 1.0)
    assert float(x.sum()) == 42.0 and float(x[:, 0, 0, :7].sum()) == 42.0
    np.testing.assert_array_equal(ops.row_count(win).cpu().numpy(), np.full(6, 7))
    d = torch.rand(50, 33, device=DEV)
    np.testing.assert_array_equal(ops.row_argmax(d, skip=1, i32=True).cpu().numpy(), (d[:, 1:].argmax(1) + 1).int().cpu().numpy())
