"""N>1 path on CPU: two `gloo` ranks shard one batch by image, compute their shard's gradients
(with the CPU oracle standing in for the device compute), all-reduce the flat gradient bucket with
GradBucketReducer (the three decoder slices from the post-accumulate hooks, the encoder slices at the end) and must
end up with the mean of the per-shard gradients - DataParallel's semantics (train.py:96-98,154-156)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _shard_grads(opt, weights, batch, rank, world):
    from oracle import subgc_oracle as O
    from subgc import parallel
    orc = O.Oracle(opt, weights, requires_grad=True)
    orc.training = True
    shard = parallel.shard_batch(batch, rank, world)
    out = O.loss_wrapper(orc, shard)
    (out["lang_loss"] + out["gpn_loss"]).backward()
    return {k: (p.grad if p.grad is not None else None) for k, p in orc.P.items()}


def _worker(rank, world, port, q):
    for p in (os.path.join(ROOT, "sub-gc_amd"), ROOT, os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from conftest import Golden
    from subgc import parallel, synthetic
    import subgc.models as models
    r, _, w = parallel.init_distributed("gloo")
    assert (r, w) == (rank, world)
    g = Golden("subgc_train")
    opt = g.opt(caption_model="topdown", gpn_drop_prob=0.0)
    weights = g.group("weights")
    model = models.setup(opt)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})
    batch = synthetic.make_train_batch(4, D=opt.att_feat_size, vocab=opt.vocab_size, seed=9, fc_size=opt.att_feat_size)
    red = parallel.GradBucketReducer(model)
    red.prepare()
    grads = _shard_grads(opt, weights, batch, rank, world)
    # feed this rank's gradients through autograd so the post-accumulate hooks fire like in a real backward;
    # decoder parameters first (that is the order a real backward produces them in)
    names = [n for n, _ in model.named_parameters() if grads.get(n) is not None]
    names.sort(key=lambda n: -model._slots[n][0])
    loss = sum((model.P(n) * grads[n]).sum() for n in names)
    loss.backward()
    # the three decoder slices (AttModel.grad_buckets) are already in flight, each sent when its last gradient arrived
    assert sorted(st for st, _ in red.issued) == ["logit", "prepare", "recurrent"] and len(red._pending) == 3
    assert all(red._fired[st] == red._need[st] for st in ("logit", "recurrent", "prepare"))
    flat = red.finish().clone()
    red.close()
    q.put((rank, flat.numpy(), {k: (None if v is None else v.numpy()) for k, v in grads.items()}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_bucket_allreduce_is_mean_of_shard_grads():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=240) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, flat0, g0), (_, flat1, g1) = res
    np.testing.assert_array_equal(flat0, flat1)                          # identical averaged bucket on both ranks
    sys.path.insert(0, os.path.join(ROOT, "sub-gc_amd"))
    from conftest import Golden
    import subgc.models as models
    g = Golden("subgc_train")
    model = models.setup(g.opt(caption_model="topdown"))
    seen_dead = 0
    for name, (o, n, shape) in model._slots.items():
        got = flat0[o:o + n].reshape(shape)
        if g0[name] is None:
            assert g1[name] is None and float(np.abs(got).max()) == 0.0     # dead parameter: zeros everywhere
            seen_dead += 1
        else:
            np.testing.assert_allclose(got, 0.5 * (g0[name] + g1[name]), rtol=1e-6, atol=1e-7, err_msg=name)
    assert seen_dead == len(g.meta["dead_params"])


def _skew_worker(rank, world, port, q):
    """Rank 0 announces every slice like a normal backward; rank 1 never announces `recurrent` (a marker that did not fire) and
    announces `prepare` and `gcn` anyway.  Both must issue the SAME collective sequence (canonical order, same sizes)."""
    for p in (os.path.join(ROOT, "sub-gc_amd"), ROOT, os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from conftest import Golden
    from subgc import parallel
    from subgc import functions as F_
    import subgc.models as models
    parallel.init_distributed("gloo")
    model = models.setup(Golden("subgc_train").opt(caption_model="topdown"))
    red = parallel.GradBucketReducer(model)
    flat = red.prepare()
    torch.manual_seed(rank)
    flat.copy_(torch.randn_like(flat))
    mine = flat.clone()
    order = ["logit", "recurrent", "prepare", "gcn"] if rank == 0 else ["logit", "prepare", "gcn"]
    early = []
    for st in order:
        F_.grads_ready(st)
        early.append([s for s, _ in red.issued])
    out = red.finish().clone()
    red.close()
    q.put((rank, [s for s, _ in red.issued], [b for _, b in red.issued], early, mine.numpy(), out.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_collective_sequence_is_rank_invariant_when_one_rank_misses_an_announcement():
    """Advisor finding (round 4): which slices went out early is rank-local; the issue ORDER and SIZES must not be."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_skew_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=240) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, seq0, by0, early0, g0, o0), (_, seq1, by1, early1, g1, o1) = res
    canon = ["logit", "recurrent", "prepare", "gcn", "fusion"]
    assert seq0 == canon and seq1 == canon and by0 == by1
    assert early0[-1] == canon[:4]                                      # rank 0 sent four slices from inside the backward
    assert early1[-1] == ["logit"]                                      # rank 1 held `prepare` / `gcn` back behind the missing `recurrent`
    np.testing.assert_array_equal(o0, o1)
    np.testing.assert_allclose(o0, 0.5 * (g0 + g1), rtol=1e-6, atol=1e-7)


def test_grad_buckets_are_contiguous_readiness_ordered_and_cover_the_flat_buffer():
    """AttModel.grad_buckets: logit -> recurrent -> prepare -> gcn -> fusion, contiguous, disjoint, covering every parameter slot, and
    every parameter lies in the slice its name says (the decoder Functions announce the slices by these names)."""
    sys.path.insert(0, os.path.join(ROOT, "sub-gc_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import Golden
    import subgc.models as models
    from subgc import functions as F_
    for gname in ("subgc_train", "fullgc_train"):
        model = models.setup(Golden(gname).opt(caption_model="topdown"))
        b = model.grad_buckets()
        assert [st for st, _, _ in b] == ["logit", "recurrent", "prepare", "gcn", "fusion"]
        spans = sorted((lo, hi) for _, lo, hi in b)
        assert spans[0][0] == 0 and spans[-1][1] == model.flat_params.numel()
        assert all(spans[i][1] == spans[i + 1][0] for i in range(4))
        where = {st: (lo, hi) for st, lo, hi in b}
        stage_of = lambda n: ("logit" if n.startswith("logit.") else
                              "prepare" if n.split(".")[0] in ("fc_embed", "att_embed", "ctx2att") else
                              "recurrent" if n in F_.PARAM_ORDER else
                              "gcn" if n.split(".")[0] in ("gcn_backbone", "gpn_layer", "read_out_proj") else "fusion")
        for n, (o, cnt, _) in model._slots.items():
            lo, hi = where[stage_of(n)]
            assert lo <= o and o + cnt <= hi, n
        assert where["fusion"][0] == 0 and where["gcn"][1] == model.decoder_offset and where["fusion"][1] == where["gcn"][0]


def test_shard_batch_splits_every_leading_dim():
    sys.path.insert(0, os.path.join(ROOT, "sub-gc_amd"))
    from subgc import parallel, synthetic
    b = synthetic.make_train_batch(8, D=16, vocab=20, n_obj_cls=10, seed=0)
    parts = [parallel.shard_batch(b, r, 4) for r in range(4)]
    for k, v in b.items():
        torch.testing.assert_close(torch.cat([p[k] for p in parts], 0), v)
    assert parts[0]["att_feats"].size(0) == 2 and parts[0]["labels"].size(0) == 10
    with pytest.raises(ValueError):
        parallel.shard_batch(b, 0, 3)


def _adam_worker(rank, world, port, q):
    """Per-slice clip norm + one sweep (GradBucketReducer(optimizer=FlatAdam)) on two gloo ranks.  The device kernels are replaced by
    torch stand-ins IN THIS TEST (the product's ops refuse CPU tensors): what is under test is the host logic -- which slices are
    summed when, that every slice is counted exactly once, that both ranks end with identical parameters."""
    for p in (os.path.join(ROOT, "sub-gc_amd"), ROOT, os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from conftest import Golden
    from subgc import ops, parallel
    from subgc import functions as F_
    import subgc.models as models
    calls = []

    def fill_(t, v):
        t.fill_(v); return t

    def sumsq(g, out):
        calls.append(("sumsq", g.data_ptr(), g.numel()))
        out += (g.double() ** 2).sum().float(); return out

    def clip_adam_step(p, g, m, v, ss, max_norm, lr, b1, b2, eps, wd, step, grad_scale=1.0, p_bf16=None, zero_grad=False):
        calls.append(("sweep", float(ss)))
        coef = grad_scale * (max_norm / max(float(ss.sqrt()) * grad_scale, max_norm))
        gi = g * coef
        m.mul_(b1).add_(gi, alpha=1 - b1); v.mul_(b2).addcmul_(gi, gi, value=1 - b2)
        p.sub_(lr / (1 - b1 ** step) * m / ((v / (1 - b2 ** step)).sqrt() + eps))
        g.zero_() if zero_grad else g.copy_(gi)

    ops.fill_, ops.sumsq, ops.clip_adam_step = fill_, sumsq, clip_adam_step
    parallel.init_distributed("gloo")
    model = models.setup(Golden("subgc_train").opt(caption_model="topdown"))
    model.invalidate_decode_caches = lambda: None
    torch.manual_seed(0)
    start = torch.randn_like(model.flat_params) * 0.1
    model.flat_params.data.copy_(start)
    results = {}
    for bucketed in (True, False):
        model.flat_params.data.copy_(start)
        adam = parallel.FlatAdam(model, lr=1e-2, clip_norm=0.5)
        red = parallel.GradBucketReducer(model, optimizer=adam if bucketed else None)
        calls.clear()
        flat = red.prepare()
        torch.manual_seed(10 + rank)
        flat.copy_(torch.randn_like(flat))                      # this rank's gradients
        for st in ("logit", "recurrent", "prepare", "gcn"):
            F_.grads_ready(st)
        red.finish(average=False)
        adam.step(grad_scale=1.0 / world, zero_grad=True)
        results[bucketed] = (model.flat_params.detach().clone(), list(calls), [st for st, _ in red.issued], sorted(b[0] for b in red.buckets))
        red.close()
    q.put((rank, results[True][0].numpy(), results[False][0].numpy(), results[True][1], results[False][1], results[True][2], results[True][3]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_per_slice_clip_norm_gives_the_same_parameters_on_two_gloo_ranks():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_adam_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=240) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, pb0, pu0, cb0, cu0, issued, stages), (_, pb1, pu1, cb1, cu1, _, _) = res
    np.testing.assert_array_equal(pb0, pb1)                              # identical parameters on both ranks
    np.testing.assert_allclose(pb0, pu0, rtol=0, atol=2e-6)             # ... and the whole-buffer norm's (fp32 summation order only)
    assert float(np.abs(pb0 - pu0).max()) < 2e-6 and float(np.abs(pu0).max()) > 0.05
    assert issued == ["logit", "recurrent", "prepare", "gcn", "fusion"]
    # bucketed: one sumsq per non-empty slice, each exactly once, covering the buffer; then ONE sweep.  unbucketed: one sumsq, one sweep
    sums = [c for c in cb0 if c[0] == "sumsq"]
    assert len(sums) == len(stages) and [c[0] for c in cb0].count("sweep") == 1 and cb0[-1][0] == "sweep"
    assert sum(c[2] for c in sums) == pb0.size and len({c[1] for c in sums}) == len(sums)
    assert [c[0] for c in cu0] == ["sumsq", "sweep"] and cu0[0][2] == pb0.size
    assert abs(cb0[-1][1] - cu0[-1][1]) <= 1e-5 * cu0[-1][1]           # the same squared norm went into the sweep
