"""Two ranks on the one MI355X of the test box (both on cuda:0; `gloo` carries the CUDA tensors because RCCL
refuses two ranks on one device): the REAL model runs LossWrapper fwd+bwd on its image shard through the HIP
path, DecoderFn's backward fires the reducer callback from the autograd thread, the flat bucket is reduced in
five readiness-ordered slices, and both ranks must end with the mean of the per-shard gradients computed in a single process."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

OPT = dict(caption_model="topdown", vocab_size=200, input_encoding_size=64, rnn_size=64, num_layers=1, drop_prob_lm=0.0,
           max_length=20, seq_length=16, fc_feat_size=48, att_feat_size=128, att_hid_size=32, use_bn=0, sampling_prob=0.0,
           use_gpn=1, embed_dim=20, gcn_dim=64, noun_fuse=1, pred_emb_type=1, gcn_layers=2, gcn_residual=2, gcn_bn=0,
           gpn_drop_prob=0.0, obj_name_path=None, rel_name_path=None, sg_obj_cnt=40, sg_pred_cnt=21)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _setup_paths():
    for p in (os.path.join(ROOT, "sub-gc_amd"), ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _model_and_batch():
    import argparse
    from subgc import synthetic
    import subgc.models as models
    torch.manual_seed(3)
    m = models.setup(argparse.Namespace(**OPT))
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "gcn_collect" in n and "weight" in n:
                p.mul_(30.0)
    batch = synthetic.make_train_batch(4, D=128, vocab=200, n_obj_cls=40, seed=11, fc_size=128)
    return m.to("cuda:0").train(), batch, models


def _shard_step(m, models, shard, reducer=None):
    lw = models.LossWrapper(m, None)
    b = {k: v.to(m.flat_params.device) for k, v in shard.items()}
    (reducer.prepare() if reducer is not None else m.flatten_grads())
    out = lw(b["fc_feats"], b["att_feats"], b["labels"], b["masks"], b["att_masks"], None, None, None, b["obj_dist"], None, b["rel_ind"],
             None, b["pred_dist"], b["gpn_obj_ind"], b["gpn_pred_ind"], b["gpn_nrel_ind"], b["gpn_pool_mtx"])
    (out["lang_loss"] + out["gpn_loss"]).backward()
    if reducer is not None:
        # the three decoder slices were sent from inside DecoderFn.backward and the GCN / sGPN slice from the fusion-output marker, in
        # readiness order, before backward returned
        launched_early = [st for st, _ in reducer.issued] == ["logit", "recurrent", "prepare", "gcn"] and len(reducer._pending) == 4
        flat = reducer.finish()
        torch.cuda.synchronize()
        return flat.clone(), launched_early
    torch.cuda.synchronize()
    return m.flat_grads.clone(), False


def _worker(rank, world, port, q):
    _setup_paths()
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from subgc import parallel
    torch.cuda.set_device(0)
    parallel.init_distributed("gloo")
    m, batch, models = _model_and_batch()
    red = parallel.GradBucketReducer(m)
    flat, early = _shard_step(m, models, parallel.shard_batch(batch, rank, world), red)
    red.close()
    q.put((rank, flat.cpu().numpy(), early))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_one_gpu_bucket_reduction_matches_mean_of_shards():
    _setup_paths()
    from subgc import parallel
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=500) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    m, batch, models = _model_and_batch()
    shard_grads = [_shard_step(m, models, parallel.shard_batch(batch, r, world))[0].cpu().numpy() for r in range(world)]
    want = 0.5 * (shard_grads[0] + shard_grads[1])
    scale = float(np.abs(want).max())
    for rank, flat, early in res:
        assert early, "the decoder slices must be in flight before the encoder backward ends"
        np.testing.assert_allclose(flat, want, atol=3e-5 * scale + 1e-8, rtol=1e-4)
    np.testing.assert_array_equal(res[0][1], res[1][1])


def _nccl_worker(port, q):
    """world size 1, backend nccl (= RCCL on ROCm): the REAL Sub_GC_Kar model (280 MB flat bucket), real fwd+bwd, the decoder
    slice all-reduced from DecoderFn.backward's callback while the encoder backward still runs, the encoder slice at the end."""
    _setup_paths()
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import argparse
    import torch.distributed as dist
    from subgc import parallel, synthetic
    import subgc.models as models
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    kar = dict(OPT, vocab_size=9487, input_encoding_size=1000, rnn_size=1000, fc_feat_size=2048, att_feat_size=2048, att_hid_size=512,
               embed_dim=300, gcn_dim=1024, sg_obj_cnt=1599, drop_prob_lm=0.5, gpn_drop_prob=0.5)
    torch.manual_seed(3)
    m = models.setup(argparse.Namespace(**kar)).to("cuda:0").train()
    batch = synthetic.make_train_batch(8, seed=11)
    m.injected_masks = None
    m._dropout_calls = 0
    plain, _ = _shard_step(m, models, batch)                        # no reducer: the reference gradients (same dropout stream below)
    m._dropout_calls = 0
    red = parallel.GradBucketReducer(m, always_reduce=True)
    assert red.active and red.overlap
    from subgc import functions as F_
    F_.trace = []                                                   # host-order record of the backward's phases and of every collective's issue point
    flat, early = _shard_step(m, models, batch, red)
    trace, F_.trace = F_.trace, None
    issued = list(red.issued)
    # a second step through the same reducer, then the fused optimizer sweep on the reduced bucket (stream order: RCCL -> Adam)
    adam = parallel.FlatAdam(m)
    before = m.flat_params.clone()
    _shard_step(m, models, batch, red)
    adam.step(grad_scale=1.0)
    torch.cuda.synchronize()
    moved = float((m.flat_params - before).abs().max())
    red.close()
    q.put((flat.numel() * 4, bool(early), float((flat - plain).abs().max()), float(plain.abs().max()), moved, dist.get_backend(), trace, issued))
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_rccl_world_size_one_pushes_the_real_280mb_bucket():
    """RCCL itself (backend "nccl") on the one-GPU box: loads librccl, creates the communicator, all-reduces the real flat
    bucket in two slices on RCCL's stream ordered against the backward kernels; one rank => the result equals the plain backward."""
    _setup_paths()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_worker, args=(_free_port(), q))
    p.start()
    nbytes, early, err, scale, moved, backend, trace, issued = q.get(timeout=800)
    p.join(120)
    assert p.exitcode == 0
    assert backend == "nccl" and nbytes > 270e6
    assert early, "the decoder slices must be in flight before the encoder backward ends"
    # WHEN each collective is issued relative to the BPTT loop (the collective is enqueued behind the kernels already on the compute
    # stream, so host issue order == earliest device start order): logit.* (38 MB) goes out BEFORE the first step of the loop is
    # enqueued, the recurrent slice right after the loop's weight-gradient products, the prepare slice before the encoder backward
    at = lambda *ev: trace.index(ev)
    steps = [e for e in trace if e[0] == "bptt_begin"][0][1]
    assert steps >= 2
    assert (at("ready", "logit") < at("issue", "logit") < at("bptt_begin", steps) < at("bptt_end") < at("issue", "recurrent") < at("issue", "prepare")
            < at("ready", "gcn") < at("issue", "gcn"))
    assert at("issue", "recurrent") == at("ready", "recurrent") + 1 and at("issue", "prepare") == at("ready", "prepare") + 1
    assert [st for st, _ in issued] == ["logit", "recurrent", "prepare", "gcn", "fusion"]
    sizes = dict(issued)
    assert sum(sizes.values()) == nbytes and sizes["logit"] > 37e6 and sizes["recurrent"] > 150e6 and sizes["fusion"] < 15e6      # every byte of the bucket travels exactly once
    assert err <= 1e-4 * scale + 1e-12, (err, scale)        # two runs of the backward differ by fp32 atomic order only
    assert 0 < moved < 1e-2


def _rccl2_worker(rank, port, q):
    """Two ranks on TWO devices over the real `nccl` (= RCCL) backend: the golden model, one shard each, readiness-ordered buckets."""
    _setup_paths()
    os.environ.update(RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from conftest import Golden
    from subgc import parallel, synthetic
    import subgc.models as models
    r, local, w = parallel.init_distributed("nccl")
    g = Golden("subgc_train")
    opt = g.opt(caption_model="topdown", gpn_drop_prob=0.0)
    m = models.setup(opt)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in g.group("weights").items()})
    m = m.to(f"cuda:{local}").train()
    batch = synthetic.make_train_batch(4, D=opt.att_feat_size, vocab=opt.vocab_size, seed=9, fc_size=opt.att_feat_size)
    plain, _ = _shard_step(m, models, parallel.shard_batch(batch, rank, 2))
    red = parallel.GradBucketReducer(m, timing=True)
    flat, early = _shard_step(m, models, parallel.shard_batch(batch, rank, 2), red)
    torch.cuda.synchronize()
    rep = red.report()
    red.close()
    q.put((rank, plain.cpu().numpy(), flat.cpu().numpy(), bool(early), dist.get_backend(), rep))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_rank_rccl_on_two_devices():
    """The real thing where the box has it: two processes, two GPUs, backend nccl (RCCL over xGMI).  Skipped on the one-GPU test box
    (tests/test_parallel_cpu.py covers the same reducer with two gloo ranks; the world-size-1 test above covers RCCL itself)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible devices")
    _setup_paths()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rccl2_worker, args=(r, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=800) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    (_, p0, f0, e0, be, rep0), (_, p1, f1, e1, _, rep1) = res
    assert be == "nccl" and e0 and e1
    np.testing.assert_array_equal(f0, f1)
    scale = float(np.abs(p0).max())
    np.testing.assert_allclose(f0, 0.5 * (p0 + p1), atol=3e-5 * scale + 1e-8, rtol=1e-4)
    for rep in (rep0, rep1):
        assert [b["stage"] for b in rep["buckets"]] == ["logit", "recurrent", "prepare", "gcn", "fusion"] and rep["exposed_ms"] >= 0


def test_one_rank_keeps_the_whole_buffer_norm_pass():
    """GradBucketReducer(optimizer=FlatAdam) on ONE rank: there is no collective to hide a slice's norm behind, and on the one compute
    stream five small norm launches cost more than one pass over the buffer (measured 5 x 36 us against 65-74) -- the whole-buffer pass in
    FlatAdam.step stays, one sumsq launch, the same parameters as without a reducer."""
    _setup_paths()
    from subgc import ops, parallel
    m, batch, models = _model_and_batch()
    start = m.flat_params.detach().clone()
    res = {}
    seen = []
    real = ops.sumsq
    for attached in (True, False):
        m.flat_params.data.copy_(start)
        m.invalidate_decode_caches()
        adam = parallel.FlatAdam(m, lr=1e-2, clip_norm=0.05)            # small enough that the clip is ACTIVE: the norm matters
        red = parallel.GradBucketReducer(m, optimizer=adam if attached else None)
        assert not red.active
        seen.clear()
        ops.sumsq = lambda g, out: (seen.append(g.numel()), real(g, out))[1]
        try:
            lw = models.LossWrapper(m, None)
            b = {k: v.to("cuda:0") for k, v in batch.items()}
            red.prepare()
            out = lw(b["fc_feats"], b["att_feats"], b["labels"], b["masks"], b["att_masks"], None, None, None, b["obj_dist"], None, b["rel_ind"],
                     None, b["pred_dist"], b["gpn_obj_ind"], b["gpn_pred_ind"], b["gpn_nrel_ind"], b["gpn_pool_mtx"])
            (out["lang_loss"] + out["gpn_loss"]).backward()
            red.finish(average=False)
            adam.step(grad_scale=1.0)
            torch.cuda.synchronize()
        finally:
            ops.sumsq = real
            red.close()
        res[attached] = (m.flat_params.detach().clone(), list(seen), float(adam.sumsq))
    (pa, seen_a, na), (pu, seen_u, nu) = res[True], res[False]
    assert seen_a == [m.flat_params.numel()] and seen_u == [m.flat_params.numel()]
    assert abs(na - nu) <= 1e-5 * nu and nu ** 0.5 > 0.05                                      # the clip was active
    # (the two backward passes differ in the last bits -- embed_bwd / pool_bwd accumulate with float atomics -- and Adam's first step turns a
    # near-zero gradient's last bit into a visible fraction of lr = 1e-2: measured 9e-6)
    assert float((pa - pu).abs().max()) <= 1e-4 and float((pu - start).abs().max()) > 1e-3


def _nccl_adam_worker(port, q):
    """world size 1, backend nccl, `always_reduce`: the code path of an N > 1 run with the optimizer attached -- every slice's clip-norm
    share is accumulated on the reducer's side stream behind ITS collective's work handle, the compute stream joins it in finish()."""
    _setup_paths()
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from subgc import ops, parallel
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    m, batch, models = _model_and_batch()
    start = m.flat_params.detach().clone()
    res = {}
    real = ops.sumsq
    for attached in (True, False):
        m.flat_params.data.copy_(start)
        m.invalidate_decode_caches()
        adam = parallel.FlatAdam(m, lr=1e-2, clip_norm=0.05)            # the clip is active: a wrong or unfinished norm moves the parameters
        red = parallel.GradBucketReducer(m, always_reduce=True, optimizer=adam if attached else None)
        seen = []
        ops.sumsq = lambda g, out: (seen.append((g.numel(), torch.cuda.current_stream().cuda_stream)), real(g, out))[1]
        try:
            for _ in range(2):                                          # two steps: the accumulator is re-armed by prepare()
                seen.clear()
                _shard_step_sum(m, models, batch, red)
                adam.step(grad_scale=1.0)
            torch.cuda.synchronize()
        finally:
            ops.sumsq = real
        side = red._side.cuda_stream if red._side is not None else None
        res[attached] = (m.flat_params.detach().cpu(), list(seen), float(adam.sumsq), side, [hi - lo for _, lo, hi in red.buckets if hi > lo])
        red.close()
    q.put((res, torch.cuda.current_stream().cuda_stream, dist.get_backend(), float((res[False][0] - start.cpu()).abs().max())))
    dist.destroy_process_group()


def _shard_step_sum(m, models, shard, reducer):
    lw = models.LossWrapper(m, None)
    b = {k: v.to(m.flat_params.device) for k, v in shard.items()}
    reducer.prepare()
    out = lw(b["fc_feats"], b["att_feats"], b["labels"], b["masks"], b["att_masks"], None, None, None, b["obj_dist"], None, b["rel_ind"],
             None, b["pred_dist"], b["gpn_obj_ind"], b["gpn_pred_ind"], b["gpn_nrel_ind"], b["gpn_pool_mtx"])
    (out["lang_loss"] + out["gpn_loss"]).backward()
    return reducer.finish(average=False)


@pytest.mark.timeout(600)
def test_rccl_slices_carry_their_clip_norm_on_the_side_stream():
    """The N > 1 step tail over the real `nccl` backend (world size 1 on the one-GPU box, `always_reduce`): with the optimizer attached
    each of the five slices adds its squared norm right behind its own collective on the reducer's side stream, FlatAdam.step finds all
    of them in and launches no norm pass of its own; the parameters after two clipped steps equal those of the detached order
    (collectives, then one whole-buffer norm pass)."""
    _setup_paths()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_adam_worker, args=(_free_port(), q))
    p.start()
    res, main_stream, backend, moved = q.get(timeout=500)
    p.join(120)
    assert p.exitcode == 0 and backend == "nccl"
    (pa, seen_a, na, side, sizes), (pu, seen_u, nu, side_u, _) = res[True], res[False]
    assert side is not None and side != main_stream and side_u is None
    assert [n for n, _ in seen_a] == sizes and all(st == side for _, st in seen_a)         # five slice passes, all on the side stream
    assert len(seen_u) == 1 and seen_u[0][0] == sum(sizes) and seen_u[0][1] == main_stream    # detached: one pass over the whole buffer
    assert abs(na - nu) <= 1e-5 * nu and nu ** 0.5 > 0.05
    assert float((pa - pu).abs().max()) <= 2e-4 and moved > 1e-3
