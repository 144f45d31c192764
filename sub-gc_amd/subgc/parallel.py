"""Data parallelism for the Sub-GC train step: one process per GPU, images sharded across ranks,
ONE flat fp32 gradient bucket all-reduced over RCCL/xGMI.

Replaces reference `train.py:96-98` (`nn.DataParallel`: per-iteration parameter broadcast of 280 MB,
input scatter, loss gather, gradient reduce-add to GPU 0).  Here parameters are replicated once,
every rank runs fwd+bwd on its own shard and the gradients - which already live contiguously in
`model.flat_grads` - are summed in five readiness-ordered collectives (GradBucketReducer):

  * logit.* (final before the BPTT loop starts) overlaps the whole recurrent backward;
  * the recurrent slice (LSTMs, h2att, alpha_net, word embedding: 54 % of the bytes) is sent when the loop's
    batched weight-gradient products are enqueued and overlaps the prepare-feature and encoder backward;
  * the prepare-feature slice follows, the GCN / sGPN slice when the gradient reaches the fusion outputs, and the
    fusion projections' slice (13 MB) when backward returns.

Averaging over ranks reproduces DataParallel's mean of per-replica losses (train.py:154-156);
BatchNorm statistics stay per rank, as they do under DataParallel.  Parameters that receive no
gradient (dead GCN units) contribute zeros identically on every rank.

Decode (reference test.py:184-185 -> eval_utils.py:98-104, one image per call on one GPU) shards the image
list round-robin (`shard_images`), every rank decodes its share with no collective on the way, and one
`all_gather_object` at the end collects token ids / log-probs / scores / kept indices (`sample_images_sharded`,
`eval_glue.caption_images`).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_* (torchrun env)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def shard_batch(batch, rank, world, sentences_per_image=5):
    """Split every tensor of a loader batch along dim 0 (all have leading dim B or 5B, which is what
    DataParallel's scatter relies on too)."""
    if world == 1:
        return batch
    B = batch["att_feats"].size(0)
    if B % world:
        raise ValueError(f"batch of {B} images does not split over {world} ranks")
    out = {}
    for k, v in batch.items():
        if not torch.is_tensor(v):
            out[k] = v
            continue
        per = v.size(0) // world
        out[k] = v[rank * per:(rank + 1) * per]
    return out


def shard_images(items, rank, world):
    """Decode-side sharding: image i of a list goes to rank i % world (round robin keeps the ranks level when the
    list is sorted by candidate count).  -> (this rank's items, their indices into `items`)."""
    idx = list(range(rank, len(items), world))
    return [items[i] for i in idx], idx


def gather_by_index(local, idx, total, group=None):
    """The ONE exchange of a sharded decode: every rank hands in its results (any picklable objects: tensors travel as host
    tensors) with their positions in the un-sharded list and gets the complete list back, in the original order."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        out = [None] * total
        for i, r in zip(idx, local):
            out[i] = r
        return out
    host = lambda o: o.cpu() if torch.is_tensor(o) else (type(o)(host(x) for x in o) if isinstance(o, (tuple, list)) else
                                                        ({k: host(v) for k, v in o.items()} if isinstance(o, dict) else o))
    parts = [None] * world
    dist.all_gather_object(parts, (list(idx), [host(r) for r in local]), group=group)
    out = [None] * total
    for ids, res in parts:
        for i, r in zip(ids, res):
            out[i] = r
    if any(r is None for r in out):
        raise RuntimeError("sharded decode: some images were decoded by no rank")
    return out


@torch.no_grad()
def sample_images_sharded(model, images, opt=None, group_size=256, group=None):
    """Decode a list of loader items on all ranks (reference: one image per call on one GPU, misc/eval_utils.py:98-104, driven
    by test.py:184-185): images round-robin across the ranks, each rank decodes its share through `model.sample_images`
    (`group_size` images per decode batch), no collective on the way, token ids / log-probs / scores / kept indices gathered
    once at the end.  Every rank returns the full per-image list of `_sample` tuples (host tensors when world > 1)."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    mine, idx = shard_images(images, rank, world)
    local = []
    for i in range(0, len(mine), group_size):
        local.extend(model.sample_images(mine[i:i + group_size], opt=dict(opt or {})))
    return gather_by_index(local, idx, len(images), group)


class GradBucketReducer:
    """All-reduce `model.flat_grads` in READINESS-ORDERED buckets (AttModel.grad_buckets): every slice of the flat gradient buffer
    is summed over the ranks the moment the backward has finished writing it, while the rest of the backward still runs --

        logit      (38 MB at Sub_GC_Kar)  final right after the criterion backward, BEFORE the BPTT loop: overlaps the whole loop
        recurrent  (152 MB: both LSTMs, h2att, alpha_net, the word embedding)  final after the loop's batched weight-gradient products
        prepare    (31 MB: fc_embed, att_embed, ctx2att)  final after the prepare-feature backward
        gcn        (46 MB: GCN units, sGPN / read-out layers)  final when the gradient reaches the fusion outputs (functions.StageMark)
        fusion     (13 MB: obj_v_proj, class embeddings and their projections)  final when backward returns (`finish`)

    The decoder Functions announce a slice through functions.on_grads_ready(stage) (they write their gradients straight into the
    bucket, so no autograd hook fires for them); the post-accumulate-grad hooks cover the generic autograd path.  Each collective is
    enqueued behind the kernels already on the compute stream (torch.distributed's stream hand-off) and runs on RCCL's own stream.
    Reference semantics: nn.DataParallel's gradient reduce-add, train.py:96-98,154-164."""

    def __init__(self, model, group=None, overlap=True, always_reduce=False, timing=False, optimizer=None):
        """`always_reduce`: issue the collectives even in a one-rank group (a one-GPU box can then exercise RCCL itself and
        the stream ordering between the backward kernels and the all-reduce; the sum over one rank is the identity).
        `timing`: bracket the step with HIP events on the compute stream (`report()`): when each slice's collective could start,
        and how long the compute stream then WAITED for the collectives after the backward's last kernel = the exposed communication.
        `optimizer` (a FlatAdam): with several ranks the squared gradient norm its global-norm clip needs is accumulated BUCKET BY BUCKET
        as each slice becomes final -- behind the slice's all-reduce, on a side
        stream, so neither the backward nor the later collectives wait for it -- instead of one pass over the whole 280 MB buffer after
        the last collective.  (One rank has nothing to hide it behind: the whole-buffer pass in `FlatAdam.step` stays.)  What cannot move is the clip + Adam sweep itself: the clip coefficient is a function of the norm of ALL
        reduced gradients (misc/utils.py:174-200 computes the total norm before it scales any gradient), so no parameter can be
        updated before the last slice has been reduced; the exposed tail of a step is therefore the last (fusion, 13 MB) collective,
        that slice's norm and the sweep."""
        self.model, self.group = model, group
        self.optimizer = optimizer
        self._side = None
        self.timing = bool(timing)
        self._ev = None
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (always_reduce and dist.is_initialized())
        self.buckets = model.grad_buckets()                     # [(stage, lo, hi)], readiness order, the fusion layers last
        self.split = model.decoder_offset
        self.overlap = overlap and self.active
        self._pending, self._launched, self._fired, self._announced = [], [], {}, set()
        self._handles = []
        self._slices = []
        self.issued = []                                        # (stage, bytes) of every collective of the current step, in issue order
        self._need = {}
        if optimizer is not None:
            optimizer.reducer = self
        if self.overlap or (optimizer is not None and overlap):
            from . import functions as F_
            F_.on_grads_ready = self._ready
            early = [(st, lo, hi) for st, lo, hi in self.buckets if st != "fusion"]
            for n, p in model.named_parameters():
                o = model._slots[n][0]
                for st, lo, hi in early:
                    if lo <= o < hi:
                        self._need[st] = self._need.get(st, 0) + 1
                        self._handles.append(p.register_post_accumulate_grad_hook(lambda _p, st=st: self._hook(st)))

    def _hook(self, stage):
        self._fired[stage] = self._fired.get(stage, 0) + 1
        if self._fired[stage] == self._need[stage]:
            self._ready(stage)

    def _issue(self, stage, lo, hi):
        from . import functions as F_
        g = self.model.flat_grads[lo:hi]
        F_.note("issue", stage)
        self.issued.append((stage, 4 * (hi - lo)))
        if self._ev is not None:                                # fires when every kernel enqueued so far is done: the collective may start
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self._ev["issue"].append(e)
        work = dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._pending.append(work)
        self._slices.append((stage, lo, hi))
        if self.optimizer is not None and g.is_cuda:
            # the slice's share of the clip norm behind ITS collective, beside everything else (the compute stream never waits for it
            # before `finish`; a host-side backend -- gloo on CPU tensors -- has no stream to put it on: `finish` adds it after the wait)
            if self._side is None:
                self._side = torch.cuda.Stream(device=g.device)
            with torch.cuda.stream(self._side):
                work.wait()
                self.optimizer.accumulate(stage, lo, hi)

    def _ready(self, stage):
        """The gradient slice `stage` is final.  Its all-reduce starts now IF every slice before it in the canonical order
        (`self.buckets`) is already out; otherwise it is only marked and goes out when its predecessors have (or in `finish`).
        RCCL pairs collectives across ranks by issue order, so the sequence has to be rank-invariant: always the canonical one,
        whatever subset of the announcements a rank's backward happened to make (advisor finding, round 4)."""
        if self.model.flat_grads is None or stage in self._launched:
            return
        if not self.active:                                     # one rank: no collective to hide a slice's norm behind -- on the one compute stream five
            return                                              # small norm launches cost MORE than one pass over the buffer (measured: 5 x 36 us vs 65-74)
        self._announced.add(stage)
        for st, lo, hi in self.buckets:
            if st in self._launched:
                continue
            if st not in self._announced or st == "fusion":      # the fusion slice is final only when backward returns
                break
            self._launched.append(st)
            if hi > lo:
                self._issue(st, lo, hi)

    def prepare(self):
        """Call before forward: (re)binds every .grad into the zeroed flat bucket."""
        self._fired, self._pending, self._launched, self.issued, self._announced = {}, [], [], [], set()
        self._slices = []
        flat = self.model.flatten_grads()
        if self.optimizer is not None:
            self.optimizer.begin_step()
        if self.timing and self.active and flat.is_cuda:
            self._ev = {"start": torch.cuda.Event(enable_timing=True), "issue": [], "bwd_end": None, "waited": []}
            self._ev["start"].record()
        return flat

    def finish(self, average=True):
        """Call after loss.backward(): reduces whatever has not been sent yet (the fusion slice; every slice when nothing
        overlapped), waits for all of it and averages (`average=False`: leave the SUM and hand `1 / world` to
        `FlatAdam.step(grad_scale=...)`, which folds it into its own sweep).

        The collective SEQUENCE must be the same on every rank (RCCL matches collectives by issue order, not by name): which
        buckets went out early is rank-local state (a marker that never fired on one rank -- no gradient reached it, an early
        return in a backward), so the leftovers are issued ONE BY ONE in the canonical `self.buckets` order, never merged and
        never re-sorted -- every rank then issues the five slices with the same sizes; a rank that announced a slice early has
        merely issued it sooner.  With overlap on, the early order IS the canonical order (logit, recurrent, prepare, gcn), so a
        rank that missed an announcement still lines up (`_ready` holds a slice back until its predecessors are out)."""
        if not self.active:
            return self.model.flat_grads
        if self.optimizer is not None and average and self.world > 1:
            raise RuntimeError("an attached optimizer accumulates the norm of the SUMMED slices: call finish(average=False) and hand 1 / world "
                               "to FlatAdam.step(grad_scale=...)")
        g = self.model.flat_grads
        for st, lo, hi in self.buckets:
            if st not in self._launched and hi > lo:
                self._launched.append(st)
                self._issue(st, lo, hi)
        if self._ev is not None:                                # the backward's last kernel (and the last slice's issue point)
            self._ev["bwd_end"] = torch.cuda.Event(enable_timing=True)
            self._ev["bwd_end"].record()
        for w, (st, lo, hi) in zip(self._pending, self._slices):
            w.wait()
            if self.optimizer is not None and not g.is_cuda:    # host-side backend: the slice's norm right behind its (blocking) wait
                self.optimizer.accumulate(st, lo, hi)
            if self._ev is not None:                            # the compute stream is past this collective
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                self._ev["waited"].append(e)
        self._pending, self._slices = [], []
        if self._side is not None:                              # the per-slice norms accumulated beside the backward
            torch.cuda.current_stream(g.device).wait_stream(self._side)
        if average and self.world > 1:
            g.mul_(1.0 / self.world)
        return g

    def report(self):
        """(timing=True; call after a device synchronisation) -> {"buckets": [{stage, bytes, issue_ms, done_by_ms}], "backward_end_ms",
        "exposed_ms"}: times on the compute stream since `prepare()`.  issue_ms = when the slice's last gradient kernel had finished
        (the collective's earliest start); done_by_ms = when the compute stream got past the wait for it; exposed_ms = what the step
        paid for communication = end of the last wait - end of the backward's own kernels."""
        ev = self._ev
        if ev is None or ev["bwd_end"] is None:
            return None
        t = lambda e: round(ev["start"].elapsed_time(e), 3)
        rows = [{"stage": st, "bytes": nb, "issue_ms": t(ei), "done_by_ms": t(ew)} for (st, nb), ei, ew in zip(self.issued, ev["issue"], ev["waited"])]
        end = t(ev["bwd_end"])
        rep = {"buckets": rows, "backward_end_ms": end, "exposed_ms": round((rows[-1]["done_by_ms"] if rows else end) - end, 3)}
        if self.optimizer is not None:
            rep["optimizer_tail"] = ("clip norm accumulated per slice behind its collective (side stream); after the last collective: the fusion "
                                     "slice's norm + ONE clip/Adam sweep (the clip coefficient needs the norm of ALL reduced slices, so no sweep can "
                                     "start earlier)")
        return rep

    def close(self):
        from . import functions as F_
        if F_.on_grads_ready == self._ready:
            F_.on_grads_ready = None
        for h in self._handles:
            h.remove()
        self._handles = []


class FlatAdam:
    """Global-norm clip + Adam over the flat bucket in one fused sweep
    (reference misc/utils.py:174-200 `clip_gradient_norm(optimizer, 10.)` + `torch.optim.Adam`).

    One difference from `torch.optim.Adam`, visible only with `weight_decay > 0` (the reference trains with 0, opts.py): torch
    skips a parameter whose `.grad` is None, the flat sweep has no such notion -- a parameter the step never touched (e.g. the
    unused `ctx2att` of a Sub-GC model) has a zero gradient slot and still receives its decay term."""

    def __init__(self, model, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, clip_norm=10.0):
        self.model, self.lr, self.betas, self.eps, self.wd, self.clip = model, lr, betas, eps, weight_decay, clip_norm
        self.m = torch.zeros_like(model.flat_params)
        self.v = torch.zeros_like(model.flat_params)
        self.sumsq = torch.zeros(1, device=model.flat_params.device)
        self.t = 0
        self.reducer = None            # a GradBucketReducer(optimizer=self) accumulates the clip norm slice by slice (begin_step / accumulate)
        self._have = None              # stages whose squared norm is already in self.sumsq for the gradients of the current step

    def begin_step(self):
        """(reducer.prepare) a new set of gradients: the norm accumulator starts from zero."""
        from . import ops
        ops.fill_(self.sumsq, 0.0)
        self._have = set()

    def accumulate(self, stage, lo, hi):
        """sumsq += |flat_grads[lo:hi]|^2 on the CURRENT stream (the slice is final -- and, with several ranks, reduced -- there)."""
        from . import ops
        if self._have is None or stage in self._have:
            return
        self._have.add(stage)
        ops.sumsq(self.model.flat_grads[lo:hi], self.sumsq)

    def step(self, grad_scale=1.0, zero_grad=False):
        """`zero_grad`: leave the gradient buffer ZEROED by the sweep itself (= this step followed by the `optimizer.zero_grad()` every
        training iteration starts with, train.py) -- the next `flatten_grads` / `GradBucketReducer.prepare` then skips its fill pass over
        the buffer.  Default: torch's semantics, the (scaled, clipped) gradients stay readable after the step."""
        from . import ops
        self.t += 1
        if self._have:                                          # some slices are in already: add whatever was not announced early (always the fusion slice)
            for st, lo, hi in self.reducer.buckets:
                if hi > lo:
                    self.accumulate(st, lo, hi)
            self._have = None
        else:                                                   # nothing accumulated (one rank, or no reducer): one pass over the whole buffer
            self._have = None
            ops.fill_(self.sumsq, 0.0)
            ops.sumsq(self.model.flat_grads, self.sumsq)
        m = self.model
        snap = m.weights_b16() if getattr(m, "bf16_storage", False) else None     # compute_dtype = bf16: refreshed in the same sweep
        ops.clip_adam_step(m.flat_params, m.flat_grads, self.m, self.v, self.sumsq, self.clip, self.lr,
                           self.betas[0], self.betas[1], self.eps, self.wd, self.t, grad_scale, p_bf16=snap, zero_grad=zero_grad)
        from . import ops as _ops
        m.__dict__["_grads_are_zero"] = (m.flat_grads.data_ptr(), m.flat_grads._version, _ops.GRAD_WRITES[0]) if zero_grad else None
        # the kernel wrote the weights through raw pointers: no torch version counter moved, so the decode-time snapshots
        # (x->gates table, K-concatenated LSTM matrices, captured hipGraphs) must be told explicitly
        m.invalidate_decode_caches()
        if snap is not None:
            m.weights_b16(fresh_from_optimizer=True)                               # ... while the bf16 weight snapshot is already current
