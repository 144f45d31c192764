"""Decode-time sub-graph selection and the token loop, for one image (the reference's contract) or MANY.

Reference: AttModel._sample (models/AttModel.py:236-326), gpn_layer test branch + subgraph_nms
(models/lib/gpn.py:83-150), driven one image per call by misc/eval_utils.py:98-104 (the loader hands
the 5 "counterparts" of a single image and gpn.py:84 asserts it).

A decode step of <= 10 kept sub-graphs streams all 152 MB of decoder weights for ~10 rows of work
(HBM weight-streaming bound below ~40 rows).  `sample_images` therefore scores the
candidates of every image in one pool/score launch, runs the node-set NMS per image, and decodes the
kept sub-graphs of ALL images as one batch; per-image results are cut back out, and everything the
reference leaves untouched after an image's own early break (AttModel.py:318-319) is zeroed again, so each
image's tuple equals what a one-image call returns.
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import torch

from .. import beam
from .. import functions as F_
from .. import ops


def _forced_pick(logp, tok, k, temp, t, seq, seqlp, it, unfinished, counts):
    """Test hook: follow a given token path (plumbing in torch; never on the product path)."""
    lp = torch.log_softmax(logp / temp, 1) if k else logp
    slp = lp.gather(1, tok.view(-1, 1)).view(-1)
    unf = (tok > 0) if t == 0 else (unfinished.bool() & (tok > 0))
    unfinished.copy_(unf.int())
    w = tok * unf.long()
    seq[:, t] = w
    seqlp[:, t] = slp
    it.copy_(w)
    counts[t] = unf.sum().int()


def score_candidates(m, X2, N, images):
    """The launch-only half of `select_subgraphs`: pool + score every candidate sub-graph of every image and run the node-set
    NMS; nothing is read back.  -> namespace(idx, lens_i, img, read_out, score, sizes, offs, keep_all, n_keep)."""
    dev, L = X2.device, m.GCN_dim
    for _, gpn_obj_ind, _, _ in images:
        if gpn_obj_ind.size(0) != 5:
            raise AssertionError("test branch of sGPN expects the 5 counterparts of ONE image (gpn.py:84)")
    # per-image pieces are views (pos slots then neg slots; the pooling weights are the diagonal of gpn_pool_mtx); every
    # device op below runs ONCE over the concatenation of all images, so the host cost does not grow with the image count
    idx = torch.cat([g[0].reshape(-1, N) for _, g, _, _ in images]).contiguous()
    w = torch.cat([p_[0].diagonal(dim1=-2, dim2=-1).reshape(-1, N) for _, _, _, p_ in images]).contiguous()
    lens_all = torch.cat([a[0].reshape(-1, N) for _, _, a, _ in images]).sum(1)
    sizes = [g.size(1) * g.size(2) for _, g, _, _ in images]
    img = torch.from_numpy(np.repeat(np.asarray([r for r, _, _, _ in images], dtype=np.int32), sizes)).to(dev)
    G = idx.size(0)
    read_out, _ = ops.pool_fwd(X2, idx, idx.stride(0), w, w.stride(0), 1, lens_all, img, G, N, L, want_argmax=False)
    if m.use_sGPN_score:
        hid = torch.empty(G, m.att_hid_size, device=dev)
        ops.gemm(read_out, m.P("gpn_layer.gpn_fc.0.weight"), hid, tb=True, bias=m.P("gpn_layer.gpn_fc.0.bias"), relu=True)
        score, _ = ops.gpn_score_fwd(hid, None, 1.0, m.P("gpn_layer.gpn_fc.3.weight"), m.P("gpn_layer.gpn_fc.3.bias"), want_loss=False)
        score = score.view(-1)
    else:
        score = torch.ones(G, device=dev)
    lens_i = lens_all.to(torch.int32)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    keep_all = n_keep = None
    if not m.sct:                                                                      # use_nms (AttModel.py:95); node-set NMS is per image (gpn.py:108-138)
        keep_all, n_keep, _ = ops.subgraph_nms_batched(score, idx, lens_i, sizes, m.gpn_nms_thres, m.gpn_max_subg)
    return SimpleNamespace(idx=idx, lens_i=lens_i, img=img, read_out=read_out, score=score, sizes=sizes, offs=offs, keep_all=keep_all,
                           n_keep=n_keep, G=G)


def select_subgraphs(m, X2, N, images, front=None, kept=None):
    """Score every candidate sub-graph of every image, keep the NMS survivors.

    images: list of (image_row, gpn_obj_ind [5,2,M,N], att_masks [5,2,M,N], gpn_pool_mtx [5,2,M,N,N]) -- only
    counterpart 0 is read, like gpn.py:86-96.  Returns a list of dicts(fc, lens, idx, img, score, keep).
    `front` / `kept`: the result of `score_candidates` and the survivor counts when the caller already has them."""
    dev, L = X2.device, m.GCN_dim
    fr = front if front is not None else score_candidates(m, X2, N, images)
    idx, lens_i, img, read_out, score, sizes, offs, keep_all, G = fr.idx, fr.lens_i, fr.img, fr.read_out, fr.score, fr.sizes, fr.offs, fr.keep_all, fr.G
    if not m.sct:
        kept = fr.n_keep.cpu().numpy().astype(np.int64) if kept is None else np.asarray(kept, dtype=np.int64)   # ONE host read for all images
        g0 = np.repeat(offs[:-1], kept)                                                # first candidate of the owning image, per survivor
        slot = g0 + (np.arange(int(kept.sum())) - np.repeat(np.cumsum(kept) - kept, kept))
        hs = torch.from_numpy(np.stack([slot, g0])).to(dev)                            # one small upload
        keep = keep_all[hs[0]]                                                         # survivors, image-local indices in original order
        glob = keep + hs[1]
    else:
        kept = np.asarray(sizes, dtype=np.int64)
        glob = torch.arange(G, device=dev)
        keep = glob - torch.from_numpy(np.repeat(offs[:-1], kept)).to(dev)
    n = glob.numel()
    fc = torch.empty(n, 2 * L, device=dev)
    if n:
        h = torch.empty(n, m.att_hid_size, device=dev)
        ops.gemm(read_out[glob], m.P("gpn_layer.read_out_proj.0.weight"), h, tb=True, bias=m.P("gpn_layer.read_out_proj.0.bias"))
        ops.gemm(h, m.P("gpn_layer.read_out_proj.1.weight"), fc, tb=True, bias=m.P("gpn_layer.read_out_proj.1.bias"))
    lens_g, idx_g, img_g, score_g = lens_i[glob], idx[glob], img[glob], score[glob]
    out, r0 = [], 0
    for k_ in kept.tolist():                                                           # per-image results are row slices (views)
        r1 = r0 + k_
        out.append(dict(keep=keep[r0:r1], fc=fc[r0:r1], lens=lens_g[r0:r1], idx=idx_g[r0:r1], img=img_g[r0:r1], score=score_g[r0:r1]))
        r0 = r1
    return out


def full_graph_rows(m, X2, N, images):
    """Full-GC baseline (AttModel.py:261-271): one row per image, mean-pooled read-out, attention over the first 36 nodes."""
    dev, L = X2.device, m.GCN_dim
    n = len(images)
    ar = torch.arange(N, device=dev).view(1, N).expand(n, N).contiguous()
    img = torch.tensor([im[0] for im in images], device=dev, dtype=torch.int32)
    mean, _ = ops.pool_fwd(X2, ar, N, torch.ones(n, N, device=dev), N, 1, torch.full((n,), float(N), device=dev), img, n, N, L, want_argmax=False)
    h = torch.empty(n, m.att_hid_size, device=dev)
    fc = torch.empty(n, 2 * L, device=dev)
    ops.gemm(mean[:, L:], m.P("read_out_proj.0.weight"), h, tb=True, bias=m.P("read_out_proj.0.bias"))
    ops.gemm(h, m.P("read_out_proj.1.weight"), fc, tb=True, bias=m.P("read_out_proj.1.bias"))
    out = []
    for i, (row, _g, att_masks, _p) in enumerate(images):
        mk = att_masks[0:1, 0, 0]
        mk[:, :36].fill_(1.0)                                                          # writes into the caller's tensor, like :148-149
        out.append(dict(fc=fc[i:i + 1], lens=mk.sum(1).to(torch.int32), idx=ar[i:i + 1], img=img[i:i + 1],
                        score=torch.ones(1, device=dev), keep=torch.arange(1, device=dev)))
    return out


class _FrontBuffers:
    """Fixed-address copies of what the selection phase of ONE image leaves for the decode graph (node states, read-outs,
    scores, node sets, NMS survivors).  The copies are queued BEFORE the one host read of a call (the survivor count), so
    after that read the host issues a single graph replay: the survivor gathers, the read-out projection and the attention-set
    preparation (`F_.Prepared`) run inside the graph instead of as ~25 eager launches with the GPU idling between them."""

    def __init__(self, m, G, N):
        """G: CAPACITY in candidate sub-graphs (a power of two): images with any smaller candidate count use the same buffers,
        hence the same captured graphs -- the count varies from image to image on real data."""
        dev, L = m.flat_params.device, m.GCN_dim
        self.G, self.N = G, N
        self.X2, self.read_out = torch.empty(N, L, device=dev), torch.empty(G, 2 * L, device=dev)
        self.keep, self.lens = torch.zeros(G, device=dev, dtype=torch.long), torch.zeros(G, device=dev, dtype=torch.int32)
        self.idx, self.score = torch.zeros(G, N, device=dev, dtype=torch.long), torch.empty(G, device=dev)

    def load(self, X2, fr):
        g = fr.G                                                                  # <= capacity; rows past g are never indexed (keep < g)
        self.X2.copy_(X2); self.read_out[:g].copy_(fr.read_out); self.keep[:g].copy_(fr.keep_all[:g])
        self.lens[:g].copy_(fr.lens_i); self.idx[:g].copy_(fr.idx); self.score[:g].copy_(fr.score)

    def prepared(self, m, n, P):
        """(inside the capture) the first n NMS survivors -> F_.Prepared, their scores and indices."""
        dev, L = self.X2.device, m.GCN_dim
        keep = self.keep[:n]
        h, fc = torch.empty(n, m.att_hid_size, device=dev), torch.empty(n, 2 * L, device=dev)
        ops.gemm(self.read_out[keep], m.P("gpn_layer.read_out_proj.0.weight"), h, tb=True, bias=m.P("gpn_layer.read_out_proj.0.bias"))
        ops.gemm(h, m.P("gpn_layer.read_out_proj.1.weight"), fc, tb=True, bias=m.P("gpn_layer.read_out_proj.1.bias"))
        pr = F_.Prepared(fc, self.X2, self.lens[keep], self.idx[keep], torch.zeros(n, device=dev, dtype=torch.int32), self.N, P, None, None, 1.0)
        return pr, self.score[keep], keep


def _front_buffers(m, G, N):
    cache = m.__dict__.setdefault("_front_cache", {})
    G = max(128, 1 << (int(G) - 1).bit_length())                                  # capacity classes: 128, 256, 512, ...
    key = (G, N, m.flat_params.data_ptr())
    if key not in cache:
        if len(cache) >= 8:
            cache.clear()
            m.__dict__.pop("_graph_cache", None)                                  # graphs captured on the dropped buffers go with them
        cache[key] = _FrontBuffers(m, G, N)
    return cache[key]


def _load_prepared(dst, pr):
    rows = pr.u.size(0)
    dst.f.copy_(pr.f)
    dst.u[:rows].copy_(pr.u); dst.v[:rows].copy_(pr.v)
    dst.off.copy_(pr.off); dst.lens.copy_(pr.lens)


class _GraphedLoop:
    """The token loop of ONE image as a replayable hipGraph (torch.cuda.CUDAGraph over the C-ABI launches).

    The reference-shaped call decodes <= 10 rows per step: ~12 kernels of 5-30 us each, 21 steps -- launch-bound.  For a
    given row count the launch sequence is static (the early break is device-side), so it is captured once on buffers
    of fixed address and replayed per image; the image's prepared features are copied into those buffers first.
    Keyed by (rows, N, attention rows capacity, k, return_att) and by the version of the flat parameter buffer (the
    K-concatenated LSTM weights inside DecodeState are snapshots of the parameters)."""

    def __init__(self, m, n, N, k, return_att, P, fb=None):
        dev = m.flat_params.device
        T, R, A = m.seq_length, m.rnn_size, m.att_hid_size
        self.n, self.N, self.T, self.k, self.return_att = n, N, T, k, return_att
        self.m, self.P, self.fb = m, P, fb                                        # fb: the graph starts from the selection's static buffers
        if fb is not None:
            self.score_out, self.keep_out = torch.empty(n, device=dev), torch.zeros(n, device=dev, dtype=torch.long)
        cap = n * N
        z = lambda *s, dt=torch.float32: torch.zeros(*s, device=dev, dtype=dt)
        self.pr = SimpleNamespace(S=n, N=N, f=z(n, R), u=z(cap, A), v=z(cap, R), off=z(n, dt=torch.int32), lens=z(n, dt=torch.int32))
        self.st = F_.DecodeState(self.pr, P, N, return_att, xt_table=m.xt_gates_table(), fuse_lstm=True, snapshots=m.decode_snapshots(), W16=m.decode_w16())
        self.seq, self.seqlp = z(n, T, dt=torch.long), z(n, T)
        self.it, self.unfinished, self.counts = z(n, dt=torch.long), z(n, dt=torch.int32), z(T, dt=torch.int32)
        self.AL = z(T + 1, n, N) if return_att else None
        self.u = z(T, n) if k else None
        self.topk_temp = m.topk_temp
        self._loop()                                                             # eager warm-up (sets kernel attributes, fills caches)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with ops.graph_capture(self.graph, dev):
            self._loop()

    def _loop(self):
        T = self.T
        if self.fb is not None:
            pr, score, keep = self.fb.prepared(self.m, self.n, self.P)
            _load_prepared(self.pr, pr)
            self.score_out.copy_(score); self.keep_out.copy_(keep)
        self.st.reset()
        for b in (self.seq, self.seqlp, self.it, self.unfinished, self.counts) + ((self.AL,) if self.AL is not None else ()):
            b.zero_()
        if self.k == 0 and self.st.fused and self.st.xt_table is not None and getattr(self.m, "decode_fused_pick", True):
            # greedy: the pick rides in the logits / attention-LSTM launches (functions.DecodeState.greedy_loop)
            self.st.greedy_loop(T, self.seq, self.seqlp, self.counts, self.AL, self.it)
            return
        for t in range(T + 1):
            logp = self.st.step(self.it, self.AL[t] if self.return_att else None, normalize=False)
            if t == T:
                break
            ops.decode_pick(logp, self.k, self.topk_temp, None if self.u is None else self.u[t], t, self.seq, self.seqlp, self.it,
                            self.unfinished, self.counts[t:t + 1], self.counts[t - 1:t] if t > 0 else None, raw=True)

    def run(self, pr, uniforms):
        if pr is not None:
            _load_prepared(self.pr, pr)
        if self.u is not None:
            self.u.copy_(uniforms.t())
        self.graph.replay()
        return self.seq.clone(), self.seqlp.clone(), self.counts, (self.AL.clone() if self.AL is not None else None)


class _GraphedBeam:
    """Beam search of ONE image (test.sh decodes Sub_GC_Kar with beam 2, Full-GC with beam 3) as a replayable hipGraph: with
    the candidate bookkeeping on the device (`subgc_beam_step`) the whole search is a static launch sequence, ~16 launches per
    step on <= 10 x beam rows, i.e. launch-bound when issued eagerly."""

    def __init__(self, m, n, N, P, opt, fb=None):
        dev = m.flat_params.device
        T, R, A = m.seq_length, m.rnn_size, m.att_hid_size
        self.m, self.n, self.P, self.fb = m, n, P, fb
        if fb is not None:
            self.score_out, self.keep_out = torch.empty(n, device=dev), torch.zeros(n, device=dev, dtype=torch.long)
        cap = n * N
        z = lambda *s, dt=torch.float32: torch.zeros(*s, device=dev, dtype=dt)
        self.pr = SimpleNamespace(S=n, N=N, f=z(n, R), u=z(cap, A), v=z(cap, R), off=z(n, dt=torch.int32), lens=z(n, dt=torch.int32))
        self.eng = beam._BatchEngine(self.pr, P, N, int(opt.get("beam_size", 10)), m.xt_gates_table(), m.decode_snapshots())
        self.ds = beam.DeviceSearch(self.eng, T, opt)
        self._loop()                                                             # eager warm-up
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with ops.graph_capture(self.graph, dev):
            self._loop()

    def _loop(self):
        if self.fb is not None:
            pr, score, keep = self.fb.prepared(self.m, self.n, self.P)
            _load_prepared(self.pr, pr)
            self.score_out.copy_(score); self.keep_out.copy_(keep)
        self.eng.refresh()
        self.ds.loop()

    def run(self, pr):
        if pr is not None:
            _load_prepared(self.pr, pr)
        self.graph.replay()
        return self.ds.collect()


def _graphed_beam(m, n, N, P, opt, fb=None):
    okey = tuple(sorted((k, v) for k, v in opt.items() if k in ("beam_size", "group_size", "diversity_lambda", "decoding_constraint", "length_penalty")))
    key = ("beam", n, N, okey, None if fb is None else fb.G) + m.weights_version()
    cache = m.__dict__.setdefault("_graph_cache", {})
    if key not in cache:
        for old in [q for q in cache if q[:5] == key[:5]]:
            del cache[old]
        if len(cache) >= 64:                                                      # states share their weight snapshots: a graph is a few MB
            cache.clear()
        cache[key] = _GraphedBeam(m, n, N, P, opt, fb)
    return cache[key]


def _graphed_loop(m, n, N, k, return_att, P, fb=None):
    key = (n, N, k, return_att, None if fb is None else fb.G) + m.weights_version()
    cache = m.__dict__.setdefault("_graph_cache", {})
    if key not in cache:
        for old in [q for q in cache if q[:5] == key[:5]]:                       # parameters changed: drop the stale snapshot
            del cache[old]
        if len(cache) >= 64:                                                      # states share their weight snapshots: a graph is a few MB
            cache.clear()
        cache[key] = _GraphedLoop(m, n, N, k, return_att, P, fb)
    return cache[key]


@torch.no_grad()
def decode_one_image(m, X2, N, image, opt, uniforms=None, forced=None):
    """The reference-shaped call (ONE image, sGPN + NMS): selection launches -> static buffers -> the one host read (how many
    sub-graphs survived) -> ONE graph replay that gathers the survivors, prepares their attention sets and runs the token
    loop / beam search.  Falls back to `select_subgraphs` + `decode` (same kernels, eager) whenever a graph does not apply."""
    T = m.seq_length
    beam_size = opt.get("beam_size", 1)
    return_att = opt.get("return_att", 0) == 1
    graphable = m.gpn and not m.sct and forced is None and getattr(m, "decode_hipgraph", True)
    fr = score_candidates(m, X2, N, [image]) if m.gpn else None
    if not graphable or fr.G == 0:
        sel = select_subgraphs(m, X2, N, [image], front=fr) if m.gpn else full_graph_rows(m, X2, N, [image])
        return decode(m, X2, N, sel, opt, uniforms, forced)[0]
    fb = _front_buffers(m, fr.G, N)
    fb.load(X2, fr)                                                               # queued before the host read below
    n = int(fr.n_keep.item())
    if n == 0 or (beam_size > 1 and n * beam_size > 128) or (beam_size <= 1 and n > 16):
        return decode(m, X2, N, select_subgraphs(m, X2, N, [image], front=fr, kept=[n]), opt, uniforms, forced)[0]
    P = m._decoder_params()
    try:
        if beam_size > 1:
            g = _graphed_beam(m, n, N, P, opt, fb)
            seq, seqlp, done = g.run(None)
            m.done_beams = done
            return (seq, seqlp, g.score_out.clone(), g.keep_out.clone())
        k = m.the_k if m.topk_sampling else 0
        if k and uniforms is None:
            uniforms = torch.rand(n, T, device=X2.device)
        g = _graphed_loop(m, n, N, k, return_att, P, fb)
    except RuntimeError as e:                                                     # capture unavailable here: same kernels, launched eagerly
        import warnings
        warnings.warn(f"hipGraph capture of the decode loop failed ({e}); decoding eagerly from now on")
        m.decode_hipgraph = False
        return decode(m, X2, N, select_subgraphs(m, X2, N, [image], front=fr, kept=[n]), opt, uniforms, forced)[0]
    seq, seqlp, counts, AL = g.run(None, uniforms)
    r = (seq, seqlp, g.score_out.clone(), g.keep_out.clone())
    if return_att:
        dead = (counts.cpu() == 0).nonzero()
        steps = int(dead[0]) + 1 if dead.numel() else T + 1
        n_max = int(g.pr.lens.max().item())
        r = r + (AL[:steps, :, :n_max].permute(1, 0, 2).contiguous(),)
    return r


@torch.no_grad()
def decode(m, X2, N, sel, opt, uniforms=None, forced=None):
    """Greedy / top-k / beam decode of the selected sub-graphs of one or many images as ONE batch.
    Returns one result tuple per image: (seq, seqLogprobs, score, keep[, att_weights])."""
    dev = X2.device
    T = m.seq_length
    return_att = opt.get("return_att", 0) == 1
    beam_size = opt.get("beam_size", 1)
    sizes = [s["keep"].numel() for s in sel]
    n = sum(sizes)
    if n == 0:
        z = lambda: (torch.zeros(0, T, device=dev, dtype=torch.long), torch.zeros(0, T, device=dev))
        return [z() + (s["score"], s["keep"]) + ((torch.zeros(0, 0, 0, device=dev),) if return_att else ()) for s in sel]
    cat = lambda k: torch.cat([s[k] for s in sel]).contiguous()
    fc, lens_k, idx_k, img_k = cat("fc"), cat("lens"), cat("idx"), cat("img")
    P = m._decoder_params()
    pr = F_.Prepared(fc, X2, lens_k, idx_k, img_k, N, P, None, None, 1.0)
    bounds = [0]
    for s in sizes:
        bounds.append(bounds[-1] + s)
    if beam_size > 1:                                                                  # AttModel.py:245-246 -> :179-234
        graphed = None
        if len(sel) == 1 and n * beam_size <= 128 and getattr(m, "decode_hipgraph", True):
            try:
                graphed = _graphed_beam(m, n, N, P, opt)
            except RuntimeError as e:                                            # capture unavailable here: same kernels, launched eagerly
                import warnings
                warnings.warn(f"hipGraph capture of the beam search failed ({e}); searching eagerly from now on")
                m.decode_hipgraph = False
        if graphed is not None:
            seq, seqlp, done = graphed.run(pr)
        else:
            seq, seqlp, done = beam.beam_decode(pr, P, N, T, opt, xt_table=m.xt_gates_table())
        m.done_beams = done if len(sel) == 1 else [done[a:b] for a, b in zip(bounds, bounds[1:])]
        return [(seq[a:b], seqlp[a:b], s["score"], s["keep"]) for s, a, b in zip(sel, bounds, bounds[1:])]
    k = m.the_k if m.topk_sampling else 0
    if k and uniforms is None and forced is None:
        uniforms = torch.rand(n, T, device=dev)
    graphed = None
    if len(sel) == 1 and forced is None and n <= 16 and getattr(m, "decode_hipgraph", True):
        # the reference-shaped call (one image, <= 10 rows): launch-bound, replayed as one hipGraph
        try:
            graphed = _graphed_loop(m, n, N, k, return_att, P)
        except RuntimeError as e:                                                # capture unavailable here: same kernels, launched eagerly
            import warnings
            warnings.warn(f"hipGraph capture of the decode loop failed ({e}); decoding eagerly from now on")
            m.decode_hipgraph = False
    if graphed is not None:
        seq, seqlp, counts, AL = graphed.run(pr, uniforms)
    else:
        st = F_.DecodeState(pr, P, N, return_att, xt_table=m.xt_gates_table(), fuse_lstm=True, snapshots=m.decode_snapshots(), W16=m.decode_w16())
        seq = torch.zeros(n, T, device=dev, dtype=torch.long)
        seqlp = torch.zeros(n, T, device=dev)
        it = torch.zeros(n, device=dev, dtype=torch.long)
        unfinished = torch.zeros(n, device=dev, dtype=torch.int32)
        counts = torch.zeros(T, device=dev, dtype=torch.int32)
        AL = torch.zeros(T + 1, n, N, device=dev) if return_att else None
        for t in range(T + 1):
            logp = st.step(it, AL[t] if return_att else None, normalize=forced is not None)
            if t == T:
                break
            if forced is not None:
                _forced_pick(logp, forced[:, t].contiguous(), k, m.topk_temp, t, seq, seqlp, it, unfinished, counts)
            else:
                ops.decode_pick(logp, k, m.topk_temp, None if uniforms is None else uniforms[:, t].contiguous(), t, seq, seqlp, it,
                                unfinished, counts[t:t + 1], counts[t - 1:t] if t > 0 else None, raw=True)
    if len(sel) == 1:
        steps = None
        if return_att:
            dead = (counts.cpu() == 0).nonzero()
            steps = [int(dead[0]) + 1 if dead.numel() else T + 1]
    else:
        # per-image early break: image i stops after the first step at which none of ITS rows is unfinished; the
        # reference writes nothing (tokens, log-probs, attention rows) beyond that step
        alive = (seq > 0).int().cumprod(1)                                             # [n, T] row still unfinished after step t
        row_img = torch.from_numpy(np.repeat(np.arange(len(sizes)), sizes)).to(dev)    # owning image of every row
        per = torch.zeros(len(sizes), T, device=dev, dtype=alive.dtype).index_add_(0, row_img, alive)   # [I, T] unfinished rows per image
        stopped = (per == 0).int()
        brk = torch.where(stopped.any(1), stopped.argmax(1), torch.full_like(stopped[:, 0], T - 1).long())   # break step per image
        tgrid = torch.arange(T, device=dev).view(1, T)
        row_brk = brk[row_img]
        seqlp = seqlp * (tgrid <= row_brk.view(-1, 1))
        steps = [min(int(b) + 1, T) + (1 if int(b) == T - 1 and not bool(s.any()) else 0) for b, s in zip(brk.cpu(), stopped.cpu())]
    out = []
    for i, (s, a, b) in enumerate(zip(sel, bounds, bounds[1:])):
        r = (seq[a:b], seqlp[a:b], s["score"], s["keep"])
        if return_att:
            n_max = int(s["lens"].max().item()) if b > a else 0
            r = r + (AL[:steps[i], a:b, :n_max].permute(1, 0, 2).contiguous(),)
        out.append(r)
    return out
