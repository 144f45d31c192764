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


def score_candidates(m, X2, N, images, fb=None):
    """The launch-only half of `select_subgraphs`: pool + score every candidate sub-graph of every image and run the node-set
    NMS; nothing is read back.  -> namespace(idx, lens_i, img, read_out, score, sizes, offs, keep_all, n_keep).
    Every device op is a C-ABI launch over the concatenation of all images (the per-image loader tensors are read in place through
    ONE address table, subgc_gpn_test_prep); `fb` (_FrontBuffers of one image): results land in its static buffers."""
    dev, L = X2.device, m.GCN_dim
    for _, gpn_obj_ind, _, _ in images:
        if gpn_obj_ind.size(0) != 5:
            raise AssertionError("test branch of sGPN expects the 5 counterparts of ONE image (gpn.py:84)")
    idx, w, lens_all, lens_i, img, offsets32, sizes, alive = ops.gpn_test_prep(images, N, dev, out=fb)
    G = idx.size(0)
    read_out, _ = ops.pool_fwd(X2, idx, idx.stride(0), w, w.stride(0), 1, lens_all, img, G, N, L, want_argmax=False,
                               out=None if fb is None else fb.read_out[:G])
    if m.use_sGPN_score and G:
        hid = torch.empty(G, m.att_hid_size, device=dev)
        ops.gemm(read_out, m.P("gpn_layer.gpn_fc.0.weight"), hid, tb=True, bias=m.P("gpn_layer.gpn_fc.0.bias"), relu=True)
        score, _ = ops.gpn_score_fwd(hid, None, 1.0, m.P("gpn_layer.gpn_fc.3.weight"), m.P("gpn_layer.gpn_fc.3.bias"), want_loss=False,
                                     out=None if fb is None else fb.score[:G].view(G, 1))
        score = score.view(-1)
    else:
        score = ops.fill_(torch.empty(G, device=dev) if fb is None else fb.score[:G], 1.0)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    keep_all = n_keep = None
    if not m.sct:                                                                      # use_nms (AttModel.py:95); node-set NMS is per image (gpn.py:108-138)
        keep_all, n_keep, _ = ops.subgraph_nms_batched(score, idx, lens_i, sizes, m.gpn_nms_thres, m.gpn_max_subg, offsets=offsets32,
                                                       keep=None if fb is None else fb.keep)
    return SimpleNamespace(idx=idx, lens_i=lens_i, img=img, read_out=read_out, score=score, sizes=sizes, offs=offs, keep_all=keep_all,
                           n_keep=n_keep, G=G, offsets32=offsets32, alive=alive)


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
        n = int(kept.sum())
        # survivors in image order: image-local indices (original order) and their global candidate rows -- one launch
        keep, glob = ops.nms_compact(keep_all, fr.n_keep, fr.offsets32, len(sizes), n)
    else:                                                                              # controllability mode (sct): every candidate, plumbing in torch
        kept = np.asarray(sizes, dtype=np.int64)
        glob = torch.arange(G, device=dev)
        keep = glob - torch.from_numpy(np.repeat(offs[:-1], kept)).to(dev)
    n = glob.numel()
    e = lambda *sh, dt=torch.float32: torch.empty(*sh, device=dev, dtype=dt)
    fc = e(n, 2 * L)
    lens_g, idx_g, img_g, score_g = e(n, dt=torch.int32), e(n, idx.size(1), dt=torch.int64), e(n, dt=torch.int32), e(n)
    if n:
        ro, h = e(n, 2 * L), e(n, m.att_hid_size)
        ops.take_rows([(read_out, ro), (lens_i, lens_g), (idx if idx.is_contiguous() else idx.contiguous(), idx_g), (img, img_g)], glob)
        ops.take_rows([(score if score.is_contiguous() else score.contiguous(), score_g)], glob)
        ops.gemm(ro, m.P("gpn_layer.read_out_proj.0.weight"), h, tb=True, bias=m.P("gpn_layer.read_out_proj.0.bias"))
        ops.gemm(h, m.P("gpn_layer.read_out_proj.1.weight"), fc, tb=True, bias=m.P("gpn_layer.read_out_proj.1.bias"))
    out, r0 = _Selection(), 0
    for k_ in kept.tolist():                                                           # per-image results are row slices (views)
        r1 = r0 + k_
        out.append(dict(keep=keep[r0:r1], fc=fc[r0:r1], lens=lens_g[r0:r1], idx=idx_g[r0:r1], img=img_g[r0:r1], score=score_g[r0:r1]))
        r0 = r1
    out.whole = dict(fc=fc, lens=lens_g, idx=idx_g, img=img_g, score=score_g, keep=keep)
    tap = m.__dict__.get("tap")
    if tap is not None:                                                                # debug tap (tests): golden-file names
        tap.update(read_out=read_out.clone(), score_all=score.clone().view(-1), keep_ind=keep.clone(), subgraph_score_raw=score_g.clone(), fc_sel=fc.clone(),
                   sel_idx=idx_g.clone(), sel_lens=lens_g.clone(), att_sel=X2.view(-1, N, L)[img_g.long().view(-1, 1), idx_g].clone())
    return out


def full_graph_rows(m, X2, N, images):
    """Full-GC baseline (AttModel.py:261-271): one row per image, mean-pooled read-out, attention over the first 36 nodes."""
    dev, L = X2.device, m.GCN_dim
    n = len(images)
    cache = m.__dict__.setdefault("_const_cache", {})
    key = ("full_graph", str(dev), n, N)
    c = cache.get(key)
    if c is None:                                                                      # shape-only index tensors, built once per (batch, N)
        ar = torch.arange(N, device=dev).view(1, N).expand(n, N).contiguous()
        c = cache[key] = dict(ar=ar, ones=torch.ones(n, N, device=dev), full=torch.full((n,), float(N), device=dev),
                              one=torch.ones(n, device=dev), zero=torch.zeros(n, device=dev, dtype=torch.long))
    ar = c["ar"]
    img = ops.upload([im[0] for im in images], torch.int32, dev)                       # one small upload (pinned, asynchronous)
    mean, _ = ops.pool_fwd(X2, ar, N, c["ones"], N, 1, c["full"], img, n, N, L, want_argmax=False)
    h = torch.empty(n, m.att_hid_size, device=dev)
    fc = torch.empty(n, 2 * L, device=dev)
    ops.gemm(mean[:, L:], m.P("read_out_proj.0.weight"), h, tb=True, bias=m.P("read_out_proj.0.bias"))
    ops.gemm(h, m.P("read_out_proj.1.weight"), fc, tb=True, bias=m.P("read_out_proj.1.bias"))
    out = _Selection()
    lens_all = torch.empty(n, device=dev, dtype=torch.int32)
    for i, (row, _g, att_masks, _p) in enumerate(images):
        mk = att_masks[0:1, 0, 0]
        ops.fill2d_(mk[:, :36], 1.0)                                                   # writes into the caller's tensor, like :148-149
        call_lens = ops.row_count(mk)
        ops.copy_(lens_all[i:i + 1], call_lens)
        out.append(dict(fc=fc[i:i + 1], lens=lens_all[i:i + 1], idx=ar[i:i + 1], img=img[i:i + 1], score=c["one"][i:i + 1], keep=c["zero"][i:i + 1]))
    out.whole = dict(fc=fc, lens=lens_all, idx=ar, img=img, score=c["one"], keep=c["zero"])
    return out


class _Selection(list):
    """Per-image selection dicts (row slices) plus `whole`: the tensors they are slices of, in image order -- what a batched decode
    reads instead of concatenating the slices again."""
    whole = None


class _FrontBuffers:
    """Fixed-address destinations of what the selection phase of ONE image leaves for the decode graph (node states, read-outs,
    scores, node sets, NMS survivors): `score_candidates(fb=...)` writes them IN PLACE, so after the one host read of a call (the
    survivor count) the host issues a single graph replay: the survivor gathers, the read-out projection and the attention-set
    preparation (`F_.Prepared`) run inside the graph instead of as ~25 eager launches with the GPU idling between them."""

    def __init__(self, m, G, N):
        """G: CAPACITY in candidate sub-graphs (a power of two): images with any smaller candidate count use the same buffers,
        hence the same captured graphs -- the count varies from image to image on real data."""
        dev, L = m.flat_params.device, m.GCN_dim
        self.G, self.N = G, N
        self.X2, self.read_out = torch.empty(N, L, device=dev), torch.empty(G, 2 * L, device=dev)
        self.keep, self.lens = ops.zero_(torch.empty(G, device=dev, dtype=torch.long)), ops.zero_(torch.empty(G, device=dev, dtype=torch.int32))
        self.idx, self.score = ops.zero_(torch.empty(G, N, device=dev, dtype=torch.long)), torch.empty(G, device=dev)
        self.img0 = ops.zero_(torch.empty(G, device=dev, dtype=torch.int32))       # every survivor belongs to the one image (row 0 of X2)

    def load(self, X2):
        ops.copy_(self.X2, X2)                                                    # everything else was written in place

    def prepared(self, m, n, P, g):
        """(inside the capture) the first n NMS survivors -> F_.Prepared in `g.pr`'s buffers, their scores / indices in g.score_out / keep_out."""
        keep = self.keep[:n]
        ops.take_rows([(self.read_out, g.ro_sel), (self.lens, g.pr.lens), (self.idx, g.idx_sel), (self.score, g.score_out)], keep)
        ops.copy_(g.keep_out, keep)
        ops.gemm(g.ro_sel, m.P("gpn_layer.read_out_proj.0.weight"), g.h_sel, tb=True, bias=m.P("gpn_layer.read_out_proj.0.bias"))
        ops.gemm(g.h_sel, m.P("gpn_layer.read_out_proj.1.weight"), g.fc_sel, tb=True, bias=m.P("gpn_layer.read_out_proj.1.bias"))
        F_.Prepared(g.fc_sel, self.X2, g.pr.lens, g.idx_sel, self.img0[:n], self.N, P, None, None, 1.0, into=g.pr)


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
    ops.copy_(dst.f, pr.f)
    ops.copy_(dst.u[:rows], pr.u); ops.copy_(dst.v[:rows], pr.v)
    ops.copy_(dst.off, pr.off); ops.copy_(dst.lens, pr.lens)


def _arena(dev, parts):
    """One fp32-word buffer carved into typed views: parts = [(name, shape, dtype)], 8-byte types first.  -> (buffer, {name: view},
    {name: (first word, words)}): the token loop zeroes / snapshots its small state tensors with ONE launch instead of one each."""
    words, views, spans, o = 0, {}, {}, 0
    for _, shape, dt in parts:
        words += int(np.prod(shape)) * (2 if dt == torch.long else 1)
    buf = torch.empty(max(words, 2), device=dev, dtype=torch.float32)
    for name, shape, dt in parts:
        n = int(np.prod(shape)) * (2 if dt == torch.long else 1)
        if dt == torch.long and o % 2:
            raise ValueError("arena: 8-byte parts first")
        views[name] = buf[o:o + n].view(dt).view(*shape)
        spans[name] = (o, n)
        o += n
    return buf, views, spans


class _GraphedLoop:
    """The token loop of ONE image as a replayable hipGraph (torch.cuda.CUDAGraph over the C-ABI launches).

    The reference-shaped call decodes <= 10 rows per step: ~12 kernels of 5-30 us each, 21 steps -- launch-bound.  For a
    given row count the launch sequence is static (the early break is device-side), so it is captured once on buffers
    of fixed address and replayed per image; the image's prepared features are written into those buffers first.
    Keyed by (rows, N, attention rows capacity, k, return_att) and by the version of the flat parameter buffer (the
    K-concatenated LSTM weights inside DecodeState are snapshots of the parameters).  Every launch inside the graph and around the
    replay is a C-ABI kernel: the small state tensors live in one arena (one zero fill, one snapshot copy per image)."""

    def __init__(self, m, n, N, k, return_att, P, fb=None):
        dev = m.flat_params.device
        T, R, A, L = m.seq_length, m.rnn_size, m.att_hid_size, m.GCN_dim
        self.n, self.N, self.T, self.k, self.return_att = n, N, T, k, return_att
        self.m, self.P, self.fb = m, P, fb                                        # fb: the graph starts from the selection's static buffers
        cap = n * N
        z = lambda *s, dt=torch.float32: ops.zero_(torch.empty(*s, device=dev, dtype=dt))
        # results first (seq, keep_out, seqlp, score_out: snapshot as one block), then the loop's scratch state
        self.arena, v, sp = _arena(dev, [("seq", (n, T), torch.long), ("keep_out", (n,), torch.long), ("it", (n,), torch.long),
                                         ("seqlp", (n, T), torch.float32), ("score_out", (n,), torch.float32),
                                         ("unfinished", (n,), torch.int32), ("counts", (T,), torch.int32)])
        self.views, self.result_words = v, sp["score_out"][0] + sp["score_out"][1]
        self.seq, self.seqlp, self.it, self.unfinished, self.counts = v["seq"], v["seqlp"], v["it"], v["unfinished"], v["counts"]
        self.score_out, self.keep_out = v["score_out"], v["keep_out"]
        self.pr = SimpleNamespace(S=n, N=N, f=z(n, R), u=z(cap, A), v=z(cap, R), off=z(n, dt=torch.int32), lens=z(n, dt=torch.int32))
        if fb is not None:
            e = lambda *s, dt=torch.float32: torch.empty(*s, device=dev, dtype=dt)
            self.ro_sel, self.h_sel, self.fc_sel, self.idx_sel = e(n, 2 * L), e(n, A), e(n, 2 * L), e(n, N, dt=torch.long)
        self.st = F_.DecodeState(self.pr, P, N, return_att, xt_table=m.xt_gates_table(), fuse_lstm=True, snapshots=m.decode_snapshots(), W16=m.decode_w16())
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
        ops.fill_(self.arena, 0.0)                                               # seq, seqlp, it, unfinished, counts (and the result slots) in one launch
        if self.AL is not None:
            ops.fill_(self.AL, 0.0)
        if self.fb is not None:
            self.fb.prepared(self.m, self.n, self.P, self)
        self.st.reset()
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

    def run(self, pr, uniforms, step_major=False):
        """-> (seq, seqlp, counts, AL, score_out, keep_out): the results are one snapshot copy of the arena's result block.
        `uniforms` (top-k sampling): [T, n] when `step_major` (drawn by `_uniforms`), else the caller's [n, T] (tests)."""
        if pr is not None:
            _load_prepared(self.pr, pr)
        if self.u is not None:
            if step_major:
                ops.copy_(self.u, uniforms)
            else:
                self.u.copy_(uniforms.t())                                       # injected uniforms: test plumbing in torch
        self.graph.replay()
        out = torch.empty(self.result_words, device=self.arena.device, dtype=torch.float32)
        ops.copy_(out, self.arena[:self.result_words])
        n, T = self.n, self.T
        o = 0
        seq = out[o:o + 2 * n * T].view(torch.long).view(n, T); o += 2 * n * T
        keep_out = out[o:o + 2 * n].view(torch.long); o += 4 * n                  # (skips `it`)
        seqlp = out[o:o + n * T].view(n, T); o += n * T
        score_out = out[o:o + n]
        AL = None
        if self.AL is not None:
            AL = ops.copy_(torch.empty_like(self.AL), self.AL)
        return seq, seqlp, self.counts, AL, score_out, keep_out


class _GraphedBeam:
    """Beam search of ONE image (test.sh decodes Sub_GC_Kar with beam 2, Full-GC with beam 3) as a replayable hipGraph: with
    the candidate bookkeeping on the device (`subgc_beam_step`) the whole search is a static launch sequence, ~16 launches per
    step on <= 10 x beam rows, i.e. launch-bound when issued eagerly."""

    def __init__(self, m, n, N, P, opt, fb=None):
        dev = m.flat_params.device
        T, R, A, L = m.seq_length, m.rnn_size, m.att_hid_size, m.GCN_dim
        self.m, self.n, self.P, self.fb = m, n, P, fb
        cap = n * N
        z = lambda *s, dt=torch.float32: ops.zero_(torch.empty(*s, device=dev, dtype=dt))
        self.pr = SimpleNamespace(S=n, N=N, f=z(n, R), u=z(cap, A), v=z(cap, R), off=z(n, dt=torch.int32), lens=z(n, dt=torch.int32))
        if fb is not None:
            e = lambda *s, dt=torch.float32: torch.empty(*s, device=dev, dtype=dt)
            self.score_out, self.keep_out = e(n), z(n, dt=torch.long)
            self.ro_sel, self.h_sel, self.fc_sel, self.idx_sel = e(n, 2 * L), e(n, A), e(n, 2 * L), e(n, N, dt=torch.long)
        self.eng = beam._BatchEngine(self.pr, P, N, int(opt.get("beam_size", 10)), m.xt_gates_table(), m.decode_snapshots())
        self.ds = beam.DeviceSearch(self.eng, T, opt)
        self._loop()                                                             # eager warm-up
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with ops.graph_capture(self.graph, dev):
            self._loop()

    def _loop(self):
        if self.fb is not None:
            self.fb.prepared(self.m, self.n, self.P, self)
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
    """The reference-shaped call (ONE image, sGPN + NMS): selection launches write into static buffers -> the one host read (how many
    sub-graphs survived) -> ONE graph replay that gathers the survivors, prepares their attention sets and runs the token
    loop / beam search.  Falls back to `select_subgraphs` + `decode` (same kernels, eager) whenever a graph does not apply."""
    T = m.seq_length
    beam_size = opt.get("beam_size", 1)
    return_att = opt.get("return_att", 0) == 1
    graphable = m.gpn and not m.sct and forced is None and getattr(m, "decode_hipgraph", True) and m.__dict__.get("tap") is None
    G_in = int(image[1].size(1) * image[1].size(2)) if m.gpn else 0
    fb = _front_buffers(m, G_in, N) if (graphable and G_in > 0) else None
    fr = score_candidates(m, X2, N, [image], fb=fb) if m.gpn else None
    if not graphable or fr.G == 0:
        sel = select_subgraphs(m, X2, N, [image], front=fr) if m.gpn else full_graph_rows(m, X2, N, [image])
        return decode(m, X2, N, sel, opt, uniforms, forced)[0]
    fb.load(X2)                                                                   # queued before the host read below
    k = m.the_k if m.topk_sampling else 0
    n_spec = min(int(m.gpn_max_subg), fr.G)
    if (beam_size <= 1 and not return_att and uniforms is None and 0 < n_spec <= 16 and getattr(m, "decode_speculate", True)):
        # SPECULATIVE replay: the NMS keeps at most gpn_max_subg sub-graphs and almost always exactly that many, so the token loop is
        # queued for n_spec rows BEFORE the host knows the survivor count -- the count travels to pinned memory ahead of the replay in
        # stream order, and the host reads it while the loop runs.  No host round trip sits between the selection and the loop any more
        # (it cost the GPU ~35 us of idling per image), and the host may run a whole image ahead.  Rows are independent in the loop; when
        # fewer sub-graphs survived, the surplus rows decoded stale candidates and are cut off, and the log-probs the reference would not
        # have written after ITS early break (AttModel.py:318-319: taken over the real rows only) are zeroed (subgc_decode_batch_finish).
        try:
            g = _graphed_loop(m, n_spec, N, k, False, m._decoder_params(), fb)
        except RuntimeError as e:
            import warnings
            warnings.warn(f"hipGraph capture of the decode loop failed ({e}); decoding eagerly from now on")
            m.decode_hipgraph = False
            n = int(fr.n_keep.item())
            return decode(m, X2, N, select_subgraphs(m, X2, N, [image], front=fr, kept=[n]), opt, uniforms, forced)[0]
        host_n = m.__dict__.get("_host_n")
        if host_n is None:
            host_n = m.__dict__["_host_n"] = torch.empty(1, dtype=torch.int32).pin_memory()
        host_n.copy_(fr.n_keep, non_blocking=True)
        arrived = torch.cuda.Event()
        arrived.record()
        seq, seqlp, _, _, score_out, keep_out = g.run(None, _uniforms(m, n_spec, T, X2.device) if k else None, step_major=True)
        arrived.synchronize()
        n = int(host_n[0])
        if n == n_spec:
            return (seq, seqlp, score_out, keep_out)
        if n == 0:
            return decode(m, X2, N, select_subgraphs(m, X2, N, [image], front=fr, kept=[0]), opt, uniforms, forced)[0]
        seq, seqlp = seq[:n], seqlp[:n]
        ops.decode_batch_finish(seq, seqlp, [0, n])
        return (seq, seqlp, score_out[:n], keep_out[:n])
    n = int(fr.n_keep.item())
    if n == 0 or (beam_size > 1 and n * beam_size > 128) or (beam_size <= 1 and n > 16):
        return decode(m, X2, N, select_subgraphs(m, X2, N, [image], front=fr, kept=[n]), opt, uniforms, forced)[0]
    P = m._decoder_params()
    try:
        if beam_size > 1:
            g = _graphed_beam(m, n, N, P, opt, fb)
            seq, seqlp, done = g.run(None)
            m.done_beams = done
            return (seq, seqlp, ops.copy_(torch.empty_like(g.score_out), g.score_out), ops.copy_(torch.empty_like(g.keep_out), g.keep_out))
        k = m.the_k if m.topk_sampling else 0
        own_u = bool(k) and uniforms is None
        if own_u:
            uniforms = _uniforms(m, n, T, X2.device)                              # step-major [T, n]
        g = _graphed_loop(m, n, N, k, return_att, P, fb)
    except RuntimeError as e:                                                     # capture unavailable here: same kernels, launched eagerly
        import warnings
        warnings.warn(f"hipGraph capture of the decode loop failed ({e}); decoding eagerly from now on")
        m.decode_hipgraph = False
        return decode(m, X2, N, select_subgraphs(m, X2, N, [image], front=fr, kept=[n]), opt, uniforms, forced)[0]
    seq, seqlp, counts, AL, score_out, keep_out = g.run(None, uniforms, step_major=own_u if k else False)
    r = (seq, seqlp, score_out, keep_out)
    if return_att:
        dead = (counts.cpu() == 0).nonzero()
        steps = int(dead[0]) + 1 if dead.numel() else T + 1
        n_max = int(g.pr.lens.max().item())
        r = r + (AL[:steps, :, :n_max].permute(1, 0, 2).contiguous(),)
    return r


def _uniforms(m, n, T, dev):
    """[T, n] (step-major) uniforms of a free-running top-k decode (AttModel.py:295-303 draws with torch.multinomial): counter-based Philox stream
    keyed by torch's seed and a per-model call counter (subgc_uniform_f32) -- reproducible under torch.manual_seed, no torch RNG kernel."""
    m.__dict__["_sample_calls"] = m.__dict__.get("_sample_calls", 0) + 1
    seed = (int(torch.initial_seed()) * 1000003 + m.__dict__["_sample_calls"]) & 0xFFFFFFFFFFFFFFFF
    return ops.uniform((T, n), seed ^ 0x3C6EF372FE94F82B, 0, dev)                     # step-major: row t = the draws of step t


@torch.no_grad()
def decode(m, X2, N, sel, opt, uniforms=None, forced=None, batch_out=None):
    """Greedy / top-k / beam decode of the selected sub-graphs of one or many images as ONE batch.
    Returns one result tuple per image: (seq, seqLogprobs, score, keep[, att_weights]).
    `batch_out` (a dict, optional): also receives the WHOLE batch's tensors in image order -- seq, seqlp, score, keep, idx (node
    lists), lens, bounds (python list of row boundaries) and, with return_att, AL (the loop's time-major attention buffer
    [T + 1, rows, N]) -- what the batched eval glue reads (eval_glue.caption_images: one ranking / grounding launch and one host copy
    per decode batch instead of per-image tensor slicing).  `batch_out["skip_att"]`: do not cut the per-image att_weights tensors."""
    dev = X2.device
    T = m.seq_length
    return_att = opt.get("return_att", 0) == 1
    beam_size = opt.get("beam_size", 1)
    sizes = [s["keep"].numel() for s in sel]
    n = sum(sizes)
    z = lambda *s, dt=torch.float32: ops.zero_(torch.empty(*s, device=dev, dtype=dt))
    if n == 0:
        if batch_out is not None:
            batch_out.update(rows=0, bounds=[0] * (len(sel) + 1))
        return [(z(0, T, dt=torch.long), z(0, T)) + (s["score"], s["keep"]) + ((z(0, 0, 0),) if return_att else ()) for s in sel]
    whole = getattr(sel, "whole", None)
    if whole is not None:                                                              # the selection's own tensors, already in image order
        fc, lens_k, idx_k, img_k = whole["fc"], whole["lens"], whole["idx"], whole["img"]
    else:
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
        if batch_out is not None:
            batch_out.update(_batch_view(sel, whole, seq, seqlp, None, idx_k, lens_k, bounds))
        return [(seq[a:b], seqlp[a:b], s["score"], s["keep"]) for s, a, b in zip(sel, bounds, bounds[1:])]
    k = m.the_k if m.topk_sampling else 0
    own_u = bool(k) and uniforms is None and forced is None
    if own_u:
        uniforms = _uniforms(m, n, T, dev)                                             # step-major [T, n]
    graphed = None
    tap = m.__dict__.get("tap")
    if len(sel) == 1 and forced is None and n <= 16 and getattr(m, "decode_hipgraph", True) and tap is None:
        # the reference-shaped call (one image, <= 10 rows): launch-bound, replayed as one hipGraph
        try:
            graphed = _graphed_loop(m, n, N, k, return_att, P)
        except RuntimeError as e:                                                # capture unavailable here: same kernels, launched eagerly
            import warnings
            warnings.warn(f"hipGraph capture of the decode loop failed ({e}); decoding eagerly from now on")
            m.decode_hipgraph = False
    if graphed is not None:
        seq, seqlp, counts, AL, _, _ = graphed.run(pr, uniforms, step_major=own_u)
    else:
        st = F_.DecodeState(pr, P, N, return_att, xt_table=m.xt_gates_table(), fuse_lstm=True, snapshots=m.decode_snapshots(), W16=m.decode_w16())
        seq, seqlp, it = z(n, T, dt=torch.long), z(n, T), z(n, dt=torch.long)
        unfinished, counts = z(n, dt=torch.int32), z(T, dt=torch.int32)
        AL = z(T + 1, n, N) if return_att else None
        ut = None
        if uniforms is not None and forced is None:                                    # step-major: a contiguous row per step for the pick kernel
            ut = uniforms if own_u else uniforms.t().contiguous()                      # (injected [n, T] uniforms: test plumbing in torch)
        if tap is not None:
            F_.tap_prepared(tap, pr, P[7])
            if AL is None:
                AL = z(T + 1, n, N)
            steps_tap = {k: [] for k in ("h_att", "h_lang", "c_att", "c_lang", "alpha", "logp")}
        for t in range(T + 1):
            logp = st.step(it, AL[t] if (return_att or tap is not None) else None, normalize=forced is not None)
            if tap is not None:                                                        # state after core step t, in the oracle's names
                R = st.R
                for key_, v_ in (("h_att", st.H1[:, R:]), ("h_lang", st.H1[:, :R]), ("c_att", st.C1[0]), ("c_lang", st.C2[0]), ("alpha", AL[t]),
                                 ("logp", logp if forced is not None else torch.log_softmax(logp, 1))):
                    steps_tap[key_].append(v_.clone())
            if t == T:
                break
            if forced is not None:
                _forced_pick(logp, forced[:, t].contiguous(), k, m.topk_temp, t, seq, seqlp, it, unfinished, counts)
            else:
                ops.decode_pick(logp, k, m.topk_temp, None if ut is None else ut[t], t, seq, seqlp, it,
                                unfinished, counts[t:t + 1], counts[t - 1:t] if t > 0 else None, raw=True)
    if tap is not None and graphed is None:
        tap.update({"step_" + key_: torch.stack(v_, 0) for key_, v_ in steps_tap.items()})
        if not return_att:
            AL = None
    want_att = return_att and not (batch_out is not None and batch_out.get("skip_att"))
    if len(sel) == 1:
        steps = None
        if want_att:
            dead = (counts.cpu() == 0).nonzero()
            steps = [int(dead[0]) + 1 if dead.numel() else T + 1]
    else:
        # per-image early break: image i stops after the first step at which none of ITS rows is unfinished; the reference writes
        # nothing (tokens, log-probs, attention rows) beyond that step: one launch masks the log-probs (subgc_decode_batch_finish)
        brk = ops.decode_batch_finish(seq, seqlp, bounds)
        steps = None
        if want_att:
            h = brk.cpu().tolist()                                                     # [I, 2]: break step, "some step had no unfinished row"
            steps = [min(b + 1, T) + (1 if b == T - 1 and not any_ else 0) for b, any_ in h]
    skip_att = False
    if batch_out is not None:
        skip_att = bool(batch_out.pop("skip_att", False))
        batch_out.update(_batch_view(sel, whole, seq, seqlp, AL if return_att else None, idx_k, lens_k, bounds))
    out = []
    for i, (s, a, b) in enumerate(zip(sel, bounds, bounds[1:])):
        r = (seq[a:b], seqlp[a:b], s["score"], s["keep"])
        if return_att and not skip_att:
            n_max = int(s["lens"].max().item()) if b > a else 0
            r = r + (AL[:steps[i], a:b, :n_max].permute(1, 0, 2).contiguous(),)
        out.append(r)
    return out


def _batch_view(sel, whole, seq, seqlp, AL, idx_k, lens_k, bounds):
    if whole is not None and "score" in whole:
        score, keep = whole["score"], whole["keep"]
    else:
        score, keep = torch.cat([s["score"].reshape(-1) for s in sel]), torch.cat([s["keep"].reshape(-1) for s in sel])
    return dict(rows=seq.size(0), seq=seq, seqlp=seqlp, score=score.reshape(-1), keep=keep.reshape(-1), idx=idx_k, lens=lens_k, bounds=list(bounds), AL=AL)
