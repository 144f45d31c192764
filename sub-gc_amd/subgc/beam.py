"""Beam search / diverse beam search on the HIP decode path.

Reference: CaptionModel.beam_search (models/CaptionModel.py:28-176) driven per sub-graph by
AttModel._sample_sentences (models/AttModel.py:179-234).  The reference walks the sub-graphs one by
one, sorts a [beam, V+1] matrix on the CPU-visible side every step and forks Python lists of states.

Here every sub-graph and every beam of one image is a row of ONE decode batch (rows = n_subgraphs x beam):
the attention rows of a sub-graph are shared by its beams through the (offset, length) table, a step is
the same kernel sequence as greedy decode, `subgc_row_topk_f32` hands back the leading beam+2 log-probs of
every row, the candidate bookkeeping (tiny: beam^2 numbers per sub-graph) runs on the host in the
reference's exact order and arithmetic (fp32 sums, stable sort over the column-major candidate list), and
the fork "beam q -> slot vix" is one row gather of the recurrent state.

Why beam+2 columns are enough: a group's augmented row differs from the raw one only at the UNK column
(-1000, CaptionModel.py:137), at the previous word (decoding_constraint, :134-135) and at the <= beam-bdash
distinct words the earlier groups picked (:33-40), all of which only move DOWN; the top bdash of the
augmented row therefore lie within the top (bdash + those) <= beam+2 of the raw row.
"""
from types import SimpleNamespace

import numpy as np
import torch

from . import functions as F_
from . import ops

_F32 = np.float32


def penalty_builder(cfg):
    """misc/utils.py:145-171."""
    if cfg == "":
        return lambda length, logprob: logprob
    kind, alpha = cfg.split("_")
    alpha = float(alpha)
    if kind == "wu":
        return lambda length, logprob: logprob / (((5 + length) ** alpha) / ((5 + 1) ** alpha))
    if kind == "avg":
        return lambda length, logprob: logprob / length
    raise ValueError(f"unknown length_penalty {cfg!r} (expected '', 'wu_<alpha>' or 'avg_<alpha>')")


class _Group:
    """One beam group of one sub-graph: the tables of CaptionModel.py:106-108."""
    __slots__ = ("seq", "lps", "sums", "done")

    def __init__(self, T, bd):
        self.seq = np.zeros((T, bd), np.int64)
        self.lps = np.zeros((T, bd), _F32)
        self.sums = np.zeros(bd, _F32)
        self.done = []


def _beam_step(grp, vals, idx, tau, bd, unk, constraint, earlier, lam):
    """CaptionModel.py:44-94 on the leading columns.  vals/idx: [bd, kk] raw log-probs (descending) and
    their word ids.  Returns the source beam of every new slot."""
    vals = vals.copy()
    if constraint and tau > 0:                                                   # :134-135
        vals[idx == grp.seq[tau - 1][:, None]] = -np.inf
    vals[idx == unk] -= _F32(1000)                                               # :137
    unaug = vals.copy()
    for prev in earlier:                                                         # add_diversity, :33-40
        for tok in prev.seq[tau]:
            vals[idx == tok] -= lam
    order = np.argsort(-vals, axis=1, kind="stable")
    rows = 1 if tau == 0 else bd
    cols = min(bd, vals.shape[1])
    cands = []
    for c in range(cols):
        for q in range(rows):
            j = order[q, c]
            cands.append((_F32(grp.sums[q] + vals[q, j]), q, int(idx[q, j]), unaug[q, j]))
    cands.sort(key=lambda x: -x[0])                                              # stable, like sorted() at :73
    prev_seq, prev_lps = grp.seq[:tau].copy(), grp.lps[:tau].copy()
    src = []
    for vix in range(bd):
        p, q, tok, r = cands[vix]
        grp.seq[:tau, vix] = prev_seq[:, q]
        grp.lps[:tau, vix] = prev_lps[:, q]
        grp.seq[tau, vix] = tok
        grp.lps[tau, vix] = r
        grp.sums[vix] = p
        src.append(q)
    return src


class _BatchEngine:
    """All sub-graphs x beams of an image batch as rows of one DecodeState (the product path)."""

    def __init__(self, pr, P, N, beam, xt_table=None, snapshots=None):
        n, dev = pr.S, pr.f.device
        self.n, self.rows, self.dev = n, n * beam, dev
        rep = torch.arange(n, device=dev).repeat_interleave(beam)
        prb = SimpleNamespace(S=self.rows, N=N, f=pr.f.index_select(0, rep).contiguous(), u=pr.u, v=pr.v,
                              off=pr.off.index_select(0, rep).contiguous(), lens=pr.lens.index_select(0, rep).contiguous())
        self.pr, self.prb, self.rep = pr, prb, rep
        self.st = F_.DecodeState(prb, P, N, False, xt_table=xt_table, fuse_lstm=True, snapshots=snapshots)    # fused for <= 32 rows
        self.V1 = self.st.V1

    def refresh(self):
        """Re-derive the per-beam rows from `pr` and restart the recurrent state (hipGraph replay: `pr` was overwritten in place)."""
        ops.take_rows([(self.pr.f, self.prb.f), (self.pr.off, self.prb.off), (self.pr.lens, self.prb.lens)], self.rep)      # one launch
        self.st.reset()

    def _topk(self, logits, kk):
        if not hasattr(self, "vals"):
            self.vals = torch.empty(self.rows, kk, device=self.dev, dtype=torch.float32)
            self.idx = torch.empty(self.rows, kk, device=self.dev, dtype=torch.int32)
        ops.row_topk(logits, kk, self.vals, self.idx, log_softmax=True)
        return self.vals.cpu().numpy(), self.idx.cpu().numpy()

    def first(self, kk):                                                         # <bos>, AttModel.py:223-227
        return self._topk(self.st.step(torch.zeros(self.rows, device=self.dev, dtype=torch.long), None, normalize=False), kk)

    def advance(self, tok, kk):
        return self._topk(self.st.step(torch.from_numpy(tok).to(self.dev), None, normalize=False), kk)

    def reorder(self, src):
        self.st.reorder(torch.from_numpy(src).to(self.dev))

    def snapshot(self):
        return [x.clone() for x in self.st.recurrent()]

    def restore(self, rows, snap):
        sel = torch.from_numpy(rows.astype(np.int64)).to(self.dev)
        for cur, saved in zip(self.st.recurrent(), snap):
            cur[sel] = saved[sel]


class _StepEngine:
    """The reference's own calling convention (CaptionModel.beam_search): a (h, c) state of shape [2, beam, R] that the
    CALLER owns, advanced through `model.get_logprobs_state` -- one sub-graph, `beam` rows."""

    def __init__(self, model, init_state, init_logprobs, args):
        self.m, self.state, self.logp, self.args = model, tuple(init_state), init_logprobs, args
        self.n, self.rows, self.dev, self.V1 = 1, init_logprobs.size(0), init_logprobs.device, init_logprobs.size(1)

    def _topk(self, logp, kk):
        vals = torch.empty(self.rows, kk, device=self.dev, dtype=torch.float32)
        idx = torch.empty(self.rows, kk, device=self.dev, dtype=torch.int32)
        ops.row_topk(logp.float().contiguous(), kk, vals, idx, log_softmax=False)
        return vals.cpu().numpy(), idx.cpu().numpy()

    def first(self, kk):
        return self._topk(self.logp, kk)

    def advance(self, tok, kk):
        logp, self.state = self.m.get_logprobs_state(torch.from_numpy(tok).to(self.dev), *self.args, self.state)
        return self._topk(logp, kk)

    def reorder(self, src):
        sel = torch.from_numpy(src.astype(np.int64)).to(self.dev)
        self.state = tuple(s.index_select(1, sel) for s in self.state)

    def snapshot(self):
        return [s.clone() for s in self.state]

    def restore(self, rows, snap):
        sel = torch.from_numpy(rows.astype(np.int64)).to(self.dev)
        self.state = tuple(s.clone() for s in self.state)
        for cur, saved in zip(self.state, snap):
            cur[:, sel] = saved[:, sel]


@torch.no_grad()
def beam_decode(pr, P, N, T, opt, xt_table=None, on_device=True):
    """Decode every sub-graph of `pr` (an F_.Prepared) with beam search.
    Returns (seq [n, T] int64, seqLogprobs [n, T] fp32, done_beams) -- CPU tensors, as the reference's are.
    `on_device` (default): the candidate bookkeeping runs in `subgc_beam_step`, the loop has no host round trip;
    False keeps it on the host (`search`, the restatement the device kernel is tested against)."""
    eng = _BatchEngine(pr, P, N, int(opt.get("beam_size", 10)), xt_table)
    return search_device(eng, T, opt) if on_device else search(eng, T, opt)


class DeviceTables:
    """The tables of CaptionModel.py:106-109 for n sub-graphs x G groups, resident on the device."""

    def __init__(self, n, G, T, bd, dev):
        z = lambda *s, dt: ops.zero_(torch.empty(*s, device=dev, dtype=dt))
        self.cap = bd * T                                                         # a group finishes at most bd beams per step
        self.seq, self.lps, self.sums = z(n, G, T, bd, dt=torch.int32), z(n, G, T, bd, dt=torch.float32), z(n, G, bd, dt=torch.float32)
        # the finished-beam tables are views of ONE 4-byte buffer: `collect` brings them to the host with a single copy (five
        # synchronising copies cost 0.15 ms of a 2.8 ms one-image search)
        cap = self.cap
        sizes = [n * G, n * G * cap * T, n * G * cap * T, n * G * cap, n * G * cap]
        self.done_all = z(sum(sizes), dt=torch.int32)
        parts, o = [], 0
        for k in sizes:
            parts.append(self.done_all[o:o + k]); o += k
        self.done_cnt = parts[0].view(n, G)
        self.done_seq, self.done_lps = parts[1].view(n, G, cap, T), parts[2].view(torch.float32).view(n, G, cap, T)
        self.done_p, self.done_len = parts[3].view(torch.float32).view(n, G, cap), parts[4].view(n, G, cap)
        self._sizes = sizes

    def done_to_host(self):
        """(cnt, seq, lps, p, len) as numpy arrays after ONE device -> host copy."""
        n, G = self.done_cnt.shape
        cap, T = self.cap, self.done_seq.size(3)
        h = self.done_all.cpu().numpy()
        out, o = [], 0
        for k in self._sizes:
            out.append(h[o:o + k]); o += k
        return (out[0].reshape(n, G), out[1].reshape(n, G, cap, T), out[2].view(np.float32).reshape(n, G, cap, T),
                out[3].view(np.float32).reshape(n, G, cap), out[4].reshape(n, G, cap))


def _check(opt, eng):
    beam = int(opt.get("beam_size", 10))
    G = int(opt.get("group_size", 1))
    if G < 1 or beam % G:
        raise ValueError(f"beam_size {beam} must be a multiple of group_size {G}")
    bd = beam // G
    if eng.rows != eng.n * beam:
        raise ValueError(f"{eng.rows} state rows for {eng.n} sub-graph(s) x beam {beam}")
    kk = min(eng.V1, beam + 2)
    if bd > kk:
        raise ValueError("beam wider than the vocabulary")
    return beam, G, bd, kk


class DeviceSearch:
    """`search` with the per-step bookkeeping in `subgc_beam_step`: every step is top-k -> beam step -> state gather ->
    decoder step on the stream (`loop`: launches only, no host read, so it can be captured in a hipGraph); the host reads
    the finished beams once, after the last step (`collect`)."""

    def __init__(self, eng, T, opt):
        self.eng, self.T, self.opt = eng, T, dict(opt)
        self.beam, self.G, self.bd, self.kk = _check(opt, eng)
        self.lam = float(_F32(opt.get("diversity_lambda", 0.5)))
        self.constraint = opt.get("decoding_constraint", 0)
        n, rows, dev, G, bd, kk = eng.n, eng.rows, eng.dev, self.G, self.bd, self.kk
        self.tb = DeviceTables(n, G, T, bd, dev)
        self.tok = ops.zero_(torch.empty(rows, device=dev, dtype=torch.long))
        self.src = ops.zero_(torch.empty(rows, device=dev, dtype=torch.int32))
        self.tv = torch.empty(rows, kk, device=dev, dtype=torch.float32)
        self.ti = torch.empty(rows, kk, device=dev, dtype=torch.int32)
        if G > 1:
            self.nv, self.ni = torch.empty_like(self.tv), torch.empty_like(self.ti)
            base = torch.arange(rows, device=dev).view(n, G, bd)
            self.group_rows = [base[:, g].reshape(-1).contiguous() for g in range(G)]

    def loop(self, fresh_tables=False):
        eng, T, G, bd, kk, tb = self.eng, self.T, self.G, self.bd, self.kk, self.tb
        n, unk = eng.n, eng.V1 - 1
        tv, ti, tok, src = self.tv, self.ti, self.tok, self.src
        if not fresh_tables:
            for x in (tb.seq, tb.lps, tb.sums, tb.done_cnt, tok):
                ops.zero_(x)
        ops.row_topk(eng.st.step(tok, None, normalize=False), kk, tv, ti, log_softmax=True)      # <bos>, AttModel.py:223-227
        if G > 1:
            init = eng.snapshot()
            tv4, ti4, nv4, ni4 = (x.view(n, G, bd, kk) for x in (tv, ti, self.nv, self.ni))
        for t in range(T + G - 1):
            ops.beam_step(tv, ti, tb, tok, src, t, T, G, bd, kk, unk, self.constraint, self.lam)
            if not any(g <= t + 1 <= T + g - 1 for g in range(G)):
                break
            if G > 1 and 0 < t < G:                                               # group t starts now: give it the post-<bos> state
                sel = self.group_rows[t]
                for cur, saved in zip(eng.st.recurrent(), init):
                    cur.index_copy_(0, sel, saved.index_select(0, sel))
            eng.st.reorder(src)
            logits = eng.st.step(tok, None, normalize=False)
            if G == 1:
                ops.row_topk(logits, kk, tv, ti, log_softmax=True)
            else:
                ops.row_topk(logits, kk, self.nv, self.ni, log_softmax=True)
                for g in range(G):
                    if g <= t <= T + g - 1:                                       # only the groups that stepped take the new rows
                        tv4[:, g], ti4[:, g] = nv4[:, g], ni4[:, g]

    def collect(self):
        tb, T, G, bd, n, opt = self.tb, self.T, self.G, self.bd, self.eng.n, self.opt
        length_penalty = penalty_builder(opt.get("length_penalty", ""))
        # one read of the finished beams; ranking (:174-175) vectorised, python objects only for the beams that are kept
        cnt, dseq, dlps, dp, dlen = tb.done_to_host()
        valid = np.arange(tb.cap)[None, None, :] < cnt[:, :, None]
        if opt.get("length_penalty", "") == "":
            p = dp.astype(np.float64)
        else:                                                                         # the reference's python-float arithmetic, entry by entry
            p = np.zeros(dp.shape, np.float64)
            for s_, g_, j_ in zip(*np.nonzero(valid)):
                p[s_, g_, j_] = length_penalty(int(dlen[s_, g_, j_]), float(dp[s_, g_, j_]))
        if n and int(cnt.min()) < bd:
            raise RuntimeError("beam search ended with fewer finished beams than slots")   # the last step finishes every slot (:151)
        order = np.argsort(np.where(valid, -p, np.inf), axis=-1, kind="stable")[:, :, :bd]
        oi = order[..., None]                                                         # numpy gathers: single-threaded, no thread-pool wake-ups
        top_seq = torch.from_numpy(np.take_along_axis(dseq, oi, 2).astype(np.int64))  # [n, G, bd, T]
        top_lps = torch.from_numpy(np.take_along_axis(dlps, oi, 2))
        top_p = np.take_along_axis(p, order, -1).reshape(-1).tolist()
        top_un = np.take_along_axis(dlps.sum(-1, dtype=_F32), order, -1).reshape(-1).tolist()
        rs, rl = top_seq.reshape(-1, T).unbind(0), top_lps.reshape(-1, T).unbind(0)
        per = G * bd
        done_beams = [[{"seq": rs[k], "logps": rl[k], "unaug_p": top_un[k], "p": top_p[k]} for k in range(s * per, (s + 1) * per)]
                      for s in range(n)]
        seq, seqlp = top_seq[:, 0, 0].clone(), top_lps[:, 0, 0].clone()
        return seq, seqlp, done_beams


@torch.no_grad()
def search_device(eng, T, opt):
    ds = DeviceSearch(eng, T, opt)
    ds.loop(fresh_tables=True)
    return ds.collect()


@torch.no_grad()
def search(eng, T, opt):
    """The bookkeeping of CaptionModel.beam_search (:97-176) over an engine that owns the recurrent state."""
    beam = int(opt.get("beam_size", 10))
    G = int(opt.get("group_size", 1))
    lam = opt.get("diversity_lambda", 0.5)
    constraint = opt.get("decoding_constraint", 0)
    length_penalty = penalty_builder(opt.get("length_penalty", ""))
    if G < 1 or beam % G:
        raise ValueError(f"beam_size {beam} must be a multiple of group_size {G}")
    lam = _F32(lam)
    bd = beam // G
    n, rows, V1 = eng.n, eng.rows, eng.V1
    if rows != n * beam:
        raise ValueError(f"{rows} state rows for {n} sub-graph(s) x beam {beam}")
    unk = V1 - 1
    kk = min(V1, beam + 2)
    if bd > kk:
        raise ValueError("beam wider than the vocabulary")
    shape = (n, G, bd, kk)
    tv, ti = (x.reshape(shape).copy() for x in eng.first(kk))
    init = eng.snapshot() if G > 1 else None
    groups = [[_Group(T, bd) for _ in range(G)] for _ in range(n)]
    base = np.arange(rows, dtype=np.int32).reshape(n, G, bd)
    for t in range(T + G - 1):
        live = [g for g in range(G) if g <= t <= T + g - 1]
        src = base.copy()
        tok = np.zeros((n, G, bd), np.int64)
        for g in live:
            tau = t - g
            for s in range(n):
                grp = groups[s][g]
                q = _beam_step(grp, tv[s, g], ti[s, g], tau, bd, unk, constraint, groups[s][:g], lam)
                src[s, g] = base[s, g][q]
                for vix in range(bd):                                            # :150-166
                    if grp.seq[tau, vix] == 0 or t == T + g - 1:
                        final = {"seq": grp.seq[:, vix].copy(), "logps": grp.lps[:, vix].copy(),
                                 "unaug_p": float(grp.lps[:, vix].sum(dtype=_F32)), "p": float(grp.sums[vix])}
                        final["p"] = length_penalty(tau + 1, final["p"])
                        grp.done.append(final)
                        grp.sums[vix] = -1000
                tok[s, g] = grp.seq[tau]
        if not any(g <= t + 1 <= T + g - 1 for g in range(G)):
            break                                                                # the reference's last step (:170-171) feeds nothing
        if G > 1 and 0 < t < G:                                                  # group t starts now: give it the post-<bos> state
            eng.restore(base[:, t].reshape(-1), init)
        eng.reorder(src.reshape(-1))
        nv, ni = (x.reshape(shape) for x in eng.advance(tok.reshape(-1), kk))
        for g in live:
            tv[:, g], ti[:, g] = nv[:, g], ni[:, g]
    seq = torch.zeros(n, T, dtype=torch.long)
    seqlp = torch.zeros(n, T, dtype=torch.float32)
    done_beams = []
    for s in range(n):
        beams = []
        for g in range(G):
            beams += sorted(groups[s][g].done, key=lambda x: -x["p"])[:bd]       # :174-175
        for b in beams:
            b["seq"], b["logps"] = torch.from_numpy(b["seq"]), torch.from_numpy(b["logps"])
        done_beams.append(beams)
        seq[s], seqlp[s] = beams[0]["seq"], beams[0]["logps"]
    return seq, seqlp, done_beams
