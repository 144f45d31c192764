"""Packed (length-sorted) teacher-forced decoder + fused criterion: the LossWrapper fast path.

The reference runs every sentence through every one of the T decoder steps (AttModel.py:157-175) and
then multiplies most of that work by a zero mask in the criterion (misc/utils.py:115-124).  A step of a
sentence whose mask is zero from there on can influence neither the loss nor any gradient, so this
Function - used only when the caller wants the LOSS and not the log-probabilities, i.e. from
LossWrapper - does what cuDNN-style packed sequences do:

  * sentences are sorted by their number of live steps (descending); at step t only the first M_t rows
    are computed (M_t is non-increasing), every per-step buffer is stored packed time-major
    (offset ot[t], M_t rows), so the batched-over-time GEMMs (x_t -> gates, logits, every weight
    gradient) run over sum_t M_t rows instead of T*S;
  * the live counts come from the label masks with ONE small device->host read (T integers) per step of training;
  * loss and gradients are identical to the unpacked path (tests/test_parity_gpu.py compares both with
    the reference's golden gradients); only the [S,T,V+1] `outputs` tensor is not produced.

Same kernels, same C ABI calls as functions.DecoderFn; only the row bookkeeping differs.
"""
from __future__ import annotations

import torch
from torch.autograd import Function

from . import functions as F_
from . import ops


class Plan:
    """The packed decoder's row plan, built by ONE launch (subgc_live_plan): sentences ordered by live steps (a step t of sentence s
    is live iff mask_t[s, t'] > 0 for some t' >= t and the reference's early break, AttModel.py:171-172, has not happened), the
    live-row counts per step and their prefix, the criterion's denominator.  The T counts are the one thing the HOST needs (launch
    dimensions); they travel to pinned memory behind an event."""

    def __init__(self, labels, mask_t):
        S, T = mask_t.shape
        self.T = T
        self.perm32, self.perm, self.inv32, self.plan, self.den = ops.live_plan(labels, mask_t)
        self.offs = self.plan[T:]                                              # int32 [T+1] on the device
        self.host = torch.empty(T, dtype=torch.int32).pin_memory()
        self.host.copy_(self.plan[:T], non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record()

    def wait(self):
        """-> list of live rows per step (blocks until the counts have arrived)."""
        self.event.synchronize()
        return [int(c) for c in self.host.tolist()]


# `Plan` issued at the START of the model's forward: the counts travel while the host is still enqueueing the encoder, so the decoder
# only waits for that event.  Reading them where they are needed is a stream synchronisation after the encoder -- the host, which runs
# a few milliseconds ahead of the GPU during the encoder, then starts the ~150 decoder launches from zero and the GPU idles behind it.
PlanAhead = Plan


def live_plan(labels, mask_t):
    """-> (perm int64 [S] on the device, live rows per step as a python list, denominator tensor): the synchronous form of `Plan`."""
    pl = Plan(labels, mask_t)
    return pl.perm, pl.wait(), pl.den


class PackedDecoderLossFn(Function):
    """(labels, fc_in, X_nodes, lens, idx, img, params...) -> masked NLL (scalar)."""

    @staticmethod
    def forward(ctx, meta, labels, fc_in, X_nodes, lens, idx, img, *P):
        N, p_drop, masks = meta["N"], meta["p"], meta.get("masks") or {}
        target, mask_t = meta["crit"]
        dev = fc_in.device
        (fc0_w, fc0_b, fc2_w, fc2_b, att_w, att_b, c2a_w, c2a_b, emb, w1i, w1h, b1i, b1h, w2i, w2h, b2i, b2h,
         h2a_w, h2a_b, an_w, an_b, lg_w, lg_b) = P
        S, T = fc_in.size(0), labels.size(1) - 1
        R, E, A, V1 = w1h.size(1), emb.size(1), h2a_w.size(0), lg_w.size(0)
        scale = 1.0 / (1.0 - p_drop) if p_drop > 0 else 1.0
        k_fc, k_att, k_xt, k_out = (masks.get(k) for k in ("fc", "att", "xt", "out"))
        new = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)

        plan = meta.get("plan") or Plan(labels, mask_t)
        M = plan.wait()
        perm = plan.perm
        T_live = sum(1 for m in M if m > 0)
        ot = [0]
        for m in M:
            ot.append(ot[-1] + m)
        rows = ot[-1]
        # sorted per-sentence inputs and the packed per-step prefixes of tokens / targets / mask: one launch (subgc_packed_rows)
        labels_p, tok_all, tgt_all, msk_all, lens_p, idx_p, img_p = ops.packed_rows(labels, target, mask_t, plan.perm32, plan.offs, lens,
                                                                                    idx if idx.stride(1) == 1 else idx.contiguous(), img)
        fc_p = ops.gather_rows(fc_in.contiguous(), plan.perm32, torch.empty(S, fc_in.size(1), device=dev, dtype=torch.float32))
        X_nodes = X_nodes.contiguous()
        W, bf = F_.bf16_twins(P, meta.get("W16"))               # GEMM-operand form of every parameter (bf16 twins under compute_dtype = bf16)
        act = lambda r, c, zero=False: ops.act_buffer((r, c), dev, bf, zero)
        # shared attention sets (Full-GC): sentence s = b*g + j of image b sits at sorted position inv32[s] of every step's rows
        pr = F_.make_prepared(meta, fc_p, X_nodes, lens_p, idx_p, img_p, N, P, k_fc, k_att, scale, W if bf else None, rows=plan.inv32)

        # scheduled sampling (AttModel.py:157-167; see functions.DecoderFn): input words, x->gates and logits go step by step
        ss = meta.get("ss")
        tokens_p = labels_p
        if ss is not None:
            sel_p, u_p = ss[1].index_select(1, perm).contiguous(), ss[2].index_select(1, perm).contiguous()
        xt = act(max(rows, 1), E)
        Gx = new(max(rows, 1), 4 * R)
        tok_flat = k_flat = None
        if rows > 0:
            # the words actually fed, in packed order (ground truth; scheduled sampling overwrites the drawn ones in place) and the
            # keep-mask rows that go with them: the backward's embedding gradient is ONE launch over them in every mode
            tok_flat = tok_all[:rows]
            k_flat = None if k_xt is None else k_xt.view(-1, E)[:rows]
        if ss is None and rows > 0:
            # all T steps' input words in packed order -> ONE embedding launch (and one in the backward) instead of one per step;
            # the dropout keep-mask is random, so its first `rows` rows serve the packed rows as they are (GENERATED masks only: a
            # caller that injects masks -- the parity tests -- never gets here, AttModel._forward runs the unpacked decoder for them,
            # where mask row (t, s) belongs to step t of sentence s)
            ops.embed_fwd(emb, tok_flat, 1, k_flat, scale, xt[:rows])
            ops.gemm(xt[:rows], W[9][:, 2 * R:], Gx[:rows], tb=True)
        Gf = new(S, 4 * R)
        ops.gemm(pr.f16 if bf else pr.f, W[9][:, R:2 * R], Gf, tb=True)
        Wc1 = F_._cat_weights(W[9][:, :R], W[10])
        Wc2 = F_._cat_weights(W[13], W[14])

        # only the state entering step 0 is zero; every other row is written by the step before it is read (checked by the
        # SUBGC_POISON_EMPTY run of the GPU suite), so ~250 MB of fills per forward shrink to ~20 MB
        m0 = M[0] if T_live > 0 else 0
        H1 = ops.act_padded((rows + S, 2 * R), dev, bf, zero_rows=m0)   # packed [h2_{t-1} | h1_{t-1}], + S rows of slack after the last step
        H2 = ops.act_padded((rows + S, 3 * R), dev, bf, zero_rows=m0)   # packed [ctx_t | h1_t | h2_{t-1}]; both with a 128-byte row pitch (ops.PITCH)
        C1, C2 = new(T + 1, S, R), new(T + 1, S, R)
        for buf in (C1[0], C2[0]):
            ops.fill_(buf, 0.0)
        Hout, G1, G2 = ops.act_padded((max(rows, 1), R), dev, bf), new(max(rows, 1), 4 * R), new(max(rows, 1), 4 * R)
        AH, AL = new(max(rows, 1), A), new(max(rows, 1), N)
        pre = new(S, 4 * R)
        QP = new(8 * S * A)
        logits = new(max(rows, 1), V1)
        # The T_live steps as ONE library call (subgc_recurrence_fwd: the same per-step entry points, issued from C with pointer
        # arithmetic); the Python loop below remains for scheduled sampling (per-step logits / draws / embeddings) and for the bench's
        # FLOP-accounting pass.
        rec = None
        if ss is None and T_live > 0 and ops.recurrence_ok():
            # the entry after the last step: where that step's state rows go (one dummy row in the slack when no row is live there)
            nxt_m = M[T_live] if T_live < len(M) else 0
            rec = ops.Recurrence(S=S, T=T_live, R=R, A=A, n_alpha=AL.size(1), bf16=int(bf), gemm_flags=ops.GEMM_MODES[ops.gemm_mode.current],
                                 keep_scale=float(scale), m=list(M[:T_live]) + [nxt_m], row0=list(ot[:T_live]) + [ot[T_live] if nxt_m > 0 else rows],
                                 hout_off=[o_ * ops.ld(Hout) for o_ in ot[:T_live]], ld_hout=ops.ld(Hout), H1=H1, ldH1=ops.ld(H1), H2=H2,
                                 ldH2=ops.ld(H2), Hout=Hout, Wc1=Wc1, ldW1=ops.ld(Wc1), Wc2=Wc2, ldW2=ops.ld(Wc2), Wq=W[17], ldWq=ops.ld(W[17]),
                                 b1i=b1i, b1h=b1h, b2i=b2i, b2h=b2h, bq=h2a_b, pre=pre, Gx=Gx, Gf=Gf, C1=C1, C2=C2, G1=G1, G2=G2, AH=AH, AL=AL,
                                 k_out=k_out, QP=QP, qp_bytes=QP.numel() * 4, w_a=an_w, b_a=an_b, lens=lens_p, **pr.recur_fields())
            ops.recurrence_fwd(rec, H1)
        for t in range(T_live if rec is None else 0):
            m, o, mn = M[t], ot[t], (M[t + 1] if t + 1 < T else 0)
            o1 = ot[t + 1]
            mn_ = max(mn, 1)                                    # row limit 0 means "all" in the C ABI: write 1 dummy row into the slack
            if ss is not None:
                if t >= 1:                                      # raw logits of every row live at step t-1; draws for the rows still live now
                    op, mp = ot[t - 1], M[t - 1]
                    ops.gemm(Hout[op:op + mp], W[21], logits[op:op + mp], tb=True, bias=lg_b)
                    ops.multinomial_rows_(logits[op:op + m], u_p[t][:m], sel_p[t][:m], ss[0], tok_flat[o:o + m])
                ops.embed_fwd(emb, tok_flat[o:o + m], 1, None if k_flat is None else k_flat[o:o + m], scale, xt[o:o + m])
                ops.gemm(xt[o:o + m], W[9][:, 2 * R:], Gx[o:o + m], tb=True)
            ops.lstm_fwd_gemm(H1[o:o + m], Wc1, pre[:m], Gx[o:o + m], Gf[:m], b1i, b1h, C1[t][:m], C1[t + 1][:m], H2[o:o + m, R:2 * R],
                              H1[o1:o1 + mn_, R:], None, 1.0, None, G1[o:o + m], m, R, rows_h=m, rows_h2=mn_)
            nq, sq = ops.gemm_planes(H2[o:o + m, R:2 * R], W[17], QP, tb=True)   # the query product stays as split-K planes: the attention
            pr.attn_fwd(AH[o:o + m], an_w, an_b, lens_p, H2[o:o + m, :R], AL[o:o + m], m, A, R, q=(QP, nq, sq, h2a_b))   # kernel sums them (+ bias) into AH

            ops.lstm_fwd_gemm(H2[o:o + m], Wc2, pre[:m], None, None, b2i, b2h, C2[t][:m], C2[t + 1][:m], H1[o1:o1 + mn_, :R],
                              H2[o1:o1 + mn_, 2 * R:], None if k_out is None else k_out[t], scale, Hout[o:o + m], G2[o:o + m], m, R,
                              rows_h=mn_, rows_h2=mn_)
        if ss is None:
            ops.gemm(Hout[:rows], W[21], logits[:rows], tb=True, bias=lg_b)
        elif T_live > 0:
            op = ot[T_live - 1]
            ops.gemm(Hout[op:rows], W[21], logits[op:rows], tb=True, bias=lg_b)
        lse = ops.row_lse(logits[:rows])                                  # log_softmax without the write: the loss needs lse and one logit per row
        # criterion over the packed rows: row ot[t] + s  <->  (sentence perm[s], step t); the denominator is the sum of ALL mask
        # entries (dead ones included: the early break can cut live mask entries off), which the plan kernel computed
        tgt_p, msk_p = tgt_all[:max(rows, 1)], msk_all[:max(rows, 1)]
        loss, nll = ops.masked_nll_fwd(logits[:rows].view(rows, 1, V1), tgt_p, msk_p, den=plan.den, lse=lse)

        ctx.meta = (N, scale, S, T, T_live, R, E, A, V1, M, ot, rows)
        ctx.masks = (k_xt, k_out)
        ctx.flat_tokens = (tok_flat, k_flat)
        ctx.W, ctx.bf = W, bf
        ctx.pr, ctx.params, ctx.aux = pr, P, (plan, tokens_p, lens_p, tgt_p, msk_p, nll, lse)      # tokens_p: the words actually fed
        ctx.save_for_backward(fc_p, X_nodes, logits, xt, Gf, Wc1, Wc2, H1, H2, C1, C2, Hout, G1, G2, AH, AL)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        N, scale, S, T, T_live, R, E, A, V1, M, ot, rows = ctx.meta
        k_xt, k_out = ctx.masks
        tok_flat, k_flat = ctx.flat_tokens
        pr, P, W, bf = ctx.pr, ctx.params, ctx.W, ctx.bf
        plan, labels_p, lens_p, tgt_p, msk_p, nll, lse = ctx.aux
        (fc_p, X_nodes, logp, xt, Gf, Wc1, Wc2, H1, H2, C1, C2, Hout, G1, G2, AH, AL) = ctx.saved_tensors
        (fc0_w, fc0_b, fc2_w, fc2_b, att_w, att_b, c2a_w, c2a_b, emb, w1i, w1h, b1i, b1h, w2i, w2h, b2i, b2h,
         h2a_w, h2a_b, an_w, an_b, lg_w, lg_b) = P
        dev = logp.device
        new = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
        zer = lambda *s: ops.zeros(*s, device=dev)
        act = lambda r, c: ops.act_buffer((r, c), dev, bf)      # gradients that only GEMMs (and bias sums) read
        opnd = (lambda t: ops.as_b16(t)) if bf else (lambda t: t)
        dst, acc, ret = [], [], []
        ops.GRAD_WRITES[0] += 1                                 # raw-pointer gradient writes follow (AttModel.flatten_grads' zero marker)
        for prm in P:                                           # accumulate straight into the flat gradient bucket when it exists
            g = prm.grad if F_.DIRECT_GRADS else None
            ok = g is not None and g.is_contiguous() and g.dtype == torch.float32 and g.device == dev
            dst.append(g if ok else None); acc.append(ok); ret.append(None)

        def out_for(i, zero=False):
            if dst[i] is None:
                dst[i] = zer(*P[i].shape) if zero else new(*P[i].shape)
                ret[i] = dst[i]
            return dst[i]

        def wgrad(i, dy, x, cols=None, bias=None, m_dev=None):      # dW_i[:, cols] (+)= dy^T x  [and db_bias (+)= column sums of dy, same launch]
            o = out_for(i)
            o = o if cols is None else o[:, cols[0]:cols[1]]
            if bias is None or not F_.FOLD_BIAS_SUMS:
                ops.gemm(dy, x, o, ta=True, accum=acc[i], m_dev=m_dev)
                if bias is not None:
                    bgrad(bias, dy, m_dev=m_dev)
            else:
                ops.wgrad(dy, x, o, out_for(bias).view(-1), accum=acc[i], db_accum=acc[bias], m_dev=m_dev)

        def bgrad(i, x, m_dev=None, also=None):
            if also is None:
                ops.colsum(x, out=out_for(i), accumulate=acc[i], m_dev=m_dev)
                return
            tmp = ops.colsum(x, m_dev=m_dev).view(1, -1)
            for j in (i, also):
                ops.copy2d(tmp, out_for(j).view(1, -1), accumulate=acc[j])

        dlogits = ops.empty_b16(max(rows, 1), V1, dev) if bf else new(max(rows, 1), V1)      # bf16: half the bytes of the largest tensor
        ops.nll_logsoftmax_bwd(logp[:rows], tgt_p, msk_p, nll, dloss.contiguous(), dlogits[:rows], None, rows, 1, V1, lse=lse)
        wgrad(21, dlogits[:rows], Hout[:rows], bias=22)
        F_.grads_ready("logit")                                 # logit.* is final: its all-reduce overlaps the whole BPTT loop
        dHout = new(max(rows, 1), R); ops.gemm(dlogits[:rows], W[21], dHout[:rows])
        del dlogits

        dP1, dP2, dAH = act(max(rows, 1), 4 * R), act(max(rows, 1), 4 * R), act(max(rows, 1), A)
        # d(v) is formed once after the loop from the kept d(ctx) rows (see DecoderFn.backward)
        defer_dv = pr.shared or (R % 4 == 0 and A % 4 == 0 and A <= 1024 and R <= 2048 and T_live > 0)      # the float4 forms' limits
        defer_du = F_.DEFER_DU and not pr.shared and defer_dv and T_live > 0      # see DecoderFn.backward
        dE = new(max(rows, 1), AL.size(1)) if defer_du else None
        du = new(pr.u.size(0), A) if defer_du else pr.new_du(A)
        dv = new(pr.v.size(0), R) if defer_dv else zer(pr.v.size(0), R)
        dCtx = new(max(rows, 1), R) if defer_dv else None
        dWa, dBa = new(max(rows, 1), A), new(max(rows, 1))     # per-(step, sentence) partials of alpha_net's gradient
        # The recurrent data-gradient products stay as split-K partial PLANES that their consumers add on load (see DecoderFn.backward).
        # Rows that are dead at step t+1 but live at step t enter the recurrence with zero state-gradient: a plane source only
        # contributes to the rows its own step had (`rows` of the window), and the cell-state ping-pong starts zeroed with a row
        # >= M[t+1] never written before step t reads it.
        PA, PB, PC = new(8 * S * 3 * R), new(8 * S * R), new(8 * S * 2 * R)
        sA = sC = None
        win = lambda st, col0: None if st is None else (st[0], st[1], col0, st[2], st[3], st[4])
        arena = zer(4 * S * R)
        dC1, dC2 = [arena[:S * R].view(S, R), arena[S * R:2 * S * R].view(S, R)], [arena[2 * S * R:3 * S * R].view(S, R), arena[3 * S * R:].view(S, R)]
        F_.note("bptt_begin", T_live)
        rec = None
        if T_live > 0 and ops.recurrence_ok():
            rec = ops.Recurrence(S=S, T=T_live, R=R, A=A, n_alpha=AL.size(1), bf16=int(bf), gemm_flags=ops.GEMM_MODES[ops.gemm_mode.current],
                                 keep_scale=float(scale), m=list(M[:T_live]) + [0], row0=list(ot[:T_live]) + [rows],
                                 dhout_off=[o_ * R for o_ in ot[:T_live]], ld_dhout=R, Wc1=Wc1, ldW1=ops.ld(Wc1), Wc2=Wc2, ldW2=ops.ld(Wc2),
                                 Wq=W[17], ldWq=ops.ld(W[17]), C1=C1, C2=C2, G1=G1, G2=G2, AH=AH, AL=AL, k_out=k_out, w_a=an_w, lens=lens_p, dHout=dHout,
                                 dP1=dP1, dP2=dP2, dAH=dAH, du=du, du_planes=du.size(0) if du.dim() == 3 else 1,
                                 du_plane_stride=du.stride(0) if du.dim() == 3 else 0, dv=None if defer_dv else dv, dWa=dWa, dBa=dBa,
                                 dCtx=dCtx if defer_dv else None, dE=dE, PA=PA, pa_bytes=PA.numel() * 4, PB=PB, pb_bytes=PB.numel() * 4, PC=PC,
                                 pc_bytes=PC.numel() * 4, dC1_in=dC1[0], dC1_out=dC1[1], dC2_in=dC2[0], dC2_out=dC2[1], **pr.recur_fields())
            ops.recurrence_bwd(rec)
        for t in range(T_live - 1, -1, -1):
            if rec is not None:
                break
            m, o = M[t], ot[t]
            nC1, cC1 = dC1; nC2, cC2 = dC2
            ops.lstm_bwd_planes(G2[o:o + m], C2[t][:m], C2[t + 1][:m], [win(sC, 0), win(sA, 2 * R)], dHout[o:o + m],
                                None if k_out is None else k_out[t], scale, nC2[:m], dP2[o:o + m], cC2[:m], m, R)
            n, st = ops.gemm_planes(dP2[o:o + m], Wc2, PA)
            sA = (PA, 3 * R, n, st, m)
            pr.attn_bwd(AH[o:o + m], an_w, lens_p, AL[o:o + m], win(sA, 0), dAH[o:o + m], du, None if defer_dv else dv,
                        dWa[o:o + m], dBa[o:o + m], m, A, R, dCtx[o:o + m] if defer_dv else None, **({"de_keep": dE[o:o + m]} if defer_du else {}))
            n, st = ops.gemm_planes(dAH[o:o + m], W[17], PB)
            ops.lstm_bwd_planes(G1[o:o + m], C1[t][:m], C1[t + 1][:m], [win(sA, R), (PB, R, 0, n, st, m), win(sC, R)], None, None, 1.0, nC1[:m],
                                dP1[o:o + m], cC1[:m], m, R)
            n, st = ops.gemm_planes(dP1[o:o + m], Wc1, PC)
            sC = (PC, 2 * R, n, st, m)
            dC1.reverse(); dC2.reverse()
        F_.note("bptt_end")

        P1, P2, H1a, H2a = dP1[:rows], dP2[:rows], H1[:rows], H2[:rows]
        wgrad(13, P2, H2a[:, :2 * R], bias=15)             # b_ih and b_hh have the same gradient: one sum rides each product
        wgrad(14, P2, H2a[:, 2 * R:], bias=16)
        wgrad(9, P1, H1a[:, :R], cols=(0, R), bias=11)
        step_off = plan.offs                                    # int32 [T+1] on the device; steps past T_live repeat `rows`
        if defer_dv:
            pr.dv_accum(AL, dCtx, step_off, max(T_live, 1), lens_p, dv, S, R)
            del dCtx
        if defer_du:
            pr.du_accum(AH, dE, step_off, max(T_live, 1), lens_p, an_w, du, S, A)
            del dE
        dGf = new(S, 4 * R)                                     # d(fc->gates) = sum over each sentence's live steps of dP1: one launch
        ops.packed_time_sum(dP1, step_off, T_live, S, dGf)
        dGf = opnd(dGf)
        wgrad(9, dGf, pr.f16 if bf else pr.f, cols=(R, 2 * R))
        wgrad(9, P1, xt[:rows], cols=(2 * R, 2 * R + E))
        wgrad(10, P1, H1a[:, R:], bias=12)
        df = new(S, R); ops.gemm(dGf, W[9][:, R:2 * R], df)
        dxt = new(max(rows, 1), E); ops.gemm(P1, W[9][:, 2 * R:], dxt[:rows])
        d_emb = out_for(8, zero=True)
        for t in range(T_live):
            if tok_flat is not None:
                break
            ops.embed_bwd(emb, labels_p[:, t], labels_p.stride(0), None if k_xt is None else k_xt[t], scale, dxt[ot[t]:ot[t + 1]], d_emb)
        if tok_flat is not None:
            ops.embed_bwd(emb, tok_flat, 1, k_flat, scale, dxt[:rows], d_emb)
        wgrad(17, dAH[:rows], H2a[:, R:2 * R], bias=18)
        ops.colsum(dWa[:rows], out=out_for(19).view(-1), accumulate=acc[19])
        ops.colsum(dBa[:rows].view(-1, 1), out=out_for(20).view(-1), accumulate=acc[20])
        F_.grads_ready("recurrent")

        dX, dfc_p = F_.prepared_backward(pr, P, W, bf, fc_p, X_nodes, pr.finish_du(du), dv, df, scale, out_for, acc, wgrad, bgrad,
                                         ctx.needs_input_grad[3], ctx.needs_input_grad[2])
        dfc_in = None if dfc_p is None else ops.gather_rows(dfc_p, plan.inv32, torch.empty_like(dfc_p))      # back to the caller's order
        ctx.pr = None
        F_.grads_ready("prepare")                               # the last decoder slice; the encoder's backward follows
        return (None, None, dfc_in, dX, None, None, None) + tuple(ret)
