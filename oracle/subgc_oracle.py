"""CPU ORACLE for the Sub-GC hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain PyTorch-CPU fp32 (+ numpy for the integer work) restatement of the algorithm of the
reference's `models/AttModel.py`, `models/lib/{gcn_backbone,graph_conv,graph_conv_unit,gpn}.py`,
`models/loss_wrapper.py` and `misc/utils.py:LanguageModelCriterion`, written op-for-op the way
the reference computes it (dense 0/1 incidence `bmm`, x5 expansion copies, python step loop,
python-set NMS).  Each function cites the reference lines it follows.

PARITY PIN: this file is checked against golden vectors produced by running the reference
itself (tests/golden/make_golden.py -> tests/golden/*.npz; tests/test_oracle_golden.py), for
train (outputs, losses, every intermediate, every parameter gradient), greedy / top-k / sct /
return_att decode, NMS and the Full-GC (BatchNorm) variant.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module, and only as the checker / the timed CPU baseline.  Nothing under `sub-gc_amd/` imports it.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- options
def opt_get(opt, name, default=None):
    return getattr(opt, name, default) if not isinstance(opt, dict) else opt.get(name, default)


class Cfg:
    """The option fields the model reads (AttModel.py:44-69,96-98)."""

    def __init__(self, opt):
        g = lambda n, d=None: opt_get(opt, n, d)
        self.vocab = g("vocab_size"); self.R = g("rnn_size"); self.E_in = g("input_encoding_size")
        self.p_lm = g("drop_prob_lm", 0.5)
        self.seq_length = g("max_length") or g("seq_length")           # AttModel.py:49
        self.L = g("gcn_dim"); self.A = g("att_hid_size")
        self.gpn = g("use_gpn", 1) == 1
        self.noun_fuse = g("noun_fuse", 1) == 1
        self.pred_emb_type = g("pred_emb_type", 1)
        self.layers = g("gcn_layers", 2); self.residual = g("gcn_residual", 2)
        self.bn = g("gcn_bn", 0) != 0
        self.test_lstm = g("test_LSTM", 0) != 0
        self.topk = g("use_topk_sampling", 0) != 0
        self.topk_temp = g("topk_temp", 0.6); self.the_k = g("the_k", 3)
        self.sct = g("sct", 0) != 0
        self.nms_thres = g("gpn_nms_thres", 0.75); self.max_subg = g("gpn_max_subg", 1)
        self.use_score = g("use_gt_subg", 0) == 0
        self.ss_prob = g("sampling_prob", 0.0)
        self.p_gpn = g("gpn_drop_prob", 0.5)      # nn.Dropout(0.5) in gpn_fc (gpn.py:27); tests pin it to 0


def _drop(x, p, training, mask):
    """nn.Dropout with an optionally injected keep-mask (1 = keep), scale 1/(1-p)."""
    if not training or p == 0.0:
        return x
    if mask is None:
        return F.dropout(x, p, True)
    return x * mask.to(x.dtype) * (1.0 / (1.0 - p))


# ----------------------------------------------------------------------------- encoder
def feat_fusion(P, cfg, obj_dist, att_feats, pred_dist):
    """AttModel.py:370-387."""
    n_obj = obj_dist.size(-1); n_pred = pred_dist.size(-1)
    if cfg.noun_fuse:
        cls = obj_dist.reshape(-1, n_obj)[:, 1:].max(1)[1] + 1                      # :376
        emb = F.linear(F.embedding(cls, P["sg_obj_embed.weight"]), P["obj_emb_proj.weight"], P["obj_emb_proj.bias"])
        x = F.linear(att_feats, P["obj_v_proj.weight"], P["obj_v_proj.bias"])        # :377
        x = torch.relu(x + emb.view(obj_dist.size(0), obj_dist.size(1), -1))        # :378
    else:
        x = F.linear(att_feats, P["obj_v_proj.weight"], P["obj_v_proj.bias"])        # :380
    flat = pred_dist.reshape(-1, n_pred)
    pc = flat[:, 1:].max(1)[1] + 1 if cfg.pred_emb_type == 1 else flat.max(1)[1]    # :383,:385
    p = F.linear(F.embedding(pc, P["sg_pred_embed.weight"]), P["pred_emb_prj.weight"], P["pred_emb_prj.bias"])
    return x, p.view(pred_dist.size(0), pred_dist.size(1), -1)


def make_map(b, N, K, rel_ind, like):
    """gcn_backbone.py:55-67: map[b, n, k, role] = 1 iff rel_ind[b, k, role] == n."""
    maps = []
    for role in (0, 1):
        m = like.new_zeros(b, N, K)
        for i in range(b):
            m[i].scatter_(0, rel_ind[i, :, role].contiguous().view(1, K), like.new_ones(1, K))
        maps.append(m)
    return maps


def collect_unit(P, pre, cfg, source, adj, bn_state, training):
    """graph_conv_unit.py:28-36 (one _Collection_Unit)."""
    out = F.linear(F.linear(source, P[pre + "fc_lft.weight"], P[pre + "fc_lft.bias"]),
                   P[pre + "fc_rgt.weight"], P[pre + "fc_rgt.bias"])
    if cfg.bn:
        dim = out.size(-1)
        out = F.batch_norm(out.view(-1, dim), bn_state[pre + "bn.running_mean"], bn_state[pre + "bn.running_var"],
                           P[pre + "bn.weight"], P[pre + "bn.bias"], training, 0.1, 1e-5).view(source.shape[0], source.shape[1], dim)
    collect = torch.bmm(adj, out)
    return torch.relu(collect / (adj.sum(2).view(collect.size(0), collect.size(1), 1) + 1e-7))


def gcn_backbone(P, cfg, x_obj, x_pred, rel_ind, bn_state=None, training=False, tap=None):
    """gcn_backbone.py:29-53 + graph_conv.py:15-34; returns the x5-expanded features."""
    b, N, L = x_obj.shape; K = x_pred.size(1)
    skip_x, skip_p = x_obj, x_pred
    if cfg.layers:
        m_s, m_o = make_map(b, N, K, rel_ind, x_obj.detach())
        for l in range(cfg.layers):
            pre = f"gcn_backbone.gcn.{l}.gcn_collect.collect_units."
            u = lambda i, src, adj: collect_unit(P, f"{pre}{i}.", cfg, src, adj, bn_state, training)
            new_x = (u(0, x_pred, m_s) + u(1, x_pred, m_o)) / 2                      # graph_conv.py:24-26
            new_p = (u(2, x_obj, m_s.transpose(1, 2)) + u(3, x_obj, m_o.transpose(1, 2))) / 2   # :29-33
            x_obj, x_pred = new_x, new_p
            if tap is not None:
                tap[f"gcn_x_layer{l}"] = x_obj; tap[f"gcn_p_layer{l}"] = x_pred
            if (l + 1) % cfg.residual == 0:                                          # gcn_backbone.py:43-47
                x_obj = x_obj + skip_x; skip_x = x_obj
                x_pred = x_pred + skip_p; skip_p = x_pred
    x5 = x_obj.view(b, 1, N, L).expand(b, 5, N, L).contiguous().view(-1, N, L)      # :50-51
    p5 = x_pred.view(b, 1, K, L).expand(b, 5, K, L).contiguous().view(-1, K, L)
    return x5, p5


# ----------------------------------------------------------------------------- sGPN
def extract_subgraph_feats(b, N, L, att_feats, gpn_obj_ind):
    """gpn.py:152-172 (node half; the predicate half is gathered by the reference but unused)."""
    hb = gpn_obj_ind.size(-2)
    bi = torch.arange(b).view(b, 1).expand(b, N * hb).contiguous().view(-1)
    pos = att_feats[bi, gpn_obj_ind[:, 0].contiguous().view(-1)]
    neg = att_feats[bi, gpn_obj_ind[:, 1].contiguous().view(-1)]
    return torch.cat((pos.view(-1, N, L), neg.view(-1, N, L)), 0)


def graph_pooling(N, gpn_att, gpn_pool_mtx, att_masks):
    """gpn.py:174-185."""
    each = gpn_pool_mtx.transpose(0, 1).contiguous().view(-1, N, N)
    clean = torch.bmm(each, gpn_att)
    mx = clean.max(1)[0]
    mean = clean.sum(1) / att_masks.transpose(0, 1).sum(-1).view(-1, 1)
    return torch.cat((mx, mean), -1)


def cal_node_iou(a, b):
    """gpn.py:140-150."""
    if a.shape[0] == 0 or b.shape[0] == 0:
        a = np.arange(a.shape[0])
    sa, sb = set(a.tolist()), set(b.tolist())
    return len(sa & sb) / float(len(sa | sb))


def subgraph_nms(score, obj_ind, masks, thres, max_subgraphs, sort_kind=None):
    """gpn.py:108-138.  `sort_kind=None` is numpy's default argsort exactly as the reference
    calls it (unstable on ties); 'stable' gives the documented tie rule (larger index first)."""
    score = np.asarray(score); obj_ind = np.asarray(obj_ind); masks = np.asarray(masks)
    order = (np.argsort(score) if sort_kind is None else np.argsort(score, kind=sort_kind))[::-1]
    sets = [np.unique(obj_ind[i][masks[i].nonzero()[0]]) for i in order]
    keep = np.ones(len(order))
    for i in range(len(order)):
        if keep[i] == 0:
            continue
        for j in range(i + 1, len(order)):
            if cal_node_iou(sets[i], sets[j]) > thres:
                keep[j] = 0
    kept_sorted = order[keep == 1]
    flag = np.zeros(len(order))
    flag[kept_sorted[:max_subgraphs]] = 1
    return flag.nonzero()[0]


def gpn_layer(P, cfg, b, N, L, gpn_obj_ind, gpn_pool_mtx, att_feats, att_masks, training, masks=None, tap=None,
              nms_sort_kind=None):
    """gpn.py:41-106 (both branches)."""
    hb = gpn_obj_ind.size(-2)
    gpn_att = extract_subgraph_feats(b, N, L, att_feats, gpn_obj_ind)
    gb = gpn_att.size(0)
    read_out = graph_pooling(N, gpn_att, gpn_pool_mtx, att_masks)
    if tap is not None:
        tap["read_out"] = read_out
    if cfg.use_score:
        hid = torch.relu(F.linear(read_out, P["gpn_layer.gpn_fc.0.weight"], P["gpn_layer.gpn_fc.0.bias"]))
        hid = _drop(hid, cfg.p_gpn, training, None if masks is None else masks.get("gpn_hid"))
        score = torch.sigmoid(F.linear(hid, P["gpn_layer.gpn_fc.3.weight"], P["gpn_layer.gpn_fc.3.bias"]))
        target = torch.cat((score.new_ones(gb // 2, 1), score.new_zeros(gb // 2, 1)), 0)
        gpn_loss = F.binary_cross_entropy(score, target)
    else:
        score = read_out.new_ones(gb, 1); gpn_loss = None
    rop = lambda r: F.linear(F.linear(r, P["gpn_layer.read_out_proj.0.weight"], P["gpn_layer.read_out_proj.0.bias"]),
                             P["gpn_layer.read_out_proj.1.weight"], P["gpn_layer.read_out_proj.1.bias"])
    if not cfg.test_lstm:                                                             # gpn.py:64-81
        sel = score.squeeze().view(2, b, hb)[0].argmax(-1)
        bi = torch.arange(b)
        sub_idx = gpn_obj_ind[:, 0][bi, sel, :].view(-1)
        att = att_feats[torch.arange(b).view(b, 1).expand(b, N).contiguous().view(-1), sub_idx, :].view(b, N, L)
        m = att_masks[:, 0][bi, sel, :]
        fc = rop(read_out.view(2, b, hb, -1)[0][bi, sel, :].detach())
        if tap is not None:
            tap.update(sel_slot=sel, att_sel=att, mask_sel=m, fc_sel=fc)
        return gpn_loss, score, att, fc, m
    assert b == 5                                                                     # gpn.py:84
    s = score.squeeze().view(2, b, hb).transpose(0, 1)[0].contiguous().view(-1)
    all_idx = gpn_obj_ind[0].contiguous().view(-1, N)
    att = att_feats[0][all_idx.view(-1), :].view(s.size(0), N, L)
    m = att_masks[0].contiguous().view(-1, N)
    fc = rop(read_out.view(2, b, hb, -1).transpose(0, 1)[0].contiguous().view(-1, read_out.size(-1)))
    keep = torch.arange(s.size(0))
    if not cfg.sct:                                                                   # use_nms (AttModel.py:95)
        keep = torch.from_numpy(subgraph_nms(s.detach().numpy(), all_idx.numpy(), m.numpy(), cfg.nms_thres,
                                             cfg.max_subg, nms_sort_kind)).long()
        s, att, fc, m = s[keep], att[keep], fc[keep], m[keep]
    return gpn_loss, s, att, fc, m, keep


# ----------------------------------------------------------------------------- decoder
def prepare_feature(P, cfg, fc, att, mask, training, masks=None):
    """AttModel.py:348-368 (+ pack_wrapper :28-36: att_embed on valid rows only, pads = 0)."""
    n_max = int(mask.long().sum(1).max())
    att = att[:, :n_max].contiguous(); mask = mask[:, :n_max].contiguous()
    g = lambda k: None if masks is None else masks.get(k)
    f = torch.relu(F.linear(fc, P["fc_embed.0.weight"], P["fc_embed.0.bias"]))
    f = _drop(torch.relu(F.linear(f, P["fc_embed.2.weight"], P["fc_embed.2.bias"])), cfg.p_lm, training, g("fc"))
    v = _drop(torch.relu(F.linear(att, P["att_embed.0.weight"], P["att_embed.0.bias"])), cfg.p_lm, training,
              None if g("att") is None else g("att")[:, :n_max])
    valid = (torch.arange(n_max).view(1, -1) < mask.long().sum(1, keepdim=True)).to(v.dtype).unsqueeze(-1)
    v = v * valid                                                                     # packed rows only; pads are 0
    u = F.linear(v, P["ctx2att.weight"], P["ctx2att.bias"])
    return f, v, u, mask


def lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh):
    """nn.LSTMCell: gate order i, f, g, o."""
    i, f, g, o = (F.linear(x, w_ih, b_ih) + F.linear(h, w_hh, b_hh)).chunk(4, 1)
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    return torch.sigmoid(o) * torch.tanh(c2), c2


def attention(P, h, v, u, mask):
    """AttModel.py:445-471."""
    att_h = F.linear(h, P["core.attention.h2att.weight"], P["core.attention.h2att.bias"])
    dot = torch.tanh(u + att_h.unsqueeze(1))
    e = F.linear(dot, P["core.attention.alpha_net.weight"], P["core.attention.alpha_net.bias"]).squeeze(-1)
    w = F.softmax(e, dim=1)
    w = w * mask.float()
    w = w / w.sum(1, keepdim=True)
    return torch.bmm(w.unsqueeze(1), v).squeeze(1), w


def core_step(P, cfg, it, f, v, u, mask, state, training, xt_mask=None, out_mask=None, tap=None):
    """AttModel.py:328-341 + TopDownCore :400-431."""
    (h1, h2), (c1, c2) = state
    xt = _drop(torch.relu(F.embedding(it, P["embed.0.weight"])), cfg.p_lm, training, xt_mask)
    h1, c1 = lstm_cell(torch.cat([h2, f, xt], 1), h1, c1, P["core.att_lstm.weight_ih"], P["core.att_lstm.weight_hh"],
                       P["core.att_lstm.bias_ih"], P["core.att_lstm.bias_hh"])
    ctx, w = attention(P, h1, v, u, mask)
    h2, c2 = lstm_cell(torch.cat([ctx, h1], 1), h2, c2, P["core.lang_lstm.weight_ih"], P["core.lang_lstm.weight_hh"],
                       P["core.lang_lstm.bias_ih"], P["core.lang_lstm.bias_hh"])
    out = _drop(h2, cfg.p_lm, training, out_mask)
    logp = F.log_softmax(F.linear(out, P["logit.weight"], P["logit.bias"]), dim=1)
    if tap is not None:
        for k, t in (("h_att", h1), ("c_att", c1), ("h_lang", h2), ("c_lang", c2), ("alpha", w), ("ctx", ctx), ("logp", logp)):
            tap.setdefault("step_" + k, []).append(t)
    return logp, ((h1, h2), (c1, c2)), w


def _zeros_state(n, R, like):
    z = lambda: like.new_zeros(n, R)
    return ((z(), z()), (z(), z()))


def _encode(P, cfg, args, bn_state, training, masks, tap, nms_sort_kind=None):
    """Everything before the step loop (AttModel.py:128-155 / :249-276)."""
    fc_feats, att_feats, att_masks, obj_dist, rel_ind, pred_dist, gpn_obj_ind, gpn_pool_mtx = args
    x, p = feat_fusion(P, cfg, obj_dist, att_feats, pred_dist)
    if tap is not None:
        tap["fusion_x"] = x; tap["fusion_p"] = p
    b, N, L = x.shape
    x5, p5 = gcn_backbone(P, cfg, x, p, rel_ind, bn_state, training, tap)
    if tap is not None:
        tap["x_obj_out"] = x5[::5]
    b5 = x5.size(0)
    keep = None
    if cfg.gpn:
        r = gpn_layer(P, cfg, b5, N, L, gpn_obj_ind, gpn_pool_mtx, x5, att_masks, training, masks, tap, nms_sort_kind)
        gpn_loss, score, att, fc, m = r[:5]
        keep = r[5] if len(r) > 5 else None
    else:                                                                             # AttModel.py:140-149 / :261-271
        gpn_loss = None; score = None
        rop = lambda r_: F.linear(F.linear(r_, P["read_out_proj.0.weight"], P["read_out_proj.0.bias"]),
                                  P["read_out_proj.1.weight"], P["read_out_proj.1.bias"])
        if cfg.sample_mode:
            att = x5[0:1]
            fc = rop(att.mean(1))
            m = att_masks[0:1, 0, 0]; m[:, :36].fill_(1.0)
            keep = torch.arange(1)
            score = torch.ones(1)
        else:
            att = x5
            fc = rop(att.mean(1).detach())
            m = att_masks[:, 0, 0]; m[:, :36].fill_(1.0)                              # in place on the caller's tensor
    return gpn_loss, score, att, fc, m, keep


class Oracle:
    """Functional model over a {reference state_dict key: tensor} parameter dict."""

    def __init__(self, opt, params, requires_grad=False):
        self.cfg = Cfg(opt)
        self.P = {}
        self.buffers = {}
        for k, v in params.items():
            t = torch.as_tensor(np.asarray(v)) if not torch.is_tensor(v) else v
            t = t.clone()
            if "running_" in k or "num_batches" in k:
                self.buffers[k] = t
            else:
                self.P[k] = t.float().requires_grad_(requires_grad)
        self.training = False

    # -- train ---------------------------------------------------------------
    def forward(self, fc_feats, att_feats, seq, att_masks=None, trip_pred=None, obj_dist=None, obj_box=None, rel_ind=None,
                pred_fmap=None, pred_dist=None, gpn_obj_ind=None, gpn_pred_ind=None, gpn_nrel_ind=None, gpn_pool_mtx=None,
                masks=None, tap=None, ss=None):
        """AttModel._forward (AttModel.py:122-177); `masks` optionally injects dropout keep-masks.
        `ss=(sel_u [T,n], u [T,n])` pins scheduled sampling (:157-167): row r of step i >= 1 is re-drawn iff
        sel_u[i, r] < ss_prob (the reference's `sample_prob < self.ss_prob`), and the draw from exp(outputs[:, i-1]) is the
        inverse CDF at u[i, r] in index order (the reference's torch.multinomial stream cannot be reproduced elsewhere)."""
        P, cfg = self.P, self.cfg
        cfg.sample_mode = False
        gpn_loss, score, att, fc, m, _ = _encode(P, cfg, (fc_feats, att_feats, att_masks, obj_dist, rel_ind, pred_dist,
                                                          gpn_obj_ind, gpn_pool_mtx), self.buffers, self.training, masks, tap)
        n = fc.size(0)
        state = _zeros_state(n, cfg.R, fc)
        outputs = fc.new_zeros(n, seq.size(1) - 1, cfg.vocab + 1)
        f, v, u, mk = prepare_feature(P, cfg, fc, att, m, self.training, masks)
        if tap is not None:
            tap.update(p_fc=f, p_att=v, pp_att=u, p_mask=mk)
        g = lambda k, i: None if masks is None or masks.get(k) is None else masks[k][:, i]
        for i in range(seq.size(1) - 1):
            it = seq[:, i].clone()                                                    # ss_prob == 0 path (:168-169)
            if self.training and i >= 1 and cfg.ss_prob > 0.0:                        # :157-167
                if ss is None:
                    raise ValueError("the oracle needs injected uniforms for scheduled sampling")
                sample_mask = ss[0][i] < cfg.ss_prob
                if sample_mask.sum() != 0:
                    prob_prev = torch.exp(outputs[:, i - 1].detach())
                    cdf = torch.cumsum(prob_prev, 1)
                    draw = (cdf <= (ss[1][i] * cdf[:, -1]).unsqueeze(1)).sum(1).clamp(max=cdf.size(1) - 1)
                    self.ss_tokens = getattr(self, "ss_tokens", {})
                    it = torch.where(sample_mask, draw, it)
                    self.ss_tokens[i] = it.clone()
            if i >= 1 and seq[:, i].sum() == 0:                                       # :171-172
                break
            logp, state, _ = core_step(P, cfg, it, f, v, u, mk, state, self.training, g("xt", i), g("out", i), tap)
            outputs[:, i] = logp
        return outputs, gpn_loss, score

    # -- decode --------------------------------------------------------------
    def sample(self, fc_feats, att_feats, att_masks=None, trip_pred=None, obj_dist=None, obj_box=None, rel_ind=None,
               pred_fmap=None, pred_dist=None, gpn_obj_ind=None, gpn_pred_ind=None, gpn_nrel_ind=None, gpn_pool_mtx=None,
               opt=None, uniforms=None, forced=None, tap=None, nms_sort_kind=None):
        """AttModel._sample (AttModel.py:236-326), beam_size == 1.  Top-k sampling draws from
        `uniforms[n, T]` by inverse CDF over the k renormalised probabilities in top-k order
        (the reference draws with torch.multinomial, which cannot be reproduced elsewhere);
        `forced[n, T]` makes the loop follow a given token path (used to pin the top-k maths)."""
        P, cfg = self.P, self.cfg
        opt = opt or {}
        return_att = opt.get("return_att", 0) == 1
        cfg.sample_mode = True
        with torch.no_grad():
            _, score, att, fc, m, keep = _encode(P, cfg, (fc_feats, att_feats, att_masks, obj_dist, rel_ind, pred_dist,
                                                           gpn_obj_ind, gpn_pool_mtx), self.buffers, False, None, tap, nms_sort_kind)
            n = fc.size(0)
            state = _zeros_state(n, cfg.R, fc)
            f, v, u, mk = prepare_feature(P, cfg, fc, att, m, False)
            T = cfg.seq_length
            seq = torch.zeros(n, T, dtype=torch.long); lps = fc.new_zeros(n, T)
            it = torch.zeros(n, dtype=torch.long)
            atts, unfinished = [], None
            for t in range(T + 1):
                logp, state, w = core_step(P, cfg, it, f, v, u, mk, state, False, tap=tap)
                atts.append(w)
                if t == T:
                    break
                if cfg.topk:                                                          # AttModel.py:295-303
                    lp = F.log_softmax(logp / float(cfg.topk_temp), dim=1)
                    top, idx = torch.topk(lp, cfg.the_k, dim=1)
                    if tap is not None:
                        tap.setdefault("topk_idx", []).append(idx); tap.setdefault("topk_lp", []).append(top)
                    if forced is not None:
                        it = forced[:, t].clone()
                        slp = lp.gather(1, it.unsqueeze(1)).view(-1)
                    else:
                        pr = torch.exp(top - torch.logsumexp(top, 1, keepdim=True))   # Categorical(logits=) renormalises
                        cdf = pr.cumsum(1)
                        uu = uniforms[:, t:t + 1] if uniforms is not None else torch.rand(n, 1)
                        pick = (uu >= cdf).sum(1).clamp(max=cfg.the_k - 1)
                        it = idx.gather(1, pick.unsqueeze(1)).view(-1)
                        slp = top.gather(1, pick.unsqueeze(1)).view(-1)
                else:
                    slp, it = torch.max(logp, 1)                                      # :306 (first max)
                    if forced is not None:
                        it = forced[:, t].clone(); slp = logp.gather(1, it.unsqueeze(1)).view(-1)
                unfinished = (it > 0) if t == 0 else unfinished * (it > 0)
                it = it * unfinished.type_as(it)
                seq[:, t] = it
                lps[:, t] = slp                                                       # un-masked after EOS (:316)
                if unfinished.sum() == 0:
                    break
        out = (seq, lps, score, keep)
        return out + (torch.stack(atts, 1),) if return_att else out


    def sample_beam(self, fc_feats, att_feats, att_masks=None, trip_pred=None, obj_dist=None, obj_box=None, rel_ind=None,
                    pred_fmap=None, pred_dist=None, gpn_obj_ind=None, gpn_pred_ind=None, gpn_nrel_ind=None, gpn_pool_mtx=None,
                    opt=None, nms_sort_kind=None):
        """AttModel._sample_sentences (AttModel.py:179-234): one beam search per kept sub-graph.
        Returns (seq, seqLogprobs, score, keep, done_beams)."""
        P, cfg = self.P, self.cfg
        opt = opt or {}
        beam = opt.get("beam_size", 10)
        cfg.sample_mode = True
        with torch.no_grad():
            _, score, att, fc, m, keep = _encode(P, cfg, (fc_feats, att_feats, att_masks, obj_dist, rel_ind, pred_dist,
                                                           gpn_obj_ind, gpn_pool_mtx), self.buffers, False, None, None, nms_sort_kind)
            f, v, u, mk = prepare_feature(P, cfg, fc, att, m, False)
            n, T = fc.size(0), cfg.seq_length
            seq = torch.zeros(n, T, dtype=torch.long); lps = fc.new_zeros(n, T)
            done = []
            for k in range(n):
                fk, vk, uk = f[k:k + 1].expand(beam, -1), v[k:k + 1].expand(beam, -1, -1), u[k:k + 1].expand(beam, -1, -1)
                mkk = mk[k:k + 1].expand(beam, -1) if mk is not None else None

                def step(it, state, fk=fk, vk=vk, uk=uk, mkk=mkk):
                    r0, r1 = state["rows"]
                    h, c = state["s"]
                    logp, (hh, cc), _ = core_step(P, cfg, it, fk[r0:r1], vk[r0:r1], uk[r0:r1], None if mkk is None else mkk[r0:r1],
                                                  ((h[0], h[1]), (c[0], c[1])), False)
                    return logp, {"s": (torch.stack(hh), torch.stack(cc)), "rows": (r0, r1)}
                z = fc.new_zeros(2, beam, cfg.R)
                logp0, st0 = step(torch.zeros(beam, dtype=torch.long), {"s": (z, z.clone()), "rows": (0, beam)})
                beams = beam_search(step, st0, logp0, T, opt)
                done.append(beams)
                seq[k], lps[k] = beams[0]["seq"], beams[0]["logps"]
        return seq, lps, score, keep, done


def length_penalty_fn(cfg):
    """misc/utils.py:145-171 (`''` -> identity, `wu_a` -> GNMT length normalisation, `avg_a` -> mean)."""
    if cfg == "":
        return lambda length, lp: lp
    kind, alpha = cfg.split("_")
    alpha = float(alpha)
    return {"wu": lambda length, lp: lp / (((5 + length) ** alpha) / ((5 + 1) ** alpha)),
            "avg": lambda length, lp: lp / length}[kind]


def beam_search(step, init_state, init_logprobs, T, opt):
    """CaptionModel.beam_search (CaptionModel.py:28-176), restated for one sub-graph.

    `step(it, state) -> (logprobs, state)` is get_logprobs_state for the rows of one group; a state is
    {"s": (h[2, rows, R], c[2, rows, R]), "rows": (first, last)}.  Every group g runs its own classical
    beam search `g` steps behind group 0; before a group sorts, the words the earlier groups hold at the
    same position are pushed down by diversity_lambda once per holder (:33-40); the UNK column is pushed
    down by 1000 (:137) and, with decoding_constraint, the previous word is removed (:134-135)."""
    beam = opt.get("beam_size", 10)
    G = opt.get("group_size", 1)
    lam = opt.get("diversity_lambda", 0.5)
    constraint = opt.get("decoding_constraint", 0)
    penalty = length_penalty_fn(opt.get("length_penalty", ""))
    bd = beam // G
    seqs = [torch.zeros(T, bd, dtype=torch.long) for _ in range(G)]
    lpss = [torch.zeros(T, bd) for _ in range(G)]
    sums = [torch.zeros(bd) for _ in range(G)]
    finished = [[] for _ in range(G)]
    h0, c0 = init_state["s"]
    states = [{"s": (h0[:, g * bd:(g + 1) * bd].clone(), c0[:, g * bd:(g + 1) * bd].clone()), "rows": (g * bd, (g + 1) * bd)}
              for g in range(G)]
    logps = [init_logprobs[g * bd:(g + 1) * bd].clone().float() for g in range(G)]
    for t in range(T + G - 1):
        for g in range(G):
            tau = t - g
            if tau < 0 or tau > T - 1:
                continue
            lp = logps[g]
            if constraint and tau > 0:
                lp.scatter_(1, seqs[g][tau - 1].unsqueeze(1), float("-inf"))
            lp[:, -1] = lp[:, -1] - 1000
            raw = lp.clone()
            for pg in range(g):
                held = seqs[pg][tau]
                for b in range(bd):
                    for j in range(bd):
                        lp[b][held[j]] = lp[b][held[j]] - lam
            # ---- one classical step (:44-94)
            ys, ix = torch.sort(lp, 1, True)
            cand = []
            for c in range(min(bd, ys.size(1))):
                for q in range(1 if tau == 0 else bd):
                    cand.append(dict(c=ix[q, c], q=q, p=sums[g][q] + ys[q, c].item(), r=raw[q, ix[q, c]]))
            cand = sorted(cand, key=lambda x: -x["p"])
            h, c_ = states[g]["s"]
            nh, nc = h.clone(), c_.clone()
            old_seq, old_lps = seqs[g][:tau].clone(), lpss[g][:tau].clone()
            for vix in range(bd):
                v = cand[vix]
                if tau >= 1:
                    seqs[g][:tau, vix] = old_seq[:, v["q"]]
                    lpss[g][:tau, vix] = old_lps[:, v["q"]]
                nh[:, vix] = h[:, v["q"]]
                nc[:, vix] = c_[:, v["q"]]
                seqs[g][tau, vix] = v["c"]
                lpss[g][tau, vix] = v["r"]
                sums[g][vix] = v["p"]
            for vix in range(bd):
                if seqs[g][tau, vix] == 0 or tau == T - 1:
                    fin = dict(seq=seqs[g][:, vix].clone(), logps=lpss[g][:, vix].clone(), unaug_p=lpss[g][:, vix].sum().item(),
                               p=penalty(tau + 1, sums[g][vix].item()))
                    finished[g].append(fin)
                    sums[g][vix] = -1000
            logps[g], states[g] = step(seqs[g][tau], {"s": (nh, nc), "rows": states[g]["rows"]})
            logps[g] = logps[g].float()
    out = []
    for g in range(G):
        out += sorted(finished[g], key=lambda x: -x["p"])[:bd]
    return out


def lm_criterion(outputs, target, mask):
    """misc/utils.py:115-124."""
    target = target[:, :outputs.size(1)]; mask = mask[:, :outputs.size(1)]
    out = -outputs.gather(2, target.unsqueeze(2)).squeeze(2) * mask
    return out.sum() / mask.sum()


def loss_wrapper(oracle, batch, masks=None, tap=None, ss=None):
    """models/loss_wrapper.py:14-27 -> {'lang_loss', 'gpn_loss'}."""
    from_keys = ("fc_feats", "att_feats", "labels", "att_masks", None, "obj_dist", None, "rel_ind", None, "pred_dist",
                 "gpn_obj_ind", "gpn_pred_ind", "gpn_nrel_ind", "gpn_pool_mtx")
    args = [None if k is None else batch[k] for k in from_keys]
    outputs, gpn_loss, score = oracle.forward(*args, masks=masks, tap=tap, ss=ss)
    lang = lm_criterion(outputs, batch["labels"][:, 1:], batch["masks"][:, 1:])
    return {"lang_loss": lang, "gpn_loss": gpn_loss, "outputs": outputs, "subgraph_score": score}


# --------------------------------------------------------------------------- eval glue (misc/eval_utils.py:105-141)
_BAD_ENDINGS = ['with', 'in', 'on', 'of', 'a', 'at', 'to', 'for', 'an', 'this', 'his', 'her', 'that', 'the']   # misc/utils.py:16-17


def decode_sequence(ix_to_word, seq, remove_bad_endings=0):
    """misc/utils.py:59-81."""
    out = []
    for i in range(seq.size(0)):
        txt = ''
        for j in range(seq.size(1)):
            ix = int(seq[i, j])
            if ix <= 0:
                break
            txt = txt + (' ' if j >= 1 else '') + ix_to_word[str(ix)]
        if remove_bad_endings:
            flag, words = 0, txt.split(' ')
            for j in range(len(words)):
                if words[-j - 1] not in _BAD_ENDINGS:
                    flag = -j
                    break
            txt = ' '.join(words[0:len(words) + flag])
        out.append(txt)
    return out


def rank_subgraphs(gpn, seqq, subgraph_score, keep_nms_ind, sct_mode=False):
    """misc/eval_utils.py:105-121."""
    if sct_mode:
        valid = int(subgraph_score.size(0) / 2)
        return seqq[:valid], subgraph_score[:valid], keep_nms_ind[:valid], keep_nms_ind[:valid].long()
    if gpn:
        sorted_score, sort_ind = torch.sort(subgraph_score, descending=True, stable=True)
        return seqq[sort_ind], sorted_score, keep_nms_ind[sort_ind], sort_ind
    return seqq, subgraph_score, keep_nms_ind, torch.arange(subgraph_score.size(0)).type_as(keep_nms_ind)


def grounding_argmax(att_weights, sort_ind, subg_index, n_words, obj_ind_this):
    """misc/grd_utils.py:36-47, the integer part: for the caption ranked `subg_index` the arg-max attention column of each of its
    first `n_words` word positions (`torch.max(att_weights[row], dim=1)[1][:len(grd_wd)]`, :42 / :46) and the full-graph node
    each stands for (`obj_ind_this[att2_ind[wd_j]]`, :56).  att_weights [n, steps, n_max]; sort_ind: eval_utils.py:106-113 (None:
    the Full-GC branch indexes att_weights[subg_index] directly, :46); obj_ind_this: the chosen sub-graph's node ids in the full
    graph, ascending (:41 `graph_mask[1].nonzero()[0]`; Full-GC: arange(36), :45).  -> (att2_ind, node_ind) int64 arrays.
    PINNED by tests/golden/grd_out.npz: what the reference's own get_grounding_material returned (make_golden.py grd_cases)."""
    row = int(subg_index) if sort_ind is None else int(sort_ind[subg_index])
    att2 = torch.max(att_weights[row], dim=1)[1][:n_words]
    obj = np.asarray(obj_ind_this)
    return att2.numpy().astype(np.int64), obj[att2.numpy()].astype(np.int64)


def grounding_material(sents, subg_index, att2_node, boxes, wd_to_lemma, lemma_det_id_dict, det_id_to_det_wd):
    """misc/grd_utils.py:38,49-58: words of the chosen sentence -> lemma -> detection class; the box of each grounded word's node."""
    out = {'clss': [], 'idx_in_sent': [], 'bbox': []}
    grd_wd = sents[subg_index].split()
    for wd_j in range(len(grd_wd)):
        if grd_wd[wd_j] not in wd_to_lemma:
            continue
        lemma = wd_to_lemma[grd_wd[wd_j]]
        if lemma in lemma_det_id_dict:
            out['bbox'].append(boxes[att2_node[wd_j]].tolist())
            out['clss'].append(det_id_to_det_wd[lemma_det_id_dict[lemma]])
            out['idx_in_sent'].append(wd_j)
    return out


# --------------------------------------------------------------------------- batch assembly (dataloaders/dataloader.py:139-157,225-367)
# PINNED: tests/golden/loader_*.npz hold what the reference's own `DataLoader.__getitem__` returned for fabricated dataset
# entries (make_golden.py loader_cases: h5py stubbed so the module imports, object made with object.__new__), for the
# sampled-sub-graph branch with seeded np.random and for `use_gt_subg`; tests/test_oracle_golden.py replays them here.
def choose_subgraphs(node_iou_mtx, thres, hb, seq_per_img=5, rng=np.random):
    """dataloader.py:229-270: the (positive, negative) sub-graph ids of every sentence's mini-batch -> mask_idx [S, hb, 2],
    indices into `subgraph_mask_list` (the +5 shift of :270 applied).  Consumes `rng` exactly as the reference consumes np.random."""
    sampled_node_iou = node_iou_mtx[:, 5:]
    mask_idx = np.full((seq_per_img, hb, 2), -1)
    pos_mask = sampled_node_iou >= thres                                                          # :234
    neg_mask = sampled_node_iou < thres
    neg_mask[:, pos_mask.nonzero()[1]] = 0                                                        # :237
    weight = pos_mask / (pos_mask.sum(0) + 1e-7)
    n_weight = (weight.T / (weight.sum(1) + 1e-7)).T
    for i in range(seq_per_img):
        pos_idx = pos_mask[i].nonzero()[0]
        if pos_idx.shape[0] < hb:                                                                 # :243-245
            to_pad = hb - pos_idx.shape[0]
            mask_idx[i, :to_pad, 0] = i - 5
            mask_idx[i, to_pad:, 0] = pos_idx
        else:                                                                                     # :246-250
            pos_weight = n_weight[i][pos_idx]
            rd_ind = rng.randint(pos_weight.shape[0], size=1)
            pos_weight[rd_ind[0]] = 1.0 - (pos_weight.sum() - pos_weight[rd_ind[0]])
            mask_idx[i, :, 0] = rng.choice(pos_idx, size=hb, replace=True, p=pos_weight)
        neg_idx = neg_mask[i].nonzero()[0]                                                        # :252-266
        if neg_idx.shape[0] < hb:
            tmp_neg_idx = (sampled_node_iou[i] <= thres).nonzero()[0]
            if tmp_neg_idx.shape[0] == 0:
                neg_idx = (sampled_node_iou[i] <= 1.0).nonzero()[0]
                mask_idx[i, :, 1] = rng.choice(neg_idx, size=hb, replace=True)
            elif neg_idx.shape[0] == 0:
                mask_idx[i, :, 1] = rng.choice(tmp_neg_idx, size=hb, replace=True)
            else:
                mask_idx[i, :, 1] = rng.choice(neg_idx, size=hb, replace=True)
        else:
            mask_idx[i, :, 1] = rng.choice(neg_idx, size=hb, replace=False)
    return mask_idx + 5                                                                           # :270


def pick_captions(label, label_start_ix, label_end_ix, ix, seq_per_img, seq_length, pyrandom=None):
    """dataloader.py:139-157 get_captions: the first `seq_per_img` captions, or draws with replacement (python `random`)."""
    import random as _random
    pyrandom = pyrandom or _random
    ix1 = label_start_ix[ix] - 1
    ix2 = label_end_ix[ix] - 1
    ncap = ix2 - ix1 + 1
    assert ncap > 0
    if ncap < seq_per_img:
        seq = np.zeros([seq_per_img, seq_length], dtype='int')
        for q in range(seq_per_img):
            ixl = pyrandom.randint(ix1, ix2)
            seq[q, :] = label[ixl, :seq_length]
        return seq
    return label[ix1: ix1 + seq_per_img, :seq_length]


def assemble_image(object_fmap, object_dist, rel_ind, pred_dist, node_masks, pred_masks, captions, obj_num, rel_num, nrel=None):
    """One image: node_masks [S, 2, hb, obj_num-1] bool, pred_masks [S, 2, hb, rel_num-1] bool, captions [S, seq_length];
    `nrel[i][side][k]` (optional): the re-indexed relation endpoints [n, 2] of the chosen sub-graphs (:303-308).
    The use_gt_subg branch (:310-327) is the same call with sentence i's own masks broadcast over (side, k)."""
    S, _, hb, _ = node_masks.shape
    gpn_obj_ind = np.full((S, 2, hb, obj_num), obj_num - 1)                                       # :276
    gpn_att_mask = np.full((S, 2, hb, obj_num), 0).astype('float32')                              # :277
    gpn_pred_ind = np.full((S, 2, hb, rel_num), rel_num - 1)                                      # :278
    gpn_nrel_ind = np.full((S, 2, hb, rel_num, 2), obj_num - 1)                                   # :279
    gpn_pool_mtx = np.zeros((S, 2, hb, obj_num, obj_num)).astype('float32')                       # :280
    for i in range(S):
        for k in range(hb):
            for side in range(2):                                                                 # :283-308 (pos then neg)
                tmp = node_masks[i, side, k].nonzero()[0]
                if tmp.shape[0] != 0:
                    gpn_obj_ind[i, side, k, :tmp.shape[0]] = tmp
                gpn_att_mask[i, side, k, :tmp.shape[0]] = 1
                gpn_pool_mtx[i, side, k, np.arange(tmp.shape[0]), np.arange(tmp.shape[0])] = 1
                tmp = pred_masks[i, side, k].nonzero()[0]
                if tmp.shape[0] != 0:
                    gpn_pred_ind[i, side, k, :tmp.shape[0]] = tmp
                if nrel is not None:
                    tmp = nrel[i][side][k]
                    if tmp.shape[0] != 0:
                        gpn_nrel_ind[i, side, k, :tmp.shape[0]] = tmp
    pad_fmap = np.full((1, obj_num, object_fmap.shape[1]), 0).astype('float32')                   # :336
    pad_dist = np.concatenate((np.ones((1, obj_num, 1)), np.zeros((1, obj_num, object_dist.shape[1] - 1))), axis=2).astype('float32')
    fc_feat = np.full((1, object_fmap.shape[1]), 0).astype('float32')
    pad_fmap[0, :obj_num - 1, :] = object_fmap                                                    # :340
    pad_dist[0, :obj_num - 1, :] = object_dist
    pad_rel = np.full((1, rel_num, rel_ind.shape[1]), obj_num - 1)                                # :349
    pad_pred = np.concatenate((np.ones((1, rel_num, 1)), np.zeros((1, rel_num, pred_dist.shape[1] - 1))), axis=2).astype('float32')
    n = min(rel_ind.shape[0], rel_num - 1)                                                        # :352
    pad_pred[0, :n, :] = pred_dist[:n]
    pad_rel[0, :n, :] = rel_ind[:n]
    Lq = captions.shape[1]
    label = np.zeros([S, Lq + 2], dtype='int64')                                                  # :356-357
    label[:, 1:Lq + 1] = captions
    nonzeros = np.array([(x != 0).sum() + 2 for x in label])
    mask = np.zeros([S, Lq + 2], dtype='float32')
    for idx in range(S):
        mask[idx, :nonzeros[idx]] = 1
    return dict(fc_feats=fc_feat, att_feats=pad_fmap, obj_dist=pad_dist, rel_ind=pad_rel, pred_dist=pad_pred, labels=label, masks=mask,
                gpn_obj_ind=gpn_obj_ind, att_masks=gpn_att_mask, gpn_pred_ind=gpn_pred_ind, gpn_nrel_ind=gpn_nrel_ind,
                gpn_pool_mtx=gpn_pool_mtx)
