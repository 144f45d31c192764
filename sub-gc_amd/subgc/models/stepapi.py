"""The reference's step-level methods on the HIP path: `_prepare_feature`, `get_logprobs_state`, `beam_search`.

Reference: AttModel._prepare_feature (models/AttModel.py:356-368), get_logprobs_state (:328-341),
TopDownCore.forward (:400-431), CaptionModel.beam_search (models/CaptionModel.py:28-176).  The product's decode
loops (`models/sampling.py`, `beam.py`) keep their state on the device in the next GEMM's operand layout and never
go through these; they exist so that code written against the reference's Python API -- a dense, clipped
`[b, n_max, .]` attention set, a caller-owned `(h[2,b,R], c[2,b,R])` state -- keeps working, through the same kernels.
Inference only (the reference's training loop does not call them either).
"""
from __future__ import annotations

from types import SimpleNamespace

import torch

from .. import beam as beam_mod
from .. import functions as F_
from .. import ops


def _eval_only(m, what):
    if m.training:
        raise NotImplementedError(f"{what}: the step-level API of the HIP path is inference-only (call model.eval())")


@torch.no_grad()
def prepare_feature(m, fc_feats, att_feats, att_masks):
    """-> (p_fc [b, R], p_att [b, n_max, R] with padded rows = 0, pp_att [b, n_max, A], att_masks [b, n_max])."""
    _eval_only(m, "_prepare_feature")
    P = m._decoder_params()
    fc0_w, fc0_b, fc2_w, fc2_b, att_w, att_b, c2a_w, c2a_b = P[:8]
    b = fc_feats.size(0)
    dev = fc_feats.device
    if att_masks is None:
        att_masks = att_feats.new_ones(att_feats.shape[:2])
    n_max = int(att_masks.long().sum(1).max().item())                            # clip_att (:348-354): one host read, like the reference
    att = att_feats[:, :n_max].contiguous()
    mask = att_masks[:, :n_max].contiguous()
    f1 = torch.empty(b, fc0_w.size(0), device=dev)
    ops.gemm(fc_feats.contiguous(), fc0_w, f1, tb=True, bias=fc0_b, relu=True)
    f = torch.empty(b, fc2_w.size(0), device=dev)
    ops.gemm(f1, fc2_w, f, tb=True, bias=fc2_b, relu=True)
    v = torch.empty(b * n_max, att_w.size(0), device=dev)
    ops.gemm(att.view(b * n_max, -1), att_w, v, tb=True, bias=att_b, relu=True)
    v.mul_(mask.reshape(-1, 1).to(v.dtype))                                      # pack_wrapper (:16-36): att_embed only on valid rows, pads = 0
    u = torch.empty(b * n_max, c2a_w.size(0), device=dev)
    ops.gemm(v, c2a_w, u, tb=True, bias=c2a_b)                                    # ctx2att on every clipped row
    return f, v.view(b, n_max, -1), u.view(b, n_max, -1), mask


@torch.no_grad()
def get_logprobs_state(m, it, fc_feats, att_feats, p_att_feats, att_masks, state, return_att=False):
    """One decoder step from a caller-owned state: -> (logprobs [b, V+1], (h[2,b,R], c[2,b,R])[, att_weights [b, n_max]])."""
    _eval_only(m, "get_logprobs_state")
    P = m._decoder_params()
    b, n_max, R = att_feats.shape
    dev = att_feats.device
    lens = (att_masks.sum(1) if att_masks is not None else torch.full((b,), n_max, device=dev)).to(torch.int32).contiguous()
    pr = SimpleNamespace(S=b, N=n_max, f=fc_feats.contiguous(), u=p_att_feats.contiguous().view(b * n_max, -1),
                         v=att_feats.contiguous().view(b * n_max, R),
                         off=(torch.arange(b, device=dev, dtype=torch.int32) * n_max).contiguous(), lens=lens)
    st = F_.DecodeState(pr, P, n_max, return_att, xt_table=m.xt_gates_table())
    h, c = state
    st.H1[:, :R] = h[1]; st.H1[:, R:] = h[0]                                      # [h_lang | h_att]
    st.H2[:, 2 * R:] = h[1]
    st.C1[0].copy_(c[0]); st.C2[0].copy_(c[1])
    alpha = torch.zeros(b, n_max, device=dev) if return_att else None
    logp = st.step(it.to(torch.long).contiguous(), alpha, normalize=True)
    new = (torch.stack([st.H1[:, R:], st.H1[:, :R]]), torch.stack([st.C1[0], st.C2[0]]))
    return (logp, new, alpha) if return_att else (logp, new)


@torch.no_grad()
def beam_search(m, init_state, init_logprobs, *args, **kwargs):
    """CaptionModel.beam_search with the reference's arguments: `args` = (fc, att, p_att, masks[, None, None]) expanded to
    `beam_size` rows, the state after <bos> and its log-probs.  Returns the list of finished beams (`done_beams`)."""
    opt = kwargs["opt"]
    eng = beam_mod._StepEngine(m, init_state, init_logprobs, tuple(args[:4]))
    _, _, done = beam_mod.search(eng, m.seq_length, opt)
    return done[0]
