"""`AttModel` / `TopDownModel`: the Sub-GC captioner behind the reference's model API, with every
contraction, gather, reduction and activation running in libsubgc_hip.so (gfx950).

Drop-in surface kept from reference `models/AttModel.py`:
  * ctor reads the same `opt` fields (AttModel.py:44-69,96-98);
  * `_forward(fc_feats, att_feats, seq, att_masks, trip_pred, obj_dist, obj_box, rel_ind, pred_fmap,
    pred_dist, gpn_obj_ind, gpn_pred_ind, gpn_nrel_ind, gpn_pool_mtx)` -> (outputs, gpn_loss, score)
    (AttModel.py:122-177) and `_sample(... , opt={})` -> (seq, seqLogprobs, score, keep_ind[, att])
    (AttModel.py:236-326);
  * attributes `.gpn .ss_prob .seq_length .vocab_size`, methods `init_hidden`;
  * `state_dict()` keys and shapes are the reference's, so checkpoints interchange.

What is different by design (MI355X-first):
  * all parameters are views into ONE flat fp32 buffer (`self.flat_params`, gradients likewise in
    `self.flat_grads`) so data-parallel training all-reduces one RCCL bucket (parallel.py) and the
    fused clip+Adam kernel sweeps one array;
  * the x5 "counterpart" expansion (gcn_backbone.py:50-51) and the gathered sub-graph tensor
    (gpn.py:159-170) are never materialised: kernels index `sentence -> image` instead;
  * dead GCN units (outputs that cannot reach any model output, SURVEY.md section 8a note) are skipped;
    their parameters stay in the state_dict and simply receive no gradient, exactly as in the
    reference;
  * the attention sets are ragged-packed once; padded rows are never computed.

Unsupported reference options fail loudly: `use_bn != 0` (BatchNorm inside att_embed, unused by
every preset; the reference's own forward raises for it: BatchNorm1d(att_feat_size) on gcn_dim-channel rows, AttModel.py:115).  Scheduled sampling (`ss_prob > 0`) runs through DecoderFn with per-step logits and its own
counter-based RNG stream; beam search lives in `subgc/beam.py`.
"""
from __future__ import annotations

import math
import os

import numpy as np
import torch
import torch.nn as nn

from .. import functions as F_
from .. import ops
from . import sampling
from . import stepapi
from .CaptionModel import CaptionModel


class _Node(nn.Module):
    """Anonymous container used to reproduce the reference's dotted state_dict names."""


def _attach(root, dotted, tensor, buffer=False):
    *path, leaf = dotted.split(".")
    mod = root
    for p in path:
        if p not in mod._modules:
            mod.add_module(p, _Node())
        mod = mod._modules[p]
    if buffer:
        mod.register_buffer(leaf, tensor)
    else:
        mod.register_parameter(leaf, tensor)


def _count_names(path, default):
    if path and os.path.exists(path):
        return int(np.load(path, encoding="latin1").shape[0])
    return default


class AttModel(CaptionModel):
    supports_fused_crit = True       # LossWrapper may pass fused_crit=(target, mask) to _forward
    packed_decoder = True            # loss-only calls skip the masked-out decoder steps (functions_packed.py)

    def __init__(self, opt):
        super().__init__()
        g = lambda n, d=None: getattr(opt, n, d)
        self.vocab_size = opt.vocab_size
        self.input_encoding_size = opt.input_encoding_size
        self.rnn_size = opt.rnn_size
        self.num_layers = g("num_layers", 1)
        self.drop_prob_lm = g("drop_prob_lm", 0.5)
        self.seq_length = g("max_length", None) or g("seq_length")
        self.fc_feat_size = opt.fc_feat_size
        self.att_feat_size = opt.att_feat_size
        self.att_hid_size = opt.att_hid_size
        self.use_bn = g("use_bn", 0)
        self.ss_prob = g("sampling_prob", 0.0)
        self.gpn = g("use_gpn", 1) == 1
        self.embed_dim = g("embed_dim", 300)
        self.GCN_dim = g("gcn_dim", 1024)
        self.noun_fuse = g("noun_fuse", 1) == 1
        self.pred_emb_type = g("pred_emb_type", 1)
        self.GCN_layers = g("gcn_layers", 2)
        self.GCN_residual = g("gcn_residual", 2)
        self.GCN_use_bn = g("gcn_bn", 0) != 0
        self.GCN_lr = 512                                     # graph_conv.py:11 (dim_lr is not plumbed)
        self.test_LSTM = g("test_LSTM", 0) != 0
        self.topk_sampling = g("use_topk_sampling", 0) != 0
        self.topk_temp = g("topk_temp", 0.6)
        self.the_k = g("the_k", 3)
        self.sct = g("sct", 0) != 0
        self.gpn_nms_thres = g("gpn_nms_thres", 0.75)
        self.gpn_max_subg = g("gpn_max_subg", 1)
        self.use_sGPN_score = g("use_gt_subg", 0) == 0
        self.gpn_drop_prob = g("gpn_drop_prob", 0.5)          # nn.Dropout(0.5) of gpn_fc (gpn.py:27)
        self.sg_obj_cnt = _count_names(g("obj_name_path"), g("sg_obj_cnt", 1599))
        self.sg_pred_cnt = _count_names(g("rel_name_path"), g("sg_pred_cnt", 21))
        if self.use_bn:
            raise NotImplementedError("use_bn != 0 (BatchNorm inside att_embed) is not on the Sub-GC presets' path")
        if self.att_hid_size % 4 or self.rnn_size % 4 or self.att_hid_size > 512 or self.rnn_size > 2048:
            raise ValueError("the attention kernels need att_hid_size % 4 == 0 (<= 512) and rnn_size % 4 == 0 (<= 2048)")
        if self.gpn and self.att_feat_size != 2 * self.GCN_dim:
            raise ValueError("att_feat_size must equal 2*gcn_dim: fc_embed consumes the [max|mean] read-out "
                             "(reference AttModel.py:109 with gpn.py:35-36,79)")
        # compute_dtype "bf16" (BASELINE configs 3 / 5): every dense contraction runs on bf16-STORED operands (weights: a bf16
        # snapshot of the fp32 masters; activations that only feed GEMMs: written bf16 by their producers), fp32 accumulation,
        # fp32 master weights / gradients / optimizer state.  Not a reference option (the reference is fp32 only).
        self.bf16_storage = str(g("compute_dtype", "fp32")).lower() in ("bf16", "bfloat16")
        if self.bf16_storage and (self.rnn_size % 8 or self.input_encoding_size % 8 or self.att_hid_size % 8 or self.GCN_dim % 8):
            raise ValueError("compute_dtype=bf16 needs rnn_size, input_encoding_size, att_hid_size and gcn_dim to be multiples of 8")
        # Full-GC only: compute the attention sets (att_embed / ctx2att over the node rows) ONCE per image and let the image's
        # sentences share them (functions.PreparedShared) instead of the reference's x5 replication (gcn_backbone.py:50-51 ->
        # AttModel.py:113-119).  That is the reference's arithmetic exactly when att_embed's Dropout is inactive (eval mode,
        # drop_prob_lm = 0); with dropout ON the reference draws an INDEPENDENT keep-mask for each of the 5 replicated copies, and
        # sharing would tie them within an image.  -1 (default, "auto"): share only when that changes nothing, i.e. a training forward
        # with dropout keeps the reference's five independent masks on replicated rows; 1: always share (tied masks: each sentence's
        # marginal is unchanged, the joint distribution is not the reference's; 7-8 % faster on Full_GC_Kar, reported beside the
        # default by bench.py); 0: never.  Not a reference option.
        self.share_attention_sets = int(g("share_attention_sets", -1))
        self.dedup_att_embed = g("dedup_att_embed", 1) != 0          # 0: att_embed on the replicated rows themselves (measurement / tests)
        # the two GCN units that read the same source run as one paired Function (concatenated fc_lft; functions.UnitPairFn); 0 = one by one
        self.pair_gcn_units = g("pair_gcn_units", 1) != 0
        self.dropout_seed = g("seed", 2019)
        self._dropout_calls = 0
        self.injected_masks = None       # tests inject {'fc','att','xt','out','gpn_hid'} keep-masks here
        self.injected_ss = None          # tests inject (selector uniforms [T,S], draw uniforms [T,S]) for scheduled sampling
        # debug tap (tests only): set to a dict and the next `model(...)` / `_sample` call files its intermediates there under the
        # oracle's / golden files' names (fusion_x, gcn_x_layer{l}, x_obj_out, read_out, att_sel, fc_sel, p_fc, p_att, pp_att,
        # step_{h_att,c_att,h_lang,c_lang,alpha,ctx,logp}); None (default): no cost.  The train forward then runs the unpacked
        # decoder, the decode the eager (not graph-replayed) loop -- same kernels.
        self.__dict__["tap"] = None
        self.__dict__["_nbt_pending"] = {}
        self._build_parameters()

    # ------------------------------------------------------------------ parameters
    def _specs(self):
        """(name, shape, init) in flat-buffer order: encoder first, then the decoder's three slices (`grad_buckets`): their
        gradients are complete first in the backward, so each is all-reduced while the rest of the backward still runs."""
        L, D, A, R, E, V1 = self.GCN_dim, self.att_feat_size, self.att_hid_size, self.rnn_size, self.input_encoding_size, self.vocab_size + 1
        Lr, Ew, FC = self.GCN_lr, self.embed_dim, self.fc_feat_size
        lin = lambda o, i: [((o, i), ("uniform", 1 / math.sqrt(i))), ((o,), ("uniform", 1 / math.sqrt(i)))]
        sp = []

        def add_lin(name, o, i, bias_zero=False):
            (ws, wi), (bs, bi) = lin(o, i)
            sp.append((name + ".weight", ws, wi)); sp.append((name + ".bias", bs, ("zero",) if bias_zero else bi))

        add_lin("obj_v_proj", L, D)
        if self.noun_fuse:
            sp.append(("sg_obj_embed.weight", (self.sg_obj_cnt, Ew), ("normal", 1.0)))
            add_lin("obj_emb_proj", L, Ew)
        sp.append(("sg_pred_embed.weight", (self.sg_pred_cnt, Ew), ("normal", 1.0)))
        add_lin("pred_emb_prj", L, Ew)
        self._gcn_first = len(sp)            # [fusion projections | GCN units, sGPN]: the second part is final before the fusion backward runs
        for l in range(self.GCN_layers):
            for ua in (0, 2):
                # the two units of a pair read the same source rows (graph_conv.py:24-25, 31-32): their fc_lft weights (and biases) lie side
                # by side, so [Wl_a ; Wl_b] is a VIEW and the pair's first product is one N = 2 * 512 launch (functions.UnitPairFn)
                pa, pb = (f"gcn_backbone.gcn.{l}.gcn_collect.collect_units.{u}." for u in (ua, ua + 1))
                sp.append((pa + "fc_lft.weight", (Lr, L), ("normal", 0.001))); sp.append((pb + "fc_lft.weight", (Lr, L), ("normal", 0.001)))
                sp.append((pa + "fc_lft.bias", (Lr,), ("zero",))); sp.append((pb + "fc_lft.bias", (Lr,), ("zero",)))
                for pre in (pa, pb):
                    sp.append((pre + "fc_rgt.weight", (L, Lr), ("normal", 0.001))); sp.append((pre + "fc_rgt.bias", (L,), ("zero",)))
                    if self.GCN_use_bn:
                        sp.append((pre + "bn.weight", (L,), ("one",))); sp.append((pre + "bn.bias", (L,), ("zero",)))
        if self.gpn:
            if self.use_sGPN_score:
                add_lin("gpn_layer.gpn_fc.0", A, 2 * L, bias_zero=True)
                add_lin("gpn_layer.gpn_fc.3", 1, A, bias_zero=True)
            add_lin("gpn_layer.read_out_proj.0", A, 2 * L, bias_zero=True)
            add_lin("gpn_layer.read_out_proj.1", 2 * L, A, bias_zero=True)
        else:
            add_lin("read_out_proj.0", A, L, bias_zero=True)
            add_lin("read_out_proj.1", 2 * L, A, bias_zero=True)
        # Decoder slices in the order the backward FINISHES them, last to first (parallel.GradBucketReducer all-reduces each slice the
        # moment it is final, so every slice must be contiguous): [prepare | recurrent | logit].  logit.* is final right after the
        # criterion backward, BEFORE the BPTT loop (functions*.py: wgrad(21)); the recurrent slice after the loop's batched weight-
        # gradient products; the prepare-feature slice (fc_embed / att_embed / ctx2att) after prepared_backward; the encoder last.
        self._decoder_first = len(sp)
        add_lin("fc_embed.0", FC, D)
        add_lin("fc_embed.2", R, FC)
        add_lin("att_embed.0", R, L)
        add_lin("ctx2att", A, R)
        self._recurrent_first = len(sp)
        sp.append(("embed.0.weight", (V1, E), ("normal", 1.0)))
        add_lin("core.attention.h2att", A, R)
        add_lin("core.attention.alpha_net", 1, A)
        k = 1 / math.sqrt(R)
        for nm, i in (("core.att_lstm", E + 2 * R), ("core.lang_lstm", 2 * R)):
            sp.append((nm + ".weight_ih", (4 * R, i), ("uniform", k))); sp.append((nm + ".weight_hh", (4 * R, R), ("uniform", k)))
            sp.append((nm + ".bias_ih", (4 * R,), ("uniform", k))); sp.append((nm + ".bias_hh", (4 * R,), ("uniform", k)))
        self._logit_first = len(sp)
        add_lin("logit", V1, R)
        return sp

    def _build_parameters(self):
        specs = self._specs()
        # 32-byte (8-float) aligned slots: every parameter can feed the vector GEMM path, and its twin at the same element
        # offset of the bf16 snapshot (`flat_params_b16`) starts on a 16-byte boundary
        offs, total = [], 0
        for _, shape, _ in specs:
            offs.append(total)
            total += (int(np.prod(shape)) + 7) // 8 * 8
        self.flat_params = torch.zeros(total, dtype=torch.float32)
        self.flat_grads = None
        self._slots = {}
        for (name, shape, init), o in zip(specs, offs):
            n = int(np.prod(shape))
            view = self.flat_params[o:o + n].view(shape)
            if init[0] == "uniform":
                view.uniform_(-init[1], init[1])
            elif init[0] == "normal":
                view.normal_(0, init[1])
            elif init[0] == "one":
                view.fill_(1.0)
            _attach(self, name, nn.Parameter(view))
            self._slots[name] = (o, n, shape)
        self.decoder_offset = offs[self._decoder_first]
        self._bucket_bounds = (0, offs[self._gcn_first], offs[self._decoder_first], offs[self._recurrent_first], offs[self._logit_first], total)
        if self.GCN_use_bn:
            for l in range(self.GCN_layers):
                for u in range(4):
                    pre = f"gcn_backbone.gcn.{l}.gcn_collect.collect_units.{u}.bn."
                    _attach(self, pre + "running_mean", torch.zeros(self.GCN_dim), buffer=True)
                    _attach(self, pre + "running_var", torch.ones(self.GCN_dim), buffer=True)
                    _attach(self, pre + "num_batches_tracked", torch.tensor(0, dtype=torch.long), buffer=True)
        self._pmap = dict(self.named_parameters())
        self._bmap = dict(self.named_buffers())

    def _apply(self, fn, recurse=True):
        """.cuda()/.to(): move the flat buffer once and re-point every parameter into it."""
        new_flat = fn(self.flat_params)
        moved = new_flat.device != self.flat_params.device or new_flat.dtype != self.flat_params.dtype
        if not moved:
            return super()._apply(fn, recurse)
        self.flat_params = new_flat
        self.flat_grads = None
        for name, p in self._pmap.items():
            o, n, shape = self._slots[name]
            p.data = self.flat_params[o:o + n].view(shape)
            p.grad = None
        for mod in self.modules():
            for k, b in mod._buffers.items():
                if b is not None:
                    mod._buffers[k] = fn(b)
        self._bmap = dict(self.named_buffers())
        return self

    def flatten_grads(self):
        """Point every .grad into one flat fp32 buffer (zeroed); dead parameters contribute zeros."""
        fresh = self.flat_grads is None or self.flat_grads.device != self.flat_params.device
        if fresh:
            self.flat_grads = torch.zeros_like(self.flat_params)
        elif self.flat_grads.is_cuda:
            # FlatAdam.step(zero_grad=True) left the buffer zeroed through raw pointers (no torch version bump): nothing to fill, unless a
            # torch op has written to it since (its version counter moved)
            # (no torch version bump, and no raw-pointer gradient write since: ops.GRAD_WRITES -- a backward that reached the buffer
            # through any entry point, not only `_forward`, moves it)
            z = self.__dict__.pop("_grads_are_zero", None)
            if z != (self.flat_grads.data_ptr(), self.flat_grads._version, ops.GRAD_WRITES[0]):
                ops.fill_(self.flat_grads, 0.0)
        else:
            self.flat_grads.zero_()
        # re-binding 100 .grad attributes costs ~0.25 ms of host time per step: keep the views, and only bind again when somebody
        # replaced or dropped one (optimizer.zero_grad(set_to_none=True), p.grad = None)
        views = self.__dict__.get("_grad_views")
        if fresh or views is None or any(p.grad is not views[name] for name, p in self._pmap.items()):
            views = {}
            for name, p in self._pmap.items():
                o, n, shape = self._slots[name]
                views[name] = p.grad = self.flat_grads[o:o + n].view(shape)
            self.__dict__["_grad_views"] = views
        return self.flat_grads

    def grad_buckets(self):
        """[(stage, lo, hi)] element ranges of the flat gradient buffer in READINESS order of the backward: "logit" (final before
        the BPTT loop starts), "recurrent" (LSTMs, h2att, alpha_net, word embedding: final after the loop's batched weight-gradient
        products), "prepare" (fc_embed, att_embed, ctx2att), "gcn" (GCN units and the sGPN / read-out layers: final when the gradient
        reaches the fusion outputs, functions.StageMark), "fusion" (obj_v_proj and the class-embedding projections: final when backward
        returns).  The decoder Functions announce the first three through functions.grads_ready(stage), `_encode`'s markers the fourth."""
        f0, g0, p0, r0, l0, end = self._bucket_bounds
        return [("logit", l0, end), ("recurrent", r0, l0), ("prepare", p0, r0), ("gcn", g0, p0), ("fusion", f0, g0)]

    def P(self, name):
        return self._pmap[name]

    # ------------------------------------------------------------------ bf16 weight snapshot (compute_dtype = bf16)
    def _all_versions(self):
        fp = self.flat_params
        return (fp.data_ptr(), fp._version, self.__dict__.get("_cache_epoch", 0)) + tuple(p._version for p in self._pmap.values())

    def weights_b16(self, fresh_from_optimizer=False):
        """The bf16 snapshot of the flat parameter buffer (same element offsets), re-cast whenever the fp32 masters may have
        changed (same version keys as the decode caches).  `fresh_from_optimizer`: the fused Adam sweep has just written it."""
        fp = self.flat_params
        buf = self.__dict__.get("_flat16")
        if buf is None or buf.device != fp.device or buf.numel() != fp.numel():
            buf = torch.empty(fp.numel(), device=fp.device, dtype=torch.bfloat16)
            self.__dict__["_flat16"], self.__dict__["_flat16_key"] = buf, None
        key = self._all_versions()
        if fresh_from_optimizer:
            self.__dict__["_flat16_key"] = key
        elif self.__dict__.get("_flat16_key") != key:
            ops.cast_bf16(fp.view(1, -1), out=buf.view(1, -1))
            self.__dict__["_flat16_key"] = key
        return buf

    def W16(self, name, flat16):
        """bf16 twin of a 2-D parameter inside `flat16` (None when its rows are not multiples of 8 elements: those few
        contractions -- the 300-d class-embedding projections -- stay on the fp32-operand GEMM)."""
        o, n, shape = self._slots[name]
        if len(shape) != 2 or shape[1] % 8:
            return None
        return flat16[o:o + n].view(shape)

    def init_hidden(self, bsz):
        w = self.P("logit.weight")
        return (w.new_zeros(self.num_layers, bsz, self.rnn_size), w.new_zeros(self.num_layers, bsz, self.rnn_size))

    # ------------------------------------------------------------------ the reference's step-level methods (models/stepapi.py)
    def _prepare_feature(self, fc_feats, att_feats, att_masks, sg_emb=None):
        return stepapi.prepare_feature(self, fc_feats, att_feats, att_masks)

    def get_logprobs_state(self, it, fc_feats, att_feats, p_att_feats, att_masks, state, sg_emb=None, p_sg_emb=None, return_att=False):
        return stepapi.get_logprobs_state(self, it, fc_feats, att_feats, p_att_feats, att_masks, state, return_att)

    def beam_search(self, init_state, init_logprobs, *args, **kwargs):
        return stepapi.beam_search(self, init_state, init_logprobs, *args, **kwargs)

    # ------------------------------------------------------------------ dropout masks
    def _rng_seed(self):
        """Philox key of this forward: opt.seed, the call counter and the data-parallel RANK (one process per GPU: without
        the rank every replica would draw the same masks / sampling uniforms for its different shard)."""
        rank = torch.distributed.get_rank() if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 0
        return ((self.dropout_seed * 1000003 + self._dropout_calls) ^ (rank * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF

    def _masks(self, shapes, device):
        """keep-masks for one training forward: injected (tests) or Philox-generated on device."""
        if not self.training:
            return {}
        if self.injected_masks is not None:
            return self.injected_masks
        out, off = {}, 0
        self._dropout_calls += 1
        seed = self._rng_seed()
        live = [(k, shape, p) for k, (shape, p) in shapes.items() if p > 0]
        if len({p for _, _, p in live}) == 1 and len(live) > 1:
            # one keep probability (drop_prob_lm everywhere, gpn_drop_prob equal to it in every preset): the masks are consecutive
            # 4-aligned segments of ONE Philox stream, so one launch over the concatenation produces the same bits as one launch each
            sizes = [(int(np.prod(shape)) + 3) // 4 * 4 for _, shape, _ in live]
            buf = ops.dropout_mask((sum(sizes),), live[0][2], seed, 0, device)
            for (k, shape, _), n in zip(live, sizes):
                out[k] = buf[off:off + int(np.prod(shape))].view(shape)
                off += n
            return out
        for k, shape, p in live:
            out[k] = ops.dropout_mask(shape, p, seed, off, device)
            off += (int(np.prod(shape)) + 3) // 4 * 4
        return out

    # ------------------------------------------------------------------ encoder
    def _gcn_liveness(self):
        Ly, res = self.GCN_layers, self.GCN_residual
        needX, needP = [False] * (Ly + 1), [False] * (Ly + 1)
        live_nodes, live_edges = [False] * Ly, [False] * Ly
        needX[Ly] = True                       # x_pred of the last layer is gathered by sGPN but never used
        for l in range(Ly - 1, -1, -1):
            live_nodes[l], live_edges[l] = needX[l + 1], needP[l + 1]
            needP[l] |= live_nodes[l]
            needX[l] |= live_edges[l]
            if (l + 1) % res == 0:
                s = (l // res) * res
                needX[s] |= live_nodes[l]
                needP[s] |= live_edges[l]
        return needX, needP, live_nodes, live_edges

    def _skip_used(self, s0, live):
        """Does the residual block that starts at layer s0 add its input to a LIVE output (nodes: live = live_nodes, edges: live_edges)?"""
        last = s0 + self.GCN_residual - 1
        return last < self.GCN_layers and live[last]

    def _unit(self, l, u, src, src16=None, w16=None, fuse_bn=False):
        """`src16` / `w16` (compute_dtype = bf16): the bf16 copy of the source rows (shared by the two units that read them) and the
        parameter-name -> bf16 twin lookup; the 512-wide hidden rows then exist in bf16 only.  `fuse_bn`: return the RAW unit output
        (bf16 under compute_dtype = bf16) -- the BatchNorm is applied by the aggregation kernel that consumes it (`_bn_args`)."""
        pre = f"gcn_backbone.gcn.{l}.gcn_collect.collect_units.{u}."
        shp = src.shape
        if w16 is not None:
            h = F_.linear(src.reshape(-1, shp[-1]), self.P(pre + "fc_lft.weight"), self.P(pre + "fc_lft.bias"), W16=w16(pre + "fc_lft.weight"),
                          x16=src16, out_b16=True)
            y = F_.linear(h, self.P(pre + "fc_rgt.weight"), self.P(pre + "fc_rgt.bias"), W16=w16(pre + "fc_rgt.weight"), out_b16=fuse_bn)
        else:
            h = F_.linear(src.reshape(-1, shp[-1]), self.P(pre + "fc_lft.weight"), self.P(pre + "fc_lft.bias"))
            y = F_.linear(h, self.P(pre + "fc_rgt.weight"), self.P(pre + "fc_rgt.bias"))
        if self.GCN_use_bn and not fuse_bn:
            y = F_.BatchNormFn.apply(y, self.P(pre + "bn.weight"), self.P(pre + "bn.bias"), self._bmap[pre + "bn.running_mean"],
                                     self._bmap[pre + "bn.running_var"], self.training)
        if self.GCN_use_bn and self.training:
            self._nbt_pending[pre] = self._nbt_pending.get(pre, 0) + 1                # num_batches_tracked, folded into the buffer lazily
        return y.view(shp[0], shp[1], -1)

    def _pair_cat(self, l, ua, flat16):
        """([Wl_a ; Wl_b], [bl_a ; bl_b], their gradient views or None, the bf16 twin or None) of the unit pair (ua, ua + 1) of layer l
        as views of the flat buffers; None when the two slots are not adjacent (L * 512 or 512 not a multiple of the 8-element slot
        alignment) -- the caller then runs the units one by one."""
        na, nb = (f"gcn_backbone.gcn.{l}.gcn_collect.collect_units.{u}.fc_lft." for u in (ua, ua + 1))
        (ow, nw, shp), (ow2, _, _) = self._slots[na + "weight"], self._slots[nb + "weight"]
        (ob, nbias, _), (ob2, _, _) = self._slots[na + "bias"], self._slots[nb + "bias"]
        if ow2 != ow + nw or ob2 != ob + nbias:
            return None
        Lr, L = shp
        fp, fg = self.flat_params, self.flat_grads
        views = self.__dict__.get("_grad_views") or {}
        bound = fg is not None and all(self.P(n + k).grad is not None and self.P(n + k).grad is views.get(n + k)
                                       for n in (na, nb) for k in ("weight", "bias"))      # the .grad views flatten_grads bound into the flat buffer
        return (fp[ow:ow + 2 * nw].view(2 * Lr, L), fp[ob:ob + 2 * nbias],
                fg[ow:ow + 2 * nw].view(2 * Lr, L) if bound else None, fg[ob:ob + 2 * nbias] if bound else None,
                None if flat16 is None or L % 8 else flat16[ow:ow + 2 * nw].view(2 * Lr, L))

    def _unit_pair(self, l, ua, src, src16, flat16, fuse_bn, out_b16):
        """Units (ua, ua + 1) of layer l on their common source `src` [B, n, L] -> (y_a, y_b) [B, n, L] (functions.UnitPairFn); the
        per-unit path (`_unit`) when the pair cannot be formed (slots not adjacent, no bf16 twin, or a BatchNorm that is not fused into
        the aggregation).  `fuse_bn` / `out_b16`: see `_unit` (raw outputs for the fused BatchNorm; bf16 outputs under compute_dtype = bf16)."""
        cat = self._pair_cat(l, ua, flat16) if self.pair_gcn_units else None
        if cat is None or (flat16 is not None and cat[4] is None) or (self.GCN_use_bn and not fuse_bn):
            xs = F_.fork(src, 2)
            w16 = None if flat16 is None else (lambda name: self.W16(name, flat16))
            return self._unit(l, ua, xs[0], src16, w16, fuse_bn or out_b16), self._unit(l, ua + 1, xs[1], src16, w16, fuse_bn or out_b16)
        pa, pb = (f"gcn_backbone.gcn.{l}.gcn_collect.collect_units.{u}." for u in (ua, ua + 1))
        shp = src.shape
        W16r = None if flat16 is None else (self.W16(pa + "fc_rgt.weight", flat16), self.W16(pb + "fc_rgt.weight", flat16))
        ya, yb = F_.UnitPairFn.apply(src.reshape(-1, shp[-1]), src16, cat, self.P(pa + "fc_lft.weight"), self.P(pa + "fc_lft.bias"),
                                     self.P(pb + "fc_lft.weight"), self.P(pb + "fc_lft.bias"), self.P(pa + "fc_rgt.weight"), self.P(pa + "fc_rgt.bias"),
                                     self.P(pb + "fc_rgt.weight"), self.P(pb + "fc_rgt.bias"), W16r, bool(fuse_bn or out_b16) and flat16 is not None)
        if self.GCN_use_bn and self.training:
            for pre in (pa, pb):
                self._nbt_pending[pre] = self._nbt_pending.get(pre, 0) + 1
        return ya.view(shp[0], shp[1], -1), yb.view(shp[0], shp[1], -1)

    def _bn_args(self, l, ua, ub):
        """(gamma_a, beta_a, gamma_b, beta_b, (running_mean_a, running_var_a, running_mean_b, running_var_b), training) of two units."""
        pa, pb = (f"gcn_backbone.gcn.{l}.gcn_collect.collect_units.{u}.bn." for u in (ua, ub))
        return (self.P(pa + "weight"), self.P(pa + "bias"), self.P(pb + "weight"), self.P(pb + "bias"),
                (self._bmap[pa + "running_mean"], self._bmap[pa + "running_var"], self._bmap[pb + "running_mean"], self._bmap[pb + "running_var"]),
                self.training)

    def _flush_bn_counters(self):
        """nn.BatchNorm1d's num_batches_tracked (a buffer of the state_dict; nothing on the path reads it): counted on the host per
        forward and added to the device buffers only when somebody looks (state_dict / load_state_dict)."""
        pend, self.__dict__["_nbt_pending"] = self.__dict__.get("_nbt_pending", {}), {}
        for pre, n in pend.items():
            self._bmap[pre + "bn.num_batches_tracked"] += n

    def state_dict(self, *args, **kwargs):
        self._flush_bn_counters()
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self.__dict__["_nbt_pending"] = {}
        return super().load_state_dict(*args, **kwargs)

    def _class_proj(self, table, lin, cls):
        """Linear(Embedding[cls]) (AttModel.py:374-377,383-386).  More rows than classes (a training batch): project the TABLE once and
        look the rows up (functions.ClassTableFn); fewer (one image at decode time: 37 rows against 1599 classes): look up, then project."""
        emb = self.P(table)
        if cls.numel() >= emb.size(0):
            return F_.ClassTableFn.apply(emb, self.P(lin + ".weight"), self.P(lin + ".bias"), cls)
        return F_.linear(F_.GatherRowsFn.apply(emb, cls), self.P(lin + ".weight"), self.P(lin + ".bias"))

    def _encode(self, att_feats, obj_dist, pred_dist, rel_ind):
        """feat_fusion + GCN (AttModel.py:370-387, gcn_backbone.py:29-53) -> X_out [B, N, L]."""
        B, N, D = att_feats.shape
        K, L = rel_ind.size(1), self.GCN_dim
        ops.ensure_workspace(att_feats.device)          # scratch for the split-K form of the M=5B recurrent GEMMs
        needX, needP, live_nodes, live_edges = self._gcn_liveness()
        att2 = att_feats.reshape(B * N, D)
        w16 = flat16 = None
        if self.bf16_storage:
            flat16 = self.weights_b16()
            w16 = lambda name: self.W16(name, flat16)
        lin16 = lambda name: {} if w16 is None or w16(name) is None else {"W16": w16(name)}      # fp32-operand GEMM where no twin exists
        # the reference's `.view(-1, sg_obj_cnt)` (AttModel.py:374,381) fails loudly on a width mismatch; here the class ids index
        # the embedding tables on the device, so a wider distribution than the table would read / scatter out of bounds
        if self.noun_fuse and obj_dist.size(-1) != self.sg_obj_cnt:
            raise ValueError(f"obj_dist has {obj_dist.size(-1)} classes, sg_obj_embed has {self.sg_obj_cnt} rows")
        if pred_dist is not None and pred_dist.size(-1) != self.sg_pred_cnt:
            raise ValueError(f"pred_dist has {pred_dist.size(-1)} classes, sg_pred_embed has {self.sg_pred_cnt} rows")
        if self.noun_fuse:
            cls = ops.row_argmax(obj_dist.reshape(B * N, -1), skip=1, i32=True)
            e = self._class_proj("sg_obj_embed.weight", "obj_emb_proj", cls)
            x = F_.linear(att2, self.P("obj_v_proj.weight"), self.P("obj_v_proj.bias"), add=e, relu=True, **lin16("obj_v_proj.weight"))
        else:
            x = F_.linear(att2, self.P("obj_v_proj.weight"), self.P("obj_v_proj.bias"), **lin16("obj_v_proj.weight"))
        x = x.view(B, N, L)
        p = None
        tap = self.__dict__.get("tap")
        if needP[0] or self.GCN_layers == 0 or tap is not None:
            pc = ops.row_argmax(pred_dist.reshape(B * K, -1), skip=1 if self.pred_emb_type == 1 else 0, i32=True)
            p = self._class_proj("sg_pred_embed.weight", "pred_emb_prj", pc).view(B, K, L)
        if tap is not None:
            tap["fusion_x"], tap["fusion_p"] = x.detach().clone(), p.detach().clone()
            if not needP[0] and self.GCN_layers > 0:
                p = None                                            # computed for the tap only: the live graph does not read it
        if self.GCN_layers == 0:
            return x
        if torch.is_grad_enabled() and F_.on_grads_ready is not None:
            # data-parallel training: when the gradient has come back to the fusion outputs, every GCN / sGPN parameter gradient is
            # final -- the "gcn" slice of the bucket goes out while the fusion layers' backward (obj_v_proj: the last product) still runs
            mark = F_.StageMark("gcn")
            x, p = mark(x), mark(p)
        rel_ind = rel_ind.contiguous()
        ptr, edges = ops.csr_build(rel_ind, N)
        skip_x, skip_p = x, p
        x16n = p16n = None
        for l in range(self.GCN_layers):
            res = (l + 1) % self.GCN_residual == 0
            new_x = new_p = None
            # a tensor that feeds several consumers is forked explicitly: the backward then adds the contributions in one launch
            s0 = (l // self.GCN_residual) * self.GCN_residual                                  # layer whose input the residual of this block adds
            # (the two units of a pair take their common source as ONE consumer: functions.UnitPairFn)
            xs = F_.fork(x, (1 if live_edges[l] else 0) + (1 if (l == s0 and self._skip_used(s0, live_nodes)) else 0)) if x is not None else ()
            ps = F_.fork(p, (1 if live_nodes[l] else 0) + (1 if (l == s0 and self._skip_used(s0, live_edges)) else 0)) if p is not None else ()
            if l == s0:
                skip_x = xs[-1] if (x is not None and self._skip_used(s0, live_nodes)) else None
                skip_p = ps[-1] if (p is not None and self._skip_used(s0, live_edges)) else None
            # one bf16 copy per source and layer (compute_dtype = bf16): written by the aggregation kernel that produced the source
            # when that was a fused-BatchNorm one (`x16n` / `p16n` of the previous layer), else a cast pass
            if w16 is not None and live_nodes[l]:
                p16 = p16n if p16n is not None else ops.as_b16(p.reshape(B * K, L))
            else:
                p16 = None
            if w16 is not None and live_edges[l]:
                x16 = x16n if x16n is not None else ops.as_b16(x.reshape(B * N, L))
            else:
                x16 = None
            x16n = p16n = None
            fuse = self.GCN_use_bn and L % 4 == 0
            raw16 = w16 is not None and not self.GCN_use_bn and L % 8 == 0                  # bf16 storage, no BatchNorm: unit outputs stay bf16
            nxt_x = w16 is not None and l + 1 < self.GCN_layers and live_edges[l + 1]          # the next layer reads new_x as a GEMM operand
            nxt_p = w16 is not None and l + 1 < self.GCN_layers and live_nodes[l + 1]
            if live_nodes[l]:
                y0, y1 = self._unit_pair(l, 0, ps[0], p16, flat16, fuse, raw16)
                if fuse:
                    r = F_.GcnNodesBnFn.apply(y0, y1, skip_x if res else None, rel_ind, ptr, edges, N, *self._bn_args(l, 0, 1), nxt_x)
                    new_x, x16n = (r[0], r[1].view(B * N, L)) if nxt_x else (r, None)
                elif raw16:
                    r = F_.GcnNodesB16Fn.apply(y0, y1, skip_x if res else None, rel_ind, ptr, edges, N, nxt_x)
                    new_x, x16n = (r[0], r[1].view(B * N, L)) if nxt_x else (r, None)
                else:
                    new_x = F_.GcnNodesFn.apply(y0, y1, skip_x if res else None, rel_ind, ptr, edges, N)
            if live_edges[l]:
                y2, y3 = self._unit_pair(l, 2, xs[0], x16, flat16, fuse, raw16)
                if fuse:
                    r = F_.GcnEdgesBnFn.apply(y2, y3, skip_p if res else None, rel_ind, ptr, edges, K, *self._bn_args(l, 2, 3), nxt_p)
                    new_p, p16n = (r[0], r[1].view(B * K, L)) if nxt_p else (r, None)
                elif raw16:
                    r = F_.GcnEdgesB16Fn.apply(y2, y3, skip_p if res else None, rel_ind, ptr, edges, K, nxt_p)
                    new_p, p16n = (r[0], r[1].view(B * K, L)) if nxt_p else (r, None)
                else:
                    new_p = F_.GcnEdgesFn.apply(y2, y3, skip_p if res else None, rel_ind, ptr, edges, K)
            x, p = new_x, new_p
            if tap is not None:                                     # dead outputs (None) are simply absent
                if x is not None:
                    tap[f"gcn_x_layer{l}"] = x.detach().float().clone()
                if p is not None:
                    tap[f"gcn_p_layer{l}"] = p.detach().float().clone()
        if tap is not None:
            tap["x_obj_out"] = x.detach().float().clone()
        return x

    # ------------------------------------------------------------------ sGPN
    def _pool(self, X2, idx, w, denom, img, N):
        return F_.SubgraphPoolFn.apply(X2, idx, w, w.stride(0), 1, denom, img, N)

    def _lin(self, x, name, **kw):
        """nn.Linear `name` on 2-D rows; under compute_dtype = bf16 on bf16-stored operands when the weight has a twin."""
        if self.bf16_storage:
            w = self.W16(name + ".weight", self.weights_b16())
            if w is not None:
                return F_.linear(x, self.P(name + ".weight"), self.P(name + ".bias"), W16=w, **kw)
        kw.pop("out_b16", None)
        return F_.linear(x, self.P(name + ".weight"), self.P(name + ".bias"), **kw)

    def _read_out_proj(self, r, prefix):
        h = self._lin(r, prefix + "read_out_proj.0", out_b16=True)              # the 512-wide hidden rows only feed the next product
        return self._lin(h, prefix + "read_out_proj.1")

    def _gpn_train(self, X2, B, gpn_obj_ind, gpn_pool_mtx, att_masks, masks):
        """gpn.py:41-81: score all (pos, neg) sub-graphs, pick the best positive one per sentence.  X2: node states [B*N, L].
        -> (gpn_loss, score [G,1], sel_idx int64 [b5,N], lens int32 [b5], fc [b5, 2L], img_s int32 [b5])."""
        L = X2.size(1)
        b5, _, hb, N = gpn_obj_ind.shape
        spi = b5 // B
        dev = X2.device
        G = 2 * b5 * hb
        # the loader's [b5, 2, hb, ...] tensors as flat pos-half / neg-half arrays: one launch (subgc_gpn_prep)
        idx, w, denom, img = ops.gpn_prep(gpn_obj_ind, gpn_pool_mtx, att_masks, spi)
        read_out = self._pool(X2, idx, w, denom, img, N)
        if self.use_sGPN_score:
            hid = self._lin(read_out, "gpn_layer.gpn_fc.0", relu=True)
            p = self.gpn_drop_prob if self.training else 0.0
            keep = masks.get("gpn_hid") if p > 0 else None
            score, gpn_loss = F_.GpnScoreFn.apply(hid, self.P("gpn_layer.gpn_fc.3.weight"), self.P("gpn_layer.gpn_fc.3.bias"), keep,
                                                  1.0 / (1.0 - p) if keep is not None else 1.0)
        else:
            score, gpn_loss = ops.fill_(torch.empty(G, 1, device=dev, dtype=torch.float32), 1.0), None
        # gpn.py:63-78: first max over the hb positive scores, that sub-graph's node list / node count / read-out row (.detach())
        sel_idx, lens, ro_sel, img_s = ops.gpn_select(score, gpn_obj_ind, att_masks, read_out.detach(), spi)
        fc = self._read_out_proj(ro_sel, "gpn_layer.")
        tap = self.__dict__.get("tap")
        if tap is not None:                                         # att_sel: the reference's gathered node rows (pads = the dummy node's row)
            tap.update(read_out=read_out.detach().clone(), fc_sel=fc.detach().clone(), sel_idx=sel_idx.clone(), sel_lens=lens.clone(),
                       att_sel=X2.detach().view(B, N, L)[img_s.long().view(-1, 1), sel_idx].clone())
        return gpn_loss, score, sel_idx, lens, fc, img_s

    def _consts(self, dev, B, b5, N):
        """Shape-only index tensors of the Full-GC branch (AttModel.py:140-149), built once per (device, batch shape)."""
        key = (str(dev), B, b5, N)
        hit = self.__dict__.setdefault("_const_cache", {}).get(key)
        if hit is None:
            spi = b5 // B
            img_s = torch.div(torch.arange(b5, device=dev, dtype=torch.int32), spi, rounding_mode="floor").to(torch.int32).contiguous()
            ar = torch.arange(N, device=dev).view(1, N)
            hit = dict(img_s=img_s, ar_B=ar.expand(B, N).contiguous(), ar_b5=ar.expand(b5, N).contiguous(), ones=torch.ones(B, N, device=dev),
                       rows=torch.arange(b5, device=dev, dtype=torch.int32),
                       full=torch.full((B,), float(N), device=dev), img_B=torch.arange(B, device=dev, dtype=torch.int32))
            if len(self.__dict__["_const_cache"]) > 16:
                self.__dict__["_const_cache"].clear()
            self.__dict__["_const_cache"][key] = hit
        return hit

    # ------------------------------------------------------------------ train forward
    def _decoder_params(self):
        return [self.P(n) for n in F_.PARAM_ORDER]

    def weights_version(self):
        """Changes whenever the decoder weights may have: the flat buffer's address and in-place version (torch ops on it)
        plus every decoder Parameter's own version counter (load_state_dict's `param.copy_`, torch.optim's in-place updates --
        a Parameter whose `.data` is a view keeps a counter of its own) plus an explicit epoch.  Keys the decode-time snapshots
        (K-concatenated LSTM weights, hipGraphs, the x->gates table).  Writes that torch cannot see bump no counter: the fused
        optimizer step (`parallel.FlatAdam.step`, raw device pointers through the C ABI) therefore calls
        `model.invalidate_decode_caches()` itself, and so must any other raw-pointer or `p.data.<op>_()` write."""
        fp = self.flat_params
        return (fp.data_ptr(), fp._version, self.__dict__.get("_cache_epoch", 0)) + tuple(p._version for p in self._decoder_params())

    def decode_snapshots(self):
        """The dict of weight snapshots shared by every decode state built on the current weights (functions.DecodeState)."""
        key = self.weights_version()
        hit = self.__dict__.get("_decode_snapshots")
        if hit is None or hit[0] != key:
            hit = (key, {})
            self.__dict__["_decode_snapshots"] = hit
        return hit[1]

    def decode_w16(self):
        """bf16 twins of the decoder parameters (aligned with functions.PARAM_ORDER) for the weight-streaming decode step under
        compute_dtype = bf16, else None: the <= 32-row step then streams half the bytes (functions.DecodeState)."""
        if not self.bf16_storage:
            return None
        flat16 = self.weights_b16()
        return [self.W16(n, flat16) for n in F_.PARAM_ORDER]

    def invalidate_decode_caches(self):
        self.__dict__["_cache_epoch"] = self.__dict__.get("_cache_epoch", 0) + 1

    @torch.no_grad()
    def xt_gates_table(self):
        """[V+1, 4R] table relu(Emb) . W_ih_att[:, 2R:]^T for decoding with frozen weights (eval mode; dropout is the identity
        there, AttModel.py:106-108): the word-input term of the attention LSTM's gates depends on the token alone, so decode
        steps look a row up instead of running embed + GEMM (functions.DecodeState).  152 MB at V+1 = 9488, R = 1000; built by one
        [V+1,E]x[E,4R] GEMM (~0.7 ms) and rebuilt whenever the flat parameter buffer changes (its version counter)."""
        fp = self.flat_params
        key = self.weights_version()
        hit = self.__dict__.get("_xt_table")
        if hit is None or hit[0] != key:
            R = self.rnn_size
            emb = torch.relu(self.P("embed.0.weight"))
            tab = torch.empty(emb.size(0), 4 * R, device=fp.device, dtype=torch.float32)
            ops.gemm(emb, self.P("core.att_lstm.weight_ih")[:, 2 * R:], tab, tb=True)
            hit = (key, tab)
            self.__dict__["_xt_table"] = hit
        return hit[1]

    def _forward(self, fc_feats, att_feats, seq, att_masks=None, trip_pred=None, obj_dist=None, obj_box=None, rel_ind=None,
                 pred_fmap=None, pred_dist=None, gpn_obj_ind=None, gpn_pred_ind=None, gpn_nrel_ind=None, gpn_pool_mtx=None,
                 fused_crit=None, need_outputs=True):
        """`fused_crit=(target, mask)` (used by LossWrapper) also evaluates LanguageModelCriterion inside the
        decoder Function, so its backward never materialises the dense d(log-probs); the value is left in
        `self.fused_lang_loss`.  The returned tuple is the reference's either way, except that with
        `need_outputs=False` (LossWrapper only wants the loss) `outputs` is None and the decoder runs packed."""
        B, N, _ = att_feats.shape
        dev = att_feats.device
        self.__dict__.pop("_grads_are_zero", None)      # a backward may follow: "the optimizer left the gradient buffer zeroed" ends here
        L, R, E = self.GCN_dim, self.rnn_size, self.input_encoding_size
        b5, T = seq.size(0), seq.size(1) - 1
        p = self.drop_prob_lm if self.training else 0.0
        hb = gpn_obj_ind.size(2) if gpn_obj_ind is not None else 1
        # the packed decoder's plan kernels keep a sentence's live steps in a 64-bit mask and the sentence order in LDS (csrc/plan.hip:
        # T <= 63, S <= 16384); longer captions / larger shards run the unpacked DecoderFn (same kernels, every step of every sentence)
        tap = self.__dict__.get("tap")
        packed = (fused_crit is not None and not need_outputs and self.packed_decoder and self.injected_masks is None and tap is None
                  and T <= F_.PACKED_MAX_STEPS and b5 <= F_.PACKED_MAX_SENTENCES)
        plan = None
        if packed:                                                                        # the packed decoder's row plan, read behind an event
            from ..functions_packed import PlanAhead
            plan = PlanAhead(seq.contiguous(), fused_crit[1])
        masks = self._masks({"fc": ((b5, R), p), "att": ((b5 * N, R), p), "xt": ((T, b5, E), p), "out": ((T, b5, R), p),
                             "gpn_hid": ((2 * b5 * hb, self.att_hid_size), self.gpn_drop_prob if (self.gpn and self.use_sGPN_score) else 0.0)}, dev)
        X = self._encode(att_feats, obj_dist, pred_dist, rel_ind)
        if self.gpn:
            Xa, Xb = F_.fork(X.reshape(B * N, L), 2)                                       # node states feed the sGPN pooling and the decoder
            gpn_loss, score, sel_idx, lens, fc, img_s = self._gpn_train(Xa, B, gpn_obj_ind, gpn_pool_mtx, att_masks, masks)
        else:                                                                             # AttModel.py:140-149
            gpn_loss = score = None
            Xb = X.reshape(B * N, L)
            c = self._consts(dev, B, b5, N)
            img_s = c["img_s"]
            mean = self._pool(Xb.detach(), c["ar_B"], c["ones"], c["full"], c["img_B"], N)[:, L:]
            mean5 = ops.gather_rows(mean, img_s, torch.empty(b5, L, device=dev, dtype=torch.float32))
            fc = self._read_out_proj(mean5, "")
            mask_sel = att_masks[:, 0, 0]
            ops.fill2d_(mask_sel[:, :36], 1.0)                                            # in place on the caller's tensor
            sel_idx = c["ar_b5"]
            lens = ops.row_count(mask_sel)
        meta = {"N": N, "p": p, "masks": masks, "crit": fused_crit, "plan": plan, "tap": tap}
        share = self.share_attention_sets == 1 or (self.share_attention_sets < 0 and p == 0.0)      # auto: only where it is exact
        if (not self.gpn and share and self.injected_masks is None and b5 % B == 0
                and F_.shared_sets_ok(b5 // B, N, self.att_hid_size, R, T)):
            meta["shared"] = {"B": B, "g": b5 // B, "rows": c["rows"]}                    # every sentence attends over its image's N node rows
        elif not self.gpn and self.dedup_att_embed and b5 >= 2 * B:
            # Full-GC on replicated rows (the reference's independent per-sentence dropout masks): att_embed's Linear + ReLU once per
            # NODE row, each sentence's copy gathered through its own keep-mask (functions.Prepared, dedup)
            meta["dedup_att_embed"] = True
        if self.bf16_storage:
            flat16 = self.weights_b16()
            meta["W16"] = [self.W16(n, flat16) for n in F_.PARAM_ORDER]
        ss_on = self.training and self.ss_prob > 0.0
        if ss_on:                                                                         # scheduled sampling, AttModel.py:157-167
            if self.injected_ss is not None:
                sel_u, u = self.injected_ss
            else:
                self._dropout_calls += 1
                seed = self._rng_seed()
                both = ops.uniform((2, T, b5), seed ^ 0x5C4ED51ED5A3B11F, 0, dev)
                sel_u, u = both[0], both[1]
            meta["ss"] = (float(self.ss_prob), sel_u.contiguous(), u.contiguous())
        if packed:
            # loss-only call (LossWrapper): length-sorted packed decoder, dead (masked-out) steps are never computed
            from ..functions_packed import PackedDecoderLossFn
            self.fused_lang_loss = PackedDecoderLossFn.apply(meta, seq.contiguous(), fc, Xb, lens, sel_idx, img_s, *self._decoder_params())
            return None, gpn_loss, score
        outputs, lang_loss = F_.DecoderFn.apply(meta, seq.contiguous(), fc, Xb, lens, sel_idx, img_s, *self._decoder_params())
        self.fused_lang_loss = lang_loss if fused_crit is not None else None
        return outputs, gpn_loss, score

    # ------------------------------------------------------------------ decode
    def _sample_sentences(self, *args, **kwargs):
        """Beam search (reference AttModel.py:179-234): same entry as `_sample`, which routes here on beam_size > 1."""
        opt = dict(kwargs.pop("opt", {}))
        if opt.get("beam_size", 10) <= 1:
            raise ValueError("_sample_sentences needs beam_size > 1")
        return self._sample(*args, opt=opt, **kwargs)

    @torch.no_grad()
    def _sample(self, fc_feats, att_feats, att_masks=None, trip_pred=None, obj_dist=None, obj_box=None, rel_ind=None,
                pred_fmap=None, pred_dist=None, gpn_obj_ind=None, gpn_pred_ind=None, gpn_nrel_ind=None, gpn_pool_mtx=None, opt={},
                uniforms=None, forced=None):
        """Greedy / top-k decode of one image (AttModel.py:236-326).  `uniforms[n, T]` (optional)
        supplies the top-k sampler's random numbers; `forced[n, T]` makes the loop follow a given
        token path (both exist so tests can pin the sampler)."""
        # the loader hands 5 identical "counterparts" of the image and the reference's test branch reads counterpart 0 only
        # (gpn.py:84-96, AttModel.py:261-271): encode that one -- a fifth of the rows, and every encoder GEMM fits the
        # weight-streaming form
        N = att_feats.size(1)
        X2 = self._encode(att_feats[:1], obj_dist[:1], pred_dist[:1], rel_ind[:1]).reshape(N, self.GCN_dim).contiguous()
        return sampling.decode_one_image(self, X2, N, (0, gpn_obj_ind, att_masks, gpn_pool_mtx), opt, uniforms, forced)

    @torch.no_grad()
    def sample_images(self, images, opt={}, batch_out=None):
        """Decode MANY images in one batch (not in the reference, whose loop is one image per call, eval_utils.py:98-104).
        `images`: list of dicts with the test loader's keys (att_feats [1,N,D], obj_dist, pred_dist, rel_ind, att_masks,
        gpn_obj_ind, gpn_pool_mtx -- the 5-counterpart layout of dataloader_test.py).  Returns one `_sample` tuple per image.
        `batch_out`: see sampling.decode (the whole batch's tensors for the batched eval glue)."""
        # counterpart 0 of every image's loader tensors, stacked into one batch by ONE table upload + four launches (subgc_gather_blocks)
        att, obj, pred, rel = ops.stack_first([[im[k] for im in images] for k in ("att_feats", "obj_dist", "pred_dist", "rel_ind")])
        I, N, _ = att.shape
        X2 = self._encode(att, obj, pred, rel).reshape(I * N, self.GCN_dim).contiguous()
        rows = [(i, im["gpn_obj_ind"], im["att_masks"], im["gpn_pool_mtx"]) for i, im in enumerate(images)]
        sel = sampling.select_subgraphs(self, X2, N, rows) if self.gpn else sampling.full_graph_rows(self, X2, N, rows)
        return sampling.decode(self, X2, N, sel, opt, batch_out=batch_out)


class TopDownModel(AttModel):
    """reference AttModel.py:476-480 (num_layers is forced to 2: att-LSTM + lang-LSTM)."""

    def __init__(self, opt):
        super().__init__(opt)
        self.num_layers = 2
