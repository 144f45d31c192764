#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE (CPU, fp32).

Runs only where /root/reference exists (the build container).  The reference's Python is
imported from where it lies, never copied; what is committed is data: weights, inputs,
expected outputs and intermediates of a tiny configuration that walks the same code as
Sub_GC_Kar / Full_GC_Kar (N=37 nodes, K=65 relations so the hard-coded 36-isms are hit).

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz + meta.json

Recipe: a scratch cwd with a stub `data/glove.6B.<dim>d.pt` so that
`misc/utils.py:348-422` finds "word vectors" (all class embeddings then come from the seeded
`normal_`), `sys.path[0] = /root/reference`, an argparse.Namespace with the fields the model
reads, synthetic inputs from subgc.synthetic.
"""
import argparse
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, os.path.join(ROOT, "sub-gc_amd"))
from subgc import synthetic  # noqa: E402

TINY = dict(rnn_size=48, input_encoding_size=48, att_feat_size=64, gcn_dim=32, fc_feat_size=40,
            att_hid_size=24, vocab_size=50, embed_dim=20, seq_length=16, max_length=20)


def ref_opt(**over):
    o = dict(caption_model="topdown", num_layers=1, drop_prob_lm=0.0, use_bn=0, sampling_prob=0.0,
             use_gpn=1, noun_fuse=1, pred_emb_type=1, gcn_layers=2, gcn_residual=2, gcn_bn=0,
             obj_name_path=os.path.join(REF, "data/object_names_1600-0-20.npy"),
             rel_name_path=os.path.join(REF, "data/predicate_names_1600-0-20.npy"), **TINY)
    o.update(over)
    return argparse.Namespace(**o)


def enter_scratch():
    d = tempfile.mkdtemp(prefix="subgc_golden_")
    os.makedirs(os.path.join(d, "data"))
    for dim in (TINY["embed_dim"], 300):
        torch.save(({"the": 0}, torch.zeros(1, dim), dim), os.path.join(d, "data", f"glove.6B.{dim}d.pt"))
    os.chdir(d)
    sys.path.insert(0, REF)


def np_(t):
    return t.detach().cpu().numpy().copy()


def build(opt, seed, gcn_scale):
    from models.AttModel import TopDownModel
    torch.manual_seed(seed)
    m = TopDownModel(opt)
    with torch.no_grad():  # make the GCN numerically visible (default init is N(0,1e-3^2))
        for n, p in m.named_parameters():
            if "gcn_collect" in n and ("fc_lft.weight" in n or "fc_rgt.weight" in n):
                p.mul_(gcn_scale)
            if "gcn_collect" in n and n.endswith("bias") and ".bn." not in n:
                p.normal_(0, 0.05)
        if hasattr(m, "gpn_layer") and hasattr(m.gpn_layer, "gpn_fc"):
            m.gpn_layer.gpn_fc[2].p = 0.0
            m.gpn_layer.gpn_fc[0].weight.mul_(4.0)   # spread the sGPN scores
            m.gpn_layer.gpn_fc[3].weight.mul_(4.0)
        # default init decodes one constant token for ever: sharpen the decoder so greedy paths
        # vary, some reach EOS (token 0) early and the unfinished-masking is exercised
        for n, p in m.named_parameters():
            if n.startswith("core.") and "lstm" in n and "weight" in n:
                p.mul_(3.0)
            if n in ("logit.weight", "core.attention.h2att.weight", "core.attention.alpha_net.weight", "ctx2att.weight"):
                p.mul_(8.0)
        m.logit.bias[0] += 1.0
    return m


class Tap:
    """Capture intermediates of one reference call without touching its source."""

    def __init__(self, model):
        self.m, self.rec, self.hooks = model, {}, []
        self.steps = dict(h_att=[], c_att=[], h_lang=[], c_lang=[], alpha=[], ctx=[], logp=[])
        m = model
        self._wrap(m, "feat_fusion", lambda out: self.rec.update(fusion_x=np_(out[0]), fusion_p=np_(out[1])))
        self._wrap(m, "_prepare_feature", lambda out: self.rec.update(
            p_fc=np_(out[0]), p_att=np_(out[1]), pp_att=np_(out[2]), p_mask=np_(out[3])))
        for l, layer in enumerate(m.gcn_backbone.gcn):
            self.hooks.append(layer.register_forward_hook(
                lambda mod, i, o, l=l: self.rec.update({f"gcn_x_layer{l}": np_(o[0]), f"gcn_p_layer{l}": np_(o[1])})))
        self.hooks.append(m.gcn_backbone.register_forward_hook(
            lambda mod, i, o: self.rec.update(x_obj_out=np_(o[0][::5]), x_pred_out=np_(o[1][::5]))))
        if m.gpn:
            self._wrap(m.gpn_layer, "graph_pooling", lambda out: self.rec.update(read_out=np_(out)))
            self.hooks.append(m.gpn_layer.register_forward_hook(self._gpn_out))
        self.hooks.append(m.core.attention.register_forward_hook(self._att))
        self.hooks.append(m.core.register_forward_hook(self._core))
        self._wrap(m, "get_logprobs_state", lambda out: self.steps["logp"].append(np_(out[0])))

    def _wrap(self, obj, name, fn):
        orig = getattr(obj, name)

        def w(*a, **k):
            out = orig(*a, **k)
            fn(out)
            return out
        setattr(obj, name, w)

    def _gpn_out(self, mod, i, o):
        self.rec.update(subgraph_score_raw=np_(o[1]), att_sel=np_(o[2]), fc_sel=np_(o[3]), mask_sel=np_(o[4]))
        if o[0] is not None:
            self.rec["gpn_loss"] = np_(o[0])
        if len(o) > 5:
            self.rec["keep_ind"] = np_(o[5])

    def _att(self, mod, i, o):
        with torch.no_grad():
            h, att_feats, p_att, masks = i[:4] if len(i) >= 4 else (tuple(i) + (None,))[:4]
            _, w = type(mod).forward(mod, h, att_feats, p_att, masks, return_att=True)
        self.steps["alpha"].append(np_(w))
        self.steps["ctx"].append(np_(o[0] if isinstance(o, tuple) else o))

    def _core(self, mod, i, o):
        st = o[1]
        self.steps["h_att"].append(np_(st[0][0])); self.steps["h_lang"].append(np_(st[0][1]))
        self.steps["c_att"].append(np_(st[1][0])); self.steps["c_lang"].append(np_(st[1][1]))

    def done(self):
        for h in self.hooks:
            h.remove()
        for k, v in self.steps.items():
            if v and len({x.shape for x in v}) == 1:            # diverse beam search steps groups of different widths
                self.rec["step_" + k] = np.stack(v, 0)
        return self.rec


def save(name, **groups):
    for g, d in groups.items():
        np.savez_compressed(os.path.join(HERE, f"{name}_{g}.npz"), **d)


def run_train(name, opt, seed, B, gcn_scale, meta):
    from misc.utils import LanguageModelCriterion
    model = build(opt, seed, gcn_scale)
    model.train()
    batch = synthetic.make_train_batch(B, D=opt.att_feat_size, vocab=opt.vocab_size, seq_length=opt.seq_length,
                                       seed=seed + 100, fc_size=opt.att_feat_size)
    weights = {k: np_(v) for k, v in model.state_dict().items()}
    tap = Tap(model)
    args = synthetic.forward_args({k: v.clone() for k, v in batch.items()})
    outputs, gpn_loss, score = model(*args)
    lang_loss = LanguageModelCriterion()(outputs, batch["labels"][:, 1:], batch["masks"][:, 1:])
    loss = lang_loss + (gpn_loss if gpn_loss is not None else 0.0)
    loss.backward()
    rec = tap.done()
    rec.update(outputs=np_(outputs), lang_loss=np_(lang_loss), loss=np_(loss))
    if score is not None:
        rec["subgraph_score"] = np_(score)
    grads = {k: np_(p.grad) for k, p in model.named_parameters() if p.grad is not None}
    meta[name] = dict(kind="train", B=B, seed=seed, gcn_scale=gcn_scale, opt={k: v for k, v in vars(opt).items() if "path" not in k},
                      dead_params=[k for k, p in model.named_parameters() if p.grad is None],
                      buffers_after={})
    # BN running stats after this one training forward (Full-GC): part of the contract
    after = {k: np_(v) for k, v in model.state_dict().items() if "running_" in k or "num_batches" in k}
    save(name, weights=weights, inputs={k: v.numpy() for k, v in batch.items()}, out=rec, grads=grads, **({"bn_after": after} if after else {}))
    return model, weights


def run_sample(name, opt, weights, seed, M, sample_opt, meta, node_pool=None, keys=None):
    from models.AttModel import TopDownModel
    torch.manual_seed(seed)
    model = TopDownModel(opt)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})
    model.eval()
    batch = synthetic.make_test_batch(M, D=opt.att_feat_size, seed=seed + 200, fc_size=opt.att_feat_size, node_pool=node_pool)
    tap = Tap(model)
    args = synthetic.sample_args({k: v.clone() for k, v in batch.items()})
    torch.manual_seed(seed + 7)
    with torch.no_grad():
        ret = model(*args, opt=dict(sample_opt), mode="sample")
    rec = tap.done()
    rec.update(seq=np_(ret[0]), seqLogprobs=np_(ret[1]), subgraph_score=np_(ret[2]), keep_ind=np_(ret[3]))
    if sample_opt.get("beam_size", 1) > 1:                      # every finished beam the reference keeps (AttModel.py:229)
        rec["done_seq"] = np.stack([np.stack([np_(b["seq"]) for b in beams]) for beams in model.done_beams])
        rec["done_logps"] = np.stack([np.stack([np_(b["logps"]) for b in beams]) for beams in model.done_beams])
        rec["done_p"] = np.array([[float(b["p"]) for b in beams] for beams in model.done_beams], np.float64)
    if len(ret) > 4:
        rec["att2_weights"] = np_(ret[4])
    if keys is not None:
        rec = {k: v for k, v in rec.items() if k in keys}
    meta[name] = dict(kind="sample", M=M, seed=seed, sample_opt=sample_opt, node_pool=node_pool,
                      opt={k: v for k, v in vars(opt).items() if "path" not in k})
    save(name, inputs={k: v.numpy() for k, v in batch.items()}, out=rec)


BEAM_KEYS = ("seq", "seqLogprobs", "subgraph_score", "keep_ind", "done_seq", "done_logps", "done_p")


def beam_cases(w, meta):
    """Beam search (CaptionModel.py:28-176).  The reference moves the chosen words with `.cuda()` (:135,171);
    on this CPU-only box that call is made the identity for the duration of the run."""
    t = dict(test_LSTM=1, gpn_nms_thres=0.75, gpn_max_subg=10)
    w = {k: v.copy() for k, v in w.items()}
    w["logit.weight"][0] *= 6.0                                  # a state-dependent <eos>: beams end at 0, 1, 3, 7, ... 20 words
    w["logit.bias"][0] -= 1.0
    save("subgc_beam", weights=w)
    orig = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        run_sample("subgc_beam3", ref_opt(**t), w, seed=12, M=16, sample_opt=dict(sample_max=1, beam_size=3), meta=meta, node_pool=14,
                   keys=BEAM_KEYS)
        run_sample("subgc_beam2_wu", ref_opt(**t), w, seed=13, M=10, sample_opt=dict(sample_max=1, beam_size=2, length_penalty="wu_0.7"),
                   meta=meta, node_pool=12, keys=BEAM_KEYS)
        run_sample("subgc_beam4_div", ref_opt(**t), w, seed=14, M=12,
                   sample_opt=dict(sample_max=1, beam_size=4, group_size=2, diversity_lambda=0.5, decoding_constraint=1,
                                   length_penalty="avg_1.0"), meta=meta, node_pool=12, keys=BEAM_KEYS)
        run_sample("subgc_beam6_div3", ref_opt(**t), w, seed=15, M=8,
                   sample_opt=dict(sample_max=1, beam_size=6, group_size=3, diversity_lambda=0.3), meta=meta, node_pool=12, keys=BEAM_KEYS)
    finally:
        torch.Tensor.cuda = orig


def ss_case(w, meta):
    """Scheduled sampling (AttModel.py:157-167) on the subgc_train weights and inputs, p = 0.25.  The two random streams the
    reference consumes there are replaced, for the duration of the forward, by injected numbers: `Tensor.uniform_` on the
    [batch] selector returns sel_u[i]; `torch.multinomial(prob_prev, 1)` returns the inverse CDF of prob_prev at u[i] in index
    order.  Everything else -- which rows are re-drawn, from which step's distribution, that the draw replaces the INPUT word
    only and carries no gradient, the early break -- is the reference's own code.  Stored: the injected numbers, the words
    actually fed at every step, outputs, losses, every parameter gradient."""
    from misc.utils import LanguageModelCriterion
    with open(os.path.join(HERE, "meta.json")) as f:
        base = json.load(f)["subgc_train"]
    opt = ref_opt(sampling_prob=0.25)
    model = build(opt, base["seed"], base["gcn_scale"])
    for k, v in model.state_dict().items():
        assert np.array_equal(np_(v), w[k]), k                    # same construction as subgc_train: no second weight file
    model.train()
    assert model.ss_prob == 0.25
    with np.load(os.path.join(HERE, "subgc_train_inputs.npz")) as z:
        batch = {k: torch.from_numpy(z[k]) for k in z.files}
    S, T = batch["labels"].shape[0], batch["labels"].shape[1] - 1
    gen = torch.Generator().manual_seed(4242)
    sel_u, u = torch.rand(T, S, generator=gen), torch.rand(T, S, generator=gen)
    state = dict(step=0, fed=[], n_uniform=0, n_multi=0)
    orig_uniform, orig_multi, orig_step = torch.Tensor.uniform_, torch.multinomial, model.get_logprobs_state

    def uniform_(self, *a, **k):
        assert tuple(self.shape) == (S,) and a == (0, 1)
        state["n_uniform"] += 1
        state["step"] += 1                                        # the reference draws the selector once per step i >= 1, in order
        return self.copy_(sel_u[state["step"]])

    def multinomial(prob, n, *a, **k):
        assert n == 1 and tuple(prob.shape) == (S, opt.vocab_size + 1)
        state["n_multi"] += 1
        cdf = torch.cumsum(prob, 1)
        draw = (cdf <= (u[state["step"]] * cdf[:, -1]).unsqueeze(1)).sum(1).clamp(max=cdf.size(1) - 1)
        return draw.view(-1, 1)

    def step(it, *a, **k):
        state["fed"].append(np_(it))
        return orig_step(it, *a, **k)

    torch.Tensor.uniform_, torch.multinomial, model.get_logprobs_state = uniform_, multinomial, step
    try:
        outputs, gpn_loss, score = model(*synthetic.forward_args({k: v.clone() for k, v in batch.items()}))
    finally:
        torch.Tensor.uniform_, torch.multinomial = orig_uniform, orig_multi
    lang_loss = LanguageModelCriterion()(outputs, batch["labels"][:, 1:], batch["masks"][:, 1:])
    loss = lang_loss + gpn_loss
    loss.backward()
    fed = np.stack(state["fed"], 0)                               # [steps run, S]
    changed = int((fed != batch["labels"][:, :fed.shape[0]].numpy().T).sum())
    assert state["n_uniform"] >= fed.shape[0] - 1 and state["n_multi"] > 0 and changed > 10, (state["n_uniform"], state["n_multi"], changed)
    rec = dict(outputs=np_(outputs), lang_loss=np_(lang_loss), gpn_loss=np_(gpn_loss), loss=np_(loss), subgraph_score=np_(score),
               sel_u=sel_u.numpy(), u=u.numpy(), fed_tokens=fed)
    grads = {k: np_(p.grad) for k, p in model.named_parameters() if p.grad is not None}
    meta["subgc_ss_train"] = dict(kind="train_ss", weights="subgc_train", inputs="subgc_train", sampling_prob=0.25, changed_words=changed,
                                  opt={k: v for k, v in vars(opt).items() if "path" not in k},
                                  dead_params=[k for k, p in model.named_parameters() if p.grad is None])
    save("subgc_ss_train", out=rec, grads=grads)


class _LoggedList(list):
    """`subgraph_mask_list` that remembers which entries the loader read, in order (how the fixture learns the sub-graph
    ids `__getitem__` drew without touching its source)."""

    def __init__(self, items):
        super().__init__(items)
        self.log = []

    def __getitem__(self, i):
        self.log.append(int(i))
        return super().__getitem__(i)


def loader_cases(meta):
    """dataloaders/dataloader.py:225-367 `DataLoader.__getitem__` driven on fabricated dataset entries: `h5py` (absent here,
    only used by `__init__`) is stubbed so the module imports, the object is made with `object.__new__` and given exactly the
    attributes `__getitem__` / `get_captions` read.  Both branches: sampled sub-graph mini-batches (seeded np.random) and
    `use_gt_subg`.  Stored per image: the fabricated raw entries, the ids the loader drew, its 13 returned arrays."""
    import random as pyrandom
    sys.modules.setdefault("h5py", types.ModuleType("h5py"))
    from dataloaders.dataloader import DataLoader
    obj_num, rel_num, hb, S, Lq, D, C, Pd = 37, 65, 2, 5, 16, 8, 7, 5
    rng = np.random.default_rng(99)
    n_img = 6
    n_cand = [9, 3, 14, 6, 2, 11]                                 # sampled candidates per image (after the 5 sentence sub-graphs)
    n_rel = [0, rel_num + 9, 17, rel_num - 1, 40, 5]              # none / more than fit / short / exactly full / ...
    n_cap = [5, 7, 3, 5, 1, 6]                                    # >= 5: the first five; < 5: random.randint with replacement
    images, labels_all, start, end = [], [], [], []
    for b in range(n_img):
        M = 5 + n_cand[b]
        iou = rng.random((S, M))
        if b == 1:
            iou[:, 5:] = 0.9                                      # every candidate positive for every sentence: no negatives at all
        if b == 4:
            iou[:, 5:] = 0.1                                      # no positives: padded with the sentence's own sub-graph (i - 5)
        if b == 2:
            iou[0, 5:] = 0.75                                     # exactly at the threshold: positive (>=) AND counted by the <= fallback
        masks = []
        for j in range(M):
            nm = rng.random(obj_num - 1) < 0.25
            if j == 6:
                nm[:] = False                                     # an empty sub-graph
            pm = rng.random(rel_num - 1) < 0.1
            nrel = rng.integers(0, obj_num - 1, size=(int(pm.sum()), 2))
            masks.append((j, nm, pm, nrel))
        sg = dict(object_fmap=rng.standard_normal((obj_num - 1, D)).astype(np.float32), object_dist=rng.random((obj_num - 1, C)).astype(np.float32),
                  pred_dist=rng.random((n_rel[b], Pd)).astype(np.float32), rel_ind=rng.integers(0, obj_num - 1, size=(n_rel[b], 2)))
        caps = rng.integers(1, 50, size=(n_cap[b], Lq))
        for r in range(n_cap[b]):
            caps[r, rng.integers(0, Lq + 1):] = 0
        start.append(len(labels_all) + 1)                         # 1-based like the h5 file
        labels_all.extend(list(caps))
        end.append(len(labels_all))
        images.append(dict(iou=iou, masks=masks, sg=sg))
    dl = object.__new__(DataLoader)
    dl.info = {"images": [{"id": 1000 + b} for b in range(n_img)]}
    dl.seq_per_img, dl.half_mini_batch, dl.obj_num, dl.rel_num, dl.seq_length, dl.thres = S, hb, obj_num, rel_num, Lq, 0.75
    dl.label = np.stack(labels_all).astype(np.int64)
    dl.label_start_ix, dl.label_end_ix = np.array(start), np.array(end)
    logged = {}

    class _Masks:
        def get(self, key):
            im = images[int(key) - 1000]
            logged[key] = _LoggedList(im["masks"])
            return {"node_iou_mtx": im["iou"].copy(), "subgraph_mask_list": logged[key]}

    class _Trip:
        def get(self, key):
            return images[int(key) - 1000]["sg"]

    dl.subgraph_mask, dl.trip_loader = _Masks(), _Trip()
    names = ("fc_feats", "att_feats", "obj_dist", "rel_ind", "pred_dist", "labels", "masks", "ix", "gpn_obj_ind", "gpn_pred_ind",
             "gpn_nrel_ind", "att_masks", "gpn_pool_mtx")
    out = {}
    branches = set()
    for gt in (0, 1):
        dl.use_gt_subg = gt
        np.random.seed(2024 + gt)
        pyrandom.seed(7 + gt)
        for b in range(n_img):
            ret = dl.__getitem__(b)
            tag = f"{'gt' if gt else 'smp'}{b}"
            for nme, arr in zip(names, ret):
                out[f"{tag}_{nme}"] = np.asarray(arr)
            log = logged[str(1000 + b)].log
            if gt:
                assert log == [i for i in range(S) for _ in range(3)]
            else:
                ids = np.array(log).reshape(S, hb, 3, 2)             # per (i, k): [obj, pred, nrel] x [pos, neg]
                assert (ids[:, :, 0] == ids[:, :, 1]).all() and (ids[:, :, 0] == ids[:, :, 2]).all()
                out[f"{tag}_mask_idx"] = ids[:, :, 0]                # [S, hb, 2] AFTER the +5 shift (:270)
                iou = images[b]["iou"][:, 5:]
                pos, neg = iou >= 0.75, iou < 0.75
                neg[:, pos.nonzero()[1]] = 0
                for i in range(S):
                    branches.add("pos_pad" if pos[i].sum() < hb else "pos_draw")
                    if neg[i].sum() >= hb:
                        branches.add("neg_plain")
                    elif (iou[i] <= 0.75).sum() == 0:
                        branches.add("neg_all")
                    elif neg[i].sum() == 0:
                        branches.add("neg_le_thres")
                    else:
                        branches.add("neg_few")
    assert branches == {"pos_pad", "pos_draw", "neg_plain", "neg_all", "neg_le_thres", "neg_few"}, branches
    raw = {}
    for b, im in enumerate(images):
        raw[f"img{b}_node_iou_mtx"] = im["iou"]
        raw[f"img{b}_node_masks"] = np.stack([m[1] for m in im["masks"]])
        raw[f"img{b}_pred_masks"] = np.stack([m[2] for m in im["masks"]])
        raw[f"img{b}_nrel"] = np.concatenate([m[3] for m in im["masks"]]).reshape(-1, 2)
        raw[f"img{b}_nrel_off"] = np.cumsum([0] + [m[3].shape[0] for m in im["masks"]])
        for k, v in im["sg"].items():
            raw[f"img{b}_{k}"] = v
    raw.update(label=dl.label, label_start_ix=dl.label_start_ix, label_end_ix=dl.label_end_ix)
    meta["loader"] = dict(kind="loader", obj_num=obj_num, rel_num=rel_num, gpn_batch=hb, seq_per_img=S, seq_length=Lq, thres=0.75,
                          n_images=n_img, np_seed=[2024, 2025], py_seed=[7, 8], branches=sorted(branches))
    save("loader", inputs=raw, out=out)


def eval_cases(meta):
    """misc/utils.py:59-81 decode_sequence (with and without REMOVE_BAD_ENDINGS) on token rows that end in function words."""
    import misc.utils as U
    rng = np.random.default_rng(21)
    words = list(U.bad_endings) + [f"w{i}" for i in range(40)]
    vocab = {str(i + 1): w for i, w in enumerate(words)}
    seq = rng.integers(1, len(words) + 1, size=(40, 12))
    for r in range(40):
        seq[r, rng.integers(0, 13):] = 0                          # random length incl. empty and full rows
    seq[3, :4] = [20, 1, 2, 3]; seq[3, 4:] = 0                    # "w.. with in on": everything after the first word is stripped
    seq[4, :3] = [1, 2, 14]; seq[4, 3:] = 0                       # only function words
    out = {}
    for flag in ("0", "1"):
        os.environ["REMOVE_BAD_ENDINGS"] = flag
        out["sents_" + flag] = np.array(U.decode_sequence(vocab, torch.from_numpy(seq)))
    os.environ["REMOVE_BAD_ENDINGS"] = "0"
    meta["eval_glue"] = dict(kind="eval", vocab=vocab)
    save("eval_glue", out=dict(seq=seq, **out))


def grd_cases(meta):
    """Grounding material (misc/grd_utils.py:13-58), collected by the eval loop at misc/eval_utils.py:143-146.  The reference's
    OWN `get_grounding_material` is run on the outputs of its own `_sample(..., return_att=1)` (ranked like eval_utils.py:105-115,
    sentences by its decode_sequence); the files it opens by image id (sub-graph node masks, detector boxes, image sizes, the
    consensus re-ranker's choice) are fabricated in the scratch directory from the synthetic test batch, the word -> lemma ->
    detection-class dictionaries make most words groundable nouns.  Stored: boxes / sizes / picks, sentences, and what the
    reference appended to grd_output (bbox, idx_in_sent, clss) -- the boxes identify the arg-max node of every grounded word."""
    import misc.utils as U
    from misc.grd_utils import get_grounding_material
    from models.AttModel import TopDownModel
    with np.load(os.path.join(HERE, "subgc_beam_weights.npz")) as z:
        w = {k: z[k].copy() for k in z.files}
    w["logit.bias"][0] += 0.5                                    # sentences of 0 .. 20 words
    with np.load(os.path.join(HERE, "fullgc_train_weights.npz")) as z:
        wf = {k: z[k].copy() for k in z.files}
    t = dict(test_LSTM=1, gpn_nms_thres=0.75, gpn_max_subg=10)
    fo = dict(use_gpn=0, noun_fuse=0, pred_emb_type=2, gcn_layers=4, gcn_residual=1, gcn_bn=1)
    V = TINY["vocab_size"]
    vocab = {str(i): f"w{i}" for i in range(1, V + 1)}
    wd_to_lemma = {f"w{i}": f"l{i}" for i in range(1, V + 1) if i % 7}              # every 7th word is unknown to the lemmatiser (:51-53)
    lemma_det = {f"l{i}": i for i in range(1, V + 1) if i % 3}                       # two lemmas in three are detection classes
    det_wd = {i: f"cls{i}" for i in range(1, V + 1)}
    cases = [("subgc", ref_opt(**t), w, True, [(24, 14, 900, 0), (5, None, 901, 0), (30, 10, 902, 2), (12, 12, 903, 1)]),
             ("fullgc", ref_opt(**fo), wf, False, [(2, None, 910, 0), (3, None, 911, 0)])]
    os.makedirs("data/flickr30k_graph_mask_1000_rm_duplicate", exist_ok=True)
    os.makedirs("data/flickr30k_sg_output_64", exist_ok=True)
    os.makedirs("m/run", exist_ok=True)
    rng = np.random.default_rng(77)
    out, info = {}, {}
    for name, opt, weights, gpn, imgs in cases:
        torch.manual_seed(3)
        model = TopDownModel(opt)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})
        model.eval()
        wh, rerank, grd_output, listing = {}, {}, {}, []
        for k_img, (M, pool, seed, pick) in enumerate(imgs):
            img_id = 5000 + 100 * (0 if gpn else 1) + k_img
            batch = synthetic.make_test_batch(M, D=opt.att_feat_size, seed=seed, fc_size=opt.att_feat_size, node_pool=pool)
            boxes = rng.random((36, 4)) * 500
            wh[img_id] = (int(rng.integers(300, 900)), int(rng.integers(300, 900)))
            rerank[img_id] = [pick, 0, 1]
            oi, am = batch["gpn_obj_ind"][0].numpy(), batch["att_masks"][0].numpy()          # [2, M, N]
            mask_list = [("gt", np.zeros(36, bool))] * 5                                      # the first 5 entries are the GT sub-graphs (:40)
            for half in range(2):
                for j in range(M):
                    nm = np.zeros(36, bool)
                    nm[oi[half, j][am[half, j] > 0]] = True
                    mask_list.append((5 + half * M + j, nm))
            np.savez(f"data/flickr30k_graph_mask_1000_rm_duplicate/{img_id}.npz", feat=np.array({"subgraph_mask_list": mask_list}, dtype=object))
            np.savez(f"data/flickr30k_sg_output_64/{img_id}.npz", feat=np.array({"boxes": boxes}, dtype=object))
            listing.append((img_id, batch, boxes, pick, M, pool, seed))
        np.save("data/flickr30k_img_wh.npy", np.array(wh, dtype=object))
        np.save("m/run/consensus_rerank_ind.npy", np.array(rerank, dtype=object))
        for img_id, batch, boxes, pick, M, pool, seed in listing:
            with torch.no_grad():
                seqq, _, score, keep, att = model(*synthetic.sample_args({k: v.clone() for k, v in batch.items()}),
                                                  opt=dict(sample_max=1, beam_size=1, return_att=1), mode="sample")
            if gpn:                                                                           # eval_utils.py:106-110
                assert len(set(score.tolist())) == score.numel()                              # no ties: stable and unstable sorts agree
                sorted_score, sort_ind = torch.sort(score, descending=True)
                seq, sorted_ind = seqq[sort_ind], keep[sort_ind]
            else:                                                                             # :112-115
                sort_ind = torch.arange(score.size(0)).type_as(keep)
                seq, sorted_ind = seqq, keep
            sents = U.decode_sequence(vocab, seq)
            for consensus in (False, True):
                grd_output = {img_id: []}
                get_grounding_material("m/run/infos.pkl", {"infos": [{"id": img_id}]}, sents, sorted_ind, att, sort_ind, wd_to_lemma,
                                       lemma_det, det_wd, grd_output, use_full_graph=not gpn, grd_sGPN_consensus=consensus)
                r = grd_output[img_id][0]
                tag = f"{name}_{img_id}_{int(consensus)}"
                out[tag + "_bbox"] = np.array(r["bbox"], np.float64).reshape(-1, 4)
                out[tag + "_idx_in_sent"] = np.array(r["idx_in_sent"], np.int64)
                out[tag + "_clss"] = np.array(r["clss"])
            out[f"{name}_{img_id}_sents"] = np.array(sents)
            out[f"{name}_{img_id}_boxes"] = boxes
            out[f"{name}_{img_id}_sorted_ind"] = np_(sorted_ind)
            info.setdefault(name, []).append(dict(id=img_id, M=M, pool=pool, seed=seed, pick=pick, wh=list(wh[img_id]),
                                                  n_sents=len(sents), n_grounded=int(len(r["bbox"]))))
        assert sum(i["n_grounded"] for i in info[name]) > 5, info[name]
    meta["grd"] = dict(kind="grounding", vocab=vocab, wd_to_lemma=wd_to_lemma, lemma_det_id_dict=lemma_det,
                       det_id_to_det_wd={str(k): v for k, v in det_wd.items()}, cases=info,
                       opt={"subgc": {k: v for k, v in vars(cases[0][1]).items() if "path" not in k},
                            "fullgc": {k: v for k, v in vars(cases[1][1]).items() if "path" not in k}},
                       weights={"subgc": "subgc_beam (+0.5 on logit.bias[0])", "fullgc": "fullgc_train"})
    save("grd", out=out)


def restore_cases(meta):
    """models/__init__.py:14-41 `optimistic_restore` run on fabricated checkpoints: (a) a "COCO" checkpoint with a larger vocabulary, one
    unknown key and one key missing, into a smaller-vocabulary network through a word map with kept (-1) rows -> False; (b) a same-shape
    checkpoint through a permuting word map -> True.  Saved: the network's values before, the checkpoint, the word map, the values after."""
    import models as ref_models
    out = {}
    np.save("data/restore_obj_names.npy", np.array(["__background__"] + ["obj%d" % i for i in range(11)]))
    np.save("data/restore_rel_names.npy", np.array(["__background__"] + ["rel%d" % i for i in range(5)]))
    small = dict(rnn_size=16, input_encoding_size=16, att_feat_size=24, gcn_dim=12, fc_feat_size=10, att_hid_size=8, gcn_layers=1,
                 obj_name_path="data/restore_obj_names.npy", rel_name_path="data/restore_rel_names.npy")
    for case, (v_net, v_ckpt, tamper) in dict(a=(30, 50, True), b=(50, 50, False)).items():
        torch.manual_seed(100 + v_net)
        net = ref_models.setup(ref_opt(vocab_size=v_net, **small))
        torch.manual_seed(200 + v_ckpt)
        ck = {k: v.clone() for k, v in ref_models.setup(ref_opt(vocab_size=v_ckpt, **small)).state_dict().items()}
        if tamper:
            ck["not_in_the_network.weight"] = torch.randn(3, 5)
            del ck["ctx2att.bias"]
        rs = np.random.RandomState(7 + v_net)
        wm = rs.randint(0, v_ckpt + 1, size=v_net + 1).astype(np.int64)
        wm[rs.rand(v_net + 1) < 0.3] = -1
        np.save("data/word_mapping.npy", wm)
        before = {k: np_(v) for k, v in net.state_dict().items()}
        ok = ref_models.optimistic_restore(net, ck)
        for k, v in before.items():
            out[f"{case}.before.{k}"] = v
        for k, v in ck.items():
            out[f"{case}.ckpt.{k}"] = np_(v)
        for k, v in net.state_dict().items():                   # only what the restore changed: the rest must equal `before`
            if not np.array_equal(np_(v), before[k]):
                out[f"{case}.after.{k}"] = np_(v)
        out[f"{case}.word_map"] = wm
        out[f"{case}.ok"] = np.array(int(ok))
        o = {k: v for k, v in vars(ref_opt(vocab_size=v_net, **small)).items() if not k.endswith("_name_path")}
        meta[f"restore_{case}"] = dict(opt=dict(o, sg_obj_cnt=12, sg_pred_cnt=6), ckpt_vocab=v_ckpt, returned=bool(ok))
    np.savez_compressed(os.path.join(HERE, "restore_out.npz"), **out)


def main():
    assert os.path.isdir(REF), "golden vectors can only be regenerated where /root/reference exists"
    enter_scratch()
    torch.set_num_threads(1)
    if "--only-eval" in sys.argv:
        with open(os.path.join(HERE, "meta.json")) as f:
            meta = json.load(f)
        eval_cases(meta)
        with open(os.path.join(HERE, "meta.json"), "w") as f:
            json.dump(meta, f, indent=1, sort_keys=True, default=str)
        return
    if "--only-grd" in sys.argv:
        with open(os.path.join(HERE, "meta.json")) as f:
            meta = json.load(f)
        grd_cases(meta)
        with open(os.path.join(HERE, "meta.json"), "w") as f:
            json.dump(meta, f, indent=1, sort_keys=True, default=str)
        return
    if "--only-restore" in sys.argv:
        with open(os.path.join(HERE, "meta.json")) as f:
            meta = json.load(f)
        restore_cases(meta)
        with open(os.path.join(HERE, "meta.json"), "w") as f:
            json.dump(meta, f, indent=1, sort_keys=True, default=str)
        return
    if "--only-loader" in sys.argv:
        with open(os.path.join(HERE, "meta.json")) as f:
            meta = json.load(f)
        loader_cases(meta)
        with open(os.path.join(HERE, "meta.json"), "w") as f:
            json.dump(meta, f, indent=1, sort_keys=True, default=str)
        return
    if "--only-ss" in sys.argv:                                 # add the scheduled-sampling case without rewriting the others
        with open(os.path.join(HERE, "meta.json")) as f:
            meta = json.load(f)
        with np.load(os.path.join(HERE, "subgc_train_weights.npz")) as z:
            w = {k: z[k] for k in z.files}
        ss_case(w, meta)
        with open(os.path.join(HERE, "meta.json"), "w") as f:
            json.dump(meta, f, indent=1, sort_keys=True, default=str)
        return
    if "--only-beam" in sys.argv:                               # add the beam cases without rewriting the others
        with open(os.path.join(HERE, "meta.json")) as f:
            meta = json.load(f)
        with np.load(os.path.join(HERE, "subgc_train_weights.npz")) as z:
            w = {k: z[k] for k in z.files}
        beam_cases(w, meta)
        with open(os.path.join(HERE, "meta.json"), "w") as f:
            json.dump(meta, f, indent=1, sort_keys=True, default=str)
        return
    meta = dict(torch=torch.__version__, numpy=np.__version__, reference="YiwuZhong/Sub-GC @ v1",
                tolerances=dict(fp32_atol=1e-4, fp32_rtol=1e-4, indices="exact"))
    # 1. Sub-GC: train (grads), then decode with the same weights
    _, w = run_train("subgc_train", ref_opt(), seed=1, B=3, gcn_scale=50.0, meta=meta)
    t = dict(test_LSTM=1, gpn_nms_thres=0.75, gpn_max_subg=10)
    run_sample("subgc_greedy", ref_opt(**t), w, seed=2, M=24, sample_opt=dict(sample_max=1, beam_size=1, return_att=1), meta=meta, node_pool=14)
    run_sample("subgc_greedy_nms55", ref_opt(**dict(t, gpn_nms_thres=0.55, gpn_max_subg=1000)), w, seed=3, M=40,
               sample_opt=dict(sample_max=1, beam_size=1), meta=meta, node_pool=10,
               keys=("seq", "seqLogprobs", "subgraph_score", "keep_ind", "subgraph_score_raw", "read_out"))
    run_sample("subgc_sct", ref_opt(**dict(t, sct=1)), w, seed=4, M=6, sample_opt=dict(sample_max=1, beam_size=1), meta=meta)
    run_sample("subgc_topk", ref_opt(**dict(t, use_topk_sampling=1, topk_temp=0.6, the_k=3)), w, seed=5, M=12,
               sample_opt=dict(sample_max=1, beam_size=1), meta=meta)
    # 2. ground-truth sub-graphs (use_gt_subg: no sGPN score / loss)
    run_train("subgc_gtsubg_train", ref_opt(use_gt_subg=1), seed=6, B=2, gcn_scale=50.0, meta=meta)
    # 3. Full-GC baseline: 4 layers, residual every layer, BatchNorm in the GCN, no sGPN
    fo = dict(use_gpn=0, noun_fuse=0, pred_emb_type=2, gcn_layers=4, gcn_residual=1, gcn_bn=1)
    _, wf = run_train("fullgc_train", ref_opt(**fo), seed=7, B=3, gcn_scale=50.0, meta=meta)
    run_sample("fullgc_greedy", ref_opt(**fo), wf, seed=8, M=2, sample_opt=dict(sample_max=1, beam_size=1), meta=meta)
    # 4. beam search / diverse beam search
    beam_cases(w, meta)
    eval_cases(meta)
    with open(os.path.join(HERE, "meta.json"), "w") as f:       # ss_case reads subgc_train's entry back
        json.dump(meta, f, indent=1, sort_keys=True, default=str)
    ss_case(w, meta)
    loader_cases(meta)
    grd_cases(meta)
    restore_cases(meta)
    with open(os.path.join(HERE, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True, default=str)
    tot = sum(os.path.getsize(os.path.join(HERE, f)) for f in os.listdir(HERE) if f.endswith(".npz"))
    print("golden written:", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")), f"{tot/1e6:.2f} MB")


if __name__ == "__main__":
    main()
