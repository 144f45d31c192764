"""compute_dtype = bf16 (BASELINE configs 3 and 5): weights snapshotted to bf16, GEMM-only activations stored bf16 by their
producers, every contraction on subgc_gemm_bf16, fp32 masters / gradients / pointwise math.  Compared with the fp32 CPU
oracle at SURVEY 8(c)'s bf16 tolerance: loss and log-probs atol 5e-2, greedy tokens equal wherever the oracle's top-1 / top-2
margin exceeds 2 x atol; gradients by direction (cosine) and norm, since every product carries 2^-9 relative rounding."""
import argparse

import numpy as np
import pytest
import torch

from oracle import subgc_oracle as O
from subgc import ops, parallel, synthetic
import subgc.models as models
from test_parity_gpu import DEV, KAR, _sharpen, build, close, run_train
from test_shapes_gpu import FLICKR, FLICKR_DATA, FULLGC

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def cosine(a, b):
    a, b = a.detach().double().flatten().cpu(), b.detach().double().flatten().cpu()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


def margin_tokens_equal(outputs, ref_outputs, min_rows):
    got = outputs.argmax(-1).cpu()
    top2 = ref_outputs.topk(2, -1).values
    sure = (top2[..., 0] - top2[..., 1]) > 0.1
    live = ref_outputs.abs().sum(-1) > 0
    assert int((sure & live).sum()) >= min_rows
    assert bool((got[sure & live] == ref_outputs.argmax(-1)[sure & live]).all())


def grads_roughly_equal(m, orc, keys, cos_min=0.995):
    for k in keys:
        g, r = m.P(k).grad, orc.P[k].grad
        assert g.dtype == torch.float32
        c = cosine(g, r)
        assert c > cos_min, (k, c)
        assert abs(float(g.norm()) / float(r.norm()) - 1.0) < 0.05, k


def every_live_gradient_close(m, ref_grads, dead=(), rel=3e-2, cos_min=0.99, norm_tol=0.05, loose=None, where=""):
    """EVERY parameter of the model against the fp32 reference gradient (round-3 review, weak #1a: the key lists above have no bias /
    BatchNorm-affine / alpha_net entries and check direction and norm only -- a wrong scale on ONE small tensor would pass):
      dead parameters (no path to the loss) are exactly zero where the reference has no gradient;
      every other tensor:  max|g - r| <= tol * max|r|  element-wise (catches a mis-scaled or mis-placed slice of a small tensor that
      a cosine over a large one hides), cosine > cos_min, | ||g|| / ||r|| - 1 | < norm_tol.
    A bias whose reference gradient VANISHES in exact arithmetic (a Linear bias in front of a BatchNorm, alpha_net.bias under the
    softmax: the reference holds ~1e-9 of fp32 noise there, own < 1e-3 of the layer's weight-gradient scale) has no scale of its own:
    it must stay noise-sized on its layer's scale, max|g - r| <= 2e-2 * max|r_weight| (measured <= 0.8e-2), and direction / norm are
    not asked of it.
    `loose`: {name fragment: tol} overrides of `rel`, stated per test with the measured value.  All violations are reported at once;
    SUBGC_GRAD_REPORT=<file> appends the full table (committed: profiles/r04_bf16_gradient_table.txt)."""
    import os
    bad, n_checked, table = [], 0, []
    ref = {}
    for k, r in ref_grads.items():
        ref[k] = None if r is None else (torch.from_numpy(r) if isinstance(r, np.ndarray) else r).detach().float().cpu()
    for k, p in m.named_parameters():
        r = ref.get(k)
        g = p.grad
        assert g is not None and g.dtype == torch.float32, k
        g = g.detach().float().cpu()
        if k in dead or r is None:
            if float(g.abs().max()) != 0.0:
                bad.append((k, "dead parameter with a gradient", float(g.abs().max())))
            continue
        own = float(r.abs().max())
        wscale = 0.0
        if k.endswith(".bias") and ref.get(k[:-4] + "weight") is not None:
            wscale = float(ref[k[:-4] + "weight"].abs().max())
        if own < 1e-3 * wscale:                                   # vanishing bias: noise-sized on the layer's scale, nothing else to check
            n_checked += 1
            err = float((g - r).abs().max()) / wscale
            table.append(f"{where}\t{k}\t{tuple(g.shape)}\terr {err:.4f}\ttol 0.02 (vanishing bias: layer scale)\tcos -\tnorm -\town {own:.3e}\tscale {wscale:.3e}")
            if err > 2e-2:
                bad.append((k, f"vanishing bias: max|g-r| / max|r_weight| = {err:.4f} (tol 0.02)"))
            continue
        if own < 1e-12:
            if float(g.abs().max()) > 1e-7:
                bad.append((k, "reference gradient is zero", float(g.abs().max())))
            continue
        n_checked += 1
        tol = rel
        for frag, t in (loose or {}).items():
            if frag in k:
                tol = max(tol, t)
        err = float((g - r).abs().max()) / own
        c = cosine(g, r)
        nr = float(g.norm()) / float(r.norm())
        table.append(f"{where}\t{k}\t{tuple(g.shape)}\terr {err:.4f}\ttol {tol}\tcos {c:.5f}\tnorm {nr:.4f}\town {own:.3e}\tscale {own:.3e}")
        if err > tol or c < cos_min or abs(nr - 1.0) > norm_tol:
            bad.append((k, f"max|g-r|/max|r| = {err:.4f} (tol {tol})", f"cos {c:.5f}", f"norm ratio {nr:.4f}"))
    if os.getenv("SUBGC_GRAD_REPORT"):
        with open(os.getenv("SUBGC_GRAD_REPORT"), "a") as f:
            f.write("\n".join(table) + "\n")
    assert not bad, (where, bad)
    return n_checked


@pytest.mark.parametrize("name", ["subgc_train", "fullgc_train"])
@pytest.mark.parametrize("packed", [True, False])
def test_golden_train_cases_in_bf16_storage(golden, name, packed):
    g = golden(name)
    m = build(g, g.group("weights"), True, compute_dtype="bf16")
    assert m.bf16_storage
    m.packed_decoder = packed
    batch, ref = g.tensors("inputs"), g.group("out")
    out, loss = run_train(m, batch)
    close(out["lang_loss"], ref["lang_loss"], "lang_loss", atol=5e-2, rtol=0)
    grads, dead = g.group("grads"), set(g.meta["dead_params"])
    worst = 1.0
    for k, p in m.named_parameters():
        if k in dead:
            assert float(p.grad.abs().max()) == 0.0, k
        elif float(np.abs(grads[k]).max()) > 1e-6:
            c = cosine(p.grad, torch.from_numpy(grads[k]))
            worst = min(worst, c)
            if k in ("logit.weight", "core.lang_lstm.weight_ih", "core.att_lstm.weight_hh", "embed.0.weight"):
                assert c > 0.995, (k, c)
    # the golden weights are sharpened (GCN x50, LSTM x3, logit x8) to make the fp32 parity tests bite; with one 2^-9 rounding
    # per product that also amplifies the bf16 noise of the small encoder gradients
    assert worst > 0.95, worst
    # every live parameter (biases, BatchNorm affine, alpha_net included) element-wise against the reference's golden gradient; the
    # encoder tensors sit behind the x50 GCN weights, where the bf16 rounding of the 512-wide hidden rows is amplified: 1e-1 there
    # Measured (profiles/r04_bf16_gradient_table.txt): decoder tensors <= 5e-2 of their own scale on the Sub-GC golden, <= 1.1e-1 on
    # the Full-GC one (fc_embed.2.*, behind the x3 LSTM weights); the encoder tensors sit behind the x50 GCN weights, and on Full-GC
    # behind four BatchNorm layers whose mean cancellation leaves the bf16 rounding of the stored hidden rows in single elements
    # (fc_rgt.weight 0.38, bn.bias 0.53 with cos 0.976): bounded at 0.25 (Sub-GC) / 0.7 (Full-GC) with direction and norm on top
    enc = 0.7 if name == "fullgc_train" else 0.25
    n = every_live_gradient_close(m, grads, dead, rel=1.5e-1 if name == "fullgc_train" else 7e-2, cos_min=0.95, norm_tol=0.1,
                                  loose={"gcn_backbone": enc, "obj_": enc, "pred_": enc, "sg_": enc, "gpn_layer": enc}, where=f"{name} packed={packed}")
    assert n >= len(grads) - len(dead) - 2
    with torch.no_grad():
        outputs, gpn_loss, score = m(*synthetic.forward_args({k: v.to(DEV) for k, v in batch.items()}))
    close(outputs, ref["outputs"], "outputs", atol=5e-2, rtol=2e-2)
    assert float((outputs.cpu() - torch.from_numpy(ref["outputs"])).abs().max()) > 1e-5      # bf16 arithmetic really ran
    assert m.weights_b16().dtype == BF and torch.equal(m.W16("logit.weight", m.weights_b16()), m.P("logit.weight").detach().to(BF))


@pytest.mark.timeout(900)
def test_flickr_stress_shape_in_bf16_storage_matches_fp32_oracle():
    """BASELINE config 5 at its stated precision: N = 101 nodes, K = 301 relations, D = 4096, L = 2048, V + 1 = 7001 (rows that are
    NOT multiples of 8 elements: padded leading dimensions and the K-tail mask of the GEMM), return_att decode."""
    torch.manual_seed(5)
    opt = argparse.Namespace(**dict(FLICKR, compute_dtype="bf16"))
    m = models.setup(opt)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    _sharpen(sd, None)
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    batch = synthetic.make_train_batch(8, vocab=7000, seed=6, **FLICKR_DATA)
    out, loss = run_train(m, batch)
    orc = O.Oracle(argparse.Namespace(**FLICKR), sd, requires_grad=True); orc.training = True
    ref = O.loss_wrapper(orc, batch)
    (ref["lang_loss"] + ref["gpn_loss"]).backward()
    close(out["lang_loss"], ref["lang_loss"], "lang_loss", atol=5e-2, rtol=0)
    close(out["gpn_loss"], ref["gpn_loss"], "gpn_loss", atol=5e-2, rtol=0)
    grads_roughly_equal(m, orc, ("logit.weight", "core.att_lstm.weight_ih", "core.lang_lstm.weight_hh", "embed.0.weight", "obj_v_proj.weight",
                                 "obj_emb_proj.weight", "gcn_backbone.gcn.0.gcn_collect.collect_units.3.fc_rgt.weight",
                                 "gcn_backbone.gcn.1.gcn_collect.collect_units.0.fc_lft.weight", "gpn_layer.gpn_fc.0.weight",
                                 "att_embed.0.weight", "ctx2att.weight", "fc_embed.0.weight", "core.attention.h2att.weight"))
    with torch.no_grad():
        outputs, _, score = m(*synthetic.forward_args({k: v.to(DEV) for k, v in batch.items()}))
    assert tuple(outputs.shape) == (40, 17, 7001)
    close(outputs, ref["outputs"].detach(), "outputs", atol=5e-2, rtol=2e-2)
    close(score, ref["subgraph_score"].detach(), "score", atol=2e-2)
    margin_tokens_equal(outputs, ref["outputs"].detach(), 100)
    # the attention-grounding output path (return_att; misc/grd_utils.py:44-47 takes the arg-max of these weights) decoded IN bf16: the
    # <= 16-row step streams the bf16 weight snapshot (functions.DecodeState: both LSTM matrices, h2att, logit) with fp32 accumulation.
    # Compared with the fp32 oracle along the SAME token path (oracle teacher-forced with the tokens the bf16 decode produced): log-probs
    # atol 5e-2, attention weights atol 2e-2, attention arg-max equal wherever the oracle's top-1 / top-2 margin exceeds 2 x atol; where the
    # free-running oracle picks another word, its margin over the bf16 word must be below 2 x atol.
    tb = synthetic.make_test_batch(5, seed=7, **FLICKR_DATA)
    sopt = dict(sample_max=1, beam_size=1, return_att=1)
    topt = argparse.Namespace(**dict(FLICKR, test_LSTM=1, sct=1, compute_dtype="bf16"))
    sd = {k: v.clone() for k, v in sd.items()}
    sd["core.attention.alpha_net.weight"] *= 10.0                  # peaky attention: the arg-max the grounding code reads has margins to test
    mt = models.setup(topt); mt.load_state_dict(sd); mt = mt.to(DEV).eval()
    ret = mt(*synthetic.sample_args({k: v.to(DEV) for k, v in tb.items()}), opt=sopt, mode="sample")
    loops = [g for g in mt._graph_cache.values() if hasattr(g, "st")]
    assert loops and loops[0].st.w16 is not None and loops[0].st.Wc1.dtype == BF           # the bf16 weight stream really ran
    orc_t = O.Oracle(argparse.Namespace(**dict(FLICKR, test_LSTM=1, sct=1)), sd)
    free = orc_t.sample(*synthetic.sample_args(tb), opt=sopt)
    forced = orc_t.sample(*synthetic.sample_args(tb), opt=sopt, forced=ret[0].cpu())
    seq_b = ret[0].cpu()
    assert seq_b.shape == (10, mt.seq_length) and int((seq_b > 0).sum()) > 40
    close(ret[1], forced[1], "seqLogprobs", atol=5e-2, rtol=0)
    assert tuple(ret[4].shape) == tuple(forced[4].shape)
    close(ret[4], forced[4], "att2_weights", atol=2e-2, rtol=0)
    top2 = forced[4].topk(2, -1).values
    sure = (top2[..., 0] - top2[..., 1]) > 4e-2
    assert int(sure.sum()) > 20 and bool((ret[4].cpu().argmax(-1)[sure] == forced[4].argmax(-1)[sure]).all())
    same = 0
    for r in range(seq_b.size(0)):
        diff = (free[0][r] != seq_b[r]).nonzero()
        if diff.numel() == 0:
            same += 1
            continue
        t0 = int(diff[0])                                          # same history up to t0: both log-probs come from the same oracle state
        assert float(free[1][r, t0] - forced[1][r, t0]) < 0.1, (r, t0)
    assert same >= 5
    close(ret[2], free[2], "score", atol=2e-2)


@pytest.mark.timeout(900)
def test_flickr_bench_size_b64_train_matches_fp32_oracle():
    """BASELINE config 5 at the size the bench line quotes (64 images = 320 sentences, N = 101, K = 301, D = 4096, L = 2048): the bf16
    storage step (bf16 unit outputs through the aggregation, raw-logit criterion, query planes, split-K forms of this batch size) against
    the fp32 CPU oracle -- losses at atol 5e-2, gradients by direction and norm."""
    torch.manual_seed(9)
    opt = argparse.Namespace(**dict(FLICKR, compute_dtype="bf16"))
    m = models.setup(opt)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    _sharpen(sd, None)
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    batch = synthetic.make_train_batch(64, vocab=7000, seed=1000, **FLICKR_DATA)
    out, loss = run_train(m, batch)
    orc = O.Oracle(argparse.Namespace(**FLICKR), sd, requires_grad=True); orc.training = True
    ref = O.loss_wrapper(orc, batch)
    (ref["lang_loss"] + ref["gpn_loss"]).backward()
    close(out["lang_loss"], ref["lang_loss"], "lang_loss", atol=5e-2, rtol=0)
    close(out["gpn_loss"], ref["gpn_loss"], "gpn_loss", atol=5e-2, rtol=0)
    grads_roughly_equal(m, orc, ("logit.weight", "core.att_lstm.weight_ih", "core.lang_lstm.weight_hh", "embed.0.weight", "obj_v_proj.weight",
                                 "gcn_backbone.gcn.0.gcn_collect.collect_units.3.fc_rgt.weight", "gcn_backbone.gcn.1.gcn_collect.collect_units.0.fc_lft.weight",
                                 "att_embed.0.weight", "ctx2att.weight", "fc_embed.0.weight", "core.attention.h2att.weight"))
    ref_g = {k: p.grad for k, p in orc.P.items()}
    # 3e-2 of each tensor's own scale everywhere except the class-embedding path (sg_obj_embed -> obj_emb_proj, 300-d, fp32-operand
    # products): a class row sums the d(x) of the few node rows of that class, each carrying the bf16 rounding of every GCN path that
    # reaches it, with no batch-sized sum to average it out -- measured 7.6e-2 / 6.5e-2, bounded at 1e-1
    every_live_gradient_close(m, ref_g, {k for k, v in ref_g.items() if v is None}, rel=3e-2, cos_min=0.99,
                              loose={"sg_obj_embed": 1e-1, "sg_pred_embed": 1e-1, "obj_emb_proj": 1e-1, "pred_emb_prj": 1e-1}, where="flickr B=64")


@pytest.mark.timeout(900)
def test_full_gc_kar_bench_size_b256_train_matches_fp32_oracle():
    """BASELINE config 3 at the size the bench line quotes (256 images = 1280 sentences, 4 GCN layers with BatchNorm over 16 640 relation /
    9 472 node rows, attention sets shared per image): the bf16 storage train step against the fp32 CPU oracle in TRAIN mode (batch
    statistics over the whole batch) -- loss at atol 5e-2, gradients by direction and norm, running statistics."""
    torch.manual_seed(4)
    opt = argparse.Namespace(**dict(FULLGC, compute_dtype="bf16"))
    m = models.setup(opt)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    _sharpen(sd, None)
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    batch = synthetic.make_train_batch(256, seed=1000)
    out, loss = run_train(m, batch)
    assert out["gpn_loss"] is None
    orc = O.Oracle(argparse.Namespace(**FULLGC), sd, requires_grad=True); orc.training = True
    ref = O.loss_wrapper(orc, batch)
    ref["lang_loss"].backward()
    close(out["lang_loss"], ref["lang_loss"], "lang_loss", atol=5e-2, rtol=0)
    grads_roughly_equal(m, orc, ("logit.weight", "core.att_lstm.weight_ih", "core.lang_lstm.weight_hh", "embed.0.weight", "obj_v_proj.weight",
                                 "gcn_backbone.gcn.0.gcn_collect.collect_units.0.fc_rgt.weight", "gcn_backbone.gcn.2.gcn_collect.collect_units.3.fc_lft.weight",
                                 "gcn_backbone.gcn.3.gcn_collect.collect_units.1.fc_lft.weight", "att_embed.0.weight", "ctx2att.weight", "fc_embed.0.weight",
                                 "core.attention.h2att.weight"), cos_min=0.99)
    ref_g = {k: p.grad for k, p in orc.P.items()}
    # Full-GC: every unit output passes a BatchNorm, so d(y) has zero column means and a weight-gradient element sum_r dy[r,i] h[r,j] is
    # what is LEFT after the mean of h[:, j] cancels -- the bf16 rounding of the stored hidden rows (relative to their mean) does not
    # cancel, which puts single elements of the GCN / class-embedding gradients up to a few 1e-1 of the tensor's scale off while direction
    # (cos > 0.98) and norm (5 %) hold (measured: fc_rgt.weight <= 0.42 at cos 0.9975, bn.bias 0.27; profiles/r04_bf16_gradient_table.txt);
    # the decoder tensors stay within 3.4e-2 (bound 5e-2), the class-embedding path within 6.1e-2 (bound 1.2e-1)
    every_live_gradient_close(m, ref_g, {k for k, v in ref_g.items() if v is None}, rel=5e-2, cos_min=0.98,
                              loose={"gcn_backbone": 6e-1, "sg_pred_embed": 1.2e-1, "pred_emb_prj": 1.2e-1}, where="full_gc_kar B=256")
    k = "gcn_backbone.gcn.1.gcn_collect.collect_units.2.bn.running_mean"
    close(m.state_dict()[k], orc.buffers[k], "running_mean", atol=2e-2, rtol=2e-2)


@pytest.mark.timeout(900)
def test_full_gc_kar_batch_256_properties_in_bf16_storage():
    """BASELINE config 3 at its stated size and precision (Full_GC_Kar, 256 images = 1280 sentences, bf16): size-independent
    properties -- log-probs normalise, padded steps are zero, the loss is finite and equals the packed path's, dead parameters get
    no gradient, BatchNorm statistics move, and one fused optimizer step leaves the bf16 snapshot equal to the rounded masters."""
    torch.manual_seed(1)
    opt = argparse.Namespace(**dict(FULLGC, drop_prob_lm=0.5, compute_dtype="bf16"))
    m = models.setup(opt).to(DEV).train()
    batch = synthetic.make_train_batch(256, seed=0)
    adam = parallel.FlatAdam(m)
    out, loss = run_train(m, batch)                                                  # packed decoder
    assert torch.isfinite(loss) and out["gpn_loss"] is None
    for k in ("gcn_backbone.gcn.3.gcn_collect.collect_units.2.fc_lft.weight",):      # Full-GC: only the last layer's units 2, 3 are dead
        assert float(m.P(k).grad.abs().max()) == 0.0
    assert float(m.P("logit.weight").grad.abs().max()) > 0 and float(m.P("gcn_backbone.gcn.0.gcn_collect.collect_units.0.fc_lft.weight").grad.abs().max()) > 0
    assert float(m.state_dict()["gcn_backbone.gcn.0.gcn_collect.collect_units.0.bn.running_mean"].abs().max()) > 0
    adam.step()
    torch.cuda.synchronize()
    snap = m.weights_b16()
    assert torch.equal(snap, m.flat_params.to(BF))                                   # written by the Adam sweep itself, no re-cast needed
    m.eval()
    with torch.no_grad():
        b = {k: v.to(DEV) for k, v in batch.items()}
        outputs, _, _ = m(*synthetic.forward_args(b))
    assert tuple(outputs.shape) == (1280, 17, 9488)
    lse = torch.logsumexp(outputs, -1)
    live = outputs.abs().sum(-1) > 0
    assert float(lse[live].abs().max()) < 1e-3 and int(live.sum()) > 1280 * 5
    assert (~live).sum() == 0 or float(outputs[~live].abs().max()) == 0.0
    # ... and against the fp32 oracle: in eval mode no row depends on another image (BatchNorm uses its running statistics), so the
    # oracle runs on the first 13 images alone (65 sentences) and must agree with the first 65 rows of the 256-image bf16 forward
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    sub = {k: v[:13 * (v.size(0) // 256)].clone() for k, v in batch.items()}
    orc = O.Oracle(argparse.Namespace(**dict(FULLGC, drop_prob_lm=0.5)), sd)
    orc.training = False
    with torch.no_grad():
        ref = O.loss_wrapper(orc, sub)
    got = outputs[:65].cpu()
    reached = ref["outputs"].abs().sum(-1) > 0                    # the oracle's early break is that of ITS 65 sentences
    close(got[reached], ref["outputs"][reached], "outputs[:65]", atol=5e-2, rtol=2e-2)
    margin_tokens_equal(got, ref["outputs"], 200)
    lab, msk = sub["labels"][:, 1:], sub["masks"][:, 1:]
    my_loss = -(got.gather(2, lab[:, :17].unsqueeze(2)).squeeze(2) * msk[:, :17] * reached).sum() / msk[:, :17].sum()
    close(my_loss, ref["lang_loss"], "lang_loss of the first 65 sentences", atol=5e-2, rtol=0)


def test_bf16_snapshot_follows_the_masters(golden):
    g = golden("subgc_train")
    m = build(g, g.group("weights"), True, compute_dtype="bf16")
    s0 = m.weights_b16().clone()
    assert m.weights_b16().data_ptr() == m.weights_b16().data_ptr()                  # cached while the masters stand
    with torch.no_grad():
        m.P("logit.weight").mul_(0.5)
    assert torch.equal(m.W16("logit.weight", m.weights_b16()), m.P("logit.weight").detach().to(BF))
    assert not torch.equal(m.weights_b16(), s0)
    assert m.W16("obj_emb_proj.weight", m.weights_b16()) is None                     # 20-column rows: no 16-byte twin, fp32-operand GEMM


@pytest.mark.timeout(900)
@pytest.mark.parametrize("shape", ["golden_dims", "full_width_b16"])
def test_bf16_training_trajectory_follows_the_fp32_hip_path(golden, shape):
    """Convergence evidence behind the loose element-wise bf16 bounds on the GCN gradients (round-5 review): 60 optimisation steps of
    Full-GC (4 GCN layers with BatchNorm, dropout off, fused clip + Adam, lr 5e-4 as train.sh) under compute_dtype = bf16 against the
    fp32 HIP path from the SAME initial weights on the same fixed batch (reference: train.py:151-166, graph_conv_unit.py:31-32).
    The fp32 path is itself not bit-reproducible (embed_bwd / pool_bwd accumulate with float atomics) and 60 Adam steps amplify that:
    on the golden dims (weights sharpened x50 / x3 / x8 for the parity tests) two fp32 runs drift 1-6 % apart after step 50 while the
    bf16 path (no atomics on its large tensors' critical path) repeats exactly.  So the fp32 path runs TWICE and its own spread is the
    yardstick: at every step  |bf16 - fp32| / fp32 < 2 % + 2 x the fp32 runs' own relative spread (and < 6 % from step 45 on);  measured (3 jobs): full width
    0.6-1.2 % at every step with an fp32 spread < 0.5 %, golden dims < 1.3 % up to step 50 and 1.5-6 % after with the fp32 spread at 1-6 %.
    The BatchNorm running statistics the runs end with are compared per layer in units of the layer's own scale (RMS column standard
    deviation for the means, mean variance for the variances) with the same yardstick."""
    import os
    steps = 60
    if shape == "golden_dims":
        g = golden("fullgc_train")
        make = lambda dt: build(g, g.group("weights"), True, compute_dtype=dt)
        batch = g.tensors("inputs")
    else:
        def make(dt):
            torch.manual_seed(31)
            return models.setup(argparse.Namespace(**dict(FULLGC, drop_prob_lm=0.0, compute_dtype=dt))).to(DEV).train()
        batch = synthetic.make_train_batch(16, seed=32)
    b = {k: v.to(DEV) for k, v in batch.items()}
    curves, stats, evals = {}, {}, {}
    init = None
    for run, dt in (("fp32", "fp32"), ("fp32_again", "fp32"), ("bf16", "bf16")):
        m = make(dt)
        if init is None:
            init = {k: v.clone() for k, v in m.state_dict().items()}
        else:
            m.load_state_dict(init)                              # the same start, whatever the constructor drew
        lw = models.LossWrapper(m, None)
        adam = parallel.FlatAdam(m, lr=5e-4)
        losses = []
        for _ in range(steps):
            m.flatten_grads()
            out = lw(b["fc_feats"], b["att_feats"], b["labels"], b["masks"], b["att_masks"], None, None, None, b["obj_dist"], None, b["rel_ind"],
                     None, b["pred_dist"], b["gpn_obj_ind"], b["gpn_pred_ind"], b["gpn_nrel_ind"], b["gpn_pool_mtx"])
            out["lang_loss"].backward()
            adam.step()
            losses.append(float(out["lang_loss"].detach()))
        curves[run] = np.array(losses)
        stats[run] = {k: v.detach().float().cpu() for k, v in m.state_dict().items() if "running_" in k}
        assert m.bf16_storage == (dt == "bf16")
        m.eval()                                                  # the running statistics' only consumer: the same batch in eval mode
        with torch.no_grad():
            ev = lw(b["fc_feats"], b["att_feats"], b["labels"], b["masks"], b["att_masks"], None, None, None, b["obj_dist"], None, b["rel_ind"],
                    None, b["pred_dist"], b["gpn_obj_ind"], b["gpn_pred_ind"], b["gpn_nrel_ind"], b["gpn_pool_mtx"])
        evals[run] = float(ev["lang_loss"])
    f, f2, h = curves["fp32"], curves["fp32_again"], curves["bf16"]
    assert np.isfinite(f).all() and np.isfinite(h).all()
    assert f[-1] < 0.8 * f[0] and h[-1] < 0.8 * h[0], (f[:3], f[-3:], h[:3], h[-3:])
    rel, own = np.abs(h - f) / np.abs(f), np.abs(f2 - f) / np.abs(f)

    def stat_dev(a, ref):
        """(worst mean deviation / layer RMS std, worst variance deviation / layer mean variance) over the BatchNorm layers"""
        wm = wv = 0.0
        for k, want in ref.items():
            if not k.endswith("running_mean"):
                continue
            kv = k[:-len("running_mean")] + "running_var"
            mv = float(ref[kv].mean())
            wm = max(wm, float((a[k] - want).abs().max()) / max(mv, 1e-12) ** 0.5)
            wv = max(wv, float((a[kv] - ref[kv]).abs().max()) / max(mv, 1e-12))
        return wm, wv

    (bm, bv), (om, ov) = stat_dev(stats["bf16"], stats["fp32"]), stat_dev(stats["fp32_again"], stats["fp32"])
    if os.environ.get("SUBGC_TRAJ_REPORT"):
        with open(os.environ["SUBGC_TRAJ_REPORT"], "a") as fh:
            fh.write(f"{shape}: fp32 {f[0]:.4f} -> {f[-1]:.4f} (again {f2[-1]:.4f}), bf16 {h[0]:.4f} -> {h[-1]:.4f}; bf16 vs fp32 max rel dev {rel.max():.4f} at step "
                     f"{int(rel.argmax())}, per decade {[round(float(rel[i:i + 10].max()), 4) for i in range(0, steps, 10)]}; fp32 vs fp32 per decade "
                     f"{[round(float(own[i:i + 10].max()), 4) for i in range(0, steps, 10)]}; running stats (mean / layer std, var / layer var): "
                     f"bf16 {bm:.4f} {bv:.4f}, fp32 again {om:.4f} {ov:.4f}; eval-mode loss fp32 {evals['fp32']:.4f} / {evals['fp32_again']:.4f}, bf16 {evals['bf16']:.4f}\n")
    # two fp32 samples do not pin the width of the fp32 path's own distribution late in the run (12 fp32 runs on the golden dims ended
    # between 2.245 and 2.330, +-2 % around their mean, the bf16 run at 2.2368 every time): from step 45 on the bound is at least 6 %
    floor = np.where(np.arange(steps) >= 45, 6e-2, 0.0)
    bound = np.maximum(2e-2 + 2.0 * np.maximum.accumulate(own), floor)
    worst = int((rel - bound).argmax())
    assert (rel < bound).all(), (worst, float(rel[worst]), float(bound[worst]))
    assert len(stats["fp32"]) >= 8                                # 4 layers x 4 units x (mean, var)
    # Running statistics.  What they are FOR is the eval-mode forward: its loss on the same batch agrees like the training losses do.
    # Two fp32 samples do not pin the width of the fp32 path's OWN eval-mode distribution either (20 fp32 runs each, round 6: golden dims
    # 2.220 ... 2.329 with the bf16 run at 2.2372 every time; full width 5.467 ... 6.107 -- the near-constant columns of the first layers
    # make the fp32 running statistics themselves irreproducible -- with the bf16 run at 5.561 ... 5.565): the bound is at least the
    # measured half-width of that distribution around the bf16 value, 6 % / 15 %.
    e_rel, e_own = abs(evals["bf16"] - evals["fp32"]) / evals["fp32"], abs(evals["fp32_again"] - evals["fp32"]) / evals["fp32"]
    e_floor = 6e-2 if shape == "golden_dims" else 15e-2
    assert e_rel < max(2e-2 + 2.0 * max(e_own, float(own.max())), e_floor), (evals, e_rel, e_own)
    # Element-wise they are a soft spot of bf16 storage and the test says how soft: a unit output whose column mean exceeds its spread is
    # stored with 2^-9 |mean| of rounding per element, so the one-pass statistics of the bf16 rows move by a fraction of the layer's
    # standard deviation -- measured 1.6 layer-std on the golden dims (x50 GCN weights; fp32 against itself: 0.07-0.19) -- while at full
    # width the near-constant columns of the first layers make even two fp32 runs differ by 20-30 layer-std, bf16 no more than they do.
    # (ten jobs, round 6, full width: bf16 20.9-29.5 layer-std / 7.5-19.6 layer-var, fp32 against itself 16.0-34.2 / 4.4-24.1 -- one
    # fp32 pair is a noisy yardstick there, so the bound is at least twice the largest fp32 self-deviation seen; golden dims: bf16
    # 1.57-1.61 / 0.93-0.95 every time, fp32 against itself 0.04-0.22 / 0.03-0.11)
    wide_m, wide_v = (0.0, 0.0) if shape == "golden_dims" else (70.0, 50.0)
    assert bm < max(2.5, 2.0 * om, wide_m) and bv < max(1.5, 2.0 * ov, wide_v), (bm, bv, om, ov)
