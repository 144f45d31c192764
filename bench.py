#!/usr/bin/env python3
"""bench.py -- Sub-GC hot path on MI355X: train fwd+bwd images/s (Sub_GC_Kar, BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One rank per GPU.  A step = one pass of the hot path over one synthetic batch that is already
resident in HBM: LossWrapper forward (fusion -> GCN -> sGPN -> attention-LSTM decoder -> NLL),
backward, and for N > 1 the RCCL all-reduce of the flat gradient bucket.  Per-GPU batch is fixed
(weak scaling); `value` = images of all ranks / max-over-ranks wall time.  Rank 0 prints ONE JSON line.

`roofline`: the dominant kernel is the fp32 MFMA GEMM (gemm_f32_kernel): every launch inside the
timed region is bracketed by HIP events on its own stream (C-ABI profiling hook), so
achieved = exact algorithmic GEMM FLOPs of the K steps / summed GEMM kernel time, against the
157.3 TFLOP/s fp32 matrix peak.  `cpu_baseline`: the CPU oracle (op-for-op restatement of the
reference, `kind: "port"`) timed on this box's host cores on a bounded sample, rank 0 at N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "sub-gc_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from subgc import _lib, ops, parallel, synthetic  # noqa: E402
import subgc.models as models  # noqa: E402

KAR = dict(caption_model="topdown", vocab_size=9487, input_encoding_size=1000, rnn_size=1000, num_layers=1, drop_prob_lm=0.5,
           max_length=20, seq_length=16, fc_feat_size=2048, att_feat_size=2048, att_hid_size=512, use_bn=0, sampling_prob=0.0,
           use_gpn=1, embed_dim=300, gcn_dim=1024, noun_fuse=1, pred_emb_type=1, gcn_layers=2, gcn_residual=2, gcn_bn=0,
           obj_name_path=None, rel_name_path=None)
MFMA_F32_PEAK_TFLOPS = 157.3            # MI355X_MICROARCH.md: fp32 matrix peak (v_mfma_f32_32x32x2_f32)
MFMA_BF16_PEAK_TFLOPS = 2500.0          # MI355X_MICROARCH.md: dense bf16 matrix peak (v_mfma_f32_32x32x16_bf16); never the 2:1-sparse 5 PF
HBM_PEAK_GBPS = 8000.0
MODEL_GFLOP_PER_IMAGE = 22.0            # live-graph fwd 7.32 GFLOP x 3 (fwd+bwd)
FULLGC = dict(KAR, use_gpn=0, noun_fuse=0, pred_emb_type=2, gcn_layers=4, gcn_residual=1, gcn_bn=1, compute_dtype="bf16")
FLICKR = dict(KAR, vocab_size=7000, fc_feat_size=4096, att_feat_size=4096, gcn_dim=2048, compute_dtype="bf16")
# --config: the headline workload (BASELINE.json configs[1]) and the two bf16 parity configs as their own contract lines
CONFIGS = {
    "kar": dict(opt=KAR, batch=128, data={}, dtype="f32", peak=MFMA_F32_PEAK_TFLOPS, gflop_img=MODEL_GFLOP_PER_IMAGE,
                kernel="gemm_f32_kernel (v_mfma_f32_32x32x2_f32)", metric="images/sec (training fwd+bwd), Sub_GC_Kar",
                workload="Sub_GC_Kar train fwd+bwd (BASELINE.json configs[1]): 128 images/GPU, 36+1 nodes, 64+1 relations, "
                         "2048-d region feats, 5 sentences/image, 2 pos + 2 neg sub-graphs/sentence, T=17, V+1=9488, dropout on"),
    # the workload the reference trains for 30 of its 35 epochs (train.sh:10,21,31: --scheduled_sampling_start 0; train.py:126-132 raises
    # ss_prob by 0.05 every 5 epochs up to 0.25; AttModel.py:158-167): the headline step with a quarter of the input words sampled
    "kar_ss25": dict(opt=dict(KAR, sampling_prob=0.25), batch=128, data={}, dtype="f32", peak=MFMA_F32_PEAK_TFLOPS, gflop_img=MODEL_GFLOP_PER_IMAGE,
                     kernel="gemm_f32_kernel (v_mfma_f32_32x32x2_f32)", metric="images/sec (training fwd+bwd), Sub_GC_Kar, scheduled sampling 0.25",
                     workload="Sub_GC_Kar train fwd+bwd as the headline, with scheduled sampling at its final probability ss_prob = 0.25 "
                              "(train.sh:10, train.py:126-132, AttModel.py:158-167): per-step logits, multinomial draws, per-step embedding and x->gates"),
    "full_gc_kar": dict(opt=FULLGC, batch=256, data={}, dtype="bf16", peak=MFMA_BF16_PEAK_TFLOPS, gflop_img=3 * 9.01,
                        kernel="gemm_bf16_kernel (v_mfma_f32_32x32x16_bf16, bf16-stored operands)",
                        metric="images/sec (training fwd+bwd), Full_GC_Kar bf16",
                        workload="Full_GC_Kar train fwd+bwd (BASELINE.json configs[2]): 256 images/GPU, full-graph 4-layer GCN with BatchNorm, "
                                 "no sGPN, attention over all 36 nodes, bf16 compute / fp32 masters, T=17, V+1=9488, dropout on"),
    "flickr": dict(opt=FLICKR, batch=64, data=dict(N=101, K=301, D=4096, n_edges=300, max_nodes=30, vocab=7000), dtype="bf16",
                   peak=MFMA_BF16_PEAK_TFLOPS, gflop_img=3 * 12.1, kernel="gemm_bf16_kernel (v_mfma_f32_32x32x16_bf16, bf16-stored operands)",
                   metric="images/sec (training fwd+bwd), Sub_GC_Flickr stress bf16",
                   workload="Flickr stress shape train fwd+bwd (BASELINE.json configs[4]): 64 images/GPU, 100+1 nodes, 300+1 relations, 4096-d "
                            "feats, gcn_dim 2048, V+1=7001, bf16 compute / fp32 masters, dropout on"),
}


def lw_args(b):
    return (b["fc_feats"], b["att_feats"], b["labels"], b["masks"], b["att_masks"], None, None, None, b["obj_dist"], None,
            b["rel_ind"], None, b["pred_dist"], b["gpn_obj_ind"], b["gpn_pred_ind"], b["gpn_nrel_ind"], b["gpn_pool_mtx"])


def effective_cores():
    """Cores this process may really use: min(affinity, cgroup CPU quota) -- os.cpu_count() alone
    reports the whole host (256 on the GPU box) while the container is capped at 16."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def cpu_decode_baseline(M, images=2):
    """The oracle's greedy decode as the reference runs it (one image per call, test.py:184-185; NMS 0.75 keeping <= 10 of the 2M
    candidate sub-graphs, 20 tokens each; the python-set NMS of gpn.py:101-150 included): tokens/s on the host cores."""
    from oracle import subgc_oracle as O
    cores = effective_cores()
    torch.set_num_threads(cores)
    opt = argparse.Namespace(**dict(KAR, test_LSTM=1, gpn_nms_thres=0.75, gpn_max_subg=10))
    torch.manual_seed(0)
    sd = {k: v.clone() for k, v in models.setup(opt).state_dict().items()}
    orc = O.Oracle(opt, sd)
    batches = [synthetic.make_test_batch(M, seed=500 + i) for i in range(images)]
    sopt = dict(sample_max=1, beam_size=1)
    tokens, t0 = 0, time.perf_counter()
    for b in batches:
        seq = orc.sample(*synthetic.sample_args(b), opt=sopt)[0]
        tokens += seq.size(0) * seq.size(1)
    dt = time.perf_counter() - t0
    return {"value": round(tokens / dt, 1), "unit": "tokens/s", "cores": cores, "kind": "port", "s_per_image": round(dt / images, 3),
            "sample": f"oracle greedy decode, {images} images looped one per call, {2 * M} candidate sub-graphs each -> NMS 0.75 -> <= 10 kept x 20 tokens"}


def cpu_train_baseline(images, iters):
    """The oracle (CPU restatement of the reference) on the host cores: fwd+bwd images/s."""
    from oracle import subgc_oracle as O
    cores = effective_cores()
    torch.set_num_threads(cores)
    opt = argparse.Namespace(**KAR)
    torch.manual_seed(0)
    sd = {k: v.clone() for k, v in models.setup(opt).state_dict().items()}
    orc = O.Oracle(opt, sd, requires_grad=True)
    orc.training = True
    batch = synthetic.make_train_batch(images, seed=77)
    times = []
    for i in range(iters + 1):
        for p in orc.P.values():
            p.grad = None
        t0 = time.perf_counter()
        out = O.loss_wrapper(orc, batch)
        (out["lang_loss"] + out["gpn_loss"]).backward()
        times.append(time.perf_counter() - t0)
    t = sorted(times[1:])[len(times[1:]) // 2]
    return {"value": round(images / t, 3), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"oracle (PyTorch-CPU restatement of the reference path), Sub_GC_Kar fp32 train fwd+bwd, B={images} images, "
                      f"median of {iters} timed iterations after 1 warm-up, torch threads={cores}"}


def decode_bench(model_sd, dev, images, M):
    """Greedy decode (reference test path: one image per call, NMS 0.75, keep 10, 20 tokens)."""
    opt = argparse.Namespace(**dict(KAR, test_LSTM=1, gpn_nms_thres=0.75, gpn_max_subg=10))
    m = models.setup(opt)
    m.load_state_dict(model_sd)
    m = m.to(dev).eval()
    batches = [{k: v.to(dev) for k, v in synthetic.make_test_batch(M, seed=500 + i).items()} for i in range(images)]
    sopt = dict(sample_max=1, beam_size=1)
    # untimed pass over a DISJOINT set of images (other seeds, same shape): the token loop is captured once per surviving-row count
    # (hipGraph) and the allocator settles, but neither the Infinity Cache nor any input-keyed state has seen the timed images
    warm = [{k: v.to(dev) for k, v in synthetic.make_test_batch(M, seed=9000 + i).items()} for i in range(min(images, 256))]
    for b in warm[:min(images, 64)]:
        m(*synthetic.sample_args(b), opt=sopt, mode="sample")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tokens = 0
    for b in batches:
        seq = m(*synthetic.sample_args(b), opt=sopt, mode="sample")[0]
        tokens += seq.size(0) * seq.size(1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"decode_tokens_per_s": round(tokens / dt, 1), "decode_ms_per_image": round(1e3 * dt / images, 3),
           "decode_config": f"greedy, {2 * M} candidate sub-graphs/image -> NMS 0.75 -> <=10 kept x 20 tokens, {images} images looped"}
    # decode roofline: the token loop streams the decoder's weights once per step (weight-streaming GEMMs, M <= 16 rows).  The
    # captured loop of the 10-row case is replayed alone and timed with events on its stream: bytes streamed per step / time per step.
    loops = [g for k, g in getattr(m, "_graph_cache", {}).items() if hasattr(g, "st") and g.n == 10]
    if loops:
        g = loops[0]
        R, A, V1 = m.rnn_size, m.att_hid_size, m.vocab_size + 1
        steps = g.T + 1
        bytes_step = 4.0 * (4 * R * 2 * R + 4 * R * 3 * R + A * R + V1 * R) + 4.0 * g.n * 4 * R      # both LSTM matrices, h2att, logit + the x->gates rows
        for _ in range(3):
            g.graph.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            g.graph.replay()
        e1.record()
        torch.cuda.synchronize()
        us_step = 1e3 * e0.elapsed_time(e1) / 20 / steps
        gbps = bytes_step / (us_step * 1e-6) / 1e9
        traffic, pmc_name = None, None
        for rnd in ("r06", "r05", "r04", "r03", "r02"):         # committed PMC passes over the same loop (tools/pmc_decode.sh)
            pmc_path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_decode.json")
            if not os.path.exists(pmc_path):
                continue
            with open(pmc_path) as f:
                kk = json.load(f)["kernels"]
            per = lambda k_: kk[k_]["hbm_fetch_bytes_per_launch"] + kk[k_]["hbm_write_bytes_per_launch"]
            if "dual_pick" in kk and getattr(m, "decode_fused_pick", True):    # round-6 loop: [logits | att gates], cell + pick, [h2att | lang hidden part], lang ctx part
                traffic, pmc_name = per("dual_pick") + per("cell_pick") + per("dual") + per("lstm"), f"{rnd}_pmc_decode.json"
                break
            if not getattr(m, "decode_fused_pick", True):
                traffic, pmc_name = 2 * per("lstm") + 2 * per("plain_1tile"), f"{rnd}_pmc_decode.json"
                break
        out["decode_roofline"] = {"bound": "hbm", "kernel": "gemm_skinny_mfma_kernel (weight streaming, M <= 16 rows)", "achieved": round(gbps, 1),
                                  "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(gbps / HBM_PEAK_GBPS, 4), "traffic": traffic,
                                  "traffic_unit": (f"bytes fetched + written past L2 per token step (profiles/{pmc_name}: rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE, tools/pmc_decode.sh)"
                                                   if pmc_name else "no committed PMC pass over this loop (tools/pmc_decode.sh)"),
                                  "bytes_per_step": round(bytes_step), "us_per_step": round(us_step, 2), "steps_per_replay": steps,
                                  "note": "whole replayed token loop of one image (10 sub-graphs; 5 launches per step -- att-LSTM cell (files the previous pick), "
                                          "[h2att | hidden part of the lang-LSTM gates], attention, lang-LSTM (context part + cell), [logits with the arg-max epilogue | "
                                          "next step's att-LSTM gates] -- + the in-graph survivor gathers) / 21 steps; "
                                          "the 120 MB of weights fit the 256 MiB Infinity Cache, so the HBM roof is generous"}
    # the same images, decoded `group` at a time as one batch (sample_images): same tokens per image, weights streamed once per step
    group = min(256, images)                              # sized for 288 GB: 2560 sub-graph rows per decode step
    m.sample_images(warm[:group], opt=sopt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tokens = 0
    for _rep in range(2):
        for i in range(0, images, group):
            for r in m.sample_images(batches[i:i + group], opt=sopt):
                tokens += r[0].size(0) * r[0].size(1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # test.sh decodes Sub_GC_Kar with --beam_size 2: the same images through beam search, one per call and batched
    bopt = dict(sample_max=1, beam_size=2)
    nb = min(images, 64)
    for b in warm[:2]:
        m(*synthetic.sample_args(b), opt=bopt, mode="sample")
    torch.cuda.synchronize()
    tb0 = time.perf_counter()
    for b in batches[:nb]:
        m(*synthetic.sample_args(b), opt=bopt, mode="sample")
    torch.cuda.synchronize()
    tb1 = time.perf_counter()
    nbb = min(images, 256)
    m.sample_images(batches[:nbb], opt=bopt)
    torch.cuda.synchronize()
    tb2 = time.perf_counter()
    m.sample_images(batches[:nbb], opt=bopt)
    torch.cuda.synchronize()
    tb3 = time.perf_counter()
    out["decode_batched_roofline"] = gemm_roofline_of(
        lambda: [m.sample_images(batches[i:i + group], opt=sopt) for i in range(0, images, group)], "greedy_batched",
        f"sample_images over {images} images, {group} per decode batch: encoder + scoring + NMS + token loop; every GEMM of one pass")
    out["decode_batched_roofline"]["whole_pass_frac"] = round(out["decode_batched_roofline"]["gemm_gflop"] / 1e3 / (dt / 2) / MFMA_F32_PEAK_TFLOPS, 4)
    out["decode_beam2_batched_roofline"] = gemm_roofline_of(lambda: m.sample_images(batches[:nbb], opt=bopt), "beam2_batched",
                                                            f"sample_images with beam_size 2 over {nbb} images as one search")
    out["decode_beam2_batched_roofline"]["whole_pass_frac"] = round(out["decode_beam2_batched_roofline"]["gemm_gflop"] / 1e3 / (tb3 - tb2) / MFMA_F32_PEAK_TFLOPS, 4)
    out.update({"decode_beam2_ms_per_image": round(1e3 * (tb1 - tb0) / nb, 3), "decode_beam2_batched_ms_per_image": round(1e3 * (tb3 - tb2) / nbb, 3),
                "decode_beam2_config": f"beam_size 2 (test.sh Sub_GC_Kar), candidate bookkeeping on the device; batched = {nbb} images per search"})
    out.update({"decode_batched_tokens_per_s": round(tokens / dt, 1), "decode_batched_ms_per_image": round(1e3 * dt / (2 * images), 3),
                "decode_batched_config": f"sample_images: {group} images per decode batch (<= {10 * group} sub-graph rows per step)"})
    return out


def pmc_traffic(config, batch, world, launches_per_step):
    """HBM bytes per GEMM launch from the committed PMC passes of THIS command (tools/pmc_traffic.sh -> profiles/): hardware
    counters cannot be read from inside the timed run, so the figure is only reported when the profiled workload matches."""
    if world != 1 or batch != CONFIGS[config]["batch"]:
        return None, "no PMC profile for this configuration"
    note = "no PMC profile for this configuration"
    for rnd in ("r06", "r05", "r04", "r03", "r02"):                        # newest committed pass whose launch count matches this build
        name = f"{rnd}_pmc_traffic.json" if config == "kar" else f"{rnd}_pmc_traffic_{config}.json"
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        try:
            with open(path) as f:
                p = json.load(f)
        except ValueError:
            note = f"profiles/{name} is not a PMC summary"
            continue
        if p.get("gemm_launches_per_step") != launches_per_step:
            note = f"profiles/{name} was taken with {p.get('gemm_launches_per_step')} launches/step, this run has {launches_per_step}"
            continue
        return round(p["traffic_bytes_per_launch"]), (f"profiles/{name}: rocprofv3 --pmc FETCH_SIZE x 2.0 (gfx950 halves wide reads; calibrated in "
                                                      "profiles/r02_pmc_calibration.txt) + WRITE_SIZE (exact), separate passes")
    return None, note


def decode_pmc(leg):
    """HBM bytes per GEMM launch of a decode leg from the committed PMC passes (tools/pmc_decode_legs.sh -> profiles/rNN_pmc_decode_legs.json)."""
    for rnd in ("r06", "r05", "r04"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_decode_legs.json")
        if os.path.exists(path):
            with open(path) as f:
                d = json.load(f).get(leg)
            if d:
                return d, f"profiles/{rnd}_pmc_decode_legs.json"
    return None, None


def gemm_roofline_of(fn, leg, note):
    """MFMA roofline block of a decode leg: `fn()` runs the leg's workload once.  One pass with every GEMM launch bracketed by HIP events
    on its stream (subgc_prof_enable: launches, summed kernel time), one pass with the exact FLOP / algorithmic-byte accounting of
    ops.gemm, both untimed; `traffic` from the committed PMC passes of the same workload when its launch count matches."""
    _lib.prof_enable("gemm", True)
    fn()
    torch.cuda.synchronize()
    _lib.prof_enable("gemm", False)
    n_launch, gemm_ms, _ = _lib.prof_collect("gemm")
    ops.FLOPS.update(on=True, gemm=0.0, gemm_bytes=0.0, gemm_calls=0)
    fn()
    torch.cuda.synchronize()
    ops.FLOPS["on"] = False
    flops, calls = ops.FLOPS["gemm"], max(ops.FLOPS["gemm_calls"], 1)
    ach = flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    pmc, src = decode_pmc(leg)
    traffic, tnote = None, "no committed PMC pass for this leg (tools/pmc_decode_legs.sh)"
    if pmc is not None:
        if pmc.get("gemm_launches_per_pass") == n_launch:
            traffic, tnote = round(pmc["traffic_bytes_per_launch"]), f"{src}: rocprofv3 --pmc FETCH_SIZE x 2.0 + WRITE_SIZE, separate passes over the same workload"
        else:
            tnote = f"{src} was taken with {pmc.get('gemm_launches_per_pass')} GEMM launches per pass, this run has {n_launch}"
    return {"bound": "mfma", "kernel": "gemm_f32_kernel (v_mfma_f32_32x32x2_f32; decode batch = the kept sub-graph rows of the step)",
            "achieved": round(ach, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_F32_PEAK_TFLOPS, 4),
            "traffic": traffic, "traffic_unit": "HBM bytes per launch", "traffic_source": tnote,
            "algorithmic_bytes_per_launch": round(ops.FLOPS["gemm_bytes"] / calls), "gemm_launches": n_launch, "avg_launch_us": round(1e3 * gemm_ms / max(n_launch, 1), 2),
            "gemm_ms": round(gemm_ms, 3), "gemm_gflop": round(flops / 1e9, 2), "note": note}


def roofline_block(cfg, batch, flops_step, alg_bytes_launch, n_launch, n_s, gemm_ms, ms_per_step, traffic, traffic_note, steps):
    """`gemm_ms`: SUM of the GEMM launches' durations over the sampled steps (HIP events on the stream the launches run on; one queue, so the
    sum is also the wall time the family held the device)."""
    achieved = flops_step * n_s / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    return {"bound": "mfma", "kernel": cfg["kernel"], "achieved": round(achieved, 2),
            "peak": cfg["peak"], "unit": "TFLOP/s", "frac": round(achieved / cfg["peak"], 4),
            "traffic": traffic, "traffic_unit": "HBM bytes per launch", "traffic_source": traffic_note,
            "algorithmic_bytes_per_launch": round(alg_bytes_launch), "launches_per_step": n_launch // n_s,
            "event_sampled_steps": f"{n_s} of the {steps} timed steps ({n_launch} launches)",
            "avg_launch_us": round(1e3 * gemm_ms / max(n_launch, 1), 2),
            "gemm_ms_per_step": round(gemm_ms / n_s, 3), "gemm_gflop_per_step": round(flops_step / 1e9, 2),
            # whole step (all kernels + gaps) against the MFMA peak, with the GEMM FLOPs the step actually EXECUTES (the reference's
            # nominal live-graph FLOPs include masked-out decoder steps the packed path never computes: not a roofline figure)
            "whole_step_frac_executed": round(flops_step / (ms_per_step * 1e-3) / 1e12 / cfg["peak"], 4)}


def train_config_leg(name, dev, steps=8, warmup=3, opt_over=None):
    """One of the OTHER BASELINE.json train configs as its own contract block (value / ms_per_step / dtype / config.workload /
    roofline), measured exactly like the headline: resident synthetic batch, LossWrapper fwd + bwd + fused clip+Adam, GEMM
    launches of two of the timed steps bracketed by HIP events, one untimed accounting step for the exact FLOPs."""
    cfg = CONFIGS[name]
    B = cfg["batch"]
    torch.manual_seed(1234)
    model = models.setup(argparse.Namespace(**dict(cfg["opt"], **(opt_over or {})))).to(dev).train()
    lw = models.LossWrapper(model, None)
    batch = {k: v.to(dev) for k, v in synthetic.make_train_batch(B, seed=1000, **cfg["data"]).items()}
    adam = parallel.FlatAdam(model)
    one = ops.fill_(torch.empty((), device=dev, dtype=torch.float32), 1.0)

    def step():
        model.flatten_grads()
        out = lw(*lw_args(batch))
        loss = models.total_loss(out)
        loss.backward(one)
        adam.step(zero_grad=True)
        return loss

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    smp = sorted({0, steps // 2})
    t0 = time.perf_counter()
    for i in range(steps):
        _lib.prof_enable("gemm", i in smp)
        loss = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    _lib.prof_enable("gemm", False)
    n_launch, gemm_ms, _ = _lib.prof_collect("gemm")
    ops.FLOPS.update(on=True, gemm=0.0, gemm_bytes=0.0, gemm_calls=0)
    step()
    torch.cuda.synchronize()
    ops.FLOPS["on"] = False
    flops_step, calls = ops.FLOPS["gemm"], max(ops.FLOPS["gemm_calls"], 1)
    ms = 1e3 * dt / steps
    traffic, note = pmc_traffic(name, B, 1, n_launch // len(smp))
    res = {"metric": cfg["metric"], "value": round(B * steps / dt, 2), "unit": "images/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
           "ms_per_step": round(ms, 3), "dtype": cfg["dtype"], "data": "synthetic",
           "config": {"workload": cfg["workload"], "images_per_gpu": B, "decoder": "packed", "includes": "fwd + bwd + fused clip+Adam"},
           "roofline": roofline_block(cfg, B, flops_step, ops.FLOPS["gemm_bytes"] / calls, n_launch, len(smp), gemm_ms, ms, traffic, note, steps),
           "final_loss": round(float(loss.item()), 4)}
    del model, lw, batch, adam, step
    torch.cuda.empty_cache()
    return res


MRNN = dict(test_LSTM=1, gpn_nms_thres=0.55, gpn_max_subg=1000, use_topk_sampling=1, topk_temp=0.6, the_k=3)      # test.sh:24-30
DECODE_MFLOP_PER_TOKEN = 76.0           # decoder forward FLOPs per sentence-token


def mrnn_decode_leg(dev, images=6, M=500, seed0=900, model_sd=None):
    """BASELINE.json configs[3]'s decode half on one GPU, as test.sh runs Sub_GC_S_MRNN: 2M = 1000 candidate sub-graphs per image,
    NMS 0.55, keep <= 1000, top-k sampling (k = 3, T = 0.6), one image per call.  -> (block, tokens, seconds)."""
    opt = argparse.Namespace(**dict(KAR, **MRNN))
    torch.manual_seed(0)
    m = models.setup(opt)
    if model_sd is not None:
        m.load_state_dict(model_sd)
    m = m.to(dev).eval()
    batches = [{k: v.to(dev) for k, v in synthetic.make_test_batch(M, seed=seed0 + i).items()} for i in range(images)]
    sopt = dict(sample_max=1, beam_size=1)
    for b in batches[:2]:
        m(*synthetic.sample_args(b), opt=sopt, mode="sample")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rows = tokens = 0
    for b in batches:
        seq = m(*synthetic.sample_args(b), opt=sopt, mode="sample")[0]
        rows += seq.size(0); tokens += seq.numel()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tf = tokens * DECODE_MFLOP_PER_TOKEN * 1e6 / dt / 1e12
    roof = gemm_roofline_of(lambda: [m(*synthetic.sample_args(b), opt=sopt, mode="sample") for b in batches], "mrnn",
                            f"the {images} timed calls again: every GEMM launch of encode + scoring + token loop (~900 kept rows per step)")
    roof["whole_call_frac"] = round(roof["gemm_gflop"] / 1e3 / dt / MFMA_F32_PEAK_TFLOPS, 4)
    roof["nominal_76_mflop_per_token_tflops"] = round(tf, 2)
    res = {"metric": "decode tokens/sec, Sub_GC_MRNN top-k sampling", "value": round(tokens / dt, 1), "unit": "tokens/s", "n_gpus": 1,
           "ms_per_step": round(1e3 * dt / images, 3), "step": "one image (one model call)", "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"Sub_GC_MRNN decode (BASELINE.json configs[3], test.sh:20-30): {2 * M} candidate sub-graphs/image -> NMS 0.55 -> "
                                  f"keep <= 1000 -> top-k sampling k=3 T=0.6, 20 tokens, one image per call, {images} images",
                      "kept_subgraphs_per_image": round(rows / images, 1)},
           "roofline": roof}
    del m, batches
    torch.cuda.empty_cache()
    return res, tokens, dt


def decode_bench_sharded(model_sd, dev, rank, world, per_rank=64, M=50):
    """N > 1: the decode half of the metric.  Weak scaling: `per_rank` images per GPU, image i of the global list on rank
    i % world (parallel.shard_images), no collective on the way, ONE gather of the token ids / log-probs / scores / kept indices
    at the end (inside the timed region).  tokens summed over ranks / max-over-ranks time."""
    opt = argparse.Namespace(**dict(KAR, test_LSTM=1, gpn_nms_thres=0.75, gpn_max_subg=10))
    m = models.setup(opt)
    m.load_state_dict(model_sd)
    m = m.to(dev).eval()
    total = per_rank * world
    idx = list(range(rank, total, world))
    mine = [{k: v.to(dev) for k, v in synthetic.make_test_batch(M, seed=500 + i).items()} for i in idx]
    sopt = dict(sample_max=1, beam_size=1)

    def fence():
        torch.cuda.synchronize()
        dist.barrier()

    def agg(tokens, dt):
        t = torch.tensor([float(tokens), 0.0], device=dev, dtype=torch.float64)
        d = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dist.all_reduce(d, op=dist.ReduceOp.MAX)
        return float(t[0].item()), float(d.item())

    for b in mine:                                                     # untimed pass: one hipGraph capture per surviving-row count
        m(*synthetic.sample_args(b), opt=sopt, mode="sample")
    m.sample_images(mine, opt=sopt)
    fence()
    t0 = time.perf_counter()                                           # the reference's shape: one image per call, per rank
    local = [m(*synthetic.sample_args(b), opt=sopt, mode="sample") for b in mine]
    full = parallel.gather_by_index(local, idx, total)
    fence()
    tok1, dt1 = agg(sum(r[0].numel() for r in local), time.perf_counter() - t0)
    assert len(full) == total
    t0 = time.perf_counter()                                           # each rank's share as one decode batch
    local = m.sample_images(mine, opt=sopt)
    full = parallel.gather_by_index(local, idx, total)
    fence()
    tok2, dt2 = agg(sum(r[0].numel() for r in local), time.perf_counter() - t0)
    roof2 = gemm_roofline_of(lambda: m.sample_images(mine, opt=sopt), "greedy_batched_sharded", f"rank 0's share ({per_rank} images) as one decode batch")
    del m, mine
    torch.cuda.empty_cache()
    leg3, tok3, dt3 = mrnn_decode_leg(dev, images=3, M=500, seed0=900 + 16 * rank, model_sd=None)
    fence()
    tok3, dt3 = agg(tok3, dt3)
    return {"decode_tokens_per_s": round(tok1 / dt1, 1), "decode_ms_per_image": round(1e3 * dt1 / per_rank, 3),
            "decode_config": f"greedy, {2 * M} candidate sub-graphs/image -> NMS 0.75 -> <=10 kept x 20 tokens, one image per call, {per_rank} images per "
                             f"rank round-robin over {world} ranks, one all_gather_object of the results at the end (timed)",
            "decode_batched_tokens_per_s": round(tok2 / dt2, 1), "decode_batched_config": f"sample_images: each rank's {per_rank} images as one decode batch, gathered at the end",
            "decode_batched_roofline": roof2, "decode_mrnn_topk_roofline": leg3["roofline"],
            "decode_mrnn_topk_tokens_per_s": round(tok3 / dt3, 1),
            "decode_mrnn_topk_config": "test.sh Sub_GC_S_MRNN: 1000 candidates/image, NMS 0.55, keep <= 1000, top-k 3 @ 0.6, one image per call, 3 images per rank"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="kar", choices=sorted(CONFIGS), help="kar = the headline (BASELINE.json configs[1]); full_gc_kar / "
                    "flickr = the bf16 parity configs 3 / 5 as their own contract lines")
    ap.add_argument("--batch", type=int, default=0, help="images per GPU (default: the config's, 128 for Sub_GC_Kar)")
    ap.add_argument("--cpu-images", type=int, default=32)
    ap.add_argument("--cpu-iters", type=int, default=6)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-decode", action="store_true")
    ap.add_argument("--decode-images", type=int, default=64, help="N > 1: images per rank of the sharded decode leg")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the other BASELINE configs' legs (clean profiles of the headline)")
    ap.add_argument("--packed-only", action="store_true", help="skip the extra unpacked-decoder leg (clean profiles)")
    ap.add_argument("--no-optimizer", action="store_true", help="time fwd + bwd (+ all-reduce) only; by default the fused clip+Adam update is inside the timed step")
    ap.add_argument("--with-optimizer", action="store_true", help="accepted for older command lines: the optimizer step is on by default")
    ap.add_argument("--pitch-f32", type=int, default=-1, help="experiment: row pitch (elements) of the fp32 recurrent operands (ops.PITCH['f32']; 32 = 128 bytes)")
    ap.add_argument("--no-fold-bias", action="store_true", help="experiment: bias gradients as separate column-sum launches (functions.FOLD_BIAS_SUMS = False)")
    ap.add_argument("--no-pair-launches", action="store_true", help="experiment: the two same-shape products of a GCN unit pair as two launches (ops.PAIR_LAUNCHES = False)")
    ap.add_argument("--share-attention-sets", type=int, default=-1, help="Full-GC configs: 1 = attention sets once per image (ties the att_embed dropout mask "
                    "across an image's sentences: NOT the reference's semantics), 0 = the reference's independent masks on replicated rows (model default)")
    ap.add_argument("--dedup", type=int, default=-1, help="experiment (Full-GC): opt.dedup_att_embed (1 = att_embed once per node row + masked gather per copy, 0 = on the replicated rows)")
    ap.add_argument("--no-p8", action="store_true", help="experiment (bf16 configs): SUBGC_GEMM_NO_P8 on every subgc_gemm_bf16 call -- the ring forms only (A/B of the eight-phase form)")
    ap.add_argument("--ss-prob", type=float, default=0.0, help="scheduled-sampling probability (train.py raises it from epoch 5; the headline workload is 0)")
    a = ap.parse_args()
    cfg = CONFIGS[a.config]
    a.batch = a.batch or cfg["batch"]
    headline = a.config == "kar"

    # SUBGC_BENCH_REHEARSAL=1: run the N > 1 code path on a ONE-GPU box (every rank on cuda:0, gloo carrying the tensors) --
    # a functional rehearsal of the launch contract, not a measurement
    rehearsal = os.environ.get("SUBGC_BENCH_REHEARSAL") == "1"
    rank, local, world = parallel.init_distributed("gloo" if rehearsal else None)
    if rehearsal:
        local = 0
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    _lib.lib()
    torch.manual_seed(1234)                     # identical replicas on every rank
    opt = argparse.Namespace(**cfg["opt"])
    if a.share_attention_sets >= 0:
        opt.share_attention_sets = a.share_attention_sets
    if a.dedup >= 0:
        opt.dedup_att_embed = a.dedup
    model = models.setup(opt).to(dev).train()
    if a.pitch_f32 >= 0:
        ops.PITCH["f32"] = a.pitch_f32
    if a.no_p8:
        ops.gemm_tune.b16_bits |= 1 << 14
    if a.no_pair_launches:
        ops.PAIR_LAUNCHES = False
    if a.no_fold_bias:
        import subgc.functions as F_
        F_.FOLD_BIAS_SUMS = False
    if a.ss_prob > 0:
        model.ss_prob = a.ss_prob
    lw = models.LossWrapper(model, None)
    batch = {k: v.to(dev) for k, v in synthetic.make_train_batch(a.batch, seed=1000 + rank, **cfg["data"]).items()}
    adam = None if a.no_optimizer else parallel.FlatAdam(model)      # a training step ends with the parameter update (misc/utils.py:174-200 + Adam)
    red = parallel.GradBucketReducer(model, optimizer=adam)          # the clip norm is accumulated slice by slice as the slices become final
    one = ops.fill_(torch.empty((), device=dev, dtype=torch.float32), 1.0)

    def step():
        red.prepare()
        out = lw(*lw_args(batch))
        loss = models.total_loss(out)
        loss.backward(one)                                # d(loss) = 1, a resident scalar (no fill launch per step)
        red.finish(average=adam is None)                 # with the optimizer on, 1/world rides in its sweep
        if adam is not None:
            adam.step(grad_scale=1.0 / world, zero_grad=True)        # step + the iteration's optimizer.zero_grad() in one sweep
        return loss

    def timed_sampled(n_steps):
        """n_steps timed steps, GEMM events on two of them (see the headline loop) -> (seconds, sampled steps, GEMM ms over them, last loss)."""
        smp = sorted({0, n_steps // 2})
        t0_ = time.perf_counter()
        for i in range(n_steps):
            _lib.prof_enable("gemm", i in smp)
            last = step()
        fence()
        dt_ = time.perf_counter() - t0_
        _lib.prof_enable("gemm", False)
        _, ms_, _ = _lib.prof_collect("gemm")
        return dt_, len(smp), ms_, last

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    # Roofline events: every GEMM launch of the SAMPLED steps of the timed region is bracketed by HIP events on its launch
    # stream.  Bracketing all K steps costs 4-5 % of the step (each event pair drains the launch pipeline between kernels), so
    # two of the K timed steps carry the events (330 launches) and the others run as the product does.
    sampled = sorted({0, a.steps // 2}) if rank == 0 else []
    t0 = time.perf_counter()
    for i in range(a.steps):
        if rank == 0:
            _lib.prof_enable("gemm", i in sampled)
        loss = step()
    fence()
    elapsed = time.perf_counter() - t0
    if rank == 0:
        _lib.prof_enable("gemm", False)
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    final_loss = float(loss.item())

    # one untimed accounting step: exact (ragged-aware) GEMM FLOPs.  EVERY rank runs it -- the step contains the gradient
    # all-reduce, so a rank-0-only step would wait for its peers forever
    ops.FLOPS.update(on=(rank == 0), gemm=0.0, gemm_bytes=0.0, gemm_calls=0)
    step()
    fence()
    ops.FLOPS["on"] = False
    if rank == 0:
        n_launch, gemm_ms, _nominal = _lib.prof_collect("gemm")
        flops_step = ops.FLOPS["gemm"]
        alg_bytes_launch = ops.FLOPS["gemm_bytes"] / max(ops.FLOPS["gemm_calls"], 1)
        n_s = max(len(sampled), 1)
        traffic, traffic_note = pmc_traffic(a.config, a.batch, world, n_launch // n_s)
        ms_per_step = 1e3 * elapsed / a.steps
        imgs = world * a.batch
        res = {
            "metric": cfg["metric"], "value": round(imgs * a.steps / elapsed, 2), "unit": "images/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": cfg["dtype"], "data": "synthetic",
            "config": {"workload": cfg["workload"],
                       "images_per_gpu": a.batch, "global_images": imgs, "parallelism": f"dp{world}" if world > 1 else "single",
                       "decoder": "packed (length-sorted, loss-only: masked-out steps skipped; identical loss and gradients)",
                       "includes": "fwd + bwd" + (" + RCCL grad all-reduce" if world > 1 else "") + (" + fused clip+Adam" if adam else "")
                                   + (f" + scheduled sampling p={a.ss_prob} (per-step logits and draws)" if a.ss_prob > 0 else "")},
            "roofline": roofline_block(cfg, a.batch, flops_step, alg_bytes_launch, n_launch, n_s, gemm_ms, ms_per_step, traffic, traffic_note, a.steps),
            "final_loss": round(final_loss, 4),
        }
        pmc_path = os.path.join(ROOT, "profiles", "r02_pmc_gemm_f32.json")
        if headline and os.path.exists(pmc_path):                 # committed PMC passes over single shapes of the same kernel
            with open(pmc_path) as f:
                res["roofline"]["pmc"] = json.load(f)
        if world == 1 and headline:
            # the HBM-bound kernel families of the same step (reported individually in GB/s against the 8 TB/s
            # roof): two extra untimed steps with every family bracketed by HIP events on its launch stream; algorithmic
            # bytes are counted by the entry points themselves (attention: sum of the ragged set sizes, counted here)
            fams = ("gcn", "pool", "attn", "lstm", "softmax")
            for f_ in fams:
                _lib.prof_enable(f_, True)
            ops.FLOPS.update(on=True, attn_bytes=0.0, pool_bytes=0.0)
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            ops.FLOPS["on"] = False
            hbm = {}
            for f_ in fams:
                _lib.prof_enable(f_, False)
                n_, ms_, work_ = _lib.prof_collect(f_)
                if f_ in ("attn", "pool"):                                          # ragged families: bytes counted on the host side
                    work_ = ops.FLOPS.get(f_ + "_bytes", 0.0)
                if n_ and ms_ > 0:
                    gbps = work_ / (ms_ * 1e-3) / 1e9
                    hbm[f_] = {"launches_per_step": n_ // 2, "ms_per_step": round(ms_ / 2, 3), "algorithmic_MB_per_step": round(work_ / 2 / 1e6, 1),
                               "achieved_GBps": round(gbps, 1), "frac_of_8TBps": round(gbps / 8000.0, 4)}
                    if f_ == "lstm":
                        # what the cell launches really move: the split-K planes of the gate products (summed on load instead of by a
                        # reduce pass), the x->gates / fc->gates terms and the gates saved for the backward ride on top of the 12-14
                        # floats per hidden unit of the algorithmic count -- the kernels run near the memory roof on THOSE bytes
                        mv = _lib.prof_last_moved(f_)
                        hbm[f_].update(moved_MB_per_step=round(mv / 2 / 1e6, 1), moved_GBps=round(mv / (ms_ * 1e-3) / 1e9, 1),
                                       moved_frac_of_8TBps=round(mv / (ms_ * 1e-3) / 1e9 / 8000.0, 4))
            res["hbm_bound_kernels"] = hbm
        if world == 1 and adam is not None:
            # the metric's literal scope, fwd + bwd without the parameter update, timed the same way (the headline includes the update)
            saved, adam = adam, None
            for _ in range(2):
                step()
            fence()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                step()
            fence()
            dt = time.perf_counter() - t0
            adam = saved
            res["fwd_bwd_only"] = {"value": round(imgs * a.steps / dt, 2), "ms_per_step": round(1e3 * dt / a.steps, 3)}
        if world == 1 and headline and not a.packed_only:
            # host side of the step: the same launch sequence on a 4-image batch -- the device work shrinks ~30x, the number of
            # launches and host reads does not, so the step time there is what the host needs to enqueue one step
            small = {k: v.to(dev) for k, v in synthetic.make_train_batch(4, seed=999).items()}
            big, batch = batch, small
            for _ in range(3):
                step()
            fence()
            t0 = time.perf_counter()
            for _ in range(10):
                step()
            fence()
            res["host_enqueue_ms_per_step"] = round(1e2 * (time.perf_counter() - t0), 3)
            res["host_enqueue_note"] = "ms per step of the identical launch sequence on a 4-image batch (device time negligible): the host-side floor of a step"
            batch = big
        if world == 1 and headline and not a.packed_only:
            # the same step with the loss-only packing switched off (every sentence runs all T steps, `outputs`
            # is materialised exactly like the reference does): reported beside the default for comparison
            model.packed_decoder = False
            for _ in range(2):
                step()
            fence()
            dt, ns2, ms2, _ = timed_sampled(a.steps)
            ops.FLOPS["on"], ops.FLOPS["gemm"] = True, 0.0
            step()
            torch.cuda.synchronize()
            ops.FLOPS["on"] = False
            ach2 = ops.FLOPS["gemm"] * ns2 / (ms2 * 1e-3) / 1e12
            res["unpacked_decoder"] = {"value": round(imgs * a.steps / dt, 2), "ms_per_step": round(1e3 * dt / a.steps, 3),
                                       "gemm_gflop_per_step": round(ops.FLOPS["gemm"] / 1e9, 2), "gemm_achieved_tflops": round(ach2, 2),
                                       "gemm_frac_of_peak": round(ach2 / MFMA_F32_PEAK_TFLOPS, 4)}
            model.packed_decoder = True
            # NOT the headline (fp32, above): the same packed step with the 128x128-tile GEMMs in the two other arithmetic
            # modes of csrc/gemm_x3.h -- "bf16x3": fp32 operands split exactly into 3 bf16 planes, 6 bf16-MFMA terms, fp32
            # accumulate (fp32-grade results); "bf16": operands rounded to bf16 (the compute type of BASELINE configs 3, 5)
            notes = {"bf16x3": "opt-in; error vs fp64 within 2x of the fp32-MFMA kernel (tests); fp32-equivalent FLOPs",
                     "bf16": "opt-in; bf16 operand rounding, fp32 accumulate/storage: a DIFFERENT precision than `value`'s fp32"}
            res["other_gemm_modes"] = {}
            for mode in ("bf16x3", "bf16"):
                with ops.gemm_mode(mode):
                    for _ in range(2):
                        step()
                    fence()
                    dt, ns3, ms3, loss3 = timed_sampled(a.steps)
                    res["other_gemm_modes"][mode] = {
                        "value": round(imgs * a.steps / dt, 2), "ms_per_step": round(1e3 * dt / a.steps, 3),
                        "gemm_algorithmic_tflops": round(flops_step * ns3 / (ms3 * 1e-3) / 1e12, 2),
                        "final_loss": round(float(loss3.item()), 4), "note": notes[mode]}
        if world == 1 and headline and not a.no_decode:
            res.update(decode_bench(model.state_dict(), dev, images=256, M=50))
        if world == 1 and headline and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_train_baseline(a.cpu_images, a.cpu_iters)
            res["speedup_vs_cpu_baseline"] = round(res["value"] / res["cpu_baseline"]["value"], 1)
            # the reference's own CPU-runnable case (B = 2) and its decode shape at M = 50 and M = 500 pairs
            res["cpu_baseline_b2"] = cpu_train_baseline(2, 5)
            if not a.no_decode:
                res["cpu_baseline_decode"] = {"M50": cpu_decode_baseline(50, 4), "M500": cpu_decode_baseline(500, 2)}
                res["decode_speedup_vs_cpu_M50"] = round(res["decode_tokens_per_s"] / res["cpu_baseline_decode"]["M50"]["value"], 1)
        if world == 1 and headline and not a.no_other_configs:
            # every other BASELINE.json config that fits one GPU, on the same clock as the headline: configs[2] and [4] as train
            # steps in bf16 (their own value / dtype / workload / roofline), configs[3]'s decode half as test.sh runs it
            oc = {}
            for name in ("full_gc_kar", "flickr"):
                oc[name + "_bf16"] = train_config_leg(name, dev, steps=8, warmup=3)
            # Full_GC_Kar: the contract line above draws the reference's five INDEPENDENT att_embed dropout masks per image (replicated
            # rows, the model default); the variant that computes the attention sets once per image ties those masks within an image
            # (share_attention_sets = 1: not the reference's joint distribution) and is reported beside it, never instead of it
            oc["full_gc_kar_bf16"]["config"]["att_embed_dropout"] = "independent per sentence (reference: gcn_backbone.py:50-51, AttModel.py:113-119)"
            tied = train_config_leg("full_gc_kar", dev, steps=8, warmup=3, opt_over={"share_attention_sets": 1})
            oc["full_gc_kar_bf16"]["variant_shared_attention_sets"] = {
                "value": tied["value"], "ms_per_step": tied["ms_per_step"], "roofline_frac": tied["roofline"]["frac"],
                "note": "share_attention_sets=1: att_embed / ctx2att once per image, one keep-mask per image instead of five (deviates from the reference under dropout)"}
            oc["kar_ss25"] = train_config_leg("kar_ss25", dev, steps=8, warmup=3)
            oc["kar_ss25"]["vs_headline_step"] = round(oc["kar_ss25"]["ms_per_step"] / res["ms_per_step"], 4)
            if not a.no_decode:
                oc["mrnn_decode_topk"] = mrnn_decode_leg(dev, images=6, M=500)[0]
            res["other_configs"] = oc
    if world > 1:
        # first-contact evidence for the readiness-ordered buckets: two extra untimed steps with the reducer's event brackets
        # on EVERY rank (a collective is a collective): when each slice could start, when the compute stream got past it, what the step
        # paid for communication.  Rank 0 reports its own view and the max exposed time over the ranks.
        red.timing = True
        comm = None
        for _ in range(2):
            step()
            torch.cuda.synchronize()
            comm = red.report()
        red.timing = False
        red._ev = None
        ex = torch.tensor([comm["exposed_ms"] if comm else 0.0], device=dev, dtype=torch.float64)
        dist.all_reduce(ex, op=dist.ReduceOp.MAX)
        if rank == 0:
            res["communication"] = {"rccl_ranks": world, "backend": dist.get_backend(), "rehearsal_not_rccl": rehearsal,
                                    "grad_bytes_per_rank": int(model.flat_params.numel()) * 4, "collectives_per_step": len(red.issued),
                                    "overlap": red.overlap, "rank0": comm, "exposed_ms_max_over_ranks": round(float(ex.item()), 3),
                                    "note": "HIP events on the compute stream: issue_ms = the slice's gradients are final (collective may start), done_by_ms = "
                                            "compute stream past the wait; exposed = last wait end - backward end"}
    sd = None
    if world > 1 and headline and not a.no_decode:                        # every rank decodes its share; rank 0 prints
        sd = decode_bench_sharded(model.state_dict(), dev, rank, world, per_rank=a.decode_images)
    if rank == 0:
        if sd is not None:
            res.update(sd)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
