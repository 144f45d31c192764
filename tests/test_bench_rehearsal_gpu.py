"""bench.py's N > 1 launch contract, rehearsed on the one-GPU test box: two ranks started by torch.distributed.run exactly
as the driver does, both on cuda:0 with gloo carrying the tensors (SUBGC_BENCH_REHEARSAL=1).  Every collective of the
timed region and of the accounting step must be entered by every rank (a rank-0-only step once deadlocked this path),
and rank 0 must print exactly one JSON line with the contract's keys."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(400)
def test_two_rank_bench_prints_one_contract_line():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, SUBGC_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "16", "--decode-images", "6"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["config"]["global_images"] == 32 and d["value"] > 0
    # the decode half of the metric at N > 1: images sharded round-robin, results gathered once, tokens summed over ranks
    for k in ("decode_tokens_per_s", "decode_batched_tokens_per_s", "decode_mrnn_topk_tokens_per_s"):
        assert d[k] > 0, k
    assert "cpu_baseline" not in d                                   # rank 0 at N = 1 only
    # first-contact evidence block of an N > 1 run: ranks, backend, every collective's issue point and the exposed communication
    c = d["communication"]
    assert c["rccl_ranks"] == 2 and c["backend"] == "gloo" and c["rehearsal_not_rccl"] is True and c["collectives_per_step"] == 5
    assert [b["stage"] for b in c["rank0"]["buckets"]] == ["logit", "recurrent", "prepare", "gcn", "fusion"]
    assert sum(b["bytes"] for b in c["rank0"]["buckets"]) == c["grad_bytes_per_rank"]
    t = [b["issue_ms"] for b in c["rank0"]["buckets"]]
    assert t == sorted(t) and t[0] > 0 and c["rank0"]["backward_end_ms"] >= t[-1] and c["exposed_ms_max_over_ranks"] >= 0


@pytest.mark.timeout(1500)
def test_eight_rank_bench_rehearsal_on_one_gpu():
    """`bench.py --gpus 8` as the driver launches it (torch.distributed.run, 8 ranks), all eight on the one GPU of the test box with
    gloo carrying the tensors: rendezvous, per-rank seeds and shards, the four readiness-ordered collectives per step, the sharded
    decode legs and their gather, memory of eight replicas -- everything but the timing."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, SUBGC_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--batch", "8", "--decode-images", "2"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1400)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 2 and d["scaling"] == "weak" and d["config"]["global_images"] == 64 and d["value"] > 0
    assert d["config"]["parallelism"].startswith("dp8")
    for k in ("decode_tokens_per_s", "decode_batched_tokens_per_s", "decode_mrnn_topk_tokens_per_s"):
        assert d[k] > 0, k


@pytest.mark.timeout(900)
@pytest.mark.parametrize("config,batch", [("full_gc_kar", 8), ("flickr", 4)])
def test_two_rank_rehearsal_of_the_bf16_configs(config, batch):
    """The other two train configs through the N > 1 path (two ranks on the one GPU, gloo): Full-GC has no sGPN, four BatchNorm layers
    and shared attention sets, Flickr the 2048-d GCN -- both go through the readiness-ordered buckets (the `gcn` slice is announced by
    the fusion-output markers of THEIR encoder graphs) and the fused Adam sweep with 1 / world folded in."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, SUBGC_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", config, "--steps", "2", "--warmup", "1",
           "--batch", str(batch)]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dtype"] == "bf16" and d["config"]["global_images"] == 2 * batch and d["value"] > 0
    assert d["final_loss"] == d["final_loss"] and d["final_loss"] > 0            # finite
