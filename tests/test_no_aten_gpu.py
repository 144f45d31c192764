"""Every FLOP and byte a training step moves on the device goes through the C ABI: one step (forward, backward, fused clip + Adam) of each
BASELINE train config, watched with a TorchDispatchMode, issues no ATen op that launches a device kernel -- views, allocations and the
pinned copy of the packed plan's row counts are all that is left to torch (boundary).  The same holds for the decode
paths the reference's test.sh exercises: one image per call (greedy and beam 2, replayed hipGraphs plus the selection launches around
them) and `sample_images` over several images -- small host <-> device copies (address tables, survivor counts, finished beams) aside."""
import argparse
import collections
import os

import pytest
import torch
from torch.utils._python_dispatch import TorchDispatchMode

import bench
import subgc.models as models
from subgc import parallel, synthetic

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.getenv("SUBGC_POISON_EMPTY") == "1", reason="the poisoned run fills every torch.empty buffer with an ATen fill_ by design")]
DEV = "cuda:0"

# ops that only make views / allocate / move the 17 plan counts to pinned memory
HARMLESS = ("view", "reshape", "empty", "as_strided", "detach", "alias", "slice", "select", "transpose", "permute", "expand", "unsqueeze",
            "squeeze", "_unsafe_view", "t.default", "unbind", "split", "narrow", "pin_memory", "is_pinned", "record_stream", "_reshape_alias",
            "lift_fresh", "unfold", "chunk", "set_", "resize_", "stride", "sym_", "is_same_size", "_local_scalar_dense")


class Watch(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.seen = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        r = func(*args, **(kwargs or {}))
        name = str(func)
        if not any(h in name for h in HARMLESS):
            ts = [t for t in (list(args) + ([r] if torch.is_tensor(r) else [])) if torch.is_tensor(t)]
            if any(t.is_cuda for t in ts):
                host_copy = ("copy_" in name or "_to_copy" in name) and any(not t.is_cuda for t in ts)   # host <-> device: a DMA, not a kernel
                if not host_copy:
                    self.seen[name] += 1
        return r


@pytest.mark.parametrize("config", ["kar", "full_gc_kar", "flickr"])
def test_train_step_issues_no_aten_device_kernel(config):
    torch.manual_seed(3)
    if config == "kar":
        opt, data, B = bench.KAR, {}, 8
    else:
        cfg = bench.CONFIGS[config]
        opt, data, B = cfg["opt"], cfg["data"], 8
    m = models.setup(argparse.Namespace(**opt)).to(DEV).train()
    lw = models.LossWrapper(m, None)
    b = {k: v.to(DEV) for k, v in synthetic.make_train_batch(B, seed=5, **data).items()}
    adam = parallel.FlatAdam(m)
    one = torch.ones((), device=DEV)

    def step():
        m.flatten_grads()
        out = lw(*bench.lw_args(b))
        models.total_loss(out).backward(one)
        adam.step()

    step()
    step()
    torch.cuda.synchronize()
    with Watch() as w:
        step()
    torch.cuda.synchronize()
    assert not w.seen, dict(w.seen)


def _decode_model():
    torch.manual_seed(3)
    opt = argparse.Namespace(**dict(bench.KAR, test_LSTM=1, gpn_nms_thres=0.75, gpn_max_subg=10))
    m = models.setup(opt).to(DEV).eval()
    images = [{k: v.to(DEV) for k, v in synthetic.make_test_batch(M, seed=40 + i).items()} for i, M in enumerate((30, 50, 12, 50))]
    return m, images


@pytest.mark.parametrize("mode", ["greedy_one_image", "beam2_one_image", "sample_images"])
def test_decode_issues_no_aten_device_kernel(mode):
    """test.sh's decode shapes (misc/eval_utils.py:98-104: one image per model call; beam 2 for Sub_GC_Kar) and the batched form: after
    the first call (graph capture, x->gates table, constant tables) a decode issues only C-ABI launches, inside the replayed graph and
    around it."""
    m, images = _decode_model()
    sopt = dict(sample_max=1, beam_size=2 if mode.startswith("beam2") else 1)

    def run():
        if mode == "sample_images":
            return m.sample_images(images, opt=sopt)
        return [m(*synthetic.sample_args(b), opt=sopt, mode="sample") for b in images]

    want = run()                                               # captures the graphs of every survivor count these images produce
    run()
    torch.cuda.synchronize()
    with Watch() as w:
        got = run()
    torch.cuda.synchronize()
    assert not w.seen, dict(w.seen)
    for a, b in zip(got, want):                                # and the watched run decoded the same captions
        assert torch.equal(torch.as_tensor(a[0]).cpu(), torch.as_tensor(b[0]).cpu())
        assert torch.equal(a[3].cpu(), b[3].cpu())
