"""No kernel of the train path reads memory that nothing wrote: a step whose every `torch.empty` / `torch.empty_like` device buffer starts
as NaN (integers as a large value) gives the same loss and gradients as the plain run.  Covers in particular the deferred d(v)
accumulation (functions.py `defer_dv`: dv is allocated with torch.empty, `attn_dv_accum` writes only the live rows, `relu_bwd` then runs
over all of them and every consumer honours the device-side row count), the packed decoder's slack rows and the split-K planes, in the
fp32 and bf16 storage modes.  (`SUBGC_POISON_EMPTY=1 python -m pytest tests -m gpu` applies the same poison to the whole suite.)"""
import contextlib

import numpy as np
import pytest
import torch

from subgc import synthetic
from test_parity_gpu import DEV, build, run_train

pytestmark = pytest.mark.gpu


@contextlib.contextmanager
def poisoned_empty():
    real = torch.empty, torch.empty_like

    def poisoned(fn):
        def wrap(*a, **k):
            t = fn(*a, **k)
            if t.is_cuda and t.numel():
                t.fill_(float("nan")) if t.is_floating_point() else (t.fill_(0x3fffffff) if t.dtype in (torch.int32, torch.int64) else t.fill_(255))
            return t
        return wrap
    torch.empty, torch.empty_like = poisoned(real[0]), poisoned(real[1])
    try:
        yield
    finally:
        torch.empty, torch.empty_like = real


def _batch(g, B, seed):
    o = g.meta["opt"]
    b = synthetic.make_train_batch(B, D=o["att_feat_size"], vocab=o["vocab_size"], n_obj_cls=o.get("sg_obj_cnt", 1599), seed=seed,
                                   fc_size=o["fc_feat_size"], min_len=1, max_len=16)
    b["labels"][3:5] = 0                                       # sentences without words: dead after step 0 in the packed decoder
    b["masks"][3:5, 2:] = 0
    return b


@pytest.mark.parametrize("packed", [True, False])
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("name", ["subgc_train", "fullgc_train"])
def test_train_step_is_unchanged_when_every_empty_buffer_starts_as_nan(golden, name, dtype, packed):
    g = golden(name)
    res = []
    for poison in (False, True):
        with (poisoned_empty() if poison else contextlib.nullcontext()):
            m = build(g, g.group("weights"), True, compute_dtype=dtype, drop_prob_lm=0.0)
            m.packed_decoder = packed
            out, loss = run_train(m, _batch(g, 4, 11))
            res.append((float(loss.detach()), {k: p.grad.clone().cpu() for k, p in m.named_parameters()}))
    (l0, g0), (l1, g1) = res
    assert np.isfinite(l1) and abs(l1 - l0) <= (1e-3 if dtype == "bf16" else 1e-5) * max(1.0, abs(l0))
    top = max(float(v.abs().max()) for v in g0.values())
    for k in g0:
        assert bool(torch.isfinite(g1[k]).all()), k
        # the only run-to-run difference allowed is the order of the fp32 atomics in the pooling backward
        np.testing.assert_allclose(g1[k].numpy(), g0[k].numpy(), atol=(2e-3 if dtype == "bf16" else 2e-6) * top, rtol=(2e-2 if dtype == "bf16" else 1e-4), err_msg=k)
