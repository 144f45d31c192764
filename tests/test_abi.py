"""CPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/subgc_hip.h declares, validates arguments on the host and fails loudly without a GPU."""
import argparse
import ctypes

import pytest
import torch

from subgc import _lib, ops
import subgc.models as models


def test_header_symbols_all_exported():
    protos = _lib.parse_header()
    assert len(protos) >= 40
    L = _lib.lib()
    for name in protos:
        assert hasattr(L, name), name
    assert L.subgc_version() == 1
    assert L.subgc_arch() == b"gfx950"


def test_host_side_argument_validation_needs_no_gpu():
    L = _lib.lib()
    rc = L.subgc_gemm_f32(0, 1, -1, 4, 4, None, 4, None, 4, None, 4, None, None, 0, None, 1.0, 0, None, None, None, None, 0, None)
    assert rc == -1 and b"negative size" in L.subgc_last_error()
    rc = L.subgc_gemm_f32(1, 1, 4, 4, 4, 1, 4, 1, 4, 1, 4, None, None, 0, None, 1.0, 0, None, None, None, None, 0, None)
    assert rc == -1 and b"transA && transB" in L.subgc_last_error()
    rc = L.subgc_gemm_f32(0, 1, 4, 4, 4, 16, 4, 16, 4, 16, 4, None, None, 0, None, 1.0, 0, None, None, None, None, 64, None)
    assert rc == -1 and b"workspace" in L.subgc_last_error()                   # bytes without a pointer
    # the bf16-operand GEMM: alignment / padding rules are checked on the host
    rc = L.subgc_gemm_bf16(0, 1, 4, 4, 12, 24, 16, 16, 16, 16, 4, None, 0, None, None, 0, None, 1.0, 0, None, None, 0, None)
    assert rc == -1 and b"16-byte aligned" in L.subgc_last_error()
    rc = L.subgc_gemm_bf16(0, 1, 4, 4, 16, 16, 20, 16, 16, 16, 4, None, 0, None, None, 0, None, 1.0, 0, None, None, 0, None)
    assert rc == -1 and b"ld %" in L.subgc_last_error()
    import ctypes
    need = ctypes.c_size_t(1)
    assert L.subgc_gemm_workspace_bytes(640, 4000, 3000, ctypes.byref(need)) == 0 and need.value >= 2 * 640 * 4000 * 4
    assert L.subgc_gemm_bf16_workspace_bytes(1280, 4000, 3000, ctypes.byref(need)) == 0 and need.value >= 1280 * 4000 * 4
    assert L.subgc_gemm_bf16_workspace_bytes(21760, 9488, 1000, ctypes.byref(need)) == 0 and need.value == 0
    rc = L.subgc_row_argmax_f32(None, 4, 2, 4, 4, None, None, None)
    assert rc == -1
    # sub-graph NMS: a sub-graph is a 256-bit node mask (64 * SUBGC_NMS_WORDS), so obj_num = 300 must be an error, not a silently
    # truncated node set (gpn.py:108-150 builds python sets of any size); both entry points, checked before any launch
    rc = L.subgc_subgraph_nms(16, 16, 300, 16, 4, 300, 0.55, 10, 16, 16, 16, 4 * 40, None)
    assert rc == -1 and b"node ids must be < 256" in L.subgc_last_error()
    rc = L.subgc_subgraph_nms_batched(16, 16, 300, 16, 16, 1, 4, 4, 300, 0.55, 10, 16, 16, 16, 4 * 40, None)
    assert rc == -1 and b"node ids must be < 256" in L.subgc_last_error()
    with pytest.raises(_lib.SubgcError):
        _lib.call("subgc_decode_pick", None, 10, 1, 10, 9, 1.0, None, 0, None, None, 4, None, None, None, None, 0, None)


def test_recurrence_struct_mirror_matches_the_library():
    """SubgcRecurrence is mirrored from the header into a ctypes.Structure (one field per declaration): same size as the compiled
    struct, and the entry points validate the block on the host."""
    import ctypes
    L = _lib.lib()
    S = _lib.parse_struct("SubgcRecurrence")
    assert ctypes.sizeof(S) == L.subgc_recurrence_sizeof() and len(S._fields_) > 60
    assert L.subgc_recurrence_fwd(None, None, 0, None) == -1 and b"null argument block" in L.subgc_last_error()
    blk = S()
    blk.S, blk.T, blk.R, blk.A, blk.n_alpha = 4, 2, 8, 4, 3
    assert L.subgc_recurrence_bwd(ctypes.addressof(blk), None) == -1 and b"null step tables" in L.subgc_last_error()
    m = (ctypes.c_int32 * 3)(4, 5, 0)                                        # rows must not grow from step to step
    r0 = (ctypes.c_int64 * 3)(0, 4, 9)
    off = (ctypes.c_int32 * 1)(0)
    blk.m, blk.row0, blk.off = ctypes.addressof(m), ctypes.addressof(r0), ctypes.addressof(off)
    # the forward reads m[T] and row0[T]: both tables must SAY they hold T + 1 entries (round-4 advisor finding)
    assert L.subgc_recurrence_fwd(ctypes.addressof(blk), None, 0, None) == -1 and b"T + 1 = 3 entries (got 0, 0)" in L.subgc_last_error()
    blk.n_m, blk.n_row0 = 3, 2
    assert L.subgc_recurrence_fwd(ctypes.addressof(blk), None, 0, None) == -1 and b"(got 3, 2)" in L.subgc_last_error()
    blk.n_row0 = 3
    assert L.subgc_recurrence_fwd(ctypes.addressof(blk), None, 0, None) == -1 and b"non-increasing" in L.subgc_last_error()


def test_product_path_refuses_cpu_tensors():
    with pytest.raises(_lib.SubgcError):
        ops.row_argmax(torch.zeros(2, 4))


def test_model_api_surface(golden):
    g = golden("subgc_train")
    m = models.setup(g.opt(caption_model="topdown"))
    assert m.gpn is True and m.ss_prob == 0.0 and m.seq_length == 20 and m.vocab_size == 50 and m.num_layers == 2
    h, c = m.init_hidden(3)
    assert h.shape == (2, 3, 48) and c.shape == (2, 3, 48)
    w = g.group("weights")
    assert set(m.state_dict().keys()) == set(w.keys())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    # parameters are views of one flat bucket
    base = m.flat_params.data_ptr()
    for n, p in m.named_parameters():
        assert base <= p.data_ptr() < base + m.flat_params.numel() * 4, n
    with pytest.raises(Exception, match="Caption model not supported"):
        models.setup(argparse.Namespace(caption_model="show_tell"))
    with pytest.raises(ValueError):                             # beam search proper is entered through mode="sample"
        m(None, None, opt={"beam_size": 1}, mode="sample_sentences")
    for name in ("_prepare_feature", "get_logprobs_state", "beam_search", "init_hidden", "sample_images"):
        assert callable(getattr(m, name)), name


@pytest.mark.parametrize("over", [dict(att_hid_size=50), dict(att_hid_size=516), dict(rnn_size=46), dict(rnn_size=2052, input_encoding_size=2052)])
def test_sizes_the_attention_kernels_cannot_serve_are_refused_at_construction(golden, over):
    """shapes outside the vector attention kernels' limits used to fail inside the first backward; they are refused up front."""
    g = golden("subgc_train")
    with pytest.raises(ValueError, match="att_hid_size"):
        models.setup(g.opt(caption_model="topdown", **over))


def test_bf16_storage_needs_8_element_rows(golden):
    g = golden("subgc_train")
    with pytest.raises(ValueError):
        models.setup(g.opt(caption_model="topdown", compute_dtype="bf16", rnn_size=44, input_encoding_size=44))
