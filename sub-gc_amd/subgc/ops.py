"""Thin tensor-level wrappers over the C ABI (include/subgc_hip.h).

Every function takes torch tensors that already live on the MI355X, hands their raw device
pointers to libsubgc_hip.so on torch's CURRENT stream and returns torch tensors.  No autograd
here (see functions.py) and no fallback: CPU tensors raise.
"""
from __future__ import annotations

import gc

import torch

from ._lib import SubgcError, call, lib

RELU, ACCUM = 1, 2
FLOPS = {"on": False, "gemm": 0.0, "gemm_bytes": 0.0, "gemm_calls": 0}


# torch.cuda.current_stream() builds a Stream object through three layers of Python (~6 us; ~190 calls per train step = 1.2 ms
# of the host's 7.5); the raw-handle getters behind it cost a fraction of a microsecond.  Same value, public path as fallback.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream_of(idx):
    return _raw_stream(idx) if _raw_stream is not None else torch.cuda.current_stream(idx).cuda_stream


def _stream():
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


GRAD_WRITES = [0]          # moved by every gradient writer that goes through raw pointers (functions._direct, the decoder backwards)
_WS = {}
WS_MBYTES = 160


def ensure_workspace(device=None, mbytes=None):
    """The split-K scratch of the CURRENT stream of `device`: one buffer per (device, stream), handed to every GEMM call as
    its `workspace, ws_bytes` arguments (the C ABI keeps no global scratch), so concurrent streams never share partial planes.
    Buffers live for the life of the process: captured hipGraphs bake their addresses in."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (idx, _stream_of(idx))
    buf = _WS.get(key)
    if buf is None:
        if torch.cuda.is_current_stream_capturing():
            raise SubgcError("no split-K workspace for the capturing stream: capture through ops.graph_capture(), which brings its own")
        buf = _WS[key] = torch.empty((mbytes or WS_MBYTES) << 20, dtype=torch.uint8, device=torch.device("cuda", idx))
    return buf


_WS_RAW = {}


def _ws(t):
    """(pointer, bytes) of the current stream's workspace on t's device -- the two `workspace, ws_bytes` call arguments."""
    idx = t.device.index
    key = (idx, _stream_of(idx))
    raw = _WS_RAW.get(key)
    if raw is None:
        buf = ensure_workspace(t.device)
        raw = _WS_RAW[key] = (buf.data_ptr(), buf.numel())
    return raw


_CAPTURE_STREAMS = {}
class graph_capture:
    """`with ops.graph_capture(graph, device): ...` = torch.cuda.graph on a dedicated capture stream of that device whose
    split-K workspace was allocated EAGERLY (outside any graph's private pool), so the addresses a captured GEMM bakes in stay
    valid for every later capture and replay.  All graphs captured through this class on one device share that one workspace:
    they must not be REPLAYED concurrently on different streams (the decode paths replay them one after the other on the current
    stream); a caller that wants concurrent replays captures on its own stream with its own `ensure_workspace`."""

    def __init__(self, graph, device):
        dev = torch.device(device)
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        st = _CAPTURE_STREAMS.get(idx)
        if st is None:
            st = _CAPTURE_STREAMS[idx] = torch.cuda.Stream(device=idx)
            with torch.cuda.stream(st):
                ensure_workspace(torch.device("cuda", idx))
        self.ctx = torch.cuda.graph(graph, stream=st)

    def __enter__(self):
        # The cyclic garbage collector must not run INSIDE a capture: a dead cycle that owns an older CUDAGraph (a dropped model's
        # graph cache) frees its private pool when collected, hipFree is illegal while a stream of the process captures
        # (capture_error_mode "global") and the error surfaces in a destructor, i.e. as abort().  torch.cuda.graph stopped
        # collecting on entry (torch.compiler.config.force_cudagraph_gc), so: collect now, keep the collector off until the end.
        self._gc_was_on = gc.isenabled()
        gc.collect()
        gc.disable()
        try:
            return self.ctx.__enter__()
        except BaseException:
            if self._gc_was_on:
                gc.enable()
            raise

    def __exit__(self, *exc):
        try:
            return self.ctx.__exit__(*exc)
        finally:
            if self._gc_was_on:
                gc.enable()


GEMM_MODES = {"f32": 1 << 4, "bf16x3": 2 << 4, "bf16": 3 << 4}       # SUBGC_GEMM_MODE_* bits of the per-call flags


class gemm_mode:
    """`with ops.gemm_mode("bf16x3"): ...` -- arithmetic of the 128x128-tile forms of the fp32-OPERAND GEMM, passed with every
    call (SUBGC_GEMM_MODE_*): "f32" (default), "bf16x3" (fp32 operands split exactly into three bf16 planes, six MFMA terms:
    fp32-grade) or "bf16" (fp32 operands rounded to bf16 in the staging path).  Storage stays fp32 in all three; the
    bf16-STORAGE path of BASELINE configs 3 / 5 is `model.bf16_storage` + bf16 tensors through `gemm` (subgc_gemm_bf16)."""

    current = "f32"

    def __init__(self, mode):
        if mode not in GEMM_MODES:
            raise SubgcError(f"unknown GEMM mode {mode!r}")
        self.mode = mode

    def __enter__(self):
        self.prev = gemm_mode.current
        gemm_mode.current = self.mode

    def __exit__(self, *exc):
        gemm_mode.current = self.prev


def set_gemm_mode(mode):
    if mode not in GEMM_MODES:
        raise SubgcError(f"unknown GEMM mode {mode!r}")
    gemm_mode.current = mode


class debug_bounds:
    """`with ops.debug_bounds(): ...` -- the library's debug bounds mode (subgc_debug_bounds): every entry point that consumes an index
    tensor it did not produce (rel_ind, gpn_obj_ind, node lists, word ids, criterion targets, class ids) validates it on the device before
    its kernels run and raises SubgcError naming the tensor, the first bad position and its value -- the way the reference fails on a bad
    loader tensor (IndexError; the asserts of gpn.py:117-118).  One launch + one stream synchronisation per checked tensor: debugging only."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        self.prev = lib().subgc_debug_bounds(int(self.on))
        return self

    def __exit__(self, *exc):
        lib().subgc_debug_bounds(int(self.prev))


class gemm_tune:
    """Measurement-script switches of the GEMM dispatch, handed over per call in `flags` (the library keeps no tunable state):
    `with ops.gemm_tune(no_splitk=True, no_skinny=True): ...` for subgc_gemm_f32, `tile=128|256|"p8"`, `no_p8`, `splits=n` for subgc_gemm_bf16."""

    f32_bits = 0
    b16_bits = 0

    TILE_BITS = {0: 0, 128: 64, 256: 128, "p8": 1 << 13}                      # SUBGC_GEMM_TILE128 / TILE256 / TILE_P8

    def __init__(self, no_splitk=False, no_skinny=False, tile=0, no_p8=False, splits=0):
        self.f32 = (64 if no_splitk else 0) | (128 if no_skinny else 0)      # SUBGC_GEMM_NO_SPLITK | SUBGC_GEMM_NO_SKINNY
        self.b16 = self.TILE_BITS[tile] | ((1 << 14) if no_p8 else 0) | ((splits & 15) << 8)    # ... | SUBGC_GEMM_NO_P8 | SUBGC_GEMM_SPLITS(n)

    def __enter__(self):
        self.prev = (gemm_tune.f32_bits, gemm_tune.b16_bits)
        gemm_tune.f32_bits, gemm_tune.b16_bits = self.f32, self.b16

    def __exit__(self, *exc):
        gemm_tune.f32_bits, gemm_tune.b16_bits = self.prev


BF16 = torch.bfloat16


def is_b16(t):
    return t is not None and t.dtype == BF16


def empty_b16(rows, cols, device, zero=False):
    """[rows, cols] bf16 view whose leading dimension is padded to a multiple of 8 elements (16-byte row segments for the GEMM)."""
    pad = (cols + 7) // 8 * 8
    if zero:                                                  # bf16 zero is the all-zero bit pattern: fill the fp32 view
        buf = fill_(torch.empty(max(rows * pad // 2, 1), device=device, dtype=torch.float32), 0.0).view(BF16)[: rows * pad].view(rows, pad)
    else:
        buf = torch.empty(rows, pad, device=device, dtype=BF16)
    return buf[:, :cols] if pad != cols else buf


def act_buffer(shape, device, bf16, zero=False):
    """A buffer that only GEMMs consume: bf16 (last dim must be a multiple of 8 so that every [.., cols] slice keeps 16-byte
    rows) or fp32."""
    if not bf16:
        return zeros(*shape, device=device) if zero else torch.empty(*shape, device=device, dtype=torch.float32)
    n = 1
    for d in shape:
        n *= d
    if shape[-1] % 8:
        raise SubgcError(f"bf16 activation buffers need a last dimension that is a multiple of 8, got {tuple(shape)}")
    if zero:
        return fill_(torch.empty(max(n // 2, 1), device=device, dtype=torch.float32), 0.0).view(BF16)[:n].view(*shape)
    return torch.empty(*shape, device=device, dtype=BF16)


# Row pitch of the activation / weight-snapshot buffers the decoder allocates for its recurrent GEMMs, in elements (0 = unpadded).
# bf16: a K-contiguous operand tile is 32 elements = 64 bytes per row; with R = 1000 the natural pitches (2000 / 4000 / 6000 bytes) put
# half of those segments across two 128-byte lines, i.e. twice the L2 requests per LDS-DMA tile (tools/gemm_bf16_bench.py --pad 64:
# 1280 x 4000 x 2000 52 -> 40 us).  Every kernel that writes these buffers takes a leading dimension, so the pad is free.
PITCH = {"bf16": 64, "f32": 0}


def pitch_of(cols, bf16):
    q = PITCH["bf16" if bf16 else "f32"]
    return (cols + q - 1) // q * q if q else cols


def act_padded(shape, device, bf16, zero_rows=0):
    """`act_buffer` with the last dimension's pitch rounded up to PITCH (a [..., cols] view of a [..., pitch] buffer); the first
    `zero_rows` rows (of the flattened leading dimensions) are zeroed, pad columns included."""
    *lead, cols = shape
    ldp = pitch_of(cols, bf16)
    if bf16 and cols % 8:
        raise SubgcError(f"bf16 activation buffers need a last dimension that is a multiple of 8, got {tuple(shape)}")
    base = torch.empty(*lead, ldp, device=device, dtype=BF16 if bf16 else torch.float32)
    if zero_rows:
        z = base.view(-1, ldp)[:zero_rows]
        fill_(z.view(torch.float32) if bf16 else z, 0.0)
    return base[..., :cols] if ldp != cols else base


def flat_rows(t3):
    """[T, S, C] (unit inner stride, stride(0) == S * stride(1)) -> the [T * S, C] view with the same row pitch."""
    T, S, C = t3.shape
    if t3.stride(2) != 1 or (T > 1 and t3.stride(0) != S * t3.stride(1)):
        raise SubgcError(f"flat_rows: not a row-pitched [T, S, C] view: {tuple(t3.shape)} / {t3.stride()}")
    return torch.as_strided(t3, (T * S, C), (t3.stride(1), 1), t3.storage_offset())


def cast_bf16(x, out=None, m_dev=None):
    """bf16 copy of a 2-D fp32 view (subgc_cast_f32_bf16); the destination's columns are padded with zeros to a multiple of 8
    (a K-contiguous GEMM operand needs K % 8 == 0).  Returns the [rows, cols_pad] bf16 tensor."""
    rows, cols = x.shape
    pad = (cols + 7) // 8 * 8
    if out is None:
        out = torch.empty(rows, pad, device=x.device, dtype=BF16)
    call("subgc_cast_f32_bf16", _ptr(x, torch.float32), ld(x), _ptr(out, BF16), ld(out), rows, cols, out.size(1), _ptr(m_dev, torch.int32), _stream())
    return out


def as_b16(x, m_dev=None):
    """The GEMM-operand form of a 2-D activation: itself when already bf16, else a bf16 copy ([rows, cols] view, padded ld)."""
    if is_b16(x):
        return x
    y = cast_bf16(x, m_dev=m_dev)
    return y[:, : x.size(1)] if y.size(1) != x.size(1) else y


def transpose_bf16(x, out=None):
    """out[c, r] = bf16(x[r, c]) (subgc_transpose_f32_bf16): the W^T snapshots; rows of `out` padded to a multiple of 8 elements."""
    rows, cols = x.shape
    if out is None:
        out = torch.zeros(cols, (rows + 7) // 8 * 8, device=x.device, dtype=BF16)[:, :rows]
    call("subgc_transpose_f32_bf16", _ptr(x, torch.float32), ld(x), _ptr(out, BF16), ld(out), rows, cols, _stream())
    return out


def _ptr(t, dtype=None):
    if t is None:
        return None
    if not t.is_cuda:
        raise SubgcError("subgc ops need device tensors (the HIP path has no CPU fallback)")
    if dtype is not None and t.dtype != dtype:
        raise SubgcError(f"expected {dtype}, got {t.dtype}")
    return t.data_ptr()


def ld(t):
    """Leading dimension of a 2-D row-major view with unit inner stride."""
    if t.dim() != 2 or (t.size(1) > 1 and t.stride(1) != 1):
        raise SubgcError(f"need a 2-D tensor with unit inner stride, got {tuple(t.shape)} / {t.stride()}")
    if t.size(0) > 1 and t.stride(0) < t.size(1):
        raise SubgcError(f"overlapping rows: shape {tuple(t.shape)} stride {t.stride()}")
    return max(t.stride(0), t.size(1))


def gemm(a, b, out, *, ta=False, tb=False, bias=None, add=None, keep=None, keep_scale=1.0, relu=False,
         accum=False, a_rows=None, c_rows=None, m_dev=None, out16=None):
    """out = epilogue(op(a) @ op(b));  a, b, out are 2-D row-major views (any leading dim).
    fp32 a, b -> subgc_gemm_f32 (arithmetic = the current `gemm_mode`); bf16 a, b -> subgc_gemm_bf16: `out` may then be fp32 or
    bf16 and `out16` an additional bf16 destination of the same result (for consumers that are GEMMs themselves)."""
    if is_b16(a) or is_b16(b):
        return _gemm_b16(a, b, out, ta, tb, bias, add, keep, keep_scale, relu, accum, a_rows, c_rows, m_dev, out16)
    if out16 is not None:
        raise SubgcError("out16 needs bf16 operands")
    M = a.size(1) if ta else a.size(0)
    K = a.size(0) if ta else a.size(1)
    N = b.size(0) if tb else b.size(1)
    Kb = b.size(1) if tb else b.size(0)
    if a_rows is not None:
        M = a_rows.numel()
    if K != Kb or (c_rows is None and out.size(0) < M) or out.size(1) != N:
        raise SubgcError(f"gemm shape mismatch: op(a)=[{M},{K}] op(b)=[{Kb},{N}] out={tuple(out.shape)}")
    if FLOPS["on"]:                      # bench.py's untimed accounting step: exact, ragged-aware algorithmic FLOPs
        r = int(m_dev.item()) if m_dev is not None else None
        me, ke = (M, min(K, r)) if (ta and r is not None) else ((min(M, r) if r is not None else M), K)
        FLOPS["gemm"] += 2.0 * me * N * ke
        # algorithmic HBM bytes of the same call: each operand read once, the result written once (+ read when accumulated into)
        FLOPS["gemm_bytes"] += 4.0 * (me * ke + ke * N + me * N * (2 if (accum or add is not None) else 1))
        FLOPS["gemm_calls"] += 1
    call("subgc_gemm_f32", int(ta), int(tb), M, N, K, _ptr(a, torch.float32), ld(a), _ptr(b, torch.float32), ld(b),
         _ptr(out, torch.float32), ld(out), _ptr(bias), _ptr(add), ld(add) if add is not None else 0,
         _ptr(keep, torch.uint8), float(keep_scale), (RELU if relu else 0) | (ACCUM if accum else 0) | GEMM_MODES[gemm_mode.current] | gemm_tune.f32_bits,
         _ptr(a_rows, torch.int32), _ptr(c_rows, torch.int32), _ptr(m_dev, torch.int32), *_ws(a), _stream())
    return out


def _gemm_b16(a, b, out, ta, tb, bias, add, keep, keep_scale, relu, accum, a_rows, c_rows, m_dev, out16):
    if not (is_b16(a) and is_b16(b)):
        raise SubgcError(f"bf16 GEMM needs both operands in bf16, got {a.dtype} / {b.dtype}")
    if a_rows is not None or c_rows is not None:
        raise SubgcError("the bf16 GEMM has no gathered / scattered row forms")
    M = a.size(1) if ta else a.size(0)
    K = a.size(0) if ta else a.size(1)
    N = b.size(0) if tb else b.size(1)
    Kb = b.size(1) if tb else b.size(0)
    if K != Kb or out.size(0) < M or out.size(1) != N:
        raise SubgcError(f"gemm shape mismatch: op(a)=[{M},{K}] op(b)=[{Kb},{N}] out={tuple(out.shape)}")
    c32, c16 = (None, out) if is_b16(out) else (out, out16)
    if c16 is not None and (c16.size(0) < M or c16.size(1) != N or not is_b16(c16)):
        raise SubgcError("out16 must be a bf16 [M, N] view")
    if FLOPS["on"]:
        r = int(m_dev.item()) if m_dev is not None else None
        me, ke = (M, min(K, r)) if (ta and r is not None) else ((min(M, r) if r is not None else M), K)
        FLOPS["gemm"] += 2.0 * me * N * ke
        FLOPS["gemm_bytes"] += 2.0 * (me * ke + ke * N) + me * N * ((4.0 * (2 if (accum or add is not None) else 1) if c32 is not None else 0.0)
                                                                   + (2.0 if c16 is not None else 0.0))
        FLOPS["gemm_calls"] += 1
    call("subgc_gemm_bf16", int(ta), int(tb), M, N, K, _ptr(a, BF16), ld(a), _ptr(b, BF16), ld(b), _ptr(c32, torch.float32),
         ld(c32) if c32 is not None else 0, _ptr(c16, BF16), ld(c16) if c16 is not None else 0, _ptr(bias, torch.float32), _ptr(add, torch.float32),
         ld(add) if add is not None else 0, _ptr(keep, torch.uint8), float(keep_scale), (RELU if relu else 0) | (ACCUM if accum else 0) | gemm_tune.b16_bits,
         _ptr(m_dev, torch.int32), *_ws(a), _stream())
    return out


PAIR_LAUNCHES = True      # the two same-shape products of a GCN unit pair in one launch (subgc_gemm_bf16_pair); False: two subgc_gemm_bf16 calls


def gemm_pair(a1, a2, b1, b2, out1, out2, *, ta=False, tb=False, bias1=None, bias2=None, relu=False, accum=False):
    """out_i = epilogue(op(a_i) @ op(b_i)) for two problems of the same shape in ONE launch (subgc_gemm_bf16_pair / subgc_gemm_f32_pair);
    shapes, leading dimensions or storage types that differ run as two `gemm` calls."""
    b16 = is_b16(a1)
    same = (PAIR_LAUNCHES and all(is_b16(t) == b16 for t in (a2, b1, b2)) and a1.shape == a2.shape and b1.shape == b2.shape and out1.shape == out2.shape
            and out1.dtype == out2.dtype and ld(a1) == ld(a2) and ld(b1) == ld(b2) and ld(out1) == ld(out2) and (bias1 is None) == (bias2 is None)
            and (b16 or (not relu and out1.dtype == torch.float32 and a1.dtype == torch.float32 and b1.dtype == torch.float32)))
    if not same:
        gemm(a1, b1, out1, ta=ta, tb=tb, bias=bias1, relu=relu, accum=accum)
        gemm(a2, b2, out2, ta=ta, tb=tb, bias=bias2, relu=relu, accum=accum)
        return out1, out2
    M = a1.size(1) if ta else a1.size(0)
    K = a1.size(0) if ta else a1.size(1)
    N = b1.size(0) if tb else b1.size(1)
    if K != (b1.size(1) if tb else b1.size(0)) or out1.size(0) < M or out1.size(1) != N:
        raise SubgcError(f"gemm_pair shape mismatch: op(a)=[{M},{K}] op(b)=[..,{N}] out={tuple(out1.shape)}")
    o16 = is_b16(out1)
    if FLOPS["on"]:
        esz = 2.0 if b16 else 4.0
        FLOPS["gemm"] += 4.0 * M * N * K
        FLOPS["gemm_bytes"] += 2.0 * (esz * (M * K + K * N) + M * N * (2.0 if o16 else 4.0 * (2 if accum else 1)))
        FLOPS["gemm_calls"] += 1
    if not b16:
        call("subgc_gemm_f32_pair", int(ta), int(tb), M, N, K, _ptr(a1, torch.float32), _ptr(a2, torch.float32), ld(a1), _ptr(b1, torch.float32),
             _ptr(b2, torch.float32), ld(b1), _ptr(out1, torch.float32), _ptr(out2, torch.float32), ld(out1), _ptr(bias1, torch.float32),
             _ptr(bias2, torch.float32), (ACCUM if accum else 0) | GEMM_MODES[gemm_mode.current] | gemm_tune.f32_bits, *_ws(a1), _stream())
        return out1, out2
    call("subgc_gemm_bf16_pair", int(ta), int(tb), M, N, K, _ptr(a1, BF16), _ptr(a2, BF16), ld(a1), _ptr(b1, BF16), _ptr(b2, BF16), ld(b1),
         None if o16 else _ptr(out1, torch.float32), None if o16 else _ptr(out2, torch.float32), 0 if o16 else ld(out1),
         _ptr(out1, BF16) if o16 else None, _ptr(out2, BF16) if o16 else None, ld(out1) if o16 else 0, _ptr(bias1, torch.float32), _ptr(bias2, torch.float32),
         (RELU if relu else 0) | (ACCUM if accum else 0) | gemm_tune.b16_bits, *_ws(a1), _stream())
    return out1, out2


def wgrad(dy, x, dW, db, accum=False, db_accum=False, m_dev=None):
    """Weight AND bias gradient of one linear layer in one call: dW (+)= dy^T x, db (+)= column sums of dy (subgc_gemm_f32_wgrad /
    subgc_gemm_bf16_wgrad: the workgroups of dW's tile column 0 add up the dy tiles they stage anyway -- no second read of dy, no
    column-sum launches).  dy [rows, M], x [rows, N] (both fp32 or both bf16), dW fp32 [M, N], db fp32 [M]; m_dev bounds the rows."""
    K, M = dy.shape
    N = x.size(1)
    if x.size(0) != K or dW.size(0) != M or dW.size(1) != N or db.numel() != M or not db.is_contiguous():
        raise SubgcError(f"wgrad shape mismatch: dy={tuple(dy.shape)} x={tuple(x.shape)} dW={tuple(dW.shape)} db={tuple(db.shape)}")
    b16 = is_b16(dy)
    if b16 != is_b16(x):
        raise SubgcError(f"wgrad needs both operands in one storage type, got {dy.dtype} / {x.dtype}")
    if FLOPS["on"]:
        ke = min(K, int(m_dev.item())) if m_dev is not None else K
        FLOPS["gemm"] += 2.0 * M * N * ke
        FLOPS["gemm_bytes"] += (2.0 if b16 else 4.0) * (M * ke + ke * N) + 4.0 * M * N * (2 if accum else 1)
        FLOPS["gemm_calls"] += 1
    if b16:
        call("subgc_gemm_bf16_wgrad", M, N, K, _ptr(dy, BF16), ld(dy), _ptr(x, BF16), ld(x), _ptr(dW, torch.float32), ld(dW), _ptr(db, torch.float32),
             (ACCUM if accum else 0) | gemm_tune.b16_bits, int(db_accum), _ptr(m_dev, torch.int32), *_ws(dy), _stream())
    else:
        call("subgc_gemm_f32_wgrad", M, N, K, _ptr(dy, torch.float32), ld(dy), _ptr(x, torch.float32), ld(x), _ptr(dW, torch.float32), ld(dW),
             _ptr(db, torch.float32), (ACCUM if accum else 0) | GEMM_MODES[gemm_mode.current] | gemm_tune.f32_bits, int(db_accum),
             _ptr(m_dev, torch.int32), *_ws(dy), _stream())
    return dW, db


def colsum_set(xs, outs, accumulate=False):
    """Column sums of several matrices: bf16 ones of one shape and leading dimension go three at a time through subgc_colsum_bf16_set (two
    launches per group instead of two per matrix); everything else through `colsum`."""
    groups = {}
    for x, o in zip(xs, outs):
        key = (tuple(x.shape), ld(x)) if (is_b16(x) and PAIR_LAUNCHES) else id(x)
        groups.setdefault(key, []).append((x, o))
    for key, items in groups.items():
        while items:
            chunk, items = items[:3], items[3:]
            if len(chunk) == 1 or not isinstance(key, tuple):
                for x, o in chunk:
                    colsum(x, out=o, accumulate=accumulate)
                continue
            x0 = chunk[0][0]
            px = [_ptr(x, BF16) for x, _ in chunk] + [None] * (3 - len(chunk))
            po = [_ptr(o, torch.float32) for _, o in chunk] + [None] * (3 - len(chunk))
            call("subgc_colsum_bf16_set", len(chunk), px[0], px[1], px[2], ld(x0), x0.size(0), x0.size(1), po[0], po[1], po[2], int(accumulate), *_ws(x0), _stream())


def colsum(x, out=None, accumulate=False, m_dev=None):
    out = torch.empty(x.size(1), device=x.device, dtype=torch.float32) if out is None else out
    if is_b16(x):
        call("subgc_colsum_bf16", _ptr(x, BF16), ld(x), x.size(0), x.size(1), _ptr(out, torch.float32), int(accumulate),
             _ptr(m_dev, torch.int32), *_ws(x), _stream())
        return out
    call("subgc_colsum_f32", _ptr(x, torch.float32), ld(x), x.size(0), x.size(1), _ptr(out), int(accumulate),
         _ptr(m_dev, torch.int32), *_ws(x), _stream())
    return out


def row_argmax(x, skip=0, want_val=False, i32=False):
    rows, cols = x.shape
    if i32:                              # class ids that index an embedding table: int32 rows for gather_rows / scatter_add_rows
        idx = torch.empty(rows, device=x.device, dtype=torch.int32)
        call("subgc_row_argmax_i32", _ptr(x, torch.float32), ld(x), rows, cols, skip, _ptr(idx), _stream())
        return idx
    idx = torch.empty(rows, device=x.device, dtype=torch.int64)
    val = torch.empty(rows, device=x.device, dtype=torch.float32) if want_val else None
    call("subgc_row_argmax_f32", _ptr(x, torch.float32), ld(x), rows, cols, skip, _ptr(idx), _ptr(val), _stream())
    return (idx, val) if want_val else idx


def csr_build(rel_ind, N):
    B, K, _ = rel_ind.shape
    rel_ind = rel_ind.contiguous()
    ptr = torch.empty(2, B, N + 1, device=rel_ind.device, dtype=torch.int32)
    edges = torch.empty(2, B, K, device=rel_ind.device, dtype=torch.int32)
    call("subgc_csr_build", _ptr(rel_ind, torch.int64), B, K, N, _ptr(ptr), _ptr(edges), _stream())
    return ptr, edges


def gcn_nodes_fwd(F0, F1, ptr, edges, skip, B, N, K, L, want_act=True):
    out = torch.empty(B, N, L, device=F0.device, dtype=torch.float32)
    act = torch.empty(B, N, L, device=F0.device, dtype=torch.uint8) if want_act else None
    call("subgc_gcn_nodes_fwd", _ptr(F0), _ptr(F1), _ptr(ptr), _ptr(edges), _ptr(skip), _ptr(out), _ptr(act), B, N, K, L, _stream())
    return out, act


def gcn_nodes_bwd(dX, act, rel_ind, ptr, B, N, K, L, bf16=False):
    dF0 = torch.empty(B, K, L, device=dX.device, dtype=BF16 if bf16 else torch.float32)
    dF1 = torch.empty_like(dF0)
    call("subgc_gcn_nodes_bwd", _ptr(dX), _ptr(act), _ptr(rel_ind), _ptr(ptr), _ptr(dF0), _ptr(dF1), int(bf16), B, N, K, L, _stream())
    return dF0, dF1


def gcn_edges_fwd(F2, F3, rel_ind, skip, B, N, K, L):
    out = torch.empty(B, K, L, device=F2.device, dtype=torch.float32)
    call("subgc_gcn_edges_fwd", _ptr(F2), _ptr(F3), _ptr(rel_ind), _ptr(skip), _ptr(out), B, N, K, L, _stream())
    return out


def gcn_edges_bwd(dP, F2, F3, ptr, edges, B, N, K, L):
    dF2 = torch.empty(B, N, L, device=dP.device, dtype=torch.float32)
    dF3 = torch.empty_like(dF2)
    call("subgc_gcn_edges_bwd", _ptr(dP), _ptr(F2), _ptr(F3), _ptr(ptr), _ptr(edges), _ptr(dF2), _ptr(dF3), B, N, K, L, _stream())
    return dF2, dF3


def bn_fwd(x, gamma, beta, running_mean, running_var, training, momentum=0.1, eps=1e-5):
    M, C = x.shape
    y = torch.empty_like(x)
    sm = torch.empty(C, device=x.device, dtype=torch.float32) if training else None
    sr = torch.empty(C, device=x.device, dtype=torch.float32) if training else None
    call("subgc_bn_fwd", _ptr(x), _ptr(y), M, C, _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var),
         _ptr(sm), _ptr(sr), int(training), float(momentum), float(eps), *_ws(x), _stream())
    return y, sm, sr


def bn_bwd(dy, x, gamma, sm, sr):
    M, C = x.shape
    dx = torch.empty_like(x)
    dg = torch.empty(C, device=x.device, dtype=torch.float32)
    db = torch.empty_like(dg)
    call("subgc_bn_bwd", _ptr(dy), _ptr(x), _ptr(gamma), _ptr(sm), _ptr(sr), _ptr(dx), _ptr(dg), _ptr(db), M, C, *_ws(x), _stream())
    return dx, dg, db


def bn_stats(x, gamma, beta, running_mean, running_var, training, momentum=0.1, eps=1e-5):
    """Batch statistics of the raw unit output x [M, C] (fp32 or bf16) -> (aff [3, C] = mean | gamma * rstd | beta, rstd [C]); the
    consumer kernels normalise on load (subgc_bn_stats).  Eval mode: aff from the running statistics."""
    M, C = x.shape
    aff = torch.empty(3, C, device=x.device, dtype=torch.float32)
    rstd = torch.empty(C, device=x.device, dtype=torch.float32)
    if x.stride(0) != C or x.stride(1) != 1:
        raise SubgcError("bn_stats: contiguous rows needed")
    call("subgc_bn_stats", _ptr(x), int(is_b16(x)), M, C, _ptr(gamma, torch.float32), _ptr(beta, torch.float32), _ptr(running_mean, torch.float32),
         _ptr(running_var, torch.float32), _ptr(aff), _ptr(rstd), int(training), float(momentum), float(eps), *_ws(x), _stream())
    return aff, rstd


def bn_stats_pair(x0, x1, gamma0, beta0, rm0, rv0, gamma1, beta1, rm1, rv1, training, momentum=0.1, eps=1e-5):
    """`bn_stats` for the two units a fused aggregation consumes (same shape and storage type, training mode) in two launches instead of
    four (subgc_bn_stats_pair) -> ((aff0, rstd0), (aff1, rstd1))."""
    if (not training or not PAIR_LAUNCHES or x0.shape != x1.shape or x0.dtype != x1.dtype or rm0 is None or rm1 is None):
        return bn_stats(x0, gamma0, beta0, rm0, rv0, training, momentum, eps), bn_stats(x1, gamma1, beta1, rm1, rv1, training, momentum, eps)
    M, C = x0.shape
    if x0.stride(0) != C or x0.stride(1) != 1 or x1.stride(0) != C or x1.stride(1) != 1:
        raise SubgcError("bn_stats_pair: contiguous rows needed")
    aff0, aff1 = torch.empty(3, C, device=x0.device, dtype=torch.float32), torch.empty(3, C, device=x0.device, dtype=torch.float32)
    rstd0, rstd1 = torch.empty(C, device=x0.device, dtype=torch.float32), torch.empty(C, device=x0.device, dtype=torch.float32)
    f32 = torch.float32
    call("subgc_bn_stats_pair", _ptr(x0), _ptr(x1), int(is_b16(x0)), M, C, _ptr(gamma0, f32), _ptr(gamma1, f32), _ptr(beta0, f32), _ptr(beta1, f32),
         _ptr(rm0, f32), _ptr(rm1, f32), _ptr(rv0, f32), _ptr(rv1, f32), _ptr(aff0), _ptr(aff1), _ptr(rstd0), _ptr(rstd1), float(momentum), float(eps),
         *_ws(x0), _stream())
    return (aff0, rstd0), (aff1, rstd1)


def bn_bwd_fused_pair(dy0, dy1, x0, x1, gamma0, gamma1, aff0, aff1, rstd0, rstd1, dg0, dg1, db0, db1, accumulate):
    """`bn_bwd_fused` for both units in two launches instead of four (subgc_bn_bwd_fused_pair) -> (dx0, dx1)."""
    M, C = x0.shape
    dx0, dx1 = torch.empty_like(x0), torch.empty_like(x1)
    f32 = torch.float32
    call("subgc_bn_bwd_fused_pair", _ptr(dy0, f32), _ptr(dy1, f32), _ptr(x0), _ptr(x1), int(is_b16(x0)), M, C, _ptr(gamma0, f32), _ptr(gamma1, f32),
         _ptr(aff0, f32), _ptr(aff1, f32), _ptr(rstd0, f32), _ptr(rstd1, f32), _ptr(dx0), _ptr(dx1), int(is_b16(dx0)), _ptr(dg0, f32), _ptr(dg1, f32),
         _ptr(db0, f32), _ptr(db1, f32), int(accumulate), *_ws(x0), _stream())
    return dx0, dx1


def bn_bwd_fused(dy, x, gamma, aff, rstd, dgamma, dbeta, accumulate):
    """-> d(x) in x's storage type; dgamma / dbeta [C] written (or added to)."""
    M, C = x.shape
    dx = torch.empty_like(x)
    call("subgc_bn_bwd_fused", _ptr(dy, torch.float32), _ptr(x), int(is_b16(x)), M, C, _ptr(gamma, torch.float32), _ptr(aff, torch.float32),
         _ptr(rstd, torch.float32), _ptr(dx), int(is_b16(dx)), _ptr(dgamma, torch.float32), _ptr(dbeta, torch.float32), int(accumulate), *_ws(x), _stream())
    return dx


def gcn_nodes_fwd_bn(F0, F1, aff0, aff1, ptr, edges, skip, B, N, K, L, want16=False):
    out = torch.empty(B, N, L, device=F0.device, dtype=torch.float32)
    out16 = torch.empty(B, N, L, device=F0.device, dtype=BF16) if want16 else None
    act = torch.empty(B, N, L, device=F0.device, dtype=torch.uint8)
    call("subgc_gcn_nodes_fwd_bn", _ptr(F0), _ptr(F1), int(is_b16(F0)), _ptr(aff0, torch.float32), _ptr(aff1, torch.float32), _ptr(ptr), _ptr(edges),
         _ptr(skip, torch.float32), _ptr(out), _ptr(out16), _ptr(act), B, N, K, L, _stream())
    return out, out16, act


def gcn_edges_fwd_bn(F2, F3, aff2, aff3, rel_ind, skip, B, N, K, L, want16=False):
    out = torch.empty(B, K, L, device=F2.device, dtype=torch.float32)
    out16 = torch.empty(B, K, L, device=F2.device, dtype=BF16) if want16 else None
    call("subgc_gcn_edges_fwd_bn", _ptr(F2), _ptr(F3), int(is_b16(F2)), _ptr(aff2, torch.float32), _ptr(aff3, torch.float32), _ptr(rel_ind, torch.int64),
         _ptr(skip, torch.float32), _ptr(out), _ptr(out16), B, N, K, L, _stream())
    return out, out16


def gcn_edges_bwd_bn(dP, F2, F3, aff2, aff3, ptr, edges, B, N, K, L, bf16=False):
    dF2 = torch.empty(B, N, L, device=dP.device, dtype=BF16 if bf16 else torch.float32)
    dF3 = torch.empty_like(dF2)
    call("subgc_gcn_edges_bwd_bn", _ptr(dP, torch.float32), _ptr(F2), _ptr(F3), int(is_b16(F2)), _ptr(aff2, torch.float32), _ptr(aff3, torch.float32),
         _ptr(ptr), _ptr(edges), _ptr(dF2), _ptr(dF3), int(bf16), B, N, K, L, _stream())
    return dF2, dF3


_IDENT_AFF = {}


def identity_aff(L, device):
    """The (mean | scale | shift) triple of the normalise-on-load aggregation kernels that changes nothing: (0 | 1 | 0)."""
    key = (L, device)
    t = _IDENT_AFF.get(key)
    if t is None:
        t = torch.zeros(3, L, device=device, dtype=torch.float32)
        t[1].fill_(1.0)
        _IDENT_AFF[key] = t
    return t


def _pool_account(denom, G, L, fwd):
    if FLOPS["on"]:                      # untimed accounting step: the valid node rows of every sub-graph, not the padded N
        rows = float(denom[:G].cpu().sum())                  # host-side sum: the accounting step launches nothing of its own
        # fwd: read the member rows, write [max | mean] (+ arg-max);  bwd: read d[max | mean] + arg-max, read-modify-write the member rows
        FLOPS["pool_bytes"] = FLOPS.get("pool_bytes", 0.0) + 4.0 * ((rows * L + 3.0 * G * L) if fwd else (3.0 * G * L + 2.0 * rows * L))


def pool_fwd(X, idx, idx_stride, w, w_g, w_i, denom, img, G, N, L, want_argmax=True, out=None):
    _pool_account(denom, G, L, True)
    out = torch.empty(G, 2 * L, device=X.device, dtype=torch.float32) if out is None else out
    am = torch.empty(G, L, device=X.device, dtype=torch.int32) if want_argmax else None
    call("subgc_subgraph_pool_fwd", _ptr(X), _ptr(idx, torch.int64), idx_stride, _ptr(w, torch.float32), w_g, w_i,
         _ptr(denom, torch.float32), _ptr(img, torch.int32), _ptr(out), _ptr(am), G, N, L, _stream())
    return out, am


def pool_bwd(dout, idx, idx_stride, w, w_g, w_i, denom, img, am, dX, G, N, L):
    _pool_account(denom, G, L, False)
    call("subgc_subgraph_pool_bwd", _ptr(dout), _ptr(idx, torch.int64), idx_stride, _ptr(w), w_g, w_i, _ptr(denom),
         _ptr(img, torch.int32), _ptr(am, torch.int32), _ptr(dX), G, N, L, _stream())
    return dX


def gpn_score_fwd(hid, keep, scale, w2, b2, want_loss=True, out=None):
    G, H = hid.shape
    score = torch.empty(G, 1, device=hid.device, dtype=torch.float32) if out is None else out
    loss = torch.empty((), device=hid.device, dtype=torch.float32) if want_loss else None
    call("subgc_gpn_score_fwd", _ptr(hid), _ptr(keep, torch.uint8), float(scale), _ptr(w2), _ptr(b2), _ptr(score), _ptr(loss), G, H, _stream())
    return score, loss


def gpn_score_bwd(hid, keep, scale, w2, score, dloss):
    G, H = hid.shape
    dhid = torch.empty_like(hid)
    dw2 = torch.empty(1, H, device=hid.device, dtype=torch.float32)
    db2 = torch.empty(1, device=hid.device, dtype=torch.float32)
    call("subgc_gpn_score_bwd", _ptr(hid), _ptr(keep, torch.uint8), float(scale), _ptr(w2), _ptr(score), _ptr(dloss), _ptr(dhid),
         _ptr(dw2), _ptr(db2), G, H, _stream())
    return dhid, dw2, db2


NMS_WORDS = 4


def subgraph_nms(score, idx, lens, thres, max_keep):
    """-> (keep[int64, M] buffer, n_keep int32[1]) both on device; caller syncs to read n_keep."""
    M, N = idx.shape
    keep = torch.empty(max(M, 1), device=score.device, dtype=torch.int64)
    n_keep = torch.zeros(1, device=score.device, dtype=torch.int32)
    scratch = torch.empty(max(M, 1) * (NMS_WORDS * 8 + 8), device=score.device, dtype=torch.uint8)
    call("subgc_subgraph_nms", _ptr(score, torch.float32), _ptr(idx, torch.int64), idx.stride(0), _ptr(lens, torch.int32), M, N,
         float(thres), int(max_keep), _ptr(keep), _ptr(n_keep), _ptr(scratch), scratch.numel(), _stream())
    return keep, n_keep


def subgraph_nms_batched(score, idx, lens, sizes, thres, max_keep, offsets=None, keep=None):
    """NMS of several images in ONE launch.  `sizes`: candidates per image (python ints, consecutive segments of score/idx/lens).
    `offsets` (device int32 [images + 1], optional): the segment offsets when the caller already has them on the device
    (subgc_gpn_test_prep writes them); `keep`: destination (int64, >= total elements).
    -> (keep int64 [total] with image b's kept indices (relative to its segment) at its segment start, n_keep int32 [images])."""
    total, N = idx.shape
    dev = score.device
    offs = [0]
    for n in sizes:
        offs.append(offs[-1] + int(n))
    if offsets is None:
        offsets = upload(offs, torch.int32, dev)                                  # one small upload
    keep = torch.empty(max(total, 1), device=dev, dtype=torch.int64) if keep is None else keep
    n_keep = zero_(torch.empty(len(sizes), device=dev, dtype=torch.int32))
    scratch = torch.empty(max(total, 1) * (NMS_WORDS * 8 + 8), device=dev, dtype=torch.uint8)
    call("subgc_subgraph_nms_batched", _ptr(score, torch.float32), _ptr(idx, torch.int64), idx.stride(0), _ptr(lens, torch.int32),
         _ptr(offsets, torch.int32), len(sizes), total, max(sizes) if sizes else 0, N, float(thres), int(max_keep), _ptr(keep, torch.int64), _ptr(n_keep),
         _ptr(scratch), scratch.numel(), _stream())
    return keep, n_keep, offs


def stack_first(groups):
    """groups: lists of per-image tensors (same shape / dtype within a list, 4- or 8-byte dtypes, contiguous).  -> one stacked tensor
    per list holding block [0] of every image ([I, *shape[1:]]), built by subgc_gather_blocks from ONE uploaded address table."""
    dev = groups[0][0].device
    ptrs, alive = [], []
    for ts in groups:
        for t in ts:
            t = t if t.is_contiguous() else t.contiguous()
            alive.append(t)
            ptrs.append(t.data_ptr())
    table = upload(ptrs, torch.int64, dev)
    outs, o = [], 0
    for ts in groups:
        I, t0 = len(ts), ts[0]
        words = t0[0].numel() * t0.element_size() // 4
        out = torch.empty((I,) + tuple(t0.shape[1:]), device=dev, dtype=t0.dtype)
        call("subgc_gather_blocks", _ptr(table[o:o + I], torch.int64), I, words, out.data_ptr(), _stream())
        outs.append(out)
        o += I
    return outs


def decode_batch_finish(seq, seqlp, bounds):
    """The per-image early break of a batched decode (subgc_decode_batch_finish): zeroes seqlp beyond each image's break step in
    place; `bounds`: python list of row bounds per image.  -> int32 [images, 2] on the device: (break step, stopped at all)."""
    dev = seq.device
    b = upload(bounds, torch.int32, dev)                                            # one small upload
    out = torch.empty(len(bounds) - 1, 2, device=dev, dtype=torch.int32)
    if not (seq.is_contiguous() and seqlp.is_contiguous()):
        raise SubgcError("decode_batch_finish: contiguous seq / seqlp")
    call("subgc_decode_batch_finish", _ptr(seq, torch.int64), _ptr(seqlp, torch.float32), _ptr(b, torch.int32), len(bounds) - 1, seq.size(1), _ptr(out),
         _stream())
    return out


_PINNED = {}


def upload(values, dtype, device):
    """A small host list -> device tensor through a PINNED staging buffer and an asynchronous copy on the current stream.  (A pageable
    `torch.tensor(...).to(device)` makes the host wait until the stream has drained -- in the one-image decode that is the whole previous
    image's token loop.)  Two staging buffers per (device, dtype) alternate; every caller reads something back from the device
    (survivor counts, finished beams) before it comes here a third time; an event per buffer makes that a guarantee."""
    n = len(values)
    key = (str(device), dtype)
    ring = _PINNED.get(key)
    if ring is None or ring[0][0][0].numel() < n:
        for _, busy in (ring[0] if ring is not None else ()):     # growing: the old staging buffers may still be the source of a copy in flight
            if busy is not None:
                busy.synchronize()
        cap = max(256, 1 << (max(n, 1) - 1).bit_length())
        ring = _PINNED[key] = [[[torch.empty(cap, dtype=dtype).pin_memory(), None] for _ in range(2)], 0]
    slots, turn = ring
    ring[1] = turn ^ 1
    buf, busy = slots[turn]
    if busy is not None:
        busy.synchronize()                                     # the copy that last read this staging buffer (long done in every caller's flow)
    stage = buf[:n]
    stage.copy_(torch.tensor(values, dtype=dtype))
    out = stage.to(device, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    slots[turn][1] = ev
    return out


def nms_compact(keep_all, n_keep, offsets32, images, total):
    """-> (keep int64 [total], glob int64 [total]): the NMS survivors in image order (subgc_nms_compact)."""
    dev = keep_all.device
    keep = torch.empty(total, device=dev, dtype=torch.int64)
    glob = torch.empty(total, device=dev, dtype=torch.int64)
    call("subgc_nms_compact", _ptr(keep_all, torch.int64), _ptr(n_keep, torch.int32), _ptr(offsets32, torch.int32), int(images), int(total),
         _ptr(keep), _ptr(glob), _stream())
    return keep, glob


def gpn_test_prep(items, N, device, out=None):
    """The sGPN test branch's input views for a list of images in ONE launch (subgc_gpn_test_prep; gpn.py:84-96 reads counterpart 0
    of the loader's five copies).  items: (row, gpn_obj_ind [5,2,M,N], att_masks [5,2,M,N], gpn_pool_mtx [5,2,M,N,N]) per image.
    One small upload (the segment offsets and the three tensor addresses per image); no torch op touches the loader tensors.
    `out`: namespace with idx / lens capacity buffers to write into (the one-image decode's static buffers).
    -> (idx int64 [G,N], w [G,N], denom fp32 [G], lens int32 [G], img int32 [G], offsets32 int32 [images+1], sizes, keepalive)"""
    sizes = [int(g.size(1) * g.size(2)) for _, g, _, _ in items]
    offs = [0]
    for n in sizes:
        offs.append(offs[-1] + n)
    G, I = offs[-1], len(items)
    alive, words = [], list(offs)
    for row, g, a, p_ in items:
        g, a, p_ = (t if t.is_contiguous() else t.contiguous() for t in (g, a, p_))
        if g.dtype != torch.int64 or a.dtype != torch.float32 or p_.dtype != torch.float32:
            raise SubgcError("gpn_test_prep: gpn_obj_ind int64, att_masks / gpn_pool_mtx fp32")
        alive += [g, a, p_]
        words += [g.data_ptr(), p_.data_ptr(), a.data_ptr(), int(row)]
    table = upload(words, torch.int64, device)                                      # the one upload of the selection phase (pinned, asynchronous)
    e = lambda *sh, dt=torch.float32: torch.empty(*sh, device=device, dtype=dt)
    idx = out.idx[:G] if out is not None else e(max(G, 0), N, dt=torch.int64)
    lens = out.lens[:G] if out is not None else e(G, dt=torch.int32)
    w, denom, img, o32 = e(G, N), e(G), e(G, dt=torch.int32), e(I + 1, dt=torch.int32)
    call("subgc_gpn_test_prep", _ptr(table, torch.int64), I, G, N, _ptr(idx, torch.int64), _ptr(w), _ptr(denom), _ptr(lens, torch.int32), _ptr(img, torch.int32),
         _ptr(o32, torch.int32), _stream())
    return idx, w, denom, lens, img, o32, sizes, (alive, table)


def _words(t):
    """A contiguous tensor of a 4- or 8-byte dtype as its [rows, 4-byte words] float32 view (bit-level row operations)."""
    if not t.is_contiguous() or t.element_size() not in (4, 8):
        raise SubgcError(f"need a contiguous 4- or 8-byte tensor, got {t.dtype} {tuple(t.shape)} / {t.stride()}")
    t2 = t.view(t.size(0), -1) if t.dim() >= 1 and t.numel() else t.reshape(max(t.size(0) if t.dim() else 1, 0), -1)
    return t2.view(torch.float32) if t2.dtype != torch.float32 else t2


def zero_(t):
    """t[...] = 0 for a contiguous tensor of any 4- or 8-byte dtype (all-zero bits; subgc_fill_f32 on the word view)."""
    if t.numel():
        fill_(_words(t.view(-1, 1) if t.dim() == 1 else t), 0.0)
    return t


def copy_(dst, src):
    """Bit copy between contiguous tensors of one dtype and shape (subgc_copy2d_f32 on the word views)."""
    if dst.dtype != src.dtype or dst.shape != src.shape:
        raise SubgcError(f"copy_: {src.dtype} {tuple(src.shape)} -> {dst.dtype} {tuple(dst.shape)}")
    if src.numel():
        a, b = _words(src.view(1, -1)), _words(dst.view(1, -1))
        call("subgc_copy2d_f32", _ptr(a), a.size(1), _ptr(b), b.size(1), 1, a.size(1), 0, _stream())
    return dst


def take_rows(pairs, rows):
    """dst[m] = src[rows[m]] for up to four (src, dst) pairs of any 4- / 8-byte dtype in ONE launch, `rows` int64 on the device
    (subgc_gather_rows_multi_i64): 1-D tensors count as one-column rows."""
    if not 1 <= len(pairs) <= 4:
        raise SubgcError("take_rows takes 1..4 (src, dst) pairs")
    M = rows.numel()
    args = []
    for src, dst in list(pairs) + [(None, None)] * (4 - len(pairs)):
        if src is None:
            args += [None, 0, None, 0, 0]
            continue
        a = _words(src.view(-1, 1) if src.dim() == 1 else src)
        b = _words(dst.view(-1, 1) if dst.dim() == 1 else dst)
        if a.size(1) != b.size(1) or b.size(0) < M:
            raise SubgcError("take_rows: row widths differ or the destination is too short")
        args += [_ptr(a), a.stride(0), _ptr(b), b.stride(0), a.size(1)]
    call("subgc_gather_rows_multi_i64", len(pairs), *args, _ptr(rows, torch.int64), M, _stream())


def pack_rows(lens, idx, img, S, N, off=None):
    dev = lens.device
    off = torch.empty(S, device=dev, dtype=torch.int32) if off is None else off
    total = torch.empty(1, device=dev, dtype=torch.int32)
    src = torch.empty(S * N, device=dev, dtype=torch.int32)
    sent = torch.empty(S * N, device=dev, dtype=torch.int32)
    call("subgc_pack_rows", _ptr(lens, torch.int32), _ptr(idx, torch.int64), idx.stride(0), _ptr(img, torch.int32), S, N,
         _ptr(off), _ptr(total), _ptr(src), _ptr(sent), _stream())
    return off, total, src, sent


def embed_fwd(table, tok, tok_stride, keep, scale, out):
    """out [n, E] fp32 or bf16 (contiguous rows)."""
    n, E = out.shape
    call("subgc_embed_fwd", _ptr(table), _ptr(tok, torch.int64), tok_stride, _ptr(keep, torch.uint8), float(scale), _ptr(out), n, E,
         table.size(0), int(is_b16(out)), _stream())
    return out


def token_rows(table, tok, out, tok_stride=1):
    """out[r] = table[tok[r]] (int64 tokens): the decode-time lookup into a per-token table (subgc_token_rows_f32)."""
    call("subgc_token_rows_f32", _ptr(table, torch.float32), ld(table), _ptr(tok, torch.int64), tok_stride, _ptr(out, torch.float32), ld(out),
         out.size(0), out.size(1), table.size(0), _stream())
    return out


def lstm_gate_perm(R, device):
    """Row order of the permuted gate matrix subgc_lstm_step_skinny reads: row 16*b + 4*g + u <- gate g of unit 4*b + u."""
    p = torch.arange(4 * R, device=device)
    return (p % 16 // 4) * R + (p // 16) * 4 + p % 4


def lstm_step_skinny(x, w_perm, c_prev, c, hs, b0=None, b1=None, add1=None, tok=None, add2=None):
    """One LSTMCell step for <= 16 rows, GEMM + cell update in one launch (subgc_lstm_step_skinny).  `hs`: up to three 2-D
    views that receive h (none may alias x)."""
    S, K = x.shape
    R = c.size(1)
    hs = list(hs) + [None] * (3 - len(hs))
    hp = []
    for h_ in hs:
        hp += [_ptr(h_, torch.float32), ld(h_) if h_ is not None else 0]
    call("subgc_lstm_step_skinny", _ptr(x, torch.float32), ld(x), _ptr(w_perm), ld(w_perm), K, S, R,
         _ptr(add1, torch.float32), ld(add1) if add1 is not None else 0, _ptr(tok, torch.int64), add1.size(0) if tok is not None else 0,
         _ptr(add2, torch.float32), ld(add2) if add2 is not None else 0, _ptr(b0, torch.float32), _ptr(b1, torch.float32),
         _ptr(c_prev, torch.float32), _ptr(c, torch.float32), *hp, int(is_b16(w_perm)), _stream())


PICK_BEST_ELEMS = 16 * 8 * 16       # uint64 slots of one `best` buffer of the fused greedy pick (subgc_hip.h)


def skinny_dual(x1, W1, out1, x2, W2, out2, bias1=None, bias2=None, unperm1_R=0, unperm2_R=0, best=None, lse_part=None):
    """Two weight-streaming products of <= 16 rows in one launch (subgc_skinny_dual): out1 = x1 W1^T (+ bias1) -- or, with `best`, the fused
    arg-max / log-sum-exp partials (packed arg-max by 64-bit atomicMax into `best`, per-workgroup (max, sum exp) into `lse_part`) instead of (or beside) out1 -- and out2 = x2 W2^T (+ bias2).  unperm*_R: W's rows are
    in the permuted LSTM gate order, the result is written gate-major."""
    S = x1.size(0)
    b16 = is_b16(W1)
    if is_b16(W2) != b16:
        raise SubgcError("skinny_dual: both weight matrices fp32 or both bf16")
    call("subgc_skinny_dual", S, _ptr(x1, torch.float32), ld(x1), _ptr(W1), ld(W1), _ptr(bias1, torch.float32), W1.size(0), x1.size(1),
         _ptr(out1, torch.float32), ld(out1) if out1 is not None else 0, int(unperm1_R), _ptr(best, torch.int64), _ptr(lse_part, torch.float32),
         _ptr(x2, torch.float32), ld(x2), _ptr(W2), ld(W2), _ptr(bias2, torch.float32), W2.size(0), x2.size(1), _ptr(out2, torch.float32), ld(out2),
         int(unperm2_R), int(b16), _stream())


def lstm_cell_pick(pre, c_prev, c, hs, b0, b1, table, add2, tok=None, pick=None, best_reset=None):
    """The attention LSTM's cell of a greedy decode step from gate-major pre-activations (subgc_lstm_cell_pick).  `pick` = (best_prev,
    unf_in, unf_out, seq, t_prev, count_out, prev_count), or None with `tok` [S] int64 (step 0)."""
    S, R = c.shape
    hs = list(hs) + [None] * (3 - len(hs))
    hp = []
    for h_ in hs:
        hp += [_ptr(h_, torch.float32), ld(h_) if h_ is not None else 0]
    if pick is not None:
        best_prev, unf_in, unf_out, seq, t_prev, count_out, prev_count = pick
        pk = [_ptr(best_prev, torch.int64), _ptr(unf_in, torch.int32), _ptr(unf_out, torch.int32), _ptr(seq, torch.int64), seq.size(1), int(t_prev),
              _ptr(count_out, torch.int32), _ptr(prev_count, torch.int32), _ptr(best_reset, torch.int64)]
    else:
        pk = [None, None, None, None, 0, 0, None, None, None]
    call("subgc_lstm_cell_pick", _ptr(pre, torch.float32), ld(pre), S, R, _ptr(table, torch.float32), ld(table), _ptr(tok, torch.int64), table.size(0),
         _ptr(add2, torch.float32), ld(add2) if add2 is not None else 0, _ptr(b0, torch.float32), _ptr(b1, torch.float32), _ptr(c_prev, torch.float32),
         _ptr(c, torch.float32), *hp, *pk, _stream())


def pick_file(best_prev, unf_in, unf_out, seq, t_prev, count_out, prev_count):
    call("subgc_pick_file", _ptr(best_prev, torch.int64), _ptr(unf_in, torch.int32), _ptr(unf_out, torch.int32), _ptr(seq, torch.int64), seq.size(0),
         seq.size(1), int(t_prev), _ptr(count_out, torch.int32), _ptr(prev_count, torch.int32), _stream())


def pick_lse_finish(lse_part, V, counts, seqlp):
    S, T = seqlp.shape
    call("subgc_pick_lse_finish", _ptr(lse_part, torch.float32), int(V), S, T, _ptr(counts, torch.int32), _ptr(seqlp, torch.float32), _stream())


def embed_bwd(table, tok, tok_stride, keep, scale, dout, dtable):
    n, E = dout.shape
    call("subgc_embed_bwd", _ptr(table), _ptr(tok, torch.int64), tok_stride, _ptr(keep, torch.uint8), float(scale), _ptr(dout),
         _ptr(dtable), n, E, table.size(0), _stream())
    return dtable


def _same_storage(*ts):
    """1 when the (non-None) h destinations are bf16, 0 when fp32; mixing is an error."""
    kinds = {is_b16(t) for t in ts if t is not None}
    if len(kinds) > 1:
        raise SubgcError("the h destinations of one LSTM call must all be fp32 or all be bf16")
    return int(kinds.pop()) if kinds else 0


def lstm_fwd(g0, g1, g2, b0, b1, c_prev, c, h, h2, keep, scale, hdrop, gates, S, R, rows_h=0, rows_h2=0):
    L = lambda t: ld(t) if t is not None else 0
    call("subgc_lstm_fwd", _ptr(g0), L(g0), _ptr(g1), L(g1), _ptr(g2), L(g2), _ptr(b0), _ptr(b1), _ptr(c_prev), _ptr(c),
         _ptr(h), L(h), _ptr(h2), L(h2), _ptr(keep, torch.uint8), float(scale), _ptr(hdrop), L(hdrop), _ptr(gates), S, R, int(rows_h),
         int(rows_h2), _same_storage(h, h2, hdrop), _stream())


def lstm_fwd_gemm(x, w, pre, g1, g2, b0, b1, c_prev, c, h, h2, keep, scale, hdrop, gates, S, R, rows_h=0, rows_h2=0):
    """gemm(x, w^T) + lstm_fwd with the split-K reduce folded into the cell kernel (subgc_lstm_fwd_gemm); `pre` [S, 4R] is scratch.
    x, w both fp32 or both bf16; h / h2 / hdrop all fp32 or all bf16."""
    L = lambda t: ld(t) if t is not None else 0
    K = x.size(1)
    xb = int(is_b16(x))
    if is_b16(w) != bool(xb):
        raise SubgcError("lstm_fwd_gemm: x and w must have the same storage type")
    if FLOPS["on"]:                      # same accounting as ops.gemm: the product is the same launch family
        eb = 2.0 if xb else 4.0
        FLOPS["gemm"] += 2.0 * S * 4 * R * K
        FLOPS["gemm_bytes"] += eb * (S * K + K * 4 * R) + 4.0 * S * 4 * R
        FLOPS["gemm_calls"] += 1
    call("subgc_lstm_fwd_gemm", _ptr(x), ld(x), _ptr(w), ld(w), K, _ptr(pre, torch.float32), ld(pre),
         _ptr(g1), L(g1), _ptr(g2), L(g2), _ptr(b0), _ptr(b1), _ptr(c_prev), _ptr(c), _ptr(h), L(h), _ptr(h2), L(h2),
         _ptr(keep, torch.uint8), float(scale), _ptr(hdrop), L(hdrop), _ptr(gates), S, R, int(rows_h), int(rows_h2),
         xb | (_same_storage(h, h2, hdrop) << 1), GEMM_MODES[gemm_mode.current], *_ws(x), _stream())


def gemm_planes(a, b, planes, *, ta=False, tb=False):
    """op(a) @ op(b) left as n fp32 partial planes planes[q][M][N] whose sum is the product (subgc_gemm_*_planes: the split-K forms
    without their reduce pass; n = 1 when the dispatch does not split).  `planes`: a flat fp32 buffer of >= 8*M*N elements.
    -> (n, M * N)"""
    import ctypes
    M = a.size(1) if ta else a.size(0)
    K = a.size(0) if ta else a.size(1)
    N = b.size(0) if tb else b.size(1)
    if K != (b.size(1) if tb else b.size(0)):
        raise SubgcError("gemm_planes: shape mismatch")
    n = ctypes.c_int(0)
    if FLOPS["on"]:
        eb = 2.0 if is_b16(a) else 4.0
        FLOPS["gemm"] += 2.0 * M * N * K
        FLOPS["gemm_bytes"] += eb * (M * K + K * N) + 4.0 * M * N
        FLOPS["gemm_calls"] += 1
    if is_b16(a) or is_b16(b):
        if not (is_b16(a) and is_b16(b)):
            raise SubgcError("gemm_planes: both operands bf16 or both fp32")
        call("subgc_gemm_bf16_planes", int(ta), int(tb), M, N, K, _ptr(a, BF16), ld(a), _ptr(b, BF16), ld(b), _ptr(planes, torch.float32), planes.numel() * 4,
             ctypes.byref(n), _stream())
    else:
        call("subgc_gemm_f32_planes", int(ta), int(tb), M, N, K, _ptr(a, torch.float32), ld(a), _ptr(b, torch.float32), ld(b), _ptr(planes, torch.float32),
             planes.numel() * 4, ctypes.byref(n), GEMM_MODES[gemm_mode.current] | gemm_tune.f32_bits, _stream())
    return n.value, M * N


RECURRENCE_IN_C = True        # the train decoder's T-step loops as one C call per direction (subgc_recurrence_fwd / _bwd); False = step by step from Python
_RECUR_T = None


class Recurrence:
    """Argument block of subgc_recurrence_fwd / subgc_recurrence_bwd (include/subgc_hip.h: SubgcRecurrence, mirrored from the header).
    Tensors go in as device pointers, python int lists (`m`, `row0`, `hout_off`, `dhout_off`) as host arrays owned by this object,
    everything else by value; fields not given stay 0 / NULL."""

    HOST = {"m": "c_int32", "row0": "c_int64", "hout_off": "c_int64", "dhout_off": "c_int64"}

    def __init__(self, **fields):
        import ctypes
        global _RECUR_T
        if _RECUR_T is None:
            from ._lib import lib, parse_struct
            _RECUR_T = parse_struct("SubgcRecurrence")
            if ctypes.sizeof(_RECUR_T) != int(lib().subgc_recurrence_sizeof()):
                raise SubgcError("SubgcRecurrence: the ctypes mirror and the library disagree about the struct layout")
        self.st = _RECUR_T()
        self._alive = []
        self.host = {}                                                  # the python lists behind the host-array fields (m, row0, ...)
        self.set(**fields)

    def set(self, **fields):
        import ctypes
        exp = _recurrence_field_types()
        for k, v in fields.items():
            if v is None:
                setattr(self.st, k, 0)                                  # NULL pointer / zero
            elif torch.is_tensor(v):
                if not v.is_cuda:
                    raise SubgcError(f"recurrence: {k} must be a device tensor")
                want = exp.get(k)
                if want is None:
                    raise SubgcError(f"recurrence: {k} is not a pointer field of SubgcRecurrence")
                if want != "any" and v.dtype != want:
                    raise SubgcError(f"recurrence: {k} is declared {want} in subgc_hip.h, got {v.dtype}")
                if want == "any" and v.dtype not in (torch.float32, torch.bfloat16, torch.uint16, torch.int16):
                    raise SubgcError(f"recurrence: {k} must be fp32 or bf16 storage, got {v.dtype}")
                if v.dim() >= 2 and v.stride(-1) != 1:
                    raise SubgcError(f"recurrence: {k} needs unit inner stride (shape {tuple(v.shape)}, stride {v.stride()})")
                if v.dim() <= 1 and not v.is_contiguous():
                    raise SubgcError(f"recurrence: {k} must be contiguous")
                setattr(self.st, k, v.data_ptr())
            elif k in self.HOST:
                arr = (getattr(ctypes, self.HOST[k]) * max(len(v), 1))(*[int(x) for x in v])
                self._alive.append(arr)
                self.host[k] = [int(x) for x in v]
                setattr(self.st, k, ctypes.addressof(arr))
                if k in ("m", "row0"):
                    setattr(self.st, "n_" + k, len(v))                  # the library checks >= T + 1 (it reads m[T] / row0[T])
            else:
                setattr(self.st, k, v)
        return self


_RECUR_TYPES = None


def _recurrence_field_types():
    """field -> expected torch dtype of every POINTER member of SubgcRecurrence, read from the header (`const float*` -> float32,
    `const int32_t*` -> int32, `const uint8_t*` -> uint8, `void*` -> "any": fp32 or bf16 storage by the `bf16` flags)."""
    global _RECUR_TYPES
    if _RECUR_TYPES is None:
        import re
        from ._lib import HEADER
        src = open(HEADER).read()
        src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
        body = re.search(r"typedef\s+struct\s+SubgcRecurrence\s*\{(.*?)\}\s*SubgcRecurrence\s*;", src, flags=re.S).group(1)
        table = {"float": torch.float32, "int32_t": torch.int32, "int64_t": torch.int64, "uint8_t": torch.uint8, "uint16_t": "any", "void": "any"}
        out = {}
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if "*" in decl:
                ty = decl.split("*")[0].replace("const", "").strip()
                out[decl.split("*")[-1].strip()] = table[ty]
        _RECUR_TYPES = out
    return _RECUR_TYPES


def recurrence_fwd(rec, like):
    import ctypes
    call("subgc_recurrence_fwd", ctypes.addressof(rec.st), *_ws(like), _stream())


def recurrence_bwd(rec):
    import ctypes
    call("subgc_recurrence_bwd", ctypes.addressof(rec.st), _stream())



def recurrence_ok():
    """The C loops apply unless a measurement mode needs the per-call Python wrappers (FLOP accounting, GEMM dispatch switches)."""
    return RECURRENCE_IN_C and not FLOPS["on"] and gemm_tune.f32_bits == 0


def lstm_bwd_planes(gates, c_prev, c, srcs, dh_drop, keep, scale, dc, dpre, dc_prev, S, R):
    """lstm_bwd whose d(h) sources are column windows of split-K plane stacks: srcs = up to three (planes, N, col0, n, stride, rows)
    -- `planes` the flat buffer gemm_planes filled ([n][rows][N]), the window starts at column col0; n = 0 entries are skipped."""
    args = []
    srcs = [x for x in srcs if x is not None and x[3] > 0]
    for k in range(3):
        if k < len(srcs):
            buf, N, col0, n, stride, rows = srcs[k]
            args += [buf.data_ptr() + 4 * col0, N, n, stride, rows]
        else:
            args += [None, 0, 0, 0, 0]
    L = lambda t: ld(t) if t is not None else 0
    call("subgc_lstm_bwd_planes", _ptr(gates), _ptr(c_prev), _ptr(c), *args, _ptr(dh_drop), L(dh_drop), _ptr(keep, torch.uint8), float(scale), _ptr(dc),
         _ptr(dpre), _ptr(dc_prev), S, R, int(is_b16(dpre)), _stream())


def lstm_bwd(gates, c_prev, c, dh_a, dh_b, dh_drop, keep, scale, dc, dpre, dc_prev, S, R):
    """dpre [S, 4R] contiguous rows, fp32 or bf16 (the gate gradients only feed GEMMs and bias sums)."""
    L = lambda t: ld(t) if t is not None else 0
    call("subgc_lstm_bwd", _ptr(gates), _ptr(c_prev), _ptr(c), _ptr(dh_a), L(dh_a), _ptr(dh_b), L(dh_b), _ptr(dh_drop), L(dh_drop),
         _ptr(keep, torch.uint8), float(scale), _ptr(dc), _ptr(dpre), _ptr(dc_prev), S, R, int(is_b16(dpre)), _stream())


def _attn_account(lens, S, A, R, passes):
    if FLOPS["on"]:                      # untimed accounting step: algorithmic bytes of the ragged attention sets
        FLOPS["attn_bytes"] = FLOPS.get("attn_bytes", 0.0) + 4.0 * float(lens[:S].cpu().sum()) * (A + R) * passes


def _uv_b16(u, v, A, R):
    """1 when the node features u [rows, A] / v [rows, R] are the bf16 tensors of compute_dtype = bf16 (both, contiguous rows)."""
    b = is_b16(u)
    if b != is_b16(v):
        raise SubgcError("attention: u and v must have the same storage type")
    if b and (ld(u) != A or ld(v) != R):
        raise SubgcError("attention: bf16 node features must have contiguous rows")
    return int(b)


def attn_fwd(u, v, ah, w_a, b_a, off, lens, ctx, alpha, S, A, R, q=None):
    """`q` = (planes, n, stride, bias): the query rows are the n split-K partial planes `gemm_planes` left in `planes` (+ bias); the
    summed rows are written to `ah` (subgc_attn_fwd_q)."""
    _attn_account(lens, S, A, R, 1)      # read u and v rows once
    if q is not None:
        planes, n, stride, bias = q
        call("subgc_attn_fwd_q", _ptr(u), _ptr(v), _ptr(planes, torch.float32), int(n), int(stride), _ptr(bias, torch.float32), _ptr(ah, torch.float32),
             _ptr(w_a), _ptr(b_a), _ptr(off, torch.int32), _ptr(lens, torch.int32), _ptr(ctx), ld(ctx), _ptr(alpha), alpha.size(1) if alpha is not None else 0,
             S, A, R, int(is_b16(ctx)) | (_uv_b16(u, v, A, R) << 1), _stream())
        return
    call("subgc_attn_fwd", _ptr(u), _ptr(v), _ptr(ah), _ptr(w_a), _ptr(b_a), _ptr(off, torch.int32), _ptr(lens, torch.int32),
         _ptr(ctx), ld(ctx), _ptr(alpha), alpha.size(1) if alpha is not None else 0, S, A, R, int(is_b16(ctx)) | (_uv_b16(u, v, A, R) << 1), _stream())


def _dctx_args(dctx):
    """d(ctx) as (pointer, ld, planes, plane stride): a 2-D view, or a window (planes, N, col0, n, stride, rows) of a plane stack."""
    if isinstance(dctx, tuple):
        buf, N, col0, n, stride, _rows = dctx
        return buf.data_ptr() + 4 * col0, N, n, stride
    return _ptr(dctx, torch.float32), ld(dctx), 1, 0


def attn_bwd(u, v, ah, w_a, off, lens, alpha, dctx, dah, du, dv, dw_a, db_a, S, A, R, dctx_keep=None, de_keep=None):
    """dv None: d(v) is deferred to one `attn_dv_accum` after the time loop; `dctx_keep` [S, R] then receives this step's d(ctx) rows.
    `de_keep` [S, n] (then du is not touched): d(u) is deferred likewise to one `attn_du_accum`.
    `dctx`: a 2-D view or a plane-stack window (see _dctx_args): the split-K planes of the data-gradient GEMM are summed on load."""
    _attn_account(lens, S, A, R, 3 if dv is not None else 2)      # read u, v; read-modify-write du (and dv)
    dp, dl, dn, ds = _dctx_args(dctx)
    if de_keep is not None:
        if de_keep.stride(0) != alpha.size(1) or alpha.stride(0) != alpha.size(1):
            raise SubgcError("attn_bwd: de_keep rows must have alpha's pitch")
        call("subgc_attn_bwd_planes_de", _ptr(u), _ptr(v), _ptr(ah), _ptr(w_a), _ptr(off, torch.int32), _ptr(lens, torch.int32), _ptr(alpha),
             alpha.size(1), dp, dl, dn, ds, _ptr(dah), _ptr(de_keep, torch.float32), _ptr(dv), _ptr(dw_a), _ptr(db_a), S, A, R,
             int(is_b16(dah)) | (_uv_b16(u, v, A, R) << 1), _ptr(dctx_keep, torch.float32), ld(dctx_keep) if dctx_keep is not None else 0, _stream())
        return
    call("subgc_attn_bwd_planes", _ptr(u), _ptr(v), _ptr(ah), _ptr(w_a), _ptr(off, torch.int32), _ptr(lens, torch.int32), _ptr(alpha),
         alpha.size(1), dp, dl, dn, ds, _ptr(dah), _ptr(du), _ptr(dv), _ptr(dw_a), _ptr(db_a), S, A, R,
         int(is_b16(dah)) | (_uv_b16(u, v, A, R) << 1), _ptr(dctx_keep, torch.float32), ld(dctx_keep) if dctx_keep is not None else 0, _stream())


def attn_dv_accum(alpha, dctx, step_off, T, off, lens, dv, S, R):
    """d(v) of all time steps in one pass (subgc_attn_dv_accum): alpha [rows, n], dctx [rows, R] hold step t's live sentences as
    rows step_off[t] .. step_off[t+1]-1; dv [sum lens, R] is overwritten."""
    call("subgc_attn_dv_accum", _ptr(alpha, torch.float32), alpha.size(1), _ptr(dctx, torch.float32), ld(dctx), _ptr(step_off, torch.int32), int(T),
         _ptr(off, torch.int32), _ptr(lens, torch.int32), _ptr(dv, torch.float32), S, R, _stream())


def attn_du_accum(u, ah, de, step_off, T, off, lens, w_a, du, S, A):
    """d(u) of all time steps in one pass (subgc_attn_du_accum): ah [rows, A] (the query rows the forward saved), de [rows, n] hold step
    t's live sentences as rows step_off[t] .. step_off[t+1]-1; du [sum lens, A] is overwritten."""
    call("subgc_attn_du_accum", _ptr(u), int(is_b16(u)), _ptr(ah, torch.float32), _ptr(de, torch.float32), de.size(1), _ptr(step_off, torch.int32), int(T),
         _ptr(off, torch.int32), _ptr(lens, torch.int32), _ptr(w_a, torch.float32), _ptr(du, torch.float32), S, A, _stream())


def attn_fwd_group(u, v, ah, w_a, b_a, rows, lens, m, B, g, Nn, ctx, alpha, A, R, q=None):
    """Attention step over per-image shared sets (subgc_attn_fwd_group): ah / ctx / alpha hold the step's rows.  `q`: see attn_fwd."""
    _attn_account_group(B, Nn, A, R, 1)
    if q is not None:
        planes, n, stride, bias = q
        call("subgc_attn_fwd_group_q", _ptr(u), _ptr(v), _ptr(planes, torch.float32), int(n), int(stride), _ptr(bias, torch.float32), _ptr(ah, torch.float32),
             _ptr(w_a), _ptr(b_a), _ptr(rows, torch.int32), _ptr(lens, torch.int32), int(m), B, g, Nn, _ptr(ctx), ld(ctx), _ptr(alpha),
             alpha.size(1) if alpha is not None else 0, A, R, int(is_b16(ctx)) | (_uv_b16(u, v, A, R) << 1), _stream())
        return
    call("subgc_attn_fwd_group", _ptr(u), _ptr(v), _ptr(ah, torch.float32), _ptr(w_a), _ptr(b_a), _ptr(rows, torch.int32), _ptr(lens, torch.int32), int(m),
         B, g, Nn, _ptr(ctx), ld(ctx), _ptr(alpha), alpha.size(1) if alpha is not None else 0, A, R, int(is_b16(ctx)) | (_uv_b16(u, v, A, R) << 1), _stream())


def attn_bwd_group(u, v, ah, w_a, rows, lens, m, B, g, Nn, alpha, dctx, dah, du, dw_a, db_a, A, R, dctx_keep):
    _attn_account_group(B, Nn, A, R, 2)
    dp, dl, dn, ds = _dctx_args(dctx)
    call("subgc_attn_bwd_group", _ptr(u), _ptr(v), _ptr(ah, torch.float32), _ptr(w_a), _ptr(rows, torch.int32), _ptr(lens, torch.int32), int(m), B, g, Nn,
         _ptr(alpha, torch.float32), alpha.size(1), dp, dl, _ptr(dah), _ptr(du, torch.float32), _ptr(dw_a), _ptr(db_a), A, R,
         int(is_b16(dah)) | (_uv_b16(u, v, A, R) << 1), _ptr(dctx_keep, torch.float32), ld(dctx_keep) if dctx_keep is not None else 0, dn, ds,
         du.size(0) if du.dim() == 3 else 1, du.stride(0) if du.dim() == 3 else 0, _stream())


def attn_group_du_planes(g):
    from ._lib import lib
    return int(lib().subgc_attn_group_du_planes(int(g)))


def attn_dv_accum_group(alpha, dctx, step_off, T, rows, B, g, Nn, dv, R):
    call("subgc_attn_dv_accum_group", _ptr(alpha, torch.float32), alpha.size(1), _ptr(dctx, torch.float32), ld(dctx), _ptr(step_off, torch.int32), int(T),
         _ptr(rows, torch.int32), B, g, Nn, _ptr(dv, torch.float32), R, _stream())


def _attn_account_group(B, Nn, A, R, passes):
    if FLOPS["on"]:                      # shared sets: every node row of u / v (and d(u)) moves once per image and step
        FLOPS["attn_bytes"] = FLOPS.get("attn_bytes", 0.0) + 4.0 * B * Nn * (A + R) * passes


def log_softmax_rows_(x, active=None):
    rows, V = x.shape
    call("subgc_log_softmax_rows", _ptr(x), ld(x), rows, V, _ptr(active, torch.int32), _stream())
    return x


def log_softmax_rows_bwd(logp, dout, dlogits, active=None):
    rows, V = logp.shape
    call("subgc_log_softmax_rows_bwd", _ptr(logp), _ptr(dout), _ptr(dlogits), ld(logp), rows, V, _ptr(active, torch.int32),
         int(is_b16(dlogits)), _stream())
    return dlogits


def row_lse(x):
    """lse[r] = logsumexp(x[r, :]) without touching x (subgc_row_lse_f32)."""
    rows, V = x.shape
    lse = torch.empty(rows, device=x.device, dtype=torch.float32)
    call("subgc_row_lse_f32", _ptr(x, torch.float32), ld(x), rows, V, _ptr(lse), _stream())
    return lse


def masked_nll_fwd(logp, target, mask, den=None, lse=None):
    """logp [S,T,V] contiguous; target/mask are [S,T] views (unit inner stride) of the label tensors.  `den` (device scalar):
    the denominator to use instead of the mask sum of the rows given.  `lse` [S*T]: logp holds raw logits (row_lse)."""
    S, T, V = logp.shape
    loss = torch.empty((), device=logp.device, dtype=torch.float32)
    scratch = torch.empty(2, device=logp.device, dtype=torch.float32)
    call("subgc_masked_nll_fwd", _ptr(logp), _ptr(target, torch.int64), target.stride(0), _ptr(mask, torch.float32), mask.stride(0),
         _ptr(loss), _ptr(scratch), S, T, V, _ptr(den, torch.float32), _ptr(lse, torch.float32), _stream())
    return loss, scratch


def live_plan(labels, mask_t):
    """Row plan of the packed decoder (subgc_live_plan) -> (perm32, perm64, inv32, plan int32 [2T+1] = counts[T] | offs[T+1], den)."""
    S, T = mask_t.shape
    dev = mask_t.device
    perm32 = torch.empty(S, device=dev, dtype=torch.int32)
    inv32 = torch.empty(S, device=dev, dtype=torch.int32)
    perm64 = torch.empty(S, device=dev, dtype=torch.int64)
    plan = torch.empty(2 * T + 1, device=dev, dtype=torch.int32)
    den = torch.empty(1, device=dev, dtype=torch.float32)
    if mask_t.stride(1) != 1 or labels.stride(1) != 1:
        raise SubgcError("live_plan: labels / mask need unit inner stride")
    call("subgc_live_plan", _ptr(labels, torch.int64), labels.stride(0), _ptr(mask_t, torch.float32), mask_t.stride(0), S, T, _ptr(perm32),
         _ptr(perm64), _ptr(inv32), _ptr(plan), plan.data_ptr() + 4 * T, _ptr(den), _stream())
    return perm32, perm64, inv32, plan, den


def packed_rows(labels, target, mask_t, perm32, offs, lens, idx, img):
    """-> (labels_p [S, cols], tok_flat [T*S], tgt_p [T*S, 1], msk_p [T*S, 1], lens_p, idx_p, img_p) in the plan's packed order."""
    S, T = mask_t.shape
    N = idx.size(1)
    dev = labels.device
    cols = labels.size(1)
    labels_p = torch.empty(S, cols, device=dev, dtype=torch.int64)
    tok = torch.empty(T * S, device=dev, dtype=torch.int64)
    tgt = torch.empty(T * S, 1, device=dev, dtype=torch.int64)
    msk = torch.empty(T * S, 1, device=dev, dtype=torch.float32)
    lens_p = torch.empty(S, device=dev, dtype=torch.int32)
    idx_p = torch.empty(S, N, device=dev, dtype=torch.int64)
    img_p = torch.empty(S, device=dev, dtype=torch.int32)
    if idx.stride(1) != 1 or target.stride(1) != 1 or mask_t.stride(1) != 1 or labels.stride(1) != 1:
        raise SubgcError("packed_rows: unit inner strides needed")
    call("subgc_packed_rows", _ptr(labels, torch.int64), labels.stride(0), _ptr(target, torch.int64), target.stride(0), _ptr(mask_t, torch.float32),
         mask_t.stride(0), _ptr(perm32, torch.int32), _ptr(offs, torch.int32), S, T, _ptr(labels_p), cols, _ptr(tok), _ptr(tgt), _ptr(msk),
         _ptr(lens, torch.int32), _ptr(idx, torch.int64), idx.stride(0), _ptr(img, torch.int32), N, _ptr(lens_p), _ptr(idx_p), _ptr(img_p), _stream())
    return labels_p, tok, tgt, msk, lens_p, idx_p, img_p


def gpn_prep(gpn_obj_ind, gpn_pool_mtx, att_masks, spi):
    """gpn.py:43-52 views as flat [G = 2*b5*hb] arrays (pos half, then neg half): idx int64 [G,N], w [G,N], denom [G], img int32 [G]."""
    b5, _, hb, N = gpn_obj_ind.shape
    dev = gpn_obj_ind.device
    G = 2 * b5 * hb
    gpn_obj_ind, gpn_pool_mtx, att_masks = gpn_obj_ind.contiguous(), gpn_pool_mtx.contiguous(), att_masks.contiguous()
    idx = torch.empty(G, N, device=dev, dtype=torch.int64)
    w = torch.empty(G, N, device=dev, dtype=torch.float32)
    denom = torch.empty(G, device=dev, dtype=torch.float32)
    img = torch.empty(G, device=dev, dtype=torch.int32)
    call("subgc_gpn_prep", _ptr(gpn_obj_ind, torch.int64), _ptr(gpn_pool_mtx, torch.float32), _ptr(att_masks, torch.float32), b5, hb, N, int(spi),
         _ptr(idx), _ptr(w), _ptr(denom), _ptr(img), _stream())
    return idx, w, denom, img


def gpn_select(score, gpn_obj_ind, att_masks, read_out, spi):
    """gpn.py:63-78 -> (sel_idx int64 [b5,N], lens int32 [b5], ro_sel [b5, W], img_s int32 [b5]): the best positive sub-graph of every
    sentence and the sentence's image."""
    b5, _, hb, N = gpn_obj_ind.shape
    dev = score.device
    W = read_out.size(1)
    sel_idx = torch.empty(b5, N, device=dev, dtype=torch.int64)
    lens = torch.empty(b5, device=dev, dtype=torch.int32)
    ro_sel = torch.empty(b5, W, device=dev, dtype=torch.float32)
    img_s = torch.empty(b5, device=dev, dtype=torch.int32)
    call("subgc_gpn_select", _ptr(score, torch.float32), _ptr(gpn_obj_ind.contiguous(), torch.int64), _ptr(att_masks.contiguous(), torch.float32),
         _ptr(read_out, torch.float32), b5, hb, N, W, _ptr(sel_idx), _ptr(lens), _ptr(ro_sel), None, int(spi), _ptr(img_s), _stream())
    return sel_idx, lens, ro_sel, img_s


def class_sum(x, cls, C):
    """-> [C, L]: sum of the rows of x [M, L] per class id (int32 cls [M]).  <= 64 classes: per-slab LDS accumulation + a column
    sum over the slabs (no atomics, fixed order); more classes: scatter-add with fp32 atomics (a few rows per class)."""
    M, L = x.shape
    if C <= 64:
        slabs = max(1, min(128, M // 64))
        part = torch.empty(slabs, C * L, device=x.device, dtype=torch.float32)
        call("subgc_class_partials", _ptr(x, torch.float32), ld(x), _ptr(cls, torch.int32), M, L, C, slabs, _ptr(part), _stream())
        return colsum(part).view(C, L)
    return scatter_add_rows(x, cls, zeros(C, L, device=x.device))


def add_n(ts):
    """Fresh tensor = sum of 2..4 same-shaped contiguous fp32 tensors (one launch)."""
    ts = [t.contiguous() for t in ts]
    out = torch.empty_like(ts[0])
    while len(ts) > 4:
        call("subgc_add_n_f32", _ptr(out), _ptr(ts[0], torch.float32), _ptr(ts[1], torch.float32), _ptr(ts[2], torch.float32), _ptr(ts[3], torch.float32),
             out.numel(), _stream())
        ts = [out] + ts[4:]
    ts = ts + [None] * (4 - len(ts))
    call("subgc_add_n_f32", _ptr(out), _ptr(ts[0], torch.float32), _ptr(ts[1], torch.float32), _ptr(ts[2], torch.float32), _ptr(ts[3], torch.float32),
         out.numel(), _stream())
    return out


def fill2d_(x, value):
    call("subgc_fill2d_f32", _ptr(x, torch.float32), ld(x), x.size(0), x.size(1), float(value), _stream())
    return x


def row_count(x):
    lens = torch.empty(x.size(0), device=x.device, dtype=torch.int32)
    call("subgc_row_count_f32", _ptr(x, torch.float32), ld(x), x.size(0), x.size(1), _ptr(lens), _stream())
    return lens


def masked_nll_bwd(target, mask, scratch, dloss, S, T, V):
    dlogp = torch.empty(S, T, V, device=mask.device, dtype=torch.float32)
    call("subgc_masked_nll_bwd", _ptr(target, torch.int64), target.stride(0), _ptr(mask), mask.stride(0), _ptr(scratch), _ptr(dloss),
         _ptr(dlogp), S, T, V, _stream())
    return dlogp


def nll_logsoftmax_bwd(logp, target, mask, scratch, dloss, dlogits, active, S, T, V, lse=None):
    call("subgc_nll_logsoftmax_bwd", _ptr(logp), _ptr(target, torch.int64), target.stride(0), _ptr(mask), mask.stride(0), _ptr(scratch),
         _ptr(dloss), _ptr(dlogits), ld(dlogits), S, T, V, _ptr(active, torch.int32), int(is_b16(dlogits)), _ptr(lse, torch.float32), _stream())
    return dlogits


def step_active(labels, T):
    S = labels.size(0)
    active = torch.empty(S * T, device=labels.device, dtype=torch.int32)
    call("subgc_step_active", _ptr(labels, torch.int64), labels.stride(0), S, T, _ptr(active), _stream())
    return active


def decode_pick(logp, k, temp, u, t, seq, seqlp, next_tok, unfinished, n_unf, prev_count=None, raw=False):
    n, V = logp.shape
    call("subgc_decode_pick", _ptr(logp), ld(logp), n, V, int(k), float(temp), _ptr(u), int(t), _ptr(seq, torch.int64), _ptr(seqlp),
         seq.size(1), _ptr(next_tok, torch.int64), _ptr(unfinished, torch.int32), _ptr(n_unf, torch.int32), _ptr(prev_count, torch.int32), int(raw), _stream())


def row_topk(x, k, vals, idx, log_softmax=True):
    """vals/idx[r, :k] = the k largest of row r (value desc, index asc); raw logits -> log-probs when log_softmax."""
    call("subgc_row_topk_f32", _ptr(x), ld(x), x.size(0), x.size(1), int(k), int(log_softmax), _ptr(vals), _ptr(idx, torch.int32), _stream())
    return vals, idx


def beam_step(tv, ti, st, tok, src, t, T, G, bd, kk, unk, constraint, lam):
    """One (diverse) beam-search step of every sub-graph on the device (subgc_beam_step).  `st`: the tables of beam.DeviceTables."""
    n = st.sums.size(0)
    call("subgc_beam_step", _ptr(tv, torch.float32), _ptr(ti, torch.int32), _ptr(st.seq, torch.int32), _ptr(st.lps, torch.float32),
         _ptr(st.sums, torch.float32), _ptr(st.done_cnt, torch.int32), _ptr(st.done_seq, torch.int32), _ptr(st.done_lps, torch.float32),
         _ptr(st.done_p, torch.float32), _ptr(st.done_len, torch.int32), _ptr(tok, torch.int64), _ptr(src, torch.int32), n, int(t), int(T),
         int(G), int(bd), int(kk), int(unk), int(bool(constraint)), float(lam), st.cap, _stream())


def uniform(shape, seed, offset, device):
    out = torch.empty(shape, device=device, dtype=torch.float32)
    call("subgc_uniform_f32", _ptr(out), out.numel(), int(seed), int(offset), _stream())
    return out


def multinomial_rows_(logits, u, sel_u, prob, tok):
    """tok[r] <- draw from softmax(logits[r]) where sel_u[r] < prob (in place; `tok` may be a strided column of int64)."""
    rows, V = logits.shape
    call("subgc_multinomial_rows", _ptr(logits, torch.float32), ld(logits), rows, V, _ptr(u, torch.float32), _ptr(sel_u, torch.float32),
         float(prob), _ptr(tok, torch.int64), tok.stride(0) if tok.dim() else 1, _stream())
    return tok


def packed_time_sum(src, offsets, T, S, dst):
    """dst[s] = sum over live steps t of src[offsets[t] + s] (packed decoder; offsets int32 [T+1] on the device)."""
    call("subgc_packed_time_sum", _ptr(src), _ptr(offsets, torch.int32), int(T), int(S), src.size(1), _ptr(dst, torch.float32),
         int(is_b16(src)), _stream())
    return dst


def rank_desc(score):
    """(sorted scores, order): stable descending sort of a 1-D fp32 score vector (eval_utils.py:106)."""
    score = score.contiguous()
    order = torch.empty(score.numel(), device=score.device, dtype=torch.int64)
    srt = torch.empty_like(score)
    call("subgc_rank_desc_f32", _ptr(score, torch.float32), score.numel(), _ptr(order), _ptr(srt), _stream())
    return srt, order


def eval_collect(score, keep, seq, bounds, identity=False, AL=None, idx=None, pick=None):
    """The eval loop's per-image work for a whole decode batch (misc/eval_utils.py:105-121; grounding: misc/grd_utils.py:36-47):
    one ranking launch (subgc_eval_rank_rows), one grounding launch when `AL` (the decode loop's attention buffer [T1, rows, N]) and
    `idx` (the kept sub-graphs' node lists [rows, N]) are given, and ONE device -> host copy of everything.
    score [rows] fp32, keep [rows] int64, seq [rows, T] int64, bounds: python list of the I + 1 row boundaries of the images,
    pick: optional per-image subg_index list (default 0 = the best-ranked caption).
    -> dict of host numpy arrays: order / score / keep (int64) / seq (int64) [rows...], and with grounding att2 / node [I, T1], n_words [I]."""
    import numpy as np
    dev = score.device
    rows, T = seq.shape
    I = len(bounds) - 1
    if bounds[-1] != rows or score.numel() != rows or keep.numel() != rows:
        raise SubgcError("eval_collect: bounds / score / keep do not cover the rows of seq")
    ground = AL is not None
    T1 = AL.size(0) if ground else 0
    seg = upload(list(bounds) + ([0] * I if pick is None else [int(p) for p in pick]), torch.int32, dev)
    words = 3 * rows + rows * T + (2 * I * T1 + I if ground else 0)
    arena = torch.empty(max(words, 1), device=dev, dtype=torch.int32)
    o = 0
    order = arena[o:o + rows]; o += rows
    score_s = arena[o:o + rows].view(torch.float32); o += rows
    keep_s = arena[o:o + rows]; o += rows
    seq_s = arena[o:o + rows * T]; o += rows * T
    max_rows = max([b - a for a, b in zip(bounds, bounds[1:])] + [0])
    if I and rows:
        call("subgc_eval_rank_rows", _ptr(score.contiguous(), torch.float32), _ptr(keep.contiguous(), torch.int64), _ptr(seq.contiguous(), torch.int64),
             T, _ptr(seg), I, max_rows, int(bool(identity)), _ptr(order), _ptr(score_s), _ptr(keep_s), _ptr(seq_s), _stream())
    if ground:
        att2 = arena[o:o + I * T1]; o += I * T1
        node = arena[o:o + I * T1]; o += I * T1
        nw = arena[o:o + I]; o += I
        if AL.stride(2) != 1 or idx.stride(1) != 1:
            raise SubgcError("eval_collect: AL / idx need unit inner strides")
        if I:
            call("subgc_grounding_argmax", _ptr(AL, torch.float32), AL.stride(0), AL.stride(1), AL.size(2), T1, _ptr(seq.contiguous(), torch.int64), T,
                 _ptr(idx, torch.int64), idx.stride(0), _ptr(seg), _ptr(order) if (rows and not identity) else None,
                 _ptr(seg[I + 1:]) if pick is not None else None, I, _ptr(att2), _ptr(node), _ptr(nw), _stream())
    host = arena.cpu().numpy()                                        # the one copy (synchronises the stream)
    o = 0
    out = {"order": host[o:o + rows].astype(np.int64)}; o += rows
    out["score"] = host[o:o + rows].view(np.float32).copy(); o += rows
    out["keep"] = host[o:o + rows].astype(np.int64); o += rows
    out["seq"] = host[o:o + rows * T].reshape(rows, T).astype(np.int64); o += rows * T
    if ground:
        out["att2"] = host[o:o + I * T1].reshape(I, T1).copy(); o += I * T1
        out["node"] = host[o:o + I * T1].reshape(I, T1).copy(); o += I * T1
        out["n_words"] = host[o:o + I].copy()
    return out


def dropout_mask(shape, p, seed, offset, device):
    keep = torch.empty(shape, device=device, dtype=torch.uint8)
    call("subgc_dropout_mask", _ptr(keep), keep.numel(), float(p), int(seed), int(offset), _stream())
    return keep


def fill_(x, value):
    call("subgc_fill_f32", _ptr(x, torch.float32), x.numel(), float(value), _stream())
    return x


def zeros(*shape, device):
    return fill_(torch.empty(*shape, device=device, dtype=torch.float32), 0.0)


def copy2d(x, y, accumulate=False):
    if is_b16(x):
        if accumulate or not is_b16(y):
            raise SubgcError("copy2d: bf16 rows are copied to bf16 rows only")
        call("subgc_copy2d_b16", _ptr(x, BF16), ld(x), _ptr(y, BF16), ld(y), x.size(0), x.size(1), _stream())
        return y
    call("subgc_copy2d_f32", _ptr(x), ld(x), _ptr(y), ld(y), x.size(0), x.size(1), int(accumulate), _stream())
    return y


def relu_bwd(dy, y, scale=1.0, out=None, bf16=False):
    """dz = dy * scale * [y > 0] (contiguous); `bf16`: dz is written bf16 (it only feeds the two gradient GEMMs)."""
    if out is None:
        out = torch.empty(y.shape, device=y.device, dtype=BF16 if bf16 else torch.float32)
    if not (y.is_contiguous() and dy.is_contiguous() and out.is_contiguous()):
        raise SubgcError("relu_bwd works on contiguous tensors")
    call("subgc_relu_bwd", _ptr(dy, torch.float32), _ptr(y), float(scale), _ptr(out), y.numel(), int(is_b16(out)) | (int(is_b16(y)) << 1), _stream())
    return out


def gather_rows(src, rows, dst, m_dev=None):
    """dst fp32 or bf16."""
    call("subgc_gather_rows", _ptr(src, torch.float32), ld(src), _ptr(rows, torch.int32), _ptr(dst), ld(dst), dst.size(0), dst.size(1),
         _ptr(m_dev, torch.int32), int(is_b16(dst)), _stream())
    return dst


def gather_rows_keep(src, rows, keep, scale, dst, m_dev=None):
    """dst[m] = src[rows[m]] * keep[m] * scale (keep: uint8 [M, L] or None); src / dst fp32 or bf16."""
    call("subgc_gather_rows_keep", _ptr(src), ld(src), _ptr(rows, torch.int32), _ptr(keep, torch.uint8), ld(keep) if keep is not None else 0, float(scale),
         _ptr(dst), ld(dst), dst.size(0), dst.size(1), _ptr(m_dev, torch.int32), int(is_b16(dst)) | (int(is_b16(src)) << 1), _stream())
    return dst


def gather_rows_multi(pairs, rows):
    """dst[m] = src[rows[m]] for up to four (src, dst) pairs of 2-D views in one launch (subgc_gather_rows_multi)."""
    if not 1 <= len(pairs) <= 4:
        raise SubgcError("gather_rows_multi takes 1..4 (src, dst) pairs")
    args = []
    for src, dst in list(pairs) + [(None, None)] * (4 - len(pairs)):
        args += [_ptr(src, torch.float32), ld(src) if src is not None else 0, _ptr(dst, torch.float32), ld(dst) if dst is not None else 0,
                 dst.size(1) if dst is not None else 0]
    call("subgc_gather_rows_multi", len(pairs), *args, _ptr(rows, torch.int32), pairs[0][1].size(0), _stream())


def scatter_add_rows(src, rows, dX, m_dev=None):
    call("subgc_scatter_add_rows", _ptr(src), ld(src), _ptr(rows, torch.int32), _ptr(dX), ld(dX), src.size(0), src.size(1),
         _ptr(m_dev, torch.int32), _stream())
    return dX


def sumsq(g, out):
    call("subgc_sumsq_f32", _ptr(g), g.numel(), _ptr(out), _stream())
    return out


def clip_adam_step(p, g, m, v, sumsq_t, max_norm, lr, beta1, beta2, eps, wd, step, grad_scale=1.0, p_bf16=None, zero_grad=False):
    call("subgc_clip_adam_step_zero" if zero_grad else "subgc_clip_adam_step", _ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), _ptr(sumsq_t), float(max_norm), float(lr),
         float(beta1), float(beta2), float(eps), float(wd), int(step), float(grad_scale), _ptr(p_bf16, BF16), _stream())
