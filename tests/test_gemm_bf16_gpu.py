"""The bf16-operand MFMA GEMM (csrc/gemm_bf16.hip, subgc_gemm_bf16) and the bf16 plumbing kernels against fp64 products of
the SAME bf16 operands: the kernel accumulates in fp32, so the only error is summation order (tolerance 2e-5 of the
result scale, like the fp32 GEMM's), plus one bf16 rounding (2^-9 relative) where the destination is bf16."""
import numpy as np
import pytest
import torch

from subgc import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


def rnd(*shape, seed=0, scale=1.0):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale).to(DEV)


def operands(mode, M, N, K, seed, lda_pad=0, ldb_pad=0):
    a = rnd(*((K, M + lda_pad) if mode == "tn" else (M, K + lda_pad)), seed=seed).to(BF)
    b = rnd(*((N, K + ldb_pad) if mode == "nt" else (K, N + ldb_pad)), seed=seed + 1).to(BF)
    a = a[:, :M] if mode == "tn" else a[:, :K]
    b = b[:, :K] if mode == "nt" else b[:, :N]
    ad = a.double().t() if mode == "tn" else a.double()
    bd = b.double().t() if mode == "nt" else b.double()
    return a, b, ad @ bd


SHAPES = [("nt", 1280, 4000, 3000), ("nt", 300, 512, 1000), ("nn", 1280, 3000, 4000), ("nn", 200, 1000, 512), ("tn", 4000, 2000, 2176),
          ("tn", 9488, 1000, 1357), ("nt", 129, 130, 72), ("nn", 5, 8, 8), ("tn", 24, 48, 3), ("nt", 4736, 1024, 2048), ("tn", 512, 1000, 21760)]


@pytest.mark.parametrize("mode,M,N,K", SHAPES)
def test_gemm_bf16_matches_fp64_product_of_the_same_operands(mode, M, N, K):
    a, b, ref = operands(mode, M, N, K, seed=M + N + K, lda_pad=8, ldb_pad=16)
    scale = float(ref.abs().max())
    out = torch.full((M, N), float("nan"), device=DEV)
    ops.gemm(a, b, out, ta=mode == "tn", tb=mode == "nt")
    assert float((out.double() - ref).abs().max()) < 2e-5 * scale
    o16 = torch.empty(M, N + 4, device=DEV, dtype=BF)[:, :N]
    ops.gemm(a, b, o16, ta=mode == "tn", tb=mode == "nt")
    assert float((o16.double() - ref).abs().max()) < 2.0 ** -8 * scale
    # the bf16 destination is the rounded fp32 result (the two calls may split K differently: last-bit fp32 differences can
    # move a rounding boundary, never more than one bf16 ulp)
    assert float(((o16.float() - out).abs() - 2.0 ** -8 * out.abs()).max()) <= 1e-6 * scale


def test_gemm_bf16_epilogues_and_second_destination():
    M, N, K = 700, 1000, 512
    a, b, ref = operands("nt", M, N, K, seed=3)
    bias, add = rnd(N, seed=5), rnd(M, N, seed=6)
    keep = (torch.rand(M, N, generator=torch.Generator().manual_seed(7)) < 0.5).to(torch.uint8).to(DEV)
    want = torch.relu(ref + bias.double() + add.double()) * keep.double() * 2.0
    out, o16 = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV, dtype=BF)
    ops.gemm(a, b, out, tb=True, bias=bias, add=add, relu=True, keep=keep, keep_scale=2.0, out16=o16)
    scale = float(want.abs().max())
    assert float((out.double() - want).abs().max()) < 2e-5 * scale
    assert torch.equal(o16, out.to(BF))
    prev = rnd(M, N, seed=8)
    acc = prev.clone()
    ops.gemm(a, b, acc, tb=True, bias=bias, accum=True)                  # split-K capable shape: the reduce kernel applies bias + accumulate
    assert float((acc.double() - (ref + bias.double() + prev.double())).abs().max()) < 2e-5 * float(ref.abs().max())


@pytest.mark.parametrize("mode", ["nt", "tn"])
def test_gemm_bf16_device_side_row_count(mode):
    """m_dev bounds the rows of the stored A: M for [M,K] operands, K for the transposed (weight-gradient) form."""
    M, N, K = (2000, 512, 1000) if mode == "nt" else (512, 1000, 2000)
    a, b, _ = operands(mode, M, N, K, seed=11)
    live = 1237
    m_dev = torch.tensor([live], device=DEV, dtype=torch.int32)
    out = torch.full((M, N), 7.0, device=DEV)
    ops.gemm(a, b, out, ta=mode == "tn", tb=mode == "nt", m_dev=m_dev)
    if mode == "nt":
        ref = a[:live].double() @ b.double().t()
        assert float((out[:live].double() - ref).abs().max()) < 2e-5 * float(ref.abs().max())
        assert float((out[live:] - 7.0).abs().max()) == 0.0              # rows past the device-side count are not written
    else:
        ref = a[:live].double().t() @ b[:live].double()
        assert float((out.double() - ref).abs().max()) < 2e-5 * float(ref.abs().max())


def test_two_streams_issue_split_k_gemms_concurrently():
    """The split-K scratch is an argument of the call (one buffer per stream on the host side), not a process global: the same
    products issued from two streams at once equal the serial results, for the fp32 and the bf16 kernels."""
    M, N, K = 640, 4000, 3000
    a32, b32 = rnd(M, K, seed=1), rnd(N, K, seed=2)
    c32, d32 = rnd(M, K, seed=3), rnd(N, K, seed=4)
    serial = []
    for x, w in ((a32, b32), (c32, d32), (a32.to(BF), b32.to(BF)), (c32.to(BF), d32.to(BF))):
        o = torch.empty(M, N, device=DEV)
        ops.gemm(x, w, o, tb=True)
        serial.append(o)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for rep in range(3):
        outs = [torch.empty(M, N, device=DEV) for _ in range(4)]
        with torch.cuda.stream(s1):
            for _ in range(4):
                ops.gemm(a32, b32, outs[0], tb=True)
                ops.gemm(a32.to(BF), b32.to(BF), outs[2], tb=True)
        with torch.cuda.stream(s2):
            for _ in range(4):
                ops.gemm(c32, d32, outs[1], tb=True)
                ops.gemm(c32.to(BF), d32.to(BF), outs[3], tb=True)
        torch.cuda.synchronize()
        for got, want in zip(outs, serial):
            assert torch.equal(got, want)
    assert len({k for k in ops._WS if k[0] == torch.cuda.current_device()}) >= 3      # default stream + the two side streams


def test_cast_transpose_and_copy_kernels():
    x = rnd(37, 300, seed=2)
    y = ops.cast_bf16(x)
    assert y.shape == (37, 304) and torch.equal(y[:, :300], x.to(BF)) and float(y[:, 300:].abs().max()) == 0.0
    m_dev = torch.tensor([20], device=DEV, dtype=torch.int32)
    z = torch.full((37, 304), 3.0, device=DEV, dtype=BF)
    ops.cast_bf16(x, out=z, m_dev=m_dev)
    assert torch.equal(z[:20, :300], x[:20].to(BF)) and float((z[20:].float() - 3.0).abs().max()) == 0.0
    big = rnd(1000, 4000, seed=3)
    assert torch.equal(ops.cast_bf16(big), big.to(BF))
    t = ops.transpose_bf16(big[:, :3001])
    assert t.shape == (3001, 1000) and t.stride(0) % 8 == 0 and torch.equal(t, big[:, :3001].t().to(BF))
    dst = torch.zeros(1000, 5000, device=DEV, dtype=BF)
    ops.copy2d(ops.cast_bf16(big), dst[:, 1000:])
    assert torch.equal(dst[:, 1000:], big.to(BF)) and float(dst[:, :1000].abs().max()) == 0.0
    g = torch.Generator().manual_seed(4)
    xb = torch.randn(777, 4000, generator=g).to(DEV).to(BF)
    cs = ops.colsum(xb)
    np.testing.assert_allclose(cs.cpu().numpy(), xb.double().sum(0).cpu().numpy(), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("mode,M,N,K", [("nt", 16640, 1024, 512), ("nn", 16640, 1024, 1024), ("nt", 16900, 1024, 264), ("nt", 8448, 512, 256)])
def test_gemm_bf16_row_cut_shapes(mode, M, N, K):
    """Shapes whose 256 x 256 tile count is a whole number of 256-workgroup rounds plus a sliver (Full_GC_Kar's 16640-row GCN products:
    260 tiles) run as two launches, the whole rounds and the remaining rows with their own plan (gemm_bf16.hip run()); every epilogue
    operand is row-offset with them."""
    a, b, ref = operands(mode, M, N, K, seed=11, lda_pad=8, ldb_pad=8)
    bias, add = rnd(N, seed=5), rnd(M, N, seed=6)
    keep = (torch.rand(M, N, generator=torch.Generator().manual_seed(7)) < 0.5).to(torch.uint8).to(DEV)
    want = torch.relu(ref + bias.double() + add.double()) * keep.double() * 2.0
    out = torch.full((M, N), float("nan"), device=DEV)
    o16 = torch.empty(M, N + 64, device=DEV, dtype=BF)[:, :N]
    ops.gemm(a, b, out, tb=mode == "nt", bias=bias, add=add, relu=True, keep=keep, keep_scale=2.0, out16=o16)
    scale = float(want.abs().max())
    assert float((out.double() - want).abs().max()) < 2e-5 * scale
    assert torch.equal(o16, out.to(BF))
    plain = torch.full((M, N), float("nan"), device=DEV)
    ops.gemm(a, b, plain, tb=mode == "nt")
    assert float((plain.double() - ref).abs().max()) < 2e-5 * float(ref.abs().max())
    acc = add.clone()
    ops.gemm(a, b, acc, tb=mode == "nt", accum=True)
    assert float((acc.double() - ref - add.double()).abs().max()) < 2e-5 * float(ref.abs().max())


@pytest.mark.parametrize("mode,M,N,K", [("nn", 16640, 512, 1024), ("tn", 1024, 512, 16640), ("nn", 9472, 512, 1024), ("tn", 1024, 512, 9472),
                                        ("nt", 700, 1000, 264), ("nn", 300, 72, 40), ("tn", 2048, 512, 6464), ("nt", 16640, 1024, 512)])
def test_gemm_bf16_pair_equals_two_single_launches(mode, M, N, K):
    """subgc_gemm_bf16_pair: two products of one shape in one launch (first / second half of the grid; planned together: tile, K parts, row
    cut) -- the same kernels on the same operands: bit for bit the results of two subgc_gemm_bf16 calls whenever both plan alike, and within
    summation-order rounding otherwise; against fp64 in any case."""
    ops_ab = [operands(mode, M, N, K, seed=31 + 7 * i, lda_pad=8, ldb_pad=8) for i in range(2)]
    (a1, b1, r1), (a2, b2, r2) = ops_ab
    bias1, bias2 = rnd(N, seed=5), rnd(N, seed=6)
    for dst in ("f32", "bf16"):
        mk = (lambda: torch.full((M, N), float("nan"), device=DEV)) if dst == "f32" else (lambda: torch.empty(M, N + 8, device=DEV, dtype=BF)[:, :N])
        o1, o2, s1, s2 = mk(), mk(), mk(), mk()
        use_bias = mode != "tn"
        ops.gemm_pair(a1, a2, b1, b2, o1, o2, ta=mode == "tn", tb=mode == "nt", bias1=bias1 if use_bias else None, bias2=bias2 if use_bias else None, relu=use_bias)
        ops.gemm(a1, b1, s1, ta=mode == "tn", tb=mode == "nt", bias=bias1 if use_bias else None, relu=use_bias)
        ops.gemm(a2, b2, s2, ta=mode == "tn", tb=mode == "nt", bias=bias2 if use_bias else None, relu=use_bias)
        for o, s_, ref, bias in ((o1, s1, r1, bias1), (o2, s2, r2, bias2)):
            want = torch.relu(ref + bias.double()) if use_bias else ref
            tol = (2e-5 if dst == "f32" else 2.0 ** -8) * float(want.abs().max())
            assert float((o.double() - want).abs().max()) < tol
            assert float((o.double() - s_.double()).abs().max()) <= tol
    if mode == "tn":                                            # accumulate into both (the weight-gradient use)
        p1, p2 = rnd(M, N, seed=8), rnd(M, N, seed=9)
        q1, q2 = p1.clone(), p2.clone()
        ops.gemm_pair(a1, a2, b1, b2, q1, q2, ta=True, accum=True)
        assert float((q1.double() - p1.double() - r1).abs().max()) < 2e-5 * float(r1.abs().max())
        assert float((q2.double() - p2.double() - r2).abs().max()) < 2e-5 * float(r2.abs().max())


def test_colsum_set_equals_single_column_sums():
    """subgc_colsum_bf16_set: up to three bf16 matrices of one shape in two launches -- the slab plan and summation order of the single
    call, so bit for bit its result; mixed shapes fall back to single calls."""
    xs = [rnd(16640, 1024, seed=i).to(BF) for i in range(3)] + [rnd(16640, 512, seed=9).to(BF), rnd(700, 1032, seed=10)[:, :1024].to(BF)]
    for acc in (False, True):
        outs = [rnd(x.size(1), seed=20 + i) for i, x in enumerate(xs)]
        want = [ops.colsum(x, out=o.clone(), accumulate=acc) for x, o in zip(xs, outs)]
        ops.colsum_set(xs, outs, accumulate=acc)
        for o, w in zip(outs, want):
            assert torch.equal(o, w)


def test_gemm_pair_falls_back_to_two_calls_when_the_problems_differ():
    """Different shapes / leading dimensions / storage types cannot share a launch: `ops.gemm_pair` then issues two `ops.gemm` calls, and with
    `ops.PAIR_LAUNCHES = False` always does -- same results either way."""
    a1, b1, r1 = operands("nt", 512, 256, 128, seed=1)
    a2, b2, r2 = operands("nt", 384, 256, 128, seed=2)                       # other M
    o1, o2 = torch.empty(512, 256, device=DEV), torch.empty(384, 256, device=DEV)
    ops.gemm_pair(a1, a2, b1, b2, o1, o2, tb=True)
    assert float((o1.double() - r1).abs().max()) < 2e-5 * float(r1.abs().max()) and float((o2.double() - r2).abs().max()) < 2e-5 * float(r2.abs().max())
    a3, b3, r3 = operands("nt", 512, 256, 128, seed=3, lda_pad=8)            # same shape, other leading dimension
    o3 = torch.empty(512, 256, device=DEV)
    ops.gemm_pair(a1, a3, b1, b3, o1, o3, tb=True)
    assert float((o3.double() - r3).abs().max()) < 2e-5 * float(r3.abs().max())
    f1, f2 = a1.float(), a3.float().contiguous()                             # fp32 operands: the fp32 pair kernel
    g1, g2 = b1.float(), b3.float().contiguous()
    p1, p2 = torch.empty(512, 256, device=DEV), torch.empty(512, 256, device=DEV)
    ops.gemm_pair(f1, f2, g1, g2, p1, p2, tb=True)
    assert float((p1.double() - r1).abs().max()) < 2e-5 * float(r1.abs().max()) and float((p2.double() - r3).abs().max()) < 2e-5 * float(r3.abs().max())
    ops.PAIR_LAUNCHES = False
    try:
        q1, q2 = torch.empty(512, 256, device=DEV), torch.empty(512, 256, device=DEV)
        ops.gemm_pair(a1, a1, b1, b1, q1, q2, tb=True)
    finally:
        ops.PAIR_LAUNCHES = True
    s1, s2 = torch.empty(512, 256, device=DEV), torch.empty(512, 256, device=DEV)
    ops.gemm_pair(a1, a1, b1, b1, s1, s2, tb=True)
    assert torch.equal(q1, q2) and torch.equal(s1, s2) and float((q1 - s1).abs().max()) <= 2e-5 * float(r1.abs().max())


# ---- the 256 x 256 x 64 eight-phase form (csrc/gemm_bf16_p8.h), forced through the per-call flag bits -------------------------------
P8_SHAPES = [("nt", 1280, 4000, 3000), ("nt", 300, 512, 1000), ("nn", 1280, 3000, 4000), ("nn", 200, 1000, 512), ("tn", 4000, 2000, 2176),
             ("tn", 9488, 1000, 1357), ("nt", 129, 130, 72), ("nn", 5, 8, 8), ("tn", 24, 48, 8), ("nt", 4736, 1024, 2048), ("tn", 512, 1000, 21760),
             ("nt", 257, 513, 64), ("nt", 256, 256, 128), ("nn", 511, 264, 200), ("tn", 264, 504, 1003), ("nt", 1000, 9488, 1000)]


@pytest.mark.parametrize("mode,M,N,K", P8_SHAPES)
def test_gemm_bf16_eight_phase_form_matches_fp64_product(mode, M, N, K):
    """Every layout, ragged M / N edges (rows past the operand are fed as zeros by the DMA's range check), K tails that end inside a 64-deep
    buffer (K % 64 in {8, 16, 40, 56}), one- and two-buffer K loops, odd and even K-tile counts."""
    a, b, ref = operands(mode, M, N, K, seed=M + N + K, lda_pad=8, ldb_pad=16)
    scale = float(ref.abs().max())
    out = torch.full((M, N), float("nan"), device=DEV)
    with ops.gemm_tune(tile="p8"):
        ops.gemm(a, b, out, ta=mode == "tn", tb=mode == "nt")
    assert float((out.double() - ref).abs().max()) < 2e-5 * scale
    o16 = torch.empty(M, N + 4, device=DEV, dtype=BF)[:, :N]
    with ops.gemm_tune(tile="p8"):
        ops.gemm(a, b, o16, ta=mode == "tn", tb=mode == "nt")
    assert float((o16.double() - ref).abs().max()) < 2.0 ** -8 * scale


@pytest.mark.parametrize("mode,M,N,K,splits", [("nt", 1280, 4000, 3000, 3), ("nn", 1280, 2000, 4000, 2), ("tn", 4000, 1000, 14593, 4),
                                                ("tn", 1024, 512, 16640, 8), ("nt", 700, 1000, 1000, 2)])
def test_gemm_bf16_eight_phase_split_k(mode, M, N, K, splits):
    """K parts of the eight-phase form: a part's range ends where the next one starts (the DMA masks at the part's end, not only at K), the
    planes are summed by the shared reduce kernel with the epilogue."""
    a, b, ref = operands(mode, M, N, K, seed=M + K)
    bias = rnd(N, seed=9)
    want = ref + bias.double()
    out = torch.full((M, N), float("nan"), device=DEV)
    with ops.gemm_tune(tile="p8", splits=splits):
        ops.gemm(a, b, out, ta=mode == "tn", tb=mode == "nt", bias=bias)
    assert float((out.double() - want).abs().max()) < 2e-5 * float(want.abs().max())


@pytest.mark.parametrize("tile,splits", [(128, 4), ("p8", 3), (0, 0)])
def test_gemm_bf16_split_k_with_relu_and_dropout_mask(tile, splits):
    """The reduce pass of a split product applies bias -> ReLU -> keep mask like the tile epilogue does (round 6: the 320-row fc projection
    of the Flickr shape, K = 4096, is 24 tiles of 128 x 128 and used to run unsplit because of its mask); fp32 and bf16 destinations of one
    call agree; tile = 0: the dispatch's own choice for this shape."""
    M, N, K = 320, 1000, 4096
    a, b, ref = operands("nt", M, N, K, seed=11)
    bias = rnd(N, seed=5)
    keep = (torch.rand(M, N, generator=torch.Generator().manual_seed(7)) < 0.5).to(torch.uint8).to(DEV)
    want = torch.relu(ref + bias.double()) * keep.double() * 2.0
    out, o16 = torch.full((M, N), float("nan"), device=DEV), torch.empty(M, N, device=DEV, dtype=BF)
    with ops.gemm_tune(tile=tile, splits=splits):
        ops.gemm(a, b, out, tb=True, bias=bias, relu=True, keep=keep, keep_scale=2.0, out16=o16)
    assert float((out.double() - want).abs().max()) < 2e-5 * float(want.abs().max())
    assert torch.equal(o16, out.to(BF))
    assert float(out[keep == 0].abs().max()) == 0.0


def test_gemm_bf16_eight_phase_epilogues_row_count_and_repeatability():
    M, N, K = 700, 1000, 512
    a, b, ref = operands("nt", M, N, K, seed=3)
    bias, add = rnd(N, seed=5), rnd(M, N, seed=6)
    keep = (torch.rand(M, N, generator=torch.Generator().manual_seed(7)) < 0.5).to(torch.uint8).to(DEV)
    want = torch.relu(ref + bias.double() + add.double()) * keep.double() * 2.0
    out, o16 = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV, dtype=BF)
    with ops.gemm_tune(tile="p8"):
        ops.gemm(a, b, out, tb=True, bias=bias, add=add, relu=True, keep=keep, keep_scale=2.0, out16=o16)
        assert float((out.double() - want).abs().max()) < 2e-5 * float(want.abs().max())
        assert torch.equal(o16, out.to(BF))
        live = 437
        m_dev = torch.tensor([live], device=DEV, dtype=torch.int32)
        o2 = torch.full((M, N), 7.0, device=DEV)
        ops.gemm(a, b, o2, tb=True, m_dev=m_dev)
        assert float((o2[:live].double() - ref[:live]).abs().max()) < 2e-5 * float(ref.abs().max())
        assert float((o2[live:] - 7.0).abs().max()) == 0.0
        # the hand-placed waits order every LDS read behind the DMA that fills it: repeated launches under load are bit-identical
        big_a, big_b, _ = operands("nt", 2048, 2304, 1000, seed=21)
        first = torch.empty(2048, 2304, device=DEV)
        ops.gemm(big_a, big_b, first, tb=True)
        for _ in range(20):
            again = torch.empty(2048, 2304, device=DEV)
            ops.gemm(big_a, big_b, again, tb=True)
            assert torch.equal(first, again)
