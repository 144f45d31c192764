"""Autograd boundary of the HIP path: torch.autograd.Function wrappers whose forward AND backward
are sequences of C-ABI launches (ops.py).  torch only owns the memory and the graph edges.

Encoder side (few launches per call) is modular: LinearFn, GcnNodesFn, GcnEdgesFn, BatchNormFn,
SubgraphPoolFn, GpnScoreFn, MaskedNLLFn.  The attention-LSTM decoder (the bulk of the FLOPs and of
the launches) is ONE Function, DecoderFn, with a hand-written BPTT: loop-invariant contractions
are hoisted out of the recurrence, every weight gradient is one batched-over-time GEMM, and the
ragged attention sets are packed once (reference: AttModel.py:122-177,328-368,400-471).
"""
from __future__ import annotations

import math

import torch
from torch.autograd import Function

from . import ops


def _direct(p, dev):
    """The existing .grad buffer of a leaf parameter if gradients may be accumulated into it in place, else None.  A caller that gets a
    buffer WILL write it through raw pointers (no torch version bump): `ops.GRAD_WRITES` is moved so that "the optimizer left the flat
    gradient buffer zeroed" (AttModel.flatten_grads' fast path) ends here, whatever entry point the backward came from."""
    g = getattr(p, "grad", None) if (p is not None and DIRECT_GRADS) else None
    ok = g is not None and g.is_contiguous() and g.dtype == torch.float32 and g.device == dev and p.is_leaf
    if ok:
        ops.GRAD_WRITES[0] += 1
    return g if ok else None


def _weight_bias_grads(dz, x, w_shape, gW, gb, need_w, need_b):
    """dW = dz^T x and db = column sums of dz for one linear layer -> (dW, db) as autograd wants them (None where the gradient was
    accumulated straight into the parameter's .grad view).  Both wanted: ONE launch (ops.wgrad)."""
    dW = db = None
    dev = dz.device
    if need_w and need_b and FOLD_BIAS_SUMS:
        oW = gW if gW is not None else torch.empty(w_shape, device=dev, dtype=torch.float32)
        ob = gb if gb is not None else torch.empty(w_shape[0], device=dev, dtype=torch.float32)
        ops.wgrad(dz, x, oW, ob.view(-1), accum=gW is not None, db_accum=gb is not None)
        return (None if gW is not None else oW), (None if gb is not None else ob)
    if need_w:
        if gW is not None:                # accumulate in the GEMM epilogue: no temporary, no separate `+=` pass by autograd
            ops.gemm(dz, x, gW, ta=True, accum=True)
        else:
            dW = torch.empty(w_shape, device=dev, dtype=torch.float32)
            ops.gemm(dz, x, dW, ta=True)
    if need_b:
        if gb is not None:
            ops.colsum(dz, out=gb, accumulate=True)
        else:
            db = ops.colsum(dz)
    return dW, db


# ------------------------------------------------------------------------------- linear
class LinearFn(Function):
    """y = [dropout]([relu](x W^T + b [+ add])) on 2-D row-major x (any leading dim).

    With `W16` (the bf16 twin of W, compute_dtype = bf16) the three products -- forward, data gradient, weight gradient -- run
    on bf16-stored operands (subgc_gemm_bf16): x may arrive as bf16 or is cast once (`x16` supplies an existing copy), the
    output is fp32, or bf16 when only GEMMs consume it (`out_b16`), or both (`want16`: second, non-differentiable output);
    d(out) is taken as fp32 or bf16, d(x) is produced in x's storage type; parameter gradients stay fp32."""

    @staticmethod
    def forward(ctx, x, W, b, add, keep, scale, relu, W16=None, x16=None, out_b16=False, want16=False):
        M, N = x.size(0), W.size(0)
        dev = x.device
        ctx.relu, ctx.scale, ctx.has_add = relu, scale, add is not None
        ctx.has_b = b is not None
        ctx.param_objs = (W, b)           # the leaf objects themselves: their .grad (a flat-bucket view) is written directly
        ctx.b16 = W16 is not None
        ctx.x_b16 = ops.is_b16(x)
        if W16 is None:
            if ops.is_b16(x) or out_b16 or want16:
                raise ops.SubgcError("LinearFn: bf16 activations need the bf16 twin of the weight")
            y = torch.empty(M, N, device=dev, dtype=torch.float32)
            ops.gemm(x, W, y, tb=True, bias=b, add=add, keep=keep, keep_scale=scale, relu=relu)
            ctx.save_for_backward(x, W, y if (relu or keep is not None) else None)
            return y
        if out_b16 and (relu or keep is not None):
            raise ops.SubgcError("LinearFn: a bf16-only output cannot carry the ReLU / dropout mask its backward needs")
        xa = x if ops.is_b16(x) else (x16 if x16 is not None else ops.as_b16(x))
        y = ops.empty_b16(M, N, dev) if out_b16 else torch.empty(M, N, device=dev, dtype=torch.float32)
        y16 = ops.empty_b16(M, N, dev) if want16 else None
        ops.gemm(xa, W16, y, tb=True, bias=b, add=add, keep=keep, keep_scale=scale, relu=relu, out16=y16)
        ctx.save_for_backward(xa, W16, y if (relu or keep is not None) else None)
        if want16:
            ctx.mark_non_differentiable(y16)
            ctx.set_materialize_grads(False)          # no zero tensor for the non-differentiable bf16 copy
            return y, y16
        return y

    @staticmethod
    def backward(ctx, dy, *_unused):
        x, W, y = ctx.saved_tensors
        dev = dy.device
        dx = dW = db = dadd = None
        gW, gb = (_direct(p, dev) for p in ctx.param_objs)
        if not ctx.b16:
            dy = dy.contiguous()
            dz = ops.relu_bwd(dy, y, ctx.scale) if y is not None else dy
            if ctx.needs_input_grad[0]:
                dx = torch.empty(x.size(0), W.size(1), device=dev, dtype=torch.float32)
                ops.gemm(dz, W, dx)
            dW, db = _weight_bias_grads(dz, x, W.shape, gW, gb, ctx.needs_input_grad[1], ctx.has_b and ctx.needs_input_grad[2])
            if ctx.has_add and ctx.needs_input_grad[3]:
                dadd = dz
            return dx, dW, db, dadd, None, None, None, None, None, None, None
        # bf16-stored operands: dz only feeds the two gradient GEMMs and the bias sum -> bf16 (fp32 too when `add` needs it)
        dz32 = None
        if y is not None:
            dy = dy.contiguous()
            if (ctx.has_add and ctx.needs_input_grad[3]) or y.size(1) % 8:
                dz32 = ops.relu_bwd(dy, y, ctx.scale)
                dz = ops.as_b16(dz32)
                dz32 = dz32 if ctx.has_add else None
            else:
                dz = ops.relu_bwd(dy, y, ctx.scale, bf16=True)          # contiguous [M, N], N % 8 == 0: a legal GEMM operand as it is
        else:
            dz = dy if ops.is_b16(dy) else ops.as_b16(dy.contiguous())
            if ctx.has_add and ctx.needs_input_grad[3]:
                dz32 = dy if not ops.is_b16(dy) else dy.float()
        if ctx.needs_input_grad[0]:
            dx = ops.empty_b16(x.size(0), W.size(1), dev) if ctx.x_b16 else torch.empty(x.size(0), W.size(1), device=dev, dtype=torch.float32)
            ops.gemm(dz, W, dx)
        Wm = ctx.param_objs[0]
        dW, db = _weight_bias_grads(dz, x, Wm.shape, gW, gb, ctx.needs_input_grad[1], ctx.has_b and ctx.needs_input_grad[2])
        if dz32 is not None:
            dadd = dz32
        return dx, dW, db, dadd, None, None, None, None, None, None, None


def linear(x, W, b=None, add=None, keep=None, scale=1.0, relu=False, W16=None, x16=None, out_b16=False, want16=False):
    shp = x.shape
    x2 = x.reshape(-1, shp[-1]) if x.dim() != 2 else x
    add2 = add.reshape(-1, add.shape[-1]) if add is not None and add.dim() != 2 else add
    if W16 is not None:                  # bf16-stored operands: 2-D callers only (the result may be a padded view)
        return LinearFn.apply(x2, W, b, add2, keep, scale, relu, W16, x16, out_b16, want16)
    y = LinearFn.apply(x2, W, b, add2, keep, scale, relu)
    return y.view(*shp[:-1], W.size(0)) if x.dim() != 2 else y


class StageMark:
    """Counts the marked tensors of one forward whose gradient has arrived; the last one announces the gradient slice `stage`
    (grads_ready): every parameter gradient DOWNSTREAM of the marked tensors is final by then."""

    def __init__(self, stage):
        self.stage, self.pending = stage, 0

    def __call__(self, x):
        if x is None or not x.requires_grad:
            return x
        self.pending += 1
        return StageMarkFn.apply(x, self)


class StageMarkFn(Function):
    """Identity whose backward tells its StageMark that this tensor's gradient is complete (it sits between a tensor and ALL its consumers)."""

    @staticmethod
    def forward(ctx, x, mark):
        ctx.mark = mark
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        m = ctx.mark
        m.pending -= 1
        if m.pending == 0:
            grads_ready(m.stage)
        return g, None


class UnitPairFn(Function):
    """The two collection units that read the SAME source rows (graph_conv.py:24-25 units 0,1 read the relation rows, :31-32 units 2,3
    the node rows; each unit = fc_rgt(fc_lft(x)), graph_conv_unit.py:28-30) as one Function:

      forward   H = x [Wl_a ; Wl_b]^T + [bl_a ; bl_b]          ONE product, N = 2 * 512: the source is read once
                y_a = H[:, :512] Wr_a^T + br_a,  y_b = H[:, 512:] Wr_b^T + br_b
      backward  dH[:, :512] = dy_a Wr_a,  dH[:, 512:] = dy_b Wr_b
                dx = dH [Wl_a ; Wl_b]                           ONE product, K = 2 * 512 (was two K = 512 products + an add)
                d[Wl_a ; Wl_b] += dH^T x                        ONE product

    `cat` = (Wl [2Lr, L], bl [2Lr], their gradient views or None, the bf16 twin of Wl or None): the two fc_lft parameters lie
    side by side in the flat buffers (AttModel._specs), so the concatenations are views.  Under compute_dtype = bf16 (`W16r` given) H
    and dH exist in bf16 only, x is taken as its bf16 copy `x16`, y is bf16 when `out_b16` (the fused aggregation reads it), d(x)
    is fp32 (it joins the other gradient contributions of the source)."""

    @staticmethod
    def forward(ctx, x, x16, cat, Wl_a, bl_a, Wl_b, bl_b, Wr_a, br_a, Wr_b, br_b, W16r, out_b16):
        Wl, bl, gWl, gbl, Wl16 = cat
        dev = x.device
        M, Lr2, L = x.size(0), Wl.size(0), Wr_a.size(0)
        Lr = Lr2 // 2
        bf = Wl16 is not None
        ctx.bf, ctx.cat, ctx.Lr = bf, cat, Lr
        ctx.params = (Wl_a, bl_a, Wl_b, bl_b, Wr_a, br_a, Wr_b, br_b)
        if bf:
            xa = x if ops.is_b16(x) else (x16 if x16 is not None else ops.as_b16(x))
            H = ops.empty_b16(M, Lr2, dev)
            ops.gemm(xa, Wl16, H, tb=True, bias=bl)
            mk = (lambda: ops.empty_b16(M, L, dev)) if out_b16 else (lambda: torch.empty(M, L, device=dev, dtype=torch.float32))
            ya, yb = mk(), mk()
            ops.gemm_pair(H[:, :Lr], H[:, Lr:], W16r[0], W16r[1], ya, yb, tb=True, bias1=br_a, bias2=br_b)      # both units' fc_rgt: one launch
            ctx.save_for_backward(xa, H, Wl16, W16r[0], W16r[1])
            return ya, yb
        H = torch.empty(M, Lr2, device=dev, dtype=torch.float32)
        ops.gemm(x, Wl, H, tb=True, bias=bl)
        ya, yb = torch.empty(M, L, device=dev, dtype=torch.float32), torch.empty(M, L, device=dev, dtype=torch.float32)
        ops.gemm_pair(H[:, :Lr], H[:, Lr:], Wr_a, Wr_b, ya, yb, tb=True, bias1=br_a, bias2=br_b)
        ctx.save_for_backward(x, H, Wl, Wr_a, Wr_b)
        return ya, yb

    @staticmethod
    def backward(ctx, dya, dyb):
        x, H, Wl, Wra, Wrb = ctx.saved_tensors
        _, _, gWl, gbl, _ = ctx.cat
        Wl_a, bl_a, Wl_b, bl_b, Wr_a, br_a, Wr_b, br_b = ctx.params
        dev, Lr, bf = x.device, ctx.Lr, ctx.bf
        if gWl is not None:
            # the concatenated gradient views were captured at FORWARD time: if a .grad was re-bound since (flatten_grads on a fresh
            # buffer, zero_grad(set_to_none=True), p.grad = None) they point into a stale buffer -- accumulate there and the gradients
            # are silently lost.  Still the live views iff the four .grad tensors alias them slot for slot; otherwise hand the
            # gradients to autograd like the non-direct branch does.
            ga, gb_, gba_, gbb_ = (_direct(q, dev) for q in (Wl_a, Wl_b, bl_a, bl_b))
            live = (ga is not None and gb_ is not None and gba_ is not None and gbb_ is not None
                    and ga.data_ptr() == gWl.data_ptr() and gb_.data_ptr() == gWl[Lr:].data_ptr()
                    and gba_.data_ptr() == gbl.data_ptr() and gbb_.data_ptr() == gbl[Lr:].data_ptr())
            if not live:
                gWl = gbl = None
        M = x.size(0)
        if bf:
            dya, dyb = (d if ops.is_b16(d) else ops.as_b16(d.contiguous()) for d in (dya, dyb))
            dH = ops.empty_b16(M, 2 * Lr, dev)
        else:
            dya, dyb = dya.contiguous(), dyb.contiguous()
            dH = torch.empty(M, 2 * Lr, device=dev, dtype=torch.float32)
        ops.gemm_pair(dya, dyb, Wra, Wrb, dH[:, :Lr], dH[:, Lr:])            # bf16: one launch for both halves (two half-filling products)
        ret = [None] * 8
        gWa, gba, gWb, gbb = _direct(Wr_a, dev), _direct(br_a, dev), _direct(Wr_b, dev), _direct(br_b, dev)
        direct = gWl is not None and DIRECT_GRADS
        if bf and ops.PAIR_LAUNCHES and direct and gbl is not None and all(g is not None for g in (gWa, gba, gWb, gbb)):
            # the two fc_rgt weight gradients: same shape, same K -> one launch (32 tiles x 8 K parts each fill half the chip); the pair's three
            # bias sums (d(y_a), d(y_b), d(H)) as one set: two launches where they are of one shape
            ops.gemm_pair(dya, dyb, H[:, :Lr], H[:, Lr:], gWa, gWb, ta=True, accum=True)
            ops.gemm(dH, x, gWl, ta=True, accum=True)
            ops.colsum_set((dya, dyb, dH), (gba, gbb, gbl.view(-1)), accumulate=True)
        else:
            for i, (W, b, dy, Hh) in enumerate(((Wr_a, br_a, dya, H[:, :Lr]), (Wr_b, br_b, dyb, H[:, Lr:]))):
                ret[4 + 2 * i], ret[5 + 2 * i] = _weight_bias_grads(dy, Hh, W.shape, _direct(W, dev), _direct(b, dev), True, True)
            dWl, dbl = _weight_bias_grads(dH, x, (2 * Lr, x.size(1)), gWl if direct else None, gbl if direct else None, True, True)
            if not direct:
                ret[0], ret[1], ret[2], ret[3] = dWl[:Lr], dbl[:Lr], dWl[Lr:], dbl[Lr:]
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, x.size(1), device=dev, dtype=torch.float32)
            ops.gemm(dH, Wl, dx)
        return (dx, None, None) + tuple(ret) + (None, None)


class ForkFn(Function):
    """x -> n aliases of x for n consumers; the backward adds their gradient contributions in ONE launch (subgc_add_n_f32) instead of
    autograd's chain of pairwise ATen adds."""

    @staticmethod
    def forward(ctx, x, n):
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        gs = [g for g in gs if g is not None]
        if not gs:
            return None, None
        return (gs[0] if len(gs) == 1 else ops.add_n(gs)), None


class SumScalarsFn(Function):
    """a + b for two scalar losses (train.py:154-156 `loss = lang_loss + gpn_loss`) through the C ABI."""

    @staticmethod
    def forward(ctx, a, b):
        return ops.add_n([a.reshape(1), b.reshape(1)]).reshape(())

    @staticmethod
    def backward(ctx, g):
        return g, g


def fork(x, n):
    return ForkFn.apply(x, n) if (n > 1 and x is not None and x.requires_grad) else (x,) * n


class GatherRowsFn(Function):
    """nn.Embedding lookup with int32 row ids (AttModel.py:376,383,385 class embeddings)."""

    @staticmethod
    def forward(ctx, table, rows):
        out = torch.empty(rows.numel(), table.size(1), device=table.device, dtype=torch.float32)
        ops.gather_rows(table, rows, out)
        ctx.save_for_backward(rows)
        ctx.shape = table.shape
        ctx.param = table
        return out

    @staticmethod
    def backward(ctx, dout):
        (rows,) = ctx.saved_tensors
        g = _direct(ctx.param, dout.device)
        if g is not None:                     # scatter straight into the flat gradient bucket (it was zeroed for this step)
            ops.scatter_add_rows(dout.contiguous(), rows, g)
            return None, None
        dt = ops.zeros(*ctx.shape, device=dout.device)
        ops.scatter_add_rows(dout.contiguous(), rows, dt)
        return dt, None


class ClassTableFn(Function):
    """Linear(Embedding[cls]) for class ids (AttModel.py:374-377, 383-386: `obj_emb_proj(sg_obj_embed(argmax))`,
    `pred_emb_prj(sg_pred_embed(argmax))`): the projection is applied to the TABLE once -- table = Emb W^T + b, [classes, L] --
    and the rows are looked up (1599 / 21 classes against thousands of rows); the backward sums d(out) per class and runs the
    two small products on the table.  Same arithmetic up to summation order."""

    @staticmethod
    def forward(ctx, emb, W, b, cls):
        C, L = emb.size(0), W.size(0)
        table = torch.empty(C, L, device=emb.device, dtype=torch.float32)
        ops.gemm(emb, W, table, tb=True, bias=b)
        out = torch.empty(cls.numel(), L, device=emb.device, dtype=torch.float32)
        ops.gather_rows(table, cls, out)
        ctx.save_for_backward(emb, W, cls)
        ctx.param_objs = (emb, W, b)
        return out

    @staticmethod
    def backward(ctx, dout):
        emb, W, cls = ctx.saved_tensors
        dev = dout.device
        C = emb.size(0)
        dtab = ops.class_sum(dout.contiguous(), cls, C)                      # [C, L]
        ge, gW, gb = (_direct(p, dev) for p in ctx.param_objs)
        demb = dW = db = None
        if ctx.needs_input_grad[0]:
            if ge is not None:
                ops.gemm(dtab, W, ge, accum=True)
            else:
                demb = torch.empty_like(emb); ops.gemm(dtab, W, demb)
        dW, db = _weight_bias_grads(dtab, emb, W.shape, gW, gb, ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        return demb, dW, db, None


# ------------------------------------------------------------------------------- GCN
class GcnNodesFn(Function):
    """nodes <- relations (graph_conv_unit.py:34-36 for units 0,1 + graph_conv.py:26 + residual)."""

    @staticmethod
    def forward(ctx, F0, F1, skip, rel_ind, ptr, edges, N):
        B, K, L = F0.shape
        F0, F1 = F0.contiguous(), F1.contiguous()
        out, act = ops.gcn_nodes_fwd(F0, F1, ptr, edges, skip.contiguous() if skip is not None else None, B, N, K, L)
        ctx.save_for_backward(act, rel_ind, ptr)
        ctx.dims = (B, N, K, L)
        ctx.has_skip = skip is not None
        return out

    @staticmethod
    def backward(ctx, dX):
        act, rel_ind, ptr = ctx.saved_tensors
        B, N, K, L = ctx.dims
        dX = dX.contiguous()
        dF0, dF1 = ops.gcn_nodes_bwd(dX, act, rel_ind, ptr, B, N, K, L)
        return dF0, dF1, (dX if ctx.has_skip else None), None, None, None, None


class GcnNodesB16Fn(Function):
    """GcnNodesFn for bf16-STORED unit outputs (compute_dtype = bf16, no BatchNorm: Sub-GC presets): the normalise-on-load kernels with the
    identity triple read the bf16 GEMM results as they are, d(y) leaves the backward in bf16 (what the Linear backward's GEMMs consume),
    and the result's bf16 copy for the next layer's GEMM comes out of the same launch (`want16`) -- no fp32 unit outputs, no cast passes."""

    @staticmethod
    def forward(ctx, y0, y1, skip, rel_ind, ptr, edges, N, want16):
        B, K, L = y0.shape
        y0, y1 = y0.contiguous(), y1.contiguous()
        ident = ops.identity_aff(L, y0.device)
        out, out16, act = ops.gcn_nodes_fwd_bn(y0, y1, ident, ident, ptr, edges, skip.contiguous() if skip is not None else None, B, N, K, L, want16)
        ctx.save_for_backward(act, rel_ind, ptr)
        ctx.dims, ctx.has_skip = (B, N, K, L), skip is not None
        if want16:
            ctx.mark_non_differentiable(out16)
            ctx.set_materialize_grads(False)
            return out, out16
        return out

    @staticmethod
    def backward(ctx, dX, *_unused):
        act, rel_ind, ptr = ctx.saved_tensors
        B, N, K, L = ctx.dims
        dX = dX.contiguous()
        dy0, dy1 = ops.gcn_nodes_bwd(dX, act, rel_ind, ptr, B, N, K, L, bf16=True)
        return dy0, dy1, (dX if ctx.has_skip else None), None, None, None, None, None


class GcnEdgesB16Fn(Function):
    """GcnEdgesFn for bf16-stored unit outputs (see GcnNodesB16Fn)."""

    @staticmethod
    def forward(ctx, y2, y3, skip, rel_ind, ptr, edges, K, want16):
        B, N, L = y2.shape
        y2, y3 = y2.contiguous(), y3.contiguous()
        ident = ops.identity_aff(L, y2.device)
        out, out16 = ops.gcn_edges_fwd_bn(y2, y3, ident, ident, rel_ind, skip.contiguous() if skip is not None else None, B, N, K, L, want16)
        ctx.save_for_backward(y2, y3, ptr, edges)
        ctx.dims, ctx.has_skip = (B, N, K, L), skip is not None
        if want16:
            ctx.mark_non_differentiable(out16)
            ctx.set_materialize_grads(False)
            return out, out16
        return out

    @staticmethod
    def backward(ctx, dP, *_unused):
        y2, y3, ptr, edges = ctx.saved_tensors
        B, N, K, L = ctx.dims
        dP = dP.contiguous()
        ident = ops.identity_aff(L, dP.device)
        dy2, dy3 = ops.gcn_edges_bwd_bn(dP, y2, y3, ident, ident, ptr, edges, B, N, K, L, bf16=True)
        return dy2, dy3, (dP if ctx.has_skip else None), None, None, None, None, None


class GcnEdgesFn(Function):
    """relations <- nodes (units 2,3; LDS-staged gather)."""

    @staticmethod
    def forward(ctx, F2, F3, skip, rel_ind, ptr, edges, K):
        B, N, L = F2.shape
        F2, F3 = F2.contiguous(), F3.contiguous()
        out = ops.gcn_edges_fwd(F2, F3, rel_ind, skip.contiguous() if skip is not None else None, B, N, K, L)
        ctx.save_for_backward(F2, F3, ptr, edges)
        ctx.dims = (B, N, K, L)
        ctx.has_skip = skip is not None
        return out

    @staticmethod
    def backward(ctx, dP):
        F2, F3, ptr, edges = ctx.saved_tensors
        B, N, K, L = ctx.dims
        dP = dP.contiguous()
        dF2, dF3 = ops.gcn_edges_bwd(dP, F2, F3, ptr, edges, B, N, K, L)
        return dF2, dF3, (dP if ctx.has_skip else None), None, None, None, None


def _bn_back(dFh, y, gamma, beta, aff, rstd):
    """BatchNorm backward of one fused source: d(normalised) -> d(raw unit output) in its storage type; the affine gradients go
    straight into the flat bucket when it exists (returns (dy, dgamma, dbeta) with None for what was written in place)."""
    M, C = y.numel() // y.size(-1), y.size(-1)
    gg, gb = _direct(gamma, dFh.device), _direct(beta, dFh.device)
    if gg is not None and gb is not None:
        return ops.bn_bwd_fused(dFh.view(M, C), y.view(M, C), gamma, aff, rstd, gg, gb, True).view(y.shape), None, None
    dg, db = torch.empty(C, device=dFh.device, dtype=torch.float32), torch.empty(C, device=dFh.device, dtype=torch.float32)
    return ops.bn_bwd_fused(dFh.view(M, C), y.view(M, C), gamma, aff, rstd, dg, db, False).view(y.shape), dg, db


def _bn_back_pair(dFa, dFb, ya, yb, ga, ba, gb_, bb, affa, affb, rstda, rstdb):
    """`_bn_back` for the two units of one fused aggregation: two launches instead of four when both write their affine gradients in place."""
    dev = dFa.device
    dirs = [_direct(t, dev) for t in (ga, ba, gb_, bb)]
    if ops.PAIR_LAUNCHES and all(d is not None for d in dirs) and ya.shape == yb.shape and ya.dtype == yb.dtype:
        M, C = ya.numel() // ya.size(-1), ya.size(-1)
        dxa, dxb = ops.bn_bwd_fused_pair(dFa.view(M, C), dFb.view(M, C), ya.view(M, C), yb.view(M, C), ga, gb_, affa, affb, rstda, rstdb,
                                         dirs[0], dirs[2], dirs[1], dirs[3], True)
        return (dxa.view(ya.shape), None, None), (dxb.view(yb.shape), None, None)
    return _bn_back(dFa, ya, ga, ba, affa, rstda), _bn_back(dFb, yb, gb_, bb, affb, rstdb)


class GcnNodesBnFn(Function):
    """nodes <- relations (GcnNodesFn) with the BatchNorm1d of the two collection units (graph_conv_unit.py:31-32) fused in: y0, y1
    are the RAW unit outputs (fp32, or bf16 under compute_dtype = bf16); the batch statistics are one pass, the aggregation kernel
    normalises on load, the backward returns d(y) in y's storage type.  `want16`: also a bf16 copy of the result (non-differentiable)."""

    @staticmethod
    def forward(ctx, y0, y1, skip, rel_ind, ptr, edges, N, g0, b0, g1, b1, stats, training, want16):
        B, K, L = y0.shape
        y0, y1 = y0.contiguous(), y1.contiguous()
        (aff0, rstd0), (aff1, rstd1) = ops.bn_stats_pair(y0.view(B * K, L), y1.view(B * K, L), g0, b0, stats[0], stats[1], g1, b1, stats[2], stats[3], training)
        out, out16, act = ops.gcn_nodes_fwd_bn(y0, y1, aff0, aff1, ptr, edges, skip.contiguous() if skip is not None else None, B, N, K, L, want16)
        ctx.save_for_backward(y0, y1, aff0, rstd0, aff1, rstd1, act, rel_ind, ptr)
        ctx.params = (g0, b0, g1, b1)
        ctx.dims, ctx.has_skip, ctx.training = (B, N, K, L), skip is not None, training
        if want16:
            ctx.mark_non_differentiable(out16)
            ctx.set_materialize_grads(False)          # else autograd hands the backward a zero tensor for the bf16 copy (an ATen fill per call)
            return out, out16
        return out

    @staticmethod
    def backward(ctx, dX, *_unused):
        y0, y1, aff0, rstd0, aff1, rstd1, act, rel_ind, ptr = ctx.saved_tensors
        if not ctx.training:
            raise RuntimeError("GcnNodesBnFn: backward in eval mode is not part of the Sub-GC path")
        B, N, K, L = ctx.dims
        g0, b0, g1, b1 = ctx.params
        dX = dX.contiguous()
        dF0, dF1 = ops.gcn_nodes_bwd(dX, act, rel_ind, ptr, B, N, K, L)
        (dy0, dg0, db0), (dy1, dg1, db1) = _bn_back_pair(dF0, dF1, y0, y1, g0, b0, g1, b1, aff0, aff1, rstd0, rstd1)
        return dy0, dy1, (dX if ctx.has_skip else None), None, None, None, None, dg0, db0, dg1, db1, None, None, None


class GcnEdgesBnFn(Function):
    """relations <- nodes (GcnEdgesFn) with the two units' BatchNorm fused in (see GcnNodesBnFn)."""

    @staticmethod
    def forward(ctx, y2, y3, skip, rel_ind, ptr, edges, K, g2, b2, g3, b3, stats, training, want16):
        B, N, L = y2.shape
        y2, y3 = y2.contiguous(), y3.contiguous()
        (aff2, rstd2), (aff3, rstd3) = ops.bn_stats_pair(y2.view(B * N, L), y3.view(B * N, L), g2, b2, stats[0], stats[1], g3, b3, stats[2], stats[3], training)
        out, out16 = ops.gcn_edges_fwd_bn(y2, y3, aff2, aff3, rel_ind, skip.contiguous() if skip is not None else None, B, N, K, L, want16)
        ctx.save_for_backward(y2, y3, aff2, rstd2, aff3, rstd3, ptr, edges)
        ctx.params = (g2, b2, g3, b3)
        ctx.dims, ctx.has_skip, ctx.training = (B, N, K, L), skip is not None, training
        if want16:
            ctx.mark_non_differentiable(out16)
            ctx.set_materialize_grads(False)          # else autograd hands the backward a zero tensor for the bf16 copy (an ATen fill per call)
            return out, out16
        return out

    @staticmethod
    def backward(ctx, dP, *_unused):
        y2, y3, aff2, rstd2, aff3, rstd3, ptr, edges = ctx.saved_tensors
        if not ctx.training:
            raise RuntimeError("GcnEdgesBnFn: backward in eval mode is not part of the Sub-GC path")
        B, N, K, L = ctx.dims
        g2, b2, g3, b3 = ctx.params
        dP = dP.contiguous()
        dF2, dF3 = ops.gcn_edges_bwd_bn(dP, y2, y3, aff2, aff3, ptr, edges, B, N, K, L)
        (dy2, dg2, db2), (dy3, dg3, db3) = _bn_back_pair(dF2, dF3, y2, y3, g2, b2, g3, b3, aff2, aff3, rstd2, rstd3)
        return dy2, dy3, (dP if ctx.has_skip else None), None, None, None, None, dg2, db2, dg3, db3, None, None, None


class BatchNormFn(Function):
    """nn.BatchNorm1d over rows (graph_conv_unit.py:31-32); running stats updated in place."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training):
        x = x.contiguous()
        y, sm, sr = ops.bn_fwd(x, gamma, beta, running_mean, running_var, training)
        ctx.training = training
        if training:
            ctx.save_for_backward(x, gamma, sm, sr)
        else:
            ctx.save_for_backward(x, gamma, running_mean, running_var)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, a, b = ctx.saved_tensors
        if not ctx.training:
            raise RuntimeError("BatchNormFn: backward in eval mode is not part of the Sub-GC path")
        dx, dg, db = ops.bn_bwd(dy.contiguous(), x, gamma, a, b)
        return dx, dg, db, None, None, None


# ------------------------------------------------------------------------------- sGPN
class SubgraphPoolFn(Function):
    """Fused gather + diagonal mask + max/mean pooling (gpn.py:152-185)."""

    @staticmethod
    def forward(ctx, X, idx, w, w_g, w_i, denom, img, N):
        Bn, L = X.shape
        G = idx.size(0)
        X = X.contiguous()
        out, am = ops.pool_fwd(X, idx, idx.stride(0), w, w_g, w_i, denom, img, G, N, L)
        ctx.save_for_backward(idx, w, denom, img, am)
        ctx.meta = (w_g, w_i, G, N, L, Bn)
        return out

    @staticmethod
    def backward(ctx, dout):
        idx, w, denom, img, am = ctx.saved_tensors
        w_g, w_i, G, N, L, Bn = ctx.meta
        dX = ops.zeros(Bn, L, device=dout.device)
        ops.pool_bwd(dout.contiguous(), idx, idx.stride(0), w, w_g, w_i, denom, img, am, dX, G, N, L)
        return dX, None, None, None, None, None, None, None


class GpnScoreFn(Function):
    """Linear(H->1) + sigmoid + BCE(mean) after the (ReLU, dropout) hidden layer (gpn.py:54-57)."""

    @staticmethod
    def forward(ctx, hid, w2, b2, keep, scale):
        hid = hid.contiguous()
        score, loss = ops.gpn_score_fwd(hid, keep, scale, w2, b2)
        ctx.save_for_backward(hid, w2, score, keep)
        ctx.scale = scale
        ctx.param_objs = (w2, b2)
        ctx.mark_non_differentiable(score)
        ctx.set_materialize_grads(False)      # no zero tensor for the (non-differentiable) score output
        return score, loss

    @staticmethod
    def backward(ctx, dscore, dloss):
        if dloss is None:
            return None, None, None, None, None
        hid, w2, score, keep = ctx.saved_tensors
        dhid, dw2, db2 = ops.gpn_score_bwd(hid, keep, ctx.scale, w2, score, dloss.contiguous())
        gw, gb = (_direct(p, hid.device) for p in ctx.param_objs)
        if gw is not None and gb is not None:           # accumulate into the flat bucket with one C-ABI launch each instead of autograd's `+=`
            ops.copy2d(dw2.view(1, -1), gw.view(1, -1), accumulate=True)
            ops.copy2d(db2.view(1, -1), gb.view(1, -1), accumulate=True)
            return dhid, None, None, None, None
        return dhid, dw2, db2, None, None


class MaskedNLLFn(Function):
    """LanguageModelCriterion (misc/utils.py:115-124)."""

    @staticmethod
    def forward(ctx, logp, target, mask):
        logp = logp.contiguous()
        loss, scratch = ops.masked_nll_fwd(logp, target, mask)
        ctx.save_for_backward(target, mask, scratch)
        ctx.shape = logp.shape
        return loss

    @staticmethod
    def backward(ctx, dloss):
        target, mask, scratch = ctx.saved_tensors
        S, T, V = ctx.shape
        return ops.masked_nll_bwd(target, mask, scratch, dloss.contiguous(), S, T, V), None, None


# ------------------------------------------------------------------------------- decoder
DIRECT_GRADS = True               # accumulate parameter gradients straight into existing .grad buffers
DEFER_DU = True                   # per-sentence attention sets: d(u) formed once after the BPTT loop (subgc_attn_du_accum) instead of read-modify-written per step
FOLD_BIAS_SUMS = True             # a layer's bias gradient rides its weight-gradient product (ops.wgrad); False: separate column-sum launches
on_grads_ready = None             # callback(stage) set by parallel.GradBucketReducer: the gradient slice `stage` ("logit", "recurrent",
                                  # "prepare"; AttModel.grad_buckets) is final and may be all-reduced while the backward goes on
trace = None                      # tests: a list that receives ("bptt_begin", steps) / ("bptt_end",) / ("ready", stage) / ("issue", bucket) in host order


def note(*event):
    if trace is not None:
        trace.append(event)


def grads_ready(stage):
    """Called by the decoder backwards when every kernel that writes the gradient slice `stage` has been enqueued."""
    note("ready", stage)
    if on_grads_ready is not None:
        on_grads_ready(stage)

PARAM_ORDER = (
    "fc_embed.0.weight", "fc_embed.0.bias", "fc_embed.2.weight", "fc_embed.2.bias",
    "att_embed.0.weight", "att_embed.0.bias", "ctx2att.weight", "ctx2att.bias", "embed.0.weight",
    "core.att_lstm.weight_ih", "core.att_lstm.weight_hh", "core.att_lstm.bias_ih", "core.att_lstm.bias_hh",
    "core.lang_lstm.weight_ih", "core.lang_lstm.weight_hh", "core.lang_lstm.bias_ih", "core.lang_lstm.bias_hh",
    "core.attention.h2att.weight", "core.attention.h2att.bias", "core.attention.alpha_net.weight",
    "core.attention.alpha_net.bias", "logit.weight", "logit.bias",
)


def bf16_twins(P, W16):
    """Per-parameter GEMM-operand form for the decoder Functions: the bf16 twin where one exists (2-D weights with 16-byte
    rows), the fp32 parameter otherwise (biases, alpha_net).  `W16` = list aligned with P (None entries allowed) or None."""
    if W16 is None:
        return list(P), False
    missing = [n for n, w, p in zip(PARAM_ORDER, W16, P) if p.dim() == 2 and p.size(0) > 1 and w is None]
    if missing:
        raise ops.SubgcError(f"compute_dtype=bf16: no 16-byte-row bf16 twin for {missing}")
    return [w if w is not None else p for w, p in zip(W16, P)], True


class Prepared:
    """Loop-invariant decoder state (AttModel._prepare_feature + pack): shared by train and decode.
    `W` = bf16_twins(P, ...)[0] switches the five products to bf16-stored operands: the fp32 results the pointwise kernels
    read (f1, f) are kept, every tensor a GEMM reads gets a bf16 twin written by its producer (`*16`), and the node features
    u, v are bf16 only (the attention kernels read them as such)."""

    dedup = False

    def __init__(self, fc_in, X_nodes, lens, idx, img, N, P, keep_fc, keep_att, scale, W=None, into=None, dedup=False):
        """`into` (decode, fp32 only): a namespace of fixed-address buffers f [S, R], u [S*N, A], v [S*N, R], off [S] that receive the
        results in place (the captured decode graphs read them there); rows past the packed total keep whatever they held -- the
        attention kernels never read them.
        `dedup` (Full-GC training under dropout): every node row of an image is in the attention set of each of its 5 sentences
        (AttModel.py:140-149 on the x5 replicated rows of gcn_backbone.py:50-51), so relu(att_embed(x)) is computed ONCE per node row
        ([B*N] instead of [5*B*N] rows in the forward product, the weight gradient and the data gradient) and each sentence's copy is
        gathered with ITS OWN dropout keep-mask row (subgc_gather_rows_keep): the reference's five independent masks, a fifth of the
        att_embed FLOPs.  v, u and everything downstream are per sentence, as in the replicated path."""
        (fc0_w, fc0_b, fc2_w, fc2_b, att_w, att_b, c2a_w, c2a_b) = P[:8]
        dev = fc_in.device
        S = fc_in.size(0)
        self.S, self.N = S, N
        self.off, self.total, self.src_row, self.sent_of = ops.pack_rows(lens, idx, img, S, N, off=None if into is None else into.off)
        self.lens = lens
        MR = S * N
        L = X_nodes.size(1)
        new = lambda r, c: torch.empty(r, c, device=dev, dtype=torch.float32)
        self.dedup = bool(dedup) and into is None
        if W is not None and ops.is_b16(W[0]):
            if into is not None:
                raise ops.SubgcError("Prepared(into=...) is the fp32 decode form")
            w0, w2, wa, wc = W[0], W[2], W[4], W[6]
            self.fc16 = ops.as_b16(fc_in)
            self.Xg = None                                     # the gathered node rows only feed GEMMs: bf16 only
            if not self.dedup:
                self.Xg16 = ops.empty_b16(MR, L, dev)
                ops.gather_rows(X_nodes, self.src_row, self.Xg16, m_dev=self.total)
            self.f1, self.f116 = new(S, fc0_w.size(0)), ops.empty_b16(S, fc0_w.size(0), dev)
            ops.gemm(self.fc16, w0, self.f1, tb=True, bias=fc0_b, relu=True, out16=self.f116)
            self.f, self.f16 = new(S, fc2_w.size(0)), ops.empty_b16(S, fc2_w.size(0), dev)
            ops.gemm(self.f116, w2, self.f, tb=True, bias=fc2_b, relu=True, keep=keep_fc, keep_scale=scale, out16=self.f16)
            # the node features v = relu(att_embed(X)) and u = ctx2att(v) exist ONLY as the bf16 tensors their GEMMs write: the
            # attention kernels read them as such (half the bytes of the two largest per-step reads), the ReLU backward takes its
            # sign test from the bf16 v (same exponent range as fp32), and no fp32 copy is ever stored
            self.v16 = ops.empty_b16(MR, att_w.size(0), dev, zero=True)
            if self.dedup:
                self.Xu16 = ops.as_b16(X_nodes)                # the unique node rows
                # ONE rounding, like the replicated path (its GEMM epilogue rounds relu(.) * keep * scale from the fp32 accumulator): a bf16
                # intermediate is exact only when the scale is a power of two (drop_prob_lm = 0.5, the presets' value) -- otherwise the
                # unique rows stay fp32 until the masked gather rounds them
                exact16 = keep_att is None or math.frexp(float(scale))[0] == 0.5
                r_u = ops.empty_b16(X_nodes.size(0), att_w.size(0), dev) if exact16 else new(X_nodes.size(0), att_w.size(0))
                ops.gemm(self.Xu16, wa, r_u, tb=True, bias=att_b, relu=True)
                ops.gather_rows_keep(r_u, self.src_row, keep_att, scale, self.v16, m_dev=self.total)
            else:
                ops.gemm(self.Xg16, wa, self.v16, tb=True, bias=att_b, relu=True, keep=keep_att, keep_scale=scale, m_dev=self.total)
            self.v = self.v16
            self.u = ops.empty_b16(MR, c2a_w.size(0), dev)
            ops.gemm(self.v16, wc, self.u, tb=True, bias=c2a_b, m_dev=self.total)
            return
        if not self.dedup:
            self.Xg = torch.empty(MR, L, device=dev, dtype=torch.float32)
            ops.gather_rows(X_nodes, self.src_row, self.Xg, m_dev=self.total)
        self.f1 = torch.empty(S, fc0_w.size(0), device=dev, dtype=torch.float32)
        ops.gemm(fc_in, fc0_w, self.f1, tb=True, bias=fc0_b, relu=True)
        self.f = torch.empty(S, fc2_w.size(0), device=dev, dtype=torch.float32) if into is None else into.f
        ops.gemm(self.f1, fc2_w, self.f, tb=True, bias=fc2_b, relu=True, keep=keep_fc, keep_scale=scale)
        self.v = ops.zeros(MR, att_w.size(0), device=dev) if into is None else into.v[:MR]
        if self.dedup:
            r = new(X_nodes.size(0), att_w.size(0))
            ops.gemm(X_nodes, att_w, r, tb=True, bias=att_b, relu=True)
            ops.gather_rows_keep(r, self.src_row, keep_att, scale, self.v, m_dev=self.total)
        else:
            ops.gemm(self.Xg, att_w, self.v, tb=True, bias=att_b, relu=True, keep=keep_att, keep_scale=scale, m_dev=self.total)
        self.u = torch.empty(MR, c2a_w.size(0), device=dev, dtype=torch.float32) if into is None else into.u[:MR]
        ops.gemm(self.v, c2a_w, self.u, tb=True, bias=c2a_b, m_dev=self.total)


    # the per-step attention calls, so that the decoder Functions do not care whether the sets are per sentence or shared per image
    shared = False

    def attn_fwd(self, ah, w_a, b_a, lens, ctx, alpha, m, A, R, q=None):
        ops.attn_fwd(self.u, self.v, ah, w_a, b_a, self.off, lens, ctx, alpha, m, A, R, q=q)

    def attn_bwd(self, ah, w_a, lens, alpha, dctx, dah, du, dv, dwa, dba, m, A, R, dctx_keep, de_keep=None):
        ops.attn_bwd(self.u, self.v, ah, w_a, self.off, lens, alpha, dctx, dah, None if de_keep is not None else du, dv, dwa, dba, m, A, R,
                     dctx_keep=dctx_keep, de_keep=de_keep)

    def dv_accum(self, alpha, dctx, step_off, T, lens, dv, S, R):
        ops.attn_dv_accum(alpha, dctx, step_off, T, self.off, lens, dv, S, R)

    def du_accum(self, ah, de, step_off, T, lens, w_a, du, S, A):
        """deferred d(u): one pass over the kept d(e) / query rows of all steps (subgc_attn_du_accum), every d(u) row written once"""
        ops.attn_du_accum(self.u, ah, de, step_off, T, self.off, lens, w_a, du, S, A)

    def recur_fields(self):
        """The attention-set fields of ops.Recurrence (per-sentence sets)."""
        return dict(shared=0, u=self.u, v=self.v, off=self.off, uv_b16=int(ops.is_b16(self.u)))

    def new_du(self, A):
        """Zeroed accumulator of d(u) for the backward's time loop."""
        return ops.zeros(self.u.size(0), A, device=self.u.device)

    def finish_du(self, du):
        return du


class PreparedShared(Prepared):
    """`Prepared` for attention sets that are SHARED by the g sentences of an image (Full-GC, AttModel.py:140-149: every sentence
    attends over all of its image's nodes; the reference replicates the node features g = 5 times, gcn_backbone.py:50-51).
    v = relu(att_embed(X)) and u = ctx2att(v) are computed ONCE per image over all B*N node rows -- no gather, no ragged packing,
    a fifth of the rows in the four products -- and the grouped attention kernels serve an image's sentences from one workgroup.
    `rows` int32 [B*g]: position of every sentence in the per-step row arrays (identity, or the packed decoder's sorted rank).
    With dropout on, the g sentences of an image share ONE keep-mask on v (the reference draws one per replicated copy): each
    sentence's marginal distribution is unchanged, the masks are tied within the image."""

    shared = True

    def __init__(self, fc_in, X_nodes, lens, rows, B, g, N, P, keep_fc, keep_att, scale, W=None):
        (fc0_w, fc0_b, fc2_w, fc2_b, att_w, att_b, c2a_w, c2a_b) = P[:8]
        dev = fc_in.device
        S = fc_in.size(0)
        self.S, self.N, self.B, self.g = S, N, B, g
        self.lens, self.rows = lens, rows
        self.total = self.off = self.src_row = None
        MR = B * N
        keep_att = None if keep_att is None else keep_att[:MR]
        new = lambda r, c: torch.empty(r, c, device=dev, dtype=torch.float32)
        if W is not None and ops.is_b16(W[0]):
            w0, w2, wa, wc = W[0], W[2], W[4], W[6]
            self.fc16 = ops.as_b16(fc_in)
            self.Xg = None
            self.Xg16 = ops.as_b16(X_nodes)
            self.f1, self.f116 = new(S, fc0_w.size(0)), ops.empty_b16(S, fc0_w.size(0), dev)
            ops.gemm(self.fc16, w0, self.f1, tb=True, bias=fc0_b, relu=True, out16=self.f116)
            self.f, self.f16 = new(S, fc2_w.size(0)), ops.empty_b16(S, fc2_w.size(0), dev)
            ops.gemm(self.f116, w2, self.f, tb=True, bias=fc2_b, relu=True, keep=keep_fc, keep_scale=scale, out16=self.f16)
            self.v16 = ops.empty_b16(MR, att_w.size(0), dev)
            ops.gemm(self.Xg16, wa, self.v16, tb=True, bias=att_b, relu=True, keep=keep_att, keep_scale=scale)
            self.v = self.v16
            self.u = ops.empty_b16(MR, c2a_w.size(0), dev)
            ops.gemm(self.v16, wc, self.u, tb=True, bias=c2a_b)
            return
        self.Xg = X_nodes
        self.f1 = new(S, fc0_w.size(0))
        ops.gemm(fc_in, fc0_w, self.f1, tb=True, bias=fc0_b, relu=True)
        self.f = new(S, fc2_w.size(0))
        ops.gemm(self.f1, fc2_w, self.f, tb=True, bias=fc2_b, relu=True, keep=keep_fc, keep_scale=scale)
        self.v = new(MR, att_w.size(0))
        ops.gemm(self.Xg, att_w, self.v, tb=True, bias=att_b, relu=True, keep=keep_att, keep_scale=scale)
        self.u = new(MR, c2a_w.size(0))
        ops.gemm(self.v, c2a_w, self.u, tb=True, bias=c2a_b)

    def attn_fwd(self, ah, w_a, b_a, lens, ctx, alpha, m, A, R, q=None):
        ops.attn_fwd_group(self.u, self.v, ah, w_a, b_a, self.rows, lens, m, self.B, self.g, self.N, ctx, alpha, A, R, q=q)

    def attn_bwd(self, ah, w_a, lens, alpha, dctx, dah, du, dv, dwa, dba, m, A, R, dctx_keep, de_keep=None):
        if dv is not None or de_keep is not None:
            raise ops.SubgcError("shared attention sets: d(v) is always deferred to dv_accum, d(u) never")
        ops.attn_bwd_group(self.u, self.v, ah, w_a, self.rows, lens, m, self.B, self.g, self.N, alpha, dctx, dah, du, dwa, dba, A, R, dctx_keep)

    def dv_accum(self, alpha, dctx, step_off, T, lens, dv, S, R):
        ops.attn_dv_accum_group(alpha, dctx, step_off, T, self.rows, self.B, self.g, self.N, dv, R)

    def recur_fields(self):
        return dict(shared=1, u=self.u, v=self.v, rows_map=self.rows, B=self.B, g=self.g, Nn=self.N, uv_b16=int(ops.is_b16(self.u)))

    def new_du(self, A):
        """One zeroed d(u) plane per workgroup that serves an image (subgc_attn_group_du_planes)."""
        return ops.zeros(ops.attn_group_du_planes(self.g), self.u.size(0), A, device=self.u.device)

    def finish_du(self, du):
        return du[0] if du.size(0) == 1 else ops.add_n([du[k] for k in range(du.size(0))])


PACKED_MAX_STEPS, PACKED_MAX_SENTENCES = 63, 16384      # csrc/plan.hip: subgc_live_plan / subgc_packed_rows


def shared_sets_ok(g, N, A, R, T=1):
    """Can the grouped attention kernels serve this shape? (csrc/attention_group.hip limits; T <= 64 is attn_dv_accum_group's step
    table).  Shapes outside fall back to the per-sentence kernels on replicated rows (attention_vec.hip), not to an error."""
    return 1 <= g <= 8 and N <= 128 and A % 4 == 0 and R % 4 == 0 and A <= 512 and R <= 1024 and T <= 64


def make_prepared(meta, fc_in, X_nodes, lens, idx, img, N, P, k_fc, k_att, scale, W, rows=None):
    sh = meta.get("shared")
    if sh is None:
        return Prepared(fc_in, X_nodes, lens, idx, img, N, P, k_fc, k_att, scale, W, dedup=bool(meta.get("dedup_att_embed")))
    if rows is None:
        rows = sh["rows"]
    return PreparedShared(fc_in, X_nodes, lens, rows, sh["B"], sh["g"], N, P, k_fc, k_att, scale, W)


def _cat_weights(w_ih_part, w_hh):
    """[W_ih(:, cols) | W_hh] as one K-contiguous operand so a recurrent step is ONE GEMM (fp32 or bf16 like its parts)."""
    R4, a = w_ih_part.shape
    out = ops.act_padded((R4, a + w_hh.size(1)), w_hh.device, ops.is_b16(w_hh))      # row pitch a multiple of 128 bytes (ops.PITCH)
    ops.copy2d(w_ih_part, out[:, :a])
    ops.copy2d(w_hh, out[:, a:])
    return out


_STEP_OFFSETS = {}


def _step_offsets(T, S, dev):
    """int32 [T + 1] = t * S on the device (the unpacked decoder's step table for dv_accum), built once per (T, S, device)."""
    key = (T, S, str(dev))
    t = _STEP_OFFSETS.get(key)
    if t is None:
        if len(_STEP_OFFSETS) > 32:
            _STEP_OFFSETS.clear()
        t = _STEP_OFFSETS[key] = torch.tensor([i * S for i in range(T + 1)], dtype=torch.int32).to(dev)
    return t


def tap_prepared(tap, pr, c2a_b):
    """Debug tap (model.tap, tests only): the prepared features in the reference's layout -- p_fc [S, R], p_att [S, n_max, R] (pad
    rows 0), pp_att [S, n_max, A] (pad rows = ctx2att's bias: the reference projects the zero rows, AttModel.py:364-366).
    Plumbing in torch, never on the product path."""
    tap["p_fc"] = pr.f.detach().clone()
    lens = pr.lens.long().cpu()
    S, n_max = pr.S, int(lens.max()) if lens.numel() else 0
    v, u = pr.v.float(), pr.u.float()
    if pr.shared:
        B, g, N = pr.B, pr.g, pr.N
        img = torch.arange(S, device=v.device) // g
        tap["p_att"] = v.view(B, N, -1)[img][:, :n_max].clone()
        tap["pp_att"] = u.view(B, N, -1)[img][:, :n_max].clone()
        return
    off = pr.off.long().cpu()
    pa = v.new_zeros(S, n_max, v.size(1))
    pp = c2a_b.detach().float().view(1, 1, -1).expand(S, n_max, -1).clone()
    for s_ in range(S):
        n, o = int(lens[s_]), int(off[s_])
        pa[s_, :n] = v[o:o + n]; pp[s_, :n] = u[o:o + n]
    tap["p_att"], tap["pp_att"] = pa, pp


class DecoderFn(Function):
    """Teacher-forced attention-LSTM decoder -> log-probabilities [S, T, V+1]."""

    @staticmethod
    def forward(ctx, meta, labels, fc_in, X_nodes, lens, idx, img, *P):
        N, p_drop, masks = meta["N"], meta["p"], meta.get("masks") or {}
        dev = fc_in.device
        (fc0_w, fc0_b, fc2_w, fc2_b, att_w, att_b, c2a_w, c2a_b, emb, w1i, w1h, b1i, b1h, w2i, w2h, b2i, b2h,
         h2a_w, h2a_b, an_w, an_b, lg_w, lg_b) = P
        S, T = fc_in.size(0), labels.size(1) - 1
        R, E, A, V1 = w1h.size(1), emb.size(1), h2a_w.size(0), lg_w.size(0)
        scale = 1.0 / (1.0 - p_drop) if p_drop > 0 else 1.0
        k_fc, k_att, k_xt, k_out = (masks.get(k) for k in ("fc", "att", "xt", "out"))
        fc_in = fc_in.contiguous(); X_nodes = X_nodes.contiguous()
        # W[i]: GEMM-operand form of parameter i -- its bf16 twin under compute_dtype = bf16 (meta["W16"]), else the parameter
        W, bf = bf16_twins(P, meta.get("W16"))
        act = lambda *shape, zero=False: ops.act_buffer(shape, dev, bf, zero)        # buffers that only GEMMs read
        pr = make_prepared(meta, fc_in, X_nodes, lens, idx, img, N, P, k_fc, k_att, scale, W if bf else None)
        f_op = pr.f16 if bf else pr.f

        # scheduled sampling (AttModel.py:157-167): meta["ss"] = (prob, sel_u [T,S], u [T,S]); the input word of step
        # t >= 1 then depends on step t-1's distribution, so embeddings / x->gates / logits are produced step by step.
        # The sampled words are constants of the graph (the reference detaches them): the backward is unchanged.
        ss = meta.get("ss")
        tokens = labels if ss is None else labels[:, :T].clone()
        xt = act(T, S, E)
        Gx = torch.empty(T * S, 4 * R, device=dev, dtype=torch.float32)
        if ss is None:
            for t in range(T):
                ops.embed_fwd(emb, labels[:, t], labels.stride(0), None if k_xt is None else k_xt[t], scale, xt[t])
            ops.gemm(xt.view(T * S, E), W[9][:, 2 * R:], Gx, tb=True)
        Gf = torch.empty(S, 4 * R, device=dev, dtype=torch.float32)
        ops.gemm(f_op, W[9][:, R:2 * R], Gf, tb=True)
        Wc1 = _cat_weights(W[9][:, :R], W[10])        # [4R, 2R]  x [h2_prev | h1_prev]
        Wc2 = _cat_weights(W[13], W[14])              # [4R, 3R]  x [ctx | h1 | h2_prev]

        H1 = ops.act_padded((T + 1, S, 2 * R), dev, bf, zero_rows=(T + 1) * S)      # 128-byte row pitch: they are the recurrent GEMMs' operands
        H2 = ops.act_padded((T + 1, S, 3 * R), dev, bf, zero_rows=(T + 1) * S)
        C1 = ops.zeros(T + 1, S, R, device=dev)
        C2 = ops.zeros(T + 1, S, R, device=dev)
        Hout = ops.act_padded((S, T, R), dev, bf)
        G1 = torch.empty(T, S, 4 * R, device=dev, dtype=torch.float32)
        G2 = torch.empty(T, S, 4 * R, device=dev, dtype=torch.float32)
        AH = torch.empty(T, S, A, device=dev, dtype=torch.float32)
        AL = torch.empty(T, S, N, device=dev, dtype=torch.float32)
        pre = torch.empty(S, 4 * R, device=dev, dtype=torch.float32)
        QP = torch.empty(8 * S * A, device=dev, dtype=torch.float32)           # split-K planes of the per-step query product
        Gx3 = Gx.view(T, S, 4 * R)
        logits = torch.empty(S * T, V1, device=dev, dtype=torch.float32)
        logits3 = logits.view(S, T, V1)
        rec = None
        if ss is None and T > 0 and ops.recurrence_ok():
            # the T steps as ONE library call (subgc_recurrence_fwd; see functions_packed.py): every step owns all S rows
            ldh = Hout.stride(1)
            rec = ops.Recurrence(S=S, T=T, R=R, A=A, n_alpha=N, bf16=int(bf), gemm_flags=ops.GEMM_MODES[ops.gemm_mode.current], keep_scale=float(scale),
                                 m=[S] * (T + 1), row0=[t * S for t in range(T + 1)], hout_off=[t * ldh for t in range(T)], ld_hout=Hout.stride(0),
                                 H1=H1, ldH1=H1.stride(1), H2=H2, ldH2=H2.stride(1), Hout=Hout, Wc1=Wc1, ldW1=ops.ld(Wc1), Wc2=Wc2, ldW2=ops.ld(Wc2),
                                 Wq=W[17], ldWq=ops.ld(W[17]), b1i=b1i, b1h=b1h, b2i=b2i, b2h=b2h, bq=h2a_b, pre=pre, Gx=Gx, Gf=Gf, C1=C1, C2=C2,
                                 G1=G1, G2=G2, AH=AH, AL=AL, k_out=k_out, QP=QP, qp_bytes=QP.numel() * 4, w_a=an_w, b_a=an_b, lens=lens,
                                 **pr.recur_fields())
            ops.recurrence_fwd(rec, H1)
        for t in range(T if rec is None else 0):
            if ss is not None:
                if t >= 1:
                    ops.gemm(Hout[:, t - 1, :], W[21], logits3[:, t - 1, :], tb=True, bias=lg_b)    # raw logits of the previous step
                    ops.multinomial_rows_(logits3[:, t - 1, :], ss[2][t], ss[1][t], ss[0], tokens[:, t])
                ops.embed_fwd(emb, tokens[:, t], tokens.stride(0), None if k_xt is None else k_xt[t], scale, xt[t])
                ops.gemm(xt[t], W[9][:, 2 * R:], Gx3[t], tb=True)
            ops.lstm_fwd_gemm(H1[t], Wc1, pre, Gx3[t], Gf, b1i, b1h, C1[t], C1[t + 1], H2[t][:, R:2 * R], H1[t + 1][:, R:], None, 1.0, None,
                              G1[t], S, R)
            nq, sq = ops.gemm_planes(H2[t][:, R:2 * R], W[17], QP, tb=True)      # the query product stays as split-K planes: the attention
            pr.attn_fwd(AH[t], an_w, an_b, lens, H2[t][:, :R], AL[t], S, A, R, q=(QP, nq, sq, h2a_b))     # kernel sums them (+ bias) into AH[t]
            ops.lstm_fwd_gemm(H2[t], Wc2, pre, None, None, b2i, b2h, C2[t], C2[t + 1], H1[t + 1][:, :R], H2[t + 1][:, 2 * R:],
                              None if k_out is None else k_out[t], scale, Hout[:, t, :], G2[t], S, R)
        if ss is None:
            ops.gemm(ops.flat_rows(Hout), W[21], logits, tb=True, bias=lg_b)
        else:
            ops.gemm(Hout[:, T - 1, :], W[21], logits3[:, T - 1, :], tb=True, bias=lg_b)
        active = ops.step_active(labels, T)
        ops.log_softmax_rows_(logits, active)

        tap = meta.get("tap")
        if tap is not None:                          # debug only (model.tap): per-step intermediates in the oracle's / golden files' names
            tap_prepared(tap, pr, c2a_b)
            f32 = lambda t: t.float().clone()
            tap.update(step_h_att=f32(H2[:T, :, R:2 * R]), step_ctx=f32(H2[:T, :, :R]), step_h_lang=f32(H1[1:T + 1, :, :R]),
                       step_c_att=C1[1:].clone(), step_c_lang=C2[1:].clone(), step_alpha=AL.clone(),
                       step_logp=logits3.permute(1, 0, 2).clone(), fed_tokens=tokens[:, :T].t().clone())
        crit = meta.get("crit")                      # (target [S,T] view, mask [S,T] view): criterion fused in
        if crit is not None:
            loss, nll_scratch = ops.masked_nll_fwd(logits.view(S, T, V1), crit[0], crit[1])
        else:
            loss, nll_scratch = torch.zeros((), device=dev), None
        ctx.meta = (N, scale, S, T, R, E, A, V1)
        ctx.masks = (k_xt, k_out)
        ctx.pr = pr
        ctx.crit = crit
        ctx.nll_scratch = nll_scratch
        ctx.params, ctx.W, ctx.bf = P, W, bf
        ctx.set_materialize_grads(False)
        ctx.tokens_used = tokens
        ctx.save_for_backward(labels, fc_in, X_nodes, lens, logits, active, xt, Gf, Wc1, Wc2, H1, H2, C1, C2, Hout, G1, G2, AH, AL)
        return logits.view(S, T, V1), loss

    @staticmethod
    def backward(ctx, dout, dloss):
        N, scale, S, T, R, E, A, V1 = ctx.meta
        k_xt, k_out = ctx.masks
        pr = ctx.pr
        (labels, fc_in, X_nodes, lens, logp, active, xt, Gf, Wc1, Wc2, H1, H2, C1, C2, Hout, G1, G2, AH, AL) = ctx.saved_tensors
        P, W, bf = ctx.params, ctx.W, ctx.bf
        (fc0_w, fc0_b, fc2_w, fc2_b, att_w, att_b, c2a_w, c2a_b, emb, w1i, w1h, b1i, b1h, w2i, w2h, b2i, b2h,
         h2a_w, h2a_b, an_w, an_b, lg_w, lg_b) = P
        dev = logp.device
        new = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
        zer = lambda *s: ops.zeros(*s, device=dev)
        act = lambda *shape: ops.act_buffer(shape, dev, bf)                 # gradients that only GEMMs (and bias sums) read
        opnd = (lambda t, m_dev=None: ops.as_b16(t, m_dev)) if bf else (lambda t, m_dev=None: t)    # fp32 tensor -> GEMM operand

        # Gradient destinations: when a parameter already owns a .grad buffer (the flat bucket of
        # AttModel.flatten_grads) the kernels accumulate straight into it and autograd gets None --
        # no temporary, no separate "+=" pass over 280 MB.
        dst, acc, ret = [], [], []
        ops.GRAD_WRITES[0] += 1
        for prm in P:
            g = prm.grad if DIRECT_GRADS else None
            ok = g is not None and g.is_contiguous() and g.dtype == torch.float32 and g.device == dev
            dst.append(g if ok else None); acc.append(ok); ret.append(None)

        def out_for(i, zero=False):
            if dst[i] is None:
                dst[i] = zer(*P[i].shape) if zero else new(*P[i].shape)
                ret[i] = dst[i]
            return dst[i]

        def wgrad(i, dy, x, cols=None, bias=None, m_dev=None):      # dW_i[:, cols] (+)= dy^T x  [and db_bias (+)= column sums of dy, same launch]
            o = out_for(i)
            o = o if cols is None else o[:, cols[0]:cols[1]]
            if bias is None or not FOLD_BIAS_SUMS:
                ops.gemm(dy, x, o, ta=True, accum=acc[i], m_dev=m_dev)
                if bias is not None:
                    bgrad(bias, dy, m_dev=m_dev)
            else:
                ops.wgrad(dy, x, o, out_for(bias).view(-1), accum=acc[i], db_accum=acc[bias], m_dev=m_dev)

        def bgrad(i, x, m_dev=None, also=None):          # db_i (+)= column sums of x  (also: a second bias with the same gradient)
            if also is None:
                ops.colsum(x, out=out_for(i), accumulate=acc[i], m_dev=m_dev)
                return
            tmp = ops.colsum(x, m_dev=m_dev).view(1, -1)
            for j in (i, also):
                ops.copy2d(tmp, out_for(j).view(1, -1), accumulate=acc[j])

        if ctx.crit is not None and dloss is not None and dout is None:
            # d(logits) only feeds the logit layer's two gradient GEMMs and its bias sum: written bf16 under compute_dtype = bf16
            dlogits = ops.empty_b16(S * T, V1, dev) if bf else new(S * T, V1)
            ops.nll_logsoftmax_bwd(logp, ctx.crit[0], ctx.crit[1], ctx.nll_scratch, dloss.contiguous(), dlogits, active, S, T, V1)
        elif dout is not None:
            dlogits = new(S * T, V1)
            if ctx.crit is not None and dloss is not None:   # the log-probabilities were ALSO used elsewhere
                ops.nll_logsoftmax_bwd(logp, ctx.crit[0], ctx.crit[1], ctx.nll_scratch, dloss.contiguous(), dlogits, active, S, T, V1)
                extra = new(S * T, V1)
                ops.log_softmax_rows_bwd(logp, dout.contiguous().view(S * T, V1), extra, active)
                dlogits.add_(extra)
            else:
                ops.log_softmax_rows_bwd(logp, dout.contiguous().view(S * T, V1), dlogits, active)
            dlogits = opnd(dlogits)
        else:
            return (None,) * (7 + len(P))
        Hout2 = ops.flat_rows(Hout)
        wgrad(21, dlogits, Hout2, bias=22)
        grads_ready("logit")                             # logit.* is final: its all-reduce overlaps the whole BPTT loop
        dHout = new(S, T, R); ops.gemm(dlogits, W[21], dHout.view(S * T, R))
        del dlogits

        dP1, dP2, dAH = act(T, S, 4 * R), act(T, S, 4 * R), act(T, S, A)
        # d(v) = sum_t alpha_t^T d(ctx_t) is formed ONCE after the loop from the kept d(ctx) rows (subgc_attn_dv_accum) instead of
        # being read and written at every step: on Full-GC (36 nodes per sentence) that was 380 of a step's 830 MB
        defer_dv = pr.shared or (R % 4 == 0 and A % 4 == 0 and A <= 1024 and R <= 2048 and T > 0)      # the float4 forms' limits
        # d(u) deferred the same way (per-sentence sets): the steps file their d(e) rows, one pass after the loop writes every d(u) row once
        defer_du = DEFER_DU and not pr.shared and defer_dv and A % 4 == 0 and A <= 1024 and T > 0
        dE = new(T, S, AL.size(2)) if defer_du else None
        du = new(pr.u.size(0), A) if defer_du else pr.new_du(A)
        dv = new(pr.v.size(0), R) if defer_dv else zer(pr.v.size(0), R)
        dCtx = new(T, S, R) if defer_dv else None
        dWa, dBa = new(T, S, A), new(T, S)                     # per-(step, sentence) partials of alpha_net's gradient
        # The three recurrent data-gradient products of a step are split-K on <= a few hundred rows; their results are read exactly
        # once (d(ctx) by the attention backward, d(h) by the two cell backwards), so they stay as partial PLANES that the consumers
        # add on load (subgc_gemm_*_planes, subgc_lstm_bwd_planes, subgc_attn_bwd_planes): no reduce launches, no summed copies.
        PA, PB, PC = new(8 * S * 3 * R), new(8 * S * R), new(8 * S * 2 * R)
        sA = sC = None                                # (planes, N, n, stride, rows) of the previous step's dP2.Wc2 / dP1.Wc1
        win = lambda st, col0: None if st is None else (st[0], st[1], col0, st[2], st[3], st[4])
        dC1 = [zer(S, R), new(S, R)]                  # [next, cur] ping-pong
        dC2 = [zer(S, R), new(S, R)]
        note("bptt_begin", T)
        rec = None
        if T > 0 and ops.recurrence_ok():
            rec = ops.Recurrence(S=S, T=T, R=R, A=A, n_alpha=N, bf16=int(bf), gemm_flags=ops.GEMM_MODES[ops.gemm_mode.current], keep_scale=float(scale),
                                 m=[S] * (T + 1), row0=[t * S for t in range(T + 1)], dhout_off=[t * R for t in range(T)], ld_dhout=T * R,
                                 Wc1=Wc1, ldW1=ops.ld(Wc1), Wc2=Wc2, ldW2=ops.ld(Wc2), Wq=W[17], ldWq=ops.ld(W[17]), C1=C1, C2=C2, G1=G1, G2=G2,
                                 AH=AH, AL=AL, k_out=k_out, w_a=an_w, lens=lens, dHout=dHout, dP1=dP1, dP2=dP2, dAH=dAH, du=du,
                                 du_planes=du.size(0) if du.dim() == 3 else 1, du_plane_stride=du.stride(0) if du.dim() == 3 else 0,
                                 dv=None if defer_dv else dv, dWa=dWa, dBa=dBa, dCtx=dCtx if defer_dv else None, dE=dE, PA=PA, pa_bytes=PA.numel() * 4,
                                 PB=PB, pb_bytes=PB.numel() * 4, PC=PC, pc_bytes=PC.numel() * 4, dC1_in=dC1[0], dC1_out=dC1[1], dC2_in=dC2[0],
                                 dC2_out=dC2[1], **pr.recur_fields())
            ops.recurrence_bwd(rec)
        for t in range(T - 1, -1, -1):
            if rec is not None:
                break
            nC1, cC1 = dC1; nC2, cC2 = dC2
            ops.lstm_bwd_planes(G2[t], C2[t], C2[t + 1], [win(sC, 0), win(sA, 2 * R)], dHout[:, t, :], None if k_out is None else k_out[t],
                                scale, nC2, dP2[t], cC2, S, R)
            n, st = ops.gemm_planes(dP2[t], Wc2, PA)                       # -> [dctx | dh1 | dh2_prev]
            sA = (PA, 3 * R, n, st, S)
            pr.attn_bwd(AH[t], an_w, lens, AL[t], win(sA, 0), dAH[t], du, None if defer_dv else dv, dWa[t], dBa[t], S, A, R,
                        dCtx[t] if defer_dv else None, **({"de_keep": dE[t]} if defer_du else {}))
            n, st = ops.gemm_planes(dAH[t], W[17], PB)                     # h1 also feeds the attention query
            ops.lstm_bwd_planes(G1[t], C1[t], C1[t + 1], [win(sA, R), (PB, R, 0, n, st, S), win(sC, R)], None, None, 1.0, nC1, dP1[t], cC1, S, R)
            n, st = ops.gemm_planes(dP1[t], Wc1, PC)                       # -> [dh2_prev | dh1_prev]
            sC = (PC, 2 * R, n, st, S)
            dC1.reverse(); dC2.reverse()
        note("bptt_end")

        if defer_dv:
            pr.dv_accum(AL[:T].view(T * S, AL.size(2)), dCtx.view(T * S, R), _step_offsets(T, S, dev), T, lens, dv, S, R)
            del dCtx                                             # [T, S, R] fp32: not kept alive through the weight-gradient products
        if defer_du:
            pr.du_accum(AH[:T].view(T * S, A), dE.view(T * S, dE.size(2)), _step_offsets(T, S, dev), T, lens, an_w, du, S, A)
            del dE
        P1, P2 = dP1.view(T * S, 4 * R), dP2.view(T * S, 4 * R)
        H1a, H2a = ops.flat_rows(H1[:T]), ops.flat_rows(H2[:T])
        wgrad(13, P2, H2a[:, :2 * R], bias=15)             # b_ih and b_hh have the same gradient: one sum rides each product
        wgrad(14, P2, H2a[:, 2 * R:], bias=16)
        wgrad(9, P1, H1a[:, :R], cols=(0, R), bias=11)
        dGf = opnd(ops.colsum(dP1.view(T, S * 4 * R)).view(S, 4 * R))
        wgrad(9, dGf, pr.f16 if bf else pr.f, cols=(R, 2 * R))
        wgrad(9, P1, xt.view(T * S, E), cols=(2 * R, 2 * R + E))
        wgrad(10, P1, H1a[:, R:], bias=12)
        df = new(S, R); ops.gemm(dGf, W[9][:, R:2 * R], df)
        dxt = new(T * S, E); ops.gemm(P1, W[9][:, 2 * R:], dxt)
        d_emb = out_for(8, zero=True)
        dxt3 = dxt.view(T, S, E)
        toks = ctx.tokens_used                       # the words actually fed (ground truth, or scheduled-sampling draws)
        for t in range(T):
            ops.embed_bwd(emb, toks[:, t], toks.stride(0), None if k_xt is None else k_xt[t], scale, dxt3[t], d_emb)
        dAH2 = dAH.view(T * S, A)
        wgrad(17, dAH2, H2a[:, R:2 * R], bias=18)
        ops.colsum(dWa.view(T * S, A), out=out_for(19).view(-1), accumulate=acc[19])
        ops.colsum(dBa.view(T * S, 1), out=out_for(20).view(-1), accumulate=acc[20])
        grads_ready("recurrent")

        dX, dfc_in = prepared_backward(pr, P, W, bf, fc_in, X_nodes, pr.finish_du(du), dv, df, scale, out_for, acc, wgrad, bgrad,
                                       ctx.needs_input_grad[3], ctx.needs_input_grad[2])
        ctx.pr = None
        grads_ready("prepare")                           # the last decoder slice; the encoder's backward follows
        return (None, None, dfc_in, dX, None, None, None) + tuple(ret)


def prepared_backward(pr, P, W, bf, fc_in, X_nodes, du, dv, df, scale, out_for, acc, wgrad, bgrad, need_dX, need_dfc):
    """Backward of `Prepared` (ctx2att, att_embed, fc_embed): du [rows, A], dv [rows, R] over the packed attention rows (fp32, the
    attention kernel accumulated them), df [S, R].  Shared by both decoder Functions; `fc_in` in the order `pr` was built in."""
    dev = du.device
    new = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
    tot = pr.total
    att_w, fc0_w, fc2_w = P[4], P[0], P[2]
    if bf:
        du16 = ops.as_b16(du)                                              # dead rows are zero: cast them all, bound the products
        ops.gemm(du16, W[6], dv, accum=True, m_dev=tot)                    # u = v W_c^T + b_c
        wgrad(6, du16, pr.v16, bias=7, m_dev=tot)
        dX = None
        if pr.dedup:
            # d(relu(att_embed)) of every COPY, summed onto the unique node rows (5 copies per row: fp32 atomics), then ONE weight-gradient
            # and ONE data-gradient product over the unique rows
            dzs = ops.relu_bwd(dv, pr.v, scale)
            dz = ops.zeros(X_nodes.size(0), dzs.size(1), device=dev)
            ops.scatter_add_rows(dzs, pr.src_row, dz, m_dev=tot)
            dz16 = ops.as_b16(dz)
            wgrad(4, dz16, pr.Xu16, bias=5)
            if need_dX:
                dX = new(X_nodes.size(0), X_nodes.size(1)); ops.gemm(dz16, W[4], dX)
            need_dX = False
        else:
            dzv = ops.relu_bwd(dv, pr.v, scale, bf16=True)
            wgrad(4, dzv, pr.Xg16, bias=5, m_dev=tot)
        if need_dX:
            dXg = new(pr.Xg16.size(0), pr.Xg16.size(1)); ops.gemm(dzv, W[4], dXg, m_dev=tot)
            if pr.shared:
                dX = dXg                                                       # the sets ARE the node rows: nothing to scatter
            else:
                dX = ops.zeros(X_nodes.size(0), X_nodes.size(1), device=dev)
                ops.scatter_add_rows(dXg, pr.src_row, dX, m_dev=tot)
        dz2 = ops.relu_bwd(df, pr.f, scale, bf16=True)
        wgrad(2, dz2, pr.f116, bias=3)
        df1 = new(pr.S, pr.f1.size(1)); ops.gemm(dz2, W[2], df1)
        dz1 = ops.relu_bwd(df1, pr.f1, 1.0, bf16=True)
        wgrad(0, dz1, pr.fc16, bias=1)
        dfc_in = None
        if need_dfc:
            dfc_in = new(pr.S, fc_in.size(1)); ops.gemm(dz1, W[0], dfc_in)
        return dX, dfc_in
    ops.gemm(du, P[6], dv, accum=True, m_dev=tot)                          # u = v W_c^T + b_c
    wgrad(6, du, pr.v, bias=7, m_dev=tot)
    dzv = ops.relu_bwd(dv, pr.v, scale)
    dX = None
    if pr.dedup:
        dz = ops.zeros(X_nodes.size(0), dzv.size(1), device=dev)
        ops.scatter_add_rows(dzv, pr.src_row, dz, m_dev=tot)
        wgrad(4, dz, X_nodes, bias=5)
        if need_dX:
            dX = new(X_nodes.size(0), X_nodes.size(1)); ops.gemm(dz, att_w, dX)
        need_dX = False
    else:
        wgrad(4, dzv, pr.Xg, bias=5, m_dev=tot)
    if need_dX:
        dXg = new(pr.Xg.size(0), pr.Xg.size(1)); ops.gemm(dzv, att_w, dXg, m_dev=tot)
        if pr.shared:
            dX = dXg
        else:
            dX = ops.zeros(X_nodes.size(0), X_nodes.size(1), device=dev)
            ops.scatter_add_rows(dXg, pr.src_row, dX, m_dev=tot)
    dz2 = ops.relu_bwd(df, pr.f, scale)
    wgrad(2, dz2, pr.f1, bias=3)
    df1 = new(pr.S, pr.f1.size(1)); ops.gemm(dz2, fc2_w, df1)
    dz1 = ops.relu_bwd(df1, pr.f1, 1.0)
    wgrad(0, dz1, fc_in, bias=1)
    dfc_in = None
    if need_dfc:
        dfc_in = new(pr.S, fc_in.size(1)); ops.gemm(dz1, fc0_w, dfc_in)
    return dX, dfc_in


# ------------------------------------------------------------------------------- decode (no grad)
class DecodeState:
    """Step-wise decoder for sampling (AttModel._sample loop body): same kernels, batch n."""

    def __init__(self, pr: Prepared, P, N, want_att, xt_table=None, fuse_lstm=False, snapshots=None, W16=None):
        """`xt_table` [V+1, 4R] (optional, frozen weights only): relu(Emb) . W_ih[:, 2R:]^T, one row per token -- the x->gates
        product of the attention LSTM looked up instead of recomputed every step (AttModel.xt_gates_table).
        `fuse_lstm` (<= 32 rows, R % 4 == 0): each LSTM cell is ONE launch, gate GEMM + cell update (subgc_lstm_step_skinny) on
        row-permuted weight snapshots; h is written into the OTHER buffer of an [H1, H1n] / [H2, H2n] pair, because the
        launch that produces it is still reading the current one.  Beam search forks the state through `reorder`, which
        gathers into a third buffer and so composes with the pairs; not for callers that write H1/H2 themselves (step API).
        `snapshots`: a dict shared by every state built on the same weights (AttModel.decode_snapshots): the K-concatenated
        (and, fused, row-permuted) LSTM matrices are built once per set of weights, not once per state / captured graph."""
        self.xt_table = xt_table
        self.fused = bool(fuse_lstm) and pr.S <= 32 and P[10].size(1) % 4 == 0
        # `W16` (compute_dtype = bf16; list aligned with P, see bf16_twins): the fused (<= 32 row) step streams bf16-STORED weights --
        # both LSTM matrices, h2att (<= 16 rows) and the logit layer -- against fp32 activations with fp32 accumulation: half the
        # bytes of the weight stream that bounds the one-image decode; cell state, attention and the pick stay fp32
        self.w16 = W16 if (self.fused and W16 is not None and all(W16[i] is not None for i in (9, 10, 13, 14, 17, 21))) else None
        (_, _, _, _, _, _, _, _, self.emb, w1i, w1h, self.b1i, self.b1h, w2i, w2h, self.b2i, self.b2h,
         self.h2a_w, self.h2a_b, self.an_w, self.an_b, self.lg_w, self.lg_b) = P
        self.pr, self.N = pr, N
        dev = pr.f.device
        S = pr.S
        R, E = w1h.size(1), self.emb.size(1)
        self.S, self.R, self.E, self.A, self.V1 = S, R, E, self.h2a_w.size(0), self.lg_w.size(0)
        key = ("perm16" if self.w16 is not None else "perm") if self.fused else "cat"
        if snapshots is not None and key in snapshots:
            self.Wc1, self.Wc2 = snapshots[key]
        else:
            if self.w16 is not None:
                self.Wc1 = _cat_weights(self.w16[9][:, :R], self.w16[10])
                self.Wc2 = _cat_weights(self.w16[13], self.w16[14])
            else:
                self.Wc1 = _cat_weights(w1i[:, :R], w1h)
                self.Wc2 = _cat_weights(w2i, w2h)
            if self.fused:
                perm = ops.lstm_gate_perm(R, dev)
                self.Wc1, self.Wc2 = self.Wc1[perm].contiguous(), self.Wc2[perm].contiguous()    # only the permuted snapshots are kept
            if snapshots is not None:
                snapshots[key] = (self.Wc1, self.Wc2)
        if self.fused:
            self.H1n, self.H2n = ops.zeros(S, 2 * R, device=dev), ops.zeros(S, 3 * R, device=dev)
        self.W1x = w1i[:, 2 * R:]
        self.W1f = w1i[:, R:2 * R]
        self.Gf = torch.empty(S, 4 * R, device=dev, dtype=torch.float32)
        ops.gemm(pr.f, self.W1f, self.Gf, tb=True)
        self.H1 = ops.zeros(S, 2 * R, device=dev)           # [h2 | h1]
        self.H2 = ops.zeros(S, 3 * R, device=dev)           # [ctx | h1 | h2]
        self.C1 = [ops.zeros(S, R, device=dev), torch.empty(S, R, device=dev)]
        self.C2 = [ops.zeros(S, R, device=dev), torch.empty(S, R, device=dev)]
        new = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
        self.xt, self.Gx, self.pre, self.ah = new(S, E), new(S, 4 * R), new(S, 4 * R), new(S, self.A)
        self.hout, self.logits = new(S, R), new(S, self.V1)
        self.want_att = want_att
        self._alt = None
        self.lg_op = self.w16[21] if (self.w16 is not None and S <= 16) else self.lg_w        # the logit matrix as the skinny launches stream it

    def _h2att(self):
        """ah = h2att(h1): the weight-streaming form on the bf16 twin when there is one (<= 16 rows), else the fp32 product."""
        R = self.R
        if self.w16 is not None and self.S <= 16:
            ops.gemm_skinny_wb16(self.H2[:, R:2 * R], self.w16[17], self.ah, bias=self.h2a_b)
        else:
            ops.gemm(self.H2[:, R:2 * R], self.h2a_w, self.ah, tb=True, bias=self.h2a_b)

    def _logits(self):
        if self.w16 is not None and self.S <= 16:
            ops.gemm_skinny_wb16(self.hout, self.w16[21], self.logits, bias=self.lg_b)
        else:
            ops.gemm(self.hout, self.lg_w, self.logits, tb=True, bias=self.lg_b)

    def reset(self):
        """Start a new decode on the SAME buffers (hipGraph replay: `pr`'s tensors were overwritten in place)."""
        ops.gemm(self.pr.f, self.W1f, self.Gf, tb=True)
        for buf in (self.H1, self.H2, self.C1[0], self.C2[0]):
            ops.fill_(buf, 0.0)

    # -- beam search support (CaptionModel.py:76-90 "rearrange recurrent states") -------------------
    def recurrent(self):
        """The tensors that carry state between steps: [h2|h1], the h2 slot of the lang-LSTM operand, c1, c2."""
        R = self.R
        return [self.H1, self.H2[:, 2 * R:], self.C1[0], self.C2[0]]

    def reorder(self, src):
        """state[row] <- state[src[row]] for every row (src: int32 [S] on the device)."""
        if self._alt is None:
            self._alt = [torch.empty_like(self.H1), torch.empty_like(self.H2), torch.empty_like(self.C1[0]), torch.empty_like(self.C2[0])]
        a1, a2, ac1, ac2 = self._alt
        R = self.R
        ops.gather_rows_multi([(self.H1, a1), (self.H2[:, 2 * R:], a2[:, 2 * R:]), (self.C1[0], ac1), (self.C2[0], ac2)], src)
        self._alt = [self.H1, self.H2, self.C1[0], self.C2[0]]
        self.H1, self.H2, self.C1[0], self.C2[0] = a1, a2, ac1, ac2

    def _step_fused(self, it, alpha_out, normalize):
        S, R, A = self.S, self.R, self.A
        pr = self.pr
        if self.xt_table is not None:
            add1, tok = self.xt_table, it
        else:
            ops.embed_fwd(self.emb, it, 1, None, 1.0, self.xt)
            ops.gemm(self.xt, self.W1x, self.Gx, tb=True)
            add1, tok = self.Gx, None
        ops.lstm_step_skinny(self.H1, self.Wc1, self.C1[0], self.C1[1], [self.H2[:, R:2 * R], self.H1n[:, R:]], self.b1i, self.b1h,
                             add1, tok, self.Gf)
        self.C1.reverse()
        self._h2att()
        ops.attn_fwd(pr.u, pr.v, self.ah, self.an_w, self.an_b, pr.off, pr.lens, self.H2[:, :R], alpha_out, S, A, R)
        ops.lstm_step_skinny(self.H2, self.Wc2, self.C2[0], self.C2[1], [self.H1n[:, :R], self.H2n[:, 2 * R:], self.hout], self.b2i, self.b2h)
        self.C2.reverse()
        self.H1, self.H1n = self.H1n, self.H1
        self.H2, self.H2n = self.H2n, self.H2
        self._logits()
        if normalize:
            ops.log_softmax_rows_(self.logits)
        return self.logits

    def greedy_loop(self, T, seq, seqlp, counts, AL, it0):
        """The whole greedy token loop (AttModel.py:282-319 with sample_max) for <= 16 rows on the fused state, with the PICK folded into
        the step's launches and -- round 6 -- the weight streams that do not depend on each other sharing launches, so that only what
        must wait sits on the step's dependent chain.  Five launches per token:

          1. [ logits(h_lang) + packed arg-max / log-sum-exp partials  |  attention-LSTM gate product H1 . Wc1^T ]   (subgc_skinny_dual)
             -- the gate product needs step t's STATE, not the word: it used to wait for the pick inside the att-LSTM launch;
          2. attention-LSTM cell: reads the pick (files seq / unfinished / the live count), adds the word's x -> gates table row
             (subgc_lstm_cell_pick);
          3. [ h2att(h_att)  |  the language LSTM's (h_att, h_lang_prev) part  [h_att, h_lang] . Wc2[:, R:]^T ]            (subgc_skinny_dual)
          4. attention;
          5. language LSTM: ctx . Wc2[:, :R]^T + the part from 3 + cell (subgc_lstm_step_skinny, K = R).

        The 32 MB of the attention LSTM and 32 of the language LSTM's 48 MB leave the chain pick -> cell -> query -> attention -> cell.
        The log-probabilities are formed once after the loop (subgc_pick_lse_finish); the last core step, whose logits the reference never
        reads, runs only as far as its attention weights are wanted (`AL`).  seq [S,T] / counts [T] zeroed by the caller, `it0` = S zeros
        (the <bos> input of step 0)."""
        if not (self.fused and self.xt_table is not None and self.S <= 16):
            raise ops.SubgcError("greedy_loop needs the fused decode state with the x->gates table and <= 16 rows")
        S, R, A, V1 = self.S, self.R, self.A, self.V1
        pr, dev = self.pr, self.H1.device
        if getattr(self, "_pick", None) is None or self._pick[2].size(0) != T:
            self._pick = (torch.zeros(2, ops.PICK_BEST_ELEMS, device=dev, dtype=torch.int64), torch.zeros(2, S, device=dev, dtype=torch.int32),
                          torch.empty(T, (V1 + 15) // 16, 16, 2, device=dev, dtype=torch.float32))
            self._preL = torch.empty(S, 4 * R, device=dev, dtype=torch.float32)
        best, unf, lse = self._pick
        preA, preL = self.pre, self._preL
        h2a_w = self.w16[17] if self.w16 is not None else self.h2a_w
        ops.zero_(best)
        # step 0's gate product has no logits launch to ride in: alone (H1 is the zero state, but the launch keeps the loop uniform)
        ops.skinny_dual(self.H2[:, R:2 * R], h2a_w, self.ah, self.H1, self.Wc1, preA, bias1=self.h2a_b, unperm2_R=R)
        for t in range(T + 1):
            last = t == T
            hs = [self.H2[:, R:2 * R], self.H1n[:, R:]]
            if t == 0:
                ops.lstm_cell_pick(preA, self.C1[0], self.C1[1], hs, self.b1i, self.b1h, self.xt_table, self.Gf, tok=it0)
            else:
                pick = (best[(t - 1) & 1], unf[(t - 2) & 1] if t >= 2 else None, unf[(t - 1) & 1], seq, t - 1, counts[t - 1:t],
                        counts[t - 2:t - 1] if t >= 2 else None)
                if last and AL is None:
                    ops.pick_file(*pick)
                    break
                ops.lstm_cell_pick(preA, self.C1[0], self.C1[1], hs, self.b1i, self.b1h, self.xt_table, self.Gf, pick=pick,
                                   best_reset=None if last else best[t & 1])
            self.C1.reverse()
            ops.skinny_dual(self.H2[:, R:2 * R], h2a_w, self.ah, self.H2[:, R:], self.Wc2[:, R:], preL, bias1=self.h2a_b, unperm2_R=R)
            ops.attn_fwd(pr.u, pr.v, self.ah, self.an_w, self.an_b, pr.off, pr.lens, self.H2[:, :R], None if AL is None else AL[t], S, A, R)
            if last:
                break
            ops.lstm_step_skinny(self.H2[:, :R], self.Wc2[:, :R], self.C2[0], self.C2[1], [self.H1n[:, :R], self.H2n[:, 2 * R:], self.hout], self.b2i,
                                 self.b2h, add2=preL)
            self.C2.reverse()
            self.H1, self.H1n = self.H1n, self.H1
            self.H2, self.H2n = self.H2n, self.H2
            ops.skinny_dual(self.hout, self.lg_op, None, self.H1, self.Wc1, preA, bias1=self.lg_b, unperm2_R=R, best=best[t & 1], lse_part=lse[t])
        ops.pick_lse_finish(lse, V1, counts, seqlp)

    def step(self, it, alpha_out, normalize=True):
        if self.fused:
            return self._step_fused(it, alpha_out, normalize)
        S, R, A = self.S, self.R, self.A
        pr = self.pr
        if self.xt_table is not None:
            ops.token_rows(self.xt_table, it, self.Gx)
        else:
            ops.embed_fwd(self.emb, it, 1, None, 1.0, self.xt)
            ops.gemm(self.xt, self.W1x, self.Gx, tb=True)
        ops.gemm(self.H1, self.Wc1, self.pre, tb=True)
        ops.lstm_fwd(self.pre, self.Gx, self.Gf, self.b1i, self.b1h, self.C1[0], self.C1[1], self.H2[:, R:2 * R], self.H1[:, R:],
                     None, 1.0, None, None, S, R)
        self.C1.reverse()
        ops.gemm(self.H2[:, R:2 * R], self.h2a_w, self.ah, tb=True, bias=self.h2a_b)
        ops.attn_fwd(pr.u, pr.v, self.ah, self.an_w, self.an_b, pr.off, pr.lens, self.H2[:, :R], alpha_out, S, A, R)
        ops.gemm(self.H2, self.Wc2, self.pre, tb=True)
        ops.lstm_fwd(self.pre, None, None, self.b2i, self.b2h, self.C2[0], self.C2[1], self.H1[:, :R], self.H2[:, 2 * R:],
                     None, 1.0, self.hout, None, S, R)
        self.C2.reverse()
        ops.gemm(self.hout, self.lg_w, self.logits, tb=True, bias=self.lg_b)
        if normalize:
            ops.log_softmax_rows_(self.logits)
        return self.logits
