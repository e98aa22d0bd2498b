"""Autograd wrappers over the libspmi kernels (include/spmi.h).

`gcn_aggregate` is the fused normalised aggregation of GraphConv.forward (DGL, called at
spiro/dtdg/models/encoder/implicitTimeEncoder/staticGraphEncoder.py:113,116); the
`*_cell` functions are one step of torch.nn.GRU/LSTM/RNN over all nodes, fused with the
NodeMemory read/write of implicitTimeModels.py:48-54,153-156.
"""
from __future__ import annotations

import threading
import weakref
from typing import Optional, Tuple

import torch

from . import _native as nv
from .graph import _ACT_CODES, aggregate


def _act_bias_bwd(g_y: torch.Tensor, y: torch.Tensor, act: Optional[str], slope: float,
                  want_gb: bool, row_scale: Optional[torch.Tensor] = None
                  ) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """gP = gY * act'(Y) and gb = column sums of gP, one pass.  `row_scale` [rows]: gP comes back
    with its rows multiplied by it (gb sums the unscaled values)."""
    if g_y.dtype != torch.float32:  # bf16 storage: elementwise torch ops, fp32 column sums
        g_p = g_y
        if act == "relu":
            g_p = torch.where(y > 0, g_y, torch.zeros_like(g_y))
        elif act == "leaky_relu":
            g_p = torch.where(y > 0, g_y, g_y * slope)
        g_b = g_p.sum(0, dtype=torch.float32) if want_gb else None
        return (g_p if row_scale is None else g_p * row_scale[:, None].to(g_p.dtype)), g_b
    m, f = g_y.shape
    if g_y.stride(1) != 1 or g_y.stride(0) != y.stride(0):
        g_y = g_y.contiguous()
        y = y.contiguous()
    g_p = torch.empty(m, f, dtype=torch.float32, device=g_y.device)
    code = {None: 0, "none": 0, "relu": 1, "leaky_relu": 2}[act]
    lib = nv.lib()
    with torch.cuda.device(g_y.device):
        # column sums slab by slab, added in slab order: the bias gradient is bit-reproducible
        g_b = ws = None
        ws_bytes = 0
        if want_gb:
            g_b = torch.empty(f, dtype=torch.float32, device=g_y.device)
            ws_bytes = lib.spmi_act_bias_bwd_workspace_bytes(m, f)
            ws = torch.empty(max(ws_bytes, 4) // 4, dtype=torch.float32, device=g_y.device)
        nv.check(lib.spmi_act_bias_bwd_ordered(nv.ptr(g_y), nv.ptr(y), m, f, g_y.stride(0), code, float(slope),
                                               nv.ptr(row_scale), nv.ptr(g_p), nv.ptr(g_b), nv.ptr(ws), ws_bytes,
                                               nv.stream()), "spmi_act_bias_bwd_ordered")
    return g_p, g_b


class GcnAggregate(torch.autograd.Function):
    """Y = act(norm_dst * A (norm_src * X) + bias) with A the snapshot's adjacency
    (rows = destinations).  Backward runs the same kernel on the by-source CSR with the
    two normalisers exchanged: dX = norm_src * A^T (norm_dst * dP).  `g` is any
    NormalisedAdjacency (a SnapshotGraph, or one rank's shard of it)."""

    @staticmethod
    def forward(ctx, x, bias, g, norm, edge_weight, act, slope):
        mode = 0 if norm in ("both", "sgc") else 1
        dst_scale = g.norm("in", mode) if norm in ("right", "both", "sgc") else None
        y = aggregate(g.csr(wait=False), x.detach(), None, g.edge_scale("fwd", norm, edge_weight), dst_scale,
                      None if bias is None else bias.detach(), act, slope)
        ctx.g, ctx.norm, ctx.edge_weight, ctx.act, ctx.slope = g, norm, edge_weight, act, slope
        ctx.has_bias = bias is not None
        ctx.save_for_backward(y if act not in (None, "none") else None)
        return y

    @staticmethod
    def backward(ctx, g_y):
        (y,) = ctx.saved_tensors
        need_x, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1] and ctx.has_bias
        g_p, g_b = g_y, None
        if y is not None or need_b:
            g_p, g_b = _act_bias_bwd(g_y, y if y is not None else g_y, ctx.act, ctx.slope, need_b)
        g_x = None
        if need_x:
            g, norm = ctx.g, ctx.norm
            mode = 0 if norm in ("both", "sgc") else 1
            out_scale = g.norm("out", mode) if norm in ("left", "both") else None
            if norm == "sgc":
                out_scale = g.norm("in", mode)
            g_x = aggregate(g.csr_t(wait=False), g_p.contiguous(), None,
                            g.edge_scale("bwd", norm, ctx.edge_weight), out_scale, None, None)
        return g_x, g_b, None, None, None, None, None


def gcn_aggregate(g, x: torch.Tensor, norm: str = "both", bias: Optional[torch.Tensor] = None,
                  edge_weight: Optional[torch.Tensor] = None, act: Optional[str] = None,
                  slope: float = 0.01) -> torch.Tensor:
    """Differentiable fused aggregation over graph `g` with DGL GraphConv's `norm` modes
    ('none', 'both', 'left', 'right') plus 'sgc' (SGConv: in-degree^-1/2 on both sides)."""
    if edge_weight is not None and edge_weight.requires_grad:
        raise nv.SpmiError("gradients with respect to edge_weight are not implemented")
    return GcnAggregate.apply(x, bias, g, norm, edge_weight, act, slope)


# --------------------------------------------------------------------------- dense layers
GEMM_TN_MIN_ROWS = 16384  # below this many rows the library GEMM is as good
# How spmi_gemm_tn multiplies: "bf16x3" = operands split exactly into three bf16 pieces, six piece
# products per term on the bf16 matrix cores (fp32-accurate, no range limit, 2-3x faster);
# "fp32" = the fp32 matrix instruction.
GEMM_TN_MODE = "bf16x3"
# Outputs made of whole 128 x 128 blocks (the layer and gate weights of the path) in three fp16
# planes per value under running power-of-two scales: three products per term instead of six.
GEMM_TN_F16_BLOCKS = True


def gemm_tn(a: torch.Tensor, b: torch.Tensor, sum_a: bool = False, sum_b: bool = False):
    """a^T b for two tall matrices sharing their row count (a weight gradient X^T dY): the
    split-K matrix-core kernel of libspmi for tall shapes with a small output (where it is
    2-3x faster than the library GEMM), the library GEMM otherwise.  With `sum_a` / `sum_b`
    the column sums of the operands come back as well, as `(product, a.sum(0) or None,
    b.sum(0) or None)`: a bias gradient is the column sum of a matrix the kernel reads anyway."""
    k, m = a.shape
    n = b.shape[1]
    sums = sum_a or sum_b
    if (k < GEMM_TN_MIN_ROWS or m * n > 512 * 512 or a.dtype != torch.float32
            or b.dtype != torch.float32 or not a.is_cuda):
        out = a.t() @ b
        return (out, a.sum(0) if sum_a else None, b.sum(0) if sum_b else None) if sums else out
    a = a if a.stride(1) == 1 else a.contiguous()
    b = b if b.stride(1) == 1 else b.contiguous()
    lib = nv.lib()
    with torch.cuda.device(a.device):
        out = torch.empty(m, n, dtype=torch.float32, device=a.device)
        sa = torch.empty(m, dtype=torch.float32, device=a.device) if sum_a else None
        sb = torch.empty(n, dtype=torch.float32, device=a.device) if sum_b else None
        ws_bytes = lib.spmi_gemm_tn_workspace_bytes(m, n, k)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=a.device)
        nv.check(lib.spmi_gemm_tn_sums(nv.ptr(a), a.stride(0), nv.ptr(b), b.stride(0), k, m, n,
                                       nv.ptr(out), n, 0, nv.ptr(sa), nv.ptr(sb),
                                       1 if GEMM_TN_MODE == "fp32" else (2 if GEMM_TN_F16_BLOCKS else 0),
                                       nv.ptr(ws), ws_bytes, nv.stream()), "spmi_gemm_tn_sums")
    return (out, sa, sb) if sums else out


def _weight_and_bias_grad(dy: torch.Tensor, x: torch.Tensor, need_w: bool, need_b: bool, dy_first: bool):
    """(dW, db) of a dense layer from dY and its input: dW = dY^T x (`dy_first`) or x^T dY,
    db = column sums of dY, taken from the same pass over dY when both are wanted."""
    g_w = g_b = None
    if need_w and need_b:
        if dy_first:
            g_w, g_b, _ = gemm_tn(dy, x, sum_a=True)
        else:
            g_w, _, g_b = gemm_tn(x, dy, sum_b=True)
    elif need_w:
        g_w = gemm_tn(dy, x) if dy_first else gemm_tn(x, dy)
    elif need_b:
        g_b = dy.sum(0)
    return g_w, g_b


def gemm_tn_pair_ok(a0: torch.Tensor, b0: torch.Tensor, a1: torch.Tensor, b1: torch.Tensor) -> bool:
    """`gemm_tn_pair` takes these operands: two products of one shape over the same rows, tall
    enough for the split-K kernel, whole 128 x 128 output blocks, layouts its loads accept."""
    ts = (a0, b0, a1, b1)
    return (GEMM_TN_MODE == "bf16x3" and a0.shape == a1.shape and b0.shape == b1.shape
            and a0.shape[0] == b0.shape[0] >= GEMM_TN_MIN_ROWS
            and all(t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1
                    and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0 for t in ts)
            and bool(nv.lib().spmi_gemm_tn_pair_supported(a0.shape[1], b0.shape[1])))


def gemm_tn_pair(a0: torch.Tensor, b0: torch.Tensor, a1: torch.Tensor, b1: torch.Tensor, sums: bool = False):
    """(a0^T b0, a1^T b1[, a0.sum(0), a1.sum(0)]) in one launch (`gemm_tn_pair_ok` holds): the
    column ranges a0 and a1 share are read from memory once."""
    k, m = a0.shape
    n = b0.shape[1]
    lib = nv.lib()
    with torch.cuda.device(a0.device):
        c0 = torch.empty(m, n, dtype=torch.float32, device=a0.device)
        c1 = torch.empty(m, n, dtype=torch.float32, device=a0.device)
        s0 = torch.empty(m, dtype=torch.float32, device=a0.device) if sums else None
        s1 = torch.empty(m, dtype=torch.float32, device=a0.device) if sums else None
        ws_bytes = lib.spmi_gemm_tn_pair_workspace_bytes(m, n, k)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=a0.device)
        nv.check(lib.spmi_gemm_tn_pair_sums(
            nv.ptr(a0), a0.stride(0), nv.ptr(b0), b0.stride(0), nv.ptr(a1), a1.stride(0), nv.ptr(b1),
            b1.stride(0), k, m, n, nv.ptr(c0), n, nv.ptr(c1), n, nv.ptr(s0), nv.ptr(s1), int(GEMM_TN_F16_BLOCKS),
            nv.ptr(ws), ws_bytes, nv.stream()), "spmi_gemm_tn_pair_sums")
    return (c0, c1, s0, s1) if sums else (c0, c1)


# Projections of a tall matrix by a small weight: "f16x3" = libspmi's memory-speed kernel on
# the fp16 matrix cores (three-plane operands, fp32-chain accuracy; rows of any finite
# magnitude, they are scaled by a power of two in the kernel; weights within +-1023),
# "library" = torch.mm (hipBLASLt, fp32 matrix instruction).
DENSE_GEMM = "f16x3"
GEMM_ROWS_MIN_ROWS = 4096


def _gemm_rows_ok(x: torch.Tensor, k: int, n: int) -> bool:
    return (DENSE_GEMM == "f16x3" and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2
            and x.shape[0] >= GEMM_ROWS_MIN_ROWS and x.shape[1] == k and x.stride(1) == 1
            and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0
            and bool(nv.lib().spmi_gemm_rows_supported(k, n)))


def gemm_rows(x: torch.Tensor, weight_nk: torch.Tensor, bias: Optional[torch.Tensor] = None,
              act: Optional[str] = None, slope: float = 0.01,
              row_scale: Optional[torch.Tensor] = None, planes: Optional[torch.Tensor] = None,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """act(x @ weight_nk.T + bias) * row_scale[:, None] for x [rows, K] and weight_nk [N, K] in
    one launch of spmi_gemm_rows; the caller has checked `_gemm_rows_ok(x, K, N)`.  `planes`:
    `weight_planes(...)` of the same weight, for callers that hold it across calls.  `out`:
    float32 [rows, N] with unit column stride to write into (e.g. the head of a larger table)."""
    n, k = weight_nk.shape
    if planes is None:
        planes = split_f16x3(weight_nk)
    b = None if bias is None else bias.detach().to(torch.float32).contiguous()
    if row_scale is not None and (row_scale.dtype != torch.float32 or row_scale.numel() != x.shape[0]
                                  or not row_scale.is_contiguous()):
        raise nv.SpmiError("gemm_rows: row_scale must be a contiguous float32 vector with one entry per row")
    with torch.cuda.device(x.device):
        if out is None:
            out = torch.empty(x.shape[0], n, dtype=torch.float32, device=x.device)
        elif (out.shape != (x.shape[0], n) or out.dtype != torch.float32 or out.stride(1) != 1
              or out.device != x.device):
            raise nv.SpmiError("gemm_rows: `out` must be float32 [rows, N] on x's device, unit column stride")
        nv.check(nv.lib().spmi_gemm_rows(nv.ptr(x), x.stride(0), x.shape[0], k, nv.ptr(planes), n,
                                         nv.ptr(b), nv.ptr(row_scale), _ACT_CODES[act], float(slope),
                                         nv.ptr(out), out.stride(0), nv.stream()), "spmi_gemm_rows")
    return out


def _gemm_rows_bf16_ok(x: torch.Tensor, k: int, n: int) -> bool:
    return (DENSE_GEMM == "f16x3" and x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 2
            and x.shape[0] >= GEMM_ROWS_MIN_ROWS and x.shape[1] == k and x.stride(1) == 1
            and x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0
            and bool(nv.lib().spmi_gemm_rows_bf16_supported(k, n)))


def gemm_rows_bf16(x: torch.Tensor, pieces: torch.Tensor, n: int, bias: Optional[torch.Tensor] = None,
                   act: Optional[str] = None, slope: float = 0.01,
                   row_scale: Optional[torch.Tensor] = None, out_dtype: torch.dtype = torch.bfloat16
                   ) -> torch.Tensor:
    """act(x @ W.T + bias) * row_scale[:, None] for x [rows, K] stored as bfloat16 and a float32
    weight W [N, K] given as its three bf16 pieces (`weight_planes(..., "bf16x3", ...)`): one
    launch of spmi_gemm_rows_bf16, products at fp32 accuracy, result rounded once to `out_dtype`
    (bfloat16 or float32).  The caller has checked `_gemm_rows_bf16_ok(x, K, N)`."""
    if out_dtype not in (torch.bfloat16, torch.float32):
        raise nv.SpmiError("gemm_rows_bf16: out_dtype must be bfloat16 or float32")
    b = None if bias is None else bias.detach().to(torch.float32).contiguous()
    if row_scale is not None and (row_scale.dtype != torch.float32 or row_scale.numel() != x.shape[0]
                                  or not row_scale.is_contiguous()):
        raise nv.SpmiError("gemm_rows_bf16: row_scale must be a contiguous float32 vector with one entry per row")
    with torch.cuda.device(x.device):
        out = torch.empty(x.shape[0], n, dtype=out_dtype, device=x.device)
        nv.check(nv.lib().spmi_gemm_rows_bf16(nv.ptr(x), x.stride(0), x.shape[0], x.shape[1], nv.ptr(pieces), n,
                                              nv.ptr(b), nv.ptr(row_scale), _ACT_CODES[act], float(slope),
                                              nv.ptr(out), n, int(out_dtype == torch.bfloat16), nv.stream()),
                 "spmi_gemm_rows_bf16")
    return out


def graphconv_inference(g, feat: torch.Tensor, weight: torch.Tensor, norm: str,
                        bias: Optional[torch.Tensor], act: Optional[str], slope: float
                        ) -> Optional[torch.Tensor]:
    """GraphConv without autograd when the projection runs on spmi_gemm_rows: the source
    normaliser is applied to the projected rows in the GEMM's epilogue (as DGL does with
    `feat_src * norm`), so the aggregation needs no per-edge weight and the snapshot no
    gather of one.  Returns None when the shapes are not covered."""
    if norm not in ("both", "left", "right", "none"):
        return None
    bf16 = feat.dtype == torch.bfloat16
    if not (_gemm_rows_bf16_ok if bf16 else _gemm_rows_ok)(feat, weight.shape[0], weight.shape[1]):
        return None
    mode = 0 if norm == "both" else 1
    scale = g.norm("out", mode) if norm in ("both", "left") else None
    if bf16:
        # bfloat16 feature storage: the rows are matrix-core operands as they are, the weight
        # keeps its float32 value (three exact bf16 pieces), the projected rows are rounded once
        rst = gemm_rows_bf16(feat, weight_planes(weight, "bf16x3", True), weight.shape[1], None, None, 0.01, scale)
    else:
        rst = gemm_rows(feat, weight.detach().t(), None, None, 0.01, scale,
                        planes=weight_planes(weight, "f16x3", True, F16X3_GEMM_ROWS_LIMIT))
    dst_scale = g.norm("in", mode) if norm in ("both", "right") else None
    return aggregate(g.csr(wait=False), rst, None, None, dst_scale,
                     None if bias is None else bias.detach(), act, slope)


class _GraphConvProjectFirst(torch.autograd.Function):
    """A whole GraphConv layer that multiplies first, under autograd, on the kernels of the
    inference path: Y = act(n_in * A (n_out * (X W)) + b) with the source normaliser in the
    projection's epilogue, so neither direction gathers or streams a per-edge weight:

      forward   Z = (X W) * n_out (spmi_gemm_rows), Y = act(n_in * (A Z) + b) (one aggregation)
      backward  P = n_in * (gY * act'(Y)) and gb in one pass (spmi_act_bias_bwd_ordered),
                gZ = n_out * (A^T P) (one aggregation over the by-source CSR) = d(X W),
                gW = X^T gZ (spmi_gemm_tn), gX = gZ W^T (spmi_gemm_rows)."""

    @staticmethod
    def forward(ctx, x, w, bias, g, norm, act, slope):
        mode = 0 if norm == "both" else 1
        src_scale = g.norm("out", mode) if norm in ("both", "left") else None
        dst_scale = g.norm("in", mode) if norm in ("both", "right") else None
        z = gemm_rows(x.detach(), w.detach().t(), None, None, 0.01, src_scale,
                      planes=weight_planes(w, "f16x3", True, F16X3_GEMM_ROWS_LIMIT))
        y = aggregate(g.csr(wait=False), z, None, None, dst_scale, None if bias is None else bias.detach(),
                      act, slope)
        ctx.g, ctx.act, ctx.slope, ctx.has_bias = g, act, slope, bias is not None
        ctx.scales = (src_scale, dst_scale)
        ctx.save_for_backward(x, w, y if act not in (None, "none") else None)
        return y

    @staticmethod
    def backward(ctx, g_y):
        x, w, y = ctx.saved_tensors
        src_scale, dst_scale = ctx.scales
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], \
            ctx.needs_input_grad[2] and ctx.has_bias
        g_p, g_b = _act_bias_bwd(g_y, y if y is not None else g_y, ctx.act if y is not None else None,
                                 ctx.slope, need_b, dst_scale)
        g_x = g_w = None
        if need_x or need_w:
            g_z = aggregate(ctx.g.csr_t(wait=False), g_p, None, None, src_scale, None, None)
            if need_w:
                g_w = gemm_tn(x, g_z)
            if need_x:
                g_x = _mm_nk(g_z, w)
        return g_x, g_w, g_b, None, None, None, None


def graphconv_project_first(g, feat: torch.Tensor, weight: torch.Tensor, norm: str,
                            bias: Optional[torch.Tensor], act: Optional[str], slope: float
                            ) -> Optional[torch.Tensor]:
    """The tracked form of `graphconv_inference` (float32 rows, shapes of spmi_gemm_rows), or None
    when the layer is not covered and runs as projection + `gcn_aggregate`."""
    if norm not in ("both", "left", "right", "none") or feat.dtype != torch.float32 \
            or weight.dtype != torch.float32 or not _gemm_rows_ok(feat, weight.shape[0], weight.shape[1]):
        return None
    return _GraphConvProjectFirst.apply(feat, weight, bias, g, norm, act, slope)


def project_rows_inference(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor],
                           act: Optional[str], slope: float) -> Optional[torch.Tensor]:
    """act(x @ w + bias) for w [K, N] in one pass over the rows (projection, bias and
    activation in the epilogue of spmi_gemm_rows), or None when the shape is not covered.
    No autograd: for layers that aggregate first and multiply afterwards."""
    if act not in (None, "none", "relu", "leaky_relu") or not _gemm_rows_ok(x, w.shape[0], w.shape[1]):
        return None
    return gemm_rows(x, w.detach().t(), bias, act, slope,
                     planes=weight_planes(w, "f16x3", True, F16X3_GEMM_ROWS_LIMIT))


def _mm(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x @ w (+ bias), w [K, N], no autograd."""
    if _gemm_rows_ok(x, w.shape[0], w.shape[1]):
        return gemm_rows(x, w.detach().t(), bias, planes=weight_planes(w, "f16x3", True, F16X3_GEMM_ROWS_LIMIT))
    return torch.addmm(bias, x, w) if bias is not None else x @ w


def _mm_nk(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x @ weight.T (+ bias), weight [N, K], no autograd."""
    if _gemm_rows_ok(x, weight.shape[1], weight.shape[0]):
        return gemm_rows(x, weight.detach(), bias, planes=weight_planes(weight, "f16x3", False, F16X3_GEMM_ROWS_LIMIT))
    return torch.nn.functional.linear(x, weight, bias)


class _RowsTimesWeight(torch.autograd.Function):
    """y = x @ w (+ bias) for a tall x; the weight gradient x^T dy runs on gemm_tn."""

    @staticmethod
    def forward(ctx, x, w, bias):
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return _mm(x.detach(), w, None if bias is None else bias.detach())

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = g.contiguous()
        g_x = _mm_nk(g, w) if ctx.needs_input_grad[0] else None
        g_w, g_b = _weight_and_bias_grad(g, x, ctx.needs_input_grad[1],
                                         ctx.has_bias and ctx.needs_input_grad[2], dy_first=False)
        return g_x, g_w, g_b


def rows_times_weight(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None
                      ) -> torch.Tensor:
    """x [rows, in] @ w [in, out] (+ bias), differentiable, tuned for rows >> in, out."""
    if x.dim() != 2:
        return torch.addmm(bias, x, w) if bias is not None else x @ w
    if not (x.requires_grad or w.requires_grad or (bias is not None and bias.requires_grad)) \
            or not torch.is_grad_enabled():
        return _mm(x.detach(), w, None if bias is None else bias.detach())
    return _RowsTimesWeight.apply(x, w, bias)


class _Linear(torch.autograd.Function):
    """torch.nn.functional.linear for a tall input; dW = dy^T x runs on gemm_tn."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return _mm_nk(x.detach(), weight, None if bias is None else bias.detach())

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        g_x = _mm(g, weight) if ctx.needs_input_grad[0] else None
        g_w, g_b = _weight_and_bias_grad(g, x, ctx.needs_input_grad[1],
                                         ctx.has_bias and ctx.needs_input_grad[2], dy_first=True)
        return g_x, g_w, g_b


def linear_rows(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None
                ) -> torch.Tensor:
    """F.linear(x, weight, bias) for x [rows, in] with rows >> in, out (differentiable)."""
    if x.dim() != 2:
        return torch.nn.functional.linear(x, weight, bias)
    if not torch.is_grad_enabled() or not (x.requires_grad or weight.requires_grad
                                           or (bias is not None and bias.requires_grad)):
        return _mm_nk(x.detach(), weight, None if bias is None else bias.detach())
    return _Linear.apply(x, weight, bias)


# --------------------------------------------------------------------------- output head
_FINAL_ACT = {None: 0, "none": 0, "sigmoid": 1, "relu": 2, "tanh": 3}


def _int_array(values):
    import ctypes

    return (ctypes.c_int32 * max(len(values), 1))(*values)


def mlp_head_supported(in_width: int, widths) -> bool:
    """Whether the one-kernel head covers these layer widths (input a multiple of 32, every
    layer at most 32 wide, weights small enough for LDS)."""
    import ctypes

    if not widths:
        return False
    rest = _int_array(list(widths[1:]))
    return bool(nv.lib().spmi_mlp_head_supported(in_width, widths[0], len(widths) - 1,
                                                 ctypes.cast(rest, ctypes.c_void_p)))


def mlp_head(x: torch.Tensor, layers=None, final_act: Optional[str] = None, packed=None
             ) -> torch.Tensor:
    """Inference pass of a small MLP over the rows of x in one launch.  `layers`: list of
    (weight [out, in], bias [out] or None, relu: bool) with BatchNorm already folded in;
    or `packed` = pack_mlp_head(layers, device), which callers with fixed weights keep, so
    that a call queues nothing but the kernel."""
    nv.require_cuda(x)
    if x.dtype != torch.float32 or x.dim() != 2:
        raise nv.SpmiError("mlp_head expects a 2-D float32 matrix")
    if x.stride(1) != 1 or x.stride(0) % 4 or x.data_ptr() % 16:
        x = x.contiguous()
    if packed is None:
        packed = pack_mlp_head(layers, x.device)
    w1t, b1p, out1, relu1, rest, outs, relus, n_rest, width = packed[:9]
    out = torch.empty(x.shape[0], width, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        nv.check(nv.lib().spmi_mlp_head_fwd(
            nv.ptr(x), x.stride(0), x.shape[0], w1t.shape[0], nv.ptr(w1t), nv.ptr(b1p), out1, relu1,
            n_rest, nv.ptr(rest), outs, relus, _FINAL_ACT[final_act], nv.ptr(out), width,
            nv.stream()), "spmi_mlp_head_fwd")
    return out


def pack_mlp_head(layers, device):
    """Device buffers in the layout spmi_mlp_head_fwd takes (see include/spmi.h)."""
    w1, b1, relu1 = layers[0]
    out1, in1 = w1.shape
    o1p = 8 if out1 <= 8 else (16 if out1 <= 16 else 32)
    w1t = torch.zeros(in1, o1p, dtype=torch.float32, device=device)
    w1t[:, :out1] = w1.detach().t()
    b1p = torch.zeros(o1p, dtype=torch.float32, device=device)
    if b1 is not None:
        b1p[:out1] = b1.detach()
    pieces, rest_out, rest_relu = [], [], []
    for w, b, relu in layers[1:]:
        pieces.append(w.detach().reshape(-1).float())
        pieces.append(torch.zeros(w.shape[0], device=device) if b is None else b.detach().float())
        rest_out.append(w.shape[0])
        rest_relu.append(int(relu))
    import ctypes

    rest = torch.cat(pieces) if pieces else None
    width = rest_out[-1] if rest_out else out1
    outs = ctypes.cast(_int_array(rest_out), ctypes.c_void_p)
    relus = ctypes.cast(_int_array(rest_relu), ctypes.c_void_p)
    # the ctypes arrays must outlive the casts: keep them on the tuple's tail
    keep = (_int_array(rest_out), _int_array(rest_relu))
    outs = ctypes.cast(keep[0], ctypes.c_void_p)
    relus = ctypes.cast(keep[1], ctypes.c_void_p)
    return (w1t, b1p, out1, int(relu1), rest, outs, relus, len(rest_out), width) + (keep,)


# --------------------------------------------------------------------------- attention
# ---- the output head in training ---------------------------------------------------------------

MLP_TRAIN_MIN_ROWS = 4096


def _pad_width(w: int) -> int:
    return 8 if w <= 8 else (16 if w <= 16 else 32)


def mlp_train_supported(in_width: int, widths) -> bool:
    """The fused training path of the head takes it: a first Linear that spmi_gemm_rows runs
    (its output padded to 32 columns) followed by blocks of widths <= 32."""
    widths = list(widths)
    return (len(widths) >= 2 and all(1 <= w <= 32 for w in widths)
            and bool(nv.lib().spmi_gemm_rows_supported(in_width, 32))
            and bool(nv.lib().spmi_gemm_rows_supported(32, in_width)))


class MlpHeadTrain(torch.autograd.Function):
    """x -> Linear_1 -> [BatchNorm (batch statistics) -> ReLU -> Dropout -> Linear_{p+1}] x k in
    libspmi's row kernels (csrc/mlp_train.hip): the same numbers as the torch modules of
    spiral_amd...simpleDecoder.MLP in train mode, a tenth of the launches.  Arguments: x
    [rows, in]; masks: per block None or the dropout factors [rows, width] (0 or 1 / (1 - p));
    bns: per block (eps, momentum, running_mean, running_var) with the running statistics
    updated in place (None: not tracked); then the parameters W_1, b_1, gamma_1, beta_1, W_2,
    b_2, ..., W_{k+1}, b_{k+1} (biases, gammas, betas may be None)."""

    @staticmethod
    def forward(ctx, x, masks, bns, *params):
        lib = nv.lib()
        k = len(bns)
        w = [params[4 * p] for p in range(k)] + [params[4 * k]]
        b = [params[4 * p + 1] for p in range(k)] + [params[4 * k + 1]]
        gamma = [params[4 * p + 2] for p in range(k)]
        beta = [params[4 * p + 3] for p in range(k)]
        m, dev = x.shape[0], x.device
        widths = [t.shape[0] for t in w]
        pads = [_pad_width(v) for v in widths]
        xd = x.detach()
        f32 = lambda t: None if t is None else t.detach().to(torch.float32).contiguous()
        w_d, b_d = [f32(t) for t in w], [f32(t) for t in b]
        with torch.cuda.device(dev):
            ws_bytes = lib.spmi_mlp_train_workspace_bytes()
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            # z_1 = Linear_1(x) on the projection kernel, its weight padded to 32 output columns
            w1 = torch.zeros(32, w_d[0].shape[1], dtype=torch.float32, device=dev)
            w1[: widths[0]] = w_d[0]
            b1 = torch.zeros(32, dtype=torch.float32, device=dev)
            if b_d[0] is not None:
                b1[: widths[0]] = b_d[0]
            # the later weights and biases zero-padded to the row widths of the matrices around them
            w_p, b_p = [None], [None]
            for q in range(1, k + 1):
                wq = torch.zeros(pads[q], pads[q - 1], dtype=torch.float32, device=dev)
                wq[: widths[q], : widths[q - 1]] = w_d[q]
                w_p.append(wq)
                bq = None
                if b_d[q] is not None:
                    bq = torch.zeros(pads[q], dtype=torch.float32, device=dev)
                    bq[: widths[q]] = b_d[q]
                b_p.append(bq)
            z1_wide = gemm_rows(xd, w1, b1)
            zs = [torch.empty(m, pads[0], dtype=torch.float32, device=dev)]
            stats = torch.zeros(k, 128, dtype=torch.float32, device=dev)
            nv.check(lib.spmi_bn_col_stats(nv.ptr(z1_wide), 32, m, widths[0], pads[0], nv.ptr(zs[0]), nv.ptr(ws),
                                           ws_bytes, nv.stream()), "spmi_bn_col_stats")
            del z1_wide
            for p in range(k):
                eps, momentum, r_mean, r_var = bns[p]
                nv.check(lib.spmi_bn_finalize(nv.ptr(ws), nv.ptr(zs[p]), m, widths[p], nv.ptr(f32(gamma[p])),
                                              nv.ptr(f32(beta[p])), float(eps), float(momentum or 0.0),
                                              nv.ptr(r_mean), nv.ptr(r_var), nv.ptr(stats[p]), nv.stream()),
                         "spmi_bn_finalize")
                out = torch.empty(m, pads[p + 1], dtype=torch.float32, device=dev)
                last = p + 1 == k
                nv.check(lib.spmi_bn_act_linear_fwd(nv.ptr(zs[p]), m, widths[p], pads[p], nv.ptr(stats[p]),
                                                    nv.ptr(masks[p]), nv.ptr(w_p[p + 1]), nv.ptr(b_p[p + 1]),
                                                    widths[p + 1], pads[p + 1], nv.ptr(out),
                                                    None if last else nv.ptr(ws), nv.stream()),
                         "spmi_bn_act_linear_fwd")
                zs.append(out)
        ctx.save_for_backward(xd, stats, w1, *zs[:-1], *[t for t in masks if t is not None], *w_p[1:])
        ctx.layout = (k, widths, pads, [t is not None for t in masks], [t is not None for t in b],
                      [t is not None for t in gamma], [t is not None for t in beta])
        return zs[-1][:, : widths[-1]]

    @staticmethod
    def backward(ctx, g_out):
        lib = nv.lib()
        k, widths, pads, has_mask, has_b, has_gamma, has_beta = ctx.layout
        saved = list(ctx.saved_tensors)
        xd, stats, w1 = saved[:3]
        zs = saved[3: 3 + k]
        rest = saved[3 + k:]
        masks = [rest.pop(0) if has_mask[p] else None for p in range(k)]
        w_next = rest                                   # W_2 .. W_{k+1}
        m, dev = xd.shape[0], xd.device
        grads = [None] * (4 * k + 2)
        with torch.cuda.device(dev):
            ws_bytes = lib.spmi_mlp_train_workspace_bytes()
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            g = torch.zeros(m, pads[k], dtype=torch.float32, device=dev)
            g[:, : widths[k]] = g_out
            for p in range(k - 1, -1, -1):
                dy = torch.empty(m, pads[p], dtype=torch.float32, device=dev)
                sums = torch.empty(3 * 32 + 32 * 32, dtype=torch.float32, device=dev)
                nv.check(lib.spmi_bn_act_linear_bwd(nv.ptr(zs[p]), m, widths[p], pads[p], nv.ptr(stats[p]),
                                                    nv.ptr(masks[p]), nv.ptr(w_next[p]), widths[p + 1],
                                                    pads[p + 1], nv.ptr(g), nv.ptr(dy), nv.ptr(ws), ws_bytes,
                                                    nv.ptr(sums), nv.stream()), "spmi_bn_act_linear_bwd")
                if ctx.needs_input_grad[3 + 4 * (p + 1)]:
                    grads[4 * (p + 1)] = sums[96:].view(32, 32)[: widths[p + 1], : widths[p]].contiguous()
                if has_b[p + 1] and ctx.needs_input_grad[3 + 4 * (p + 1) + 1]:
                    grads[4 * (p + 1) + 1] = sums[64: 64 + widths[p + 1]].clone()
                if has_gamma[p] and ctx.needs_input_grad[3 + 4 * p + 2]:
                    grads[4 * p + 2] = sums[32: 32 + widths[p]].clone()
                if has_beta[p] and ctx.needs_input_grad[3 + 4 * p + 3]:
                    grads[4 * p + 3] = sums[: widths[p]].clone()
                nv.check(lib.spmi_bn_bwd_apply(nv.ptr(dy), nv.ptr(zs[p]), m, widths[p], pads[p], nv.ptr(stats[p]),
                                               nv.ptr(sums), nv.ptr(dy), nv.stream()), "spmi_bn_bwd_apply")
                g = dy                                    # dz_p, in place of dy
            # Linear_1: dz_1 [rows, pad] (zero past the width) against x
            g_x = None
            if ctx.needs_input_grad[0]:
                if pads[0] == 32:
                    g32 = g
                else:
                    g32 = torch.zeros(m, 32, dtype=torch.float32, device=dev)
                    g32[:, : pads[0]] = g
                g_x = gemm_rows(g32, w1.t().contiguous())
            need_w1, need_b1 = ctx.needs_input_grad[3], has_b[0] and ctx.needs_input_grad[4]
            g_w1, g_b1 = _weight_and_bias_grad(g[:, : widths[0]], xd, need_w1, need_b1, dy_first=True)
            grads[0], grads[1] = g_w1, g_b1
        return (g_x, None, None, *grads)


def mlp_head_train(x: torch.Tensor, layers, masks) -> torch.Tensor:
    """The training forward of a head made of `layers` = [(linear, batch_norm), ..., (linear, None)]
    (torch modules) with the dropout factors `masks` (one per batch norm, None for no dropout);
    differentiable in x and every parameter, running statistics updated as BatchNorm1d does."""
    bns, params = [], []
    for lin, bn in layers[:-1]:
        tracked = bn.track_running_stats and bn.running_mean is not None
        bns.append((bn.eps, bn.momentum, bn.running_mean if tracked else None,
                    bn.running_var if tracked else None))
        params += [lin.weight, lin.bias, bn.weight, bn.bias]
        if tracked and bn.num_batches_tracked is not None:
            bn.num_batches_tracked.add_(1)
    params += [layers[-1][0].weight, layers[-1][0].bias]
    return MlpHeadTrain.apply(x, list(masks), bns, *params)


def _heavy(csr):
    return nv.ptr(csr.heavy_rows), csr.n_heavy, csr.heavy_thresh


def edge_softmax_scores(csr, el: torch.Tensor, er: torch.Tensor, slope: float, keep_raw: bool):
    """attn[h, j] = softmax over the entries j of a row of LeakyReLU(el[src_j, h] + er[row, h]);
    returns (attn, raw) head-major [H, E] in CSR order (raw = logits before LeakyReLU)."""
    nv.require_cuda(el, er)
    heads = el.shape[1]
    attn = torch.empty(heads, csr.nnz, dtype=torch.float32, device=el.device)
    raw = torch.empty_like(attn) if keep_raw else None
    with torch.cuda.device(el.device):
        nv.check(nv.lib().spmi_gat_edge_softmax(
            nv.ptr(csr.indptr), nv.ptr(csr.indices), csr.n_rows, csr.nnz, nv.ptr(el), nv.ptr(er),
            heads, float(slope), nv.ptr(attn), nv.ptr(raw), *_heavy(csr), nv.stream()),
            "spmi_gat_edge_softmax")
    return attn, raw


def _csr_to_csr_t_positions(g) -> torch.Tensor:
    """For every position of the by-source CSR, the position of the same edge in the
    by-destination CSR (cached on the graph)."""
    perm = getattr(g, "_t2f", None)
    if perm is None:
        csr, csr_t = g.csr(want_eid=True), g.csr_t(want_eid=True)
        inv = torch.empty(csr.nnz, dtype=torch.int64, device=csr.eid.device)
        inv[csr.eid.long()] = torch.arange(csr.nnz, device=csr.eid.device)
        perm = g._t2f = inv[csr_t.eid.long()]
    return perm


class GatAggregate(torch.autograd.Function):
    """out[v, h, :] = act(sum_{u->v} a_uv^h ft[u, h, :] + bias) with a = edge softmax of
    LeakyReLU(el[u] + er[v]) over the incoming edges of v (dgl GATConv.forward between its
    dense projections).  ft [N, H*D], el / er [N, H]."""

    @staticmethod
    def forward(ctx, ft, el, er, bias, g, heads, slope, act, act_slope, attn_drop):
        csr = g.csr()
        n, width = ft.shape
        d = width // heads
        need_grad = _needs_grad(ctx, 3)
        ft_c, el_c, er_c = ft.detach().contiguous(), el.detach().contiguous(), er.detach().contiguous()
        attn, raw = edge_softmax_scores(csr, el_c, er_c, slope, need_grad)
        # attention dropout (dgl GATConv: `attn_drop(edge_softmax(graph, e))`): the weights that
        # survive are scaled by 1 / (1 - p); `used` is what the weighted sum multiplies by
        keep = None
        if attn_drop > 0.0:
            keep = (torch.rand_like(attn) >= attn_drop).to(attn.dtype) / (1.0 - attn_drop)
        used = attn if keep is None else attn * keep
        out = torch.empty(csr.n_rows, width, dtype=torch.float32, device=ft.device)
        b = None if bias is None else bias.detach()
        for h in range(heads):
            cols = slice(h * d, (h + 1) * d)
            aggregate(csr, ft_c[:, cols], None, used[h], None, None if b is None else b[cols], act,
                      act_slope, out=out[:, cols])
        ctx.g, ctx.heads, ctx.slope, ctx.act, ctx.act_slope = g, heads, slope, act, act_slope
        ctx.has_bias = bias is not None
        ctx.save_for_backward(ft_c, attn, raw, out if act not in (None, "none") else None, keep)
        ctx.mark_non_differentiable(used)
        return out, used

    @staticmethod
    def backward(ctx, g_out, _g_attn):
        ft, attn, raw, out, keep = ctx.saved_tensors
        g, heads = ctx.g, ctx.heads
        need_b = ctx.needs_input_grad[3] and ctx.has_bias
        g_p, g_b = g_out.contiguous(), None
        if out is not None or need_b:
            g_p, g_b = _act_bias_bwd(g_p, out if out is not None else g_p, ctx.act, ctx.act_slope,
                                     need_b)
        csr, csr_t = g.csr(), g.csr_t()
        t2f = _csr_to_csr_t_positions(g)
        csr, csr_t = g.csr(), g.csr_t()  # the calls above may have rebuilt them with edge ids
        n_src, width = ft.shape
        d = width // heads
        lib = nv.lib()
        if not lib.spmi_sddmm_dot_supported(heads, d):
            raise nv.SpmiError(f"attention backward needs a head width that is a power of two below "
                               f"64 or a multiple of 64 (got {d}) and heads*width <= 512")
        # gradient of the projected source rows through the weighted sum
        attn_t = (attn if keep is None else attn * keep).index_select(1, t2f)
        g_ft = torch.empty_like(ft)
        for h in range(heads):
            cols = slice(h * d, (h + 1) * d)
            aggregate(csr_t, g_p[:, cols], None, attn_t[h], None, None, None, out=g_ft[:, cols])
        with torch.cuda.device(ft.device):
            d_attn = torch.empty_like(attn)
            nv.check(lib.spmi_sddmm_dot(nv.ptr(csr.indptr), nv.ptr(csr.indices), csr.n_rows, csr.nnz,
                                        nv.ptr(g_p), g_p.stride(0), nv.ptr(ft), ft.stride(0), heads, d,
                                        nv.ptr(d_attn), nv.stream()), "spmi_sddmm_dot")
            if keep is not None:
                d_attn = d_attn * keep   # through the dropout: dropped weights pass no gradient
            d_raw = torch.empty_like(attn)
            g_er = torch.empty(csr.n_rows, heads, dtype=torch.float32, device=ft.device)
            nv.check(lib.spmi_gat_edge_softmax_bwd(
                nv.ptr(csr.indptr), csr.n_rows, csr.nnz, nv.ptr(attn), nv.ptr(raw), nv.ptr(d_attn),
                heads, float(ctx.slope), nv.ptr(d_raw), nv.ptr(g_er), *_heavy(csr), nv.stream()),
                "spmi_gat_edge_softmax_bwd")
            d_raw_t = d_raw.index_select(1, t2f)
            g_el = torch.empty(n_src, heads, dtype=torch.float32, device=ft.device)
            nv.check(lib.spmi_segment_sum(nv.ptr(csr_t.indptr), csr_t.n_rows, csr_t.nnz,
                                          nv.ptr(d_raw_t), heads, nv.ptr(g_el), *_heavy(csr_t),
                                          nv.stream()), "spmi_segment_sum")
        return g_ft, g_el, g_er, g_b, None, None, None, None, None, None


def gat_aggregate(g, ft: torch.Tensor, el: torch.Tensor, er: torch.Tensor, heads: int,
                  negative_slope: float = 0.2, bias: Optional[torch.Tensor] = None,
                  act: Optional[str] = None, act_slope: float = 0.01, attn_drop: float = 0.0):
    """Differentiable attention aggregation; returns (out [N, H*D], attn [H, E] in the order
    of g.csr()).  `attn_drop` > 0 drops attention weights with that probability (training
    only: the caller passes 0 in eval mode); the returned weights are the ones used."""
    nv.require_cuda(ft, el, er, bias)
    if ft.dtype != torch.float32 or ft.dim() != 2 or ft.shape[1] % heads:
        raise nv.SpmiError(f"gat_aggregate expects float32 [N, heads*D] features, got {tuple(ft.shape)}")
    return _apply_tracked(GatAggregate, (ft, el, er, bias), ft, el, er, bias, g, heads, float(negative_slope), act,
                          float(act_slope), float(attn_drop))


# --------------------------------------------------------------------------- recurrent cells
# Inside Function.forward grad mode is always off and ctx.needs_input_grad is True for every
# parameter that requires grad, torch.no_grad() or not.  Whether a call is really tracked
# (and must keep gate values for its backward) is therefore decided by the wrappers below,
# before apply(): an inference step must not allocate and write M x 5H floats per layer.
_track = threading.local()


def _apply_tracked(fn, tensors, *args, x_bounded: bool = False):
    prev = getattr(_track, "on", True), getattr(_track, "x_bounded", False)
    _track.on = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)
    _track.x_bounded = bool(x_bounded)
    try:
        return fn.apply(*args)
    finally:
        _track.on, _track.x_bounded = prev


def _needs_grad(ctx, n: int) -> bool:
    return getattr(_track, "on", True) and any(ctx.needs_input_grad[:n])


def _rows(mem: torch.Tensor, ids: Optional[torch.Tensor], m: int) -> torch.Tensor:
    """Dense [M,H] view/copy of the memory rows a batch reads (GEMM operand)."""
    if ids is None:
        return mem[:m]
    out = torch.empty(m, mem.shape[1], dtype=torch.float32, device=mem.device)
    with torch.cuda.device(mem.device):
        nv.check(nv.lib().spmi_gather_rows(nv.ptr(mem), mem.stride(0), nv.ptr(ids), m,
                                           mem.shape[1], nv.ptr(out), nv.stream()),
                 "spmi_gather_rows")
    return out


def _check_cell_args(x, mem, ids):
    nv.require_cuda(x, mem, ids)
    if x.dtype != torch.float32 or mem.dtype != torch.float32:
        raise nv.SpmiError("recurrent cells compute in float32")
    if not mem.is_contiguous():
        raise nv.SpmiError("node memory layer must be contiguous")
    if ids is not None and (ids.dtype != torch.int64 or ids.numel() != x.shape[0]):
        raise nv.SpmiError("ids must be int64 with one entry per batch row")
    if ids is None and x.shape[0] > mem.shape[0]:
        raise nv.SpmiError("batch larger than the node memory")


_CELL_CODE = {"gru": 0, "lstm": 1, "rnn": 2}
USE_FUSED_CELLS = True  # one-kernel layer step when the shapes allow it (I % 32, H % 32)


def _fused_ok(x, w_hh) -> bool:
    return (USE_FUSED_CELLS and x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0
            and bool(nv.lib().spmi_cell_fused_supported(x.shape[1], w_hh.shape[1])))


# How the fused layer step multiplies:
#   "f16x3"  = every operand as the fp16 planes fp16(v), fp16(v/64), fp16((v - fp16(v))*64);
#              three matrix instructions per term; sums land 3-8x closer to float64 than an
#              fp32 FMA chain.  The state and the weights must lie within +-65504 (a larger
#              one shows up as inf/NaN in its output row); rows of x of any finite magnitude
#              are scaled by a power of two inside the kernel;
#   "bf16x3"  = fp32 operands split exactly into three bf16 pieces, six piece products per
#               term (fp32-accurate over the whole fp32 range, twice the matrix work);
#   "fp32"    = the fp32 matrix instruction (an exact fp32 FMA chain).
CELL_GEMM = "f16x3"
# Whole-table GRU steps (every node, inference) on the kernel that keeps the gate weights in
# LDS and writes the new state to a second slab (csrc/cell_rows.hip).
USE_GRU_ROWS = True


def split_bf16x3(w: torch.Tensor) -> torch.Tensor:
    """The weight matrix [rows, cols] as exact bf16 pieces (hi + mid + lo == w), tiled
    [cols/16, 3, 2, rows, 8] (K-step, piece, half of the K-step, row, k) for the
    split-precision cell kernel."""
    w = w.detach().contiguous()
    rows, cols = w.shape
    out = torch.empty(cols // 16, 3, 2, rows, 8, dtype=torch.bfloat16, device=w.device)
    with torch.cuda.device(w.device):
        nv.check(nv.lib().spmi_split_bf16x3(nv.ptr(w), rows, cols, nv.ptr(out), nv.stream()),
                 "spmi_split_bf16x3")
    return out


def split_f16x3(w: torch.Tensor, amax_slot: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The weight matrix [rows, cols] as the planes fp16(w), fp16(w/64), fp16((w - fp16(w))*64),
    tiled [cols/16, 3, 2, rows, 8] (K-step, plane, half of the K-step, row, k) for the
    three-product kernels.  `w` may be any 2-D view with positive strides (a `.t()` view is read
    through its strides, not copied).  `amax_slot`: a zeroed int32 device scalar that receives
    the bit pattern of the largest |w| (NaN included) from the same launch."""
    w = w.detach()
    if w.dim() != 2 or w.dtype != torch.float32 or min(w.stride()) < 1:
        w = w.contiguous().float()
    rows, cols = w.shape
    out = torch.empty(cols // 16, 3, 2, rows, 8, dtype=torch.float16, device=w.device)
    with torch.cuda.device(w.device):
        nv.check(nv.lib().spmi_split_f16x3_strided(
            nv.ptr(w), rows, cols, w.stride(0), w.stride(1), nv.ptr(out), nv.ptr(amax_slot),
            nv.stream()), "spmi_split_f16x3_strided")
    return out


# Operand planes of a weight are kept per tensor object for as long as the tensor is alive and
# unchanged (same version counter, same storage address): an inference loop then launches no
# split kernel and no transpose copy per call, a training loop re-splits after every update.
#
# What the key cannot see: an in-place edit made through `.data` (`p.data.mul_(..)`,
# `p.data -= lr * g`) bumps no version counter and moves no storage.  Code that updates weights
# that way calls `invalidate_weight_caches()` afterwards; the modules of this package call it
# from `train()` / `eval()` and after `load_state_dict`.
_plane_cache: dict = {}
_cache_epoch = 0

# |w| limits of the fp16-plane kernels: a weight beyond them turns into +-inf in its first
# plane and every output row into inf/NaN.
F16X3_CELL_LIMIT = 65504.0
F16X3_GEMM_ROWS_LIMIT = 1023.0


def invalidate_weight_caches() -> None:
    """Forget every cached operand image of a weight (the fp16/bf16 planes of the dense
    kernels, the folded and packed layers of the one-kernel MLP head): the next use rebuilds
    them from the weights as they are then.  Needed after edits the cache keys cannot see,
    i.e. in-place changes through `.data`."""
    global _cache_epoch
    _cache_epoch += 1
    _plane_cache.clear()


def cache_epoch() -> int:
    return _cache_epoch


class WeightCacheHooks:
    """Mixin for the modules whose inference kernels read cached operand images of their
    weights: switching between train() and eval() and loading a state dict drop the images, so
    that a weight edited through `.data` in between (an optimiser written by hand, a restored
    checkpoint) is never multiplied in its old form."""

    def train(self, mode: bool = True):
        if bool(mode) != self.training:   # a call that changes nothing keeps the images
            invalidate_weight_caches()
        return super().train(mode)

    def _load_from_state_dict(self, *args, **kwargs):
        # runs for this module also when a parent's load_state_dict recurses into it
        super()._load_from_state_dict(*args, **kwargs)
        invalidate_weight_caches()


def _amax_slot(device: torch.device) -> torch.Tensor:
    """A zeroed int32 device scalar for one domain check, valid on the current stream
    (`_native.zeroed_slot`: no fill launch per check)."""
    return nv.zeroed_slot(device, 1)


class _DomainCheck:
    """max |w| of a weight on its way to pinned host memory, so that the domain check of the
    fp16-plane kernels never makes the host wait for the device: the value is looked at the
    first time the planes are used again after the copy has landed.  `slot`: the scalar the
    split kernel left the bit pattern of max |w| in."""

    def __init__(self, slot: torch.Tensor, limit: float):
        self.limit = limit
        self.host = torch.empty(1, dtype=torch.float32, pin_memory=True)
        self.host.copy_(slot.view(torch.float32), non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record()

    def done(self, wait: bool = False) -> bool:
        if wait:
            self.event.synchronize()
        elif not self.event.query():
            return False
        amax = float(self.host[0])
        if not amax <= self.limit:
            raise nv.SpmiError(
                f"a weight of magnitude {amax:g} is outside the fp16-plane kernels' domain "
                f"(|w| <= {self.limit:g}): results computed from it are inf/NaN; set "
                "ops.CELL_GEMM = 'bf16x3' or 'fp32' / ops.DENSE_GEMM = 'library' for this model")
        return True


def weight_planes(w: torch.Tensor, kind: str = "f16x3", transposed: bool = False,
                  limit: Optional[float] = None) -> torch.Tensor:
    """`split_f16x3` / `split_bf16x3` of `w` (of `w.T` when `transposed`), cached.  `limit`: the
    largest |w| the consuming kernel takes.  It is checked whenever planes are (re)built, without
    a host wait: the maximum travels to the host behind the split and is read at a later use of
    the same weight (or when its planes are rebuilt), where an out-of-domain weight raises."""
    # the limit is part of the key: a weight whose planes were checked against a kernel with a
    # wide domain is split (and checked) again for a kernel with a narrower one
    slot_id = (id(w), kind, transposed, limit)
    key = (w._version, w.data_ptr(), tuple(w.shape), w.stride(), w.device)
    slot = _plane_cache.get(slot_id)
    if slot is not None and slot[0]() is w:
        if slot[3] is not None and slot[3].done(wait=slot[1] != key):
            slot = _plane_cache[slot_id] = slot[:3] + (None,)
        if slot[1] == key:
            return slot[2]
    src = w.detach().t() if transposed else w.detach()
    check = None
    if kind != "f16x3":
        planes = split_bf16x3(src)
    elif limit is not None and src.numel():
        with torch.cuda.device(w.device):
            slot = _amax_slot(w.device)
            planes = split_f16x3(src, slot)
            check = _DomainCheck(slot, limit)
    else:
        planes = split_f16x3(src)
    ref = weakref.ref(w, lambda _r, k=slot_id: _plane_cache.pop(k, None))
    _plane_cache[slot_id] = (ref, key, planes, check)
    return planes


def _fused_cell(kind, x, w_ih, w_hh, b_ih, b_hh, h_mem, c_mem, ids, mem_write, h_out, c_out, saved):
    """Both gate GEMMs + gates + memory write-back in one launch (csrc/cell_split.hip or
    csrc/cell_fused.hip, see CELL_GEMM)."""
    if CELL_GEMM in ("bf16x3", "f16x3"):
        name = "spmi_cell_split_fwd" if CELL_GEMM == "bf16x3" else "spmi_cell_f16x3_fwd"
        ws_ih = weight_planes(w_ih, CELL_GEMM, False, F16X3_CELL_LIMIT)
        ws_hh = weight_planes(w_hh, CELL_GEMM, False, F16X3_CELL_LIMIT)
        with torch.cuda.device(x.device):
            nv.check(getattr(nv.lib(), name)(
                _CELL_CODE[kind], nv.ptr(x), x.stride(0), x.shape[1], nv.ptr(ws_ih), nv.ptr(ws_hh),
                nv.ptr(b_ih), nv.ptr(b_hh), nv.ptr(h_mem), nv.ptr(c_mem), nv.ptr(ids), x.shape[0],
                w_hh.shape[1], nv.ptr(h_out), nv.ptr(c_out), int(mem_write), nv.ptr(saved),
                *(() if CELL_GEMM == "bf16x3" else (int(getattr(_track, "x_bounded", False)),)),
                nv.stream()), name)
        return
    if CELL_GEMM != "fp32":
        raise ValueError(f"unknown CELL_GEMM {CELL_GEMM!r}")
    wt_ih = w_ih.detach().t().contiguous()
    wt_hh = w_hh.detach().t().contiguous()
    with torch.cuda.device(x.device):
        nv.check(nv.lib().spmi_cell_fused_fwd(
            _CELL_CODE[kind], nv.ptr(x), x.stride(0), x.shape[1], nv.ptr(wt_ih), nv.ptr(wt_hh),
            nv.ptr(b_ih), nv.ptr(b_hh), nv.ptr(h_mem), nv.ptr(c_mem), nv.ptr(ids), x.shape[0],
            w_hh.shape[1], nv.ptr(h_out), nv.ptr(c_out), int(mem_write), nv.ptr(saved),
            nv.stream()), "spmi_cell_fused_fwd")


def _whole_table_ok(supported: str, x: torch.Tensor, w_hh: torch.Tensor, params) -> bool:
    """Common conditions of the whole-table cell kernels (csrc/cell_rows.hip): fp16-plane
    arithmetic selected, a layout the kernel reads, a shape `supported` (the library's predicate
    of that name) accepts, and nothing tracked for a backward pass.  The kernels keep no graph,
    so one trainable tensor among x and the layer's parameters under grad mode sends the layer
    to the tracked cell instead."""
    tracked = torch.is_grad_enabled() and any(
        t is not None and t.requires_grad for t in (x, w_hh) + tuple(params))
    return (USE_FUSED_CELLS and USE_GRU_ROWS and CELL_GEMM == "f16x3" and x.is_cuda
            and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
            and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0 and not tracked
            and bool(getattr(nv.lib(), supported)(x.shape[1], w_hh.shape[1])))


def gru_rows_ok(x: torch.Tensor, w_hh: torch.Tensor, *params: Optional[torch.Tensor]) -> bool:
    """The whole-table GRU kernel takes this layer.  `params`: the layer's other parameters
    (w_ih, b_ih, b_hh)."""
    return _whole_table_ok("spmi_gru_rows_supported", x, w_hh, params)


def gru_rows(x: torch.Tensor, w_ih, w_hh, b_ih, b_hh, h_in: torch.Tensor, h_out: torch.Tensor,
             x_bounded: bool = False) -> torch.Tensor:
    """One GRU layer step for every row: reads the state slab `h_in` [M, H], writes the new
    state to `h_out` [M, H] (a different slab: the kernel cannot update in place) and returns
    it.  Inference only (no gate values are kept)."""
    _check_cell_args(x, h_in, None)
    m, hd = x.shape[0], w_hh.shape[1]
    if h_in.shape != (m, hd) or h_out.shape != (m, hd) or not h_out.is_contiguous() \
            or h_out.dtype != torch.float32 or h_out.device != x.device:
        raise nv.SpmiError("gru_rows: h_in and h_out must be contiguous float32 [M, H] slabs on x's device")
    ws_ih = weight_planes(w_ih, "f16x3", False, F16X3_CELL_LIMIT)
    ws_hh = weight_planes(w_hh, "f16x3", False, F16X3_CELL_LIMIT)
    with torch.cuda.device(x.device):
        nv.check(nv.lib().spmi_gru_rows_fwd(
            nv.ptr(x), x.stride(0), x.shape[1], nv.ptr(ws_ih), nv.ptr(ws_hh), nv.ptr(b_ih),
            nv.ptr(b_hh), nv.ptr(h_in), m, hd, nv.ptr(h_out), int(x_bounded), nv.stream()),
            "spmi_gru_rows_fwd")
    return h_out


def rnn_rows_ok(x: torch.Tensor, w_hh: torch.Tensor, *params: Optional[torch.Tensor]) -> bool:
    """The whole-table tanh-RNN kernel takes this layer (conditions as for `gru_rows_ok`)."""
    return _whole_table_ok("spmi_rnn_rows_supported", x, w_hh, params)


def rnn_rows(x: torch.Tensor, w_ih, w_hh, b_ih, b_hh, h_in: torch.Tensor, h_out: torch.Tensor,
             x_bounded: bool = False) -> torch.Tensor:
    """One tanh-RNN layer step for every row: reads the state slab `h_in` [M, H], writes the new
    state to `h_out` [M, H] (a different slab) and returns it.  Inference only."""
    _check_cell_args(x, h_in, None)
    m, hd = x.shape[0], w_hh.shape[1]
    if h_in.shape != (m, hd) or h_out.shape != (m, hd) or not h_out.is_contiguous() \
            or h_out.dtype != torch.float32 or h_out.device != x.device:
        raise nv.SpmiError("rnn_rows: h_in and h_out must be contiguous float32 [M, H] slabs on x's device")
    ws_ih = weight_planes(w_ih, "f16x3", False, F16X3_CELL_LIMIT)
    ws_hh = weight_planes(w_hh, "f16x3", False, F16X3_CELL_LIMIT)
    with torch.cuda.device(x.device):
        nv.check(nv.lib().spmi_rnn_rows_fwd(
            nv.ptr(x), x.stride(0), x.shape[1], nv.ptr(ws_ih), nv.ptr(ws_hh), nv.ptr(b_ih),
            nv.ptr(b_hh), nv.ptr(h_in), m, hd, nv.ptr(h_out), int(x_bounded), nv.stream()),
            "spmi_rnn_rows_fwd")
    return h_out


def lstm_rows_ok(x: torch.Tensor, w_hh: torch.Tensor, *params: Optional[torch.Tensor]) -> bool:
    """The whole-table LSTM kernel takes this layer (conditions as for `gru_rows_ok`)."""
    return _whole_table_ok("spmi_lstm_rows_supported", x, w_hh, params)


def lstm_rows(x: torch.Tensor, w_ih, w_hh, b_ih, b_hh, h_in: torch.Tensor, c_in: torch.Tensor,
              h_out: torch.Tensor, c_out: torch.Tensor, x_bounded: bool = False):
    """One LSTM layer step for every row: reads the state slabs `h_in`, `c_in` [M, H], writes
    `h_out` (a different slab from h_in) and `c_out` (may be c_in) and returns them.  Inference
    only (no gate values are kept)."""
    _check_cell_args(x, h_in, None)
    nv.require_cuda(c_in, h_out, c_out)
    m, hd = x.shape[0], w_hh.shape[1]
    for t in (h_in, c_in, h_out, c_out):
        if t.shape != (m, hd) or not t.is_contiguous() or t.dtype != torch.float32 or t.device != x.device:
            raise nv.SpmiError("lstm_rows: state slabs must be contiguous float32 [M, H] on x's device")
    ws_ih = weight_planes(w_ih, "f16x3", False, F16X3_CELL_LIMIT)
    ws_hh = weight_planes(w_hh, "f16x3", False, F16X3_CELL_LIMIT)
    with torch.cuda.device(x.device):
        nv.check(nv.lib().spmi_lstm_rows_fwd(
            nv.ptr(x), x.stride(0), x.shape[1], nv.ptr(ws_ih), nv.ptr(ws_hh), nv.ptr(b_ih),
            nv.ptr(b_hh), nv.ptr(h_in), nv.ptr(c_in), m, hd, nv.ptr(h_out), nv.ptr(c_out),
            int(x_bounded), nv.stream()), "spmi_lstm_rows_fwd")
    return h_out, c_out


# GRU layers hand their gate gradients to the dense kernels as one [rows, 4H] matrix instead of
# two [rows, 3H] ones (see GruCell._backward_packed).
GRU_PACKED_BACKWARD = True
# ... and take both weight gradients of a layer from one launch over that matrix (spmi_gemm_tn_pair_sums).
GRU_PAIRED_WEIGHT_GRADS = True


def _gru_grads_from_packed(ctx, g4, xd, h_old, w_ih):
    """(g_x, g_w_ih, g_w_hh, g_b_ih, g_b_hh) of a GRU layer from its packed gate gradients
    g4 = [n | r | z | n*r] ([M, 4H], see GruCell._backward_packed), the layer's input rows and
    its old state."""
    hd = g4.shape[1] // 4
    g_i, g_hh = g4[:, : 3 * hd], g4[:, hd:]
    nrz = lambda t: torch.cat([t[2 * hd:], t[: 2 * hd]])      # torch's r, z, n -> n, r, z
    rzn = lambda t: torch.cat([t[hd:], t[:hd]])               # and back
    g_x = None
    if ctx.needs_input_grad[0]:
        w_nrz = nrz(w_ih.detach())                             # [3H, I]
        g_x = gemm_rows(g_i, w_nrz.t()) if _gemm_rows_ok(g_i, 3 * hd, w_ih.shape[1]) else g_i @ w_nrz
    need_b = ctx.has_bias and (ctx.needs_input_grad[3] or ctx.needs_input_grad[4])
    need_bhh = ctx.has_bias and ctx.needs_input_grad[4]
    if (GRU_PAIRED_WEIGHT_GRADS and ctx.needs_input_grad[1] and ctx.needs_input_grad[2]
            and gemm_tn_pair_ok(g_i, xd, g_hh, h_old)):
        # both weight gradients from one launch: the r and z columns of g4 are read once
        g_wih, g_whh, g_bih, g_bhh = gemm_tn_pair(g_i, xd, g_hh, h_old, sums=True)
        if not need_b:
            g_bih = None
        if not need_bhh:
            g_bhh = None
    else:
        g_wih, g_bih = _weight_and_bias_grad(g_i, xd, ctx.needs_input_grad[1], need_b, dy_first=True)
        g_whh, g_bhh = _weight_and_bias_grad(g_hh, h_old, ctx.needs_input_grad[2], need_bhh, dy_first=True)
    if g_wih is not None:
        g_wih = rzn(g_wih)
    g_bih = rzn(g_bih) if (g_bih is not None and ctx.has_bias and ctx.needs_input_grad[3]) else None
    return g_x, g_wih, g_whh, g_bih, g_bhh


class GruCell(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, h_mem, ids, mem_write):
        m, hd = x.shape[0], w_hh.shape[1]
        xd = x.detach()
        need_grad = _needs_grad(ctx, 5)
        saved = torch.empty(m, 5 * hd, dtype=torch.float32, device=x.device) if need_grad else None
        h_out = torch.empty(m, hd, dtype=torch.float32, device=x.device)
        if _fused_ok(xd, w_hh):
            _fused_cell("gru", xd, w_ih, w_hh, b_ih, b_hh, h_mem, None, ids, mem_write, h_out, None,
                        saved)
        else:
            h_old = _rows(h_mem, ids, m)
            gi = xd @ w_ih.detach().t()
            gh = h_old @ w_hh.detach().t()
            with torch.cuda.device(x.device):
                nv.check(nv.lib().spmi_gru_gate_fwd(
                    nv.ptr(gi), nv.ptr(gh), nv.ptr(b_ih), nv.ptr(b_hh), nv.ptr(h_mem), nv.ptr(ids), m,
                    hd, nv.ptr(h_out), int(mem_write), nv.ptr(saved), nv.stream()),
                    "spmi_gru_gate_fwd")
        ctx.save_for_backward(xd, w_ih, w_hh, saved)
        ctx.has_bias = b_ih is not None
        return h_out

    @staticmethod
    def _backward_packed(ctx, g_h):
        """The backward with the gate gradients as one [M, 4H] matrix [n | r | z | n*r] (one
        write, read back as two overlapping [M, 3H] views): columns 0..3H are the gradient of
        x W_ih^T with the gates in the order n, r, z, columns H..4H that of h W_hh^T in torch's
        order r, z, n.  dX comes from spmi_gemm_rows with the rows of W_ih reordered to match."""
        xd, w_ih, w_hh, saved = ctx.saved_tensors
        m, hd = g_h.shape
        g4 = torch.empty(m, 4 * hd, dtype=torch.float32, device=g_h.device)
        with torch.cuda.device(g_h.device):
            nv.check(nv.lib().spmi_gru_gate_bwd_packed(nv.ptr(g_h), nv.ptr(saved), m, hd, nv.ptr(g4),
                                                       None, nv.stream()), "spmi_gru_gate_bwd_packed")
        return _gru_grads_from_packed(ctx, g4, xd, saved[:, 4 * hd:], w_ih) + (None, None, None)

    @staticmethod
    def backward(ctx, g_h):
        xd, w_ih, w_hh, saved = ctx.saved_tensors
        m, hd = g_h.shape
        g_h = g_h.contiguous()
        if GRU_PACKED_BACKWARD and hd % 4 == 0:
            return GruCell._backward_packed(ctx, g_h)
        g_gi = torch.empty(m, 3 * hd, dtype=torch.float32, device=g_h.device)
        g_gh = torch.empty(m, 3 * hd, dtype=torch.float32, device=g_h.device)
        with torch.cuda.device(g_h.device):
            nv.check(nv.lib().spmi_gru_gate_bwd(nv.ptr(g_h), nv.ptr(saved), m, hd, nv.ptr(g_gi),
                                                nv.ptr(g_gh), None, nv.stream()),
                     "spmi_gru_gate_bwd")
        h_old = saved[:, 4 * hd:]
        g_x = g_gi @ w_ih if ctx.needs_input_grad[0] else None
        need_b = ctx.has_bias and (ctx.needs_input_grad[3] or ctx.needs_input_grad[4])
        need_bhh = ctx.has_bias and ctx.needs_input_grad[4]
        # bias gradients = column sums of the gate gradients, taken by the weight-gradient
        # kernels on their way through g_gi / g_gh
        g_wih, g_bih = _weight_and_bias_grad(g_gi, xd, ctx.needs_input_grad[1], need_b, dy_first=True)
        g_whh, s_gh = _weight_and_bias_grad(g_gh, h_old, ctx.needs_input_grad[2], need_bhh, dy_first=True)
        g_bhh = None
        if need_bhh:
            # the r and z columns of g_gi and g_gh are the same numbers
            g_bhh = torch.cat([g_bih[: 2 * hd], s_gh[2 * hd:]])
        if not (ctx.has_bias and ctx.needs_input_grad[3]):
            g_bih = None
        return g_x, g_wih, g_whh, g_bih, g_bhh, None, None, None


def gru_rows_tracked_ok(x: torch.Tensor, w_hh: torch.Tensor) -> bool:
    """The whole-table GRU kernel takes this layer as a tracked step (`gru_rows_tracked`): the
    conditions of `gru_rows_ok` without the one that nothing be tracked."""
    return (USE_FUSED_CELLS and USE_GRU_ROWS and GRU_PACKED_BACKWARD and CELL_GEMM == "f16x3" and x.is_cuda
            and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
            and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0
            and bool(nv.lib().spmi_gru_rows_supported(x.shape[1], w_hh.shape[1])))


class GruRowsTracked(torch.autograd.Function):
    """One GRU layer step over every row of the state matrix `h_in` [M, H] on the kernel that
    keeps the gate weights in LDS, with the gate values kept for the backward pass.  The new
    state is the returned tensor (storage of its own); `h_in` is read, never written, and is
    what the backward pass reads the old state from."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, h_in, x_bounded):
        m, hd = x.shape[0], w_hh.shape[1]
        xd = x.detach()
        _check_cell_args(xd, h_in, None)
        if h_in.shape != (m, hd):
            raise nv.SpmiError("gru_rows_tracked: h_in must be the [M, H] state of the rows of x")
        need_grad = _needs_grad(ctx, 5)
        saved = torch.empty(m, 4, hd, dtype=torch.float32, device=x.device) if need_grad else None
        h_out = torch.empty(m, hd, dtype=torch.float32, device=x.device)
        ws_ih = weight_planes(w_ih, "f16x3", False, F16X3_CELL_LIMIT)
        ws_hh = weight_planes(w_hh, "f16x3", False, F16X3_CELL_LIMIT)
        with torch.cuda.device(x.device):
            if saved is None:
                nv.check(nv.lib().spmi_gru_rows_fwd(
                    nv.ptr(xd), xd.stride(0), xd.shape[1], nv.ptr(ws_ih), nv.ptr(ws_hh), nv.ptr(b_ih),
                    nv.ptr(b_hh), nv.ptr(h_in), m, hd, nv.ptr(h_out), int(x_bounded), nv.stream()),
                    "spmi_gru_rows_fwd")
            else:
                nv.check(nv.lib().spmi_gru_rows_fwd_saved(
                    nv.ptr(xd), xd.stride(0), xd.shape[1], nv.ptr(ws_ih), nv.ptr(ws_hh), nv.ptr(b_ih),
                    nv.ptr(b_hh), nv.ptr(h_in), m, hd, nv.ptr(h_out), nv.ptr(saved), int(x_bounded),
                    nv.stream()), "spmi_gru_rows_fwd_saved")
        ctx.save_for_backward(xd, w_ih, w_hh, saved, h_in)
        ctx.has_bias = b_ih is not None
        return h_out

    @staticmethod
    def backward(ctx, g_h):
        xd, w_ih, w_hh, saved, h_in = ctx.saved_tensors
        m, hd = g_h.shape
        g_h = g_h.contiguous()
        g4 = torch.empty(m, 4 * hd, dtype=torch.float32, device=g_h.device)
        with torch.cuda.device(g_h.device):
            nv.check(nv.lib().spmi_gru_gate_bwd_packed_q(
                nv.ptr(g_h), nv.ptr(saved), nv.ptr(h_in), h_in.stride(0), m, hd, nv.ptr(g4), nv.stream()),
                "spmi_gru_gate_bwd_packed_q")
        return _gru_grads_from_packed(ctx, g4, xd, h_in, w_ih) + (None, None)


def gru_rows_tracked(x, w_ih, w_hh, b_ih, b_hh, h_in: torch.Tensor, x_bounded: bool = False) -> torch.Tensor:
    """One GRU layer step for every row under autograd: the state matrix `h_in` [M, H] is read
    (and kept for the backward pass: it must not be written before that pass has run), the new
    state comes back as a tensor of its own."""
    return _apply_tracked(GruRowsTracked, (x, w_ih, w_hh, b_ih, b_hh), x, w_ih, w_hh, b_ih, b_hh, h_in,
                          bool(x_bounded))


def _whole_table_tracked_ok(supported: str, x: torch.Tensor, w_hh: torch.Tensor) -> bool:
    """The conditions of `_whole_table_ok` without the one that nothing be tracked."""
    return (USE_FUSED_CELLS and USE_GRU_ROWS and CELL_GEMM == "f16x3" and x.is_cuda
            and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
            and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0
            and bool(getattr(nv.lib(), supported)(x.shape[1], w_hh.shape[1])))


def _cell_grads_from_gates(ctx, g_g, xd, h_old, w_ih):
    """(g_x, g_w_ih, g_w_hh, g_b_ih, g_b_hh) of an LSTM or tanh-RNN layer from the gradient of
    its gate pre-activations g_g [M, G] (the same matrix for the input and the state side), the
    layer's input rows and its old state.  Both weight gradients and the bias gradient come
    from one launch where the shapes allow it."""
    g_x = None
    if ctx.needs_input_grad[0]:
        wt = w_ih.detach()                                      # [G, I]
        g_x = gemm_rows(g_g, wt.t()) if _gemm_rows_ok(g_g, wt.shape[0], wt.shape[1]) else g_g @ wt
    need_b = ctx.has_bias and (ctx.needs_input_grad[3] or ctx.needs_input_grad[4])
    if (GRU_PAIRED_WEIGHT_GRADS and ctx.needs_input_grad[1] and ctx.needs_input_grad[2]
            and gemm_tn_pair_ok(g_g, xd, g_g, h_old)):
        g_wih, g_whh, g_b, _ = gemm_tn_pair(g_g, xd, g_g, h_old, sums=True)
        if not need_b:
            g_b = None
    else:
        g_wih, g_b = _weight_and_bias_grad(g_g, xd, ctx.needs_input_grad[1], need_b, dy_first=True)
        g_whh = gemm_tn(g_g, h_old) if ctx.needs_input_grad[2] else None
    return g_x, g_wih, g_whh, g_b, g_b


def lstm_rows_tracked_ok(x: torch.Tensor, w_hh: torch.Tensor) -> bool:
    """The whole-table LSTM kernel takes this layer as a tracked step (`lstm_rows_tracked`)."""
    return _whole_table_tracked_ok("spmi_lstm_rows_supported", x, w_hh)


class LstmRowsTracked(torch.autograd.Function):
    """One LSTM layer step over every row of the state matrices `h_in`, `c_in` [M, H] on the
    kernel that keeps the gate weights in LDS, with the gate values kept for the backward pass.
    The new states are the returned tensors (storage of their own); h_in and c_in are read,
    never written.  The cell state carries no gradient to the caller (the node memory is a
    buffer between steps, as in LstmCell)."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, h_in, c_in, x_bounded):
        m, hd = x.shape[0], w_hh.shape[1]
        xd = x.detach()
        _check_cell_args(xd, h_in, None)
        nv.require_cuda(c_in)
        for t in (h_in, c_in):
            if t.shape != (m, hd) or not t.is_contiguous() or t.dtype != torch.float32 or t.device != x.device:
                raise nv.SpmiError("lstm_rows_tracked: h_in and c_in must be the contiguous float32 [M, H] "
                                   "states of the rows of x")
        need_grad = _needs_grad(ctx, 5)
        saved = torch.empty(m, 6 * hd, dtype=torch.float32, device=x.device) if need_grad else None
        h_out = torch.empty(m, hd, dtype=torch.float32, device=x.device)
        c_out = torch.empty(m, hd, dtype=torch.float32, device=x.device)
        ws_ih = weight_planes(w_ih, "f16x3", False, F16X3_CELL_LIMIT)
        ws_hh = weight_planes(w_hh, "f16x3", False, F16X3_CELL_LIMIT)
        with torch.cuda.device(x.device):
            if saved is None:
                nv.check(nv.lib().spmi_lstm_rows_fwd(
                    nv.ptr(xd), xd.stride(0), xd.shape[1], nv.ptr(ws_ih), nv.ptr(ws_hh), nv.ptr(b_ih),
                    nv.ptr(b_hh), nv.ptr(h_in), nv.ptr(c_in), m, hd, nv.ptr(h_out), nv.ptr(c_out),
                    int(x_bounded), nv.stream()), "spmi_lstm_rows_fwd")
            else:
                nv.check(nv.lib().spmi_lstm_rows_fwd_saved(
                    nv.ptr(xd), xd.stride(0), xd.shape[1], nv.ptr(ws_ih), nv.ptr(ws_hh), nv.ptr(b_ih),
                    nv.ptr(b_hh), nv.ptr(h_in), nv.ptr(c_in), m, hd, nv.ptr(h_out), nv.ptr(c_out),
                    nv.ptr(saved), int(x_bounded), nv.stream()), "spmi_lstm_rows_fwd_saved")
        ctx.save_for_backward(xd, w_ih, w_hh, saved, h_in)
        ctx.has_bias = b_ih is not None
        ctx.mark_non_differentiable(c_out)
        return h_out, c_out

    @staticmethod
    def backward(ctx, g_h, _g_c):
        xd, w_ih, w_hh, saved, h_in = ctx.saved_tensors
        m, hd = g_h.shape
        g_h = g_h.contiguous()
        g_g = torch.empty(m, 4 * hd, dtype=torch.float32, device=g_h.device)
        with torch.cuda.device(g_h.device):
            nv.check(nv.lib().spmi_lstm_gate_bwd(nv.ptr(g_h), None, nv.ptr(saved), m, hd,
                                                 nv.ptr(g_g), None, nv.stream()),
                     "spmi_lstm_gate_bwd")
        return _cell_grads_from_gates(ctx, g_g, xd, h_in, w_ih) + (None, None, None)


def lstm_rows_tracked(x, w_ih, w_hh, b_ih, b_hh, h_in: torch.Tensor, c_in: torch.Tensor,
                      x_bounded: bool = False):
    """One LSTM layer step for every row under autograd: the state matrices `h_in`, `c_in`
    [M, H] are read (h_in is kept for the backward pass: it must not be written before that
    pass has run), the new states (h', c') come back as tensors of their own."""
    return _apply_tracked(LstmRowsTracked, (x, w_ih, w_hh, b_ih, b_hh), x, w_ih, w_hh, b_ih, b_hh, h_in,
                          c_in, bool(x_bounded))


def rnn_rows_tracked_ok(x: torch.Tensor, w_hh: torch.Tensor) -> bool:
    """The whole-table tanh-RNN kernel takes this layer as a tracked step (`rnn_rows_tracked`)."""
    return _whole_table_tracked_ok("spmi_rnn_rows_supported", x, w_hh)


class RnnRowsTracked(torch.autograd.Function):
    """One tanh-RNN layer step over every row of the state matrix `h_in` [M, H] on the kernel
    that keeps the weights in LDS.  The backward pass needs the output and the old state only,
    so the forward is the inference kernel as it is."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, h_in, x_bounded):
        m, hd = x.shape[0], w_hh.shape[1]
        xd = x.detach()
        h_out = torch.empty(m, hd, dtype=torch.float32, device=x.device)
        rnn_rows(xd, w_ih, w_hh, b_ih, b_hh, h_in, h_out, x_bounded=x_bounded)
        ctx.save_for_backward(xd, w_ih, w_hh, h_out, h_in)
        ctx.has_bias = b_ih is not None
        return h_out

    @staticmethod
    def backward(ctx, g_h):
        xd, w_ih, w_hh, h_out, h_in = ctx.saved_tensors
        m, hd = g_h.shape
        g_h = g_h.contiguous()
        g_g = torch.empty(m, hd, dtype=torch.float32, device=g_h.device)
        with torch.cuda.device(g_h.device):
            nv.check(nv.lib().spmi_rnn_gate_bwd(nv.ptr(g_h), nv.ptr(h_out), m, hd, nv.ptr(g_g),
                                                nv.stream()), "spmi_rnn_gate_bwd")
        return _cell_grads_from_gates(ctx, g_g, xd, h_in, w_ih) + (None, None)


def rnn_rows_tracked(x, w_ih, w_hh, b_ih, b_hh, h_in: torch.Tensor, x_bounded: bool = False) -> torch.Tensor:
    """One tanh-RNN layer step for every row under autograd: the state matrix `h_in` [M, H] is
    read (and kept for the backward pass), the new state comes back as a tensor of its own."""
    return _apply_tracked(RnnRowsTracked, (x, w_ih, w_hh, b_ih, b_hh), x, w_ih, w_hh, b_ih, b_hh, h_in,
                          bool(x_bounded))


class LstmCell(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, h_mem, c_mem, ids, mem_write):
        m, hd = x.shape[0], w_hh.shape[1]
        xd = x.detach()
        need_grad = _needs_grad(ctx, 5)
        fused = _fused_ok(xd, w_hh)
        h_old = _rows(h_mem, ids, m) if (need_grad or not fused) else None
        if need_grad and (mem_write or ids is not None):
            h_old = h_old.clone()  # the backward needs h_old after the memory rows are overwritten
        saved = torch.empty(m, 6 * hd, dtype=torch.float32, device=x.device) if need_grad else None
        h_out = torch.empty(m, hd, dtype=torch.float32, device=x.device)
        c_out = torch.empty(m, hd, dtype=torch.float32, device=x.device)
        if fused:
            _fused_cell("lstm", xd, w_ih, w_hh, b_ih, b_hh, h_mem, c_mem, ids, mem_write, h_out,
                        c_out, saved)
        else:
            gi = xd @ w_ih.detach().t()
            gh = h_old @ w_hh.detach().t()
            with torch.cuda.device(x.device):
                nv.check(nv.lib().spmi_lstm_gate_fwd(
                    nv.ptr(gi), nv.ptr(gh), nv.ptr(b_ih), nv.ptr(b_hh), nv.ptr(h_mem), nv.ptr(c_mem),
                    nv.ptr(ids), m, hd, nv.ptr(h_out), nv.ptr(c_out), int(mem_write), nv.ptr(saved),
                    nv.stream()), "spmi_lstm_gate_fwd")
        ctx.save_for_backward(xd, w_ih, w_hh, saved, h_old if need_grad else None)
        ctx.has_bias = b_ih is not None
        ctx.mark_non_differentiable(c_out)
        return h_out, c_out

    @staticmethod
    def backward(ctx, g_h, _g_c):
        xd, w_ih, w_hh, saved, h_old = ctx.saved_tensors
        m, hd = g_h.shape
        g_h = g_h.contiguous()
        g_g = torch.empty(m, 4 * hd, dtype=torch.float32, device=g_h.device)
        with torch.cuda.device(g_h.device):
            nv.check(nv.lib().spmi_lstm_gate_bwd(nv.ptr(g_h), None, nv.ptr(saved), m, hd,
                                                 nv.ptr(g_g), None, nv.stream()),
                     "spmi_lstm_gate_bwd")
        g_x = g_g @ w_ih if ctx.needs_input_grad[0] else None
        need_b = ctx.has_bias and (ctx.needs_input_grad[3] or ctx.needs_input_grad[4])
        g_wih, g_b = _weight_and_bias_grad(g_g, xd, ctx.needs_input_grad[1], need_b, dy_first=True)
        g_whh = gemm_tn(g_g, h_old) if ctx.needs_input_grad[2] else None
        return g_x, g_wih, g_whh, g_b, g_b, None, None, None, None


class RnnCell(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, h_mem, ids, mem_write):
        m, hd = x.shape[0], w_hh.shape[1]
        xd = x.detach()
        need_grad = _needs_grad(ctx, 5)
        fused = _fused_ok(xd, w_hh)
        h_old = _rows(h_mem, ids, m) if (need_grad or not fused) else None
        if need_grad and (mem_write or ids is not None):
            h_old = h_old.clone()
        h_out = torch.empty(m, hd, dtype=torch.float32, device=x.device)
        if fused:
            _fused_cell("rnn", xd, w_ih, w_hh, b_ih, b_hh, h_mem, None, ids, mem_write, h_out, None,
                        None)
        else:
            gi = xd @ w_ih.detach().t()
            gh = h_old @ w_hh.detach().t()
            with torch.cuda.device(x.device):
                nv.check(nv.lib().spmi_rnn_gate_fwd(
                    nv.ptr(gi), nv.ptr(gh), nv.ptr(b_ih), nv.ptr(b_hh), nv.ptr(h_mem), nv.ptr(ids), m,
                    hd, nv.ptr(h_out), int(mem_write), nv.stream()), "spmi_rnn_gate_fwd")
        ctx.save_for_backward(xd, w_ih, w_hh, h_out, h_old if need_grad else None)
        ctx.has_bias = b_ih is not None
        return h_out

    @staticmethod
    def backward(ctx, g_h):
        xd, w_ih, w_hh, h_out, h_old = ctx.saved_tensors
        m, hd = g_h.shape
        g_h = g_h.contiguous()
        g_g = torch.empty(m, hd, dtype=torch.float32, device=g_h.device)
        with torch.cuda.device(g_h.device):
            nv.check(nv.lib().spmi_rnn_gate_bwd(nv.ptr(g_h), nv.ptr(h_out), m, hd, nv.ptr(g_g),
                                                nv.stream()), "spmi_rnn_gate_bwd")
        g_x = g_g @ w_ih if ctx.needs_input_grad[0] else None
        need_b = ctx.has_bias and (ctx.needs_input_grad[3] or ctx.needs_input_grad[4])
        g_wih, g_b = _weight_and_bias_grad(g_g, xd, ctx.needs_input_grad[1], need_b, dy_first=True)
        g_whh = gemm_tn(g_g, h_old) if ctx.needs_input_grad[2] else None
        return g_x, g_wih, g_whh, g_b, g_b, None, None, None


class GruCellDense(torch.autograd.Function):
    """GRU step with the previous state as an ordinary differentiable tensor (no node
    memory): used where the recurrent state is itself a trained quantity, e.g. a weight
    matrix that evolves over snapshots."""

    @staticmethod
    def forward(ctx, x, h, w_ih, w_hh, b_ih, b_hh):
        m, hd = x.shape[0], w_hh.shape[1]
        xd, hdn = x.detach().contiguous(), h.detach().contiguous()
        saved = torch.empty(m, 5 * hd, dtype=torch.float32, device=x.device)
        h_out = torch.empty(m, hd, dtype=torch.float32, device=x.device)
        if _fused_ok(xd, w_hh):
            _fused_cell("gru", xd, w_ih, w_hh, b_ih, b_hh, hdn, None, None, False, h_out, None, saved)
        else:
            gi = xd @ w_ih.detach().t()
            gh = hdn @ w_hh.detach().t()
            with torch.cuda.device(x.device):
                nv.check(nv.lib().spmi_gru_gate_fwd(
                    nv.ptr(gi), nv.ptr(gh), nv.ptr(b_ih), nv.ptr(b_hh), nv.ptr(hdn), None, m, hd,
                    nv.ptr(h_out), 0, nv.ptr(saved), nv.stream()), "spmi_gru_gate_fwd")
        ctx.save_for_backward(xd, w_ih, w_hh, saved)
        ctx.has_bias = b_ih is not None
        return h_out

    @staticmethod
    def backward(ctx, g_out):
        xd, w_ih, w_hh, saved = ctx.saved_tensors
        m, hd = g_out.shape
        g_out = g_out.contiguous()
        g_gi = torch.empty(m, 3 * hd, dtype=torch.float32, device=g_out.device)
        g_gh = torch.empty(m, 3 * hd, dtype=torch.float32, device=g_out.device)
        g_hd = torch.empty(m, hd, dtype=torch.float32, device=g_out.device)
        with torch.cuda.device(g_out.device):
            nv.check(nv.lib().spmi_gru_gate_bwd(nv.ptr(g_out), nv.ptr(saved), m, hd, nv.ptr(g_gi),
                                                nv.ptr(g_gh), nv.ptr(g_hd), nv.stream()),
                     "spmi_gru_gate_bwd")
        h_old = saved[:, 4 * hd:]
        g_x = g_gi @ w_ih if ctx.needs_input_grad[0] else None
        g_h = g_hd + g_gh @ w_hh if ctx.needs_input_grad[1] else None
        g_wih, g_bih = _weight_and_bias_grad(g_gi, xd, ctx.needs_input_grad[2],
                                             ctx.has_bias and ctx.needs_input_grad[4], dy_first=True)
        g_whh, g_bhh = _weight_and_bias_grad(g_gh, h_old, ctx.needs_input_grad[3],
                                             ctx.has_bias and ctx.needs_input_grad[5], dy_first=True)
        return g_x, g_h, g_wih, g_whh, g_bih, g_bhh


def gru_cell_dense(x, h, w_ih, w_hh, b_ih=None, b_hh=None):
    """h' = GRU(x, h) for dense [M, I] / [M, H] float32 CUDA tensors, differentiable in both."""
    nv.require_cuda(x, h)
    if x.dtype != torch.float32 or h.dtype != torch.float32:
        raise nv.SpmiError("recurrent cells compute in float32")
    return GruCellDense.apply(x, h, w_ih, w_hh, b_ih, b_hh)


def gru_cell(x, w_ih, w_hh, b_ih, b_hh, h_mem, ids=None, mem_write=True, x_bounded=False):
    """One GRU step for a batch of nodes.  h_mem is one layer of the node memory
    ([n_nodes, H], read at rows `ids`, overwritten there with the detached new state when
    mem_write).  Returns the new hidden state [M, H] (differentiable w.r.t. x and weights;
    the memory is a constant, as in the reference where it never carries a graph).
    `x_bounded`: the caller knows |x| < 2^15 (e.g. x is another cell's output), which spares the
    default kernel its pass over the magnitudes of x; same for lstm_cell and rnn_cell."""
    _check_cell_args(x, h_mem, ids)
    return _apply_tracked(GruCell, (x, w_ih, w_hh, b_ih, b_hh), x.contiguous(), w_ih, w_hh, b_ih, b_hh, h_mem, ids,
                          mem_write, x_bounded=x_bounded)


def lstm_cell(x, w_ih, w_hh, b_ih, b_hh, h_mem, c_mem, ids=None, mem_write=True, x_bounded=False):
    _check_cell_args(x, h_mem, ids)
    _check_cell_args(x, c_mem, ids)
    return _apply_tracked(LstmCell, (x, w_ih, w_hh, b_ih, b_hh), x.contiguous(), w_ih, w_hh, b_ih, b_hh, h_mem,
                          c_mem, ids, mem_write, x_bounded=x_bounded)


def rnn_cell(x, w_ih, w_hh, b_ih, b_hh, h_mem, ids=None, mem_write=True, x_bounded=False):
    _check_cell_args(x, h_mem, ids)
    return _apply_tracked(RnnCell, (x, w_ih, w_hh, b_ih, b_hh), x.contiguous(), w_ih, w_hh, b_ih, b_hh, h_mem, ids,
                          mem_write, x_bounded=x_bounded)
