"""Device-resident snapshot graph: the object that stands where the reference holds a
`dgl.DGLGraph` (`Snapshot.observation`, spiro/dtdg/types.py:57-59,191).

It keeps the COO edge list the reference builds (`dgl.graph((src, dst), num_nodes)`) and
materialises, lazily and once per snapshot, what DGL materialises inside libdgl:

* the by-destination CSR (forward aggregation) and the by-source CSR (backward), each a
  stable grouping with `eid` = original edge position;
* in/out degrees (row-pointer differences, no atomics) and the GraphConv normalisers;
* the load-balance plan of the aggregation kernel (hub rows cut into chunks).

Only the DGLGraph methods the reference's torch path touches are provided
(spiro/dtdg/types.py:67-106,191,241,247; tests/dtdg/test_types_torch.py:49-50).
All compute runs in libspmi on the GPU; a graph held on the CPU is a plain container
and raises when asked to compute.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from . import _native as nv

HEAVY_THRESH = 128  # rows with more nonzeros than this are summed chunk-wise by whole workgroups (swept: 0.78 ms at 128 against 0.83 ms at 512 for 1M x 10M x 128)
HEAVY_CHUNK = 256   # nonzeros per chunk (one workgroup, four waves of 64)

_ACT_CODES = {None: 0, "none": 0, "relu": 1, "leaky_relu": 2}
_DTYPE_CODES = {torch.float32: 0, torch.bfloat16: 1}

# Optional launch timing of the aggregation (bench.py): when a list is installed, every
# call of `aggregate` appends (start_event, end_event, nnz, n_rows, F) recorded on the
# stream the kernels are launched on.
_launch_log: Optional[list] = None


def set_launch_log(log: Optional[list]) -> None:
    global _launch_log
    _launch_log = log


def _i64(t: torch.Tensor) -> torch.Tensor:
    return t if t.dtype == torch.int64 else t.to(torch.int64)


class CSR:
    """One grouping of a snapshot's edges on the device (int32, see include/spmi.h).

    The status word of the build and the two sizes of the hub-row plan come back from the
    device through a pinned buffer.  A CSR made with `defer=True` does not wait for them:
    `ready()` (called by `n_heavy` / `n_chunks`, hence by every aggregation) does, and raises
    there if the edge list held an id out of range.  Until then the host keeps queueing work
    behind the build instead of idling on it."""

    def __init__(self, indptr: torch.Tensor, indices: torch.Tensor, eid: Optional[torch.Tensor],
                 n_rows: int, n_cols: int, nnz: int, heavy_rows: Optional[torch.Tensor] = None,
                 heavy_chunk_ptr: Optional[torch.Tensor] = None, n_heavy: int = 0, n_chunks: int = 0,
                 heavy_thresh: int = HEAVY_THRESH, chunk: int = HEAVY_CHUNK, pending=None):
        self.indptr, self.indices, self.eid = indptr, indices, eid  # eid: None unless asked for
        self.n_rows, self.n_cols, self.nnz = n_rows, n_cols, nnz
        self.heavy_rows, self.heavy_chunk_ptr = heavy_rows, heavy_chunk_ptr
        self._n_heavy, self._n_chunks = n_heavy, n_chunks
        self.heavy_thresh, self.chunk = heavy_thresh, chunk
        self._bad_ids = False
        self._pending = pending  # (pinned int32[3], event, device int32[3]) until ready()
        self._partial: Dict[int, torch.Tensor] = {}
        self._norms: Dict[int, torch.Tensor] = {}  # normaliser of the rows, per mode
        self._other_norms: Dict[int, torch.Tensor] = {}  # normaliser of the other side, where the build made it

    def ready(self) -> "CSR":
        if self._pending is not None:
            host, event, _ = self._pending
            event.synchronize()
            status, self._n_heavy, self._n_chunks = host.tolist()
            self._pending = None
            self._bad_ids = status != 0
        if self._bad_ids:  # every later use reports it again
            raise nv.SpmiError(f"edge list holds a node id outside [0, {self.n_rows}) / "
                               f"[0, {self.n_cols})")
        return self

    @property
    def n_heavy(self) -> int:
        return self.ready()._n_heavy

    @property
    def n_chunks(self) -> int:
        return self.ready()._n_chunks

    def partial_ws(self, feat: int) -> Optional[torch.Tensor]:
        if self.n_heavy == 0:
            return None
        ws = self._partial.get(feat)
        if ws is None:
            ws = torch.empty(self.n_chunks * feat, dtype=torch.float32, device=self.indptr.device)
            self._partial = {feat: ws}  # keep one size: layers of a model share F
        return ws

    def degrees(self) -> torch.Tensor:
        """int64 row degrees, like DGLGraph.in_degrees()."""
        return (self.indptr[1:] - self.indptr[:-1]).to(torch.int64)

    def other_norm(self, mode: int) -> Optional[torch.Tensor]:
        return self._other_norms.get(mode)

    def norm(self, mode: int) -> torch.Tensor:
        out = self._norms.get(mode)
        if out is None:
            out = torch.empty(self.n_rows, dtype=torch.float32, device=self.indptr.device)
            nv.check(nv.lib().spmi_norm_from_indptr(nv.ptr(self.indptr), self.n_rows, mode,
                                                    nv.ptr(out), nv.stream()), "spmi_norm_from_indptr")
            self._norms[mode] = out
        return out


def _plan(indptr: torch.Tensor, n_rows: int, e: int, meta: torch.Tensor, heavy_thresh: int,
          chunk: int):
    """Queue the hub-row plan of the aggregation; meta[1:3] receives its two sizes."""
    cap = e // heavy_thresh + 2
    heavy_rows = torch.empty(cap, dtype=torch.int32, device=indptr.device)
    heavy_ptr = torch.empty(cap, dtype=torch.int32, device=indptr.device)
    nv.check(nv.lib().spmi_spmm_plan(nv.ptr(indptr), n_rows, heavy_thresh, chunk,
                                     nv.ptr(heavy_rows), nv.ptr(heavy_ptr), cap, nv.ptr(meta[1:]),
                                     nv.stream()), "spmi_spmm_plan")
    return heavy_rows, heavy_ptr


def _zeroed_meta(device: torch.device) -> torch.Tensor:
    """int32[3] (status, n_heavy, n_chunks) of one build, zeroed on the current stream without a
    fill of its own (`_native.zeroed_slot`)."""
    return nv.zeroed_slot(device, 4)[:3]


def _fetch_meta(meta: torch.Tensor):
    """Queue the copy of a build's (status, n_heavy, n_chunks) to pinned host memory and an
    event behind it; the device tensor rides along so its memory outlives the copy."""
    host = torch.empty(3, dtype=torch.int32, pin_memory=True)
    host.copy_(meta, non_blocking=True)
    event = torch.cuda.Event()
    event.record()
    return host, event, meta


def build_csr(key: torch.Tensor, other: torch.Tensor, n_key: int, n_other: int,
              heavy_thresh: int = HEAVY_THRESH, chunk: int = HEAVY_CHUNK,
              want_eid: bool = False, defer: bool = False, norm_mode: Optional[int] = None,
              other_norm: bool = False) -> CSR:
    """COO -> CSR grouped by `key` plus the aggregation plan, one library call.  The status word
    and the plan sizes cost one host sync: here, or with `defer` at the CSR's first `ready()`.
    Edge ids travel through the sort only when `want_eid` (edge weights / edge features need
    them; plain GCN does not).  `norm_mode` (0 or 1): the normaliser of the key side is written
    by the same kernels that write the row pointers and kept on the CSR (`CSR.norm(mode)`).
    `other_norm`: also the normaliser of the OTHER side in that mode (out-degree based, beside a
    by-destination CSR), where the library can make it in the same pass over the id arrays
    (`CSR.other_norm(mode)`, None where it cannot: the caller counts the id column instead)."""
    nv.require_cuda(key, other)
    key, other = _i64(key).contiguous(), _i64(other).contiguous()
    dev = key.device
    e = key.numel()
    lib = nv.lib()
    with torch.cuda.device(dev):
        indptr = torch.empty(n_key + 1, dtype=torch.int32, device=dev)
        indices = torch.empty(e, dtype=torch.int32, device=dev)
        eid = torch.empty(e, dtype=torch.int32, device=dev) if want_eid else None
        norm = torch.empty(n_key, dtype=torch.float32, device=dev) if norm_mode is not None else None
        onorm = None
        if other_norm and norm_mode is not None and lib.spmi_snapshot_csr_norm_other_supported(e, n_key, n_other):
            onorm = torch.empty(n_other, dtype=torch.float32, device=dev)
        cap = e // heavy_thresh + 2
        heavy_rows = torch.empty(cap, dtype=torch.int32, device=dev)
        heavy_ptr = torch.empty(cap, dtype=torch.int32, device=dev)
        meta = _zeroed_meta(dev)  # status, n_heavy, n_chunks
        ws_bytes = lib.spmi_snapshot_csr_workspace_bytes(e, n_key)
        ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
        nv.check(lib.spmi_snapshot_csr(nv.ptr(key), nv.ptr(other), e, n_key, n_other, nv.ptr(indptr),
                                       nv.ptr(indices), nv.ptr(eid), nv.ptr(norm), nv.ptr(onorm),
                                       0 if norm_mode is None else int(norm_mode), heavy_thresh, chunk,
                                       nv.ptr(heavy_rows), nv.ptr(heavy_ptr), cap, nv.ptr(ws), ws_bytes,
                                       nv.ptr(meta), nv.stream()), "spmi_snapshot_csr")
        pending = _fetch_meta(meta)
    csr = CSR(indptr, indices, eid, n_key, n_other, e, heavy_rows, heavy_ptr, 0, 0, heavy_thresh,
              chunk, pending)
    if norm is not None:
        csr._norms[int(norm_mode)] = norm
    if onorm is not None:
        csr._other_norms[int(norm_mode)] = onorm
    return csr if defer else csr.ready()


class CumulativeCSR:
    """One grouping of a whole time-sorted edge table, from which the CSR of any time
    prefix `edges[:e_cut]` is cut without sorting again (the reference rebuilds the graph
    of every prefix from scratch, spiro/dtdg/types.py:183-198).  The grouping is stable, so
    the edges of a prefix are a prefix of every row."""

    def __init__(self, key: torch.Tensor, other: torch.Tensor, n_key: int, n_other: int):
        self.full = build_csr(key, other, n_key, n_other, want_eid=True)

    def _check(self, e_cut: int, n_rows: int) -> None:
        if not (0 <= e_cut <= self.full.nnz and 1 <= n_rows <= self.full.n_rows):
            raise nv.SpmiError(f"prefix (e_cut={e_cut}, n_rows={n_rows}) outside the table "
                               f"({self.full.nnz} edges, {self.full.n_rows} rows)")

    def prefix(self, e_cut: int, n_rows: int, n_cols: int, want_eid: bool = False,
               heavy_thresh: int = HEAVY_THRESH, chunk: int = HEAVY_CHUNK,
               defer: bool = False) -> CSR:
        """CSR of the first `e_cut` edges over rows [0, n_rows): bit-identical to
        `build_csr(key[:e_cut], other[:e_cut], n_rows, n_cols)`."""
        self._check(e_cut, n_rows)
        full, lib = self.full, nv.lib()
        dev = full.indptr.device
        with torch.cuda.device(dev):
            indptr = torch.empty(n_rows + 1, dtype=torch.int32, device=dev)
            indices = torch.empty(e_cut, dtype=torch.int32, device=dev)
            eid = torch.empty(e_cut, dtype=torch.int32, device=dev) if want_eid else None
            meta = _zeroed_meta(dev)
            ws_bytes = lib.spmi_csr_prefix_workspace_bytes(n_rows)
            ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
            nv.check(lib.spmi_csr_prefix(nv.ptr(full.indptr), nv.ptr(full.indices), nv.ptr(full.eid),
                                         n_rows, n_cols, e_cut, nv.ptr(indptr), nv.ptr(indices),
                                         nv.ptr(eid), nv.ptr(ws), ws_bytes, nv.ptr(meta),
                                         nv.stream()), "spmi_csr_prefix")
            heavy_rows, heavy_ptr = _plan(indptr, n_rows, e_cut, meta, heavy_thresh, chunk)
            pending = _fetch_meta(meta)
        csr = CSR(indptr, indices, eid, n_rows, n_cols, e_cut, heavy_rows, heavy_ptr, 0, 0,
                  heavy_thresh, chunk, pending)
        return csr if defer else csr.ready()

    def prefix_degree(self, e_cut: int, n_rows: int) -> torch.Tensor:
        """int64 degrees of rows [0, n_rows) in the first `e_cut` edges."""
        self._check(e_cut, n_rows)
        full = self.full
        with torch.cuda.device(full.indptr.device):
            out = torch.empty(n_rows, dtype=torch.int64, device=full.indptr.device)
            nv.check(nv.lib().spmi_csr_prefix_degree(nv.ptr(full.indptr), nv.ptr(full.eid), n_rows,
                                                     e_cut, nv.ptr(out), nv.stream()),
                     "spmi_csr_prefix_degree")
        return out


def degree(ids: torch.Tensor, n: int, validated: bool = False, out: Optional[torch.Tensor] = None
           ) -> torch.Tensor:
    """deg[v] = occurrences of v in ids (int64): ids grouped into buckets of 4096 nodes and
    counted in LDS (n <= 2^24), keys-only sort + run boundaries beyond.  `validated` says the
    caller does not need the range check reported here (an id outside [0, n) is clamped, never
    dereferenced), so the call does not wait for the device.  `out`: a contiguous int64 [n]
    tensor to write into (a caller that counts every snapshot keeps one)."""
    nv.require_cuda(ids)
    ids = _i64(ids).contiguous()
    dev = ids.device
    lib = nv.lib()
    with torch.cuda.device(dev):
        if out is None:
            out = torch.empty(n, dtype=torch.int64, device=dev)
        elif out.dtype != torch.int64 or out.numel() != n or not out.is_contiguous():
            raise nv.SpmiError("degree(out=...) expects a contiguous int64 tensor of n elements")
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        ws_bytes = lib.spmi_degree_workspace_bytes(ids.numel(), n)
        ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
        nv.check(lib.spmi_degree(nv.ptr(ids), ids.numel(), n, nv.ptr(out), nv.ptr(ws), ws_bytes,
                                 nv.ptr(status), nv.stream()), "spmi_degree")
        if not validated and status.item() != 0:
            raise nv.SpmiError(f"node id outside [0, {n})")
    return out


def norm_from_ids32(ids: torch.Tensor, n: int, mode: int) -> torch.Tensor:
    """clamp(count of v in ids, 1)^-0.5 (mode 0) or its reciprocal (mode 1) for v in [0, n), from
    an int32 id column the library wrote itself (a CSR's `indices`)."""
    nv.require_cuda(ids)
    if ids.dtype != torch.int32 or not ids.is_contiguous():
        raise nv.SpmiError("norm_from_ids32 expects a contiguous int32 id column")
    lib = nv.lib()
    with torch.cuda.device(ids.device):
        out = torch.empty(n, dtype=torch.float32, device=ids.device)
        ws_bytes = lib.spmi_degree_workspace_bytes(ids.numel(), n)
        ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=ids.device)
        nv.check(lib.spmi_norm_from_ids32(nv.ptr(ids), ids.numel(), n, mode, nv.ptr(out), None, nv.ptr(ws),
                                          ws_bytes, nv.stream()), "spmi_norm_from_ids32")
    return out


def norm_from_degree(deg: torch.Tensor, mode: int) -> torch.Tensor:
    """clamp(deg, 1)^-0.5 (mode 0) or 1/clamp(deg, 1) (mode 1) as fp32, one launch."""
    nv.require_cuda(deg)
    deg = _i64(deg).contiguous()
    with torch.cuda.device(deg.device):
        out = torch.empty(deg.numel(), dtype=torch.float32, device=deg.device)
        nv.check(nv.lib().spmi_norm_from_degree(nv.ptr(deg), deg.numel(), mode, nv.ptr(out),
                                                nv.stream()), "spmi_norm_from_degree")
    return out


def aggregate(csr: CSR, x: torch.Tensor, src_scale: Optional[torch.Tensor] = None,
              edge_w: Optional[torch.Tensor] = None, dst_scale: Optional[torch.Tensor] = None,
              bias: Optional[torch.Tensor] = None, act: Optional[str] = None,
              slope: float = 0.01, out: Optional[torch.Tensor] = None,
              out_dtype: Optional[torch.dtype] = None, accumulate: bool = False,
              finalize: bool = True) -> torch.Tensor:
    """Y = act(dst_scale * (A (src_scale * X) with edge weights) + bias), one fused launch
    (plus two for hub rows).  X and Y are stored as float32 or bfloat16 (sums are float32;
    Y defaults to X's type); scales and bias are float32.  No autograd here; see
    ops.GcnAggregate.  `accumulate` adds the sums to what `out` holds and `finalize=False`
    leaves out scaling, bias and activation: a sum over several batches of sources is a chain
    of such calls (float32 `out`), the last one with `finalize=True`."""
    nv.require_cuda(x, src_scale, edge_w, dst_scale, bias, out)
    if x.dim() != 2 or x.dtype not in _DTYPE_CODES:
        raise nv.SpmiError(f"aggregate expects a 2-D float32/bfloat16 feature matrix, got "
                           f"{tuple(x.shape)} {x.dtype}")
    if x.shape[0] != csr.n_cols:
        raise nv.SpmiError(f"feature matrix has {x.shape[0]} rows, the graph addresses {csr.n_cols}")
    if x.stride(1) != 1:
        x = x.contiguous()
    feat = x.shape[1]
    for name, t, n in (("src_scale", src_scale, csr.n_cols), ("edge_w", edge_w, csr.nnz),
                       ("dst_scale", dst_scale, csr.n_rows), ("bias", bias, feat)):
        if t is not None and (t.dtype != torch.float32 or t.numel() != n or not t.is_contiguous()):
            raise nv.SpmiError(f"{name} must be contiguous float32 with {n} elements")
    if accumulate and out is None:
        raise nv.SpmiError("aggregate(accumulate=True) needs the `out` tensor of the earlier calls")
    if out is None:
        partial_sums = accumulate or not finalize
        out = torch.empty(csr.n_rows, feat, dtype=out_dtype or (torch.float32 if partial_sums else x.dtype),
                          device=x.device)
    elif out.shape != (csr.n_rows, feat) or out.dtype not in _DTYPE_CODES or out.stride(1) != 1:
        raise nv.SpmiError("bad `out` tensor")
    with torch.cuda.device(x.device):
        partial = csr.partial_ws(feat)
        if _launch_log is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        nv.check(nv.lib().spmi_gcn_spmm_part(
            nv.ptr(csr.indptr), nv.ptr(csr.indices), csr.n_rows, csr.n_cols, csr.nnz,
            nv.ptr(x), _DTYPE_CODES[x.dtype], x.stride(0), feat, nv.ptr(src_scale), nv.ptr(edge_w),
            nv.ptr(dst_scale), nv.ptr(bias), _ACT_CODES[act], float(slope), nv.ptr(out),
            _DTYPE_CODES[out.dtype], out.stride(0), nv.ptr(csr.heavy_rows),
            nv.ptr(csr.heavy_chunk_ptr), csr.n_heavy, csr.n_chunks, csr.heavy_thresh, csr.chunk,
            nv.ptr(partial), int(accumulate), int(finalize), nv.stream()), "spmi_gcn_spmm_part")
        if _launch_log is not None:
            ev1.record()
            _launch_log.append((ev0, ev1, csr.nnz, csr.n_rows, feat))
    return out


class NormalisedAdjacency:
    """What the aggregation needs from a graph object: csr() / csr_t() / norm(side, mode).
    On top of those this mixin caches the per-nonzero weights the fused kernel streams
    in CSR order, so that the source-side normaliser is a coalesced 4-byte stream per
    edge instead of a dependent random gather inside the kernel."""

    def _edge_scale_cache(self) -> dict:
        cache = getattr(self, "_edge_scales", None)
        if cache is None:
            cache = self._edge_scales = {}
        return cache

    def edge_scale(self, direction: str, norm: str, edge_weight: Optional[torch.Tensor] = None
                   ) -> Optional[torch.Tensor]:
        """Weights w_j in CSR order.  'fwd': by-destination CSR, w_j = norm_out[src_j] for
        norm in ('left','both'); 'bwd': by-source CSR, w_j = norm_in[dst_j] for norm in
        ('right','both').  norm 'sgc' (SGConv) scales both sides by in-degree^-1/2, so both
        directions use norm_in.  A user edge weight (edge-id order) is permuted and
        multiplied in."""
        mode = 0 if norm in ("both", "sgc") else 1
        if norm == "sgc":
            scaled, side = True, "in"
        elif direction == "fwd":
            scaled, side = norm in ("left", "both"), "out"
        else:
            scaled, side = norm in ("right", "both"), "in"
        key = (direction, norm)
        cache = self._edge_scale_cache()
        w = cache.get(key)
        if w is None and scaled:
            # wait for the build's verdict first: after a failed range check `indices` holds
            # unchecked or unwritten entries, which must not reach a gather
            csr = self.csr() if direction == "fwd" else self.csr_t()
            w = cache[key] = self.norm(side, mode).index_select(0, csr.indices)
        if edge_weight is not None:
            csr = self.csr(want_eid=True) if direction == "fwd" else self.csr_t(want_eid=True)
            p = edge_weight.detach().to(torch.float32).reshape(-1).index_select(0, csr.eid)
            w = p if w is None else w * p
        return w


class EdgeTableIndex:
    """Table-wide groupings (by destination and by source) of a time-sorted edge table,
    built on first use and shared by every prefix snapshot of the table."""

    def __init__(self, src: torch.Tensor, dst: torch.Tensor, n_nodes: int):
        self._src, self._dst, self._n = src, dst, int(n_nodes)
        self._by_dst: Optional[CumulativeCSR] = None
        self._by_src: Optional[CumulativeCSR] = None

    def by_dst(self) -> CumulativeCSR:
        if self._by_dst is None:
            self._by_dst = CumulativeCSR(self._dst, self._src, self._n, self._n)
        return self._by_dst

    def by_src(self) -> CumulativeCSR:
        if self._by_src is None:
            self._by_src = CumulativeCSR(self._src, self._dst, self._n, self._n)
        return self._by_src


class _FrameDict(dict):
    """ndata / edata: feature name -> tensor, checked against the number of rows."""

    def __init__(self, rows: int, what: str):
        super().__init__()
        self._rows, self._what = rows, what

    def __setitem__(self, key, value):
        if value.shape[0] != self._rows:
            raise ValueError(f"Expect number of features to match number of {self._what} "
                             f"(len(u)). Got {value.shape[0]} and {self._rows} instead.")
        super().__setitem__(key, value)


class SnapshotGraph(NormalisedAdjacency):
    """Stand-in for the `dgl.DGLGraph` of one snapshot (homogeneous, directed, multi-edges
    and self-loops kept, edge ids = positions in the COO list)."""

    def __init__(self, src: torch.Tensor, dst: torch.Tensor, num_nodes: Optional[int] = None,
                 table: Optional["EdgeTableIndex"] = None):
        """`table`: set when (src, dst) are the first len(src) rows of a time-sorted edge
        table that keeps table-wide groupings; the CSRs are then cut from those."""
        src, dst = torch.as_tensor(src), torch.as_tensor(dst)
        if src.shape != dst.shape or src.dim() != 1:
            raise ValueError("src and dst must be 1-D tensors of equal length")
        if src.device != dst.device:
            raise ValueError("src and dst must live on the same device")
        self._src, self._dst = _i64(src), _i64(dst)
        if num_nodes is None:
            num_nodes = int(max(src.max().item(), dst.max().item())) + 1 if src.numel() else 0
        self._n = int(num_nodes)
        self.ndata = _FrameDict(self._n, "nodes")
        self.edata = _FrameDict(src.numel(), "edges")
        self._csr: Optional[CSR] = None      # grouped by destination
        self._csr_t: Optional[CSR] = None    # grouped by source
        self._out_deg: Optional[torch.Tensor] = None
        self._norms: Dict[Tuple[str, int], torch.Tensor] = {}
        self._table = table

    # ---- DGLGraph surface used by the reference -----------------------------------
    @property
    def device(self) -> torch.device:
        return self._src.device

    def num_nodes(self) -> int:
        return self._n

    def number_of_nodes(self) -> int:
        return self._n

    def num_edges(self) -> int:
        return self._src.numel()

    def number_of_edges(self) -> int:
        return self._src.numel()

    def edges(self):
        return self._src, self._dst

    def to(self, device, **kwargs) -> "SnapshotGraph":
        device = torch.device(device)
        if device == self.device:
            return self
        g = SnapshotGraph(self._src.to(device, **kwargs), self._dst.to(device, **kwargs), self._n)
        for k, v in self.ndata.items():
            g.ndata[k] = v.to(device, **kwargs)
        for k, v in self.edata.items():
            g.edata[k] = v.to(device, **kwargs)
        return g

    def in_degrees(self) -> torch.Tensor:
        return self.csr().degrees()

    def out_degrees(self) -> torch.Tensor:
        if self._csr_t is not None:
            return self._csr_t.degrees()
        if self._out_deg is None:
            if self._table is not None:
                self._out_deg = self._table.by_src().prefix_degree(self._src.numel(), self._n)
            else:
                # the by-destination build has range-checked both id arrays
                self._out_deg = degree(self._src, self._n, validated=self._csr is not None)
        return self._out_deg

    def add_self_loop(self) -> "SnapshotGraph":
        """dgl.add_self_loop: append (v, v) for every node after the existing edges; node
        features are kept, edge features of the new edges are zero-filled."""
        loops = torch.arange(self._n, dtype=torch.int64, device=self.device)
        g = SnapshotGraph(torch.cat([self._src, loops]), torch.cat([self._dst, loops]), self._n)
        for k, v in self.ndata.items():
            g.ndata[k] = v
        for k, v in self.edata.items():
            pad = torch.zeros((self._n,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
            g.edata[k] = torch.cat([v, pad])
        return g

    def remove_edges(self, eids) -> None:
        """DGLGraph.remove_edges: drop the edges with the given ids, in place; the ids of the
        remaining edges are compacted in order (used by the link-prediction loop of
        examples/tf/link_prediction_sliding_window.py:105-106 to hide the positives)."""
        eids = torch.as_tensor(eids, dtype=torch.int64, device=self.device).reshape(-1)
        keep = torch.ones(self._src.numel(), dtype=torch.bool, device=self.device)
        keep[eids] = False
        self._src, self._dst = self._src[keep], self._dst[keep]
        self._table = None  # no longer a prefix of the table
        edata = _FrameDict(self._src.numel(), "edges")
        for k, v in self.edata.items():
            edata[k] = v[keep]
        self.edata = edata
        self._csr = self._csr_t = self._out_deg = None
        self._norms = {}
        self._edge_scales = {}
        self._t2f = None

    def successors(self, v: int) -> torch.Tensor:
        """Destination ids of the edges leaving node v, in edge order."""
        csr_t = self.csr_t()
        lo, hi = csr_t.indptr[v: v + 2].tolist()
        return csr_t.indices[lo:hi].to(torch.int64)

    def predecessors(self, v: int) -> torch.Tensor:
        csr = self.csr()
        lo, hi = csr.indptr[v: v + 2].tolist()
        return csr.indices[lo:hi].to(torch.int64)

    # ---- device structures ------------------------------------------------------------
    def csr(self, want_eid: bool = False, wait: bool = True) -> CSR:
        """By-destination CSR.  `wait=False` returns it with its status word and plan sizes
        still in flight (CSR.ready() collects them; every aggregation does)."""
        if self._csr is None or (want_eid and self._csr.eid is None):
            if self._table is not None and self._src.numel() > 0:
                self._csr = self._table.by_dst().prefix(self._src.numel(), self._n, self._n, want_eid,
                                                        defer=True)
            else:
                # GraphConv's default norm ('both') wants degree^-1/2 of the rows: written by the
                # kernels that write the row pointers, for nothing
                # (and, when no backward pass will build the by-source CSR, the source side's too)
                self._csr = build_csr(self._dst, self._src, self._n, self._n, want_eid=want_eid,
                                      defer=True, norm_mode=0, other_norm=not torch.is_grad_enabled())
        return self._csr.ready() if wait else self._csr

    def csr_t(self, want_eid: bool = False, wait: bool = True) -> CSR:
        if self._csr_t is None or (want_eid and self._csr_t.eid is None):
            if self._table is not None and self._src.numel() > 0:
                self._csr_t = self._table.by_src().prefix(self._src.numel(), self._n, self._n,
                                                          want_eid, defer=True)
            else:
                self._csr_t = build_csr(self._src, self._dst, self._n, self._n, want_eid=want_eid,
                                        defer=True, norm_mode=0)
        return self._csr_t.ready() if wait else self._csr_t

    def norm(self, side: str, mode: int) -> torch.Tensor:
        """GraphConv normaliser: side 'in' (destination, in-degree) or 'out' (source,
        out-degree); mode 0 = clamp(deg,1)^-0.5, 1 = 1/clamp(deg,1)."""
        key = (side, mode)
        if key not in self._norms:
            if side == "in":
                self._norms[key] = self.csr(wait=False).norm(mode)
            elif self._csr_t is not None or torch.is_grad_enabled():
                self._norms[key] = self.csr_t(wait=False).norm(mode)
            else:
                # The by-destination build range-checks both id arrays, and the aggregation
                # that uses this normaliser collects its verdict.  Its source column (int32,
                # ids clamped into range) is counted instead of a second pass over the int64 ids.
                csr = self.csr(wait=False)
                if csr.other_norm(mode) is not None:
                    self._norms[key] = csr.other_norm(mode)
                elif self._out_deg is None and self._table is None:
                    self._norms[key] = norm_from_ids32(csr.indices, self._n, mode)
                else:
                    self._norms[key] = norm_from_degree(self.out_degrees(), mode)
        return self._norms[key]


def graph(data, num_nodes: Optional[int] = None, device=None) -> SnapshotGraph:
    """`dgl.graph((src, dst), num_nodes=...)` as the reference calls it (types.py:191)."""
    src, dst = data
    src, dst = torch.as_tensor(src), torch.as_tensor(dst)
    if device is not None:
        src, dst = src.to(device), dst.to(device)
    return SnapshotGraph(src, dst, num_nodes)
