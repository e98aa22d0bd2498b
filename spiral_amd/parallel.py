"""One snapshot across the GPUs of a node: destination-row blocks, one process per GPU.

The reference has no distributed code; this is the part of the path that shards.  Rank q
owns a contiguous block of destination nodes (cut at equal numbers of nonzeros, because
degrees are power-law), the rows of X, of the node memory and of the output for those
nodes, and the CSR rows of that block.  Per GCN layer there is exactly one exchange:

  forward   every rank receives the source rows its own edges reference, and only those,
            from the ranks that own them (one all-to-all over RCCL: on xGMI that is a direct
            transfer on every one of the 7 links of a GPU at once, where a ring all-gather
            moves everything through one link), then runs the fused aggregation over its
            rows; the indices of its CSR address a compact source table
            [own block | rows received];
  backward  the local transposed aggregation produces a partial gradient for every row
            of that table; the remote part travels back the same way and the owners add it
            to their rows.

How much that saves depends on how thin a rank's edges are spread over the node range: at
E/P edges per rank over N nodes with E/P close to N (8 ranks, 10 edges per node) a power-law
snapshot names about half of the remote rows from any one rank, so half of what an
all-gather would move stays at home; at 2 ranks nearly every row is named and the gain is
the direct transfer alone.  The transfer can be cut into `stages`
batches, each holding a slice of every peer's rows (so every link is busy in every batch):
the aggregation over the sources of batch k runs while batch k+1 is on the wire
(`spmi_gcn_spmm_part` adds batch after batch into the same output).  The layer is also
available in two halves (`graphconv_project` -> `exchange_rows_async` ...
`graphconv_aggregate`) so that independent work, e.g. the recurrent update of the previous
snapshot, runs while the rows travel, and a layer whose input never changes and is held by
every rank needs no exchange at all (`graphconv_replicated`).

Weight and bias gradients are summed with one all-reduce after the backward
(`allreduce_gradients`).  The recurrent update and the MLP are row-local: no exchange.
"""
from __future__ import annotations

import copy
from dataclasses import dataclass
from types import SimpleNamespace
from typing import List, Optional

import torch
import torch.distributed as dist

from . import graph as G
from . import ops

# Kernel entry points used below.  They are the libspmi-backed functions; the CPU tests of
# the exchange logic (gloo, no GPU) substitute checker implementations here.
kernels = SimpleNamespace(build_csr=G.build_csr, aggregate=G.aggregate, degree=G.degree,
                          act_bias_bwd=ops._act_bias_bwd)


def edge_balanced_bounds(in_degree: torch.Tensor, world: int) -> List[int]:
    """Row cuts [b_0=0, ..., b_world=N] such that every block holds about E/world nonzeros."""
    n = in_degree.numel()
    if world == 1 or n == 0:
        return [0] + [n] * world
    csum = torch.cumsum(in_degree.to(torch.int64), 0)
    total = int(csum[-1])
    targets = torch.arange(1, world, dtype=torch.int64, device=csum.device) * total // world
    cuts = (torch.searchsorted(csum, targets, right=False) + 1).clamp_(max=n).tolist()
    bounds = [0] + cuts + [n]
    for i in range(1, len(bounds)):
        bounds[i] = max(bounds[i], bounds[i - 1])
    if n >= world:
        # no empty block, even when one hub row holds more than its share of the edges (the
        # device kernels take at least one row)
        for i in range(1, world):
            bounds[i] = min(max(bounds[i], bounds[i - 1] + 1), n - (world - i))
    return bounds


@dataclass
class RowPartition:
    """Block bounds shared by all ranks plus this rank's position."""

    n: int
    bounds: List[int]
    rank: int
    world: int

    def block(self, q: Optional[int] = None):
        q = self.rank if q is None else q
        return self.bounds[q], self.bounds[q + 1]

    def owner(self, ids: torch.Tensor) -> torch.Tensor:
        """Rank that owns each global node id."""
        b = torch.tensor(self.bounds[1:-1], dtype=torch.int64, device=ids.device)
        return torch.bucketize(ids, b, right=True)

    def workspace(self, device: torch.device) -> "PlanWorkspace":
        """The node-range scratch of this partition's exchange plans on `device`, made once."""
        cache = self.__dict__.setdefault("_workspaces", {})
        key = (device.type, device.index)
        if key not in cache:
            cache[key] = PlanWorkspace(self, device)
        return cache[key]


class PlanWorkspace:
    """What an exchange plan needs over the whole node range, allocated once per partition and
    device and reused by every snapshot (a plan leaves it as it found it):

    deg     int64[N]  how many of the plan's edges leave every node (scratch)
    mark    bool[N]   the remote ids a plan's edges name (scratch)
    csum    int32[N]  inclusive scan of mark (scratch)
    colmap  int64[N]  table row of a node id: the own block's entries never change, a plan
                      overwrites exactly the remote entries it reads afterwards

    A plan is built on the current stream; plans of one partition must not be built on two
    streams at once."""

    def __init__(self, part: RowPartition, device: torch.device):
        lo, hi = part.block()
        self.deg = torch.empty(part.n, dtype=torch.int64, device=device)
        self.mark = torch.zeros(part.n, dtype=torch.bool, device=device)
        self.csum = torch.empty(part.n, dtype=torch.int32, device=device)
        self.colmap = torch.full((part.n,), -1, dtype=torch.int64, device=device)
        self.colmap[lo:hi] = torch.arange(hi - lo, dtype=torch.int64, device=device)
        self.bounds_t = torch.tensor(part.bounds, dtype=torch.int64, device=device)
        ends = self.bounds_t[1:]
        self.end_idx = (ends - 1).clamp_(min=0)       # last node of every block ...
        self.end_ok = (ends > 0).to(torch.int64)      # ... where there is one at or below it


def _slice_bounds(count: int, stages: int) -> List[int]:
    """Cut `count` rows into `stages` consecutive slices; both ends of a transfer compute the
    same cuts from the same count."""
    return [count * k // stages for k in range(stages + 1)]


class ExchangePlan:
    """Which source rows travel where for one snapshot.

    Built from the global source ids of this rank's edges.  The source table this rank
    aggregates from is [own block (n_local rows, in place) | batch 0 | batch 1 | ...]; batch k
    holds slice k of the rows needed from every peer, peers in rank order, rows in ascending
    id order.  `cols` gives every edge its row of that table.

    Cost per snapshot: one count of the edges' sources over the node range (the degree kernels,
    which also gives every named row its number of edges here), a scan over the marks, both in
    persistent buffers (no allocation of that size), ONE host round trip (the P x P matrix of row counts, gathered on
    the device and copied once: it holds what this rank needs from every peer and what every
    peer needs from it), and one all-to-all of the requests.  A request carries the row number
    inside its owner's block and this rank's count of edges that leave the row, so the owners
    can add up out-degrees without any vector over the whole node range (`out_degree_table`)."""

    def __init__(self, src: torch.Tensor, part: RowPartition, group=None, stages: int = 1,
                 data_group=None):
        """`group`: communicator of the plan's own small collectives; `data_group`: communicator
        of the row exchanges.  A caller that keeps the exchange of one snapshot in flight while
        it builds the plan of the next passes two different ones, so that the small messages do
        not queue behind the rows (collectives of one communicator run in order)."""
        self.part, self.group, self.stages = part, group, max(int(stages), 1)
        self.data_group = data_group
        dev = src.device
        world, rank = part.world, part.rank
        lo, hi = part.block()
        self.n_local = hi - lo
        src = src.to(torch.int64)
        ws = part.workspace(dev)

        # distinct remote sources by mark and scan over the node range (no sort), their counts
        # per owner read off the scan at the block ends
        # edges of this rank that leave every node: the marks of the named rows and, later, the
        # counts the out-degrees are added up from.  An id outside the node range is clamped by
        # the count and stopped by the bounds check of the gather through colmap below.
        self._own_partial = None
        if world > 1:
            kernels.degree(src, part.n, validated=True, out=ws.deg)
            self._own_partial = ws.deg[lo:hi].clone()
            torch.gt(ws.deg, 0, out=ws.mark)
            ws.mark[lo:hi] = False                                # own rows stay where they are
            torch.cumsum(ws.mark, 0, dtype=torch.int32, out=ws.csum)
            upto = ws.csum.index_select(0, ws.end_idx).to(torch.int64) * ws.end_ok
            need_dev = torch.diff(upto, prepend=upto.new_zeros(1))
            matrix = torch.empty(world * world, dtype=torch.int64, device=dev)
            dist.all_gather_into_tensor(matrix, need_dev, group=group)
            counts = matrix.view(world, world).cpu()              # the one host round trip
            need = counts[rank].tolist()                          # rows needed from rank q (0 for self)
            wanted = counts[:, rank].tolist()                     # rows rank p wants from this rank
            self.n_remote = sum(need)
            uniq = torch.nonzero_static(ws.mark, size=self.n_remote).reshape(-1)  # ascending ids
        else:
            need, wanted, self.n_remote = [0], [0], 0
            uniq = torch.empty(0, dtype=torch.int64, device=dev)
        self.n_table = self.n_local + self.n_remote
        cut = [sum(need[:q]) for q in range(world + 1)]
        self.need_per_peer, self.send_per_peer = need, wanted

        # table row of every needed id: batch-major, then owner, then ascending id
        peers = [q for q in range(world) if q != rank]
        need_cuts = {q: _slice_bounds(need[q], self.stages) for q in peers}
        self.recv_counts = [[need_cuts[q][k + 1] - need_cuts[q][k] if q != rank else 0
                             for q in range(world)] for k in range(self.stages)]
        self.batch_rows = [sum(c) for c in self.recv_counts]
        self.batch_start = [self.n_local + sum(self.batch_rows[:k]) for k in range(self.stages + 1)]
        seg_pos, seg_row = [], []                             # per (owner, slice): start in uniq, start in table
        for q in peers:
            for k in range(self.stages):
                seg_pos.append(cut[q] + need_cuts[q][k])
                seg_row.append(self.batch_start[k] + sum(self.recv_counts[k][:q]))
        if self.n_remote:
            pos = torch.arange(self.n_remote, dtype=torch.int64, device=dev)
            seg_pos_t = torch.tensor(seg_pos, dtype=torch.int64, device=dev)
            seg_row_t = torch.tensor(seg_row, dtype=torch.int64, device=dev)
            seg = torch.bucketize(pos, seg_pos_t[1:], right=True)
            table_row = pos - seg_pos_t[seg] + seg_row_t[seg]
        else:
            table_row = torch.empty(0, dtype=torch.int64, device=dev)
        self._uniq, self._table_row = uniq, table_row
        ws.colmap[uniq] = table_row
        self.cols = ws.colmap.index_select(0, src)            # table row of every edge's source
        self._table_ids = self._out_degree_table = None
        # this rank's edges out of every named row (a single rank names none and counts on demand)
        self._uniq_partial = ws.deg.index_select(0, uniq) if world > 1 else None
        self._src = src if world == 1 else None

        # tell every owner which of its rows are wanted here, and how many edges leave them here
        self.send_counts = [[0] * world for _ in range(self.stages)]
        self.send_index = torch.empty(0, dtype=torch.int64, device=dev)
        self._asked = self._asked_count = self.send_index
        if world > 1:
            request = uniq - ws.bounds_t[part.owner(uniq)]    # row inside the owner's block
            message = torch.stack([request, self._uniq_partial], 1).to(torch.int32)   # both below 2^31
            asked = torch.empty(sum(wanted), 2, dtype=torch.int32, device=dev)
            dist.all_to_all_single(asked, message, wanted, need, group=group)
            self._asked, self._asked_count = asked[:, 0].to(torch.int64), asked[:, 1].to(torch.int64)
            # reorder the requests batch-major, the order the rows are sent in
            starts = [sum(wanted[:p]) for p in range(world)]
            pieces = []
            for k in range(self.stages):
                for p in range(world):
                    c = _slice_bounds(wanted[p], self.stages)
                    self.send_counts[k][p] = c[k + 1] - c[k]
                    pieces.append(self._asked[starts[p] + c[k]: starts[p] + c[k + 1]])
            self.send_index = torch.cat(pieces) if self.stages > 1 else self._asked
        self.send_batch_start = [sum(sum(c) for c in self.send_counts[:k]) for k in range(self.stages + 1)]
        self._return_csrs = None

    def columns(self, src: torch.Tensor) -> torch.Tensor:
        """Row of this rank's source table for global source ids (own ids or ids the plan's
        edges name).  The plan's own edges have theirs in `cols`; this is the lookup for other
        id lists, by binary search in the plan's sorted remote ids."""
        src = src.to(torch.int64)
        lo, hi = self.part.block()
        own = (src >= lo) & (src < hi)
        if self.n_remote == 0:
            return src - lo
        pos = torch.searchsorted(self._uniq, src).clamp_(max=self.n_remote - 1)
        return torch.where(own, src - lo, self._table_row[pos])

    def batch_of_column(self, cols: torch.Tensor) -> torch.Tensor:
        """Batch whose arrival completes a column: own rows and batch 0 are 0."""
        b = torch.tensor(self.batch_start[1:-1], dtype=torch.int64, device=cols.device)
        return torch.bucketize(cols, b, right=True)

    @property
    def table_ids(self) -> torch.Tensor:
        """Global id of every row of the source table (own block first), e.g. to look up
        per-node vectors; made on first use (layers with an exchange never need it)."""
        if self._table_ids is None:
            lo, hi = self.part.block()
            dev = self._uniq.device
            order = torch.empty(self.n_remote, dtype=torch.int64, device=dev)
            order[self._table_row - self.n_local] = self._uniq
            self._table_ids = torch.cat([torch.arange(lo, hi, dtype=torch.int64, device=dev), order])
        return self._table_ids

    def out_degree_table(self) -> torch.Tensor:
        """Out-degree in the whole graph (int64) of every row of this rank's source table.
        A rank counts the edges that leave a row among its own edges; the counts of remote rows
        travelled to the rows' owners with the requests; the owners add what they were sent to
        their own counts and answer every request with the total: one more all-to-all of
        n_remote numbers, against an all-reduce over the whole node range."""
        if self._out_degree_table is None:
            if self.part.world == 1:
                self._out_degree_table = kernels.degree(self._src, self.part.n, validated=True).to(torch.int64)
            else:
                own = self._own_partial.clone()
                own.index_add_(0, self._asked, self._asked_count)   # integer adds: order-free
                reply = own.index_select(0, self._asked)
                back = torch.empty(self.n_remote, dtype=torch.int64, device=own.device)
                dist.all_to_all_single(back, reply, self.need_per_peer, self.send_per_peer, group=self.group)
                table = torch.empty(self.n_table, dtype=torch.int64, device=own.device)
                table[: self.n_local] = own
                table[self._table_row] = back
                self._out_degree_table = table
        return self._out_degree_table

    def return_csrs(self) -> list:
        """For the backward pass: per batch, the rows that come back from the peers grouped by
        the own row they add to (a CSR with one column per returned row, columns numbered over
        all batches), so that the owner's additions are one aggregation launch per batch with a
        fixed order of summation inside a row (peers in rank order) and no atomics."""
        if self._return_csrs is None:
            self._return_csrs = []
            total = self.send_index.numel()
            for k in range(self.stages):
                a, b = self.send_batch_start[k], self.send_batch_start[k + 1]
                if a == b:
                    self._return_csrs.append(None)
                    continue
                pos = torch.arange(a, b, dtype=torch.int64, device=self.send_index.device)
                self._return_csrs.append(kernels.build_csr(self.send_index[a:b], pos, self.n_local, total))
        return self._return_csrs


class PendingTable:
    """A source table whose remote batches may still be on the wire.  `wait(k)` returns once
    batches 0..k have arrived (k defaults to all); the transfers run on the communicator's
    own stream, so whatever the caller launches in between overlaps them."""

    def __init__(self, table: torch.Tensor, works: list, keep, prescaled: bool):
        self.table, self._works, self._keep, self.prescaled = table, list(works), keep, prescaled
        self._done = 0

    def wait(self, k: Optional[int] = None) -> torch.Tensor:
        upto = len(self._works) if k is None else min(k + 1, len(self._works))
        while self._done < upto:
            if self._works[self._done] is not None:
                self._works[self._done].wait()
            self._done += 1
        if self._done == len(self._works):
            self._keep = None
        return self.table


def exchange_rows_async(z_local: torch.Tensor, plan: ExchangePlan,
                        payload_dtype: Optional[torch.dtype] = None,
                        table: Optional[torch.Tensor] = None, prescaled: bool = False) -> PendingTable:
    """Start moving the rows of `z_local` (this rank's block, [n_local, F]) that other ranks
    need, and the rows this rank needs from them, into its source table.  One all-to-all per
    batch of the plan, queued back to back.  `table`: a [n_table, F] tensor whose head already
    holds `z_local` (the projection wrote it there), otherwise one is allocated and the block
    copied in.  payload_dtype=torch.bfloat16 halves the bytes on the wire; the aggregation then
    reads bfloat16 rows and still sums in float32 (a documented precision trade, off by
    default)."""
    z = z_local.detach()
    dtype = payload_dtype or z.dtype
    f = z.shape[1]
    if table is None or table.dtype != dtype:
        table = torch.empty(plan.n_table, f, dtype=dtype, device=z.device)
        table[: plan.n_local] = z
    works = []
    send = None
    if plan.part.world > 1:
        send = z.index_select(0, plan.send_index)
        if send.dtype != dtype:
            send = send.to(dtype)
        for k in range(plan.stages):
            out = table[plan.batch_start[k]: plan.batch_start[k + 1]]
            inp = send[plan.send_batch_start[k]: plan.send_batch_start[k + 1]]
            works.append(dist.all_to_all_single(out, inp, plan.recv_counts[k], plan.send_counts[k],
                                                group=plan.data_group, async_op=True))
    else:
        works = [None] * plan.stages
    return PendingTable(table, works, send, prescaled)


def _return_rows_async(g_remote: torch.Tensor, plan: ExchangePlan):
    """Start the reverse transfer of the backward pass: `g_remote` [n_remote, F] is the gradient
    of the remote part of this rank's source table (batch-major, as the rows arrived); every
    batch goes back to the owners of its rows with one all-to-all, queued back to back.
    Returns (buffer that receives the rows coming back to this rank, one handle per batch)."""
    f = g_remote.shape[1]
    back = g_remote.new_empty(plan.send_index.numel(), f)
    works = []
    for k in range(plan.stages):
        inp = g_remote[plan.batch_start[k] - plan.n_local: plan.batch_start[k + 1] - plan.n_local]
        out = back[plan.send_batch_start[k]: plan.send_batch_start[k + 1]]
        works.append(dist.all_to_all_single(out, inp, plan.send_counts[k], plan.recv_counts[k],
                                            group=plan.data_group, async_op=True))
    return back, works


def _add_returned_rows(g_local: torch.Tensor, back: torch.Tensor, works: list, plan: ExchangePlan) -> torch.Tensor:
    """Add the rows that came back to the own block's gradient, batch by batch as they land:
    one aggregation launch per batch over the plan's return CSR (row = own row, columns = the
    returned rows that belong to it, peers in rank order), which adds into `g_local` in place.
    No atomics, a fixed order inside every row: bit-reproducible."""
    csrs = plan.return_csrs()
    for k, work in enumerate(works):
        if work is not None:
            work.wait()
        if csrs[k] is not None:
            kernels.aggregate(csrs[k], back, None, None, None, None, None, 0.01, out=g_local,
                              accumulate=True, finalize=False)
    return g_local


def _return_rows(g_table: torch.Tensor, plan: ExchangePlan) -> torch.Tensor:
    """The reverse transfer for a gradient over the whole source table [n_table, F]: the remote
    part goes back to the owners, who add what they receive to their own block's part.
    Returns the gradient of this rank's block [n_local, F]."""
    g_local = g_table[: plan.n_local].clone()
    if plan.part.world == 1:
        return g_local
    back, works = _return_rows_async(g_table[plan.n_local:], plan)
    return _add_returned_rows(g_local, back, works, plan)


def _with_indices(csr, indices: torch.Tensor, n_cols: int):
    """The same grouping (row pointers, hub-row plan) over another numbering of the columns."""
    out = copy.copy(csr)
    out.indices, out.n_cols = indices, n_cols
    return out


class ShardedSnapshot(G.NormalisedAdjacency):
    """This rank's destination block of one snapshot: the edges that end in the block, their
    sources renumbered to rows of the compact source table of the exchange plan."""

    def __init__(self, src: torch.Tensor, dst: torch.Tensor, part: RowPartition, group=None,
                 stages: int = 1, data_group=None):
        """src/dst: the edges whose destination lies in this rank's block (global ids).
        `group` carries the per-snapshot bookkeeping (plan, out-degrees), `data_group` the row
        exchanges (see ExchangePlan)."""
        self.part, self.group = part, group
        lo, hi = part.block()
        self.n_local = hi - lo
        self.src_global = src.to(torch.int64)
        self.plan = ExchangePlan(self.src_global, part, group, stages, data_group)
        self.cols = self.plan.cols
        self.dst_rows = dst.to(torch.int64) - lo
        self._csr = self._csr_t = self._batch_csrs = self._global = None
        self._csr_t_parts = None
        self._norms = {}
        self._batch_scales = {}

    @classmethod
    def from_global(cls, src, dst, part: RowPartition, group=None, stages: int = 1,
                    data_group=None) -> "ShardedSnapshot":
        """Keep the edges of a full edge list that end in this rank's block."""
        lo, hi = part.block()
        keep = (dst >= lo) & (dst < hi)
        return cls(src[keep], dst[keep], part, group, stages, data_group)

    def csr(self, want_eid: bool = False, wait: bool = True):
        if self._csr is None or (want_eid and self._csr.eid is None):
            self._csr = kernels.build_csr(self.dst_rows, self.cols, self.n_local, self.plan.n_table,
                                          want_eid=want_eid)
        return self._csr

    def csr_t(self, want_eid: bool = False, wait: bool = True):
        if self._csr_t is None:
            self._csr_t = kernels.build_csr(self.cols, self.dst_rows, self.plan.n_table, self.n_local)
        return self._csr_t

    def csr_t_parts(self):
        """The by-source grouping in two pieces, for a backward pass that sends the remote rows'
        gradient on its way before it computes its own: (edges whose source row is remote,
        grouped by table row - n_local; edges whose source is an own row, grouped by it)."""
        if self._csr_t_parts is None:
            remote = self.cols >= self.n_local
            own = ~remote
            self._csr_t_parts = (
                None if self.plan.n_remote == 0 else
                kernels.build_csr(self.cols[remote] - self.n_local, self.dst_rows[remote],
                                  self.plan.n_remote, self.n_local),
                kernels.build_csr(self.cols[own], self.dst_rows[own], self.n_local, self.n_local))
        return self._csr_t_parts

    def batch_csrs(self) -> list:
        """One CSR per batch of the plan: the edges whose source row arrives with that batch
        (own rows count to batch 0)."""
        if self._batch_csrs is None:
            if self.plan.stages == 1:
                self._batch_csrs = [self.csr()]
            else:
                batch = self.plan.batch_of_column(self.cols)
                self._batch_csrs = []
                for k in range(self.plan.stages):
                    keep = batch == k
                    self._batch_csrs.append(kernels.build_csr(self.dst_rows[keep], self.cols[keep],
                                                              self.n_local, self.plan.n_table))
        return self._batch_csrs

    def batch_edge_scale(self, k: int, norm: str) -> Optional[torch.Tensor]:
        """Per-nonzero source normaliser of batch k's CSR (cached per snapshot and norm)."""
        if norm not in ("left", "both"):
            return None
        key = (k, norm)
        if key not in self._batch_scales:
            csr = self.batch_csrs()[k]
            self._batch_scales[key] = self.norm("out", 0 if norm == "both" else 1).index_select(
                0, csr.indices.to(torch.int64))
        return self._batch_scales[key]

    def out_degree_table(self) -> torch.Tensor:
        """Out-degree in the whole graph of every row of this rank's source table (int64)."""
        return self.plan.out_degree_table()

    def _norm_of(self, deg: torch.Tensor, mode: int) -> torch.Tensor:
        d = deg.to(torch.float32).clamp_(min=1)
        return d.sqrt_().reciprocal_() if mode == 0 else d.reciprocal_()

    def norm(self, side: str, mode: int) -> torch.Tensor:
        """'in': this block's destination normaliser (local rows are complete).
        'out': source normaliser of every row of this rank's source table."""
        key = (side, mode)
        if key not in self._norms:
            if side == "in":
                self._norms[key] = self._norm_of(self.csr().degrees(), mode)
            else:
                self._norms[key] = self._norm_of(self.out_degree_table(), mode)
        return self._norms[key]

    def norm_global(self, mode: int) -> torch.Tensor:
        """Source normaliser as a vector over the whole node range [N]: the nodes this rank's
        edges name (and its own) hold their value, every other entry is 1 and is never read by
        this rank's aggregations.  Allocates over the node range; the layers use the table
        form (`norm('out', mode)`)."""
        key = ("global", mode)
        if key not in self._norms:
            full = torch.ones(self.part.n, dtype=torch.float32, device=self.cols.device)
            full[self.plan.table_ids] = self.norm("out", mode)
            self._norms[key] = full
        return self._norms[key]

    def norm_out_local(self, mode: int) -> torch.Tensor:
        """Source normaliser of this rank's own rows [n_local]: what the owner multiplies its
        projected rows by before they travel."""
        return self.norm("out", mode)[: self.n_local]

    def global_view(self) -> "_GlobalSources":
        """The same block with sources numbered by global node id, for inputs every rank holds
        in full."""
        if self._global is None:
            self._global = _GlobalSources(self)
        return self._global


class _GlobalSources(G.NormalisedAdjacency):
    """A ShardedSnapshot whose CSR addresses a matrix with one row per node of the graph."""

    def __init__(self, shard: ShardedSnapshot):
        self.shard = shard
        self._csr = self._csr_t = None

    def csr(self, want_eid: bool = False, wait: bool = True):
        if self._csr is None:
            base = self.shard.csr()
            cols = self.shard.plan.table_ids.index_select(0, base.indices.to(torch.int64)).to(base.indices.dtype)
            self._csr = _with_indices(base, cols, self.shard.part.n)
        return self._csr

    def csr_t(self, want_eid: bool = False, wait: bool = True):
        if self._csr_t is None:
            self._csr_t = kernels.build_csr(self.shard.src_global, self.shard.dst_rows, self.shard.part.n,
                                            self.shard.n_local)
        return self._csr_t

    def norm(self, side: str, mode: int) -> torch.Tensor:
        return self.shard.norm("in", mode) if side == "in" else self.shard.norm_global(mode)

    def edge_scale(self, direction: str, norm: str, edge_weight: Optional[torch.Tensor] = None):
        """Forward weights from the table form of the source normaliser (the block's CSR with
        table column numbers has the same order of nonzeros), so that nothing over the whole
        node range is built for a layer that only runs forward."""
        if direction == "fwd" and edge_weight is None and norm in ("left", "both"):
            cache = self._edge_scale_cache()
            key = (direction, norm)
            if key not in cache:
                base = self.shard.csr()
                cache[key] = self.shard.norm("out", 0 if norm == "both" else 1).index_select(
                    0, base.indices.to(torch.int64))
            return cache[key]
        return super().edge_scale(direction, norm, edge_weight)


def _aggregate_batches(shard: ShardedSnapshot, pending: PendingTable, norm: str, dst_scale, bias, act,
                       slope) -> torch.Tensor:
    """The aggregation of a source table into this rank's destination block, batch by batch as
    the batches land (the sum over the sources of batch k is added while batch k+1 is on the
    wire); the last launch applies destination scale, bias and activation."""
    csrs = shard.batch_csrs()
    rst = None
    for k, csr in enumerate(csrs):
        table = pending.wait(k)
        edge_w = None if pending.prescaled else shard.batch_edge_scale(k, norm)
        if len(csrs) == 1:
            rst = kernels.aggregate(csr, table, None, edge_w, dst_scale, bias, act, slope,
                                    out_dtype=torch.float32)
        else:
            rst = kernels.aggregate(csr, table, None, edge_w, dst_scale, bias, act, slope, out=rst,
                                    out_dtype=torch.float32, accumulate=k > 0, finalize=k == len(csrs) - 1)
    return rst


class _ShardedAggregate(torch.autograd.Function):
    """The exchange + aggregation of one layer under autograd.

    forward   the batches of the plan are queued back to back and aggregated as they arrive
              (`_aggregate_batches`, the chain the inference path uses);
    backward  the gradient of the REMOTE rows of the source table is computed first and sent on
              its way (asynchronous all-to-alls, one per batch); the gradient of the own rows is
              computed while it travels; the rows that come back are added batch by batch with
              one launch each (`_add_returned_rows`)."""

    @staticmethod
    def forward(ctx, x_local, bias, shard: ShardedSnapshot, norm, act, slope):
        mode = 0 if norm == "both" else 1
        dst_scale = shard.norm("in", mode) if norm in ("right", "both") else None
        y = _aggregate_batches(shard, exchange_rows_async(x_local, shard.plan), norm, dst_scale,
                               None if bias is None else bias.detach(), act, slope)
        ctx.shard, ctx.norm, ctx.act, ctx.slope, ctx.has_bias = shard, norm, act, slope, bias is not None
        ctx.save_for_backward(y if act not in (None, "none") else None)
        return y

    @staticmethod
    def backward(ctx, g_y):
        (y,) = ctx.saved_tensors
        shard, norm = ctx.shard, ctx.norm
        need_b = ctx.needs_input_grad[1] and ctx.has_bias
        g_p, g_b = g_y, None
        if y is not None or need_b:
            g_p, g_b = kernels.act_bias_bwd(g_y, y if y is not None else g_y, ctx.act, ctx.slope,
                                            need_b)
        g_x = None
        if ctx.needs_input_grad[0]:
            mode = 0 if norm == "both" else 1
            plan, n_local = shard.plan, shard.n_local
            g_p = g_p.contiguous()
            out_scale = shard.norm("out", mode) if norm in ("left", "both") else None
            in_w = shard.norm("in", mode) if norm in ("right", "both") else None
            if plan.part.world == 1:
                edge_w = shard.edge_scale("bwd", norm)
                g_table = kernels.aggregate(shard.csr_t(), g_p, None, edge_w, out_scale, None, None)
                g_x = _return_rows(g_table, plan)
            else:
                remote_t, own_t = shard.csr_t_parts()

                def weights(csr):
                    return None if in_w is None else in_w.index_select(0, csr.indices.to(torch.int64))

                if remote_t is None:  # no edge of this block leaves a remote row: nothing to send,
                    g_remote = g_p.new_empty(0, g_p.shape[1])  # but the collectives are everyone's
                else:
                    g_remote = kernels.aggregate(remote_t, g_p, None, weights(remote_t),
                                                 None if out_scale is None else out_scale[n_local:].contiguous(),
                                                 None, None)
                back, works = _return_rows_async(g_remote, plan)
                g_own = kernels.aggregate(own_t, g_p, None, weights(own_t),
                                          None if out_scale is None else out_scale[:n_local].contiguous(),
                                          None, None)
                g_x = _add_returned_rows(g_own, back, works, plan)
        return g_x, g_b, None, None, None, None


class _TableAggregate(torch.autograd.Function):
    """ops.GcnAggregate through the `kernels` indirection of this module: the aggregation of
    a matrix this rank already holds in full over any NormalisedAdjacency `g` (no exchange)."""

    @staticmethod
    def forward(ctx, x, bias, g, norm, act, slope):
        mode = 0 if norm == "both" else 1
        dst_scale = g.norm("in", mode) if norm in ("right", "both") else None
        y = kernels.aggregate(g.csr(), x.detach(), None, g.edge_scale("fwd", norm), dst_scale,
                              None if bias is None else bias.detach(), act, slope)
        ctx.g, ctx.norm, ctx.act, ctx.slope, ctx.has_bias = g, norm, act, slope, bias is not None
        ctx.save_for_backward(y if act not in (None, "none") else None)
        return y

    @staticmethod
    def backward(ctx, g_y):
        (y,) = ctx.saved_tensors
        g, norm = ctx.g, ctx.norm
        need_b = ctx.needs_input_grad[1] and ctx.has_bias
        g_p, g_b = g_y, None
        if y is not None or need_b:
            g_p, g_b = kernels.act_bias_bwd(g_y, y if y is not None else g_y, ctx.act, ctx.slope, need_b)
        g_x = None
        if ctx.needs_input_grad[0]:
            mode = 0 if norm == "both" else 1
            out_scale = g.norm("out", mode) if norm in ("left", "both") else None
            g_x = kernels.aggregate(g.csr_t(), g_p.contiguous(), None, g.edge_scale("bwd", norm),
                                    out_scale, None, None)
        return g_x, g_b, None, None, None, None


def sharded_aggregate(shard: ShardedSnapshot, x_local: torch.Tensor, norm: str = "both",
                      bias=None, act=None, slope: float = 0.01) -> torch.Tensor:
    """Y_local = act(norm_in * A_local (norm_out * exchanged(X_local)) + bias), differentiable."""
    return _ShardedAggregate.apply(x_local, bias, shard, norm, act, slope)


def _split_activation(conv):
    from .nn.graphconv import _classify_activation

    return _classify_activation(conv._activation)


def _tracked(conv, *tensors) -> bool:
    return torch.is_grad_enabled() and (any(t is not None and t.requires_grad for t in tensors) or any(
        p is not None and p.requires_grad for p in (conv.weight, conv.bias)))


def _finish(conv, rst: torch.Tensor, code, slope, unfused) -> torch.Tensor:
    """What follows the aggregation when the layer aggregates first (in_feats < out_feats)."""
    import torch.nn.functional as F

    weight = conv.weight
    if weight is not None and not _tracked(conv, rst):
        fused = ops.project_rows_inference(rst, weight, conv.bias, code, slope)
        if fused is not None:  # projection, bias and activation in one pass over the rows
            return fused if unfused is None else unfused(fused)
    if weight is not None:
        rst = ops.rows_times_weight(rst, weight, conv.bias)
    elif conv.bias is not None:
        rst = rst + conv.bias
    if code == "relu":
        rst = F.relu(rst)
    elif code == "leaky_relu":
        rst = F.leaky_relu(rst, slope)
    return rst if unfused is None else unfused(rst)


def sharded_graphconv(conv, shard: ShardedSnapshot, feat_local: torch.Tensor) -> torch.Tensor:
    """A spiral_amd GraphConv layer applied to this rank's rows of a sharded snapshot:
    same parameters and order of operations as GraphConv.forward, with the exchange in
    front of the aggregation (after X W when in_feats >= out_feats, so the smaller matrix
    travels).  Differentiable; without autograd it takes the two-half path below."""
    if not _tracked(conv, feat_local):
        return graphconv_aggregate(conv, shard, graphconv_project(conv, feat_local, shard))
    norm = conv._norm
    code, slope, unfused = _split_activation(conv)
    if conv._in_feats >= conv._out_feats:
        if conv.weight is not None:
            feat_local = ops.rows_times_weight(feat_local, conv.weight)
        rst = sharded_aggregate(shard, feat_local, norm, conv.bias, code, slope)
        return rst if unfused is None else unfused(rst)
    return _finish(conv, sharded_aggregate(shard, feat_local, norm), code, slope, unfused)


# ---- the same layer in two halves, for callers that overlap the exchange -----------------
def graphconv_project(conv, feat_local: torch.Tensor, shard: ShardedSnapshot,
                      payload_dtype: Optional[torch.dtype] = None) -> PendingTable:
    """First half of a layer without autograd: produce the rows that have to travel (X W when
    the layer multiplies first, X itself otherwise) and start their exchange.  When the
    projection runs on spmi_gemm_rows it writes straight into the head of the source table and
    applies the source normaliser in its epilogue, so the aggregation needs no per-edge
    weight."""
    plan, norm = shard.plan, conv._norm
    first = conv._in_feats >= conv._out_feats and conv.weight is not None
    feat_local = feat_local.detach()
    if not first:
        return exchange_rows_async(feat_local, plan, payload_dtype)
    w = conv.weight.detach()
    mode = 0 if norm == "both" else 1
    if ops._gemm_rows_ok(feat_local, w.shape[0], w.shape[1]):
        scale = shard.norm_out_local(mode).contiguous() if norm in ("both", "left") else None
        table = None
        if payload_dtype in (None, torch.float32):
            table = torch.empty(plan.n_table, w.shape[1], dtype=torch.float32, device=feat_local.device)
        z = ops.gemm_rows(feat_local, w.t(), None, None, 0.01, scale,
                          planes=ops.weight_planes(conv.weight, "f16x3", True, ops.F16X3_GEMM_ROWS_LIMIT),
                          out=None if table is None else table[: plan.n_local])
        return exchange_rows_async(z, plan, payload_dtype, table=table, prescaled=True)
    return exchange_rows_async(feat_local @ w, plan, payload_dtype)


def graphconv_aggregate(conv, shard: ShardedSnapshot, pending: PendingTable) -> torch.Tensor:
    """Second half without autograd: the aggregation of the source table into this rank's
    destination block, batch by batch as the batches arrive, then bias/activation (and W when
    the layer aggregates first).  Sums are float32 whatever the payload type."""
    norm = conv._norm
    code, slope, unfused = _split_activation(conv)
    first = conv._in_feats >= conv._out_feats
    mode = 0 if norm == "both" else 1
    dst_scale = shard.norm("in", mode) if norm in ("right", "both") else None
    bias = conv.bias.detach() if (first and conv.bias is not None) else None
    rst = _aggregate_batches(shard, pending, norm, dst_scale, bias, code if first else None, slope)
    if first:
        return rst if unfused is None else unfused(rst)
    return _finish(conv, rst, code, slope, unfused)


def graphconv_replicated(conv, shard: ShardedSnapshot, x_full: torch.Tensor) -> torch.Tensor:
    """A layer whose input every rank holds in full ([N, F] in node order, e.g. the node table
    of a VEInvariantDTDG, which never changes): no exchange, and no redundant work either,
    because the rows are aggregated into this rank's block first and only the block is
    projected: A (X W) = (A X) W, the order DGL itself takes when in_feats <= out_feats.
    Projecting the replicated matrix first would cost every rank the GEMM of the whole graph.
    Differentiable with respect to the layer's parameters."""
    code, slope, unfused = _split_activation(conv)
    rst = _TableAggregate.apply(x_full, None, shard.global_view(), conv._norm, None, 0.01)
    return _finish(conv, rst, code, slope, unfused)


def allreduce_gradients(module: torch.nn.Module, group=None) -> None:
    """Sum parameter gradients over ranks in one flat all-reduce (they are a few KB-MB).
    The sum is the gradient of the global loss only if every rank's loss is its share of one
    sum normalised by the global row count (e.g. `F.mse_loss(..., reduction="sum") / N_total`):
    a per-rank mean would weight rows by the inverse size of their block."""
    grads = [p.grad for p in module.parameters() if p.grad is not None]
    if not grads or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, group=group)
    off = 0
    for g in grads:
        g.copy_(flat[off: off + g.numel()].view_as(g))
        off += g.numel()
