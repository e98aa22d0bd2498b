"""CPU oracle for the spiro.dtdg per-snapshot GCN message-passing path.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py, never by spiral_amd/.

Two layers:

* thin ctypes wrappers over oracle/_build/liboracle.so (spmi_oracle.c), the C
  restatement used for the bit-exact integer checks and as the timed CPU baseline;
* numpy / torch restatements of the layers above it, each citing the reference
  lines it follows.  The arithmetic of GraphConv lives in DGL 0.8.0.post1
  (reference poetry.lock:206-207), which is not vendored; `graphconv_forward`
  follows DGL's published GraphConv.forward step by step.

Pinning status
  degree / CSR / citation feature / anchors: pinned by the reference's own
    fixtures (tests/dtdg/test_types_torch.py:12-14,36-56; tests/core/test_utils.py:47-53).
  GRU / LSTM / RNN / MLP: pinned by goldens generated from the reference's own torch
    decoders (tests/golden/make_golden.py).
  GraphConv aggregation, losses, gradients: PARITY UNPINNED by the reference (it holds
    no numeric fixture and DGL is absent); guarded by three independent restatements
    that must agree (C sequential, numpy add.at, scipy CSR).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile spmi_oracle.c with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(
        os.path.join(_HERE, "spmi_oracle.c")
    ):
        subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.orc_max_threads.restype = ctypes.c_int
        _lib.orc_degree.restype = ctypes.c_int
        _lib.orc_coo_to_csr.restype = ctypes.c_int
    return _lib


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def max_threads() -> int:
    return int(lib().orc_max_threads())


# ----------------------------------------------------------------------------- integer path
def degree(ids: np.ndarray, n: int) -> np.ndarray:
    """DGLGraph.in_degrees()/out_degrees() (spiro/dtdg/types.py:241)."""
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    out = np.zeros(n, dtype=np.int64)
    rc = lib().orc_degree(_p(ids), ctypes.c_int64(ids.size), ctypes.c_int64(n), _p(out))
    if rc != 0:
        raise ValueError("node id out of range")
    return out


def coo_to_csr(key: np.ndarray, other: np.ndarray, n: int, n_other: Optional[int] = None
               ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """dgl.graph((src,dst)) + COOToCSR (spiro/dtdg/types.py:191): stable grouping by key."""
    key = np.ascontiguousarray(key, dtype=np.int64)
    other = np.ascontiguousarray(other, dtype=np.int64)
    n_other = n if n_other is None else n_other
    e = key.size
    indptr = np.zeros(n + 1, dtype=np.int32)
    indices = np.zeros(e, dtype=np.int32)
    eid = np.zeros(e, dtype=np.int32)
    rc = lib().orc_coo_to_csr(_p(key), _p(other), ctypes.c_int64(e), ctypes.c_int64(n),
                              ctypes.c_int64(n_other), _p(indptr), _p(indices), _p(eid))
    if rc != 0:
        raise ValueError("node id out of range" if rc == -1 else "out of memory")
    return indptr, indices, eid


def norm_from_indptr(indptr: np.ndarray, mode: str = "rsqrt") -> np.ndarray:
    """degs.float().clamp(min=1) ** -0.5 (norm='both') or 1/degs (DGL GraphConv.forward)."""
    indptr = np.ascontiguousarray(indptr, dtype=np.int32)
    n = indptr.size - 1
    out = np.zeros(n, dtype=np.float32)
    lib().orc_norm(_p(indptr), ctypes.c_int64(n), ctypes.c_int(0 if mode == "rsqrt" else 1), _p(out))
    return out


_ACT = {None: 0, "none": 0, "relu": 1, "leaky_relu": 2}


def aggregate(indptr, indices, x, src_scale=None, edge_w=None, dst_scale=None, bias=None,
              act=None, slope: float = 0.01, threads: int = 0) -> np.ndarray:
    """Sequential CSR-order fp32 aggregation with the GraphConv pre/post scaling (C)."""
    indptr = np.ascontiguousarray(indptr, dtype=np.int32)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    x = np.ascontiguousarray(x, dtype=np.float32)
    n_rows = indptr.size - 1
    f = x.shape[1]
    y = np.empty((n_rows, f), dtype=np.float32)

    def c(a):
        return None if a is None else np.ascontiguousarray(a, dtype=np.float32)

    ss, ew, ds, b = c(src_scale), c(edge_w), c(dst_scale), c(bias)
    lib().orc_aggregate(_p(indptr), _p(indices), ctypes.c_int64(n_rows), _p(x), ctypes.c_int64(f),
                        ctypes.c_int64(f), _p(ss), _p(ew), _p(ds), _p(b), ctypes.c_int(_ACT[act]),
                        ctypes.c_float(slope), _p(y), ctypes.c_int64(f), ctypes.c_int(threads))
    return y


def aggregate_add_at(src, dst, n_dst, x, w=None) -> np.ndarray:
    """Independent restatement #2: scatter-add over the COO list (np.add.at)."""
    msg = x[np.asarray(src)]
    if w is not None:
        msg = msg * np.asarray(w, dtype=np.float32)[:, None]
    out = np.zeros((n_dst, x.shape[1]), dtype=np.float32)
    np.add.at(out, np.asarray(dst), msg)
    return out


def aggregate_scipy(src, dst, n_dst, n_src, x, w=None) -> np.ndarray:
    """Independent restatement #3: scipy CSR (duplicates summed) times dense."""
    import scipy.sparse as sp

    vals = np.ones(len(src), dtype=np.float32) if w is None else np.asarray(w, dtype=np.float32)
    a = sp.coo_matrix((vals, (np.asarray(dst), np.asarray(src))), shape=(n_dst, n_src)).tocsr()
    return np.asarray(a @ x, dtype=np.float32)


def _activation(y: np.ndarray, act, slope=0.01) -> np.ndarray:
    if act in (None, "none"):
        return y
    if act == "relu":
        return np.where(y > 0, y, np.float32(0)).astype(np.float32)
    if act == "leaky_relu":
        return np.where(y > 0, y, y * np.float32(slope)).astype(np.float32)
    raise ValueError(act)


def graphconv_forward(src, dst, n, x, weight=None, bias=None, norm="both", act=None,
                      slope: float = 0.01, allow_zero_in_degree: bool = False, edge_weight=None
                      ) -> np.ndarray:
    """DGL 0.8 GraphConv.forward, step by step, on CPU in fp32.

    Called by the reference at staticGraphEncoder.py:113,116.  Order of operations:
    zero-in-degree check; source scaling by out-degree norm ('left','both'); X W first
    iff in_feats > out_feats; sum aggregation (multi-edges counted, CSR order);
    destination scaling by in-degree norm ('right','both'); + bias; activation.
    """
    if norm not in ("none", "both", "right", "left"):
        raise ValueError(f"Invalid norm value {norm}")
    src = np.asarray(src, dtype=np.int64)
    dst = np.asarray(dst, dtype=np.int64)
    x = np.asarray(x, dtype=np.float32)
    in_deg = degree(dst, n)
    if not allow_zero_in_degree and (in_deg == 0).any():
        raise ValueError("There are 0-in-degree nodes in the graph")
    indptr, indices, eid = coo_to_csr(dst, src, n, n)
    ew = None if edge_weight is None else np.asarray(edge_weight, dtype=np.float32)[eid]
    feat = x
    if norm in ("left", "both"):
        degs = np.maximum(degree(src, n).astype(np.float32), np.float32(1))
        nrm = (degs ** np.float32(-0.5)) if norm == "both" else (np.float32(1) / degs)
        feat = (feat * nrm[:, None]).astype(np.float32)
    in_feats = x.shape[1]
    out_feats = in_feats if weight is None else weight.shape[1]
    if in_feats > out_feats and weight is not None:
        feat = (feat @ weight).astype(np.float32)
        rst = aggregate(indptr, indices, feat, edge_w=ew)
    else:
        rst = aggregate(indptr, indices, feat, edge_w=ew)
        if weight is not None:
            rst = (rst @ weight).astype(np.float32)
    if norm in ("right", "both"):
        degs = np.maximum(in_deg.astype(np.float32), np.float32(1))
        nrm = (degs ** np.float32(-0.5)) if norm == "both" else (np.float32(1) / degs)
        rst = (rst * nrm[:, None]).astype(np.float32)
    if bias is not None:
        rst = (rst + np.asarray(bias, dtype=np.float32)).astype(np.float32)
    return _activation(rst, act, slope)


def graphconv_torch(src, dst, n, x, weight, bias, norm="both", act=None, slope=0.01,
                    edge_weight=None):
    """The same layer in differentiable torch ops (index_add_), for gradient parity.

    x / weight / bias / edge_weight are torch tensors (requires_grad as wanted)."""
    import torch

    src_t = torch.as_tensor(np.asarray(src), dtype=torch.long)
    dst_t = torch.as_tensor(np.asarray(dst), dtype=torch.long)
    feat = x
    if norm in ("left", "both"):
        degs = torch.bincount(src_t, minlength=n).float().clamp(min=1)
        nrm = degs.pow(-0.5) if norm == "both" else 1.0 / degs
        feat = feat * nrm[:, None]

    def agg(f):
        msg = f[src_t]
        if edge_weight is not None:
            msg = msg * edge_weight[:, None]
        return torch.zeros(n, f.shape[1], dtype=f.dtype).index_add_(0, dst_t, msg)

    if weight is not None and x.shape[1] > weight.shape[1]:
        rst = agg(feat @ weight)
    else:
        rst = agg(feat)
        if weight is not None:
            rst = rst @ weight
    if norm in ("right", "both"):
        degs = torch.bincount(dst_t, minlength=n).float().clamp(min=1)
        nrm = degs.pow(-0.5) if norm == "both" else 1.0 / degs
        rst = rst * nrm[:, None]
    if bias is not None:
        rst = rst + bias
    if act == "relu":
        rst = torch.relu(rst)
    elif act == "leaky_relu":
        rst = torch.nn.functional.leaky_relu(rst, slope)
    return rst


def sgconv_torch(src, dst, n, x, weight, bias, k=1):
    """DGL 0.8 SGConv.forward in differentiable torch ops: k hops of
    feat*norm -> sum aggregation -> feat*norm with norm = in_degree.clamp(1)^-1/2, then the
    Linear (weight [out,in]).  Wrapped by the reference's SGCN (staticGraphEncoder.py:135)."""
    import torch

    src_t = torch.as_tensor(np.asarray(src), dtype=torch.long)
    dst_t = torch.as_tensor(np.asarray(dst), dtype=torch.long)
    nrm = torch.bincount(dst_t, minlength=n).float().clamp(min=1).pow(-0.5)[:, None]
    feat = x
    for _ in range(k):
        feat = feat * nrm
        feat = torch.zeros(n, feat.shape[1], dtype=feat.dtype).index_add_(0, dst_t, feat[src_t])
        feat = feat * nrm
    out = feat @ weight.t()
    return out if bias is None else out + bias


def sageconv_gcn_torch(src, dst, n, x, weight, bias, act=None, slope=0.01):
    """DGL 0.8 SAGEConv.forward, aggregator 'gcn', in differentiable torch ops:
    h_neigh = (sum of in-neighbours + self) / (in_degree + 1); fc_neigh (weight [out,in], no
    bias of its own) runs first iff in > out; then + bias, activation."""
    import torch

    src_t = torch.as_tensor(np.asarray(src), dtype=torch.long)
    dst_t = torch.as_tensor(np.asarray(dst), dtype=torch.long)
    before = x.shape[1] > weight.shape[0]
    h = x @ weight.t() if before else x
    neigh = torch.zeros(n, h.shape[1], dtype=h.dtype).index_add_(0, dst_t, h[src_t])
    degs = torch.bincount(dst_t, minlength=n).to(h.dtype)
    h_neigh = (neigh + h) / (degs[:, None] + 1)
    if not before:
        h_neigh = h_neigh @ weight.t()
    rst = h_neigh if bias is None else h_neigh + bias
    if act == "relu":
        rst = torch.relu(rst)
    elif act == "leaky_relu":
        rst = torch.nn.functional.leaky_relu(rst, slope)
    return rst


def sageconv_mean_torch(src, dst, n, x, w_self, w_neigh, bias, act=None, slope=0.01):
    """DGL 0.8 SAGEConv.forward, aggregator 'mean', in differentiable torch ops:
    h_neigh = mean of the in-neighbours (0 without in-edges); fc_neigh (no bias of its own) runs
    first iff in > out; rst = fc_self(x) + h_neigh + bias, activation."""
    import torch

    src_t = torch.as_tensor(np.asarray(src), dtype=torch.long)
    dst_t = torch.as_tensor(np.asarray(dst), dtype=torch.long)
    before = x.shape[1] > w_neigh.shape[0]
    h = x @ w_neigh.t() if before else x
    neigh = torch.zeros(n, h.shape[1], dtype=h.dtype).index_add_(0, dst_t, h[src_t])
    degs = torch.bincount(dst_t, minlength=n).to(h.dtype).clamp(min=1)
    h_neigh = neigh / degs[:, None]
    if not before:
        h_neigh = h_neigh @ w_neigh.t()
    rst = x @ w_self.t() + h_neigh
    if bias is not None:
        rst = rst + bias
    if act == "relu":
        rst = torch.relu(rst)
    elif act == "leaky_relu":
        rst = torch.nn.functional.leaky_relu(rst, slope)
    return rst


def gatconv_torch(src, dst, n, x, fc_weight, attn_l, attn_r, bias, heads, negative_slope=0.2,
                  act=None, slope=0.01, res_weight=None, residual=False):
    """DGL 0.8 GATConv.forward in differentiable torch index ops (GATConv is the layer the
    reference's GAT encoder stacks, staticGraphEncoder.py:170,174): ft = fc(x) viewed
    [N,H,D]; el/er = (ft * attn).sum(-1); e = LeakyReLU(el[src] + er[dst]); a = softmax of e
    over the incoming edges of each destination; out = sum a * ft[src] (+ residual) + bias,
    activation.  Returns (out [N,H,D], a [E,H,1] in edge order)."""
    import torch

    src_t = torch.as_tensor(np.asarray(src), dtype=torch.long)
    dst_t = torch.as_tensor(np.asarray(dst), dtype=torch.long)
    d = fc_weight.shape[0] // heads
    ft = (x @ fc_weight.t()).view(-1, heads, d)
    el = (ft * attn_l).sum(-1)
    er = (ft * attn_r).sum(-1)
    e = torch.nn.functional.leaky_relu(el[src_t] + er[dst_t], negative_slope)  # [E,H]
    idx = dst_t[:, None].expand(-1, heads)
    m = torch.full((n, heads), -torch.inf, dtype=e.dtype).scatter_reduce(0, idx, e.detach(), "amax",
                                                                         include_self=True)
    p = torch.exp(e - m[dst_t])
    denom = torch.zeros(n, heads, dtype=e.dtype).index_add_(0, dst_t, p)
    a = p / denom[dst_t]
    out = torch.zeros(n, heads, d, dtype=ft.dtype).index_add_(0, dst_t, a[:, :, None] * ft[src_t])
    if residual:
        res = x if res_weight is None else x @ res_weight.t()
        out = out + res.view(n, heads, d)
    if bias is not None:
        out = out + bias.view(1, heads, d)
    if act == "relu":
        out = torch.relu(out)
    elif act == "leaky_relu":
        out = torch.nn.functional.leaky_relu(out, slope)
    return out, a[:, :, None]


def evolve_gcn_o_torch(snapshots, n, x, init_weights, grus, biases, act="leaky_relu",
                       detach_state=True):
    """EvolveGCN-O over a list of snapshots [(src, dst), ...] in differentiable torch ops.
    Per snapshot and layer: W_t = GRU(input=W_{t-1}, h0=W_{t-1}) with rows as batch
    (`grus[l]` is a torch.nn.GRU(cols, cols, batch_first=True)), then GraphConv with the
    external weight W_t (graphconv_torch).  Not in the reference; this IS the definition
    the GPU path is held to.  Returns the list of per-snapshot outputs."""
    state = list(init_weights)
    outs = []
    for src, dst in snapshots:
        new = []
        for w, gru in zip(state, grus):
            o, _ = gru(w[:, None, :], w[None].contiguous())
            new.append(o[:, 0, :])
        h = x
        for w, b in zip(new, biases):
            h = graphconv_torch(src, dst, n, h, w, b, act=act)
        outs.append(h)
        state = [w.detach() for w in new] if detach_state else new
    return outs


# ----------------------------------------------------------------------------- container
def is_sorted(a: np.ndarray) -> bool:
    """spiro/core/utils.py:24-33."""
    a = np.asarray(a)
    return bool(np.all(a[:-1] <= a[1:]))


def get_anchor_position(arr: np.ndarray, anchors) -> np.ndarray:
    """spiro/core/utils.py:36-55: last position (1-based) of each anchor in a sorted array."""
    pos = np.searchsorted(arr, anchors)[1:]
    return np.hstack((pos, [arr.shape[0]]))


def dispatch(edges: np.ndarray, nodes: np.ndarray, n_timestamps: int, t: int):
    """VEInvariantDTDG.dispatcher (spiro/dtdg/types.py:183-198): cumulative prefix of the
    time-sorted tables.  Returns (src, dst, num_nodes, node_feat or None, edge_feat or None)."""
    na = get_anchor_position(nodes[:, 0], range(n_timestamps))
    ea = get_anchor_position(edges[:, 0], range(n_timestamps))
    e = edges[: ea[t]]
    v = nodes[: na[t]]
    src = e[:, 1].astype(np.int64)
    dst = e[:, 2].astype(np.int64)
    nf = v[:, 1:].astype(np.float32) if v.shape[1] > 1 else None
    ef = e[:, 3:].astype(np.float32) if e.shape[1] > 3 else None
    return src, dst, v.shape[0], nf, ef


def citation_dispatch(edges: np.ndarray, nodes: np.ndarray, n_timestamps: int, t: int):
    """CitationGraph.dispatcher (spiro/dtdg/types.py:230-248), including its quirk: the
    feature matrix that already lost its time column is sliced by node_dimension again
    (:244-245), so the first real feature is dropped and the last column is the in-degree
    of snapshot t-1, zero-padded for nodes that did not exist yet."""
    src, dst, n, nf, ef = dispatch(edges, nodes, n_timestamps, t)
    citation = np.zeros((n, 1))
    if t > 0:
        _, pdst, pn, _, _ = dispatch(edges, nodes, n_timestamps, t - 1)
        citation[:pn, 0] = degree(pdst, pn)
    this_nodes = np.hstack((nf.astype(np.float64), citation))
    feat = this_nodes[:, 1:].astype(np.float32)
    return src, dst, n, feat, ef


def citation_tables_loops(graph_data, node_time: str, edges_key="edge_index", nodes_key="node_feat"):
    """spiro/dtdg/dataloader.py:50-70 step by step (the per-edge loop and the id dictionary
    included): (edges [t, src, dst], nodes [t, feat...], timestamps).  Parity unpinned: the
    reference's test of this function downloads ogbn-arxiv (tests/dtdg/test_dataloader_torch.py)
    and holds no fixture; what pins this restatement is the reading of those twenty lines."""
    timestamps = np.unique(graph_data[node_time])
    t_index = np.searchsorted(timestamps, graph_data[node_time][:, 0])
    n = graph_data[nodes_key].shape[0]
    t_index = np.reshape(t_index, (n, 1))
    node_id = np.reshape(np.array(range(n)), (n, 1))
    nodes = np.hstack((node_id, t_index, graph_data[nodes_key]))
    nodes = nodes[nodes[:, 1].argsort()]
    relabel = {old: new for old, new in zip(nodes[:, 0], np.array(range(n)))}
    src_l, dst_l = [], []
    for i in range(len(graph_data[edges_key][0])):
        n1 = relabel[graph_data[edges_key][0][i]]
        n2 = relabel[graph_data[edges_key][1][i]]
        src_l.append(max(n1, n2))
        dst_l.append(min(n1, n2))
    sources, dest = np.array(src_l), np.array(dst_l)
    edge_time = np.array([nodes[s][1] for s in sources])
    edges = np.vstack((edge_time, sources, dest)).transpose()
    edges = edges[edges[:, 0].argsort()]
    return edges, nodes[:, 1:], timestamps


def prepare_citation_task(edges, nodes, n_timestamps, start_t=1, validating_snapshots=5,
                          minimum_citation=5):
    """spiro/automl/tf/prepare_dataset.py:97-122 on the numpy container: (targets, labels),
    dicts over t.  Parity unpinned: the reference's only test of it
    (tests/automl/batch_generator_test.py) counts batches and runs on TensorFlow."""
    feat = lambda t: citation_dispatch(edges, nodes, n_timestamps, t)
    targets, labels = {}, {}
    times, v = n_timestamps, validating_snapshots
    for t in range(start_t, times - v - 1):
        n = feat(t)[2]
        later = feat(t + v)[3]
        target = np.where(later[np.arange(n), -1] > minimum_citation)[0]
        new_citation = feat(t + 1)[3][:n, -1]
        targets[t] = target.astype(np.int32)
        labels[t] = new_citation[target]
    target = targets[times - 2 - v]
    for t in range(times - v - 1, times - 1):
        n = feat(t)[2]
        new_citation = feat(t + 1)[3][:n, -1]
        targets[t] = target
        labels[t] = new_citation[target]
    return targets, labels


# ----------------------------------------------------------------------------- recurrent cells
def _sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x.astype(np.float32)))).astype(np.float32)


def gru_cell(x, h, w_ih, w_hh, b_ih, b_hh):
    """torch.nn.GRU one step (gate order r,z,n), as run by SequentialDecoder.forward
    (implicitTimeModels.py:154) with seq_len 1."""
    hd = h.shape[1]
    gi = x @ w_ih.T + (0 if b_ih is None else b_ih)
    gh = h @ w_hh.T + (0 if b_hh is None else b_hh)
    r = _sigmoid(gi[:, :hd] + gh[:, :hd])
    z = _sigmoid(gi[:, hd:2 * hd] + gh[:, hd:2 * hd])
    n = np.tanh(gi[:, 2 * hd:] + r * gh[:, 2 * hd:]).astype(np.float32)
    return ((1 - z) * n + z * h).astype(np.float32)


def lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh):
    """torch.nn.LSTM one step (gate order i,f,g,o), LSTM.forward (implicitTimeModels.py:225)."""
    hd = h.shape[1]
    g = x @ w_ih.T + h @ w_hh.T + (0 if b_ih is None else b_ih) + (0 if b_hh is None else b_hh)
    i = _sigmoid(g[:, :hd])
    f = _sigmoid(g[:, hd:2 * hd])
    gg = np.tanh(g[:, 2 * hd:3 * hd]).astype(np.float32)
    o = _sigmoid(g[:, 3 * hd:])
    c2 = (f * c + i * gg).astype(np.float32)
    return (o * np.tanh(c2)).astype(np.float32), c2


def rnn_cell(x, h, w_ih, w_hh, b_ih, b_hh):
    """torch.nn.RNN(tanh) one step (implicitTimeModels.py:284-286)."""
    return np.tanh(x @ w_ih.T + h @ w_hh.T + (0 if b_ih is None else b_ih)
                   + (0 if b_hh is None else b_hh)).astype(np.float32)


def recurrent_step(kind: str, x, h0, params, c0=None):
    """Stack of n_layers cells with seq_len 1: input of layer k is h' of layer k-1.
    params[k] = (w_ih, w_hh, b_ih, b_hh).  Returns (out, h_new[L,M,H][, c_new])."""
    hs, cs = [], []
    inp = x.astype(np.float32)
    for k, (w_ih, w_hh, b_ih, b_hh) in enumerate(params):
        if kind == "gru":
            h = gru_cell(inp, h0[k], w_ih, w_hh, b_ih, b_hh)
        elif kind == "rnn":
            h = rnn_cell(inp, h0[k], w_ih, w_hh, b_ih, b_hh)
        elif kind == "lstm":
            h, c = lstm_cell(inp, h0[k], c0[k], w_ih, w_hh, b_ih, b_hh)
            cs.append(c)
        else:
            raise ValueError(kind)
        hs.append(h)
        inp = h
    if kind == "lstm":
        return inp, np.stack(hs), np.stack(cs)
    return inp, np.stack(hs)


def mlp_forward(x, layers, training: bool = False, eps: float = 1e-5):
    """MLP of torch/simpleDecoder.py:45-79 with dropout inactive.
    layers: list of dicts {'w','b'} for Linear and {'gamma','beta','mean','var'} for BatchNorm1d,
    in module order.  training=True uses batch statistics (biased variance)."""
    h = x.astype(np.float32)
    for ly in layers:
        if "w" in ly:
            h = (h @ ly["w"].T + ly["b"]).astype(np.float32)
        elif "gamma" in ly:
            if training:
                mean = h.mean(0)
                var = h.var(0)
            else:
                mean, var = ly["mean"], ly["var"]
            h = ((h - mean) / np.sqrt(var + eps) * ly["gamma"] + ly["beta"]).astype(np.float32)
            h = np.maximum(h, 0)  # the ReLU that follows every BatchNorm1d
    return h


# ----------------------------------------------------------------------------- edge task
def process_target_ids(ids: np.ndarray):
    """tf/sequentialDecoder/baseModel.py:32-46: rank-1 ids pass through; rank-2 (n_edges, 2)
    pairs become (distinct nodes, pairs re-indexed into that list).  The reference keeps
    first-occurrence order (tf.unique); the order only permutes rows that are re-indexed
    consistently, so sorted order gives the same per-edge result."""
    ids = np.asarray(ids)
    if ids.ndim == 1:
        return ids, ids
    if ids.ndim != 2:
        raise ValueError("ids must be of rank 1 or 2")
    nodes, inverse = np.unique(ids.reshape(-1), return_inverse=True)
    return nodes, inverse.reshape(-1, 2)


def edge_head_forward(kind, embs, pairs, mem_h, params, mlp_layers, mem_c=None, sigmoid=True):
    """Edge-task decoder forward (RnnFamily.call, tf/.../baseModel.py:154-164 + MLP.call,
    tf/simpleDecoder.py:90-97): recurrent step on the distinct nodes of `pairs` (memory rows
    read and written there), pair-wise concatenation, MLP (eval-mode BatchNorm), sigmoid.
    Returns (prediction [n_edges, out], new mem_h[, new mem_c])."""
    nodes, pid = process_target_ids(pairs)
    x = embs[nodes]
    if kind == "lstm":
        out, h, c = recurrent_step(kind, x, mem_h[:, nodes], params, mem_c[:, nodes])
        mem_c = mem_c.copy()
        mem_c[:, nodes] = c
    else:
        out, h = recurrent_step(kind, x, mem_h[:, nodes], params)
    mem_h = mem_h.copy()
    mem_h[:, nodes] = h
    z = out[pid.reshape(-1)].reshape(pid.shape[0], -1)
    y = mlp_forward(z, mlp_layers, training=False)
    if sigmoid:
        y = _sigmoid(y)
    return (y, mem_h, mem_c) if kind == "lstm" else (y, mem_h)


def mrr_numpy(pred_pos: np.ndarray, pred_neg: np.ndarray) -> float:
    """Mean reciprocal rank as ogb 1.3.3's `_eval_mrr` publishes it (the evaluator the reference
    calls at examples/tf/link_prediction_sliding_window.py:142-146): every positive against its
    own row of negatives, rank = 1 + mean of the optimistic and the pessimistic count of
    negatives scored above it.  ogb is not installable here: parity unpinned."""
    pos = np.asarray(pred_pos, dtype=np.float64).reshape(-1, 1)
    neg = np.asarray(pred_neg, dtype=np.float64)
    rank = 0.5 * ((neg > pos).sum(1) + (neg >= pos).sum(1)) + 1.0
    return float((1.0 / rank).mean())


def link_prediction_loop_torch(edges, nodes, n_timestamps, tasks, train_ts, eval_t, convs, kind, rnn_params,
                               mlp_layers, mem_h0, mem_c0=None, lr=1e-3, epochs=1, loss_order="bce",
                               eval_negatives_per_node=1000, positives_per_node=2, bn_momentum=0.1,
                               bn_eps=1e-5):
    """The link-prediction training loop of examples/tf/link_prediction_sliding_window.py:99-146
    on the torch path, restated with plain torch-CPU tensor ops (differentiable, float32):

      per epoch: node memory back to its initial value (:101, injected instead of re-drawn);
      for t in train_ts (:104): snapshot t (prefix of the edge table, types.py:183-198) minus the
      positive edges of its task (:105-106), GraphConv stack (leaky_relu), recurrent step on the
      distinct nodes of the target pairs with the detached state written back
      (tf/sequentialDecoder/baseModel.py:32-46,154-164), pair-wise concatenation + MLP with
      train-mode BatchNorm + sigmoid (tf/simpleDecoder.py:90-97), loss, Adam(lr, eps 1e-8) (:98);
      afterwards the held-out snapshot `eval_t` in inference mode (BatchNorm running statistics,
      memory not written) and the MRR of its positives against `eval_negatives_per_node`
      negatives per evaluated node (:128-146).

    loss_order: "bce" = binary cross-entropy of the prediction against the label (what the
    script means); "swapped" = the script as written (:109 passes (prediction, label) to a
    Keras loss whose signature is (y_true, y_pred)): -mean(p log l' + (1-p) log(1-l')) with the
    label clipped to [1e-7, 1-1e-7], which is linear in the prediction.

    tasks[t] = (pairs int64 [m, 2], labels float [m], ids of the positive edges in the table).
    convs = [(W [in,out], b)], rnn_params[k] = (w_ih, w_hh, b_ih, b_hh), mlp_layers = list of
    {'w','b'} / {'gamma','beta','mean','var'} dicts in module order (a ReLU follows every
    BatchNorm, a sigmoid the last Linear); all numpy, copied.  Returns a dict with the loss of
    every step, the gradients of the first step, the final parameters, the evaluation
    predictions and the MRR."""
    import torch

    def leaf(a):
        return torch.tensor(np.asarray(a, dtype=np.float32), requires_grad=True)

    convs_t = [(leaf(w), leaf(b)) for w, b in convs]
    rnn_t = [tuple(leaf(a) for a in layer) for layer in rnn_params]
    mlp_t = []
    for ly in mlp_layers:
        if "w" in ly:
            mlp_t.append({"w": leaf(ly["w"]), "b": leaf(ly["b"])})
        else:
            mlp_t.append({"gamma": leaf(ly["gamma"]), "beta": leaf(ly["beta"]),
                          "mean": torch.tensor(np.asarray(ly["mean"], dtype=np.float32)),
                          "var": torch.tensor(np.asarray(ly["var"], dtype=np.float32))})
    params = [t for pair in convs_t for t in pair] + [t for layer in rnn_t for t in layer] + \
             [ly[k] for ly in mlp_t for k in ("w", "b", "gamma", "beta") if k in ly]
    opt = torch.optim.Adam(params, lr=lr, eps=1e-8)

    def cell(x, h, c, w_ih, w_hh, b_ih, b_hh):
        hd = h.shape[1]
        gi, gh = x @ w_ih.T + b_ih, h @ w_hh.T + b_hh
        if kind == "gru":
            r = torch.sigmoid(gi[:, :hd] + gh[:, :hd])
            z = torch.sigmoid(gi[:, hd:2 * hd] + gh[:, hd:2 * hd])
            n = torch.tanh(gi[:, 2 * hd:] + r * gh[:, 2 * hd:])
            return (1 - z) * n + z * h, None
        if kind == "rnn":
            return torch.tanh(gi + gh), None
        g = gi + gh
        i, f = torch.sigmoid(g[:, :hd]), torch.sigmoid(g[:, hd:2 * hd])
        c2 = f * c + i * torch.tanh(g[:, 2 * hd:3 * hd])
        return torch.sigmoid(g[:, 3 * hd:]) * torch.tanh(c2), c2

    def head(z, training):
        h = z
        for ly in mlp_t:
            if "w" in ly:
                h = h @ ly["w"].T + ly["b"]
            else:
                if training:
                    mean, var = h.mean(0), h.var(0, unbiased=False)
                    with torch.no_grad():
                        ly["mean"] = (1 - bn_momentum) * ly["mean"] + bn_momentum * mean
                        ly["var"] = (1 - bn_momentum) * ly["var"] + bn_momentum * h.var(0, unbiased=True)
                else:
                    mean, var = ly["mean"], ly["var"]
                h = torch.relu((h - mean) / torch.sqrt(var + bn_eps) * ly["gamma"] + ly["beta"])
        return torch.sigmoid(h)

    def forward(t, mem_h, mem_c, training):
        pairs, _, positive = tasks[t]
        src, dst, n, feat, _ = dispatch(edges, nodes, n_timestamps, t)
        keep = np.ones(len(src), dtype=bool)
        keep[np.asarray(positive, dtype=np.int64)] = False
        h = torch.tensor(np.asarray(feat, dtype=np.float32))
        for w, b in convs_t:
            h = graphconv_torch(src[keep], dst[keep], n, h, w, b, act="leaky_relu")
        ids, pid = process_target_ids(np.asarray(pairs))
        ids_t = torch.from_numpy(ids.astype(np.int64))
        x = h[ids_t]
        new_h, new_c = [], []
        for k, layer in enumerate(rnn_t):
            x, c = cell(x, mem_h[k, ids_t], None if mem_c is None else mem_c[k, ids_t], *layer)
            new_h.append(x.detach())
            new_c.append(None if c is None else c.detach())
        if training:
            for k in range(len(rnn_t)):
                mem_h[k, ids_t] = new_h[k]
                if mem_c is not None:
                    mem_c[k, ids_t] = new_c[k]
        z = x[torch.from_numpy(pid.reshape(-1).astype(np.int64))].reshape(pid.shape[0], -1)
        return head(z, training).squeeze(-1)

    def loss_of(p, label):
        y = torch.tensor(np.asarray(label, dtype=np.float32))
        if loss_order == "swapped":
            yc = y.clamp(1e-7, 1 - 1e-7)
            return -(p * torch.log(yc) + (1 - p) * torch.log(1 - yc)).mean()
        return -(y * torch.log(p).clamp(min=-100) + (1 - y) * torch.log(1 - p).clamp(min=-100)).mean()

    losses, first_grads = [], None
    for _ in range(epochs):
        mem_h = torch.tensor(np.asarray(mem_h0, dtype=np.float32))
        mem_c = None if mem_c0 is None else torch.tensor(np.asarray(mem_c0, dtype=np.float32))
        for t in train_ts:
            opt.zero_grad()
            loss = loss_of(forward(t, mem_h, mem_c, True), tasks[t][1])
            loss.backward()
            if first_grads is None:
                first_grads = [q.grad.detach().numpy().copy() for q in params]
            opt.step()
            losses.append(float(loss.detach()))
    with torch.no_grad():
        predict = forward(eval_t, mem_h, mem_c, False).numpy()
    n_pos = len(tasks[eval_t][2])
    n_eval = n_pos // positives_per_node
    neg = predict[n_pos:].reshape(n_eval, eval_negatives_per_node)
    return {"losses": losses, "first_grads": first_grads,
            "params": [q.detach().numpy() for q in params],
            "eval_predict": predict, "eval_loss": float(loss_of(torch.from_numpy(predict), tasks[eval_t][1])),
            "mrr": mrr_numpy(predict[:n_pos], np.repeat(neg, positives_per_node, axis=0)),
            "mem_h": mem_h.numpy()}


def neg_sampling(pos_inds, n_items, n_samp, rng=np.random):
    """spiro/core/dataio.py:29-35 (pos_inds sorted)."""
    pos_inds = np.asarray(pos_inds)
    raw = rng.randint(0, n_items - len(pos_inds), size=n_samp)
    return raw + np.searchsorted(pos_inds - np.arange(len(pos_inds)), raw, side="right")


# ----------------------------------------------------------------------------- synthetic data
def powerlaw_snapshot(n: int, e: int, t: int = 0, k_dst: float = 3.0, k_src: float = 1.5,
                      perm: Optional[np.ndarray] = None):
    """Synthetic power-law snapshot: dst = pi(floor(n*u1^k_dst)), src = pi(floor(n*u2^k_src)),
    u ~ U[0,1) from PCG64(1234+t); pi is a fixed permutation (seed 99).  Multi-edges and
    self-loops are kept, as DGL keeps them."""
    rng = np.random.Generator(np.random.PCG64(1234 + t))
    if perm is None:
        perm = np.random.Generator(np.random.PCG64(99)).permutation(n)
    dst = perm[np.minimum((n * rng.random(e) ** k_dst).astype(np.int64), n - 1)]
    src = perm[np.minimum((n * rng.random(e) ** k_src).astype(np.int64), n - 1)]
    return src.astype(np.int64), dst.astype(np.int64)
