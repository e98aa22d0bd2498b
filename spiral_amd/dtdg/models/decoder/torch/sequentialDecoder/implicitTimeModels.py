"""Sequential decoders with per-node memory; API of
spiro/dtdg/models/decoder/torch/sequentialDecoder/implicitTimeModels.py:28-286.

The time recurrence runs across snapshot calls through NodeMemory (sequence length is
always 1 inside a call).  Here the memory lives on the GPU and the gate kernels read and
overwrite its rows in place; parameters are held by a torch.nn.GRU/LSTM/RNN module so
names, shapes and initialisation match the reference's state dict, but the step itself is
computed by libspmi (library GEMMs + fused gate kernels).
"""
from __future__ import annotations

import copy
import weakref
from typing import List, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from spiral_amd import ops
from spiral_amd.dtdg.models.decoder.torch.simpleDecoder import SimpleDecoder


_identity_verdicts: dict = {}  # id(tensor) -> (weakref, version, data_ptr, numel, verdict)


def _is_identity(ids, n: int) -> bool:
    """True when ids is arange(n) (or None, which means the same).  Tensors made by
    `dispatcher` carry a tag and CPU tensors are compared on the host; any other GPU tensor
    is compared with arange on the device once (one small kernel and one round trip) and the
    verdict is kept for as long as that tensor object lives unchanged, so a hand-made
    `torch.arange(n, device=...)` takes the same kernels as the dispatcher's ids."""
    if ids is None:
        return True
    if not isinstance(ids, torch.Tensor):
        return len(ids) == n and list(ids) == list(range(n))
    if ids.dim() != 1 or ids.numel() != n:
        return False
    if getattr(ids, "_spmi_identity", False):
        return True
    if ids.device.type == "cpu":
        return bool(torch.equal(ids, torch.arange(n, dtype=ids.dtype)))
    slot = _identity_verdicts.get(id(ids))
    state = (ids._version, ids.data_ptr(), n)
    if slot is not None and slot[0]() is ids and slot[1] == state:
        return slot[2]
    verdict = bool(torch.equal(ids, torch.arange(n, dtype=ids.dtype, device=ids.device)))
    ref = weakref.ref(ids, lambda _r, k=id(ids): _identity_verdicts.pop(k, None))
    _identity_verdicts[id(ids)] = (ref, state, verdict)
    return verdict


class NodeMemory:
    """(n_layers, n_nodes, hidden_d) table of hidden states, one row per node."""

    def __init__(self, n_nodes: int, hidden_d: int, n_layers: int, device=None):
        self.n_nodes, self.hidden_d, self.n_layers = n_nodes, hidden_d, n_layers
        self._device = torch.device(device) if device is not None else torch.device("cpu")
        self.reset_state()

    # Largest state magnitude for which a layer's output may be handed to the next layer as
    # "bounded" (|x| < 2^15, the recurrent kernels then skip their magnitude pass over x): a new
    # state is a convex combination of a gate value in [-1, 1] and the old state, so its
    # magnitude never exceeds max(1, |old state|).
    X_BOUND = 2.0 ** 15

    def reset_state(self):
        """Fresh standard-normal state, drawn on the CPU generator like the reference
        (implicitTimeModels.py:46) so that a seeded run yields the same values."""
        self.memory = torch.randn(self.n_layers, self.n_nodes, self.hidden_d).to(self._device)
        self._bounded = True  # standard normal draws
        self._spare = None

    @property
    def memory(self) -> torch.Tensor:
        """The (n_layers, n_nodes, hidden_d) table.  After tracked whole-table steps the state
        lives in one matrix per layer (`set_layers`); the first reader of the table stacks them
        into a table of its own, so nothing written to the table reaches tensors an autograd
        graph still holds."""
        if self._layers is not None:
            self._memory = torch.stack(self._layers)
            self._layers = None
        return self._memory

    @memory.setter
    def memory(self, table: torch.Tensor):
        """A table assigned from outside (a restored checkpoint, a test fixture) has unknown
        magnitudes until `bounded()` looks at it."""
        self._memory = table
        self._layers = None
        self._bounded = None

    def layer(self, k: int) -> torch.Tensor:
        """The [n_nodes, hidden_d] state of layer k, to be READ only: it may be the output of a
        tracked step that an autograd graph holds.  Writers go through `memory`."""
        return self._layers[k] if self._layers is not None else self._memory[k]

    def rows(self) -> int:
        return (self._layers[0] if self._layers is not None else self._memory).shape[-2]

    def set_layers(self, layers) -> None:
        """The new state of every node, one detached [n_nodes, hidden_d] matrix per layer,
        becomes the memory as it is (no copy).  The matrices are never written afterwards: the
        whole-table steps write their results elsewhere and everything else sees a stacked copy
        (`memory`).  Boundedness carries over (see X_BOUND)."""
        self._layers = [t.detach() for t in layers]
        self._memory = None

    def bounded(self, wait: bool = True) -> bool:
        """Whether every stored state is below X_BOUND in magnitude.  The cell kernels keep the
        property (see X_BOUND), so the question only comes up after the table was assigned or
        written through `update_memory`.  `wait=True` answers it with one device round trip.
        `wait=False` never makes the host wait: the maximum is queued to pinned memory once and
        the answer is "no" (callers then take the path that assumes nothing) until a later
        call finds it landed."""
        if self._bounded is None:
            m = self.memory
            if m.numel() == 0:
                self._bounded = True
            elif wait or not m.is_cuda:
                self._bounded = bool(float(m.abs().max()) < self.X_BOUND)
                self._bound_check = None
            else:
                check = getattr(self, "_bound_check", None)
                if check is None or check[0] is not m:
                    host = torch.empty(1, dtype=torch.float32, pin_memory=True)
                    host.copy_(m.abs().max().reshape(1), non_blocking=True)
                    event = torch.cuda.Event()
                    event.record()
                    check = self._bound_check = (m, host, event)
                if not check[2].query():
                    return False
                self._bounded = bool(float(check[1][0]) < self.X_BOUND)
                self._bound_check = None
        return self._bounded

    def spare(self) -> torch.Tensor:
        """A second table of the memory's shape and device (contents undefined), for steps
        that write the new state of every node beside the old one."""
        sp = getattr(self, "_spare", None)
        if sp is None or sp.shape != self.memory.shape or sp.device != self.memory.device:
            sp = self._spare = torch.empty_like(self.memory)
        return sp

    def swap_spare(self):
        """The spare table holds the new state of every node: it becomes the memory."""
        self._memory, self._spare = self._spare, self._memory

    def update_memory(self, new_memory: torch.Tensor, inx):
        self.memory[:, inx, :] = new_memory.to(self.memory.device)
        self._bounded = None
        self._bound_check = None   # a check in flight looked at the table before this write

    def get_memory(self, inx) -> torch.Tensor:
        return self.memory[:, inx, :]

    @property
    def device(self):
        return (self._layers[0] if self._layers is not None else self._memory).device

    def to(self, device, **kwargs):
        device = torch.device(device) if device is not None else None
        if device is None or self.device == device:
            return self
        ret = copy.copy(self)
        ret._device = device
        bounded = self._bounded
        ret.memory = self.memory.to(device, **kwargs)
        ret._bounded = bounded
        ret._spare = None
        return ret


class SequentialDecoder(ops.WeightCacheHooks, nn.Module):
    """Gathers embeddings and state by ids, runs one recurrent step, writes the detached
    state back (only in training mode), then applies the simple decoder."""

    _cell_kind = ""

    def __init__(self, hidden_d: int, n_nodes: int, n_layers: int, simple_decoder: SimpleDecoder):
        super().__init__()
        self.n_nodes = n_nodes
        self.hidden_d = hidden_d
        self.n_layers = n_layers
        self.mini_batch = False
        self.simple_decoder = simple_decoder
        self.memory_h = NodeMemory(n_nodes, hidden_d, n_layers)
        self.training_mode = True
        self.base_model = None

    def set_mini_batch(self, mini_batch: bool = True):
        self.mini_batch = mini_batch

    @property
    def device(self):
        return next(self.base_model.parameters()).device

    def _memories(self) -> List[str]:
        return ["memory_h"]

    def to(self, device=None, **kwargs):
        """Moves parameters and node memories; returns self (torch.nn.Module semantics)."""
        super().to(device, **kwargs)
        for name in self._memories():
            setattr(self, name, getattr(self, name).to(device))
        return self

    def cuda(self, device=None):
        return self.to(torch.device("cuda" if device is None else device))

    def training(self):  # noqa: D401
        """The reference's name for `train_mode` (implicitTimeModels.py:132-134).  On an instance
        it is hidden by the boolean `training` attribute that torch.nn.Module sets, there as here,
        so `decoder.training()` raises TypeError in both; call `train_mode()`."""
        self.training_mode = True

    def train_mode(self):
        """Write the new state back to the node memory on every step (the default)."""
        self.training_mode = True

    def eval_mode(self):
        self.training_mode = False

    def reset_memory_state(self):
        for name in self._memories():
            getattr(self, name).reset_state()

    def _check_base_model(self):
        bm = self.base_model
        if bm.bidirectional or getattr(bm, "proj_size", 0):
            raise NotImplementedError("bidirectional / projected recurrent layers are not supported")

    def _layer_params(self, k: int):
        bm = self.base_model
        w_ih, w_hh = getattr(bm, f"weight_ih_l{k}"), getattr(bm, f"weight_hh_l{k}")
        b_ih = getattr(bm, f"bias_ih_l{k}") if bm.bias else None
        b_hh = getattr(bm, f"bias_hh_l{k}") if bm.bias else None
        return w_ih, w_hh, b_ih, b_hh

    def _step(self, x: torch.Tensor, ids: Optional[torch.Tensor]) -> torch.Tensor:
        raise NotImplementedError

    def forward(self, in_state: Tuple[torch.Tensor, List[int]]):
        node_embs, ids = in_state
        if isinstance(ids, torch.Tensor) and ids.dim() == 2:
            return self._forward_edges(node_embs, ids)
        ids_t = None
        if not _is_identity(ids, node_embs.shape[0]) or node_embs.shape[0] > self.n_nodes:
            ids_t = torch.as_tensor(ids, dtype=torch.int64, device=node_embs.device)
            if not self.mini_batch:
                node_embs = node_embs[ids_t]
        for name in self._memories():
            mem = getattr(self, name)
            if mem.device != node_embs.device:
                setattr(self, name, mem.to(node_embs.device))
        out = self._step(node_embs.reshape(-1, self.input_d), ids_t)
        return self.simple_decoder((out.view(-1, self.hidden_d), ids))

    def _forward_edges(self, node_embs: torch.Tensor, pairs: torch.Tensor):
        """Edge task: `pairs` is (n_edges, 2) node ids.  The recurrent step runs once per
        distinct node of the pairs (state read and written at those rows), and the simple
        decoder receives the pairs re-indexed into that list, as the reference's
        TensorFlow decoders do (tf/sequentialDecoder/baseModel.py:32-46,154-164)."""
        pairs = pairs.to(node_embs.device)
        nodes, inverse = torch.unique(pairs.reshape(-1), sorted=True, return_inverse=True)
        for name in self._memories():
            mem = getattr(self, name)
            if mem.device != node_embs.device:
                setattr(self, name, mem.to(node_embs.device))
        out = self._step(node_embs.index_select(0, nodes).reshape(-1, self.input_d), nodes)
        return self.simple_decoder((out.view(-1, self.hidden_d), inverse.reshape(-1, 2)))

    def _between_layers(self, x: torch.Tensor, k: int) -> torch.Tensor:
        bm = self.base_model
        if bm.dropout > 0 and bm.training and k < self.n_layers - 1:
            x = F.dropout(x, bm.dropout, training=True)
        return x


class GRU(SequentialDecoder):
    _cell_kind = "gru"

    def __init__(self, input_d: int, hidden_d: int, n_nodes: int, n_layers: int,
                 simple_decoder: SimpleDecoder, **kwargs):
        super().__init__(hidden_d, n_nodes, n_layers, simple_decoder)
        self.input_d = input_d
        self.base_model = nn.GRU(input_size=input_d, hidden_size=hidden_d, batch_first=True,
                                 num_layers=n_layers, **kwargs)
        self._check_base_model()

    def _step(self, x, ids):
        mem = self.memory_h
        # Every node, nothing tracked: the kernel that keeps the gate weights in LDS.  It
        # cannot update rows in place, so the new state goes to the memory's second table and
        # the two tables trade places afterwards; the returned rows are the new state itself.
        inner_bounded = mem.bounded(wait=False)  # layers k > 0 read the state of layer k - 1
        every_node = (ids is None and x.shape[0] == mem.rows()
                      and self._layer_params(0)[0].shape[1] == x.shape[1])
        params = [self._layer_params(k) for k in range(self.n_layers)]
        tracked = torch.is_grad_enabled() and (x.requires_grad or any(
            t is not None and t.requires_grad for ps in params for t in ps))
        if every_node and tracked and all(ops.gru_rows_tracked_ok(x if k == 0 else mem.layer(k), params[k][1])
                                          for k in range(self.n_layers)):
            # Every node, under autograd: the same kernel with the gate values kept.  Each layer's
            # new state is a tensor of its own (the graph's), and the memory adopts it as it is.
            new = []
            for k in range(self.n_layers):
                h = ops.gru_rows_tracked(x, *params[k], mem.layer(k), x_bounded=k > 0 and inner_bounded)
                new.append(h)
                x = self._between_layers(h, k)
            if self.training_mode:
                mem.set_layers(new)
            return x
        whole = (every_node and not tracked
                 and all(ops.gru_rows_ok(x if k == 0 else mem.memory[k], self._layer_params(k)[1],
                                         *self._layer_params(k)[::2], self._layer_params(k)[3])
                         for k in range(self.n_layers)))
        if whole:
            new = mem.spare() if self.training_mode else None
            for k in range(self.n_layers):
                out = new[k] if new is not None else torch.empty_like(mem.memory[k])
                x = ops.gru_rows(x, *self._layer_params(k), mem.memory[k], out,
                                 x_bounded=k > 0 and inner_bounded)
                x = self._between_layers(x, k)
            if new is not None:
                mem.swap_spare()
            return x
        for k in range(self.n_layers):
            x = ops.gru_cell(x, *self._layer_params(k), mem.memory[k], ids,
                             mem_write=self.training_mode, x_bounded=k > 0 and inner_bounded)
            x = self._between_layers(x, k)
        return x


class RNN(SequentialDecoder):
    _cell_kind = "rnn"

    def __init__(self, input_d: int, hidden_d: int, n_nodes: int, n_layers: int,
                 simple_decoder: SimpleDecoder, **kwargs):
        super().__init__(hidden_d, n_nodes, n_layers, simple_decoder)
        self.input_d = input_d
        if kwargs.get("nonlinearity", "tanh") != "tanh":
            raise NotImplementedError("only the tanh RNN cell is supported")
        self.base_model = nn.RNN(input_size=input_d, hidden_size=hidden_d, batch_first=True,
                                 num_layers=n_layers, **kwargs)
        self._check_base_model()

    def _step(self, x, ids):
        mem = self.memory_h
        every_node = (ids is None and x.shape[0] == mem.rows()
                      and self._layer_params(0)[0].shape[1] == x.shape[1])
        params = [self._layer_params(k) for k in range(self.n_layers)]
        tracked = torch.is_grad_enabled() and (x.requires_grad or any(
            t is not None and t.requires_grad for ps in params for t in ps))
        if every_node and tracked and all(ops.rnn_rows_tracked_ok(x if k == 0 else mem.layer(k), params[k][1])
                                          for k in range(self.n_layers)):
            # every node, under autograd: the same kernel, its output kept by the graph and
            # adopted by the memory as it is (see GRU._step)
            new = []
            for k in range(self.n_layers):
                h = ops.rnn_rows_tracked(x, *params[k], mem.layer(k), x_bounded=k > 0)
                new.append(h)
                x = self._between_layers(h, k)
            if self.training_mode:
                mem.set_layers(new)
            return x
        # every node, nothing tracked: the kernel that keeps the weights in LDS (see GRU._step)
        whole = (every_node and not tracked
                 and all(ops.rnn_rows_ok(x if k == 0 else mem.memory[k], self._layer_params(k)[1],
                                         *self._layer_params(k)[::2], self._layer_params(k)[3])
                         for k in range(self.n_layers)))
        if whole:
            new = mem.spare() if self.training_mode else None
            for k in range(self.n_layers):
                out = new[k] if new is not None else torch.empty_like(mem.memory[k])
                x = ops.rnn_rows(x, *self._layer_params(k), mem.memory[k], out, x_bounded=k > 0)
                x = self._between_layers(x, k)
            if new is not None:
                mem.swap_spare()
            return x
        for k in range(self.n_layers):
            # tanh output: bounded by 1 whatever the state holds
            x = ops.rnn_cell(x, *self._layer_params(k), self.memory_h.memory[k], ids,
                             mem_write=self.training_mode, x_bounded=k > 0)
            x = self._between_layers(x, k)
        return x


class LSTM(SequentialDecoder):
    _cell_kind = "lstm"

    def __init__(self, input_d: int, hidden_d: int, n_nodes: int, n_layers: int,
                 simple_decoder: SimpleDecoder, **kwargs):
        super().__init__(hidden_d, n_nodes, n_layers, simple_decoder)
        self.input_d = input_d
        self.base_model = nn.LSTM(input_size=input_d, hidden_size=hidden_d, batch_first=True,
                                  num_layers=n_layers, **kwargs).float()
        self._check_base_model()
        self.memory_c = NodeMemory(n_nodes, hidden_d, n_layers)

    def _memories(self) -> List[str]:
        return ["memory_h", "memory_c"]

    def _step(self, x, ids):
        mh, mc = self.memory_h, self.memory_c
        every_node = (ids is None and x.shape[0] == mh.rows()
                      and self._layer_params(0)[0].shape[1] == x.shape[1])
        params = [self._layer_params(k) for k in range(self.n_layers)]
        tracked = torch.is_grad_enabled() and (x.requires_grad or any(
            t is not None and t.requires_grad for ps in params for t in ps))
        if every_node and tracked and all(ops.lstm_rows_tracked_ok(x if k == 0 else mh.layer(k), params[k][1])
                                          for k in range(self.n_layers)):
            # every node, under autograd: the same kernel with the gate values kept; both
            # memories adopt the layers' new states as they are (see GRU._step)
            new_h, new_c = [], []
            for k in range(self.n_layers):
                # h' = sigmoid(o) * tanh(c'): bounded by 1 whatever the state holds
                h, c = ops.lstm_rows_tracked(x, *params[k], mh.layer(k), mc.layer(k), x_bounded=k > 0)
                new_h.append(h)
                new_c.append(c)
                x = self._between_layers(h, k)
            if self.training_mode:
                mh.set_layers(new_h)
                mc.set_layers(new_c)
                mc._bounded = None  # c' = f c + i g may grow by one a step
            return x
        # Every node, nothing tracked: the kernel that keeps the gate weights in LDS (see
        # GRU._step).  The new hidden state goes to the memory's second table; the cell state
        # is updated in place when the step writes back, into a scratch table otherwise.
        whole = (every_node and not tracked
                 and all(ops.lstm_rows_ok(x if k == 0 else mh.memory[k], self._layer_params(k)[1],
                                          *self._layer_params(k)[::2], self._layer_params(k)[3])
                         for k in range(self.n_layers)))
        if whole:
            new_h = mh.spare() if self.training_mode else None
            for k in range(self.n_layers):
                h_out = new_h[k] if new_h is not None else torch.empty_like(mh.memory[k])
                c_out = mc.memory[k] if self.training_mode else torch.empty_like(mc.memory[k])
                # h' = sigmoid(o) * tanh(c'): bounded by 1 whatever the state holds
                x, _c = ops.lstm_rows(x, *self._layer_params(k), mh.memory[k], mc.memory[k], h_out, c_out,
                                      x_bounded=k > 0)
                x = self._between_layers(x, k)
            if new_h is not None:
                mh.swap_spare()
                mc._bounded = None
            return x
        for k in range(self.n_layers):
            # h' = sigmoid(o) * tanh(c'): bounded by 1 whatever the state holds
            x, _c = ops.lstm_cell(x, *self._layer_params(k), self.memory_h.memory[k],
                                  self.memory_c.memory[k], ids, mem_write=self.training_mode,
                                  x_bounded=k > 0)
            x = self._between_layers(x, k)
        return x
