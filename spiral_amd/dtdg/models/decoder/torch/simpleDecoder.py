"""Output heads; API of spiro/dtdg/models/decoder/torch/simpleDecoder.py:24-79.
Training runs the layers as torch modules (weight gradients on spmi_gemm_tn); inference on
the GPU runs the whole head as one libspmi kernel when its widths allow it."""
from abc import abstractmethod
from typing import List, Tuple

import torch
from torch import nn

from spiral_amd.nn.linear import TallLinear
from spiral_amd.ops import WeightCacheHooks, cache_epoch


class SimpleDecoder(WeightCacheHooks, nn.Module):
    def __init__(self):
        super().__init__()
        self.layers = nn.ModuleList()

    @abstractmethod
    def forward(self, in_sample: Tuple[torch.Tensor, List]) -> torch.Tensor:
        ...


_ACTIVATIONS = {"linear": None, None: None, "sigmoid": nn.Sigmoid, "relu": nn.ReLU, "tanh": nn.Tanh}


class MLP(SimpleDecoder):
    """[Linear, BatchNorm1d, ReLU, Dropout] per entry of embed_dims, then Linear to
    output_dim.

    Node task: ids of rank 1 are not used (simpleDecoder.py:78-79).  Edge task: ids of
    rank 2, shape (n_edges, 2), index rows of the embedding; the two rows of a pair are
    concatenated before the first layer, so input_dim is twice the embedding width.  The
    reference implements the edge task, and the `activation` of the last layer, only in
    its TensorFlow MLP (spiro/dtdg/models/decoder/tf/simpleDecoder.py:45-97)."""

    def __init__(self, input_dim: int, embed_dims: List[int], dropout: float = 0.5, output_dim: int = 1,
                 activation: str = "linear"):
        super().__init__()
        if activation not in _ACTIVATIONS:
            raise ValueError(f"unsupported activation {activation!r}")
        width = input_dim
        for embed_dim in embed_dims:
            self.layers.extend([TallLinear(width, embed_dim), nn.BatchNorm1d(embed_dim), nn.ReLU(),
                                nn.Dropout(p=dropout)])
            width = embed_dim
        self.layers.append(TallLinear(width, output_dim))
        if _ACTIVATIONS[activation] is not None:
            self.layers.append(_ACTIVATIONS[activation]())
        self.mlp = nn.Sequential(*self.layers)

    def forward(self, in_state):
        node_embs, ids = in_state
        if isinstance(ids, torch.Tensor) and ids.dim() == 2:
            pairs = ids.to(node_embs.device).reshape(-1)
            node_embs = node_embs.index_select(0, pairs).reshape(ids.shape[0], -1)
        elif isinstance(ids, torch.Tensor) and ids.dim() > 2:
            from spiral_amd.core.errors import DimensionError

            raise DimensionError("The index to predict in the input must be of rank 1 for node prediction "
                                 f"or rank 2 for edge prediction. But get an input index of rank {ids.dim()}")
        if self._fused_eval_ok(node_embs):
            from spiral_amd import ops

            packed = self._packed_head(node_embs.device)
            return ops.mlp_head(node_embs, final_act=self._final_act, packed=packed)
        blocks = self._train_blocks(node_embs)
        if blocks is not None:
            return self._fused_train(node_embs, *blocks)
        return self.mlp(node_embs)

    # ---- training path on libspmi's row kernels --------------------------------------------
    # As torch modules the narrow layers of the head are some forty launches per step over
    # [rows, 5..20] matrices (3 ms of a 17 ms training step at a million rows); the fused path
    # runs a block per launch with the same arithmetic: batch statistics, running statistics,
    # and dropout factors drawn from torch's generator in the modules' order.
    FUSE_TRAIN = True

    def _train_blocks(self, x: torch.Tensor):
        """([(linear, batch_norm), ..., (linear, None)], [dropout p per block], final activation
        module or None) when the training forward can run fused, else None."""
        if (not self.FUSE_TRAIN or not self.training or not torch.is_grad_enabled() or not x.is_cuda
                or x.dtype != torch.float32 or x.dim() != 2):
            return None
        from spiral_amd import ops

        if x.shape[0] < ops.MLP_TRAIN_MIN_ROWS:
            return None
        mods, blocks, drops, i, final = list(self.layers), [], [], 0, None
        while i < len(mods):
            if not isinstance(mods[i], nn.Linear):
                if i != len(mods) - 1 or not blocks or blocks[-1][1] is not None:
                    return None
                final = mods[i]
                break
            lin, j = mods[i], i + 1
            if j < len(mods) and isinstance(mods[j], nn.BatchNorm1d):
                bn = mods[j]
                if not (j + 1 < len(mods) and isinstance(mods[j + 1], nn.ReLU)) or bn.momentum is None \
                        or not bn.training:
                    return None
                j += 2
                p = 0.0
                if j < len(mods) and isinstance(mods[j], nn.Dropout):
                    p = mods[j].p if mods[j].training else 0.0
                    j += 1
                blocks.append((lin, bn))
                drops.append(p)
            else:
                blocks.append((lin, None))
            i = j
        if len(blocks) < 2 or blocks[-1][1] is not None or any(bn is None for _, bn in blocks[:-1]):
            return None
        widths = [lin.out_features for lin, _ in blocks]
        if not ops.mlp_train_supported(x.shape[1], widths) or not ops._gemm_rows_ok(x, x.shape[1], 32):
            return None
        return blocks, drops, final

    def _fused_train(self, x, blocks, drops, final):
        from spiral_amd import ops

        masks = []
        for (lin, _), p in zip(blocks[:-1], drops):
            if p > 0.0:
                ones = torch.ones(x.shape[0], lin.out_features, dtype=torch.float32, device=x.device)
                masks.append(nn.functional.dropout(ones, p, True))
            else:
                masks.append(None)
        out = ops.mlp_head_train(x, blocks, masks)
        return final(out) if final is not None else out

    # ---- one-kernel inference path ---------------------------------------------------------
    # At the benchmark shape (1M rows, 128 -> 10 -> 20 -> 10 -> 5 -> 1) this path makes a step
    # 0.32-0.35 ms shorter (recurrent stage 3.52-3.64 ms against 3.90-4.01 ms).  Callers that
    # time short windows should know one thing: a full Python garbage collection costs
    # 35-40 ms with torch loaded, and with this path it tended to fall inside a 32-step
    # window where the torch layers' runs had it just outside; gc.freeze() after building the
    # model removes it for both.
    FUSE_EVAL = True

    def _fused_eval_ok(self, x: torch.Tensor) -> bool:
        if (not self.FUSE_EVAL or self.training or torch.is_grad_enabled() or not x.is_cuda
                or x.dtype != torch.float32 or x.dim() != 2 or x.shape[0] < 4096):
            return False
        from spiral_amd import ops

        linears = [m for m in self.layers if isinstance(m, nn.Linear)]
        return ops.mlp_head_supported(x.shape[1], [m.out_features for m in linears])

    def _packed_head(self, device):
        """Device buffers of the folded layers in the kernel's layout, rebuilt only when
        `_folded_layers` was (same version key), so a forward queues nothing but the kernel."""
        from spiral_amd import ops

        layers = self._folded_layers()
        cached = getattr(self, "_pack_cache", None)
        if cached is None or cached[0] is not layers or cached[1] != device:
            cached = self._pack_cache = (layers, device, ops.pack_mlp_head(layers, device))
        return cached[2]

    def _folded_layers(self):
        """(weight, bias, relu) per Linear with the eval-mode BatchNorm that follows it folded
        in: y = (W x + b - mean) * gamma / sqrt(var + eps) + beta.  Cached until a parameter
        or a running statistic changes."""
        tensors = getattr(self, "_fold_tensors", None)
        if tensors is None:
            tensors = self._fold_tensors = [t for m in self.layers
                                            for t in list(m.parameters()) + list(m.buffers())]
        # one integer per call: in-place updates bump _version, re-assignment or a move to
        # another device changes data_ptr; edits through .data change neither and need
        # ops.invalidate_weight_caches(), which moves the epoch
        key = cache_epoch()
        for t in tensors:
            key = key * 1000003 + t._version * 31 + t.data_ptr()
        cached = getattr(self, "_fold_cache", None)
        if cached is not None and cached[0] == key:
            return cached[1]
        out, mods, i = [], list(self.layers), 0
        self._final_act = None
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.Linear):
                w, b = m.weight.detach(), (m.bias.detach() if m.bias is not None else None)
                relu = False
                j = i + 1
                if j < len(mods) and isinstance(mods[j], nn.BatchNorm1d):
                    bn = mods[j]
                    scale = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
                    shift = bn.bias.detach() - bn.running_mean * scale
                    w = w * scale[:, None]
                    b = shift if b is None else b * scale + shift
                    j += 1
                if j < len(mods) and isinstance(mods[j], nn.ReLU) and j + 1 < len(mods):
                    relu = True
                    j += 1
                if j < len(mods) and isinstance(mods[j], nn.Dropout):
                    j += 1
                out.append((w.contiguous(), b, relu))
                i = j
            else:  # the activation after the last Linear
                self._final_act = {nn.Sigmoid: "sigmoid", nn.ReLU: "relu", nn.Tanh: "tanh"}[type(m)]
                i += 1
        self._fold_cache = (key, out)
        return out
