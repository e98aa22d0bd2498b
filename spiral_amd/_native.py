"""ctypes binding of libspmi.so, the C-ABI HIP library (include/spmi.h).

There is no CPU fallback: if the shared library is missing, every entry point raises.
Pointers are raw device addresses (`tensor.data_ptr()`), the stream is torch's current
HIP stream; torch's caching allocator owns all memory.
"""
from __future__ import annotations

import ctypes
import threading
import os
from typing import Optional

import torch

# SPMI_LIB_PATH: another build of the same library, for A/B timing of kernel variants
_LIB_PATH = os.environ.get("SPMI_LIB_PATH") or os.path.join(
    os.path.dirname(os.path.abspath(__file__)), "lib", "libspmi.so")
_lib: Optional[ctypes.CDLL] = None

c_i64, c_i32, c_int, c_f32, c_vp, c_sz = (
    ctypes.c_int64, ctypes.c_int32, ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t)

# name -> (restype, argtypes); mirrors include/spmi.h one to one
_SIGNATURES = {
    "spmi_version": (c_int, []),
    "spmi_last_error": (ctypes.c_char_p, []),
    "spmi_degree_workspace_bytes": (c_sz, [c_i64, c_i64]),
    "spmi_degree": (c_int, [c_vp, c_i64, c_i64, c_vp, c_vp, c_sz, c_vp, c_vp]),
    "spmi_coo_to_csr_workspace_bytes": (c_sz, [c_i64, c_i64]),
    "spmi_coo_to_csr": (c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_sz,
                                c_vp, c_vp]),
    "spmi_snapshot_csr_workspace_bytes": (c_sz, [c_i64, c_i64]),
    "spmi_snapshot_csr_norm_other_supported": (c_int, [c_i64, c_i64, c_i64]),
    "spmi_snapshot_csr": (c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_i32,
                                  c_i32, c_vp, c_vp, c_i64, c_vp, c_sz, c_vp, c_vp]),
    "spmi_csr_prefix_workspace_bytes": (c_sz, [c_i64]),
    "spmi_csr_prefix": (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_sz,
                                c_vp, c_vp]),
    "spmi_csr_prefix_degree": (c_int, [c_vp, c_vp, c_i64, c_i64, c_vp, c_vp]),
    "spmi_norm_from_indptr": (c_int, [c_vp, c_i64, c_int, c_vp, c_vp]),
    "spmi_norm_from_degree": (c_int, [c_vp, c_i64, c_int, c_vp, c_vp]),
    "spmi_norm_from_ids32": (c_int, [c_vp, c_i64, c_i64, c_int, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "spmi_spmm_plan": (c_int, [c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_i64, c_vp, c_vp]),
    "spmi_gcn_spmm": (c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp, c_vp,
                              c_vp, c_vp, c_int, c_f32, c_vp, c_i64, c_vp, c_vp, c_i32, c_i32,
                              c_i32, c_i32, c_vp, c_vp]),
    "spmi_gcn_spmm_ex": (c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_int, c_i64, c_i64, c_vp,
                                 c_vp, c_vp, c_vp, c_int, c_f32, c_vp, c_int, c_i64, c_vp, c_vp,
                                 c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "spmi_gcn_spmm_part": (c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_int, c_i64, c_i64, c_vp,
                                   c_vp, c_vp, c_vp, c_int, c_f32, c_vp, c_int, c_i64, c_vp, c_vp,
                                   c_i32, c_i32, c_i32, c_i32, c_vp, c_int, c_int, c_vp]),
    "spmi_gcn_spmm_static": (c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_int, c_i64, c_i64, c_vp,
                                     c_vp, c_vp, c_vp, c_int, c_f32, c_vp, c_int, c_i64, c_vp, c_vp,
                                     c_i32, c_i32, c_i32, c_i32, c_vp]),
    "spmi_act_bias_bwd": (c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_int, c_f32, c_vp, c_vp, c_vp]),
    "spmi_act_bias_bwd_workspace_bytes": (c_sz, [c_i64, c_i64]),
    "spmi_act_bias_bwd_ordered": (c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_int, c_f32, c_vp, c_vp, c_vp, c_vp,
                                          c_sz, c_vp]),
    "spmi_gather_rows": (c_int, [c_vp, c_i64, c_vp, c_i64, c_i64, c_vp, c_vp]),
    "spmi_gru_gate_fwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_int,
                                  c_vp, c_vp]),
    "spmi_gru_gate_bwd": (c_int, [c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "spmi_gru_gate_bwd_packed": (c_int, [c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp]),
    "spmi_gru_gate_bwd_packed_q": (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp]),
    "spmi_lstm_gate_fwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp,
                                   c_vp, c_int, c_vp, c_vp]),
    "spmi_lstm_gate_bwd": (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp]),
    "spmi_rnn_gate_fwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_int,
                                  c_vp]),
    "spmi_rnn_gate_bwd": (c_int, [c_vp, c_vp, c_i64, c_i64, c_vp, c_vp]),
    "spmi_split_bf16x3": (c_int, [c_vp, c_i64, c_i64, c_vp, c_vp]),
    "spmi_cell_split_fwd": (c_int, [c_int, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                    c_vp, c_i64, c_i64, c_vp, c_vp, c_int, c_vp, c_vp]),
    "spmi_split_f16x3": (c_int, [c_vp, c_i64, c_i64, c_vp, c_vp]),
    "spmi_split_f16x3_strided": (c_int, [c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp]),
    "spmi_gemm_rows_supported": (c_int, [c_i64, c_i64]),
    "spmi_gemm_rows_bf16_supported": (c_int, [c_i64, c_i64]),
    "spmi_gemm_rows_bf16": (c_int, [c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp, c_int, c_f32, c_vp, c_i64,
                                    c_int, c_vp]),
    "spmi_gemm_rows": (c_int, [c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp, c_int, c_f32, c_vp,
                               c_i64, c_vp]),
    "spmi_cell_f16x3_fwd": (c_int, [c_int, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                    c_vp, c_i64, c_i64, c_vp, c_vp, c_int, c_vp, c_int, c_vp]),
    "spmi_gru_rows_supported": (c_int, [c_i64, c_i64]),
    "spmi_gru_rows_fwd": (c_int, [c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp,
                                  c_int, c_vp]),
    "spmi_gru_rows_fwd_saved": (c_int, [c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp,
                                        c_vp, c_int, c_vp]),
    "spmi_rnn_rows_supported": (c_int, [c_i64, c_i64]),
    "spmi_rnn_rows_fwd": (c_int, [c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp,
                                  c_int, c_vp]),
    "spmi_lstm_rows_supported": (c_int, [c_i64, c_i64]),
    "spmi_lstm_rows_fwd": (c_int, [c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64,
                                   c_vp, c_vp, c_int, c_vp]),
    "spmi_lstm_rows_fwd_saved": (c_int, [c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64,
                                         c_vp, c_vp, c_vp, c_int, c_vp]),
    "spmi_gat_edge_softmax": (c_int, [c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_i32, c_f32, c_vp, c_vp,
                                      c_vp, c_i32, c_i32, c_vp]),
    "spmi_gat_edge_softmax_bwd": (c_int, [c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_i32, c_f32, c_vp,
                                          c_vp, c_vp, c_i32, c_i32, c_vp]),
    "spmi_segment_sum": (c_int, [c_vp, c_i64, c_i64, c_vp, c_i32, c_vp, c_vp, c_i32, c_i32, c_vp]),
    "spmi_sddmm_dot_supported": (c_int, [c_i32, c_i32]),
    "spmi_sddmm_dot": (c_int, [c_vp, c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_i64, c_i32, c_i32,
                               c_vp, c_vp]),
    "spmi_gemm_tn_workspace_bytes": (c_sz, [c_i64, c_i64, c_i64]),
    "spmi_gemm_tn": (c_int, [c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_i64, c_int, c_vp,
                             c_sz, c_vp]),
    "spmi_gemm_tn_sums": (c_int, [c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_i64, c_int, c_vp,
                                  c_vp, c_int, c_vp, c_sz, c_vp]),
    "spmi_gemm_tn_pair_supported": (c_int, [c_i64, c_i64]),
    "spmi_gemm_tn_pair_workspace_bytes": (c_sz, [c_i64, c_i64, c_i64]),
    "spmi_gemm_tn_pair_sums": (c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_i64,
                                       c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_int, c_vp, c_sz, c_vp]),
    "spmi_mlp_head_supported": (c_int, [c_i64, c_i32, c_i32, c_vp]),
    "spmi_mlp_head_fwd": (c_int, [c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp,
                                  c_vp, c_i32, c_vp, c_i64, c_vp]),
    "spmi_mlp_train_workspace_bytes": (c_sz, []),
    "spmi_bn_col_stats": (c_int, [c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp, c_sz, c_vp]),
    "spmi_bn_finalize": (c_int, [c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp]),
    "spmi_bn_act_linear_fwd": (c_int, [c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp,
                                       c_vp, c_vp]),
    "spmi_bn_act_linear_bwd": (c_int, [c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_vp,
                                       c_vp, c_sz, c_vp, c_vp]),
    "spmi_bn_bwd_apply": (c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "spmi_cell_fused_supported": (c_int, [c_i64, c_i64]),
    "spmi_cell_fused_fwd": (c_int, [c_int, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                    c_vp, c_i64, c_i64, c_vp, c_vp, c_int, c_vp, c_vp]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


class SpmiError(RuntimeError):
    """A libspmi entry point returned a negative status."""


def lib_path() -> str:
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    """Load libspmi.so (once).  Raises if it has not been built: the product path has no
    other implementation to fall back to."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise SpmiError(
                f"{_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950); spiral_amd has no CPU fallback")
        handle = ctypes.CDLL(_LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().spmi_last_error()
        raise SpmiError(f"{what} failed with status {rc}: {msg.decode() if msg else ''}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Device address of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def require_cuda(*tensors: Optional[torch.Tensor]) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise SpmiError(
                "spiral_amd kernels run on the GPU only (got a CPU tensor); there is no CPU path")


# ---- zeroed device words, without a fill per use ---------------------------------------------
_slot_pools: dict = {}
_slot_lock = threading.Lock()


def zeroed_slot(device, width: int):
    """`width` zeroed int32 device words for the stream that is current on `device`.

    Kernels here report through a few words the caller has zeroed (status, plan sizes, the bit
    pattern of a maximum).  A fill per use would be a launch of its own, so the words are slices
    of a block of 256 that one fill zeroes.  That fill runs on the stream that was current when
    the block was made, so a block only ever serves that stream: work queued on another stream
    is not ordered behind the fill and takes a block of its own.  A slice is handed out once; a
    block lives as long as one of its slices.  Safe to call from several threads."""
    import torch

    dev = torch.device(device)
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream, width)
    with _slot_lock:
        pool = _slot_pools.get(key)
        if pool is None or pool[1] >= pool[0].shape[0]:
            pool = _slot_pools[key] = [torch.zeros(256, width, dtype=torch.int32, device=dev), 0]
        slot = pool[0][pool[1]]
        pool[1] += 1
    return slot
