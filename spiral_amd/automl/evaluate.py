"""Ranking metrics of the link-prediction task.

The reference evaluates link prediction with `ogb.linkproppred.Evaluator(name='ogbl-citation2')`
(examples/tf/link_prediction_sliding_window.py:142-146; ogb 1.3.3 per poetry.lock:744-745):
every positive pair is ranked against its own list of negative pairs and the mean of the
reciprocal ranks is reported.  ogb is not installable here, so this is a restatement of its
published `_eval_mrr`: rank = 1 + (number of negatives scored above the positive), ties
counted half (the mean of the optimistic and the pessimistic rank); without ties that equals
the position of the positive in the descending sort.  Parity with ogb is unpinned (no fixture
of the reference holds an MRR value); tests/test_host_logic.py checks the definition on
hand-computed cases.
"""
from __future__ import annotations

from typing import Dict

import torch


def mrr(y_pred_pos: torch.Tensor, y_pred_neg: torch.Tensor) -> Dict[str, torch.Tensor]:
    """`y_pred_pos` [n]: scores of the positive pairs; `y_pred_neg` [n, k]: scores of the k
    negatives ranked against each.  Returns per-positive lists like ogb's evaluator:
    'mrr_list', 'hits@1_list', 'hits@3_list', 'hits@10_list' (float32, on the inputs' device)."""
    y_pred_pos = torch.as_tensor(y_pred_pos)
    y_pred_neg = torch.as_tensor(y_pred_neg)
    if y_pred_neg.dim() != 2 or y_pred_pos.reshape(-1).shape[0] != y_pred_neg.shape[0]:
        raise ValueError(f"expected y_pred_pos [n] and y_pred_neg [n, k], got {tuple(y_pred_pos.shape)} "
                         f"and {tuple(y_pred_neg.shape)}")
    pos = y_pred_pos.reshape(-1, 1)
    optimistic = (y_pred_neg > pos).sum(dim=1)
    pessimistic = (y_pred_neg >= pos).sum(dim=1)
    rank = 0.5 * (optimistic + pessimistic).to(torch.float32) + 1.0
    return {
        "mrr_list": 1.0 / rank,
        "hits@1_list": (rank <= 1).to(torch.float32),
        "hits@3_list": (rank <= 3).to(torch.float32),
        "hits@10_list": (rank <= 10).to(torch.float32),
    }


def link_prediction_mrr(predict: torch.Tensor, n_pos: int, negatives_per_node: int = 1000,
                        positives_per_node: int = 2) -> float:
    """Mean reciprocal rank of one evaluation pass laid out as the reference's script lays it
    out (link_prediction_sliding_window.py:128-146): `predict` holds the scores of the `n_pos`
    positive pairs first (`positives_per_node` consecutive pairs per evaluated node), then
    `negatives_per_node` negative pairs per evaluated node; every positive of a node is ranked
    against that node's negatives."""
    predict = torch.as_tensor(predict).reshape(-1)
    if n_pos % positives_per_node:
        raise ValueError("n_pos must be a multiple of positives_per_node")
    n_nodes = n_pos // positives_per_node
    neg = predict[n_pos:]
    if neg.numel() != n_nodes * negatives_per_node:
        raise ValueError(f"expected {n_nodes * negatives_per_node} negative scores, got {neg.numel()}")
    neg = neg.reshape(n_nodes, negatives_per_node).repeat_interleave(positives_per_node, dim=0)
    return float(mrr(predict[:n_pos], neg)["mrr_list"].mean())
