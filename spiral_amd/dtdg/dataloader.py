"""OGB citation datasets as DTDG containers (reference: spiro/dtdg/dataloader.py).

`preprocess_citation_graph` turns the dictionary an OGB dataset hands out (edge_index (2,|E|),
node_feat (|V|,f), a per-node time column (|V|,1)) into the tables `CitationGraph` is built
from: nodes relabelled in time order as rows [t, feat...], edges oriented newer -> older as
rows [t, src, dst] sorted by the time of their source.  The reference walks the edge list in a
Python loop with a dictionary lookup per end point (dataloader.py:57-68); here the same tables
come from array indexing, in O(|E|) numpy work.

Both sorts are the reference's own `argsort()` calls on arrays of the same values and dtype
(dataloader.py:56, :70), so nodes of one year and edges of one year end up in the same order
as the reference leaves them on the same numpy build (that order is numpy's default,
unstable sort; it is not specified further by the reference).

The OGB packages download by dataset name; they are imported when a dataset is asked for, not
when this module is imported.
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np
from numpy import ndarray

from spiral_amd.core.logger import get_logger
from spiral_amd.dtdg.types import CitationGraph, DiscreteGraph

EDGES = "edge_index"
NODES = "node_feat"

SRC = 0
DST = 1


def citation_tables(graph_data: Dict[str, ndarray], node_time: str):
    """(edges [t, src, dst], nodes [t, feat...], timestamps) of a citation dataset, as numpy
    arrays (dataloader.py:50-70)."""
    feat = graph_data[NODES]
    n = feat.shape[0]
    when = graph_data[node_time]
    timestamps = np.unique(when)
    t_index = np.searchsorted(timestamps, when[:, 0]).reshape(n, 1)
    ids = np.arange(n).reshape(n, 1)
    nodes = np.hstack((ids, t_index, feat))
    nodes = nodes[nodes[:, 1].argsort()]  # [old id, t, feat...], time order
    new_id = np.empty(n, dtype=np.int64)
    new_id[nodes[:, 0].astype(np.int64)] = np.arange(n, dtype=np.int64)
    a = new_id[np.asarray(graph_data[EDGES][SRC], dtype=np.int64)]
    b = new_id[np.asarray(graph_data[EDGES][DST], dtype=np.int64)]
    sources = np.maximum(a, b)  # the newer paper cites the older one
    dest = np.minimum(a, b)
    edge_time = nodes[sources, 1]  # an edge appears with its source
    edges = np.vstack((edge_time, sources, dest)).transpose()
    edges = edges[edges[:, 0].argsort()]
    return edges, nodes[:, 1:], timestamps


def preprocess_citation_graph(graph_data: Dict[str, ndarray], node_time: str, device=None) -> CitationGraph:
    """A citation dataset as a CitationGraph (dataloader.py:36-71).  `node_time` names the key
    of the (|V|, 1) column with the time a node was added."""
    edges, nodes, timestamps = citation_tables(graph_data, node_time)
    return CitationGraph(edges, nodes, timestamps, device=device)


def _ogb_linkdata_loader(name: str) -> Dict:
    from ogb.linkproppred import LinkPropPredDataset  # downloads by name on first use

    return LinkPropPredDataset(name)[0]


def _ogb_nodedata_loader(name: str) -> Dict:
    """Node-property datasets carry the label apart from the features; the label becomes the
    last feature column (dataloader.py:79-84)."""
    from ogb.nodeproppred import NodePropPredDataset

    graph, label = NodePropPredDataset(name)[0]
    graph[NODES] = np.hstack((graph[NODES], label))
    return graph


# dataset name -> (loader, preprocessing, key of the node time column)
OGB_DATASETS = {
    "ogbl-citation2": (_ogb_linkdata_loader, preprocess_citation_graph, "node_year"),
    "ogbn-arxiv": (_ogb_nodedata_loader, preprocess_citation_graph, "node_year"),
    "ogbn-papers100M": (_ogb_nodedata_loader, preprocess_citation_graph, "node_year"),
}


def supported_ogb_datasets() -> List[str]:
    """Names `ogb_dataset` takes (dataloader.py:95-101)."""
    return list(OGB_DATASETS)


def ogb_dataset(name: str) -> DiscreteGraph:
    """Load (downloading if needed) and preprocess an OGB dataset (dataloader.py:104-118).
    An unknown name ends the program, as in the reference; a missing `ogb` package raises
    ImportError from the loader."""
    if name not in OGB_DATASETS:
        get_logger().error("Dataset %s is not supported." % (name))
        raise SystemExit("Fatal error happens!")
    loader, preprocess, time_key = OGB_DATASETS[name]
    return preprocess(loader(name), time_key)
