"""cphnsw_mi355x — MI355X-native query path of CP-HNSW behind the reference's Python API.

    from cphnsw_mi355x import CPIndex      # drop-in for `from cphnsw import CPIndex`
"""
from .index import (CPIndex, FastScanStream, encode_edges, heap_ops_debug, knn_bruteforce,  # noqa: F401
                    select_neighbors_debug)

__all__ = ["CPIndex", "FastScanStream", "encode_edges", "heap_ops_debug", "knn_bruteforce", "select_neighbors_debug"]
