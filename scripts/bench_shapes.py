"""Algorithmic bytes per unit of the bench configs, shared by the profile scripts."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def _cfg(name):
    import bench
    return bench.CONFIGS[name]


def bytes_per_dist(name, bits=0):
    c = _cfg(name)
    bits = bits or c["bits"]
    D = 1 << (c["dim"] - 1).bit_length()
    return D * bits // 8 + (20 if bits > 1 else 18)


def bytes_per_exact(name):
    c = _cfg(name)
    return 4 * (1 << (c["dim"] - 1).bit_length()) + 4
