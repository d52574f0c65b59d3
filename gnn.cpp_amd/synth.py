"""Deterministic synthetic inputs: counter-based SplitMix64 so that this
container, the GPU box's host and the HIP generator kernel (csrc/gnnx_synth.hip) produce
bit-identical graphs and features from a seed -- no data files are shipped.

The reference ships no dataset and seeds its RNG from time() (utils.cpp:6), so inputs are ours.
"""
import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(x):
    """Finaliser of SplitMix64 applied to a uint64 array (stateless, counter-based)."""
    with np.errstate(over="ignore"):
        z = (x + _GOLDEN).astype(np.uint64)
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _stream(seed, n, offset=0):
    with np.errstate(over="ignore"):
        i = np.arange(offset, offset + n, dtype=np.uint64)
        return splitmix64(splitmix64(np.uint64(seed)) ^ (i * _GOLDEN))


def uniform_pm1(seed, shape, scale=1.0):
    """fp32 U[-scale, scale): top 24 bits of the stream -> exactly representable, portable."""
    n = int(np.prod(shape))
    r = (_stream(seed, n) >> np.uint64(40)).astype(np.float32)  # 24 bits
    v = r * np.float32(2.0 ** -23) - np.float32(1.0)
    if scale != 1.0:
        v = v * np.float32(scale)
    return v.reshape(shape)


def uniform_edges(seed, n_nodes, n_edges):
    """Uniform random endpoints (the Cora-sized config C2)."""
    r = _stream(seed, 2 * n_edges)
    src = ((r[:n_edges] >> np.uint64(32)) % np.uint64(n_nodes)).astype(np.int32)
    dst = ((r[n_edges:] >> np.uint64(32)) % np.uint64(n_nodes)).astype(np.int32)
    return src, dst


def rmat_edges(seed, n_nodes, n_edges, a=0.57, b=0.19, c=0.19, first_edge=0):
    """R-MAT edge list; (a,b,c,d) quadrant probabilities, one 32-bit draw per level, ids folded with
    `% n_nodes` when n_nodes is not a power of two.  Bit-identical to gnnx_rmat_edges (csrc/gnnx_synth.hip).
    Edge e, level l draws the high 32 bits of splitmix64(key ^ ((e*64 + l) * GOLDEN))."""
    scale = max(1, int(np.ceil(np.log2(max(2, n_nodes)))))
    ta = np.uint64(int(a * 4294967296.0))
    tb = np.uint64(int((a + b) * 4294967296.0))
    tc = np.uint64(int((a + b + c) * 4294967296.0))
    key = splitmix64(np.uint64(seed))
    e = np.arange(first_edge, first_edge + n_edges, dtype=np.uint64)
    src = np.zeros(n_edges, dtype=np.uint64)
    dst = np.zeros(n_edges, dtype=np.uint64)
    with np.errstate(over="ignore"):
        for l in range(scale):
            r = splitmix64(key ^ ((e * np.uint64(64) + np.uint64(l)) * _GOLDEN)) >> np.uint64(32)
            sbit = (r >= tb).astype(np.uint64)                      # quadrants c,d -> lower half (src bit 1)
            dbit = (((r >= ta) & (r < tb)) | (r >= tc)).astype(np.uint64)  # quadrants b,d -> right half
            src = (src << np.uint64(1)) | sbit
            dst = (dst << np.uint64(1)) | dbit
    return (src % np.uint64(n_nodes)).astype(np.int32), (dst % np.uint64(n_nodes)).astype(np.int32)


KARATE_EDGES_1BASED = [
    (2, 1), (3, 1), (3, 2), (4, 1), (4, 2), (4, 3), (5, 1), (6, 1), (7, 1), (7, 5), (7, 6), (8, 1), (8, 2), (8, 3),
    (8, 4), (9, 1), (9, 3), (10, 3), (11, 1), (11, 5), (11, 6), (12, 1), (13, 1), (13, 4), (14, 1), (14, 2), (14, 3),
    (14, 4), (17, 6), (17, 7), (18, 1), (18, 2), (20, 1), (20, 2), (22, 1), (22, 2), (26, 24), (26, 25), (28, 3),
    (28, 24), (28, 25), (29, 3), (30, 24), (30, 27), (31, 2), (31, 9), (32, 1), (32, 25), (32, 26), (32, 29), (33, 3),
    (33, 9), (33, 15), (33, 16), (33, 19), (33, 21), (33, 23), (33, 24), (33, 30), (33, 31), (33, 32), (34, 9),
    (34, 10), (34, 14), (34, 15), (34, 16), (34, 19), (34, 20), (34, 21), (34, 23), (34, 24), (34, 27), (34, 28),
    (34, 29), (34, 30), (34, 31), (34, 32), (34, 33)]


def karate_edges():
    """Zachary's karate club, 34 nodes, 78 undirected = 156 directed edges (BASELINE.json configs[0])."""
    u = np.array([p[0] - 1 for p in KARATE_EDGES_1BASED], dtype=np.int32)
    v = np.array([p[1] - 1 for p in KARATE_EDGES_1BASED], dtype=np.int32)
    return np.concatenate([u, v]), np.concatenate([v, u])
