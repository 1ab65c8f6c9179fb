"""Oracle: geodesic farthest-point sampling in plain Python (heapq Dijkstra), restating
/root/reference/deltaconv/cpp/sampling.cpp:5-81.  TEST INFRASTRUCTURE ONLY (small inputs).

Parity status: the reference's C++ cannot be built here (geometry-central and Eigen are un-vendored
submodules) and its only test (test/geometry/test_fps.py:8-28) checks counts,
uniqueness and the two ValueErrors -- so value-level parity for this component is UNPINNED; this
restatement and the C++ one are checked against each other and against those properties."""
import heapq

import numpy as np


def knn_sets(points, k=10):
    d = ((points[:, None, :] - points[None, :, :]) ** 2).sum(-1)
    np.fill_diagonal(d, np.inf)
    return np.argsort(d, axis=1, kind="stable")[:, :k]


def geodesic_fps(points, n_samples, start):
    points = np.asarray(points, dtype=np.float64)
    n = points.shape[0]
    nbr = knn_sets(points, min(10, n - 1))
    D = np.full(n, np.inf)
    out = [int(start)]
    for _ in range(1, n_samples):
        src = out[-1]
        D[src] = 0.0
        heap = [(0.0, src)]
        while heap:
            du, u = heapq.heappop(heap)
            for v in [u] + list(nbr[u]):
                nd = du + float(np.linalg.norm(points[v] - points[u]))
                if nd < D[v]:
                    D[v] = nd
                    heapq.heappush(heap, (nd, int(v)))
        out.append(int(np.argmax(D)))          # first index of the maximum
    return np.array(out, dtype=np.int32)
