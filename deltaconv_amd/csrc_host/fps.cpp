// Geodesic farthest-point sampling on the k=10 nearest-neighbour graph -- dependency-free C++17
// restatement of the only native component the reference owns
// (/root/reference/deltaconv/cpp/sampling.cpp:5-81, core.cpp:16-25; there: geometry-central's
// nanoflann kNN + Eigen + pybind11, all un-vendored).  Host-side, once per shape at dataset
// preparation time (reference: transforms/geodesic_fps.py:14-43) -- not on the per-step GPU path.
//
// Algorithm (kept exactly): kNN graph with k=10 plus a self loop per point; distance vector D
// initialised to +inf and NEVER reset between rounds; round i runs Dijkstra (binary heap, lazy
// deletion) from the previous sample, updating D where shorter; the next sample is the FIRST index
// of max(D).  Start: random (std::random_device, as sampling.cpp:34-40) or fixed by `seed >= 0`.
#include <stdint.h>
#include <algorithm>
#include <cmath>
#include <limits>
#include <queue>
#include <random>
#include <utility>
#include <vector>

namespace {

struct P3 { double x, y, z; };

// k nearest neighbours of every point (excluding itself), brute force with partial selection.
std::vector<int> knn_graph(const std::vector<P3>& p, int k) {
    const int n = (int)p.size();
    const int kk = std::min(k, n - 1);
    std::vector<int> nbr((size_t)n * (kk > 0 ? kk : 0));
#pragma omp parallel
    {
        std::vector<std::pair<double, int>> d(n > 0 ? n - 1 : 0);
#pragma omp for schedule(static)
        for (int i = 0; i < n; ++i) {
            int m = 0;
            for (int j = 0; j < n; ++j) {
                if (j == i) continue;
                const double dx = p[j].x - p[i].x, dy = p[j].y - p[i].y, dz = p[j].z - p[i].z;
                d[m++] = {dx * dx + dy * dy + dz * dz, j};
            }
            if (kk > 0) {
                std::partial_sort(d.begin(), d.begin() + kk, d.end());
                for (int s = 0; s < kk; ++s) nbr[(size_t)i * kk + s] = d[s].second;
            }
        }
    }
    return nbr;
}

// sampling.cpp:56-81
void dijkstra(const std::vector<P3>& p, int source, const std::vector<int>& nbr, int kk, std::vector<double>& D) {
    using VP = std::pair<double, int>;  // (distance, vertex): std::greater orders by distance first
    std::priority_queue<VP, std::vector<VP>, std::greater<VP>> q;
    D[source] = 0.0;
    q.push({0.0, source});
    while (!q.empty()) {
        const VP cur = q.top();
        q.pop();
        const int u = cur.second;
        const P3 pu = p[u];
        for (int s = -1; s < kk; ++s) {  // s = -1: the self loop the reference inserts first (sampling.cpp:13-14)
            const int v = s < 0 ? u : nbr[(size_t)u * kk + s];
            const double dx = p[v].x - pu.x, dy = p[v].y - pu.y, dz = p[v].z - pu.z;
            const double nd = cur.first + std::sqrt(dx * dx + dy * dy + dz * dz);
            if (nd < D[v]) {
                D[v] = nd;
                q.push({nd, v});
            }
        }
    }
}

}  // namespace

extern "C" {

// points: [n,3] float64 row-major; out: [num_samples] int32.  seed < 0 -> std::random_device start.
// Returns 0, or -1 on bad arguments.  Mirrors geodesicFPS(vMat, nSamples) (core.cpp:16-25).
__attribute__((visibility("default"))) int dc_geodesic_fps(const double* points, int32_t n, int32_t num_samples,
                                                           int64_t seed, int32_t* out) {
    if (!points || !out || n < 1 || num_samples < 1) return -1;
    std::vector<P3> p(n);
    for (int i = 0; i < n; ++i) p[i] = P3{points[3 * i], points[3 * i + 1], points[3 * i + 2]};
    const int kk = std::min(10, n - 1);
    const std::vector<int> nbr = knn_graph(p, 10);
    std::vector<double> D(n, std::numeric_limits<double>::infinity());
    int start;
    if (seed < 0) {
        std::random_device rd;
        std::mt19937 gen(rd());
        start = std::uniform_int_distribution<>(0, n - 1)(gen);
    } else {
        std::mt19937 gen((uint32_t)seed);
        start = std::uniform_int_distribution<>(0, n - 1)(gen);
    }
    out[0] = start;
    for (int i = 1; i < num_samples; ++i) {
        dijkstra(p, out[i - 1], nbr, kk, D);
        int best = 0;  // first index of the maximum, as Eigen's maxCoeff(&index) (sampling.cpp:47-49)
        for (int j = 1; j < n; ++j)
            if (D[j] > D[best]) best = j;
        out[i] = best;
    }
    return 0;
}

__attribute__((visibility("default"))) int32_t dc_host_version(void) { return 100; }
}
