/* deltaconv_host.h -- C ABI of libdeltaconv_host.so (host-side, no GPU).
 * Replaces the reference's pybind11 module `deltaconv_bindings` (deltaconv/cpp/core.cpp:16-35). */
#ifndef DELTACONV_HOST_H
#define DELTACONV_HOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
int32_t dc_host_version(void);
/* geodesicFPS(vMat: float64[V,3], nSamples) -> int32[nSamples]  (deltaconv/cpp/core.cpp:16-25,
 * sampling.cpp:21-53).  seed < 0: random start (std::random_device, as the reference); seed >= 0:
 * reproducible start.  Returns 0, -1 on bad arguments. */
int dc_geodesic_fps(const double* points, int32_t n, int32_t num_samples, int64_t seed, int32_t* out);
#ifdef __cplusplus
}
#endif
#endif
