// One finalisation per thread, nothing else: the ISA of these kernels is what tools/count_f64_ops.py counts.
// probe<M, false>: the pair kernels' fused epilogue (dst_device.hpp: series logarithms, reciprocal multiplies);
// probe<M, true>:  the reference's operation order with the table logarithm (the text path).
#include "../../distance_amd/csrc/dst_device.hpp"

namespace dst {
template <int M, bool CLOSE>
__global__ void probe(const uint32_t *__restrict__ tallies, const uint4 *__restrict__ counts, double *__restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t o[4] = {tallies[4 * i], tallies[4 * i + 1], tallies[4 * i + 2], tallies[4 * i + 3]};
    out[i] = finalize_pair<M, CLOSE>(o, counts[2 * i], counts[2 * i + 1]);
}
template __global__ void probe<DST_RAW, false>(const uint32_t *, const uint4 *, double *);
template __global__ void probe<DST_JC69, false>(const uint32_t *, const uint4 *, double *);
template __global__ void probe<DST_K80, false>(const uint32_t *, const uint4 *, double *);
template __global__ void probe<DST_TN93, false>(const uint32_t *, const uint4 *, double *);
template __global__ void probe<DST_JC69, true>(const uint32_t *, const uint4 *, double *);
template __global__ void probe<DST_K80, true>(const uint32_t *, const uint4 *, double *);
template __global__ void probe<DST_TN93, true>(const uint32_t *, const uint4 *, double *);
}  // namespace dst
