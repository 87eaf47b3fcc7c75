// dst_device.hpp — device-side helpers shared by the pair kernels (dst_kernels.hip: dense bit-plane
// path, dst_consensus.hip: consensus-delta path): canonical-order indexing and the finalisation math
// (tallies -> f64 in the reference's operation order, src/measures.rs:68, 76, 109-112, 118-190).
// Both files are built with -ffp-contract=off: rustc never fuses a*b+c.
#pragma once
#include "dst_internal.h"

namespace dst {
namespace {

__device__ __forceinline__ uint64_t tri_row_start(uint64_t n, uint64_t i)
{
    return i * (2 * n - i - 1) / 2;
}

// ---- finalisation math: tallies -> f64 in the reference's operation order ---------------------
__device__ __forceinline__ double fin_raw(uint32_t n, uint32_t d)  // src/measures.rs:68
{
    return (double)n / (double)d;
}

__device__ __forceinline__ double fin_jc69(uint32_t n, uint32_t d)  // src/measures.rs:72-77
{
    const double p = fin_raw(n, d);
    return -0.75 * log(1.0 - (4.0 / 3.0) * p);
}

__device__ __forceinline__ double fin_k80(uint32_t count_L, uint32_t ts, uint32_t tv)  // :109-112
{
    const double P = (double)ts / (double)count_L;
    const double Q = (double)tv / (double)count_L;
    return -0.5 * log((1.0 - 2.0 * P - Q) * sqrt(1.0 - 2.0 * Q));
}

// counts = {A, T, G, C}; sums keep the reference's operand order (target first), :118-190
__device__ __forceinline__ double fin_tn93(uint32_t count_L, uint32_t count_d, uint32_t count_P1,
                                           uint32_t count_P2, uint4 qc, uint4 tc)
{
    const uint64_t L = (uint64_t)qc.x + qc.y + qc.z + qc.w + tc.x + tc.y + tc.z + tc.w;
    const double g_A = ((double)tc.x + (double)qc.x) / (double)L;
    const double g_C = ((double)tc.w + (double)qc.w) / (double)L;
    const double g_G = ((double)tc.z + (double)qc.z) / (double)L;
    const double g_T = ((double)tc.y + (double)qc.y) / (double)L;
    const double g_R = ((double)tc.x + (double)qc.x + (double)tc.z + (double)qc.z) / (double)L;
    const double g_Y = ((double)tc.w + (double)qc.w + (double)tc.y + (double)qc.y) / (double)L;
    const double k1 = 2.0 * g_A * g_G / g_R;
    const double k2 = 2.0 * g_T * g_C / g_Y;
    const double k3 = 2.0 * (g_R * g_Y - g_A * g_G * g_Y / g_R - g_T * g_C * g_R / g_Y);
    const double P1 = (double)count_P1 / (double)count_L;
    const double P2 = (double)count_P2 / (double)count_L;
    const double Q = (double)(count_d - (count_P1 + count_P2)) / (double)count_L;
    const double w1 = 1.0 - P1 / k1 - Q / (2.0 * g_R);
    const double w2 = 1.0 - P2 / k2 - Q / (2.0 * g_Y);
    const double w3 = 1.0 - Q / (2.0 * g_R * g_Y);
    double d = -k1 * log(w1) - k2 * log(w2) - k3 * log(w3);
    if (d == 0.0)
        d = 0.0;
    return d;
}

constexpr int OUT_TALLY = -1;      // uint32 x NT site tallies per pair
constexpr int OUT_INT = -2;        // int64 (n / n_high)
constexpr int OUT_TALLY_ADD = -3;  // split-L launch: atomicAdd partial tallies into a zeroed buffer
constexpr int OUT_INT_ADD = -4;    // split-L launch: atomicAdd partial counts into zeroed int64
constexpr int OUT_TALLY16 = -5;    // uint16 x NT per pair (alignments shorter than 65,536 sites)
// OUT >= 0: the measure id whose f64 distance the epilogue writes

template <int MEASURE>
__device__ __forceinline__ double finalize_pair(const uint32_t *o, uint4 qc, uint4 tc)
{
    if constexpr (MEASURE == DST_RAW)
        return fin_raw(o[0], o[1]);
    else if constexpr (MEASURE == DST_JC69)
        return fin_jc69(o[0], o[1]);
    else if constexpr (MEASURE == DST_K80)
        return fin_k80(o[0], o[1], o[2]);
    else
        return fin_tn93(o[0], o[1], o[2], o[3], qc, tc);
}

}  // namespace
}  // namespace dst
