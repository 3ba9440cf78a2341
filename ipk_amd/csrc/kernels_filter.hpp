// kernels_filter.hpp -- MIF0 filter values of a database shard.
//
// Follows mif0_filter::calc_filter_values (ipk/src/filter.cpp:20-23,55-119), per k-mer, in double:
//   s_i = (float) min(pow(10, log_score_i), 1.0)                         (logscore_to_score, :20-23)
//   S   = sum_i s_i + (N - n) * threshold                                 (:66-84)
//   t   = threshold / S ;  h(x) = -x log2 x                               (:87-88, :55-58)
//   H   = N h(t) ; for every entry: H = H - h(t) + h(s_i / S)             (:91-108)
//   fv  = S * (H - log2 N)                                                (:110-115)
// N = total number of groups (= node count of the original tree, db_builder.cpp:261), n = entries of
// the k-mer.  One wavefront per k-mer: the lanes evaluate pow / log2 of 64 entries at a time, then the
// partial results are ADDED IN ENTRY ORDER by a wave-uniform serial loop (readlane), i.e. with the very
// association of the reference's two sequential loops (:66-84, :91-108) -- the double value, its float
// narrowing and hence the k-mer order follow the reference formula operation for operation; what can still
// differ is the last bit of the device libm's pow/log2 against the host's.
#pragma once
#include "dcla_device.hpp"

namespace ipkgpu {

// lane j's double, broadcast (j wave-uniform)
__device__ __forceinline__ double lane_value(double v, uint32_t j)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, (int)j);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), (int)j);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

__device__ __forceinline__ double mif0_score(uint32_t score_bits)
{
    double s = pow(10.0, (double)__uint_as_float(score_bits));
    if (1.0 < s) s = 1.0;                                  // std::min(a, b) = (b < a) ? b : a
    return (double)(float)s;                               // the function returns score_type (float)
}

__device__ __forceinline__ double shannon(double x) { return -x * log2(x); }

__global__ __launch_bounds__(256) void mif0_kernel(const uint64_t* __restrict__ key_off, const uint2* __restrict__ entries,
                                                   uint64_t n_keys, double N, double threshold,
                                                   double* __restrict__ fv64, float* __restrict__ fv32, uint64_t first)
{
    const uint64_t key = first + (uint64_t)blockIdx.x * 4 + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform
    if (key >= n_keys) return;
    const uint32_t lane = lane_id();
    const uint64_t a = key_off[key], b = key_off[key + 1];
    double score_sum = 0.0;                                                      // filter.cpp:66-80, in entry order
    for (uint64_t base = a; base < b; base += 64) {
        const uint64_t i = base + lane;
        const double s = i < b ? mif0_score(entries[i].y) : 0.0;
        const uint32_t cnt = (uint32_t)min((uint64_t)64, b - base);
        for (uint32_t j = 0; j < cnt; ++j) score_sum += lane_value(s, j);
    }
    const double n = (double)(b - a);
    const double S = score_sum + (N - n) * threshold;                             // :84
    const double ht = shannon(threshold / S);                                     // :87-88
    double H = N * ht;                                                            // :91
    for (uint64_t base = a; base < b; base += 64) {
        const uint64_t i = base + lane;
        const double tv = i < b ? shannon(mif0_score(entries[i].y) / S) : 0.0;   // :103-104
        const uint32_t cnt = (uint32_t)min((uint64_t)64, b - base);
        for (uint32_t j = 0; j < cnt; ++j) H = H - ht + lane_value(tv, j);        // :106
    }
    const double fv = S * (H - log2(N));                                          // :109-114
    if (lane == 0) { fv64[key] = fv; fv32[key] = (float)fv; }
}

// sort key: (order-preserving code of the float filter value) << 32 | position of the k-mer in the shard.
// Keys are ascending in the shard, so ties on the filter value keep ascending key order (the
// reference's std::sort leaves ties in unspecified order, db_builder.cpp:281-284).
__global__ __launch_bounds__(256) void filter_sortkey_kernel(const float* __restrict__ fv32, uint64_t n,
                                                             unsigned long long* __restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    out[i] = ((unsigned long long)enc_score_bits(__float_as_uint(fv32[i])) << 32) | (unsigned long long)i;
}
__global__ __launch_bounds__(256) void filter_order_kernel(const unsigned long long* __restrict__ sorted, uint64_t n,
                                                           uint32_t* __restrict__ order)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) order[i] = (uint32_t)sorted[i];
}

}  // namespace ipkgpu
