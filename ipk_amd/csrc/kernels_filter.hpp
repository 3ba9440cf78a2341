// kernels_filter.hpp -- MIF0 filter values of a database shard.
//
// Follows mif0_filter::calc_filter_values (ipk/src/filter.cpp:20-23,55-119), per k-mer, in double:
//   s_i = (float) min(pow(10, log_score_i), 1.0)                         (logscore_to_score, :20-23)
//   S   = sum_i s_i + (N - n) * threshold                                 (:66-84)
//   t   = threshold / S ;  h(x) = -x log2 x                               (:87-88, :55-58)
//   H   = N h(t) ; for every entry: H = H - h(t) + h(s_i / S)             (:91-108)
//   fv  = S * (H - log2 N)                                                (:110-115)
// N = total number of groups (= node count of the original tree, db_builder.cpp:261), n = entries of
// the k-mer.  One wavefront per k-mer; lanes stride the entries, partial sums are combined with
// shuffles, so the summation ORDER differs from the reference's sequential loop: values agree to
// ~1e-13 relative, not bit for bit (tests use 1e-9).  pow/log2 are the device libm's.
#pragma once
#include "dcla_device.hpp"

namespace ipkgpu {

__device__ __forceinline__ double wave_sum(double v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
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
                                                   double* __restrict__ fv64, float* __restrict__ fv32)
{
    const uint64_t key = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (key >= n_keys) return;
    const uint32_t lane = lane_id();
    const uint64_t a = key_off[key], b = key_off[key + 1];
    double part = 0.0;
    for (uint64_t i = a + lane; i < b; i += 64) part += mif0_score(entries[i].y);
    const double n = (double)(b - a);
    const double S = wave_sum(part) + (N - n) * threshold;
    const double ht = shannon(threshold / S);
    double hp = 0.0;
    for (uint64_t i = a + lane; i < b; i += 64) hp += shannon(mif0_score(entries[i].y) / S) - ht;
    const double H = N * ht + wave_sum(hp);
    const double fv = S * (H - log2(N));
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
