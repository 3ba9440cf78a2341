// kernels_dbfile.hpp -- packs the k-mer records of the database file on the device (layout: ipk_format.hpp).
//
// Device analogue of the serialisation loop at db_builder.cpp:323-327: `for (kmer, fv) in kmer_order: save_phylo_kmer(...)`.
// The file body is produced in filter order straight from the key-major database arrays; the host only writes bytes.
#pragma once
#include "dcla_device.hpp"
#include "ipk_format.hpp"

namespace ipkgpu {

// rec_bytes[i] = size of the record of the i-th k-mer in filter order (u32: a k-mer has at most one entry per branch group, and
// a batch holds < 2^22 groups -- ipkgpu_db_write refuses databases whose largest record would not fit)
__global__ __launch_bounds__(256) void db_record_sizes_kernel(const uint32_t* __restrict__ order, const uint64_t* __restrict__ key_off,
                                                              uint64_t n_keys, uint32_t* __restrict__ rec_bytes, uint32_t* __restrict__ too_big)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_keys) return;
    const uint32_t k = order[i];
    const uint64_t n = key_off[k + 1] - key_off[k];
    if (n >= (1ull << 28)) atomicOr(too_big, 1u);
    rec_bytes[i] = (uint32_t)ipkfmt::record_bytes(n);
}

// One wavefront per k-mer of [i_lo, i_hi) (positions in filter order): head words by lane 0, entries by all lanes.
// `out` receives the body bytes [rec_off[i_lo], rec_off[i_hi]).
__global__ __launch_bounds__(256) void db_pack_kernel(const uint32_t* __restrict__ order, const uint32_t* __restrict__ keys,
                                                      const uint64_t* __restrict__ key_off, const uint2* __restrict__ entries,
                                                      const float* __restrict__ fv, const uint64_t* __restrict__ rec_off,
                                                      uint64_t i_lo, uint64_t i_hi, unsigned char* __restrict__ out)
{
    const uint64_t i = i_lo + (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= i_hi) return;
    const uint32_t lane = lane_id();
    const uint32_t k = order[i];
    const uint64_t a = key_off[k], n = key_off[k + 1] - a;
    unsigned char* dst = out + (rec_off[i] - rec_off[i_lo]);
    if (lane == 0) {
        uint32_t w[4];
        ipkfmt::record_head(keys[k], __float_as_uint(fv[k]), n, w);
        // records are 16 + 8 n bytes: a head is 8-byte aligned, not 16 -- two 8-byte stores, not one uint4
        reinterpret_cast<uint2*>(dst)[0] = make_uint2(w[0], w[1]);
        reinterpret_cast<uint2*>(dst)[1] = make_uint2(w[2], w[3]);
    }
    uint2* e = reinterpret_cast<uint2*>(dst + ipkfmt::RECORD_HEAD_BYTES);
    for (uint64_t j = lane; j < n; j += 64) e[j] = entries[a + j];                // (branch, score bits) = the entry's bytes
}

}  // namespace ipkgpu
