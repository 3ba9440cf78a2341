// kernels_dbfile.hpp -- packs the k-mer records of the database file on the device (layout: ipk_format.hpp).
//
// Device analogue of the serialisation loop at db_builder.cpp:323-327: `for (kmer, fv) in kmer_order: save_phylo_kmer(...)`.
// The file body is produced in filter order straight from the key-major database arrays; the host only writes bytes.
#pragma once
#include "dcla_device.hpp"
#include "ipk_format.hpp"

namespace ipkgpu {

// rec_bytes[i] = size of the record of the i-th k-mer in filter order (u32: a k-mer has < 2^28 entries)
__global__ __launch_bounds__(256) void db_record_sizes_kernel(const uint32_t* __restrict__ order, const uint64_t* __restrict__ key_off,
                                                              uint64_t n_keys, uint32_t* __restrict__ rec_bytes)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_keys) return;
    const uint32_t k = order[i];
    rec_bytes[i] = (uint32_t)ipkfmt::record_bytes(key_off[k + 1] - key_off[k]);
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
        *reinterpret_cast<uint4*>(dst) = make_uint4(w[0], w[1], w[2], w[3]);      // records are 8-byte sized: heads 8-byte aligned
    }
    uint2* e = reinterpret_cast<uint2*>(dst + ipkfmt::RECORD_HEAD_BYTES);
    for (uint64_t j = lane; j < n; j += 64) e[j] = entries[a + j];                // (branch, score bits) = the entry's bytes
}

}  // namespace ipkgpu
