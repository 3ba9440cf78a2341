// comp_table.hpp -- read side of the compressed per-group tables.
#pragma once
#include "dcla_device.hpp"

namespace ipkgpu {

// The compressed table form of the exact-partition variant (kernels_score.hpp, reduce_ranges_kernel<.., COMPRESS>):
// occupancy bits + rank per 64-slot block + the non-empty slots' score codes in place in the pool.
struct CompTable {
    const uint32_t* mask;      // [groups][mask_words]
    const uint32_t* rank;      // [groups][mask_words / 2]
    const uint64_t* vaddr;     // [groups][mask_words / 2]: address of the block's first value (= values(g, b) + rank)
    const uint2* pool;         // values of (g, b) start at (u32*)(pool + off[(g * NB + b) * stride])
    const uint64_t* off;
    uint64_t mask_words;
    uint32_t NB, stride, TBL;
    // score code of slot x (a multiple of 64 plus xl) of group g; 0 = empty
    __device__ __forceinline__ uint32_t slot(uint32_t g, uint64_t x64, uint32_t xl) const
    {
        const uint64_t blk = x64 >> 6;
        const uint32_t* mw = mask + (size_t)g * mask_words + 2 * blk;
        const uint64_t m = (uint64_t)mw[0] | ((uint64_t)mw[1] << 32);
        if (!((m >> xl) & 1ull)) return 0u;
        const uint32_t b = (uint32_t)(x64 / TBL);
        const uint32_t* vals = reinterpret_cast<const uint32_t*>(pool + off[((size_t)g * NB + b) * stride]);
        return vals[rank[(size_t)g * (mask_words / 2) + blk] + (uint32_t)__popcll(m & ((1ull << xl) - 1ull))];
    }
};

}  // namespace ipkgpu
