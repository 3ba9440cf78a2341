// micro_atomics.hip -- measures the scatter max-reduce primitives the scoring kernel can use:
//   (1) global atomicMax(u32) at random slots of a T-byte table
//   (2) LDS atomicMax(u32) at random slots of a 64 KiB table, flushed with plain stores
// Build: hipcc -O3 --offload-arch=gfx950 -o gpurun_out/micro_atomics tools/micro_atomics.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ uint32_t hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// each thread performs `per` atomics; slot = hash(gid, i) & mask within table `tab_id`
__global__ void k_global_atomic(uint32_t* table, uint32_t mask, uint32_t n_tables, uint32_t per, int precheck)
{
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t* tab = table + (size_t)(blockIdx.x % n_tables) * ((size_t)mask + 1);
    for (uint32_t i = 0; i < per; ++i) {
        const uint32_t h = hash32(gid * 131u + i * 2654435761u);
        const uint32_t v = hash32(h) | 1u;
        uint32_t* p = tab + (h & mask);
        if (precheck) { if (__builtin_nontemporal_load(p) >= v) continue; }
        atomicMax(p, v);
    }
}

__global__ void k_lds_atomic(uint32_t* out, uint32_t per)
{
    __shared__ uint32_t tab[16384];
    for (uint32_t i = threadIdx.x; i < 16384; i += blockDim.x) tab[i] = 0;
    __syncthreads();
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t i = 0; i < per; ++i) {
        const uint32_t h = hash32(gid * 131u + i * 2654435761u);
        atomicMax(&tab[h & 16383u], hash32(h) | 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 16384; i += blockDim.x) out[(size_t)blockIdx.x * 16384 + i] = tab[i];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main()
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const uint32_t blocks = 8192, threads = 256, per = 256;
    const double n_atomics = (double)blocks * threads * per;
    // (1) global: table sizes 4 MiB (one DNA k=10 group), 64 MiB, 1 GiB total spread over n tables
    struct Cfg { uint32_t log2_slots; uint32_t n_tables; };
    Cfg cfgs[] = {{20, 1}, {20, 8}, {20, 256}, {24, 1}, {24, 16}, {26, 4}};
    for (Cfg c : cfgs) {
        const size_t slots = (size_t)1 << c.log2_slots;
        uint32_t* t; CK(hipMalloc(&t, slots * c.n_tables * 4));
        for (int pre = 0; pre < 2; ++pre) {
            CK(hipMemset(t, 0, slots * c.n_tables * 4));
            hipLaunchKernelGGL(k_global_atomic, dim3(blocks), dim3(threads), 0, 0, t, (uint32_t)(slots - 1), c.n_tables, 8u, pre);
            CK(hipDeviceSynchronize());
            CK(hipMemset(t, 0, slots * c.n_tables * 4));
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(k_global_atomic, dim3(blocks), dim3(threads), 0, 0, t, (uint32_t)(slots - 1), c.n_tables, per, pre);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            printf("global atomicMax u32: table %6.0f MiB x %3u tables precheck=%d : %8.3f ms  %7.2f G atomics/s\n",
                   slots * 4 / 1048576.0, c.n_tables, pre, ms, n_atomics / ms / 1e6);
        }
        CK(hipFree(t));
    }
    // (2) LDS
    {
        uint32_t* out; CK(hipMalloc(&out, (size_t)blocks * 16384 * 4));
        hipLaunchKernelGGL(k_lds_atomic, dim3(blocks), dim3(threads), 0, 0, out, 8u);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k_lds_atomic, dim3(blocks), dim3(threads), 0, 0, out, per);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("LDS atomicMax u32 (64 KiB table/WG, flush 64 KiB): %8.3f ms  %7.2f G atomics/s\n", ms, n_atomics / ms / 1e6);
        CK(hipFree(out));
    }
    return 0;
}
