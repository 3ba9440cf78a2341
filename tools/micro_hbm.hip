// micro_hbm.hip -- what HBM3E gives plain streaming kernels on this part: read-only, write-only (8 and 16 bytes per lane),
// copy, and scattered write runs of 144 B .. 36 KB at random places (144 B: a key's entries of one 64-group tile at AA k=6).
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/micro_hbm tools/micro_hbm.hip && /tmp/micro_hbm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void k_read(const uint4* __restrict__ a, size_t n, uint4* sink)
{
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = a[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    if (acc.x == 0x12345678u && acc.y == 1u) *sink = acc;
}
__global__ void k_write16(uint4* __restrict__ a, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        a[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
__global__ void k_write8(uint2* __restrict__ a, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        a[i] = make_uint2((uint32_t)i, 1);
}
__global__ void k_copy(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
// each wave writes runs of L entries (8 B each) at pseudo-random run slots; every slot is written once
template <int L>
__global__ void k_runs(uint2* __restrict__ a, size_t n_runs)
{
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    for (size_t r = wave; r < n_runs; r += nwaves) {
        const size_t slot = (r * 2654435761ull) % n_runs;
#pragma unroll
        for (int o = 0; o < L; o += 64)
            if (o + lane < L) a[slot * L + o + lane] = make_uint2((uint32_t)r, lane);
    }
}

template <class F> static float time_ms(F&& f, int reps = 5)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b); return ms / reps;
}

int main()
{
    const size_t bytes = 16ull << 30;
    uint4 *a = nullptr, *b = nullptr, *sink = nullptr;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&sink, 16) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    const int grid = 256 * 16, bs = 256;
    const size_t n16 = bytes / 16, n8 = bytes / 8;
    float t;
    t = time_ms([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(bs), 0, 0, a, n16, sink); });
    printf("read   16 B/lane : %7.2f ms  %6.2f TB/s\n", t, bytes / t / 1e9);
    t = time_ms([&] { hipLaunchKernelGGL(k_write16, dim3(grid), dim3(bs), 0, 0, a, n16); });
    printf("write  16 B/lane : %7.2f ms  %6.2f TB/s\n", t, bytes / t / 1e9);
    t = time_ms([&] { hipLaunchKernelGGL(k_write8, dim3(grid), dim3(bs), 0, 0, (uint2*)a, n8); });
    printf("write   8 B/lane : %7.2f ms  %6.2f TB/s\n", t, bytes / t / 1e9);
    t = time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(bs), 0, 0, a, b, n16); });
    printf("copy   16 B/lane : %7.2f ms  %6.2f TB/s (read + write)\n", t, 2.0 * bytes / t / 1e9);
#define RUNS(L) { const size_t n_runs = bytes / (8 * L) - 1; \
        t = time_ms([&] { hipLaunchKernelGGL(k_runs<L>, dim3(grid), dim3(bs), 0, 0, (uint2*)a, n_runs); }); \
        printf("%5d-B runs at random places: %6.2f ms  %6.2f TB/s of useful bytes\n", 8 * L, t, (double)n_runs * 8 * L / t / 1e9); }
    RUNS(18) RUNS(36) RUNS(72) RUNS(144) RUNS(576) RUNS(4608)
    return 0;
}
