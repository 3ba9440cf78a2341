// What does ONE dependent float addition cost on this part?  prefix_max_kernel's running sum (matrix::preprocess, window.cpp:16-27) is a
// chain of `sites` dependent v_add_f32 in one lane; the kernel measures ~9.5 ns per addition.  This times the bare chain: N dependent
// additions in registers (no LDS, no loads), one wavefront per CU and one per SIMD-full CU, by s_memtime and by wall clock.
// build: hipcc -O3 --offload-arch=gfx950 -o /tmp/micro_addchain tools/micro_addchain.hip ; run: /tmp/micro_addchain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void chain(float* out, const float* in, int n, unsigned long long* cyc)
{
    float acc = in[threadIdx.x & 63];
    const float x0 = in[64], x1 = in[65], x2 = in[66], x3 = in[67];
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; i += 16) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(x0));
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(x1));
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(x2));
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(x3));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main()
{
    const int n = 1 << 20;
    float *in, *out; unsigned long long* cyc;
    const size_t max_threads = (size_t)2048 * 1024;                    // the largest launch below: every thread stores one float
    (void)hipMalloc(&in, 1024); (void)hipMalloc(&out, max_threads * sizeof(float)); (void)hipMalloc(&cyc, 2048 * sizeof(unsigned long long));
    std::vector<float> h(256, 1e-3f); (void)hipMemcpy(in, h.data(), 1024, hipMemcpyHostToDevice);
    for (int threads : {64, 256, 1024}) for (int blocks : {1, 256, 2048}) {
        if ((size_t)threads * blocks > max_threads || blocks > 2048) return 1;          // (operand sizes checked on the host before any launch)
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        hipLaunchKernelGGL(chain, dim3(blocks), dim3(threads), 0, 0, out, in, 1024, cyc);
        (void)hipEventRecord(a); hipLaunchKernelGGL(chain, dim3(blocks), dim3(threads), 0, 0, out, in, n, cyc); (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%4d threads x %4d blocks: %.2f ns per dependent add (wall), %.2f counter ticks per add\n", threads, blocks, ms * 1e6 / n, (double)c / n);
    }
    return 0;
}
