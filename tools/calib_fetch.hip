// calib_fetch.hip -- calibrates rocprofv3 FETCH_SIZE for THIS path's access pattern (MI355X_MICROARCH.md, HBM section:
// "calibrate on a known byte count in your own access pattern before trusting an absolute").
// Every lane does `iters` dependent-free random loads: 16 B (dwordx4) + 8 B + 4 B from ONE random 128-byte line of a
// buffer much larger than the Infinity Cache -- the same shape as the search kernel's per-epoch request.
// Known traffic: lines touched = threads * iters (distinct with overwhelming probability).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

__global__ void gather_kernel(const char* buf, uint64_t n_lines, int iters, int mode, uint64_t* sink) {
    uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t s = tid * 0x9E3779B97F4A7C15ull + 12345;
    uint64_t acc = 0;
    for (int i = 0; i < iters; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const char* line = buf + (s % n_lines) * 128;
        uint32_t o = (uint32_t)(s >> 40) % 48;
        uint4 v; __builtin_memcpy(&v, line + o, 16);
        acc += v.x ^ v.w;
        if (mode >= 1) { acc += *(const uint64_t*)(line + 64 + 8 * ((s >> 50) & 3)); acc += *(const uint32_t*)(line + 96 + 4 * ((s >> 50) & 3)); }
    }
    if (acc == 0x1234567) sink[0] = acc;
}
__global__ void stream_kernel(const uint4* buf, uint64_t n16, uint64_t* sink) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t acc = 0;
    for (; i < n16; i += stride) { uint4 v = buf[i]; acc += v.x ^ v.w; }
    if (acc == 0x1234567) sink[0] = acc;
}

int main(int argc, char** argv) {
    uint64_t bytes = 4ull << 30; int iters = 64; int threads = 1 << 22;
    char* buf; uint64_t* sink;
    hipMalloc(&buf, bytes); hipMalloc(&sink, 8);
    hipMemset(buf, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; mode++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(gather_kernel, dim3(threads / 256), dim3(256), 0, 0, buf, bytes / 128, iters, mode, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double lines = (double)threads * iters;
        printf("gather mode %d: %.0f lines touched (%.3f GB at 128 B/line, %.3f GB at 64 B/line), %.3f ms, %.1f G lines/s\n", mode, lines,
               lines * 128 / 1e9, lines * 64 / 1e9, ms, lines / ms / 1e6);
    }
    hipEventRecord(e0);
    hipLaunchKernelGGL(stream_kernel, dim3(4096), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("stream: %.3f GB read, %.3f ms, %.1f GB/s\n", bytes / 1e9, ms, bytes / 1e6 / ms);
    return 0;
}
