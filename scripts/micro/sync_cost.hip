// sync_cost.hip -- what one mid-call read-back costs on this box: (a) copy + stream sync, (b) zero-copy write + stream
// sync, (c) zero-copy write + flag, host polls the flag.  Build: hipcc --offload-arch=gfx950 -O2 -o sync_cost sync_cost.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_work(uint32_t *d, uint32_t v) { if (threadIdx.x == 0) d[0] = v; }
__global__ void k_pub(uint32_t *d, volatile uint32_t *host, uint32_t v) {
    if (threadIdx.x < 32) host[1 + threadIdx.x] = v + threadIdx.x;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) { __hip_atomic_store((uint32_t *)host, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
}
int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    uint32_t *d; CK(hipMalloc(&d, 4096));
    uint32_t *h; CK(hipHostMalloc(&h, 4096, hipHostMallocMapped));
    uint32_t *hd; CK(hipHostGetDevicePointer((void **)&hd, h, 0));
    const int N = 2000;
    for (int mode = 0; mode < 3; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 1; i <= N; i++) {
                if (mode == 0) { k_work<<<1, 64, 0, st>>>(d, i); CK(hipMemcpyAsync(h, d, 4, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); }
                else if (mode == 1) { k_work<<<1, 64, 0, st>>>(hd, i); CK(hipStreamSynchronize(st)); }
                else { k_pub<<<1, 64, 0, st>>>(d, hd, (uint32_t)(i + rep * N)); while (__atomic_load_n((volatile uint32_t *)h, __ATOMIC_ACQUIRE) != (uint32_t)(i + rep * N)) { } }
            }
            CK(hipStreamSynchronize(st));
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
            if (rep) printf("mode %d (%s): %.2f us per launch+readback\n", mode, mode == 0 ? "memcpyAsync + streamSync" : mode == 1 ? "zero-copy store + streamSync" : "zero-copy store + flag, host polls", us);
        }
    }
    return 0;
}
