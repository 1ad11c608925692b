// Workgroup dispatch-rate probe for gfx950: how long does a grid of N small workgroups take when each one does
// (almost) nothing?   hipcc --offload-arch=gfx950 -O3 -o tools/bin/dispatch_rate_probe tools/dispatch_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

__global__ void k_empty(uint32_t* out, int spin) {
    extern __shared__ uint8_t sm[];
    uint32_t a = threadIdx.x;
    for (int i = 0; i < spin; i++) asm volatile("v_add_u32 %0, %0, 1" : "+v"(a));
    if (a == 0xFFFFFFFFu) { sm[threadIdx.x] = 1; out[0] = sm[0]; }
}

static double time_ms(dim3 grid, dim3 block, size_t shm, uint32_t* out, int spin) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_empty, grid, block, shm, 0, out, spin); (void)hipDeviceSynchronize();
    std::vector<float> t;
    for (int r = 0; r < 7; r++) {
        (void)hipEventRecord(e0); hipLaunchKernelGGL(k_empty, grid, block, shm, 0, out, spin); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[3];
}

int main() {
    uint32_t* out; (void)hipMalloc(&out, 64);
    const int nwg = 1224 * 32;
    for (int threads : {64, 128, 256})
        for (size_t shm : {(size_t)0, (size_t)4096, (size_t)9216})
            for (int spin : {0, 600, 2400}) {
                const double t = time_ms(dim3(1224, 32), dim3(threads), shm, out, spin);
                printf("grid 1224x32 (%d WGs) x %3d threads, LDS %5zu B, spin %4d VALU: %.1f us -> %.1f WGs/us\n", nwg, threads, shm, spin, t * 1e3, nwg / (t * 1e3));
            }
    for (int n : {4096, 16384, 65536, 262144}) {
        const double t = time_ms(dim3(n), dim3(64), 0, out, 0);
        printf("grid %6d x 64 threads, no LDS, empty: %.1f us -> %.1f WGs/us\n", n, t * 1e3, n / (t * 1e3));
    }
    return 0;
}
