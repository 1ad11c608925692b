// Which engine carries a device-to-host hipMemcpyAsync of a result block (64 KB .. 4 MB, pinned destination) issued on
// a stream behind a kernel -- the DMA engines or the runtime's shader blit (__amd_rocclr_copyBuffer in a kernel trace)?
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/d2h_route_probe tools/d2h_route_probe.hip
//   rocprofv3 --kernel-trace --memory-copy-trace --stats -d out -- tools/bin/d2h_route_probe [with_h2d]
// Run it under different runtime settings (GPU_FORCE_BLIT_COPY_SIZE, HSA_ENABLE_SDMA_COPY_SIZE_OVERRIDE, ...) and read the
// kernel stats: a copyBuffer kernel per copy = shader route.  Prints the event-bracketed time per copy and GB/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void k_spin(uint32_t* p, int us) { /* keeps a stream busy for ~us microseconds (100 MHz clock) */
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)us * 100ull) __builtin_amdgcn_s_sleep(8);
    if (us < 0) p[0] = 1;
}
__global__ void k_fill(uint32_t* p, size_t n, uint32_t v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + (uint32_t)i;
}

int main(int argc, char** argv) {
    const bool with_h2d = argc > 1 && !strcmp(argv[1], "with_h2d");
    const size_t MAXB = 4u << 20;
    uint32_t *d, *d_up;
    uint8_t *h, *h_up;
    CHECK(hipMalloc(&d, MAXB));
    CHECK(hipMalloc(&d_up, 16u << 20));
    CHECK(hipHostMalloc(&h, MAXB, hipHostMallocDefault));
    CHECK(hipHostMalloc(&h_up, 16u << 20, hipHostMallocDefault));
    hipStream_t st, st2;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("with_h2d=%d\n", (int)with_h2d);
    if (!(argc > 3 && (!strcmp(argv[1], "pipeline") || !strcmp(argv[1], "busy"))))
    for (size_t bytes : {(size_t)64 << 10, (size_t)256 << 10, (size_t)1 << 20, (size_t)2 << 20, (size_t)4 << 20}) {
        std::vector<float> t;
        for (int r = 0; r < 60; r++) {
            hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, st, d, bytes / 4, (uint32_t)r);
            if (with_h2d) CHECK(hipMemcpyAsync(d_up, h_up, 15u << 20, hipMemcpyHostToDevice, st2));
            CHECK(hipEventRecord(e0, st));
            CHECK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, st));
            CHECK(hipEventRecord(e1, st));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 10) t.push_back(ms);
            if (((uint32_t*)h)[bytes / 4 - 1] != (uint32_t)r + (uint32_t)(bytes / 4 - 1)) { printf("DATA MISMATCH\n"); return 1; }
            CHECK(hipStreamSynchronize(st2));
        }
        std::sort(t.begin(), t.end());
        printf("D2H %7zu KB: median %.1f us (min %.1f) -> %.1f GB/s\n", bytes >> 10, t[t.size() / 2] * 1e3, t[0] * 1e3,
               (double)bytes / (t[t.size() / 2] * 1e-3) / 1e9);
    }
    /* the pipeline's shape: N streams, each kernel -> three D2H copies (528 B, 0.9 MB, 1 MB) with hipMemcpyDefault or
     * hipMemcpyDeviceToHost, several iterations in flight.   d2h_route_probe pipeline <streams> <default|d2h> */
    if (argc > 3 && !strcmp(argv[1], "pipeline")) {
        const int ns = atoi(argv[2]);
        const hipMemcpyKind kind = !strcmp(argv[3], "default") ? hipMemcpyDefault : hipMemcpyDeviceToHost;
        std::vector<hipStream_t> ss(ns);
        std::vector<uint32_t*> dd(ns);
        std::vector<uint8_t*> hh(ns);
        for (int i = 0; i < ns; i++) {
            CHECK(hipStreamCreateWithFlags(&ss[i], hipStreamNonBlocking));
            CHECK(hipMalloc(&dd[i], MAXB));
            CHECK(hipHostMalloc(&hh[i], MAXB, hipHostMallocDefault));
        }
        CHECK(hipEventRecord(e0, ss[0]));
        const int iters = 200;
        for (int r = 0; r < iters; r++) {
            for (int i = 0; i < ns; i++) {
                hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, ss[i], dd[i], (size_t)(512 << 10), (uint32_t)r);
                CHECK(hipMemcpyAsync(hh[i], dd[i], 528, kind, ss[i]));
                CHECK(hipMemcpyAsync(hh[i] + 4096, dd[i] + 1024, 900 << 10, kind, ss[i]));
                CHECK(hipMemcpyAsync(hh[i] + (1 << 20), dd[i] + (256 << 10), 1 << 20, kind, ss[i]));
            }
            if ((r & 3) == 3)
                for (int i = 0; i < ns; i++) CHECK(hipStreamSynchronize(ss[i]));
        }
        for (int i = 0; i < ns; i++) CHECK(hipStreamSynchronize(ss[i]));
        CHECK(hipEventRecord(e1, ss[0]));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("pipeline %d streams kind %s: %.1f us per (kernel + 3 copies) per stream-iteration, %.1f GB/s aggregate\n", ns, argv[3],
               ms * 1e3 / iters, (double)ns * iters * (528 + (900 << 10) + (1 << 20)) / (ms * 1e-3) / 1e9);
    }
    /* the pipeline's real condition: the copy is enqueued while a LONG kernel of the same stream is still running.
     *   d2h_route_probe busy <streams> <same|xstream>   same: copy on the kernel's stream; xstream: on ONE shared copy
     *   stream that waits for an event recorded behind the kernel */
    if (argc > 3 && !strcmp(argv[1], "busy")) {
        const int ns = atoi(argv[2]);
        const bool xs = !strcmp(argv[3], "xstream");
        std::vector<hipStream_t> ss(ns);
        std::vector<uint32_t*> dd(ns);
        std::vector<uint8_t*> hh(ns);
        std::vector<hipEvent_t> ev(ns), evc(ns);
        hipStream_t cs;
        CHECK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
        for (int i = 0; i < ns; i++) {
            CHECK(hipStreamCreateWithFlags(&ss[i], hipStreamNonBlocking));
            CHECK(hipMalloc(&dd[i], MAXB));
            CHECK(hipHostMalloc(&hh[i], MAXB, hipHostMallocDefault));
            CHECK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
            CHECK(hipEventCreateWithFlags(&evc[i], hipEventDisableTiming));
        }
        CHECK(hipEventRecord(e0, ss[0]));
        const int iters = 100;
        for (int r = 0; r < iters; r++) {
            for (int i = 0; i < ns; i++) {
                if (xs && r) CHECK(hipStreamWaitEvent(ss[i], evc[i], 0)); /* the next kernel overwrites what the copy reads */
                hipLaunchKernelGGL(k_spin, dim3(64), dim3(64), 0, ss[i], dd[i], 150);
                hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, ss[i], dd[i], (size_t)(512 << 10), (uint32_t)r);
                if (xs) {
                    CHECK(hipEventRecord(ev[i], ss[i]));
                    CHECK(hipStreamWaitEvent(cs, ev[i], 0));
                    CHECK(hipMemcpyAsync(hh[i], dd[i], 2 << 20, hipMemcpyDeviceToHost, cs));
                    CHECK(hipEventRecord(evc[i], cs));
                } else {
                    CHECK(hipMemcpyAsync(hh[i], dd[i], 2 << 20, hipMemcpyDeviceToHost, ss[i]));
                }
            }
            if ((r & 3) == 3) {
                for (int i = 0; i < ns; i++) CHECK(hipStreamSynchronize(ss[i]));
                CHECK(hipStreamSynchronize(cs));
            }
        }
        for (int i = 0; i < ns; i++) CHECK(hipStreamSynchronize(ss[i]));
        CHECK(hipStreamSynchronize(cs));
        CHECK(hipEventRecord(e1, ss[0]));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("busy %d streams %s: %.1f us per iteration (150-us kernel + 2 MB copy per stream)\n", ns, argv[3], ms * 1e3 / iters);
    }
    return 0;
}
