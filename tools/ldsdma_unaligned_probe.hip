#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
// lane i copies 16 bytes from g + misalign + i*stride into LDS (contiguous), then the block writes LDS out
__global__ void k(const uint8_t* g, int misalign, int stride, uint8_t* out) {
    extern __shared__ __align__(16) uint8_t sm[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint8_t* p = g + misalign + (size_t)(wv * 64 + lane) * stride;
    if (lane != 7) // an exec-masked lane must leave its LDS slot untouched
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                         (__attribute__((address_space(3))) void*)(sm + wv * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) out[i] = sm[i];
}
__global__ void kz(uint8_t* out) { extern __shared__ uint8_t sm[]; for (int i = threadIdx.x; i < 2048; i += blockDim.x) sm[i] = 0xEE; __syncthreads(); out[threadIdx.x] = sm[threadIdx.x]; }
int main() {
    const int N = 1 << 20;
    uint8_t* h = (uint8_t*)malloc(N);
    for (int i = 0; i < N; i++) h[i] = (uint8_t)(i * 131 + (i >> 8) * 7);
    uint8_t *d, *o;
    hipMalloc(&d, N); hipMalloc(&o, 4096);
    hipMemcpy(d, h, N, hipMemcpyHostToDevice);
    uint8_t ho[2048];
    int bad = 0;
    for (int mis = 0; mis < 8; mis++)
        for (int stride : {16, 48, 1241, 17}) {
            hipMemset(o, 0, 4096);
            hipLaunchKernelGGL(k, dim3(1), dim3(128), 2048, 0, d, mis, stride, o);
            hipMemcpy(ho, o, 2048, hipMemcpyDeviceToHost);
            int b = 0;
            for (int l = 0; l < 128; l++) {
                if ((l & 63) == 7) continue;
                for (int j = 0; j < 16; j++) b += ho[l * 16 + j] != h[mis + (size_t)l * stride + j];
            }
            printf("misalign %d stride %4d: %s (%d wrong bytes)\n", mis, stride, b ? "FAIL" : "ok", b);
            bad += b;
        }
    printf(bad ? "LDSDMA_UNALIGNED_FAIL\n" : "LDSDMA_UNALIGNED_OK\n");
    return bad != 0;
}
