// Issue-rate probe for gfx950: how many cycles does a SIMD need per wave64 VALU / SALU / LDS instruction when 8 waves
// per SIMD keep it busy, and do the instruction classes overlap?      hipcc --offload-arch=gfx950 -O3 -o tools/bin/issue_rate_probe tools/issue_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define REP8(x) x x x x x x x x
#define VALU8 REP8("v_pk_min_u16 %0, %0, %1\n\t" "v_pk_max_u16 %2, %2, %1\n\t" "v_and_b32 %3, %3, %1\n\t" "v_or_b32 %4, %4, %1\n\t")
#define SALU8 REP8("s_add_u32 %0, %0, %1\n\t" "s_and_b32 %2, %2, %1\n\t" "s_or_b32 %3, %3, %1\n\t" "s_xor_b32 %4, %4, %1\n\t")

__global__ void k_valu(uint32_t* out, int iters) {
    uint32_t a = threadIdx.x, b = 3, c = 5, d = 7, e = 9;
    for (int i = 0; i < iters; i++) asm volatile(VALU8 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e));
    if (a + c + d + e == 0x12345) out[0] = a;
}
__global__ void k_valu_perm(uint32_t* out, int iters) { /* v_perm + pk_sub clamp + mbcnt + alignbyte mix */
    uint32_t a = threadIdx.x, b = 3, c = 5, d = 7, e = 9;
    for (int i = 0; i < iters; i++)
        asm volatile(REP8("v_perm_b32 %0, %0, %1, %1\n\t" "v_pk_sub_u16 %2, %2, %1 clamp\n\t" "v_alignbyte_b32 %3, %3, %1, 1\n\t" "v_mbcnt_lo_u32_b32 %4, %1, %4\n\t")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e));
    if (a + c + d + e == 0x12345) out[0] = a;
}
__global__ void k_salu(uint32_t* out, int iters) {
    uint32_t a = blockIdx.x, b = 3, c = 5, d = 7, e = 9;
    for (int i = 0; i < iters; i++) asm volatile(SALU8 : "+s"(a), "+s"(b), "+s"(c), "+s"(d), "+s"(e));
    if (a + c + d + e == 0x12345) out[0] = a;
}
__global__ void k_mix(uint32_t* out, int iters) { /* 32 VALU + 32 SALU per iteration, independent */
    uint32_t a = threadIdx.x, b = 3, c = 5, d = 7, e = 9;
    uint32_t sa = blockIdx.x, sb = 3, sc = 5, sd = 7, se = 9;
    for (int i = 0; i < iters; i++)
        asm volatile(REP8("v_pk_min_u16 %0, %0, %1\n\t" "s_add_u32 %5, %5, %6\n\t" "v_pk_max_u16 %2, %2, %1\n\t" "s_and_b32 %7, %7, %6\n\t"
                          "v_and_b32 %3, %3, %1\n\t" "s_or_b32 %8, %8, %6\n\t" "v_or_b32 %4, %4, %1\n\t" "s_xor_b32 %9, %9, %6\n\t")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+s"(sa), "+s"(sb), "+s"(sc), "+s"(sd), "+s"(se));
    if (a + c + d + e + sa + sc + sd + se == 0x12345) out[0] = a;
}
__global__ void k_lds_u8(uint32_t* out, int iters, int stride) { /* 32 ds_read_u8 per iteration */
    __shared__ uint8_t sm[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) sm[i] = (uint8_t)i;
    __syncthreads();
    uint32_t addr = (threadIdx.x * stride) & 4095, acc = 0;
    const uint8_t* p = sm + addr;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 32; k++) acc += p[(k * 53) & 4095];
        asm volatile("" : "+v"(acc));
    }
    if (acc == 0x12345) out[0] = acc;
}
__global__ void k_mix_lds(uint32_t* out, int iters, int stride) { /* 32 ds_read_u8 + 32 VALU pk ops per iteration */
    __shared__ uint8_t sm[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) sm[i] = (uint8_t)i;
    __syncthreads();
    uint32_t addr = (threadIdx.x * stride) & 4095, acc = 0;
    uint32_t a = threadIdx.x, b = 3, c = 5, d = 7, e = 9;
    const uint8_t* p = sm + addr;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 32; k++) acc += p[(k * 53) & 4095];
        asm volatile(REP8("v_pk_min_u16 %0, %0, %1\n\t" "v_pk_max_u16 %2, %2, %1\n\t" "v_and_b32 %3, %3, %1\n\t") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e));
        asm volatile("" : "+v"(acc));
    }
    if (acc + a + c + d == 0x12345) out[0] = acc;
}

template <class F>
static double time_ms(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    std::vector<float> t;
    for (int r = 0; r < 5; r++) {
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[2];
}

int main() {
    uint32_t* out; hipMalloc(&out, 64);
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("reported clock %d kHz\n", clk);
    const int iters = 4000;
    for (int wps = 1; wps <= 8; wps *= 2) { /* waves per SIMD */
        const int nwg = 256 * 4 * wps;
        const double per_simd_instr = (double)wps * iters * 32;
        double t;
        t = time_ms([&] { hipLaunchKernelGGL(k_valu, dim3(nwg), dim3(64), 0, 0, out, iters); });
        printf("wps %d  valu      : %.3f ms  -> %.2f ns per wave-instr per SIMD\n", wps, t, t * 1e6 / per_simd_instr);
        t = time_ms([&] { hipLaunchKernelGGL(k_valu_perm, dim3(nwg), dim3(64), 0, 0, out, iters); });
        printf("wps %d  valu perm : %.3f ms  -> %.2f ns per wave-instr per SIMD\n", wps, t, t * 1e6 / per_simd_instr);
        t = time_ms([&] { hipLaunchKernelGGL(k_salu, dim3(nwg), dim3(64), 0, 0, out, iters); });
        printf("wps %d  salu      : %.3f ms  -> %.2f ns per instr per SIMD\n", wps, t, t * 1e6 / per_simd_instr);
        t = time_ms([&] { hipLaunchKernelGGL(k_mix, dim3(nwg), dim3(64), 0, 0, out, iters); });
        printf("wps %d  valu+salu : %.3f ms  -> %.2f ns per (valu+salu) pair per SIMD\n", wps, t, t * 1e6 / per_simd_instr);
        for (int stride : {1, 4, 48, 67}) {
            t = time_ms([&] { hipLaunchKernelGGL(k_lds_u8, dim3(nwg), dim3(64), 0, 0, out, iters / 4, stride); });
            printf("wps %d  lds u8 stride %2d : %.3f ms -> %.2f ns per ds_read per SIMD\n", wps, stride, t, t * 1e6 / (per_simd_instr / 4));
        }
        t = time_ms([&] { hipLaunchKernelGGL(k_mix_lds, dim3(nwg), dim3(64), 0, 0, out, iters / 4, 67); });
        printf("wps %d  lds u8(67)+24 valu: %.3f ms -> %.2f ns per ds_read per SIMD\n", wps, t, t * 1e6 / (per_simd_instr / 4));
    }
    return 0;
}
