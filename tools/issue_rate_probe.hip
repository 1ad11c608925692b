// Issue-rate probe for gfx950 (round 3 rewrite).  Question: how many shader cycles does one SIMD need per wave64
// instruction of a given kind when 1, 2, 4 or 8 waves per SIMD keep it busy -- and do plain 32-bit VALU operations issue
// faster than packed 16-bit ones (which would decide the arithmetic of the FAST score network)?
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/issue_rate_probe tools/issue_rate_probe.hip && tools/bin/issue_rate_probe
//
// Every row is ONE instruction kind: a block of 32 of them over EIGHT independent destination registers (a register is
// rewritten every 8th instruction), ITERS blocks per wave.  The clock is read INSIDE the kernel: s_memtime (shader cycles)
// around the loop of every wave, s_memrealtime (100 MHz) beside it, so the row shows cycles and the clock the chip ran at;
// the host checks hipGetLastError after every launch and prints the event time as a cross-check.  Workgroups are 256
// threads (one wave per SIMD), wps workgroups per CU; HW_ID is recorded and the waves per (CU, SIMD) are counted to confirm
// the placement the arithmetic assumes.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct WaveRec { unsigned long long cyc, rt, r0, r1; unsigned hwid, pad; };

#define R4(x) x x x x
/* 8 independent destinations %0..%7, one shared source %8 (and %9 where three inputs are needed) */
#define BLK8(OP, TAIL) \
    OP " %0, %0, %8" TAIL "\n\t" OP " %1, %1, %8" TAIL "\n\t" OP " %2, %2, %8" TAIL "\n\t" OP " %3, %3, %8" TAIL "\n\t" \
    OP " %4, %4, %8" TAIL "\n\t" OP " %5, %5, %8" TAIL "\n\t" OP " %6, %6, %8" TAIL "\n\t" OP " %7, %7, %8" TAIL "\n\t"
#define BLK8_3(OP, TAIL) \
    OP " %0, %0, %8, %9" TAIL "\n\t" OP " %1, %1, %8, %9" TAIL "\n\t" OP " %2, %2, %8, %9" TAIL "\n\t" OP " %3, %3, %8, %9" TAIL "\n\t" \
    OP " %4, %4, %8, %9" TAIL "\n\t" OP " %5, %5, %8, %9" TAIL "\n\t" OP " %6, %6, %8, %9" TAIL "\n\t" OP " %7, %7, %8, %9" TAIL "\n\t"

#define PROLOGUE()                                                                                          \
    unsigned long long t0, t1, r0, r1;                                                                      \
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) : : "memory")
#define EPILOGUE(sink)                                                                                      \
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) : : "memory"); \
    if ((threadIdx.x & 63) == 0) {                                                                          \
        unsigned hw;                                                                                        \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                                    \
        WaveRec w; w.cyc = t1 - t0; w.rt = r1 - r0; w.r0 = r0; w.r1 = r1; w.hwid = hw; w.pad = 0;                                 \
        rec[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = w;                                       \
    }                                                                                                       \
    if ((sink) == 0x12345u) out[0] = (sink)

#define VKERNEL(NAME, BODY)                                                                                 \
    __global__ void __launch_bounds__(256) NAME(uint32_t* out, WaveRec* rec, int iters) {                   \
        uint32_t a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3, e = a + 4, f = a + 5, g = a + 6, h = a + 7; \
        uint32_t s = 0x00030005u, s2 = 0x07060504u;                                                         \
        PROLOGUE();                                                                                         \
        for (int i = 0; i < iters; i++)                                                                     \
            asm volatile(R4(BODY) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(s), "v"(s2)); \
        EPILOGUE(a + b + c + d + e + f + g + h);                                                            \
    }

VKERNEL(k_add_u32, BLK8("v_add_u32", ""))
VKERNEL(k_and_b32, BLK8("v_and_b32", ""))
VKERNEL(k_xor_b32, BLK8("v_xor_b32", ""))
VKERNEL(k_min_u32, BLK8("v_min_u32", ""))
VKERNEL(k_min3_i32, BLK8_3("v_min3_i32", ""))
VKERNEL(k_max3_u32, BLK8_3("v_max3_u32", ""))
VKERNEL(k_pk_min_u16, BLK8("v_pk_min_u16", ""))
VKERNEL(k_pk_max_u16, BLK8("v_pk_max_u16", ""))
VKERNEL(k_pk_sub_u16c, BLK8("v_pk_sub_u16", " clamp"))
VKERNEL(k_pk_add_u16, BLK8("v_pk_add_u16", ""))
VKERNEL(k_perm_b32, BLK8_3("v_perm_b32", ""))
VKERNEL(k_alignbyte, BLK8_3("v_alignbyte_b32", ""))
VKERNEL(k_bfe_u32, BLK8_3("v_bfe_u32", ""))
VKERNEL(k_lshl_or, BLK8_3("v_lshl_or_b32", ""))
VKERNEL(k_and_or, BLK8_3("v_and_or_b32", ""))
VKERNEL(k_add3, BLK8_3("v_add3_u32", ""))
VKERNEL(k_mad_u32_u24, BLK8_3("v_mad_u32_u24", ""))
VKERNEL(k_fma_f32, BLK8_3("v_fma_f32", ""))
VKERNEL(k_add_f32, BLK8("v_add_f32", ""))
VKERNEL(k_bcnt, BLK8("v_bcnt_u32_b32", ""))
VKERNEL(k_mbcnt, BLK8("v_mbcnt_lo_u32_b32", ""))
VKERNEL(k_sad_u16, BLK8_3("v_sad_u16", ""))
VKERNEL(k_sad_u8, BLK8_3("v_sad_u8", ""))
VKERNEL(k_dot4_u8, BLK8_3("v_dot4_u32_u8", ""))
VKERNEL(k_mul_lo, BLK8("v_mul_lo_u32", ""))
VKERNEL(k_mul_u24, BLK8("v_mul_u32_u24", ""))
VKERNEL(k_min_u16, BLK8("v_min_u16", ""))
VKERNEL(k_max_u16, BLK8("v_max_u16", ""))
VKERNEL(k_min_i16, BLK8("v_min_i16", ""))
VKERNEL(k_sub_u16, BLK8("v_sub_u16", ""))
VKERNEL(k_add_u16, BLK8("v_add_u16", ""))
VKERNEL(k_or_b32, BLK8("v_or_b32", ""))
VKERNEL(k_sub_u32, BLK8("v_sub_u32", ""))
VKERNEL(k_max_u32, BLK8("v_max_u32", ""))
VKERNEL(k_min_i32, BLK8("v_min_i32", ""))
VKERNEL(k_lshlrev, BLK8("v_lshlrev_b32", ""))
VKERNEL(k_lshrrev, BLK8("v_lshrrev_b32", ""))
VKERNEL(k_ashrrev, BLK8("v_ashrrev_i32", ""))
VKERNEL(k_min_f32, BLK8("v_min_f32", ""))
VKERNEL(k_max_f32, BLK8("v_max_f32", ""))
VKERNEL(k_mul_f32, BLK8("v_mul_f32", ""))
VKERNEL(k_sub_f32, BLK8("v_sub_f32", ""))
VKERNEL(k_cndmask, BLK8("v_cndmask_b32", ", vcc"))
VKERNEL(k_lshl_add, BLK8_3("v_lshl_add_u32", ""))
VKERNEL(k_xad, BLK8_3("v_xad_u32", ""))
VKERNEL(k_med3_i32, BLK8_3("v_med3_i32", ""))
VKERNEL(k_pk_lshlrev_b16, BLK8("v_pk_lshlrev_b16", ""))
VKERNEL(k_mov_b32, "v_mov_b32 %0, %8\n\tv_mov_b32 %1, %8\n\tv_mov_b32 %2, %8\n\tv_mov_b32 %3, %8\n\tv_mov_b32 %4, %8\n\tv_mov_b32 %5, %8\n\tv_mov_b32 %6, %8\n\tv_mov_b32 %7, %8\n\t")
VKERNEL(k_cmp_vcc, "v_cmp_lt_u32 vcc, %0, %8\n\tv_cmp_lt_u32 vcc, %1, %8\n\tv_cmp_lt_u32 vcc, %2, %8\n\tv_cmp_lt_u32 vcc, %3, %8\n\tv_cmp_lt_u32 vcc, %4, %8\n\tv_cmp_lt_u32 vcc, %5, %8\n\tv_cmp_lt_u32 vcc, %6, %8\n\tv_cmp_lt_u32 vcc, %7, %8\n\t")
VKERNEL(k_cvt_ubyte, "v_cvt_f32_ubyte0 %0, %8\n\tv_cvt_f32_ubyte1 %1, %8\n\tv_cvt_f32_ubyte2 %2, %8\n\tv_cvt_f32_ubyte3 %3, %8\n\tv_cvt_f32_ubyte0 %4, %8\n\tv_cvt_f32_ubyte1 %5, %8\n\tv_cvt_f32_ubyte2 %6, %8\n\tv_cvt_f32_ubyte3 %7, %8\n\t")
/* DPP / SDWA forms as the FAST kernel uses them */
VKERNEL(k_add_dpp, BLK8("v_add_u32_dpp", " row_shr:1 row_mask:0xf bank_mask:0xf"))
VKERNEL(k_min_sdwa, BLK8("v_min_u16_sdwa", " dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_0"))

/* v_cmp into SGPR pairs: 8 independent destinations */
__global__ void __launch_bounds__(256) k_cmp_sgpr(uint32_t* out, WaveRec* rec, int iters) {
    uint32_t a = threadIdx.x, s = 77;
    unsigned long long m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0, m5 = 0, m6 = 0, m7 = 0;
    PROLOGUE();
    for (int i = 0; i < iters; i++)
        asm volatile(R4("v_cmp_lt_u32 %0, %8, %9\n\tv_cmp_gt_u32 %1, %8, %9\n\tv_cmp_lt_u32 %2, %8, %9\n\tv_cmp_gt_u32 %3, %8, %9\n\t"
                        "v_cmp_lt_u32 %4, %8, %9\n\tv_cmp_gt_u32 %5, %8, %9\n\tv_cmp_lt_u32 %6, %8, %9\n\tv_cmp_gt_u32 %7, %8, %9\n\t")
                     : "+s"(m0), "+s"(m1), "+s"(m2), "+s"(m3), "+s"(m4), "+s"(m5), "+s"(m6), "+s"(m7) : "v"(a), "v"(s));
    EPILOGUE((uint32_t)(m0 + m1 + m2 + m3 + m4 + m5 + m6 + m7));
}

/* packed f32: 64-bit register pairs */
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) k_pk_fma_f32(uint32_t* out, WaveRec* rec, int iters) {
    f2 a = {1.f, 2.f}, b = a, c = a, d = a, e = a, f = a, g = a, h = a, s = {1.0001f, 0.9999f}, s2 = {0.5f, 0.25f};
    a.x = (float)threadIdx.x;
    PROLOGUE();
    for (int i = 0; i < iters; i++)
        asm volatile(R4(BLK8_3("v_pk_fma_f32", "")) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(s), "v"(s2));
    EPILOGUE((uint32_t)(a.x + b.x + c.x + d.x + e.x + f.x + g.x + h.x));
}

/* 64-bit integer multiply-adds (what size_t address arithmetic compiles to) and FP64: 64-bit register pairs */
__global__ void __launch_bounds__(256) k_mad_u64_u32(uint32_t* out, WaveRec* rec, int iters) {
    unsigned long long a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3, e = a + 4, f = a + 5, g = a + 6, h = a + 7;
    uint32_t s = 0x00030005u, s2 = 0x07060504u;
    PROLOGUE();
    for (int i = 0; i < iters; i++)
        asm volatile(R4("v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\tv_mad_u64_u32 %2, vcc, %8, %9, %2\n\t"
                        "v_mad_u64_u32 %3, vcc, %8, %9, %3\n\tv_mad_u64_u32 %4, vcc, %8, %9, %4\n\tv_mad_u64_u32 %5, vcc, %8, %9, %5\n\t"
                        "v_mad_u64_u32 %6, vcc, %8, %9, %6\n\tv_mad_u64_u32 %7, vcc, %8, %9, %7\n\t")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(s), "v"(s2) : "vcc");
    EPILOGUE((uint32_t)(a + b + c + d + e + f + g + h));
}
__global__ void __launch_bounds__(256) k_mad_i64_i32(uint32_t* out, WaveRec* rec, int iters) {
    unsigned long long a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3, e = a + 4, f = a + 5, g = a + 6, h = a + 7;
    uint32_t s = 0x00030005u, s2 = 0x07060504u;
    PROLOGUE();
    for (int i = 0; i < iters; i++)
        asm volatile(R4("v_mad_i64_i32 %0, vcc, %8, %9, %0\n\tv_mad_i64_i32 %1, vcc, %8, %9, %1\n\tv_mad_i64_i32 %2, vcc, %8, %9, %2\n\t"
                        "v_mad_i64_i32 %3, vcc, %8, %9, %3\n\tv_mad_i64_i32 %4, vcc, %8, %9, %4\n\tv_mad_i64_i32 %5, vcc, %8, %9, %5\n\t"
                        "v_mad_i64_i32 %6, vcc, %8, %9, %6\n\tv_mad_i64_i32 %7, vcc, %8, %9, %7\n\t")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(s), "v"(s2) : "vcc");
    EPILOGUE((uint32_t)(a + b + c + d + e + f + g + h));
}
__global__ void __launch_bounds__(256) k_fma_f64(uint32_t* out, WaveRec* rec, int iters) {
    double a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3, e = a + 4, f = a + 5, g = a + 6, h = a + 7, s = 1.0000001, s2 = 0.5;
    PROLOGUE();
    for (int i = 0; i < iters; i++)
        asm volatile(R4(BLK8_3("v_fma_f64", "")) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(s), "v"(s2));
    EPILOGUE((uint32_t)(a + b + c + d + e + f + g + h));
}
__global__ void __launch_bounds__(256) k_lshl_add_u64(uint32_t* out, WaveRec* rec, int iters) {
    unsigned long long a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3, e = a + 4, f = a + 5, g = a + 6, h = a + 7, s = 12345;
    PROLOGUE();
    for (int i = 0; i < iters; i++)
        asm volatile(R4("v_lshl_add_u64 %0, %0, 1, %8\n\tv_lshl_add_u64 %1, %1, 1, %8\n\tv_lshl_add_u64 %2, %2, 1, %8\n\tv_lshl_add_u64 %3, %3, 1, %8\n\t"
                        "v_lshl_add_u64 %4, %4, 1, %8\n\tv_lshl_add_u64 %5, %5, 1, %8\n\tv_lshl_add_u64 %6, %6, 1, %8\n\tv_lshl_add_u64 %7, %7, 1, %8\n\t")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(s));
    EPILOGUE((uint32_t)(a + b + c + d + e + f + g + h));
}
VKERNEL(k_mul_hi_u32, BLK8("v_mul_hi_u32", ""))
VKERNEL(k_mul_hi_u24, BLK8("v_mul_hi_u32_u24", ""))
VKERNEL(k_dot2_i16, BLK8_3("v_dot2_i32_i16", ""))
VKERNEL(k_dot2_u16, BLK8_3("v_dot2_u32_u16", ""))

/* SALU: 8 independent SGPR chains */
__global__ void __launch_bounds__(256) k_salu(uint32_t* out, WaveRec* rec, int iters) {
    uint32_t a = blockIdx.x, b = a + 1, c = a + 2, d = a + 3, e = a + 4, f = a + 5, g = a + 6, h = a + 7, s = 3;
    PROLOGUE();
    for (int i = 0; i < iters; i++)
        asm volatile(R4("s_add_u32 %0, %0, %8\n\ts_and_b32 %1, %1, %8\n\ts_or_b32 %2, %2, %8\n\ts_xor_b32 %3, %3, %8\n\t"
                        "s_add_u32 %4, %4, %8\n\ts_and_b32 %5, %5, %8\n\ts_or_b32 %6, %6, %8\n\ts_xor_b32 %7, %7, %8\n\t")
                     : "+s"(a), "+s"(b), "+s"(c), "+s"(d), "+s"(e), "+s"(f), "+s"(g), "+s"(h) : "s"(s) : "scc");
    EPILOGUE(a + b + c + d + e + f + g + h);
}
/* 32 VALU (v_add_u32) + 32 SALU per block, interleaved one to one, all independent */
__global__ void __launch_bounds__(256) k_valu_salu(uint32_t* out, WaveRec* rec, int iters) {
    uint32_t a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3, s = 5;
    uint32_t sa = blockIdx.x, sb = sa + 1, sc = sa + 2, sd = sa + 3, ss = 3;
    PROLOGUE();
    for (int i = 0; i < iters; i++)
        asm volatile(R4(R4("v_add_u32 %0, %0, %8\n\ts_add_u32 %4, %4, %9\n\tv_add_u32 %1, %1, %8\n\ts_and_b32 %5, %5, %9\n\t"
                        "v_add_u32 %2, %2, %8\n\ts_or_b32 %6, %6, %9\n\tv_add_u32 %3, %3, %8\n\ts_xor_b32 %7, %7, %9\n\t") "")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(sa), "+s"(sb), "+s"(sc), "+s"(sd) : "v"(s), "s"(ss) : "scc");
    EPILOGUE(a + b + c + d + sa + sb + sc + sd);
}
/* the mix of round 2's row: two packed + two plain per four instructions */
VKERNEL(k_mix_r02, "v_pk_min_u16 %0, %0, %8\n\tv_pk_max_u16 %1, %1, %8\n\tv_and_b32 %2, %2, %8\n\tv_or_b32 %3, %3, %8\n\t"
                   "v_pk_min_u16 %4, %4, %8\n\tv_pk_max_u16 %5, %5, %8\n\tv_and_b32 %6, %6, %8\n\tv_or_b32 %7, %7, %8\n\t")

/* LDS byte reads beside VALU, as in FAST: 32 ds_read_u8 (+ 32 v_add_u32) per block */
template <bool WITH_VALU>
__global__ void __launch_bounds__(256) k_lds_u8(uint32_t* out, WaveRec* rec, int iters, int stride) {
    __shared__ uint8_t sm[4 * 8192];
    uint8_t* my = sm + (threadIdx.x >> 6) * 8192;
    for (int i = threadIdx.x & 63; i < 8192; i += 64) my[i] = (uint8_t)i;
    __syncthreads();
    const uint32_t addr = (uint32_t)(my - sm) + (((threadIdx.x & 63) * stride) & 4095);
    uint32_t acc = 0, a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3, s = 5;
    PROLOGUE();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 32; k++) {
            uint32_t v;
            asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"((k * 53) & 4095));
            if (WITH_VALU) asm volatile("v_add_u32 %0, %0, %1" : "+v"((k & 3) == 0 ? a : (k & 3) == 1 ? b : (k & 3) == 2 ? c : d) : "v"(s));
            asm volatile("s_waitcnt lgkmcnt(15)");
            acc ^= v;
        }
    }
    EPILOGUE(acc + a + b + c + d);
}

struct Row { const char* name; void (*fn)(uint32_t*, WaveRec*, int); int per_block; };

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    uint32_t* out; CHECK(hipMalloc(&out, 64));
    const int maxw = 256 * 4 * 8;
    WaveRec* rec; CHECK(hipMalloc(&rec, sizeof(WaveRec) * maxw));
    std::vector<WaveRec> h(maxw);
    hipDeviceProp_t pr; CHECK(hipGetDeviceProperties(&pr, 0));
    printf("device %s, %d CUs, reported clock %d kHz; iters %d x 32 instructions per wave\n", pr.gcnArchName, pr.multiProcessorCount, pr.clockRate, iters);
    printf("cyc/instr chip-wide = (last wave end - first wave start, 100 MHz s_memrealtime) x shader MHz / 100 / (instructions per wave x waves per SIMD); median wave = the same from one wave's own s_memtime span\n");
    /* argv[2] = "mul": only the multiply / 64-bit / FP64 rows added in round 4 (what address arithmetic compiles to) */
    const bool only_mul = argc > 2 && std::string(argv[2]) == "mul";
    const Row rows_mul[] = {
        {"v_add_u32", k_add_u32, 32}, {"v_mul_u32_u24", k_mul_u24, 32}, {"v_mad_u32_u24", k_mad_u32_u24, 32}, {"v_mul_hi_u32_u24", k_mul_hi_u24, 32},
        {"v_mul_lo_u32", k_mul_lo, 32}, {"v_mul_hi_u32", k_mul_hi_u32, 32}, {"v_mad_u64_u32", k_mad_u64_u32, 32}, {"v_mad_i64_i32", k_mad_i64_i32, 32},
        {"v_lshl_add_u64", k_lshl_add_u64, 32}, {"v_fma_f32", k_fma_f32, 32}, {"v_fma_f64", k_fma_f64, 32}, {"v_dot2_i32_i16", k_dot2_i16, 32},
        {"v_dot2_u32_u16", k_dot2_u16, 32}, {"v_dot4_u32_u8", k_dot4_u8, 32},
    };
    const Row rows[] = {
        {"v_add_u32", k_add_u32, 32}, {"v_and_b32", k_and_b32, 32}, {"v_xor_b32", k_xor_b32, 32}, {"v_min_u32", k_min_u32, 32},
        {"v_min3_i32", k_min3_i32, 32}, {"v_max3_u32", k_max3_u32, 32}, {"v_pk_min_u16", k_pk_min_u16, 32}, {"v_pk_max_u16", k_pk_max_u16, 32},
        {"v_pk_sub_u16 clamp", k_pk_sub_u16c, 32}, {"v_pk_add_u16", k_pk_add_u16, 32}, {"v_min_u16", k_min_u16, 32},
        {"v_max_u16", k_max_u16, 32}, {"v_min_i16", k_min_i16, 32}, {"v_sub_u16", k_sub_u16, 32}, {"v_add_u16", k_add_u16, 32},
        {"v_or_b32", k_or_b32, 32}, {"v_sub_u32", k_sub_u32, 32}, {"v_max_u32", k_max_u32, 32}, {"v_min_i32", k_min_i32, 32},
        {"v_lshlrev_b32", k_lshlrev, 32}, {"v_lshrrev_b32", k_lshrrev, 32}, {"v_ashrrev_i32", k_ashrrev, 32},
        {"v_min_f32", k_min_f32, 32}, {"v_max_f32", k_max_f32, 32}, {"v_mul_f32", k_mul_f32, 32}, {"v_sub_f32", k_sub_f32, 32},
        {"v_cndmask_b32", k_cndmask, 32}, {"v_lshl_add_u32", k_lshl_add, 32}, {"v_xad_u32", k_xad, 32}, {"v_med3_i32", k_med3_i32, 32},
        {"v_pk_lshlrev_b16", k_pk_lshlrev_b16, 32},
        {"v_mov_b32", k_mov_b32, 32}, {"v_cmp -> vcc", k_cmp_vcc, 32}, {"v_cvt_f32_ubyteN", k_cvt_ubyte, 32},
        {"v_min_u16 sdwa", k_min_sdwa, 32}, {"v_add_u32 dpp row_shr", k_add_dpp, 32}, {"v_perm_b32", k_perm_b32, 32},
        {"v_alignbyte_b32", k_alignbyte, 32}, {"v_bfe_u32", k_bfe_u32, 32}, {"v_lshl_or_b32", k_lshl_or, 32}, {"v_and_or_b32", k_and_or, 32},
        {"v_add3_u32", k_add3, 32}, {"v_mad_u32_u24", k_mad_u32_u24, 32}, {"v_mul_u32_u24", k_mul_u24, 32}, {"v_mul_lo_u32", k_mul_lo, 32},
        {"v_bcnt_u32_b32", k_bcnt, 32}, {"v_mbcnt_lo", k_mbcnt, 32}, {"v_sad_u16", k_sad_u16, 32}, {"v_sad_u8", k_sad_u8, 32},
        {"v_dot4_u32_u8", k_dot4_u8, 32}, {"v_cmp -> sgpr pair", k_cmp_sgpr, 32}, {"v_add_f32", k_add_f32, 32}, {"v_fma_f32", k_fma_f32, 32},
        {"v_pk_fma_f32", k_pk_fma_f32, 32}, {"mix r02 (2 pk + 2 plain)", k_mix_r02, 32}, {"s_add/and/or/xor", k_salu, 32},
        {"v_add_u32 + salu 1:1 (pairs)", k_valu_salu, 64},
    };
    auto run = [&](const char* name, int wps, int per_block, int iters, auto launch) {
        const int nwg = 256 * wps, nw = nwg * 4;
        CHECK(hipMemset(rec, 0, sizeof(WaveRec) * maxw));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        launch(nwg); CHECK(hipGetLastError()); CHECK(hipDeviceSynchronize()); /* warm-up */
        CHECK(hipEventRecord(e0)); launch(nwg); CHECK(hipGetLastError()); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(h.data(), rec, sizeof(WaveRec) * nw, hipMemcpyDeviceToHost));
        std::vector<double> cyc, mhz; std::map<unsigned, int> per_simd;
        unsigned long long rmin = ~0ull, rmax = 0;
        for (int i = 0; i < nw; i++) {
            if (!h[i].cyc) continue;
            rmin = std::min(rmin, h[i].r0); rmax = std::max(rmax, h[i].r1);
            cyc.push_back((double)h[i].cyc);
            if (h[i].rt) mhz.push_back((double)h[i].cyc / (double)h[i].rt * 100.0);
            /* HW_ID: simd [5:4], cu [11:8], sh [12], se [15:13] (+ xcc from the dispatch: unknown here, counted modulo) */
            per_simd[(h[i].hwid >> 4) & 0xFFF]++;
        }
        if (cyc.empty()) { printf("%-30s wps %d: NO WAVE RECORDS (kernel did not run)\n", name, wps); return; }
        std::sort(cyc.begin(), cyc.end()); std::sort(mhz.begin(), mhz.end());
        /* chip-wide: first wave's start to last wave's end on the 100 MHz clock, converted with the measured shader clock;
         * every SIMD issued iters * 32 * wps instructions (pairs) in that span */
        const double clk = mhz.empty() ? 0.0 : mhz[mhz.size() / 2];
        const double span_cyc = (double)(rmax - rmin) * clk / 100.0;
        const double per = span_cyc / ((double)iters * 32 * wps);
        printf("%-30s wps %d: %6.2f cyc/%s chip-wide  (median wave %.2f)  clock %4.0f MHz  event %.3f ms  waves %zu\n", name, wps, per,
               per_block == 64 ? "pair " : "instr", cyc[cyc.size() / 2] / ((double)iters * 32 * wps), clk, ms, cyc.size());
    };
    if (only_mul) {
        for (const Row& r : rows_mul)
            for (int wps : {1, 2, 4, 8})
                run(r.name, wps, r.per_block, iters, [&](int nwg) { hipLaunchKernelGGL(r.fn, dim3(nwg), dim3(256), 0, 0, out, rec, iters); });
        return 0;
    }
    for (const Row& r : rows)
        for (int wps : {1, 2, 4, 8})
            run(r.name, wps, r.per_block, iters, [&](int nwg) { hipLaunchKernelGGL(r.fn, dim3(nwg), dim3(256), 0, 0, out, rec, iters); });
    for (int stride : {1, 4, 48, 67})
        for (int wps : {1, 2, 4}) {
            char nm[64];
            snprintf(nm, sizeof nm, "ds_read_u8 stride %d", stride);
            run(nm, wps, 32, iters / 4, [&](int nwg) { hipLaunchKernelGGL(k_lds_u8<false>, dim3(nwg), dim3(256), 0, 0, out, rec, iters / 4, stride); });
            snprintf(nm, sizeof nm, "ds_read_u8 s%d + v_add 1:1", stride);
            run(nm, wps, 32, iters / 4, [&](int nwg) { hipLaunchKernelGGL(k_lds_u8<true>, dim3(nwg), dim3(256), 0, 0, out, rec, iters / 4, stride); });
        }
    return 0;
}
