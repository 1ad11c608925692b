/* vslam_kernels.hip -- hand-written gfx950 kernels of the ORB front-end.
 *
 * All integer stages are exact; the few float operations (fastAtan2 polynomial, pattern rotation) are
 * written with explicit non-fused intrinsics and the file is built with -ffp-contract=off.
 * wave = 64 everywhere.  Reference lines each kernel reproduces are cited at the kernel.
 */
#include "vslam_kernels.h"
#include "vslam_wave.h"

#include "../../include/vslam_orb_pattern.h"
#include "vslam_trig.h"

/* ------------------------------------------------------------------------------------------------
 * helpers
 * ---------------------------------------------------------------------------------------------- */
__device__ __forceinline__ const uint8_t* level_base(const uint8_t* pyr, size_t slot_stride,
                                                     const BatchSrc& src, const LevelGeom& lg, int level,
                                                     int slot, int* pitch) {
    if (level == 0) {
        *pitch = (int)src.pitch0[slot];
        return src.l0[slot];
    }
    *pitch = lg.pitch;
    return pyr + (size_t)slot * slot_stride + lg.off;
}

/* ------------------------------------------------------------------------------------------------
 * K1  pyramid level l from level l-1: cv::resize(INTER_LINEAR) 8u, fixed point 11 bits
 *     (FExtractor::ComputePyramid, fextractor.cpp:1135-1160 -> OpenCV resizeGeneric_ HResizeLinear /
 *     VResizeLinear<uchar,int,short>).  One thread per output pixel; tables hold, per dx, the two source
 *     columns and the two short coefficients, per dy the two clipped source rows and coefficients.
 * ---------------------------------------------------------------------------------------------- */
__global__ void __launch_bounds__(256)
k_resize_level(uint8_t* pyr, size_t slot_stride, BatchSrc src, LevelGeom sg, LevelGeom dg, int src_level,
               const uint16_t* __restrict__ xtab, const int16_t* __restrict__ xa,
               const uint16_t* __restrict__ ytab, const int16_t* __restrict__ yb) {
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int slot = blockIdx.z;
    if (dx >= dg.w || dy >= dg.h) return;
    int spitch;
    const uint8_t* S = level_base(pyr, slot_stride, src, sg, src_level, slot, &spitch);
    const int sx0 = xtab[2 * dx], sx1 = xtab[2 * dx + 1];
    const int a0 = xa[2 * dx], a1 = xa[2 * dx + 1];
    const int sy0 = ytab[2 * dy], sy1 = ytab[2 * dy + 1];
    const int b0 = yb[2 * dy], b1 = yb[2 * dy + 1];
    const uint8_t* r0 = S + (size_t)sy0 * spitch;
    const uint8_t* r1 = S + (size_t)sy1 * spitch;
    const int h0 = r0[sx0] * a0 + r0[sx1] * a1;
    const int h1 = r1[sx0] * a0 + r1[sx1] * a1;
    const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
    uint8_t* D = pyr + (size_t)slot * slot_stride + dg.off;
    D[(size_t)dy * dg.pitch + dx] = (uint8_t)v;
}

/* ------------------------------------------------------------------------------------------------
 * K4  orientation + descriptor, one wave per keypoint
 *     IC_Angle (fextractor.cpp:68-95): int32 moments over the radius-15 disc of the UN-blurred level,
 *     cv::fastAtan2 polynomial in float; computeOrbDescriptor (fextractor.cpp:99-138): rotate the 512
 *     pattern points with a=cosf, b=sinf (glibc-exact, vslam_trig.h), cvRound (round-half-even), sample
 *     the BLURRED level, 256 comparisons -> 4 wave ballots = 32 bytes.
 * ---------------------------------------------------------------------------------------------- */
/* IC_Angle weights for the dword form of the moment sum: the 31 x 31 patch is read as 31 rows x 8 unaligned dwords
 * (columns u = -15 .. 16); item i = row * 8 + dword has two packed byte weights per pixel: wm = 1 inside the radius-15
 * disc (|u| <= umax[|v|], fextractor.cpp:75-92), wu = (u + 15) inside, both 0 outside.  Then
 *   m10 = sum u I = sum (u + 15) I - 15 sum I,   m01 = sum v I = sum_rows v * (sum_row I)      (exact int32). */
__device__ uint32_t g_mom_wu[256];
__device__ uint32_t g_mom_wm[256];
__device__ uint32_t g_mom_wr[256]; /* row (0 .. 30) in every byte inside the disc: sum of row * I by the same dot4 */

void vk_upload_disc(const int umax[16]) {
    uint32_t wu[256], wm[256], wr[256];
    for (int i = 0; i < 256; i++) {
        const int row = i >> 3, dw = i & 7, v = row - 15;
        wu[i] = wm[i] = wr[i] = 0;
        if (row > 30) continue;
        for (int j = 0; j < 4; j++) {
            const int u = 4 * dw + j - 15;
            const bool inside = u <= 15 && abs(u) <= umax[abs(v)];
            if (inside) {
                wu[i] |= (uint32_t)(u + 15) << (8 * j);
                wm[i] |= 1u << (8 * j);
                wr[i] |= (uint32_t)row << (8 * j);
            }
        }
    }
    hipMemcpyToSymbol(HIP_SYMBOL(g_mom_wu), wu, sizeof(wu));
    hipMemcpyToSymbol(HIP_SYMBOL(g_mom_wm), wm, sizeof(wm));
    hipMemcpyToSymbol(HIP_SYMBOL(g_mom_wr), wr, sizeof(wr));
}

__device__ __forceinline__ float fast_atan2_deg(float y, float x, int fma) {
    /* OpenCV mathfuncs_core atan_f32(); constants folded in float exactly as the C source does */
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    const float eps = (float)2.2204460492503131e-16;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, eps));
        c2 = __fmul_rn(c, c);
        if (fma) a = __fmul_rn(__fmaf_rn(__fmaf_rn(__fmaf_rn(p7, c2, p5), c2, p3), c2, p1), c);
        else
            a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, eps));
        c2 = __fmul_rn(c, c);
        if (fma) a = __fmaf_rn(-__fmaf_rn(__fmaf_rn(__fmaf_rn(p7, c2, p5), c2, p3), c2, p1), c, 90.f);
        else
            a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

/* One wave = one keypoint at a time, DESC_KPW keypoints one after the other, software-pipelined: the global loads of
 * keypoint k+1 (its raw 31x31 patch as 4 dwords per lane, its blurred 37x37 support as 6 dwords per lane) are issued
 * before keypoint k is computed, so the memory latency hides behind the trigonometry and the sampling of the previous
 * keypoint.
 *   moments   : v_dot4_u32_u8 of the raw dwords against the lane's packed disc weights (registers, fixed per lane).
 *   descriptor: the blurred support goes to the wave's LDS tile (37 rows x 40 bytes, coalesced dword rows); the 512
 *               rotated pattern points (|x|,|y| <= 13 -> radius <= 18 after rotation) are sampled from LDS.  The
 *               previous version gathered 749 + 512 single bytes per keypoint from global memory: ~300 cache-line
 *               accesses per keypoint against ~90 now.
 * Keypoints are >= 19 px from the level border (EDGE_THRESHOLD), so every patch byte is inside the level; the dword
 * rows may overrun the 37-px support by up to 3 bytes, still inside the row (or its padding). */
#ifndef DESC_KPW
#define DESC_KPW 4          /* keypoints per wave (2 and 8 measured: see DESIGN.md section 8) */
#endif
#define DESC_TP 40          /* LDS tile pitch: 10 dwords cover 18 + 18 + 1 columns from a dword-aligned start */
#define DESC_TROWS 37
#define DESC_TILE_BYTES (DESC_TP * DESC_TROWS + 8)

/* everything a keypoint needs from the kernel-argument tables (level geometry, level-0 source of its slot), looked up
 * ONCE and unconditionally per keypoint of the wave: look-ups under the `k < nk` conditions below make the compiler copy
 * both argument structs to scratch memory (1.3 KB per lane) */
struct DescAddr {
    const uint8_t* raw; /* top-left of the raw 31x31 patch */
    const uint8_t* blr; /* dword-aligned start of the blurred 37-row support */
    int pitch, bpitch;
    float scale;
};
__device__ __forceinline__ DescAddr desc_addr(const uint8_t* __restrict__ pyr, const uint8_t* __restrict__ blur,
                                              size_t slot_stride, const BatchSrc& src, const PyramidGeom& g, const SelKp s) {
    const LevelGeom lg = g.lv[s.level];
    DescAddr A;
    const uint8_t* img = level_base(pyr, slot_stride, src, lg, s.level, s.slot, &A.pitch);
    A.raw = img + (size_t)((int)s.y - 15) * A.pitch + ((int)s.x - 15);
    A.blr = blur + (size_t)s.slot * slot_stride + lg.off + (size_t)((int)s.y - 18) * lg.pitch + (((int)s.x - 18) & ~3);
    A.bpitch = lg.pitch;
    A.scale = lg.scale;
    return A;
}
__device__ __forceinline__ void desc_issue_raw(const DescAddr& A, int lane, uint32_t raw[4]) {
#pragma unroll
    for (int k = 0; k < 4; k++) { /* item i = lane + 64 k -> row i >> 3, dword i & 7 (rows 0..30 of 31) */
        const int i = lane + 64 * k;
        raw[k] = 0;
        if (i < 248) raw[k] = *(const uint32_t*)(A.raw + mad24u_s((uint32_t)(i >> 3), (uint32_t)A.pitch, 4u * (uint32_t)(i & 7))); /* unaligned dword */
    }
}
__device__ __forceinline__ void desc_issue_blur(const DescAddr& A, int lane, uint32_t blr[6]) {
#pragma unroll
    for (int k = 0; k < 6; k++) { /* item i = lane + 64 k -> row i / 10, dword i % 10 (370 items) */
        const int i = lane + 64 * k;
        const int row = (i * 205) >> 11; /* i / 10 for i < 1024 */
        blr[k] = 0;
        if (i < 370) blr[k] = *(const uint32_t*)(A.blr + mad24u_s((uint32_t)row, (uint32_t)A.bpitch, (uint32_t)mad24i(row, -10, i) << 2));
    }
}

/* IC_Angle moments of one keypoint: every lane ends up with the wave's sums.  Three dot4 chains per lane -- sum of (u + 15) I,
 * of row I and of I over the lane's four dwords (the row of a dword is a per-lane constant, so it sits in the byte weights
 * wr = row * wm) -- then m10 = A - 15 S, m01 = R - 15 S: twelve dot4 and three adds instead of eight dot4, eight multiplies
 * and their adds (and no hand-placed instruction right behind a dot4: it would sit inside the dot -> VALU wait states only
 * the compiler keeps track of) */
__device__ __forceinline__ void desc_moments(const uint32_t raw[4], const uint32_t wu[4], const uint32_t wm[4],
                                             const uint32_t wr[4], int* m01_out, int* m10_out) {
    uint32_t A = 0, R = 0, S = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        A = __builtin_amdgcn_udot4(raw[k], wu[k], A, false);
        R = __builtin_amdgcn_udot4(raw[k], wr[k], R, false);
        S = __builtin_amdgcn_udot4(raw[k], wm[k], S, false);
    }
    const int s15 = (int)((S << 4) - S);
    int m10 = (int)A - s15, m01 = (int)R - s15;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        m10 += __shfl_xor(m10, o, 64);
        m01 += __shfl_xor(m01, o, 64);
    }
    *m01_out = m01;
    *m10_out = m10;
}

/* descriptor of one keypoint from its blurred support (already in registers) and its (angle, cos, sin) */
__device__ __forceinline__ void desc_sample(const DescAddr& A, const SelKp s, const uint32_t blr[6], uint8_t* tile,
                                            const char4 pat[4], float angle, float a, float b, vslam_kp* kps, uint8_t* desc,
                                            int cap, int lane) {
    /* blurred support -> LDS (LDS operations of one wave execute in order: no barrier between these stores and the
     * sampling reads below, nor between the reads of one keypoint and the stores of the next) */
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const int i = lane + 64 * k;
        if (i < 370) ((uint32_t*)tile)[i] = blr[k];
    }
    const int xo = 18 + ((((int)s.x - 18) & 3)) + 18 * DESC_TP; /* tile byte of the keypoint centre */
    unsigned long long w[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const char4 pt = pat[q];
        const float x0 = (float)pt.x, y0 = (float)pt.y, x1 = (float)pt.z, y1 = (float)pt.w;
        const int r0 = __float2int_rn(__fadd_rn(__fmul_rn(x0, b), __fmul_rn(y0, a)));
        const int c0 = __float2int_rn(__fsub_rn(__fmul_rn(x0, a), __fmul_rn(y0, b)));
        const int r1 = __float2int_rn(__fadd_rn(__fmul_rn(x1, b), __fmul_rn(y1, a)));
        const int c1 = __float2int_rn(__fsub_rn(__fmul_rn(x1, a), __fmul_rn(y1, b)));
        /* |r| <= 18: row * 40 + column as ONE v_mad_i32_i24 */
        const int t0 = tile[xo + mad24i(r0, DESC_TP, c0)], t1 = tile[xo + mad24i(r1, DESC_TP, c1)];
        w[q] = __ballot(t0 < t1);
    }
    vslam_kp* okp = kps + (size_t)s.slot * cap + s.out;
    if (lane == 0) { /* fextractor.cpp:828-838 (octave, size), :1114-1116 (pt *= scale) */
        vslam_kp o;
        o.x = s.level ? __fmul_rn((float)s.x, A.scale) : (float)s.x;
        o.y = s.level ? __fmul_rn((float)s.y, A.scale) : (float)s.y;
        o.size = (float)(int)__fmul_rn(31.f, A.scale);
        o.angle = angle;
        o.response = (float)s.response;
        o.octave = s.level;
        o.class_id = -1;
        *okp = o;
    }
    if (lane < 4) {
        unsigned long long* od = (unsigned long long*)(desc + ((size_t)s.slot * cap + s.out) * 32);
        od[lane] = lane == 0 ? w[0] : lane == 1 ? w[1] : lane == 2 ? w[2] : w[3];
    }
}

/* the per-lane constants of a wave: disc weights of its four raw-patch items, its four pattern point pairs */
__device__ __forceinline__ void desc_lane_tables(const int8_t* __restrict__ pattern, int lane, uint32_t wu[4],
                                                 uint32_t wm[4], uint32_t wr[4], char4 pat[4]) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
        wu[k] = g_mom_wu[lane + 64 * k];
        wm[k] = g_mom_wm[lane + 64 * k];
        wr[k] = g_mom_wr[lane + 64 * k];
        pat[k] = ((const char4*)pattern)[k * 64 + lane]; /* descriptor bit k*64 + lane */
    }
}

/* keypoints sel[k0 .. kend) of one list, at most DESC_KPW, on one wave -- in three phases:
 *   A  the raw patches of ALL the wave's keypoints are loaded (4 dwords per lane each) and reduced to their moments;
 *      lane k keeps keypoint k's (m01, m10);
 *   B  fastAtan2 and the glibc-exact cosf/sinf -- 40 + 81 double-precision instructions, a third of this kernel's
 *      issue time when they ran once per keypoint on 64 identical lanes -- run ONCE, lane k working on keypoint k;
 *   C  per keypoint: blurred support -> LDS tile, 512 rotated samples, 4 ballots; the support of keypoint k+1 is loaded
 *      while keypoint k is sampled. */
template <int KPW>
__device__ __forceinline__ void describe_run(const uint8_t* __restrict__ pyr, const uint8_t* __restrict__ blur,
                                             size_t slot_stride, const BatchSrc& src, const PyramidGeom& g,
                                             const SelKp* __restrict__ sel, int k0, int kend,
                                             const int8_t* __restrict__ pattern, vslam_kp* kps, uint8_t* desc, int cap,
                                             int atan_fma, uint8_t* tile) {
    const int lane = threadIdx.x & 63;
    if (k0 >= kend) return; /* wave-uniform */
    const int nk = kend - k0; /* 1 .. KPW, wave-uniform */
    uint32_t wu[4], wm[4], wr[4];
    char4 pat[4];
    desc_lane_tables(pattern, lane, wu, wm, wr, pat);
    SelKp s[KPW];
    DescAddr A[KPW];
    uint32_t raw[KPW][4];
#pragma unroll
    for (int k = 0; k < KPW; k++) { /* slots past the wave's last keypoint repeat it (looked up, never loaded) */
        s[k] = sel[k0 + min(k, nk - 1)];
        A[k] = desc_addr(pyr, blur, slot_stride, src, g, s[k]);
    }
#pragma unroll
    for (int k = 0; k < KPW; k++)
        if (k < nk) desc_issue_raw(A[k], lane, raw[k]);
    uint32_t blr[6], blrn[6];
    desc_issue_blur(A[0], lane, blr);
    int my01 = 0, my10 = 0;
#pragma unroll
    for (int k = 0; k < KPW; k++)
        if (k < nk) {
            int m01, m10;
            desc_moments(raw[k], wu, wm, wr, &m01, &m10);
            if (lane == k) {
                my01 = m01;
                my10 = m10;
            }
        }
    const float angle = fast_atan2_deg((float)my01, (float)my10, atan_fma);
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    const float rad = __fmul_rn(angle, factorPI);
    const float ca = vslam_trig::glibc_cosf(rad), sb = vslam_trig::glibc_sinf(rad);
#pragma unroll
    for (int k = 0; k < KPW; k++)
        if (k < nk) {
            if (k + 1 < nk) desc_issue_blur(A[min(k + 1, KPW - 1)], lane, blrn);
            const float ang_k = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(angle), k));
            const float a_k = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ca), k));
            const float b_k = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sb), k));
            desc_sample(A[k], s[k], blr, tile, pat, ang_k, a_k, b_k, kps, desc, cap, lane);
            if (k + 1 < nk) {
#pragma unroll
                for (int j = 0; j < 6; j++) blr[j] = blrn[j];
            }
        }
}

/* host-selected keypoints (quadtree on the host): one flat list for the batch */
__global__ void __launch_bounds__(256)
k_orient_describe(const uint8_t* __restrict__ pyr, const uint8_t* __restrict__ blur, size_t slot_stride,
                  BatchSrc src, PyramidGeom g, const SelKp* __restrict__ sel, int nsel,
                  const int8_t* __restrict__ pattern, vslam_kp* kps, uint8_t* desc, int cap, int atan_fma) {
    __shared__ __align__(16) uint8_t s_tile[4][DESC_TILE_BYTES];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); /* scalar keypoint records, as in the _dev form */
    const int k0 = (blockIdx.x * 4 + wave) * DESC_KPW;
    describe_run<DESC_KPW>(pyr, blur, slot_stride, src, g, sel, k0, min(k0 + DESC_KPW, nsel), pattern, kps, desc, cap, atan_fma,
                           s_tile[wave]);
}

/* device-selected keypoints (k_octree + k_assign_out): per-slot lists, counts read from HBM.  KPW keypoints per wave:
 * DESC_KPW for batches (2 and 8 measured: -6 % / -10 % in the pipeline), ONE for one or two images, where the launch is a
 * frame's latency and four times the waves finish sooner (batch-1 latency -4 us) */
#ifndef DESC_WPB
#define DESC_WPB 4 /* waves per workgroup of k_orient_describe_dev (independent of each other; 1 and 2 measured) */
#endif
template <int KPW>
__global__ void __launch_bounds__(64 * DESC_WPB)
k_orient_describe_dev(const uint8_t* __restrict__ pyr, const uint8_t* __restrict__ blur, size_t slot_stride,
                      BatchSrc src, PyramidGeom g, const SelKp* __restrict__ sel,
                      const int32_t* __restrict__ slot_counts, const int8_t* __restrict__ pattern, vslam_kp* kps,
                      uint8_t* desc, int cap, int atan_fma, int bps, int nwork, int prio) {
    __shared__ __align__(16) uint8_t s_tile[DESC_WPB][DESC_TILE_BYTES];
    wave_prio_raise(prio);
    /* XCD-aware order: workgroups b and b+8 share an XCD/L2.  The (slot, keypoint-block) work list is
     * slot-major and level-major inside a slot, so handing XCD k the k-th contiguous eighth keeps one image
     * (or a few of its levels) per L2 instead of streaming every pyramid through all eight. */
    const int per_xcd = (nwork + 7) >> 3;
    const int w = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (w >= nwork) return;
    const int slot = w / bps;
    /* wave-uniform, and said so: the keypoint records then come through scalar loads and every address derived from them is
     * scalar arithmetic instead of 64-bit vector multiply-add chains in 4 x 30 more VGPRs */
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int k0 = ((w - slot * bps) * DESC_WPB + wave) * KPW;
    const int n = slot_counts[slot * 4];
    describe_run<KPW>(pyr, blur, slot_stride, src, g, sel + (size_t)slot * cap, k0, min(k0 + KPW, n), pattern, kps, desc,
                      cap, atan_fma, s_tile[wave]);
}

/* ------------------------------------------------------------------------------------------------
 * K5  all-pairs 256-bit Hamming (FMatcher::DescriptorDistance, fmatcher.cpp:2859-2875)
 *     64 queries per workgroup (one per lane, 8 dwords in VGPRs); the 4 waves each sweep a quarter of a
 *     256-descriptor train tile staged in LDS (all lanes read the same address: LDS broadcast), xor +
 *     v_bcnt_u32_b32 accumulate.  Outputs either the dense u8 matrix or per-(query,tile) top-2 partials.
 * ---------------------------------------------------------------------------------------------- */
#define HAM_TQ 64
#define HAM_TT 256

__global__ void __launch_bounds__(256)
k_hamming_matrix(const uint32_t* __restrict__ q, int nq, const uint32_t* __restrict__ t, int nt,
                 uint8_t* __restrict__ out) {
    __shared__ uint4 s_t[HAM_TT * 2];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int qi = blockIdx.x * HAM_TQ + lane;
    const int t0 = blockIdx.y * HAM_TT;
    for (int i = tid; i < HAM_TT * 2; i += 256) {
        const int ti = t0 + (i >> 1);
        s_t[i] = ti < nt ? ((const uint4*)t)[(size_t)ti * 2 + (i & 1)] : make_uint4(0, 0, 0, 0);
    }
    uint4 qa = make_uint4(0, 0, 0, 0), qb = qa;
    if (qi < nq) {
        qa = ((const uint4*)q)[(size_t)qi * 2];
        qb = ((const uint4*)q)[(size_t)qi * 2 + 1];
    }
    __syncthreads();
    if (qi >= nq) return;
    for (int j = wv * 64; j < wv * 64 + 64; j++) {
        const int ti = t0 + j;
        if (ti >= nt) break;
        const uint4 ta = s_t[2 * j], tb = s_t[2 * j + 1];
        int d = __popc(qa.x ^ ta.x) + __popc(qa.y ^ ta.y) + __popc(qa.z ^ ta.z) + __popc(qa.w ^ ta.w) +
                __popc(qb.x ^ tb.x) + __popc(qb.y ^ tb.y) + __popc(qb.z ^ tb.z) + __popc(qb.w ^ tb.w);
        out[(size_t)qi * nt + ti] = (uint8_t)min(d, 255);
    }
}

/* diagnostics: evaluate the device float helpers on arrays (tests pin them against glibc / the oracle) */
__global__ void k_dbg_sincos(const float* __restrict__ x, int n, float* s, float* c) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    s[i] = vslam_trig::glibc_sinf(x[i]);
    c[i] = vslam_trig::glibc_cosf(x[i]);
}
__global__ void k_dbg_atan2(const float* __restrict__ y, const float* __restrict__ x, int n, int fma, float* a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = fast_atan2_deg(y[i], x[i], fma);
}
__global__ void k_dbg_logf(const float* __restrict__ x, int n, float* y) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = vslam_trig::glibc_logf(x[i]);
}
void vk_dbg_logf(hipStream_t st, const float* x, int n, float* y) {
    if (n > 0) hipLaunchKernelGGL(k_dbg_logf, dim3((n + 255) / 256), dim3(256), 0, st, x, n, y);
}
void vk_dbg_sincos(hipStream_t st, const float* x, int n, float* s, float* c) {
    if (n > 0) hipLaunchKernelGGL(k_dbg_sincos, dim3((n + 255) / 256), dim3(256), 0, st, x, n, s, c);
}
void vk_dbg_atan2(hipStream_t st, const float* y, const float* x, int n, int fma, float* a) {
    if (n > 0) hipLaunchKernelGGL(k_dbg_atan2, dim3((n + 255) / 256), dim3(256), 0, st, y, x, n, fma, a);
}

/* ------------------------------------------------------------------------------------------------
 * launch wrappers (plain functions so the host file needs no kernel syntax)
 * ---------------------------------------------------------------------------------------------- */
void vk_resize_level(hipStream_t st, uint8_t* pyr, size_t slot_stride, const BatchSrc& src,
                     const LevelGeom& sg, const LevelGeom& dg, int src_level, const uint16_t* xtab,
                     const int16_t* xa, const uint16_t* ytab, const int16_t* yb, int nslots) {
    dim3 grid((dg.w + 63) / 64, (dg.h + 3) / 4, nslots);
    hipLaunchKernelGGL(k_resize_level, grid, dim3(256), 0, st, pyr, slot_stride, src, sg, dg, src_level, xtab,
                       xa, ytab, yb);
}

void vk_orient_describe(hipStream_t st, const uint8_t* pyr, const uint8_t* blur, size_t slot_stride,
                        const BatchSrc& src, const PyramidGeom& g, const SelKp* sel, int nsel,
                        const int8_t* pattern, vslam_kp* kps, uint8_t* desc, int cap, int atan_fma) {
    if (nsel <= 0) return;
    const int per_wg = 4 * DESC_KPW;
    hipLaunchKernelGGL(k_orient_describe, dim3((nsel + per_wg - 1) / per_wg), dim3(256), 0, st, pyr, blur, slot_stride, src, g,
                       sel, nsel, pattern, kps, desc, cap, atan_fma);
}

void vk_orient_describe_dev(hipStream_t st, const uint8_t* pyr, const uint8_t* blur, size_t slot_stride,
                            const BatchSrc& src, const PyramidGeom& g, const SelKp* sel,
                            const int32_t* slot_counts, const int8_t* pattern, vslam_kp* kps, uint8_t* desc,
                            int cap, int atan_fma, int nslots, int prio, int kpw_override) {
    /* keypoints per wave: 4, one after the other with the next one's loads in flight (110 VGPRs, 4 waves per SIMD), for batches;
     * 1 (48 VGPRs, 8 waves per SIMD, four times the waves) for one or two images; vslam_tuning.desc_kpw forces either */
    const int kpw = kpw_override == 1 ? 1 : kpw_override == 2 ? 2 : kpw_override == DESC_KPW ? DESC_KPW : nslots <= 2 ? 1 : DESC_KPW;
    const int per_wg = DESC_WPB * kpw;
    const int bps = (cap + per_wg - 1) / per_wg, nwork = bps * nslots;
    if (kpw == 1)
        hipLaunchKernelGGL(k_orient_describe_dev<1>, dim3(((nwork + 7) / 8) * 8), dim3(64 * DESC_WPB), 0, st, pyr, blur, slot_stride,
                           src, g, sel, slot_counts, pattern, kps, desc, cap, atan_fma, bps, nwork, prio);
    else if (kpw == 2)
        hipLaunchKernelGGL(k_orient_describe_dev<2>, dim3(((nwork + 7) / 8) * 8), dim3(64 * DESC_WPB), 0, st, pyr, blur, slot_stride,
                           src, g, sel, slot_counts, pattern, kps, desc, cap, atan_fma, bps, nwork, prio);
    else
        hipLaunchKernelGGL(k_orient_describe_dev<DESC_KPW>, dim3(((nwork + 7) / 8) * 8), dim3(64 * DESC_WPB), 0, st, pyr, blur, slot_stride,
                           src, g, sel, slot_counts, pattern, kps, desc, cap, atan_fma, bps, nwork, prio);
}

void vk_hamming_matrix(hipStream_t st, const uint8_t* q, int nq, const uint8_t* t, int nt, uint8_t* out) {
    if (nq <= 0 || nt <= 0) return;
    dim3 grid((nq + HAM_TQ - 1) / HAM_TQ, (nt + HAM_TT - 1) / HAM_TT);
    hipLaunchKernelGGL(k_hamming_matrix, grid, dim3(256), 0, st, (const uint32_t*)q, nq, (const uint32_t*)t, nt,
                       out);
}

/* ------------------------------------------------------------------------------------------------
 * The brute-force matcher for P independent problems in ONE launch (the cv::BFMatcher::knnMatch(.., 2) sites,
 * frame.cpp:1167-1174: e.g. the stereo pairs of a step), shaped to be bound by the integer ALU rather than by launches:
 *   - a workgroup owns 64 queries (one per lane, 8 dwords in VGPRs, the same in all four waves) and a RANGE of the
 *     train descriptors (grid.y splits the train set only as far as the GPU needs workgroups);
 *   - the range is walked in tiles of 256 descriptors, double-buffered in LDS: the next tile's global loads are issued
 *     before the current tile is computed and stored after it, one barrier per tile;
 *   - wave w sweeps descriptors [64w, 64w + 64) of a tile: two broadcast ds_read_b128, 8 v_xor + 8 v_bcnt (the count
 *     accumulates in the instruction), key = dist << 16 | index, and the running two smallest keys by
 *     k2 = med3(k1, k2, key), k1 = min(k1, key): 19 VALU instructions per 64 pairs, 16 of them the arithmetic itself;
 *   - the four waves' pairs are merged through LDS, the ranges' by k_hamming_top2_merge_batch.
 * Ties resolve to the lower train index, as in the reference's ascending loops.
 * ---------------------------------------------------------------------------------------------- */
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
/* popcount that accumulates in the instruction (hipcc turns `d += __popc(x)` into eight plain v_bcnt and a tree of v_add3:
 * 24.5 instead of 19 instructions per 64 pairs) */
__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc) {
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}
__device__ __forceinline__ uint32_t top2_key(uint32_t dist, uint32_t s_index) { /* dist << 16 | index; index wave-uniform (SGPR) */
    uint32_t r;
    asm("v_lshl_or_b32 %0, %1, 16, %2" : "=v"(r) : "v"(dist), "s"(s_index));
    return r;
}
__device__ __forceinline__ uint32_t hamming256(const uint4& qa, const uint4& qb, const uint4& ta, const uint4& tb) {
    uint32_t d = bcnt_acc(qa.x ^ ta.x, 0u);
    d = bcnt_acc(qa.y ^ ta.y, d);
    d = bcnt_acc(qa.z ^ ta.z, d);
    d = bcnt_acc(qa.w ^ ta.w, d);
    d = bcnt_acc(qb.x ^ tb.x, d);
    d = bcnt_acc(qb.y ^ tb.y, d);
    d = bcnt_acc(qb.z ^ tb.z, d);
    return bcnt_acc(qb.w ^ tb.w, d);
}
__global__ void __launch_bounds__(256)
k_hamming_top2_batch(Top2Jobs jobs, int nsplit, uint32_t* __restrict__ part /* [row][nsplit][2] */) {
    __shared__ uint4 s_t[2][HAM_TT * 2];
    __shared__ uint32_t s_k[4][HAM_TQ][2];
    const Top2Job jb = jobs.job[blockIdx.z];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if ((int)blockIdx.x * HAM_TQ >= jb.nq) return; /* block-uniform: problems of different sizes share the grid */
    const int qi = blockIdx.x * HAM_TQ + lane;
    /* this workgroup's train range: whole tiles, the ranges of the nsplit workgroups cover [0, nt) */
    const int ntile = (jb.nt + HAM_TT - 1) / HAM_TT, tps = (ntile + nsplit - 1) / nsplit;
    const int tile0 = (int)blockIdx.y * tps, tile1 = min(tile0 + tps, ntile);
    uint4 qa = make_uint4(0, 0, 0, 0), qb = qa;
    if (qi < jb.nq) {
        qa = ((const uint4*)jb.q)[(size_t)qi * 2];
        qb = ((const uint4*)jb.q)[(size_t)qi * 2 + 1];
    }
    const uint4* T4 = (const uint4*)jb.t;
    const int n4 = jb.nt * 2; /* uint4 elements of the train array */
    uint4 r0 = make_uint4(0, 0, 0, 0), r1 = r0;
    auto fetch = [&](int tile) { /* two uint4 per thread: elements tid and tid + 256 of the tile's 512 */
        const int e0 = tile * (HAM_TT * 2) + tid, e1 = e0 + 256;
        r0 = e0 < n4 ? T4[e0] : make_uint4(0, 0, 0, 0);
        r1 = e1 < n4 ? T4[e1] : make_uint4(0, 0, 0, 0);
    };
    uint32_t k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;
    if (tile0 < tile1) {
        fetch(tile0);
        s_t[0][tid] = r0;
        s_t[0][tid + 256] = r1;
    }
    __syncthreads();
    for (int tile = tile0; tile < tile1; tile++) {
        const int b = (tile - tile0) & 1;
        if (tile + 1 < tile1) fetch(tile + 1); /* in flight while this tile is computed */
        const int tb0 = __builtin_amdgcn_readfirstlane(tile * HAM_TT + wv * 64); /* first train index of this wave's 64 */
        const int cnt = min(64, jb.nt - tb0);             /* wave-uniform */
        const uint4* st = &s_t[b][(wv * 64) * 2];
        if (cnt == 64) {
#pragma unroll 8
            for (int j = 0; j < 64; j++) {
                const uint4 ta = st[2 * j], tb = st[2 * j + 1];
                const uint32_t key = top2_key(hamming256(qa, qb, ta, tb), (uint32_t)(tb0 + j));
                k2 = umed3(k1, k2, key);
                k1 = min(k1, key);
            }
        } else {
            for (int j = 0; j < cnt; j++) {
                const uint4 ta = st[2 * j], tb = st[2 * j + 1];
                const uint32_t key = top2_key(hamming256(qa, qb, ta, tb), (uint32_t)(tb0 + j));
                k2 = umed3(k1, k2, key);
                k1 = min(k1, key);
            }
        }
        if (tile + 1 < tile1) { /* the other buffer was last read one iteration ago, behind the barrier below */
            s_t[b ^ 1][tid] = r0;
            s_t[b ^ 1][tid + 256] = r1;
        }
        __syncthreads();
    }
    s_k[wv][lane][0] = k1;
    s_k[wv][lane][1] = k2;
    __syncthreads();
    if (wv == 0 && qi < jb.nq) {
        uint32_t b1 = 0xFFFFFFFFu, b2 = 0xFFFFFFFFu;
#pragma unroll
        for (int w2 = 0; w2 < 4; w2++)
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const uint32_t key = s_k[w2][lane][e];
                b2 = umed3(b1, b2, key);
                b1 = min(b1, key);
            }
        const size_t o = ((size_t)(jb.row0 + (uint32_t)qi) * nsplit + blockIdx.y) * 2;
        part[o] = b1;
        part[o + 1] = b2;
    }
}

__global__ void k_hamming_top2_merge_batch(const uint32_t* __restrict__ part, int nrows, int nsplit,
                                           int32_t* __restrict__ idx2, int32_t* __restrict__ dist2) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    uint32_t b1 = 0xFFFFFFFFu, b2 = 0xFFFFFFFFu;
    for (int i = 0; i < nsplit * 2; i++) {
        const uint32_t key = part[(size_t)r * nsplit * 2 + i];
        b2 = umed3(b1, b2, key);
        b1 = min(b1, key);
    }
    idx2[2 * r] = b1 == 0xFFFFFFFFu ? -1 : (int)(b1 & 0xFFFF);
    dist2[2 * r] = b1 == 0xFFFFFFFFu ? 0x7FFFFFFF : (int)(b1 >> 16);
    idx2[2 * r + 1] = b2 == 0xFFFFFFFFu ? -1 : (int)(b2 & 0xFFFF);
    dist2[2 * r + 1] = b2 == 0xFFFFFFFFu ? 0x7FFFFFFF : (int)(b2 >> 16);
}

/* how many workgroups share a problem's train set: as few as fill the GPU with eight waves per SIMD (8 workgroups per CU;
 * with 4 the 16 x 2000 x 2000 launch ran 42.7 us, 81 % of its issue time), never more than it has tiles */
int vk_hamming_top2_batch_split(int nprob, int max_nq, int max_nt) {
    const int qt = std::max(1, (max_nq + HAM_TQ - 1) / HAM_TQ), ntile = std::max(1, (max_nt + HAM_TT - 1) / HAM_TT);
    const int want = (2048 + nprob * qt - 1) / (nprob * qt);
    return std::max(1, std::min(want, ntile));
}

void vk_hamming_top2_batch(hipStream_t st, const Top2Jobs& jobs, int nprob, int max_nq, int nrows, int nsplit, uint32_t* part,
                           int32_t* idx2, int32_t* dist2) {
    if (nprob <= 0 || nrows <= 0) return;
    if (max_nq > 0)
        hipLaunchKernelGGL(k_hamming_top2_batch, dim3((max_nq + HAM_TQ - 1) / HAM_TQ, nsplit, nprob), dim3(256), 0, st, jobs, nsplit,
                           part);
    hipLaunchKernelGGL(k_hamming_top2_merge_batch, dim3((nrows + 255) / 256), dim3(256), 0, st, part, nrows, nsplit, idx2, dist2);
}

