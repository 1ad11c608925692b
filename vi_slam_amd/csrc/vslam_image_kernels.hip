/* vslam_image_kernels.hip -- the image-streaming kernels of the extractor: pyramid, FAST cells, 7x7 blur.
 *
 * k_resize_level_v2  one thread = four consecutive output pixels of a row (cv::resize INTER_LINEAR 8u fixed point).
 * k_fast_cells_v3    one workgroup per 30-px FAST cell: packed 4-pixel compass pre-test that writes its survivor lists
 *                    as it goes, one pass of packed two-pixel score networks on dense lanes, cell-local NMS,
 *                    raster-ordered compaction.
 * k_blur7_v2         marching-rows separable 7x7: one WAVE owns a 256-px-wide strip (64 lanes x 4 px, loaded as one
 *                    dword per lane = 256 B coalesced), gets its neighbours' dwords with two wave shuffles, does
 *                    the row pass with v_dot4_u32_u8 against packed tap constants (10 dot4 per 4 px, no byte
 *                    extraction) and keeps a 7-row ring of row-pass results in VGPRs for the column pass.  No LDS,
 *                    no barriers; every input byte is read once per strip (+6 halo rows per 32).
 * (The kernel names keep their generation suffix so that profiles of different rounds line up; the first
 * generations are in the git history only.)
 */
#include "vslam_kernels.h"
#include "vslam_wave.h"

__device__ __forceinline__ const uint8_t* level_base_v2(const uint8_t* pyr, size_t slot_stride,
                                                        const BatchSrc& src, const LevelGeom& lg, int level,
                                                        int slot, int* pitch) {
    if (level == 0) {
        *pitch = (int)src.pitch0[slot];
        return src.l0[slot];
    }
    *pitch = lg.pitch;
    return pyr + (size_t)slot * slot_stride + lg.off;
}

__device__ __forceinline__ int refl101_v2(int p, int len) {
    if (p < 0) p = -p;
    if (p >= len) p = 2 * (len - 1) - p;
    return min(max(p, 0), len - 1);
}

/* ------------------------------------------------------------------------------------------------
 * blur
 * ---------------------------------------------------------------------------------------------- */
#define BV_OUTW 248  /* output columns per strip: lanes 1..62 of 64, 4 px each */

__device__ __forceinline__ uint32_t udot4(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_udot4(a, b, c, false);
}

/* row pass of the 4 pixels of one lane: L, C, R = dwords of the left neighbour, this lane, right neighbour
 * (byte 0 = lowest x).  T* = taps packed per byte position; results are exact (<= 255*256). */
struct BlurTapsV2 {
    uint32_t l0, c0, l1, c1, r1, l2, c2, r2, c3, r3; /* packed byte weights */
    uint32_t k[7];
};

#ifndef BLUR_WPB
#define BLUR_WPB 4 /* wave tasks per workgroup of k_blur7_v2 (independent of each other; 1 and 2 measured) */
#endif
/* Column pass on PAIRS of consecutive rows (round 4): the row-pass results are at most 255 * 256 and fit 16 bits, so two
 * rows of one pixel share a register and v_dot2_u32_u16 applies two taps at once: a window of seven rows is three pairs
 * and the newest row -- 3 dot2 + 1 mad per pixel and one v_lshl_or to make the new pair, instead of 4 multiplies and 6 adds.
 * Every consecutive pair (r, r+1) is made once, when row r+1 arrives, and serves as the first, second and third pair of
 * three successive windows; six pairs are alive at a time, so the row loop is unrolled by six. */
typedef unsigned short ushort2b __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t udot2_u16(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2b, a), __builtin_bit_cast(ushort2b, b), c, false);
}
__global__ void __launch_bounds__(64 * BLUR_WPB)
k_blur7_v2(const uint8_t* __restrict__ pyr, size_t slot_stride, BatchSrc src, PyramidGeom g, uint8_t* blur,
           const uint32_t* __restrict__ tasks, int ntasks, int nslots, int rows_per_task, BlurTapsV2 T) {
    const int lane = threadIdx.x & 63;
    /* XCD-aware order (workgroups b, b+8 share an L2): the (slot, task) list is cut into 8 contiguous parts */
    const int tpb = (ntasks + BLUR_WPB - 1) / BLUR_WPB, nwork = tpb * nslots, per_xcd = (nwork + 7) >> 3;
    const int wk = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (wk >= nwork) return;
    const int slot = wk / tpb;
    const int task = (wk - slot * tpb) * BLUR_WPB + (threadIdx.x >> 6);
    if (task >= ntasks) return; /* wave-uniform */
    /* level << 24 | rowchunk << 12 | strip; wave-uniform by construction -- say so, or every loop bound and
     * branch below is treated as divergent */
    const uint32_t td = (uint32_t)__builtin_amdgcn_readfirstlane((int)tasks[task]);
    const int level = td >> 24, rc = (td >> 12) & 0xFFF, strip = td & 0xFFF;
    const LevelGeom lg = g.lv[level];
    int pitch;
    const uint8_t* img = level_base_v2(pyr, slot_stride, src, lg, level, slot, &pitch);
    uint8_t* out = blur + (size_t)slot * slot_stride + lg.off;
    const int w = lg.w, h = lg.h;
    /* the last strip is shifted left so that it ends at the image edge (overlap is rewritten identically) */
    const int ox0 = min(strip * BV_OUTW, max(w - BV_OUTW, 0)); /* first output column of the strip */
    const int x = ox0 - 4 + lane * 4;                          /* first of this lane's 4 columns */
    const int y0 = rc * rows_per_task;
    const int nrows = min(rows_per_task, h - y0);
    /* Branch-free row fetch: every lane reads the 8-byte window [bc, bc+8) of the row (inside the row for any
     * w >= 8) and picks its four columns -- reflected once at the image border -- with one v_perm_b32.  (A
     * divergent "interior dword / border bytes" choice made the compiler wait for each load where it was
     * issued, i.e. one full memory latency per row.) */
    int bx[4];
#pragma unroll
    for (int k = 0; k < 4; k++) bx[k] = refl101_v2(x + k, w);
    const int bc = min(min(min(bx[0], bx[1]), min(bx[2], bx[3])), w - 8);
    const uint32_t sel = (uint32_t)(bx[0] - bc) | ((uint32_t)(bx[1] - bc) << 8) | ((uint32_t)(bx[2] - bc) << 16) |
                         ((uint32_t)(bx[3] - bc) << 24);

    auto load_row = [&](int yy) -> uint2 { /* raw window; the v_perm happens where the row is consumed */
        /* 32-bit offsets: the row term is scalar arithmetic, the lane term one add, the load takes the scalar base */
        const uint32_t* row = (const uint32_t*)(img + ((uint32_t)refl101_v2(yy, h) * (uint32_t)pitch + (uint32_t)bc));
        return make_uint2(row[0], row[1]);
    };
    auto hpass = [&](uint2 raw, uint32_t* o) {
        const uint32_t Cw = __builtin_amdgcn_perm(raw.y, raw.x, sel);
        /* neighbours' dwords by DPP wave shifts (lanes 0 / 63 are halo lanes whose outputs are never stored) */
        const uint32_t Lw = (uint32_t)__builtin_amdgcn_update_dpp((int)Cw, (int)Cw, 0x138, 0xf, 0xf, false); /* wave_shr:1 */
        const uint32_t Rw = (uint32_t)__builtin_amdgcn_update_dpp((int)Cw, (int)Cw, 0x130, 0xf, 0xf, false); /* wave_shl:1 */
        o[0] = udot4(Lw, T.l0, udot4(Cw, T.c0, 0));
        o[1] = udot4(Lw, T.l1, udot4(Cw, T.c1, udot4(Rw, T.r1, 0)));
        o[2] = udot4(Lw, T.l2, udot4(Cw, T.c2, udot4(Rw, T.r2, 0)));
        o[3] = udot4(Cw, T.c3, udot4(Rw, T.r3, 0));
    };
    /* taps of the pairs: low half = the earlier row */
    const uint32_t W01 = T.k[0] | (T.k[1] << 16), W23 = T.k[2] | (T.k[3] << 16), W45 = T.k[4] | (T.k[5] << 16);
    uint32_t pr[6][4]; /* pair k % 6 = rows (k, k + 1), counted from y0 - 3 */
    uint32_t prev[4];  /* the newest row's results (the low half of the next pair) */
    /* prologue: rows y0-3 .. y0+2 = rows 0 .. 5, pairs 0 .. 4 */
    {
        uint32_t o[4];
        hpass(load_row(y0 - 3), prev);
#pragma unroll
        for (int r = 1; r < 6; r++) {
            hpass(load_row(y0 - 3 + r), o);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                pr[r - 1][c] = prev[c] | (o[c] << 16);
                prev[c] = o[c];
            }
        }
    }
    const bool writer = lane >= 1 && lane <= 62 && x < w; /* halo lanes and lanes past the image do not store */
    /* input rows are fetched six iterations ahead (a ring of 6 dwords pairs): with only a few waves per SIMD one
     * row of look-ahead left the wave waiting on HBM every iteration */
    uint2 nw[6];
#pragma unroll
    for (int u = 0; u < 6; u++) nw[u] = load_row(y0 + 3 + u);
    for (int rb = 0; rb < nrows; rb += 6) {
#pragma unroll
        for (int u = 0; u < 6; u++) {
            const int r = rb + u;
            if (r < nrows) { /* wave-uniform */
                uint32_t o[4];
                hpass(nw[u], o); /* row r + 6 of the count from y0 - 3: the newest row of output row r's window */
                if (r + 6 < nrows) nw[u] = load_row(y0 + r + 9);
                uint32_t px[4];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    /* window rows r .. r+6: pairs r, r+2, r+4 and the newest row */
                    uint32_t acc = __umul24(T.k[6], o[c]) + 32768u;
                    acc = udot2_u16(pr[(u + 4) % 6][c], W45, acc);
                    acc = udot2_u16(pr[(u + 2) % 6][c], W23, acc);
                    acc = udot2_u16(pr[u][c], W01, acc);
                    px[c] = min(acc >> 16, 255u);
                    pr[(u + 5) % 6][c] = prev[c] | (o[c] << 16); /* pair r+5 = rows (r+5, r+6): used from the next row on */
                    prev[c] = o[c];
                }
                if (writer) {
                    uint8_t* op = out + ((uint32_t)(y0 + r) * (uint32_t)lg.pitch + (uint32_t)x); /* writer lanes: x >= 0 */
                    if (x + 3 < w) *(uint32_t*)op = px[0] | (px[1] << 8) | (px[2] << 16) | (px[3] << 24);
                    else {
                        op[0] = (uint8_t)px[0];
                        if (x + 1 < w) op[1] = (uint8_t)px[1];
                        if (x + 2 < w) op[2] = (uint8_t)px[2];
                    }
                }
            }
        }
    }
}

void vk_blur7_v2(hipStream_t st, const uint8_t* pyr, size_t slot_stride, const BatchSrc& src, const PyramidGeom& g,
                 uint8_t* blur, const uint32_t* tasks, int ntasks, const int32_t taps[7], int rows_per_task,
                 int nslots) {
    BlurTapsV2 T;
    const uint32_t k0 = taps[0], k1 = taps[1], k2 = taps[2], k3 = taps[3], k4 = taps[4], k5 = taps[5], k6 = taps[6];
    auto pk = [](uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3) { return b0 | (b1 << 8) | (b2 << 16) | (b3 << 24); };
    /* pixel j of the lane sits at byte j of C; its window covers bytes j-3 .. j+3 of the L|C|R stream */
    T.l0 = pk(0, k0, k1, k2); T.c0 = pk(k3, k4, k5, k6);
    T.l1 = pk(0, 0, k0, k1);  T.c1 = pk(k2, k3, k4, k5); T.r1 = pk(k6, 0, 0, 0);
    T.l2 = pk(0, 0, 0, k0);   T.c2 = pk(k1, k2, k3, k4); T.r2 = pk(k5, k6, 0, 0);
    T.c3 = pk(k0, k1, k2, k3); T.r3 = pk(k4, k5, k6, 0);
    for (int i = 0; i < 7; i++) T.k[i] = taps[i];
    const int nwork = ((ntasks + BLUR_WPB - 1) / BLUR_WPB) * nslots;
    hipLaunchKernelGGL(k_blur7_v2, dim3(((nwork + 7) / 8) * 8), dim3(64 * BLUR_WPB), 0, st, pyr, slot_stride, src, g, blur, tasks, ntasks,
                       nslots, rows_per_task, T);
}

/* ------------------------------------------------------------------------------------------------
 * FAST cells
 * ---------------------------------------------------------------------------------------------- */

/* hipcc splits min(a,min(b,c)) into shared two-input mins; the three-input forms halve the network */
__device__ __forceinline__ int imin3v(int a, int b, int c) {
    int r;
    asm("v_min3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ int imax3v(int a, int b, int c) {
    int r;
    asm("v_max3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

/* One polarity of the FAST score: max over the 16 nine-pixel arcs of min_k d_k, with d_k = sign*(v - ring_k).
 * sign = +1 -> dark corners (cornerScore's a0), sign = -1 -> bright corners (its -b0).  The OpenCV score is
 * max(dark, bright) - 1. */
template <int SIGN, int P> /* P: LDS pitch of the window */
__device__ __forceinline__ int fast_half_score(const uint8_t* c) {
    const int v = c[0];
    int d[16];
#define RD(k, off) d[k] = SIGN > 0 ? v - (int)c[off] : (int)c[off] - v
    RD(0, 3 * P);   RD(1, 3 * P + 1);   RD(2, 2 * P + 2);    RD(3, P + 3);
    RD(4, 3);        RD(5, -P + 3);      RD(6, -2 * P + 2);   RD(7, -3 * P + 1);
    RD(8, -3 * P);  RD(9, -3 * P - 1);  RD(10, -2 * P - 2);  RD(11, -P - 3);
    RD(12, -3);      RD(13, P - 3);      RD(14, 2 * P - 2);   RD(15, 3 * P - 1);
#undef RD
    int lo3[16];
#pragma unroll
    for (int k = 0; k < 16; k++) lo3[k] = imin3v(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
    int A = -256;
#pragma unroll
    for (int k = 0; k < 16; k += 2)
        A = imax3v(A, imin3v(lo3[k], lo3[(k + 3) & 15], lo3[(k + 6) & 15]),
                   imin3v(lo3[k + 1], lo3[(k + 4) & 15], lo3[(k + 7) & 15]));
    return A;
}

/* The same for TWO pixels at once, in the halves of packed u16 registers (v_pk_sub_u16 clamp / v_pk_min_u16 /
 * v_pk_max_u16), returned as max(score half, 0) per half.  A nine-pixel arc that starts in one half of the ring ends
 * in the other one, so with suffix minima S and prefix minima Pm of the two ring halves
 *     arc starting at element i of a half = min(S_thishalf[i], Pm_otherhalf[i]):
 * 26 + 16 minima and 15 maxima for both pixels, against 2 x (32 three-input minima + 8 maxima) one by one.  Clamping
 * the differences at 0 does not change max(score - 1, 0).
 * (Unaligned LDS reads -- the ring rows as dwords / 8 bytes at x-1, x-2, x-3, 14 instead of 34 LDS instructions per pair --
 * work on gfx950 but are served lane by lane: the kernel took 283 instead of 115 us.  Byte reads it is.) */
__device__ __forceinline__ uint32_t pk_sub_sat16(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_min16(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_max16(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <int SIGN, int P>
__device__ __forceinline__ uint32_t fast_pair_score(const uint8_t* ca, const uint8_t* cb) {
#define RING(off) (SIGN > 0 ? pk_sub_sat16(v, (uint32_t)ca[off] | ((uint32_t)cb[off] << 16)) \
                            : pk_sub_sat16((uint32_t)ca[off] | ((uint32_t)cb[off] << 16), v))
    const uint32_t v = (uint32_t)ca[0] | ((uint32_t)cb[0] << 16);
    uint32_t S1[8], P1[8], S2[8], P2[8];
    /* ring 4..11: the centre row's right end, up the right side, the top, down the left side to row -1 */
    S1[0] = RING(3);           S1[1] = RING(-P + 3);      S1[2] = RING(-2 * P + 2);  S1[3] = RING(-3 * P + 1);
    S1[4] = RING(-3 * P);      S1[5] = RING(-3 * P - 1);  S1[6] = RING(-2 * P - 2);  S1[7] = RING(-P - 3);
    P1[0] = S1[0];
#pragma unroll
    for (int i = 1; i < 7; i++) P1[i] = pk_min16(P1[i - 1], S1[i]);
#pragma unroll
    for (int i = 6; i >= 0; i--) S1[i] = pk_min16(S1[i], S1[i + 1]);
    P1[7] = S1[0];
    __builtin_amdgcn_sched_barrier(0); /* keeps the second half's loads below the first half's minima: <= 64 VGPRs */
    /* ring 12..15, 0..3: down the left side from the centre row, the bottom, up the right side to row +1 */
    S2[0] = RING(-3);          S2[1] = RING(P - 3);       S2[2] = RING(2 * P - 2);   S2[3] = RING(3 * P - 1);
    S2[4] = RING(3 * P);       S2[5] = RING(3 * P + 1);   S2[6] = RING(2 * P + 2);   S2[7] = RING(P + 3);
#undef RING
    P2[0] = S2[0];
#pragma unroll
    for (int i = 1; i < 7; i++) P2[i] = pk_min16(P2[i - 1], S2[i]);
#pragma unroll
    for (int i = 6; i >= 0; i--) S2[i] = pk_min16(S2[i], S2[i + 1]);
    P2[7] = S2[0];
    uint32_t best = pk_min16(S1[0], P2[0]);
#pragma unroll
    for (int i = 1; i < 8; i++) best = pk_max16(best, pk_min16(S1[i], P2[i]));
#pragma unroll
    for (int i = 0; i < 8; i++) best = pk_max16(best, pk_min16(S2[i], P1[i]));
    return best;
}

/* ------------------------------------------------------------------------------------------------
 * pyramid level: one thread = four consecutive output pixels of a row (same arithmetic as the generic
 * one-pixel-per-thread k_resize_level in vslam_kernels.hip, cv::resize INTER_LINEAR 8u).  All eight taps of a row lie in one 8-byte window of the source
 * row (host table: window start per quad), so a thread issues two 8-byte loads instead of sixteen byte loads;
 * a v_perm_b32 per output picks its two taps as packed u16 and v_dot2_i32_i16 does the horizontal pass.
 * ---------------------------------------------------------------------------------------------- */
typedef short short2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int hdot2(uint32_t taps, uint32_t coef) {
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, taps), __builtin_bit_cast(short2v, coef), 0, false);
}
/* Four horizontal passes at once in the three-address form with the constant 0 as accumulator: from the builtin hipcc makes the
 * two-address v_dot2c_i32_i16 and a v_mov_b32 0 in front of every one of them.  A dot instruction's result may not be read by
 * another kind of VALU instruction for three wait states (nor overwritten by one for four), which the compiler only knows of
 * its own dots: the s_nop 3 closes the block (the compiler's own sequences carry such nops too). */
__device__ __forceinline__ void hdot2x4(uint32_t t0, uint32_t t1, uint32_t t2, uint32_t t3, uint4 cf, int* h0, int* h1, int* h2,
                                        int* h3) {
    int r0, r1, r2, r3;
    asm("v_dot2_i32_i16 %0, %4, %8, 0\n\t"
        "v_dot2_i32_i16 %1, %5, %9, 0\n\t"
        "v_dot2_i32_i16 %2, %6, %10, 0\n\t"
        "v_dot2_i32_i16 %3, %7, %11, 0\n\t"
        "s_nop 3"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
        : "v"(t0), "v"(t1), "v"(t2), "v"(t3), "v"(cf.x), "v"(cf.y), "v"(cf.z), "v"(cf.w));
    *h0 = r0; *h1 = r1; *h2 = r2; *h3 = r3;
}

__global__ void __launch_bounds__(256)
k_resize_level_v2(uint8_t* pyr, size_t slot_stride, BatchSrc src, LevelGeom sg, LevelGeom dg, int src_level,
                  const uint16_t* __restrict__ qbase, const ResizeQuad* __restrict__ quads,
                  const uint16_t* __restrict__ ytab, const int16_t* __restrict__ yb, int nq) {
    const int q = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int slot = blockIdx.z;
    if (q >= nq || dy >= dg.h) return;
    int spitch;
    const uint8_t* S = level_base_v2(pyr, slot_stride, src, sg, src_level, slot, &spitch);
    const int sy0 = ytab[2 * dy], sy1 = ytab[2 * dy + 1];
    const int b0 = yb[2 * dy], b1 = yb[2 * dy + 1];
    const int base = qbase[q];
    const uint4 sel = *(const uint4*)quads[q].sel;
    const uint4 cf = *(const uint4*)quads[q].coef;
    const uint32_t* r0 = (const uint32_t*)(S + (size_t)sy0 * spitch + base); /* unaligned dwords, in-row */
    const uint32_t* r1 = (const uint32_t*)(S + (size_t)sy1 * spitch + base);
    const uint32_t lo0 = r0[0], hi0 = r0[1], lo1 = r1[0], hi1 = r1[1];
    uint32_t out = 0;
#define RZ_ONE(j, SEL, CF)                                                                                 \
    {                                                                                                      \
        const int h0 = hdot2(__builtin_amdgcn_perm(hi0, lo0, SEL), CF);                                    \
        const int h1 = hdot2(__builtin_amdgcn_perm(hi1, lo1, SEL), CF);                                    \
        const int v = (((__mul24(b0, h0 >> 4)) >> 16) + ((__mul24(b1, h1 >> 4)) >> 16) + 2) >> 2;         \
        out |= (uint32_t)(v & 0xFF) << (8 * (j));                                                          \
    }
    RZ_ONE(0, sel.x, cf.x)
    RZ_ONE(1, sel.y, cf.y)
    RZ_ONE(2, sel.z, cf.z)
    RZ_ONE(3, sel.w, cf.w)
#undef RZ_ONE
    /* rows are padded to a multiple of 128 bytes, so the last quad may spill into the padding */
    *(uint32_t*)(pyr + (size_t)slot * slot_stride + dg.off + (size_t)dy * dg.pitch + 4 * q) = out;
}

void vk_resize_level_v2(hipStream_t st, uint8_t* pyr, size_t slot_stride, const BatchSrc& src, const LevelGeom& sg,
                        const LevelGeom& dg, int src_level, const uint16_t* qbase, const ResizeQuad* quads,
                        const uint16_t* ytab, const int16_t* yb, int nslots) {
    const int nq = (dg.w + 3) / 4;
    dim3 grid((nq + 63) / 64, (dg.h + 3) / 4, nslots);
    hipLaunchKernelGGL(k_resize_level_v2, grid, dim3(256), 0, st, pyr, slot_stride, src, sg, dg, src_level, qbase,
                       quads, ytab, yb, nq);
}

/* ------------------------------------------------------------------------------------------------
 * Fused pyramid: levels l0+1 .. l0+nl in ONE launch (FExtractor::ComputePyramid, fextractor.cpp:1135-1160).
 * Level l is resized from level l-1, so the seven per-level launches were a chain of dependent kernels whose small
 * levels cannot fill 256 CUs.  Here a workgroup owns a spatial tile: it stages the tile of the source level (plus the
 * halo the cascade needs) in LDS, computes its tile of level l0+1 into LDS, from that level l0+2, ... and stores the
 * part of every level it owns to HBM.  Neighbouring tiles RECOMPUTE each other's halos (6-11 % more pixels) instead
 * of exchanging them, so there is no inter-workgroup dependency.  The plan (compute / store range per tile and level,
 * LDS offsets) comes from vslam::build_pyramid_group and is validated on the CPU by emulate_pyramid_group.
 * Arithmetic = k_resize_level_v2: lane = quad of four output pixels, all eight taps of a source row inside one 8-byte
 * window starting at qbase[q]; in LDS the window is cut out of three aligned dwords with v_alignbyte_b32.
 * 8 levels = 2 launches (levels 1-3 from level 0, levels 4-7 from level 3).
 * ---------------------------------------------------------------------------------------------- */
/* 16 bytes of a source row starting at column `col`; only `nvalid` (< 16) of them may be read -- the last row of the
 * caller's level-0 image may end with the buffer.  No loop: whole dwords, then at most three single bytes, all
 * independent loads (one memory latency, not sixteen). */
__device__ __forceinline__ uint4 pyr_load16_tail(const uint8_t* p, int nvalid) {
    uint32_t t[4] = {0, 0, 0, 0};
    const int nd = nvalid >> 2, nbyte = nvalid & 3;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (i < nd) t[i] = *(const uint32_t*)(p + 4 * i);
        else if (i == nd) {
            if (nbyte > 0) t[i] |= (uint32_t)p[4 * i];
            if (nbyte > 1) t[i] |= (uint32_t)p[4 * i + 1] << 8;
            if (nbyte > 2) t[i] |= (uint32_t)p[4 * i + 2] << 16;
        }
    }
    return make_uint4(t[0], t[1], t[2], t[3]);
}

struct PyrTileU { /* PyrTileDev in scalar registers */
    int c0, nc, r0, nr, sq0, sq1, sr0, sr1;
    uint32_t lds_off, pitch, rt_off;
};
__device__ __forceinline__ PyrTileU pyr_tile_uniform(const PyrTileDev& t) {
    PyrTileU u;
#define PTU(f) u.f = __builtin_amdgcn_readfirstlane((int)t.f)
    PTU(c0); PTU(nc); PTU(r0); PTU(nr); PTU(sq0); PTU(sq1); PTU(sr0); PTU(sr1);
#undef PTU
    u.lds_off = (uint32_t)__builtin_amdgcn_readfirstlane((int)t.lds_off);
    u.pitch = (uint32_t)__builtin_amdgcn_readfirstlane((int)t.pitch);
    u.rt_off = (uint32_t)__builtin_amdgcn_readfirstlane((int)t.rt_off);
    return u;
}

template <int NT> /* threads per workgroup: NT / 64 waves share the rows of a tile */
__global__ void __launch_bounds__(NT)
k_pyramid_group(uint8_t* pyr, size_t slot_stride, BatchSrc src, PyrGroupDev G, int nslots, uint8_t* reset_cand,
                size_t cand_stride, int32_t* reset_err) {
    extern __shared__ __align__(16) uint8_t psm[];
    constexpr int NW = NT / 64;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); /* wave-uniform, and said so: the row loops go scalar */
    /* the pass's first launch also zeroes the (total, overflow) header of every slot's candidate buffer and the
     * quadtree's error word (FAST and the quadtree come later on the stream): no launch of its own for 33 stores */
    if (reset_cand && blockIdx.x == 0) {
        if ((int)threadIdx.x < nslots) *(uint2*)(reset_cand + (size_t)threadIdx.x * cand_stride) = make_uint2(0u, 0u);
        if (reset_err && threadIdx.x < 4) reset_err[threadIdx.x] = 0;
    }
    /* XCD-aware order: the (slot, tile) list is cut into 8 contiguous parts, one per XCD (workgroups b, b+8 share an L2),
     * so the tiles of one image -- whose source halos overlap -- are fetched through one L2 */
    const int nwork = G.ntiles * nslots, per_xcd = (nwork + 7) >> 3;
    const int wk = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (wk >= nwork) return;
    const int slot = wk / G.ntiles, tile = wk - slot * G.ntiles;
    const PyrTileDev* T = G.tiles + (size_t)tile * (G.nl + 1);
    const PyrTileDev S0 = T[0];
    if (S0.nr <= 0 || S0.nc <= 0) return; /* a tile that owns nothing (more tiles than quads on a tiny level) */

    /* Everything a level needs besides pixels is fetched NOW, so that no global-memory latency sits inside the per-level
     * loops: the lane's quad record of every level (registers) and the row tables of every level (LDS). */
    uint4 qsel[VSLAM_PYR_GROUP_LEVELS], qcf[VSLAM_PYR_GROUP_LEVELS];
    int qloc[VSLAM_PYR_GROUP_LEVELS];
#pragma unroll
    for (int j = 1; j <= VSLAM_PYR_GROUP_LEVELS; j++) {
        qsel[j - 1] = qcf[j - 1] = make_uint4(0, 0, 0, 0);
        qloc[j - 1] = 0;
        if (j <= G.nl) {
            const PyrTileDev D = T[j];
            if (lane < (D.nc >> 2)) {
                const int q = (D.c0 >> 2) + lane;
                qloc[j - 1] = (int)G.qbase[j - 1][q] - T[j - 1].c0;
                qsel[j - 1] = *(const uint4*)G.quads[j - 1][q].sel;
                qcf[j - 1] = *(const uint4*)G.quads[j - 1][q].coef;
            }
            uint2* rt = (uint2*)(psm + D.rt_off);
            for (int i = threadIdx.x; i < D.nr; i += NT) {
                const int r = D.r0 + i;
                rt[i] = make_uint2(*(const uint32_t*)(G.ytab[j - 1] + 2 * r), *(const uint32_t*)(G.yb[j - 1] + 2 * r));
            }
        }
    }
    {   /* stage the source tile: lane = one 16-byte chunk of a row, several rows per wave instruction; the loads of
         * four such instructions are issued before the first LDS store */
        int spitch;
        const uint8_t* img = level_base_v2(pyr, slot_stride, src, G.lg[0], G.l0, slot, &spitch);
        /* bytes of a row that may be read: internal levels are padded to their pitch; of the caller's level-0 image
         * only the LAST row ends with the buffer (reading a few bytes into the next row is harmless) */
        const int readable = G.l0 == 0 ? G.readable_w0 : spitch;
        const int last_row = G.lg[0].h - 1;
        const uint32_t s0pitch = (uint32_t)__builtin_amdgcn_readfirstlane((int)S0.pitch);
        const int nch = (int)s0pitch >> 4, rpi = 64 / nch;
        const int rsub = lane / nch, ch = lane - rsub * nch;
        const bool lact = rsub < rpi;
        const int col = S0.c0 + 16 * ch;
        for (int kb = 0; kb * NW * rpi < S0.nr; kb += 4) {
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int row = ((kb + u) * NW + wave) * rpi + rsub;
                v[u] = make_uint4(0, 0, 0, 0);
                if (lact && row < S0.nr) {
                    const uint8_t* gp = img + mad24u_s((uint32_t)(S0.r0 + row), (uint32_t)spitch, (uint32_t)col); /* 32-bit, full rate */
                    if (col + 16 <= readable || (G.l0 == 0 && S0.r0 + row < last_row)) v[u] = *(const uint4*)gp;
                    else if (col < readable) v[u] = pyr_load16_tail(gp, readable - col);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int row = ((kb + u) * NW + wave) * rpi + rsub;
                if (lact && row < S0.nr) *(uint4*)(psm + mad24u_s((uint32_t)row, s0pitch, S0.lds_off + 16u * (uint32_t)ch)) = v[u];
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 1; j <= VSLAM_PYR_GROUP_LEVELS; j++) {
        if (j > G.nl) break; /* uniform */
        /* the tile records are the same for every lane -- said so field by field (16-bit fields come through vector loads):
         * row addresses are then scalar multiplies instead of vector ones in the row loop */
        const PyrTileU S = pyr_tile_uniform(T[j - 1]), D = pyr_tile_uniform(T[j]);
        if (lane < (D.nc >> 2) && D.nr > 0) {
            const int q = (D.c0 >> 2) + lane;
            const int loc = qloc[j - 1];
            const uint32_t sh = (uint32_t)(loc & 3);
            const uint4 sel = qsel[j - 1], cf = qcf[j - 1];
            const int sp = (int)S.pitch, dp = (int)D.pitch;
            const int sbase = (int)S.lds_off + (loc & ~3) - __mul24((int)S.r0, sp); /* + source row * sp */
            const int dbase = (int)D.lds_off + 4 * lane - __mul24((int)D.r0, dp);   /* + row * dp */
            const bool qstore = q >= D.sq0 && q < D.sq1;
            uint8_t* gout = pyr + (size_t)slot * slot_stride + G.lg[j].off + 4 * (size_t)q;
            const int gpitch = G.lg[j].pitch;
            const uint2* rt = (const uint2*)(psm + D.rt_off) - D.r0; /* indexed by the absolute row */
            /* A wave owns a contiguous band of output rows and walks it top to bottom.  Consecutive output rows share a
             * source row (sy1 of row r is sy0 of row r + 1 for 5 rows out of 6 at scale 1.2), so the horizontal pass
             * of a source row -- (h >> 4) of the lane's four pixels -- is kept in registers and reused: 1.2 instead of
             * 2 horizontal passes per output row. */
            const int band = (D.nr + NW - 1) / NW;
            const int rb = D.r0 + wave * band, re = min(rb + band, D.r0 + D.nr);
            int prev_sy = -1;
            int hp0 = 0, hp1 = 0, hp2 = 0, hp3 = 0;
#define PYR_HPASS(SY, H0, H1, H2, H3)                                                                        \
    {                                                                                                         \
        const uint32_t* pr_ = (const uint32_t*)(psm + sbase + __mul24((int)(SY), sp));                        \
        const uint32_t x0_ = pr_[0], x1_ = pr_[1], x2_ = pr_[2];                                             \
        const uint32_t lo_ = __builtin_amdgcn_alignbyte(x1_, x0_, sh), hi_ = __builtin_amdgcn_alignbyte(x2_, x1_, sh); \
        hdot2x4(__builtin_amdgcn_perm(hi_, lo_, sel.x), __builtin_amdgcn_perm(hi_, lo_, sel.y),                \
                __builtin_amdgcn_perm(hi_, lo_, sel.z), __builtin_amdgcn_perm(hi_, lo_, sel.w), cf, &H0, &H1, &H2, &H3); \
        H0 >>= 4; H1 >>= 4; H2 >>= 4; H3 >>= 4;                                                               \
    }
            for (int r = rb; r < re; r++) {
                const uint2 e = rt[r];
                /* wave-uniform by construction (LDS broadcast of the row table): scalar branches */
                const int sy0 = __builtin_amdgcn_readfirstlane((int)(e.x & 0xFFFF));
                const int sy1 = __builtin_amdgcn_readfirstlane((int)(e.x >> 16));
                const int b0 = (int)(int16_t)(e.y & 0xFFFF), b1 = (int)e.y >> 16; /* 0 .. 2048 */
                int g0, g1, g2, g3;
                if (sy0 == prev_sy) {
                    g0 = hp0; g1 = hp1; g2 = hp2; g3 = hp3;
                } else
                    PYR_HPASS(sy0, g0, g1, g2, g3)
                if (sy1 != sy0) PYR_HPASS(sy1, hp0, hp1, hp2, hp3)
                else {
                    hp0 = g0; hp1 = g1; hp2 = g2; hp3 = g3;
                }
                prev_sy = sy1;
                /* vertical pass: both factors fit 24 bits (b <= 2048, h >> 4 <= 32640): full-rate v_mul_i32_i24 */
                const uint32_t v0 = (uint32_t)(((__mul24(b0, g0) >> 16) + (__mul24(b1, hp0) >> 16) + 2) >> 2);
                const uint32_t v1 = (uint32_t)(((__mul24(b0, g1) >> 16) + (__mul24(b1, hp1) >> 16) + 2) >> 2);
                const uint32_t v2 = (uint32_t)(((__mul24(b0, g2) >> 16) + (__mul24(b1, hp2) >> 16) + 2) >> 2);
                const uint32_t v3 = (uint32_t)(((__mul24(b0, g3) >> 16) + (__mul24(b1, hp3) >> 16) + 2) >> 2);
                const uint32_t out = (v0 & 0xFF) | ((v1 & 0xFF) << 8) | ((v2 & 0xFF) << 16) | (v3 << 24);
                *(uint32_t*)(psm + dbase + __mul24(r, dp)) = out;
                if (qstore && r >= D.sr0 && r < D.sr1) *(uint32_t*)(gout + (size_t)r * gpitch) = out;
            }
#undef PYR_HPASS
        }
        __syncthreads();
    }
}

void vk_pyramid_group(hipStream_t st, uint8_t* pyr, size_t slot_stride, const BatchSrc& src, const PyrGroupDev& G,
                      size_t lds_bytes, int nslots, const vslam_tuning& T, uint8_t* reset_cand, size_t cand_stride,
                      int32_t* reset_err) {
    const int nwork = G.ntiles * nslots;
    const int nt = T.pyr_threads == 512 ? 512 : 256; /* waves per tile (A/B runs) */
    if (nt == 512)
        hipLaunchKernelGGL(k_pyramid_group<512>, dim3(((nwork + 7) / 8) * 8), dim3(512), lds_bytes, st, pyr, slot_stride, src, G, nslots,
                           reset_cand, cand_stride, reset_err);
    else
        hipLaunchKernelGGL(k_pyramid_group<256>, dim3(((nwork + 7) / 8) * 8), dim3(256), lds_bytes, st, pyr, slot_stride, src, G, nslots,
                           reset_cand, cand_stride, reset_err);
}

/* ------------------------------------------------------------------------------------------------
 * FAST cells.  What keeps the instruction count of the pre-test and the survivor bookkeeping down:
 *   - the window is staged one column to the left (LDS column = window column + 1), so the four interior
 *     pixels x = 4q..4q+3 of a row and their up/down compass pixels are aligned dwords and the left/right ones
 *     come out of two v_alignbyte_b32;
 *   - the compass pre-test runs on FOUR pixels per thread in packed u16 arithmetic: "(down|up) & (right|left) darker
 *     than v - T" is max(min(d,u), min(r,l)) < v - T, the bright case min(max(d,u), max(r,l)) > v + T: six
 *     v_pk_min/max_u16 and two v_pk_sub_u16 clamp per polarity pair and pixel pair;
 *   - survivors are written to the lists inside the sweep: per pixel column one compare per polarity straight into an
 *     SGPR lane mask, ranks from v_mbcnt, totals from s_bcnt1, ONE LDS atomic per wave and sweep, one store under the
 *     lane mask for the dark and the bright-only lanes together (they are disjoint);
 *   - the lists are disjoint (dark incl. the flagged "both polarities possible", bright-only), a thread takes TWO
 *     entries in the halves of packed registers, and with the usual list lengths each wave runs one network once.
 * ---------------------------------------------------------------------------------------------- */
typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_sub_sat(uint32_t a, uint32_t b) { /* per u16 half: max(a - b, 0) */
    uint32_t r;
    asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_add_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_or0(uint32_t v) { /* lanes without a source read 0 */
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    v += dpp_or0<0x111, 0xf>(v); /* row_shr:1 */
    v += dpp_or0<0x112, 0xf>(v); /* row_shr:2 */
    v += dpp_or0<0x114, 0xf>(v); /* row_shr:4 */
    v += dpp_or0<0x118, 0xf>(v); /* row_shr:8 */
    v += dpp_or0<0x142, 0xa>(v); /* row_bcast:15 -> rows 1,3 */
    v += dpp_or0<0x143, 0xc>(v); /* row_bcast:31 -> rows 2,3 */
    return v;
}

/* LDS byte address of a __shared__ object (the low half of its flat address) */
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)p; }
/* lane masks of "low / high u16 half != 0" straight into an SGPR pair (inactive lanes read 0) */
__device__ __forceinline__ uint64_t half_lo_nonzero(uint32_t x) {
    uint64_t m;
    asm("v_cmp_ne_u32_sdwa %0, %1, %2 src0_sel:WORD_0 src1_sel:DWORD" : "=s"(m) : "v"(x), "v"(0u));
    return m;
}
__device__ __forceinline__ uint64_t half_hi_nonzero(uint32_t x) {
    uint64_t m;
    asm("v_cmp_lt_u32_e64 %0, %1, %2" : "=s"(m) : "s"(0xFFFFu), "v"(x));
    return m;
}
__device__ __forceinline__ uint32_t lane_select(uint64_t m, uint32_t a, uint32_t b) { /* a for the lanes of m, else b */
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
    return r;
}
__device__ __forceinline__ uint32_t lane_rank(uint64_t m) { /* set bits of m below this lane */
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
/* ds_write_b16 of `code` by the lanes of `m` (a subset of exec) to per-lane addresses, then -- only if mX is not
 * empty -- of codeX by the lanes of mX (a subset of m: same address, overwrites; LDS executes a wave's stores in order) */
__device__ __forceinline__ void lds_store_b16_masked2(uint64_t m, uint64_t mX, uint32_t at, uint32_t code, uint32_t codeX) {
    uint64_t save;
    asm volatile("s_mov_b64 %0, exec\n\t"
                 "s_mov_b64 exec, %1\n\tds_write_b16 %3, %4\n\t"
                 "s_cmp_eq_u64 %2, 0\n\ts_cbranch_scc1 1f\n\t"
                 "s_mov_b64 exec, %2\n\tds_write_b16 %3, %5\n"
                 "1:\n\ts_mov_b64 exec, %0"
                 : "=&s"(save) : "s"(m), "s"(mX), "v"(at), "v"(code), "v"(codeX) : "memory", "scc");
}
/* LDS fetch-and-add without the compiler's wave-aggregation wrapper (the caller is a single lane already) */
__device__ __forceinline__ uint32_t lds_add_rtn(uint32_t addr, uint32_t v) {
    uint32_t r;
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr), "v"(v) : "memory");
    return r;
}

#ifdef VSLAM_FAST_WGREC /* diagnostic build: one record per workgroup of the LAST launch: s_memtime at entry and exit,
 * s_memrealtime (100 MHz) at entry, HW_ID (16 bits) | XCC_ID << 16 | (s_memrealtime at exit - entry) << 32 --
 * tools/fast_occupancy.py turns them into workgroups in flight per CU over time */
#define FAST_WG_RECORDS 65536
__device__ unsigned long long g_fast_wg[FAST_WG_RECORDS * 4];
extern "C" int vslam_dbg_fast_wg_records(unsigned long long* out, int n) {
    if (n > FAST_WG_RECORDS) n = FAST_WG_RECORDS;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fast_wg), sizeof(unsigned long long) * 4 * (size_t)n) != hipSuccess) return -3;
    return n;
}
#endif
#ifdef VSLAM_FAST_STAMPS /* diagnostic build: cycles per phase of wave 0 of every workgroup, summed (tools/fast_stamps.py) */
__device__ unsigned long long g_fast_stamps[16];
#define FSTAMP(k) do { if (tid == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); atomicAdd(&g_fast_stamps[k], t_ - fst_); fst_ = t_; } } while (0)
extern "C" int vslam_dbg_fast_stamps(unsigned long long* out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_fast_stamps), sizeof(g_fast_stamps)) != hipSuccess) return -3;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_fast_stamps), z, sizeof(z)) != hipSuccess) return -3;
    }
    return 0;
}
#else
#define FSTAMP(k) do { } while (0)
#endif
#ifdef VSLAM_FAST_WGREC
#define FAST_WGREC_END()                                                                                              \
    do {                                                                                                              \
        if (tid == 0) {                                                                                               \
            const unsigned rec = blockIdx.y * gridDim.x + blockIdx.x;                                                 \
            if (rec < FAST_WG_RECORDS) {                                                                              \
                unsigned hw, xcc;                                                                                     \
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                                      \
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                                    \
                g_fast_wg[rec * 4 + 0] = fst_entry_;                                                                  \
                g_fast_wg[rec * 4 + 1] = __builtin_amdgcn_s_memtime();                                                \
                g_fast_wg[rec * 4 + 2] = fst_real_;                                                                   \
                const unsigned long long dreal = __builtin_amdgcn_s_memrealtime() - fst_real_;                        \
                g_fast_wg[rec * 4 + 3] = (unsigned long long)(hw & 0xFFFFu) | ((unsigned long long)(xcc & 0xFu) << 16) | (dreal << 32); \
            }                                                                                                         \
        }                                                                                                             \
    } while (0)
#else
#define FAST_WGREC_END() do { } while (0)
#endif
#ifndef FAST_XCD_CHUNK
#define FAST_XCD_CHUNK 16
#endif
template <int NT, int P> /* P: LDS pitch, 48 for windows up to 42 px, else 72 */
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_num_sgpr(80)))
k_fast_cells_v3(const uint8_t* __restrict__ pyr, size_t slot_stride, BatchSrc src, PyramidGeom g,
                const CellDesc* __restrict__ cells, uint8_t* cand_region, size_t cand_stride, int ncells,
                int iniTh, int minTh, int tile_rows, int lcap) {
    extern __shared__ __align__(16) uint8_t smem3[];
    uint8_t* win = smem3;                        /* tile_rows x P; window column c at LDS column c + 1 */
    uint8_t* sc = win + tile_rows * P;          /* (tile_rows-4) x P, interior at (1..ih, 1..iw) */
    uint32_t* keep = (uint32_t*)(sc + (tile_rows - 4) * P); /* 2 words per interior row */
    /* survivor lists (codes ly << 6 | x): dark pixels, with bit 15 set where the bright polarity is possible too, and
     * bright-only pixels; disjoint sets, each at most the pixels of the largest cell interior (lcap entries) */
    uint16_t* listD = (uint16_t*)(keep + (tile_rows - 6) * 2);
    uint16_t* listB = listD + lcap;
    __shared__ uint32_t s_cnt; /* nD+nX | nB << 16 */
    __shared__ uint32_t s_wave_tot[NT / 64];
    __shared__ int s_any;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
#ifdef VSLAM_FAST_STAMPS
    unsigned long long fst_ = __builtin_amdgcn_s_memtime();
#endif
#ifdef VSLAM_FAST_WGREC
    const unsigned long long fst_entry_ = __builtin_amdgcn_s_memtime(), fst_real_ = __builtin_amdgcn_s_memrealtime();
#endif
    const int slot = blockIdx.y;
    /* XCD-aware cell order: workgroups b and b+8 share an XCD (and its L2).  XCD k takes every eighth CHUNK of
     * FAST_XCD_CHUNK consecutive cells -- neighbouring cells, whose windows overlap by 6 px and share image rows, hit
     * the same L2 -- and the chunks are dealt round-robin, because the cost of a cell depends on its pyramid level and
     * texture: with one contiguous eighth of the cell list per XCD the slowest XCD ran 122 us and the fastest 87
     * (per-workgroup records, tools/fast_occupancy.py). */
    const int xcd_i = (int)(blockIdx.x >> 3);
    /* ... rotated by the image slot, so that the partial last round of chunks lands on a different XCD per image */
    const int cell = ((xcd_i / FAST_XCD_CHUNK) * 8 + (int)((blockIdx.x + blockIdx.y) & 7)) * FAST_XCD_CHUNK + xcd_i % FAST_XCD_CHUNK;
    if (cell >= ncells) return;
    /* the cell record through the SCALAR cache: as four dwords (a 16-bit field alone becomes a vector load, i.e. one
     * more full memory round trip in front of the window fetch -- the prologue's latency is half of a wave's life) */
    CellDesc cd;
    {
        const uint4 raw = ((const uint4*)cells)[cell];
        cd.level = (uint16_t)(raw.x & 0xFFFF);
        cd.x0 = (uint16_t)(raw.x >> 16);
        cd.y0 = (uint16_t)(raw.y & 0xFFFF);
        cd.x1 = (uint16_t)(raw.y >> 16);
        cd.y1 = (uint16_t)(raw.z & 0xFFFF);
        cd.pad = 0;
        cd.base = raw.w;
    }
    const int level = cd.level;
    const LevelGeom lg = g.lv[level];
    int pitch;
    const uint8_t* img = level_base_v2(pyr, slot_stride, src, lg, level, slot, &pitch);
    const int ww = cd.x1 - cd.x0, wh = cd.y1 - cd.y0;
    const int iw = ww - 6, ih = wh - 6;
    const int nwords = ih * 2;

    /* stage columns x0-1 .. (x0 >= 16; windows end >= 13 px before the row end): LPR lanes x 8 bytes per row
     * (= the LDS pitch), RPS rows per sweep.  ALL loads of the window are issued before anything waits for one, and the
     * tiles are zeroed while they are in flight.  In-kernel stamps (tools/fast_stamps.py, -DVSLAM_FAST_STAMPS) put 36 %
     * of wave 0's life into this prologue, 27 % into the pre-test sweeps and 20 % into the six barriers -- but the
     * waiting is hidden by the other seven waves of the SIMD: a workgroup that walks several cells and prefetches the
     * next window while it works was built twice -- register staging (+20 VGPRs: 6 waves per SIMD) and LDS-DMA
     * (global_load_lds_dwordx4 into a second tile, 7 waves per SIMD) -- and both were SLOWER (117-160 us against 111):
     * what this kernel is short of is issue slots per instruction, not latency cover. */
    constexpr int LPR = P / 8, RPS = NT / LPR;        /* 8-byte lanes per row, rows per sweep */
    /* sweeps kept in registers at once: two cover the 42 rows of the narrow-pitch geometries (KITTI, 1080p: windows of
     * 36-38 rows); taller windows take the row loop below */
    constexpr int NSW = P == 48 ? (42 + RPS - 1) / RPS : ((134 + RPS - 1) / RPS > 4 ? 4 : (134 + RPS - 1) / RPS);
    const int srow = tid / LPR, scol = (tid - srow * LPR) * 8;
    const bool stager = tid < RPS * LPR && scol < ww + 1;
    const uint8_t* gsrc = img + (size_t)cd.y0 * pitch + cd.x0 - 1;
    uint2 wreg[NSW];
#pragma unroll
    for (int u = 0; u < NSW; u++) {
        const int y = srow + u * RPS;
        wreg[u] = make_uint2(0u, 0u);
        /* uniform base + 32-bit lane offset: one v_mad_u32_u24 per load instead of a 64-bit multiply-add chain */
        if (stager && y < wh) wreg[u] = *(const uint2*)(gsrc + (uint32_t)(__umul24((uint32_t)y, (uint32_t)pitch) + (uint32_t)scol));
    }
    /* score tile and keep words are contiguous: zeroed in 16-byte stores (the tail may run into the lists, which
     * are filled later) */
    for (int i = tid; i < ((tile_rows - 4) * P + nwords * 4 + 15) / 16; i += NT) ((uint4*)sc)[i] = make_uint4(0u, 0u, 0u, 0u);
    if (tid == 0) {
        s_any = 0;
        s_cnt = 0u;
    }
#pragma unroll
    for (int u = 0; u < NSW; u++) {
        const int y = srow + u * RPS;
        if (stager && y < wh) *(uint2*)(win + y * P + scol) = wreg[u];
    }
    if (stager) /* taller windows than NSW sweeps cover (cells of unusual geometries): the rest row by row */
        for (int y = srow + NSW * RPS; y < wh; y += RPS)
            *(uint2*)(win + y * P + scol) = *(const uint2*)(gsrc + (uint32_t)(__umul24((uint32_t)y, (uint32_t)pitch) + (uint32_t)scol));
    FSTAMP(0);
    __syncthreads();
    FSTAMP(1);

    const int QW = (iw + 3) >> 2;               /* quads per interior row */
    const int qsh = QW > 8 ? 4 : 3;             /* 16 or 8 quad columns per sweep */
    const int qx = tid & ((1 << qsh) - 1), qly = tid >> qsh;
    const int rem = iw - 4 * qx;                /* pixels of this lane's quad inside the interior (<= 0: none) */
    const uint32_t* W32 = (const uint32_t*)win;
    int T = iniTh;
    for (int stage = 0; stage < 2; stage++) {
        /* pixels outside the interior (the last quad's tail, rows past ih in the last sweep) are tested against a
         * threshold nothing passes (v - 0x7FFF saturates to 0, v + 0x7FFF is above every pixel): no validity masks,
         * no compares for them inside the sweep.  Halves of TTv[par]: pixels par and par + 2 of the quad */
        uint32_t TTv[2];
#pragma unroll
        for (int par = 0; par < 2; par++)
            TTv[par] = (par < rem ? (uint32_t)T : 0x7FFFu) | ((par + 2 < rem ? (uint32_t)T : 0x7FFFu) << 16);
        for (int ly0 = 0; ly0 < ih; ly0 += (NT >> qsh)) { /* block-uniform trip count */
            const int ly = ly0 + qly;
            const bool rowok = ly < ih;
            /* every lane computes (rows past the interior and quads past QW read other parts of the LDS allocation;
             * their results are dropped by rowok / rem below): no branch inside the sweep */
            const uint32_t* rowc = W32 + (ly + 3) * (P / 4) + qx;
            const uint32_t A0 = rowc[0], C = rowc[1], E = rowc[2];
            const uint32_t U = W32[ly * (P / 4) + qx + 1], Dn = W32[(ly + 6) * (P / 4) + qx + 1];
            const uint32_t Lf = __builtin_amdgcn_alignbyte(C, A0, 1); /* columns x-3 */
            const uint32_t Rt = __builtin_amdgcn_alignbyte(E, C, 3);  /* columns x+3 */
#define EVN(x) __builtin_amdgcn_perm(0u, (x), 0x0c020c00u) /* pixels 0,2 as u16 halves */
#define ODD(x) __builtin_amdgcn_perm(0u, (x), 0x0c030c01u) /* pixels 1,3 */
            uint32_t passD[2], passB[2];
#pragma unroll
            for (int par = 0; par < 2; par++) {
                const uint32_t v = par ? ODD(C) : EVN(C);
                const uint32_t TT = rowok ? TTv[par] : 0x7FFF7FFFu;
                const uint32_t vm = pk_sub_sat(v, TT), vp = pk_add(v, TT);
                const uint32_t u = par ? ODD(U) : EVN(U), d = par ? ODD(Dn) : EVN(Dn);
                const uint32_t l = par ? ODD(Lf) : EVN(Lf), r = par ? ODD(Rt) : EVN(Rt);
                /* dark: (down|up) & (right|left) darker than v - T  <=>  max(min(d,u), min(r,l)) < v - T;
                 * bright: min(max(d,u), max(r,l)) > v + T -- six min/max + two saturating subtractions per polarity
                 * pair instead of eight subtractions, four ORs and two mins */
                passD[par] = pk_sub_sat(vm, pk_max(pk_min(d, u), pk_min(r, l)));
                passB[par] = pk_sub_sat(pk_min(pk_max(d, u), pk_max(r, l)), vp);
            }
#undef EVN
#undef ODD
            /* survivors go straight to the lists: per pixel column j of the quad one compare per polarity whose
             * lane mask lands in an SGPR pair, the lane's rank from v_mbcnt, the wave's totals from s_bcnt1 on the
             * scalar side, ONE LDS atomic per wave and sweep for the list space, stores under the lane mask.  (A
             * per-stage compaction pass over a byte-per-quad mask array -- wave scan, per-lane bit loops -- was ~150
             * instructions per wave on top of the bit extraction here; this is ~45 per sweep, and a barrier less.) */
            uint64_t mD[4], mB[4], mX[4];
            uint32_t totD = 0, totB = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint64_t dk = (j >> 1) ? half_hi_nonzero(passD[j & 1]) : half_lo_nonzero(passD[j & 1]);
                const uint64_t br = (j >> 1) ? half_hi_nonzero(passB[j & 1]) : half_lo_nonzero(passB[j & 1]);
                mD[j] = dk;       /* dark, or both polarities possible: the latter are listed once, here, flagged */
                mX[j] = dk & br;
                mB[j] = br & ~dk;
                totD += (uint32_t)__popcll(mD[j]);
                totB += (uint32_t)__popcll(mB[j]);
            }
            if (totD | totB) { /* wave-uniform */
                uint32_t base = 0;
                if (lane == 0) base = lds_add_rtn(lds_addr(&s_cnt), totD | (totB << 16));
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                const uint32_t aD = lds_addr(listD) + 2u * (base & 0xFFFFu), aB = lds_addr(listB) + 2u * (base >> 16);
                const uint32_t code0 = (uint32_t)((ly << 6) + 4 * qx);
                uint32_t pD = 0, pB = 0; /* entries of the pixel columns before j */
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t code = code0 + j;
                    /* dark and bright-only lanes are disjoint: ONE store with a per-lane address */
                    const uint32_t atD = aD + 2u * pD + 2u * lane_rank(mD[j]), atB = aB + 2u * pB + 2u * lane_rank(mB[j]);
                    lds_store_b16_masked2(mD[j] | mB[j], mX[j], lane_select(mD[j], atD, atB), code, code | 0x8000u);
                    pD += (uint32_t)__popcll(mD[j]);
                    pB += (uint32_t)__popcll(mB[j]);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* the masked stores above are not tracked by the compiler */
        FSTAMP(2);
        FSTAMP(3);
        FSTAMP(4);
        __syncthreads();
        FSTAMP(5);
        int nD, nB;
        {
            const uint32_t tot = s_cnt;
            nD = (int)(tot & 0xFFFFu);
            nB = (int)(tot >> 16);
        }
        const int ntot = nD + nB;
        /* one pass of the networks over the disjoint lists (no two threads touch the same score byte), TWO list
         * entries per thread in the halves of packed u16 registers.  Dark pairs are handed out from thread 0 upwards,
         * bright ones from thread NT-1 downwards, so with the usual ~110 + ~110 entries each wave runs one network
         * once; a flagged dark entry (both polarities possible, rare) makes its thread run both */
        const int sc_off = tile_rows * P - 2 * P - 3; /* score byte of a pixel relative to its window byte */
        for (int base = 0; base < max(nD, nB); base += 2 * NT) {
            const int iD = base + 2 * tid, iB = base + 2 * (NT - 1 - tid);
            if (iD < nD) {
                const uint32_t e2 = *(const uint32_t*)(listD + iD);
                const uint32_t eA = e2 & 0xFFFFu, eB = iD + 1 < nD ? e2 >> 16 : eA;
                const uint8_t* cA = win + (((eA >> 6) & 0x1FFu) + 3) * P + (eA & 63u) + 4;
                const uint8_t* cB = win + (((eB >> 6) & 0x1FFu) + 3) * P + (eB & 63u) + 4;
                uint32_t a = fast_pair_score<1, P>(cA, cB);
                if ((eA | eB) & 0x8000u) {
                    const uint32_t b = fast_pair_score<-1, P>(cA, cB);
                    const uint32_t m = ((eA & 0x8000u) ? 0xFFFFu : 0u) | ((eB & 0x8000u) ? 0xFFFF0000u : 0u);
                    a = pk_max(a, b & m);
                }
                a = pk_sub_sat(a, 0x00010001u); /* OpenCV's score: max(dark, bright) - 1, not below 0 */
                ((uint8_t*)cA)[sc_off] = (uint8_t)a;
                ((uint8_t*)cB)[sc_off] = (uint8_t)(a >> 16);
            }
            if (iB < nB) {
                const uint32_t e2 = *(const uint32_t*)(listB + iB);
                const uint32_t eA = e2 & 0xFFFFu, eB = iB + 1 < nB ? e2 >> 16 : eA;
                const uint8_t* cA = win + ((eA >> 6) + 3) * P + (eA & 63u) + 4;
                const uint8_t* cB = win + ((eB >> 6) + 3) * P + (eB & 63u) + 4;
                const uint32_t a = pk_sub_sat(fast_pair_score<-1, P>(cA, cB), 0x00010001u);
                ((uint8_t*)cA)[sc_off] = (uint8_t)a;
                ((uint8_t*)cB)[sc_off] = (uint8_t)(a >> 16);
            }
        }
        FSTAMP(6);
        __syncthreads();
        FSTAMP(7);
        /* NMS at T only where a score exists (listed pixels with a score below T cannot suppress anything) */
        int any = 0;
        for (int i = tid; i < ntot; i += NT) {
            const int code = (i < nD ? listD[i] : listB[i - nD]) & 0x7FFF;
            const int ly = code >> 6, x = code & 63;
            const uint8_t* q = sc + (ly + 1) * P + x + 1;
            const int s = q[0];
            if (s >= T) {
                const int mx = max(max(max((int)q[-P - 1], (int)q[-P]), max((int)q[-P + 1], (int)q[-1])),
                                   max(max((int)q[1], (int)q[P - 1]), max((int)q[P], (int)q[P + 1])));
                if (s > mx) {
                    atomicOr(&keep[ly * 2 + (x >> 5)], 1u << (x & 31));
                    any = 1;
                }
            }
        }
        if (any) s_any = 1; /* benign race, all writers store 1 */
        FSTAMP(8);
        __syncthreads();
        FSTAMP(9);
        if (s_any || stage == 1 || minTh == iniTh) break; /* block-uniform */
        /* empty at iniThFAST: the whole cell again at minThFAST (fextractor.cpp:800-807).  Scores already in
         * the tile belong to pixels that are listed again (the pre-test is monotone in T) and get rewritten. */
        T = minTh;
        if (tid == 0) s_cnt = 0u;
        __syncthreads();
    }

    /* ordered compaction: thread t owns the keep words [t*WPT, (t+1)*WPT) -- word w = (row w>>1, columns (w&1)*32 ..) --
     * so raster order is kept across threads */
    uint32_t* hdr = (uint32_t*)(cand_region + (size_t)slot * cand_stride);
    CellOut* cout = (CellOut*)(hdr + 2);
    uint32_t* cand = (uint32_t*)(cout + ncells);
    if (nwords <= 64) {
        /* the usual cell (interiors of up to 32 rows): wave 0 owns every keep word, one per lane -- the other waves are
         * done, no barrier, no cross-wave offsets */
        if (wv != 0) return;
        uint32_t bits = lane < nwords ? keep[lane] : 0u;
        const uint32_t c1 = (uint32_t)__popc(bits);
        const uint32_t in1 = wave_incl_scan(c1);
        const uint32_t total1 = (uint32_t)__builtin_amdgcn_readlane((int)in1, 63);
        if (lane == 0) {
            cout[cell].base = cd.base;
            cout[cell].count = total1;
        }
        if (bits) {
            uint32_t o = cd.base + in1 - c1;
            const int kly = lane >> 1, kxb = (lane & 1) * 32;
            const int ox = cd.x0 + 3 - VSLAM_BORDER + kxb, oy = cd.y0 + 3 - VSLAM_BORDER + kly;
            const uint8_t* srow_ = sc + (kly + 1) * P + kxb + 1;
            while (bits) {
                const int k = __ffs(bits) - 1;
                bits &= bits - 1;
                cand[o++] = ((uint32_t)srow_[k] << 24) | ((uint32_t)oy << 12) | (uint32_t)(ox + k);
            }
        }
        FSTAMP(12);
        FAST_WGREC_END();
        return;
    }
    const int WPT = (nwords + NT - 1) / NT; /* 1 for NT = 256 */
    uint32_t cnt = 0;
    for (int j = 0; j < WPT; j++) {
        const int w = tid * WPT + j;
        if (w < nwords) cnt += __popc(keep[w]);
    }
    const uint32_t incl = wave_incl_scan(cnt);
    if (lane == 63) s_wave_tot[wv] = incl;
    FSTAMP(10);
    __syncthreads();
    FSTAMP(11);
    uint32_t wave_off = 0, total = 0;
#pragma unroll
    for (int k = 0; k < NT / 64; k++) {
        if (k < wv) wave_off += s_wave_tot[k];
        total += s_wave_tot[k];
    }
    if (tid == 0) {
        cout[cell].base = cd.base;
        cout[cell].count = total;
    }
    if (cnt != 0) {
        uint32_t o = cd.base + wave_off + incl - cnt;
        for (int j = 0; j < WPT; j++) {
            const int w = tid * WPT + j;
            if (w >= nwords) break;
            uint32_t bits = keep[w];
            const int kly = w >> 1, kxb = (w & 1) * 32;
            const int ox = cd.x0 + 3 - VSLAM_BORDER + kxb, oy = cd.y0 + 3 - VSLAM_BORDER + kly;
            while (bits) {
                const int k = __ffs(bits) - 1;
                bits &= bits - 1;
                const uint32_t s = sc[(kly + 1) * P + kxb + k + 1];
                cand[o++] = (s << 24) | ((uint32_t)oy << 12) | (uint32_t)(ox + k);
            }
        }
    }
    FSTAMP(12);
    FAST_WGREC_END();
}

/* threads per cell: a cell is 900 pixels, and ~150 of the ~420 instructions a thread executes do not depend on how many
 * pixels it owns (prologue, NMS bookkeeping, compaction), so fewer, busier threads per cell cost fewer instructions in
 * total; VSLAM_FAST_NT = 64 | 128 | 256 selects the variant for A/B runs */
static int fast_v3_nt(const vslam_tuning& T) {
    const int v = T.fast_threads;
    return (v == 64 || v == 128 || v == 256) ? v : 128;
}

void vk_fast_cells_v3(hipStream_t st, const uint8_t* pyr, size_t slot_stride, const BatchSrc& src,
                      const PyramidGeom& g, const CellDesc* cells, int ncells, uint8_t* cand_region,
                      size_t cand_stride, int iniTh, int minTh, int tile_rows, int max_window_w, int max_px, int nslots,
                      const vslam_tuning& T) {
    /* survivor lists: dark-only + "both" share one list (from both ends), bright-only the other; neither can hold
     * more entries than the largest cell interior has pixels.  Windows up to 42 px wide (KITTI, 1080p: 38) fit an
     * LDS pitch of 48 bytes instead of 72: 9.7 KB per cell, 16 cells resident per CU. */
    const int lcap = (max_px + 7) & ~3;
    const bool force72 = T.fast_pitch == 72; /* the wider pitch everywhere, for A/B runs */
    const int P = (max_window_w <= 42 && !force72) ? 48 : 72;
    const int nt = fast_v3_nt(T);
    /* window + score tile + keep words + list; not less than what the pre-test sweep's idle lanes may READ (rows up
     * to a sweep's height below the window, quads past the last one: results dropped, but the addresses stay inside
     * the allocation) */
    const size_t shm = std::max((size_t)tile_rows * P + (size_t)(tile_rows - 4) * P + (size_t)(tile_rows - 6) * 8 + (size_t)lcap * 4 + 16,
                                (size_t)(tile_rows + nt / 8 + 1) * P + 128);
    const int lds_pad = std::max(0, tune_or(T.fast_lds_pad, 0)); /* extra LDS per workgroup (occupancy experiments) */
    const dim3 grid((ncells + 8 * FAST_XCD_CHUNK - 1) / (8 * FAST_XCD_CHUNK) * (8 * FAST_XCD_CHUNK), nslots);
    const int it = std::min(iniTh, 256), mt = std::min(minTh, 256);
#define FAST3_LAUNCH(NT_, P_)                                                                                              \
    hipLaunchKernelGGL((k_fast_cells_v3<NT_, P_>), grid, dim3(NT_), shm + lds_pad, st, pyr, slot_stride, src, g, cells, cand_region, \
                       cand_stride, ncells, it, mt, tile_rows, lcap)
    if (P == 48) {
        if (nt == 64) FAST3_LAUNCH(64, 48);
        else if (nt == 128) FAST3_LAUNCH(128, 48);
        else FAST3_LAUNCH(256, 48);
    } else {
        if (nt == 64) FAST3_LAUNCH(64, 72);
        else if (nt == 128) FAST3_LAUNCH(128, 72);
        else FAST3_LAUNCH(256, 72);
    }
#undef FAST3_LAUNCH
}

/* ------------------------------------------------------------------------------------------------
 * FAST bands (round 4): one workgroup per BAND -- up to four consecutive cells of one cell row -- instead of one per cell.
 * Why: k_fast_cells_v3 spent ~45 % of its instructions on things that exist once per WORKGROUP (cell record, level
 * geometry, staging addresses, zeroing, loop set-up, the output prologue: ~190 of ~1220 wave-instructions per cell in two
 * waves) or once per WINDOW (the 6-px ring of every 30-px cell is staged 1.42 times).  A band stages ONE window of
 * (ncell * wcell + 6) x (hcell + 6) px (1.05 x 1.19), decodes one record, and its four waves share every loop:
 *   - pre-test: the same packed four-pixel compass test, lanes dealt to (row, quad) with 32 / 16 / 8 quads per row, so a
 *     wave sweeps two, four or eight whole rows per iteration and skips iterations that lie below the band;
 *   - ONE survivor list for the band (dark from the front, bright-only from the back of one array that can hold every
 *     pixel of the rows swept at once -- taller bands than that are swept in row chunks that overlap by one row, so that
 *     the list-driven NMS of a chunk never needs a score of the next one);
 *   - one pass of the two-pixel score networks over the band's list: dark pairs from thread 0 up, bright pairs right
 *     behind them, so all but one wave run a single polarity;
 *   - cell-local NMS (fextractor.cpp:800: every cell is its own cv::FAST call) through a per-column flag table -- first /
 *     last column of its cell -- that masks the neighbours across a cell border;
 *   - "empty at iniThFAST => the whole cell at minThFAST" (fextractor.cpp:800-807) per cell: the second stage sweeps only
 *     the quad columns of the band's empty cells;
 *   - output: wave w writes cell w's candidates in raster order into the cell's fixed segment (same layout as v3).
 * Scores do not depend on the cell a pixel belongs to, so the result is identical to one cv::FAST per cell.
 * ---------------------------------------------------------------------------------------------- */
#define FB_MAXC 4     /* cells per band */
#ifndef FB_XCD_CHUNK
#define FB_XCD_CHUNK 4
#endif
#ifdef VSLAM_FAST_COUNT /* diagnostic build: how often a wave executes each block of k_fast_bands (tools/fast_band_counts.py
 * multiplies by the blocks' static instruction counts: the per-phase instruction budget) */
__device__ unsigned long long g_fb_cnt[32];
#define FCNT(k)                                                                                       \
    do {                                                                                              \
        const unsigned long long m_ = __ballot(1);                                                    \
        if ((int)(threadIdx.x & 63) == __ffsll((unsigned long long)m_) - 1) atomicAdd(&g_fb_cnt[k], 1ull); \
    } while (0)
#define FCNTN(k, n)                                                                                   \
    do {                                                                                              \
        const unsigned long long m_ = __ballot(1);                                                    \
        if ((int)(threadIdx.x & 63) == __ffsll((unsigned long long)m_) - 1) atomicAdd(&g_fb_cnt[k], (unsigned long long)(n)); \
    } while (0)
extern "C" int vslam_dbg_fast_band_counts(unsigned long long* out32, int reset) {
    if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_fb_cnt), sizeof(g_fb_cnt)) != hipSuccess) return -3;
    if (reset) {
        unsigned long long z[32] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_fb_cnt), z, sizeof(z)) != hipSuccess) return -3;
    }
    return 0;
}
#else
#define FCNT(k) do { } while (0)
#define FCNTN(k, n) do { } while (0)
#endif
/* ds_write_b16 of `code` by the lanes of mA to atA, then by the lanes of mB to atB (masks are subsets of exec) */
__device__ __forceinline__ void lds_store_b16_two(uint64_t mA, uint64_t mB, uint32_t atA, uint32_t atB, uint32_t code) {
    uint64_t save;
    asm volatile("s_mov_b64 %0, exec\n\t"
                 "s_mov_b64 exec, %1\n\tds_write_b16 %3, %5\n\t"
                 "s_mov_b64 exec, %2\n\tds_write_b16 %4, %5\n\t"
                 "s_mov_b64 exec, %0"
                 : "=&s"(save) : "s"(mA), "s"(mB), "v"(atA), "v"(atB), "v"(code) : "memory");
}
__device__ __forceinline__ uint32_t pk_mul_lo(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t sub_twice(uint32_t a, uint32_t s_base) { /* s_base - 2 * a; s_base wave-uniform (an SGPR operand) */
    uint32_t r;
    asm("v_mad_i32_i24 %0, %1, -2, %2" : "=v"(r) : "v"(a), "s"(s_base));
    return r;
}
__device__ __forceinline__ uint32_t lane_rank_from(uint64_t m, uint32_t start) { /* start + set bits of m below this lane */
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, start));
}

template <int NT, int P> /* P: LDS pitch = interior columns + 8: 136 (bands of up to 128 columns) or 72 (up to 64) */
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_num_sgpr(80)))
k_fast_bands(const uint8_t* __restrict__ pyr, size_t slot_stride, BatchSrc src, PyramidGeom g,
             const BandDesc* __restrict__ bands, int nbands, const uint4* __restrict__ classes,
             const CellDesc* __restrict__ cells, uint8_t* cand_region, size_t cand_stride, int ncells, int iniTh, int minTh,
             int nslots, int lds_total, int by_image) {
    extern __shared__ __align__(16) uint8_t smemb[];
    constexpr int KW = (P - 8) / 32; /* keep words per interior row */
    constexpr int LOGW = P == 136 ? 7 : 6; /* log2 of the interior columns a row can have */
    constexpr int NW = NT / 64;
    __shared__ uint32_t s_cnt[2]; /* nD | nB << 16 of the chunk being swept; the other one is zeroed meanwhile */
    __shared__ uint32_t s_any;    /* bit c: cell c of the band kept a corner */
    __shared__ uint32_t s_nq;     /* stage 2: quad columns to sweep */

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6); /* wave-uniform, and said so: loop bounds and per-cell records go scalar */
    /* XCD-aware order (workgroups b and b + 8 share an XCD and its L2).  Full batches: XCD k takes the images k, k + 8, ...
     * whole -- neighbouring bands share six window columns, the next cell row shares six window rows, and the eight XCDs
     * get the same mix of levels -- so every image byte is fetched into one L2 once.  Fewer than eight images: chunks of
     * FB_XCD_CHUNK bands dealt round-robin (rotated by the slot), as k_fast_cells_v3 does with cells. */
    int slot, band;
    {
        const int xi = (int)(blockIdx.x >> 3), xk = (int)(blockIdx.x & 7);
        if (by_image) {
            const int sl = xi / nbands;
            band = xi - sl * nbands;
            slot = xk + 8 * sl;
        } else {
            slot = blockIdx.y;
            band = ((xi / FB_XCD_CHUNK) * 8 + (int)((blockIdx.x + blockIdx.y) & 7)) * FB_XCD_CHUNK + xi % FB_XCD_CHUNK;
        }
        if (slot >= nslots || band >= nbands) return;
    }
    FCNT(0);
    const uint4 bd = ((const uint4*)bands)[band]; /* scalar cache */
    const int cell0 = (int)bd.x;
    const int level = (int)(bd.y & 15u), ncell = (int)((bd.y >> 4) & 15u), wcell = (int)((bd.y >> 8) & 255u);
    const int bclass = (int)(bd.y >> 16);
    const int bx0 = (int)(bd.z & 0xFFFFu), by0 = (int)(bd.z >> 16);
    const int ww = (int)(bd.w & 0xFFFFu), wh = (int)(bd.w >> 16);
    const int iw = ww - 6, ih = wh - 6;
    const LevelGeom lg = g.lv[level];
    int pitch;
    const uint8_t* img = level_base_v2(pyr, slot_stride, src, lg, level, slot, &pitch);

    /* LDS: window | score tile (ih + 2 rows, interior at row 1, column 1) | keep words | column tables | survivor list */
    const int whe = (wh + 1) & ~1; /* 136 * even: the score tile stays 16-byte aligned */
    uint8_t* win = smemb;          /* window column c at LDS column c + 1 */
    uint8_t* sc = win + whe * P;
    uint32_t* keep = (uint32_t*)(sc + (((ih + 2) * P + 15) & ~15));
    uint8_t* cellbit = (uint8_t*)(keep + ih * KW); /* 1 << (cell of interior column x), 0 outside the interior */
    uint8_t* cellfl = cellbit + 136;                  /* bit 0: first column of its cell, bit 1: last column */
    uint8_t* qtab = cellfl + 136;                     /* stage 2: the quad columns of the empty cells */
    uint16_t* list = (uint16_t*)(qtab + 32);
    const int lcap = ((lds_total - (int)((uint8_t*)list - smemb)) >> 1) & ~1; /* entries; even */

    /* stage the window: columns bx0 - 1 .. of wh rows, 17 eight-byte lanes per row; every load is issued before the first
     * LDS store, and the tiles are zeroed / the tables fetched while they are in flight */
    constexpr int LPR = P / 8, RPS = NT / LPR;
    constexpr int NSW = (52 + RPS - 1) / RPS > 4 ? 4 : (52 + RPS - 1) / RPS; /* sweeps kept in registers: windows of up to 52 rows */
    const int srow = tid / LPR, scol = (tid - srow * LPR) * 8;
    const bool stager = tid < RPS * LPR && scol < ww + 1;
    const uint8_t* gsrc = img + (size_t)by0 * pitch + bx0 - 1;
    uint2 wreg[NSW];
#pragma unroll
    for (int u = 0; u < NSW; u++) {
        const int y = srow + u * RPS;
        wreg[u] = make_uint2(0u, 0u);
        if (stager && y < wh) wreg[u] = *(const uint2*)(gsrc + (uint32_t)(__umul24((uint32_t)y, (uint32_t)pitch) + (uint32_t)scol));
    }
    {
        /* the band's column tables (host: one 272-byte record per (cell pitch, interior width) class) */
        uint4 tabv = make_uint4(0u, 0u, 0u, 0u);
        if (tid < 17) tabv = classes[bclass * 17 + tid];
        const int nz = (int)(((uint8_t*)(keep + ih * KW) - sc) >> 4); /* score tile + keep words, 16-byte stores */
        for (int i = tid; i < nz; i += NT) ((uint4*)sc)[i] = make_uint4(0u, 0u, 0u, 0u);
        if (tid < 17) ((uint4*)cellbit)[tid] = tabv;
        if (tid >= 64 && tid < 72) ((uint32_t*)qtab)[tid - 64] = 0u;
        if (tid == 0) {
            s_cnt[0] = 0u;
            s_cnt[1] = 0u;
            s_any = 0u;
        }
    }
#pragma unroll
    for (int u = 0; u < NSW; u++) {
        const int y = srow + u * RPS;
        if (stager && y < wh) *(uint2*)(win + y * P + scol) = wreg[u];
    }
    if (stager)
        for (int y = srow + NSW * RPS; y < wh; y += RPS)
            *(uint2*)(win + y * P + scol) = *(const uint2*)(gsrc + (uint32_t)(__umul24((uint32_t)y, (uint32_t)pitch) + (uint32_t)scol));
    __syncthreads();

    const uint32_t* W32 = (const uint32_t*)win;
    const int QW = (iw + 3) >> 2; /* quad columns of the band */
    /* rows swept at once.  A pixel that passes both compass tests is listed twice (dark and bright), every other survivor
     * once: the list holds every pixel of lcap / (P - 8) rows once -- the optimistic chunk, all of a usual band -- and of half
     * as many rows twice -- the chunk after an overflow, which the two counters show before anything reads the list */
    int CR = min(ih, lcap >> LOGW);
    const int sc_off = whe * P - 2 * P - 3; /* score byte of a pixel relative to its window byte */
    const uint32_t list_lo = lds_addr(list), list_hi = lds_addr(list) + 2u * (uint32_t)(lcap - 1);
    int T = iniTh;
    uint32_t act = 0xFFu; /* cells swept in this stage */
    int nq = QW;
    int par = 0;          /* which s_cnt the current chunk counts in */
    for (int stage = 0; stage < 2; stage++) {
        /* lanes per row: 32, 16 or 8 quad slots; a wave sweeps 2, 4 or 8 whole rows per iteration */
        const int sh = nq > 16 ? 5 : nq > 8 ? 4 : 3;
        const int qs = tid & ((1 << sh) - 1), qly = tid >> sh;
        const int qx = stage ? (int)qtab[qs] : qs; /* slots past nq read column 0 and pass nothing */
        /* per-pixel thresholds: T inside the cells swept in this stage, else one nothing passes (v - 0x7FFF saturates to 0,
         * v + 0x7FFF is above every pixel): no validity masks inside the sweep.  Halves of TTv[par]: pixels par, par + 2 */
        uint32_t TTv[2];
        {
            uint32_t m = *(const uint32_t*)(cellbit + 4 * qx) & (act * 0x01010101u);
            if (qs >= nq) m = 0u;
            const uint32_t dT = (uint32_t)(0x7FFF - T) * 0x00010001u;
#pragma unroll
            for (int pp = 0; pp < 2; pp++) {
                const uint32_t h = pk_min(__builtin_amdgcn_perm(0u, m, pp ? 0x0c030c01u : 0x0c020c00u), 0x00010001u);
                TTv[pp] = pk_sub_sat(0x7FFF7FFFu, pk_mul_lo(h, dT));
            }
        }
        const int RPI = NT >> sh;         /* rows per iteration of the workgroup */
        const int wrow = (wv * 64) >> sh; /* first row of this wave inside an iteration */
        const uint32_t codeq = (uint32_t)((qly << 8) + 4 * qx);
        for (int r0 = 0;;) {
            const int r1 = min(r0 + CR, ih);
            for (int ly0 = r0; ly0 + wrow < r1; ly0 += RPI) { /* wave-uniform: iterations below the chunk are skipped */
                FCNT(1);
                const int ly = ly0 + qly;
                if (ly >= r1) continue; /* rows below the chunk: those lanes sit the iteration out (lane 0 never does) */
                const uint32_t* rowc = W32 + (ly + 3) * (P / 4) + qx;
                const uint32_t A0 = rowc[0], C = rowc[1], E = rowc[2];
                const uint32_t U = W32[ly * (P / 4) + qx + 1], Dn = W32[(ly + 6) * (P / 4) + qx + 1];
                const uint32_t Lf = __builtin_amdgcn_alignbyte(C, A0, 1); /* columns x-3 */
                const uint32_t Rt = __builtin_amdgcn_alignbyte(E, C, 3);  /* columns x+3 */
#define EVN(x) __builtin_amdgcn_perm(0u, (x), 0x0c020c00u) /* pixels 0,2 as u16 halves */
#define ODD(x) __builtin_amdgcn_perm(0u, (x), 0x0c030c01u) /* pixels 1,3 */
                uint32_t passD[2], passB[2];
#pragma unroll
                for (int pp = 0; pp < 2; pp++) {
                    const uint32_t v = pp ? ODD(C) : EVN(C);
                    const uint32_t TT = TTv[pp];
                    const uint32_t vm = pk_sub_sat(v, TT), vp = pk_add(v, TT);
                    const uint32_t u = pp ? ODD(U) : EVN(U), d = pp ? ODD(Dn) : EVN(Dn);
                    const uint32_t l = pp ? ODD(Lf) : EVN(Lf), r = pp ? ODD(Rt) : EVN(Rt);
                    passD[pp] = pk_sub_sat(vm, pk_max(pk_min(d, u), pk_min(r, l)));
                    passB[pp] = pk_sub_sat(pk_min(pk_max(d, u), pk_max(r, l)), vp);
                }
#undef EVN
#undef ODD
                /* survivors go straight to the lists: per pixel column j one compare per polarity whose lane mask lands in an
                 * SGPR pair, ranks from v_mbcnt, totals from s_bcnt1, ONE LDS atomic per wave and iteration for the list space,
                 * two stores under the lane masks.  The dark list grows from the front, the bright one from the back. */
                uint64_t mD[4], mB[4];
                uint32_t totD = 0, totB = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    mD[j] = (j >> 1) ? half_hi_nonzero(passD[j & 1]) : half_lo_nonzero(passD[j & 1]);
                    mB[j] = (j >> 1) ? half_hi_nonzero(passB[j & 1]) : half_lo_nonzero(passB[j & 1]);
                    totD += (uint32_t)__popcll(mD[j]);
                    totB += (uint32_t)__popcll(mB[j]);
                }
                if (totD | totB) { /* wave-uniform */
                    FCNT(2);
                    /* the list space of this wave's survivors: one returning LDS add by lane 0.  Its round trip is covered
                     * by the ranks, which do not need the base: the add is issued here and waited for behind them */
                    uint32_t base = 0;
                    if (lane == 0)
                        asm volatile("ds_add_rtn_u32 %0, %1, %2" : "=v"(base) : "v"(lds_addr(&s_cnt[par])), "v"(totD | (totB << 16)) : "memory");
                    const uint32_t code0 = codeq + ((uint32_t)ly0 << 8);
                    uint32_t rD[4], rB[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        rD[j] = lane_rank(mD[j]);
                        rB[j] = lane_rank(mB[j]);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(base) : : "memory");
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    /* the entries of the pixel columns before j go into the (scalar) base of column j */
                    uint32_t aD = list_lo + 2u * (base & 0xFFFFu), aB = list_hi - 2u * (base >> 16);
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        lds_store_b16_two(mD[j], mB[j], aD + 2u * rD[j], sub_twice(rB[j], aB), code0 + j);
                        aD += 2u * (uint32_t)__popcll(mD[j]);
                        aB -= 2u * (uint32_t)__popcll(mB[j]);
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* the masked stores above are not tracked by the compiler */
            __syncthreads();
            int nD, nB;
            {
                const uint32_t tot = s_cnt[par];
                nD = (int)(tot & 0xFFFFu);
                nB = (int)(tot >> 16);
            }
            if (nD + nB > lcap) { /* block-uniform: the two lists ran into each other -- these rows again, half as many */
                CR = max(2, lcap >> (LOGW + 1));
                par ^= 1; /* the other counter is zero; this one is zeroed one chunk later, like every used one */
                continue;
            }
            if (tid == 0) s_cnt[par ^ 1] = 0u; /* the next chunk's / stage's counter: its last readers passed a barrier ago */
            FCNT(11);
            if (wv == 0) { FCNTN(13, nD); FCNTN(14, nB); }
            /* networks: TWO list entries per thread in the halves of packed registers.  First every dark pair, then -- behind a
             * barrier -- every bright pair, which keeps the larger of its score and the one already in the tile: a pixel on
             * both lists ends with max(dark, bright), and no wave runs a second network for the few lanes that hold one */
            const int nDp = (nD + 1) >> 1, nBp = (nB + 1) >> 1;
            /* a thread keeps the codes of the first pair it scores on either list and runs the NMS of those four pixels itself,
             * behind the barrier: no second walk of the list for them (0xFFFF: no entry; its row 255 lies below every band) */
            uint32_t keptD = 0xFFFFFFFFu, keptB = 0xFFFFFFFFu;
            for (int u = tid; u < nDp; u += NT) {
                FCNT(3);
                const uint32_t e2 = *(const uint32_t*)(list + 2 * u);
                const uint32_t eA = e2 & 0xFFFFu, eB = 2 * u + 1 < nD ? e2 >> 16 : eA;
                if (u == tid) keptD = eA | (eB << 16);
                const uint8_t* cA = win + ((eA >> 8) + 3) * P + (eA & 255u) + 4;
                const uint8_t* cB = win + ((eB >> 8) + 3) * P + (eB & 255u) + 4;
                const uint32_t a = pk_sub_sat(fast_pair_score<1, P>(cA, cB), 0x00010001u); /* OpenCV's score: max(..) - 1, not below 0 */
                ((uint8_t*)cA)[sc_off] = (uint8_t)a;
                ((uint8_t*)cB)[sc_off] = (uint8_t)(a >> 16);
            }
            __syncthreads();
            for (int u = tid; u < nBp; u += NT) { /* bright entries 2u (high half) and 2u + 1 (low half), counted from the back */
                FCNT(5);
                const uint32_t e2 = *(const uint32_t*)(list + lcap - 2 - 2 * u);
                const uint32_t eA = e2 >> 16, eB = 2 * u + 1 < nB ? e2 & 0xFFFFu : eA;
                if (u == tid) keptB = eA | (eB << 16);
                const uint8_t* cA = win + ((eA >> 8) + 3) * P + (eA & 255u) + 4;
                const uint8_t* cB = win + ((eB >> 8) + 3) * P + (eB & 255u) + 4;
                const uint32_t old = (uint32_t)cA[sc_off] | ((uint32_t)cB[sc_off] << 16);
                const uint32_t a = pk_max(pk_sub_sat(fast_pair_score<-1, P>(cA, cB), 0x00010001u), old);
                ((uint8_t*)cA)[sc_off] = (uint8_t)a;
                ((uint8_t*)cB)[sc_off] = (uint8_t)(a >> 16);
            }
            __syncthreads();
            /* NMS at T where a score exists (listed pixels below T cannot suppress anything), for the rows whose 3x3
             * neighbourhood is complete: all of a chunk but its last row, which the next chunk lists again */
            const bool last = r1 >= ih;
            const int nhi = last ? ih - 1 : r1 - 2;
            auto nms_one = [&](uint32_t code) {
                FCNT(6);
                const int ly = (int)(code >> 8), x = (int)(code & 255u);
                if (ly > nhi) return;
                const uint8_t* q = sc + (ly + 1) * P + x + 1;
                const int s = q[0];
                if (s >= T) {
                    FCNT(7);
                    const int fl = cellfl[x];
                    /* every cell is its own cv::FAST call: scores across a cell border count as 0 */
                    int ml = max(max((int)q[-P - 1], (int)q[-1]), (int)q[P - 1]);
                    int mr = max(max((int)q[-P + 1], (int)q[1]), (int)q[P + 1]);
                    if (fl & 1) ml = 0;
                    if (fl & 2) mr = 0;
                    const int mx = max(max((int)q[-P], (int)q[P]), max(ml, mr));
                    if (s > mx) {
                        atomicOr(&keep[ly * KW + (x >> 5)], 1u << (x & 31));
                        atomicOr(&s_any, (uint32_t)cellbit[x]);
                    }
                }
            };
            nms_one(keptD & 0xFFFFu);
            nms_one(keptD >> 16);
            nms_one(keptB & 0xFFFFu);
            nms_one(keptB >> 16);
            /* entries beyond the first pair of every thread (more than 2 * NT on a list: dense texture, 1080p) */
            for (int i = 2 * NT + tid; i < nD; i += NT) nms_one(list[i]);
            for (int i = 2 * NT + tid; i < nB; i += NT) nms_one(list[lcap - 1 - i]);
            par ^= 1;
            if (last) break;
            __syncthreads(); /* the next chunk's sweep overwrites the list */
            r0 = r1 - 1;
        }
        __syncthreads();
        /* cells empty at iniThFAST: again, whole, at minThFAST (fextractor.cpp:800-807).  Scores already in the tile
         * belong to pixels that are listed again (the pre-test is monotone in T) and are recomputed to the same value. */
        const uint32_t empty = ((1u << ncell) - 1u) & ~s_any;
        if (stage == 1 || minTh == iniTh || empty == 0u) break; /* block-uniform */
        FCNT(10);
        T = minTh;
        act = empty;
        if (wv == 0) { /* the quad columns that touch an empty cell, in order */
            const uint32_t cb = lane < QW ? *(const uint32_t*)(cellbit + 4 * lane) : 0u;
            const bool on = (cb & (empty * 0x01010101u)) != 0u;
            const uint64_t m = __ballot(on);
            if (on) qtab[lane_rank(m)] = (uint8_t)lane;
            if (lane == 0) s_nq = (uint32_t)__popcll(m);
        }
        __syncthreads();
        nq = (int)s_nq;
    }

    /* ordered output: wave w owns cell w; lane = (row, 32-column half) of a pass of 32 rows, so raster order is the
     * lane order and one wave scan places every candidate */
    uint32_t* hdr = (uint32_t*)(cand_region + (size_t)slot * cand_stride);
    CellOut* cout = (CellOut*)(hdr + 2);
    uint32_t* cand = (uint32_t*)(cout + ncells);
    for (int c = wv; c < ncell; c += NW) {
        const int cx0 = c * wcell, cw = min(wcell, iw - cx0);
        const uint32_t cbase = ((const uint4*)cells)[cell0 + c].w; /* CellDesc::base */
        const int half = lane & 1;
        const int b0 = cx0 + 32 * half, nb = cw - 32 * half; /* first column and columns of this lane's half */
        const int ox = bx0 + 3 - VSLAM_BORDER + b0, oy = by0 + 3 - VSLAM_BORDER;
        uint32_t running = 0;
        for (int rb = 0; rb < ih; rb += 32) {
            FCNT(8);
            const int row = rb + (lane >> 1);
            uint32_t bits = 0u;
            if (row < ih && nb > 0) {
                const uint32_t* kr = keep + row * KW;
                const int w = b0 >> 5;
                const uint32_t lo = kr[w], hi = w + 1 < KW ? kr[w + 1] : 0u;
                bits = __builtin_amdgcn_alignbit(hi, lo, (uint32_t)(b0 & 31));
                if (nb < 32) bits &= (1u << nb) - 1u;
            }
            const uint32_t c1 = (uint32_t)__popc(bits);
            const uint32_t in1 = wave_incl_scan(c1);
            const uint32_t total1 = (uint32_t)__builtin_amdgcn_readlane((int)in1, 63);
            if (bits) {
                uint32_t o = cbase + running + in1 - c1;
                const uint8_t* srow_ = sc + (row + 1) * P + b0 + 1;
                while (bits) {
                    FCNT(9);
                    const int k = __ffs(bits) - 1;
                    bits &= bits - 1;
                    cand[o++] = ((uint32_t)srow_[k] << 24) | ((uint32_t)(oy + row) << 12) | (uint32_t)(ox + k);
                }
            }
            running += total1;
        }
        if (lane == 0) {
            cout[cell0 + c].base = cbase;
            cout[cell0 + c].count = running;
        }
    }
}

/* LDS a band of wh window rows needs before its survivor list (k_fast_bands' layout); P = LDS pitch */
static size_t fast_band_fixed_lds(int wh, int P) {
    const int ih = wh - 6, whe = (wh + 1) & ~1, KW = (P - 8) / 32;
    return (size_t)whe * P + (size_t)((((ih + 2) * P) + 15) & ~15) + (size_t)ih * KW * 4 + 136 + 136 + 32;
}

int vk_fast_bands_check(int max_wh, int max_iw, int max_cells_per_band) {
    /* the kernel's limits: 128 interior columns, codes ly << 8 | x with ly < 128, FB_MAXC cells per band */
    if (max_iw > 128 || max_wh > 6 + 127 || max_cells_per_band > FB_MAXC) return -1;
    return 0;
}

void vk_fast_bands(hipStream_t st, const uint8_t* pyr, size_t slot_stride, const BatchSrc& src, const PyramidGeom& g,
                   const BandDesc* bands, int nbands, const uint8_t* classes, const CellDesc* cells, int ncells, uint8_t* cand_region,
                   size_t cand_stride, int iniTh, int minTh, int max_wh, int max_iw, int nslots, const vslam_tuning& T) {
    /* one shape: bands of up to 128 interior columns (four 31-px cells) by four waves, LDS pitch 136.  (A second shape --
     * bands of up to 64 columns by two waves, pitch 72: the same work per cell behind barriers of two waves instead of four --
     * was built in round 4 and measured no faster: 90.6 vs 89.7 us per 32 KITTI images, 41.5 M instead of 39.2 M
     * instructions for the shorter rows' idle lanes; removed.) */
    (void)max_iw;
    constexpr int P = 136, NT = 256;
    /* LDS per workgroup: 32 waves per CU fit when a four-wave band takes at most 20 KB.  The list
     * gets what the window, the score tile and the tables leave: enough for every pixel of the usual band (cells of up to
     * ~34 rows: one chunk); taller bands are swept in chunks (the kernel derives the chunk height from the space it
     * finds).  Never less than the tallest band needs for chunks of 8 rows with every pixel listed twice, nor than what a
     * sweep's idle lanes may READ (rows up to an iteration's height below the window: results dropped, but the
     * addresses stay inside the allocation). */
    const size_t budget = 20480 - 64; /* 64: the kernel's static variables */
    size_t lds = std::max(budget, fast_band_fixed_lds(max_wh, P) + 2 * (size_t)(8 * 2 * (P - 8)));
    lds = std::max(lds, (size_t)(max_wh + NT / 8 + 1) * P + 256);
    lds = (lds + 15) & ~(size_t)15;
    const int lds_pad = std::max(0, tune_or(T.fast_lds_pad, 0)); /* extra LDS per workgroup (occupancy experiments) */
    const int it = std::min(iniTh, 256), mt = std::min(minTh, 256);
    const int by_image = nslots >= 8 ? 1 : 0;
    dim3 grid;
    if (by_image) grid = dim3((unsigned)(8 * nbands * ((nslots + 7) / 8)), 1);
    else grid = dim3((unsigned)((nbands + 8 * FB_XCD_CHUNK - 1) / (8 * FB_XCD_CHUNK) * (8 * FB_XCD_CHUNK)), (unsigned)nslots);
#define FB_LAUNCH(NT_, P_)                                                                                                   \
    hipLaunchKernelGGL((k_fast_bands<NT_, P_>), grid, dim3(NT_), lds + lds_pad, st, pyr, slot_stride, src, g, bands, nbands, \
                       (const uint4*)classes, cells, cand_region, cand_stride, ncells, it, mt, nslots, (int)lds, by_image)
    FB_LAUNCH(256, 136);
#undef FB_LAUNCH
}
