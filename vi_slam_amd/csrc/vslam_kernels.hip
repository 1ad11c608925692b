/* vslam_kernels.hip -- hand-written gfx950 kernels of the ORB front-end.
 *
 * All integer stages are exact; the few float operations (fastAtan2 polynomial, pattern rotation) are
 * written with explicit non-fused intrinsics and the file is built with -ffp-contract=off.
 * wave = 64 everywhere.  Reference lines each kernel reproduces are cited at the kernel.
 */
#include "vslam_kernels.h"

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

__device__ __forceinline__ int reflect101(int p, int len) {
    /* BORDER_REFLECT_101 for |overshoot| < len (radius 3 vs len >= 7 always holds here) */
    if (p < 0) p = -p;
    if (p >= len) p = 2 * (len - 1) - p;
    return p;
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
 * K2  FAST-9/16 per 30-px cell with threshold fallback and cell-local 3x3 NMS
 *     (FExtractor::ComputeKeyPointsOctTree cell loop, fextractor.cpp:780-819; cv::FAST + cornerScore).
 *
 * One workgroup per executed cell.  The cell window (interior + 3-px ring) is staged in LDS with
 * coalesced row reads; every interior pixel gets its FAST score S = max over the 16 nine-pixel arcs of
 * the Bresenham ring of min |diff| (both polarities) - 1, computed branch-free with min3/max3 networks.
 * "Corner at threshold T" == S >= T and OpenCV's response == S, so ONE score tile serves both the
 * iniThFAST pass and the minThFAST fallback.  NMS: a corner survives iff S is strictly greater than its
 * 8 neighbours, where positions outside the cell interior count 0 (cv::FAST runs on the cell
 * sub-image).  Survivors are emitted in raster order into a segment obtained with one global atomic per
 * cell; the host (or a later kernel) walks cells in index order, which reproduces vToDistributeKeys.
 * ---------------------------------------------------------------------------------------------- */
__device__ __forceinline__ int imin3(int a, int b, int c) { return min(a, min(b, c)); }
__device__ __forceinline__ int imax3(int a, int b, int c) { return max(a, max(b, c)); }

/* ring offsets in OpenCV order (fast_score.cpp makeOffsets, patternSize 16): (dx,dy) */
#define RING_AT(c, P, k)                                                                                   \
    ((k) == 0 ? (c)[3 * (P)] : (k) == 1 ? (c)[3 * (P) + 1] : (k) == 2 ? (c)[2 * (P) + 2]                   \
     : (k) == 3 ? (c)[(P) + 3] : (k) == 4 ? (c)[3] : (k) == 5 ? (c)[-(P) + 3]                              \
     : (k) == 6 ? (c)[-2 * (P) + 2] : (k) == 7 ? (c)[-3 * (P) + 1] : (k) == 8 ? (c)[-3 * (P)]              \
     : (k) == 9 ? (c)[-3 * (P) - 1] : (k) == 10 ? (c)[-2 * (P) - 2] : (k) == 11 ? (c)[-(P) - 3]            \
     : (k) == 12 ? (c)[-3] : (k) == 13 ? (c)[(P) - 3] : (k) == 14 ? (c)[2 * (P) - 2] : (c)[3 * (P) - 1])

__device__ __forceinline__ int fast_score(const uint8_t* c, const int P) {
    const int v = c[0];
    int d[16];
#pragma unroll
    for (int k = 0; k < 16; k++) d[k] = v - (int)RING_AT(c, P, k);
    int lo3[16], hi3[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        lo3[k] = imin3(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
        hi3[k] = imax3(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
    }
    int A = -256, B = 256; /* A = max_arcs min d (dark), B = min_arcs max d (bright = -B) */
#pragma unroll
    for (int k = 0; k < 16; k++) {
        A = max(A, imin3(lo3[k], lo3[(k + 3) & 15], lo3[(k + 6) & 15]));
        B = min(B, imax3(hi3[k], hi3[(k + 3) & 15], hi3[(k + 6) & 15]));
    }
    const int s = max(A, -B) - 1;
    return s < 0 ? 0 : s; /* 0..254 */
}

__global__ void __launch_bounds__(256)
k_fast_cells(const uint8_t* __restrict__ pyr, size_t slot_stride, BatchSrc src, PyramidGeom g,
             const CellDesc* __restrict__ cells, uint8_t* cand_region, size_t cand_stride, int ncells,
             int cand_cap, int iniTh, int minTh, int tile_pitch, int tile_rows) {
    extern __shared__ __align__(16) uint8_t smem[];
    uint8_t* win = smem;                                              /* tile_rows x tile_pitch */
    uint8_t* sc = win + tile_rows * tile_pitch;                       /* (tile_rows-4) x tile_pitch */
    uint32_t* keep = (uint32_t*)(sc + (tile_rows - 4) * tile_pitch);  /* ceil(max_px/32) words */
    __shared__ uint32_t s_wave_tot[4];
    __shared__ int s_any_ini;

    const int tid = threadIdx.x;
    const int slot = blockIdx.y;
    const CellDesc cd = cells[blockIdx.x];
    const int level = cd.level;
    const LevelGeom lg = g.lv[level];
    int pitch;
    const uint8_t* img = level_base(pyr, slot_stride, src, lg, level, slot, &pitch);
    const int ww = cd.x1 - cd.x0, wh = cd.y1 - cd.y0; /* window */
    const int iw = ww - 6, ih = wh - 6;               /* interior */
    const int npx = iw * ih;
    const int nwords = (npx + 31) >> 5;

    /* stage the window: consecutive lanes read consecutive bytes of a row */
    for (int i = tid; i < ww * wh; i += 256) {
        const int y = i / ww, x = i - y * ww;
        win[y * tile_pitch + x] = img[(size_t)(cd.y0 + y) * pitch + cd.x0 + x];
    }
    /* zero the score tile (its 1-px frame must read 0) and the keep mask */
    for (int i = tid; i < (ih + 2) * tile_pitch; i += 256) sc[i] = 0;
    for (int i = tid; i < nwords; i += 256) keep[i] = 0;
    if (tid == 0) s_any_ini = 0;
    __syncthreads();

    /* scores of the interior */
    {
        int ly = tid / iw, lx = tid - ly * iw;
        const int sy = 256 / iw, sx = 256 - sy * iw;
        for (int p = tid; p < npx; p += 256) {
            const uint8_t* c = win + (ly + 3) * tile_pitch + lx + 3;
            sc[(ly + 1) * tile_pitch + lx + 1] = (uint8_t)fast_score(c, tile_pitch);
            ly += sy;
            lx += sx;
            if (lx >= iw) { lx -= iw; ly++; }
        }
    }
    __syncthreads();

    /* strict local maxima with S >= minTh; note whether any of them reaches iniTh */
    {
        int ly = tid / iw, lx = tid - ly * iw;
        const int sy = 256 / iw, sx = 256 - sy * iw;
        int any_ini = 0;
        for (int p = tid; p < npx; p += 256) {
            const uint8_t* q = sc + (ly + 1) * tile_pitch + lx + 1;
            const int s = q[0];
            if (s >= minTh) {
                const int P = tile_pitch;
                const int m = max(max(max((int)q[-P - 1], (int)q[-P]), max((int)q[-P + 1], (int)q[-1])),
                                  max(max((int)q[1], (int)q[P - 1]), max((int)q[P], (int)q[P + 1])));
                if (s > m) {
                    atomicOr(&keep[p >> 5], 1u << (p & 31));
                    if (s >= iniTh) any_ini = 1;
                }
            }
            ly += sy;
            lx += sx;
            if (lx >= iw) { lx -= iw; ly++; }
        }
        if (any_ini) s_any_ini = 1; /* benign race: all writers store 1 */
    }
    __syncthreads();
    const int T = s_any_ini ? iniTh : minTh;

    /* ordered compaction: thread w owns keep word w (nwords <= 256 is guaranteed by the host) */
    uint32_t bits = 0;
    if (tid < nwords) {
        uint32_t b = keep[tid];
        if (T != minTh) { /* drop survivors below iniTh */
            uint32_t r = b;
            while (r) {
                const int k = __ffs(r) - 1;
                r &= r - 1;
                const int p = (tid << 5) + k;
                const int ly = p / iw, lx = p - ly * iw;
                if (sc[(ly + 1) * tile_pitch + lx + 1] < T) b &= ~(1u << k);
            }
        }
        bits = b;
    }
    const uint32_t cnt = __popc(bits);
    /* block exclusive scan of cnt */
    uint32_t incl = cnt;
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) s_wave_tot[wv] = incl;
    __syncthreads();
    uint32_t wave_off = 0, total = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (k < wv) wave_off += s_wave_tot[k];
        total += s_wave_tot[k];
    }
    uint32_t* hdr = (uint32_t*)(cand_region + (size_t)slot * cand_stride);
    CellOut* cout = (CellOut*)(hdr + 2);
    uint32_t* cand = (uint32_t*)(cout + ncells);
    if (tid == 0) {
        cout[blockIdx.x].base = cd.base;
        cout[blockIdx.x].count = total;
    }
    (void)cand_cap;
    (void)hdr;
    const uint32_t base = cd.base;
    if (bits == 0) return;
    uint32_t o = base + wave_off + incl - cnt;
    const int ox = cd.x0 + 3 - VSLAM_BORDER, oy = cd.y0 + 3 - VSLAM_BORDER;
    while (bits) {
        const int k = __ffs(bits) - 1;
        bits &= bits - 1;
        const int p = (tid << 5) + k;
        const int ly = p / iw, lx = p - ly * iw;
        const uint32_t s = sc[(ly + 1) * tile_pitch + lx + 1];
        cand[o++] = (s << 24) | ((uint32_t)(oy + ly) << 12) | (uint32_t)(ox + lx);
    }
}

/* ------------------------------------------------------------------------------------------------
 * K3  GaussianBlur 7x7 sigma 2, CV_8U fixed point, BORDER_REFLECT_101 (fextractor.cpp:1085-1086 ->
 *     OpenCV GaussianBlurFixedPoint): row pass u8*u8.8 -> u16 exact, column pass -> (acc + 2^15) >> 16.
 *     64x16 output tile per workgroup, input tile + halo staged in LDS, row-pass result kept in LDS.
 * ---------------------------------------------------------------------------------------------- */
#define BLUR_TW 64
#define BLUR_TH 16
struct BlurTaps { int32_t k[7]; };

__global__ void __launch_bounds__(256)
k_blur7(const uint8_t* __restrict__ pyr, size_t slot_stride, BatchSrc src, PyramidGeom g, uint8_t* blur,
        const uint32_t* __restrict__ tiles, BlurTaps taps) {
    __shared__ uint8_t s_in[(BLUR_TH + 6) * (BLUR_TW + 8)];
    __shared__ uint16_t s_h[(BLUR_TH + 6) * BLUR_TW];
    const int tid = threadIdx.x;
    const int slot = blockIdx.y;
    const uint32_t td = tiles[blockIdx.x]; /* level << 24 | ty << 12 | tx */
    const int level = td >> 24, ty = (td >> 12) & 0xFFF, tx = td & 0xFFF;
    const LevelGeom lg = g.lv[level];
    int pitch;
    const uint8_t* img = level_base(pyr, slot_stride, src, lg, level, slot, &pitch);
    const int x0 = tx * BLUR_TW, y0 = ty * BLUR_TH;
    const int IW = BLUR_TW + 6, IP = BLUR_TW + 8, IH = BLUR_TH + 6;
    for (int i = tid; i < IW * IH; i += 256) {
        const int y = i / IW, x = i - y * IW;
        /* positions past the image only feed outputs that are discarded: clamp keeps them in bounds */
        const int gx = min(max(reflect101(x0 + x - 3, lg.w), 0), lg.w - 1);
        const int gy = min(max(reflect101(y0 + y - 3, lg.h), 0), lg.h - 1);
        s_in[y * IP + x] = img[(size_t)gy * pitch + gx];
    }
    __syncthreads();
    for (int i = tid; i < BLUR_TW * IH; i += 256) {
        const int y = i >> 6, x = i & 63;
        const uint8_t* r = s_in + y * IP + x;
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < 7; k++) acc += (uint32_t)taps.k[k] * r[k];
        s_h[y * BLUR_TW + x] = (uint16_t)min(acc, 0xFFFFu);
    }
    __syncthreads();
    uint8_t* out = blur + (size_t)slot * slot_stride + lg.off;
    for (int i = tid; i < BLUR_TW * BLUR_TH; i += 256) {
        const int y = i >> 6, x = i & 63;
        if (x0 + x >= lg.w || y0 + y >= lg.h) continue;
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < 7; k++) acc += (uint32_t)taps.k[k] * s_h[(y + k) * BLUR_TW + x];
        const uint32_t v = (acc + 32768u) >> 16;
        out[(size_t)(y0 + y) * lg.pitch + x0 + x] = (uint8_t)min(v, 255u);
    }
}

/* ------------------------------------------------------------------------------------------------
 * K4  orientation + descriptor, one wave per keypoint
 *     IC_Angle (fextractor.cpp:68-95): int32 moments over the radius-15 disc of the UN-blurred level,
 *     cv::fastAtan2 polynomial in float; computeOrbDescriptor (fextractor.cpp:99-138): rotate the 512
 *     pattern points with a=cosf, b=sinf (glibc-exact, vslam_trig.h), cvRound (round-half-even), sample
 *     the BLURRED level, 256 comparisons -> 4 wave ballots = 32 bytes.
 * ---------------------------------------------------------------------------------------------- */
__constant__ int8_t c_disc_u[768];
__constant__ int8_t c_disc_v[768];
__constant__ int c_disc_n;

void vk_upload_disc(const int8_t* u, const int8_t* v, int n) {
    hipMemcpyToSymbol(HIP_SYMBOL(c_disc_u), u, n);
    hipMemcpyToSymbol(HIP_SYMBOL(c_disc_v), v, n);
    hipMemcpyToSymbol(HIP_SYMBOL(c_disc_n), &n, sizeof(int));
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

__device__ __forceinline__ void orient_describe_one(const uint8_t* __restrict__ pyr, const uint8_t* __restrict__ blur,
                                                    size_t slot_stride, const BatchSrc& src, const PyramidGeom& g,
                                                    const SelKp s, const int8_t* __restrict__ pattern, vslam_kp* kps,
                                                    uint8_t* desc, int cap, int atan_fma) {
    const int lane = threadIdx.x & 63;
    const LevelGeom lg = g.lv[s.level];
    int pitch;
    const uint8_t* img = level_base(pyr, slot_stride, src, lg, s.level, s.slot, &pitch);
    const uint8_t* center = img + (size_t)s.y * pitch + s.x;

    int m10 = 0, m01 = 0;
    for (int i = lane; i < c_disc_n; i += 64) {
        const int u = c_disc_u[i], v = c_disc_v[i];
        const int val = center[v * pitch + u];
        m10 += u * val;
        m01 += v * val;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        m10 += __shfl_xor(m10, o, 64);
        m01 += __shfl_xor(m01, o, 64);
    }
    const float angle = fast_atan2_deg((float)m01, (float)m10, atan_fma);

    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    const float rad = __fmul_rn(angle, factorPI);
    const float a = vslam_trig::glibc_cosf(rad), b = vslam_trig::glibc_sinf(rad);
    const uint8_t* bc = blur + (size_t)s.slot * slot_stride + lg.off + (size_t)s.y * lg.pitch + s.x;
    const int bp = lg.pitch;
    unsigned long long w[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int pair = q * 64 + lane; /* descriptor bit index */
        const char4 pt = ((const char4*)pattern)[pair];
        const float x0 = (float)pt.x, y0 = (float)pt.y, x1 = (float)pt.z, y1 = (float)pt.w;
        const int r0 = __float2int_rn(__fadd_rn(__fmul_rn(x0, b), __fmul_rn(y0, a)));
        const int c0 = __float2int_rn(__fsub_rn(__fmul_rn(x0, a), __fmul_rn(y0, b)));
        const int r1 = __float2int_rn(__fadd_rn(__fmul_rn(x1, b), __fmul_rn(y1, a)));
        const int c1 = __float2int_rn(__fsub_rn(__fmul_rn(x1, a), __fmul_rn(y1, b)));
        const int t0 = bc[r0 * bp + c0], t1 = bc[r1 * bp + c1];
        w[q] = __ballot(t0 < t1);
    }
    vslam_kp* okp = kps + (size_t)s.slot * cap + s.out;
    if (lane == 0) { /* fextractor.cpp:828-838 (octave, size), :1114-1116 (pt *= scale) */
        vslam_kp o;
        o.x = s.level ? __fmul_rn((float)s.x, lg.scale) : (float)s.x;
        o.y = s.level ? __fmul_rn((float)s.y, lg.scale) : (float)s.y;
        o.size = (float)(int)__fmul_rn(31.f, lg.scale);
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

/* host-selected keypoints (quadtree on the host): one flat list for the batch */
__global__ void __launch_bounds__(256)
k_orient_describe(const uint8_t* __restrict__ pyr, const uint8_t* __restrict__ blur, size_t slot_stride,
                  BatchSrc src, PyramidGeom g, const SelKp* __restrict__ sel, int nsel,
                  const int8_t* __restrict__ pattern, vslam_kp* kps, uint8_t* desc, int cap, int atan_fma) {
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= nsel) return; /* wave-uniform */
    orient_describe_one(pyr, blur, slot_stride, src, g, sel[k], pattern, kps, desc, cap, atan_fma);
}

/* device-selected keypoints (k_octree + k_assign_out): per-slot lists, counts read from HBM */
__global__ void __launch_bounds__(256)
k_orient_describe_dev(const uint8_t* __restrict__ pyr, const uint8_t* __restrict__ blur, size_t slot_stride,
                      BatchSrc src, PyramidGeom g, const SelKp* __restrict__ sel,
                      const int32_t* __restrict__ slot_counts, const int8_t* __restrict__ pattern, vslam_kp* kps,
                      uint8_t* desc, int cap, int atan_fma, int bps, int nwork) {
    /* XCD-aware order: workgroups b and b+8 share an XCD/L2.  The (slot, keypoint-block) work list is
     * slot-major and level-major inside a slot, so handing XCD k the k-th contiguous eighth keeps one image
     * (or a few of its levels) per L2 instead of streaming every pyramid through all eight. */
    const int per_xcd = (nwork + 7) >> 3;
    const int w = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (w >= nwork) return;
    const int slot = w / bps;
    const int k = (w - slot * bps) * 4 + (threadIdx.x >> 6);
    if (k >= slot_counts[slot * 4]) return; /* wave-uniform */
    orient_describe_one(pyr, blur, slot_stride, src, g, sel[(size_t)slot * cap + k], pattern, kps, desc, cap,
                        atan_fma);
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

/* partial top-2 per (query, train tile): key = dist << 16 | train index (smaller key = better; ties
 * resolve to the lower train index, i.e. first-wins in ascending order like the reference loops). */
__global__ void __launch_bounds__(256)
k_hamming_top2_partial(const uint32_t* __restrict__ q, int nq, const uint32_t* __restrict__ t, int nt,
                       uint32_t* __restrict__ part /* [nq][ntiles][2] */, int ntiles) {
    __shared__ uint4 s_t[HAM_TT * 2];
    __shared__ uint32_t s_k[4][HAM_TQ][2];
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
    uint32_t k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;
    for (int j = wv * 64; j < wv * 64 + 64; j++) {
        const int ti = t0 + j;
        if (ti >= nt) break;
        const uint4 ta = s_t[2 * j], tb = s_t[2 * j + 1];
        const uint32_t d = __popc(qa.x ^ ta.x) + __popc(qa.y ^ ta.y) + __popc(qa.z ^ ta.z) +
                           __popc(qa.w ^ ta.w) + __popc(qb.x ^ tb.x) + __popc(qb.y ^ tb.y) +
                           __popc(qb.z ^ tb.z) + __popc(qb.w ^ tb.w);
        const uint32_t key = (d << 16) | (uint32_t)ti;
        if (key < k1) { k2 = k1; k1 = key; }
        else if (key < k2) k2 = key;
    }
    s_k[wv][lane][0] = k1;
    s_k[wv][lane][1] = k2;
    __syncthreads();
    if (wv == 0 && qi < nq) {
        uint32_t b1 = 0xFFFFFFFFu, b2 = 0xFFFFFFFFu;
#pragma unroll
        for (int w2 = 0; w2 < 4; w2++)
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const uint32_t key = s_k[w2][lane][e];
                if (key < b1) { b2 = b1; b1 = key; }
                else if (key < b2) b2 = key;
            }
        part[((size_t)qi * ntiles + blockIdx.y) * 2] = b1;
        part[((size_t)qi * ntiles + blockIdx.y) * 2 + 1] = b2;
    }
}

__global__ void k_hamming_top2_merge(const uint32_t* __restrict__ part, int nq, int ntiles,
                                     int32_t* __restrict__ idx2, int32_t* __restrict__ dist2) {
    const int qi = blockIdx.x * blockDim.x + threadIdx.x;
    if (qi >= nq) return;
    uint32_t b1 = 0xFFFFFFFFu, b2 = 0xFFFFFFFFu;
    for (int i = 0; i < ntiles * 2; i++) {
        const uint32_t key = part[(size_t)qi * ntiles * 2 + i];
        if (key < b1) { b2 = b1; b1 = key; }
        else if (key < b2) b2 = key;
    }
    idx2[2 * qi] = b1 == 0xFFFFFFFFu ? -1 : (int)(b1 & 0xFFFF);
    dist2[2 * qi] = b1 == 0xFFFFFFFFu ? 0x7FFFFFFF : (int)(b1 >> 16);
    idx2[2 * qi + 1] = b2 == 0xFFFFFFFFu ? -1 : (int)(b2 & 0xFFFF);
    dist2[2 * qi + 1] = b2 == 0xFFFFFFFFu ? 0x7FFFFFFF : (int)(b2 >> 16);
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

void vk_fast_cells(hipStream_t st, const uint8_t* pyr, size_t slot_stride, const BatchSrc& src,
                   const PyramidGeom& g, const CellDesc* cells, int ncells, uint8_t* cand_region,
                   size_t cand_stride, int cand_cap, int iniTh, int minTh, int tile_pitch, int tile_rows,
                   int max_px, int nslots) {
    const size_t shm = (size_t)tile_rows * tile_pitch + (size_t)(tile_rows - 4) * tile_pitch +
                       (size_t)((max_px + 31) / 32) * 4 + 16;
    hipLaunchKernelGGL(k_fast_cells, dim3(ncells, nslots), dim3(256), shm, st, pyr, slot_stride, src, g, cells,
                       cand_region, cand_stride, ncells, cand_cap, iniTh, minTh, tile_pitch, tile_rows);
}

void vk_blur7(hipStream_t st, const uint8_t* pyr, size_t slot_stride, const BatchSrc& src,
              const PyramidGeom& g, uint8_t* blur, const uint32_t* tiles, int ntiles, const int32_t taps[7],
              int nslots) {
    BlurTaps t;
    for (int i = 0; i < 7; i++) t.k[i] = taps[i];
    hipLaunchKernelGGL(k_blur7, dim3(ntiles, nslots), dim3(256), 0, st, pyr, slot_stride, src, g, blur, tiles, t);
}

void vk_orient_describe(hipStream_t st, const uint8_t* pyr, const uint8_t* blur, size_t slot_stride,
                        const BatchSrc& src, const PyramidGeom& g, const SelKp* sel, int nsel,
                        const int8_t* pattern, vslam_kp* kps, uint8_t* desc, int cap, int atan_fma) {
    if (nsel <= 0) return;
    hipLaunchKernelGGL(k_orient_describe, dim3((nsel + 3) / 4), dim3(256), 0, st, pyr, blur, slot_stride, src, g,
                       sel, nsel, pattern, kps, desc, cap, atan_fma);
}

void vk_orient_describe_dev(hipStream_t st, const uint8_t* pyr, const uint8_t* blur, size_t slot_stride,
                            const BatchSrc& src, const PyramidGeom& g, const SelKp* sel,
                            const int32_t* slot_counts, const int8_t* pattern, vslam_kp* kps, uint8_t* desc,
                            int cap, int atan_fma, int nslots) {
    const int bps = (cap + 3) / 4, nwork = bps * nslots;
    hipLaunchKernelGGL(k_orient_describe_dev, dim3(((nwork + 7) / 8) * 8), dim3(256), 0, st, pyr, blur, slot_stride,
                       src, g, sel, slot_counts, pattern, kps, desc, cap, atan_fma, bps, nwork);
}

void vk_hamming_matrix(hipStream_t st, const uint8_t* q, int nq, const uint8_t* t, int nt, uint8_t* out) {
    if (nq <= 0 || nt <= 0) return;
    dim3 grid((nq + HAM_TQ - 1) / HAM_TQ, (nt + HAM_TT - 1) / HAM_TT);
    hipLaunchKernelGGL(k_hamming_matrix, grid, dim3(256), 0, st, (const uint32_t*)q, nq, (const uint32_t*)t, nt,
                       out);
}

int vk_hamming_top2_tiles(int nt) { return (nt + HAM_TT - 1) / HAM_TT; }

void vk_hamming_top2(hipStream_t st, const uint8_t* q, int nq, const uint8_t* t, int nt, uint32_t* part,
                     int32_t* idx2, int32_t* dist2) {
    if (nq <= 0) return;
    const int ntiles = vk_hamming_top2_tiles(nt);
    if (ntiles > 0) {
        dim3 grid((nq + HAM_TQ - 1) / HAM_TQ, ntiles);
        hipLaunchKernelGGL(k_hamming_top2_partial, grid, dim3(256), 0, st, (const uint32_t*)q, nq,
                           (const uint32_t*)t, nt, part, ntiles);
    }
    hipLaunchKernelGGL(k_hamming_top2_merge, dim3((nq + 255) / 256), dim3(256), 0, st, part, nq, ntiles, idx2,
                       dist2);
}
