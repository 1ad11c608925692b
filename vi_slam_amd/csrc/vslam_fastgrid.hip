/* vslam_fastgrid.hip -- the grid FAST detector behind vi_slam::geometry::FAST::detect (include/vslam_fastgrid.h).
 *
 * What the reference runs (all CUDA, warp = 32, one launch per pyramid level and stage):
 *   K5 image_halfsample_gpu_kernel           thirdparty/vilib/visual_lib/src/preprocess/pyramid_gpu.cu:76-96
 *   K1 fast_gpu_calculate_lut_kernel          .../feature_detection/fast/fast_gpu_cuda_tools.cu:142-166   (64 Ki-entry LUT)
 *   K2 fast_gpu_calc_corner_response_kernel   .../fast_gpu_cuda_tools.cu:244-420    (float response image in global memory)
 *   K3 detector_base_gpu_grid_nms_kernel      .../detector_base_gpu_cuda_tools.cu:700-878 (reads the response 9x per pixel)
 * What runs here:
 *   k_fg_halfsample  one thread = 4 output pixels from two 8-byte loads per source row, whole batch per launch
 *   k_fg_detect      ONE launch for every level, cell and image: a workgroup owns a grid cell, walks the levels,
 *                    stages the cell's window (+1 px NMS halo +3 px Bresenham ring) in LDS, computes the response
 *                    of the cell and its halo into LDS (the float response image never exists in HBM), does the
 *                    3x3 suppression and the cell arg-max there and merges the levels in registers.  K1's table is
 *                    replaced by a shift-and-AND run test on the 16-bit mask (same predicate).
 * Which of several equal maxima of a cell is reported is decided in the reference by K3's launch geometry: a thread
 * per column keeps its topmost maximum, a 32-lane __shfl_down_sync tree prefers the lane whose 5-bit index reads
 * smallest when bit-reversed, warps and then levels are merged in ascending order with strict '>'.  The arg-max
 * key below carries exactly that priority (tie_rule 0); tie_rule 1 is plain raster order (rosten::FASTCPU<true>).
 */
#include "../../include/vslam_fastgrid.h"
#include "vslam_ctx.h"
#include "vslam_wave.h"

#define FG_MAX_LEVELS 8
#define FG_MAX_BATCH 64

struct FgLevel {
    int32_t w, h, pitch;
    uint32_t pad;
    size_t base, bytes; /* level-major layout: image s of level l starts at base + s * bytes */
};
struct FgGeom {
    FgLevel lv[FG_MAX_LEVELS];
    int32_t cw, ch, n_cols, n_rows, min_level, max_level, hb, vb, dhb, dvb, arc, score, tie;
    float thr;
};
struct FgPtrs {
    const uint8_t* p[FG_MAX_BATCH];
};

struct vslam_fg {
    vslam_fg_params p;
    FgGeom G;
    int cells = 0;
    size_t pyr_bytes = 0;
    hipStream_t stream = nullptr;
    uint8_t *d_pyr = nullptr, *h_img = nullptr, *d_grid = nullptr, *h_grid = nullptr;
    float* d_resp = nullptr;
    int last_n = 0;
};

/* ---------------------------------------------------------------------------------------------- */
__global__ void __launch_bounds__(256)
k_fg_gather(FgPtrs src, size_t src_pitch, uint8_t* pyr, FgLevel d) { /* device images -> level 0 */
    const int x16 = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), s = blockIdx.z;
    if (y >= d.h || x16 * 16 >= d.w) return;
    const uint8_t* sp = src.p[s] + (size_t)y * src_pitch + (size_t)x16 * 16;
    uint8_t* dp = pyr + d.base + (size_t)s * d.bytes + (size_t)y * d.pitch + (size_t)x16 * 16;
    const int n = min(16, d.w - x16 * 16);
    if (n == 16 && (((uintptr_t)sp) & 15) == 0) {
        *(uint4*)dp = *(const uint4*)sp;
    } else if (n == 16 && (((uintptr_t)sp) & 3) == 0) {
        const uint32_t* s4 = (const uint32_t*)sp;
        *(uint4*)dp = make_uint4(s4[0], s4[1], s4[2], s4[3]);
    } else {
        for (int i = 0; i < n; i++) dp[i] = sp[i];
    }
}

/* K5: (a + b + c + d) >> 2 */
__global__ void __launch_bounds__(256)
k_fg_halfsample(uint8_t* pyr, FgLevel s, FgLevel d) {
    const int q = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), img = blockIdx.z;
    if (y >= d.h || 4 * q >= d.w) return;
    const uint8_t* sp = pyr + s.base + (size_t)img * s.bytes + (size_t)(2 * y) * s.pitch + (size_t)q * 8;
    const uint2 t = *(const uint2*)sp, b = *(const uint2*)(sp + s.pitch); /* pitches are multiples of 64 */
    uint32_t out = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t tw = i < 2 ? t.x : t.y, bw = i < 2 ? b.x : b.y;
        const int sh = (i & 1) * 16;
        const uint32_t v = ((tw >> sh) & 0xFF) + ((tw >> (sh + 8)) & 0xFF) + ((bw >> sh) & 0xFF) + ((bw >> (sh + 8)) & 0xFF);
        out |= (v >> 2) << (8 * i);
    }
    *(uint32_t*)(pyr + d.base + (size_t)img * d.bytes + (size_t)y * d.pitch + (size_t)q * 4) = out;
}

/* a circular run of >= arc ones in the low 16 bits (fast_gpu_is_corner, fast_gpu_cuda_tools.cu:97-114) */
__device__ __forceinline__ bool fg_is_corner(uint32_t m, int arc) {
    if (__popc(m) < arc) return false;
    const uint32_t d = m | (m << 16);
    uint32_t r = d;
    for (int k = 1; k < arc; k++) r &= d >> k;
    return (r & 0xFFFFu) != 0;
}
__device__ __forceinline__ uint32_t fg_sign(float v) { return __float_as_uint(v) >> 31; } /* signbit() */

/* fast_gpu_prechecks (fast_gpu_cuda_tools.cu:116-139): true = cannot be a corner.  Exact for arcs >= 9 (which
 * the constructor asserts): an arc of 9 contains one pixel of every opposite pair. */
__device__ __forceinline__ bool fg_precheck_fails(const uint8_t* p, int wp, float thr) {
    const float c = (float)p[0];
    const float ct = __fadd_rn(c, thr), c_t = __fsub_rn(c, thr);
    float a = (float)p[-3], b = (float)p[3];
    if ((fg_sign(__fsub_rn(a, c_t)) | fg_sign(__fsub_rn(b, c_t)) | fg_sign(__fsub_rn(ct, a)) | fg_sign(__fsub_rn(ct, b))) == 0) return true;
    a = (float)p[3 * wp];
    b = (float)p[-3 * wp];
    return (fg_sign(__fsub_rn(a, c_t)) | fg_sign(__fsub_rn(b, c_t)) | fg_sign(__fsub_rn(ct, a)) | fg_sign(__fsub_rn(ct, b))) == 0;
}

/* K2 for one pixel that passed the prechecks; its window address is p (LDS, pitch wp) */
__device__ float fg_response_px(const uint8_t* p, int wp, float thr, int arc, int score) {
    const float c = (float)p[0];
    const float ct = __fadd_rn(c, thr), c_t = __fsub_rn(c, thr);
    /* ring order of bresenham_circle_offset_pitch (:41-95) */
    const int off[16] = {3 * wp,      3 * wp - 1,  2 * wp - 2,  wp - 3,  -3,     -wp - 3,    -2 * wp - 2, -3 * wp - 1,
                         -3 * wp,     -3 * wp + 1, -2 * wp + 2, -wp + 3, 3,      wp + 3,     2 * wp + 2,  3 * wp + 1};
    float px[16];
    uint32_t dark = 0, bright = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        px[i] = (float)p[off[i]];
        dark |= fg_sign(__fsub_rn(px[i], c_t)) << i;
        bright |= fg_sign(__fsub_rn(ct, px[i])) << i;
    }
    if (!(fg_is_corner(dark, arc) || fg_is_corner(bright, arc))) return 0.0f;
    if (score == VSLAM_FG_SUM_OF_ABS_DIFF_ALL) {
        float r = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; i++) r = __fadd_rn(r, fabsf(__fsub_rn(px[i], c)));
        return r;
    }
    if (score == VSLAM_FG_SUM_OF_ABS_DIFF_ON_ARC) {
        float rb = 0.0f, rd = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const float ad = __fsub_rn(fabsf(__fsub_rn(px[i], c)), thr);
            rd = __fadd_rn(rd, (dark >> i) & 1u ? ad : 0.0f);
            rb = __fadd_rn(rb, (bright >> i) & 1u ? ad : 0.0f);
        }
        return fmaxf(rb, rd);
    }
    float mn = __fadd_rn(thr, 1.0f), mx = 255.0f; /* MAX_THRESHOLD: binary search, :386-415 */
    while (mn <= mx) {
        const float med = floorf(__fmul_rn(__fadd_rn(mn, mx), 0.5f));
        const float mct = __fadd_rn(c, med), mc_t = __fsub_rn(c, med);
        uint32_t dk = 0, br = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            dk |= fg_sign(__fsub_rn(px[i], mc_t)) << i;
            br |= fg_sign(__fsub_rn(mct, px[i])) << i;
        }
        if (fg_is_corner(dk, arc) || fg_is_corner(br, arc)) mn = __fadd_rn(med, 1.0f);
        else mx = __fsub_rn(med, 1.0f);
    }
    return mx;
}

__device__ __forceinline__ uint32_t fg_brev5(uint32_t v) { return __brev(v) >> 27; }

template <int NT>
__global__ void __launch_bounds__(NT)
k_fg_detect(const uint8_t* __restrict__ pyr, FgGeom G, uint8_t* grid, float* resp_out, int resp_level, int resp_slot) {
    extern __shared__ __align__(16) uint8_t fgsm[];
    __shared__ unsigned long long s_best;
    __shared__ int s_nlist;
    const int tid = threadIdx.x, slot = blockIdx.y;
    const int ncell = G.n_cols * G.n_rows;
    const int per_xcd = (ncell + 7) >> 3; /* workgroups b and b+8 share an XCD: neighbouring cells per L2 */
    const int cell = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (cell >= ncell) return;
    const int cy = cell / G.n_cols, cx = cell - cy * G.n_cols;
    float bestS = 0.0f, bestX = 0.0f, bestY = 0.0f;
    int bestL = -1;
    for (int l = G.min_level; l < G.max_level; l++) { /* every condition below is workgroup-uniform */
        const FgLevel lg = G.lv[l];
        const int cwl = G.cw >> l, chl = G.ch >> l;
        if (cwl < 1 || chl < 1) break;
        const int x0 = cwl * cx, y0 = chl * cy;
        if (lg.w < 7 || lg.h < 7 || x0 >= lg.w || y0 >= lg.h) continue;
        const uint8_t* img = pyr + lg.base + (size_t)slot * lg.bytes;
        const int WP = cwl + 8, WH = chl + 8, RP = cwl + 2, RH = chl + 2;
        uint8_t* win = fgsm;
        float* respS = (float*)(fgsm + ((WP * WH + 15) & ~15));
        uint16_t* list = (uint16_t*)(respS + RP * RH);
        const uint32_t mRP = ((1u << 20) + RP - 1) / RP; /* i / RP == (i * mRP) >> 20 for i < 2^20 / RP */
        /* window origin x0 - 4 is a multiple of 4 (cell widths are), rows are 64-byte aligned: aligned dwords.
         * Bytes outside the image are never used by a pixel inside the detection border, so the dword and row
         * indices are merely clamped into the allocation. */
        if (cwl >= 4) {
            const int WD = WP >> 2, maxd = (lg.pitch >> 2) - 1, d0 = (x0 - 4) >> 2; /* arithmetic shift: -1 for x0 = 0 */
            const uint32_t mD = ((1u << 20) + WD - 1) / WD;
            for (int i = tid; i < WD * WH; i += NT) {
                /* 24-bit multiply-adds throughout (i < 2^20, the magic numbers < 2^21, rows and pitches < 2^24): one
                 * instruction where the 32-bit / size_t forms take a multiply plus adds or a 64-bit chain */
                const int wy = (int)(mad24u((uint32_t)i, mD, 0u) >> 20), wd = i - (int)mad24u((uint32_t)wy, (uint32_t)WD, 0u);
                const int gy = min(max(y0 - 4 + wy, 0), lg.h - 1), gd = min(max(d0 + wd, 0), maxd);
                ((uint32_t*)win)[i] = *(const uint32_t*)(img + mad24u((uint32_t)gy, (uint32_t)lg.pitch, 4u * (uint32_t)gd));
            }
        } else { /* 1- and 2-pixel cells of very coarse levels */
            for (int i = tid; i < WP * WH; i += NT) {
                const int wy = i / WP, wx = i - wy * WP;
                const int gx = min(max(x0 - 4 + wx, 0), lg.w - 1), gy = min(max(y0 - 4 + wy, 0), lg.h - 1);
                win[i] = img[(size_t)gy * lg.pitch + gx];
            }
        }
        if (tid == 0) {
            s_best = 0ull;
            s_nlist = 0;
        }
        __syncthreads();
        /* K2 on the cell and its 1-px halo, in two passes: the cheap prechecks on every pixel, the survivors
         * (a quarter of the pixels on textured images) compacted into a list so that the expensive part -- ring
         * gather, masks, arc test, score -- runs on dense lanes */
        for (int i = tid; i < RP * RH; i += NT) {
            const int ry = (int)(mad24u((uint32_t)i, mRP, 0u) >> 20), rx = i - (int)mad24u((uint32_t)ry, (uint32_t)RP, 0u);
            const int gx = x0 - 1 + rx, gy = y0 - 1 + ry;
            respS[i] = 0.0f;
            const bool cand = gx >= G.dhb && gy >= G.dvb && gx < lg.w - G.dhb && gy < lg.h - G.dvb &&
                              !fg_precheck_fails(win + mad24u((uint32_t)(ry + 3), (uint32_t)WP, (uint32_t)(rx + 3)), WP, G.thr);
            const unsigned long long m = __ballot(cand);
            if (m) { /* wave-uniform */
                const int lane = tid & 63;
                int base = 0;
                if (lane == 0) base = atomicAdd(&s_nlist, __popcll(m));
                base = __shfl(base, 0, 64);
                if (cand) list[base + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)i;
            }
        }
        __syncthreads();
        const int nlist = s_nlist;
        for (int k = tid; k < nlist; k += NT) {
            const int i = list[k];
            const int ry = (int)(mad24u((uint32_t)i, mRP, 0u) >> 20), rx = i - (int)mad24u((uint32_t)ry, (uint32_t)RP, 0u);
            respS[i] = fg_response_px(win + mad24u((uint32_t)(ry + 3), (uint32_t)WP, (uint32_t)(rx + 3)), WP, G.thr, G.arc, G.score);
        }
        __syncthreads();
        if (resp_out && l == resp_level && slot == resp_slot)
            for (int i = tid; i < RP * RH; i += NT) {
                const int ry = i / RP, rx = i - ry * RP;
                const int gx = x0 - 1 + rx, gy = y0 - 1 + ry;
                if (rx >= 1 && rx <= cwl && ry >= 1 && ry <= chl && gx < lg.w && gy < lg.h) resp_out[(size_t)gy * lg.w + gx] = respS[i];
            }
        /* K3: 3x3 suppression (strictly_greater) + cell arg-max with the reference's tie order */
        const int bdx = cwl, bdy = max(1, min(128 / cwl, chl)); /* K3's block, detector_base_gpu_cuda_tools.cu:898-903 */
        const int yoff = max(0, G.vb - chl * cy);
        const int cshift = 31 - __clz(cwl); /* cell widths are powers of two */
        for (int i = tid; i < cwl * chl; i += NT) {
            const int py = i >> cshift, px = i & (cwl - 1);
            const int gx = x0 + px, gy = y0 + py;
            if (py < yoff || gx < G.hb || gx >= lg.w - G.hb || gy >= lg.h - G.vb) continue;
            const float* rp = respS + mad24u((uint32_t)(py + 1), (uint32_t)RP, (uint32_t)(px + 1));
            float c = rp[0];
            if (!(c > 0.0f)) continue;
#pragma unroll
            for (int dy = -1; dy <= 1; dy++)
#pragma unroll
                for (int dx = -1; dx <= 1; dx++)
                    if (dx || dy) c = __fmul_rn(c, __fmul_rn(-0.5f, __fadd_rn(-1.0f, copysignf(1.0f, __fsub_rn(rp[dy * RP + dx], c)))));
            if (!(c > 0.0f)) continue;
            uint32_t prio;
            if (G.tie == 0) {
                const int ty = (py - yoff) & (bdy - 1); /* bdy is a power of two */
                const uint32_t t = (uint32_t)(px + bdx * ty);
                prio = ((t >> 5) << 17) | (fg_brev5(t & 31u) << 12) | (uint32_t)py;
            } else {
                prio = ((uint32_t)py << 12) | (uint32_t)px;
            }
            atomicMax(&s_best, ((unsigned long long)__float_as_uint(c) << 32) | (unsigned long long)(0xFFFFFFFFu - prio));
        }
        __syncthreads();
        if (tid == 0 && s_best) {
            const float r = __uint_as_float((uint32_t)(s_best >> 32));
            const uint32_t prio = 0xFFFFFFFFu - (uint32_t)s_best;
            int px, py;
            if (G.tie == 0) {
                const uint32_t t = ((prio >> 17) << 5) | fg_brev5((prio >> 12) & 31u);
                px = (int)(t % (uint32_t)bdx);
                py = (int)(prio & 0xFFFu);
            } else {
                px = (int)(prio & 0xFFFu);
                py = (int)(prio >> 12);
            }
            if (bestS < r) { /* levels in ascending order, strict: the finer level keeps a tie (:871-876) */
                const float scale = (float)(1 << l);
                bestS = r;
                bestX = __fmul_rn((float)(x0 + px), scale);
                bestY = __fmul_rn((float)(y0 + py), scale);
                bestL = l;
            }
        }
        __syncthreads();
    }
    if (tid == 0) { /* DetectorBaseGPU's SoA grid: pos (float2) | score | level */
        uint8_t* g = grid + (size_t)slot * ncell * 16;
        ((float2*)g)[cell] = make_float2(bestX, bestY);
        ((float*)(g + (size_t)ncell * 8))[cell] = bestS;
        ((int32_t*)(g + (size_t)ncell * 12))[cell] = bestL;
    }
}

/* ---------------------------------------------------------------------------------------------- host */
static size_t fg_lds_bytes(const FgGeom& G) {
    size_t m = 0;
    for (int l = G.min_level; l < G.max_level; l++) {
        const int cwl = G.cw >> l, chl = G.ch >> l;
        if (cwl < 1 || chl < 1) break;
        m = std::max(m, (size_t)(((cwl + 8) * (chl + 8) + 15) & ~15) + (size_t)(cwl + 2) * (chl + 2) * 6);
    }
    return m;
}

extern "C" int vslam_fg_create(const vslam_fg_params* p, vslam_fg** out) {
    if (!p || !out) {
        g_err = "invalid arguments";
        return VSLAM_ERR_INVALID;
    }
    *out = nullptr;
    if (p->image_width < 16 || p->image_height < 16 || p->image_width > 16384 || p->image_height > 16384 ||
        (p->cell_size_width != 32 && p->cell_size_width != 64) || (p->cell_size_height != 32 && p->cell_size_height != 64) ||
        p->min_level < 0 || p->max_level <= p->min_level || p->max_level > FG_MAX_LEVELS || p->horizontal_border < 0 ||
        p->vertical_border < 0 || p->min_arc_length < 9 || p->min_arc_length > 12 || p->score < 0 || p->score > 2 ||
        p->tie_rule < 0 || p->tie_rule > 1 || p->max_batch < 1 || p->max_batch > FG_MAX_BATCH || !(p->threshold >= 0.0f)) {
        g_err = "vslam_fg_create: unsupported parameters";
        return VSLAM_ERR_INVALID;
    }
    /* the reference's own assertions: pyramid_pool.cpp:58-59, detector_base_gpu.cpp:62 */
    if ((p->image_width % (1 << (p->max_level - 1))) || (p->image_height % (1 << (p->max_level - 1))) ||
        (p->cell_size_height % (1 << (p->max_level - 1)))) {
        g_err = "vslam_fg_create: image and cell sizes must be divisible by 2^(max_level-1)";
        return VSLAM_ERR_INVALID;
    }
    if ((p->cell_size_height >> (p->max_level - 1)) > 0xFFF) return VSLAM_ERR_INVALID;
    vslam_fg* fg = new vslam_fg();
    fg->p = *p;
    FgGeom& G = fg->G;
    memset(&G, 0, sizeof(G));
    G.cw = p->cell_size_width;
    G.ch = p->cell_size_height;
    G.n_cols = (p->image_width + G.cw - 1) / G.cw; /* detector_base.cpp:54-55 */
    G.n_rows = (p->image_height + G.ch - 1) / G.ch;
    G.min_level = p->min_level;
    G.max_level = p->max_level;
    G.hb = std::max(3, p->horizontal_border); /* fast_gpu.cpp:66-67 */
    G.vb = std::max(3, p->vertical_border);
    G.dhb = std::max(3, p->horizontal_border - 1); /* fast_gpu.cpp:72-73 (DETECTOR_BASE_NMS_SIZE / 2 = 1) */
    G.dvb = std::max(3, p->vertical_border - 1);
    G.arc = p->min_arc_length;
    G.score = p->score;
    G.tie = p->tie_rule;
    G.thr = p->threshold;
    fg->cells = G.n_cols * G.n_rows;
    size_t off = 0;
    for (int l = 0; l < G.max_level; l++) {
        FgLevel& L = G.lv[l];
        L.w = p->image_width >> l; /* pyramid_pool.cpp:61-62 */
        L.h = p->image_height >> l;
        L.pitch = (L.w + 8 + 63) & ~63;
        L.bytes = (size_t)L.pitch * L.h;
        L.base = off;
        off += L.bytes * p->max_batch;
    }
    fg->pyr_bytes = off + 256;
#define FG_TRY(call)                                                      \
    do {                                                                  \
        hipError_t e_ = (call);                                           \
        if (e_ != hipSuccess) {                                           \
            g_err = std::string(#call) + ": " + hipGetErrorString(e_);    \
            vslam_fg_destroy(fg);                                         \
            return VSLAM_ERR_HIP;                                         \
        }                                                                 \
    } while (0)
    FG_TRY(hipSetDevice(p->device));
    FG_TRY(hipStreamCreateWithFlags(&fg->stream, hipStreamNonBlocking));
    FG_TRY(hipMalloc((void**)&fg->d_pyr, fg->pyr_bytes));
    FG_TRY(hipMemset(fg->d_pyr, 0, fg->pyr_bytes));
    FG_TRY((hipError_t)vslam_pinned_alloc((void**)&fg->h_img, G.lv[0].bytes * p->max_batch));
    memset(fg->h_img, 0, G.lv[0].bytes * p->max_batch);
    FG_TRY(hipMalloc((void**)&fg->d_grid, (size_t)fg->cells * 16 * p->max_batch));
    FG_TRY((hipError_t)vslam_pinned_alloc((void**)&fg->h_grid, (size_t)fg->cells * 16 * p->max_batch));
    FG_TRY(hipMalloc((void**)&fg->d_resp, (size_t)p->image_width * p->image_height * 4));
    FG_TRY(hipDeviceSynchronize());
#undef FG_TRY
    *out = fg;
    return VSLAM_OK;
}

extern "C" void vslam_fg_destroy(vslam_fg* fg) {
    if (!fg) return;
    (void)hipSetDevice(fg->p.device);
    if (fg->stream) (void)hipStreamSynchronize(fg->stream);
    if (fg->d_pyr) (void)hipFree(fg->d_pyr);
    if (fg->h_img) (void)hipHostFree(fg->h_img);
    if (fg->d_grid) (void)hipFree(fg->d_grid);
    if (fg->h_grid) (void)hipHostFree(fg->h_grid);
    if (fg->d_resp) (void)hipFree(fg->d_resp);
    if (fg->stream) (void)hipStreamDestroy(fg->stream);
    delete fg;
}

extern "C" int vslam_fg_grid(const vslam_fg* fg, int* n_cols, int* n_rows) {
    if (!fg) return VSLAM_ERR_INVALID;
    if (n_cols) *n_cols = fg->G.n_cols;
    if (n_rows) *n_rows = fg->G.n_rows;
    return VSLAM_OK;
}

static int fg_nt() { /* threads per cell, see vk_fast_cells_v3; VSLAM_FG_NT = 64 | 128 | 256 for A/B runs */
    const int v = vslam_process_tuning().fg_threads; /* process-wide */
    return (v == 64 || v == 128 || v == 256) ? v : 128;
}

static void fg_launch_detect(vslam_fg* fg, int n, float* resp_out, int resp_level, int resp_slot) {
    const FgGeom& G = fg->G;
    const dim3 grid(((fg->cells + 7) / 8) * 8, n);
    const size_t lds = fg_lds_bytes(G);
    switch (fg_nt()) {
        case 64:
            hipLaunchKernelGGL(k_fg_detect<64>, grid, dim3(64), lds, fg->stream, fg->d_pyr, G, fg->d_grid, resp_out, resp_level,
                               resp_slot);
            break;
        case 128:
            hipLaunchKernelGGL(k_fg_detect<128>, grid, dim3(128), lds, fg->stream, fg->d_pyr, G, fg->d_grid, resp_out,
                               resp_level, resp_slot);
            break;
        default:
            hipLaunchKernelGGL(k_fg_detect<256>, grid, dim3(256), lds, fg->stream, fg->d_pyr, G, fg->d_grid, resp_out,
                               resp_level, resp_slot);
    }
}

extern "C" int vslam_fg_detect_batch(vslam_fg* fg, int n, const uint8_t* const* imgs, size_t pitch, int on_device, float* pos,
                                     float* score, int32_t* level) {
    if (!fg || n < 1 || n > fg->p.max_batch || !imgs || !pos || !score || !level || pitch < (size_t)fg->p.image_width) {
        g_err = "invalid arguments";
        return VSLAM_ERR_INVALID;
    }
    for (int s = 0; s < n; s++)
        if (!imgs[s]) {
            g_err = "null image";
            return VSLAM_ERR_INVALID;
        }
    HIPCHK(hipSetDevice(fg->p.device));
    const FgGeom& G = fg->G;
    const FgLevel& L0 = G.lv[0];
    hipStream_t st = fg->stream;
    if (on_device) {
        FgPtrs P;
        memset(&P, 0, sizeof(P));
        for (int s = 0; s < n; s++) P.p[s] = imgs[s];
        hipLaunchKernelGGL(k_fg_gather, dim3(((L0.w + 15) / 16 + 63) / 64, (L0.h + 3) / 4, n), dim3(256), 0, st, P, pitch,
                           fg->d_pyr, L0);
    } else { /* pageable rows -> pinned staging in the device layout -> one copy kernel (see vslam_fe.hip) */
        for (int s = 0; s < n; s++)
            for (int y = 0; y < L0.h; y++) memcpy(fg->h_img + (size_t)s * L0.bytes + (size_t)y * L0.pitch, imgs[s] + (size_t)y * pitch, L0.w);
        CopyRanges R;
        memset(&R, 0, sizeof(R));
        R.dst[0] = fg->d_pyr + L0.base;
        R.src[0] = fg->h_img;
        R.bytes[0] = L0.bytes * n;
        R.n = 1;
        vk_copy_ranges(st, R);
    }
    for (int l = 1; l < G.max_level; l++) {
        const FgLevel& D = G.lv[l];
        hipLaunchKernelGGL(k_fg_halfsample, dim3(((D.w + 3) / 4 + 63) / 64, (D.h + 3) / 4, n), dim3(256), 0, st, fg->d_pyr,
                           G.lv[l - 1], D);
    }
    fg_launch_detect(fg, n, nullptr, -1, -1);
    CopyRanges R;
    memset(&R, 0, sizeof(R));
    R.dst[0] = fg->h_grid;
    R.src[0] = fg->d_grid;
    R.bytes[0] = (size_t)fg->cells * 16 * n;
    R.n = 1;
    vk_copy_ranges(st, R);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    fg->last_n = n;
    const int C = fg->cells;
    for (int s = 0; s < n; s++) {
        const uint8_t* g = fg->h_grid + (size_t)s * C * 16;
        memcpy(pos + (size_t)s * C * 2, g, (size_t)C * 8);
        memcpy(score + (size_t)s * C, g + (size_t)C * 8, (size_t)C * 4);
        memcpy(level + (size_t)s * C, g + (size_t)C * 12, (size_t)C * 4);
    }
    return VSLAM_OK;
}

extern "C" int vslam_fg_detect(vslam_fg* fg, const uint8_t* img_host, size_t pitch, float* pos, float* score, int32_t* level) {
    const uint8_t* one[1] = {img_host};
    return vslam_fg_detect_batch(fg, 1, one, pitch, 0, pos, score, level);
}

extern "C" int vslam_fg_level_copy(vslam_fg* fg, int slot, int level, uint8_t* dst, size_t dst_pitch, int* w, int* h) {
    if (!fg || slot < 0 || slot >= fg->last_n || level < 0 || level >= fg->G.max_level) {
        g_err = "invalid arguments";
        return VSLAM_ERR_INVALID;
    }
    const FgLevel& L = fg->G.lv[level];
    if (w) *w = L.w;
    if (h) *h = L.h;
    if (!dst) return VSLAM_OK;
    if (dst_pitch < (size_t)L.w) return VSLAM_ERR_INVALID;
    HIPCHK(hipSetDevice(fg->p.device));
    HIPCHK(hipMemcpy2DAsync(dst, dst_pitch, fg->d_pyr + L.base + (size_t)slot * L.bytes, L.pitch, L.w, L.h, hipMemcpyDeviceToHost,
                            fg->stream));
    HIPCHK(hipStreamSynchronize(fg->stream));
    return VSLAM_OK;
}

extern "C" int vslam_fg_response_copy(vslam_fg* fg, int slot, int level, float* dst) {
    if (!fg || !dst || slot < 0 || slot >= fg->last_n || level < fg->G.min_level || level >= fg->G.max_level) {
        g_err = "invalid arguments";
        return VSLAM_ERR_INVALID;
    }
    const FgLevel& L = fg->G.lv[level];
    HIPCHK(hipSetDevice(fg->p.device));
    HIPCHK(hipMemsetAsync(fg->d_resp, 0, (size_t)L.w * L.h * 4, fg->stream));
    fg_launch_detect(fg, fg->last_n, fg->d_resp, level, slot);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(dst, fg->d_resp, (size_t)L.w * L.h * 4, hipMemcpyDeviceToHost, fg->stream));
    HIPCHK(hipStreamSynchronize(fg->stream));
    return VSLAM_OK;
}
