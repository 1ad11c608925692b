/* oracle/fastgrid_oracle.cpp -- TEST INFRASTRUCTURE ONLY (see orb_oracle.h).
 *
 * CPU restatement of the detector behind vi_slam::geometry::FAST::detect (src/geometry/fast_cuda.cpp:70-132),
 * i.e. vilib::FASTGPU on a half-sampled pyramid:
 *   K5  image_halfsample_gpu_kernel          thirdparty/vilib/visual_lib/src/preprocess/pyramid_gpu.cu:76-96
 *   K2  fast_gpu_calc_corner_response_kernel  .../feature_detection/fast/fast_gpu_cuda_tools.cu:244-420
 *   K3  detector_base_gpu_grid_nms_kernel     .../feature_detection/detector_base_gpu_cuda_tools.cu:700-878
 *   host: FASTGPU::FASTGPU / detectBase       .../feature_detection/fast/fast_gpu.cpp:52-126
 *         DetectorBaseGPU::processResponse    .../feature_detection/detector_base_gpu.cpp:155-187
 *
 * The reference kernels are CUDA and cannot run here.  What pins this restatement:
 *   - per level, the set of NMS survivors with their scores equals the reference's own CPU detector
 *     (rosten::fastN_detect_nonmax<false|true>, compiled from the reference into oracle/_ref) for integer
 *     thresholds -- tests/test_fastgrid_oracle.py;
 *   - the GPU-subset-of-CPU property the reference itself asserts (test/src/feature_detection/test_fast.cpp:212-245).
 * What is restated from the kernel text alone: which of several equal maxima of a cell wins.  K3 resolves that
 * by its launch geometry (column per thread, 32-lane __shfl_down_sync tree, warps in order, levels in order);
 * grid_nms() below walks exactly those threads, lanes and warps.
 */
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>

namespace orbo {

/* K5: dst(x, y) = (s(2x,2y) + s(2x+1,2y) + s(2x,2y+1) + s(2x+1,2y+1)) >> 2; level sizes are original >> l
 * (pyramid_pool.cpp:61-62) */
void fg_halfsample(const uint8_t* src, int sw, int sh, size_t spitch, uint8_t* dst, size_t dpitch) {
    const int dw = sw >> 1, dh = sh >> 1;
    for (int y = 0; y < dh; y++) {
        const uint8_t* t = src + (size_t)(2 * y) * spitch;
        const uint8_t* b = t + spitch;
        for (int x = 0; x < dw; x++)
            dst[(size_t)y * dpitch + x] = (uint8_t)(((unsigned)t[2 * x] + t[2 * x + 1] + b[2 * x] + b[2 * x + 1]) >> 2);
    }
}

/* fast_gpu_is_corner (fast_gpu_cuda_tools.cu:97-114): a circular run of >= min_arc_length ones in 16 bits */
static bool fg_is_corner(unsigned address, int min_arc_length) {
    int ones = __builtin_popcount(address);
    if (ones < min_arc_length) return false;
    if (ones == 16) return true; /* the CUDA loop ends on lones = 32 >= arc as well */
    unsigned d = address | (address << 16);
    while (ones > 0) {
        d <<= __builtin_clz(d);
        int lones = (~d) ? __builtin_clz(~d) : 32;
        if (lones >= min_arc_length) return true;
        d <<= lones;
        ones -= lones;
        if (!d) break;
    }
    return false;
}

static const int FG_RING[16][2] = {/* bresenham_circle_offset_pitch, :41-95: (dx, dy) of ring index i */
    {0, 3}, {-1, 3}, {-2, 2}, {-3, 1}, {-3, 0}, {-3, -1}, {-2, -2}, {-1, -3},
    {0, -3}, {1, -3}, {2, -2}, {3, -1}, {3, 0}, {3, 1}, {2, 2}, {1, 3}};

static bool fg_corner_quick(const float* px, float c, float thr, int arc) { /* fast_gpu_is_corner_quick, :178-242 */
    const float ct = c + thr, c_t = c - thr;
    unsigned dark = 0, bright = 0;
    for (int i = 0; i < 16; i++) {
        dark += std::signbit(px[i] - c_t) ? (1u << i) : 0;
        bright += std::signbit(ct - px[i]) ? (1u << i) : 0;
    }
    return fg_is_corner(dark, arc) || fg_is_corner(bright, arc);
}

/* K2 for one pixel (the caller has checked the detection border) */
static float fg_response_px(const uint8_t* p, size_t pitch, float threshold, int arc, int score) {
    const float c = (float)p[0];
    const float ct = c + threshold, c_t = c - threshold;
    { /* fast_gpu_prechecks, :116-139 */
        float a = (float)p[-3], b = (float)p[3];
        if ((std::signbit(a - c_t) | std::signbit(b - c_t) | std::signbit(ct - a) | std::signbit(ct - b)) == 0) return 0.0f;
        a = (float)p[3 * (std::ptrdiff_t)pitch];
        b = (float)p[-3 * (std::ptrdiff_t)pitch];
        if ((std::signbit(a - c_t) | std::signbit(b - c_t) | std::signbit(ct - a) | std::signbit(ct - b)) == 0) return 0.0f;
    }
    float px[16];
    unsigned dark = 0, bright = 0;
    for (int i = 0; i < 16; i++) {
        px[i] = (float)p[(std::ptrdiff_t)FG_RING[i][1] * (std::ptrdiff_t)pitch + FG_RING[i][0]];
        dark += std::signbit(px[i] - c_t) ? (1u << i) : 0;
        bright += std::signbit(ct - px[i]) ? (1u << i) : 0;
    }
    if (!(fg_is_corner(dark, arc) || fg_is_corner(bright, arc))) return 0.0f;
    if (score == 0) { /* SUM_OF_ABS_DIFF_ALL */
        float r = 0.0f;
        for (int i = 0; i < 16; i++) r += std::fabs(px[i] - c);
        return r;
    }
    if (score == 1) { /* SUM_OF_ABS_DIFF_ON_ARC: every ring pixel beyond the threshold, not just the arc */
        float rb = 0.0f, rd = 0.0f;
        for (int i = 0; i < 16; i++) {
            const float ad = std::fabs(px[i] - c) - threshold;
            rd += (dark & (1u << i)) ? ad : 0.0f;
            rb += (bright & (1u << i)) ? ad : 0.0f;
        }
        return std::fmax(rb, rd);
    }
    float mn = threshold + 1, mx = 255.0f; /* MAX_THRESHOLD, :386-415 */
    while (mn <= mx) {
        const float med = std::floor((mn + mx) * 0.5f);
        if (fg_corner_quick(px, c, med, arc)) mn = med + 1.0f;
        else mx = med - 1.0f;
    }
    return mx;
}

/* K2 over a level: response = 0 outside [hb, w-hb) x [vb, h-vb) */
void fg_response(const uint8_t* img, int w, int h, size_t pitch, int hb, int vb, float threshold, int arc, int score,
                 float* resp /* w*h */) {
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float r = 0.0f;
            if (x >= hb && y >= vb && x < w - hb && y < h - vb) r = fg_response_px(img + (size_t)y * pitch + x, pitch, threshold, arc, score);
            resp[(size_t)y * w + x] = r;
        }
}

/* K3 for one level.  tie_rule 0: the kernel's own thread geometry; 1: raster order inside the cell (what
 * rosten::FASTCPU<true> + DetectorBase::addFeaturePoint keep, detector_base.cpp:96-110) */
void fg_grid_nms(int level, int min_level, int w, int h, int hb, int vb, int cw, int ch, int n_cols, int n_rows,
                 const float* resp, float* pos /* 2*cells */, float* score, int* lvl, int tie_rule) {
    const int cwl = cw >> level, chl = ch >> level;
    if (cwl < 1 || chl < 1) return;
    const int bdx = cwl, bdy = std::max(1, std::min(128 / cwl, chl)); /* detector_base_gpu_cuda_tools.cu:898-903 */
    const int nthreads = bdx * bdy, warp_cnt = (nthreads + 31) >> 5;
    const float scale = (float)(1 << level);
    auto nms = [&](int x, int gy) -> float { /* strictly_greater = true: survives iff > all 8 neighbours */
        float c = resp[(size_t)gy * w + x];
        for (int dy = -1; dy <= 1; dy++)
            for (int dx = -1; dx <= 1; dx++) {
                if (!dx && !dy) continue;
                const int xx = x + dx, yy = gy + dy;
                const float nb = (xx >= 0 && yy >= 0 && xx < w && yy < h) ? resp[(size_t)yy * w + xx] : 0.0f;
                c *= -0.5f * (-1.0f + std::copysign(1.0f, nb - c));
            }
        return c;
    };
    struct Cand { float r, x, y; };
    std::vector<Cand> th(warp_cnt * 32);
    for (int by = 0; by < n_rows; by++)
        for (int bx = 0; bx < n_cols; bx++) {
            const int cell = n_cols * by + bx;
            if (level == min_level) score[cell] = 0.0f;
            float best_r = 0.0f, best_x = 0.0f, best_y = 0.0f;
            if (tie_rule == 1) {
                for (int gy = std::max(vb, chl * by); gy < std::min(h - vb, chl * (by + 1)); gy++)
                    for (int x = std::max(hb, cwl * bx); x < std::min(w - hb, cwl * (bx + 1)); x++) {
                        const float c = nms(x, gy);
                        if (c > best_r) { best_r = c; best_x = (float)x; best_y = (float)gy; }
                    }
            } else {
                for (auto& t : th) t = Cand{0.0f, 0.0f, 0.0f}; /* lanes beyond the block contribute nothing */
                for (int ty = 0; ty < bdy; ty++)
                    for (int tx = 0; tx < bdx; tx++) {
                        const int x = cwl * bx + tx, y = chl * by + ty;
                        Cand c{0.0f, (float)x, 0.0f};
                        if (x < w && y < h && x >= hb && x < w - hb) {
                            const int ctb = vb - chl * by;
                            const int yoff = ctb > 0 ? ctb : 0;
                            int gy = y + yoff, box_line = ty + yoff, max_y = 0;
                            for (; box_line < chl && gy < h - vb; box_line += bdy, gy += bdy) {
                                const float v = nms(x, gy);
                                if (v > c.r) { c.r = v; max_y = gy; }
                            }
                            c.y = (float)max_y;
                        }
                        th[tx + bdx * ty] = c;
                    }
                for (int wi = 0; wi < warp_cnt; wi++) { /* __shfl_down_sync tree, offsets 16..1 */
                    Cand* L = &th[wi * 32];
                    for (int off = 16; off > 0; off >>= 1) {
                        Cand nxt[32];
                        for (int l = 0; l < 32; l++) {
                            nxt[l] = L[l];
                            if (l + off < 32 && L[l + off].r > L[l].r) nxt[l] = L[l + off];
                        }
                        memcpy(L, nxt, sizeof(nxt));
                    }
                }
                best_r = th[0].r; best_x = th[0].x; best_y = th[0].y;
                for (int wi = 1; wi < warp_cnt; wi++)
                    if (th[wi * 32].r > best_r) { best_r = th[wi * 32].r; best_x = th[wi * 32].x; best_y = th[wi * 32].y; }
            }
            if (score[cell] < best_r) {
                score[cell] = best_r;
                pos[2 * cell] = best_x * scale;
                pos[2 * cell + 1] = best_y * scale;
                lvl[cell] = level;
            }
        }
}

/* FASTGPU::detect on one image: pyramid (levels 0 .. max_level-1), K2 + K3 for min_level <= l < max_level.
 * pos/score/level: n_cols*n_rows cells in DetectorBaseGPU's grid layout; cells without a corner: score 0,
 * pos (0,0), level -1 (the reference leaves stale values there and never reads them). */
void fg_detect(const uint8_t* img, int w, int h, size_t pitch, int cw, int ch, int min_level, int max_level, int hborder,
               int vborder, float threshold, int arc, int score_kind, int tie_rule, float* pos, float* score, int* lvl) {
    const int n_cols = (w + cw - 1) / cw, n_rows = (h + ch - 1) / ch;
    const int hb = std::max(3, hborder), vb = std::max(3, vborder);         /* fast_gpu.cpp:66-67 */
    const int dhb = std::max(3, hborder - 1), dvb = std::max(3, vborder - 1); /* :72-73 */
    for (int i = 0; i < n_cols * n_rows; i++) { pos[2 * i] = pos[2 * i + 1] = 0.0f; score[i] = 0.0f; lvl[i] = -1; }
    std::vector<uint8_t> lv((size_t)w * h);
    for (int y = 0; y < h; y++) memcpy(&lv[(size_t)y * w], img + (size_t)y * pitch, w);
    int lw = w, lh = h;
    for (int l = 0; l < max_level; l++) {
        if (l > 0) {
            std::vector<uint8_t> d((size_t)(lw >> 1) * (lh >> 1));
            fg_halfsample(lv.data(), lw, lh, lw, d.data(), lw >> 1);
            lv.swap(d);
            lw >>= 1;
            lh >>= 1;
        }
        if (l < min_level || lw < 7 || lh < 7) continue;
        std::vector<float> resp((size_t)lw * lh);
        fg_response(lv.data(), lw, lh, lw, dhb, dvb, threshold, arc, score_kind, resp.data());
        fg_grid_nms(l, min_level, lw, lh, hb, vb, cw, ch, n_cols, n_rows, resp.data(), pos, score, lvl, tie_rule);
    }
}

} // namespace orbo

extern "C" {
void orbo_fg_halfsample(const uint8_t* src, int sw, int sh, size_t spitch, uint8_t* dst, size_t dpitch) {
    orbo::fg_halfsample(src, sw, sh, spitch, dst, dpitch);
}
void orbo_fg_response(const uint8_t* img, int w, int h, size_t pitch, int hb, int vb, float threshold, int arc, int score,
                      float* resp) {
    orbo::fg_response(img, w, h, pitch, hb, vb, threshold, arc, score, resp);
}
void orbo_fg_detect(const uint8_t* img, int w, int h, size_t pitch, int cw, int ch, int min_level, int max_level, int hborder,
                    int vborder, float threshold, int arc, int score_kind, int tie_rule, float* pos, float* score, int* lvl) {
    orbo::fg_detect(img, w, h, pitch, cw, ch, min_level, max_level, hborder, vborder, threshold, arc, score_kind, tie_rule, pos,
                    score, lvl);
}
}
