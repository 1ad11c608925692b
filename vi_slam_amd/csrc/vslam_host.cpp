/* vslam_host.cpp -- see vslam_host.h.  Reference lines are cited per function. */
#include "vslam_host.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace vslam {

/* cvRound / cvFloor / cvCeil as OpenCV 4.2 defines them on x86-64 (fast_math.hpp): SSE round-to-nearest-even */
static inline int cv_round(float v) { return (int)lrintf(v); }
static inline int cv_round(double v) { return (int)lrint(v); }
static inline int cv_floor(float v) {
    int i = (int)v;
    return i - (i > v);
}
static inline int cv_ceil(float v) {
    int i = (int)v;
    return i + (i < v);
}
static inline int16_t sat_short(float v) {
    int iv = cv_round(v);
    return (int16_t)std::min(std::max(iv, (int)SHRT_MIN), (int)SHRT_MAX);
}

void build_tables(int nfeatures, float scaleFactorF, int nlevels, ExtractorTables& t) {
    /* fextractor.cpp:401-461; scaleFactor is a double member initialised from the float argument */
    const double scaleFactor = scaleFactorF;
    t.nlevels = nlevels;
    t.scale.assign(nlevels, 1.0f);
    t.sigma2.assign(nlevels, 1.0f);
    for (int i = 1; i < nlevels; i++) {
        t.scale[i] = (float)(t.scale[i - 1] * scaleFactor);
        t.sigma2[i] = t.scale[i] * t.scale[i];
    }
    t.inv_scale.resize(nlevels);
    t.inv_sigma2.resize(nlevels);
    for (int i = 0; i < nlevels; i++) {
        t.inv_scale[i] = 1.0f / t.scale[i];
        t.inv_sigma2[i] = 1.0f / t.sigma2[i];
    }
    t.quota.assign(nlevels, 0);
    const float factor = (float)(1.0f / scaleFactor);
    float want = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int level = 0; level < nlevels - 1; level++) {
        t.quota[level] = cv_round(want);
        sum += t.quota[level];
        want *= factor;
    }
    t.quota[nlevels - 1] = std::max(nfeatures - sum, 0);

    /* umax (fextractor.cpp:445-460) */
    const int HP = 15;
    int v, v0;
    const int vmax = cv_floor(HP * sqrtf(2.f) / 2 + 1);
    const int vmin = cv_ceil(HP * sqrtf(2.f) / 2);
    const double hp2 = HP * HP;
    for (v = 0; v <= vmax; ++v) t.umax[v] = cv_round(sqrt(hp2 - v * v));
    for (v = HP, v0 = 0; v >= vmin; --v) {
        while (t.umax[v0] == t.umax[v0 + 1]) ++v0;
        t.umax[v] = v0;
        ++v0;
    }
    t.disc_u.clear();
    t.disc_v.clear();
    for (v = -HP; v <= HP; v++) {
        const int d = t.umax[std::abs(v)];
        for (int u = -d; u <= d; u++) {
            t.disc_u.push_back((int8_t)u);
            t.disc_v.push_back((int8_t)v);
        }
    }
}

void level_size(const ExtractorTables& t, int w, int h, int level, int* lw, int* lh) {
    const float scale = t.inv_scale[level];
    *lw = cv_round((float)w * scale);
    *lh = cv_round((float)h * scale);
}

void build_resize_tables(int sw, int sh, int dw, int dh, ResizeTables& r) {
    /* OpenCV 4.2 resize.cpp, INTER_LINEAR, depth 8U: coefficients in 11-bit fixed point */
    const int ONE = 2048;
    const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
    r.xtab.resize(2 * dw);
    r.xa.resize(2 * dw);
    r.ytab.resize(2 * dh);
    r.yb.resize(2 * dh);
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; } /* dx >= xmax: D = S[sx] * ONE */
        r.xtab[2 * dx] = (uint16_t)sx;
        r.xtab[2 * dx + 1] = (uint16_t)std::min(sx + 1, sw - 1);
        r.xa[2 * dx] = sat_short((1.f - fx) * ONE);
        r.xa[2 * dx + 1] = sat_short(fx * ONE);
    }
    auto clip = [](int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; };
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        r.ytab[2 * dy] = (uint16_t)clip(sy, 0, sh);
        r.ytab[2 * dy + 1] = (uint16_t)clip(sy + 1, 0, sh);
        r.yb[2 * dy] = sat_short((1.f - fy) * ONE);
        r.yb[2 * dy + 1] = sat_short(fy * ONE);
    }
}

bool build_resize_quads(const ResizeTables& r, int sw, int dw, std::vector<uint16_t>& qbase,
                        std::vector<uint32_t>& quads) {
    if (sw < 8) return false;
    const int nq = (dw + 3) / 4;
    qbase.assign(nq, 0);
    quads.assign((size_t)nq * 8, 0);
    for (int q = 0; q < nq; q++) {
        int lo = 1 << 30, hi = -1;
        for (int j = 0; j < 4; j++) {
            const int dx = std::min(4 * q + j, dw - 1); /* outputs past the row end repeat the last column */
            lo = std::min(lo, (int)r.xtab[2 * dx]);
            hi = std::max(hi, (int)r.xtab[2 * dx + 1]);
        }
        const int base = std::min(lo, sw - 8); /* the 8-byte window must stay inside the source row */
        if (hi - base > 7) return false;
        qbase[q] = (uint16_t)base;
        for (int j = 0; j < 4; j++) {
            const int dx = std::min(4 * q + j, dw - 1);
            const uint32_t o0 = r.xtab[2 * dx] - base, o1 = r.xtab[2 * dx + 1] - base;
            quads[(size_t)q * 8 + j] = o0 | (0x0cu << 8) | (o1 << 16) | (0x0cu << 24);
            quads[(size_t)q * 8 + 4 + j] = (uint32_t)(uint16_t)r.xa[2 * dx] | ((uint32_t)(uint16_t)r.xa[2 * dx + 1] << 16);
        }
    }
    return true;
}


/* ------------------------------------------------------------------ fused pyramid plan (k_pyramid_group) */
static inline int div_floor4(int x) { return x >> 2; }

static bool plan_with(const std::vector<const PyrLevelTables*>& tabs, int l0, int ntx, int nty, size_t max_lds,
                      PyrGroupPlan& plan) {
    const int nl = (int)tabs.size();
    plan.l0 = l0;
    plan.nl = nl;
    plan.ntx = ntx;
    plan.nty = nty;
    plan.tiles.assign((size_t)ntx * nty * (nl + 1), PyrTileLevel{});
    size_t lds_max = 0;
    for (int ty = 0; ty < nty; ty++)
        for (int tx = 0; tx < ntx; tx++) {
            PyrTileLevel* T = &plan.tiles[((size_t)ty * ntx + tx) * (nl + 1)];
            /* compute ranges, deepest level first: C_j = hull(own_j, need of level j+1) */
            int cq0[VSLAM_MAX_LEVELS + 1], cq1[VSLAM_MAX_LEVELS + 1], cr0[VSLAM_MAX_LEVELS + 1], cr1[VSLAM_MAX_LEVELS + 1];
            int nq0 = -1, nq1 = -1, nr0 = -1, nr1 = -1; /* need on the current level from the deeper one (empty: -1) */
            for (int j = nl; j >= 0; j--) {
                int oq0 = 0, oq1 = 0, or0 = 0, or1 = 0;
                if (j >= 1) {
                    const PyrLevelTables& t = *tabs[j - 1];
                    const int nq = (t.dw + 3) / 4;
                    oq0 = (int)((long long)tx * nq / ntx);
                    oq1 = (int)((long long)(tx + 1) * nq / ntx);
                    or0 = (int)((long long)ty * t.dh / nty);
                    or1 = (int)((long long)(ty + 1) * t.dh / nty);
                    T[j].sq0 = (int16_t)oq0; T[j].sq1 = (int16_t)oq1; T[j].sr0 = (int16_t)or0; T[j].sr1 = (int16_t)or1;
                }
                const bool own = j >= 1 && oq1 > oq0 && or1 > or0, need = nq1 > nq0 && nr1 > nr0;
                if (!own && !need) { cq0[j] = cq1[j] = cr0[j] = cr1[j] = 0; }
                else if (own && need) {
                    cq0[j] = std::min(oq0, nq0); cq1[j] = std::max(oq1, nq1);
                    cr0[j] = std::min(or0, nr0); cr1[j] = std::max(or1, nr1);
                } else if (own) { cq0[j] = oq0; cq1[j] = oq1; cr0[j] = or0; cr1[j] = or1; }
                else { cq0[j] = nq0; cq1[j] = nq1; cr0[j] = nr0; cr1[j] = nr1; }
                nq0 = nq1 = nr0 = nr1 = -1;
                if (j >= 1 && cq1[j] > cq0[j]) { /* what computing C_j reads of level j-1 */
                    const PyrLevelTables& t = *tabs[j - 1];
                    const int dxl = std::min(4 * cq1[j], t.dw) - 1;
                    const int lo = t.qbase[cq0[j]];               /* windows start here (<= the first tap) */
                    const int hi = t.r.xtab[2 * dxl + 1];          /* last tap column */
                    nq0 = div_floor4(lo);
                    nq1 = div_floor4(hi) + 1;
                    nr0 = t.r.ytab[2 * cr0[j]];
                    nr1 = t.r.ytab[2 * (cr1[j] - 1) + 1] + 1;
                }
            }
            size_t off = 0;
            for (int j = 0; j <= nl; j++) {
                T[j].c0 = (int16_t)(4 * cq0[j]);
                T[j].nc = (int16_t)(4 * (cq1[j] - cq0[j]));
                T[j].r0 = (int16_t)cr0[j];
                T[j].nr = (int16_t)(cr1[j] - cr0[j]);
                T[j].pitch = (uint32_t)((T[j].nc + 8 + 15) & ~15);
                T[j].lds_off = (uint32_t)off;
                off += (size_t)T[j].pitch * T[j].nr;
                if (j >= 1 && T[j].nc / 4 > 64) return false; /* lane = quad */
                if (j == 0 && T[j].pitch / 16 > 64) return false; /* staging: lane = 16-byte chunk of a row */
            }
            for (int j = 1; j <= nl; j++) {
                T[j].rt_off = (uint32_t)off;
                off += ((size_t)T[j].nr * 8 + 15) & ~(size_t)15;
            }
            lds_max = std::max(lds_max, off + 16);
        }
    plan.lds_bytes = lds_max;
    return lds_max <= max_lds;
}

bool build_pyramid_group(const std::vector<const PyrLevelTables*>& tabs, int l0, size_t max_lds, PyrGroupPlan& plan, int rows_override) {
    if (tabs.empty() || tabs.size() > VSLAM_MAX_LEVELS) return false;
    for (const PyrLevelTables* t : tabs)
        if (!t || t->qbase.empty()) return false; /* a level without the quad table: per-level launches */
    const int nq1 = (tabs[0]->dw + 3) / 4, h1 = tabs[0]->dh;
    /* output rows of the first computed level per tile (vslam_tuning.pyr_rows for A/B runs).  Taller tiles recompute less
     * halo and amortise the per-tile prologue over more rows: 28 -> 48 rows is +3 % mono, +2 % stereo, +3 % at 1080p in the
     * pipeline (36 / 40 / 48 / 56 / 64 measured; the LDS limit of the plan, 64 KB, is the brake at 64) */
    int rows = 48;
    if (rows_override >= 0) rows = std::min(64, std::max(4, rows_override));
    const int nty = std::max(1, (h1 + rows - 1) / rows);
#ifndef VSLAM_PYR_TILE_QUADS
#define VSLAM_PYR_TILE_QUADS 52 /* widest tile tried first: quads of the first computed level (a wave's lanes, halo included) */
#endif
    for (int ntx = std::max(1, (nq1 + VSLAM_PYR_TILE_QUADS - 1) / VSLAM_PYR_TILE_QUADS); ntx <= std::max(1, nq1 / 8); ntx++)
        if (plan_with(tabs, l0, ntx, nty, max_lds, plan)) return true;
    return false;
}

int emulate_pyramid_group(const PyrGroupPlan& plan, const std::vector<const PyrLevelTables*>& tabs,
                          const uint8_t* src, int sstride, int readable_w, std::vector<uint8_t*>& dst,
                          const std::vector<int>& dstride) {
    const int nl = plan.nl;
    std::vector<uint8_t> lds(plan.lds_bytes), ok(plan.lds_bytes);
    for (int tile = 0; tile < plan.ntx * plan.nty; tile++) {
        const PyrTileLevel* T = &plan.tiles[(size_t)tile * (nl + 1)];
        std::fill(lds.begin(), lds.end(), (uint8_t)0xCD);
        std::fill(ok.begin(), ok.end(), (uint8_t)0);
        if (T[0].nr <= 0 || T[0].nc <= 0) continue;
        /* stage the source tile: dwords of columns [c0, c0 + pitch), those inside the readable row width */
        for (int r = 0; r < T[0].nr; r++)
            for (int c = 0; c < (int)T[0].pitch; c++) {
                const int col = T[0].c0 + c;
                const size_t a = T[0].lds_off + (size_t)r * T[0].pitch + c;
                if (a >= lds.size()) return -1;
                if (col < readable_w) {
                    lds[a] = src[(size_t)(T[0].r0 + r) * sstride + col];
                    ok[a] = 1;
                }
            }
        for (int j = 1; j <= nl; j++) {
            const PyrLevelTables& t = *tabs[j - 1];
            const PyrTileLevel &S = T[j - 1], &D = T[j];
            for (int r = D.r0; r < D.r0 + D.nr; r++) {
                const int sy0 = t.r.ytab[2 * r], sy1 = t.r.ytab[2 * r + 1], b0 = t.r.yb[2 * r], b1 = t.r.yb[2 * r + 1];
                if (sy0 < S.r0 || sy1 >= S.r0 + S.nr) return -2;
                for (int lane = 0; lane < D.nc / 4; lane++) {
                    const int q = D.c0 / 4 + lane;
                    const int loc = (int)t.qbase[q] - S.c0;
                    if (loc < 0) return -3;
                    const int dwb = loc & ~3, sh = loc & 3;
                    if (dwb + 12 > (int)S.pitch) return -4;
                    uint32_t out = 0;
                    for (int k = 0; k < 4; k++) {
                        const uint32_t sel = t.quads[(size_t)q * 8 + k], cf = t.quads[(size_t)q * 8 + 4 + k];
                        const int o0 = sel & 0xFF, o1 = (sel >> 16) & 0xFF;
                        const int a0 = (int16_t)(cf & 0xFFFF), a1 = (int16_t)(cf >> 16);
                        int h[2];
                        for (int rr = 0; rr < 2; rr++) {
                            const size_t base = S.lds_off + (size_t)((rr ? sy1 : sy0) - S.r0) * S.pitch + dwb + sh;
                            if (base + 8 > lds.size()) return -5;
                            if (!ok[base + o0] || !ok[base + o1]) return -6; /* a tap that was never staged/computed */
                            h[rr] = lds[base + o0] * a0 + lds[base + o1] * a1;
                        }
                        const int v = (((b0 * (h[0] >> 4)) >> 16) + ((b1 * (h[1] >> 4)) >> 16) + 2) >> 2;
                        out |= (uint32_t)(v & 0xFF) << (8 * k);
                    }
                    const size_t da = D.lds_off + (size_t)(r - D.r0) * D.pitch + 4 * lane;
                    if (da + 4 > lds.size()) return -7;
                    for (int k = 0; k < 4; k++) {
                        lds[da + k] = (uint8_t)(out >> (8 * k));
                        ok[da + k] = 1;
                    }
                    if (q >= D.sq0 && q < D.sq1 && r >= D.sr0 && r < D.sr1)
                        for (int k = 0; k < 4; k++)
                            if (4 * q + k < dstride[j]) dst[j][(size_t)r * dstride[j] + 4 * q + k] = (uint8_t)(out >> (8 * k));
                }
            }
        }
    }
    return 0;
}

void build_cells(int level, int lw, int lh, std::vector<HostCell>& out) {
    /* fextractor.cpp:764-797 */
    const float W = 30;
    const int minBorderX = VSLAM_FAST_BORDER, minBorderY = VSLAM_FAST_BORDER;
    const int maxBorderX = lw - VSLAM_FAST_BORDER, maxBorderY = lh - VSLAM_FAST_BORDER;
    const float width = (float)(maxBorderX - minBorderX);
    const float height = (float)(maxBorderY - minBorderY);
    const int nCols = (int)(width / W);
    const int nRows = (int)(height / W);
    if (nCols < 1 || nRows < 1) return;
    const int wCell = (int)std::ceil(width / nCols);
    const int hCell = (int)std::ceil(height / nRows);
    for (int i = 0; i < nRows; i++) {
        const float iniY = (float)(minBorderY + i * hCell);
        float maxY = iniY + hCell + 6;
        if (iniY >= maxBorderY - 3) continue;
        if (maxY > maxBorderY) maxY = (float)maxBorderY;
        for (int j = 0; j < nCols; j++) {
            const float iniX = (float)(minBorderX + j * wCell);
            float maxX = iniX + wCell + 6;
            if (iniX >= maxBorderX - 6) continue;
            if (maxX > maxBorderX) maxX = (float)maxBorderX;
            HostCell c;
            c.level = (uint16_t)level;
            c.x0 = (uint16_t)iniX;
            c.y0 = (uint16_t)iniY;
            c.x1 = (uint16_t)maxX;
            c.y1 = (uint16_t)maxY;
            if (c.x1 - c.x0 < 7 || c.y1 - c.y0 < 7) continue; /* cv::FAST finds nothing in < 7 px */
            out.push_back(c);
        }
    }
}

void build_bands(const std::vector<HostCell>& cells, int max_cells, int max_width, std::vector<HostBand>& out) {
    const size_t n = cells.size();
    size_t i = 0;
    while (i < n) {
        /* the cells of one cell row: same level, same window rows */
        size_t e = i + 1;
        while (e < n && cells[e].level == cells[i].level && cells[e].y0 == cells[i].y0 && cells[e].y1 == cells[i].y1 &&
               cells[e].x0 > cells[e - 1].x0)
            e++;
        const int nrow = (int)(e - i);
        /* pitch of the cells in x; a row of one cell is its own pitch */
        const int wcell = nrow > 1 ? cells[i + 1].x0 - cells[i].x0 : std::max(1, cells[i].x1 - cells[i].x0 - 6);
        const int per = std::max(1, std::min(max_cells, max_width / std::max(wcell, 1)));
        for (size_t b = i; b < e; b += (size_t)per) {
            const size_t last = std::min(e, b + (size_t)per) - 1;
            HostBand hb;
            hb.cell0 = (uint32_t)b;
            hb.level = cells[b].level;
            hb.ncell = (uint16_t)(last - b + 1);
            hb.wcell = (uint16_t)wcell;
            hb.x0 = cells[b].x0;
            hb.y0 = cells[b].y0;
            hb.ww = (uint16_t)(cells[last].x1 - cells[b].x0);
            hb.wh = (uint16_t)(cells[b].y1 - cells[b].y0);
            out.push_back(hb);
        }
        i = e;
    }
}

/* ------------------------------------------------------------------ quadtree distribution */
namespace {
struct QNode {
    int x0, y0, x1, y1; /* UL.x, UL.y, UR.x(=BR.x), BR.y(=BL.y) */
    int begin, count;   /* range in the permutation array (vKeys, order preserved) */
    int prev, next;     /* std::list links */
    bool noMore;
};

struct QTree {
    std::vector<QNode> pool; /* index == creation order (the tie-break key) */
    std::vector<int> perm, tmp;
    const Cand* c;
    int head = -1, tail = -1, size = 0;

    int push_front(const QNode& n) {
        int id = (int)pool.size();
        pool.push_back(n);
        pool[id].prev = -1;
        pool[id].next = head;
        if (head >= 0) pool[head].prev = id;
        head = id;
        if (tail < 0) tail = id;
        size++;
        return id;
    }
    int push_back(const QNode& n) {
        int id = (int)pool.size();
        pool.push_back(n);
        pool[id].next = -1;
        pool[id].prev = tail;
        if (tail >= 0) pool[tail].next = id;
        tail = id;
        if (head < 0) head = id;
        size++;
        return id;
    }
    void erase(int id) {
        const int p = pool[id].prev, nx = pool[id].next;
        if (p >= 0) pool[p].next = nx; else head = nx;
        if (nx >= 0) pool[nx].prev = p; else tail = p;
        size--;
    }
    /* ExtractorNode::DivideNode (fextractor.cpp:472-528): children in order n1..n4, each keeps the
     * parent's key order; returns child ids (or -1 if empty) after push_front in that order. */
    void divide(int id, int child[4]) {
        const QNode P = pool[id];
        const int halfX = (P.x1 - P.x0 + 1) >> 1; /* ceil((float)d/2), d >= 0 */
        const int halfY = (P.y1 - P.y0 + 1) >> 1;
        const int mx = P.x0 + halfX, my = P.y0 + halfY;
        int cnt[4] = {0, 0, 0, 0};
        for (int i = P.begin; i < P.begin + P.count; i++) {
            const Cand& k = c[perm[i]];
            const int q = (k.x < mx) ? ((k.y < my) ? 0 : 2) : ((k.y < my) ? 1 : 3);
            cnt[q]++;
        }
        int off[4] = {P.begin, P.begin + cnt[0], P.begin + cnt[0] + cnt[1], P.begin + cnt[0] + cnt[1] + cnt[2]};
        int w[4] = {off[0], off[1], off[2], off[3]};
        for (int i = P.begin; i < P.begin + P.count; i++) {
            const Cand& k = c[perm[i]];
            const int q = (k.x < mx) ? ((k.y < my) ? 0 : 2) : ((k.y < my) ? 1 : 3);
            tmp[w[q]++] = perm[i];
        }
        memcpy(&perm[P.begin], &tmp[P.begin], sizeof(int) * P.count);
        const int bx[4][4] = {{P.x0, P.y0, mx, my}, {mx, P.y0, P.x1, my}, {P.x0, my, mx, P.y1}, {mx, my, P.x1, P.y1}};
        for (int q = 0; q < 4; q++) {
            child[q] = -1;
            if (cnt[q] == 0) continue;
            QNode n;
            n.x0 = bx[q][0]; n.y0 = bx[q][1]; n.x1 = bx[q][2]; n.y1 = bx[q][3];
            n.begin = off[q];
            n.count = cnt[q];
            n.noMore = cnt[q] == 1;
            n.prev = n.next = -1;
            child[q] = push_front(n);
        }
    }
};
} // namespace

bool distribute_octree(const Cand* cands, int n, int W, int H, int N, std::vector<Cand>& out) {
    out.clear();
    const int nIni = (int)std::round((float)W / (float)H);
    if (nIni < 1) return false;
    if (n == 0) return true;
    const float hX = (float)W / nIni;
    QTree T;
    T.c = cands;
    T.perm.resize(n);
    T.tmp.resize(n);
    T.pool.reserve(4 * (size_t)std::max(N, 16) + 64);

    /* initial nodes and stable bucketing of the candidates (fextractor.cpp:543-561) */
    std::vector<int> bucket(n), cnt(nIni + 1, 0);
    for (int i = 0; i < n; i++) {
        int b = (int)((float)cands[i].x / hX);
        if (b >= nIni) b = nIni - 1;
        bucket[i] = b;
        cnt[b + 1]++;
    }
    for (int b = 0; b < nIni; b++) cnt[b + 1] += cnt[b];
    {
        std::vector<int> w(cnt.begin(), cnt.end() - 1);
        for (int i = 0; i < n; i++) T.perm[w[bucket[i]]++] = i;
    }
    for (int i = 0; i < nIni; i++) {
        QNode ni;
        ni.x0 = (int)(hX * (float)i);
        ni.x1 = (int)(hX * (float)(i + 1));
        ni.y0 = 0;
        ni.y1 = H;
        ni.begin = cnt[i];
        ni.count = cnt[i + 1] - cnt[i];
        ni.noMore = ni.count == 1;
        ni.prev = ni.next = -1;
        if (ni.count == 0) continue; /* erased at fextractor.cpp:572-573 */
        T.push_back(ni);
    }

    typedef std::pair<int, int> SP; /* (size, node id); id order == creation order */
    std::vector<SP> vSize, vPrev;
    bool bFinish = false;
    while (!bFinish) {
        int prevSize = T.size;
        int nToExpand = 0;
        vSize.clear();
        for (int cur = T.head; cur >= 0;) {
            const int next = T.pool[cur].next;
            if (!T.pool[cur].noMore) {
                int ch[4];
                T.divide(cur, ch);
                for (int q = 0; q < 4; q++)
                    if (ch[q] >= 0 && T.pool[ch[q]].count > 1) {
                        nToExpand++;
                        vSize.push_back(SP(T.pool[ch[q]].count, ch[q]));
                    }
                T.erase(cur);
            }
            cur = next;
        }
        if (T.size >= N || T.size == prevSize) {
            bFinish = true;
        } else if (T.size + nToExpand * 3 > N) {
            while (!bFinish) {
                prevSize = T.size;
                vPrev = vSize;
                vSize.clear();
                std::sort(vPrev.begin(), vPrev.end()); /* ascending (size, creation id) */
                for (int j = (int)vPrev.size() - 1; j >= 0; j--) {
                    int ch[4];
                    T.divide(vPrev[j].second, ch);
                    for (int q = 0; q < 4; q++)
                        if (ch[q] >= 0 && T.pool[ch[q]].count > 1) vSize.push_back(SP(T.pool[ch[q]].count, ch[q]));
                    T.erase(vPrev[j].second);
                    if (T.size >= N) break;
                }
                if (T.size >= N || T.size == prevSize) bFinish = true;
            }
        }
    }
    out.reserve(T.size);
    for (int cur = T.head; cur >= 0; cur = T.pool[cur].next) {
        const QNode& nd = T.pool[cur];
        int best = T.perm[nd.begin];
        int maxResponse = cands[best].response;
        for (int k = 1; k < nd.count; k++) {
            const int id = T.perm[nd.begin + k];
            if (cands[id].response > maxResponse) {
                best = id;
                maxResponse = cands[id].response;
            }
        }
        out.push_back(cands[best]);
    }
    return true;
}

/* ------------------------------------------------------------------ matcher host logic */
uint32_t oct_key_path(int x, int y, int W, int H, int depth) {
    const int nIni = (int)std::round((float)W / (float)H);
    const float hX = (float)W / (float)nIni;
    int b = (int)((float)x / hX); /* fextractor.cpp:557-561 */
    b = std::min(b, nIni - 1);
    int x0 = (int)(hX * (float)b), x1 = (int)(hX * (float)(b + 1)), y0 = 0, y1 = H;
    uint32_t code = 0;
    for (int d = 0; d < depth; d++) { /* DivideNode: halfX = ceil((UR.x - UL.x) / 2); kp.x < n1.UR.x, kp.y < n1.BR.y */
        const int mx = x0 + ((x1 - x0 + 1) >> 1), my = y0 + ((y1 - y0 + 1) >> 1);
        const int qx = x < mx ? 0 : 1, qy = y < my ? 0 : 1;
        code = (code << 2) | (uint32_t)qx | ((uint32_t)qy << 1);
        if (qx) x0 = mx; else x1 = mx;
        if (qy) y0 = my; else y1 = my;
    }
    return ((uint32_t)b << (2 * depth)) | code;
}

void build_oct_lut(int W, int H, int D, std::vector<uint32_t>& xs, std::vector<uint32_t>& ys) {
    const int nIni = (int)std::round((float)W / (float)H);
    const float hX = (float)W / (float)nIni;
    xs.assign((size_t)W + 1, 0u);
    ys.assign((size_t)H + 1, 0u);
    for (int x = 0; x <= W; x++) {
        int b = (int)((float)x / hX);
        b = std::min(b, nIni - 1);
        int x0 = (int)(hX * (float)b), x1 = (int)(hX * (float)(b + 1));
        uint32_t code = 0;
        for (int d = 0; d < D; d++) {
            const int mx = x0 + ((x1 - x0 + 1) >> 1);
            const int q = x < mx ? 0 : 1;
            code = (code << 2) | (uint32_t)q;
            if (q) x0 = mx; else x1 = mx;
        }
        xs[x] = ((uint32_t)b << (2 * D)) | code;
    }
    for (int y = 0; y <= H; y++) {
        int y0 = 0, y1 = H;
        uint32_t code = 0;
        for (int d = 0; d < D; d++) {
            const int my = y0 + ((y1 - y0 + 1) >> 1);
            const int q = y < my ? 0 : 1;
            code = (code << 2) | ((uint32_t)q << 1);
            if (q) y0 = my; else y1 = my;
        }
        ys[y] = code;
    }
}

void compute_three_maxima(const int* hs, int L, int& ind1, int& ind2, int& ind3) {
    int max1 = 0, max2 = 0, max3 = 0; /* fmatcher.cpp:2813-2854 */
    for (int i = 0; i < L; i++) {
        const int s = hs[i];
        if (s > max1) {
            max3 = max2; max2 = max1; max1 = s;
            ind3 = ind2; ind2 = ind1; ind1 = i;
        } else if (s > max2) {
            max3 = max2; max2 = s;
            ind3 = ind2; ind2 = i;
        } else if (s > max3) {
            max3 = s;
            ind3 = i;
        }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

void FrameGrid::build(const vslam_kp* k, int n_, int imgW, int imgH) {
    kps = k;
    n = n_;
    minX = 0.0f; maxX = (float)imgW; minY = 0.0f; maxY = (float)imgH; /* frame.cpp:814-820 */
    invW = (float)COLS / (maxX - minX);                                /* frame.cpp:322-323 */
    invH = (float)ROWS / (maxY - minY);
    std::vector<int> cellOf(n, -1);
    cell_start.assign(COLS * ROWS + 1, 0);
    for (int i = 0; i < n; i++) { /* PosInGrid, frame.cpp:746-756 */
        const int px = (int)std::round((k[i].x - minX) * invW);
        const int py = (int)std::round((k[i].y - minY) * invH);
        if (px < 0 || px >= COLS || py < 0 || py >= ROWS) continue;
        cellOf[i] = px * ROWS + py;
        cell_start[cellOf[i] + 1]++;
    }
    for (int cidx = 0; cidx < COLS * ROWS; cidx++) cell_start[cidx + 1] += cell_start[cidx];
    cell_items.resize(cell_start[COLS * ROWS]);
    std::vector<int> w(cell_start.begin(), cell_start.end() - 1);
    for (int i = 0; i < n; i++)
        if (cellOf[i] >= 0) cell_items[w[cellOf[i]]++] = i; /* ascending index inside a cell */
}

void FrameGrid::query(float x, float y, float r, int minLevel, int maxLevel, std::vector<int>& out) const {
    out.clear(); /* frame.cpp:678-744 */
    const int nMinCellX = std::max(0, (int)std::floor((x - minX - r) * invW));
    if (nMinCellX >= COLS) return;
    const int nMaxCellX = std::min(COLS - 1, (int)std::ceil((x - minX + r) * invW));
    if (nMaxCellX < 0) return;
    const int nMinCellY = std::max(0, (int)std::floor((y - minY - r) * invH));
    if (nMinCellY >= ROWS) return;
    const int nMaxCellY = std::min(ROWS - 1, (int)std::ceil((y - minY + r) * invH));
    if (nMaxCellY < 0) return;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
            const int cidx = ix * ROWS + iy;
            for (int e = cell_start[cidx]; e < cell_start[cidx + 1]; e++) {
                const int idx = cell_items[e];
                const vslam_kp& kp = kps[idx];
                if (bCheckLevels) {
                    if (kp.octave < minLevel) continue;
                    if (maxLevel >= 0 && kp.octave > maxLevel) continue;
                }
                const float distx = kp.x - x, disty = kp.y - y;
                if (std::fabs(distx) < r && std::fabs(disty) < r) out.push_back(idx);
            }
        }
}

int search_for_initialization_replay(const vslam_kp* kps1, int n1, const vslam_kp* kps2, int n2,
                                     const uint8_t* dmat, const int* row_of_i1, const int* col_of_i2,
                                     int ncols, int imgW, int imgH, float* prevMatched, int32_t* vnMatches12,
                                     int windowSize, float mfNNratio, bool checkOri) {
    /* fmatcher.cpp:983-1098 */
    const int TH_LOW = 50, HISTO_LENGTH = 30;
    int nmatches = 0;
    for (int i = 0; i < n1; i++) vnMatches12[i] = -1;
    std::vector<int> rotHist[30];
    const float factor = 1.0f / HISTO_LENGTH;
    std::vector<int> vMatchedDistance(n2, INT_MAX), vnMatches21(n2, -1);
    FrameGrid grid2;
    grid2.build(kps2, n2, imgW, imgH);
    std::vector<int> vIndices2;
    for (int i1 = 0; i1 < n1; i1++) {
        const int level1 = kps1[i1].octave;
        if (level1 > 0) continue;
        grid2.query(prevMatched[2 * i1], prevMatched[2 * i1 + 1], (float)windowSize, level1, level1, vIndices2);
        if (vIndices2.empty()) continue;
        const uint8_t* drow = dmat + (size_t)row_of_i1[i1] * ncols;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int i2 : vIndices2) {
            const int dist = drow[col_of_i2[i2]];
            if (vMatchedDistance[i2] <= dist) continue;
            if (dist < bestDist) {
                bestDist2 = bestDist;
                bestDist = dist;
                bestIdx2 = i2;
            } else if (dist < bestDist2) {
                bestDist2 = dist;
            }
        }
        if (bestDist <= TH_LOW) {
            if (bestDist < (float)bestDist2 * mfNNratio) {
                if (vnMatches21[bestIdx2] >= 0) {
                    vnMatches12[vnMatches21[bestIdx2]] = -1;
                    nmatches--;
                }
                vnMatches12[i1] = bestIdx2;
                vnMatches21[bestIdx2] = i1;
                vMatchedDistance[bestIdx2] = bestDist;
                nmatches++;
                if (checkOri) {
                    float rot = kps1[i1].angle - kps2[bestIdx2].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)std::round(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    rotHist[bin].push_back(i1);
                }
            }
        }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        int sizes[30];
        for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int)rotHist[i].size();
        compute_three_maxima(sizes, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx1 : rotHist[i])
                if (vnMatches12[idx1] >= 0) {
                    vnMatches12[idx1] = -1;
                    nmatches--;
                }
        }
    }
    for (int i1 = 0; i1 < n1; i1++)
        if (vnMatches12[i1] >= 0) {
            prevMatched[2 * i1] = kps2[vnMatches12[i1]].x;
            prevMatched[2 * i1 + 1] = kps2[vnMatches12[i1]].y;
        }
    return nmatches;
}

} // namespace vslam

/* ---------------------------------------------------------------------------------------------
 * C hooks for the CPU test-suite (libvslam_host.so); not part of the product ABI.
 * ------------------------------------------------------------------------------------------- */
extern "C" {

int vslamh_tables(int nfeatures, float scale, int nlevels, float* sf, float* isf, float* s2, float* is2,
                  int* quota, int* umax16, int* disc_n) {
    vslam::ExtractorTables t;
    vslam::build_tables(nfeatures, scale, nlevels, t);
    for (int i = 0; i < nlevels; i++) {
        sf[i] = t.scale[i]; isf[i] = t.inv_scale[i]; s2[i] = t.sigma2[i]; is2[i] = t.inv_sigma2[i];
        quota[i] = t.quota[i];
    }
    for (int i = 0; i < 16; i++) umax16[i] = t.umax[i];
    *disc_n = (int)t.disc_u.size();
    return 0;
}

int vslamh_level_size(int nfeatures, float scale, int nlevels, int w, int h, int level, int* lw, int* lh) {
    vslam::ExtractorTables t;
    vslam::build_tables(nfeatures, scale, nlevels, t);
    vslam::level_size(t, w, h, level, lw, lh);
    return 0;
}

/* CPU evaluation of the resize tables exactly as k_resize_level consumes them */
int vslamh_resize_with_tables(const uint8_t* src, int sw, int sh, size_t sstride, uint8_t* dst, int dw, int dh,
                              size_t dstride) {
    vslam::ResizeTables r;
    vslam::build_resize_tables(sw, sh, dw, dh, r);
    for (int dy = 0; dy < dh; dy++)
        for (int dx = 0; dx < dw; dx++) {
            const uint8_t* r0 = src + (size_t)r.ytab[2 * dy] * sstride;
            const uint8_t* r1 = src + (size_t)r.ytab[2 * dy + 1] * sstride;
            const int sx0 = r.xtab[2 * dx], sx1 = r.xtab[2 * dx + 1], a0 = r.xa[2 * dx], a1 = r.xa[2 * dx + 1];
            const int h0 = r0[sx0] * a0 + r0[sx1] * a1, h1 = r1[sx0] * a0 + r1[sx1] * a1;
            const int b0 = r.yb[2 * dy], b1 = r.yb[2 * dy + 1];
            dst[(size_t)dy * dstride + dx] = (uint8_t)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
        }
    return 0;
}

/* The fused pyramid plan evaluated on the CPU: builds the tables and group plans exactly as vslam_fe_create does
 * (groups of levels 1-3, 4-7, ...) and runs emulate_pyramid_group.  out = all levels >= 1 back to back, level l at
 * offset sum of w*h of the levels before it, tightly packed.  Returns the number of groups, or a negative code. */
int vslamh_pyramid_fused(const uint8_t* img, int w, int h, size_t stride, int nfeatures, float scale, int nlevels,
                         uint8_t* out, int* lds_bytes_max, int* tiles_total) {
    vslam::ExtractorTables t;
    vslam::build_tables(nfeatures, scale, nlevels, t);
    std::vector<vslam::PyrLevelTables> tabs(nlevels);
    std::vector<int> lw(nlevels), lh(nlevels);
    for (int l = 0; l < nlevels; l++) vslam::level_size(t, w, h, l, &lw[l], &lh[l]);
    for (int l = 1; l < nlevels; l++) {
        vslam::PyrLevelTables& T = tabs[l];
        T.sw = lw[l - 1]; T.sh = lh[l - 1]; T.dw = lw[l]; T.dh = lh[l];
        vslam::build_resize_tables(T.sw, T.sh, T.dw, T.dh, T.r);
        if (!vslam::build_resize_quads(T.r, T.sw, T.dw, T.qbase, T.quads)) return -100;
    }
    /* level images with the product's pitch (multiple of 128) */
    std::vector<std::vector<uint8_t>> lv(nlevels);
    std::vector<int> pitch(nlevels);
    for (int l = 0; l < nlevels; l++) {
        pitch[l] = (lw[l] + 127) & ~127;
        lv[l].assign((size_t)pitch[l] * lh[l], 0xEE);
    }
    for (int y = 0; y < h; y++) memcpy(&lv[0][(size_t)y * pitch[0]], img + (size_t)y * stride, w);
    int ngroups = 0;
    *lds_bytes_max = 0;
    *tiles_total = 0;
    for (int l0 = 0; l0 + 1 < nlevels;) {
        const int nl = std::min(l0 == 0 ? 3 : 4, nlevels - 1 - l0);
        std::vector<const vslam::PyrLevelTables*> gt;
        for (int j = 1; j <= nl; j++) gt.push_back(&tabs[l0 + j]);
        vslam::PyrGroupPlan plan;
        if (!vslam::build_pyramid_group(gt, l0, 64 * 1024, plan)) return -101;
        std::vector<uint8_t*> dst(nl + 1, nullptr);
        std::vector<int> ds(nl + 1, 0);
        for (int j = 1; j <= nl; j++) { dst[j] = lv[l0 + j].data(); ds[j] = pitch[l0 + j]; }
        /* level 0 may be the caller's image (readable up to w only); internal levels are readable up to their pitch */
        const int rc = vslam::emulate_pyramid_group(plan, gt, lv[l0].data(), pitch[l0], l0 == 0 ? w : pitch[l0], dst, ds);
        if (rc) return rc;
        *lds_bytes_max = std::max(*lds_bytes_max, (int)plan.lds_bytes);
        *tiles_total += plan.ntx * plan.nty;
        ngroups++;
        l0 += nl;
    }
    size_t o = 0;
    for (int l = 1; l < nlevels; l++)
        for (int y = 0; y < lh[l]; y++) {
            memcpy(out + o, &lv[l][(size_t)y * pitch[l]], lw[l]);
            o += lw[l];
        }
    return ngroups;
}

int vslamh_cells(int level, int lw, int lh, uint16_t* out5, int cap) {
    std::vector<vslam::HostCell> c;
    vslam::build_cells(level, lw, lh, c);
    for (int i = 0; i < (int)c.size() && i < cap; i++) {
        out5[5 * i] = c[i].level; out5[5 * i + 1] = c[i].x0; out5[5 * i + 2] = c[i].y0;
        out5[5 * i + 3] = c[i].x1; out5[5 * i + 4] = c[i].y1;
    }
    return (int)c.size();
}

/* bands of all cells of one level: out8 rows = (cell0, level, ncell, wcell, x0, y0, ww, wh) */
int vslamh_bands(int level, int lw, int lh, int max_cells, int max_width, uint32_t* out8, int cap) {
    std::vector<vslam::HostCell> c;
    vslam::build_cells(level, lw, lh, c);
    std::vector<vslam::HostBand> b;
    vslam::build_bands(c, max_cells, max_width, b);
    for (int i = 0; i < (int)b.size() && i < cap; i++) {
        uint32_t* o = out8 + 8 * i;
        o[0] = b[i].cell0; o[1] = b[i].level; o[2] = b[i].ncell; o[3] = b[i].wcell;
        o[4] = b[i].x0; o[5] = b[i].y0; o[6] = b[i].ww; o[7] = b[i].wh;
    }
    return (int)b.size();
}

/* keys: n x (x, y, response) int32 triples; returns count, writes triples */
int vslamh_octree(const int32_t* xyr, int n, int W, int H, int N, int32_t* out_xyr, int cap) {
    std::vector<vslam::Cand> c(n), o;
    for (int i = 0; i < n; i++) {
        c[i].x = (int16_t)xyr[3 * i]; c[i].y = (int16_t)xyr[3 * i + 1]; c[i].response = (uint8_t)xyr[3 * i + 2];
    }
    if (!vslam::distribute_octree(c.data(), n, W, H, N, o)) return -1;
    for (int i = 0; i < (int)o.size() && i < cap; i++) {
        out_xyr[3 * i] = o[i].x; out_xyr[3 * i + 1] = o[i].y; out_xyr[3 * i + 2] = o[i].response;
    }
    return (int)o.size();
}

int vslamh_grid_query(const vslam_kp* kps, int n, int W, int H, float x, float y, float r, int minL, int maxL,
                      int* out, int cap) {
    vslam::FrameGrid g;
    g.build(kps, n, W, H);
    std::vector<int> v;
    g.query(x, y, r, minL, maxL, v);
    for (int i = 0; i < (int)v.size() && i < cap; i++) out[i] = v[i];
    return (int)v.size();
}

/* dense host replay with a caller-provided full n1 x n2 u8 distance matrix */
int vslamh_search_init(const vslam_kp* kps1, int n1, const vslam_kp* kps2, int n2, const uint8_t* dmat_full,
                       int W, int H, float* prevMatched, int32_t* matches12, int window, float nnratio,
                       int checkOri) {
    std::vector<int> rows(n1), cols(n2);
    for (int i = 0; i < n1; i++) rows[i] = i;
    for (int i = 0; i < n2; i++) cols[i] = i;
    return vslam::search_for_initialization_replay(kps1, n1, kps2, n2, dmat_full, rows.data(), cols.data(), n2,
                                                   W, H, prevMatched, matches12, window, nnratio, checkOri != 0);
}

/* quadtree path tables of k_octree_v4 against the literal halvings: returns the number of (x, y) whose table path differs */
int vslamh_oct_lut_check(int W, int H, int D) {
    std::vector<uint32_t> xs, ys;
    vslam::build_oct_lut(W, H, D, xs, ys);
    int bad = 0;
    for (int y = 0; y <= H; y++)
        for (int x = 0; x <= W; x++) bad += (xs[x] | ys[y]) != vslam::oct_key_path(x, y, W, H, D);
    return bad;
}
unsigned vslamh_oct_key_path(int x, int y, int W, int H, int depth) { return vslam::oct_key_path(x, y, W, H, depth); }

} /* extern "C" */

/* ------------------------------------------------------------------ ComputeBoW, host half
 * DBoW3::Vocabulary::transform (Vocabulary.cpp:754-826) after the per-feature tree walk: BowVector::addWeight /
 * addIfNotExist in feature order (std::map, double), the "divide by the number of words" step when the scoring
 * object does not normalise, BowVector::normalize (BowVector.cpp), FeatureVector::addFeature. */
extern "C" int vslam_bow_assemble(int weighting, int norm, const int32_t* word_id, const double* weight,
                                  const int32_t* node_id, int n, int32_t* bow_ids, double* bow_vals, int* n_bow,
                                  int32_t* fv_nodes, int32_t* fv_off, int32_t* fv_feat, int* n_fv) {
    if (n < 0 || (n && (!word_id || !weight || !node_id || !bow_ids || !bow_vals || !fv_nodes || !fv_off || !fv_feat)) ||
        !n_bow || !n_fv || weighting < 0 || weighting > 3 || norm < 0 || norm > 2)
        return -1;
    /* std::map semantics without the tree: a STABLE sort of the kept feature indices by word id (by node id) keeps
     * the feature order inside each key, so every word's weights are summed in exactly the order addWeight saw them */
    std::vector<int32_t> kept;
    kept.reserve(n);
    for (int i = 0; i < n; i++)
        if (weight[i] > 0) kept.push_back(i); /* not stopped */
    const bool tf = weighting == 0 || weighting == 1; /* TF_IDF, TF */
    std::vector<int32_t> byWord(kept);
    std::stable_sort(byWord.begin(), byWord.end(), [&](int32_t a, int32_t b) { return word_id[a] < word_id[b]; });
    int k = 0;
    for (size_t p = 0; p < byWord.size();) {
        const int32_t w = word_id[byWord[p]];
        double v = 0.0;
        bool first = true;
        size_t q = p;
        for (; q < byWord.size() && word_id[byWord[q]] == w; q++) {
            if (tf) v = first ? weight[byWord[q]] : v + weight[byWord[q]]; /* insert, then += */
            else if (first) v = weight[byWord[q]];                       /* addIfNotExist */
            first = false;
        }
        bow_ids[k] = w;
        bow_vals[k++] = v;
        p = q;
    }
    if (tf && k > 0 && norm == 0) {
        const double nd = (double)k;
        for (int i = 0; i < k; i++) bow_vals[i] /= nd;
    }
    if (norm) {
        double nv = 0.0;
        if (norm == 1) for (int i = 0; i < k; i++) nv += std::fabs(bow_vals[i]);
        else {
            for (int i = 0; i < k; i++) nv += bow_vals[i] * bow_vals[i];
            nv = std::sqrt(nv);
        }
        if (nv > 0.0) for (int i = 0; i < k; i++) bow_vals[i] /= nv;
    }
    *n_bow = k;
    std::vector<int32_t> byNode(kept);
    std::stable_sort(byNode.begin(), byNode.end(), [&](int32_t a, int32_t b) { return node_id[a] < node_id[b]; });
    int f = 0, off = 0;
    for (size_t p = 0; p < byNode.size();) {
        const int32_t nd = node_id[byNode[p]];
        fv_nodes[f] = nd;
        fv_off[f++] = off;
        for (; p < byNode.size() && node_id[byNode[p]] == nd; p++) fv_feat[off++] = byNode[p];
    }
    fv_off[f] = off;
    *n_fv = f;
    return 0;
}
