/* oracle/orb_oracle.cpp -- TEST INFRASTRUCTURE ONLY (see orb_oracle.h for the parity status).
 *
 * Plain C++17, no dependencies.  Every function cites the reference lines it restates; the OpenCV 4.2
 * primitives the reference calls are restated from their published algorithms (modules/imgproc
 * resize.cpp, smooth.dispatch.cpp + fixedpoint.inl.hpp, modules/features2d fast.cpp + fast_score.cpp,
 * modules/core mathfuncs_core.simd.hpp, fast_math.hpp).
 *
 * Build: see oracle/Makefile (-O2 -ffp-contract=off: the reference's own objects are x86-64 baseline
 * builds, CMakeLists.txt:14-15 has no -march, so no FMA contraction in fextractor.cpp arithmetic).
 */
#include "orb_oracle.h"
#include <limits>
#include <map>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>
#include <list>
#include <utility>

#include "../include/vslam_orb_pattern.h"

namespace orbo {

static const int PATCH_SIZE = 31;      /* fextractor.cpp:64 */
static const int HALF_PATCH_SIZE = 15; /* :65 */
static const int EDGE_THRESHOLD = 19;  /* :66 */

/* ------------------------------------------------------------------ cvRound & friends */
int cv_round_f(float v) { return (int)lrintf(v); } /* SSE cvtss2si under the default rounding mode */
int cv_round_d(double v) { return (int)lrint(v); }
static inline int cv_floor_f(float v) {
    int i = (int)v;
    return i - (i > v);
}
static inline int cv_floor_d(double v) {
    int i = (int)v;
    return i - (i > v);
}
static inline int cv_ceil_f(float v) {
    int i = (int)v;
    return i + (i < v);
}
static inline short sat_short_from_float(float v) {
    int iv = cv_round_f(v);
    return (short)(iv < SHRT_MIN ? SHRT_MIN : iv > SHRT_MAX ? SHRT_MAX : iv);
}

/* ------------------------------------------------------------------ cv::resize, INTER_LINEAR, CV_8UC1
 * OpenCV 4.2 resize.cpp: hal::resize -> resizeGeneric_<HResizeLinear<uchar,int,short,2048>,
 * VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>>.  IPP is not taken for 8u linear unless
 * useIPP_NotExact().  Coefficients: INTER_RESIZE_COEF_BITS = 11. */
void resize_linear_u8(const uint8_t* src, int sw, int sh, size_t sstride, uint8_t* dst, int dw, int dh,
                      size_t dstride) {
    const int COEF_SCALE = 2048;
    const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;

    std::vector<int> xofs(dw), yofs(dh);
    std::vector<short> ialpha(dw * 2), ibeta(dh * 2);
    int xmax = dw;
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor_f(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= sw) {
            xmax = std::min(xmax, dx);
            if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        }
        xofs[dx] = sx;
        ialpha[dx * 2] = sat_short_from_float((1.f - fx) * COEF_SCALE);
        ialpha[dx * 2 + 1] = sat_short_from_float(fx * COEF_SCALE);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor_f(fy);
        fy -= sy;
        yofs[dy] = sy;
        ibeta[dy * 2] = sat_short_from_float((1.f - fy) * COEF_SCALE);
        ibeta[dy * 2 + 1] = sat_short_from_float(fy * COEF_SCALE);
    }
    std::vector<int> r0(dw), r1(dw);
    auto hresize = [&](int sy, std::vector<int>& D) {
        const uint8_t* S = src + (size_t)sy * sstride;
        int dx = 0;
        for (; dx < xmax; dx++) {
            int sx = xofs[dx];
            D[dx] = S[sx] * ialpha[dx * 2] + S[sx + 1] * ialpha[dx * 2 + 1];
        }
        for (; dx < dw; dx++) D[dx] = S[xofs[dx]] * COEF_SCALE;
    };
    auto clip = [](int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; };
    for (int dy = 0; dy < dh; dy++) {
        int sy0 = clip(yofs[dy], 0, sh), sy1 = clip(yofs[dy] + 1, 0, sh);
        hresize(sy0, r0);
        hresize(sy1, r1);
        const int b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
        uint8_t* D = dst + (size_t)dy * dstride;
        for (int x = 0; x < dw; x++)
            D[x] = (uint8_t)((((b0 * (r0[x] >> 4)) >> 16) + ((b1 * (r1[x] >> 4)) >> 16) + 2) >> 2);
    }
}

/* ------------------------------------------------------------------ cv::FAST (TYPE_9_16)
 * OpenCV 4.2 fast.cpp FAST_t<16> + fast_score.cpp cornerScore<16>. */
static const int RING[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                                {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

static int corner_score16(const uint8_t* p, const int* pixel, int threshold) {
    const int N = 25;
    int v = p[0];
    int d[N];
    for (int k = 0; k < N; k++) d[k] = v - p[pixel[k]];
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = std::min(d[k + 1], d[k + 2]);
        a = std::min(a, d[k + 3]);
        if (a <= a0) continue;
        for (int q = 4; q <= 8; q++) a = std::min(a, d[k + q]);
        a0 = std::max(a0, std::min(a, d[k]));
        a0 = std::max(a0, std::min(a, d[k + 9]));
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = std::max(d[k + 1], d[k + 2]);
        b = std::max(b, d[k + 3]);
        b = std::max(b, d[k + 4]);
        b = std::max(b, d[k + 5]);
        if (b >= b0) continue;
        b = std::max(b, d[k + 6]);
        b = std::max(b, d[k + 7]);
        b = std::max(b, d[k + 8]);
        b0 = std::min(b0, std::max(b, d[k]));
        b0 = std::min(b0, std::max(b, d[k + 9]));
    }
    return -b0 - 1;
}

void fast_detect(const uint8_t* img, int w, int h, size_t stride, int threshold, bool nonmax,
                 std::vector<KeyPoint>& out) {
    out.clear();
    const int K = 8, N = 25;
    int pixel[25];
    for (int k = 0; k < 16; k++) pixel[k] = RING[k][0] + RING[k][1] * (int)stride;
    for (int k = 16; k < N; k++) pixel[k] = pixel[k - 16];
    threshold = std::min(std::max(threshold, 0), 255);
    if (w < 7 || h < 7) return;
    uint8_t threshold_tab[512];
    for (int i = -255; i <= 255; i++)
        threshold_tab[i + 255] = (uint8_t)(i < -threshold ? 1 : i > threshold ? 2 : 0);

    std::vector<uint8_t> bufs(3 * (size_t)w, 0);
    std::vector<int> cps(3 * ((size_t)w + 1), 0);
    uint8_t* buf[3] = {bufs.data(), bufs.data() + w, bufs.data() + 2 * w};
    int* cpbuf[3] = {cps.data(), cps.data() + (w + 1), cps.data() + 2 * (w + 1)};

    for (int i = 3; i < h - 2; i++) {
        const uint8_t* ptr = img + (size_t)i * stride + 3;
        uint8_t* curr = buf[(i - 3) % 3];
        int* cornerpos = cpbuf[(i - 3) % 3] + 1;
        memset(curr, 0, w);
        int ncorners = 0;
        if (i < h - 3) {
            for (int j = 3; j < w - 3; j++, ptr++) {
                const int v = ptr[0];
                const uint8_t* tab = &threshold_tab[0] - v + 255;
                /* FAST_t<16> pre-test: a 9-arc always contains one pixel of each opposite pair */
                int d = tab[ptr[pixel[0]]] | tab[ptr[pixel[8]]];
                if (d == 0) continue;
                d &= tab[ptr[pixel[2]]] | tab[ptr[pixel[10]]];
                d &= tab[ptr[pixel[4]]] | tab[ptr[pixel[12]]];
                d &= tab[ptr[pixel[6]]] | tab[ptr[pixel[14]]];
                if (d == 0) continue;
                d &= tab[ptr[pixel[1]]] | tab[ptr[pixel[9]]];
                d &= tab[ptr[pixel[3]]] | tab[ptr[pixel[11]]];
                d &= tab[ptr[pixel[5]]] | tab[ptr[pixel[13]]];
                d &= tab[ptr[pixel[7]]] | tab[ptr[pixel[15]]];
                const int vd = v - threshold, vb = v + threshold;
                bool corner = false;
                if (d & 1) {
                    int count = 0;
                    for (int k = 0; k < N; k++) { /* >= 9 contiguous darker */
                        if (ptr[pixel[k]] < vd) {
                            if (++count > K) { corner = true; break; }
                        } else
                            count = 0;
                    }
                }
                if (!corner && (d & 2)) {
                    int count = 0;
                    for (int k = 0; k < N; k++) { /* >= 9 contiguous brighter */
                        if (ptr[pixel[k]] > vb) {
                            if (++count > K) { corner = true; break; }
                        } else
                            count = 0;
                    }
                }
                if (corner) {
                    cornerpos[ncorners++] = j;
                    if (nonmax) curr[j] = (uint8_t)corner_score16(ptr, pixel, threshold);
                }
            }
        }
        cornerpos[-1] = ncorners;
        if (i == 3) continue;
        const uint8_t* prev = buf[(i - 4 + 3) % 3];
        const uint8_t* pprev = buf[(i - 5 + 3) % 3];
        cornerpos = cpbuf[(i - 4 + 3) % 3] + 1;
        ncorners = cornerpos[-1];
        for (int k = 0; k < ncorners; k++) {
            int j = cornerpos[k];
            int score = prev[j];
            if (!nonmax || (score > prev[j + 1] && score > prev[j - 1] && score > pprev[j - 1] &&
                            score > pprev[j] && score > pprev[j + 1] && score > curr[j - 1] &&
                            score > curr[j] && score > curr[j + 1])) {
                KeyPoint kp;
                kp.x = (float)j;
                kp.y = (float)(i - 1);
                kp.size = 7.f;
                kp.angle = -1.f;
                kp.response = (float)score;
                kp.octave = 0;
                kp.class_id = -1;
                out.push_back(kp);
            }
        }
    }
}

/* ------------------------------------------------------------------ cv::GaussianBlur 7x7, sigma 2, CV_8U
 * OpenCV 4.2 smooth.dispatch.cpp: a non-submatrix CV_8U source takes GaussianBlurFixedPoint; row pass in
 * ufixedpoint16 (8 fractional bits, exact), column pass in ufixedpoint32, rounding (acc + 2^15) >> 16. */
static inline int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * (len - 1) - p;
    }
    return p;
}

void gaussian_blur7(const Image& src, Image& dst, const int taps[7]) {
    const int w = src.w, h = src.h;
    dst.w = w;
    dst.h = h;
    dst.px.assign((size_t)w * h, 0);
    std::vector<uint16_t> tmp((size_t)w * h);
    const uint32_t k0 = taps[0], k1 = taps[1], k2 = taps[2], k3 = taps[3], k4 = taps[4], k5 = taps[5], k6 = taps[6];
    for (int y = 0; y < h; y++) {
        const uint8_t* s = src.row(y);
        uint16_t* t = tmp.data() + (size_t)y * w;
        auto edge = [&](int x) {
            uint32_t acc = 0;
            for (int k = 0; k < 7; k++) acc += (uint32_t)taps[k] * s[reflect101(x + k - 3, w)];
            t[x] = (uint16_t)std::min<uint32_t>(acc, 0xFFFF); /* ufixedpoint16 adds saturate */
        };
        int x = 0;
        for (; x < std::min(3, w); x++) edge(x);
        for (; x < w - 3; x++) { /* interior: no border handling, vectorisable */
            const uint32_t acc = k0 * s[x - 3] + k1 * s[x - 2] + k2 * s[x - 1] + k3 * s[x] + k4 * s[x + 1] +
                                 k5 * s[x + 2] + k6 * s[x + 3];
            t[x] = (uint16_t)std::min<uint32_t>(acc, 0xFFFF);
        }
        for (; x < w; x++) edge(x);
    }
    for (int y = 0; y < h; y++) {
        const uint16_t* r[7];
        for (int k = 0; k < 7; k++) r[k] = tmp.data() + (size_t)reflect101(y + k - 3, h) * w;
        uint8_t* d = dst.row(y);
        for (int x = 0; x < w; x++) {
            const uint32_t acc = k0 * r[0][x] + k1 * r[1][x] + k2 * r[2][x] + k3 * r[3][x] + k4 * r[4][x] +
                                 k5 * r[5][x] + k6 * r[6][x];
            const uint32_t v = (acc + 32768u) >> 16;
            d[x] = (uint8_t)std::min<uint32_t>(v, 255);
        }
    }
}

/* ------------------------------------------------------------------ cv::fastAtan2 (degrees)
 * OpenCV 4.2 mathfuncs_core.simd.hpp atan_f32(). */
float fast_atan2(float y, float x, int fma) {
    static const float p1 = 0.9997878412794807f * (float)(180 / M_PI);
    static const float p3 = -0.3258083974640975f * (float)(180 / M_PI);
    static const float p5 = 0.1555786518463281f * (float)(180 / M_PI);
    static const float p7 = -0.04432655554792128f * (float)(180 / M_PI);
    const float eps = (float)2.2204460492503131e-16; /* (float)DBL_EPSILON */
    float ax = std::fabs(x), ay = std::fabs(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        if (fma) a = fmaf(fmaf(fmaf(p7, c2, p5), c2, p3), c2, p1) * c;
        else a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        if (fma) a = fmaf(-fmaf(fmaf(fmaf(p7, c2, p5), c2, p3), c2, p1), c, 90.f);
        else a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* ------------------------------------------------------------------ FMatcher::DescriptorDistance */
int descriptor_distance(const uint8_t* a, const uint8_t* b) { /* fmatcher.cpp:2859-2875 */
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t pa, pb;
        memcpy(&pa, a + 4 * i, 4);
        memcpy(&pb, b + 4 * i, 4);
        uint32_t v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

/* ------------------------------------------------------------------ ExtractorNode / DistributeOctTree */
namespace {
struct Node { /* ExtractorNode, fextractor.h:13-24 */
    std::vector<KeyPoint> vKeys;
    int ULx = 0, ULy = 0, URx = 0, URy = 0, BLx = 0, BLy = 0, BRx = 0, BRy = 0;
    std::list<Node>::iterator lit;
    bool bNoMore = false;
    long seq = 0; /* creation order; replaces the heap-pointer tie-break of fextractor.cpp:675 */
};

void divide_node(const Node& p, Node& n1, Node& n2, Node& n3, Node& n4) { /* fextractor.cpp:472-528 */
    const int halfX = (int)std::ceil((float)(p.URx - p.ULx) / 2);
    const int halfY = (int)std::ceil((float)(p.BRy - p.ULy) / 2);
    n1.ULx = p.ULx; n1.ULy = p.ULy;
    n1.URx = p.ULx + halfX; n1.URy = p.ULy;
    n1.BLx = p.ULx; n1.BLy = p.ULy + halfY;
    n1.BRx = p.ULx + halfX; n1.BRy = p.ULy + halfY;
    n2.ULx = n1.URx; n2.ULy = n1.URy;
    n2.URx = p.URx; n2.URy = p.URy;
    n2.BLx = n1.BRx; n2.BLy = n1.BRy;
    n2.BRx = p.URx; n2.BRy = p.ULy + halfY;
    n3.ULx = n1.BLx; n3.ULy = n1.BLy;
    n3.URx = n1.BRx; n3.URy = n1.BRy;
    n3.BLx = p.BLx; n3.BLy = p.BLy;
    n3.BRx = n1.BRx; n3.BRy = p.BLy;
    n4.ULx = n3.URx; n4.ULy = n3.URy;
    n4.URx = n2.BRx; n4.URy = n2.BRy;
    n4.BLx = n3.BRx; n4.BLy = n3.BRy;
    n4.BRx = p.BRx; n4.BRy = p.BRy;
    for (const KeyPoint& kp : p.vKeys) {
        if (kp.x < n1.URx) {
            if (kp.y < n1.BRy) n1.vKeys.push_back(kp);
            else n3.vKeys.push_back(kp);
        } else if (kp.y < n1.BRy)
            n2.vKeys.push_back(kp);
        else
            n4.vKeys.push_back(kp);
    }
    if (n1.vKeys.size() == 1) n1.bNoMore = true;
    if (n2.vKeys.size() == 1) n2.bNoMore = true;
    if (n3.vKeys.size() == 1) n3.bNoMore = true;
    if (n4.vKeys.size() == 1) n4.bNoMore = true;
}
} // namespace

std::vector<KeyPoint> distribute_octree(const std::vector<KeyPoint>& keys, int minX, int maxX, int minY,
                                        int maxY, int N) {
    std::vector<KeyPoint> result;
    const int nIni = (int)std::round((float)(maxX - minX) / (maxY - minY));
    if (nIni < 1) return result; /* the reference indexes an empty vector here (UB); we return nothing */
    const float hX = (float)(maxX - minX) / nIni;
    long seq = 0;
    std::list<Node> lNodes;
    std::vector<Node*> ini(nIni);
    for (int i = 0; i < nIni; i++) {
        Node ni;
        ni.ULx = (int)(hX * (float)i); ni.ULy = 0;
        ni.URx = (int)(hX * (float)(i + 1)); ni.URy = 0;
        ni.BLx = ni.ULx; ni.BLy = maxY - minY;
        ni.BRx = ni.URx; ni.BRy = maxY - minY;
        ni.seq = seq++;
        lNodes.push_back(ni);
        ini[i] = &lNodes.back();
    }
    for (const KeyPoint& kp : keys) {
        int idx = (int)(kp.x / hX);
        if (idx >= nIni) idx = nIni - 1; /* unreachable for in-range keypoints; guards the UB */
        ini[idx]->vKeys.push_back(kp);
    }
    for (auto lit = lNodes.begin(); lit != lNodes.end();) {
        if (lit->vKeys.size() == 1) { lit->bNoMore = true; ++lit; }
        else if (lit->vKeys.empty()) lit = lNodes.erase(lit);
        else ++lit;
    }

    typedef std::pair<int, Node*> SP;
    auto later_created_first = [](const SP& a, const SP& b) { /* ascending; consumed from the back */
        if (a.first != b.first) return a.first < b.first;
        return a.second->seq < b.second->seq;
    };
    auto push_child = [&](Node& n, std::vector<SP>& out, int* nToExpand) {
        if (n.vKeys.empty()) return;
        n.seq = seq++;
        lNodes.push_front(n);
        if (n.vKeys.size() > 1) {
            if (nToExpand) ++*nToExpand;
            out.push_back(SP((int)n.vKeys.size(), &lNodes.front()));
            lNodes.front().lit = lNodes.begin();
        }
    };

    bool bFinish = false;
    std::vector<SP> vSizeAndPointerToNode;
    while (!bFinish) {
        int prevSize = (int)lNodes.size();
        int nToExpand = 0;
        vSizeAndPointerToNode.clear();
        for (auto lit = lNodes.begin(); lit != lNodes.end();) {
            if (lit->bNoMore) { ++lit; continue; }
            Node n1, n2, n3, n4;
            divide_node(*lit, n1, n2, n3, n4);
            push_child(n1, vSizeAndPointerToNode, &nToExpand);
            push_child(n2, vSizeAndPointerToNode, &nToExpand);
            push_child(n3, vSizeAndPointerToNode, &nToExpand);
            push_child(n4, vSizeAndPointerToNode, &nToExpand);
            lit = lNodes.erase(lit);
        }
        if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) {
            bFinish = true;
        } else if (((int)lNodes.size() + nToExpand * 3) > N) {
            while (!bFinish) {
                prevSize = (int)lNodes.size();
                std::vector<SP> prev = vSizeAndPointerToNode;
                vSizeAndPointerToNode.clear();
                std::sort(prev.begin(), prev.end(), later_created_first);
                for (int j = (int)prev.size() - 1; j >= 0; j--) {
                    Node n1, n2, n3, n4;
                    divide_node(*prev[j].second, n1, n2, n3, n4);
                    push_child(n1, vSizeAndPointerToNode, nullptr);
                    push_child(n2, vSizeAndPointerToNode, nullptr);
                    push_child(n3, vSizeAndPointerToNode, nullptr);
                    push_child(n4, vSizeAndPointerToNode, nullptr);
                    lNodes.erase(prev[j].second->lit);
                    if ((int)lNodes.size() >= N) break;
                }
                if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) bFinish = true;
            }
        }
    }
    result.reserve(lNodes.size());
    for (const Node& n : lNodes) {
        const KeyPoint* best = &n.vKeys[0];
        float maxResponse = best->response;
        for (size_t k = 1; k < n.vKeys.size(); k++)
            if (n.vKeys[k].response > maxResponse) {
                best = &n.vKeys[k];
                maxResponse = n.vKeys[k].response;
            }
        result.push_back(*best);
    }
    return result;
}

/* ------------------------------------------------------------------ FExtractor */
Extractor::Extractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST,
                     const Knobs& _knobs)
    : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_iniThFAST),
      minThFAST(_minThFAST), knobs(_knobs) { /* fextractor.cpp:401-461 */
    mvScaleFactor.resize(nlevels);
    mvLevelSigma2.resize(nlevels);
    mvScaleFactor[0] = 1.0f;
    mvLevelSigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
        mvScaleFactor[i] = (float)(mvScaleFactor[i - 1] * scaleFactor); /* float * double member */
        mvLevelSigma2[i] = mvScaleFactor[i] * mvScaleFactor[i];
    }
    mvInvScaleFactor.resize(nlevels);
    mvInvLevelSigma2.resize(nlevels);
    for (int i = 0; i < nlevels; i++) {
        mvInvScaleFactor[i] = 1.0f / mvScaleFactor[i];
        mvInvLevelSigma2[i] = 1.0f / mvLevelSigma2[i];
    }
    mvImagePyramid.resize(nlevels);
    mnFeaturesPerLevel.resize(nlevels);
    float factor = (float)(1.0f / scaleFactor);
    float nDesired = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sumFeatures = 0;
    for (int level = 0; level < nlevels - 1; level++) {
        mnFeaturesPerLevel[level] = cv_round_f(nDesired);
        sumFeatures += mnFeaturesPerLevel[level];
        nDesired *= factor;
    }
    mnFeaturesPerLevel[nlevels - 1] = std::max(nfeatures - sumFeatures, 0);

    umax.resize(HALF_PATCH_SIZE + 1);
    int v, v0, vmax = cv_floor_f(HALF_PATCH_SIZE * sqrtf(2.f) / 2 + 1);
    int vmin = cv_ceil_f(HALF_PATCH_SIZE * sqrtf(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) umax[v] = cv_round_d(sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
}

void Extractor::ComputePyramid(const uint8_t* img, int w, int h, size_t stride) { /* :1135-1160 */
    for (int level = 0; level < nlevels; ++level) {
        float scale = mvInvScaleFactor[level];
        int sw = cv_round_f((float)w * scale), sh = cv_round_f((float)h * scale);
        Image& L = mvImagePyramid[level];
        L.w = sw;
        L.h = sh;
        L.px.assign((size_t)sw * sh, 0);
        if (level != 0) {
            const Image& P = mvImagePyramid[level - 1];
            resize_linear_u8(P.px.data(), P.w, P.h, P.w, L.px.data(), sw, sh, sw);
        } else {
            for (int y = 0; y < h; y++) memcpy(L.row(y), img + (size_t)y * stride, w);
        }
        /* the 19-px reflect-101 border of :1150-1156 is never read on this path */
    }
}

static float ic_angle(const Image& im, float px, float py, const std::vector<int>& u_max, int fma) {
    int m_01 = 0, m_10 = 0; /* fextractor.cpp:68-95 */
    const int cx = cv_round_f(px), cy = cv_round_f(py);
    const uint8_t* center = im.row(cy) + cx;
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
    const int step = im.w;
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
        int v_sum = 0;
        int d = u_max[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * step], val_minus = center[u - v * step];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return fast_atan2((float)m_01, (float)m_10, fma);
}

void Extractor::ComputeKeyPointsOctTree() { /* fextractor.cpp:756-844 */
    allKeypoints.assign(nlevels, std::vector<KeyPoint>());
    vToDistribute.assign(nlevels, std::vector<KeyPoint>());
    const float W = 30;
    for (int level = 0; level < nlevels; ++level) {
        const Image& im = mvImagePyramid[level];
        const int minBorderX = EDGE_THRESHOLD - 3;
        const int minBorderY = minBorderX;
        const int maxBorderX = im.w - EDGE_THRESHOLD + 3;
        const int maxBorderY = im.h - EDGE_THRESHOLD + 3;
        std::vector<KeyPoint>& vToDistributeKeys = vToDistribute[level];
        const float width = (float)(maxBorderX - minBorderX);
        const float height = (float)(maxBorderY - minBorderY);
        const int nCols = (int)(width / W);
        const int nRows = (int)(height / W);
        if (nCols < 1 || nRows < 1) continue; /* reference divides by zero here; too-small level */
        const int wCell = (int)std::ceil(width / nCols);
        const int hCell = (int)std::ceil(height / nRows);
        std::vector<KeyPoint> vKeysCell;
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
                const int x0 = (int)iniX, y0 = (int)iniY, x1 = (int)maxX, y1 = (int)maxY;
                fast_detect(im.row(y0) + x0, x1 - x0, y1 - y0, im.w, iniThFAST, true, vKeysCell);
                if (vKeysCell.empty())
                    fast_detect(im.row(y0) + x0, x1 - x0, y1 - y0, im.w, minThFAST, true, vKeysCell);
                for (KeyPoint& kp : vKeysCell) {
                    kp.x += j * wCell;
                    kp.y += i * hCell;
                    vToDistributeKeys.push_back(kp);
                }
            }
        }
        std::vector<KeyPoint>& keypoints = allKeypoints[level];
        keypoints = distribute_octree(vToDistributeKeys, minBorderX, maxBorderX, minBorderY, maxBorderY,
                                      mnFeaturesPerLevel[level]);
        const int scaledPatchSize = (int)(PATCH_SIZE * mvScaleFactor[level]);
        for (KeyPoint& kp : keypoints) {
            kp.x += minBorderX;
            kp.y += minBorderY;
            kp.octave = level;
            kp.size = (float)scaledPatchSize;
        }
    }
    for (int level = 0; level < nlevels; ++level)
        for (KeyPoint& kp : allKeypoints[level])
            kp.angle = ic_angle(mvImagePyramid[level], kp.x, kp.y, umax, knobs.atan_fma);
}

static void compute_orb_descriptor(const KeyPoint& kpt, const Image& img, uint8_t* desc) {
    /* fextractor.cpp:98-138.  cos/sin resolve to the float overloads (cosf/sinf of the host libm). */
    const float factorPI = (float)(M_PI / 180.f);
    float angle = (float)kpt.angle * factorPI;
    float a = cosf(angle), b = sinf(angle);
    const uint8_t* center = img.row(cv_round_f(kpt.y)) + cv_round_f(kpt.x);
    const int step = img.w;
    const signed char* pat = VSLAM_ORB_PATTERN;
    auto get = [&](int idx) -> int {
        const float px = (float)pat[2 * idx], py = (float)pat[2 * idx + 1];
        return center[cv_round_f(px * b + py * a) * step + cv_round_f(px * a - py * b)];
    };
    for (int i = 0; i < 32; ++i, pat += 32) {
        int val = 0;
        for (int k = 0; k < 8; k++) {
            int t0 = get(2 * k), t1 = get(2 * k + 1);
            val |= (t0 < t1) << k;
        }
        desc[i] = (uint8_t)val;
    }
}

int Extractor::compute(const uint8_t* img, int w, int h, size_t stride, int lap0, int lap1,
                       std::vector<KeyPoint>& kps, std::vector<uint8_t>& desc) { /* :1034-1133 */
    if (!img || w <= 0 || h <= 0) return -1;
    ComputePyramid(img, w, h, stride);
    ComputeKeyPointsOctTree();
    int nkeypoints = 0;
    for (int level = 0; level < nlevels; ++level) nkeypoints += (int)allKeypoints[level].size();
    kps.assign(nkeypoints, KeyPoint());
    desc.assign((size_t)nkeypoints * 32, 0);
    mvBlurred.assign(nlevels, Image());
    int monoIndex = 0, stereoIndex = nkeypoints - 1;
    for (int level = 0; level < nlevels; ++level) {
        std::vector<KeyPoint>& keypoints = allKeypoints[level];
        if (keypoints.empty()) continue;
        gaussian_blur7(mvImagePyramid[level], mvBlurred[level], knobs.gauss_taps);
        std::vector<uint8_t> d(keypoints.size() * 32);
        for (size_t i = 0; i < keypoints.size(); i++)
            compute_orb_descriptor(keypoints[i], mvBlurred[level], &d[i * 32]);
        float scale = mvScaleFactor[level];
        for (size_t i = 0; i < keypoints.size(); i++) {
            KeyPoint kp = keypoints[i]; /* allKeypoints keeps level coordinates for the tests */
            if (level != 0) {
                kp.x *= scale;
                kp.y *= scale;
            }
            int dstIdx;
            if (kp.x >= lap0 && kp.x <= lap1) dstIdx = stereoIndex--;
            else dstIdx = monoIndex++;
            kps[dstIdx] = kp;
            memcpy(&desc[(size_t)dstIdx * 32], &d[i * 32], 32);
        }
    }
    return monoIndex;
}

/* ------------------------------------------------------------------ Frame::ComputeStereoMatches */
void compute_stereo_matches(const Extractor& exL, const Extractor& exR, const std::vector<KeyPoint>& kpsL,
                            const std::vector<uint8_t>& descL, const std::vector<KeyPoint>& kpsR,
                            const std::vector<uint8_t>& descR, float mbf, float fx,
                            std::vector<float>& mvuRight, std::vector<float>& mvDepth,
                            std::vector<int>* outBestIdxR, std::vector<int>* outBestSad) {
    /* frame.cpp:823-997 */
    const int N = (int)kpsL.size();
    mvuRight.assign(N, -1.0f);
    mvDepth.assign(N, -1.0f);
    if (outBestIdxR) outBestIdxR->assign(N, -1);
    if (outBestSad) outBestSad->assign(N, -1);
    const int TH_HIGH = 100, TH_LOW = 50;
    const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
    const int nRows = exL.mvImagePyramid[0].h;
    std::vector<std::vector<int>> vRowIndices(nRows);
    const int Nr = (int)kpsR.size();
    for (int iR = 0; iR < Nr; iR++) {
        const KeyPoint& kp = kpsR[iR];
        const float kpY = kp.y;
        const float r = 2.0f * exL.mvScaleFactor[kp.octave];
        const int maxr = (int)std::ceil(kpY + r);
        const int minr = (int)std::floor(kpY - r);
        for (int yi = minr; yi <= maxr; yi++)
            if (yi >= 0 && yi < nRows) vRowIndices[yi].push_back(iR); /* reference: unchecked */
    }
    const float mb = mbf / fx; /* frame.cpp:157 */
    const float minZ = mb;
    const float minD = 0;
    const float maxD = mbf / minZ;
    std::vector<std::pair<int, int>> vDistIdx;
    for (int iL = 0; iL < N; iL++) {
        const KeyPoint& kpL = kpsL[iL];
        const int levelL = kpL.octave;
        const float vL = kpL.y, uL = kpL.x;
        const std::vector<int>& vCandidates = vRowIndices[(size_t)vL];
        if (vCandidates.empty()) continue;
        const float minU = uL - maxD, maxU = uL - minD;
        if (maxU < 0) continue;
        int bestDist = TH_HIGH;
        int bestIdxR = 0;
        const uint8_t* dL = &descL[(size_t)iL * 32];
        for (int iR : vCandidates) {
            const KeyPoint& kpR = kpsR[iR];
            if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
            const float uR = kpR.x;
            if (uR >= minU && uR <= maxU) {
                const int dist = descriptor_distance(dL, &descR[(size_t)iR * 32]);
                if (dist < bestDist) {
                    bestDist = dist;
                    bestIdxR = iR;
                }
            }
        }
        if (bestDist < thOrbDist) {
            if (outBestIdxR) (*outBestIdxR)[iL] = bestIdxR;
            const float uR0 = kpsR[bestIdxR].x;
            const float sf = exL.mvInvScaleFactor[kpL.octave];
            const float scaleduL = std::round(kpL.x * sf);
            const float scaledvL = std::round(kpL.y * sf);
            const float scaleduR0 = std::round(uR0 * sf);
            const int w = 5;
            const Image& imL = exL.mvImagePyramid[kpL.octave];
            const Image& imR = exR.mvImagePyramid[kpL.octave];
            const int cyL = (int)scaledvL, cxL = (int)scaleduL, cxR0 = (int)scaleduR0;
            int bestSad = INT_MAX, bestincR = 0;
            const int L = 5;
            float vDists[2 * 5 + 1];
            const float iniu = scaleduR0 + L - w;
            const float endu = scaleduR0 + L + w + 1;
            if (iniu < 0 || endu >= imR.w) continue;
            /* The reference reads sub-matrices of the bordered pyramid buffers, so an index a few pixels
             * outside the level lands in its reflect-101 border (fextractor.cpp:1150-1156); restate that
             * with reflect101() instead of carrying the border.  Keypoints sit >= 19 px inside, so in
             * practice every index is interior. */
            auto pxL = [&](int x, int y) { return (int)imL.row(reflect101(y, imL.h))[reflect101(x, imL.w)]; };
            auto pxR = [&](int x, int y) { return (int)imR.row(reflect101(y, imR.h))[reflect101(x, imR.w)]; };
            const int cL = pxL(cxL, cyL);
            for (int incR = -L; incR <= +L; incR++) {
                const int cxR = cxR0 + incR;
                const int cR = pxR(cxR, cyL);
                long sad = 0; /* cv::norm(IL,IR,NORM_L1) on centre-subtracted float patches: exact ints */
                for (int dy = -w; dy <= w; dy++)
                    for (int dx = -w; dx <= w; dx++)
                        sad += std::abs((pxL(cxL + dx, cyL + dy) - cL) - (pxR(cxR + dx, cyL + dy) - cR));
                float dist = (float)sad;
                if (dist < bestSad) {
                    bestSad = (int)dist;
                    bestincR = incR;
                }
                vDists[L + incR] = dist;
            }
            if (bestincR == -L || bestincR == L) continue;
            const float dist1 = vDists[L + bestincR - 1];
            const float dist2 = vDists[L + bestincR];
            const float dist3 = vDists[L + bestincR + 1];
            const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
            if (deltaR < -1 || deltaR > 1) continue;
            float bestuR = exL.mvScaleFactor[kpL.octave] * ((float)scaleduR0 + (float)bestincR + deltaR);
            float disparity = (uL - bestuR);
            if (disparity >= minD && disparity < maxD) {
                if (disparity <= 0) {
                    disparity = 0.01;
                    bestuR = uL - 0.01;
                }
                mvDepth[iL] = mbf / disparity;
                mvuRight[iL] = bestuR;
                vDistIdx.push_back(std::pair<int, int>(bestSad, iL));
                if (outBestSad) (*outBestSad)[iL] = bestSad;
            }
        }
    }
    if (vDistIdx.empty()) return; /* reference reads vDistIdx[0] of an empty vector (UB) */
    std::sort(vDistIdx.begin(), vDistIdx.end());
    const float median = (float)vDistIdx[vDistIdx.size() / 2].first;
    const float thDist = 1.5f * 1.4f * median;
    for (int i = (int)vDistIdx.size() - 1; i >= 0; i--) {
        if (vDistIdx[i].first < thDist) break;
        mvuRight[vDistIdx[i].second] = -1;
        mvDepth[vDistIdx[i].second] = -1;
    }
}

/* ------------------------------------------------------------------ Frame grid */
FrameGrid::FrameGrid(const std::vector<KeyPoint>& k, int imgW, int imgH) : kps(&k) {
    mnMinX = 0.0f; /* frame.cpp:814-820 (no distortion) */
    mnMaxX = (float)imgW;
    mnMinY = 0.0f;
    mnMaxY = (float)imgH;
    invW = (float)COLS / (mnMaxX - mnMinX); /* frame.cpp:322-323 */
    invH = (float)ROWS / (mnMaxY - mnMinY);
    for (int i = 0; i < (int)k.size(); i++) { /* AssignFeaturesToGrid + PosInGrid, :386-414,746-756 */
        int posX = (int)std::round((k[i].x - mnMinX) * invW);
        int posY = (int)std::round((k[i].y - mnMinY) * invH);
        if (posX < 0 || posX >= COLS || posY < 0 || posY >= ROWS) continue;
        cell[posX][posY].push_back(i);
    }
}

std::vector<int> FrameGrid::GetFeaturesInArea(float x, float y, float r, int minLevel, int maxLevel) const {
    std::vector<int> vIndices; /* frame.cpp:678-744 */
    const float factorX = r, factorY = r;
    const int nMinCellX = std::max(0, (int)std::floor((x - mnMinX - factorX) * invW));
    if (nMinCellX >= COLS) return vIndices;
    const int nMaxCellX = std::min(COLS - 1, (int)std::ceil((x - mnMinX + factorX) * invW));
    if (nMaxCellX < 0) return vIndices;
    const int nMinCellY = std::max(0, (int)std::floor((y - mnMinY - factorY) * invH));
    if (nMinCellY >= ROWS) return vIndices;
    const int nMaxCellY = std::min(ROWS - 1, (int)std::ceil((y - mnMinY + factorY) * invH));
    if (nMaxCellY < 0) return vIndices;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++)
            for (int idx : cell[ix][iy]) {
                const KeyPoint& kp = (*kps)[idx];
                if (bCheckLevels) {
                    if (kp.octave < minLevel) continue;
                    if (maxLevel >= 0 && kp.octave > maxLevel) continue;
                }
                const float distx = kp.x - x, disty = kp.y - y;
                if (std::fabs(distx) < factorX && std::fabs(disty) < factorY) vIndices.push_back(idx);
            }
    return vIndices;
}

/* ------------------------------------------------------------------ FMatcher::SearchForInitialization */
static void compute_three_maxima(const std::vector<int>* histo, int L, int& ind1, int& ind2, int& ind3) {
    int max1 = 0, max2 = 0, max3 = 0; /* fmatcher.cpp:2813-2854 */
    for (int i = 0; i < L; i++) {
        const int s = (int)histo[i].size();
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

int search_for_initialization(const std::vector<KeyPoint>& kps1, const std::vector<uint8_t>& desc1,
                              const std::vector<KeyPoint>& kps2, const std::vector<uint8_t>& desc2,
                              int imgW, int imgH, std::vector<float>& prevMatched,
                              std::vector<int>& vnMatches12, int windowSize, float mfNNratio,
                              bool checkOri) { /* fmatcher.cpp:983-1098 */
    const int TH_LOW = 50, HISTO_LENGTH = 30;
    int nmatches = 0;
    vnMatches12.assign(kps1.size(), -1);
    std::vector<int> rotHist[30];
    const float factor = 1.0f / HISTO_LENGTH;
    std::vector<int> vMatchedDistance(kps2.size(), INT_MAX);
    std::vector<int> vnMatches21(kps2.size(), -1);
    FrameGrid grid2(kps2, imgW, imgH);
    for (size_t i1 = 0; i1 < kps1.size(); i1++) {
        const KeyPoint& kp1 = kps1[i1];
        int level1 = kp1.octave;
        if (level1 > 0) continue;
        std::vector<int> vIndices2 = grid2.GetFeaturesInArea(prevMatched[2 * i1], prevMatched[2 * i1 + 1],
                                                             (float)windowSize, level1, level1);
        if (vIndices2.empty()) continue;
        const uint8_t* d1 = &desc1[i1 * 32];
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int i2 : vIndices2) {
            int dist = descriptor_distance(d1, &desc2[(size_t)i2 * 32]);
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
                vnMatches21[bestIdx2] = (int)i1;
                vMatchedDistance[bestIdx2] = bestDist;
                nmatches++;
                if (checkOri) {
                    float rot = kps1[i1].angle - kps2[bestIdx2].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)std::round(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    rotHist[bin].push_back((int)i1);
                }
            }
        }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        compute_three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx1 : rotHist[i])
                if (vnMatches12[idx1] >= 0) {
                    vnMatches12[idx1] = -1;
                    nmatches--;
                }
        }
    }
    for (size_t i1 = 0; i1 < vnMatches12.size(); i1++)
        if (vnMatches12[i1] >= 0) {
            prevMatched[2 * i1] = kps2[vnMatches12[i1]].x;
            prevMatched[2 * i1 + 1] = kps2[vnMatches12[i1]].y;
        }
    return nmatches;
}


/* ---- cv::Mat float algebra as the reference's expressions evaluate it.  `A*x + t` is ONE cv::gemm call
 * (MatExpr folds the addition in as beta*C); for CV_32F OpenCV's GEMMSingleMul<float,double> accumulates in
 * double and rounds once: d = float(sum_k double(a_k)*double(b_k) + double(c)).  OpenCV is not in this image, so
 * this is restated from its published source and kept behind a knob (gemmDouble = 0: plain float arithmetic). */
static inline float gemm_row(const float* r, const float* x, float t, int dbl) {
    if (dbl) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)r[k] * (double)x[k];
        return (float)(s + (double)t);
    }
    float s = 0;
    for (int k = 0; k < 3; k++) s += r[k] * x[k];
    return s + t;
}

int search_by_projection_mappoints(const std::vector<MapPointTrack>& mps, const std::vector<uint8_t>& mpDesc,
                                   const std::vector<KeyPoint>& curKps, const std::vector<uint8_t>& curDesc,
                                   const std::vector<float>& mvuRight, const std::vector<uint8_t>& occupied0,
                                   const std::vector<float>& scaleFactors, int imgW, int imgH, float th, float nnratio,
                                   std::vector<int>& matchCur) { /* fmatcher.cpp:321-411, Nleft == -1 */
    const int TH_HIGH = 100;
    int nmatches = 0;
    const bool bFactor = th != 1.0;
    const int N2 = (int)curKps.size();
    matchCur.assign(N2, -1);
    std::vector<uint8_t> occ(N2, 0); /* F.mvpMapPoints[idx] && Observations() > 0 */
    for (int i = 0; i < N2 && i < (int)occupied0.size(); i++) occ[i] = occupied0[i];
    FrameGrid grid(curKps, imgW, imgH);
    for (size_t iMP = 0; iMP < mps.size(); iMP++) {
        const MapPointTrack& mp = mps[iMP];
        if (!(mp.flags & 1)) continue; /* !mbTrackInView / far point / isBad() */
        const int nPredictedLevel = mp.level;
        float r = mp.viewCos > 0.998 ? 2.5f : 4.0f; /* RadiusByViewingCos, fmatcher.cpp:493-499 */
        if (bFactor) r *= th;
        const std::vector<int> vIndices = grid.GetFeaturesInArea(mp.projX, mp.projY, r * scaleFactors[nPredictedLevel],
                                                                 nPredictedLevel - 1, nPredictedLevel);
        if (vIndices.empty()) continue;
        const uint8_t* MPdescriptor = &mpDesc[32 * iMP];
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int idx : vIndices) {
            if (occ[idx]) continue;
            if (mvuRight[idx] > 0) {
                const float er = std::fabs(mp.projXR - mvuRight[idx]);
                if (er > r * scaleFactors[nPredictedLevel]) continue;
            }
            const int dist = descriptor_distance(MPdescriptor, &curDesc[32 * (size_t)idx]);
            if (dist < bestDist) {
                bestDist2 = bestDist;
                bestDist = dist;
                bestLevel2 = bestLevel;
                bestLevel = curKps[idx].octave;
                bestIdx = idx;
            } else if (dist < bestDist2) {
                bestLevel2 = curKps[idx].octave;
                bestDist2 = dist;
            }
        }
        if (bestDist <= TH_HIGH) {
            if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
            if (bestLevel != bestLevel2 || bestDist <= nnratio * bestDist2) {
                matchCur[bestIdx] = (int)iMP;
                if (mp.flags & 2) occ[bestIdx] = 1;
                nmatches++;
            }
        }
    }
    return nmatches;
}

int distinctive_descriptor(const uint8_t* desc, int n) { /* mappoint.cpp:358-385 */
    if (n <= 0) return -1;
    const size_t N = (size_t)n;
    std::vector<float> Distances(N * N);
    for (size_t i = 0; i < N; i++) {
        Distances[i * N + i] = 0;
        for (size_t j = i + 1; j < N; j++) {
            const int distij = descriptor_distance(desc + 32 * i, desc + 32 * j);
            Distances[i * N + j] = (float)distij;
            Distances[j * N + i] = (float)distij;
        }
    }
    int BestMedian = INT_MAX, BestIdx = 0;
    for (size_t i = 0; i < N; i++) {
        std::vector<int> vDists(Distances.begin() + i * N, Distances.begin() + (i + 1) * N);
        std::sort(vDists.begin(), vDists.end());
        const int median = vDists[(size_t)(0.5 * (N - 1))];
        if (median < BestMedian) {
            BestMedian = median;
            BestIdx = (int)i;
        }
    }
    return BestIdx;
}

void bow_transform_feature(const Vocabulary& v, const uint8_t* feature, int levelsup, int& word, double& weight,
                           int& nid) { /* Vocabulary.cpp:838-878 */
    const int nid_level = v.L - levelsup;
    nid = 0;                 /* if (nid_level <= 0) *nid = 0 (root); otherwise set on the way down */
    int final_id = 0, current_level = 0;
    do {
        ++current_level;
        double best_d = std::numeric_limits<double>::max();
        const int c0 = v.childStart[final_id], cn = v.childCount[final_id];
        int next = final_id;
        for (int j = 0; j < cn; j++) {
            const int id = v.childIds[c0 + j];
            const double d = (double)descriptor_distance(feature, &v.desc[32 * (size_t)id]); /* DescManip::distance */
            if (d < best_d) {
                best_d = d;
                next = id;
            }
        }
        final_id = next;
        if (current_level == nid_level) nid = final_id;
    } while (v.childCount[final_id] != 0);
    word = v.wordId[final_id];
    weight = v.weight[final_id];
}

void bow_transform(const Vocabulary& v, const uint8_t* desc, int n, int levelsup, BowResult& out) {
    /* Vocabulary.cpp:754-826 with BowVector (std::map<WordId, WordValue>) and FeatureVector semantics */
    std::map<int, double> bow;
    std::map<int, std::vector<unsigned>> fv;
    const bool tf = v.weighting == 0 || v.weighting == 1;
    for (int i = 0; i < n; i++) {
        int id, nid;
        double w;
        bow_transform_feature(v, desc + 32 * (size_t)i, levelsup, id, w, nid);
        if (w > 0) { /* not stopped */
            if (tf) bow[id] += w;                     /* addWeight */
            else if (!bow.count(id)) bow[id] = w;     /* addIfNotExist */
            fv[nid].push_back((unsigned)i);
        }
    }
    if (tf && !bow.empty() && v.norm == 0) {
        const double nd = (double)bow.size();
        for (auto& kv : bow) kv.second /= nd;
    }
    if (v.norm) { /* BowVector::normalize */
        double norm = 0.0;
        if (v.norm == 1) for (auto& kv : bow) norm += std::fabs(kv.second);
        else {
            for (auto& kv : bow) norm += kv.second * kv.second;
            norm = std::sqrt(norm);
        }
        if (norm > 0.0) for (auto& kv : bow) kv.second /= norm;
    }
    out.words.clear(); out.values.clear(); out.nodes.clear(); out.features.clear();
    for (auto& kv : bow) { out.words.push_back(kv.first); out.values.push_back(kv.second); }
    for (auto& kv : fv) { out.nodes.push_back(kv.first); out.features.push_back(kv.second); }
}

int search_by_bow(const std::vector<KeyPoint>& kfKps, const std::vector<uint8_t>& kfDesc, const std::vector<uint8_t>& kfFlags,
                  const std::vector<int>& kfNodes, const std::vector<int>& kfOff, const std::vector<int>& kfFeat,
                  const std::vector<KeyPoint>& fKps, const std::vector<uint8_t>& fDesc, const std::vector<int>& fNodes,
                  const std::vector<int>& fOff, const std::vector<int>& fFeat, float mfNNratio, bool checkOri,
                  std::vector<int>& vpMapPointMatches) { /* fmatcher.cpp:546-748, F.Nleft == -1 */
    const int TH_LOW = 50, HISTO_LENGTH = 30;
    vpMapPointMatches.assign(fKps.size(), -1);
    int nmatches = 0;
    std::vector<int> rotHist[30];
    const float factor = 1.0f / HISTO_LENGTH;
    size_t KFit = 0, Fit = 0;
    const size_t KFend = kfNodes.size(), Fend = fNodes.size();
    while (KFit != KFend && Fit != Fend) {
        if (kfNodes[KFit] == fNodes[Fit]) {
            for (int a = kfOff[KFit]; a < kfOff[KFit + 1]; a++) {
                const int realIdxKF = kfFeat[a];
                if (!kfFlags[realIdxKF]) continue; /* !pMP || pMP->isBad() */
                const uint8_t* dKF = &kfDesc[32 * (size_t)realIdxKF];
                int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
                for (int b = fOff[Fit]; b < fOff[Fit + 1]; b++) {
                    const int realIdxF = fFeat[b];
                    if (vpMapPointMatches[realIdxF] >= 0) continue;
                    const int dist = descriptor_distance(dKF, &fDesc[32 * (size_t)realIdxF]);
                    if (dist < bestDist1) {
                        bestDist2 = bestDist1;
                        bestDist1 = dist;
                        bestIdxF = realIdxF;
                    } else if (dist < bestDist2) {
                        bestDist2 = dist;
                    }
                }
                if (bestDist1 <= TH_LOW) {
                    if ((float)bestDist1 < mfNNratio * (float)bestDist2) {
                        vpMapPointMatches[bestIdxF] = realIdxKF;
                        if (checkOri) {
                            float rot = kfKps[realIdxKF].angle - fKps[bestIdxF].angle;
                            if (rot < 0.0) rot += 360.0f;
                            int bin = (int)std::round(rot * factor);
                            if (bin == HISTO_LENGTH) bin = 0;
                            rotHist[bin].push_back(bestIdxF);
                        }
                        nmatches++;
                    }
                }
            }
            KFit++;
            Fit++;
        } else if (kfNodes[KFit] < fNodes[Fit]) {
            KFit = std::lower_bound(kfNodes.begin(), kfNodes.end(), fNodes[Fit]) - kfNodes.begin();
        } else {
            Fit = std::lower_bound(fNodes.begin(), fNodes.end(), kfNodes[KFit]) - fNodes.begin();
        }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        compute_three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx : rotHist[i]) {
                vpMapPointMatches[idx] = -1;
                nmatches--;
            }
        }
    }
    return nmatches;
}

int search_by_bow_keyframes(const std::vector<KeyPoint>& vKeysUn1, const std::vector<uint8_t>& Descriptors1,
                            const std::vector<uint8_t>& flags1, const std::vector<int>& nodes1, const std::vector<int>& off1,
                            const std::vector<int>& feat1, const std::vector<KeyPoint>& vKeysUn2,
                            const std::vector<uint8_t>& Descriptors2, const std::vector<uint8_t>& flags2,
                            const std::vector<int>& nodes2, const std::vector<int>& off2, const std::vector<int>& feat2,
                            float mfNNratio, bool checkOri, std::vector<int>& vpMatches12) { /* fmatcher.cpp:1100-1240 */
    const int TH_LOW = 50, HISTO_LENGTH = 30;
    vpMatches12.assign(vKeysUn1.size(), -1);
    std::vector<bool> vbMatched2(vKeysUn2.size(), false);
    std::vector<int> rotHist[30];
    const float factor = 1.0f / HISTO_LENGTH;
    int nmatches = 0;
    size_t f1it = 0, f2it = 0;
    while (f1it != nodes1.size() && f2it != nodes2.size()) {
        if (nodes1[f1it] == nodes2[f2it]) {
            for (int a = off1[f1it]; a < off1[f1it + 1]; a++) {
                const int idx1 = feat1[a];
                if (!flags1[idx1]) continue; /* !pMP1 || pMP1->isBad() */
                const uint8_t* d1 = &Descriptors1[32 * (size_t)idx1];
                int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
                for (int b = off2[f2it]; b < off2[f2it + 1]; b++) {
                    const int idx2 = feat2[b];
                    if (vbMatched2[idx2] || !flags2[idx2]) continue;
                    const int dist = descriptor_distance(d1, &Descriptors2[32 * (size_t)idx2]);
                    if (dist < bestDist1) {
                        bestDist2 = bestDist1;
                        bestDist1 = dist;
                        bestIdx2 = idx2;
                    } else if (dist < bestDist2) {
                        bestDist2 = dist;
                    }
                }
                if (bestDist1 < TH_LOW) {
                    if ((float)bestDist1 < mfNNratio * (float)bestDist2) {
                        vpMatches12[idx1] = bestIdx2;
                        vbMatched2[bestIdx2] = true;
                        if (checkOri) {
                            float rot = vKeysUn1[idx1].angle - vKeysUn2[bestIdx2].angle;
                            if (rot < 0.0) rot += 360.0f;
                            int bin = (int)std::round(rot * factor);
                            if (bin == HISTO_LENGTH) bin = 0;
                            rotHist[bin].push_back(idx1);
                        }
                        nmatches++;
                    }
                }
            }
            f1it++;
            f2it++;
        } else if (nodes1[f1it] < nodes2[f2it]) {
            f1it = std::lower_bound(nodes1.begin(), nodes1.end(), nodes2[f2it]) - nodes1.begin();
        } else {
            f2it = std::lower_bound(nodes2.begin(), nodes2.end(), nodes1[f1it]) - nodes2.begin();
        }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        compute_three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx : rotHist[i]) {
                vpMatches12[idx] = -1;
                nmatches--;
            }
        }
    }
    return nmatches;
}

/* ------------------------------------------------------------------ glibc logf */
float glibc_logf(float x) {
    static const struct { double invc, logc; } T[16] = {
        {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
        {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},  {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
        {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
        {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
        {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1p+0, 0x0p+0},
        {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
        {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3},
        {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},  {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};
    static const double Ln2 = 0x1.62e42fefa39efp-1;
    static const double A[3] = {-0x1.00ea348b88334p-2, 0x1.5575b0be00b6ap-2, -0x1.ffffef20a4123p-2};
    uint32_t ix;
    memcpy(&ix, &x, 4);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) { /* 0, inf, nan, negative, subnormal */
        if (ix * 2 == 0) return -INFINITY;
        if (ix == 0x7f800000u) return x;
        if ((ix & 0x80000000u) || ix * 2 >= 0xff000000u) return NAN;
        const float xs = x * 0x1p23f;
        memcpy(&ix, &xs, 4);
        ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) % 16);
    const int k = (int32_t)tmp >> 23;
    const uint32_t iz = ix - (tmp & (0x1ffu << 23));
    float zf;
    memcpy(&zf, &iz, 4);
    const double z = (double)zf;
    const double r = z * T[i].invc - 1;
    const double y0 = T[i].logc + (double)k * Ln2;
    const double r2 = r * r;
    double y = A[1] * r + A[2];
    y = A[0] * r2 + y;
    y = y * r2 + (y0 + r);
    return (float)y;
}

/* ------------------------------------------------------------------ FMatcher::SearchForTriangulation */
int search_for_triangulation(const std::vector<KeyPoint>& vKeysUn1, const std::vector<uint8_t>& Descriptors1,
                             const std::vector<uint8_t>& hasMp1, const std::vector<float>& mvuRight1,
                             const std::vector<int>& nodes1, const std::vector<int>& off1, const std::vector<int>& feat1,
                             const std::vector<KeyPoint>& vKeysUn2, const std::vector<uint8_t>& Descriptors2,
                             const std::vector<uint8_t>& hasMp2, const std::vector<float>& mvuRight2,
                             const std::vector<int>& nodes2, const std::vector<int>& off2, const std::vector<int>& feat2,
                             const std::vector<float>& mvScaleFactors2, const std::vector<float>& mvLevelSigma2_2,
                             const TriArgs& A, std::vector<int>& vMatches12) { /* fmatcher.cpp:1242-1482 */
    const int TH_LOW = 50, HISTO_LENGTH = 30;
    const float* F12 = A.F12;
    int nmatches = 0;
    vMatches12.assign(vKeysUn1.size(), -1);
    std::vector<int> rotHist[30];
    const float factor = 1.0f / HISTO_LENGTH;
    size_t f1it = 0, f2it = 0;
    while (f1it != nodes1.size() && f2it != nodes2.size()) {
        if (nodes1[f1it] == nodes2[f2it]) {
            for (int i1 = off1[f1it]; i1 < off1[f1it + 1]; i1++) {
                const int idx1 = feat1[i1];
                if (hasMp1[idx1]) continue; /* :1313 */
                const bool bStereo1 = mvuRight1[idx1] >= 0; /* !mpCamera2 && ... :1318 */
                if (A.onlyStereo && !bStereo1) continue;
                const KeyPoint& kp1 = vKeysUn1[idx1];
                const uint8_t* d1 = &Descriptors1[32 * (size_t)idx1];
                int bestDist = TH_LOW, bestIdx2 = -1;
                for (int i2 = off2[f2it]; i2 < off2[f2it + 1]; i2++) {
                    const int idx2 = feat2[i2];
                    if (hasMp2[idx2]) continue; /* vbMatched2 stays false throughout, :1344 */
                    const bool bStereo2 = mvuRight2[idx2] >= 0;
                    if (A.onlyStereo && !bStereo2) continue;
                    const int dist = descriptor_distance(d1, &Descriptors2[32 * (size_t)idx2]);
                    if (dist > TH_LOW || dist > bestDist) continue;
                    const KeyPoint& kp2 = vKeysUn2[idx2];
                    if (!bStereo1 && !bStereo2) { /* :1366-1374 */
                        const float distex = A.epx - kp2.x;
                        const float distey = A.epy - kp2.y;
                        if (distex * distex + distey * distey < 100 * mvScaleFactors2[kp2.octave]) continue;
                    }
                    bool ok = A.coarse != 0;
                    if (!ok) { /* Pinhole::epipolarConstrain, pinhole.cpp:128-142 */
                        const float a = kp1.x * F12[0] + kp1.y * F12[3] + F12[6];
                        const float b = kp1.x * F12[1] + kp1.y * F12[4] + F12[7];
                        const float c = kp1.x * F12[2] + kp1.y * F12[5] + F12[8];
                        const float num = a * kp2.x + b * kp2.y + c;
                        const float den = a * a + b * b;
                        if (den != 0) {
                            const float dsqr = num * num / den;
                            ok = dsqr < 3.84 * mvLevelSigma2_2[kp2.octave];
                        }
                    }
                    if (ok) {
                        bestIdx2 = idx2;
                        bestDist = dist;
                    }
                }
                if (bestIdx2 >= 0) {
                    vMatches12[idx1] = bestIdx2;
                    nmatches++;
                    if (A.checkOri) {
                        float rot = kp1.angle - vKeysUn2[bestIdx2].angle;
                        if (rot < 0.0) rot += 360.0f;
                        int bin = (int)std::round(rot * factor);
                        if (bin == HISTO_LENGTH) bin = 0;
                        rotHist[bin].push_back(idx1);
                    }
                }
            }
            f1it++;
            f2it++;
        } else if (nodes1[f1it] < nodes2[f2it]) {
            f1it = std::lower_bound(nodes1.begin(), nodes1.end(), nodes2[f2it]) - nodes1.begin();
        } else {
            f2it = std::lower_bound(nodes2.begin(), nodes2.end(), nodes1[f1it]) - nodes2.begin();
        }
    }
    if (A.checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        compute_three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx : rotHist[i]) {
                vMatches12[idx] = -1;
                nmatches--;
            }
        }
    }
    return nmatches;
}

/* ------------------------------------------------------------------ FMatcher::Fuse, search half */
void fuse_search(const std::vector<FusePoint>& pts, const std::vector<uint8_t>& mpDesc, const std::vector<KeyPoint>& kfKps,
                 const std::vector<uint8_t>& kfDesc, const std::vector<float>& mvuRight,
                 const std::vector<float>& mvScaleFactors, const std::vector<float>& mvInvLevelSigma2, const FuseArgs& A,
                 std::vector<int>& bestIdxOut, std::vector<int>& bestDistOut) { /* fmatcher.cpp:1953-2090, :2144-2222 */
    const int nMPs = (int)pts.size();
    bestIdxOut.assign(nMPs, -1);
    bestDistOut.assign(nMPs, 256);
    FrameGrid grid(kfKps, A.imgW, A.imgH); /* KeyFrame copies the Frame's grid, keyframe.cpp:40-60 */
    const int mnScaleLevels = (int)mvScaleFactors.size();
    for (int i = 0; i < nMPs; i++) {
        const FusePoint& mp = pts[i];
        if (!mp.valid) continue;
        const float* p3Dw = mp.pos;
        const float xc = gemm_row(A.Rcw + 0, p3Dw, A.tcw[0], A.gemmDouble); /* Rcw*p3Dw + tcw */
        const float yc = gemm_row(A.Rcw + 3, p3Dw, A.tcw[1], A.gemmDouble);
        const float zc = gemm_row(A.Rcw + 6, p3Dw, A.tcw[2], A.gemmDouble);
        if (zc < 0.0f) continue;
        const float invz = 1 / zc;
        const float u = A.fx * xc / zc + A.cx; /* Pinhole::project, pinhole.cpp:13-16 */
        const float v = A.fy * yc / zc + A.cy;
        if (!(u >= 0.0f && u < (float)A.imgW && v >= 0.0f && v < (float)A.imgH)) continue; /* KeyFrame::IsInImage */
        const float ur = u - A.bf * invz;
        const float PO[3] = {p3Dw[0] - A.Ow[0], p3Dw[1] - A.Ow[1], p3Dw[2] - A.Ow[2]};
        double n2 = 0; /* cv::norm(NORM_L2) on CV_32F accumulates in double */
        for (int k = 0; k < 3; k++) n2 += (double)PO[k] * (double)PO[k];
        const float dist3D = (float)std::sqrt(n2);
        if (dist3D < mp.minDistance || dist3D > mp.maxDistance) continue;
        double dot = 0; /* cv::Mat::dot on CV_32F accumulates in double */
        for (int k = 0; k < 3; k++) dot += (double)PO[k] * (double)mp.normal[k];
        if (dot < 0.5 * dist3D) continue;
        const float ratio = mp.maxDistance / dist3D; /* MapPoint::PredictScale, mappoint.cpp:506-521 */
        /* `int nScale = ceil(...)`: out-of-range and NaN conversions are what cvttss2si returns on the reference's
         * x86-64 build (INT_MIN), spelled out here instead of relying on undefined behaviour */
        const float lv = std::ceil(glibc_logf(ratio) / A.logScaleFactor);
        int nPredictedLevel = (lv != lv || lv >= 2147483648.0f || lv < -2147483648.0f) ? INT_MIN : (int)lv;
        if (nPredictedLevel < 0) nPredictedLevel = 0;
        else if (nPredictedLevel >= mnScaleLevels) nPredictedLevel = mnScaleLevels - 1;
        const float radius = A.th * mvScaleFactors[nPredictedLevel];
        const std::vector<int> vIndices = grid.GetFeaturesInArea(u, v, radius, -1, -1);
        if (vIndices.empty()) continue;
        const uint8_t* dMP = &mpDesc[32 * (size_t)i];
        int bestDist = A.sim3 ? INT_MAX : 256, bestIdx = -1;
        for (int idx : vIndices) {
            const KeyPoint& kp = kfKps[idx];
            const int kpLevel = kp.octave;
            if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
            if (!A.sim3) {
                if (mvuRight[idx] >= 0) { /* :2053-2066 */
                    const float ex = u - kp.x, ey = v - kp.y, er = ur - mvuRight[idx];
                    const float e2 = ex * ex + ey * ey + er * er;
                    if (e2 * mvInvLevelSigma2[kpLevel] > 7.8) continue;
                } else {
                    const float ex = u - kp.x, ey = v - kp.y;
                    const float e2 = ex * ex + ey * ey;
                    if (e2 * mvInvLevelSigma2[kpLevel] > 5.99) continue;
                }
            }
            const int dist = descriptor_distance(dMP, &kfDesc[32 * (size_t)idx]);
            if (dist < bestDist) {
                bestDist = dist;
                bestIdx = idx;
            }
        }
        if (bestIdx >= 0) {
            bestIdxOut[i] = bestIdx;
            bestDistOut[i] = bestDist;
        }
    }
}

/* ------------------------------------------------------------------ SearchByProjection(CurrentFrame, pKF, ...) */
int search_by_projection_keyframe(const float Tcw[12], const float Ow[3], float fx, float fy, float cx, float cy, float th,
                                  int ORBdist, float mfLogScaleFactor, bool checkOri, int imgW, int imgH, int gemmDouble,
                                  const std::vector<KeyPoint>& mvKeysUn, const std::vector<uint8_t>& flags,
                                  const std::vector<float>& x3Dws, const std::vector<float>& minDist,
                                  const std::vector<float>& maxDist, const std::vector<uint8_t>& mpDesc,
                                  const std::vector<KeyPoint>& curKps, const std::vector<uint8_t>& curDesc,
                                  const std::vector<uint8_t>& occupied0, const std::vector<float>& mvScaleFactors,
                                  std::vector<int>& matchCur) { /* fmatcher.cpp:2689-2811 */
    const int HISTO_LENGTH = 30;
    int nmatches = 0;
    const int N2 = (int)curKps.size();
    matchCur.assign(N2, -1);
    std::vector<uint8_t> has(N2, 0); /* CurrentFrame.mvpMapPoints[i2] != NULL */
    for (int i = 0; i < N2 && i < (int)occupied0.size(); i++) has[i] = occupied0[i];
    std::vector<int> rotHist[30];
    const float factor = 1.0f / HISTO_LENGTH;
    FrameGrid grid(curKps, imgW, imgH);
    const int mnScaleLevels = (int)mvScaleFactors.size();
    const float mnMinX = 0.0f, mnMaxX = (float)imgW, mnMinY = 0.0f, mnMaxY = (float)imgH;
    for (size_t i = 0; i < mvKeysUn.size(); i++) {
        if (!(flags[i] & 1)) continue;
        const float* x3Dw = &x3Dws[3 * i];
        const float xc = gemm_row(Tcw + 0, x3Dw, Tcw[3], gemmDouble);
        const float yc = gemm_row(Tcw + 4, x3Dw, Tcw[7], gemmDouble);
        const float zc = gemm_row(Tcw + 8, x3Dw, Tcw[11], gemmDouble);
        const float u = fx * xc / zc + cx; /* mpCamera->project(x3Dc): no depth test in this overload */
        const float v = fy * yc / zc + cy;
        if (u < mnMinX || u > mnMaxX) continue;
        if (v < mnMinY || v > mnMaxY) continue;
        const float PO[3] = {x3Dw[0] - Ow[0], x3Dw[1] - Ow[1], x3Dw[2] - Ow[2]};
        double n2 = 0;
        for (int k = 0; k < 3; k++) n2 += (double)PO[k] * (double)PO[k];
        const float dist3D = (float)std::sqrt(n2); /* cv::norm */
        if (dist3D < minDist[i] || dist3D > maxDist[i]) continue;
        const float ratio = maxDist[i] / dist3D; /* MapPoint::PredictScale(dist3D, &CurrentFrame), mappoint.cpp:523-538 */
        const float lv = std::ceil(glibc_logf(ratio) / mfLogScaleFactor);
        int nPredictedLevel = (lv != lv || lv >= 2147483648.0f || lv < -2147483648.0f) ? INT_MIN : (int)lv;
        if (nPredictedLevel < 0) nPredictedLevel = 0;
        else if (nPredictedLevel >= mnScaleLevels) nPredictedLevel = mnScaleLevels - 1;
        const float radius = th * mvScaleFactors[nPredictedLevel];
        const std::vector<int> vIndices2 = grid.GetFeaturesInArea(u, v, radius, nPredictedLevel - 1, nPredictedLevel + 1);
        if (vIndices2.empty()) continue;
        const uint8_t* dMP = &mpDesc[32 * i];
        int bestDist = 256, bestIdx2 = -1;
        for (int i2 : vIndices2) {
            if (has[i2]) continue;
            const int dist = descriptor_distance(dMP, &curDesc[32 * (size_t)i2]);
            if (dist < bestDist) {
                bestDist = dist;
                bestIdx2 = i2;
            }
        }
        if (bestDist <= ORBdist) {
            matchCur[bestIdx2] = (int)i;
            has[bestIdx2] = 1;
            nmatches++;
            if (checkOri) {
                float rot = mvKeysUn[i].angle - curKps[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(bestIdx2);
            }
        }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        compute_three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int idx : rotHist[i]) {
                    matchCur[idx] = -1;
                    nmatches--;
                }
    }
    return nmatches;
}

/* ------------------------------------------------------------------ SearchByProjection(pKF, Scw, ...) x2 */
int search_by_projection_sim3(const float Tcw[12], const float Ow[3], float fx, float fy, float cx, float cy, int th,
                              float ratioHamming, float mfLogScaleFactor, int projVariant, int imgW, int imgH, int gemmDouble,
                              const std::vector<uint8_t>& flags, const std::vector<float>& x3Dws,
                              const std::vector<float>& normals, const std::vector<float>& minDist,
                              const std::vector<float>& maxDist, const std::vector<uint8_t>& mpDesc,
                              const std::vector<KeyPoint>& mvKeysUn, const std::vector<uint8_t>& mDescriptors,
                              const std::vector<uint8_t>& matched0, const std::vector<float>& mvScaleFactors,
                              std::vector<int>& matchKf) { /* fmatcher.cpp:750-863, :865-981 */
    const int TH_LOW = 50;
    int nmatches = 0;
    const int N = (int)mvKeysUn.size();
    matchKf.assign(N, -1);
    std::vector<uint8_t> vpMatched(N, 0);
    for (int i = 0; i < N && i < (int)matched0.size(); i++) vpMatched[i] = matched0[i];
    FrameGrid grid(mvKeysUn, imgW, imgH);
    const int mnScaleLevels = (int)mvScaleFactors.size();
    for (size_t iMP = 0; iMP < flags.size(); iMP++) {
        if (!(flags[iMP] & 1)) continue;
        const float* p3Dw = &x3Dws[3 * iMP];
        const float xc = gemm_row(Tcw + 0, p3Dw, Tcw[3], gemmDouble);
        const float yc = gemm_row(Tcw + 4, p3Dw, Tcw[7], gemmDouble);
        const float zc = gemm_row(Tcw + 8, p3Dw, Tcw[11], gemmDouble);
        if (zc < 0.0) continue;
        float u, v;
        if (projVariant == 0) { /* pKF->mpCamera->project(cv::Point3f(x, y, z)) */
            u = fx * xc / zc + cx;
            v = fy * yc / zc + cy;
        } else { /* :908-913 */
            const float invz = 1 / zc;
            const float x = xc * invz, y = yc * invz;
            u = fx * x + cx;
            v = fy * y + cy;
        }
        if (!(u >= 0.0f && u < (float)imgW && v >= 0.0f && v < (float)imgH)) continue; /* KeyFrame::IsInImage */
        const float PO[3] = {p3Dw[0] - Ow[0], p3Dw[1] - Ow[1], p3Dw[2] - Ow[2]};
        double n2 = 0;
        for (int k = 0; k < 3; k++) n2 += (double)PO[k] * (double)PO[k];
        const float dist = (float)std::sqrt(n2);
        if (dist < minDist[iMP] || dist > maxDist[iMP]) continue;
        double dot = 0;
        for (int k = 0; k < 3; k++) dot += (double)PO[k] * (double)normals[3 * iMP + k];
        if (dot < 0.5 * dist) continue;
        const float ratio = maxDist[iMP] / dist;
        const float lv = std::ceil(glibc_logf(ratio) / mfLogScaleFactor);
        int nPredictedLevel = (lv != lv || lv >= 2147483648.0f || lv < -2147483648.0f) ? INT_MIN : (int)lv;
        if (nPredictedLevel < 0) nPredictedLevel = 0;
        else if (nPredictedLevel >= mnScaleLevels) nPredictedLevel = mnScaleLevels - 1;
        const float radius = th * mvScaleFactors[nPredictedLevel];
        const std::vector<int> vIndices = grid.GetFeaturesInArea(u, v, radius, -1, -1);
        if (vIndices.empty()) continue;
        const uint8_t* dMP = &mpDesc[32 * iMP];
        int bestDist = 256, bestIdx = -1;
        for (int idx : vIndices) {
            if (vpMatched[idx]) continue;
            const int kpLevel = mvKeysUn[idx].octave;
            if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
            const int d = descriptor_distance(dMP, &mDescriptors[32 * (size_t)idx]);
            if (d < bestDist) {
                bestDist = d;
                bestIdx = idx;
            }
        }
        if (bestDist <= TH_LOW * ratioHamming) {
            vpMatched[bestIdx] = 1;
            matchKf[bestIdx] = (int)iMP;
            nmatches++;
        }
    }
    return nmatches;
}

/* ------------------------------------------------------------------ SearchBySim3, one direction */
void search_by_sim3_direction(const float Ra[9], const float ta[3], const float Rb[9], const float tb[3], float fx, float fy,
                              float cx, float cy, float th, float mfLogScaleFactor, int imgW, int imgH, int gemmDouble,
                              const std::vector<uint8_t>& valid, const std::vector<float>& x3Dws,
                              const std::vector<float>& minDist, const std::vector<float>& maxDist,
                              const std::vector<uint8_t>& mpDesc, const std::vector<KeyPoint>& mvKeysUn,
                              const std::vector<uint8_t>& mDescriptors, const std::vector<float>& mvScaleFactors,
                              std::vector<int>& vnMatch) { /* fmatcher.cpp:2291-2368 */
    const int TH_HIGH = 100;
    vnMatch.assign(valid.size(), -1);
    FrameGrid grid(mvKeysUn, imgW, imgH);
    const int mnScaleLevels = (int)mvScaleFactors.size();
    for (size_t i1 = 0; i1 < valid.size(); i1++) {
        if (!valid[i1]) continue;
        const float* p3Dw = &x3Dws[3 * i1];
        const float c1[3] = {gemm_row(Ra + 0, p3Dw, ta[0], gemmDouble), gemm_row(Ra + 3, p3Dw, ta[1], gemmDouble),
                             gemm_row(Ra + 6, p3Dw, ta[2], gemmDouble)}; /* p3Dc1 = R1w*p3Dw + t1w */
        const float c2[3] = {gemm_row(Rb + 0, c1, tb[0], gemmDouble), gemm_row(Rb + 3, c1, tb[1], gemmDouble),
                             gemm_row(Rb + 6, c1, tb[2], gemmDouble)}; /* p3Dc2 = sR21*p3Dc1 + t21 */
        if (c2[2] < 0.0) continue;
        const float invz = 1.0 / c2[2];
        const float x = c2[0] * invz, y = c2[1] * invz;
        const float u = fx * x + cx, v = fy * y + cy;
        if (!(u >= 0.0f && u < (float)imgW && v >= 0.0f && v < (float)imgH)) continue;
        double n2 = 0;
        for (int k = 0; k < 3; k++) n2 += (double)c2[k] * (double)c2[k];
        const float dist3D = (float)std::sqrt(n2); /* cv::norm(p3Dc2) */
        if (dist3D < minDist[i1] || dist3D > maxDist[i1]) continue;
        const float lv = std::ceil(glibc_logf(maxDist[i1] / dist3D) / mfLogScaleFactor);
        int nPredictedLevel = (lv != lv || lv >= 2147483648.0f || lv < -2147483648.0f) ? INT_MIN : (int)lv;
        if (nPredictedLevel < 0) nPredictedLevel = 0;
        else if (nPredictedLevel >= mnScaleLevels) nPredictedLevel = mnScaleLevels - 1;
        const float radius = th * mvScaleFactors[nPredictedLevel];
        const std::vector<int> vIndices = grid.GetFeaturesInArea(u, v, radius, -1, -1);
        if (vIndices.empty()) continue;
        const uint8_t* dMP = &mpDesc[32 * i1];
        int bestDist = INT_MAX, bestIdx = -1;
        for (int idx : vIndices) {
            const KeyPoint& kp = mvKeysUn[idx];
            if (kp.octave < nPredictedLevel - 1 || kp.octave > nPredictedLevel) continue;
            const int dist = descriptor_distance(dMP, &mDescriptors[32 * (size_t)idx]);
            if (dist < bestDist) {
                bestDist = dist;
                bestIdx = idx;
            }
        }
        if (bestDist <= TH_HIGH) vnMatch[i1] = bestIdx;
    }
}

bool unproject_stereo(const KeyPoint& kpUn, float z, const float Twc[12], float cx, float cy, float invfx, float invfy,
                      int gemmDouble, float out[3]) { /* frame.cpp:1023-1037 */
    if (!(z > 0)) return false;
    const float u = kpUn.x, v = kpUn.y;
    const float x = (u - cx) * z * invfx;
    const float y = (v - cy) * z * invfy;
    const float x3Dc[3] = {x, y, z};
    for (int r = 0; r < 3; r++) out[r] = gemm_row(Twc + 4 * r, x3Dc, Twc[4 * r + 3], gemmDouble); /* mRwc*x3Dc+mOw */
    return true;
}

void projection_direction(const ProjFrameArgs& a, bool& bForward, bool& bBackward) {
    /* twc = -Rcw.t()*tcw (gemm with alpha = -1, no C); tlc = Rlw*twc + tlw; fmatcher.cpp:2482-2495 */
    float twc[3], tlc[3];
    for (int r = 0; r < 3; r++) {
        const float col[3] = {a.Tcw[0 * 4 + r], a.Tcw[1 * 4 + r], a.Tcw[2 * 4 + r]};
        const float t[3] = {a.Tcw[3], a.Tcw[7], a.Tcw[11]};
        if (a.gemmDouble) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += (double)col[k] * (double)t[k];
            twc[r] = (float)(s * -1.0);
        } else {
            float s = 0;
            for (int k = 0; k < 3; k++) s += col[k] * t[k];
            twc[r] = -s;
        }
    }
    for (int r = 0; r < 3; r++) tlc[r] = gemm_row(a.Tlw + 4 * r, twc, a.Tlw[4 * r + 3], a.gemmDouble);
    bForward = tlc[2] > a.mb && !a.bMono;
    bBackward = -tlc[2] > a.mb && !a.bMono;
}

int search_by_projection_frame(const ProjFrameArgs& a, const std::vector<KeyPoint>& lastKps,
                               const std::vector<uint8_t>& flags, const std::vector<float>& x3Dw,
                               const std::vector<uint8_t>& mpDesc, const std::vector<KeyPoint>& curKps,
                               const std::vector<uint8_t>& curDesc, const std::vector<float>& mvuRight,
                               const std::vector<uint8_t>& occupied0, const std::vector<float>& scaleFactors,
                               std::vector<int>& matchCur) { /* fmatcher.cpp:2471-2687, Nleft == -1 */
    const int TH_HIGH = 100, HISTO_LENGTH = 30;
    int nmatches = 0;
    std::vector<int> rotHist[30];
    const float factor = 1.0f / HISTO_LENGTH;
    bool bForward, bBackward;
    projection_direction(a, bForward, bBackward);
    const int N = (int)lastKps.size(), N2 = (int)curKps.size();
    matchCur.assign(N2, -1);
    std::vector<uint8_t> occ(N2, 0); /* mvpMapPoints[i2] != NULL && Observations() > 0 */
    for (int i = 0; i < N2 && i < (int)occupied0.size(); i++) occ[i] = occupied0[i];
    FrameGrid grid(curKps, a.imgW, a.imgH);
    const float mnMinX = 0.f, mnMinY = 0.f, mnMaxX = (float)a.imgW, mnMaxY = (float)a.imgH;
    for (int i = 0; i < N; i++) {
        if (!(flags[i] & 1)) continue; /* pMP && !mvbOutlier[i] */
        const float* Xw = &x3Dw[3 * (size_t)i];
        float x3Dc[3];
        for (int r = 0; r < 3; r++) x3Dc[r] = gemm_row(a.Tcw + 4 * r, Xw, a.Tcw[4 * r + 3], a.gemmDouble);
        const float invzc = (float)(1.0 / (double)x3Dc[2]); /* const float invzc = 1.0/x3Dc.at<float>(2) */
        if (invzc < 0) continue;
        /* Pinhole::project(cv::Point3f), pinhole.cpp:13-16 */
        const float u = a.fx * x3Dc[0] / x3Dc[2] + a.cx;
        const float v = a.fy * x3Dc[1] / x3Dc[2] + a.cy;
        if (u < mnMinX || u > mnMaxX) continue;
        if (v < mnMinY || v > mnMaxY) continue;
        const int nLastOctave = lastKps[i].octave;
        const float radius = a.th * scaleFactors[nLastOctave];
        std::vector<int> vIndices2;
        if (bForward) vIndices2 = grid.GetFeaturesInArea(u, v, radius, nLastOctave, -1);
        else if (bBackward) vIndices2 = grid.GetFeaturesInArea(u, v, radius, 0, nLastOctave);
        else vIndices2 = grid.GetFeaturesInArea(u, v, radius, nLastOctave - 1, nLastOctave + 1);
        if (vIndices2.empty()) continue;
        const uint8_t* dMP = &mpDesc[32 * (size_t)i];
        int bestDist = 256, bestIdx2 = -1;
        for (int i2 : vIndices2) {
            if (occ[i2]) continue;
            if (mvuRight[i2] > 0) {
                const float ur = u - a.mbf * invzc;
                const float er = std::fabs(ur - mvuRight[i2]);
                if (er > radius) continue;
            }
            const int dist = descriptor_distance(dMP, &curDesc[32 * (size_t)i2]);
            if (dist < bestDist) {
                bestDist = dist;
                bestIdx2 = i2;
            }
        }
        if (bestDist <= TH_HIGH) {
            matchCur[bestIdx2] = i;
            if (flags[i] & 2) occ[bestIdx2] = 1;
            nmatches++;
            if (a.checkOri) {
                float rot = lastKps[i].angle - curKps[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(bestIdx2);
            }
        }
    }
    if (a.checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        compute_three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int b = 0; b < HISTO_LENGTH; b++)
            if (b != ind1 && b != ind2 && b != ind3)
                for (int idx : rotHist[b]) {
                    matchCur[idx] = -1;
                    nmatches--;
                }
    }
    return nmatches;
}

} // namespace orbo
