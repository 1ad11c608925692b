/* oracle/orb_oracle_c.cpp -- TEST INFRASTRUCTURE ONLY.  Flat C bindings of orb_oracle.h for ctypes. */
#include <cmath>
#include <cstring>

#include "orb_oracle.h"

using namespace orbo;

extern "C" {

void* orbo_create(int nfeatures, float scale, int nlevels, int iniTh, int minTh, const int* taps7,
                  int atan_fma) {
    Knobs k;
    if (taps7) memcpy(k.gauss_taps, taps7, sizeof(k.gauss_taps));
    k.atan_fma = atan_fma;
    return new Extractor(nfeatures, scale, nlevels, iniTh, minTh, k);
}
void orbo_destroy(void* h) { delete (Extractor*)h; }

int orbo_tables(void* h, float* sf, float* isf, float* s2, float* is2, int* quota, int* umax16) {
    Extractor* e = (Extractor*)h;
    for (int i = 0; i < e->nlevels; i++) {
        sf[i] = e->mvScaleFactor[i];
        isf[i] = e->mvInvScaleFactor[i];
        s2[i] = e->mvLevelSigma2[i];
        is2[i] = e->mvInvLevelSigma2[i];
        quota[i] = e->mnFeaturesPerLevel[i];
    }
    for (int i = 0; i < 16; i++) umax16[i] = e->umax[i];
    return e->nlevels;
}

/* returns monoIndex (or -1); *n_out = number of keypoints; fails with -2 if cap is too small */
int orbo_compute(void* h, const uint8_t* img, int w, int hh, size_t stride, int lap0, int lap1,
                 KeyPoint* kps, uint8_t* desc, int cap, int* n_out) {
    Extractor* e = (Extractor*)h;
    std::vector<KeyPoint> k;
    std::vector<uint8_t> d;
    int mono = e->compute(img, w, hh, stride, lap0, lap1, k, d);
    if (mono < 0) { *n_out = 0; return mono; }
    *n_out = (int)k.size();
    if ((int)k.size() > cap) return -2;
    if (!k.empty()) {
        memcpy(kps, k.data(), k.size() * sizeof(KeyPoint));
        memcpy(desc, d.data(), d.size());
    }
    return mono;
}

int orbo_pyramid_only(void* h, const uint8_t* img, int w, int hh, size_t stride) {
    ((Extractor*)h)->ComputePyramid(img, w, hh, stride);
    return 0;
}

int orbo_level_size(void* h, int level, int* w, int* hh) {
    Extractor* e = (Extractor*)h;
    if (level < 0 || level >= e->nlevels) return -1;
    *w = e->mvImagePyramid[level].w;
    *hh = e->mvImagePyramid[level].h;
    return 0;
}
int orbo_level_copy(void* h, int level, int blurred, uint8_t* dst) {
    Extractor* e = (Extractor*)h;
    if (level < 0 || level >= e->nlevels) return -1;
    const Image& im = blurred ? e->mvBlurred[level] : e->mvImagePyramid[level];
    if (im.px.empty()) return -2;
    memcpy(dst, im.px.data(), im.px.size());
    return 0;
}
int orbo_candidates(void* h, int level, KeyPoint* out, int cap) {
    Extractor* e = (Extractor*)h;
    const std::vector<KeyPoint>& v = e->vToDistribute[level];
    if (out && (int)v.size() <= cap && !v.empty()) memcpy(out, v.data(), v.size() * sizeof(KeyPoint));
    return (int)v.size();
}
int orbo_level_keys(void* h, int level, KeyPoint* out, int cap) {
    Extractor* e = (Extractor*)h;
    const std::vector<KeyPoint>& v = e->allKeypoints[level];
    if (out && (int)v.size() <= cap && !v.empty()) memcpy(out, v.data(), v.size() * sizeof(KeyPoint));
    return (int)v.size();
}

/* ---- primitives ---- */
void orbo_resize(const uint8_t* src, int sw, int sh, size_t sstride, uint8_t* dst, int dw, int dh,
                 size_t dstride) {
    resize_linear_u8(src, sw, sh, sstride, dst, dw, dh, dstride);
}
int orbo_fast_detect(const uint8_t* img, int w, int h, size_t stride, int th, int nonmax, KeyPoint* out,
                     int cap) {
    std::vector<KeyPoint> v;
    fast_detect(img, w, h, stride, th, nonmax != 0, v);
    if (out && (int)v.size() <= cap && !v.empty()) memcpy(out, v.data(), v.size() * sizeof(KeyPoint));
    return (int)v.size();
}
void orbo_blur7(const uint8_t* src, int w, int h, uint8_t* dst, const int* taps7) {
    Image s, d;
    s.w = w;
    s.h = h;
    s.px.assign(src, src + (size_t)w * h);
    Knobs k;
    gaussian_blur7(s, d, taps7 ? taps7 : k.gauss_taps);
    memcpy(dst, d.px.data(), d.px.size());
}
float orbo_fast_atan2(float y, float x, int fma) { return fast_atan2(y, x, fma); }
int orbo_descriptor_distance(const uint8_t* a, const uint8_t* b) { return descriptor_distance(a, b); }
float orbo_cosf(float x) { return cosf(x); } /* the host libm the reference would call */
float orbo_sinf(float x) { return sinf(x); }
int orbo_cv_round_f(float v) { return cv_round_f(v); }

int orbo_distribute_octree(const KeyPoint* keys, int n, int minX, int maxX, int minY, int maxY, int N,
                           KeyPoint* out, int cap) {
    std::vector<KeyPoint> v(keys, keys + n);
    std::vector<KeyPoint> r = distribute_octree(v, minX, maxX, minY, maxY, N);
    if (out && (int)r.size() <= cap && !r.empty()) memcpy(out, r.data(), r.size() * sizeof(KeyPoint));
    return (int)r.size();
}

/* all-pairs distance matrix, for matcher parity tests */
void orbo_hamming_matrix(const uint8_t* a, int na, const uint8_t* b, int nb, uint16_t* out) {
    for (int i = 0; i < na; i++)
        for (int j = 0; j < nb; j++)
            out[(size_t)i * nb + j] = (uint16_t)descriptor_distance(a + (size_t)i * 32, b + (size_t)j * 32);
}

void orbo_stereo(void* hL, void* hR, const KeyPoint* kpsL, int nL, const uint8_t* descL,
                 const KeyPoint* kpsR, int nR, const uint8_t* descR, float bf, float fx, float* uRight,
                 float* depth, int* bestIdxR, int* bestSad) {
    std::vector<KeyPoint> kl(kpsL, kpsL + nL), kr(kpsR, kpsR + nR);
    std::vector<uint8_t> dl(descL, descL + (size_t)nL * 32), dr(descR, descR + (size_t)nR * 32);
    std::vector<float> u, d;
    std::vector<int> bi, bs;
    compute_stereo_matches(*(Extractor*)hL, *(Extractor*)hR, kl, dl, kr, dr, bf, fx, u, d, &bi, &bs);
    if (nL) {
        memcpy(uRight, u.data(), nL * sizeof(float));
        memcpy(depth, d.data(), nL * sizeof(float));
        if (bestIdxR) memcpy(bestIdxR, bi.data(), nL * sizeof(int));
        if (bestSad) memcpy(bestSad, bs.data(), nL * sizeof(int));
    }
}

int orbo_search_init(const KeyPoint* kps1, int n1, const uint8_t* desc1, const KeyPoint* kps2, int n2,
                     const uint8_t* desc2, int imgW, int imgH, float* prevMatchedXY, int* matches12,
                     int windowSize, float nnratio, int checkOri) {
    std::vector<KeyPoint> k1(kps1, kps1 + n1), k2(kps2, kps2 + n2);
    std::vector<uint8_t> d1(desc1, desc1 + (size_t)n1 * 32), d2(desc2, desc2 + (size_t)n2 * 32);
    std::vector<float> pm(prevMatchedXY, prevMatchedXY + 2 * (size_t)n1);
    std::vector<int> m;
    int nm = search_for_initialization(k1, d1, k2, d2, imgW, imgH, pm, m, windowSize, nnratio,
                                       checkOri != 0);
    if (n1) {
        memcpy(prevMatchedXY, pm.data(), pm.size() * sizeof(float));
        memcpy(matches12, m.data(), m.size() * sizeof(int));
    }
    return nm;
}

int orbo_grid_query(const KeyPoint* kps, int n, int imgW, int imgH, float x, float y, float r, int minL,
                    int maxL, int* out, int cap) {
    std::vector<KeyPoint> k(kps, kps + n);
    FrameGrid g(k, imgW, imgH);
    std::vector<int> v = g.GetFeaturesInArea(x, y, r, minL, maxL);
    if (out && (int)v.size() <= cap && !v.empty()) memcpy(out, v.data(), v.size() * sizeof(int));
    return (int)v.size();
}

/* args: 12 Tcw, 12 Tlw, fx, fy, cx, cy, mbf, mb, th (31 floats); iargs: bMono, checkOri, imgW, imgH, gemmDouble */
int orbo_search_by_projection_frame(const float* fargs, const int* iargs, const KeyPoint* lastKps, int n,
                                    const uint8_t* flags, const float* x3Dw, const uint8_t* mpDesc,
                                    const KeyPoint* curKps, int n2, const uint8_t* curDesc, const float* mvuRight,
                                    const uint8_t* occupied, const float* scaleFactors, int nlevels, int* matchCur,
                                    int* dir_out) {
    ProjFrameArgs a;
    memcpy(a.Tcw, fargs, 48);
    memcpy(a.Tlw, fargs + 12, 48);
    a.fx = fargs[24]; a.fy = fargs[25]; a.cx = fargs[26]; a.cy = fargs[27]; a.mbf = fargs[28]; a.mb = fargs[29];
    a.th = fargs[30];
    a.bMono = iargs[0]; a.checkOri = iargs[1]; a.imgW = iargs[2]; a.imgH = iargs[3]; a.gemmDouble = iargs[4];
    std::vector<KeyPoint> lk(lastKps, lastKps + n), ck(curKps, curKps + n2);
    std::vector<uint8_t> fl(flags, flags + n), md(mpDesc, mpDesc + (size_t)n * 32), cd(curDesc, curDesc + (size_t)n2 * 32);
    std::vector<float> xw(x3Dw, x3Dw + (size_t)n * 3), ur(mvuRight, mvuRight + n2), sf(scaleFactors, scaleFactors + nlevels);
    std::vector<uint8_t> occ;
    if (occupied) occ.assign(occupied, occupied + n2);
    std::vector<int> m;
    const int nm = search_by_projection_frame(a, lk, fl, xw, md, ck, cd, ur, occ, sf, m);
    if (n2) memcpy(matchCur, m.data(), (size_t)n2 * sizeof(int));
    if (dir_out) {
        bool f, b;
        projection_direction(a, f, b);
        dir_out[0] = f;
        dir_out[1] = b;
    }
    return nm;
}

int orbo_search_by_projection_mappoints(const MapPointTrack* mps, int n, const uint8_t* mpDesc, const KeyPoint* curKps,
                                        int n2, const uint8_t* curDesc, const float* mvuRight, const uint8_t* occupied,
                                        const float* scaleFactors, int nlevels, int imgW, int imgH, float th,
                                        float nnratio, int* matchCur) {
    std::vector<MapPointTrack> m(mps, mps + n);
    std::vector<KeyPoint> ck(curKps, curKps + n2);
    std::vector<uint8_t> md(mpDesc, mpDesc + (size_t)n * 32), cd(curDesc, curDesc + (size_t)n2 * 32), occ;
    std::vector<float> ur(mvuRight, mvuRight + n2), sf(scaleFactors, scaleFactors + nlevels);
    if (occupied) occ.assign(occupied, occupied + n2);
    std::vector<int> out;
    const int nm = search_by_projection_mappoints(m, md, ck, cd, ur, occ, sf, imgW, imgH, th, nnratio, out);
    if (n2) memcpy(matchCur, out.data(), (size_t)n2 * sizeof(int));
    return nm;
}

void orbo_distinctive_descriptors(const uint8_t* desc, const int* offsets, int nsets, int* best) {
    for (int s = 0; s < nsets; s++) best[s] = distinctive_descriptor(desc + 32 * (size_t)offsets[s], offsets[s + 1] - offsets[s]);
}

/* vocabulary as flat arrays; outputs: per-feature word/weight/nid and the assembled vectors (caller arrays of n) */
int orbo_bow_transform(int L, int weighting, int norm, int nNodes, const int* childStart, const int* childCount,
                       const int* childIds, int nChildIds, const uint8_t* nodeDesc, const double* nodeWeight,
                       const int* nodeWord, const uint8_t* desc, int n, int levelsup, int* fWord, double* fWeight,
                       int* fNid, int* bowIds, double* bowVals, int* nBow, int* fvNodes, int* fvOff, int* fvFeat,
                       int* nFv) {
    Vocabulary v;
    v.L = L; v.weighting = weighting; v.norm = norm;
    v.childStart.assign(childStart, childStart + nNodes);
    v.childCount.assign(childCount, childCount + nNodes);
    v.childIds.assign(childIds, childIds + nChildIds);
    v.wordId.assign(nodeWord, nodeWord + nNodes);
    v.desc.assign(nodeDesc, nodeDesc + (size_t)nNodes * 32);
    v.weight.assign(nodeWeight, nodeWeight + nNodes);
    for (int i = 0; i < n; i++) bow_transform_feature(v, desc + 32 * (size_t)i, levelsup, fWord[i], fWeight[i], fNid[i]);
    BowResult r;
    bow_transform(v, desc, n, levelsup, r);
    *nBow = (int)r.words.size();
    for (size_t i = 0; i < r.words.size(); i++) { bowIds[i] = r.words[i]; bowVals[i] = r.values[i]; }
    *nFv = (int)r.nodes.size();
    int off = 0;
    for (size_t i = 0; i < r.nodes.size(); i++) {
        fvNodes[i] = r.nodes[i];
        fvOff[i] = off;
        for (unsigned f : r.features[i]) fvFeat[off++] = (int)f;
    }
    fvOff[r.nodes.size()] = off;
    return 0;
}

int orbo_search_by_bow(const KeyPoint* kfKps, int nKF, const uint8_t* kfDesc, const uint8_t* kfFlags, const int* kfNodes,
                       const int* kfOff, const int* kfFeat, int nKFnodes, const KeyPoint* fKps, int nF,
                       const uint8_t* fDesc, const int* fNodes, const int* fOff, const int* fFeat, int nFnodes,
                       float nnratio, int checkOri, int* matchF) {
    std::vector<KeyPoint> kk(kfKps, kfKps + nKF), fk(fKps, fKps + nF);
    std::vector<uint8_t> kd(kfDesc, kfDesc + (size_t)nKF * 32), fd(fDesc, fDesc + (size_t)nF * 32), kfl(kfFlags, kfFlags + nKF);
    std::vector<int> kn(kfNodes, kfNodes + nKFnodes), ko(kfOff, kfOff + nKFnodes + 1), kf(kfFeat, kfFeat + kfOff[nKFnodes]);
    std::vector<int> fn(fNodes, fNodes + nFnodes), fo(fOff, fOff + nFnodes + 1), ff(fFeat, fFeat + fOff[nFnodes]);
    std::vector<int> m;
    const int nm = search_by_bow(kk, kd, kfl, kn, ko, kf, fk, fd, fn, fo, ff, nnratio, checkOri != 0, m);
    if (nF) memcpy(matchF, m.data(), (size_t)nF * sizeof(int));
    return nm;
}

int orbo_search_by_bow_keyframes(const KeyPoint* kps1, int n1, const uint8_t* desc1, const uint8_t* flags1, const int* nodes1,
                                 const int* off1, const int* feat1, int nn1, const KeyPoint* kps2, int n2,
                                 const uint8_t* desc2, const uint8_t* flags2, const int* nodes2, const int* off2,
                                 const int* feat2, int nn2, float nnratio, int checkOri, int* match12) {
    std::vector<KeyPoint> k1(kps1, kps1 + n1), k2(kps2, kps2 + n2);
    std::vector<uint8_t> d1(desc1, desc1 + (size_t)n1 * 32), d2(desc2, desc2 + (size_t)n2 * 32), f1(flags1, flags1 + n1),
        f2(flags2, flags2 + n2);
    std::vector<int> a1(nodes1, nodes1 + nn1), o1(off1, off1 + nn1 + 1), e1(feat1, feat1 + off1[nn1]);
    std::vector<int> a2(nodes2, nodes2 + nn2), o2(off2, off2 + nn2 + 1), e2(feat2, feat2 + off2[nn2]);
    std::vector<int> m;
    const int nm = search_by_bow_keyframes(k1, d1, f1, a1, o1, e1, k2, d2, f2, a2, o2, e2, nnratio, checkOri != 0, m);
    if (n1) memcpy(match12, m.data(), (size_t)n1 * sizeof(int));
    return nm;
}

float orbo_logf(float x) { return glibc_logf(x); }

int orbo_search_for_triangulation(const KeyPoint* kps1, int n1, const uint8_t* desc1, const uint8_t* hasMp1,
                                  const float* uRight1, const int* nodes1, const int* off1, const int* feat1, int nn1,
                                  const KeyPoint* kps2, int n2, const uint8_t* desc2, const uint8_t* hasMp2,
                                  const float* uRight2, const int* nodes2, const int* off2, const int* feat2, int nn2,
                                  const float* scaleFactors2, const float* levelSigma2_2, int nlevels, const float* F12,
                                  float epx, float epy, int onlyStereo, int coarse, int checkOri, int* match12) {
    std::vector<KeyPoint> k1(kps1, kps1 + n1), k2(kps2, kps2 + n2);
    std::vector<uint8_t> d1(desc1, desc1 + (size_t)n1 * 32), d2(desc2, desc2 + (size_t)n2 * 32), f1(hasMp1, hasMp1 + n1),
        f2(hasMp2, hasMp2 + n2);
    std::vector<float> u1(uRight1, uRight1 + n1), u2(uRight2, uRight2 + n2), sf(scaleFactors2, scaleFactors2 + nlevels),
        ls(levelSigma2_2, levelSigma2_2 + nlevels);
    std::vector<int> a1(nodes1, nodes1 + nn1), o1(off1, off1 + nn1 + 1), e1(feat1, feat1 + off1[nn1]);
    std::vector<int> a2(nodes2, nodes2 + nn2), o2(off2, off2 + nn2 + 1), e2(feat2, feat2 + off2[nn2]);
    TriArgs A;
    memcpy(A.F12, F12, sizeof(A.F12));
    A.epx = epx;
    A.epy = epy;
    A.onlyStereo = onlyStereo;
    A.coarse = coarse;
    A.checkOri = checkOri;
    std::vector<int> m;
    const int nm = search_for_triangulation(k1, d1, f1, u1, a1, o1, e1, k2, d2, f2, u2, a2, o2, e2, sf, ls, A, m);
    if (n1) memcpy(match12, m.data(), (size_t)n1 * sizeof(int));
    return nm;
}

/* pts: n x {pos[3], normal[3], minDistance, maxDistance (float), valid (int)} = 36 bytes; args: FuseArgs */
void orbo_fuse_search(const FusePoint* pts, int n, const uint8_t* mpDesc, const KeyPoint* kfKps, int nKF,
                      const uint8_t* kfDesc, const float* kfURight, const float* scaleFactors,
                      const float* invLevelSigma2, int nlevels, const FuseArgs* args, int* bestIdx, int* bestDist) {
    std::vector<FusePoint> p(pts, pts + n);
    std::vector<uint8_t> md(mpDesc, mpDesc + (size_t)n * 32), kd(kfDesc, kfDesc + (size_t)nKF * 32);
    std::vector<KeyPoint> kk(kfKps, kfKps + nKF);
    std::vector<float> ur(kfURight, kfURight + nKF), sf(scaleFactors, scaleFactors + nlevels),
        is(invLevelSigma2, invLevelSigma2 + nlevels);
    std::vector<int> bi, bd;
    fuse_search(p, md, kk, kd, ur, sf, is, *args, bi, bd);
    if (n) {
        memcpy(bestIdx, bi.data(), (size_t)n * sizeof(int));
        memcpy(bestDist, bd.data(), (size_t)n * sizeof(int));
    }
}

int orbo_search_by_projection_keyframe(const float* Tcw, const float* Ow, const float* cam /*fx,fy,cx,cy*/, float th,
                                       int ORBdist, float logScaleFactor, int checkOri, int imgW, int imgH, int gemmDouble,
                                       const KeyPoint* kfKps, int nKF, const uint8_t* flags, const float* x3Dw,
                                       const float* minDist, const float* maxDist, const uint8_t* mpDesc,
                                       const KeyPoint* curKps, int nCur, const uint8_t* curDesc, const uint8_t* occupied,
                                       const float* scaleFactors, int nlevels, int* matchCur) {
    std::vector<KeyPoint> kk(kfKps, kfKps + nKF), ck(curKps, curKps + nCur);
    std::vector<uint8_t> fl(flags, flags + nKF), md(mpDesc, mpDesc + (size_t)nKF * 32), cd(curDesc, curDesc + (size_t)nCur * 32),
        oc;
    if (occupied) oc.assign(occupied, occupied + nCur);
    std::vector<float> x(x3Dw, x3Dw + (size_t)nKF * 3), mn(minDist, minDist + nKF), mx(maxDist, maxDist + nKF),
        sf(scaleFactors, scaleFactors + nlevels);
    std::vector<int> m;
    const int nm = search_by_projection_keyframe(Tcw, Ow, cam[0], cam[1], cam[2], cam[3], th, ORBdist, logScaleFactor,
                                                 checkOri != 0, imgW, imgH, gemmDouble, kk, fl, x, mn, mx, md, ck, cd, oc, sf, m);
    if (nCur) memcpy(matchCur, m.data(), (size_t)nCur * sizeof(int));
    return nm;
}

int orbo_search_by_projection_sim3(const float* Tcw, const float* Ow, const float* cam, int th, float ratioHamming,
                                   float logScaleFactor, int projVariant, int imgW, int imgH, int gemmDouble, int nPoints,
                                   const uint8_t* flags, const float* x3Dw, const float* normals, const float* minDist,
                                   const float* maxDist, const uint8_t* mpDesc, const KeyPoint* kfKps, int nKF,
                                   const uint8_t* kfDesc, const uint8_t* matched, const float* scaleFactors, int nlevels,
                                   int* matchKf) {
    std::vector<KeyPoint> kk(kfKps, kfKps + nKF);
    std::vector<uint8_t> fl(flags, flags + nPoints), md(mpDesc, mpDesc + (size_t)nPoints * 32), kd(kfDesc, kfDesc + (size_t)nKF * 32),
        mt;
    if (matched) mt.assign(matched, matched + nKF);
    std::vector<float> x(x3Dw, x3Dw + (size_t)nPoints * 3), nr(normals, normals + (size_t)nPoints * 3),
        mn(minDist, minDist + nPoints), mx(maxDist, maxDist + nPoints), sf(scaleFactors, scaleFactors + nlevels);
    std::vector<int> m;
    const int nm = search_by_projection_sim3(Tcw, Ow, cam[0], cam[1], cam[2], cam[3], th, ratioHamming, logScaleFactor,
                                             projVariant, imgW, imgH, gemmDouble, fl, x, nr, mn, mx, md, kk, kd, mt, sf, m);
    if (nKF) memcpy(matchKf, m.data(), (size_t)nKF * sizeof(int));
    return nm;
}

void orbo_search_by_sim3_direction(const float* Ra, const float* ta, const float* Rb, const float* tb, const float* cam,
                                   float th, float logScaleFactor, int imgW, int imgH, int gemmDouble, int n,
                                   const uint8_t* valid, const float* x3Dw, const float* minDist, const float* maxDist,
                                   const uint8_t* mpDesc, const KeyPoint* kfKps, int nKF, const uint8_t* kfDesc,
                                   const float* scaleFactors, int nlevels, int* vnMatch) {
    std::vector<KeyPoint> kk(kfKps, kfKps + nKF);
    std::vector<uint8_t> va(valid, valid + n), md(mpDesc, mpDesc + (size_t)n * 32), kd(kfDesc, kfDesc + (size_t)nKF * 32);
    std::vector<float> x(x3Dw, x3Dw + (size_t)n * 3), mn(minDist, minDist + n), mx(maxDist, maxDist + n),
        sf(scaleFactors, scaleFactors + nlevels);
    std::vector<int> m;
    search_by_sim3_direction(Ra, ta, Rb, tb, cam[0], cam[1], cam[2], cam[3], th, logScaleFactor, imgW, imgH, gemmDouble, va, x,
                             mn, mx, md, kk, kd, sf, m);
    if (n) memcpy(vnMatch, m.data(), (size_t)n * sizeof(int));
}

/* x3dw: n x 3 out, flags: n out (0 / 1) */
void orbo_unproject_stereo(const KeyPoint* kps, int n, const float* depth, const float* Twc, float cx, float cy,
                           float invfx, float invfy, int gemmDouble, float* x3dw, uint8_t* flags) {
    for (int i = 0; i < n; i++) {
        float o[3] = {0, 0, 0};
        flags[i] = unproject_stereo(kps[i], depth[i], Twc, cx, cy, invfx, invfy, gemmDouble, o) ? 1 : 0;
        memcpy(x3dw + 3 * i, o, 12);
    }
}

} /* extern "C" */
