/* Exercises include/vslam_shim.hpp the way frame.cpp / tracking.cpp use the reference classes:
 *   two mono frames  -> FExtractor::compute x2 + FMatcher::SearchForInitialization
 *   one stereo pair  -> two extractors + ComputeStereoMatches
 * Input: raw u8 images written by the pytest driver; output: one line of JSON with counts and FNV-1a checksums
 * that the driver compares with the ctypes path (and through it with the oracle).
 *   shim_demo W H a.raw b.raw left.raw right.raw nfeatures
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <vector>
#include <string>

#include "vslam_shim.hpp"

using namespace vi_slam_amd::geometry;

static Mat8u load(const char* path, int w, int h) {
    Mat8u m;
    m.create(h, w);
    FILE* f = std::fopen(path, "rb");
    if (!f || std::fread(m.data, 1, (size_t)w * h, f) != (size_t)w * h) {
        std::fprintf(stderr, "cannot read %s\n", path);
        std::exit(2);
    }
    std::fclose(f);
    return m;
}

static unsigned long long fnv(const void* p, size_t n, unsigned long long h = 1469598103934665603ull) {
    const unsigned char* b = (const unsigned char*)p;
    for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 1099511628211ull;
    return h;
}

int main(int argc, char** argv) {
    if (argc < 8) return 2;
    const int w = std::atoi(argv[1]), h = std::atoi(argv[2]), nf = std::atoi(argv[7]);
    try {
        /* ---- mono initialisation: tracking.cpp:2290-2330 */
        FExtractor ini1(nf, 1.2f, 8, 20, 7), ini2(nf, 1.2f, 8, 20, 7);
        Mat8u a = load(argv[3], w, h), b = load(argv[4], w, h), mask;
        std::vector<KeyPoint> k1, k2;
        Mat8u d1, d2;
        std::vector<int> lap = {0, 1000};
        const int mono1 = ini1.compute(a, mask, k1, d1, lap);
        const int mono2 = ini2.compute(b, mask, k2, d2, lap);
        Mat8u empty;
        std::vector<KeyPoint> kx;
        Mat8u dx;
        const int rc_empty = ini1.compute(empty, mask, kx, dx, lap); /* -1, and must not disturb ini1's frame */
        std::vector<Point2f> prev(k1.size());
        for (size_t i = 0; i < k1.size(); i++) prev[i] = k1[i].pt;
        std::vector<int> m12;
        FrameView F1, F2;
        F1.ukeypoints = &k1; F1.extractor = &ini1; F1.mnMaxX = w; F1.mnMaxY = h;
        F2.ukeypoints = &k2; F2.extractor = &ini2; F2.mnMaxX = w; F2.mnMaxY = h;
        FMatcher matcher(0.9f, true);
        const int nm = matcher.SearchForInitialization(F1, F2, prev, m12, 100);
        const int dd = k1.size() > 1 ? FMatcher::DescriptorDistance(d1.ptr(0), d1.ptr(1)) : -1;

        /* ---- stereo: frame.cpp:102-132 */
        FExtractor left(nf, 1.2f, 8, 20, 7), right(nf, 1.2f, 8, 20, 7);
        Mat8u L = load(argv[5], w, h), R = load(argv[6], w, h);
        std::vector<KeyPoint> kL, kR;
        Mat8u dL, dR;
        std::vector<int> nolap = {0, 0};
        left.compute(L, mask, kL, dL, nolap);
        right.compute(R, mask, kR, dR, nolap);
        std::vector<float> uR, depth;
        ComputeStereoMatches(left, right, 386.1448f, 718.856f, (int)kL.size(), uR, depth);
        int nst = 0;
        for (float u : uR) nst += u >= 0.f;
        /* ---- tracking: SearchByProjection(right-now = frame b, last = frame a with fake stereo points) */
        std::vector<uint8_t> fl(k1.size(), 3), mpd((size_t)k1.size() * 32);
        std::vector<float> xw((size_t)k1.size() * 3);
        for (size_t i = 0; i < k1.size(); i++) {
            const float z = 8.0f;
            xw[3 * i] = (k1[i].pt.x - w / 2.0f) * z * (1.0f / 500.0f);
            xw[3 * i + 1] = (k1[i].pt.y - h / 2.0f) * z * (1.0f / 500.0f);
            xw[3 * i + 2] = z;
            std::memcpy(&mpd[32 * i], d1.ptr((int)i), 32);
        }
        FMatcher::CurrentFrameView cv;
        cv.frame = F2;
        const float Tc[12] = {1, 0, 0, 3.0f / 500.0f * 8.0f, 0, 1, 0, 1.0f / 500.0f * 8.0f, 0, 0, 1, 0};
        std::memcpy(cv.Tcw, Tc, sizeof(Tc));
        cv.fx = 500.f; cv.fy = 500.f; cv.cx = w / 2.0f; cv.cy = h / 2.0f; cv.mbf = 40.f; cv.mb = 0.08f;
        FMatcher::LastFrameView lv;
        lv.ukeypoints = &k1;
        const float Tl[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
        std::memcpy(lv.Tlw, Tl, sizeof(Tl));
        lv.mapPointFlags = &fl; lv.mapPointWorldPos = &xw; lv.mapPointDescriptors = &mpd;
        std::vector<int> mpi;
        const int nsbp = matcher.SearchByProjection(cv, lv, 15.f, true, mpi);

        int lw = 0, lh = 0;
        std::vector<uint8_t> lvl3 = left.ImagePyramidLevel(3, &lw, &lh);

        /* ---- the grid detector of fast_cuda.cpp:70-132 on frame a (width cut to a multiple of 4) */
        FAST fast;
        std::vector<KeyPoint> kf;
        fast.detect(a.data, w & ~3, h & ~3, a.step, kf);
        FASTGPU multi((size_t)(w & ~3), (size_t)(h & ~3), 32, 32, 0, 3, 0, 0, 10.0f, 10, VSLAM_FG_SUM_OF_ABS_DIFF_ON_ARC);
        multi.detect(a.data, a.step);
        std::vector<double> flat;
        for (size_t i = 0; i < multi.getPoints().size(); i++)
            if (multi.isOccupied(i)) {
                flat.push_back(multi.getPoints()[i].x_);
                flat.push_back(multi.getPoints()[i].y_);
                flat.push_back(multi.getPoints()[i].score_);
                flat.push_back((double)multi.getPoints()[i].level_);
            }

        /* ---- the C ABI directly: switches as per-context parameters, a staged upload consumed with imgs == NULL
         * (twice, so that the second pass replays the captured graph), quadtree statistics, and the descriptor
         * half of ComputeStereoFishEyeMatches on the stereo pair above */
        vslam_tuning tn;
        vslam_tuning_init(&tn);
        tn.oct_fine_depth = 2; /* a shallow quadtree grid for THIS context only: same keypoints, more sub-grid splits */
        tn.stream_priority = 1;
        vslam_fe_params cp;
        std::memset(&cp, 0, sizeof(cp));
        cp.width = w; cp.height = h; cp.nfeatures = nf; cp.scale_factor = 1.2f; cp.nlevels = 8;
        cp.ini_th_fast = 20; cp.min_th_fast = 7; cp.device = 0; cp.max_batch = 1; cp.tuning = &tn;
        vslam_fe* cfe = nullptr;
        if (vslam_fe_create(&cp, &cfe) != VSLAM_OK) throw std::runtime_error(vslam_last_error());
        void* pin = nullptr;
        if (vslam_host_alloc((size_t)w * h, &pin) != VSLAM_OK) throw std::runtime_error(vslam_last_error());
        std::memcpy(pin, a.data, (size_t)w * h);
        const uint8_t* pimgs[1] = {(const uint8_t*)pin};
        const int ccap = vslam_fe_capacity(cfe);
        std::vector<vslam_kp> ck((size_t)ccap);
        std::vector<uint8_t> cd((size_t)ccap * 32);
        vslam_kp* ckp[1] = {ck.data()};
        uint8_t* cdp[1] = {cd.data()};
        int staged_ok = 1, cn = 0, cmono = 0;
        for (int rep = 0; rep < 2; rep++) {
            if (vslam_fe_stage_images_async(cfe, 1, pimgs, (size_t)w, VSLAM_IMGS_PINNED) != VSLAM_OK ||
                vslam_fe_extract_batch(cfe, 1, nullptr, 0, VSLAM_IMGS_STAGED, 0, 1000, ckp, cdp, ccap, &cn, &cmono) != VSLAM_OK)
                throw std::runtime_error(vslam_last_error());
            staged_ok = staged_ok && cn == (int)k1.size() && cmono == mono1 &&
                        !std::memcmp(ck.data(), k1.data(), (size_t)cn * sizeof(vslam_kp)) &&
                        !std::memcmp(cd.data(), d1.data, (size_t)cn * 32);
        }
        unsigned long long oprob = 0, odeep = 0;
        vslam_fe_octree_stats(cfe, &oprob, &odeep, nullptr);
        vslam_host_free(pin);
        vslam_fe_destroy(cfe);
        const uint8_t *fdl = nullptr, *fdr = nullptr;
        int fnl = 0, fnr = 0, fish_n = 0;
        vslam_fe_slot_buffers(left.context(), 0, nullptr, &fdl, &fnl);
        vslam_fe_slot_buffers(right.context(), 0, nullptr, &fdr, &fnr);
        std::vector<int32_t> fish((size_t)(fnl > 0 ? fnl : 1));
        if (vslam_stereo_fisheye_candidates(left.context(), fdl, fnl, 0, fdr, fnr, 0, fish.data(), nullptr, nullptr, &fish_n) != VSLAM_OK)
            throw std::runtime_error(vslam_last_error());

        std::printf("{\"staged_ok\": %d, \"oct_problems\": %llu, \"oct_deep\": %llu, \"fish_n\": %d, \"fish\": %llu, ", staged_ok, oprob,
                    odeep, fish_n, fnv(fish.data(), (size_t)fnl * 4));
        std::printf("\"n1\": %zu, \"n2\": %zu, \"mono1\": %d, \"mono2\": %d, \"rc_empty\": %d, \"nmatches\": %d, "
                    "\"dd01\": %d, \"kp1\": %llu, \"desc1\": %llu, \"kp2\": %llu, \"desc2\": %llu, \"m12\": %llu, "
                    "\"prev\": %llu, \"nL\": %zu, \"nR\": %zu, \"nstereo\": %d, \"uR\": %llu, \"depth\": %llu, "
                    "\"lvl3\": [%d, %d, %llu], \"levels\": %d, \"sf7\": %.9g, \"nsbp\": %d, \"sbp\": %llu, "
                    "\"fast_n\": %zu, \"fast_kp\": %llu, \"fast3_n\": %zu, \"fast3\": %llu}\n",
                    k1.size(), k2.size(), mono1, mono2, rc_empty, nm, dd, fnv(k1.data(), k1.size() * sizeof(KeyPoint)),
                    fnv(d1.data, (size_t)d1.rows * 32), fnv(k2.data(), k2.size() * sizeof(KeyPoint)),
                    fnv(d2.data, (size_t)d2.rows * 32), fnv(m12.data(), m12.size() * 4),
                    fnv(prev.data(), prev.size() * 8), kL.size(), kR.size(), nst, fnv(uR.data(), uR.size() * 4),
                    fnv(depth.data(), depth.size() * 4), lw, lh, fnv(lvl3.data(), lvl3.size()), left.GetLevels(),
                    (double)left.GetScaleFactors()[7], nsbp, fnv(mpi.data(), mpi.size() * 4), kf.size(),
                    fnv(kf.data(), kf.size() * sizeof(KeyPoint)), multi.count(), fnv(flat.data(), flat.size() * 8));
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
