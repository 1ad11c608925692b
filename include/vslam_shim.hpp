/* vslam_shim.hpp -- header-only C++ face of libvslam_fe.so with the reference's class names and signatures.
 *
 * What a maintainer of KMS-TEAM/vi_slam includes instead of (the bodies of)
 *   include/vi_slam/geometry/fextractor.h:26-91   -> FExtractor
 *   include/vi_slam/geometry/fmatcher.h:70-147    -> FMatcher (hot-path subset + the two pinhole SearchByProjection
 *                                                     overloads used by tracking)
 *   src/datastructures/frame.cpp:823-997          -> ComputeStereoMatches
 * Everything forwards to the C ABI in vslam_fe.h; nothing is computed here.
 *
 * cv:: types: with -DVSLAM_SHIM_WITH_OPENCV the signatures take cv::Mat / cv::KeyPoint / cv::Point2f exactly as
 * the reference does.  Without OpenCV (this repository's test build) POD stand-ins with the same layout are
 * used: KeyPoint == cv::KeyPoint (28 bytes), Point2f == cv::Point2f, Mat8u == a CV_8UC1 cv::Mat
 * (rows, cols, step, data).
 *
 * Error behaviour follows the reference where it has one (compute() returns -1 on an empty image,
 * fextractor.cpp:1037-1038); failures of the library surface as std::runtime_error(vslam_last_error()).
 */
#ifndef VSLAM_SHIM_HPP
#define VSLAM_SHIM_HPP

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <memory>
#include <vector>

#include "vslam_fe.h"
#include "vslam_fastgrid.h"

#ifdef VSLAM_SHIM_WITH_OPENCV
#include <opencv2/core/core.hpp>
#endif

#ifndef VSLAM_SHIM_NAMESPACE
#define VSLAM_SHIM_NAMESPACE vi_slam_amd
#endif

namespace VSLAM_SHIM_NAMESPACE {
namespace geometry {

#ifdef VSLAM_SHIM_WITH_OPENCV
typedef cv::KeyPoint KeyPoint;
typedef cv::Point2f Point2f;
#else
struct Point2f {
    float x, y;
};
struct KeyPoint { /* field order of cv::KeyPoint */
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
};
/* CV_8UC1 matrix stand-in: non-owning view or owning buffer */
struct Mat8u {
    int rows = 0, cols = 0;
    size_t step = 0;
    uint8_t* data = nullptr;
    std::vector<uint8_t> own;
    Mat8u() {}
    Mat8u(int r, int c, uint8_t* d, size_t s) : rows(r), cols(c), step(s), data(d) {}
    void create(int r, int c) {
        rows = r;
        cols = c;
        step = (size_t)c;
        own.assign((size_t)r * c, 0);
        data = own.data();
    }
    void release() {
        rows = cols = 0;
        step = 0;
        own.clear();
        data = nullptr;
    }
    bool empty() const { return !data || rows == 0 || cols == 0; }
    uint8_t* ptr(int r) { return data + (size_t)r * step; }
    const uint8_t* ptr(int r) const { return data + (size_t)r * step; }
};
#endif
static_assert(sizeof(KeyPoint) == sizeof(vslam_kp), "KeyPoint must be layout-compatible with vslam_kp");
static_assert(sizeof(Point2f) == 8, "Point2f is two packed floats");

inline void check(int rc) {
    if (rc != VSLAM_OK) throw std::runtime_error(std::string("libvslam_fe: ") + vslam_last_error());
}

/* ---------------------------------------------------------------------------------------------------
 * FExtractor (fextractor.h:26-91).  The image size becomes known at the first compute(), as in the
 * reference (mvImagePyramid is sized there, fextractor.cpp:1135-1160); the device context is created then
 * and re-created if the size changes.  One instance per concurrent stream, like the reference
 * (tracking.cpp:1087-1093 allocates Left / Right / Ini extractors).
 * ------------------------------------------------------------------------------------------------- */
class FExtractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    FExtractor(int nfeatures_, float scaleFactor_, int nlevels_, int iniThFAST_, int minThFAST_, int device_ = 0,
               int max_batch_ = 1)
        : nfeatures(nfeatures_), scaleFactor(scaleFactor_), nlevels(nlevels_), iniThFAST(iniThFAST_),
          minThFAST(minThFAST_), device(device_), max_batch(max_batch_) {
        /* scale tables exactly as fextractor.cpp:406-422 (float products, scaleFactor held as double) */
        mvScaleFactor.resize(nlevels);
        mvLevelSigma2.resize(nlevels);
        mvScaleFactor[0] = 1.0f;
        mvLevelSigma2[0] = 1.0f;
        for (int i = 1; i < nlevels; i++) {
            mvScaleFactor[i] = (float)(mvScaleFactor[i - 1] * scaleFactor);
            mvLevelSigma2[i] = mvScaleFactor[i] * mvScaleFactor[i];
        }
        mvInvScaleFactor.resize(nlevels);
        mvInvLevelSigma2.resize(nlevels);
        for (int i = 0; i < nlevels; i++) {
            mvInvScaleFactor[i] = 1.0f / mvScaleFactor[i];
            mvInvLevelSigma2[i] = 1.0f / mvLevelSigma2[i];
        }
    }
    ~FExtractor() {
        if (fe_) vslam_fe_destroy(fe_);
    }
    FExtractor(const FExtractor&) = delete;
    FExtractor& operator=(const FExtractor&) = delete;

#ifdef VSLAM_SHIM_WITH_OPENCV
    int compute(cv::InputArray image, cv::InputArray /*mask*/, std::vector<cv::KeyPoint>& keypoints,
                cv::OutputArray descriptors, std::vector<int>& vLappingArea) {
        if (image.empty()) return -1;
        cv::Mat im = image.getMat();
        CV_Assert(im.type() == CV_8UC1);
        std::vector<uint8_t> d;
        int n = 0;
        const int mono = run(im.data, im.cols, im.rows, im.step, keypoints, d, vLappingArea, &n);
        if (n == 0) descriptors.release();
        else cv::Mat(n, 32, CV_8U, d.data()).copyTo(descriptors);
        return mono;
    }
#else
    /* mask is ignored, as in the reference ("Mask is ignored in the current implementation") */
    int compute(const Mat8u& image, const Mat8u& /*mask*/, std::vector<KeyPoint>& keypoints, Mat8u& descriptors,
                std::vector<int>& vLappingArea) {
        if (image.empty()) return -1; /* fextractor.cpp:1037-1038 */
        std::vector<uint8_t> d;
        int n = 0;
        const int mono = run(image.data, image.cols, image.rows, image.step, keypoints, d, vLappingArea, &n);
        if (n == 0) descriptors.release();
        else {
            descriptors.create(n, 32);
            std::memcpy(descriptors.data, d.data(), (size_t)n * 32);
        }
        return mono;
    }
#endif

    int GetLevels() { return nlevels; }
    float GetScaleFactor() { return (float)scaleFactor; }
    std::vector<float> GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    /* mvImagePyramid[level] (fextractor.h:64), fetched from HBM on demand; borderless w x h, tight rows */
    std::vector<uint8_t> ImagePyramidLevel(int level, int* w, int* h) const {
        if (!fe_) throw std::runtime_error("FExtractor: no image has been processed yet");
        check(vslam_fe_level_size(fe_, level, w, h));
        std::vector<uint8_t> out((size_t)*w * *h);
        check(vslam_fe_level_copy(fe_, 0, level, 0, out.data(), (size_t)*w));
        return out;
    }

    /* the device context (NULL before the first compute): device-side consumers -- FMatcher, stereo -- use it */
    vslam_fe* context() const { return fe_; }

protected:
    int run(const uint8_t* data, int cols, int rows, size_t step, std::vector<KeyPoint>& keypoints,
            std::vector<uint8_t>& desc, std::vector<int>& vLappingArea, int* n_out) {
        if (!fe_ || cols != w_ || rows != h_) {
            if (fe_) vslam_fe_destroy(fe_);
            fe_ = nullptr;
            vslam_fe_params p;
            std::memset(&p, 0, sizeof(p));
            p.width = cols;
            p.height = rows;
            p.nfeatures = nfeatures;
            p.scale_factor = (float)scaleFactor;
            p.nlevels = nlevels;
            p.ini_th_fast = iniThFAST;
            p.min_th_fast = minThFAST;
            p.device = device;
            p.max_batch = max_batch;
            check(vslam_fe_create(&p, &fe_));
            w_ = cols;
            h_ = rows;
        }
        const int cap = vslam_fe_capacity(fe_); /* quota + the quadtree's overshoot, see vslam_fe.h */
        keypoints.resize(cap);
        desc.resize((size_t)cap * 32);
        int n = 0, mono = 0;
        const int lap0 = vLappingArea.size() > 0 ? vLappingArea[0] : 0, lap1 = vLappingArea.size() > 1 ? vLappingArea[1] : 0;
        check(vslam_fe_extract(fe_, data, step, lap0, lap1, reinterpret_cast<vslam_kp*>(keypoints.data()), desc.data(),
                               cap, &n, &mono));
        keypoints.resize(n);
        *n_out = n;
        return mono;
    }

    int nfeatures;
    double scaleFactor;
    int nlevels;
    int iniThFAST;
    int minThFAST;
    int device, max_batch;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    vslam_fe* fe_ = nullptr;
    int w_ = 0, h_ = 0;
};

/* What the matchers read of a Frame (frame.h:71-91): undistorted keypoints, the image bounds of its grid,
 * and where its descriptors are in HBM (the extractor that produced them still holds them). */
struct FrameView {
    const std::vector<KeyPoint>* ukeypoints = nullptr; /* Frame::ukeypoints_ */
    const FExtractor* extractor = nullptr;             /* descriptors_ live in its slot 0 */
    int mnMaxX = 0, mnMaxY = 0;                        /* image bounds (no distortion: cols, rows) */
};

/* ---------------------------------------------------------------------------------------------------
 * FMatcher (fmatcher.h:70-147), hot-path subset
 * ------------------------------------------------------------------------------------------------- */
class FMatcher {
public:
    static const int TH_LOW = 50, TH_HIGH = 100, HISTO_LENGTH = 30; /* fmatcher.cpp:313-315 */

    FMatcher(float nnratio = 0.6f, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

    /* fmatcher.cpp:2859-2875: 256-bit Hamming distance of two descriptor rows */
    static int DescriptorDistance(const uint8_t* a, const uint8_t* b) {
        int dist = 0;
        for (int i = 0; i < 8; i++) {
            uint32_t pa, pb;
            std::memcpy(&pa, a + 4 * i, 4);
            std::memcpy(&pb, b + 4 * i, 4);
            dist += __builtin_popcount(pa ^ pb);
        }
        return dist;
    }

    /* fmatcher.cpp:983-1098 */
    int SearchForInitialization(const FrameView& F1, const FrameView& F2, std::vector<Point2f>& vbPrevMatched,
                                std::vector<int>& vnMatches12, int windowSize = 10) {
        const std::vector<KeyPoint>& k1 = *F1.ukeypoints;
        const std::vector<KeyPoint>& k2 = *F2.ukeypoints;
        vnMatches12.assign(k1.size(), -1);
        if (k1.empty()) return 0;
        if (vbPrevMatched.size() != k1.size()) throw std::invalid_argument("vbPrevMatched.size() != F1 keypoints");
        const uint8_t *d1 = nullptr, *d2 = nullptr;
        int n1 = 0, n2 = 0, nm = 0;
        check(vslam_fe_slot_buffers(F1.extractor->context(), 0, nullptr, &d1, &n1));
        check(vslam_fe_slot_buffers(F2.extractor->context(), 0, nullptr, &d2, &n2));
        if (n1 != (int)k1.size() || n2 != (int)k2.size())
            throw std::runtime_error("FMatcher: the extractor's slot no longer holds this frame's descriptors");
        check(vslam_search_for_initialization(F2.extractor->context(), reinterpret_cast<const vslam_kp*>(k1.data()), d1,
                                              n1, reinterpret_cast<const vslam_kp*>(k2.data()), d2, n2, F2.mnMaxX,
                                              F2.mnMaxY, reinterpret_cast<float*>(vbPrevMatched.data()),
                                              vnMatches12.data(), windowSize, mfNNratio, mbCheckOrientation ? 1 : 0,
                                              &nm));
        return nm;
    }

    /* fmatcher.cpp:2471-2687 (pinhole frames).  What the function reads of the two Frames:
     *   cur : pose rows [Rcw | tcw] (T_w_c_ as the reference uses it), intrinsics, mbf, mb, mvuRight, grid bounds,
     *         and the extractor slot that still holds its keypoints/descriptors in HBM
     *   last: ukeypoints_ (octave, angle), pose, and per keypoint its MapPoint (flags bit0 = present and not an
     *         outlier, bit1 = Observations() > 0, world position, descriptor)
     * mvpMapPointIndex[i2] = index into the last frame whose MapPoint ends up in CurrentFrame.mvpMapPoints[i2]. */
    struct CurrentFrameView {
        FrameView frame;
        float Tcw[12];
        float fx, fy, cx, cy, mbf, mb;
        const std::vector<float>* mvuRight = nullptr; /* NULL: monocular */
    };
    struct LastFrameView {
        const std::vector<KeyPoint>* ukeypoints = nullptr;
        float Tlw[12];
        const std::vector<uint8_t>* mapPointFlags = nullptr;
        const std::vector<float>* mapPointWorldPos = nullptr;    /* 3 per keypoint */
        const std::vector<uint8_t>* mapPointDescriptors = nullptr; /* 32 per keypoint */
    };
    int SearchByProjection(const CurrentFrameView& cur, const LastFrameView& last, const float th, const bool bMono,
                           std::vector<int>& mvpMapPointIndex) {
        vslam_proj_params p;
        std::memset(&p, 0, sizeof(p));
        std::memcpy(p.Tcw, cur.Tcw, sizeof(p.Tcw));
        p.fx = cur.fx; p.fy = cur.fy; p.cx = cur.cx; p.cy = cur.cy; p.mbf = cur.mbf; p.th = th;
        p.check_orientation = mbCheckOrientation ? 1 : 0;
        p.img_w = cur.frame.mnMaxX;
        p.img_h = cur.frame.mnMaxY;
        check(vslam_projection_direction(cur.Tcw, last.Tlw, cur.mb, bMono ? 1 : 0, 0, &p.forward, &p.backward));
        const vslam_kp* dk = nullptr;
        const uint8_t* dd = nullptr;
        int n2 = 0, nm = 0;
        check(vslam_fe_slot_buffers(cur.frame.extractor->context(), 0, &dk, &dd, &n2));
        mvpMapPointIndex.assign(n2, -1);
        const int n = (int)last.ukeypoints->size();
        check(vslam_search_by_projection_frame(cur.frame.extractor->context(), &p,
                                               reinterpret_cast<const vslam_kp*>(last.ukeypoints->data()), n,
                                               last.mapPointFlags->data(), last.mapPointWorldPos->data(),
                                               last.mapPointDescriptors->data(), dk, dd, n2,
                                               cur.mvuRight ? cur.mvuRight->data() : nullptr, nullptr,
                                               mvpMapPointIndex.data(), &nm));
        return nm;
    }

    /* fmatcher.cpp:321-411 (pinhole frames): vpMapPoints as vslam_mp_track records + descriptors; occupied marks
     * keypoints of F whose mvpMapPoints entry already holds a MapPoint with observations */
    int SearchByProjection(const FrameView& F, const std::vector<float>* mvuRight, const std::vector<vslam_mp_track>& vpMapPoints,
                           const std::vector<uint8_t>& mapPointDescriptors, const std::vector<uint8_t>* occupied,
                           const float th, std::vector<int>& mvpMapPointIndex) {
        const vslam_kp* dk = nullptr;
        const uint8_t* dd = nullptr;
        int n2 = 0, nm = 0;
        check(vslam_fe_slot_buffers(F.extractor->context(), 0, &dk, &dd, &n2));
        mvpMapPointIndex.assign(n2, -1);
        check(vslam_search_by_projection_mappoints(F.extractor->context(), vpMapPoints.data(), mapPointDescriptors.data(),
                                                   (int)vpMapPoints.size(), dk, dd, n2,
                                                   mvuRight ? mvuRight->data() : nullptr,
                                                   occupied ? occupied->data() : nullptr, F.mnMaxX, F.mnMaxY, th,
                                                   mfNNratio, mvpMapPointIndex.data(), &nm));
        return nm;
    }

protected:
    float mfNNratio;
    bool mbCheckOrientation;
};

/* Frame::ComputeStereoMatches (frame.cpp:823-997): fills mvuRight / mvDepth (size N = left keypoints) from the
 * last frames processed by the two extractors (the reference's mpORBextractorLeft / Right). */
inline void ComputeStereoMatches(const FExtractor& left, const FExtractor& right, float mbf, float fx, int N,
                                 std::vector<float>& mvuRight, std::vector<float>& mvDepth) {
    mvuRight.assign(N, -1.0f);
    mvDepth.assign(N, -1.0f);
    if (N == 0) return;
    check(vslam_stereo_match(left.context(), 0, right.context(), 0, mbf, fx, mvuRight.data(), mvDepth.data()));
}

/* ---------------------------------------------------------------------------------------------------
 * The grid FAST detector of fast_cuda.h:19-25 / fast_cuda.cpp:70-132.  FASTGPU carries vilib::FASTGPU's
 * constructor (fast_gpu.h:46-56) and DetectorBase's read side (detector_base.h:48-57,75-80): detect() on an
 * 8-bit image replaces `Frame(image, 0, levels)` + `detect(frame->pyramid_)`; getPoints() holds one FeaturePoint
 * per grid cell and isOccupied(i) says whether cell i found a corner (OccupancyGrid2D::isOccupied).
 * FAST::detect is the reference's wrapper with its fixed parameters (fast_cuda.cpp:24-39: one level, 32x32
 * cells, epsilon 10, arc 10, SUM_OF_ABS_DIFF_ON_ARC); unlike the reference, which prints the grid and leaves
 * `keypoints` empty, it returns the occupied cells as key points (pt, response = score, octave = level).
 * ------------------------------------------------------------------------------------------------- */
class FASTGPU {
public:
    struct FeaturePoint {
        double x_, y_, score_;
        unsigned int level_;
    };
    FASTGPU(std::size_t image_width, std::size_t image_height, std::size_t cell_size_width, std::size_t cell_size_height,
            std::size_t min_level, std::size_t max_level, std::size_t horizontal_border, std::size_t vertical_border,
            float threshold, int min_arc_length, int score, int device = 0) {
        vslam_fg_params p;
        std::memset(&p, 0, sizeof(p));
        p.image_width = (int32_t)image_width;
        p.image_height = (int32_t)image_height;
        p.cell_size_width = (int32_t)cell_size_width;
        p.cell_size_height = (int32_t)cell_size_height;
        p.min_level = (int32_t)min_level;
        p.max_level = (int32_t)max_level;
        p.horizontal_border = (int32_t)horizontal_border;
        p.vertical_border = (int32_t)vertical_border;
        p.threshold = threshold;
        p.min_arc_length = min_arc_length;
        p.score = score;
        p.tie_rule = 0;
        p.device = device;
        p.max_batch = 1;
        check(vslam_fg_create(&p, &fg_));
        int nc = 0, nr = 0;
        vslam_fg_grid(fg_, &nc, &nr);
        n_cols_ = (std::size_t)nc;
        n_rows_ = (std::size_t)nr;
        reset();
    }
    ~FASTGPU() { vslam_fg_destroy(fg_); }
    FASTGPU(const FASTGPU&) = delete;
    FASTGPU& operator=(const FASTGPU&) = delete;

    void reset() { /* DetectorBase::reset + the constructor's keypoints_ fill (detector_base.cpp:67,76-84) */
        keypoints_.assign(n_cols_ * n_rows_, FeaturePoint{0.0, 0.0, 0.0, (unsigned int)-1});
        occupied_.assign(n_cols_ * n_rows_, 0);
    }
    void detect(const uint8_t* image, std::size_t pitch) { /* detectBase + processGrid, detector_base_gpu.cpp:204-218 */
        const std::size_t n = n_cols_ * n_rows_;
        pos_.resize(2 * n);
        score_.resize(n);
        level_.resize(n);
        check(vslam_fg_detect(fg_, image, pitch, pos_.data(), score_.data(), level_.data()));
        for (std::size_t i = 0; i < n; i++)
            if (score_[i] > 0.0f) {
                keypoints_[i] = FeaturePoint{(double)pos_[2 * i], (double)pos_[2 * i + 1], (double)score_[i], (unsigned int)level_[i]};
                occupied_[i] = 1;
            }
    }
    const std::vector<FeaturePoint>& getPoints() const { return keypoints_; }
    bool isOccupied(std::size_t i) const { return occupied_[i] != 0; }
    std::size_t count() const {
        std::size_t c = 0;
        for (uint8_t o : occupied_) c += o;
        return c;
    }
    std::size_t getCellCountHorizontal() const { return n_cols_; }
    std::size_t getCellCountVertical() const { return n_rows_; }

private:
    vslam_fg* fg_ = nullptr;
    std::size_t n_cols_ = 0, n_rows_ = 0;
    std::vector<FeaturePoint> keypoints_;
    std::vector<uint8_t> occupied_;
    std::vector<float> pos_, score_;
    std::vector<int32_t> level_;
};

class FAST {
public:
    /* image: 8-bit grey (the reference converts BGR first, fast_cuda.cpp:45-47) */
    void detect(const uint8_t* image, int width, int height, std::size_t pitch, std::vector<KeyPoint>& keypoints) {
        if (!det_ || w_ != width || h_ != height) {
            det_.reset(new FASTGPU((std::size_t)width, (std::size_t)height, 32, 32, 0, 1, 0, 0, 10.0f, 10,
                                   VSLAM_FG_SUM_OF_ABS_DIFF_ON_ARC));
            w_ = width;
            h_ = height;
        }
        det_->reset();
        det_->detect(image, pitch);
        keypoints.clear();
        const std::vector<FASTGPU::FeaturePoint>& pts = det_->getPoints();
        for (std::size_t i = 0; i < pts.size(); i++) {
            if (!det_->isOccupied(i)) continue;
            KeyPoint k;
            std::memset(&k, 0, sizeof(k));
            k.pt.x = (float)pts[i].x_;
            k.pt.y = (float)pts[i].y_;
            k.size = 7.0f;
            k.angle = -1.0f;
            k.response = (float)pts[i].score_;
            k.octave = (int)pts[i].level_;
            k.class_id = -1;
            keypoints.push_back(k);
        }
    }
    const FASTGPU* detector() const { return det_.get(); }

private:
    std::unique_ptr<FASTGPU> det_;
    int w_ = 0, h_ = 0;
};

} /* namespace geometry */
} /* namespace VSLAM_SHIM_NAMESPACE */
#endif /* VSLAM_SHIM_HPP */
