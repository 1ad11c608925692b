/* oracle/orb_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement ("oracle-R") of the reference's visual front-end hot path.  Nothing in the
 * product (vi_slam_amd/, include/) may include, link or call this; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as the checker.
 *
 * PARITY STATUS: "parity unpinned" against the reference as a whole -- the reference has no test or
 * golden vector for this path and its pixel arithmetic lives in OpenCV 4.2, which is
 * neither vendored under /root/reference nor installed.  What IS pinned:
 *   - the FAST stage, against the reference's own Rosten FAST sources compiled into oracle/_ref/
 *     (thirdparty/vilib/visual_lib/src/feature_detection/fast/rosten/{fast,fast_9,nonmax}.cpp);
 *   - DescriptorDistance, against hardware popcount;
 *   - constructor tables (quotas, umax, scale factors) against the values in SURVEY.md 8.
 * The OpenCV 4.2.0 primitives (resize INTER_LINEAR 8u, FAST 9/16 + NMS, GaussianBlur 8u fixed point,
 * fastAtan2, cvRound) are restated from their published algorithm; every build/version dependent
 * constant is a named knob in orbo::Knobs.
 */
#ifndef ORB_ORACLE_H
#define ORB_ORACLE_H

#include <cstddef>
#include <cstdint>
#include <map>
#include <vector>

namespace orbo {

/* Same field order and size as cv::KeyPoint (28 bytes). */
struct KeyPoint {
    float x, y, size, angle, response;
    int octave, class_id;
};
static_assert(sizeof(KeyPoint) == 28, "KeyPoint must mirror cv::KeyPoint");

struct Knobs {
    /* cv::GaussianBlur(7x7, sigma=2) CV_8U fixed-point taps (8 fractional bits).  OpenCV >= 3.4.9/4.2.0
     * diffuses the rounding error so the taps sum to 256: {18,34,48,56,48,34,18}.  Older 3.4/4.x rounded
     * each tap independently: {18,34,49,55,49,34,18} (sum 257). */
    int gauss_taps[7] = {18, 34, 48, 56, 48, 34, 18};
    /* cv::fastAtan2 polynomial: 0 = separate mul/add (baseline-ISA build), 1 = FMA-contracted Horner
     * (what GCC emits for OpenCV's AVX2/FMA3 dispatch unit and on aarch64). */
    int atan_fma = 0;
};

struct Image {
    int w = 0, h = 0;
    std::vector<uint8_t> px; /* contiguous, stride == w */
    const uint8_t* row(int y) const { return px.data() + (size_t)y * w; }
    uint8_t* row(int y) { return px.data() + (size_t)y * w; }
};

/* ---- OpenCV primitive restatements (each cites the call site in the reference) ---- */
int cv_round_f(float v);   /* cvRound(float): round-half-even  (fextractor.cpp:72,106,110,433,1140) */
int cv_round_d(double v);  /* cvRound(double)                  (fextractor.cpp:451) */
void resize_linear_u8(const uint8_t* src, int sw, int sh, size_t sstride, uint8_t* dst, int dw, int dh,
                      size_t dstride); /* cv::resize(..., INTER_LINEAR), fextractor.cpp:1148 */
void fast_detect(const uint8_t* img, int w, int h, size_t stride, int threshold, bool nonmax,
                 std::vector<KeyPoint>& out); /* cv::FAST(img,kps,th,true), fextractor.cpp:800-806 */
void gaussian_blur7(const Image& src, Image& dst, const int taps[7]); /* fextractor.cpp:1086 */
float fast_atan2(float y, float x, int fma);                        /* fextractor.cpp:94 */
int descriptor_distance(const uint8_t* a, const uint8_t* b);       /* fmatcher.cpp:2859-2875 */

std::vector<KeyPoint> distribute_octree(const std::vector<KeyPoint>& keys, int minX, int maxX, int minY,
                                        int maxY, int N); /* fextractor.cpp:530-754 */

class Extractor { /* FExtractor, fextractor.h:26-91 */
public:
    Extractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST,
              const Knobs& knobs = Knobs());

    /* FExtractor::compute (fextractor.cpp:1034-1133).  Returns monoIndex, or -1 on empty image.
     * desc is N x 32 bytes row-major. */
    int compute(const uint8_t* img, int w, int h, size_t stride, int lap0, int lap1,
                std::vector<KeyPoint>& kps, std::vector<uint8_t>& desc);

    int nfeatures;
    double scaleFactor;
    int nlevels, iniThFAST, minThFAST;
    Knobs knobs;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    std::vector<int> mnFeaturesPerLevel, umax;

    /* State after compute(), exposed for stage-wise parity tests. */
    std::vector<Image> mvImagePyramid;                   /* borderless level images (fextractor.h:64) */
    std::vector<Image> mvBlurred;                        /* blurred clones, only levels with keypoints */
    std::vector<std::vector<KeyPoint>> vToDistribute;    /* per level FAST candidates, coords minus border */
    std::vector<std::vector<KeyPoint>> allKeypoints;     /* per level after octree+orientation, level coords */

    void ComputePyramid(const uint8_t* img, int w, int h, size_t stride);
    void ComputeKeyPointsOctTree();
};

/* Frame::ComputeStereoMatches (frame.cpp:823-997).  Pyramids come from the two extractors.
 * uRight/depth are resized to kpsL.size() and filled with -1 for unmatched. */
void compute_stereo_matches(const Extractor& exL, const Extractor& exR, const std::vector<KeyPoint>& kpsL,
                            const std::vector<uint8_t>& descL, const std::vector<KeyPoint>& kpsR,
                            const std::vector<uint8_t>& descR, float bf, float fx,
                            std::vector<float>& uRight, std::vector<float>& depth,
                            std::vector<int>* bestIdxR = nullptr, std::vector<int>* bestSad = nullptr);

/* Frame grid (frame.cpp:386-414, 678-756) for an undistorted W x H image. */
struct FrameGrid {
    static const int COLS = 64, ROWS = 48; /* frame.h:42-43 */
    float mnMinX, mnMaxX, mnMinY, mnMaxY, invW, invH;
    std::vector<int> cell[COLS][ROWS];
    const std::vector<KeyPoint>* kps;
    FrameGrid(const std::vector<KeyPoint>& k, int imgW, int imgH);
    std::vector<int> GetFeaturesInArea(float x, float y, float r, int minLevel, int maxLevel) const;
};

/* FMatcher::SearchForInitialization (fmatcher.cpp:983-1098).  Returns nmatches. */
int search_for_initialization(const std::vector<KeyPoint>& kps1, const std::vector<uint8_t>& desc1,
                              const std::vector<KeyPoint>& kps2, const std::vector<uint8_t>& desc2,
                              int imgW, int imgH, std::vector<float>& prevMatchedXY /* 2*N1, in/out */,
                              std::vector<int>& matches12, int windowSize, float nnratio, bool checkOri);

/* FMatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono) (fmatcher.cpp:2471-2687)
 * for pinhole frames (Nleft == -1).  What it reads of the two frames is passed explicitly. */
struct ProjFrameArgs {
    float Tcw[12];   /* CurrentFrame pose rows [Rcw | tcw] (the reference's T_w_c_ member, used as Rcw/tcw) */
    float Tlw[12];   /* LastFrame pose, only for bForward / bBackward */
    float fx, fy, cx, cy, mbf, mb;
    float th;
    int bMono, checkOri;
    int imgW, imgH;  /* mnMinX = mnMinY = 0, mnMaxX = imgW, mnMaxY = imgH (no distortion) */
    int gemmDouble;  /* knob: cv::gemm accumulates CV_32F products in double (OpenCV GEMMSingleMul<float,double>) */
};
/* last frame: per keypoint i -- flags bit0 = has a MapPoint that is not an outlier, bit1 = that MapPoint has
 * Observations() > 0; x3Dw = its world position; mpDesc = its descriptor; lastKps = keypoints_ (octave) /
 * ukeypoints_ (angle).  current frame: ukeypoints_, descriptors_, mvuRight (-1 = none), occupied = initial
 * "mvpMapPoints[i2] with Observations() > 0" (normally all 0: tracking.cpp fills NULL before the call).
 * Output: matchCur[i2] = index i of the last-frame keypoint whose MapPoint was assigned, -1 = none.  Returns
 * nmatches exactly as the reference counts it. */
int search_by_projection_frame(const ProjFrameArgs& a, const std::vector<KeyPoint>& lastKps,
                               const std::vector<uint8_t>& flags, const std::vector<float>& x3Dw,
                               const std::vector<uint8_t>& mpDesc, const std::vector<KeyPoint>& curKps,
                               const std::vector<uint8_t>& curDesc, const std::vector<float>& mvuRight,
                               const std::vector<uint8_t>& occupied, const std::vector<float>& scaleFactors,
                               std::vector<int>& matchCur);
/* FMatcher::SearchByProjection(Frame& F, const vector<MapPoint*>& vpMapPoints, th, bFarPoints, thFarPoints)
 * (fmatcher.cpp:321-411, pinhole frames: Nleft == -1) -- the local-map matcher of SearchLocalPoints.
 * Per MapPoint what Frame::isInFrustum left in it: */
struct MapPointTrack {
    float projX, projY, projXR, viewCos; /* mTrackProjX, mTrackProjY, mTrackProjXR, mTrackViewCos */
    int level;                           /* mnTrackScaleLevel */
    unsigned flags; /* bit0: mbTrackInView && !isBad() && !(bFarPoints && mTrackDepth > thFarPoints); bit1: Observations() > 0 */
};
int search_by_projection_mappoints(const std::vector<MapPointTrack>& mps, const std::vector<uint8_t>& mpDesc,
                                   const std::vector<KeyPoint>& curKps, const std::vector<uint8_t>& curDesc,
                                   const std::vector<float>& mvuRight, const std::vector<uint8_t>& occupied,
                                   const std::vector<float>& scaleFactors, int imgW, int imgH, float th, float nnratio,
                                   std::vector<int>& matchCur);

/* MapPoint::ComputeDistinctiveDescriptors (mappoint.cpp:322-390): index of the descriptor with the least median
 * Hamming distance to the others (median = sorted[int(0.5*(N-1))], first wins); -1 for an empty set. */
int distinctive_descriptor(const uint8_t* desc, int n);

/* Frame::ComputeBoW (frame.cpp:455-461) = DBoW3::Vocabulary::transform(features, BowVector&, FeatureVector&, 4)
 * (thirdparty/DBoW3/DBoW3/src/Vocabulary.cpp:754-826, :838-878; BowVector.cpp addWeight/normalize;
 * FeatureVector.cpp:31-45; DescManip.cpp:92-119).  The vocabulary is a flat copy of m_nodes: node i has children
 * childIds[childStart[i] .. +childCount[i]) in their stored order (order decides ties), a 32-byte descriptor,
 * and -- if it is a leaf -- a word id and a weight. */
struct Vocabulary {
    int L = 0;                       /* m_L */
    int weighting = 0;               /* 0 TF_IDF, 1 TF, 2 IDF, 3 BINARY (DBoW3::WeightingType) */
    int norm = 1;                    /* scoring object's mustNormalize: 0 none, 1 L1, 2 L2 */
    std::vector<int> childStart, childCount, childIds, wordId;
    std::vector<uint8_t> desc;
    std::vector<double> weight;
};
void bow_transform_feature(const Vocabulary& v, const uint8_t* feature, int levelsup, int& word, double& weight,
                           int& nid);
struct BowResult {
    std::vector<int> words;              /* BowVector keys, ascending */
    std::vector<double> values;          /* BowVector values */
    std::vector<int> nodes;              /* FeatureVector keys, ascending */
    std::vector<std::vector<unsigned>> features; /* FeatureVector values */
};
void bow_transform(const Vocabulary& v, const uint8_t* desc, int n, int levelsup, BowResult& out);

/* FMatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vector<MapPoint*>& vpMapPointMatches) (fmatcher.cpp:546-748,
 * pinhole: F.Nleft == -1).  FeatureVectors as (ascending node ids, offsets, feature indices); kfFlags[i] != 0 iff
 * vpMapPointsKF[i] exists and is not bad.  matchF[iF] = KeyFrame feature index whose MapPoint lands in
 * vpMapPointMatches[iF], -1 = NULL.  Returns nmatches. */
int search_by_bow(const std::vector<KeyPoint>& kfKps, const std::vector<uint8_t>& kfDesc, const std::vector<uint8_t>& kfFlags,
                  const std::vector<int>& kfNodes, const std::vector<int>& kfOff, const std::vector<int>& kfFeat,
                  const std::vector<KeyPoint>& fKps, const std::vector<uint8_t>& fDesc, const std::vector<int>& fNodes,
                  const std::vector<int>& fOff, const std::vector<int>& fFeat, float nnratio, bool checkOri,
                  std::vector<int>& matchF);

/* FMatcher::SearchByBoW(KeyFrame*, KeyFrame*, vector<MapPoint*>& vpMatches12) (fmatcher.cpp:1100-1240):
 * match12[idx1] = idx2 or -1; flags: MapPoint exists and is not bad. */
int search_by_bow_keyframes(const std::vector<KeyPoint>& kps1, const std::vector<uint8_t>& desc1,
                            const std::vector<uint8_t>& flags1, const std::vector<int>& nodes1, const std::vector<int>& off1,
                            const std::vector<int>& feat1, const std::vector<KeyPoint>& kps2,
                            const std::vector<uint8_t>& desc2, const std::vector<uint8_t>& flags2,
                            const std::vector<int>& nodes2, const std::vector<int>& off2, const std::vector<int>& feat2,
                            float nnratio, bool checkOri, std::vector<int>& match12);

/* Frame::UnprojectStereo (frame.cpp:1023-1037): returns false (cv::Mat()) when mvDepth[i] <= 0 */
bool unproject_stereo(const KeyPoint& kpUn, float z, const float Twc[12], float cx, float cy, float invfx, float invfy,
                      int gemmDouble, float out[3]);
/* bForward / bBackward of the same function (fmatcher.cpp:2482-2495) */
void projection_direction(const ProjFrameArgs& a, bool& bForward, bool& bBackward);

/* glibc logf (sysdeps/ieee754/flt-32/e_logf.c with its 16-entry table, glibc 2.35): `log(ratio)` on a float under
 * `using namespace std` (mappoint.cpp:11,514) is std::log(float) = logf.  The restatement was compared against the
 * platform libm for every positive finite float (tests/test_oracle.py samples it again). */
float glibc_logf(float x);

/* FMatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo, bCoarse) (fmatcher.cpp:1242-1482;
 * SearchForTriangulation_ :1484-1725 is the same walk on cv::Matx) for pinhole KeyFrames without a second camera
 * (mpCamera2 == NULL, NLeft == -1).  hasMp[i] != 0 iff GetMapPoint(i) != NULL; F12 is the matrix
 * Pinhole::epipolarConstrain builds for the pair (pinhole.cpp:121-143; it does not depend on the keypoints), ep the
 * epipole pKF2->mpCamera->project(R2w*Cw+t2w).  match12[idx1] = idx2 or -1 (vbMatched2 is never set in the
 * reference, so two idx1 may share an idx2).  Returns nmatches. */
struct TriArgs {
    float F12[9];
    float epx, epy;
    int onlyStereo, coarse, checkOri;
};
int search_for_triangulation(const std::vector<KeyPoint>& kps1, const std::vector<uint8_t>& desc1,
                             const std::vector<uint8_t>& hasMp1, const std::vector<float>& uRight1,
                             const std::vector<int>& nodes1, const std::vector<int>& off1, const std::vector<int>& feat1,
                             const std::vector<KeyPoint>& kps2, const std::vector<uint8_t>& desc2,
                             const std::vector<uint8_t>& hasMp2, const std::vector<float>& uRight2,
                             const std::vector<int>& nodes2, const std::vector<int>& off2, const std::vector<int>& feat2,
                             const std::vector<float>& scaleFactors2, const std::vector<float>& levelSigma2_2,
                             const TriArgs& a, std::vector<int>& match12);

/* The search half of FMatcher::Fuse (fmatcher.cpp:1918-2119 with bRight = false, and the Sim3 overload
 * :2121-2243): per MapPoint the projection gates, MapPoint::PredictScale (mappoint.cpp:506-521),
 * KeyFrame::GetFeaturesInArea (keyframe.cpp:656-699) and the best keypoint of the window.  What the reference does
 * with the result (Replace / AddObservation / vpReplacePoint) mutates the map and stays with the caller.
 * bestIdx[i] = -1 when a gate rejects the point or no keypoint of the window passes; bestDist[i] = 256 then. */
struct FusePoint {
    float pos[3];    /* GetWorldPos() */
    float normal[3]; /* GetNormal() */
    float minDistance, maxDistance; /* Get{Min,Max}DistanceInvariance() */
    int valid;       /* pMP && !isBad() && !IsInKeyFrame(pKF) (Sim3: !isBad() && !spAlreadyFound.count(pMP)) */
};
struct FuseArgs {
    float Rcw[9], tcw[3], Ow[3];
    float fx, fy, cx, cy, bf, th, logScaleFactor;
    int imgW, imgH; /* mnMaxX, mnMaxY; mnMinX = mnMinY = 0 */
    int sim3;       /* 1: no chi2 gate (fmatcher.cpp:2205-2222) */
    int gemmDouble;
};
void fuse_search(const std::vector<FusePoint>& pts, const std::vector<uint8_t>& mpDesc, const std::vector<KeyPoint>& kfKps,
                 const std::vector<uint8_t>& kfDesc, const std::vector<float>& kfURight,
                 const std::vector<float>& scaleFactors, const std::vector<float>& invLevelSigma2, const FuseArgs& a,
                 std::vector<int>& bestIdx, std::vector<int>& bestDist);

/* FMatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const set<MapPoint*>& sAlreadyFound, th, ORBdist)
 * (fmatcher.cpp:2689-2811), pinhole: flags[i] & 1 iff vpMPs[i] && !isBad() && !sAlreadyFound.count(vpMPs[i]);
 * occupied: CurrentFrame.mvpMapPoints[i2] != NULL on entry.  matchCur[i2] = i or -1; returns nmatches. */
int search_by_projection_keyframe(const float Tcw[12], const float Ow[3], float fx, float fy, float cx, float cy, float th,
                                  int ORBdist, float logScaleFactor, bool checkOri, int imgW, int imgH, int gemmDouble,
                                  const std::vector<KeyPoint>& kfKps, const std::vector<uint8_t>& flags,
                                  const std::vector<float>& x3Dw, const std::vector<float>& minDist,
                                  const std::vector<float>& maxDist, const std::vector<uint8_t>& mpDesc,
                                  const std::vector<KeyPoint>& curKps, const std::vector<uint8_t>& curDesc,
                                  const std::vector<uint8_t>& occupied, const std::vector<float>& scaleFactors,
                                  std::vector<int>& matchCur);

/* FMatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, vpPoints, vpMatched, th, ratioHamming) (fmatcher.cpp:750-863;
 * projVariant 0) and the vpPointsKFs / vpMatchedKF overload (:865-981; projVariant 1: u = fx*(x*(1/z))+cx).
 * flags[i] & 1 iff !isBad() && !spAlreadyFound.count(vpPoints[i]); matched0: vpMatched[idx] != NULL on entry.
 * matchKf[idx] = iMP for the keypoints assigned by this call, else -1; returns nmatches. */
int search_by_projection_sim3(const float Tcw[12], const float Ow[3], float fx, float fy, float cx, float cy, int th,
                              float ratioHamming, float logScaleFactor, int projVariant, int imgW, int imgH, int gemmDouble,
                              const std::vector<uint8_t>& flags, const std::vector<float>& x3Dw,
                              const std::vector<float>& normals, const std::vector<float>& minDist,
                              const std::vector<float>& maxDist, const std::vector<uint8_t>& mpDesc,
                              const std::vector<KeyPoint>& kfKps, const std::vector<uint8_t>& kfDesc,
                              const std::vector<uint8_t>& matched0, const std::vector<float>& scaleFactors,
                              std::vector<int>& matchKf);

/* One direction of FMatcher::SearchBySim3 (fmatcher.cpp:2291-2368 with Ra|ta = R1w|t1w, Rb|tb = sR21|t21 and pKF2's
 * keypoints; :2371-2448 with R2w|t2w, sR12|t12 and pKF1's): vnMatch[i] = best keypoint (bestDist <= TH_HIGH) or -1.
 * valid[i]: the keypoint has a MapPoint that is not bad and is not matched yet. */
void search_by_sim3_direction(const float Ra[9], const float ta[3], const float Rb[9], const float tb[3], float fx, float fy,
                              float cx, float cy, float th, float logScaleFactor, int imgW, int imgH, int gemmDouble,
                              const std::vector<uint8_t>& valid, const std::vector<float>& x3Dw,
                              const std::vector<float>& minDist, const std::vector<float>& maxDist,
                              const std::vector<uint8_t>& mpDesc, const std::vector<KeyPoint>& kfKps,
                              const std::vector<uint8_t>& kfDesc, const std::vector<float>& scaleFactors,
                              std::vector<int>& vnMatch);

} // namespace orbo

#endif
