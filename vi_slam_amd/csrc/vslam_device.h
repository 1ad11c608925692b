/* vslam_device.h -- structs shared by the host context (vslam_fe.hip) and the kernels. */
#ifndef VSLAM_DEVICE_H
#define VSLAM_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vslam_fe.h"

#define VSLAM_EDGE 19        /* EDGE_THRESHOLD, fextractor.cpp:66 */
#define VSLAM_BORDER 16      /* EDGE_THRESHOLD-3: FAST cells start here, fextractor.cpp:764 */
#define VSLAM_HALF_PATCH 15  /* HALF_PATCH_SIZE, fextractor.cpp:65 */

/* One pyramid level of the context geometry (identical for every image slot). */
struct LevelGeom {
    int32_t w, h;        /* level size: cvRound(cols*inv), cvRound(rows*inv), fextractor.cpp:1139-1140 */
    int32_t pitch;       /* bytes per row in the slot buffers (multiple of 128) */
    uint32_t off;        /* byte offset of the level inside a slot's pyramid / blur buffer */
    float scale;         /* mvScaleFactor[level] */
    int32_t pad[3];
};

struct PyramidGeom {
    int32_t nlevels;
    int32_t pad[3];
    LevelGeom lv[VSLAM_MAX_LEVELS];
};

/* Level-0 sources of the current batch: internal staging or caller device memory (zero copy). */
struct BatchSrc {
    const uint8_t* l0[VSLAM_MAX_BATCH];
    uint32_t pitch0[VSLAM_MAX_BATCH];
};

/* One executed FAST cell (fextractor.cpp:780-819): window [x0,x1) x [y0,y1) in level coordinates. */
struct CellDesc {
    uint16_t level, x0, y0, x1, y1, pad;
    uint32_t base; /* first entry of the cell's fixed candidate segment: capacity ceil(iw/2)*ceil(ih/2), the
                      most strict 3x3 local maxima an iw x ih interior can hold -> no allocation atomics */
};

/* One band of k_fast_bands (vslam::HostBand): consecutive cells of one cell row that share one staged window.  16 bytes,
 * read as one scalar dwordx4. */
struct BandDesc {
    uint32_t cell0; /* first cell (index into the CellDesc table / the slot's CellOut table) */
    uint32_t lnw;   /* level | ncell << 4 | wcell << 8 | column-table class << 16 */
    uint32_t xy;    /* x0 | y0 << 16 */
    uint32_t wh;    /* window width | height << 16 */
};

/* Per-slot candidate region, contiguous so one D2H moves header + cell table + candidates:
 *   uint32 unused; uint32 overflow; CellOut cells[ncells]; uint32 cand[cap]
 * cand = (score << 24) | ((y-16) << 12) | (x-16), level coordinates. */
struct CellOut {
    uint32_t base, count;
};

/* A keypoint chosen by the quadtree, as the orientation/descriptor kernel wants it. */
struct SelKp {
    uint16_t x, y;     /* level coordinates */
    uint8_t level, slot;
    uint8_t response;  /* FAST score (cv::KeyPoint::response is this integer as float) */
    uint8_t pad;
    uint32_t out;      /* index into the slot's output arrays (lapping-area order, fextractor.cpp:1118-1127) */
};

/* One part of a level's key-counting walk (k_oct_count): the keys with ylo <= y < yhi -- whole rows of leaves of the implicit
 * quadtree, so every leaf is counted by exactly one part -- which lie in the FAST cells [ca, cb) (cell rows that overlap the
 * y range; a cell row on a boundary is read by both parts, each taking its own keys). */
struct OctPart {
    int32_t ylo, yhi, ca, cb;
};
#define VSLAM_OCT_MAX_PARTS 8

/* Per-level parameters of the GPU quadtree distribution (k_octree). */
struct OctParams {
    int32_t N[VSLAM_MAX_LEVELS];      /* mnFeaturesPerLevel */
    int32_t H[VSLAM_MAX_LEVELS];      /* maxBorderY - minBorderY */
    int32_t nIni[VSLAM_MAX_LEVELS];   /* round(W/H) initial nodes */
    float hX[VSLAM_MAX_LEVELS];       /* (float)W / nIni */
    int32_t cellFirst[VSLAM_MAX_LEVELS + 1];
    int32_t selOff[VSLAM_MAX_LEVELS]; /* offset of the level's result list inside a slot's sel_xyr */
    int32_t selStride;                /* entries per slot in sel_xyr */
    int32_t maxNodes;                 /* list capacity (LDS) */
    int32_t ptsCap;                   /* entries per slot in the key ping-pong arrays */
    int32_t maxIter;                  /* split passes allowed (64; lower only for timing experiments) */
    int32_t wavePrio;                 /* != 0: the quadtree / output-order waves raise their priority (vslam_tuning.wave_prio & 1) */
    void* dbg;                        /* timing stamps (diagnostic builds only) */
    /* k_octree_v4: keys are counted ONCE into the leaves of an implicit quadtree of depth fineD below the initial nodes
     * (nIni << 2*fineD fine cells, <= 16384, counters and prefix sums in LDS) and sorted by leaf; a node of depth
     * d <= fineD is a run of 4^(fineD-d) fine cells, so the child counts of every split pass are differences of one
     * prefix-sum array; a key's leaf is lut[lutOff + x] | lut[lutOff + lutW + y] (vslam::build_oct_lut) */
    int32_t fineD[VSLAM_MAX_LEVELS];
    int32_t lutOff[VSLAM_MAX_LEVELS];  /* uint32 offset of the level's x table in lut; the y table follows it */
    int32_t lutW[VSLAM_MAX_LEVELS];    /* entries of the x table (maxBorderX - minBorderX + 1) */
    int32_t fineLdsOff;                /* the fine arrays live in LDS at this byte offset of the dynamic allocation ... */
    int32_t fineLdsBytes;              /* ... and take this many bytes (the largest level's two arrays) */
    const uint32_t* lut;               /* device; nullptr: no fine grid (walk-per-pass kernel only) */
    /* k_oct_count (vslam_tuning.oct_precount): walk 1 of k_octree_v4 as a launch of its own, a level's keys dealt to up to
     * eight workgroups by leaf row; the leaf counters travel through fineCnt[slot][fineCntOff[level] + leaf] */
    const OctPart* parts;              /* [level][VSLAM_OCT_MAX_PARTS]; nullptr: walk 1 inside k_octree_v4 */
    uint32_t* fineCnt;
    int32_t fineCntStride;             /* words per slot */
    int32_t maxPartCells;              /* most FAST cells a part spans (LDS of k_oct_count) */
    int32_t partBits[VSLAM_MAX_LEVELS];   /* log2(parts of the level) = leading y decisions of the leaf index that select the part */
    int32_t fineCntOff[VSLAM_MAX_LEVELS]; /* word offset of the level's counters inside a slot */
};

/* One stereo pair for the matcher kernels (Frame::ComputeStereoMatches). */
struct StereoJob {
    const vslam_kp* kpsL;
    const uint8_t* descL;
    const vslam_kp* kpsR;
    const uint8_t* descR;
    const int32_t* cntL; /* device: number of left / right keypoints */
    const int32_t* cntR;
    int32_t slotL, slotR;
};
#define VSLAM_MAX_STEREO_JOBS 32
struct StereoJobs {
    StereoJob job[VSLAM_MAX_STEREO_JOBS];
};

/* One (frame 1, frame 2) pair of the batched dense-distance kernel. */
struct MatJob {
    const uint8_t* desc1;
    const uint8_t* desc2;
    int32_t nr, nc;             /* octave-0 keypoints of frame 1 / frame 2 */
    uint32_t idx_off1, idx_off2; /* offsets into the uploaded index list */
    uint32_t q_off, t_off;       /* row offsets of the gathered descriptors in the scratch buffer */
    size_t out_off;              /* byte offset of this pair's nr x nc matrix */
};
#define VSLAM_MAX_MAT_JOBS 64
struct MatJobs {
    MatJob job[VSLAM_MAX_MAT_JOBS];
};

/* One problem of the batched brute-force matcher (k_hamming_top2_batch): nq query against nt train descriptors. */
struct Top2Job {
    const uint32_t* q;
    const uint32_t* t;
    int32_t nq, nt;
    uint32_t row0;   /* first row of this problem in the partial / result arrays (queries of all problems back to back) */
    uint32_t pad;
};
#define VSLAM_MAX_TOP2_JOBS 32
struct Top2Jobs {
    Top2Job job[VSLAM_MAX_TOP2_JOBS];
};

/* k_resize_level_v2: four consecutive output pixels of a row.  base[] (separate u16 array) is the first source
 * column of an 8-byte window that holds all eight taps; sel[j] is the v_perm_b32 selector that pulls output
 * j's two taps out of that window as packed u16 (tap0 | tap1 << 16); coef[j] = a0 | a1 << 16 (11-bit fixed
 * point, cv::resize INTER_LINEAR 8u). */
struct ResizeQuad {
    uint32_t sel[4];
    uint32_t coef[4];
};

/* Fused pyramid (k_pyramid_group, vslam_image_kernels.hip): one entry per (tile, level of the group); same layout as
 * vslam::PyrTileLevel (vslam_host.h), which the host planner fills. */
struct PyrTileDev {
    int16_t c0, nc, r0, nr, sq0, sq1, sr0, sr1;
    uint32_t lds_off, pitch, rt_off;
};
#define VSLAM_PYR_GROUP_LEVELS 4
struct PyrGroupDev { /* by-value kernel argument */
    int32_t l0, nl, ntiles, readable_w0; /* readable_w0: bytes of a source row that may be read (level 0: the image width) */
    const PyrTileDev* tiles;             /* [ntiles][nl + 1] */
    const uint16_t* qbase[VSLAM_PYR_GROUP_LEVELS];
    const ResizeQuad* quads[VSLAM_PYR_GROUP_LEVELS];
    const uint16_t* ytab[VSLAM_PYR_GROUP_LEVELS];
    const int16_t* yb[VSLAM_PYR_GROUP_LEVELS];
    LevelGeom lg[VSLAM_PYR_GROUP_LEVELS + 1]; /* lg[0] = source level */
};

/* vslam_mp_track: per MapPoint, what Frame::isInFrustum left in it */
struct MpTrack {
    float projX, projY, projXR, viewCos;
    int32_t level;
    uint32_t flags;
};

/* vslam_fuse_point / the arguments of one FMatcher::Fuse search (vslam_fuse_search) */
struct FusePoint {
    float pos[3], normal[3];
    float minDistance, maxDistance;
    int32_t valid;
};
struct FuseArgsDev {
    float Rcw[9], tcw[3], Ow[3];
    float Rb[9], tb[3]; /* sim3 == 2 (SearchBySim3): second transform, p2 = Rb * (Rcw * p + tcw) + tb */
    float fx, fy, cx, cy, bf, th, logScaleFactor;
    int32_t imgW, imgH, sim3, gemmFloat, nlevels, nPoints, nKF;
    float scale[VSLAM_MAX_LEVELS], invSigma2[VSLAM_MAX_LEVELS];
    const FusePoint* pts;
    const uint8_t* mpDesc;
    const vslam_kp* kfKps;
    const uint8_t* kfDesc;
    const float* kfURight;
    int32_t *bestIdx, *bestDist;
};

/* One SearchByProjection(CurrentFrame, LastFrame) problem; every pointer is a DEVICE pointer.  nLast / nCur are
 * the counts, or -- when nLastPtr / nCurPtr are set -- capacities with the real counts read from HBM. */
struct SbpProj;
struct SbpJobDev {
    float Tcw[12];
    float fx, fy, cx, cy, mbf, th;
    int32_t forward, backward, checkOri, imgW, imgH, gemmFloat;
    int32_t nLast, nCur;
    const vslam_kp* lastKps;
    const int32_t* nLastPtr;
    const uint8_t* flags;
    const float* x3Dw;
    const uint8_t* mpDesc;
    const vslam_kp* curKps;
    const uint8_t* curDesc;
    const int32_t* nCurPtr;
    const float* uRight;      /* may be null */
    const uint8_t* occupied0; /* may be null */
    SbpProj* proj;            /* scratch: nLast projection records */
    uint32_t* topm;           /* scratch: nLast x M keys */
    int32_t* matchCur;        /* out: nCur */
    int32_t* nmatches;        /* out: 1 */
    int32_t* needSeq;         /* scratch: k_sbp_resolve -> k_sbp_replay hand-over flag */
    /* mode 1 = SearchByProjection(F, vpMapPoints): queries come pre-projected (Frame::isInFrustum) */
    const MpTrack* mps;
    float nnratio;
    int32_t mode;
};
/* mode 2 = SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (relocalisation; one job per call): the
 * KeyFrame's MapPoints are projected with PredictScale; accept threshold ORBdist instead of TH_HIGH */
struct SbpKfDev {
    int32_t thHigh; /* accept threshold + 1; 0 = TH_HIGH */
    float logScaleFactor;
    const float* minDist;
    const float* maxDist;
    float ow[3];
    /* mode 3 = SearchByProjection(pKF, Scw, vpPoints, [vpPointsKFs,] vpMatched, [vpMatchedKF,] th, ratioHamming)
     * (loop closing): depth test, viewing-angle test against the MapPoint normals, levels [l-1, l];
     * projKind 0: Pinhole::project (fmatcher.cpp:796), 1: invz = 1/z, u = fx*(x*invz)+cx (:908-913) */
    int32_t projKind;
    const float* normals;
};
#define VSLAM_MAX_SBP_JOBS 16
struct SbpJobs { /* by-value kernel argument (< 4 KB) */
    SbpJobDev job[VSLAM_MAX_SBP_JOBS];
    float scale[VSLAM_MAX_LEVELS];
    int32_t nlevels, M;
    SbpKfDev kf;
};

/* k_unproject_stereo: one stereo pair's left keypoints -> world points (Frame::UnprojectStereo) */
struct UnprojJob {
    float Twc[12]; /* rows [mRwc | mOw] */
    const vslam_kp* kps;
    const int32_t* nPtr;
    const float* depth;
    float* x3Dw;
    uint8_t* flags;
};
struct UnprojJobs {
    UnprojJob job[VSLAM_MAX_SBP_JOBS];
    float cx, cy, invfx, invfy;
    int32_t cap, observations, gemmFloat;
};

/* Up to four dword-granular ranges for k_copy_ranges (src == nullptr: zero-fill). */
struct CopyRanges {
    void* dst[4];
    const void* src[4];
    size_t bytes[4];
    int n;
};

/* One (frame 1, frame 2) problem of k_si_topm / k_si_replay; every pointer is a device pointer. */
struct InitJob {
    const vslam_kp* k1;
    const uint8_t* d1;
    const int32_t* cnt1;
    const vslam_kp* k2;
    const uint8_t* d2;
    const int32_t* cnt2;
    const float* prev; /* vbPrevMatched (2*n1 floats) or NULL = frame 1's keypoint positions */
};
struct InitJobs {
    InitJob job[VSLAM_MAX_MAT_JOBS];
};

#endif
