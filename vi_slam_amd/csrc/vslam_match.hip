/* vslam_match.hip -- matcher half of the C ABI: Frame::ComputeStereoMatches on the device and the
 * host-replayed FMatcher::SearchForInitialization (distances from the device). */
#include "vslam_ctx.h"

/* ------------------------------------------------------------------ stereo */
struct StereoScratch { /* carved out of fe->d_stereo */
    uint32_t* best;
    float* uRight;
    float* depth;
    int32_t* sad;
    uint8_t* rows; /* row table of the right keypoints (vk_stereo_rows_bytes) */
    int max_band;
};

static int stereo_scratch(vslam_fe* fe, int njobs, int capR, StereoScratch* s) {
    const size_t n = (size_t)njobs * fe->cap;
    /* rows of a right keypoint's band: floor(y - 2s) .. ceil(y + 2s), s <= the coarsest level's scale factor */
    s->max_band = 2 * (int)std::ceil(2.0f * fe->tab.scale[fe->p.nlevels - 1]) + 3;
    const size_t rows_bytes = vk_stereo_rows_bytes(njobs, fe->p.height, s->max_band, capR);
    int rc = vslam_ensure(&fe->d_stereo, &fe->stereo_bytes, n * 16 + rows_bytes + 16);
    if (rc) return rc;
    s->best = (uint32_t*)fe->d_stereo;
    s->uRight = (float*)(s->best + n);
    s->depth = s->uRight + n;
    s->sad = (int32_t*)(s->depth + n);
    s->rows = (uint8_t*)(s->sad + n);
    if (fe->h_stereo_bytes < n * 8) {
        if (fe->h_stereo) HIPCHK(hipHostFree(fe->h_stereo));
        fe->h_stereo = nullptr;
        fe->h_stereo_bytes = 0;
        HIPCHK((hipError_t)vslam_pinned_alloc((void**)&fe->h_stereo, n * 8));
        fe->h_stereo_bytes = n * 8;
    }
    return VSLAM_OK;
}

/* enqueue the matcher kernels and the D2H of mvuRight/mvDepth (whole capacity: the keypoint counts may
 * not be known on the host yet); nothing waits */
static int enqueue_stereo(vslam_fe* feL, vslam_fe* feR, int npairs, const int* slotsL, const int* slotsR,
                          float bf, float fx, bool in_block = false) {
    if (feL->p.device != feR->p.device || feL->p.width != feR->p.width || feL->p.height != feR->p.height ||
        feL->p.nlevels != feR->p.nlevels || feL->p.scale_factor != feR->p.scale_factor) {
        g_err = "left/right extractors must share device and geometry";
        return VSLAM_ERR_INVALID;
    }
    StereoJobs jobs;
    memset(&jobs, 0, sizeof(jobs));
    for (int j = 0; j < npairs; j++) {
        const int sL = slotsL[j], sR = slotsR[j];
        if (sL < 0 || sL >= feL->B || sR < 0 || sR >= feR->B || !feL->src.l0[sL] || !feR->src.l0[sR]) {
            g_err = "slot not extracted";
            return VSLAM_ERR_INVALID;
        }
        StereoJob& jb = jobs.job[j];
        jb.kpsL = feL->d_kps + (size_t)sL * feL->cap;
        jb.descL = feL->d_desc + (size_t)sL * feL->cap * 32;
        jb.kpsR = feR->d_kps + (size_t)sR * feR->cap;
        jb.descR = feR->d_desc + (size_t)sR * feR->cap * 32;
        jb.cntL = feL->d_counts + sL * 4;
        jb.cntR = feR->d_counts + sR * 4;
        jb.slotL = sL;
        jb.slotR = sR;
    }
    if (feL != feR) HIPCHK(vslam_stream_wait(feR->stream)); /* right results must be complete */
    StereoScratch sc;
    int rc = stereo_scratch(feL, npairs, feR->cap, &sc);
    feL->stereo_capR = feR->cap;
    if (rc) return rc;
    if (in_block) { /* mvuRight | mvDepth behind the extraction's results in the context's result block: one delivery */
        sc.uRight = (float*)(feL->d_res + feL->res_feat_bytes);
        sc.depth = sc.uRight + (size_t)npairs * feL->cap;
    }
    feL->d_stereo_u = sc.uRight; /* device-side consumers (UnprojectStereo, vslam_stereo_points_buffers) read them here */
    feL->d_stereo_depth = sc.depth;
    /* frame.cpp:853-855: mb = mbf/fx (frame.cpp:157), minZ = mb, maxD = mbf/minZ */
    const float mb = bf / fx;
    const float maxD = bf / mb;
    hipStream_t st = feL->stream;
    vk_stereo(st, jobs, npairs, feL->cap, feR->cap, feL->geom, feL->d_pyr, feL->slot_stride, feL->src, feR->d_pyr,
              feR->slot_stride, feR->src, bf, maxD, sc.best, sc.uRight, sc.depth, sc.sad, feL->cap, sc.max_band, sc.rows,
              wave_prio_on(feL->tune, 4));
    HIPCHK(hipGetLastError());
    const size_t n = (size_t)npairs * feL->cap;
    CopyRanges R;
    memset(&R, 0, sizeof(R));
    if (in_block && feL->deliver_deferred) {
        /* the extraction of this call left its delivery to us (want_host = 2): counts | keypoints | descriptors | mvuRight |
         * mvDepth are contiguous -> ONE transfer for the whole stereo step */
        feL->deliver_deferred = false;
        R.dst[0] = feL->h_res;
        R.src[0] = feL->d_res;
        R.bytes[0] = feL->res_feat_bytes + n * 8;
        feL->h_stereo_cur = (float*)(feL->h_res + feL->res_feat_bytes);
    } else {
        R.dst[0] = in_block ? (void*)(feL->h_res + feL->res_feat_bytes) : (void*)feL->h_stereo;
        R.src[0] = sc.uRight; /* uRight | depth */
        R.bytes[0] = n * 8;
        feL->h_stereo_cur = in_block ? (float*)(feL->h_res + feL->res_feat_bytes) : feL->h_stereo;
    }
    R.n = 1;
    vslam_count_delivery(feL, vk_copy_ranges(st, R, feL->tune), R);
    HIPCHK(hipGetLastError());
    feL->stereo_pairs = npairs;
    for (int j = 0; j < npairs; j++) feL->stereo_slotL[j] = slotsL[j];
    return VSLAM_OK;
}

static void deliver_stereo(vslam_fe* feL, float* const* u_right, float* const* depth) {
    const size_t n = (size_t)feL->stereo_pairs * feL->cap;
    for (int j = 0; j < feL->stereo_pairs; j++) {
        const int nL = feL->n_out[feL->stereo_slotL[j]];
        if (!nL) continue;
        if (u_right && u_right[j]) memcpy(u_right[j], feL->h_stereo_cur + (size_t)j * feL->cap, (size_t)nL * 4);
        if (depth && depth[j]) memcpy(depth[j], feL->h_stereo_cur + n + (size_t)j * feL->cap, (size_t)nL * 4);
    }
}

extern "C" int vslam_stereo_match_batch(vslam_fe* feL, vslam_fe* feR, int npairs, const int* slotsL,
                                        const int* slotsR, float bf, float fx, float* const* u_right,
                                        float* const* depth) {
    if (!feL || !feR || npairs < 1 || npairs > VSLAM_MAX_STEREO_JOBS || !slotsL || !slotsR || !u_right || !depth) {
        g_err = "invalid arguments";
        return VSLAM_ERR_INVALID;
    }
    HIPCHK(hipSetDevice(feL->p.device));
    int rc = enqueue_stereo(feL, feR, npairs, slotsL, slotsR, bf, fx);
    if (rc) return rc;
    HIPCHK(vslam_stream_wait(feL->stream));
    deliver_stereo(feL, u_right, depth);
    return VSLAM_OK;
}

extern "C" int vslam_stereo_match(vslam_fe* feL, int sL, vslam_fe* feR, int sR, float bf, float fx,
                                  float* u_right, float* depth) {
    float* u[1] = {u_right};
    float* d[1] = {depth};
    return vslam_stereo_match_batch(feL, feR, 1, &sL, &sR, bf, fx, u, d);
}

/* The extraction + stereo section of Frame::Frame(stereo) (frame.cpp:102-132) for npairs frames in one
 * enqueue: images are L0,R0,L1,R1,...; slot 2j = left, 2j+1 = right of pair j. */
extern "C" int vslam_frame_stereo_batch_async(vslam_fe* fe, int npairs, const uint8_t* const* imgs, size_t pitch,
                                              int imgs_on_device, float bf, float fx, int want_host) {
    if (!fe || npairs < 1 || npairs > VSLAM_MAX_STEREO_JOBS || 2 * npairs > fe->B ||
        (imgs_on_device != VSLAM_IMGS_STAGED && (!imgs || pitch < (size_t)fe->p.width)) || imgs_on_device < 0 ||
        imgs_on_device > VSLAM_IMGS_STAGED) {
        g_err = "invalid arguments";
        return VSLAM_ERR_INVALID;
    }
    /* a full batch whose results go to the host: mvuRight / mvDepth live behind the extraction's results in the result
     * block (the region SearchForInitialization's outputs use in a context that runs that matcher: whoever comes first
     * in a context's life owns it) and the whole step leaves in one transfer */
    const bool blk = want_host && fe->dev_octree && 2 * npairs == fe->B && fe->res_init_bytes >= (size_t)npairs * fe->cap * 8 &&
                     fe->block_region_owner != VSLAM_REGION_INIT && fe->d_res;
    if (blk) fe->block_region_owner = VSLAM_REGION_STEREO;
    /* stereo frames: the contexts of a pipelined caller are independent of each other and the step is the sum of the wide
     * kernels' single-context times; the descriptor kernel alone is 8-10 % shorter with one keypoint per wave (48 VGPRs, 8 waves
     * per SIMD) than with four (110 VGPRs, 4 waves, the trigonometry of four keypoints in one pass: fewer instructions, which
     * is what the VALU-bound mono pipeline wants): KITTI stereo +3.3 %, profiles/r04_describe_kpw_ab.txt */
    fe->desc_kpw_hint = 1;
    int rc = vslam_enqueue_extract(fe, 2 * npairs, imgs, pitch, imgs_on_device, 0, 0, blk ? 2 : (want_host != 0));
    fe->desc_kpw_hint = -1;
    if (rc == VSLAM_OK) {
        int sl[VSLAM_MAX_STEREO_JOBS], sr[VSLAM_MAX_STEREO_JOBS];
        for (int j = 0; j < npairs; j++) {
            sl[j] = 2 * j;
            sr[j] = 2 * j + 1;
        }
        rc = enqueue_stereo(fe, fe, npairs, sl, sr, bf, fx, blk);
    }
    if (rc != VSLAM_OK) hipStreamSynchronize(fe->stream);
    return rc;
}

extern "C" int vslam_frame_stereo_wait(vslam_fe* fe, vslam_kp* const* kps, uint8_t* const* desc, int cap, int* n,
                                       float* const* u_right, float* const* depth) {
    if (!fe || fe->last_nimg < 2) {
        g_err = "nothing enqueued";
        return VSLAM_ERR_INVALID;
    }
    int rc = vslam_finish_extract(fe, fe->last_nimg);
    if (rc != VSLAM_OK) return rc;
    rc = vslam_deliver(fe, fe->last_nimg, kps, desc, cap, n, nullptr);
    if (rc != VSLAM_OK) return rc;
    deliver_stereo(fe, u_right, depth);
    return VSLAM_OK;
}

/* ------------------------------------------------------------------ SearchForInitialization on the device */
static int init_scratch(vslam_fe* fe, int npairs) {
    const size_t per = (size_t)fe->cap * 4 + (size_t)fe->cap * 8 + 16;
    const bool stereo_owns = fe->init_in_block && fe->block_region_owner == VSLAM_REGION_STEREO;
    if (per * npairs <= fe->init_bytes && !stereo_owns) { /* the result block's own region: up to max_batch pairs */
        if (fe->init_in_block) fe->block_region_owner = VSLAM_REGION_INIT;
        return VSLAM_OK;
    }
    if (fe->init_in_block) { /* more pairs than image slots, or the stereo matcher's outputs live there: buffers of their own from now on */
        fe->init_in_block = false;
        fe->d_init = nullptr;
        fe->h_init = nullptr;
        fe->init_bytes = fe->h_init_bytes = 0;
    }
    int rc = vslam_ensure((void**)&fe->d_init, &fe->init_bytes, per * npairs);
    if (rc) return rc;
    if (fe->h_init_bytes < per * npairs) {
        if (fe->h_init) HIPCHK(hipHostFree(fe->h_init));
        fe->h_init = nullptr;
        fe->h_init_bytes = 0;
        HIPCHK((hipError_t)vslam_pinned_alloc((void**)&fe->h_init, per * npairs));
        fe->h_init_bytes = per * npairs;
    }
    return VSLAM_OK;
}

extern "C" int vslam_search_init_dev_async(vslam_fe* fe, int npairs, const vslam_init_job* jobs, int img_w,
                                           int img_h, int window, float nnratio, int check_orientation) {
    if (!fe || npairs < 1 || npairs > VSLAM_MAX_MAT_JOBS || !jobs) {
        g_err = "invalid arguments";
        return VSLAM_ERR_INVALID;
    }
    if (!(nnratio >= 0.2f)) { /* see the distance clamp in vslam_init_kernel.hip */
        g_err = "SearchForInitialization on the device needs nnratio >= 0.2";
        return VSLAM_ERR_UNSUPPORTED;
    }
    HIPCHK(hipSetDevice(fe->p.device));
    InitJobs J;
    memset(&J, 0, sizeof(J));
    for (int j = 0; j < npairs; j++) {
        if (!jobs[j].dev_kps1 || !jobs[j].dev_desc1 || !jobs[j].dev_n1 || !jobs[j].dev_kps2 || !jobs[j].dev_desc2 ||
            !jobs[j].dev_n2) {
            g_err = "null device pointer in job";
            return VSLAM_ERR_INVALID;
        }
        J.job[j].k1 = jobs[j].dev_kps1;
        J.job[j].d1 = jobs[j].dev_desc1;
        J.job[j].cnt1 = jobs[j].dev_n1;
        J.job[j].k2 = jobs[j].dev_kps2;
        J.job[j].d2 = jobs[j].dev_desc2;
        J.job[j].cnt2 = jobs[j].dev_n2;
        J.job[j].prev = jobs[j].dev_prev_matched;
    }
    int rc = init_scratch(fe, npairs);
    if (rc) return rc;
    /* octave-0 keypoints of frame 2 kept in LDS: the level-0 quota (+ the quadtree's overshoot) bounds them
     * for extractor output; foreign inputs are clipped to the context's capacity */
    const int max_c2 = std::min(fe->cap, std::max(fe->tab.quota[0] + 8, 64));
    if (max_c2 > 4096) { /* candidate slot is a 12-bit field of the ordering key */
        g_err = "SearchForInitialization on the device supports at most 4096 octave-0 keypoints per frame";
        return VSLAM_ERR_UNSUPPORTED;
    }
    const size_t lds = vk_search_init_lds(fe->cap, max_c2);
    if (lds > 150 * 1024) {
        g_err = "SearchForInitialization: keypoint capacity too large for the LDS-resident matcher";
        return VSLAM_ERR_UNSUPPORTED;
    }
    if (!fe->init_lds_set) {
        if (vk_search_init_set_max_lds(150 * 1024) != 0) {
            g_err = "hipFuncSetAttribute(k_si_*) failed";
            return VSLAM_ERR_HIP;
        }
        fe->init_lds_set = true;
    }
    /* sorted candidate prefix per query; VSLAM_INIT_TOPM=<1..16> (tests force the re-scan path with 2) */
    const int M = std::min(16, std::max(1, tune_or(fe->tune.init_topm, 8)));
    rc = vslam_ensure((void**)&fe->d_init_scratch, &fe->init_scratch_bytes,
                      vk_search_init_scratch_bytes(npairs, max_c2, M) + 16);
    if (rc) return rc;
    if (!fe->d_init_fb) {
        HIPCHK(hipMalloc((void**)&fe->d_init_fb, 16));
        HIPCHK(hipMemset(fe->d_init_fb, 0, 16));
    }
    const size_t nm = (size_t)npairs * fe->cap;
    int32_t* d_m = (int32_t*)fe->d_init;
    float* d_p = (float*)(d_m + nm);
    int32_t* d_n = (int32_t*)(d_p + 2 * nm);
    vk_search_init(fe->stream, J, npairs, fe->cap, img_w, img_h, window, nnratio, check_orientation, d_m, d_p, d_n,
                   max_c2, M, fe->d_init_scratch, fe->d_init_fb, fe->tune);
    HIPCHK(hipGetLastError());
    CopyRanges R;
    memset(&R, 0, sizeof(R));
    if (fe->deliver_deferred && fe->init_in_block) {
        /* the extraction of this step asked for its delivery to be deferred (want_host = 2): counts, keypoints, descriptors
         * and the matcher's outputs are contiguous in the result block -> ONE transfer for the whole step */
        fe->deliver_deferred = false;
        R.dst[0] = fe->h_res;
        R.src[0] = fe->d_res;
        R.bytes[0] = fe->res_feat_bytes + ((nm * 12 + (size_t)npairs * 4 + 15) & ~(size_t)15);
    } else {
        R.dst[0] = fe->h_init;
        R.src[0] = fe->d_init;
        R.bytes[0] = nm * 12 + (size_t)npairs * 4;
    }
    R.n = 1;
    vslam_count_delivery(fe, vk_copy_ranges(fe->stream, R, fe->tune), R);
    HIPCHK(hipGetLastError());
    fe->init_pairs = npairs;
    return VSLAM_OK;
}

extern "C" int vslam_search_init_dev_wait(vslam_fe* fe, const int* n1, int32_t* const* matches12,
                                          float* const* prev_matched, int* nmatches) {
    if (!fe || fe->init_pairs < 1) {
        g_err = "nothing enqueued";
        return VSLAM_ERR_INVALID;
    }
    HIPCHK(hipSetDevice(fe->p.device));
    HIPCHK(vslam_stream_wait(fe->stream));
    const int npairs = fe->init_pairs;
    const size_t nm = (size_t)npairs * fe->cap;
    const int32_t* h_m = (const int32_t*)fe->h_init;
    const float* h_p = (const float*)(h_m + nm);
    const int32_t* h_n = (const int32_t*)(h_p + 2 * nm);
    for (int j = 0; j < npairs; j++) {
        const int n = n1 ? std::min(n1[j], fe->cap) : 0;
        if (matches12 && matches12[j] && n) memcpy(matches12[j], h_m + (size_t)j * fe->cap, (size_t)n * 4);
        if (prev_matched && prev_matched[j] && n) memcpy(prev_matched[j], h_p + (size_t)j * fe->cap * 2, (size_t)n * 8);
        if (nmatches) nmatches[j] = h_n[j];
    }
    for (int j = 0; j < npairs; j++)
        if (h_n[j] < 0) { /* k_si_topm found more octave-0 keypoints than its lists hold: nothing was truncated silently */
            g_err = "SearchForInitialization on the device: more octave-0 keypoints in a frame than the context's "
                    "level-0 quota allows (use the host-keypoint entry point, which falls back to the host replay)";
            return VSLAM_ERR_CAPACITY;
        }
    return VSLAM_OK;
}

/* ------------------------------------------------------------------ SearchForInitialization */
extern "C" int vslam_search_for_initialization_batch(vslam_fe* fe, int npairs, const vslam_kp* const* kps1,
                                                     const uint8_t* const* dev_desc1, const int* n1,
                                                     const vslam_kp* const* kps2, const uint8_t* const* dev_desc2,
                                                     const int* n2, int img_w, int img_h,
                                                     float* const* prev_matched, int32_t* const* matches12,
                                                     int window, float nnratio, int check_orientation,
                                                     int* nmatches) {
    if (!fe || npairs < 1 || npairs > VSLAM_MAX_MAT_JOBS || !kps1 || !dev_desc1 || !n1 || !kps2 || !dev_desc2 ||
        !n2 || !prev_matched || !matches12 || !nmatches) {
        g_err = "invalid arguments";
        return VSLAM_ERR_INVALID;
    }
    for (int j = 0; j < npairs; j++)
        if (n1[j] < 0 || n2[j] < 0 || (n1[j] && (!kps1[j] || !dev_desc1[j] || !prev_matched[j] || !matches12[j])) ||
            (n2[j] && (!kps2[j] || !dev_desc2[j]))) {
            g_err = "invalid arguments";
            return VSLAM_ERR_INVALID;
        }
    HIPCHK(hipSetDevice(fe->p.device));
    {
        /* default: the whole matcher on the GPU (k_si_topm + k_si_replay).  VSLAM_INIT_MATCH=host keeps the distance
         * matrices on the GPU and replays the order-dependent part on the host (cross-check path). */
        const bool host_mode = fe->tune.init_match_host == 1;
        bool fits = true;
        for (int j = 0; j < npairs; j++) fits = fits && n1[j] <= fe->cap && n2[j] <= fe->cap;
        fits = fits && fe->tab.quota[0] + 8 <= 4096 && nnratio >= 0.2f;
        /* the device matcher keeps at most max_c2 octave-0 keypoints per frame (sized for THIS context's level-0
         * quota); keypoints from another extractor configuration can exceed that while n <= cap: host replay then */
        const int max_c2 = std::min(fe->cap, std::max(fe->tab.quota[0] + 8, 64));
        for (int j = 0; j < npairs && fits; j++) {
            int o1 = 0, o2 = 0;
            for (int i = 0; i < n1[j]; i++) o1 += kps1[j][i].octave == 0;
            for (int i = 0; i < n2[j]; i++) o2 += kps2[j][i].octave == 0;
            fits = o1 <= max_c2 && o2 <= max_c2;
        }
        if (!host_mode && fits) {
            /* upload keypoints, counts and vbPrevMatched of every pair, run, download */
            size_t bytes = 0;
            for (int j = 0; j < npairs; j++) bytes += ((size_t)(n1[j] + n2[j]) * sizeof(vslam_kp) + (size_t)n1[j] * 8 + 16 + 63) & ~(size_t)63;
            int rc = vslam_ensure((void**)&fe->d_tmp_desc[1], &fe->tmp_desc_bytes[1], bytes + 64);
            if (rc) return rc;
            std::vector<uint8_t> stage(bytes + 64);
            std::vector<vslam_init_job> jobs(npairs);
            size_t off = 0;
            for (int j = 0; j < npairs; j++) {
                uint8_t* base = fe->d_tmp_desc[1] + off;
                uint8_t* hb = stage.data() + off;
                int32_t cnt[4] = {n1[j], n2[j], 0, 0};
                memcpy(hb, cnt, 16);
                memcpy(hb + 16, kps1[j], (size_t)n1[j] * sizeof(vslam_kp));
                memcpy(hb + 16 + (size_t)n1[j] * sizeof(vslam_kp), kps2[j], (size_t)n2[j] * sizeof(vslam_kp));
                const size_t poff = 16 + (size_t)(n1[j] + n2[j]) * sizeof(vslam_kp);
                memcpy(hb + poff, prev_matched[j], (size_t)n1[j] * 8);
                jobs[j].dev_n1 = (const int32_t*)base;
                jobs[j].dev_n2 = (const int32_t*)base + 1;
                jobs[j].dev_kps1 = (const vslam_kp*)(base + 16);
                jobs[j].dev_kps2 = (const vslam_kp*)(base + 16 + (size_t)n1[j] * sizeof(vslam_kp));
                jobs[j].dev_prev_matched = (const float*)(base + poff);
                jobs[j].dev_desc1 = dev_desc1[j];
                jobs[j].dev_desc2 = dev_desc2[j];
                off += (poff + (size_t)n1[j] * 8 + 63) & ~(size_t)63;
            }
            HIPCHK(hipMemcpyAsync(fe->d_tmp_desc[1], stage.data(), off, hipMemcpyHostToDevice, fe->stream));
            HIPCHK(vslam_stream_wait(fe->stream)); /* stage is pageable and goes out of scope */
            rc = vslam_search_init_dev_async(fe, npairs, jobs.data(), img_w, img_h, window, nnratio, check_orientation);
            if (rc) return rc;
            return vslam_search_init_dev_wait(fe, n1, matches12, prev_matched, nmatches);
        }
    }
    /* only octave-0 keypoints take part (fmatcher.cpp:999-1003: level1 > 0 -> continue; window query
     * restricted to [level1, level1]) */
    std::vector<std::vector<int>> row_of(npairs), col_of(npairs);
    std::vector<int32_t> idx;
    MatJobs jobs;
    memset(&jobs, 0, sizeof(jobs));
    size_t rows_total = 0, out_total = 0;
    int maxr = 0, maxc = 0;
    for (int j = 0; j < npairs; j++) {
        MatJob& jb = jobs.job[j];
        row_of[j].assign(n1[j], -1);
        col_of[j].assign(n2[j], -1);
        jb.idx_off1 = (uint32_t)idx.size();
        int nr = 0, nc = 0;
        for (int i = 0; i < n1[j]; i++)
            if (kps1[j][i].octave == 0) { row_of[j][i] = nr++; idx.push_back(i); }
        jb.idx_off2 = (uint32_t)idx.size();
        for (int i = 0; i < n2[j]; i++)
            if (kps2[j][i].octave == 0) { col_of[j][i] = nc++; idx.push_back(i); }
        jb.desc1 = dev_desc1[j];
        jb.desc2 = dev_desc2[j];
        jb.nr = nr;
        jb.nc = nc;
        jb.q_off = (uint32_t)rows_total;
        jb.t_off = (uint32_t)(rows_total + nr);
        jb.out_off = out_total;
        rows_total += (size_t)nr + nc;
        out_total += ((size_t)nr * nc + 15) & ~(size_t)15;
        maxr = std::max(maxr, nr);
        maxc = std::max(maxc, nc);
    }
    std::vector<uint8_t> dmat(std::max<size_t>(out_total, 16));
    if (maxr && maxc) {
        int rc;
        const size_t ib = (idx.size() * 4 + 255) & ~(size_t)255;
        if ((rc = vslam_ensure((void**)&fe->d_tmp_desc[0], &fe->tmp_desc_bytes[0], ib + rows_total * 32))) return rc;
        if ((rc = vslam_ensure((void**)&fe->d_dmat, &fe->dmat_bytes, out_total))) return rc;
        int32_t* d_idx = (int32_t*)fe->d_tmp_desc[0];
        uint8_t* d_rows = fe->d_tmp_desc[0] + ib;
        hipStream_t st = fe->stream;
        HIPCHK(hipMemcpyAsync(d_idx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice, st));
        vk_hamming_matrix_batch(st, jobs, npairs, maxr, maxc, d_idx, d_rows, fe->d_dmat);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(dmat.data(), fe->d_dmat, out_total, hipMemcpyDeviceToHost, st));
        HIPCHK(vslam_stream_wait(st));
    }
    /* the stealing / ratio / rotation-histogram logic is order dependent: replayed on the host, one
     * pair per worker */
    fe->pool->parallel_for(npairs, [&](int j) {
        const MatJob& jb = jobs.job[j];
        nmatches[j] = vslam::search_for_initialization_replay(
            kps1[j], n1[j], kps2[j], n2[j], dmat.data() + jb.out_off, row_of[j].data(), col_of[j].data(),
            std::max(jb.nc, 1), img_w, img_h, prev_matched[j], matches12[j], window, nnratio, check_orientation != 0);
    });
    return VSLAM_OK;
}

extern "C" int vslam_search_for_initialization(vslam_fe* fe, const vslam_kp* kps1, const uint8_t* dev_desc1,
                                               int n1, const vslam_kp* kps2, const uint8_t* dev_desc2, int n2,
                                               int img_w, int img_h, float* prev_matched, int32_t* matches12,
                                               int window, float nnratio, int check_orientation,
                                               int* nmatches) {
    /* the reference tolerates empty frames: nothing to match */
    static const vslam_kp no_kp = {0, 0, 0, 0, 0, 0, 0};
    static const uint8_t* no_desc = (const uint8_t*)&no_kp;
    static float no_prev[2];
    static int32_t no_match[1];
    const vslam_kp* k1 = n1 ? kps1 : &no_kp;
    const vslam_kp* k2 = n2 ? kps2 : &no_kp;
    const uint8_t* d1 = n1 ? dev_desc1 : no_desc;
    const uint8_t* d2 = n2 ? dev_desc2 : no_desc;
    float* pm = n1 ? prev_matched : no_prev;
    int32_t* m = n1 ? matches12 : no_match;
    return vslam_search_for_initialization_batch(fe, 1, &k1, &d1, &n1, &k2, &d2, &n2, img_w, img_h, &pm, &m, window,
                                                 nnratio, check_orientation, nmatches);
}

/* ------------------------------------------------------------------ diagnostics */
static int dbg_buffers(vslam_fe* fe, int n, float** a, float** b, float** c) {
    int rc = vslam_ensure((void**)&fe->d_tmp_desc[1], &fe->tmp_desc_bytes[1], (size_t)n * 12);
    if (rc) return rc;
    *a = (float*)fe->d_tmp_desc[1];
    *b = *a + n;
    *c = *b + n;
    return VSLAM_OK;
}

extern "C" int vslam_dbg_sincos(vslam_fe* fe, const float* x, int n, float* sin_out, float* cos_out) {
    if (!fe || n < 0 || (n && (!x || !sin_out || !cos_out))) return VSLAM_ERR_INVALID;
    if (!n) return VSLAM_OK;
    HIPCHK(hipSetDevice(fe->p.device));
    float *dx, *ds, *dc;
    int rc = dbg_buffers(fe, n, &dx, &ds, &dc);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(dx, x, (size_t)n * 4, hipMemcpyHostToDevice, fe->stream));
    vk_dbg_sincos(fe->stream, dx, n, ds, dc);
    HIPCHK(hipMemcpyAsync(sin_out, ds, (size_t)n * 4, hipMemcpyDeviceToHost, fe->stream));
    HIPCHK(hipMemcpyAsync(cos_out, dc, (size_t)n * 4, hipMemcpyDeviceToHost, fe->stream));
    HIPCHK(vslam_stream_wait(fe->stream));
    return VSLAM_OK;
}

extern "C" int vslam_dbg_logf(vslam_fe* fe, const float* x, int n, float* y) {
    if (!fe || n < 0 || (n && (!x || !y))) return VSLAM_ERR_INVALID;
    if (!n) return VSLAM_OK;
    HIPCHK(hipSetDevice(fe->p.device));
    float *dx, *dy, *dz;
    int rc = dbg_buffers(fe, n, &dx, &dy, &dz);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(dx, x, (size_t)n * 4, hipMemcpyHostToDevice, fe->stream));
    vk_dbg_logf(fe->stream, dx, n, dy);
    HIPCHK(hipMemcpyAsync(y, dy, (size_t)n * 4, hipMemcpyDeviceToHost, fe->stream));
    HIPCHK(vslam_stream_wait(fe->stream));
    return VSLAM_OK;
}

extern "C" int vslam_dbg_fast_atan2(vslam_fe* fe, const float* y, const float* x, int n, int fma, float* deg) {
    if (!fe || n < 0 || (n && (!x || !y || !deg))) return VSLAM_ERR_INVALID;
    if (!n) return VSLAM_OK;
    HIPCHK(hipSetDevice(fe->p.device));
    float *dy, *dx, *da;
    int rc = dbg_buffers(fe, n, &dy, &dx, &da);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(dy, y, (size_t)n * 4, hipMemcpyHostToDevice, fe->stream));
    HIPCHK(hipMemcpyAsync(dx, x, (size_t)n * 4, hipMemcpyHostToDevice, fe->stream));
    vk_dbg_atan2(fe->stream, dy, dx, n, fma, da);
    HIPCHK(hipMemcpyAsync(deg, da, (size_t)n * 4, hipMemcpyDeviceToHost, fe->stream));
    HIPCHK(vslam_stream_wait(fe->stream));
    return VSLAM_OK;
}

extern "C" int vslam_dbg_search_init_fallbacks(vslam_fe* fe, int* count) {
    if (!fe || !count) return VSLAM_ERR_INVALID;
    *count = 0;
    if (!fe->d_init_fb) return VSLAM_OK;
    HIPCHK(hipSetDevice(fe->p.device));
    HIPCHK(vslam_stream_wait(fe->stream));
    HIPCHK(hipMemcpy(count, fe->d_init_fb, 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(fe->d_init_fb, 0, 4));
    return VSLAM_OK;
}

/* rounds / queries / pairs of the replay wave since the last call (k_si_replay's own counters; read-and-reset) */
extern "C" int vslam_dbg_search_init_replay_stats(vslam_fe* fe, int* rounds, int* queries, int* pairs) {
    if (!fe) return VSLAM_ERR_INVALID;
    int v[4] = {0, 0, 0, 0};
    if (fe->d_init_fb) {
        HIPCHK(hipSetDevice(fe->p.device));
        HIPCHK(vslam_stream_wait(fe->stream));
        HIPCHK(hipMemcpy(v, fe->d_init_fb, 16, hipMemcpyDeviceToHost));
        HIPCHK(hipMemset(fe->d_init_fb + 1, 0, 12));
    }
    if (rounds) *rounds = v[1];
    if (queries) *queries = v[2];
    if (pairs) *pairs = v[3];
    return VSLAM_OK;
}

/* ------------------------------------------------------------------ SearchByProjection(CurrentFrame, LastFrame) */
extern "C" int vslam_projection_direction(const float* Tcw, const float* Tlw, float mb, int mono, int gemm_float,
                                          int* forward, int* backward) {
    if (!Tcw || !Tlw || !forward || !backward) return VSLAM_ERR_INVALID;
    /* twc = -Rcw.t()*tcw ; tlc = Rlw*twc + tlw (fmatcher.cpp:2482-2492), each ONE cv::gemm */
    float twc[3], tlc2;
    for (int r = 0; r < 3; r++) {
        if (!gemm_float) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += (double)Tcw[k * 4 + r] * (double)Tcw[k * 4 + 3];
            twc[r] = (float)(s * -1.0);
        } else {
            float s = 0;
            for (int k = 0; k < 3; k++) s += Tcw[k * 4 + r] * Tcw[k * 4 + 3];
            twc[r] = -s;
        }
    }
    if (!gemm_float) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)Tlw[8 + k] * (double)twc[k];
        tlc2 = (float)(s + (double)Tlw[11]);
    } else {
        float s = 0;
        for (int k = 0; k < 3; k++) s += Tlw[8 + k] * twc[k];
        tlc2 = s + Tlw[11];
    }
    *forward = (tlc2 > mb && !mono) ? 1 : 0;
    *backward = (-tlc2 > mb && !mono) ? 1 : 0;
    return VSLAM_OK;
}

/* extras of the KeyFrame overload (fmatcher.cpp:2689-2811); null for SearchByProjection(CurrentFrame, LastFrame) */
struct SbpKfExtras {
    const float *min_dist, *max_dist;
    float ow[3], log_scale_factor;
    int orb_dist;
    const float* normals = nullptr; /* non-null: the Scw overloads (mode 3) */
    int proj_kind = 0;
};

static int sbp_frame_impl(vslam_fe* fe, const vslam_proj_params* p, const vslam_kp* last_kps_host, int n_last,
                          const uint8_t* last_flags, const float* last_x3dw, const uint8_t* mp_desc_host,
                          const vslam_kp* dev_cur_kps, const uint8_t* dev_cur_desc, int n_cur,
                          const float* cur_u_right_host, const uint8_t* cur_occupied_host, const SbpKfExtras* kf,
                          int32_t* match_cur, int* nmatches) {
    if (!fe || !p || n_last < 0 || n_cur < 0 || !nmatches || (n_cur && (!dev_cur_kps || !dev_cur_desc || !match_cur)) ||
        (n_last && (!last_kps_host || !last_flags || !last_x3dw || !mp_desc_host)) || p->img_w < 1 || p->img_h < 1) {
        g_err = "invalid arguments";
        return VSLAM_ERR_INVALID;
    }
    *nmatches = 0;
    if (n_cur == 0 || n_last == 0) {
        for (int i = 0; i < n_cur; i++) match_cur[i] = -1;
        return VSLAM_OK;
    }
    if (n_cur > 4096) { /* i2 is a 12-bit field of the ordering key */
        g_err = "SearchByProjection on the device supports at most 4096 current-frame keypoints";
        return VSLAM_ERR_UNSUPPORTED;
    }
    HIPCHK(hipSetDevice(fe->p.device));
    const int M = std::min(16, std::max(1, tune_or(fe->tune.sbp_topm, 8)));
    const size_t lds = std::max(vk_sbp_rank_lds(n_cur), vk_sbp_replay_lds(n_cur, n_last));
    if (lds > 150 * 1024) {
        g_err = "SearchByProjection: frame too large for the LDS-resident matcher";
        return VSLAM_ERR_UNSUPPORTED;
    }
    if (!fe->proj_lds_set) {
        if (vk_sbp_set_max_lds(150 * 1024) != 0) {
            g_err = "hipFuncSetAttribute(k_sbp_*) failed";
            return VSLAM_ERR_HIP;
        }
        fe->proj_lds_set = true;
    }
    /* one pinned block -> one upload: kps | x3Dw | mpDesc | flags | uRight | occupied, 16-byte aligned parts */
    auto al = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t o_kps = 0, o_x = al(o_kps + (size_t)n_last * sizeof(vslam_kp)), o_d = al(o_x + (size_t)n_last * 12),
                 o_f = al(o_d + (size_t)n_last * 32), o_u = al(o_f + (size_t)n_last), o_o = al(o_u + (size_t)n_cur * 4),
                 o_mn = al(o_o + (size_t)n_cur), o_mx = al(o_mn + (kf ? (size_t)n_last * 4 : 0)),
                 o_nr = al(o_mx + (kf ? (size_t)n_last * 4 : 0)),
                 in_bytes = al(o_nr + ((kf && kf->normals) ? (size_t)n_last * 12 : 0));
    const size_t o_scr = in_bytes, o_m = al(o_scr + vk_sbp_scratch_bytes(n_last, M)), o_n = al(o_m + (size_t)n_cur * 4),
                 total = o_n + 16;
    int rc = vslam_ensure((void**)&fe->d_proj, &fe->proj_bytes, total);
    if (rc) return rc;
    if ((rc = vslam_ensure_pinned(&fe->h_proj, &fe->h_proj_bytes, total))) return rc;
    if (!fe->d_init_fb) {
        HIPCHK(hipMalloc((void**)&fe->d_init_fb, 16));
        HIPCHK(hipMemset(fe->d_init_fb, 0, 16));
    }
    uint8_t* h = fe->h_proj;
    memcpy(h + o_kps, last_kps_host, (size_t)n_last * sizeof(vslam_kp));
    memcpy(h + o_x, last_x3dw, (size_t)n_last * 12);
    memcpy(h + o_d, mp_desc_host, (size_t)n_last * 32);
    memcpy(h + o_f, last_flags, (size_t)n_last);
    if (cur_u_right_host) memcpy(h + o_u, cur_u_right_host, (size_t)n_cur * 4);
    if (cur_occupied_host) memcpy(h + o_o, cur_occupied_host, (size_t)n_cur);
    if (kf) {
        for (int i = 0; i < n_last; i++) h[o_f + i] = (last_flags[i] & 1) ? 3 : 0; /* every accepted MapPoint blocks its keypoint */
        memcpy(h + o_mn, kf->min_dist, (size_t)n_last * 4);
        memcpy(h + o_mx, kf->max_dist, (size_t)n_last * 4);
        if (kf->normals) memcpy(h + o_nr, kf->normals, (size_t)n_last * 12);
    }
    hipStream_t st = fe->stream;
    HIPCHK(hipMemcpyAsync(fe->d_proj, h, in_bytes, hipMemcpyHostToDevice, st));
    static_assert(sizeof(SbpJobs) <= 4000, "SbpJobs travels as a kernel argument");
    SbpJobs JS;
    memset(&JS, 0, sizeof(JS));
    for (int l = 0; l < fe->p.nlevels; l++) JS.scale[l] = fe->tab.scale[l];
    JS.nlevels = fe->p.nlevels;
    JS.M = M;
    uint8_t* d = fe->d_proj;
    SbpJobDev& J = JS.job[0];
    memcpy(J.Tcw, p->Tcw, sizeof(J.Tcw));
    J.fx = p->fx; J.fy = p->fy; J.cx = p->cx; J.cy = p->cy; J.mbf = p->mbf; J.th = p->th;
    J.forward = p->forward; J.backward = p->backward; J.checkOri = p->check_orientation;
    J.imgW = p->img_w; J.imgH = p->img_h; J.gemmFloat = p->gemm_float;
    J.nLast = n_last; J.nCur = n_cur;
    J.lastKps = (const vslam_kp*)(d + o_kps);
    J.flags = d + o_f;
    J.x3Dw = (const float*)(d + o_x);
    J.mpDesc = d + o_d;
    J.curKps = dev_cur_kps;
    J.curDesc = dev_cur_desc;
    J.uRight = cur_u_right_host ? (const float*)(d + o_u) : nullptr;
    J.occupied0 = cur_occupied_host ? d + o_o : nullptr;
    J.proj = (SbpProj*)(d + o_scr);
    J.topm = (uint32_t*)(d + o_scr + vk_sbp_proj_bytes(n_last));
    J.matchCur = (int32_t*)(d + o_m);
    J.nmatches = (int32_t*)(d + o_n);
    J.needSeq = (int32_t*)(d + o_n) + 1;
    if (kf) {
        J.mode = kf->normals ? 3 : 2;
        JS.kf.projKind = kf->proj_kind;
        JS.kf.normals = kf->normals ? (const float*)(d + o_nr) : nullptr;
        JS.kf.thHigh = kf->orb_dist + 1;
        JS.kf.logScaleFactor = kf->log_scale_factor;
        JS.kf.minDist = (const float*)(d + o_mn);
        JS.kf.maxDist = (const float*)(d + o_mx);
        for (int i = 0; i < 3; i++) JS.kf.ow[i] = kf->ow[i];
    }
    const bool seq_mode = fe->tune.sbp_sequential == 1; /* skip the parallel resolution (cross-check path) */
    vk_search_by_projection(st, JS, 1, n_last, n_cur, fe->d_init_fb, seq_mode);
    HIPCHK(hipGetLastError());
    CopyRanges R;
    memset(&R, 0, sizeof(R));
    R.dst[0] = h + o_m;
    R.src[0] = d + o_m;
    R.bytes[0] = (o_n + 16) - o_m;
    R.n = 1;
    vk_copy_ranges(st, R);
    HIPCHK(hipGetLastError());
    HIPCHK(vslam_stream_wait(st));
    memcpy(match_cur, h + o_m, (size_t)n_cur * 4);
    *nmatches = *(const int32_t*)(h + o_n);
    return VSLAM_OK;
}

extern "C" int vslam_search_by_projection_frame(vslam_fe* fe, const vslam_proj_params* p,
                                                const vslam_kp* last_kps_host, int n_last, const uint8_t* last_flags,
                                                const float* last_x3dw, const uint8_t* mp_desc_host,
                                                const vslam_kp* dev_cur_kps, const uint8_t* dev_cur_desc, int n_cur,
                                                const float* cur_u_right_host, const uint8_t* cur_occupied_host,
                                                int32_t* match_cur, int* nmatches) {
    return sbp_frame_impl(fe, p, last_kps_host, n_last, last_flags, last_x3dw, mp_desc_host, dev_cur_kps, dev_cur_desc, n_cur,
                          cur_u_right_host, cur_occupied_host, nullptr, match_cur, nmatches);
}

/* FMatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const vector<MapPoint*>& vpPoints, vector<MapPoint*>& vpMatched,
 * int th, float ratioHamming) (fmatcher.cpp:750-863) and the overload that also records the KeyFrames
 * (:865-981; its projection is spelled differently, proj_variant 1): mode 3 of the projection kernels */
extern "C" int vslam_search_by_projection_sim3(vslam_fe* fe, const vslam_proj_params* p, const float* Ow,
                                               float log_scale_factor, float ratio_hamming, int proj_variant,
                                               const uint8_t* mp_flags, const float* mp_x3dw, const float* mp_normals,
                                               const float* mp_min_dist, const float* mp_max_dist,
                                               const uint8_t* mp_desc_host, int n_points, const vslam_kp* dev_kf_kps,
                                               const uint8_t* dev_kf_desc, int n_kf, const uint8_t* kf_matched_host,
                                               int32_t* match_kf, int* nmatches) {
    if (!Ow || !p || (n_points > 0 && (!mp_min_dist || !mp_max_dist || !mp_normals)) || proj_variant < 0 || proj_variant > 1 ||
        !(ratio_hamming >= 0.0f)) {
        g_err = "invalid arguments";
        return VSLAM_ERR_INVALID;
    }
    if (n_points > 4096) {
        g_err = "SearchByProjection on the device supports at most 4096 candidate MapPoints per call";
        return VSLAM_ERR_UNSUPPORTED;
    }
    /* bestDist <= TH_LOW*ratioHamming with an integer bestDist: the float product, floored */
    const float lim = 50 * ratio_hamming;
    int thr = lim >= 255.0f ? 255 : (int)floorf(lim);
    if (thr < 0) thr = 0;
    SbpKfExtras kf;
    kf.min_dist = mp_min_dist;
    kf.max_dist = mp_max_dist;
    for (int i = 0; i < 3; i++) kf.ow[i] = Ow[i];
    kf.log_scale_factor = log_scale_factor;
    kf.orb_dist = thr;
    kf.normals = mp_normals;
    kf.proj_kind = proj_variant;
    vslam_proj_params q = *p;
    q.check_orientation = 0; /* these overloads have no rotation histogram */
    std::vector<vslam_kp> dummy((size_t)std::max(n_points, 1));
    memset(dummy.data(), 0, dummy.size() * sizeof(vslam_kp));
    return sbp_frame_impl(fe, &q, dummy.data(), n_points, mp_flags, mp_x3dw, mp_desc_host, dev_kf_kps, dev_kf_desc, n_kf, nullptr,
                          kf_matched_host, &kf, match_kf, nmatches);
}

/* FMatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const set<MapPoint*>& sAlreadyFound, th, ORBdist)
 * (fmatcher.cpp:2689-2811): same ranking / resolution kernels, mode 2 */
extern "C" int vslam_search_by_projection_keyframe(vslam_fe* fe, const vslam_proj_params* p, const float* Ow,
                                                   float log_scale_factor, int orb_dist, const vslam_kp* kf_kps_host,
                                                   int n_kf, const uint8_t* mp_flags, const float* mp_x3dw,
                                                   const float* mp_min_dist, const float* mp_max_dist,
                                                   const uint8_t* mp_desc_host, const vslam_kp* dev_cur_kps,
                                                   const uint8_t* dev_cur_desc, int n_cur,
                                                   const uint8_t* cur_occupied_host, int32_t* match_cur, int* nmatches) {
    if (!Ow || (n_kf > 0 && (!mp_min_dist || !mp_max_dist)) || orb_dist < 1 || orb_dist > 255) {
        g_err = "invalid arguments";
        return VSLAM_ERR_INVALID;
    }
    if (n_kf > 4096) {
        g_err = "SearchByProjection on the device supports at most 4096 KeyFrame keypoints";
        return VSLAM_ERR_UNSUPPORTED;
    }
    SbpKfExtras kf;
    kf.min_dist = mp_min_dist;
    kf.max_dist = mp_max_dist;
    for (int i = 0; i < 3; i++) kf.ow[i] = Ow[i];
    kf.log_scale_factor = log_scale_factor;
    kf.orb_dist = orb_dist;
    return sbp_frame_impl(fe, p, kf_kps_host, n_kf, mp_flags, mp_x3dw, mp_desc_host, dev_cur_kps, dev_cur_desc, n_cur, nullptr,
                          cur_occupied_host, &kf, match_cur, nmatches);
}

static int sbp_prepare(vslam_fe* fe, int* M_out) {
    *M_out = std::min(16, std::max(1, tune_or(fe->tune.sbp_topm, 8)));
    if (!fe->proj_lds_set) {
        if (vk_sbp_set_max_lds(150 * 1024) != 0) {
            g_err = "hipFuncSetAttribute(k_sbp_*) failed";
            return VSLAM_ERR_HIP;
        }
        fe->proj_lds_set = true;
    }
    if (!fe->d_init_fb) {
        HIPCHK(hipMalloc((void**)&fe->d_init_fb, 16));
        HIPCHK(hipMemset(fe->d_init_fb, 0, 16));
    }
    return VSLAM_OK;
}

extern "C" int vslam_search_by_projection_dev_async(vslam_fe* fe, int njobs, const vslam_sbp_job* jobs) {
    if (!fe || njobs < 1 || njobs > VSLAM_MAX_SBP_JOBS || !jobs) {
        g_err = "invalid arguments";
        return VSLAM_ERR_INVALID;
    }
    const int cap = fe->cap;
    if (cap > 4096) {
        g_err = "SearchByProjection on the device supports at most 4096 keypoints per frame";
        return VSLAM_ERR_UNSUPPORTED;
    }
    if (std::max(vk_sbp_rank_lds(cap), vk_sbp_replay_lds(cap, cap)) > 150 * 1024) {
        g_err = "SearchByProjection: frame too large for the LDS-resident matcher";
        return VSLAM_ERR_UNSUPPORTED;
    }
    HIPCHK(hipSetDevice(fe->p.device));
    int M;
    int rc = sbp_prepare(fe, &M);
    if (rc) return rc;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    /* per job: proj | topm ; then all match tables and counts together (one result copy) */
    const size_t per_scr = al(vk_sbp_scratch_bytes(cap, M));
    const size_t o_res = per_scr * njobs, res_bytes = al((size_t)njobs * cap * 4 + (size_t)njobs * 8);
    rc = vslam_ensure((void**)&fe->d_sbp, &fe->sbp_bytes, o_res + res_bytes);
    if (rc) return rc;
    if (fe->h_sbp_bytes < res_bytes) {
        if (fe->h_sbp) HIPCHK(hipHostFree(fe->h_sbp));
        fe->h_sbp = nullptr;
        fe->h_sbp_bytes = 0;
        HIPCHK((hipError_t)vslam_pinned_alloc((void**)&fe->h_sbp, res_bytes));
        fe->h_sbp_bytes = res_bytes;
    }
    SbpJobs JS;
    memset(&JS, 0, sizeof(JS));
    for (int l = 0; l < fe->p.nlevels; l++) JS.scale[l] = fe->tab.scale[l];
    JS.nlevels = fe->p.nlevels;
    JS.M = M;
    int32_t* d_match = (int32_t*)(fe->d_sbp + o_res);
    int32_t* d_nm = d_match + (size_t)njobs * cap;
    for (int j = 0; j < njobs; j++) {
        const vslam_sbp_job& s = jobs[j];
        if (!s.dev_last_kps || !s.dev_n_last || !s.dev_last_flags || !s.dev_last_x3dw || !s.dev_mp_desc ||
            !s.dev_cur_kps || !s.dev_cur_desc || !s.dev_n_cur || s.p.img_w < 1 || s.p.img_h < 1) {
            g_err = "null device pointer in job";
            return VSLAM_ERR_INVALID;
        }
        SbpJobDev& J = JS.job[j];
        memcpy(J.Tcw, s.p.Tcw, sizeof(J.Tcw));
        J.fx = s.p.fx; J.fy = s.p.fy; J.cx = s.p.cx; J.cy = s.p.cy; J.mbf = s.p.mbf; J.th = s.p.th;
        J.forward = s.p.forward; J.backward = s.p.backward; J.checkOri = s.p.check_orientation;
        J.imgW = s.p.img_w; J.imgH = s.p.img_h; J.gemmFloat = s.p.gemm_float;
        J.nLast = cap; J.nCur = cap;
        J.lastKps = s.dev_last_kps; J.nLastPtr = s.dev_n_last; J.flags = s.dev_last_flags; J.x3Dw = s.dev_last_x3dw;
        J.mpDesc = s.dev_mp_desc; J.curKps = s.dev_cur_kps; J.curDesc = s.dev_cur_desc; J.nCurPtr = s.dev_n_cur;
        J.uRight = s.dev_cur_u_right; J.occupied0 = s.dev_cur_occupied;
        uint8_t* scr = fe->d_sbp + per_scr * j;
        J.proj = (SbpProj*)scr;
        J.topm = (uint32_t*)(scr + vk_sbp_proj_bytes(cap));
        J.matchCur = d_match + (size_t)j * cap;
        J.nmatches = d_nm + j;
        J.needSeq = d_nm + njobs + j;
    }
    const bool seq_mode = fe->tune.sbp_sequential == 1;
    vk_search_by_projection(fe->stream, JS, njobs, cap, cap, fe->d_init_fb, seq_mode);
    HIPCHK(hipGetLastError());
    CopyRanges R;
    memset(&R, 0, sizeof(R));
    R.dst[0] = fe->h_sbp;
    R.src[0] = d_match;
    R.bytes[0] = (size_t)njobs * cap * 4 + (size_t)njobs * 4;
    R.n = 1;
    vk_copy_ranges(fe->stream, R);
    HIPCHK(hipGetLastError());
    fe->sbp_jobs = njobs;
    return VSLAM_OK;
}

extern "C" int vslam_search_by_projection_dev_wait(vslam_fe* fe, const int* n_cur, int32_t* const* match_cur,
                                                   int* nmatches) {
    if (!fe || fe->sbp_jobs < 1) {
        g_err = "nothing enqueued";
        return VSLAM_ERR_INVALID;
    }
    HIPCHK(hipSetDevice(fe->p.device));
    HIPCHK(vslam_stream_wait(fe->stream));
    const int njobs = fe->sbp_jobs, cap = fe->cap;
    const int32_t* h_m = (const int32_t*)fe->h_sbp;
    const int32_t* h_n = h_m + (size_t)njobs * cap;
    for (int j = 0; j < njobs; j++) {
        if (match_cur && match_cur[j] && n_cur) memcpy(match_cur[j], h_m + (size_t)j * cap, (size_t)std::min(n_cur[j], cap) * 4);
        if (nmatches) nmatches[j] = h_n[j];
    }
    return VSLAM_OK;
}

extern "C" int vslam_stereo_points_dev_async(vslam_fe* fe, int npairs, const float* Twc, float cx, float cy,
                                             float invfx, float invfy, int observations, int gemm_float) {
    if (!fe || npairs < 1 || npairs > VSLAM_MAX_SBP_JOBS || npairs > fe->stereo_pairs || !Twc) {
        g_err = "invalid arguments (npairs must not exceed the pairs of the last stereo enqueue, <= 16)";
        return VSLAM_ERR_INVALID;
    }
    HIPCHK(hipSetDevice(fe->p.device));
    if (!fe->d_x3dw) {
        HIPCHK(hipMalloc((void**)&fe->d_x3dw, (size_t)fe->B * fe->cap * 12));
        HIPCHK(hipMalloc((void**)&fe->d_mpflags, (size_t)fe->B * fe->cap));
    }
    StereoScratch sc;
    int rc = stereo_scratch(fe, fe->stereo_pairs, fe->stereo_capR, &sc); /* same carving as the enqueue that filled it */
    if (rc) return rc;
    UnprojJobs U;
    memset(&U, 0, sizeof(U));
    U.cx = cx; U.cy = cy; U.invfx = invfx; U.invfy = invfy;
    U.cap = fe->cap; U.observations = observations; U.gemmFloat = gemm_float;
    for (int j = 0; j < npairs; j++) {
        UnprojJob& J = U.job[j];
        memcpy(J.Twc, Twc + 12 * j, sizeof(J.Twc));
        const int sl = fe->stereo_slotL[j];
        J.kps = fe->d_kps + (size_t)sl * fe->cap;
        J.nPtr = fe->d_counts + sl * 4;
        J.depth = fe->d_stereo_depth + (size_t)j * fe->cap;
        J.x3Dw = fe->d_x3dw + (size_t)j * fe->cap * 3;
        J.flags = fe->d_mpflags + (size_t)j * fe->cap;
    }
    vk_unproject_stereo(fe->stream, U, npairs);
    HIPCHK(hipGetLastError());
    return VSLAM_OK;
}

extern "C" int vslam_stereo_points_buffers(vslam_fe* fe, int pair, const float** dev_x3dw, const uint8_t** dev_flags,
                                           const float** dev_u_right, const float** dev_depth) {
    if (!fe || pair < 0 || pair >= fe->B || pair >= VSLAM_MAX_SBP_JOBS) return VSLAM_ERR_INVALID;
    HIPCHK(hipSetDevice(fe->p.device));
    if (!fe->d_x3dw) {
        HIPCHK(hipMalloc((void**)&fe->d_x3dw, (size_t)fe->B * fe->cap * 12));
        HIPCHK(hipMalloc((void**)&fe->d_mpflags, (size_t)fe->B * fe->cap));
    }
    if (dev_x3dw) *dev_x3dw = fe->d_x3dw + (size_t)pair * fe->cap * 3;
    if (dev_flags) *dev_flags = fe->d_mpflags + (size_t)pair * fe->cap;
    if (dev_u_right || dev_depth) {
        if (fe->stereo_pairs < 1 || pair >= fe->stereo_pairs) {
            g_err = "no stereo result for this pair yet";
            return VSLAM_ERR_INVALID;
        }
        StereoScratch sc;
        int rc = stereo_scratch(fe, fe->stereo_pairs, fe->stereo_capR, &sc);
        if (rc) return rc;
        if (dev_u_right) *dev_u_right = fe->d_stereo_u + (size_t)pair * fe->cap;
        if (dev_depth) *dev_depth = fe->d_stereo_depth + (size_t)pair * fe->cap;
    }
    return VSLAM_OK;
}

/* ------------------------------------------------------------------ SearchByProjection(F, vpMapPoints) */
extern "C" int vslam_search_by_projection_mappoints(vslam_fe* fe, const vslam_mp_track* mps_host,
                                                    const uint8_t* mp_desc_host, int n_mp, const vslam_kp* dev_cur_kps,
                                                    const uint8_t* dev_cur_desc, int n_cur,
                                                    const float* cur_u_right_host, const uint8_t* cur_occupied_host,
                                                    int img_w, int img_h, float th, float nnratio, int32_t* match_cur,
                                                    int* nmatches) {
    static_assert(sizeof(vslam_mp_track) == sizeof(MpTrack), "vslam_mp_track layout");
    if (!fe || n_mp < 0 || n_cur < 0 || !nmatches || (n_cur && (!dev_cur_kps || !dev_cur_desc || !match_cur)) ||
        (n_mp && (!mps_host || !mp_desc_host)) || img_w < 1 || img_h < 1) {
        g_err = "invalid arguments";
        return VSLAM_ERR_INVALID;
    }
    *nmatches = 0;
    if (n_cur == 0 || n_mp == 0) {
        for (int i = 0; i < n_cur; i++) match_cur[i] = -1;
        return VSLAM_OK;
    }
    if (n_cur > 4096 || n_mp > 4096) {
        g_err = "SearchByProjection on the device supports at most 4096 keypoints / MapPoints per call";
        return VSLAM_ERR_UNSUPPORTED;
    }
    if (!(nnratio >= 0.4f)) { /* keys clamp distances to 255; exact for bestDist <= 100 when 0.4 * 255 > 100 */
        g_err = "SearchByProjection on the device needs nnratio >= 0.4";
        return VSLAM_ERR_UNSUPPORTED;
    }
    HIPCHK(hipSetDevice(fe->p.device));
    int M;
    int rc = sbp_prepare(fe, &M);
    if (rc) return rc;
    auto al = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t o_mp = 0, o_d = al(o_mp + (size_t)n_mp * sizeof(MpTrack)), o_f = al(o_d + (size_t)n_mp * 32),
                 o_u = al(o_f + (size_t)n_mp), o_o = al(o_u + (size_t)n_cur * 4), in_bytes = al(o_o + (size_t)n_cur);
    const size_t o_scr = in_bytes, o_m = al(o_scr + vk_sbp_scratch_bytes(n_mp, M)), o_n = al(o_m + (size_t)n_cur * 4),
                 total = o_n + 16;
    rc = vslam_ensure((void**)&fe->d_proj, &fe->proj_bytes, total);
    if (rc) return rc;
    if ((rc = vslam_ensure_pinned(&fe->h_proj, &fe->h_proj_bytes, total))) return rc;
    uint8_t* h = fe->h_proj;
    memcpy(h + o_mp, mps_host, (size_t)n_mp * sizeof(MpTrack));
    memcpy(h + o_d, mp_desc_host, (size_t)n_mp * 32);
    for (int i = 0; i < n_mp; i++) h[o_f + i] = (uint8_t)(mps_host[i].flags & 3u);
    if (cur_u_right_host) memcpy(h + o_u, cur_u_right_host, (size_t)n_cur * 4);
    if (cur_occupied_host) memcpy(h + o_o, cur_occupied_host, (size_t)n_cur);
    hipStream_t st = fe->stream;
    uint8_t* d = fe->d_proj;
    {
        CopyRanges R;
        memset(&R, 0, sizeof(R));
        R.dst[0] = d;
        R.src[0] = h;
        R.bytes[0] = in_bytes;
        R.n = 1;
        vk_copy_ranges(st, R); /* pinned -> HBM by a kernel (cheaper to enqueue than hipMemcpyAsync) */
    }
    SbpJobs JS;
    memset(&JS, 0, sizeof(JS));
    for (int l = 0; l < fe->p.nlevels; l++) JS.scale[l] = fe->tab.scale[l];
    JS.nlevels = fe->p.nlevels;
    JS.M = M;
    SbpJobDev& J = JS.job[0];
    J.mode = 1;
    J.th = th;
    J.nnratio = nnratio;
    J.imgW = img_w;
    J.imgH = img_h;
    J.nLast = n_mp;
    J.nCur = n_cur;
    J.mps = (const MpTrack*)(d + o_mp);
    J.mpDesc = d + o_d;
    J.flags = d + o_f;
    J.curKps = dev_cur_kps;
    J.curDesc = dev_cur_desc;
    J.uRight = cur_u_right_host ? (const float*)(d + o_u) : nullptr;
    J.occupied0 = cur_occupied_host ? d + o_o : nullptr;
    J.proj = (SbpProj*)(d + o_scr);
    J.topm = (uint32_t*)(d + o_scr + vk_sbp_proj_bytes(n_mp));
    J.matchCur = (int32_t*)(d + o_m);
    J.nmatches = (int32_t*)(d + o_n);
    J.needSeq = (int32_t*)(d + o_n) + 1;
    const bool seq_mode = fe->tune.sbp_sequential == 1;
    vk_search_by_projection(st, JS, 1, n_mp, n_cur, fe->d_init_fb, seq_mode);
    HIPCHK(hipGetLastError());
    CopyRanges R;
    memset(&R, 0, sizeof(R));
    R.dst[0] = h + o_m;
    R.src[0] = d + o_m;
    R.bytes[0] = (o_n + 16) - o_m;
    R.n = 1;
    vk_copy_ranges(st, R);
    HIPCHK(hipGetLastError());
    HIPCHK(vslam_stream_wait(st));
    memcpy(match_cur, h + o_m, (size_t)n_cur * 4);
    *nmatches = *(const int32_t*)(h + o_n);
    return VSLAM_OK;
}

/* ------------------------------------------------------------------ MapPoint::ComputeDistinctiveDescriptors */
extern "C" int vslam_distinctive_descriptors(vslam_fe* fe, const uint8_t* desc_host, const int32_t* offsets, int nsets,
                                             int32_t* best) {
    if (!fe || nsets < 0 || (nsets && (!offsets || !best))) {
        g_err = "invalid arguments";
        return VSLAM_ERR_INVALID;
    }
    if (nsets == 0) return VSLAM_OK;
    int maxN = 0;
    for (int s = 0; s < nsets; s++) {
        const int n = offsets[s + 1] - offsets[s];
        if (n < 0 || offsets[s] < 0) {
            g_err = "offsets must be non-negative and ascending";
            return VSLAM_ERR_INVALID;
        }
        maxN = std::max(maxN, n);
    }
    if (maxN > 2048) {
        g_err = "ComputeDistinctiveDescriptors on the device supports at most 2048 observations per MapPoint";
        return VSLAM_ERR_UNSUPPORTED;
    }
    const int total = offsets[nsets];
    if (total > 0 && !desc_host) {
        g_err = "invalid arguments";
        return VSLAM_ERR_INVALID;
    }
    HIPCHK(hipSetDevice(fe->p.device));
    auto al = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t o_d = 0, o_o = al((size_t)total * 32), o_b = al(o_o + (size_t)(nsets + 1) * 4),
                 bytes = al(o_b + (size_t)nsets * 4);
    int rc = vslam_ensure((void**)&fe->d_proj, &fe->proj_bytes, bytes);
    if (rc) return rc;
    if (fe->h_proj_bytes < bytes) {
        if (fe->h_proj) HIPCHK(hipHostFree(fe->h_proj));
        fe->h_proj = nullptr;
        fe->h_proj_bytes = 0;
        HIPCHK((hipError_t)vslam_pinned_alloc((void**)&fe->h_proj, bytes));
        fe->h_proj_bytes = bytes;
    }
    uint8_t *h = fe->h_proj, *d = fe->d_proj;
    if (total) memcpy(h + o_d, desc_host, (size_t)total * 32);
    memcpy(h + o_o, offsets, (size_t)(nsets + 1) * 4);
    if (!fe->dist_lds_set && vk_distinctive_set_max_lds(2048 * 32) != 0) {
        g_err = "hipFuncSetAttribute(k_distinctive) failed";
        return VSLAM_ERR_HIP;
    }
    fe->dist_lds_set = true;
    hipStream_t st = fe->stream;
    CopyRanges R;
    memset(&R, 0, sizeof(R));
    R.dst[0] = d;
    R.src[0] = h;
    R.bytes[0] = o_b;
    R.n = 1;
    vk_copy_ranges(st, R);
    vk_distinctive(st, d + o_d, (const int32_t*)(d + o_o), nsets, maxN, (int32_t*)(d + o_b));
    R.dst[0] = h + o_b;
    R.src[0] = d + o_b;
    R.bytes[0] = (size_t)nsets * 4;
    vk_copy_ranges(st, R);
    HIPCHK(hipGetLastError());
    HIPCHK(vslam_stream_wait(st));
    memcpy(best, h + o_b, (size_t)nsets * 4);
    return VSLAM_OK;
}

/* ---- the search half of FMatcher::Fuse (fmatcher.cpp:1918-2119, :2121-2243): k_fuse_rank ---- */
extern "C" int vslam_fuse_search(vslam_fe* fe, const vslam_fuse_params* p, const vslam_fuse_point* points_host,
                                 const uint8_t* mp_desc_host, int n_points, const vslam_kp* dev_kf_kps,
                                 const uint8_t* dev_kf_desc, int n_kf, const float* kf_u_right_host, int32_t* best_idx,
                                 int32_t* best_dist) {
    static_assert(sizeof(vslam_fuse_point) == sizeof(FusePoint), "vslam_fuse_point layout");
    if (!fe || !p || n_points < 0 || n_kf < 0 || (n_points && (!points_host || !mp_desc_host || !best_idx || !best_dist)) ||
        (n_kf && (!dev_kf_kps || !dev_kf_desc)) || p->img_w < 1 || p->img_h < 1) {
        g_err = "invalid arguments";
        return VSLAM_ERR_INVALID;
    }
    for (int i = 0; i < n_points; i++) {
        best_idx[i] = -1;
        best_dist[i] = 256;
    }
    if (n_points == 0 || n_kf == 0) return VSLAM_OK;
    if (n_kf > 4096) {
        g_err = "Fuse on the device supports at most 4096 keypoints per KeyFrame";
        return VSLAM_ERR_UNSUPPORTED;
    }
    HIPCHK(hipSetDevice(fe->p.device));
    int M;
    int rc = sbp_prepare(fe, &M);
    if (rc) return rc;
    auto al = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t o_p = 0, o_d = al(o_p + (size_t)n_points * sizeof(FusePoint)), o_u = al(o_d + (size_t)n_points * 32),
                 in_bytes = al(o_u + (size_t)n_kf * 4);
    const size_t o_bi = in_bytes, o_bd = al(o_bi + (size_t)n_points * 4), total = al(o_bd + (size_t)n_points * 4);
    rc = vslam_ensure((void**)&fe->d_proj, &fe->proj_bytes, total);
    if (rc) return rc;
    if ((rc = vslam_ensure_pinned(&fe->h_proj, &fe->h_proj_bytes, total))) return rc;
    uint8_t *h = fe->h_proj, *d = fe->d_proj;
    memcpy(h + o_p, points_host, (size_t)n_points * sizeof(FusePoint));
    memcpy(h + o_d, mp_desc_host, (size_t)n_points * 32);
    for (int i = 0; i < n_kf; i++) ((float*)(h + o_u))[i] = kf_u_right_host ? kf_u_right_host[i] : -1.0f;
    hipStream_t st = fe->stream;
    CopyRanges R;
    memset(&R, 0, sizeof(R));
    R.dst[0] = d;
    R.src[0] = h;
    R.bytes[0] = in_bytes;
    R.n = 1;
    vk_copy_ranges(st, R);
    FuseArgsDev A;
    memset(&A, 0, sizeof(A));
    for (int i = 0; i < 9; i++) A.Rcw[i] = p->Rcw[i];
    for (int i = 0; i < 3; i++) {
        A.tcw[i] = p->tcw[i];
        A.Ow[i] = p->Ow[i];
    }
    for (int i = 0; i < 9; i++) A.Rb[i] = p->Rb[i];
    for (int i = 0; i < 3; i++) A.tb[i] = p->tb[i];
    A.fx = p->fx; A.fy = p->fy; A.cx = p->cx; A.cy = p->cy; A.bf = p->bf; A.th = p->th;
    A.logScaleFactor = p->log_scale_factor;
    A.imgW = p->img_w; A.imgH = p->img_h; A.sim3 = p->sim3; A.gemmFloat = p->gemm_float;
    A.nlevels = fe->p.nlevels; A.nPoints = n_points; A.nKF = n_kf;
    for (int l = 0; l < fe->p.nlevels; l++) {
        A.scale[l] = fe->tab.scale[l];
        A.invSigma2[l] = fe->tab.inv_sigma2[l];
    }
    A.pts = (const FusePoint*)(d + o_p);
    A.mpDesc = d + o_d;
    A.kfKps = dev_kf_kps;
    A.kfDesc = dev_kf_desc;
    A.kfURight = (const float*)(d + o_u);
    A.bestIdx = (int32_t*)(d + o_bi);
    A.bestDist = (int32_t*)(d + o_bd);
    vk_fuse_search(st, A);
    R.dst[0] = h + o_bi;
    R.src[0] = d + o_bi;
    R.bytes[0] = total - o_bi;
    vk_copy_ranges(st, R);
    HIPCHK(hipGetLastError());
    HIPCHK(vslam_stream_wait(st));
    memcpy(best_idx, h + o_bi, (size_t)n_points * 4);
    memcpy(best_dist, h + o_bd, (size_t)n_points * 4);
    return VSLAM_OK;
}
