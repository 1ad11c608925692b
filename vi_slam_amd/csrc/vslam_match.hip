/* vslam_match.hip -- matcher half of the C ABI: Frame::ComputeStereoMatches on the device and the
 * host-replayed FMatcher::SearchForInitialization (distances from the device). */
#include "vslam_ctx.h"

/* ------------------------------------------------------------------ stereo */
struct StereoScratch { /* carved out of fe->d_stereo */
    uint32_t* best;
    float* uRight;
    float* depth;
    int32_t* sad;
};

static int stereo_scratch(vslam_fe* fe, int njobs, StereoScratch* s) {
    const size_t n = (size_t)njobs * fe->cap;
    int rc = vslam_ensure(&fe->d_stereo, &fe->stereo_bytes, n * 16);
    if (rc) return rc;
    s->best = (uint32_t*)fe->d_stereo;
    s->uRight = (float*)(s->best + n);
    s->depth = s->uRight + n;
    s->sad = (int32_t*)(s->depth + n);
    return VSLAM_OK;
}

extern "C" int vslam_stereo_match_batch(vslam_fe* feL, vslam_fe* feR, int npairs, const int* slotsL,
                                        const int* slotsR, float bf, float fx, float* const* u_right,
                                        float* const* depth) {
    if (!feL || !feR || npairs < 1 || npairs > VSLAM_MAX_STEREO_JOBS || !slotsL || !slotsR || !u_right || !depth) {
        g_err = "invalid arguments";
        return VSLAM_ERR_INVALID;
    }
    if (feL->p.device != feR->p.device || feL->p.width != feR->p.width || feL->p.height != feR->p.height ||
        feL->p.nlevels != feR->p.nlevels || feL->p.scale_factor != feR->p.scale_factor) {
        g_err = "left/right extractors must share device and geometry";
        return VSLAM_ERR_INVALID;
    }
    HIPCHK(hipSetDevice(feL->p.device));
    StereoJobs jobs;
    memset(&jobs, 0, sizeof(jobs));
    int maxNL = 0, maxNR = 0;
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
        jb.nL = feL->n_out[sL];
        jb.nR = feR->n_out[sR];
        jb.slotL = sL;
        jb.slotR = sR;
        maxNL = std::max(maxNL, jb.nL);
        maxNR = std::max(maxNR, jb.nR);
    }
    if (feL != feR) HIPCHK(hipStreamSynchronize(feR->stream)); /* right results must be complete */
    StereoScratch sc;
    int rc = stereo_scratch(feL, npairs, &sc);
    if (rc) return rc;
    /* frame.cpp:853-855: mb = mbf/fx (frame.cpp:157), minZ = mb, maxD = mbf/minZ */
    const float mb = bf / fx;
    const float maxD = bf / mb;
    hipStream_t st = feL->stream;
    if (maxNL > 0) {
        vk_stereo(st, jobs, npairs, maxNL, maxNR, feL->geom, feL->d_pyr, feL->slot_stride, feL->src, feR->d_pyr,
                  feR->slot_stride, feR->src, bf, maxD, sc.best, sc.uRight, sc.depth, sc.sad, feL->cap);
        HIPCHK(hipGetLastError());
        for (int j = 0; j < npairs; j++) {
            if (!jobs.job[j].nL) continue;
            HIPCHK(hipMemcpyAsync(u_right[j], sc.uRight + (size_t)j * feL->cap, (size_t)jobs.job[j].nL * 4,
                                  hipMemcpyDeviceToHost, st));
            HIPCHK(hipMemcpyAsync(depth[j], sc.depth + (size_t)j * feL->cap, (size_t)jobs.job[j].nL * 4,
                                  hipMemcpyDeviceToHost, st));
        }
    }
    HIPCHK(hipStreamSynchronize(st));
    return VSLAM_OK;
}

extern "C" int vslam_stereo_match(vslam_fe* feL, int sL, vslam_fe* feR, int sR, float bf, float fx,
                                  float* u_right, float* depth) {
    float* u[1] = {u_right};
    float* d[1] = {depth};
    return vslam_stereo_match_batch(feL, feR, 1, &sL, &sR, bf, fx, u, d);
}

/* ------------------------------------------------------------------ SearchForInitialization */
extern "C" int vslam_search_for_initialization(vslam_fe* fe, const vslam_kp* kps1, const uint8_t* dev_desc1,
                                               int n1, const vslam_kp* kps2, const uint8_t* dev_desc2, int n2,
                                               int img_w, int img_h, float* prev_matched, int32_t* matches12,
                                               int window, float nnratio, int check_orientation,
                                               int* nmatches) {
    if (!fe || n1 < 0 || n2 < 0 || (n1 && (!kps1 || !dev_desc1 || !prev_matched || !matches12)) ||
        (n2 && (!kps2 || !dev_desc2)) || !nmatches) {
        g_err = "invalid arguments";
        return VSLAM_ERR_INVALID;
    }
    HIPCHK(hipSetDevice(fe->p.device));
    /* only octave-0 keypoints take part (fmatcher.cpp:999-1003: level1 > 0 -> continue; window query
     * restricted to [level1, level1]) */
    std::vector<int> row_of(n1, -1), col_of(n2, -1);
    std::vector<int32_t> rows, cols;
    for (int i = 0; i < n1; i++)
        if (kps1[i].octave == 0) { row_of[i] = (int)rows.size(); rows.push_back(i); }
    for (int i = 0; i < n2; i++)
        if (kps2[i].octave == 0) { col_of[i] = (int)cols.size(); cols.push_back(i); }
    const int nr = (int)rows.size(), nc = (int)cols.size();
    std::vector<uint8_t> dmat((size_t)std::max(nr, 1) * std::max(nc, 1));
    if (nr && nc) {
        int rc;
        const size_t ib = (size_t)(nr + nc) * 4;
        if ((rc = vslam_ensure((void**)&fe->d_tmp_desc[0], &fe->tmp_desc_bytes[0], (size_t)(nr + nc) * 32 + ib))) return rc;
        uint8_t* d_q = fe->d_tmp_desc[0];
        uint8_t* d_t = d_q + (size_t)nr * 32;
        int32_t* d_idx = (int32_t*)(d_t + (size_t)nc * 32);
        hipStream_t st = fe->stream;
        HIPCHK(hipMemcpyAsync(d_idx, rows.data(), (size_t)nr * 4, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(d_idx + nr, cols.data(), (size_t)nc * 4, hipMemcpyHostToDevice, st));
        vk_gather_rows32(st, dev_desc1, d_idx, nr, d_q);
        vk_gather_rows32(st, dev_desc2, d_idx + nr, nc, d_t);
        if ((rc = vslam_ensure((void**)&fe->d_dmat, &fe->dmat_bytes, (size_t)nr * nc))) return rc;
        vk_hamming_matrix(st, d_q, nr, d_t, nc, fe->d_dmat);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(dmat.data(), fe->d_dmat, (size_t)nr * nc, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    *nmatches = vslam::search_for_initialization_replay(kps1, n1, kps2, n2, dmat.data(), row_of.data(),
                                                        col_of.data(), std::max(nc, 1), img_w, img_h,
                                                        prev_matched, matches12, window, nnratio,
                                                        check_orientation != 0);
    return VSLAM_OK;
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
    HIPCHK(hipStreamSynchronize(fe->stream));
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
    HIPCHK(hipStreamSynchronize(fe->stream));
    return VSLAM_OK;
}
