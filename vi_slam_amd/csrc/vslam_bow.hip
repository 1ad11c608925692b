/* vslam_bow.hip -- Frame::ComputeBoW (frame.cpp:455-461): DBoW3::Vocabulary::transform(features, BowVector&,
 * FeatureVector&, levelsup) (thirdparty/DBoW3/DBoW3/src/Vocabulary.cpp:754-878).
 *
 * The per-feature part -- walking the vocabulary tree by "closest child in Hamming distance, first child wins
 * ties" -- runs on the GPU: 16 lanes per feature (one child per lane, k = 10 for the ORB vocabulary), 4 features per
 * wave, the arg-min as a 4-step shuffle reduction on dist << 20 | child position.  What is left (accumulating the
 * word weights in a std::map in feature order, the L1/L2 normalisation in double, grouping feature indices by node)
 * is order-dependent double arithmetic on ~2000 items and stays on the host: vslam_bow_assemble (vslam_host.cpp).
 */
#include "vslam_ctx.h"
#include "vslam_kernels.h"

struct vslam_voc {
    int device = 0;
    int nNodes = 0, L = 0, weighting = 0, norm = 1;
    int32_t *d_childStart = nullptr, *d_childCount = nullptr, *d_childIds = nullptr, *d_word = nullptr;
    uint8_t* d_desc = nullptr;
    double* d_weight = nullptr;
};

struct BowJobsDev {
    const uint8_t* desc[VSLAM_MAX_BATCH];
    const int32_t* cnt[VSLAM_MAX_BATCH]; /* null: n[] holds the count */
    int32_t n[VSLAM_MAX_BATCH];
    int32_t* word[VSLAM_MAX_BATCH];
    double* weight[VSLAM_MAX_BATCH];
    int32_t* nid[VSLAM_MAX_BATCH];
};

__global__ void __launch_bounds__(256)
k_bow_descend(BowJobsDev J, const int32_t* __restrict__ childStart, const int32_t* __restrict__ childCount,
              const int32_t* __restrict__ childIds, const uint8_t* __restrict__ nodeDesc,
              const double* __restrict__ nodeWeight, const int32_t* __restrict__ nodeWord, int L, int levelsup) {
    const int job = blockIdx.y;
    const int n = J.cnt[job] ? min(*J.cnt[job], J.n[job]) : J.n[job];
    const int f = blockIdx.x * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
    if (f >= n) return; /* whole 16-lane group */
    const uint4 fa = ((const uint4*)J.desc[job])[(size_t)f * 2], fb = ((const uint4*)J.desc[job])[(size_t)f * 2 + 1];
    const int nid_level = L - levelsup;
    int final_id = 0, level = 0, nid = 0; /* nid_level <= 0: the root */
    do { /* Vocabulary.cpp:849-871 */
        ++level;
        const int c0 = childStart[final_id], cn = childCount[final_id];
        uint32_t best = 0xFFFFFFFFu;
        for (int j = sub; j < cn; j += 16) {
            const int id = childIds[c0 + j];
            const uint4 da = ((const uint4*)nodeDesc)[(size_t)id * 2], db = ((const uint4*)nodeDesc)[(size_t)id * 2 + 1];
            const uint32_t d = __popc(fa.x ^ da.x) + __popc(fa.y ^ da.y) + __popc(fa.z ^ da.z) + __popc(fa.w ^ da.w) +
                               __popc(fb.x ^ db.x) + __popc(fb.y ^ db.y) + __popc(fb.z ^ db.z) + __popc(fb.w ^ db.w);
            best = min(best, (d << 20) | (uint32_t)j); /* d < best_d in child order: first wins */
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, o, 16));
        final_id = childIds[c0 + (int)(best & 0xFFFFFu)];
        if (level == nid_level) nid = final_id;
    } while (childCount[final_id] != 0);
    if (sub == 0) {
        J.word[job][f] = nodeWord[final_id];
        J.weight[job][f] = nodeWeight[final_id];
        J.nid[job][f] = nid;
    }
}

template <typename T>
static int up(T** dst, const void* src, size_t bytes) {
    HIPCHK(hipMalloc((void**)dst, bytes ? bytes : 8));
    if (bytes) HIPCHK(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    return VSLAM_OK;
}

/* DBoW3::Vocabulary::load(filename) (Vocabulary.cpp:1084-1112; called by core::System, system.cpp:76): parse the file on
 * the host (vslam_voc_file.cpp) and upload the node table. */
extern "C" int vslam_voc_load(int device, const char* path, vslam_voc** out) {
    if (!out) {
        g_err = "invalid arguments";
        return VSLAM_ERR_INVALID;
    }
    *out = nullptr;
    vslam_voc_file* f = nullptr;
    int rc = vslam_voc_file_open(path, &f);
    if (rc != VSLAM_OK) {
        g_err = vslam_voc_file_last_error();
        return rc;
    }
    int L = 0, weighting = 0, norm = 0, n_nodes = 0, n_child = 0;
    const int32_t *cs, *cc, *ci, *wi;
    const uint8_t* nd;
    const double* nw;
    vslam_voc_file_info(f, nullptr, &L, nullptr, &weighting, &norm, &n_nodes, nullptr, &n_child, nullptr);
    vslam_voc_file_arrays(f, &cs, &cc, &ci, &nd, &nw, &wi);
    rc = vslam_voc_create(device, L, weighting, norm, n_nodes, cs, cc, ci, n_child, nd, nw, wi, out);
    vslam_voc_file_close(f);
    return rc;
}

extern "C" void vslam_voc_destroy(vslam_voc* v) {
    if (!v) return;
    hipSetDevice(v->device);
    hipFree(v->d_childStart);
    hipFree(v->d_childCount);
    hipFree(v->d_childIds);
    hipFree(v->d_word);
    hipFree(v->d_desc);
    hipFree(v->d_weight);
    delete v;
}

extern "C" int vslam_voc_create(int device, int depth_levels, int weighting, int norm, int n_nodes,
                                const int32_t* child_start, const int32_t* child_count, const int32_t* child_ids,
                                int n_child_ids, const uint8_t* node_desc, const double* node_weight,
                                const int32_t* node_word_id, vslam_voc** out) {
    if (!out || n_nodes < 2 || !child_start || !child_count || !child_ids || !node_desc || !node_weight ||
        !node_word_id || depth_levels < 1 || weighting < 0 || weighting > 3 || norm < 0 || norm > 2) {
        g_err = "invalid arguments";
        return VSLAM_ERR_INVALID;
    }
    /* the walk must terminate: children of a node must exist and point forward (DBoW3 numbers nodes in creation
     * order, parents first), and the root must not be a leaf */
    if (child_count[0] < 1) {
        g_err = "empty vocabulary";
        return VSLAM_ERR_INVALID;
    }
    for (int i = 0; i < n_nodes; i++) {
        if (child_count[i] < 0 || child_count[i] > (1 << 20) || child_start[i] < 0 ||
            (long long)child_start[i] + child_count[i] > n_child_ids) {
            g_err = "vocabulary: child range out of bounds";
            return VSLAM_ERR_INVALID;
        }
        for (int j = 0; j < child_count[i]; j++) {
            const int c = child_ids[child_start[i] + j];
            if (c <= i || c >= n_nodes) {
                g_err = "vocabulary: child ids must be larger than their parent's and inside the node table";
                return VSLAM_ERR_INVALID;
            }
        }
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device >= ndev) {
        g_err = "no usable HIP device (this library has no CPU fallback)";
        return VSLAM_ERR_NO_DEVICE;
    }
    HIPCHK(hipSetDevice(device));
    vslam_voc* v = new vslam_voc();
    v->device = device;
    v->nNodes = n_nodes;
    v->L = depth_levels;
    v->weighting = weighting;
    v->norm = norm;
    int rc;
    if ((rc = up(&v->d_childStart, child_start, (size_t)n_nodes * 4)) || (rc = up(&v->d_childCount, child_count, (size_t)n_nodes * 4)) ||
        (rc = up(&v->d_childIds, child_ids, (size_t)n_child_ids * 4)) || (rc = up(&v->d_word, node_word_id, (size_t)n_nodes * 4)) ||
        (rc = up(&v->d_desc, node_desc, (size_t)n_nodes * 32)) || (rc = up(&v->d_weight, node_weight, (size_t)n_nodes * 8))) {
        vslam_voc_destroy(v);
        return rc;
    }
    *out = v;
    return VSLAM_OK;
}

extern "C" int vslam_voc_info(const vslam_voc* v, int* depth_levels, int* weighting, int* norm, int* n_nodes) {
    if (!v) return VSLAM_ERR_INVALID;
    if (depth_levels) *depth_levels = v->L;
    if (weighting) *weighting = v->weighting;
    if (norm) *norm = v->norm;
    if (n_nodes) *n_nodes = v->nNodes;
    return VSLAM_OK;
}

static int bow_buffers(vslam_fe* fe, int nslots, int per) {
    const size_t bytes = (size_t)nslots * per * 16;
    int rc = vslam_ensure((void**)&fe->d_bow, &fe->bow_bytes, bytes);
    if (rc) return rc;
    if (fe->h_bow_bytes < bytes) {
        if (fe->h_bow) HIPCHK(hipHostFree(fe->h_bow));
        fe->h_bow = nullptr;
        fe->h_bow_bytes = 0;
        HIPCHK((hipError_t)vslam_pinned_alloc((void**)&fe->h_bow, bytes));
        fe->h_bow_bytes = bytes;
    }
    return VSLAM_OK;
}

/* layout of d_bow / h_bow for `nslots` jobs of `per` features: weight f64[nslots][per] | word i32[..] | nid i32[..] */
static void bow_launch(vslam_fe* fe, const vslam_voc* v, int njobs, const uint8_t* const* desc, const int32_t* const* cnt,
                       const int* n, int per, int levelsup) {
    BowJobsDev J;
    memset(&J, 0, sizeof(J));
    double* dw = (double*)fe->d_bow;
    int32_t* dword = (int32_t*)(dw + (size_t)njobs * per);
    int32_t* dnid = dword + (size_t)njobs * per;
    int maxn = 0;
    for (int j = 0; j < njobs; j++) {
        J.desc[j] = desc[j];
        J.cnt[j] = cnt ? cnt[j] : nullptr;
        J.n[j] = n[j];
        J.weight[j] = dw + (size_t)j * per;
        J.word[j] = dword + (size_t)j * per;
        J.nid[j] = dnid + (size_t)j * per;
        maxn = std::max(maxn, n[j]);
    }
    if (maxn > 0)
        hipLaunchKernelGGL(k_bow_descend, dim3((maxn + 15) / 16, njobs), dim3(256), 0, fe->stream, J, v->d_childStart,
                           v->d_childCount, v->d_childIds, v->d_desc, v->d_weight, v->d_word, v->L, levelsup);
    CopyRanges R;
    memset(&R, 0, sizeof(R));
    R.dst[0] = fe->h_bow;
    R.src[0] = fe->d_bow;
    R.bytes[0] = (size_t)njobs * per * 16;
    R.n = 1;
    vk_copy_ranges(fe->stream, R);
}

extern "C" int vslam_bow_transform(vslam_fe* fe, const vslam_voc* voc, const uint8_t* dev_desc, int n, int levelsup,
                                   int32_t* word_id, double* weight, int32_t* node_id) {
    if (!fe || !voc || n < 0 || (n && (!dev_desc || !word_id || !weight || !node_id)) || voc->device != fe->p.device) {
        g_err = "invalid arguments (the vocabulary must live on the context's device)";
        return VSLAM_ERR_INVALID;
    }
    if (n == 0) return VSLAM_OK;
    HIPCHK(hipSetDevice(fe->p.device));
    int rc = bow_buffers(fe, 1, n);
    if (rc) return rc;
    const uint8_t* d[1] = {dev_desc};
    bow_launch(fe, voc, 1, d, nullptr, &n, n, levelsup);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(fe->stream));
    const double* hw = (const double*)fe->h_bow;
    const int32_t* hword = (const int32_t*)(hw + n);
    memcpy(weight, hw, (size_t)n * 8);
    memcpy(word_id, hword, (size_t)n * 4);
    memcpy(node_id, hword + n, (size_t)n * 4);
    return VSLAM_OK;
}

extern "C" int vslam_bow_transform_slots_async(vslam_fe* fe, const vslam_voc* voc, int first_slot, int nslots,
                                               int levelsup) {
    if (!fe || !voc || first_slot < 0 || nslots < 1 || first_slot + nslots > fe->B || voc->device != fe->p.device) {
        g_err = "invalid arguments";
        return VSLAM_ERR_INVALID;
    }
    HIPCHK(hipSetDevice(fe->p.device));
    int rc = bow_buffers(fe, nslots, fe->cap);
    if (rc) return rc;
    const uint8_t* d[VSLAM_MAX_BATCH];
    const int32_t* c[VSLAM_MAX_BATCH];
    int n[VSLAM_MAX_BATCH];
    for (int j = 0; j < nslots; j++) {
        d[j] = fe->d_desc + (size_t)(first_slot + j) * fe->cap * 32;
        c[j] = fe->d_counts + (first_slot + j) * 4;
        n[j] = fe->cap;
    }
    bow_launch(fe, voc, nslots, d, c, n, fe->cap, levelsup);
    HIPCHK(hipGetLastError());
    fe->bow_jobs = nslots;
    return VSLAM_OK;
}

extern "C" int vslam_bow_transform_slots_wait(vslam_fe* fe, const int* n, int32_t* const* word_id, double* const* weight,
                                              int32_t* const* node_id) {
    if (!fe || fe->bow_jobs < 1 || !n) {
        g_err = "nothing enqueued";
        return VSLAM_ERR_INVALID;
    }
    HIPCHK(hipSetDevice(fe->p.device));
    HIPCHK(hipStreamSynchronize(fe->stream));
    const int nj = fe->bow_jobs, per = fe->cap;
    const double* hw = (const double*)fe->h_bow;
    const int32_t* hword = (const int32_t*)(hw + (size_t)nj * per);
    const int32_t* hnid = hword + (size_t)nj * per;
    for (int j = 0; j < nj; j++) {
        const int m = std::min(n[j], per);
        if (weight && weight[j]) memcpy(weight[j], hw + (size_t)j * per, (size_t)m * 8);
        if (word_id && word_id[j]) memcpy(word_id[j], hword + (size_t)j * per, (size_t)m * 4);
        if (node_id && node_id[j]) memcpy(node_id[j], hnid + (size_t)j * per, (size_t)m * 4);
    }
    return VSLAM_OK;
}

/* ==================================================================================================
 * FMatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vector<MapPoint*>& vpMapPointMatches)
 * (fmatcher.cpp:546-748, pinhole frames).  Only features under the same vocabulary node are compared, and a
 * frame feature belongs to exactly one node, so the "already matched" bookkeeping never crosses nodes: one wave
 * per shared node walks that node's KeyFrame features in order (the sequential part is ~20 steps), its lanes hold
 * the node's frame features.  best = wave-min of dist << 16 | position (first wins), second = multiset second
 * minimum; TH_LOW and the ratio test as in the reference.  The rotation histogram spans all nodes: k_sbow_finish.
 * ================================================================================================== */
#include "vslam_wave.h"
#define SBOW_TH_LOW 50
#define SBOW_HISTO 30
#define SBOW_MAXC 32 /* candidates per lane: 2048 frame features under one node */

struct SbowArgs {
    const uint8_t *kfDesc, *fDesc, *kfFlags;
    const uint8_t* fFlags; /* KeyFrame-KeyFrame overload: candidate must carry a good MapPoint; null otherwise */
    int32_t strictTh;      /* KeyFrame-KeyFrame overload tests bestDist1 < TH_LOW, KeyFrame-Frame <= TH_LOW */
    int32_t outByQuery;    /* 1: out[idx1] = idx2 (vpMatches12); 0: out[idxF] = idxKF (vpMapPointMatches) */
    int32_t nOut;
    const float *kfAngle, *fAngle;
    const int32_t *kfNodes, *kfOff, *kfFeat, *fNodes, *fOff, *fFeat;
    int32_t nKFnodes, nFnodes, nF, checkOri;
    float nnratio;
    int32_t* matchF;   /* nOut, initialised to -1 by the launch wrapper */
    uint8_t* matchBin; /* nOut */
    int32_t* nmatches;
    int32_t* overflow;
};

__global__ void __launch_bounds__(64) k_sbow_nodes(SbowArgs A) {
    const int lane = threadIdx.x;
    const int kn = blockIdx.x;
    if (kn >= A.nKFnodes) return;
    const int node = A.kfNodes[kn];
    int lo = 0, hi = A.nFnodes; /* lower_bound in F's node list */
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (A.fNodes[mid] < node) lo = mid + 1;
        else hi = mid;
    }
    if (lo >= A.nFnodes || A.fNodes[lo] != node) return;
    const int f0 = A.fOff[lo], cF = A.fOff[lo + 1] - f0;
    if (cF > 64 * SBOW_MAXC) {
        if (lane == 0) atomicExch(A.overflow, 1);
        return;
    }
    uint32_t occ = 0; /* bit c: my c-th candidate already carries a MapPoint */
    const float factor = 1.0f / SBOW_HISTO;
    for (int a = A.kfOff[kn]; a < A.kfOff[kn + 1]; a++) {
        const int realIdxKF = A.kfFeat[a];
        if (!A.kfFlags[realIdxKF]) continue; /* !pMP || pMP->isBad() */
        const uint4 da = ((const uint4*)A.kfDesc)[(size_t)realIdxKF * 2], db = ((const uint4*)A.kfDesc)[(size_t)realIdxKF * 2 + 1];
        uint32_t k1 = 0xFFFFFFFFu, d2 = 256u; /* my best key, my second-best distance */
        for (int c = 0, b = lane; b < cF; c++, b += 64) {
            if (occ & (1u << c)) continue; /* vpMapPointMatches[realIdxF] / vbMatched2[idx2] */
            const int realIdxF = A.fFeat[f0 + b];
            if (A.fFlags && !A.fFlags[realIdxF]) continue; /* !pMP2 || pMP2->isBad() */
            const uint4 ta = ((const uint4*)A.fDesc)[(size_t)realIdxF * 2], tb = ((const uint4*)A.fDesc)[(size_t)realIdxF * 2 + 1];
            const uint32_t dist = __popc(da.x ^ ta.x) + __popc(da.y ^ ta.y) + __popc(da.z ^ ta.z) + __popc(da.w ^ ta.w) +
                                  __popc(db.x ^ tb.x) + __popc(db.y ^ tb.y) + __popc(db.z ^ tb.z) + __popc(db.w ^ tb.w);
            const uint32_t key = (dist << 16) | (uint32_t)b;
            if (key < k1) {
                if (k1 != 0xFFFFFFFFu) d2 = min(d2, k1 >> 16);
                k1 = key;
            } else {
                d2 = min(d2, dist);
            }
        }
        const uint32_t g = wave_min_u32(k1);
        if (g == 0xFFFFFFFFu) continue; /* no free candidate: bestDist1 stays 256 */
        const uint32_t contrib = (k1 == g) ? d2 : (k1 == 0xFFFFFFFFu ? 256u : min(k1 >> 16, 256u));
        const uint32_t second = wave_min_u32(contrib);
        const uint32_t bestDist1 = g >> 16;
        const bool under = A.strictTh ? bestDist1 < SBOW_TH_LOW : bestDist1 <= SBOW_TH_LOW;
        if (under && (float)(int)bestDist1 < __fmul_rn(A.nnratio, (float)(int)second)) {
            const int bpos = (int)(g & 0xFFFFu);
            if ((bpos & 63) == lane) occ |= 1u << (bpos >> 6);
            if (lane == 0) {
                const int bestIdxF = A.fFeat[f0 + bpos];
                const int oi = A.outByQuery ? realIdxKF : bestIdxF;
                A.matchF[oi] = A.outByQuery ? bestIdxF : realIdxKF;
                uint8_t bin = 255;
                if (A.checkOri) {
                    float rot = __fsub_rn(A.kfAngle[realIdxKF], A.fAngle[bestIdxF]);
                    if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
                    int bi = (int)roundf(__fmul_rn(rot, factor));
                    if (bi == SBOW_HISTO) bi = 0;
                    bin = (uint8_t)bi;
                }
                A.matchBin[oi] = bin;
            }
        }
    }
}

__global__ void __launch_bounds__(256) k_sbow_finish(SbowArgs A) {
    __shared__ int s_hist[SBOW_HISTO];
    __shared__ int s_n, s_removed;
    const int tid = threadIdx.x;
    if (tid < SBOW_HISTO) s_hist[tid] = 0;
    if (tid == 0) {
        s_n = 0;
        s_removed = 0;
    }
    __syncthreads();
    int mine = 0;
    for (int i = tid; i < A.nOut; i += 256)
        if (A.matchF[i] >= 0) {
            mine++;
            if (A.checkOri) atomicAdd(&s_hist[A.matchBin[i]], 1);
        }
    atomicAdd(&s_n, mine);
    __syncthreads();
    if (A.checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0; /* ComputeThreeMaxima, fmatcher.cpp:2813-2854 */
        for (int i = 0; i < SBOW_HISTO; i++) {
            const int sv = s_hist[i];
            if (sv > max1) {
                max3 = max2; max2 = max1; max1 = sv;
                ind3 = ind2; ind2 = ind1; ind1 = i;
            } else if (sv > max2) {
                max3 = max2; max2 = sv;
                ind3 = ind2; ind2 = i;
            } else if (sv > max3) {
                max3 = sv;
                ind3 = i;
            }
        }
        if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) { ind3 = -1; }
        int rem = 0;
        for (int i = tid; i < A.nOut; i += 256)
            if (A.matchF[i] >= 0) {
                const int b = A.matchBin[i];
                if (b != ind1 && b != ind2 && b != ind3) {
                    A.matchF[i] = -1;
                    rem++;
                }
            }
        atomicAdd(&s_removed, rem);
    }
    __syncthreads();
    if (tid == 0) A.nmatches[0] = s_n - s_removed;
}

__global__ void k_fill_i32(int32_t* p, int n, int32_t v) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

static int search_by_bow_impl(vslam_fe* fe, const vslam_kp* kf_kps_host, const uint8_t* dev_kf_desc,
                              const uint8_t* kf_flags_host, int n_kf, const int32_t* kf_fv_nodes,
                              const int32_t* kf_fv_off, const int32_t* kf_fv_feat, int n_kf_nodes,
                              const vslam_kp* f_kps_host, const uint8_t* dev_f_desc, const uint8_t* f_flags_host, int n_f,
                              const int32_t* f_fv_nodes, const int32_t* f_fv_off, const int32_t* f_fv_feat,
                              int n_f_nodes, float nnratio, int check_orientation, int kf_kf, int32_t* match_f,
                              int* nmatches) {
    const int n_out = kf_kf ? n_kf : n_f;
    if (!fe || n_kf < 0 || n_f < 0 || n_kf_nodes < 0 || n_f_nodes < 0 || !nmatches ||
        (n_kf && (!kf_kps_host || !dev_kf_desc || !kf_flags_host)) || (n_f && (!f_kps_host || !dev_f_desc)) ||
        (n_out && !match_f) || (kf_kf && n_f && !f_flags_host) ||
        (n_kf_nodes && (!kf_fv_nodes || !kf_fv_off || !kf_fv_feat)) || (n_f_nodes && (!f_fv_nodes || !f_fv_off || !f_fv_feat))) {
        g_err = "invalid arguments";
        return VSLAM_ERR_INVALID;
    }
    *nmatches = 0;
    for (int i = 0; i < n_out; i++) match_f[i] = -1;
    if (n_kf == 0 || n_f == 0 || n_kf_nodes == 0 || n_f_nodes == 0) return VSLAM_OK;
    for (int j = 0; j < n_kf_nodes; j++)
        for (int a = kf_fv_off[j]; a < kf_fv_off[j + 1]; a++)
            if (kf_fv_feat[a] < 0 || kf_fv_feat[a] >= n_kf) {
                g_err = "KeyFrame FeatureVector index out of range";
                return VSLAM_ERR_INVALID;
            }
    for (int j = 0; j < n_f_nodes; j++)
        for (int a = f_fv_off[j]; a < f_fv_off[j + 1]; a++)
            if (f_fv_feat[a] < 0 || f_fv_feat[a] >= n_f) {
                g_err = "Frame FeatureVector index out of range";
                return VSLAM_ERR_INVALID;
            }
    HIPCHK(hipSetDevice(fe->p.device));
    auto al = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const int tk = kf_fv_off[n_kf_nodes], tf = f_fv_off[n_f_nodes];
    const size_t o_kfl = 0, o_ka = al(o_kfl + (size_t)n_kf), o_fa = al(o_ka + (size_t)n_kf * 4),
                 o_kn = al(o_fa + (size_t)n_f * 4), o_ko = al(o_kn + (size_t)n_kf_nodes * 4),
                 o_kf = al(o_ko + (size_t)(n_kf_nodes + 1) * 4), o_fn = al(o_kf + (size_t)tk * 4),
                 o_fo = al(o_fn + (size_t)n_f_nodes * 4), o_ff = al(o_fo + (size_t)(n_f_nodes + 1) * 4),
                 o_ffl = al(o_ff + (size_t)tf * 4), in_bytes = al(o_ffl + (size_t)n_f);
    const size_t o_m = in_bytes, o_b = al(o_m + (size_t)n_out * 4), o_n = al(o_b + (size_t)n_out), total = o_n + 16;
    int rc = vslam_ensure((void**)&fe->d_proj, &fe->proj_bytes, total);
    if (rc) return rc;
    if ((rc = vslam_ensure_pinned(&fe->h_proj, &fe->h_proj_bytes, total))) return rc;
    uint8_t *h = fe->h_proj, *d = fe->d_proj;
    memcpy(h + o_kfl, kf_flags_host, (size_t)n_kf);
    for (int i = 0; i < n_kf; i++) ((float*)(h + o_ka))[i] = kf_kps_host[i].angle;
    for (int i = 0; i < n_f; i++) ((float*)(h + o_fa))[i] = f_kps_host[i].angle;
    memcpy(h + o_kn, kf_fv_nodes, (size_t)n_kf_nodes * 4);
    memcpy(h + o_ko, kf_fv_off, (size_t)(n_kf_nodes + 1) * 4);
    memcpy(h + o_kf, kf_fv_feat, (size_t)tk * 4);
    memcpy(h + o_fn, f_fv_nodes, (size_t)n_f_nodes * 4);
    memcpy(h + o_fo, f_fv_off, (size_t)(n_f_nodes + 1) * 4);
    memcpy(h + o_ff, f_fv_feat, (size_t)tf * 4);
    if (f_flags_host) memcpy(h + o_ffl, f_flags_host, (size_t)n_f);
    hipStream_t st = fe->stream;
    CopyRanges R;
    memset(&R, 0, sizeof(R));
    R.dst[0] = d;
    R.src[0] = h;
    R.bytes[0] = in_bytes;
    R.n = 1;
    vk_copy_ranges(st, R);
    SbowArgs A;
    memset(&A, 0, sizeof(A));
    A.kfDesc = dev_kf_desc; A.fDesc = dev_f_desc; A.kfFlags = d + o_kfl;
    A.kfAngle = (const float*)(d + o_ka); A.fAngle = (const float*)(d + o_fa);
    A.kfNodes = (const int32_t*)(d + o_kn); A.kfOff = (const int32_t*)(d + o_ko); A.kfFeat = (const int32_t*)(d + o_kf);
    A.fNodes = (const int32_t*)(d + o_fn); A.fOff = (const int32_t*)(d + o_fo); A.fFeat = (const int32_t*)(d + o_ff);
    A.fFlags = f_flags_host ? d + o_ffl : nullptr;
    A.strictTh = kf_kf; A.outByQuery = kf_kf; A.nOut = n_out;
    A.nKFnodes = n_kf_nodes; A.nFnodes = n_f_nodes; A.nF = n_f; A.checkOri = check_orientation; A.nnratio = nnratio;
    A.matchF = (int32_t*)(d + o_m); A.matchBin = d + o_b; A.nmatches = (int32_t*)(d + o_n);
    A.overflow = (int32_t*)(d + o_n) + 1;
    hipLaunchKernelGGL(k_fill_i32, dim3((n_out + 255) / 256), dim3(256), 0, st, A.matchF, n_out, -1);
    hipLaunchKernelGGL(k_fill_i32, dim3(1), dim3(256), 0, st, A.nmatches, 4, 0);
    hipLaunchKernelGGL(k_sbow_nodes, dim3(n_kf_nodes), dim3(64), 0, st, A);
    hipLaunchKernelGGL(k_sbow_finish, dim3(1), dim3(256), 0, st, A);
    R.dst[0] = h + o_m;
    R.src[0] = d + o_m;
    R.bytes[0] = (o_n + 16) - o_m;
    vk_copy_ranges(st, R);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    if (((const int32_t*)(h + o_n))[1]) {
        g_err = "SearchByBoW: more than 2048 frame features under one vocabulary node";
        return VSLAM_ERR_UNSUPPORTED;
    }
    memcpy(match_f, h + o_m, (size_t)n_out * 4);
    *nmatches = *(const int32_t*)(h + o_n);
    return VSLAM_OK;
}

extern "C" int vslam_search_by_bow(vslam_fe* fe, const vslam_kp* kf_kps_host, const uint8_t* dev_kf_desc,
                                   const uint8_t* kf_flags_host, int n_kf, const int32_t* kf_fv_nodes,
                                   const int32_t* kf_fv_off, const int32_t* kf_fv_feat, int n_kf_nodes,
                                   const vslam_kp* f_kps_host, const uint8_t* dev_f_desc, int n_f,
                                   const int32_t* f_fv_nodes, const int32_t* f_fv_off, const int32_t* f_fv_feat,
                                   int n_f_nodes, float nnratio, int check_orientation, int32_t* match_f,
                                   int* nmatches) {
    return search_by_bow_impl(fe, kf_kps_host, dev_kf_desc, kf_flags_host, n_kf, kf_fv_nodes, kf_fv_off, kf_fv_feat,
                              n_kf_nodes, f_kps_host, dev_f_desc, nullptr, n_f, f_fv_nodes, f_fv_off, f_fv_feat, n_f_nodes,
                              nnratio, check_orientation, 0, match_f, nmatches);
}

extern "C" int vslam_search_by_bow_keyframes(vslam_fe* fe, const vslam_kp* kps1_host, const uint8_t* dev_desc1,
                                             const uint8_t* flags1_host, int n1, const int32_t* fv1_nodes,
                                             const int32_t* fv1_off, const int32_t* fv1_feat, int n1_nodes,
                                             const vslam_kp* kps2_host, const uint8_t* dev_desc2,
                                             const uint8_t* flags2_host, int n2, const int32_t* fv2_nodes,
                                             const int32_t* fv2_off, const int32_t* fv2_feat, int n2_nodes,
                                             float nnratio, int check_orientation, int32_t* match12, int* nmatches) {
    return search_by_bow_impl(fe, kps1_host, dev_desc1, flags1_host, n1, fv1_nodes, fv1_off, fv1_feat, n1_nodes, kps2_host,
                              dev_desc2, flags2_host, n2, fv2_nodes, fv2_off, fv2_feat, n2_nodes, nnratio,
                              check_orientation, 1, match12, nmatches);
}

/* ==================================================================================================
 * FMatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo, bCoarse)
 * (fmatcher.cpp:1242-1482; SearchForTriangulation_ :1484-1725 is the same walk on cv::Matx), pinhole KeyFrames
 * without a second camera.  The reference never sets vbMatched2, so every KeyFrame-1 feature is an independent
 * query: among the features of the same vocabulary node in KeyFrame 2 that have no MapPoint, pass the stereo
 * filter, the epipole distance test and Pinhole::epipolarConstrain (pinhole.cpp:121-143), the one with the least
 * distance <= TH_LOW wins and the LAST one wins a tie (`dist > bestDist` skips, equality replaces).  One wave per
 * KeyFrame-1 feature (a wave per node would serialise the few heavy nodes: 1.0 ms -> see DESIGN.md); its lanes hold
 * the node's KeyFrame-2 features, key = dist << 20 | (0xFFFFF - position) -> wave min.
 * ================================================================================================== */
struct StriArgs {
    const uint8_t *desc1, *desc2, *hasMp1, *hasMp2;
    const float *uRight1, *uRight2;
    const vslam_kp *kps1, *kps2;
    const int32_t *nodes1, *off1, *feat1, *nodes2, *off2, *feat2;
    int32_t nNodes1, nNodes2, nFeat1, onlyStereo, coarse, checkOri, nlevels;
    float F12[9], epx, epy;
    float scale2[VSLAM_MAX_LEVELS], sigma2_2[VSLAM_MAX_LEVELS]; /* pKF2->mvScaleFactors, mvLevelSigma2 */
    int32_t* match12;  /* n1, initialised to -1 */
    uint8_t* matchBin; /* n1 */
};

__global__ void __launch_bounds__(256) k_stri_queries(StriArgs A) {
    const int lane = threadIdx.x & 63;
    const int a1 = blockIdx.x * 4 + (threadIdx.x >> 6); /* position in KeyFrame 1's flattened FeatureVector */
    if (a1 >= A.nFeat1) return; /* wave-uniform */
    int lo = 0, hi = A.nNodes1; /* node of a1: last kn with off1[kn] <= a1 (empty nodes cannot own a position) */
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (A.off1[mid] <= a1) lo = mid;
        else hi = mid;
    }
    const int node = A.nodes1[lo];
    lo = 0, hi = A.nNodes2;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (A.nodes2[mid] < node) lo = mid + 1;
        else hi = mid;
    }
    if (lo >= A.nNodes2 || A.nodes2[lo] != node) return;
    const int f0 = A.off2[lo], c2 = A.off2[lo + 1] - f0;
    const float factor = 1.0f / SBOW_HISTO;
    const int idx1 = A.feat1[a1];
    if (A.hasMp1[idx1]) return; /* pMP1 */
    const bool bStereo1 = A.uRight1[idx1] >= 0.f;
    if (A.onlyStereo && !bStereo1) return;
    const vslam_kp kp1 = A.kps1[idx1];
    const uint4 da = ((const uint4*)A.desc1)[(size_t)idx1 * 2], db = ((const uint4*)A.desc1)[(size_t)idx1 * 2 + 1];
    /* epipolar line l = x1' F12 (pinhole.cpp:129-131) */
    const float ea = __fadd_rn(__fadd_rn(__fmul_rn(kp1.x, A.F12[0]), __fmul_rn(kp1.y, A.F12[3])), A.F12[6]);
    const float eb = __fadd_rn(__fadd_rn(__fmul_rn(kp1.x, A.F12[1]), __fmul_rn(kp1.y, A.F12[4])), A.F12[7]);
    const float ec = __fadd_rn(__fadd_rn(__fmul_rn(kp1.x, A.F12[2]), __fmul_rn(kp1.y, A.F12[5])), A.F12[8]);
    const float den = __fadd_rn(__fmul_rn(ea, ea), __fmul_rn(eb, eb));
    uint32_t best = 0xFFFFFFFFu;
    for (int b = lane; b < c2; b += 64) {
        const int idx2 = A.feat2[f0 + b];
        if (A.hasMp2[idx2]) continue; /* pMP2 (vbMatched2 is never set) */
        const bool bStereo2 = A.uRight2[idx2] >= 0.f;
        if (A.onlyStereo && !bStereo2) continue;
        const uint4 ta = ((const uint4*)A.desc2)[(size_t)idx2 * 2], tb = ((const uint4*)A.desc2)[(size_t)idx2 * 2 + 1];
        const uint32_t dist = __popc(da.x ^ ta.x) + __popc(da.y ^ ta.y) + __popc(da.z ^ ta.z) + __popc(da.w ^ ta.w) +
                              __popc(db.x ^ tb.x) + __popc(db.y ^ tb.y) + __popc(db.z ^ tb.z) + __popc(db.w ^ tb.w);
        if (dist > SBOW_TH_LOW) continue;
        const vslam_kp kp2 = A.kps2[idx2];
        const int oct2 = min(max(kp2.octave, 0), A.nlevels - 1);
        if (!bStereo1 && !bStereo2) { /* too close to the epipole, :1366-1374 */
            const float dex = __fsub_rn(A.epx, kp2.x), dey = __fsub_rn(A.epy, kp2.y);
            if (__fadd_rn(__fmul_rn(dex, dex), __fmul_rn(dey, dey)) < __fmul_rn(100.f, A.scale2[oct2])) continue;
        }
        if (!A.coarse) {
            if (den == 0.f) continue;
            const float num = __fadd_rn(__fadd_rn(__fmul_rn(ea, kp2.x), __fmul_rn(eb, kp2.y)), ec);
            const float dsqr = __fdiv_rn(__fmul_rn(num, num), den);
            if (!((double)dsqr < __dmul_rn(3.84, (double)A.sigma2_2[oct2]))) continue;
        }
        best = min(best, (dist << 20) | (0xFFFFFu - (uint32_t)b));
    }
    const uint32_t g = wave_min_u32(best);
    if (g == 0xFFFFFFFFu) return;
    if (lane == 0) {
        const int bestIdx2 = A.feat2[f0 + (int)(0xFFFFFu - (g & 0xFFFFFu))];
        A.match12[idx1] = bestIdx2;
        uint8_t bin = 255;
        if (A.checkOri) {
            float rot = __fsub_rn(kp1.angle, A.kps2[bestIdx2].angle);
            if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
            int bi = (int)roundf(__fmul_rn(rot, factor));
            if (bi == SBOW_HISTO) bi = 0;
            bin = (uint8_t)bi;
        }
        A.matchBin[idx1] = bin;
    }
}

static int fv_in_range(const int32_t* off, const int32_t* feat, int nnodes, int n) {
    for (int j = 0; j < nnodes; j++)
        for (int a = off[j]; a < off[j + 1]; a++)
            if (feat[a] < 0 || feat[a] >= n) return 0;
    return 1;
}

extern "C" int vslam_search_for_triangulation(vslam_fe* fe, const vslam_tri_params* p, const vslam_kp* kps1_host,
                                              const uint8_t* dev_desc1, const uint8_t* has_mp1_host,
                                              const float* u_right1_host, int n1, const int32_t* fv1_nodes,
                                              const int32_t* fv1_off, const int32_t* fv1_feat, int n1_nodes,
                                              const vslam_kp* kps2_host, const uint8_t* dev_desc2,
                                              const uint8_t* has_mp2_host, const float* u_right2_host, int n2,
                                              const int32_t* fv2_nodes, const int32_t* fv2_off, const int32_t* fv2_feat,
                                              int n2_nodes, int32_t* match12, int* nmatches) {
    if (!fe || !p || n1 < 0 || n2 < 0 || n1_nodes < 0 || n2_nodes < 0 || !nmatches ||
        (n1 && (!kps1_host || !dev_desc1 || !has_mp1_host || !u_right1_host || !match12)) ||
        (n2 && (!kps2_host || !dev_desc2 || !has_mp2_host || !u_right2_host)) ||
        (n1_nodes && (!fv1_nodes || !fv1_off || !fv1_feat)) || (n2_nodes && (!fv2_nodes || !fv2_off || !fv2_feat))) {
        g_err = "invalid arguments";
        return VSLAM_ERR_INVALID;
    }
    *nmatches = 0;
    for (int i = 0; i < n1; i++) match12[i] = -1;
    if (n1 == 0 || n2 == 0 || n1_nodes == 0 || n2_nodes == 0) return VSLAM_OK;
    if (!fv_in_range(fv1_off, fv1_feat, n1_nodes, n1) || !fv_in_range(fv2_off, fv2_feat, n2_nodes, n2)) {
        g_err = "FeatureVector index out of range";
        return VSLAM_ERR_INVALID;
    }
    if (fv2_off[n2_nodes] > 0xFFFFF) {
        g_err = "SearchForTriangulation: FeatureVector too long";
        return VSLAM_ERR_UNSUPPORTED;
    }
    HIPCHK(hipSetDevice(fe->p.device));
    auto al = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const int t1 = fv1_off[n1_nodes], t2 = fv2_off[n2_nodes];
    const size_t o_k1 = 0, o_k2 = al(o_k1 + (size_t)n1 * sizeof(vslam_kp)), o_h1 = al(o_k2 + (size_t)n2 * sizeof(vslam_kp)),
                 o_h2 = al(o_h1 + (size_t)n1), o_u1 = al(o_h2 + (size_t)n2), o_u2 = al(o_u1 + (size_t)n1 * 4),
                 o_n1 = al(o_u2 + (size_t)n2 * 4), o_o1 = al(o_n1 + (size_t)n1_nodes * 4),
                 o_f1 = al(o_o1 + (size_t)(n1_nodes + 1) * 4), o_n2 = al(o_f1 + (size_t)t1 * 4),
                 o_o2 = al(o_n2 + (size_t)n2_nodes * 4), o_f2 = al(o_o2 + (size_t)(n2_nodes + 1) * 4),
                 in_bytes = al(o_f2 + (size_t)t2 * 4);
    const size_t o_m = in_bytes, o_b = al(o_m + (size_t)n1 * 4), o_n = al(o_b + (size_t)n1), total = o_n + 16;
    int rc = vslam_ensure((void**)&fe->d_proj, &fe->proj_bytes, total);
    if (rc) return rc;
    if ((rc = vslam_ensure_pinned(&fe->h_proj, &fe->h_proj_bytes, total))) return rc;
    uint8_t *h = fe->h_proj, *d = fe->d_proj;
    memcpy(h + o_k1, kps1_host, (size_t)n1 * sizeof(vslam_kp));
    memcpy(h + o_k2, kps2_host, (size_t)n2 * sizeof(vslam_kp));
    memcpy(h + o_h1, has_mp1_host, (size_t)n1);
    memcpy(h + o_h2, has_mp2_host, (size_t)n2);
    memcpy(h + o_u1, u_right1_host, (size_t)n1 * 4);
    memcpy(h + o_u2, u_right2_host, (size_t)n2 * 4);
    memcpy(h + o_n1, fv1_nodes, (size_t)n1_nodes * 4);
    memcpy(h + o_o1, fv1_off, (size_t)(n1_nodes + 1) * 4);
    memcpy(h + o_f1, fv1_feat, (size_t)t1 * 4);
    memcpy(h + o_n2, fv2_nodes, (size_t)n2_nodes * 4);
    memcpy(h + o_o2, fv2_off, (size_t)(n2_nodes + 1) * 4);
    memcpy(h + o_f2, fv2_feat, (size_t)t2 * 4);
    hipStream_t st = fe->stream;
    CopyRanges R;
    memset(&R, 0, sizeof(R));
    R.dst[0] = d;
    R.src[0] = h;
    R.bytes[0] = in_bytes;
    R.n = 1;
    vk_copy_ranges(st, R);
    StriArgs A;
    memset(&A, 0, sizeof(A));
    A.desc1 = dev_desc1; A.desc2 = dev_desc2; A.hasMp1 = d + o_h1; A.hasMp2 = d + o_h2;
    A.uRight1 = (const float*)(d + o_u1); A.uRight2 = (const float*)(d + o_u2);
    A.kps1 = (const vslam_kp*)(d + o_k1); A.kps2 = (const vslam_kp*)(d + o_k2);
    A.nodes1 = (const int32_t*)(d + o_n1); A.off1 = (const int32_t*)(d + o_o1); A.feat1 = (const int32_t*)(d + o_f1);
    A.nodes2 = (const int32_t*)(d + o_n2); A.off2 = (const int32_t*)(d + o_o2); A.feat2 = (const int32_t*)(d + o_f2);
    A.nNodes1 = n1_nodes; A.nNodes2 = n2_nodes; A.nFeat1 = t1; A.onlyStereo = p->only_stereo; A.coarse = p->coarse;
    A.checkOri = p->check_orientation; A.nlevels = fe->p.nlevels;
    for (int i = 0; i < 9; i++) A.F12[i] = p->F12[i];
    A.epx = p->ep_x; A.epy = p->ep_y;
    for (int l = 0; l < fe->p.nlevels; l++) {
        A.scale2[l] = fe->tab.scale[l];
        A.sigma2_2[l] = fe->tab.sigma2[l];
    }
    A.match12 = (int32_t*)(d + o_m); A.matchBin = d + o_b;
    SbowArgs Fz; /* histogram + ComputeThreeMaxima + count: k_sbow_finish */
    memset(&Fz, 0, sizeof(Fz));
    Fz.matchF = A.match12; Fz.matchBin = A.matchBin; Fz.nOut = n1; Fz.checkOri = p->check_orientation;
    Fz.nmatches = (int32_t*)(d + o_n);
    hipLaunchKernelGGL(k_fill_i32, dim3((n1 + 255) / 256), dim3(256), 0, st, A.match12, n1, -1);
    hipLaunchKernelGGL(k_fill_i32, dim3(1), dim3(256), 0, st, Fz.nmatches, 4, 0);
    if (t1 > 0) hipLaunchKernelGGL(k_stri_queries, dim3((t1 + 3) / 4), dim3(256), 0, st, A);
    hipLaunchKernelGGL(k_sbow_finish, dim3(1), dim3(256), 0, st, Fz);
    R.dst[0] = h + o_m;
    R.src[0] = d + o_m;
    R.bytes[0] = (o_n + 16) - o_m;
    vk_copy_ranges(st, R);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    memcpy(match12, h + o_m, (size_t)n1 * 4);
    *nmatches = *(const int32_t*)(h + o_n);
    return VSLAM_OK;
}
