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
        HIPCHK(hipHostMalloc((void**)&fe->h_bow, bytes, hipHostMallocDefault));
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
