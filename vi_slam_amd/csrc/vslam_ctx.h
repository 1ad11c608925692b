/* vslam_ctx.h -- private: the vslam_fe context (HBM layout, scratch, worker pool) shared by
 * vslam_fe.hip (extractor ABI) and vslam_match.hip (matcher ABI). */
#ifndef VSLAM_CTX_H
#define VSLAM_CTX_H
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/vslam_orb_pattern.h"
#include "vslam_host.h"
#include "vslam_pool.h"
#include "vslam_kernels.h"
#include "vslam_tuning.h"

std::string& vslam_err();
#define g_err (vslam_err())

#define HIPCHK(call)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            g_err = std::string(#call) + ": " + hipGetErrorString(e_);                           \
            return VSLAM_ERR_HIP;                                                                \
        }                                                                                        \
    } while (0)

/* ------------------------------------------------------------------ context */
enum { VSLAM_REGION_NONE = 0, VSLAM_REGION_INIT = 1, VSLAM_REGION_STEREO = 2 };
struct vslam_fe {
    vslam_fe_params p;
    vslam_tuning tune; /* the context's behaviour switches, resolved once by vslam_fe_create (vslam_tuning.h) */
    vslam::ExtractorTables tab;
    PyramidGeom geom;
    size_t slot_stride = 0; /* bytes per slot in d_pyr / d_blur */
    int B = 1, cap = 0;

    hipStream_t stream = nullptr;
    hipEvent_t ev_cand = nullptr;
    hipEvent_t ev_x = nullptr; /* cross-context ordering (vslam_fe_wait_for) */
    hipEvent_t ev_user[4] = {};  /* vslam_fe_event_record / _wait */
    int desc_kpw_hint = -1;           /* keypoints per wave of the descriptor kernel for the pass being enqueued (-1: by batch size) */
    hipEvent_t ev_fast = nullptr;     /* recorded behind the FAST launch of every pass once a gate refers to this context */
    vslam_fe* fast_gate = nullptr;    /* vslam_fe_set_fast_gate: this context's FAST waits for that context's last FAST */
    bool fast_gated_by_someone = false;
    std::vector<vslam_fe*> gate_waiters; /* contexts whose fast_gate is this one: un-gated when this context is destroyed */

    uint8_t* d_pyr = nullptr;
    uint8_t* d_blur = nullptr;
    /* resize tables, one set per destination level >= 1 */
    uint16_t* d_xtab[VSLAM_MAX_LEVELS] = {};
    int16_t* d_xa[VSLAM_MAX_LEVELS] = {};
    uint16_t* d_ytab[VSLAM_MAX_LEVELS] = {};
    int16_t* d_yb[VSLAM_MAX_LEVELS] = {};
    uint16_t* d_qbase[VSLAM_MAX_LEVELS] = {};    /* k_resize_level_v2 quad tables (nullptr: generic k_resize_level) */
    ResizeQuad* d_quads[VSLAM_MAX_LEVELS] = {};
    /* fused pyramid: groups of levels per launch (k_pyramid_group); empty -> one launch per level */
    struct PyrGroupCtx {
        PyrGroupDev dev;
        size_t lds_bytes;
        PyrTileDev* d_tiles;
    };
    std::vector<PyrGroupCtx> pyr_groups;
    /* FAST cells */
    std::vector<vslam::HostCell> cells;
    int level_cell_first[VSLAM_MAX_LEVELS + 1] = {};
    CellDesc* d_cells = nullptr;
    int tile_pitch = 0, tile_rows = 0, max_px = 0;
    /* FAST bands (k_fast_bands): up to four cells of a cell row per workgroup; nbands == 0: not available for this geometry */
    OctPart* d_oct_parts = nullptr; /* k_oct_count: parts of every level; nullptr: walk 1 inside k_octree_v4 */
    uint32_t* d_oct_cnt = nullptr;  /* leaf counters [B][sum of the levels' leaves] */
    int oct_maxcells = 0;
    BandDesc* d_bands = nullptr;
    uint8_t* d_band_classes = nullptr; /* 272-byte column tables, BandDesc::lnw >> 16 indexes them */
    int nbands = 0, band_max_wh = 0, band_max_iw = 0;
    /* candidates: per slot [total, overflow, CellOut[ncells], cand[cand_cap]] */
    uint8_t* d_cand = nullptr;
    uint8_t* h_cand = nullptr; /* pinned */
    size_t cand_stride = 0;
    int cand_cap = 0;
    int32_t taps[7];
    uint32_t* d_blur_tasks = nullptr;
    int n_blur_tasks = 0;
    int oct_threads = 1024; /* threads per quadtree problem (k_octree_v4), chosen at creation */
    int blur_rows = 32; /* output rows per wave task of k_blur7_v2 (VSLAM_BLUR_ROWS) */
    /* selection + outputs */
    SelKp* d_sel = nullptr;
    SelKp* h_sel = nullptr; /* pinned, B*cap */
    uint8_t* d_res = nullptr;  /* the context's result block: counts (res_counts_bytes) | kps B*cap | desc B*cap*32 */
    uint8_t* h_res = nullptr;  /* pinned mirror, same layout */
    size_t res_bytes = 0, res_counts_bytes = 0, res_feat_bytes = 0, res_init_bytes = 0;
    bool init_in_block = false;    /* d_init / h_init are the block's own matcher region (not allocations of their own) */
    /* result deliveries of the extraction / SearchForInitialization paths so far (vslam_fe_delivery_stats) */
    unsigned long long n_deliveries = 0, n_delivery_bytes = 0, graph_deliveries = 0, graph_delivery_bytes = 0;
    bool deliver_deferred = false; /* an extraction with want_host = 2 is waiting for the matcher call that delivers both */
    vslam_kp* d_kps = nullptr; /* views into d_res / h_res */
    uint8_t* d_desc = nullptr;
    vslam_kp* h_kps = nullptr;
    uint8_t* h_desc = nullptr;
    int8_t* d_pattern = nullptr;
    BatchSrc src;
    int n_out[VSLAM_MAX_BATCH] = {};
    int mono_out[VSLAM_MAX_BATCH] = {};
    /* quadtree statistics (vslam_fe_octree_stats): (slot, level) problems distributed on the device so far, on how many
     * of them k_octree_v4 split nodes below its fine grid, and the per-slot level masks of the last pass */
    unsigned long long oct_problems = 0, oct_deep = 0;
    uint32_t oct_last_mask[VSLAM_MAX_BATCH] = {};
    int32_t pack_hdr[VSLAM_MAX_BATCH][4] = {}; /* headers of vslam_fe_pack_slots while the copy is in flight */
    std::vector<std::vector<vslam::Cand>> sel_level; /* [slot*nlevels + level] */
    std::vector<std::vector<vslam::Cand>> cand_level;
    /* matcher scratch (grown on demand) */
    uint8_t* h_top2 = nullptr; /* pinned: idx2 | dist2 of the batched brute-force matcher */
    size_t h_top2_bytes = 0;
    uint32_t* d_part = nullptr;
    size_t part_bytes = 0;
    int32_t* d_idx2 = nullptr;
    int32_t* d_dist2 = nullptr;
    size_t top2_cap = 0;
    uint8_t* d_dmat = nullptr;
    size_t dmat_bytes = 0;
    uint8_t* d_tmp_desc[2] = {nullptr, nullptr};
    size_t tmp_desc_bytes[2] = {0, 0};
    /* device SearchForInitialization: matches12 | prevMatched | nmatches of every pair */
    uint8_t* d_init = nullptr;
    size_t init_bytes = 0;
    uint8_t* h_init = nullptr; /* pinned */
    size_t h_init_bytes = 0;
    int init_pairs = 0;
    bool init_lds_set = false;
    uint8_t* d_init_scratch = nullptr; /* k_si_topm -> k_si_replay: compacted octave-0 lists + sorted prefixes */
    size_t init_scratch_bytes = 0;
    int* d_init_fb = nullptr; /* number of full re-scans in k_si_replay / k_sbp_replay (diagnostics) */
    uint8_t* d_proj = nullptr;  /* SearchByProjection: uploaded last-frame arrays | scratch | results */
    size_t proj_bytes = 0;
    uint8_t* h_proj = nullptr;  /* pinned mirror (inputs, then results) */
    size_t h_proj_bytes = 0;
    bool proj_lds_set = false, dist_lds_set = false;
    uint8_t* d_bow = nullptr;   /* ComputeBoW: per-feature weight | word | node for nslots x cap features */
    size_t bow_bytes = 0;
    uint8_t* h_bow = nullptr;   /* pinned mirror */
    size_t h_bow_bytes = 0;
    int bow_jobs = 0;
    bool use_graph = true;          /* host-image passes replay a captured HIP graph (VSLAM_GRAPH=0 disables) */
    hipGraphExec_t graph_exec = nullptr;
    long long graph_key = 0;
    size_t graph_pitch = 0;                          /* pinned-image passes: the captured pull reads these */
    const uint8_t* graph_imgs[VSLAM_MAX_BATCH] = {};
    int graph_lap0 = 0, graph_lap1 = 0;
    uint8_t* h_img = nullptr;   /* pinned staging for host images: B x height x level-0 pitch */
    size_t h_img_pitch = 0;     /* row pitch of the images staged last: the level-0 pitch, or the width for dense sources */
    uint8_t* d_stage = nullptr; /* device staging of pinned caller images copied by the DMA engines (VSLAM_H2D=sdma) */
    size_t d_stage_bytes = 0;
    uint8_t* d_sbp = nullptr;   /* batched device-resident SearchByProjection: scratch + results per job */
    size_t sbp_bytes = 0;
    uint8_t* h_sbp = nullptr;   /* pinned results */
    size_t h_sbp_bytes = 0;
    int sbp_jobs = 0, sbp_M = 8;
    float* d_x3dw = nullptr;    /* stereo points per pair: B x cap x 3 */
    uint8_t* d_mpflags = nullptr; /* B x cap */
    /* stereo scratch */
    void* d_stereo = nullptr;
    size_t stereo_bytes = 0;
    float* h_stereo = nullptr; /* pinned: [uRight | depth] x pairs x cap */
    float *d_stereo_u = nullptr, *d_stereo_depth = nullptr; /* mvuRight / mvDepth of the last stereo pass on the device */
    float* h_stereo_cur = nullptr; /* where the last stereo pass delivered: h_stereo, or the result block's matcher region */
    int block_region_owner = 0;    /* VSLAM_REGION_*: which matcher's outputs live in the result block's extra region */
    size_t h_stereo_bytes = 0;
    int stereo_pairs = 0;
    int stereo_capR = 0; /* slot capacity of the right context of the last stereo enqueue (scratch carving) */
    int stereo_slotL[VSLAM_MAX_STEREO_JOBS] = {};

    /* GPU quadtree distribution */
    bool dev_octree = false;
    OctParams oct;
    uint32_t* d_pts[2] = {nullptr, nullptr};  /* B x cand_cap each: keys in key order | k_octree_v4: fine cell and rank of every key */
    uint16_t* d_nid = nullptr;                /* k_octree_v2 only: node (list index) of every key */
    uint8_t* d_oct_sorted = nullptr;          /* k_octree_v4: {key, position} sorted by fine cell, B x cand_cap x 8 bytes */
    uint32_t* d_oct_lut = nullptr;            /* k_octree_v4: per level the x and y path tables (vslam::build_oct_lut) */
    int32_t* d_oct_redo = nullptr;            /* k_octree_v4: (slot, level) split nodes finer than the grid, B x VSLAM_MAX_LEVELS */
    uint32_t* d_sel_xyr = nullptr;            /* per slot / level result lists */
    int32_t* d_sel_cnt = nullptr;
    int32_t* d_counts = nullptr;              /* [slot][4] = n, monoIndex, -, - ; then [B*4] = error flags */
    int32_t* h_counts = nullptr;              /* pinned mirror */
    bool cand_on_host = false;                /* h_cand holds the last batch's candidates */
    int last_nimg = 0;

    /* optional HIP-event timing of the kernel stages (bench.py roofline): resize x7, fast, blur, describe,
     * quadtree (+ output order) */
    bool profiling = false;
    hipEvent_t ev_prof[10] = {};
    double prof_ms[5] = {0, 0, 0, 0, 0};
    long prof_batches = 0, prof_images = 0;

    WorkerPool* pool = nullptr;
};


int vslam_ensure(void** p, size_t* have, size_t want);
int vslam_ensure_pinned(uint8_t** p, size_t* have, size_t want); /* grow-only pinned host buffer */
int vslam_pinned_alloc(void** p, size_t bytes);                  /* hipHostMalloc on the device's NUMA node; returns a hipError_t */
int vslam_enqueue_extract(vslam_fe* fe, int nimg, const uint8_t* const* imgs, size_t pitch, int on_device,
                          int lap0, int lap1, int want_host);
int vslam_finish_extract(vslam_fe* fe, int nimg);
int vslam_deliver(vslam_fe* fe, int nimg, vslam_kp* const* kps, uint8_t* const* desc, int cap, int* n,
                  int* mono_index);


/* Host wait for a stream.  hipStreamSynchronize blocks on an interrupt after a short spin; VSLAM_WAIT=spin polls
 * hipStreamQuery instead (a core per waiting thread, lower wake-up latency). */
static inline hipError_t vslam_stream_wait(hipStream_t st) {
    const bool spin = vslam_process_tuning().wait_spin == 1; /* process-wide switch, resolved once (vslam_tuning.h) */
    if (!spin) return hipStreamSynchronize(st);
    hipError_t r;
    while ((r = hipStreamQuery(st)) == hipErrorNotReady) {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    return r;
}

/* bookkeeping of the result deliveries (vslam_fe_delivery_stats): `ops` copy operations carrying the ranges of R */
static inline void vslam_count_delivery(vslam_fe* fe, int ops, const CopyRanges& R) {
    fe->n_deliveries += (unsigned long long)ops;
    for (int r = 0; r < R.n; r++) fe->n_delivery_bytes += R.bytes[r];
}

#endif
