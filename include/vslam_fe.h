/* include/vslam_fe.h -- C ABI of the MI355X-native visual front-end (libvslam_fe.so).
 *
 * Drop-in boundary for ONE hot path of KMS-TEAM/vi_slam: ORB extraction + 256-bit Hamming matching.
 * Every entry point names the reference interface it replaces (paths relative to the reference root).
 * Plain pointers and sizes only; all functions return VSLAM_OK (0) or a negative error code and never
 * abort (the reference asserts / exit()s, see fextractor.cpp:1041, cuda_common.h:49-82).
 *
 * The library is HIP-only: there is no CPU fallback.  If no gfx950 device is usable, vslam_fe_create()
 * fails with VSLAM_ERR_NO_DEVICE.
 *
 * Threading: a vslam_fe context is stateful (it owns the image pyramids, like FExtractor owns
 * mvImagePyramid, fextractor.h:64) and is NOT re-entrant; use one context per concurrent stream, exactly
 * as the reference allocates Left/Right/Ini extractors (tracking.cpp:1087-1093).  Different contexts may
 * be driven from different host threads (frame.cpp:107-108).
 */
#ifndef VSLAM_FE_H
#define VSLAM_FE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSLAM_OK 0
#define VSLAM_ERR_INVALID (-1)     /* bad argument / empty image (FExtractor::compute returns -1, :1037) */
#define VSLAM_ERR_NO_DEVICE (-2)   /* no usable HIP device */
#define VSLAM_ERR_HIP (-3)         /* a HIP runtime call failed; see vslam_last_error() */
#define VSLAM_ERR_CAPACITY (-4)    /* caller buffer or an internal candidate buffer too small */
#define VSLAM_ERR_UNSUPPORTED (-5) /* geometry the reference itself cannot process (e.g. nIni == 0) */
#define VSLAM_ERR_COMM (-6)        /* RCCL missing or a collective failed; see vslam_last_error() */

#define VSLAM_MAX_LEVELS 16
#define VSLAM_MAX_BATCH 64

/* values of the imgs_on_device argument (where the input images live) */
#define VSLAM_IMGS_HOST 0
#define VSLAM_IMGS_DEVICE 1
#define VSLAM_IMGS_PINNED 2
#define VSLAM_IMGS_STAGED 3 /* already in the slots' level 0: put there by vslam_fe_stage_images_async (imgs is ignored) */

/* flags for vslam_fe_params.flags: OpenCV build-dependent arithmetic the reference inherits */
#define VSLAM_FLAG_ATAN_FMA 1u /* cv::fastAtan2 Horner polynomial FMA-contracted (AVX2/FMA3 dispatch, aarch64) */
/* Run FExtractor::DistributeOctTree on the host worker pool instead of the GPU kernel (k_octree).  The
 * library picks this by itself only when a level's node list cannot fit LDS (nfeatures > ~11000). */
#define VSLAM_FLAG_HOST_OCTREE 2u

/* Same 28-byte layout and field order as cv::KeyPoint (pt.x, pt.y, size, angle, response, octave,
 * class_id) so std::vector<cv::KeyPoint> storage can be handed over directly. */
typedef struct vslam_kp {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} vslam_kp;

/* Behaviour switches of a context.  Every field: -1 = library default (vslam_tuning_init sets all of them to -1).
 * vslam_fe_create resolves them ONCE: a field the caller set wins, then the process default, which is read from the
 * environment variable named in the comment a single time per process (under std::call_once; documented for A/B runs of
 * whole programs), then the built-in default.  Nothing reads the environment after that, and no switch is cached in
 * unsynchronised statics, so two threads creating and using two contexts (frame.cpp:107-108) share no mutable state. */
typedef struct vslam_tuning {
    int32_t pyramid_per_level;    /* VSLAM_PYRAMID=levels: 1 = one launch per pyramid level instead of the fused groups */
    int32_t pyr_rows;             /* VSLAM_PYR_ROWS: rows of the first computed level per fused-pyramid tile (4..64, default 28) */
    int32_t pyr_threads;          /* VSLAM_PYR_NT: 256 | 512 threads per fused-pyramid tile (default 256) */
    int32_t blur_rows;            /* VSLAM_BLUR_ROWS: output rows per wave task of the blur (8..512, default 32) */
    int32_t fast_threads;         /* VSLAM_FAST_NT: 64 | 128 | 256 threads per FAST cell (default 128) */
    int32_t fast_pitch;           /* VSLAM_FAST_PITCH: 72 forces the wide LDS pitch of the FAST window */
    int32_t fast_lds_pad;         /* VSLAM_FAST_LDS_PAD: extra LDS bytes per FAST workgroup (occupancy experiments) */
    int32_t octree_walk_kernel;   /* VSLAM_OCTREE=v2: 1 = the walk-per-pass quadtree kernel only */
    int32_t oct_fine_depth;       /* VSLAM_OCT_FINE_D: depth of the one-walk kernel's fine grid (tests force 1 or 3) */
    int32_t oct_lds_budget_kb;    /* VSLAM_OCT_LDS_BUDGET_KB: LDS a quadtree workgroup may take (16..150, default 128) */
    int32_t oct_regkeys;          /* VSLAM_OCT_REGKEYS: 0 | 1 keys in registers between the key walks (default: batches <= 2) */
    int32_t oct_max_iter;         /* VSLAM_OCT_MAXITER: split-pass limit (default 64) */
    int32_t oct_debug;            /* VSLAM_OCT_DBG: 1 = allocate the stamp buffer of diagnostic builds */
    int32_t graphs;               /* VSLAM_GRAPH: 0 = never capture / replay HIP graphs */
    int32_t h2d_route;            /* VSLAM_H2D=pull|sdma: 1 = kernel reads pinned memory, 2 = DMA engines (default by batch size) */
    int32_t d2h_route;            /* VSLAM_D2H=kernel|sdma: 1 = copy kernel, 2 = hipMemcpyAsync for every result transfer */
    int32_t copy_wgs;             /* VSLAM_COPY_WGS: workgroup cap of the copy kernel (default 16) */
    int32_t pull_depth;           /* VSLAM_PULL_DEPTH: loads in flight per lane of the pull kernel */
    int32_t init_topm;            /* VSLAM_INIT_TOPM: sorted candidate prefix per query of SearchForInitialization (1..16, default 8) */
    int32_t init_match_host;      /* VSLAM_INIT_MATCH=host: 1 = order-dependent replay on the host (cross-check path) */
    int32_t sbp_topm;             /* VSLAM_SBP_TOPM: the same for SearchByProjection */
    int32_t sbp_sequential;       /* VSLAM_SBP_MODE=seq: 1 = skip the parallel resolution (cross-check path) */
    int32_t si_queries_per_block; /* VSLAM_SI_QPB: 8 | 16 | 32 queries per k_si_topm workgroup (default 16) */
    int32_t fg_threads;           /* VSLAM_FG_NT: 64 | 128 | 256 threads per grid-detector cell (default 128) */
    int32_t wait_spin;            /* VSLAM_WAIT=spin: 1 = host waits poll hipStreamQuery instead of blocking */
    int32_t numa;                 /* VSLAM_NUMA: 0 = do not allocate pinned memory from the CPUs next to the device */
    int32_t host_prof;            /* VSLAM_HOST_PROF: 1 = host-side wall time per API phase, printed at destroy */
    int32_t stream_priority;      /* VSLAM_STREAM_PRIORITY: 0 normal (default), 1 low, 2 high priority of the context's HIP stream: a
                                     priority of its own gives the context hardware queues it does not share with the host
                                     application's other streams -- and, being a priority, lets the context's kernels
                                     pre-empt the application's default-priority work: opt in (bench.py does), see
                                     INTEGRATION.md */
    int32_t stage_split_event;    /* 0..3: uploads of more than one image go as two transfers with the context's user event of that
                                     index (vslam_fe_event_wait) recorded between them; see vslam_fe_stage_images_async */
    int32_t oct_threads;          /* VSLAM_OCT_THREADS: 256 | 512 | 1024 threads per quadtree problem (default: 1024 for contexts of
                                     one or two images, 512 for frames above a megapixel, else 256) */
    int32_t fast_kernel;          /* VSLAM_FAST_KERNEL: 3 = one workgroup per FAST cell (k_fast_cells_v3; default for contexts of one
                                     or two images), 4 = one workgroup per band of cells of a cell row sharing one staged
                                     window (k_fast_bands; default for batches) */
    int32_t fast_band_cells;      /* VSLAM_FAST_BAND_CELLS: cells per band of k_fast_bands (1..4, default 4; fewer where 4 cell
                                     interiors are wider than 128 px) */
    int32_t wave_prio;            /* VSLAM_WAVE_PRIO: bit mask of kernel classes that raise their wave priority (s_setprio 3)
                                     at entry: 1 = quadtree + output order, 2 = orientation/descriptors, 4 = matchers
                                     (k_si_*, k_stereo_*), 8 = result packing; default 0 */
    int32_t oct_precount;         /* VSLAM_OCT_PRECOUNT: 1 = the quadtree's counting walk as a launch of its own with a level's keys dealt
                                     to up to eight workgroups (k_oct_count); 0 (default) = inside the quadtree workgroup */
    int32_t desc_kpw;             /* VSLAM_DESC_KPW: 1 | 4 keypoints per wave of the descriptor kernel (default: 1 for contexts of one or
                                     two images, else 4) */
    int32_t reserved[1];
} vslam_tuning;
void vslam_tuning_init(vslam_tuning* t); /* every field = -1 (library default) */

typedef struct vslam_fe_params {
    int32_t width, height;   /* level-0 image size (reference: image.cols/rows at compute()) */
    int32_t nfeatures;       /* ORBextractor.nFeatures   (tracking.cpp:1021-1085) */
    float scale_factor;      /* ORBextractor.scaleFactor */
    int32_t nlevels;         /* ORBextractor.nLevels     */
    int32_t ini_th_fast;     /* ORBextractor.iniThFAST   */
    int32_t min_th_fast;     /* ORBextractor.minThFAST   */
    int32_t device;          /* HIP device ordinal */
    int32_t max_batch;       /* image slots processed per batched call, 1..VSLAM_MAX_BATCH */
    uint32_t flags;          /* VSLAM_FLAG_* */
    int32_t gauss_taps[7];   /* all zero -> OpenCV 4.2 taps {18,34,48,56,48,34,18} */
    const vslam_tuning* tuning; /* NULL: library defaults; copied by vslam_fe_create */
} vslam_fe_params;

typedef struct vslam_fe vslam_fe;

/* ---------------------------------------------------------------- extractor (FExtractor) */

/* FExtractor::FExtractor (fextractor.cpp:401-461): allocates pyramids for max_batch image slots, builds
 * scale tables, per-level quotas, resize coefficient tables and the FAST cell list. */
int vslam_fe_create(const vslam_fe_params* params, vslam_fe** out);
/* Change switches of an existing context between calls (fields >= 0 of *t overwrite the context's; like every other
 * call on a context this is not re-entrant).  Takes effect for what is consulted per call -- the transport routes
 * (h2d_route, d2h_route, pull_depth, copy_wgs), the matcher switches (init_topm, init_match_host, sbp_topm,
 * sbp_sequential, si_queries_per_block), oct_regkeys, the FAST / pyramid launch shapes, graphs = 0; switches that shaped
 * the context's buffers at creation (pyramid plan, blur rows, quadtree grid depth and LDS placement) stay as created. */
int vslam_fe_set_tuning(vslam_fe* fe, const vslam_tuning* t);
void vslam_fe_destroy(vslam_fe* fe);
const char* vslam_last_error(void);

/* FExtractor::GetLevels / GetScaleFactors / GetInverseScaleFactors / GetScaleSigmaSquares /
 * GetInverseScaleSigmaSquares (fextractor.h:42-62) + mnFeaturesPerLevel.  Arrays of nlevels; any may be NULL. */
int vslam_fe_tables(const vslam_fe* fe, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                    int32_t* features_per_level);

/* FExtractor::compute (fextractor.h:38-40, fextractor.cpp:1034-1133) for one host image, synchronous.
 *   img/pitch : CV_8UC1 rows.   lap0/lap1 : vLappingArea (frame.cpp:107-108 passes {0,0}, :289 {0,1000}).
 *   kps/desc  : caller storage for cap keypoints / cap*32 descriptor bytes; cap >= vslam_fe_capacity(fe).
 *   *n        : keypoints written;  *mono_index : the reference's return value.
 * Uses image slot 0. */
int vslam_fe_extract(vslam_fe* fe, const uint8_t* img, size_t pitch, int lap0, int lap1, vslam_kp* kps,
                     uint8_t* desc, int cap, int* n, int* mono_index);

/* Batched form: nimg (<= max_batch) images in one pass of the kernels; slot i <- imgs[i].
 *   imgs_on_device == VSLAM_IMGS_DEVICE (1): imgs[i] are device pointers, used in place as level 0 (zero
 *   copy).  They must stay valid until the LAST consumer of this pass's level 0 has run: the stereo refinement of
 *   vslam_stereo_match(_batch) / vslam_frame_stereo_* and vslam_fe_level_copy(level 0) read them again -- i.e.
 *   until the next extraction on this context, or until the context is destroyed.
 *   == VSLAM_IMGS_HOST (0): pageable host pointers; rows are copied into the context's pinned staging by its
 *   worker pool, then pulled over PCIe.   == VSLAM_IMGS_PINNED (2): host pointers into pinned (hipHostMalloc /
 *   hipHostRegister'ed, e.g. vslam_host_alloc) memory that the GPU reads directly -- no host-side copy at all;
 *   the memory must stay untouched until the pass has been waited for.
 *   kps/desc/n/mono_index are host arrays of nimg entries;
 *   kps[i] holds cap keypoints, desc[i] cap*32 bytes.  kps/desc may be NULL to keep results on the
 *   device only (read them with vslam_fe_slot_buffers). */
int vslam_fe_extract_batch(vslam_fe* fe, int nimg, const uint8_t* const* imgs, size_t pitch,
                           int imgs_on_device, int lap0, int lap1, vslam_kp* const* kps,
                           uint8_t* const* desc, int cap, int* n, int* mono_index);

/* Upload only: pull nimg pinned (VSLAM_IMGS_PINNED) or pageable (VSLAM_IMGS_HOST) host images into level 0 of slots
 * 0..nimg-1, enqueued on fe's stream, and return.  A following extraction with imgs_on_device == VSLAM_IMGS_STAGED uses
 * them.  Splitting the upload from the extraction lets a pipelined caller start the PCIe transfer of the next pass
 * before the GPU-side dependencies of the extraction itself (e.g. another context's matcher still reading this
 * context's previous RESULTS -- the upload only overwrites level 0, which nothing outside the context reads). */
int vslam_fe_stage_images_async(vslam_fe* fe, int nimg, const uint8_t* const* imgs, size_t pitch, int imgs_where);

/* The same in two halves, so a caller can keep several contexts (streams) in flight: _async enqueues the
 * whole pass and returns without waiting for the GPU (with the device quadtree nothing in it touches the
 * host); _wait blocks until that pass is done and delivers the results like vslam_fe_extract_batch.
 * want_host = 1 also enqueues the D2H of keypoints and descriptors (0: they stay in HBM, only the counts travel).
 * want_host = 2 (full batches): the delivery is DEFERRED to the device SearchForInitialization that follows on this
 * context (vslam_search_init_dev_async with at most max_batch pairs): counts, keypoints, descriptors and the matcher's
 * outputs lie in one block and leave in ONE transfer behind the matcher; if no such call follows, _wait delivers.
 * Device images: lifetime as above; pinned host images must stay untouched until _wait returns. */
int vslam_fe_extract_batch_async(vslam_fe* fe, int nimg, const uint8_t* const* imgs, size_t pitch,
                                 int imgs_on_device, int lap0, int lap1, int want_host);
int vslam_fe_extract_wait(vslam_fe* fe, vslam_kp* const* kps, uint8_t* const* desc, int cap, int* n,
                          int* mono_index);

/* FExtractor::mvImagePyramid[level] (fextractor.h:64; read by frame.cpp:830,920,932,937).  Copies the
 * borderless level image of a slot into dst (dst_pitch >= level width).  blurred != 0 returns the
 * GaussianBlur'ed clone used for the descriptors (fextractor.cpp:1085-1086). */
int vslam_fe_level_size(const vslam_fe* fe, int level, int* w, int* h);

/* Keypoint capacity of one image slot = the most keypoints one extraction can return: nfeatures + 4*nlevels + 8
 * rounded up to a multiple of 4, or -- for very small nfeatures -- the exact bound of the quadtree
 * (sum over levels of max(quota + 3, 4 * initial nodes)) if that is larger.  Size caller arrays with it. */
int vslam_fe_capacity(const vslam_fe* fe);
int vslam_fe_level_copy(vslam_fe* fe, int slot, int level, int blurred, uint8_t* dst, size_t dst_pitch);

/* Stage taps for parity tests and profilers: per-level FAST candidates of the last extract of a slot in
 * the reference's vToDistributeKeys order (fextractor.cpp:769-817): x,y relative to the 16-px border,
 * response = FAST score.  Returns the count (or a negative error); writes at most cap entries. */
int vslam_fe_candidates(vslam_fe* fe, int slot, int level, vslam_kp* out, int cap);

/* Device-resident outputs of the last extract of a slot (valid until the next extract on this context):
 * keypoints (vslam_kp[n]) and descriptors (n*32 bytes) in HBM, for device-side matching / RCCL. */
int vslam_fe_slot_buffers(vslam_fe* fe, int slot, const vslam_kp** dev_kps, const uint8_t** dev_desc,
                          int* n);

/* Host-side views of the same results: the context's pinned staging that the result kernel fills
 * (vslam_kp[cap] / cap*32 bytes per slot).  After vslam_fe_extract_wait / vslam_frame_stereo_wait with
 * kps == desc == NULL (n != NULL) the first n[slot] entries are valid until the next enqueue on this context --
 * a consumer that reads them in place saves the copy into its own arrays. */
int vslam_fe_slot_host_views(vslam_fe* fe, int slot, const vslam_kp** host_kps, const uint8_t** host_desc);

/* GPU-side ordering between two contexts of one device: work enqueued on `waiter` after this call runs
 * after everything enqueued on `signal` so far (event record + stream wait, no host synchronisation). */
int vslam_fe_wait_for(vslam_fe* waiter, vslam_fe* signal);
/* Finer grain: four user events per context.  _record marks the current end of fe's stream; _wait makes the
 * waiter's stream wait for the last recorded instance of the signal context's event idx (0..3). */
int vslam_fe_event_record(vslam_fe* fe, int idx);
int vslam_fe_event_wait(vslam_fe* waiter, vslam_fe* signal, int idx);

/* Pacing between contexts kept in flight: with a gate set, the FAST launch of every later pass of `fe` starts only when the
 * FAST launch of `signal`'s latest pass has finished (GPU-side event; no host synchronisation).  FAST takes a CU's whole
 * LDS while it runs; chaining the contexts' FAST launches keeps two of them from being resident at once.  NULL removes it.
 * Destroying `signal` removes the gate (later passes of `fe` run un-gated); do not destroy it from another thread while a
 * pass of `fe` is being enqueued.  Passes of gated contexts are not replayed from captured graphs. */
int vslam_fe_set_fast_gate(vslam_fe* fe, vslam_fe* signal);

/* Stream the context launches on (hipStream_t as void*), for event timing by the caller. */
void* vslam_fe_stream(vslam_fe* fe);

/* Pack the results of slots 0..nslots-1 into caller device memory (e.g. this rank's send buffer of an
 * RCCL exchange): per slot `slot_bytes` >= 16 + cap*60 laid out as
 *   int32 n, mono_index, cap, 0 | vslam_kp[cap] | uint8 desc[cap][32]      (cap = vslam_fe_capacity(fe))
 * Returns after the copies have completed. */
int vslam_fe_pack_slots(vslam_fe* fe, int nslots, void* dev_dst, size_t slot_bytes);
/* the same for slots first .. first+nslots-1 (packed from offset 0 of dev_dst) */
int vslam_fe_pack_slot_range(vslam_fe* fe, int first, int nslots, void* dev_dst, size_t slot_bytes);
/* enqueue-only variant: the copies run on fe's stream after the pass that produced the slots; the caller
 * orders later readers (stream sync / event) itself */
int vslam_fe_pack_slot_range_async(vslam_fe* fe, int first, int nslots, void* dev_dst, size_t slot_bytes);

/* ---------------------------------------------------------------- multi-GPU exchange step (RCCL over xGMI)
 *
 * The reference is single-GPU/CPU; BASELINE.json's north_star shards frames one per GPU and names
 * ONE exchange: the predecessor frame's descriptors + keypoint geometry for the cross-frame matchers
 * (FMatcher::SearchForInitialization fmatcher.cpp:983-1098, SearchByProjection(Current, Last) :2471-2687).
 * One process per GPU; rank 0 makes an id (vslam_comm_unique_id), the launcher distributes its 128 bytes (any
 * channel: torch.distributed store, MPI, a file), every rank calls vslam_comm_create.  The exchange functions only
 * ENQUEUE on fe's stream (behind vslam_fe_pack_slot_range_async, ahead of the matcher): no host synchronisation.
 * RCCL is loaded at run time (librccl.so.1); without it these return VSLAM_ERR_COMM and everything else works. */
#define VSLAM_COMM_ID_BYTES 128
typedef struct vslam_comm vslam_comm;
int vslam_comm_unique_id(uint8_t id[VSLAM_COMM_ID_BYTES]);
int vslam_comm_create(int device, int rank, int world, const uint8_t id[VSLAM_COMM_ID_BYTES], vslam_comm** out);
void vslam_comm_destroy(vslam_comm* comm);
int vslam_comm_rank(const vslam_comm* comm);
int vslam_comm_world(const vslam_comm* comm);
/* Ring shift: `bytes` of dev_send go to rank+1, dev_recv receives rank-1's block (frame g lives on rank g % world,
 * so every predecessor frame lives on the left neighbour): one ncclSend/ncclRecv pair in a group. */
int vslam_exchange_ring(vslam_fe* fe, vslam_comm* comm, const void* dev_send, void* dev_recv, size_t bytes);
/* The north_star-literal variant: all-gather, dev_recv_all = world x bytes_per_rank (rank r's block at r*bytes). */
int vslam_exchange_allgather(vslam_fe* fe, vslam_comm* comm, const void* dev_send, void* dev_recv_all,
                             size_t bytes_per_rank);

/* Pinned host memory for VSLAM_IMGS_PINNED inputs (hipHostMalloc: the GPU reads it over PCIe without a staging copy). */
int vslam_host_alloc(size_t bytes, void** out);
void vslam_host_free(void* p);

/* Stage timing with HIP events on the context's stream (the reference's REGISTER_TIMES spans,
 * frame.cpp:103-132, broken down per kernel stage): stage_ms[0..4] = pyramid (7 launches), FAST cells,
 * Gaussian blur, orientation+descriptor, quadtree distribution + output order (0 with the host quadtree),
 * accumulated over `batches` batched calls / `images` images since profiling was switched on. */
int vslam_fe_set_profiling(vslam_fe* fe, int on);
int vslam_fe_get_profile(vslam_fe* fe, double stage_ms[5], long* batches, long* images);

/* ---------------------------------------------------------------- matcher (FMatcher / Frame) */

/* FMatcher::DescriptorDistance (fmatcher.h:77, fmatcher.cpp:2859-2875) over device arrays: all-pairs
 * 256-bit Hamming, two nearest train descriptors per query (the cv::BFMatcher::knnMatch(…,2) sites,
 * frame.cpp:1167-1174; fmatcher.cpp:204-229).  q/t: nq*32 / nt*32 bytes in HBM.  idx2/dist2: host arrays
 * of nq*2 (first = best; ties -> lower train index; missing -> idx -1, dist 2^31-1).
 * Runs on fe's stream and device. */
int vslam_hamming_top2(vslam_fe* fe, const uint8_t* dev_q, int nq, const uint8_t* dev_t, int nt,
                       int32_t* idx2, int32_t* dist2);

/* The same for nprob (<= 32) INDEPENDENT problems in one launch -- e.g. the brute-force matches of all stereo pairs of a
 * step (frame.cpp:1167-1174 once per frame): problem p matches dev_q[p] (nq[p] x 32 bytes in HBM) against dev_t[p]
 * (nt[p] <= 65535).  idx2[p] / dist2[p]: host arrays of nq[p] * 2 entries, semantics as above. */
int vslam_hamming_top2_batch(vslam_fe* fe, int nprob, const uint8_t* const* dev_q, const int32_t* nq,
                             const uint8_t* const* dev_t, const int32_t* nt, int32_t* const* idx2, int32_t* const* dist2);
/* Enqueue-only form for pipelines and profilers: the same launch on fe's stream, nothing waited for or copied; the results
 * stay in the context's device arrays (rows of all problems back to back).  Sizes as in a previous vslam_hamming_top2_batch
 * on this context (which allocates); VSLAM_ERR_INVALID otherwise. */
int vslam_hamming_top2_batch_dev_async(vslam_fe* fe, int nprob, const uint8_t* const* dev_q, const int32_t* nq,
                                       const uint8_t* const* dev_t, const int32_t* nt);

/* Frame::ComputeStereoFishEyeMatches (frame.cpp:1149-1174), descriptor half: cv::BFMatcher(NORM_HAMMING).knnMatch of the
 * lapping-area descriptors -- left rows [mono_left, n_left) against right rows [mono_right, n_right), k = 2 -- and the
 * ratio test `m[0].distance < m[1].distance * 0.7` (a float times a double literal: compared in double).
 * left_to_right[i] (n_left entries, host): index of the right keypoint (in the FULL right list, i.e. + mono_right) for every
 * left keypoint that passes, else -1; best_dist / second_dist (nullable): the two Hamming distances; *n_candidates = the
 * reference's descMatches.  The triangulation that decides mvLeftToRightMatch / mvDepth from these pairs
 * (KannalaBrandt8::TriangulateMatches, :1176-1188) belongs to the camera model and stays with the caller. */
int vslam_stereo_fisheye_candidates(vslam_fe* fe, const uint8_t* dev_desc_left, int n_left, int mono_left,
                                    const uint8_t* dev_desc_right, int n_right, int mono_right, int32_t* left_to_right,
                                    int32_t* best_dist, int32_t* second_dist, int* n_candidates);

/* Dense distance matrix (nq x nt, uint8, 255 = distance >= 255) on the device -> host; used by the
 * order-dependent matchers whose sequential part is replayed on the host. */
int vslam_hamming_matrix(vslam_fe* fe, const uint8_t* dev_q, int nq, const uint8_t* dev_t, int nt,
                         uint8_t* out);

/* Frame::ComputeStereoMatches (frame.h:293, frame.cpp:823-997) for the pair (feL slot sL, feR slot sR)
 * after both have been extracted.  bf = Camera.bf, fx = Camera.fx.  u_right/depth: host arrays of the
 * left keypoint count (mvuRight, mvDepth; -1 = no match).  feL and feR may be the same context. */
int vslam_stereo_match(vslam_fe* feL, int sL, vslam_fe* feR, int sR, float bf, float fx, float* u_right,
                       float* depth);

/* The same for npairs (<= 32) stereo pairs in one pass of the kernels: pair j = (feL slot slotsL[j],
 * feR slot slotsR[j]); u_right[j]/depth[j] are host arrays of that pair's left keypoint count. */
int vslam_stereo_match_batch(vslam_fe* feL, vslam_fe* feR, int npairs, const int* slotsL, const int* slotsR,
                             float bf, float fx, float* const* u_right, float* const* depth);

/* The extraction + stereo-match section of Frame::Frame(imLeft, imRight, ...) (frame.cpp:102-132) for
 * npairs (<= 32, 2*npairs <= max_batch) stereo frames in ONE enqueue on fe's stream: imgs = L0,R0,L1,R1,...
 * (slot 2j = left, 2j+1 = right of frame j; vLappingArea {0,0} as the reference passes, frame.cpp:107-108).
 * _async returns without waiting; _wait delivers keypoints/descriptors of all 2*npairs images (arrays of
 * 2*npairs entries, may be NULL) and mvuRight/mvDepth of the npairs left images. */
int vslam_frame_stereo_batch_async(vslam_fe* fe, int npairs, const uint8_t* const* imgs, size_t pitch,
                                   int imgs_on_device, float bf, float fx, int want_host);
int vslam_frame_stereo_wait(vslam_fe* fe, vslam_kp* const* kps, uint8_t* const* desc, int cap, int* n,
                            float* const* u_right, float* const* depth);

/* FMatcher::SearchForInitialization (fmatcher.h:106, fmatcher.cpp:983-1098).  Frame 1 / frame 2
 * keypoints+descriptors are device arrays (e.g. from vslam_fe_slot_buffers, or a slot of an RCCL
 * exchange buffer); kps1_host/kps2_host are the same keypoints on the host.  prev_matched: 2*n1 floats
 * in/out (vbPrevMatched).  matches12: n1 ints out.  Returns the match count in *nmatches. */
int vslam_search_for_initialization(vslam_fe* fe, const vslam_kp* kps1_host, const uint8_t* dev_desc1,
                                    int n1, const vslam_kp* kps2_host, const uint8_t* dev_desc2, int n2,
                                    int img_w, int img_h, float* prev_matched, int32_t* matches12,
                                    int window, float nnratio, int check_orientation, int* nmatches);

/* npairs (<= 32) independent SearchForInitialization problems in one pass: the dense distance matrices of
 * all pairs come from one kernel launch, the order-dependent replays run in parallel on the host pool.
 * Every per-pair argument of the single form becomes an array of npairs entries. */
int vslam_search_for_initialization_batch(vslam_fe* fe, int npairs, const vslam_kp* const* kps1_host,
                                          const uint8_t* const* dev_desc1, const int* n1,
                                          const vslam_kp* const* kps2_host, const uint8_t* const* dev_desc2,
                                          const int* n2, int img_w, int img_h, float* const* prev_matched,
                                          int32_t* const* matches12, int window, float nnratio,
                                          int check_orientation, int* nmatches);

/* Device-resident form: every pointer of a job is a DEVICE pointer -- keypoints/descriptors as returned by
 * vslam_fe_slot_buffers (or a slot of an RCCL exchange buffer packed by vslam_fe_pack_slots), the keypoint
 * counts as int32 in HBM (vslam_fe_slot_count_ptr, or the header word of a packed slot).  The whole matcher
 * (window query, Hamming distances, stealing, ratio test, rotation histogram) runs in one kernel, one wave per
 * pair, on fe's stream; _async returns without waiting, _wait delivers n1[j] entries of vnMatches12 and the
 * updated vbPrevMatched per pair plus the match counts.  dev_prev_matched == NULL means "frame 1's keypoint
 * positions" (tracking.cpp:2281-2288 initialises vbPrevMatched that way). */
typedef struct vslam_init_job {
    const vslam_kp* dev_kps1;
    const uint8_t* dev_desc1;
    const int32_t* dev_n1;
    const vslam_kp* dev_kps2;
    const uint8_t* dev_desc2;
    const int32_t* dev_n2;
    const float* dev_prev_matched;
} vslam_init_job;
int vslam_search_init_dev_async(vslam_fe* fe, int npairs, const vslam_init_job* jobs, int img_w, int img_h,
                                int window, float nnratio, int check_orientation);
int vslam_search_init_dev_wait(vslam_fe* fe, const int* n1, int32_t* const* matches12, float* const* prev_matched,
                               int* nmatches);
/* device address of a slot's keypoint count (int32), valid for the life of the context */
int vslam_fe_slot_count_ptr(vslam_fe* fe, int slot, const int32_t** dev_n);

/* ---------------------------------------------------------------- diagnostics */

/* FMatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono)
 * (fmatcher.h, fmatcher.cpp:2471-2687; pinhole frames, Nleft == -1) -- the tracking matcher of
 * TrackWithMotionModel (tracking.cpp:2728).  What the function reads of the two frames is passed explicitly:
 *   p            pose of CurrentFrame as rows [Rcw | tcw] (the reference's T_w_c_ member is used that way),
 *                pinhole intrinsics, mbf, th, bForward/bBackward (vslam_projection_direction), image bounds of
 *                CurrentFrame's grid (mnMinX = mnMinY = 0, mnMaxX = img_w, mnMaxY = img_h)
 *   last frame   host arrays of n_last entries: keypoints (octave from keypoints_, angle from ukeypoints_),
 *                flags (bit0: mvpMapPoints[i] != NULL && !mvbOutlier[i]; bit1: that MapPoint has
 *                Observations() > 0), world positions (3 floats), MapPoint descriptors (32 bytes)
 *   current      DEVICE keypoints/descriptors (vslam_fe_slot_buffers), n_cur <= 4096; host mvuRight (NULL =
 *                monocular, all -1) and the initial "mvpMapPoints[i2] with Observations() > 0" mask (NULL = none:
 *                tracking.cpp fills mvpMapPoints with NULL before the call)
 * Output: match_cur[i2] = index i of the last-frame keypoint whose MapPoint ends up in mvpMapPoints[i2], or -1;
 * *nmatches as the reference counts it. */
typedef struct vslam_proj_params {
    float Tcw[12];
    float fx, fy, cx, cy, mbf, th;
    int32_t forward, backward;
    int32_t check_orientation;
    int32_t img_w, img_h;
    int32_t gemm_float; /* 0: Rcw*x+tcw as cv::gemm evaluates it (double accumulation); 1: float arithmetic */
} vslam_proj_params;
int vslam_projection_direction(const float* Tcw, const float* Tlw, float mb, int mono, int gemm_float,
                               int* forward, int* backward);
int vslam_search_by_projection_frame(vslam_fe* fe, const vslam_proj_params* p, const vslam_kp* last_kps_host,
                                     int n_last, const uint8_t* last_flags, const float* last_x3dw,
                                     const uint8_t* mp_desc_host, const vslam_kp* dev_cur_kps,
                                     const uint8_t* dev_cur_desc, int n_cur, const float* cur_u_right_host,
                                     const uint8_t* cur_occupied_host, int32_t* match_cur, int* nmatches);

/* FMatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const set<MapPoint*>& sAlreadyFound, th, ORBdist)
 * (fmatcher.cpp:2689-2811) -- the matcher of Tracking::Relocalization.  p carries T_w_c_'s Rcw | tcw in Tcw, the camera,
 * th, check_orientation, the image bounds and gemm_float (forward / backward / mbf are not used); Ow = -Rcw^T tcw as the
 * function computes it (:2695); log_scale_factor = CurrentFrame.mfLogScaleFactor; orb_dist 1..255.  Per KeyFrame
 * keypoint i: kf_kps_host = pKF->mvKeysUn, mp_flags[i] & 1 iff vpMPs[i] && !isBad() && !sAlreadyFound.count(it),
 * world position, Get{Min,Max}DistanceInvariance(), descriptor.  cur_occupied_host marks CurrentFrame.mvpMapPoints
 * entries that are not NULL.  match_cur[i2] = i (pKF's keypoint whose MapPoint is written to mvpMapPoints[i2]) or -1. */
int vslam_search_by_projection_keyframe(vslam_fe* fe, const vslam_proj_params* p, const float* Ow, float log_scale_factor,
                                        int orb_dist, const vslam_kp* kf_kps_host, int n_kf, const uint8_t* mp_flags,
                                        const float* mp_x3dw, const float* mp_min_dist, const float* mp_max_dist,
                                        const uint8_t* mp_desc_host, const vslam_kp* dev_cur_kps,
                                        const uint8_t* dev_cur_desc, int n_cur, const uint8_t* cur_occupied_host,
                                        int32_t* match_cur, int* nmatches);

/* FMatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const vector<MapPoint*>& vpPoints, vector<MapPoint*>& vpMatched,
 * int th, float ratioHamming) (fmatcher.cpp:750-863; proj_variant 0) and the overload that also fills vpMatchedKF from
 * vpPointsKFs (:865-981; it projects as fx*(x*(1/z))+cx, proj_variant 1) -- the loop-closing matchers.  p carries
 * Rcw | tcw (= sRcw/scw, Scw's translation/scw, :760-763) in Tcw, the camera, th, the image bounds and gemm_float;
 * Ow = -Rcw^T tcw; log_scale_factor = pKF->mfLogScaleFactor.  Per candidate MapPoint: flag (!isBad() &&
 * !spAlreadyFound.count(pMP)), world position, normal, Get{Min,Max}DistanceInvariance(), descriptor.
 * kf_matched_host[idx] != 0 iff vpMatched[idx] != NULL on entry.  match_kf[idx] = iMP (vpMatched[idx] = vpPoints[iMP],
 * vpMatchedKF[idx] = vpPointsKFs[iMP]) or -1 for the keypoints this call assigns; *nmatches as returned. */
int vslam_search_by_projection_sim3(vslam_fe* fe, const vslam_proj_params* p, const float* Ow, float log_scale_factor,
                                    float ratio_hamming, int proj_variant, const uint8_t* mp_flags, const float* mp_x3dw,
                                    const float* mp_normals, const float* mp_min_dist, const float* mp_max_dist,
                                    const uint8_t* mp_desc_host, int n_points, const vslam_kp* dev_kf_kps,
                                    const uint8_t* dev_kf_desc, int n_kf, const uint8_t* kf_matched_host,
                                    int32_t* match_kf, int* nmatches);

/* FMatcher::SearchByProjection(Frame& F, const vector<MapPoint*>& vpMapPoints, th, bFarPoints, thFarPoints)
 * (fmatcher.cpp:321-411; pinhole frames) -- the local-map matcher of Tracking::SearchLocalPoints.  Each MapPoint
 * arrives with what Frame::isInFrustum left in it (mappoint.h:73-81):
 *   flags bit0 = mbTrackInView && !isBad() && !(bFarPoints && mTrackDepth > thFarPoints), bit1 = Observations() > 0.
 * cur_occupied_host marks keypoints whose mvpMapPoints entry already holds a MapPoint with observations (the
 * matches of TrackWithMotionModel).  match_cur[idx] = index of the MapPoint written to F.mvpMapPoints[idx], or -1.
 * nnratio is FMatcher::mfNNratio (>= 0.4 on the device). */
typedef struct vslam_mp_track {
    float proj_x, proj_y, proj_xr, view_cos; /* mTrackProjX, mTrackProjY, mTrackProjXR, mTrackViewCos */
    int32_t level;                           /* mnTrackScaleLevel */
    uint32_t flags;
} vslam_mp_track;
int vslam_search_by_projection_mappoints(vslam_fe* fe, const vslam_mp_track* mps_host, const uint8_t* mp_desc_host,
                                         int n_mp, const vslam_kp* dev_cur_kps, const uint8_t* dev_cur_desc,
                                         int n_cur, const float* cur_u_right_host, const uint8_t* cur_occupied_host,
                                         int img_w, int img_h, float th, float nnratio, int32_t* match_cur,
                                         int* nmatches);

/* Device-resident, batched form of the same matcher: every pointer of a job is a DEVICE pointer; counts are
 * int32 in HBM (vslam_fe_slot_count_ptr); keypoint arrays hold up to the context's capacity (<= 4096).  Up to 16
 * jobs per call run in one pass of the two kernels on fe's stream.  _async returns without waiting; _wait
 * delivers n_cur[j] entries of match_cur[j] and nmatches[j]. */
typedef struct vslam_sbp_job {
    vslam_proj_params p;
    const vslam_kp* dev_last_kps;
    const int32_t* dev_n_last;
    const uint8_t* dev_last_flags;
    const float* dev_last_x3dw;
    const uint8_t* dev_mp_desc;
    const vslam_kp* dev_cur_kps;
    const uint8_t* dev_cur_desc;
    const int32_t* dev_n_cur;
    const float* dev_cur_u_right;    /* may be NULL */
    const uint8_t* dev_cur_occupied; /* may be NULL */
} vslam_sbp_job;
int vslam_search_by_projection_dev_async(vslam_fe* fe, int njobs, const vslam_sbp_job* jobs);
int vslam_search_by_projection_dev_wait(vslam_fe* fe, const int* n_cur, int32_t* const* match_cur, int* nmatches);

/* Frame::UnprojectStereo (frame.cpp:1023-1037) on the device for all left keypoints of the stereo pairs of fe's
 * last vslam_frame_stereo_batch_async / vslam_stereo_match_batch: the stereo points that UpdateLastFrame
 * (tracking.cpp) turns into MapPoints for the next frame's SearchByProjection.  Twc: npairs x 12 floats, rows
 * [mRwc | mOw] per pair.  flags[i] = 0 (no depth) or 1 | (observations ? 2 : 0).  Enqueued on fe's stream;
 * vslam_stereo_points_buffers returns the per-pair device arrays (stable for the life of the context). */
int vslam_stereo_points_dev_async(vslam_fe* fe, int npairs, const float* Twc, float cx, float cy, float invfx,
                                  float invfy, int observations, int gemm_float);
int vslam_stereo_points_buffers(vslam_fe* fe, int pair, const float** dev_x3dw, const uint8_t** dev_flags,
                                const float** dev_u_right, const float** dev_depth);

/* MapPoint::ComputeDistinctiveDescriptors (mappoint.cpp:322-390) for nsets MapPoints in one pass: set s owns the
 * descriptors desc_host[offsets[s] .. offsets[s+1]) (32 bytes each, its observations in the reference's iteration
 * order); best[s] = index inside the set of the descriptor with the least median distance to the others
 * (median = sorted[int(0.5*(N-1))], first wins), -1 for an empty set.  At most 2048 descriptors per set. */
int vslam_distinctive_descriptors(vslam_fe* fe, const uint8_t* desc_host, const int32_t* offsets, int nsets,
                                  int32_t* best);

/* Frame::ComputeBoW (frame.cpp:455-461) = DBoW3::Vocabulary::transform(features, BowVector&, FeatureVector&,
 * levelsup) (thirdparty/DBoW3/DBoW3/src/Vocabulary.cpp:754-878).
 * vslam_voc_create uploads a flat copy of DBoW3's m_nodes: node i has the children child_ids[child_start[i] ..
 * + child_count[i]) in their stored order (it decides ties), a 32-byte descriptor, and for leaves a word id and a
 * weight; depth_levels = m_L; weighting: 0 TF_IDF, 1 TF, 2 IDF, 3 BINARY; norm (the scoring object's
 * mustNormalize): 0 none, 1 L1, 2 L2.  Node ids must be larger than their parent's (DBoW3's creation order).
 * vslam_bow_transform walks the tree for n device-resident descriptors and returns per feature the word id, the
 * word weight and the node id at level depth_levels - levelsup; the _slots forms do the same for image slots of the
 * context (counts stay in HBM) in one launch.  vslam_bow_assemble (pure host) turns those into the BowVector
 * (ascending word ids / values) and the FeatureVector (ascending node ids, fv_off[n_fv+1] offsets into fv_feat)
 * exactly as DBoW3 does (std::map accumulation in feature order, double arithmetic); arrays hold n entries
 * (fv_off: n + 1). */
typedef struct vslam_voc vslam_voc;
int vslam_voc_create(int device, int depth_levels, int weighting, int norm, int n_nodes, const int32_t* child_start,
                     const int32_t* child_count, const int32_t* child_ids, int n_child_ids, const uint8_t* node_desc,
                     const double* node_weight, const int32_t* node_word_id, vslam_voc** out);
void vslam_voc_destroy(vslam_voc* voc);
/* DBoW3::Vocabulary::load(const std::string&) (thirdparty/DBoW3/DBoW3/src/Vocabulary.cpp:1084-1112), the call
 * core::System makes at start-up (src/core/system.cpp:76).  vslam_voc_load reads the file and uploads it; the
 * vslam_voc_file_* functions are its host half (no GPU needed): open parses, info / arrays expose the flat node table
 * in vslam_voc_create's layout (pointers stay valid until close).  Formats, tried in the reference's order: DBoW3's
 * binary stream (magic 88877711233; format 1 plain, 2 in QuickLZ level-1 chunks -- what Vocabulary::save writes by
 * default under any file name), then the text form for names containing ".txt" (format 3; weights keep float
 * precision as in load_fromtxt, :1372-1446).  cv::FileStorage YAML/XML vocabularies and descriptors other than
 * 1 x 32 CV_8U return VSLAM_ERR_UNSUPPORTED; a damaged file returns VSLAM_ERR_INVALID (every read is bounds-checked).
 * scoring is DBoW3's ScoringType (0 L1_NORM .. 5 DOT_PRODUCT); norm is what its scoring object's mustNormalize says,
 * in vslam_voc_create's encoding.  vslam_voc_file_last_error: message of the last failed vslam_voc_file_* call of
 * this thread (vslam_voc_load copies it into vslam_last_error). */
typedef struct vslam_voc_file vslam_voc_file;
int vslam_voc_file_open(const char* path, vslam_voc_file** out);
void vslam_voc_file_close(vslam_voc_file* f);
int vslam_voc_file_info(const vslam_voc_file* f, int* branching, int* depth_levels, int* scoring, int* weighting,
                        int* norm, int* n_nodes, int* n_words, int* n_child_ids, int* format);
int vslam_voc_file_arrays(const vslam_voc_file* f, const int32_t** child_start, const int32_t** child_count,
                          const int32_t** child_ids, const uint8_t** node_desc, const double** node_weight,
                          const int32_t** node_word_id);
const char* vslam_voc_file_last_error(void);
int vslam_voc_load(int device, const char* path, vslam_voc** out);
/* test hook of the reader: decode one QuickLZ 1.5 level-1 packet (header + payload) of at most n bytes into dst;
 * returns the decoded size and the packet's size in *used, -1 for a damaged packet or a short dst. */
long vslam_dbg_qlz_decode(const uint8_t* packet, size_t n, uint8_t* dst, size_t cap, size_t* used);
int vslam_voc_info(const vslam_voc* voc, int* depth_levels, int* weighting, int* norm, int* n_nodes);
int vslam_bow_transform(vslam_fe* fe, const vslam_voc* voc, const uint8_t* dev_desc, int n, int levelsup,
                        int32_t* word_id, double* weight, int32_t* node_id);
int vslam_bow_transform_slots_async(vslam_fe* fe, const vslam_voc* voc, int first_slot, int nslots, int levelsup);
int vslam_bow_transform_slots_wait(vslam_fe* fe, const int* n, int32_t* const* word_id, double* const* weight,
                                   int32_t* const* node_id);
int vslam_bow_assemble(int weighting, int norm, const int32_t* word_id, const double* weight, const int32_t* node_id,
                       int n, int32_t* bow_ids, double* bow_vals, int* n_bow, int32_t* fv_nodes, int32_t* fv_off,
                       int32_t* fv_feat, int* n_fv);

/* FMatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vector<MapPoint*>& vpMapPointMatches) (fmatcher.cpp:546-748,
 * pinhole frames).  FeatureVectors as produced by vslam_bow_assemble (ascending node ids, offsets, feature
 * indices); kf_flags[i] != 0 iff pKF's i-th keypoint has a MapPoint that is not bad; keypoints only contribute
 * their angles; descriptors are device arrays.  match_f[iF] = KeyFrame feature index whose MapPoint is written to
 * vpMapPointMatches[iF], -1 = NULL; *nmatches as the reference counts it. */
int vslam_search_by_bow(vslam_fe* fe, const vslam_kp* kf_kps_host, const uint8_t* dev_kf_desc,
                        const uint8_t* kf_flags_host, int n_kf, const int32_t* kf_fv_nodes, const int32_t* kf_fv_off,
                        const int32_t* kf_fv_feat, int n_kf_nodes, const vslam_kp* f_kps_host,
                        const uint8_t* dev_f_desc, int n_f, const int32_t* f_fv_nodes, const int32_t* f_fv_off,
                        const int32_t* f_fv_feat, int n_f_nodes, float nnratio, int check_orientation,
                        int32_t* match_f, int* nmatches);

/* FMatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12) (fmatcher.cpp:1100-1240):
 * both sides carry MapPoint flags (exists and not bad), the threshold is bestDist1 < TH_LOW, and the result is
 * indexed by the first KeyFrame: match12[idx1] = idx2 (vpMatches12[idx1] = vpMapPoints2[idx2]) or -1. */
int vslam_search_by_bow_keyframes(vslam_fe* fe, const vslam_kp* kps1_host, const uint8_t* dev_desc1,
                                  const uint8_t* flags1_host, int n1, const int32_t* fv1_nodes, const int32_t* fv1_off,
                                  const int32_t* fv1_feat, int n1_nodes, const vslam_kp* kps2_host,
                                  const uint8_t* dev_desc2, const uint8_t* flags2_host, int n2,
                                  const int32_t* fv2_nodes, const int32_t* fv2_off, const int32_t* fv2_feat,
                                  int n2_nodes, float nnratio, int check_orientation, int32_t* match12, int* nmatches);

/* FMatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo, bCoarse) (fmatcher.cpp:1242-1482)
 * and its cv::Matx twin SearchForTriangulation_ (:1484-1725, the one LocalMapping::CreateNewMapPoints calls,
 * localmapping.cpp:447) for pinhole KeyFrames without a second camera.  F12 is the matrix
 * Pinhole::epipolarConstrain builds for the pair, K1^-T [t12]x R12 K2^-1 (pinhole.cpp:121-127, row-major; it does
 * not depend on the keypoints, so the caller computes it once with its own cv::Mat algebra); (ep_x, ep_y) is the
 * epipole pKF2->mpCamera->project(R2w*Cw + t2w) (fmatcher.cpp:1249-1254).  kps*_host are mvKeysUn, has_mp*[i] != 0
 * iff GetMapPoint(i) != NULL, u_right* are mvuRight; FeatureVectors as produced by vslam_bow_assemble; the scale
 * tables of pKF2 are the context's (vslam_fe_tables).  match12[idx1] = idx2 or -1 -- vMatchedPairs is the list of
 * (idx1, match12[idx1]) with match12[idx1] >= 0 in ascending idx1; the reference never marks KeyFrame-2 features
 * as taken, so an idx2 may appear more than once.  *nmatches as the reference returns it. */
typedef struct vslam_tri_params {
    float F12[9];
    float ep_x, ep_y;
    int32_t only_stereo, coarse, check_orientation;
} vslam_tri_params;
int vslam_search_for_triangulation(vslam_fe* fe, const vslam_tri_params* p, const vslam_kp* kps1_host,
                                   const uint8_t* dev_desc1, const uint8_t* has_mp1_host, const float* u_right1_host,
                                   int n1, const int32_t* fv1_nodes, const int32_t* fv1_off, const int32_t* fv1_feat,
                                   int n1_nodes, const vslam_kp* kps2_host, const uint8_t* dev_desc2,
                                   const uint8_t* has_mp2_host, const float* u_right2_host, int n2,
                                   const int32_t* fv2_nodes, const int32_t* fv2_off, const int32_t* fv2_feat,
                                   int n2_nodes, int32_t* match12, int* nmatches);

/* The search half of FMatcher::Fuse(pKF, vpMapPoints, th, bRight = false) (fmatcher.cpp:1918-2119) and of
 * Fuse(pKF, Scw, vpPoints, th, vpReplacePoint) (:2121-2243; sim3 = 1, Rcw/tcw/Ow as the function derives them from
 * Scw at :2130-2134): for every MapPoint the projection gates, MapPoint::PredictScale, the window of
 * KeyFrame::GetFeaturesInArea and the keypoint with the least descriptor distance (first of the window order wins).
 * best_idx[i] = -1 (best_dist[i] = 256) when a gate rejects the point or no keypoint passes; distances above 255 are
 * reported as 255.  The caller walks the results in order and does what the reference does when
 * best_dist[i] <= TH_LOW (50): Replace / AddObservation + AddMapPoint / vpReplacePoint -- re-checking isBad() and
 * IsInKeyFrame() at that moment, because those are the only inputs an earlier iteration can change.
 * valid = pMP && !pMP->isBad() && !pMP->IsInKeyFrame(pKF) when the call is made. */
typedef struct vslam_fuse_point {
    float pos[3];       /* GetWorldPos() */
    float normal[3];    /* GetNormal() */
    float min_distance; /* GetMinDistanceInvariance() */
    float max_distance; /* GetMaxDistanceInvariance() */
    int32_t valid;
} vslam_fuse_point;
typedef struct vslam_fuse_params {
    float Rcw[9], tcw[3], Ow[3];
    float fx, fy, cx, cy, bf, th;
    float log_scale_factor; /* pKF->mfLogScaleFactor */
    int32_t img_w, img_h;   /* mnMaxX, mnMaxY (mnMinX = mnMinY = 0: undistorted pinhole images) */
    int32_t sim3;           /* 0 / 1: the two Fuse overloads; 2: one direction of SearchBySim3, see below */
    int32_t gemm_float;     /* as in vslam_proj_params */
    float Rb[9], tb[3];     /* sim3 == 2 only */
} vslam_fuse_params;
/* FMatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th) (fmatcher.cpp:2245-2469) is two such searches
 * (sim3 = 2) and an agreement check.  Direction 1 -> 2: Rcw | tcw = R1w | t1w, Rb | tb = sR21 | t21 (:2262-2264),
 * the points are pKF1's MapPoints that are not bad and not matched yet, the KeyFrame is pKF2; direction 2 -> 1 with
 * R2w | t2w and sR12 | t12.  A point is p2 = Rb*(Rcw*p + tcw) + tb, depth >= 0, u = fx*(x*(1.0/z)) + cx, IsInImage,
 * |p2| within the scale-invariance range, PredictScale, window, levels [l-1, l], least distance (first wins); Ow,
 * normals, bf are unused.  The caller keeps vnMatch1[i1] = best_idx where best_dist <= TH_HIGH (100), likewise
 * vnMatch2, and accepts (i1, idx2) iff vnMatch2[idx2] == i1 (:2453-2466). */
int vslam_fuse_search(vslam_fe* fe, const vslam_fuse_params* p, const vslam_fuse_point* points_host,
                      const uint8_t* mp_desc_host, int n_points, const vslam_kp* dev_kf_kps,
                      const uint8_t* dev_kf_desc, int n_kf, const float* kf_u_right_host, int32_t* best_idx,
                      int32_t* best_dist);

/* Evaluate the device float helpers on host arrays (round trip through HBM): the glibc-exact sinf/cosf
 * used for the rBRIEF rotation (fextractor.cpp:103-104) and cv::fastAtan2 (fextractor.cpp:94). */
int vslam_dbg_sincos(vslam_fe* fe, const float* x, int n, float* sin_out, float* cos_out);
int vslam_dbg_fast_atan2(vslam_fe* fe, const float* y, const float* x, int n, int fma, float* deg);
/* glibc logf as MapPoint::PredictScale uses it (mappoint.cpp:514); normal positive inputs, NaN otherwise */
int vslam_dbg_logf(vslam_fe* fe, const float* x, int n, float* y);
/* Quadtree statistics of a context: how many (slot, level) DistributeOctTree problems (fextractor.cpp:530-754) ran on the
 * device so far, on how many of them nodes had to be split BELOW the kernel's fine grid (keys closer together than a fine
 * cell -- real images do that on sparse levels; resolved inside the kernel, same result, a little slower), and --
 * last_level_masks, max_batch words or NULL -- bit l of word s set if that happened on level l of slot s in the last pass.
 * Updated when a pass's results are collected. */
int vslam_fe_octree_stats(const vslam_fe* fe, unsigned long long* problems, unsigned long long* split_below_grid,
                          uint32_t* last_level_masks);
/* Result deliveries of a context so far: how many copy operations (a runtime copy or one launch of the copy kernel) the
 * extraction and SearchForInitialization paths put on the stream towards the host, and the bytes they carried.  A full
 * batch with want_host = 1 is one operation, the matcher's outputs another; with want_host = 2 the step is ONE. */
int vslam_fe_delivery_stats(const vslam_fe* fe, unsigned long long* transfers, unsigned long long* bytes);
/* In-kernel time stamps of the quadtree kernel (100 MHz ticks; out64[63] = count).  Only a library built with
 * -DVSLAM_OCT_STAMPS and a context created under VSLAM_OCT_DBG=1 records them; otherwise VSLAM_ERR_INVALID. */
int vslam_dbg_octree_stamps(vslam_fe* fe, unsigned long long* out64);
/* Number of queries for which the device SearchForInitialization had to re-scan the whole window because the
 * sorted candidate prefix (length VSLAM_INIT_TOPM, default 8) was exhausted; read-and-reset, synchronises. */
int vslam_dbg_search_init_fallbacks(vslam_fe* fe, int* count);
/* Rounds of the replay wave, queries and pairs it decided since the last call (read-and-reset, synchronises): the
 * sequential half of SearchForInitialization commits `queries / rounds` queries per round on average. */
int vslam_dbg_search_init_replay_stats(vslam_fe* fe, int* rounds, int* queries, int* pairs);

#ifdef __cplusplus
}
#endif
#endif /* VSLAM_FE_H */
