/* vslam_init_kernel.hip -- FMatcher::SearchForInitialization (fmatcher.cpp:983-1098) entirely on the GPU.
 *
 * The reference loop is sequential over the octave-0 keypoints of frame 1 (a later query may steal an
 * earlier query's match, and a candidate whose current owner is at least as close is skipped).  Only that
 * skip/steal bookkeeping is order-dependent; the expensive part -- which frame-2 keypoints lie in a query's
 * window and how far their descriptors are -- is not.  Two kernels:
 *
 *   k_si_topm   (parallel, one wave per query, all pairs at once): every window candidate gets
 *               key = min(dist,255) << 24 | gridcell << 12 | slot, i.e. (distance, position in
 *               Frame::GetFeaturesInArea's output: grid cell column-major, ascending index inside a cell,
 *               frame.cpp:712-741).  The M smallest keys of each query are written in ascending order.
 *   k_si_replay (one wave per pair, queries in index order): the first two list entries that are not skipped
 *               (vMatchedDistance[i2] <= dist, fmatcher.cpp:1022) are bestDist/bestIdx2 and bestDist2 --
 *               "first wins" on equal distance is the key order, bestDist2 is the multiset second minimum.
 *               A list is a sorted prefix of all candidates, so this is exact whenever two survivors are found
 *               (or the list is not full, or the last list entry already decides the ratio test); otherwise
 *               the wave re-scans that query's window in full (rare; forced in tests with a tiny M).
 *               Acceptance (TH_LOW, ratio test in float), stealing, the 30-bin rotation histogram and
 *               ComputeThreeMaxima follow the reference literally.
 *
 * Distances are clamped to 255 inside the key (the only other value is 256): an accepted best and every
 * owner's distance are <= TH_LOW = 50, so the clamp can only matter through the ratio test against a second
 * of exactly 256, and nnratio * 255 > 50 for every nnratio >= 0.2 -- the host wrapper refuses smaller ratios.
 */
#include "vslam_kernels.h"
#include "vslam_wave.h"
#include "vslam_trig.h"

#define SI_TH_LOW 50
#define SI_HISTO 30
#define SI_GRID_COLS 64 /* FRAME_GRID_COLS, frame.h:42 */
#define SI_GRID_ROWS 48 /* FRAME_GRID_ROWS, frame.h:43 */
#define SI_QPB_MAX 32   /* queries per k_si_topm workgroup: run-time (4 waves x 2..8), see vk_search_init */
#define SI_MAX_M 16

struct SiCand { /* one octave-0 keypoint of frame 2 */
    float x, y, angle;
    uint16_t cell; /* gridX * 64 + gridY (gridY < 48) */
    uint16_t idx;  /* index in frame 2 */
};
struct SiCandL { /* the same in k_si_topm's LDS list, which is sorted by grid column: the position c in the
                   keypoint-ordered list (what the keys and k_si_replay refer to) instead of the angle */
    float x, y;
    uint32_t c;
    uint16_t cell;
    uint16_t idx;
};
struct SiQuery { /* one octave-0 keypoint of frame 1 */
    float px, py, angle; /* vbPrevMatched position, keypoint angle */
    uint32_t idx;
};

/* per-pair scratch in HBM: [c1, c2, pad, pad] | SiCand[max_c2] | SiQuery[max_c2] | topm[max_c2][M] */
__host__ __device__ inline size_t si_pair_bytes(int max_c2, int M) {
    return 16 + (size_t)max_c2 * (sizeof(SiCand) + sizeof(SiQuery) + 4 * (size_t)M);
}

struct SiWindow {
    int minX, maxX, minY, maxY;
    bool empty;
};
/* GetFeaturesInArea(x, y, r, 0, 0): cell range (frame.cpp:686-708) */
__device__ __forceinline__ SiWindow si_window(float px, float py, float r, float invW, float invH) {
    SiWindow w;
    w.minX = max(0, (int)floorf(__fmul_rn(__fsub_rn(px, r), invW)));
    w.maxX = min(SI_GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(px, r), invW)));
    w.minY = max(0, (int)floorf(__fmul_rn(__fsub_rn(py, r), invH)));
    w.maxY = min(SI_GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(py, r), invH)));
    w.empty = w.minX >= SI_GRID_COLS || w.maxX < 0 || w.minY >= SI_GRID_ROWS || w.maxY < 0;
    return w;
}
__device__ __forceinline__ bool si_in_window(const SiCand& cd, const SiWindow& w, float px, float py, float r) {
    const int gx = cd.cell >> 6, gy = cd.cell & 63;
    if (gx < w.minX || gx > w.maxX || gy < w.minY || gy > w.maxY) return false;
    const float distx = __fsub_rn(cd.x, px), disty = __fsub_rn(cd.y, py);
    return fabsf(distx) < r && fabsf(disty) < r;
}
__device__ __forceinline__ bool si_in_window(const SiCandL& cd, const SiWindow& w, float px, float py, float r) {
    const int gx = cd.cell >> 6, gy = cd.cell & 63;
    if (gx < w.minX || gx > w.maxX || gy < w.minY || gy > w.maxY) return false;
    const float distx = __fsub_rn(cd.x, px), disty = __fsub_rn(cd.y, py);
    return fabsf(distx) < r && fabsf(disty) < r;
}
__device__ __forceinline__ uint32_t si_hamming(const uint4& da, const uint4& db, const uint4& ta, const uint4& tb) {
    return __popc(da.x ^ ta.x) + __popc(da.y ^ ta.y) + __popc(da.z ^ ta.z) + __popc(da.w ^ ta.w) +
           __popc(db.x ^ tb.x) + __popc(db.y ^ tb.y) + __popc(db.z ^ tb.z) + __popc(db.w ^ tb.w);
}

/* ------------------------------------------------------------------------------------------------
 * phase A: grid (chunks of SI_QPB queries, pairs), 256 threads.  Every workgroup compacts the octave-0
 * keypoints of both frames in index order (block scan per 256), keeps frame 2's in LDS, and ranks the window
 * candidates of its own queries.  Chunk 0 also publishes the compacted candidate list and the counts.
 * ---------------------------------------------------------------------------------------------- */
__global__ void __launch_bounds__(256)
k_si_topm(InitJobs jobs, int cap, int imgW, int imgH, int window, int max_c2, int M, uint8_t* scratch, int SI_QPB, int prio) {
    extern __shared__ __align__(16) uint8_t sism[];
    wave_prio_raise(prio);
    SiCandL* cand = (SiCandL*)sism;                  /* max_c2, sorted by grid column */
    uint32_t* wkeys = (uint32_t*)(cand + max_c2);    /* 4 waves x max_c2: window keys of the current query */
    SiCandL* tmpc = (SiCandL*)wkeys;                 /* ... and, before that, the list in keypoint order */
    __shared__ SiQuery s_q[SI_QPB_MAX];
    __shared__ int s_wcnt[4];
    __shared__ int s_col[SI_GRID_COLS + 1]; /* candidates per grid column, then the columns' first positions */
    __shared__ int s_fill[SI_GRID_COLS];
    if (threadIdx.x <= SI_GRID_COLS) s_col[threadIdx.x] = 0;
    if (threadIdx.x < SI_GRID_COLS) s_fill[threadIdx.x] = 0;
    __syncthreads();
    const InitJob jb = jobs.job[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n1 = min(*jb.cnt1, cap), n2 = min(*jb.cnt2, cap);
    uint8_t* sp = scratch + (size_t)blockIdx.y * si_pair_bytes(max_c2, M);
    int32_t* hdr = (int32_t*)sp;
    SiCand* gcand = (SiCand*)(sp + 16);
    SiQuery* gqry = (SiQuery*)(gcand + max_c2);
    uint32_t* topm = (uint32_t*)(gqry + max_c2);
    const int q0 = blockIdx.x * SI_QPB;

    /* Frame grid of frame 2 (frame.cpp:322-323, 746-756): mnMinX = 0, mnMaxX = cols (no distortion) */
    const float invW = __fdiv_rn((float)SI_GRID_COLS, (float)imgW);
    const float invH = __fdiv_rn((float)SI_GRID_ROWS, (float)imgH);

    int c2 = 0;
    for (int b = 0; b < n2; b += 256) {
        const int i = b + tid;
        bool take = false;
        vslam_kp k;
        int gx = 0, gy = 0;
        if (i < n2) {
            k = jb.k2[i];
            if (k.octave == 0) {
                gx = (int)roundf(__fmul_rn(k.x, invW));
                gy = (int)roundf(__fmul_rn(k.y, invH));
                take = !(gx < 0 || gx >= SI_GRID_COLS || gy < 0 || gy >= SI_GRID_ROWS); /* PosInGrid */
            }
        }
        const unsigned long long m = __ballot(take);
        if (lane == 0) s_wcnt[wave] = __popcll(m);
        __syncthreads();
        int pos = c2 + __popcll(m & ((1ull << lane) - 1ull));
        for (int w = 0; w < wave; w++) pos += s_wcnt[w];
        if (take && pos < max_c2) {
            SiCand cd;
            cd.x = k.x;
            cd.y = k.y;
            cd.angle = k.angle;
            cd.cell = (uint16_t)(gx * 64 + gy);
            cd.idx = (uint16_t)i;
            if (blockIdx.x == 0) gcand[pos] = cd;
            SiCandL cl;
            cl.x = k.x;
            cl.y = k.y;
            cl.c = (uint32_t)pos;
            cl.cell = cd.cell;
            cl.idx = cd.idx;
            tmpc[pos] = cl;
            atomicAdd(&s_col[gx], 1);
        }
        c2 += s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
        __syncthreads();
    }
    const int c2_raw = c2;
    c2 = min(c2, max_c2);
    /* sort the list by grid column (order inside a column is irrelevant: keys are unique), so that a query only visits
     * the columns of its window -- a sixth of the list at window 100 on a KITTI frame -- instead of testing all of it */
    if (wave == 0) {
        const int v = s_col[lane];
        int inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        s_col[lane] = inc - v;
        if (lane == 63) s_col[SI_GRID_COLS] = inc;
    }
    __syncthreads();
    for (int i = tid; i < c2; i += 256) {
        const SiCandL e = tmpc[i];
        const int gx = e.cell >> 6;
        cand[s_col[gx] + atomicAdd(&s_fill[gx], 1)] = e;
    }
    __syncthreads();
    int c1 = 0;
    for (int b = 0; b < n1; b += 256) {
        const int i = b + tid;
        vslam_kp k;
        bool take = false;
        if (i < n1) {
            k = jb.k1[i];
            take = k.octave == 0; /* level1 > 0 -> continue, fmatcher.cpp:999-1001 */
        }
        const unsigned long long m = __ballot(take);
        if (lane == 0) s_wcnt[wave] = __popcll(m);
        __syncthreads();
        int pos = c1 + __popcll(m & ((1ull << lane) - 1ull));
        for (int w = 0; w < wave; w++) pos += s_wcnt[w];
        if (take && pos >= q0 && pos < q0 + SI_QPB && pos < max_c2) {
            SiQuery q;
            q.px = jb.prev ? jb.prev[2 * i] : k.x;
            q.py = jb.prev ? jb.prev[2 * i + 1] : k.y;
            q.angle = k.angle;
            q.idx = (uint32_t)i;
            s_q[pos - q0] = q;
            gqry[pos] = q;
        }
        c1 += s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
        __syncthreads();
    }
    const int c1_raw = c1;
    c1 = min(c1, max_c2);
    if (blockIdx.x == 0 && tid == 0) {
        hdr[0] = c1;
        hdr[1] = c2;
        /* more octave-0 keypoints than the LDS-resident lists hold (keypoints of a foreign extractor configuration):
         * the result would be silently truncated, so k_si_replay reports the pair as failed instead */
        hdr[2] = (c1_raw > max_c2 || c2_raw > max_c2) ? 1 : 0;
    }

    const float r = (float)window;
    uint32_t* wk = wkeys + (size_t)wave * max_c2;
    for (int t = q0 + wave; t < min(c1, q0 + SI_QPB); t += 4) {
        const SiQuery qq = s_q[t - q0];
        const SiWindow win = si_window(qq.px, qq.py, r, invW, invH);
        uint32_t mine = 0xFFFFFFFFu; /* lane j < M ends up with the j-th smallest key */
        if (!win.empty) {
            const uint4 da = ((const uint4*)jb.d1)[(size_t)qq.idx * 2];
            const uint4 db = ((const uint4*)jb.d1)[(size_t)qq.idx * 2 + 1];
            int nk = 0; /* window candidates, compacted (their order is irrelevant: keys are unique) */
            const int lo = s_col[win.minX], hi = s_col[win.maxX + 1]; /* 0 <= minX <= maxX <= 63 for a non-empty window */
            for (int cb = lo; cb < hi; cb += 64) {
                const int c = cb + lane;
                bool in = false;
                uint32_t key = 0;
                if (c < hi) {
                    const SiCandL cd = cand[c];
                    if (si_in_window(cd, win, qq.px, qq.py, r)) {
                        const uint4 ta = ((const uint4*)jb.d2)[(size_t)cd.idx * 2];
                        const uint4 tb = ((const uint4*)jb.d2)[(size_t)cd.idx * 2 + 1];
                        const uint32_t dist = si_hamming(da, db, ta, tb);
                        key = (min(dist, 255u) << 24) | ((uint32_t)cd.cell << 12) | cd.c;
                        in = true;
                    }
                }
                const unsigned long long m = __ballot(in);
                if (in) wk[nk + __popcll(m & ((1ull << lane) - 1ull))] = key;
                nk += __popcll(m);
            }
            /* M rounds of "smallest key not yet taken" (keys are unique; same wave, LDS is in order) */
            uint32_t lower = 0;
            for (int j = 0; j < M; j++) {
                uint32_t lm = 0xFFFFFFFFu;
                for (int e = lane; e < nk; e += 64) {
                    const uint32_t kv = wk[e];
                    if (kv >= lower) lm = min(lm, kv);
                }
                const uint32_t g = wave_min_u32(lm);
                if (g == 0xFFFFFFFFu) break;
                if (lane == j) mine = g;
                lower = g + 1u;
            }
        }
        if (lane < M) topm[(size_t)t * M + lane] = mine;
    }
}

/* vMatchedDistance as the replay keeps it: per frame-2 slot up to four (writer + 1) << 8 | distance entries (0xFFFFFFFF =
 * empty; writer field 0 = merged entries of queries that everybody still pending comes after).  A query sees the
 * acceptances of EARLIER queries only -- later queries may have committed before it (see k_si_replay). */
__device__ __forceinline__ uint32_t sir_od_eff(const uint4 L, uint32_t q) {
    uint32_t od = 0x7FFFFFFFu;
    if ((L.x >> 8) <= q) od = min(od, L.x & 0xFFu);
    if ((L.y >> 8) <= q) od = min(od, L.y & 0xFFu);
    if ((L.z >> 8) <= q) od = min(od, L.z & 0xFFu);
    if ((L.w >> 8) <= q) od = min(od, L.w & 0xFFu);
    return od;
}

/* Full re-scan of one query's window (the reference loop body, fmatcher.cpp:1003-1035) by ONE wave: used when the query's
 * sorted prefix ran out.  Returns the best key; *second_out = bestDist2 (0x7FFFFFFF if none). */
__device__ uint32_t si_full_scan(const InitJob& jb, const SiCand* gcand, const uint4* slotlog, int c2, int lane,
                                 const SiQuery& qq, uint32_t qpos, float r, float invW, float invH, uint32_t* second_out) {
    const SiWindow win = si_window(qq.px, qq.py, r, invW, invH);
    uint32_t bestKey = 0xFFFFFFFFu, second = 0x7FFFFFFFu; /* per lane: best key, smallest other distance */
    if (!win.empty) {
        const uint4 da = ((const uint4*)jb.d1)[(size_t)qq.idx * 2];
        const uint4 db = ((const uint4*)jb.d1)[(size_t)qq.idx * 2 + 1];
        for (int c = lane; c < c2; c += 64) {
            const SiCand cd = gcand[c];
            if (!si_in_window(cd, win, qq.px, qq.py, r)) continue;
            const uint4 ta = ((const uint4*)jb.d2)[(size_t)cd.idx * 2];
            const uint4 tb = ((const uint4*)jb.d2)[(size_t)cd.idx * 2 + 1];
            const uint32_t dist = min(si_hamming(da, db, ta, tb), 255u);
            if (sir_od_eff(slotlog[c], qpos) <= dist) continue; /* vMatchedDistance[i2] <= dist, fmatcher.cpp:1022 */
            const uint32_t key = (dist << 24) | ((uint32_t)cd.cell << 12) | (uint32_t)c;
            if (key < bestKey) {
                if (bestKey != 0xFFFFFFFFu) second = min(second, bestKey >> 24);
                bestKey = key;
            } else {
                second = min(second, dist);
            }
        }
    }
    const uint32_t gBest = wave_min_u32(bestKey);
    /* second minimum over the multiset: lanes that do not hold the winner contribute their own best */
    uint32_t contrib = second;
    if (bestKey != gBest && bestKey != 0xFFFFFFFFu) contrib = min(contrib, bestKey >> 24);
    *second_out = wave_min_u32(contrib);
    return gBest;
}

/* append (writer q, distance d) to a slot's entries; `merge`: the writer is the smallest pending query, so every entry
 * present comes from an earlier query and is visible to everybody still pending -- they collapse into one.  Returns
 * false if the slot is full and merging is not allowed. */
__device__ __forceinline__ bool sir_slot_append(uint4* slotlog, uint32_t slot, uint32_t q, uint32_t d, bool merge) {
    uint4 L = slotlog[slot];
    const uint32_t e = ((q + 1u) << 8) | d;
    if (L.x == 0xFFFFFFFFu) L.x = e;
    else if (L.y == 0xFFFFFFFFu) L.y = e;
    else if (L.z == 0xFFFFFFFFu) L.z = e;
    else if (L.w == 0xFFFFFFFFu) L.w = e;
    else if (merge) {
        const uint32_t dm = min(min(L.x & 0xFFu, L.y & 0xFFu), min(L.z & 0xFFu, L.w & 0xFFu));
        L = make_uint4(dm, e, 0xFFFFFFFFu, 0xFFFFFFFFu);
    } else {
        return false;
    }
    slotlog[slot] = L;
    return true;
}

/* ------------------------------------------------------------------------------------------------
 * phase B: one workgroup per pair; the sequential loop of fmatcher.cpp:1003-1049 decided in a few rounds.
 *
 * The loop carries ONE piece of state, vMatchedDistance, and a query's decision is a function of its FIRST TWO
 * candidates (in key order) that are not skipped (vMatchedDistance[i2] <= dist, :1022); skipped ones never come back,
 * the array only decreases.  So a query's decision is final as soon as no EARLIER undecided query can still accept one
 * of those two slots -- and an undecided query can only ever accept a slot that is in its sorted prefix, not skipped,
 * with a distance <= TH_LOW (:1037).  Every round, all undecided queries at once:
 *   1. each evaluates the loop body against the acceptances of the queries BEFORE it (the per-slot entries carry their
 *      writer, so an acceptance committed early by a later query is invisible to it) and publishes its index at every
 *      slot it might still accept (atomicMin);
 *   2. a query whose best and second-best slots show no smaller index commits: appends its acceptance to the slot's
 *      entries and to the log.  The smallest undecided query always commits, so the rounds end; on consecutive KITTI-like
 *      frames 217 (436) queries take 4-6 rounds (tests/tools/replay_sim.py), where committing only the prefix in front of
 *      the first dependent query of a 64-query window (round 3's first version) took 30 (48) and one query per step 217.
 * Two rare cases wait until they are the smallest undecided query: a query whose sorted prefix ran out and that needs the
 * re-scan of its whole window (its second-best lies beyond the list) -- and while an undecided query could accept a slot
 * BEYOND its list (full list, last listed distance <= TH_LOW), nothing behind it commits.
 * Ownership ("last acceptor wins", which is what the reference's steal/undo amounts to), vnMatches12 and the rotation
 * histogram are rebuilt from the log in parallel afterwards.
 * ---------------------------------------------------------------------------------------------- */
#ifndef SIR_NT
#define SIR_NT 256
#endif
#define SIR_PEND 1u
#define SIR_ACC 2u
#define SIR_SCAN 4u
template <int MAXM> /* unrolled length of a query's sorted prefix: 8 (the default M) or SI_MAX_M */
__global__ void __launch_bounds__(SIR_NT)
k_si_replay(InitJobs jobs, int cap, int imgW, int imgH, int window, float nnratio, int checkOri,
            int32_t* matches_out /* [pair][cap] */, float* prev_out /* [pair][2*cap] */,
            int32_t* nmatch_out /* [pair] */, int max_c2, int M, const uint8_t* scratch, int* fallbacks, int keys_lds_bytes,
            int prio) {
    extern __shared__ __align__(16) uint8_t sism[];
    wave_prio_raise(prio);
    const InitJob jb = jobs.job[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63;
    const int n1 = min(*jb.cnt1, cap);
    const uint8_t* sp = scratch + (size_t)blockIdx.x * si_pair_bytes(max_c2, M);
    const int32_t* hdr = (const int32_t*)sp;
    const SiCand* gcand = (const SiCand*)(sp + 16);
    const SiQuery* gqry = (const SiQuery*)(gcand + max_c2);
    const uint32_t* topm = (const uint32_t*)(gqry + max_c2);
    const int c1 = hdr[0], c2 = hdr[1];
    /* LDS: slotlog | lister[2] | kfirst | ksecond | alog | m12 | qstate | rotBin */
    uint4* slotlog = (uint4*)sism;                                /* vMatchedDistance with writers, max_c2 */
    int32_t* lister0 = (int32_t*)(slotlog + max_c2);              /* smallest undecided query that may still accept the slot */
    int32_t* lister1 = lister0 + max_c2;                          /* (two buffers: one is reset while the other is in use) */
    uint32_t* kfirst = (uint32_t*)(lister1 + max_c2);             /* this round's best / second-best key of a query */
    uint32_t* ksecond = kfirst + max_c2;
    uint32_t* alog = ksecond + max_c2;                            /* accepted (query position << 12 | slot), any order */
    int32_t* m12 = (int32_t*)(alog + max_c2);                     /* vnMatches12, cap */
    uint8_t* qstate = (uint8_t*)(m12 + cap);                      /* SIR_* flags per query, max_c2 */
    uint8_t* rotBin = qstate + max_c2;                            /* bin of an accepted query, 255 = none; cap */
    /* the queries' sorted prefixes are read once per round: kept in LDS when they fit what the host set aside */
    uint32_t* keys_lds = (uint32_t*)(sism + (((size_t)max_c2 * 37 + (size_t)cap * 5 + 15) & ~(size_t)15));
    const bool keys_in_lds = (size_t)c1 * M * 4 <= (size_t)keys_lds_bytes;
    const uint32_t* keysrc = keys_in_lds ? keys_lds : topm;
    if (keys_in_lds)
        for (int i = tid; i < c1 * M; i += SIR_NT) keys_lds[i] = topm[i];
    __shared__ int s_hist[SI_HISTO];
    __shared__ int s_minpend[2], s_minwild[2], s_scanq, s_nlog, s_cnt;

    int32_t* mo = matches_out + (size_t)blockIdx.x * cap;
    float* po = prev_out + (size_t)blockIdx.x * cap * 2;
    const float invW = __fdiv_rn((float)SI_GRID_COLS, (float)imgW);
    const float invH = __fdiv_rn((float)SI_GRID_ROWS, (float)imgH);
    const float r = (float)window;

    for (int c = tid; c < c2; c += SIR_NT) {
        slotlog[c] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
        lister0[c] = 0x7FFFFFFF;
        lister1[c] = 0x7FFFFFFF;
    }
    for (int q = tid; q < c1; q += SIR_NT) qstate[q] = (uint8_t)SIR_PEND;
    for (int i = tid; i < n1; i += SIR_NT) {
        m12[i] = -1;
        rotBin[i] = 255;
    }
    if (tid < SI_HISTO) s_hist[tid] = 0;
    if (tid == 0) {
        s_minpend[0] = s_minpend[1] = 0x7FFFFFFF;
        s_minwild[0] = s_minwild[1] = 0x7FFFFFFF;
        s_scanq = -1;
        s_nlog = 0;
        s_cnt = 0;
    }
    __syncthreads();

    int nfb = 0, nrounds = 0;
    for (int par = 0;; par ^= 1) {
        int32_t* lister = par ? lister1 : lister0;
        if (tid == 0) s_scanq = -1; /* everybody has read last round's value (it is only set in part 2, behind a barrier) */
        /* ---- 1: the reference's loop body (fmatcher.cpp:1003-1039) for every undecided query, as the queries before it left the state */
        for (int q = tid; q < c1; q += SIR_NT) {
            if (!(qstate[q] & SIR_PEND)) continue;
            uint32_t key[MAXM];
#pragma unroll
            for (int j = 0; j < MAXM; j++) key[j] = j < M ? keysrc[(size_t)q * M + j] : 0xFFFFFFFFu;
            int nvalid = 0;
            uint32_t gBest = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu, klast = 0u;
#pragma unroll
            for (int j = 0; j < MAXM; j++) {
                if (j < M && key[j] != 0xFFFFFFFFu) {
                    nvalid++;
                    klast = key[j];
                    const uint32_t slot = key[j] & 0xFFFu, d = key[j] >> 24;
                    if (!(sir_od_eff(slotlog[slot], (uint32_t)q) <= d)) { /* not skipped */
                        if (d <= SI_TH_LOW) atomicMin(&lister[slot], q); /* may still accept this slot */
                        if (gBest == 0xFFFFFFFFu) gBest = key[j];
                        else if (k2 == 0xFFFFFFFFu) k2 = key[j];
                    }
                }
            }
            const bool full = nvalid == M;
            const uint32_t dlast = klast >> 24;
            const bool wild = full && dlast <= SI_TH_LOW; /* could accept a slot beyond its list once the list is used up */
            bool accept = false, scan = false;
            if (gBest == 0xFFFFFFFFu) {
                scan = wild; /* else: vIndices2 empty, everything skipped, or nothing within TH_LOW beyond the list */
            } else if ((gBest >> 24) <= SI_TH_LOW) {
                uint32_t bestDist2 = k2 == 0xFFFFFFFFu ? 0x7FFFFFFFu : (k2 >> 24);
                if (k2 == 0xFFFFFFFFu && full) {
                    /* the second survivor lies beyond the list: it is at least as far as the last entry */
                    if ((float)(int)(gBest >> 24) < __fmul_rn((float)(int)dlast, nnratio)) bestDist2 = dlast; /* accepted whatever the true second is */
                    else scan = true;
                }
                if (!scan) accept = (float)(int)(gBest >> 24) < __fmul_rn((float)(int)bestDist2, nnratio);
            }
            kfirst[q] = gBest;
            ksecond[q] = k2;
            qstate[q] = (uint8_t)(SIR_PEND | (accept ? SIR_ACC : 0u) | (scan ? SIR_SCAN : 0u));
            atomicMin(&s_minpend[par], q);
            if (wild) atomicMin(&s_minwild[par], q);
        }
        __syncthreads();
        const int minp = s_minpend[par], minw = s_minwild[par];
        if (minp == 0x7FFFFFFF) break; /* nothing undecided (block-uniform) */
        nrounds++;
        /* ---- 2: commit what no earlier undecided query can change; reset the other round's buffers */
        {
            int32_t* other = par ? lister0 : lister1;
            for (int c = tid; c < c2; c += SIR_NT) other[c] = 0x7FFFFFFF;
            if (tid == 0) {
                s_minpend[par ^ 1] = 0x7FFFFFFF;
                s_minwild[par ^ 1] = 0x7FFFFFFF;
            }
        }
        for (int q = tid; q < c1; q += SIR_NT) {
            const uint32_t fl = qstate[q];
            if (!(fl & SIR_PEND)) continue;
            if (fl & SIR_SCAN) {
                if (q == minp) s_scanq = q; /* everything in front of it is decided: re-scan below */
                continue;
            }
            const uint32_t k1 = kfirst[q], k2 = ksecond[q];
            bool blocked = q > minw;
            if (k1 != 0xFFFFFFFFu && lister[k1 & 0xFFFu] < q) blocked = true;
            if (k2 != 0xFFFFFFFFu && lister[k2 & 0xFFFu] < q) blocked = true;
            if (blocked) continue;
            if (fl & SIR_ACC) {
                if (!sir_slot_append(slotlog, k1 & 0xFFFu, (uint32_t)q, k1 >> 24, q == minp)) continue; /* slot full: wait to be first */
                alog[atomicAdd(&s_nlog, 1)] = ((uint32_t)q << 12) | (k1 & 0xFFFu);
            }
            qstate[q] = 0;
        }
        __syncthreads();
        /* ---- 3 (rare): the smallest undecided query's sorted prefix ran out -- re-scan its whole window, one wave
         * (fmatcher.cpp:1003-1035 literally); it sees the acceptances of the queries before it only */
        const int sq = s_scanq;
        if (sq >= 0) { /* block-uniform */
            if (tid < 64) {
                uint32_t d2;
                const uint32_t gb = si_full_scan(jb, gcand, slotlog, c2, lane, gqry[sq], (uint32_t)sq, r, invW, invH, &d2);
                if (tid == 0) {
                    if (gb != 0xFFFFFFFFu && (gb >> 24) <= SI_TH_LOW && (float)(int)(gb >> 24) < __fmul_rn((float)(int)d2, nnratio)) {
                        sir_slot_append(slotlog, gb & 0xFFFu, (uint32_t)sq, gb >> 24, true);
                        alog[atomicAdd(&s_nlog, 1)] = ((uint32_t)sq << 12) | (gb & 0xFFFu);
                    }
                    qstate[sq] = 0;
                }
            }
            nfb++;
            __syncthreads();
        }
    }
    __syncthreads();
    const int nlog = s_nlog;
    if (fallbacks && tid == 0) { /* [0] re-scans (vslam_dbg_search_init_fallbacks), [1..3] rounds, queries, pairs */
        if (nfb) atomicAdd(fallbacks, nfb);
        atomicAdd(fallbacks + 1, nrounds);
        atomicAdd(fallbacks + 2, c1);
        atomicAdd(fallbacks + 3, 1);
    }
    /* ownership: the last acceptor of a slot (in query order) keeps it (fmatcher.cpp:1041-1049 undoes the previous owner) */
    int32_t* ownerQ = lister0;
    for (int c = tid; c < c2; c += SIR_NT) ownerQ[c] = -1;
    __syncthreads();
    for (int e = tid; e < nlog; e += SIR_NT) atomicMax(&ownerQ[alog[e] & 0xFFFu], (int)(alog[e] >> 12));
    __syncthreads();
    const float factor = 1.0f / SI_HISTO;
    for (int e = tid; e < nlog; e += SIR_NT) {
        const uint32_t le = alog[e];
        const SiQuery q = gqry[le >> 12];
        const SiCand cd = gcand[le & 0xFFFu];
        if (ownerQ[le & 0xFFFu] == (int)(le >> 12)) m12[q.idx] = cd.idx;
        if (checkOri) {
            float rot = __fsub_rn(q.angle, cd.angle);
            if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
            int bin = (int)roundf(__fmul_rn(rot, factor));
            if (bin == SI_HISTO) bin = 0;
            rotBin[q.idx] = (uint8_t)bin;
            atomicAdd(&s_hist[bin], 1); /* rotHist[bin].push_back(i1): never removed, even if stolen later */
        }
    }
    __syncthreads();
    /* nmatches is counted from vnMatches12 at the end: the reference's running count equals it */
    if (checkOri) {
        /* ComputeThreeMaxima (fmatcher.cpp:2813-2854) */
        int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
        for (int i = 0; i < SI_HISTO; i++) {
            const int s = s_hist[i];
            if (s > max1) {
                max3 = max2; max2 = max1; max1 = s;
                ind3 = ind2; ind2 = ind1; ind1 = i;
            } else if (s > max2) {
                max3 = max2; max2 = s;
                ind3 = ind2; ind2 = i;
            } else if (s > max3) {
                max3 = s;
                ind3 = i;
            }
        }
        if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) { ind3 = -1; }
        for (int i = tid; i < n1; i += SIR_NT) {
            const int b = rotBin[i];
            if (b != 255 && b != ind1 && b != ind2 && b != ind3) m12[i] = -1;
        }
        __syncthreads();
    }
    int cnt = 0;
    /* four keypoints per thread and pass: their (dependent) global loads are in flight together */
    for (int i0 = tid; i0 < n1; i0 += 4 * SIR_NT) {
        int m[4];
        float2 o[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + u * SIR_NT;
            m[u] = i < n1 ? m12[i] : -1;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + u * SIR_NT;
            o[u] = make_float2(0.0f, 0.0f);
            if (i < n1) {
                /* vbPrevMatched update (fmatcher.cpp:1093-1095) */
                if (m[u] >= 0) o[u] = make_float2(jb.k2[m[u]].x, jb.k2[m[u]].y);
                else if (jb.prev) o[u] = make_float2(jb.prev[2 * i], jb.prev[2 * i + 1]);
                else o[u] = make_float2(jb.k1[i].x, jb.k1[i].y);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + u * SIR_NT;
            if (i < n1) {
                mo[i] = m[u];
                po[2 * i] = o[u].x;
                po[2 * i + 1] = o[u].y;
                cnt += m[u] >= 0;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if (lane == 0 && cnt) atomicAdd(&s_cnt, cnt);
    __syncthreads();
    if (tid == 0) nmatch_out[blockIdx.x] = hdr[2] ? -1 : s_cnt; /* -1: capacity overflow (vslam_search_init_dev_wait) */
}

size_t vk_search_init_scratch_bytes(int npairs, int max_c2, int M) { return (size_t)npairs * si_pair_bytes(max_c2, M); }

static size_t si_topm_lds(int max_c2) { return (size_t)max_c2 * (sizeof(SiCand) + 4 * 4); }
/* state (37 bytes per octave-0 keypoint, 5 per keypoint) + the sorted prefixes of up to max_c2 queries if that stays within 64 KB */
static size_t si_replay_keys_lds(int max_c2, int M) {
    const size_t want = (size_t)max_c2 * M * 4;
    return want <= (64u << 10) ? want : 0;
}
static size_t si_replay_lds(int cap, int max_c2, int M = 0) {
    return (((size_t)max_c2 * 37 + (size_t)cap * 5 + 15) & ~(size_t)15) + si_replay_keys_lds(max_c2, M) + 16;
}

size_t vk_search_init_lds(int cap, int max_c2) { return std::max(si_topm_lds(max_c2), si_replay_lds(cap, max_c2)); } /* without the optional key cache */

int vk_search_init_set_max_lds(size_t bytes) {
    int rc = (int)hipFuncSetAttribute((const void*)k_si_topm, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (rc) return rc;
    rc = (int)hipFuncSetAttribute((const void*)k_si_replay<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (rc) return rc;
    return (int)hipFuncSetAttribute((const void*)k_si_replay<SI_MAX_M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

void vk_search_init(hipStream_t st, const InitJobs& jobs, int npairs, int cap, int imgW, int imgH, int window,
                    float nnratio, int checkOri, int32_t* matches_out, float* prev_out, int32_t* nmatch_out,
                    int max_c2, int M, uint8_t* scratch, int* fallbacks, const vslam_tuning& T) {
    if (npairs <= 0) return;
    /* queries per workgroup: every workgroup first rebuilds the pair's candidate list and finds its queries (a scan of
     * both frames' keypoints), so fewer, longer workgroups repeat less of that; VSLAM_SI_QPB = 8 | 16 | 32 for A/B runs */
    const int qv = T.si_queries_per_block;
    const int qpb = (qv == 8 || qv == 16 || qv == 32) ? qv : 16;
    const int chunks = (max_c2 + qpb - 1) / qpb;
    const int prio = T.wave_prio > 0 && (T.wave_prio & 4) ? 1 : 0;
    hipLaunchKernelGGL(k_si_topm, dim3(chunks, npairs), dim3(256), si_topm_lds(max_c2), st, jobs, cap, imgW, imgH,
                       window, max_c2, M, scratch, qpb, prio);
    /* the key cache only if the whole allocation stays within what vk_search_init_set_max_lds allowed (150 KB) */
    int keys_lds = (int)si_replay_keys_lds(max_c2, M);
    if (si_replay_lds(cap, max_c2, M) > (150u << 10)) keys_lds = 0;
    const size_t rlds = keys_lds ? si_replay_lds(cap, max_c2, M) : si_replay_lds(cap, max_c2);
    if (M <= 8)
        hipLaunchKernelGGL(k_si_replay<8>, dim3(npairs), dim3(SIR_NT), rlds, st, jobs, cap, imgW, imgH,
                           window, nnratio, checkOri, matches_out, prev_out, nmatch_out, max_c2, M, scratch, fallbacks, keys_lds, prio);
    else
        hipLaunchKernelGGL(k_si_replay<SI_MAX_M>, dim3(npairs), dim3(SIR_NT), rlds, st, jobs, cap, imgW, imgH,
                           window, nnratio, checkOri, matches_out, prev_out, nmatch_out, max_c2, M, scratch, fallbacks, keys_lds, prio);
}

/* ==================================================================================================
 * FMatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono)
 * (fmatcher.cpp:2471-2687, pinhole frames: Nleft == -1) -- the per-frame tracking matcher of
 * TrackWithMotionModel (tracking.cpp:2728).  Same decomposition as SearchForInitialization above:
 *
 *   k_sbp_rank   (parallel, one wave per last-frame keypoint): project its MapPoint into the current frame
 *                (Rcw*x3Dw+tcw as ONE cv::gemm: products accumulated in double, rounded once; 1.0/z in
 *                double; Pinhole::project in float), query the window (Frame::GetFeaturesInArea with the
 *                forward / backward / +-1 octave rule), apply the order-free mvuRight gate, and write the M
 *                smallest keys = dist << 24 | gridcell << 12 | i2 in ascending order ("first wins" on equal
 *                distance is the key order).
 *   k_sbp_replay (one wave, last-frame keypoints in index order): the only sequential state is "current
 *                keypoint i2 already carries a MapPoint with observations" (the skip at fmatcher.cpp:2545-
 *                2547); the first list entry that is not occupied is bestIdx2.  An exhausted full list falls
 *                back to a full re-scan of that query.  Assignments go to a log; mvpMapPoints ("last writer
 *                wins"), the rotation histogram, ComputeThreeMaxima and nmatches are rebuilt from it.
 * ================================================================================================== */
#define SBP_TH_HIGH 100
#define SBP_QPB 16 /* queries per k_sbp_rank workgroup (4 waves x 4) */

struct SbpCand { /* one keypoint of the current frame */
    float x, y, uRight;
    uint16_t cell;
    uint16_t octave;
};
struct SbpProj { /* projection record of one last-frame keypoint (k_sbp_rank -> k_sbp_replay's re-scan) */
    float u, v, radius, ur;
    int32_t minLevel, maxLevel, valid, pad;
};

__device__ __forceinline__ float sbp_gemm_row(const float* r, float x0, float x1, float x2, float t, int flt) {
    if (!flt) { /* GEMMSingleMul<float,double>: s += double(a)*double(b), then float(s + double(c)) */
        double s = __dmul_rn((double)r[0], (double)x0);
        s = __dadd_rn(s, __dmul_rn((double)r[1], (double)x1));
        s = __dadd_rn(s, __dmul_rn((double)r[2], (double)x2));
        return (float)__dadd_rn(s, (double)t);
    }
    float s = __fmul_rn(r[0], x0);
    s = __fadd_rn(s, __fmul_rn(r[1], x1));
    s = __fadd_rn(s, __fmul_rn(r[2], x2));
    return __fadd_rn(s, t);
}

__device__ __forceinline__ bool sbp_level_ok(int octave, int minLevel, int maxLevel) { /* frame.cpp:712-728 */
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    if (!bCheckLevels) return true;
    if (octave < minLevel) return false;
    if (maxLevel >= 0 && octave > maxLevel) return false;
    return true;
}

/* the order-free part of the candidate loop body (fmatcher.cpp:2541-2556): window, level, mvuRight gate */
__device__ __forceinline__ bool sbp_candidate_ok(const SbpCand& cd, const SiWindow& w, const SbpProj& pr) {
    const int gx = cd.cell >> 6, gy = cd.cell & 63;
    if (gx < w.minX || gx > w.maxX || gy < w.minY || gy > w.maxY) return false;
    if (!sbp_level_ok(cd.octave, pr.minLevel, pr.maxLevel)) return false;
    const float distx = __fsub_rn(cd.x, pr.u), disty = __fsub_rn(cd.y, pr.v);
    if (!(fabsf(distx) < pr.radius && fabsf(disty) < pr.radius)) return false;
    if (cd.uRight > 0.f) { /* CurrentFrame.mvuRight[i2] > 0 */
        const float er = fabsf(__fsub_rn(pr.ur, cd.uRight));
        if (er > pr.radius) return false;
    }
    return true;
}

__device__ __forceinline__ SbpCand sbp_make_cand(const vslam_kp& k, float uRight, float invW, float invH, bool* in_grid) {
    SbpCand c;
    c.x = k.x;
    c.y = k.y;
    c.uRight = uRight;
    const int gx = (int)roundf(__fmul_rn(k.x, invW)), gy = (int)roundf(__fmul_rn(k.y, invH));
    *in_grid = !(gx < 0 || gx >= SI_GRID_COLS || gy < 0 || gy >= SI_GRID_ROWS); /* PosInGrid, frame.cpp:746-756 */
    c.cell = (uint16_t)(*in_grid ? gx * 64 + gy : 0xFFFF);
    c.octave = (uint16_t)k.octave;
    return c;
}

__global__ void __launch_bounds__(256)
k_sbp_rank(SbpJobs JS) {
    extern __shared__ __align__(16) uint8_t sbsm[];
    SbpCand* cand = (SbpCand*)sbsm; /* nCur */
    const SbpJobDev& J = JS.job[blockIdx.y];
    const int M = JS.M;
    const int nLast = J.nLastPtr ? min(*J.nLastPtr, J.nLast) : J.nLast;
    const int nCur = J.nCurPtr ? min(*J.nCurPtr, J.nCur) : J.nCur;
    if ((int)blockIdx.x * SBP_QPB >= nLast) return; /* grid is sized for the capacity */
    const vslam_kp* __restrict__ lastKps = J.lastKps;
    const vslam_kp* __restrict__ curKps = J.curKps;
    const uint8_t* __restrict__ curDesc = J.curDesc;
    const uint8_t* __restrict__ mpDesc = J.mpDesc;
    const float* __restrict__ uRight = J.uRight;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float invW = __fdiv_rn((float)SI_GRID_COLS, (float)J.imgW);
    const float invH = __fdiv_rn((float)SI_GRID_ROWS, (float)J.imgH);
    for (int i = tid; i < nCur; i += 256) {
        bool ing;
        cand[i] = sbp_make_cand(curKps[i], uRight ? uRight[i] : -1.f, invW, invH, &ing);
    }
    __syncthreads();
    for (int q = blockIdx.x * SBP_QPB + wave; q < min(nLast, (int)(blockIdx.x + 1) * SBP_QPB); q += 4) {
        SbpProj pr;
        pr.valid = 0;
        pr.u = pr.v = pr.radius = pr.ur = 0.f;
        pr.minLevel = pr.maxLevel = 0;
        pr.pad = 0;
        if (J.mode == 1) { /* pre-projected MapPoints (fmatcher.cpp:327-350) */
            const MpTrack mp = J.mps[q];
            if (mp.flags & 1) {
                float r = (double)mp.viewCos > 0.998 ? 2.5f : 4.0f; /* RadiusByViewingCos, fmatcher.cpp:493-499 */
                if (J.th != 1.0f) r = __fmul_rn(r, J.th);
                pr.u = mp.projX;
                pr.v = mp.projY;
                pr.radius = __fmul_rn(r, JS.scale[min(max(mp.level, 0), JS.nlevels - 1)]);
                pr.ur = mp.projXR;
                pr.minLevel = mp.level - 1;
                pr.maxLevel = mp.level;
                pr.valid = 1;
            }
        } else if (J.mode == 2) { /* KeyFrame MapPoints into the current frame, fmatcher.cpp:2705-2743 */
            if (J.flags[q] & 1) { /* pMP && !isBad() && !sAlreadyFound.count(pMP) */
                const float X = J.x3Dw[3 * q], Y = J.x3Dw[3 * q + 1], Z = J.x3Dw[3 * q + 2];
                const float xc = sbp_gemm_row(J.Tcw + 0, X, Y, Z, J.Tcw[3], J.gemmFloat);
                const float yc = sbp_gemm_row(J.Tcw + 4, X, Y, Z, J.Tcw[7], J.gemmFloat);
                const float zc = sbp_gemm_row(J.Tcw + 8, X, Y, Z, J.Tcw[11], J.gemmFloat);
                const float u = __fadd_rn(__fdiv_rn(__fmul_rn(J.fx, xc), zc), J.cx); /* no depth test here */
                const float v = __fadd_rn(__fdiv_rn(__fmul_rn(J.fy, yc), zc), J.cy);
                if (!(u < 0.f || u > (float)J.imgW) && !(v < 0.f || v > (float)J.imgH)) {
                    /* Ow = -Rcw^T tcw is cv::Mat algebra on the caller's side: it arrives in JS.kf.ow */
                    const float p0 = __fsub_rn(X, JS.kf.ow[0]), p1 = __fsub_rn(Y, JS.kf.ow[1]), p2 = __fsub_rn(Z, JS.kf.ow[2]);
                    double n2 = __dmul_rn((double)p0, (double)p0); /* cv::norm: double accumulation */
                    n2 = __dadd_rn(n2, __dmul_rn((double)p1, (double)p1));
                    n2 = __dadd_rn(n2, __dmul_rn((double)p2, (double)p2));
                    const float dist3D = (float)__dsqrt_rn(n2);
                    const float mn = JS.kf.minDist[q], mx = JS.kf.maxDist[q];
                    if (!(dist3D < mn || dist3D > mx)) {
                        const float lv = ceilf(__fdiv_rn(vslam_trig::glibc_logf(__fdiv_rn(mx, dist3D)), JS.kf.logScaleFactor));
                        const int level = (lv != lv || lv >= 2147483648.0f || lv < 0.f) ? 0 : min((int)lv, JS.nlevels - 1);
                        pr.u = u;
                        pr.v = v;
                        pr.radius = __fmul_rn(J.th, JS.scale[level]);
                        pr.minLevel = level - 1;
                        pr.maxLevel = level + 1;
                        pr.valid = 1;
                    }
                }
            }
        } else if (J.mode == 3) { /* candidate MapPoints into a KeyFrame under Scw, fmatcher.cpp:773-825 / :890-941 */
            if (J.flags[q] & 1) { /* !pMP->isBad() && !spAlreadyFound.count(pMP) */
                const float X = J.x3Dw[3 * q], Y = J.x3Dw[3 * q + 1], Z = J.x3Dw[3 * q + 2];
                const float xc = sbp_gemm_row(J.Tcw + 0, X, Y, Z, J.Tcw[3], J.gemmFloat);
                const float yc = sbp_gemm_row(J.Tcw + 4, X, Y, Z, J.Tcw[7], J.gemmFloat);
                const float zc = sbp_gemm_row(J.Tcw + 8, X, Y, Z, J.Tcw[11], J.gemmFloat);
                if (!(zc < 0.0f)) {
                    float u, v;
                    if (JS.kf.projKind == 0) {
                        u = __fadd_rn(__fdiv_rn(__fmul_rn(J.fx, xc), zc), J.cx);
                        v = __fadd_rn(__fdiv_rn(__fmul_rn(J.fy, yc), zc), J.cy);
                    } else {
                        const float invz = __fdiv_rn(1.0f, zc);
                        u = __fadd_rn(__fmul_rn(J.fx, __fmul_rn(xc, invz)), J.cx);
                        v = __fadd_rn(__fmul_rn(J.fy, __fmul_rn(yc, invz)), J.cy);
                    }
                    if (u >= 0.f && u < (float)J.imgW && v >= 0.f && v < (float)J.imgH) { /* KeyFrame::IsInImage */
                        const float p0 = __fsub_rn(X, JS.kf.ow[0]), p1 = __fsub_rn(Y, JS.kf.ow[1]), p2 = __fsub_rn(Z, JS.kf.ow[2]);
                        double n2 = __dmul_rn((double)p0, (double)p0);
                        n2 = __dadd_rn(n2, __dmul_rn((double)p1, (double)p1));
                        n2 = __dadd_rn(n2, __dmul_rn((double)p2, (double)p2));
                        const float dist3D = (float)__dsqrt_rn(n2);
                        const float mn = JS.kf.minDist[q], mx = JS.kf.maxDist[q];
                        bool ok = !(dist3D < mn || dist3D > mx);
                        if (ok) { /* PO.dot(Pn) < 0.5*dist: cv::Mat::dot accumulates in double */
                            const float* Pn = JS.kf.normals + 3 * q;
                            double dot = __dmul_rn((double)p0, (double)Pn[0]);
                            dot = __dadd_rn(dot, __dmul_rn((double)p1, (double)Pn[1]));
                            dot = __dadd_rn(dot, __dmul_rn((double)p2, (double)Pn[2]));
                            ok = !(dot < __dmul_rn(0.5, (double)dist3D));
                        }
                        if (ok) {
                            const float lv = ceilf(__fdiv_rn(vslam_trig::glibc_logf(__fdiv_rn(mx, dist3D)), JS.kf.logScaleFactor));
                            const int level = (lv != lv || lv >= 2147483648.0f || lv < 0.f) ? 0 : min((int)lv, JS.nlevels - 1);
                            pr.u = u;
                            pr.v = v;
                            pr.radius = __fmul_rn(J.th, JS.scale[level]);
                            pr.minLevel = level - 1;
                            pr.maxLevel = level;
                            pr.valid = 1;
                        }
                    }
                }
            }
        } else if (J.flags[q] & 1) { /* pMP && !LastFrame.mvbOutlier[i] */
            const float X = J.x3Dw[3 * q], Y = J.x3Dw[3 * q + 1], Z = J.x3Dw[3 * q + 2];
            const float xc = sbp_gemm_row(J.Tcw + 0, X, Y, Z, J.Tcw[3], J.gemmFloat);
            const float yc = sbp_gemm_row(J.Tcw + 4, X, Y, Z, J.Tcw[7], J.gemmFloat);
            const float zc = sbp_gemm_row(J.Tcw + 8, X, Y, Z, J.Tcw[11], J.gemmFloat);
            const float invzc = (float)__ddiv_rn(1.0, (double)zc);
            if (!(invzc < 0.f)) {
                const float u = __fadd_rn(__fdiv_rn(__fmul_rn(J.fx, xc), zc), J.cx); /* pinhole.cpp:13-16 */
                const float v = __fadd_rn(__fdiv_rn(__fmul_rn(J.fy, yc), zc), J.cy);
                if (!(u < 0.f || u > (float)J.imgW) && !(v < 0.f || v > (float)J.imgH)) {
                    const int oct = lastKps[q].octave;
                    pr.u = u;
                    pr.v = v;
                    pr.radius = __fmul_rn(J.th, JS.scale[min(max(oct, 0), JS.nlevels - 1)]);
                    pr.ur = __fsub_rn(u, __fmul_rn(J.mbf, invzc));
                    if (J.forward) { pr.minLevel = oct; pr.maxLevel = -1; }
                    else if (J.backward) { pr.minLevel = 0; pr.maxLevel = oct; }
                    else { pr.minLevel = oct - 1; pr.maxLevel = oct + 1; }
                    pr.valid = 1;
                }
            }
        }
        /* per-lane sorted prefix (ascending), then an M-way merge across the wave */
        uint32_t best[SI_MAX_M];
#pragma unroll
        for (int j = 0; j < SI_MAX_M; j++) best[j] = 0xFFFFFFFFu;
        if (pr.valid) { /* wave-uniform */
            const SiWindow win = si_window(pr.u, pr.v, pr.radius, invW, invH);
            if (!win.empty) {
                const uint4 da = ((const uint4*)mpDesc)[(size_t)q * 2], db = ((const uint4*)mpDesc)[(size_t)q * 2 + 1];
                for (int c = lane; c < nCur; c += 64) {
                    const SbpCand cd = cand[c];
                    if (cd.cell == 0xFFFF || !sbp_candidate_ok(cd, win, pr)) continue;
                    const uint4 ta = ((const uint4*)curDesc)[(size_t)c * 2], tb = ((const uint4*)curDesc)[(size_t)c * 2 + 1];
                    uint32_t key = (min(si_hamming(da, db, ta, tb), 255u) << 24) | ((uint32_t)cd.cell << 12) | (uint32_t)c;
#pragma unroll
                    for (int j = 0; j < SI_MAX_M; j++) { /* insertion: keeps best[] ascending, drops the largest */
                        const uint32_t lo = min(key, best[j]);
                        key = max(key, best[j]);
                        best[j] = lo;
                    }
                }
            }
        }
        uint32_t mine = 0xFFFFFFFFu;
        for (int j = 0; j < M; j++) {
            const uint32_t g = wave_min_u32(best[0]);
            if (g == 0xFFFFFFFFu) break;
            if (lane == j) mine = g;
            if (best[0] == g) { /* pop (keys are unique: they contain i2) */
#pragma unroll
                for (int t = 0; t + 1 < SI_MAX_M; t++) best[t] = best[t + 1];
                best[SI_MAX_M - 1] = 0xFFFFFFFFu;
            }
        }
        if (lane < M) J.topm[(size_t)q * M + lane] = mine;
        if (lane == 0) J.proj[q] = pr;
    }
}

/* full re-scan of one query's window in candidate order (the reference loop body) -> best key or ~0 */
__device__ uint32_t sbp_full_scan(const SbpProj& pr, const vslam_kp* curKps, const float* uRight, int nCur,
                                  const uint8_t* mpDescQ, const uint8_t* curDesc, const uint32_t* occupied, float invW,
                                  float invH, int lane) {
    uint32_t bestKey = 0xFFFFFFFFu;
    const SiWindow win = si_window(pr.u, pr.v, pr.radius, invW, invH);
    if (!win.empty) {
        const uint4 da = ((const uint4*)mpDescQ)[0], db = ((const uint4*)mpDescQ)[1];
        for (int c = lane; c < nCur; c += 64) {
            bool ing;
            const SbpCand cd = sbp_make_cand(curKps[c], uRight ? uRight[c] : -1.f, invW, invH, &ing);
            if (!ing || occupied[c] || !sbp_candidate_ok(cd, win, pr)) continue;
            const uint4 ta = ((const uint4*)curDesc)[(size_t)c * 2], tb = ((const uint4*)curDesc)[(size_t)c * 2 + 1];
            bestKey = min(bestKey, (min(si_hamming(da, db, ta, tb), 255u) << 24) | ((uint32_t)cd.cell << 12) | (uint32_t)c);
        }
    }
    return wave_min_u32(bestKey);
}

/* ------------------------------------------------------------------------------------------------
 * The sequential rule "query q takes the first candidate of its sorted list that no earlier query with
 * observations has taken" is a serial dictatorship, and a serial dictatorship is the fixpoint of deferred
 * acceptance with one common priority order: every query points at a candidate; a candidate is held by the
 * lowest-index blocking query pointing at it; whoever sees a lower-index holder moves on to its next candidate.
 * Holders' indices only decrease, so a rejection is final and the fixpoint is the sequential result.  One
 * 1024-thread workgroup per job iterates this in parallel (a handful of rounds) instead of one wave walking
 * 2000 queries in order.  Only when some query runs off a FULL sorted prefix (it would need candidates that
 * k_sbp_rank did not keep) the job is handed to k_sbp_replay, which walks it sequentially with full re-scans.
 * ---------------------------------------------------------------------------------------------- */
#define SBP_RT 1024
#define SBP_QPT 4 /* queries per thread: capacity 4096 */
__global__ void __launch_bounds__(SBP_RT)
k_sbp_resolve(SbpJobs JS, int forceSeq) {
    extern __shared__ __align__(16) uint8_t sbsm[];
    const SbpJobDev& J = JS.job[blockIdx.x];
    const uint32_t thHigh = JS.kf.thHigh ? (uint32_t)(JS.kf.thHigh - 1) : (uint32_t)SBP_TH_HIGH;
    const int M = JS.M;
    const int nLast = J.nLastPtr ? min(*J.nLastPtr, J.nLast) : J.nLast;
    const int nCur = J.nCurPtr ? min(*J.nCurPtr, J.nCur) : J.nCur;
    int32_t* holder = (int32_t*)sbsm; /* nCur: lowest-index blocking query pointing here; -1 = occupied from the start */
    int32_t* owner = holder + nCur;   /* nCur: highest-index query assigned here (mvpMapPoints: last writer wins) */
    __shared__ int s_changed, s_seq, s_nlog, s_removed;
    __shared__ int s_hist[SI_HISTO];
    const int tid = threadIdx.x;
    if (tid == 0) {
        s_seq = forceSeq;
        s_nlog = 0;
        s_removed = 0;
    }
    if (tid < SI_HISTO) s_hist[tid] = 0;
    for (int c = tid; c < nCur; c += SBP_RT) {
        holder[c] = (J.occupied0 && J.occupied0[c]) ? -1 : 0x7FFFFFFF;
        owner[c] = -1;
    }
    uint32_t key[SBP_QPT];
    int ptr[SBP_QPT];
    bool blocking[SBP_QPT];
#pragma unroll
    for (int k = 0; k < SBP_QPT; k++) {
        const int q = tid + k * SBP_RT;
        ptr[k] = 0;
        key[k] = q < nLast ? J.topm[(size_t)q * M] : 0xFFFFFFFFu;
        blocking[k] = q < nLast && (J.flags[q] & 2);
    }
    __syncthreads();
    for (int round = 0; round < 4096 && !s_seq; round++) { /* s_seq / s_changed are block-uniform at the barriers */
        if (tid == 0) s_changed = 0;
#pragma unroll
        for (int k = 0; k < SBP_QPT; k++)
            if (key[k] != 0xFFFFFFFFu && blocking[k] && (key[k] >> 24) <= thHigh)
                atomicMin(&holder[key[k] & 0xFFF], tid + k * SBP_RT);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SBP_QPT; k++) {
            const int q = tid + k * SBP_RT;
            if (key[k] != 0xFFFFFFFFu && holder[key[k] & 0xFFF] < q) { /* taken by an earlier query: next candidate */
                ptr[k]++;
                if (ptr[k] >= M) {
                    s_seq = 1; /* ran off a full prefix */
                    key[k] = 0xFFFFFFFFu;
                } else {
                    key[k] = J.topm[(size_t)q * M + ptr[k]]; /* ~0: the list is complete and exhausted */
                }
                s_changed = 1;
            }
        }
        __syncthreads();
        if (!s_changed) break;
        __syncthreads();
    }
    __syncthreads();
    if (tid == 0) *J.needSeq = s_seq;
    if (s_seq) return;
    /* assignments */
    uint32_t bin[SBP_QPT];
    const float factor = 1.0f / SI_HISTO;
#pragma unroll
    for (int k = 0; k < SBP_QPT; k++) {
        const int q = tid + k * SBP_RT;
        bin[k] = 255;
        if (key[k] != 0xFFFFFFFFu && (key[k] >> 24) <= thHigh) {
            const int i2 = (int)(key[k] & 0xFFF);
            atomicMax(&owner[i2], q);
            atomicAdd(&s_nlog, 1);
            bin[k] = 254;
            if (J.checkOri) {
                float rot = __fsub_rn(J.lastKps[q].angle, J.curKps[i2].angle);
                if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
                int b = (int)roundf(__fmul_rn(rot, factor));
                if (b == SI_HISTO) b = 0;
                bin[k] = (uint32_t)b;
                atomicAdd(&s_hist[b], 1);
            }
        }
    }
    __syncthreads();
    for (int c = tid; c < nCur; c += SBP_RT) J.matchCur[c] = owner[c];
    __syncthreads();
    if (J.checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0; /* ComputeThreeMaxima, fmatcher.cpp:2813-2854 */
        for (int i = 0; i < SI_HISTO; i++) {
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
#pragma unroll
        for (int k = 0; k < SBP_QPT; k++) {
            const int b = (int)bin[k];
            if (b < SI_HISTO && b != ind1 && b != ind2 && b != ind3) { /* fmatcher.cpp:2668-2681 */
                J.matchCur[key[k] & 0xFFF] = -1;
                atomicAdd(&s_removed, 1);
            }
        }
    }
    __syncthreads();
    if (tid == 0) J.nmatches[0] = s_nlog - s_removed;
}

__global__ void __launch_bounds__(64)
k_sbp_replay(SbpJobs JS, int* fallbacks) {
    extern __shared__ __align__(16) uint8_t sbsm[];
    const SbpJobDev& J = JS.job[blockIdx.x];
    const uint32_t thHigh = JS.kf.thHigh ? (uint32_t)(JS.kf.thHigh - 1) : (uint32_t)SBP_TH_HIGH;
    const int M = JS.M;
    const int nLast = J.nLastPtr ? min(*J.nLastPtr, J.nLast) : J.nLast;
    const int nCur = J.nCurPtr ? min(*J.nCurPtr, J.nCur) : J.nCur;
    const vslam_kp* __restrict__ lastKps = J.lastKps;
    const vslam_kp* __restrict__ curKps = J.curKps;
    const uint8_t* __restrict__ curDesc = J.curDesc;
    const uint8_t* __restrict__ mpDesc = J.mpDesc;
    const uint8_t* __restrict__ flags = J.flags;
    const float* __restrict__ uRight = J.uRight;
    const uint8_t* __restrict__ occupied0 = J.occupied0;
    const SbpProj* __restrict__ proj = J.proj;
    const uint32_t* __restrict__ topm = J.topm;
    int32_t* __restrict__ matchCur = J.matchCur;
    int32_t* __restrict__ nmatches_out = J.nmatches;
    struct { int checkOri, imgW, imgH; } A = {J.checkOri, J.imgW, J.imgH};
    uint32_t* occupied = (uint32_t*)sbsm;              /* nCur: mvpMapPoints[i2] with Observations() > 0 */
    int32_t* ownerEntry = (int32_t*)(occupied + nCur); /* nCur: last log entry that wrote mvpMapPoints[i2] */
    uint32_t* alog = (uint32_t*)(ownerEntry + nCur);   /* nLast: (query << 12 | i2), in order */
    uint8_t* logBin = (uint8_t*)(alog + nLast);        /* nLast: rotation bin of a log entry */
    uint32_t* kbuf = (uint32_t*)(sbsm + (((size_t)nCur * 8 + (size_t)nLast * 5 + 15) & ~(size_t)15)); /* 64 x M */
    __shared__ int s_hist[SI_HISTO];
    const int lane = threadIdx.x;
    if (*J.needSeq == 0) return; /* k_sbp_resolve finished this job */
    const float invW = __fdiv_rn((float)SI_GRID_COLS, (float)A.imgW);
    const float invH = __fdiv_rn((float)SI_GRID_ROWS, (float)A.imgH);
    for (int c = lane; c < nCur; c += 64) {
        occupied[c] = occupied0 ? occupied0[c] : 0u;
        ownerEntry[c] = -1;
        matchCur[c] = -1;
    }
    if (lane < SI_HISTO) s_hist[lane] = 0;
    __syncthreads();

    int nlog = 0, nfb = 0;
    for (int qb = 0; qb < nLast; qb += 64) {
        const int nq = min(64, nLast - qb);
        __syncthreads();
        for (int i = lane; i < nq * M; i += 64) kbuf[i] = topm[(size_t)qb * M + i];
        const uint32_t myflags = lane < nq ? flags[qb + lane] : 0u; /* lane tq holds query qb+tq's flags */
        __syncthreads();
        uint32_t keyN = lane < M ? kbuf[lane] : 0xFFFFFFFFu;
        uint32_t occN = keyN != 0xFFFFFFFFu ? occupied[keyN & 0xFFF] : 0u;
        for (int tq = 0; tq < nq; tq++) {
            const uint32_t key = keyN, occ = occN;
            if (tq + 1 < nq) {
                keyN = lane < M ? kbuf[(tq + 1) * M + lane] : 0xFFFFFFFFu;
                occN = keyN != 0xFFFFFFFFu ? occupied[keyN & 0xFFF] : 0u;
            }
            const bool valid = key != 0xFFFFFFFFu;
            const unsigned long long mv = __ballot(valid), mk = __ballot(valid && !occ);
            if (mv == 0) continue; /* vIndices2 empty, or every candidate failed an order-free gate */
            uint32_t gBest;
            if (mk) {
                const int first = __builtin_amdgcn_readfirstlane(__ffsll((long long)mk) - 1);
                gBest = (uint32_t)__builtin_amdgcn_readlane((int)key, first);
            } else if (__popcll(mv) < M) {
                continue; /* the list is complete and every entry is occupied: bestDist stays 256 */
            } else {
                nfb++;
                gBest = sbp_full_scan(proj[qb + tq], curKps, uRight, nCur, mpDesc + (size_t)(qb + tq) * 32, curDesc,
                                      occupied, invW, invH, lane);
                if (gBest == 0xFFFFFFFFu) continue;
            }
            if ((gBest >> 24) > thHigh) continue; /* bestDist <= TH_HIGH (ORBdist in the KeyFrame overload) */
            const uint32_t i2 = gBest & 0xFFF;
            const uint32_t fl = (uint32_t)__builtin_amdgcn_readlane((int)myflags, tq);
            if (lane == 0) {
                alog[nlog] = ((uint32_t)(qb + tq) << 12) | i2;
                if (fl & 2) occupied[i2] = 1u; /* pMP->Observations() > 0: later queries skip i2 */
            }
            nlog++;
            if ((fl & 2) && keyN != 0xFFFFFFFFu && (keyN & 0xFFF) == i2) occN = 1u;
        }
    }
    __syncthreads();
    if (fallbacks && lane == 0 && nfb) atomicAdd(fallbacks, nfb);
    /* mvpMapPoints[bestIdx2] = pMP: the last writer stays */
    for (int e = lane; e < nlog; e += 64) atomicMax(&ownerEntry[alog[e] & 0xFFF], e);
    __syncthreads();
    const float factor = 1.0f / SI_HISTO;
    for (int e = lane; e < nlog; e += 64) {
        const uint32_t le = alog[e];
        const int q = (int)(le >> 12), i2 = (int)(le & 0xFFF);
        if (ownerEntry[i2] == e) matchCur[i2] = q;
        if (A.checkOri) {
            float rot = __fsub_rn(lastKps[q].angle, curKps[i2].angle);
            if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
            int bin = (int)roundf(__fmul_rn(rot, factor));
            if (bin == SI_HISTO) bin = 0;
            logBin[e] = (uint8_t)bin;
            atomicAdd(&s_hist[bin], 1);
        }
    }
    __syncthreads();
    int removed = 0;
    if (A.checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0; /* ComputeThreeMaxima, fmatcher.cpp:2813-2854 */
        for (int i = 0; i < SI_HISTO; i++) {
            const int s = s_hist[i];
            if (s > max1) {
                max3 = max2; max2 = max1; max1 = s;
                ind3 = ind2; ind2 = ind1; ind1 = i;
            } else if (s > max2) {
                max3 = max2; max2 = s;
                ind3 = ind2; ind2 = i;
            } else if (s > max3) {
                max3 = s;
                ind3 = i;
            }
        }
        if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) { ind3 = -1; }
        /* every entry of a rejected bin clears mvpMapPoints[idx] and decrements nmatches (fmatcher.cpp:2668-2681) */
        for (int e = lane; e < nlog; e += 64) {
            const int b = logBin[e];
            if (b != ind1 && b != ind2 && b != ind3) {
                matchCur[alog[e] & 0xFFF] = -1;
                removed++;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) removed += __shfl_xor(removed, o, 64);
    if (lane == 0) nmatches_out[0] = nlog - removed;
}

size_t vk_sbp_rank_lds(int nCur) { return (size_t)nCur * sizeof(SbpCand); }
size_t vk_sbp_replay_lds(int nCur, int nLast) {
    return (((size_t)nCur * 8 + (size_t)nLast * 5 + 15) & ~(size_t)15) + 64 * SI_MAX_M * 4;
}
size_t vk_sbp_scratch_bytes(int nLast, int M) { return (size_t)nLast * (sizeof(SbpProj) + 4 * (size_t)M); }
size_t vk_sbp_proj_bytes(int nLast) { return (size_t)nLast * sizeof(SbpProj); }
__global__ void k_sbpm_resolve(SbpJobs JS, int forceSeq); /* defined below */
__global__ void k_fuse_rank(FuseArgsDev A);
int vk_sbp_set_max_lds(size_t bytes) {
    int rc = (int)hipFuncSetAttribute((const void*)k_sbp_resolve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (rc) return rc;
    rc = (int)hipFuncSetAttribute((const void*)k_sbpm_resolve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (rc) return rc;
    rc = (int)hipFuncSetAttribute((const void*)k_sbp_rank, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (rc) return rc;
    rc = (int)hipFuncSetAttribute((const void*)k_fuse_rank, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (rc) return rc;
    return (int)hipFuncSetAttribute((const void*)k_sbp_replay, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

/* ------------------------------------------------------------------------------------------------
 * SearchByProjection(Frame& F, const vector<MapPoint*>&, th, ...) (fmatcher.cpp:321-411): same ranking, but a
 * query now looks at its first TWO free candidates (best / second with their octaves, ratio test only when both
 * lie on the same level).  A query's proposal can therefore appear or vanish while the occupancy around it
 * changes, so the resolution is a plain fixpoint: every round rebuilds "lowest-index blocking query that chose
 * this keypoint" from the current choices and lets every query choose again; by induction on the query index the
 * unique fixpoint is the sequential result.  Same hand-over to a sequential wave when a list prefix runs out.
 * ---------------------------------------------------------------------------------------------- */
#define SBPM_M 8
__global__ void __launch_bounds__(SBP_RT)
k_sbpm_resolve(SbpJobs JS, int forceSeq) {
    extern __shared__ __align__(16) uint8_t sbsm[];
    const SbpJobDev& J = JS.job[blockIdx.x];
    const int M = min(JS.M, SBPM_M);
    const int nLast = J.nLastPtr ? min(*J.nLastPtr, J.nLast) : J.nLast;
    const int nCur = J.nCurPtr ? min(*J.nCurPtr, J.nCur) : J.nCur;
    int32_t* holder = (int32_t*)sbsm;
    int32_t* owner = holder + nCur;
    uint8_t* oct = (uint8_t*)(owner + nCur);
    __shared__ int s_changed, s_seq, s_nlog;
    const int tid = threadIdx.x;
    if (tid == 0) {
        s_seq = forceSeq;
        s_nlog = 0;
        s_changed = 0;
    }
    for (int c = tid; c < nCur; c += SBP_RT) {
        oct[c] = (uint8_t)J.curKps[c].octave;
        owner[c] = -1;
    }
    uint32_t keys[SBP_QPT][SBPM_M];
    int choice[SBP_QPT];
    bool blocking[SBP_QPT], full[SBP_QPT];
#pragma unroll
    for (int k = 0; k < SBP_QPT; k++) {
        const int q = tid + k * SBP_RT;
        choice[k] = -1;
        blocking[k] = q < nLast && (J.flags[q] & 2);
        full[k] = true;
#pragma unroll
        for (int j = 0; j < SBPM_M; j++) {
            keys[k][j] = (q < nLast && j < M) ? J.topm[(size_t)q * JS.M + j] : 0xFFFFFFFFu;
            if (j < M && keys[k][j] == 0xFFFFFFFFu) full[k] = false;
        }
    }
    __syncthreads();
    for (int round = 0; round < 1024 && !s_seq; round++) {
        for (int c = tid; c < nCur; c += SBP_RT) holder[c] = (J.occupied0 && J.occupied0[c]) ? -1 : 0x7FFFFFFF;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SBP_QPT; k++)
            if (choice[k] >= 0 && blocking[k]) atomicMin(&holder[choice[k]], tid + k * SBP_RT);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SBP_QPT; k++) {
            const int q = tid + k * SBP_RT;
            if (keys[k][0] == 0xFFFFFFFFu) continue; /* no candidates at all */
            int found = 0, best = -1, bd = 256, bl = -1, sd = 256, sl = -1;
#pragma unroll
            for (int j = 0; j < SBPM_M; j++) {
                const uint32_t key = keys[k][j];
                if (key == 0xFFFFFFFFu || found == 2) continue;
                const int c = (int)(key & 0xFFF);
                if (holder[c] < q) continue; /* held by an earlier query (or occupied from the start) */
                if (found == 0) {
                    best = c;
                    bd = (int)(key >> 24);
                    bl = oct[c];
                } else {
                    sd = (int)(key >> 24);
                    sl = oct[c];
                }
                found++;
            }
            int nc = -1;
            if (found == 0) {
                if (full[k]) s_seq = 1; /* free candidates may exist beyond the prefix */
            } else if (bd <= SBP_TH_HIGH) {
                if (found == 2) {
                    if (!(bl == sl && (float)bd > __fmul_rn(J.nnratio, (float)sd))) nc = best;
                } else if (!full[k]) {
                    nc = best; /* bestLevel2 stays -1 */
                } else {
                    uint32_t klast = 0xFFFFFFFFu;
#pragma unroll
                    for (int j = 0; j < SBPM_M; j++)
                        if (j == M - 1) klast = keys[k][j];
                    const int dlast = (int)(klast >> 24);
                    if ((float)bd <= __fmul_rn(J.nnratio, (float)dlast)) nc = best; /* whatever the second is */
                    else s_seq = 1;
                }
            }
            if (nc != choice[k]) {
                choice[k] = nc;
                s_changed = 1;
            }
        }
        __syncthreads();
        const int ch = s_changed;
        __syncthreads();
        if (tid == 0) s_changed = 0;
        if (!ch) break;
        if (round == 1023 && tid == 0) s_seq = 1;
        __syncthreads();
    }
    __syncthreads();
    if (tid == 0) *J.needSeq = s_seq;
    if (s_seq) return;
#pragma unroll
    for (int k = 0; k < SBP_QPT; k++)
        if (choice[k] >= 0) {
            atomicMax(&owner[choice[k]], tid + k * SBP_RT); /* F.mvpMapPoints[bestIdx] = pMP: last writer */
            atomicAdd(&s_nlog, 1);
        }
    __syncthreads();
    for (int c = tid; c < nCur; c += SBP_RT) J.matchCur[c] = owner[c];
    if (tid == 0) J.nmatches[0] = s_nlog;
}

/* sequential cross-check / fallback of the same function: one wave, MapPoints in index order, every window
 * scanned in full (two smallest free keys by two wave reductions) */
__global__ void __launch_bounds__(64)
k_sbpm_replay(SbpJobs JS, int* fallbacks) {
    extern __shared__ __align__(16) uint8_t sbsm[];
    const SbpJobDev& J = JS.job[blockIdx.x];
    if (*J.needSeq == 0) return;
    const int nLast = J.nLastPtr ? min(*J.nLastPtr, J.nLast) : J.nLast;
    const int nCur = J.nCurPtr ? min(*J.nCurPtr, J.nCur) : J.nCur;
    uint32_t* occupied = (uint32_t*)sbsm;
    const int lane = threadIdx.x;
    const float invW = __fdiv_rn((float)SI_GRID_COLS, (float)J.imgW);
    const float invH = __fdiv_rn((float)SI_GRID_ROWS, (float)J.imgH);
    for (int c = lane; c < nCur; c += 64) {
        occupied[c] = J.occupied0 ? J.occupied0[c] : 0u;
        J.matchCur[c] = -1;
    }
    __syncthreads();
    int nm = 0, nscan = 0;
    for (int q = 0; q < nLast; q++) {
        const SbpProj pr = J.proj[q];
        if (!pr.valid) continue;
        nscan++;
        const SiWindow win = si_window(pr.u, pr.v, pr.radius, invW, invH);
        if (win.empty) continue;
        const uint4 da = ((const uint4*)J.mpDesc)[(size_t)q * 2], db = ((const uint4*)J.mpDesc)[(size_t)q * 2 + 1];
        uint32_t k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu; /* this lane's two smallest free keys */
        for (int c = lane; c < nCur; c += 64) {
            bool ing;
            const SbpCand cd = sbp_make_cand(J.curKps[c], J.uRight ? J.uRight[c] : -1.f, invW, invH, &ing);
            if (!ing || occupied[c] || !sbp_candidate_ok(cd, win, pr)) continue;
            const uint4 ta = ((const uint4*)J.curDesc)[(size_t)c * 2], tb = ((const uint4*)J.curDesc)[(size_t)c * 2 + 1];
            const uint32_t key = (min(si_hamming(da, db, ta, tb), 255u) << 24) | ((uint32_t)cd.cell << 12) | (uint32_t)c;
            if (key < k1) { k2 = k1; k1 = key; }
            else if (key < k2) k2 = key;
        }
        const uint32_t g1 = wave_min_u32(k1);
        if (g1 == 0xFFFFFFFFu) continue;
        const uint32_t g2 = wave_min_u32(k1 == g1 ? k2 : k1);
        const int bd = (int)(g1 >> 24), best = (int)(g1 & 0xFFF);
        if (bd > SBP_TH_HIGH) continue;
        const int bl = J.curKps[best].octave;
        bool accept = true;
        if (g2 != 0xFFFFFFFFu) {
            const int sd = (int)(g2 >> 24), sl = J.curKps[g2 & 0xFFF].octave;
            if (bl == sl && (float)bd > __fmul_rn(J.nnratio, (float)sd)) accept = false;
        }
        if (!accept) continue;
        if (lane == 0) {
            J.matchCur[best] = q;
            if (J.flags[q] & 2) occupied[best] = 1u;
        }
        nm++;
        __syncthreads();
    }
    __syncthreads();
    if (lane == 0) {
        J.nmatches[0] = nm;
        if (fallbacks && nscan) atomicAdd(fallbacks, nscan);
    }
}

size_t vk_sbpm_resolve_lds(int nCur) { return (size_t)nCur * 9 + 16; }

size_t vk_sbp_resolve_lds(int nCur) { return (size_t)nCur * 8; }

void vk_search_by_projection(hipStream_t st, const SbpJobs& JS, int njobs, int maxLast, int maxCur, int* fallbacks,
                             int forceSeq) {
    if (njobs <= 0) return;
    if (maxLast > 0)
        hipLaunchKernelGGL(k_sbp_rank, dim3((maxLast + SBP_QPB - 1) / SBP_QPB, njobs), dim3(256),
                           vk_sbp_rank_lds(maxCur), st, JS);
    if (JS.job[0].mode == 1) {
        hipLaunchKernelGGL(k_sbpm_resolve, dim3(njobs), dim3(SBP_RT), vk_sbpm_resolve_lds(maxCur), st, JS, forceSeq);
        hipLaunchKernelGGL(k_sbpm_replay, dim3(njobs), dim3(64), (size_t)maxCur * 4, st, JS, fallbacks);
        return;
    }
    hipLaunchKernelGGL(k_sbp_resolve, dim3(njobs), dim3(SBP_RT), vk_sbp_resolve_lds(maxCur), st, JS, forceSeq);
    hipLaunchKernelGGL(k_sbp_replay, dim3(njobs), dim3(64), vk_sbp_replay_lds(maxCur, maxLast), st, JS, fallbacks);
}

/* ------------------------------------------------------------------------------------------------
 * Frame::UnprojectStereo (frame.cpp:1023-1037) for every left keypoint of up to VSLAM_MAX_SBP_JOBS stereo
 * pairs: z = mvDepth[i] > 0  ->  x3Dc = ((u-cx)*z*invfx, (v-cy)*z*invfy, z), x3Dw = mRwc*x3Dc + mOw (one
 * cv::gemm, see sbp_gemm_row).  These are the points UpdateLastFrame turns into (temporal) MapPoints for
 * TrackWithMotionModel, i.e. the last-frame side of SearchByProjection; flags = has-a-point | observations.
 * ---------------------------------------------------------------------------------------------- */
__global__ void __launch_bounds__(256)
k_unproject_stereo(UnprojJobs U) {
    const UnprojJob& J = U.job[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n = min(*J.nPtr, U.cap);
    if (i >= U.cap) return;
    float X = 0.f, Y = 0.f, Z = 0.f;
    uint8_t fl = 0;
    if (i < n) {
        const float z = J.depth[i];
        if (z > 0.f) {
            const vslam_kp k = J.kps[i];
            const float x = __fmul_rn(__fmul_rn(__fsub_rn(k.x, U.cx), z), U.invfx);
            const float y = __fmul_rn(__fmul_rn(__fsub_rn(k.y, U.cy), z), U.invfy);
            X = sbp_gemm_row(J.Twc + 0, x, y, z, J.Twc[3], U.gemmFloat);
            Y = sbp_gemm_row(J.Twc + 4, x, y, z, J.Twc[7], U.gemmFloat);
            Z = sbp_gemm_row(J.Twc + 8, x, y, z, J.Twc[11], U.gemmFloat);
            fl = (uint8_t)(1 | (U.observations ? 2 : 0));
        }
    }
    J.x3Dw[3 * i] = X;
    J.x3Dw[3 * i + 1] = Y;
    J.x3Dw[3 * i + 2] = Z;
    J.flags[i] = fl;
}

void vk_unproject_stereo(hipStream_t st, const UnprojJobs& U, int njobs) {
    if (njobs <= 0) return;
    hipLaunchKernelGGL(k_unproject_stereo, dim3((U.cap + 255) / 256, njobs), dim3(256), 0, st, U);
}

/* ------------------------------------------------------------------------------------------------
 * The search half of FMatcher::Fuse (fmatcher.cpp:1918-2119 with bRight = false; Sim3 overload :2121-2243):
 * per MapPoint  Rcw*p + tcw (one cv::gemm, sbp_gemm_row), depth >= 0, Pinhole::project, KeyFrame::IsInImage,
 * cv::norm(p - Ow) inside the scale-invariance range, viewing angle (cv::Mat::dot, both accumulate in double),
 * MapPoint::PredictScale (mappoint.cpp:506-521: ceil(logf(max/dist) / mfLogScaleFactor), glibc logf), radius =
 * th * scale[level], KeyFrame::GetFeaturesInArea (keyframe.cpp:656-699), level gate, chi2 gate (7.8 with a right
 * coordinate, 5.99 without; not in the Sim3 overload), then the least Hamming distance, first of the window order
 * wins: key = dist << 24 | cell << 12 | index -> wave min.  Nothing here depends on other MapPoints; the map
 * mutation that follows in the reference (Replace / AddObservation / vpReplacePoint) stays with the caller.
 * One wave per MapPoint, the KeyFrame's keypoints in LDS as for SearchByProjection.
 * ---------------------------------------------------------------------------------------------- */
#define FUSE_QPB 16
__global__ void __launch_bounds__(256)
k_fuse_rank(FuseArgsDev A) {
    extern __shared__ __align__(16) uint8_t sbsm[];
    SbpCand* cand = (SbpCand*)sbsm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float invW = __fdiv_rn((float)SI_GRID_COLS, (float)A.imgW);
    const float invH = __fdiv_rn((float)SI_GRID_ROWS, (float)A.imgH);
    for (int i = tid; i < A.nKF; i += 256) {
        bool ing;
        cand[i] = sbp_make_cand(A.kfKps[i], A.kfURight[i], invW, invH, &ing);
    }
    __syncthreads();
    for (int q = blockIdx.x * FUSE_QPB + wave; q < min(A.nPoints, (int)(blockIdx.x + 1) * FUSE_QPB); q += 4) {
        const FusePoint mp = A.pts[q];
        uint32_t best = 0xFFFFFFFFu;
        bool go = mp.valid != 0;
        float u = 0.f, v = 0.f, ur = 0.f, radius = 0.f;
        int level = 0;
        if (go && A.sim3 == 2) { /* SearchBySim3, one direction (fmatcher.cpp:2301-2332 / :2381-2412) */
            const float x1 = sbp_gemm_row(A.Rcw + 0, mp.pos[0], mp.pos[1], mp.pos[2], A.tcw[0], A.gemmFloat);
            const float y1 = sbp_gemm_row(A.Rcw + 3, mp.pos[0], mp.pos[1], mp.pos[2], A.tcw[1], A.gemmFloat);
            const float z1 = sbp_gemm_row(A.Rcw + 6, mp.pos[0], mp.pos[1], mp.pos[2], A.tcw[2], A.gemmFloat);
            const float xc = sbp_gemm_row(A.Rb + 0, x1, y1, z1, A.tb[0], A.gemmFloat);
            const float yc = sbp_gemm_row(A.Rb + 3, x1, y1, z1, A.tb[1], A.gemmFloat);
            const float zc = sbp_gemm_row(A.Rb + 6, x1, y1, z1, A.tb[2], A.gemmFloat);
            go = !(zc < 0.0f);
            if (go) {
                const float invz = (float)__ddiv_rn(1.0, (double)zc); /* const float invz = 1.0/z */
                u = __fadd_rn(__fmul_rn(A.fx, __fmul_rn(xc, invz)), A.cx);
                v = __fadd_rn(__fmul_rn(A.fy, __fmul_rn(yc, invz)), A.cy);
                go = u >= 0.0f && u < (float)A.imgW && v >= 0.0f && v < (float)A.imgH;
            }
            if (go) {
                double n2 = __dmul_rn((double)xc, (double)xc); /* cv::norm(p3Dc2) */
                n2 = __dadd_rn(n2, __dmul_rn((double)yc, (double)yc));
                n2 = __dadd_rn(n2, __dmul_rn((double)zc, (double)zc));
                const float dist3D = (float)__dsqrt_rn(n2);
                go = !(dist3D < mp.minDistance || dist3D > mp.maxDistance);
                if (go) {
                    const float lv = ceilf(__fdiv_rn(vslam_trig::glibc_logf(__fdiv_rn(mp.maxDistance, dist3D)), A.logScaleFactor));
                    level = (lv != lv || lv >= 2147483648.0f || lv < 0.f) ? 0 : min((int)lv, A.nlevels - 1);
                    radius = __fmul_rn(A.th, A.scale[level]);
                }
            }
        } else if (go) { /* wave-uniform */
            const float xc = sbp_gemm_row(A.Rcw + 0, mp.pos[0], mp.pos[1], mp.pos[2], A.tcw[0], A.gemmFloat);
            const float yc = sbp_gemm_row(A.Rcw + 3, mp.pos[0], mp.pos[1], mp.pos[2], A.tcw[1], A.gemmFloat);
            const float zc = sbp_gemm_row(A.Rcw + 6, mp.pos[0], mp.pos[1], mp.pos[2], A.tcw[2], A.gemmFloat);
            go = !(zc < 0.0f);
            if (go) {
                const float invz = __fdiv_rn(1.0f, zc);
                u = __fadd_rn(__fdiv_rn(__fmul_rn(A.fx, xc), zc), A.cx);
                v = __fadd_rn(__fdiv_rn(__fmul_rn(A.fy, yc), zc), A.cy);
                go = u >= 0.0f && u < (float)A.imgW && v >= 0.0f && v < (float)A.imgH;
                ur = __fsub_rn(u, __fmul_rn(A.bf, invz));
            }
            if (go) {
                const float p0 = __fsub_rn(mp.pos[0], A.Ow[0]), p1 = __fsub_rn(mp.pos[1], A.Ow[1]),
                            p2 = __fsub_rn(mp.pos[2], A.Ow[2]);
                double n2 = __dmul_rn((double)p0, (double)p0);
                n2 = __dadd_rn(n2, __dmul_rn((double)p1, (double)p1));
                n2 = __dadd_rn(n2, __dmul_rn((double)p2, (double)p2));
                const float dist3D = (float)__dsqrt_rn(n2);
                go = !(dist3D < mp.minDistance || dist3D > mp.maxDistance);
                if (go) {
                    double dot = __dmul_rn((double)p0, (double)mp.normal[0]);
                    dot = __dadd_rn(dot, __dmul_rn((double)p1, (double)mp.normal[1]));
                    dot = __dadd_rn(dot, __dmul_rn((double)p2, (double)mp.normal[2]));
                    go = !(dot < __dmul_rn(0.5, (double)dist3D));
                }
                if (go) {
                    const float ratio = __fdiv_rn(mp.maxDistance, dist3D);
                    /* NaN and out-of-range quotients convert to INT_MIN on the reference's x86-64 build
                     * (cvttss2si) and so clamp to level 0 */
                    const float lv = ceilf(__fdiv_rn(vslam_trig::glibc_logf(ratio), A.logScaleFactor));
                    level = (lv != lv || lv >= 2147483648.0f || lv < 0.f) ? 0 : min((int)lv, A.nlevels - 1);
                    radius = __fmul_rn(A.th, A.scale[level]);
                }
            }
        }
        if (go) {
            const SiWindow win = si_window(u, v, radius, invW, invH);
            if (!win.empty) {
                const uint4 da = ((const uint4*)A.mpDesc)[(size_t)q * 2], db = ((const uint4*)A.mpDesc)[(size_t)q * 2 + 1];
                for (int c = lane; c < A.nKF; c += 64) {
                    const SbpCand cd = cand[c];
                    if (cd.cell == 0xFFFF) continue;
                    const int gx = cd.cell >> 6, gy = cd.cell & 63;
                    if (gx < win.minX || gx > win.maxX || gy < win.minY || gy > win.maxY) continue;
                    const float ex = __fsub_rn(cd.x, u), ey = __fsub_rn(cd.y, v);
                    if (!(fabsf(ex) < radius && fabsf(ey) < radius)) continue;
                    const int kpLevel = cd.octave;
                    if (kpLevel < level - 1 || kpLevel > level) continue;
                    if (!A.sim3) {
                        /* ex = uv.x - kpx in the reference; only its square is used */
                        const float is2 = A.invSigma2[min(kpLevel, A.nlevels - 1)];
                        const float exx = __fsub_rn(u, cd.x), eyy = __fsub_rn(v, cd.y);
                        if (cd.uRight >= 0.f) {
                            const float er = __fsub_rn(ur, cd.uRight);
                            const float e2 = __fadd_rn(__fadd_rn(__fmul_rn(exx, exx), __fmul_rn(eyy, eyy)), __fmul_rn(er, er));
                            if ((double)__fmul_rn(e2, is2) > 7.8) continue;
                        } else {
                            const float e2 = __fadd_rn(__fmul_rn(exx, exx), __fmul_rn(eyy, eyy));
                            if ((double)__fmul_rn(e2, is2) > 5.99) continue;
                        }
                    }
                    const uint4 ta = ((const uint4*)A.kfDesc)[(size_t)c * 2], tb = ((const uint4*)A.kfDesc)[(size_t)c * 2 + 1];
                    best = min(best, (min(si_hamming(da, db, ta, tb), 255u) << 24) | ((uint32_t)cd.cell << 12) | (uint32_t)c);
                }
            }
        }
        const uint32_t g = wave_min_u32(best);
        if (lane == 0) {
            A.bestIdx[q] = g == 0xFFFFFFFFu ? -1 : (int32_t)(g & 0xFFFu);
            A.bestDist[q] = g == 0xFFFFFFFFu ? 256 : (int32_t)(g >> 24);
        }
    }
}

void vk_fuse_search(hipStream_t st, const FuseArgsDev& A) {
    if (A.nPoints <= 0) return;
    hipLaunchKernelGGL(k_fuse_rank, dim3((A.nPoints + FUSE_QPB - 1) / FUSE_QPB), dim3(256), vk_sbp_rank_lds(A.nKF), st, A);
}
