/* vslam_init_kernel.hip -- FMatcher::SearchForInitialization (fmatcher.cpp:983-1098) entirely on the GPU.
 *
 * The reference loop is sequential over the octave-0 keypoints of frame 1 (a later query may steal an
 * earlier query's match, and a candidate whose current owner is at least as close is skipped), so ONE WAVE
 * owns a frame pair and walks the queries in index order.  Everything inside one query is order-free once
 * each candidate carries the position it would have in Frame::GetFeaturesInArea's output (grid cell
 * column-major, ascending index inside a cell; frame.cpp:712-741): with key = dist << 28 | cell << 16 | slot
 *   bestDist/bestIdx2 = min key over the non-skipped window candidates ("first wins" on equal distance),
 *   bestDist2         = smallest distance among the remaining ones (multiset second minimum),
 * both wave min-reductions.  Acceptance (TH_LOW, ratio test in float), stealing, the 30-bin rotation
 * histogram and ComputeThreeMaxima follow the reference literally.
 *
 * Frame 2's octave-0 keypoints (position, grid cell, owner, owner distance) live in LDS; descriptors are
 * read from HBM (L2) only for candidates inside the window.
 */
#include "vslam_kernels.h"

#define SI_TH_LOW 50
#define SI_HISTO 30
#define SI_GRID_COLS 64 /* FRAME_GRID_COLS, frame.h:42 */
#define SI_GRID_ROWS 48 /* FRAME_GRID_ROWS, frame.h:43 */

/* wave64 min-reduction on DPP (no LDS traffic, a few cycles per step instead of a ds_bpermute round trip):
 * quad swaps, row rotations, then row_bcast:15 / row_bcast:31; the result sits in lane 63 and is returned
 * wave-uniform through v_readlane. */
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    v = min(v, dpp_mov<0xb1, 0xf>(v));  /* quad_perm [1,0,3,2] */
    v = min(v, dpp_mov<0x4e, 0xf>(v));  /* quad_perm [2,3,0,1] */
    v = min(v, dpp_mov<0x124, 0xf>(v)); /* row_ror:4 */
    v = min(v, dpp_mov<0x128, 0xf>(v)); /* row_ror:8 */
    v = min(v, dpp_mov<0x142, 0xa>(v)); /* row_bcast:15 into rows 1,3 */
    v = min(v, dpp_mov<0x143, 0xc>(v)); /* row_bcast:31 into rows 2,3 */
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

struct SiCand { /* one octave-0 keypoint of frame 2 */
    float x, y, angle;
    uint16_t cell; /* gridX * 64 + gridY (gridY < 48) */
    uint16_t idx;  /* index in frame 2 */
};
struct SiQuery { /* one octave-0 keypoint of frame 1 */
    float px, py, angle; /* vbPrevMatched position, keypoint angle */
    uint32_t idx;
};

__global__ void __launch_bounds__(64)
k_search_init(InitJobs jobs, int cap, int imgW, int imgH, int window, float nnratio, int checkOri,
              int32_t* matches_out /* [pair][cap] */, float* prev_out /* [pair][2*cap] */,
              int32_t* nmatch_out /* [pair] */, int max_c2, int lds_desc) {
    extern __shared__ __align__(16) uint8_t sism[];
    const InitJob jb = jobs.job[blockIdx.x];
    const int lane = threadIdx.x;
    const int n1 = min(*jb.cnt1, cap), n2 = min(*jb.cnt2, cap);
    /* LDS: [desc2 | desc1] (only if lds_desc) | cand | query | owner | ownerDist | m12 | rotBin */
    uint4* ldesc2 = (uint4*)sism;                                      /* max_c2 x 32 B */
    uint4* ldesc1 = ldesc2 + (lds_desc ? 2 * max_c2 : 0);              /* max_c2 x 32 B */
    SiCand* cand = (SiCand*)(ldesc1 + (lds_desc ? 2 * max_c2 : 0));    /* max_c2 */
    SiQuery* qry = (SiQuery*)(cand + max_c2);                          /* max_c2 (frame 1 has the same quota) */
    int32_t* owner = (int32_t*)(qry + max_c2);                         /* vnMatches21 (query index) */
    int32_t* ownerDist = owner + max_c2;                               /* vMatchedDistance */
    int32_t* m12 = ownerDist + max_c2;                                 /* vnMatches12, cap */
    uint8_t* rotBin = (uint8_t*)(m12 + cap);                           /* bin of an accepted query, 255 = none */
    __shared__ int s_hist[SI_HISTO];

    int32_t* mo = matches_out + (size_t)blockIdx.x * cap;
    float* po = prev_out + (size_t)blockIdx.x * cap * 2;

    /* Frame grid of frame 2 (frame.cpp:322-323, 746-756): mnMinX = 0, mnMaxX = cols (no distortion) */
    const float invW = __fdiv_rn((float)SI_GRID_COLS, (float)imgW);
    const float invH = __fdiv_rn((float)SI_GRID_ROWS, (float)imgH);

    /* ---- compaction of the octave-0 keypoints of both frames, order preserved (wave scan per 64) */
    int c2 = 0;
    for (int b = 0; b < n2; b += 64) {
        const int i = b + lane;
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
        if (take) {
            const int pos = c2 + __popcll(m & ((1ull << lane) - 1ull));
            if (pos < max_c2) {
                cand[pos].x = k.x;
                cand[pos].y = k.y;
                cand[pos].angle = k.angle;
                cand[pos].cell = (uint16_t)(gx * 64 + gy);
                cand[pos].idx = (uint16_t)i;
                owner[pos] = -1;
                ownerDist[pos] = 0x7FFFFFFF;
                if (lds_desc) {
                    ldesc2[2 * pos] = ((const uint4*)jb.d2)[(size_t)i * 2];
                    ldesc2[2 * pos + 1] = ((const uint4*)jb.d2)[(size_t)i * 2 + 1];
                }
            }
        }
        c2 += __popcll(m);
    }
    c2 = min(c2, max_c2);
    int c1 = 0;
    for (int b = 0; b < n1; b += 64) {
        const int i = b + lane;
        vslam_kp k;
        bool take = false;
        if (i < n1) {
            k = jb.k1[i];
            take = k.octave == 0; /* level1 > 0 -> continue, fmatcher.cpp:999-1001 */
        }
        const unsigned long long m = __ballot(take);
        if (take) {
            const int pos = c1 + __popcll(m & ((1ull << lane) - 1ull));
            if (pos < max_c2) {
                qry[pos].px = jb.prev ? jb.prev[2 * i] : k.x;
                qry[pos].py = jb.prev ? jb.prev[2 * i + 1] : k.y;
                qry[pos].angle = k.angle;
                qry[pos].idx = (uint32_t)i;
                if (lds_desc) {
                    ldesc1[2 * pos] = ((const uint4*)jb.d1)[(size_t)i * 2];
                    ldesc1[2 * pos + 1] = ((const uint4*)jb.d1)[(size_t)i * 2 + 1];
                }
            }
        }
        c1 += __popcll(m);
    }
    c1 = min(c1, max_c2);
    for (int i = lane; i < n1; i += 64) {
        m12[i] = -1;
        rotBin[i] = 255;
    }
    if (lane < SI_HISTO) s_hist[lane] = 0;
    __syncthreads();

    const float r = (float)window;
    const float factor = 1.0f / SI_HISTO;
    for (int t = 0; t < c1; t++) {
        const SiQuery qq = qry[t];
        const int i1 = (int)qq.idx;
        const float px = qq.px, py = qq.py;
        /* GetFeaturesInArea(x, y, r, 0, 0): cell range (frame.cpp:686-708) */
        const int nMinCellX = max(0, (int)floorf(__fmul_rn(__fsub_rn(px, r), invW)));
        const int nMaxCellX = min(SI_GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(px, r), invW)));
        const int nMinCellY = max(0, (int)floorf(__fmul_rn(__fsub_rn(py, r), invH)));
        const int nMaxCellY = min(SI_GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(py, r), invH)));
        if (nMinCellX >= SI_GRID_COLS || nMaxCellX < 0 || nMinCellY >= SI_GRID_ROWS || nMaxCellY < 0) continue;
        uint4 da, db;
        if (lds_desc) {
            da = ldesc1[2 * t];
            db = ldesc1[2 * t + 1];
        } else {
            da = ((const uint4*)jb.d1)[(size_t)i1 * 2];
            db = ((const uint4*)jb.d1)[(size_t)i1 * 2 + 1];
        }

        /* key = min(dist,63) << 24 | cell << 12 | slot: distances above TH_LOW are rejected anyway, so the
         * clamp cannot change an accepted match; true distances are tracked beside the key */
        uint32_t bestKey = 0xFFFFFFFFu, bestD = 0x7FFFFFFFu;
        uint32_t second = 0x7FFFFFFFu; /* smallest distance among this lane's other candidates */
        for (int c = lane; c < c2; c += 64) {
            const SiCand cd = cand[c];
            const int gx = cd.cell >> 6, gy = cd.cell & 63;
            if (gx < nMinCellX || gx > nMaxCellX || gy < nMinCellY || gy > nMaxCellY) continue;
            const float distx = __fsub_rn(cd.x, px), disty = __fsub_rn(cd.y, py);
            if (!(fabsf(distx) < r && fabsf(disty) < r)) continue;
            uint4 ta, tb;
            if (lds_desc) {
                ta = ldesc2[2 * c];
                tb = ldesc2[2 * c + 1];
            } else {
                ta = ((const uint4*)jb.d2)[(size_t)cd.idx * 2];
                tb = ((const uint4*)jb.d2)[(size_t)cd.idx * 2 + 1];
            }
            const uint32_t dist = __popc(da.x ^ ta.x) + __popc(da.y ^ ta.y) + __popc(da.z ^ ta.z) +
                                  __popc(da.w ^ ta.w) + __popc(db.x ^ tb.x) + __popc(db.y ^ tb.y) +
                                  __popc(db.z ^ tb.z) + __popc(db.w ^ tb.w);
            if ((uint32_t)ownerDist[c] <= dist) continue; /* vMatchedDistance[i2] <= dist, fmatcher.cpp:1022 */
            const uint32_t key = (min(dist, 63u) << 24) | ((uint32_t)cd.cell << 12) | (uint32_t)c;
            /* compare on (true distance, grid order): identical to the key order whenever dist < 63 */
            if (dist < bestD || (dist == bestD && key < bestKey)) {
                second = min(second, bestD);
                bestKey = key;
                bestD = dist;
            } else {
                second = min(second, dist);
            }
        }
        const uint32_t gBest = wave_min_u32(bestKey);
        if (gBest == 0xFFFFFFFFu) continue; /* vIndices2 empty or everything skipped: bestDist = INT_MAX */
        if ((gBest >> 24) > SI_TH_LOW) continue; /* bestDist > TH_LOW (clamped distances are all > 50) */
        /* second minimum over the multiset: lanes that do not hold the winner contribute their own best */
        uint32_t contrib = second;
        if (bestKey != gBest) contrib = min(contrib, bestD);
        const uint32_t bestDist2 = wave_min_u32(contrib); /* 0x7FFFFFFF == INT_MAX when there is none */
        const int bestDist = (int)(gBest >> 24);
        const int slot2 = (int)(gBest & 0xFFF);
        if (bestDist <= SI_TH_LOW && (float)bestDist < __fmul_rn((float)(int)bestDist2, nnratio)) {
            /* wave-uniform branch; lane 0 updates the shared state */
            if (lane == 0) {
                const int prevOwner = owner[slot2];
                if (prevOwner >= 0) m12[prevOwner] = -1;
                m12[i1] = cand[slot2].idx;
                owner[slot2] = i1;
                ownerDist[slot2] = bestDist;
                if (checkOri) {
                    float rot = __fsub_rn(qq.angle, cand[slot2].angle);
                    if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
                    int bin = (int)roundf(__fmul_rn(rot, factor));
                    if (bin == SI_HISTO) bin = 0;
                    rotBin[i1] = (uint8_t)bin;
                    s_hist[bin]++; /* rotHist[bin].push_back(i1): never removed, even if stolen later */
                }
            }
            __syncthreads(); /* one wave: orders lane 0's LDS updates before the next query's reads */
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
        for (int i = lane; i < n1; i += 64) {
            const int b = rotBin[i];
            if (b != 255 && b != ind1 && b != ind2 && b != ind3) m12[i] = -1;
        }
        __syncthreads();
    }
    int cnt = 0;
    for (int i = lane; i < n1; i += 64) {
        const int m = m12[i];
        mo[i] = m;
        /* vbPrevMatched update (fmatcher.cpp:1093-1095) */
        float ox = jb.prev ? jb.prev[2 * i] : jb.k1[i].x, oy = jb.prev ? jb.prev[2 * i + 1] : jb.k1[i].y;
        if (m >= 0) {
            ox = jb.k2[m].x;
            oy = jb.k2[m].y;
            cnt++;
        }
        po[2 * i] = ox;
        po[2 * i + 1] = oy;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if (lane == 0) nmatch_out[blockIdx.x] = cnt;
}

size_t vk_search_init_lds(int cap, int max_c2, int lds_desc) {
    return (size_t)max_c2 * (sizeof(SiCand) + sizeof(SiQuery) + 8 + (lds_desc ? 64 : 0)) + (size_t)cap * (4 + 1) + 64;
}

int vk_search_init_set_max_lds(size_t bytes) {
    return (int)hipFuncSetAttribute((const void*)k_search_init, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

void vk_search_init(hipStream_t st, const InitJobs& jobs, int npairs, int cap, int imgW, int imgH, int window,
                    float nnratio, int checkOri, int32_t* matches_out, float* prev_out, int32_t* nmatch_out,
                    int max_c2, int lds_desc) {
    if (npairs <= 0) return;
    hipLaunchKernelGGL(k_search_init, dim3(npairs), dim3(64), vk_search_init_lds(cap, max_c2, lds_desc), st, jobs, cap,
                       imgW, imgH, window, nnratio, checkOri, matches_out, prev_out, nmatch_out, max_c2, lds_desc);
}
