/* vslam_octree_kernel.hip -- FExtractor::DistributeOctTree (fextractor.cpp:530-754) on the GPU.
 *
 * One workgroup (256, 512 or 1024 threads: vk_octree) per (image slot, pyramid level).  The reference algorithm is a sequential
 * walk over a std::list, but every pass of it splits a whole generation of nodes, and the only
 * order-dependent facts are (a) the relative order of the keys inside a node (DivideNode keeps it),
 * (b) the order of the list and (c) the "largest node first, stop at N" rule.  All three are prefix sums:
 *
 *  - keys stay where the gather put them and carry a node label; a split needs the per-child key COUNT of every
 *    expandable node (a histogram) and then relabels each key; the relative order of the keys inside a node only
 *    matters for the final "first key with the maximal response" = arg-max on (response, ~position);
 *  - children are push_front'ed in creation order and survivors keep their relative order, so after a
 *    pass   list = reverse(children in creation order) ++ survivors.  Nodes are stored IN LIST ORDER and
 *    rebuilt each pass from two scans (children created before me / survivors before me);
 *  - phase 2 (fextractor.cpp:664-729) sorts the expandable nodes by (size, node address) and stops once
 *    lNodes.size() >= N: rank by counting, prefix sum of (children-1) in that order, cut where the
 *    running list size reaches N.  Heap addresses are not reproducible; the tie-break is "created later
 *    first" == smaller list index first (same rule as the CPU oracle and vslam_host.cpp).
 *
 * Integer arithmetic only, except the initial bucket index (int)(x / hX) which is an IEEE float division
 * exactly as in the reference (fextractor.cpp:560).
 */
#include "vslam_kernels.h"
#include "vslam_wave.h"

#include <mutex>

#ifndef OT
#define OT 1024
#endif
#define OKPT 16  /* k_octree_v2: keys per thread kept in registers (problems up to 16384 keys) */
#define OBATCH 8 /* k_octree_v2: keys per thread and batch when streaming a larger problem */
#ifndef O4BATCH
#define O4BATCH 8 /* k_octree_v4: keys per thread and batch of the streaming walks (loads in flight per lane) */
#endif
typedef unsigned long long u64;

struct ONode { /* 16 bytes, one entry of the list */
    int16_t x0, y0, x1, y1;
    uint32_t begin; /* first key of the node in the key array */
    uint32_t cf;    /* count << 1 | noMore */
};
#define ND_COUNT(nd) ((nd).cf >> 1)
#define ND_NOMORE(nd) ((nd).cf & 1u)

/* packed per-quadrant counters: quadrants 0..2 in three 21-bit fields of a u64 (up to 2M keys per level);
 * quadrant 3 is what remains of the node */
#define FB 21
#define FMASK ((1ull << FB) - 1)
__device__ __forceinline__ unsigned long long onehot(int q) { return q < 3 ? 1ull << (FB * q) : 0ull; }
__device__ __forceinline__ uint32_t fld(unsigned long long v, int q) { return (uint32_t)((v >> (FB * q)) & FMASK); }
/* number of non-empty children given the packed counts of quadrants 0..2 and the node's key count */
__device__ __forceinline__ uint32_t nchildren(unsigned long long c, uint32_t count) {
    const uint32_t c0 = fld(c, 0), c1 = fld(c, 1), c2 = fld(c, 2);
    return (c0 != 0) + (c1 != 0) + (c2 != 0) + (count - c0 - c1 - c2 != 0);
}

template <typename T, int NT = OT>
__device__ __forceinline__ T block_excl_scan(T v, T* s_wave, T* total) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    T inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const T t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) s_wave[wv] = inc;
    __syncthreads();
    T woff = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < NT / 64; k++) {
        const T x = s_wave[k];
        if (k < wv) woff += x;
        tot += x;
    }
    __syncthreads();
    *total = tot;
    return woff + inc - v;
}

__device__ __forceinline__ int quadrant(uint32_t pt, const ONode& nd) {
    /* DivideNode, fextractor.cpp:474-517: halfX = ceil((UR.x-UL.x)/2); kp.x < n1.UR.x, kp.y < n1.BR.y */
    const int x = pt & 0xFFF, y = (pt >> 12) & 0xFFF;
    const int mx = nd.x0 + ((nd.x1 - nd.x0 + 1) >> 1), my = nd.y0 + ((nd.y1 - nd.y0 + 1) >> 1);
    return (x < mx ? 0 : 1) | (y < my ? 0 : 2);
}

/* ------------------------------------------------------------------------------------------------
 * The keys never move.  What a pass needs from the keys is only
 * the per-child COUNT of every expandable node (which children exist, which can still be split, the sort key of
 * phase 2) -- a histogram (64-bit LDS atomics on the packed counters) -- and afterwards the keys' new node
 * labels.  The final "best response, first wins" is a segmented arg-max on (response, ~position).  All key
 * walks are strided by the workgroup size, i.e. coalesced (the first generation, a scan-based stable partition
 * of the key array per pass, gave each thread a contiguous chunk -- 64 cache lines per wave load, 5.6 ms per
 * launch at 1080p / 100 k candidates; it is in the git history).
 * ---------------------------------------------------------------------------------------------- */
/* what one (slot, level) problem of the walk-per-pass distribution needs from OctParams: plain values, so that the body
 * is independent of the parameter block's layout */
struct OctWalkLevel {
    int32_t N, H, nIni, c0, c1, selOff, selStride, maxNodes, ptsCap, maxIter;
    float hX;
    void* dbg;
};
__device__ __forceinline__ OctWalkLevel oct_walk_level(const OctParams& P, int level) {
    OctWalkLevel L;
    L.N = P.N[level]; L.H = P.H[level]; L.nIni = P.nIni[level]; L.hX = P.hX[level];
    L.c0 = P.cellFirst[level]; L.c1 = P.cellFirst[level + 1];
    L.selOff = P.selOff[level]; L.selStride = P.selStride; L.maxNodes = P.maxNodes; L.ptsCap = P.ptsCap;
    L.maxIter = P.maxIter; L.dbg = P.dbg;
    return L;
}

/* osm: the workgroup's dynamic LDS; s_w32[NT / 64] and s_ctl[4]: scratch words in LDS (scan partials; size, M, nexp, cut) */
template <int NT, bool REGKEYS> /* threads of the workgroup; REGKEYS: problems up to OKPT * NT keys live in registers */
__device__ __forceinline__ void
oct_walk_body(const uint8_t* __restrict__ cand_region, size_t cand_stride, int ncells, const OctWalkLevel P, int level, int slot,
              uint32_t* pts_a, uint16_t* nid_a, size_t pts_stride, uint32_t* sel_xyr, int32_t* sel_cnt, int32_t* err_flag,
              uint8_t* osm, uint32_t* s_w32, int* s_ctl) {
#define s_size s_ctl[0]
#define s_M s_ctl[1]
#define s_nexp s_ctl[2]
#define s_cut s_ctl[3]
    constexpr int WBATCH = REGKEYS ? OBATCH : 2; /* keys per thread and batch when streaming */
    const int MAXN = P.maxNodes;
    ONode* cur = (ONode*)osm;
    ONode* nxt = cur + MAXN;
    u64* Sbeg = (u64*)(nxt + MAXN);       /* v2: only used as the arg-max array of the final selection */
    u64* Cnt = Sbeg + MAXN;               /* packed per-quadrant key counts of a node (histogram) */
    uint16_t* cb = (uint16_t*)(Cnt + MAXN);   /* list index of a processed node's FIRST created child */
    uint16_t* newIdx = cb + MAXN;             /* list index of a survivor after the pass */
    uint16_t* prank = newIdx + MAXN;          /* processing rank of an expandable node */
    uint16_t* ordv = prank + MAXN;            /* node at processing rank r (phase 2) */
    const int tid = threadIdx.x;
#ifdef VSLAM_OCT_STAMPS /* diagnostic build (make EXTRA_HIPFLAGS=-DVSLAM_OCT_STAMPS): where the level-0 workgroup
                           of slot 0 spends its time; read with vslam_dbg_octree_stamps / tools/octree_stamps.py */
    int dbgn = 0;
    unsigned long long* DBG = (unsigned long long*)P.dbg;
#define STAMP() do { if (DBG && tid == 0 && level == 0 && slot == 0 && dbgn < 60) DBG[dbgn++] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define STAMP() do { } while (0)
#endif
    STAMP();
    const int N = P.N;
    const uint32_t* hdr = (const uint32_t*)(cand_region + (size_t)slot * cand_stride);
    const CellOut* cout = (const CellOut*)(hdr + 2);
    const uint32_t* cand = (const uint32_t*)(cout + ncells);
    uint32_t* pa = pts_a + (size_t)slot * pts_stride;
    uint16_t* na = nid_a + (size_t)slot * pts_stride;
    uint32_t* out = sel_xyr + (size_t)slot * P.selStride + P.selOff;
    int32_t* ocnt = sel_cnt + slot * VSLAM_MAX_LEVELS + level;

    /* ---- 0. gather this level's candidates in cell order (vToDistributeKeys, fextractor.cpp:809-817):
     * a wave per cell, coalesced.  A key's position in this array IS its rank in the reference's key order. */
    const int c0 = P.c0, c1 = P.c1;
    uint32_t before = 0;
    for (int c = tid; c < c0; c += NT) before += cout[c].count;
    uint32_t off0;
    {
        uint32_t tot;
        block_excl_scan<uint32_t, NT>(before, s_w32, &tot);
        off0 = tot;
    }
    const int ncl = c1 - c0, K = (ncl + NT - 1) / NT;
    uint32_t mine = 0;
    for (int k = 0; k < K; k++) {
        const int c = c0 + tid * K + k;
        if (c < c1) mine += cout[c].count;
    }
    uint32_t ntot;
    uint32_t woff = block_excl_scan<uint32_t, NT>(mine, s_w32, &ntot);
    const int n = (int)ntot;
    if (off0 + ntot > (uint32_t)P.ptsCap || n >= (1 << FB) || hdr[1] != 0) {
        if (tid == 0) {
            atomicOr(err_flag, 1);
            *ocnt = 0;
        }
        return;
    }
    pa += off0; na += off0;
    uint32_t* coff = (uint32_t*)nxt; /* cell offsets, borrowed from the second node array (ncl <= 4 * MAXN) */
    for (int k = 0; k < K; k++) {
        const int c = c0 + tid * K + k;
        if (c < c1) {
            coff[c - c0] = woff;
            woff += cout[c].count;
        }
    }
    __syncthreads();
    {
        const int lane = tid & 63, wv = tid >> 6;
        for (int c = c0 + wv; c < c1; c += NT / 64) {
            const uint32_t* q = cand + cout[c].base;
            const uint32_t cnt = cout[c].count, o = coff[c - c0];
            for (uint32_t e = lane; e < cnt; e += 64) pa[o + e] = q[e];
        }
    }
    if (n == 0) {
        if (tid == 0) *ocnt = 0;
        return;
    }
    __syncthreads();

    STAMP();

    /* Key walks.  A problem of up to OKPT * 1024 keys (every KITTI-size level) keeps its keys and their node labels in
     * REGISTERS for the whole kernel: after this one read the walks touch no global memory at all.  Larger problems
     * (1080p level 0: ~100 k keys) stream keys and labels through registers in batches of WBATCH per thread, all
     * loads of a batch issued before the first use (one L2 round trip per batch instead of one per key). */
    const bool inReg = REGKEYS && n <= OKPT * NT;
    uint32_t keyR[OKPT];
    uint32_t nidR[OKPT];
    if (inReg) {
#pragma unroll
        for (int k = 0; k < OKPT; k++) {
            const int i = tid + k * NT;
            keyR[k] = i < n ? pa[i] : 0u;
            nidR[k] = 0u;
        }
    }
    /* body(i, key, nid&): called once per key; writesNid: the walk changes labels */
    auto walk = [&](auto body, bool writesNid) {
        if (inReg) {
#pragma unroll
            for (int k = 0; k < OKPT; k++) {
                const int i = tid + k * NT;
                if (i < n) body(i, keyR[k], nidR[k]);
            }
        } else {
            for (int base = tid; base < n; base += WBATCH * NT) {
                uint32_t kk[WBATCH], nn[WBATCH];
#pragma unroll
                for (int j = 0; j < WBATCH; j++) {
                    const int i = base + j * NT;
                    kk[j] = i < n ? pa[i] : 0u;
                    nn[j] = i < n ? (uint32_t)na[i] : 0u;
                }
#pragma unroll
                for (int j = 0; j < WBATCH; j++) {
                    const int i = base + j * NT;
                    if (i < n) {
                        const uint32_t before = nn[j];
                        body(i, kk[j], nn[j]);
                        if (writesNid && nn[j] != before) na[i] = (uint16_t)nn[j];
                    }
                }
            }
        }
    };

    /* ---- 1. initial nodes: stable bucketing by (int)(x / hX) (fextractor.cpp:534-576) */
    const int nIni = P.nIni;
    const float hX = P.hX;
    const int Hh = P.H;
    __shared__ uint32_t s_bcnt[64], s_bidx[64];
    if (tid < 64) s_bcnt[tid] = 0;
    __syncthreads();
    walk([&](int, uint32_t key, uint32_t&) {
        int b = (int)__fdiv_rn((float)(key & 0xFFF), hX);
        b = min(b, nIni - 1);
        /* a handful of buckets: one LDS atomic per bucket and wave instead of 64 colliding ones */
        const int lane = tid & 63;
        for (int j = 0; j < nIni; j++) {
            const unsigned long long m = __ballot(b == j);
            if (m && lane == __ffsll((long long)m) - 1) atomicAdd(&s_bcnt[j], (uint32_t)__popcll(m));
        }
    }, false);
    __syncthreads();
    if (tid == 0) {
        int li = 0;
        for (int b = 0; b < nIni; b++) {
            const uint32_t cb0 = s_bcnt[b];
            s_bidx[b] = (uint32_t)li;
            if (cb0) { /* empty initial nodes are erased (fextractor.cpp:572-573) */
                ONode nd;
                nd.x0 = (int16_t)(int)__fmul_rn(hX, (float)b);
                nd.x1 = (int16_t)(int)__fmul_rn(hX, (float)(b + 1));
                nd.y0 = 0;
                nd.y1 = (int16_t)Hh;
                nd.begin = 0;
                nd.cf = (cb0 << 1) | (cb0 == 1 ? 1u : 0u);
                cur[li++] = nd;
            }
        }
        s_size = li;
    }
    __syncthreads();
    walk([&](int i, uint32_t key, uint32_t& nid) { /* keys never move: every key carries the list index of its node */
        int b = (int)__fdiv_rn((float)(key & 0xFFF), hX);
        b = min(b, nIni - 1);
        nid = s_bidx[b];
        if (!inReg) na[i] = (uint16_t)nid; /* first label: always stored */
    }, false);
    __syncthreads();

    STAMP();
    /* ---- 2. split passes */
    int phase = 1;
    const int KN = (MAXN + NT - 1) / NT;
    for (int iter = 0; iter < P.maxIter; iter++) {
        const int size0 = s_size;
        /* A. children's key counts of every expandable node: a histogram, no key moves */
        for (int k = 0; k < KN; k++) {
            const int v = tid * KN + k;
            if (v < size0) Cnt[v] = 0ull;
        }
        __syncthreads();
        walk([&](int, uint32_t key, uint32_t& nid) {
            const int v = (int)nid;
            const ONode nd = cur[v];
            if (!ND_NOMORE(nd)) {
                const int q = quadrant(key, nd);
                if (q < 3) atomicAdd(&Cnt[v], 1ull << (FB * q));
            }
        }, false);
        __syncthreads();
        STAMP();
        /* D. node level */
        /* D1. processing rank of every expandable node */
        uint32_t nexp_mine = 0;
        for (int k = 0; k < KN; k++) {
            const int v = tid * KN + k;
            if (v < size0 && !ND_NOMORE(cur[v])) nexp_mine++;
        }
        uint32_t nexp;
        uint32_t rbase = block_excl_scan<uint32_t, NT>(nexp_mine, s_w32, &nexp);
        if (nexp == 0) break; /* nothing expandable: lNodes.size() == prevSize -> finish */
        if (phase == 1) {
            for (int k = 0; k < KN; k++) {
                const int v = tid * KN + k;
                if (v < size0 && !ND_NOMORE(cur[v])) {
                    prank[v] = (uint16_t)rbase;
                    ordv[rbase] = (uint16_t)v;
                    rbase++;
                }
            }
        } else {
            /* descending (count, "created later" == smaller list index) */
            for (int k = 0; k < KN; k++) {
                const int v = tid * KN + k;
                if (v < size0 && !ND_NOMORE(cur[v])) {
                    const uint32_t cv = ND_COUNT(cur[v]);
                    uint32_t r = 0;
#pragma unroll 8
                    for (int u = 0; u < size0; u++) {
                        const uint32_t cfu = cur[u].cf; /* count << 1 | noMore */
                        r += (!(cfu & 1u) && ((cfu >> 1) > cv || ((cfu >> 1) == cv && u < v))) ? 1u : 0u;
                    }
                    prank[v] = (uint16_t)r;
                    ordv[r] = (uint16_t)v;
                }
            }
        }
        __syncthreads();
        STAMP();
        /* D2. in processing order: children created before me, running list size -> cut */
        const int KE = ((int)nexp + NT - 1) / NT;
        uint32_t chl = 0;
        for (int k = 0; k < KE; k++) {
            const int r = tid * KE + k;
            if (r < (int)nexp) {
                const int v = ordv[r];
                chl += nchildren(Cnt[v], ND_COUNT(cur[v]));
            }
        }
        uint32_t chtot;
        uint32_t chbase = block_excl_scan<uint32_t, NT>(chl, s_w32, &chtot);
        if (tid == 0) s_cut = (int)nexp; /* number of processed parents */
        __syncthreads();
        {
            /* parent r is processed iff size0 + sum_{r'<r}(nch-1) < N (phase 2); phase 1: all */
            uint32_t cb_run = chbase;
            for (int k = 0; k < KE; k++) {
                const int r = tid * KE + k;
                if (r < (int)nexp) {
                    const int v = ordv[r];
                    const uint32_t nch = nchildren(Cnt[v], ND_COUNT(cur[v]));
                    if (phase == 2 && size0 + (int)cb_run - r >= N) atomicMin(&s_cut, r);
                    cb[v] = (uint16_t)cb_run; /* children created before this parent (creation rank base) */
                    cb_run += nch;
                }
            }
        }
        __syncthreads();
        const int ncut = s_cut;
        /* M = children of processed parents; new size */
        if (tid == 0) {
            int M;
            if (ncut >= (int)nexp) M = (int)chtot;
            else M = cb[ordv[ncut]];
            s_M = M;
            s_size = size0 + M - ncut;
            s_nexp = 0;
        }
        __syncthreads();
        const int M = s_M;
        STAMP();
        /* D3. survivors: list index after the pass */
        uint32_t sv = 0;
        for (int k = 0; k < KN; k++) {
            const int v = tid * KN + k;
            if (v < size0) {
                const bool processed = !ND_NOMORE(cur[v]) && prank[v] < ncut;
                if (!processed) sv++;
            }
        }
        uint32_t svtot;
        uint32_t svbase = block_excl_scan<uint32_t, NT>(sv, s_w32, &svtot);
        int nexp_children = 0;
        for (int k = 0; k < KN; k++) {
            const int v = tid * KN + k;
            if (v >= size0) continue;
            const ONode nd = cur[v];
            const bool processed = !ND_NOMORE(nd) && prank[v] < ncut;
            if (!processed) {
                newIdx[v] = (uint16_t)(M + svbase);
                nxt[M + svbase] = nd;
                svbase++;
            } else {
                const u64 c = Cnt[v];
                const int mx = nd.x0 + ((nd.x1 - nd.x0 + 1) >> 1), my = nd.y0 + ((nd.y1 - nd.y0 + 1) >> 1);
                int kq = 0;
                const int first = M - 1 - (int)cb[v]; /* list index of the first created child */
                const uint32_t c3 = ND_COUNT(nd) - fld(c, 0) - fld(c, 1) - fld(c, 2);
                for (int q = 0; q < 4; q++) {
                    const uint32_t cq = q < 3 ? fld(c, q) : c3;
                    if (!cq) continue;
                    ONode ch;
                    ch.x0 = (q & 1) ? (int16_t)mx : nd.x0;
                    ch.x1 = (q & 1) ? nd.x1 : (int16_t)mx;
                    ch.y0 = (q & 2) ? (int16_t)my : nd.y0;
                    ch.y1 = (q & 2) ? nd.y1 : (int16_t)my;
                    ch.begin = 0;
                    ch.cf = (cq << 1) | (cq == 1 ? 1u : 0u);
                    nxt[first - kq] = ch;
                    if (cq > 1) nexp_children++;
                    kq++;
                }
                cb[v] = (uint16_t)first;
            }
        }
        if (nexp_children) atomicAdd(&s_nexp, nexp_children);
        __syncthreads();
        STAMP();
        /* E. relabel every key with the list index of the node that holds it after the pass */
        walk([&](int, uint32_t key, uint32_t& nid) {
            const int v = (int)nid;
            const ONode nd = cur[v];
            if (ND_NOMORE(nd) || prank[v] >= ncut) {
                nid = newIdx[v];
            } else {
                const int q = quadrant(key, nd);
                const u64 c = Cnt[v];
                int kq = 0;
                for (int q2 = 0; q2 < q; q2++) kq += fld(c, q2) != 0;
                nid = (uint32_t)(cb[v] - kq);
            }
        }, true);
        __syncthreads();
        { ONode* t = cur; cur = nxt; nxt = t; }
        STAMP();
        /* F. loop control (fextractor.cpp:658-729) */
        const int size = s_size, nToExpand = s_nexp;
        __syncthreads();
        if (size >= N || size == size0) break;
        if (phase == 1 && size + nToExpand * 3 > N) phase = 2;
    }

    STAMP();
    /* ---- 3. best response per node, first in key order wins (fextractor.cpp:732-751): a segmented arg-max,
     * key = response << 32 | ~position; output in list order */
    const int size = s_size;
    u64* best = Sbeg;
    for (int v = tid; v < size; v += NT) best[v] = 0ull;
    __syncthreads();
    walk([&](int i, uint32_t key, uint32_t& nid) {
        atomicMax(&best[nid], ((u64)(key >> 24) << 32) | (u64)(0xFFFFFFFFu - (uint32_t)i));
    }, false);
    __syncthreads();
    for (int v = tid; v < size; v += NT) /* every listed node holds at least one key */
        out[v] = best[v] ? pa[0xFFFFFFFFu - (uint32_t)(best[v] & 0xFFFFFFFFull)] : 0u;
#ifdef VSLAM_OCT_STAMPS
    __syncthreads();
    STAMP();
    if (DBG && tid == 0 && level == 0 && slot == 0) DBG[63] = dbgn;
#endif
    if (tid == 0) *ocnt = size;
#undef s_size
#undef s_M
#undef s_nexp
#undef s_cut
}

/* the distribution as a kernel of its own (VSLAM_OCTREE=v2, and contexts whose fine grid was not allocated): one workgroup
 * per problem; grid (slots, levels): workgroups go to XCDs round-robin by linear id, so the heavy level-0 problems of a
 * batch spread over all eight XCDs instead of piling onto XCD 0 (which grid (levels, slots) did) */
__global__ void __launch_bounds__(OT)
k_octree_v2(const uint8_t* __restrict__ cand_region, size_t cand_stride, int ncells, OctParams P, uint32_t* pts_a, uint16_t* nid_a,
            size_t pts_stride, uint32_t* sel_xyr, int32_t* sel_cnt, int32_t* err_flag) {
    extern __shared__ __align__(16) uint8_t osm[];
    __shared__ uint32_t s_w32[OT / 64];
    __shared__ int s_ctl[4];
    wave_prio_raise(P.wavePrio);
    const int level = blockIdx.y, slot = blockIdx.x;
    oct_walk_body<OT, true>(cand_region, cand_stride, ncells, oct_walk_level(P, level), level, slot, pts_a, nid_a, pts_stride, sel_xyr,
                      sel_cnt, err_flag, osm, s_w32, s_ctl);
}

/* ------------------------------------------------------------------------------------------------
 * k_octree_v4: the distribution with TWO key walks in all, no walk per pass, and no limit on how finely keys cluster.
 *
 * DivideNode's boundaries depend on the node alone (midpoints, fextractor.cpp:474-475), never on the keys, so the
 * quadtree below an initial node is a fixed implicit tree; and because x and y are halved separately, a key's path to
 * depth D (one 2-bit quadrant per depth) is two table look-ups, path = xs[x] | ys[y] (vslam::build_oct_lut).
 *   walk 1  every key is read ONCE from the FAST cells' segments (position in cell order = the reference's key order),
 *           its leaf of depth D ("fine cell", nIni * 4^D of them, counters in LDS) is counted with a returning LDS
 *           atomic: the old count is the key's rank inside its fine cell;
 *           -> one prefix sum over the fine cells;
 *   walk 2  every key is stored at sorted[PS[cell] + rank]: the keys are now sorted by fine cell, so a node of depth
 *           d <= D IS the contiguous run of the 4^(D-d) fine cells under its path, in the prefix array and in the keys.
 *   passes  k_octree_v2's node logic (list order, "largest first until N", creation ranks); the child counts a pass
 *           needs are four differences of the prefix array, per NODE, not per key.  A node DEEPER than the grid (keys
 *           closer together than a fine cell: real images do that on the sparse top levels, where every node is split
 *           down to single keys) lies inside one fine cell: its few keys are enumerated from the sorted array and
 *           tested against the node's path by walking the halvings -- exact at any depth, no fallback kernel.
 *   select  "best response, first key wins" (fextractor.cpp:732-751): four lanes per final node reduce its run of
 *           the sorted array (response, then original position); no atomics, no owner table.
 * One workgroup per (slot, level): the quadtree stays a narrow kernel that hides behind the grid-filling ones.
 * History (git): v3 kept an owner table instead of sorted keys (gather + count + owner fill + select walks, fine cell by D
 * dependent halvings per key: 52 / 89 / 251 us for the level-0 problem at KITTI N=1000 / 2000 / 1080p N=4000) and
 * handed problems whose keys cluster below the grid to the walk-per-pass code in k_assign_out -- which the reference's
 * own test images (hut_stereo 752x480) triggered on 5-30 % of their (slot, level) problems.
 * ---------------------------------------------------------------------------------------------- */
#define ND4_DEPTH(nd) ((int)((nd).cf >> 28))
#define ND4_COUNT(nd) (((nd).cf & 0x0FFFFFFFu) >> 1)

__device__ __forceinline__ uint32_t oct_key_path(uint32_t key, float hX, int nIni, int Hh, int depth) {
    const int x = key & 0xFFF, y = (key >> 12) & 0xFFF;
    int b = (int)__fdiv_rn((float)x, hX); /* initial node, fextractor.cpp:557-561 */
    b = min(b, nIni - 1);
    int x0 = (int)__fmul_rn(hX, (float)b), x1 = (int)__fmul_rn(hX, (float)(b + 1)), y0 = 0, y1 = Hh;
    uint32_t code = 0;
    for (int d = 0; d < depth; d++) { /* DivideNode: halfX = ceil((UR.x - UL.x) / 2); kp.x < n1.UR.x, kp.y < n1.BR.y */
        const int mx = x0 + ((x1 - x0 + 1) >> 1), my = y0 + ((y1 - y0 + 1) >> 1);
        const int qx = x < mx ? 0 : 1, qy = y < my ? 0 : 1;
        code = (code << 2) | (uint32_t)qx | ((uint32_t)qy << 1);
        if (qx) x0 = mx; else x1 = mx;
        if (qy) y0 = my; else y1 = my;
    }
    return ((uint32_t)b << (2 * depth)) | code;
}

/* inclusive scan over a 64-cell tile, lane = cell */
__device__ __forceinline__ uint32_t wave_incl_add(uint32_t v) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(v, o, 64);
        if ((int)(threadIdx.x & 63) >= o) v += t;
    }
    return v;
}

/* REGKEYS: keys (and their cell | rank words) of problems up to OKPT * 1024 keys stay in registers between the two walks
 * (1-2 images: latency); otherwise both are written to the slot's scratch by walk 1 and re-read, coalesced, by walk 2
 * (batches: fewer VGPRs, so that foreign waves fit next to a 1024-thread workgroup). */
/* PRE (round 4, vslam_tuning.oct_precount): walk 1 has run as k_oct_count -- the keys of the level are in pa[] in key order, their
 * leaf | rank << 16 in aux[], the leaf counters in P.fineCnt -- and this kernel starts from the counts. */
template <bool REGKEYS, int OTV, bool PRE = false> /* OTV: threads of the workgroup (256 / 512 / 1024, vk_octree) */
__global__ void __launch_bounds__(OTV)
k_octree_v4(const uint8_t* __restrict__ cand_region, size_t cand_stride, int ncells, OctParams P, uint32_t* keys_a,
            uint32_t* aux_a, uint2* sorted_a, size_t pts_stride, uint32_t* sel_xyr, int32_t* sel_cnt, int32_t* err_flag,
            int32_t* deep_flags) {
    extern __shared__ __align__(16) uint8_t osm[];
    wave_prio_raise(P.wavePrio);
    const int MAXN = P.maxNodes;
    ONode* cur = (ONode*)osm;
    ONode* nxt = cur + MAXN;
    uint32_t* Kq = (uint32_t*)(nxt + MAXN);   /* phase 2: sort key of every node (8 bytes per node reserved, 4 used) */
    u64* Cnt = (u64*)(nxt + MAXN) + MAXN;     /* packed per-quadrant key counts of a node */
    uint16_t* cb = (uint16_t*)(Cnt + MAXN);   /* list index of a processed node's FIRST created child */
    uint16_t* newIdx = cb + MAXN;             /* (k_octree_v2 only) */
    uint16_t* prank = newIdx + MAXN;          /* processing rank of an expandable node */
    uint16_t* ordv = prank + MAXN;            /* node at processing rank r */
    __shared__ uint32_t s_w32[OTV / 64];
    __shared__ int s_size, s_M, s_nexp, s_cut, s_deep;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int level = blockIdx.y, slot = blockIdx.x;
    const int N = P.N[level];
    const uint32_t* hdr = (const uint32_t*)(cand_region + (size_t)slot * cand_stride);
    const CellOut* cout = (const CellOut*)(hdr + 2);
    const uint32_t* cand = (const uint32_t*)(cout + ncells);
    uint32_t* pa = keys_a + (size_t)slot * pts_stride;
    uint32_t* aux = aux_a + (size_t)slot * pts_stride;   /* fine cell | rank in the cell << 16 of every key */
    uint2* sorted = sorted_a + (size_t)slot * pts_stride; /* {key, position in key order}, sorted by fine cell */
    uint32_t* out = sel_xyr + (size_t)slot * P.selStride + P.selOff[level];
    int32_t* ocnt = sel_cnt + slot * VSLAM_MAX_LEVELS + level;
    if (tid == 0) {
        deep_flags[slot * VSLAM_MAX_LEVELS + level] = 0;
        s_deep = 0;
    }
#ifdef VSLAM_OCT_STAMPS
    int dbgn3 = 0;
    unsigned long long* DBG3 = (unsigned long long*)P.dbg;
#define STAMP3() do { if (DBG3 && tid == 0 && level == 0 && slot == 0 && dbgn3 < 60) DBG3[dbgn3++] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define STAMP3() do { } while (0)
#endif
    STAMP3();

    /* ---- 0. this level's candidates in cell order (vToDistributeKeys, fextractor.cpp:809-817): a key's position in
     * that order is its rank in the reference's key order.  One binary search in the cells' offsets (LDS) for a lane's
     * first position, then it walks on cell by cell. */
    const int c0 = P.cellFirst[level], c1 = P.cellFirst[level + 1];
    uint32_t off0 = 0;
    if (c0 > 0) { /* keys of the levels in front of this one (level 0: none, no scan) */
        uint32_t before = 0;
        for (int c = tid; c < c0; c += OTV) before += cout[c].count;
        uint32_t tot;
        block_excl_scan<uint32_t, OTV>(before, s_w32, &tot);
        off0 = tot;
    }
    const int ncl = c1 - c0, K = (ncl + OTV - 1) / OTV;
    uint32_t mine = 0;
    for (int k = 0; k < K; k++) {
        const int c = c0 + tid * K + k;
        if (c < c1) mine += cout[c].count;
    }
    uint32_t ntot;
    uint32_t woff = block_excl_scan<uint32_t, OTV>(mine, s_w32, &ntot);
    const int n = (int)ntot;
    if (off0 + ntot > (uint32_t)P.ptsCap || n >= (1 << 20) || hdr[1] != 0) {
        if (tid == 0) {
            atomicOr(err_flag, 1);
            *ocnt = 0;
        }
        return;
    }
    pa += off0;
    aux += off0;
    sorted += off0;
    /* cell offsets (+ sentinel) and the cells' segment bases, borrowed from the node arrays (2 * ncl + 1 words; the host
     * sizes MAXN >= ncl / 4, i.e. 14 * MAXN words of node arrays): the walk needs ONE global round trip (the keys) */
    uint32_t* coff = (uint32_t*)nxt;
    uint32_t* cbas = coff + ncl + 1;
    if (!PRE) {
        for (int k = 0; k < K; k++) {
            const int c = c0 + tid * K + k;
            if (c < c1) {
                const CellOut co = cout[c];
                coff[c - c0] = woff;
                cbas[c - c0] = co.base;
                woff += co.count;
            }
        }
        if (tid == 0) coff[ncl] = ntot;
    }
    if (n == 0) {
        if (tid == 0) *ocnt = 0;
        return;
    }
    const int nIni = P.nIni[level];
    const float hX = P.hX[level];
    const int Hh = P.H[level];
    const int D = P.fineD[level];
    const int cells = nIni << (2 * D);
    uint32_t* Hc = (uint32_t*)(osm + P.fineLdsOff); /* key counts of the fine cells */
    uint32_t* PS = Hc + cells + 1;                  /* their exclusive prefix sums */
    const uint32_t* __restrict__ xs = P.lut + P.lutOff[level];
    const uint32_t* __restrict__ ys = xs + P.lutW[level];
    if (PRE) { /* the counters k_oct_count left for this (slot, level) */
        const uint32_t* __restrict__ Hg = P.fineCnt + (size_t)slot * P.fineCntStride + P.fineCntOff[level];
        for (int i = tid; i <= cells; i += OTV) Hc[i] = i < cells ? Hg[i] : 0u;
    } else {
        for (int i = tid; i <= cells; i += OTV) Hc[i] = 0u;
    }
    /* the path tables are cold (another XCD's L2 or HBM) the first time a workgroup touches them: start pulling their
     * lines now, the key loads below hide the round trip */
    uint32_t warm = 0u;
    if (!PRE && tid * 16 < P.lutW[level] + Hh + 1) warm = xs[tid * 16]; /* consumed (by nothing) behind walk 1 */
    __syncthreads();

    /* positions are dealt to WAVES in contiguous chunks of EW (a multiple of 64) and to the lanes of a wave interleaved:
     * lane l holds positions wbeg + 64 k + l, so that a wave's loads (mostly one FAST cell segment after the other) and
     * its stores are coalesced */
    const bool inReg = !PRE && REGKEYS && n <= OKPT * OTV;
    const int KW = (n + OTV - 1) / OTV, EW = KW * 64; /* keys per lane, positions per wave */
    const int wbeg = wv * EW;
    uint32_t keyR[OKPT];
    uint32_t auxR[OKPT];
#pragma unroll
    for (int k = 0; k < OKPT; k++) keyR[k] = auxR[k] = 0u;

    /* ---- 1. walk 1: read every key, count it into its fine cell; the counter's old value is its rank in the cell */
    if (!PRE) {
        const int p0 = wbeg + lane;
        int c = 0;
        if (p0 < n) {
            int lo = 0, hi = ncl - 1; /* last cell with coff <= p0 (empty cells share an offset: the last one holds it) */
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (coff[mid] <= (uint32_t)p0) lo = mid;
                else hi = mid - 1;
            }
            c = lo;
        }
        uint32_t cbase = p0 < n ? cbas[c] : 0u, cfirst = p0 < n ? coff[c] : 0u, cnext = p0 < n ? coff[c + 1] : 0u;
        if (inReg) {
#pragma unroll
            for (int k = 0; k < OKPT; k++) {
                const int i = p0 + 64 * k;
                if (k < KW && i < n) {
                    while ((uint32_t)i >= cnext) { /* on to the cell that holds position i */
                        c++;
                        cfirst = cnext;
                        cnext = coff[c + 1];
                        cbase = cbas[c];
                    }
                    keyR[k] = cand[cbase + ((uint32_t)i - cfirst)];
                }
            }
#pragma unroll
            for (int k = 0; k < OKPT; k++)
                if (k < KW && p0 + 64 * k < n) auxR[k] = xs[keyR[k] & 0xFFF] | ys[(keyR[k] >> 12) & 0xFFF];
#pragma unroll
            for (int k = 0; k < OKPT; k++)
                if (k < KW && p0 + 64 * k < n) auxR[k] |= atomicAdd(&Hc[auxR[k]], 1u) << 16;
        } else {
            for (int kb = 0; kb < KW; kb += O4BATCH) { /* O4BATCH loads in flight, then the table look-ups, then the atomics */
                uint32_t kk[O4BATCH], ff[O4BATCH];
#pragma unroll
                for (int j = 0; j < O4BATCH; j++) {
                    const int i = p0 + 64 * (kb + j);
                    kk[j] = 0u;
                    if (kb + j < KW && i < n) {
                        while ((uint32_t)i >= cnext) {
                            c++;
                            cfirst = cnext;
                            cnext = coff[c + 1];
                            cbase = cbas[c];
                        }
                        kk[j] = cand[cbase + ((uint32_t)i - cfirst)];
                    }
                }
#pragma unroll
                for (int j = 0; j < O4BATCH; j++)
                    ff[j] = (kb + j < KW && p0 + 64 * (kb + j) < n) ? xs[kk[j] & 0xFFF] | ys[(kk[j] >> 12) & 0xFFF] : 0u;
#pragma unroll
                for (int j = 0; j < O4BATCH; j++)
                    if (kb + j < KW && p0 + 64 * (kb + j) < n) ff[j] |= atomicAdd(&Hc[ff[j]], 1u) << 16;
#pragma unroll
                for (int j = 0; j < O4BATCH; j++)
                    if (kb + j < KW && p0 + 64 * (kb + j) < n) {
                        pa[p0 + 64 * (kb + j)] = kk[j];
                        aux[p0 + 64 * (kb + j)] = ff[j];
                    }
            }
        }
    }
    asm volatile("" ::"v"(warm));
    __syncthreads(); /* coff (in the node arrays) is free again; the counters are complete */
    STAMP3();
    {   /* exclusive prefix sums of the fine counts, LDS to LDS */
        const int ntile = (cells + 63) >> 6, tpw = (ntile + OTV / 64 - 1) / (OTV / 64); /* tiles per wave */
        const int t1 = min((wv + 1) * tpw, ntile);
        uint32_t carry = 0;
        for (int t = wv * tpw; t < t1; t++) {
            const int c = t * 64 + lane;
            const uint32_t h = c < cells ? Hc[c] : 0u;
            /* a key's rank inside its fine cell travels in 16 bits next to the cell index (walk 1); the host sizes the
             * grid so that no cell can hold that many keys, this is the runtime guard behind that bound: the pass is
             * reported (VSLAM_ERR_CAPACITY), never silently mis-sorted */
            if (h >= 65536u) atomicOr(err_flag, 1);
            const uint32_t inc = wave_incl_add(h);
            if (c < cells) PS[c] = carry + inc - h;
            carry += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        }
        if (lane == 0) s_w32[wv] = carry;
        __syncthreads();
        uint32_t wo = 0;
        for (int k = 0; k < wv; k++) wo += s_w32[k];
        if (wo)
            for (int t = wv * tpw; t < t1; t++) {
                const int c = t * 64 + lane;
                if (c < cells) PS[c] += wo;
            }
        if (tid == 0) PS[cells] = (uint32_t)n;
    }
    __syncthreads();
    STAMP3();

    /* ---- 2. walk 2: the keys sorted by fine cell (order inside a cell: whatever the atomics gave; the selection below
     * decides by original position, which travels with the key) */
    if (inReg) {
#pragma unroll
        for (int k = 0; k < OKPT; k++) {
            const int i = wbeg + lane + 64 * k;
            if (k < KW && i < n) sorted[PS[auxR[k] & 0xFFFFu] + (auxR[k] >> 16)] = make_uint2(keyR[k], (uint32_t)i);
        }
    } else {
        for (int base = tid; base < n; base += O4BATCH * OTV) {
            uint32_t kk[O4BATCH], ff[O4BATCH];
#pragma unroll
            for (int j = 0; j < O4BATCH; j++) {
                const int i = base + j * OTV;
                kk[j] = i < n ? pa[i] : 0u;
                ff[j] = i < n ? aux[i] : 0u;
            }
#pragma unroll
            for (int j = 0; j < O4BATCH; j++) {
                const int i = base + j * OTV;
                if (i < n) sorted[PS[ff[j] & 0xFFFFu] + (ff[j] >> 16)] = make_uint2(kk[j], (uint32_t)i);
            }
        }
    }
    /* initial nodes; ONode.begin = path code of the node (root << 2 depth | quadrants), depth in cf[31:28] */
    if (tid == 0) {
        int li = 0;
        for (int b = 0; b < nIni; b++) {
            const uint32_t cb0 = PS[(b + 1) << (2 * D)] - PS[b << (2 * D)];
            if (cb0) { /* empty initial nodes are erased (fextractor.cpp:572-573) */
                ONode nd;
                nd.x0 = (int16_t)(int)__fmul_rn(hX, (float)b);
                nd.x1 = (int16_t)(int)__fmul_rn(hX, (float)(b + 1));
                nd.y0 = 0;
                nd.y1 = (int16_t)Hh;
                nd.begin = (uint32_t)b;
                nd.cf = (cb0 << 1) | (cb0 == 1 ? 1u : 0u);
                cur[li++] = nd;
            }
        }
        s_size = li;
    }
    __syncthreads(); /* also: the sorted keys (global memory, this CU's stores) are complete for the whole workgroup */
    STAMP3();

    /* ---- 3. split passes: k_octree_v2's node logic; the children's key counts come from the prefix sums */
    int phase = 1;
    const int KN = (MAXN + OTV - 1) / OTV;
    for (int iter = 0; iter < P.maxIter; iter++) {
        const int size0 = s_size;
        /* A. children's key counts of every expandable node; phase 2 also needs every node's sort key */
        for (int k = 0; k < KN; k++) {
            const int v = tid * KN + k;
            if (v < size0) {
                const ONode nd = cur[v];
                u64 c = 0ull;
                uint32_t kq = 0u;
                if (!ND_NOMORE(nd)) {
                    const int depth = ND4_DEPTH(nd);
                    const uint32_t path = nd.begin;
                    if (depth < D) {
                        const uint32_t cb0 = path << (2 * (D - depth)), q4 = 1u << (2 * (D - depth - 1));
                        const uint32_t p0 = PS[cb0], p1 = PS[cb0 + q4], p2 = PS[cb0 + 2 * q4], p3 = PS[cb0 + 3 * q4];
                        c = (u64)(p1 - p0) | ((u64)(p2 - p1) << FB) | ((u64)(p3 - p2) << (2 * FB));
                    } else { /* finer than the grid: the node's keys are among the few of ONE fine cell */
                        s_deep = 1; /* statistics only (benign race: all writers store 1) */
                        const uint32_t f = path >> (2 * (depth - D));
                        for (uint32_t j = PS[f]; j < PS[f + 1]; j++) {
                            const uint32_t kp = oct_key_path(sorted[j].x, hX, nIni, Hh, depth + 1);
                            if ((kp >> 2) == path) c += onehot((int)(kp & 3u));
                        }
                    }
                    /* descending (count, "created later" == smaller list index): one unsigned compare */
                    kq = (ND4_COUNT(nd) << 12) | (uint32_t)(4095 - v);
                }
                Cnt[v] = c;
                Kq[v] = kq;
            }
        }
        /* pad the sort keys to a multiple of four (the rank loop reads them four at a time) */
        if (tid < 4 && size0 + tid < ((size0 + 3) & ~3)) Kq[size0 + tid] = 0u;
        __syncthreads();
        /* D1. processing rank of every expandable node */
        uint32_t nexp_mine = 0;
        for (int k = 0; k < KN; k++) {
            const int v = tid * KN + k;
            if (v < size0 && !ND_NOMORE(cur[v])) nexp_mine++;
        }
        uint32_t nexp;
        uint32_t rbase = block_excl_scan<uint32_t, OTV>(nexp_mine, s_w32, &nexp);
        if (nexp == 0) break; /* nothing expandable: lNodes.size() == prevSize -> finish */
        if (phase == 1) {
            for (int k = 0; k < KN; k++) {
                const int v = tid * KN + k;
                if (v < size0 && !ND_NOMORE(cur[v])) {
                    prank[v] = (uint16_t)rbase;
                    ordv[rbase] = (uint16_t)v;
                    rbase++;
                }
            }
        } else {
            /* rank = number of nodes with a larger sort key (nodes that cannot be split have key 0) */
            const uint4* K4 = (const uint4*)Kq;
            const int n4 = (size0 + 3) >> 2;
            for (int k = 0; k < KN; k++) {
                const int v = tid * KN + k;
                if (v < size0 && !ND_NOMORE(cur[v])) {
                    const uint32_t kv = Kq[v];
                    uint32_t r = 0;
#pragma unroll 4
                    for (int u = 0; u < n4; u++) {
                        const uint4 q = K4[u];
                        r += (q.x > kv) + (q.y > kv) + (q.z > kv) + (q.w > kv);
                    }
                    prank[v] = (uint16_t)r;
                    ordv[r] = (uint16_t)v;
                }
            }
        }
        __syncthreads();
        /* D2. in processing order: children created before me, running list size -> cut */
        const int KE = ((int)nexp + OTV - 1) / OTV;
        uint32_t chl = 0;
        for (int k = 0; k < KE; k++) {
            const int r = tid * KE + k;
            if (r < (int)nexp) {
                const int v = ordv[r];
                chl += nchildren(Cnt[v], ND4_COUNT(cur[v]));
            }
        }
        uint32_t chtot;
        uint32_t chbase = block_excl_scan<uint32_t, OTV>(chl, s_w32, &chtot);
        if (tid == 0) s_cut = (int)nexp; /* number of processed parents */
        __syncthreads();
        {
            /* parent r is processed iff size0 + sum_{r'<r}(nch-1) < N (phase 2); phase 1: all */
            uint32_t cb_run = chbase;
            for (int k = 0; k < KE; k++) {
                const int r = tid * KE + k;
                if (r < (int)nexp) {
                    const int v = ordv[r];
                    const uint32_t nch = nchildren(Cnt[v], ND4_COUNT(cur[v]));
                    if (phase == 2 && size0 + (int)cb_run - r >= N) atomicMin(&s_cut, r);
                    cb[v] = (uint16_t)cb_run; /* children created before this parent (creation rank base) */
                    cb_run += nch;
                }
            }
        }
        __syncthreads();
        const int ncut = s_cut;
        if (tid == 0) {
            int M;
            if (ncut >= (int)nexp) M = (int)chtot;
            else M = cb[ordv[ncut]];
            s_M = M;
            s_size = size0 + M - ncut;
            s_nexp = 0;
        }
        __syncthreads();
        const int M = s_M;
        /* D3. survivors keep their relative order behind the children (created in reverse) */
        uint32_t sv = 0;
        for (int k = 0; k < KN; k++) {
            const int v = tid * KN + k;
            if (v < size0) {
                const bool processed = !ND_NOMORE(cur[v]) && prank[v] < ncut;
                if (!processed) sv++;
            }
        }
        uint32_t svtot;
        uint32_t svbase = block_excl_scan<uint32_t, OTV>(sv, s_w32, &svtot);
        int nexp_children = 0;
        for (int k = 0; k < KN; k++) {
            const int v = tid * KN + k;
            if (v >= size0) continue;
            const ONode nd = cur[v];
            const bool processed = !ND_NOMORE(nd) && prank[v] < ncut;
            if (!processed) {
                nxt[M + svbase] = nd;
                svbase++;
            } else {
                const u64 c = Cnt[v];
                const int mx = nd.x0 + ((nd.x1 - nd.x0 + 1) >> 1), my = nd.y0 + ((nd.y1 - nd.y0 + 1) >> 1);
                const uint32_t depth1 = (uint32_t)ND4_DEPTH(nd) + 1u;
                int kq = 0;
                const int first = M - 1 - (int)cb[v]; /* list index of the first created child */
                const uint32_t c3 = ND4_COUNT(nd) - fld(c, 0) - fld(c, 1) - fld(c, 2);
                for (int q = 0; q < 4; q++) {
                    const uint32_t cq = q < 3 ? fld(c, q) : c3;
                    if (!cq) continue;
                    ONode ch;
                    ch.x0 = (q & 1) ? (int16_t)mx : nd.x0;
                    ch.x1 = (q & 1) ? nd.x1 : (int16_t)mx;
                    ch.y0 = (q & 2) ? (int16_t)my : nd.y0;
                    ch.y1 = (q & 2) ? nd.y1 : (int16_t)my;
                    ch.begin = (nd.begin << 2) | (uint32_t)q;
                    ch.cf = (depth1 << 28) | (cq << 1) | (cq == 1 ? 1u : 0u);
                    nxt[first - kq] = ch;
                    if (cq > 1) nexp_children++;
                    kq++;
                }
            }
        }
        if (nexp_children) atomicAdd(&s_nexp, nexp_children);
        __syncthreads();
        { ONode* t = cur; cur = nxt; nxt = t; }
        /* F. loop control (fextractor.cpp:658-729) */
        const int size = s_size, nToExpand = s_nexp;
        __syncthreads();
        STAMP3(); /* one stamp per split pass */
        if (size >= N || size == size0) break;
        if (phase == 1 && size + nToExpand * 3 > N) phase = 2;
    }

    STAMP3();
    /* ---- 4. best response per node, first in key order wins (fextractor.cpp:732-751): four lanes per node reduce its
     * run of the sorted keys on (response, ~position) */
    const int size = s_size;
    for (int v0 = 0; v0 < size; v0 += OTV / 4) {
        const int v = v0 + (tid >> 2), sub = tid & 3;
        u64 best = 0ull;
        uint32_t bkey = 0u;
        if (v < size) {
            const ONode nd = cur[v];
            const int depth = ND4_DEPTH(nd);
            const uint32_t path = nd.begin;
            if (depth <= D) {
                const uint32_t lo = PS[path << (2 * (D - depth))], hi = PS[(path + 1u) << (2 * (D - depth))];
                for (uint32_t j0 = lo + (uint32_t)sub; j0 < hi; j0 += 16) { /* four loads in flight per lane */
                    uint2 r[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) r[u] = j0 + 4 * u < hi ? sorted[j0 + 4 * u] : make_uint2(0u, 0xFFFFFFFFu);
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const u64 a = ((u64)(r[u].x >> 24) << 32) | (u64)(0xFFFFFFFFu - r[u].y);
                        if (j0 + 4 * u < hi && a >= best) { /* >=: a key of response 0 at position 0xFFFFFFFF cannot exist, so 'best == 0' means none yet */
                            best = a;
                            bkey = r[u].x;
                        }
                    }
                }
            } else { /* a node inside one fine cell: test the cell's keys against the node's path */
                const uint32_t f = path >> (2 * (depth - D));
                for (uint32_t j = PS[f] + (uint32_t)sub; j < PS[f + 1]; j += 4) {
                    const uint2 r = sorted[j];
                    if (oct_key_path(r.x, hX, nIni, Hh, depth) == path) {
                        const u64 a = ((u64)(r.x >> 24) << 32) | (u64)(0xFFFFFFFFu - r.y);
                        if (a >= best) {
                            best = a;
                            bkey = r.x;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int o = 1; o < 4; o <<= 1) { /* the node's four lanes are neighbours in the wave */
            const u64 ob = __shfl_xor(best, o, 64);
            const uint32_t ok = __shfl_xor(bkey, o, 64);
            if (ob > best) {
                best = ob;
                bkey = ok;
            }
        }
        if (v < size && sub == 0) out[v] = bkey; /* every listed node holds at least one key */
    }
#ifdef VSLAM_OCT_STAMPS
    __syncthreads();
    STAMP3();
    if (DBG3 && tid == 0 && level == 0 && slot == 0) DBG3[63] = dbgn3;
#endif
    if (tid == 0) {
        *ocnt = size;
        if (s_deep) deep_flags[slot * VSLAM_MAX_LEVELS + level] = 1;
    }
}

/* ------------------------------------------------------------------------------------------------
 * output order of a slot (fextractor.cpp:1071-1129): level-major; a keypoint whose scaled x lies in
 * [lap0, lap1] takes the next free index from the tail, the others from the head.  One workgroup per
 * slot; writes the SelKp list the orientation/descriptor kernel consumes and the slot's counts.
 * ---------------------------------------------------------------------------------------------- */
#define AO_T 256 /* a small workgroup finds a free CU slot quickly next to the other streams' kernels */
__global__ void __launch_bounds__(AO_T)
k_assign_out(OctParams P, PyramidGeom g, uint32_t* sel_xyr, int32_t* sel_cnt,
             int lap0, int lap1, SelKp* sel, int32_t* slot_counts /* [slot][4]: n, mono, deep-level mask, 0 */, int cap,
             int32_t* err_flag, const int32_t* deep_flags) {
    __shared__ uint32_t s_w32[AO_T / 64];
    __shared__ int s_lvl_off[VSLAM_MAX_LEVELS + 1];
    wave_prio_raise(P.wavePrio);
    const int tid = threadIdx.x, slot = blockIdx.x;
    const int L = g.nlevels;
    uint32_t redo_mask = 0; /* levels of this slot on which k_octree_v4 split nodes finer than its grid (vslam_fe_octree_stats) */
    if (deep_flags && tid == 0)
        for (int lv = 0; lv < L; lv++) redo_mask |= deep_flags[slot * VSLAM_MAX_LEVELS + lv] ? 1u << lv : 0u;
    if (tid == 0) {
        int acc = 0;
        for (int l = 0; l < L; l++) {
            s_lvl_off[l] = acc;
            acc += sel_cnt[slot * VSLAM_MAX_LEVELS + l];
        }
        s_lvl_off[L] = acc;
    }
    __syncthreads();
    const int nk = s_lvl_off[L];
    if (nk > cap) {
        if (tid == 0) {
            atomicOr(err_flag, 2);
            slot_counts[slot * 4] = 0;
            slot_counts[slot * 4 + 1] = 0;
            slot_counts[slot * 4 + 2] = (int32_t)redo_mask;
        }
        return;
    }
    const int Cc = (nk + AO_T - 1) / AO_T;
    const int i0 = min(tid * Cc, nk), i1 = min(i0 + Cc, nk);
    const uint32_t* base = sel_xyr + (size_t)slot * P.selStride;
    uint32_t lapc = 0;
    int l = 0;
    for (int i = i0; i < i1; i++) {
        while (i >= s_lvl_off[l + 1]) l++;
        const uint32_t p = base[P.selOff[l] + (i - s_lvl_off[l])];
        float px = (float)((int)(p & 0xFFF) + VSLAM_BORDER);
        if (l) px = __fmul_rn(px, g.lv[l].scale);
        lapc += (px >= (float)lap0 && px <= (float)lap1) ? 1u : 0u;
    }
    uint32_t laptot;
    uint32_t lapbefore = block_excl_scan<uint32_t, AO_T>(lapc, s_w32, &laptot);
    l = 0;
    for (int i = i0; i < i1; i++) {
        while (i >= s_lvl_off[l + 1]) l++;
        const uint32_t p = base[P.selOff[l] + (i - s_lvl_off[l])];
        const int lx = (int)(p & 0xFFF) + VSLAM_BORDER, ly = (int)((p >> 12) & 0xFFF) + VSLAM_BORDER;
        float px = (float)lx;
        if (l) px = __fmul_rn(px, g.lv[l].scale);
        const bool inlap = px >= (float)lap0 && px <= (float)lap1;
        SelKp k;
        k.x = (uint16_t)lx;
        k.y = (uint16_t)ly;
        k.level = (uint8_t)l;
        k.slot = (uint8_t)slot;
        k.response = (uint8_t)(p >> 24);
        k.pad = 0;
        k.out = inlap ? (uint32_t)(nk - 1 - (int)lapbefore) : (uint32_t)(i - (int)lapbefore);
        if (inlap) lapbefore++;
        sel[(size_t)slot * cap + i] = k;
    }
    if (tid == 0) {
        slot_counts[slot * 4] = nk;
        slot_counts[slot * 4 + 1] = nk - (int)laptot; /* monoIndex */
        slot_counts[slot * 4 + 2] = (int32_t)redo_mask;
    }
}

size_t vk_octree_lds_bytes(int maxNodes) { return (size_t)maxNodes * (16 + 16 + 8 + 8 + 2 + 2 + 2 + 2) + 64; }

/* ------------------------------------------------------------------------------------------------
 * k_oct_count (round 4): walk 1 of k_octree_v4 -- every key read once and counted into its leaf, the
 * counter's old value being its rank in the leaf -- as a launch of its own in which a level's keys are dealt to up to eight
 * workgroups.  The parts are ROWS OF LEAVES: a key's part is the leading y decisions of its leaf index (P.partBits), i.e. a
 * y range (OctPart), so every leaf is counted by exactly one workgroup -- its rank needs no merging -- and a part reads the
 * FAST cell rows that overlap its y range (a row on a boundary is read by both parts; each takes its own keys).  The
 * level-0 problem of a 1080p frame (65 k keys, 55 us of look-ups and LDS atomics on ONE CU inside k_octree_v4) becomes
 * eight workgroups of 8 k keys.  Outputs: pa[] (keys in key order), aux[] (leaf | rank << 16) and the leaf counters.
 * ---------------------------------------------------------------------------------------------- */
#define OCNT_T 256
__global__ void __launch_bounds__(OCNT_T)
k_oct_count(const uint8_t* __restrict__ cand_region, size_t cand_stride, int ncells, OctParams P, uint32_t* keys_a,
            uint32_t* aux_a, size_t pts_stride) {
    extern __shared__ __align__(16) uint8_t csm[];
    __shared__ u64 s_w64[OCNT_T / 64];
    __shared__ uint32_t s_w32[OCNT_T / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int part = blockIdx.x, level = blockIdx.y, slot = blockIdx.z;
    const int kb = P.partBits[level];
    if (part >= (1 << kb)) return;
    const OctPart pt = P.parts[level * VSLAM_OCT_MAX_PARTS + part];
    const uint32_t* hdr = (const uint32_t*)(cand_region + (size_t)slot * cand_stride);
    const CellOut* cout = (const CellOut*)(hdr + 2);
    const uint32_t* cand = (const uint32_t*)(cout + ncells);
    const int c0 = P.cellFirst[level], c1 = P.cellFirst[level + 1];
    /* keys of the levels in front of this one | of this level in front of the part's cells | of this level: three 21-bit
     * sums in one scan (a level of 2^20 keys or more is refused by k_octree_v4 anyway) */
    u64 sums = 0ull;
    for (int c = tid; c < c1; c += OCNT_T) {
        const u64 cnt = cout[c].count;
        sums += c < c0 ? cnt : (cnt << 42) | (c < pt.ca ? cnt << 21 : 0ull);
    }
    u64 tot3;
    block_excl_scan<u64, OCNT_T>(sums, s_w64, &tot3);
    const uint32_t off0 = (uint32_t)(tot3 & 0x1FFFFFull), Sa = (uint32_t)((tot3 >> 21) & 0x1FFFFFull);
    const uint32_t n = (uint32_t)(tot3 >> 42);
    if (off0 + n > (uint32_t)P.ptsCap || n >= (1u << 20) || hdr[1] != 0 || n == 0) return; /* k_octree_v4 reports it */
    uint32_t* pa = keys_a + (size_t)slot * pts_stride + off0;
    uint32_t* aux = aux_a + (size_t)slot * pts_stride + off0;
    const int D = P.fineD[level];
    const int cells = P.nIni[level] << (2 * D);
    uint32_t* Hl = (uint32_t*)csm;          /* this part's leaf counters (all leaves; only its own are ever touched) */
    uint32_t* coff = Hl + cells + 1;        /* first local position of every cell of the part (+ sentinel) */
    uint32_t* cbas = coff + P.maxPartCells + 1;
    const int ncl = pt.cb - pt.ca, K = (ncl + OCNT_T - 1) / OCNT_T;
    uint32_t mine = 0;
    for (int k = 0; k < K; k++) {
        const int c = pt.ca + tid * K + k;
        if (c < pt.cb) mine += cout[c].count;
    }
    uint32_t npart;
    uint32_t woff = block_excl_scan<uint32_t, OCNT_T>(mine, s_w32, &npart);
    for (int k = 0; k < K; k++) {
        const int c = pt.ca + tid * K + k;
        if (c < pt.cb) {
            const CellOut co = cout[c];
            coff[c - pt.ca] = woff;
            cbas[c - pt.ca] = co.base;
            woff += co.count;
        }
    }
    if (tid == 0) coff[ncl] = npart;
    for (int i = tid; i <= cells; i += OCNT_T) Hl[i] = 0u;
    const uint32_t* __restrict__ xs = P.lut + P.lutOff[level];
    const uint32_t* __restrict__ ys = xs + P.lutW[level];
    __syncthreads();
    /* local positions are dealt to waves in contiguous chunks and to the lanes of a wave interleaved, as in k_octree_v4 */
    const int np = (int)npart;
    const int KW = (np + OCNT_T - 1) / OCNT_T, EW = KW * 64;
    const int p0 = wv * EW + lane;
    if (np > 0) {
        int c = 0;
        if (p0 < np) {
            int lo = 0, hi = ncl - 1; /* last cell with coff <= p0 */
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (coff[mid] <= (uint32_t)p0) lo = mid;
                else hi = mid - 1;
            }
            c = lo;
        }
        uint32_t cbase = p0 < np ? cbas[c] : 0u, cfirst = p0 < np ? coff[c] : 0u, cnext = p0 < np ? coff[c + 1] : 0u;
        for (int kb0 = 0; kb0 < KW; kb0 += O4BATCH) {
            uint32_t kk[O4BATCH], ff[O4BATCH];
            bool own[O4BATCH];
#pragma unroll
            for (int j = 0; j < O4BATCH; j++) {
                const int i = p0 + 64 * (kb0 + j);
                kk[j] = 0u;
                own[j] = false;
                if (kb0 + j < KW && i < np) {
                    while ((uint32_t)i >= cnext) {
                        c++;
                        cfirst = cnext;
                        cnext = coff[c + 1];
                        cbase = cbas[c];
                    }
                    kk[j] = cand[cbase + ((uint32_t)i - cfirst)];
                }
            }
#pragma unroll
            for (int j = 0; j < O4BATCH; j++) {
                const int i = p0 + 64 * (kb0 + j);
                const int y = (int)((kk[j] >> 12) & 0xFFF);
                own[j] = kb0 + j < KW && i < np && y >= pt.ylo && y < pt.yhi;
                ff[j] = own[j] ? xs[kk[j] & 0xFFF] | ys[y] : 0u;
            }
#pragma unroll
            for (int j = 0; j < O4BATCH; j++)
                if (own[j]) ff[j] |= atomicAdd(&Hl[ff[j]], 1u) << 16;
#pragma unroll
            for (int j = 0; j < O4BATCH; j++)
                if (own[j]) {
                    const uint32_t pos = Sa + (uint32_t)(p0 + 64 * (kb0 + j)); /* position in the level's key order */
                    pa[pos] = kk[j];
                    aux[pos] = ff[j];
                }
        }
    }
    __syncthreads();
    /* the counters of this part's leaves: leaf index = root << 2D | (y_d << 1 | x_d) per depth, the part = its first kb y bits */
    uint32_t* Hg = P.fineCnt + (size_t)slot * P.fineCntStride + P.fineCntOff[level];
    for (int leaf = tid; leaf < cells; leaf += OCNT_T) {
        int pb = 0;
        for (int t = 0; t < kb; t++) pb = (pb << 1) | (int)(((uint32_t)leaf >> (2 * D - 1 - 2 * t)) & 1u);
        if (pb == part) Hg[leaf] = Hl[leaf];
    }
}

size_t vk_oct_count_lds(int maxcells, int maxPartCells) { return ((size_t)maxcells + 1 + 2 * ((size_t)maxPartCells + 1)) * 4 + 16; }

int vk_oct_count_set_max_lds(size_t bytes) {
    static std::mutex mu;
    static size_t have_dev[64] = {0};
    std::lock_guard<std::mutex> lk(mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return (int)hipErrorInvalidDevice;
    if (bytes <= have_dev[dev]) return 0;
    const int rc = (int)hipFuncSetAttribute((const void*)k_oct_count, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (rc) return rc;
    have_dev[dev] = bytes;
    return 0;
}

void vk_oct_count(hipStream_t st, const uint8_t* cand_region, size_t cand_stride, int ncells, const OctParams& P, uint32_t* keys_a,
                  uint32_t* aux_a, size_t pts_stride, int nlevels, int nslots, int maxcells) {
    hipLaunchKernelGGL(k_oct_count, dim3(VSLAM_OCT_MAX_PARTS, nlevels, nslots), dim3(OCNT_T), vk_oct_count_lds(maxcells, P.maxPartCells), st,
                       cand_region, cand_stride, ncells, P, keys_a, aux_a, pts_stride);
}

void vk_octree(hipStream_t st, const uint8_t* cand_region, size_t cand_stride, int ncells, const OctParams& P,
               uint32_t* keys_a, uint32_t* aux_a, uint16_t* nid_a, void* sorted_a, size_t pts_stride,
               uint32_t* sel_xyr, int32_t* sel_cnt, int32_t* err_flag, int nlevels, int nslots, int32_t* deep_flags,
               int regkeys /* vslam_tuning.oct_regkeys: -1 by batch size, 0 | 1 forced */, int threads /* 256 | 512 | 1024 */,
               int prio /* vslam_tuning.wave_prio & 1 */) {
    const dim3 grid(nslots, nlevels);
    OctParams Pq = P;
    Pq.wavePrio = prio;
    if (P.lut && deep_flags) /* k_octree_v4 */
    {
        /* Keys in registers save walk 2 its re-read, at 32 VGPRs per thread; a 1024-thread workgroup then leaves less room
         * for foreign waves on its CU.  With several contexts in flight the neighbours matter more than the re-read
         * (default: registers only for one or two images, where latency is what counts). */
        const bool rk = regkeys < 0 ? nslots <= 2 : regkeys == 1;
        const size_t lds = (size_t)P.fineLdsOff + (size_t)P.fineLdsBytes;
        /* Threads per problem.  1024 finish a single frame's level-0 problem soonest (and hold its keys in registers); in a
         * batch every (slot, level) problem has a workgroup of its own anyway, and a SMALL workgroup leaves the CU's wave
         * slots and issue cycles to the other contexts' kernels: 256 instead of 1024 threads is +4 % mono, +7 % stereo,
         * +13 % mono at 2000 features in the pipeline; 1080p, with 65 k keys on level 0, wants 512 (+2 %; 256: -2 %) */
        const int th = rk ? 1024 : (threads == 256 || threads == 512) ? threads : 1024;
#define OCT4_LAUNCH(RK, TH, PRE)                                                                                          \
    hipLaunchKernelGGL((k_octree_v4<RK, TH, PRE>), grid, dim3(TH), lds, st, cand_region, cand_stride, ncells, Pq, keys_a, aux_a, \
                       (uint2*)sorted_a, pts_stride, sel_xyr, sel_cnt, err_flag, deep_flags)
        if (Pq.parts) { /* walk 1 ran as k_oct_count: the keys are not in registers, whatever the batch size */
            const int tp = rk ? 1024 : th;
            if (tp == 256) OCT4_LAUNCH(false, 256, true);
            else if (tp == 512) OCT4_LAUNCH(false, 512, true);
            else OCT4_LAUNCH(false, 1024, true);
        } else if (rk) OCT4_LAUNCH(true, 1024, false);
        else if (th == 256) OCT4_LAUNCH(false, 256, false);
        else if (th == 512) OCT4_LAUNCH(false, 512, false);
        else OCT4_LAUNCH(false, 1024, false);
#undef OCT4_LAUNCH
    }
    else
        hipLaunchKernelGGL(k_octree_v2, grid, dim3(OT), vk_octree_lds_bytes(P.maxNodes), st, cand_region, cand_stride, ncells, Pq,
                           keys_a, nid_a, pts_stride, sel_xyr, sel_cnt, err_flag);
}

void vk_assign_out(hipStream_t st, const OctParams& P, const PyramidGeom& g, uint32_t* sel_xyr, int32_t* sel_cnt, int lap0,
                   int lap1, SelKp* sel, int32_t* slot_counts, int cap, int32_t* err_flag, int nslots,
                   const int32_t* deep_flags, int prio) {
    OctParams Pq = P;
    Pq.wavePrio = prio;
    hipLaunchKernelGGL(k_assign_out, dim3(nslots), dim3(AO_T), 0, st, Pq, g, sel_xyr, sel_cnt, lap0, lap1, sel,
                       slot_counts, cap, err_flag, deep_flags);
}

/* hipFuncAttributeMaxDynamicSharedMemorySize is a property of the FUNCTION, shared by every context of the process: only
 * ever raise it (a small context created after a large one must not lower the limit the large one launches with) */
int vk_octree_set_max_lds(size_t bytes) {
    /* ... and of the DEVICE: hipFuncSetAttribute acts on the current device, so the raised limit is remembered per
     * device (a context on device 1 created after one on device 0 must set it again) */
    static std::mutex mu;
    static size_t have_dev[64] = {0};
    std::lock_guard<std::mutex> lk(mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return (int)hipErrorInvalidDevice;
    size_t& have = have_dev[dev];
    if (bytes <= have) return 0;
    const void* fns[8] = {(const void*)k_octree_v2, (const void*)k_octree_v4<true, 1024>, (const void*)k_octree_v4<false, 1024>,
                          (const void*)k_octree_v4<false, 512>, (const void*)k_octree_v4<false, 256>,
                          (const void*)k_octree_v4<false, 1024, true>, (const void*)k_octree_v4<false, 512, true>,
                          (const void*)k_octree_v4<false, 256, true>};
    for (int i = 0; i < 8; i++) {
        const int rc = (int)hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (rc) return rc;
    }
    have = bytes;
    return 0;
}
