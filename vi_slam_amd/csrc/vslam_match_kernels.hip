/* vslam_match_kernels.hip -- stereo L<->R matching (Frame::ComputeStereoMatches, frame.cpp:823-997) and
 * helpers for the host-replayed matchers.  Integer stages exact; the float tail (parabola, disparity,
 * depth) uses explicit round-to-nearest intrinsics, no fusion.
 */
#include "vslam_kernels.h"
#include "vslam_wave.h"
#include <cstring>
#include <cstdlib>

__device__ __forceinline__ const uint8_t* level_base2(const uint8_t* pyr, size_t slot_stride,
                                                      const BatchSrc& src, const LevelGeom& lg, int level,
                                                      int slot, int* pitch) {
    if (level == 0) {
        *pitch = (int)src.pitch0[slot];
        return src.l0[slot];
    }
    *pitch = lg.pitch;
    return pyr + (size_t)slot * slot_stride + lg.off;
}

/* keypoint counts of a stereo job live in HBM (written by k_assign_out or uploaded by the host quadtree
 * path), so the whole frame can be enqueued before the host knows them */
#define JNL(j) (*(j).cntL)
#define JNR(j) (*(j).cntR)

__device__ __forceinline__ int refl101(int p, int len) {
    if (p < 0) p = -p;
    if (p >= len) p = 2 * (len - 1) - p;
    return min(max(p, 0), len - 1);
}

/* ------------------------------------------------------------------------------------------------
 * S1  the row table and the masked Hamming arg-min (frame.cpp:840-906).
 *
 * The reference buckets the right keypoints by image row -- keypoint iR is pushed onto vRowIndices[yi] for every row
 * yi of its band [floor(yR - 2 s_R), ceil(yR + 2 s_R)] -- and a left keypoint only looks at the bucket of its own
 * row: ~35 candidates out of 2000.  The first version of this kernel tested ALL pairs (4 M gates per KITTI stereo
 * frame, 171 us per 16 frames under load); this one builds the same buckets:
 *   k_stereo_rows  one workgroup per stereo pair: counting sort of (row, iR) entries in LDS -> rowStart[row], items[]
 *                  (order inside a bucket is free: the arg-min key dist << 16 | iR picks the lowest iR on ties, which
 *                  is what "first candidate wins" means for buckets filled in iR order), and a compact {uR, octave}
 *                  record per right keypoint;
 *   k_stereo_best  one WAVE per left keypoint, lane = candidate of its row bucket: gates |octR - octL| <= 1 and
 *                  uL - maxD <= uR <= uL, 256-bit Hamming distance for the lanes that pass, wave minimum.
 * ---------------------------------------------------------------------------------------------- */
#ifndef SROWS_T
#define SROWS_T 1024
#endif
__global__ void __launch_bounds__(SROWS_T)
k_stereo_rows(StereoJobs jobs, PyramidGeom g, int nrows, int max_band, uint32_t* row_start /* [job][nrows + 1] */,
              uint16_t* items /* [job][cap * max_band] */, float2* rattr /* [job][cap] */, int cap /* of the RIGHT context */,
              int prio) {
    extern __shared__ uint32_t s_cnt[]; /* nrows + 1 */
    wave_prio_raise(prio);
    __shared__ uint32_t s_w32[SROWS_T / 64];
    const StereoJob jb = jobs.job[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nR = min(JNR(jb), cap);
    uint32_t* rs = row_start + (size_t)blockIdx.x * (nrows + 1);
    uint16_t* it = items + (size_t)blockIdx.x * cap * max_band;
    float2* ra = rattr + (size_t)blockIdx.x * cap;
    for (int r = tid; r <= nrows; r += SROWS_T) s_cnt[r] = 0;
    __syncthreads();
    for (int iR = tid; iR < nR; iR += SROWS_T) {
        const vslam_kp k = jb.kpsR[iR];
        const float r = __fmul_rn(2.0f, g.lv[k.octave].scale);
        const int maxr = min((int)ceilf(__fadd_rn(k.y, r)), nrows - 1), minr = max((int)floorf(__fsub_rn(k.y, r)), 0);
        for (int yi = minr; yi <= maxr; yi++) atomicAdd(&s_cnt[yi], 1u);
        ra[iR] = make_float2(k.x, __int_as_float(k.octave));
    }
    __syncthreads();
    /* exclusive scan over the rows: thread t owns rows [t*E, (t+1)*E) */
    const int E = (nrows + SROWS_T) / SROWS_T;
    uint32_t sum = 0;
    for (int j = 0; j < E; j++) {
        const int r = tid * E + j;
        if (r < nrows) sum += s_cnt[r];
    }
    uint32_t inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) s_w32[wv] = inc;
    __syncthreads();
    uint32_t run = inc - sum;
    for (int k = 0; k < wv; k++) run += s_w32[k];
    for (int j = 0; j < E; j++) {
        const int r = tid * E + j;
        if (r < nrows) {
            const uint32_t c = s_cnt[r];
            rs[r] = run;
            s_cnt[r] = run; /* becomes the fill cursor */
            run += c;
        }
    }
    if (tid == SROWS_T - 1) rs[nrows] = run;
    __syncthreads();
    for (int iR = tid; iR < nR; iR += SROWS_T) {
        const vslam_kp k = jb.kpsR[iR];
        const float r = __fmul_rn(2.0f, g.lv[k.octave].scale);
        const int maxr = min((int)ceilf(__fadd_rn(k.y, r)), nrows - 1), minr = max((int)floorf(__fsub_rn(k.y, r)), 0);
        for (int yi = minr; yi <= maxr; yi++) it[atomicAdd(&s_cnt[yi], 1u)] = (uint16_t)iR;
    }
}

__global__ void __launch_bounds__(256)
k_stereo_best(StereoJobs jobs, float maxD, int nrows, int max_band, const uint32_t* __restrict__ row_start,
              const uint16_t* __restrict__ items, const float2* __restrict__ rattr, uint32_t* best, int cap, int capR, int prio) {
    wave_prio_raise(prio);
    const StereoJob jb = jobs.job[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const int iL = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); /* wave-uniform, and said so */
    if (iL >= JNL(jb)) return;
    const uint32_t* rs = row_start + (size_t)blockIdx.y * (nrows + 1);
    const uint16_t* it = items + (size_t)blockIdx.y * capR * max_band;
    const float2* ra = rattr + (size_t)blockIdx.y * capR;
    const vslam_kp k = jb.kpsL[iL];             /* wave-uniform: scalar loads */
    const uint4 qa = ((const uint4*)jb.descL)[(size_t)iL * 2], qb = ((const uint4*)jb.descL)[(size_t)iL * 2 + 1];
    const int row = (int)k.y;                   /* vRowIndices[vL]: float -> size_t truncation */
    const float minU = __fsub_rn(k.x, maxD), maxU = __fsub_rn(k.x, 0.f);
    uint32_t bestKey = 100u << 16;              /* bestDist = TH_HIGH, bestIdxR = 0 (frame.cpp:879-880) */
    if (!(maxU < 0) && row >= 0 && row < nrows) { /* frame.cpp:876-877 */
        const uint32_t s = rs[row], e = rs[row + 1];
        for (uint32_t c0 = s; c0 < e; c0 += 64) { /* wave-uniform trip count: one turn for nearly every keypoint */
            const uint32_t c = c0 + lane;
            if (c < e) {
                const int iR = it[c];
                const float2 a = ra[iR];
                const int oR = __float_as_int(a.y);
                if (oR >= k.octave - 1 && oR <= k.octave + 1 && a.x >= minU && a.x <= maxU) {
                    const uint4 ta = ((const uint4*)jb.descR)[(size_t)iR * 2], tb = ((const uint4*)jb.descR)[(size_t)iR * 2 + 1];
                    const uint32_t d = __popc(qa.x ^ ta.x) + __popc(qa.y ^ ta.y) + __popc(qa.z ^ ta.z) +
                                       __popc(qa.w ^ ta.w) + __popc(qb.x ^ tb.x) + __popc(qb.y ^ tb.y) +
                                       __popc(qb.z ^ tb.z) + __popc(qb.w ^ tb.w);
                    bestKey = min(bestKey, (d << 16) | (uint32_t)iR); /* dist < bestDist, first (lowest) iR on ties */
                }
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bestKey = min(bestKey, (uint32_t)__shfl_xor((int)bestKey, o, 64));
    if (lane == 0) best[(size_t)blockIdx.y * cap + iL] = bestKey;
}

/* ------------------------------------------------------------------------------------------------
 * S2  sub-pixel refinement (frame.cpp:908-980): 11x11 centre-subtracted L1 patch distance for
 *     incR in [-5,5] at the left keypoint's octave, parabola fit, disparity / depth.
 *     16 lanes per keypoint: lane s < 11 evaluates shift s-5.
 * ---------------------------------------------------------------------------------------------- */
__global__ void __launch_bounds__(256)
k_stereo_refine(StereoJobs jobs, PyramidGeom g, const uint8_t* pyrL, size_t strideL, BatchSrc srcL,
                const uint8_t* pyrR, size_t strideR, BatchSrc srcR, float mbf, float maxD,
                const uint32_t* best, float* uRight, float* depth, int32_t* sad, int cap, int prio) {
    __shared__ float s_d[16][12];
    wave_prio_raise(prio);
    const StereoJob jb = jobs.job[blockIdx.y];
    const int grp = threadIdx.x >> 4, sub = threadIdx.x & 15;
    const int iL = blockIdx.x * 16 + grp;
    const bool live = iL < JNL(jb);
    const size_t o = (size_t)blockIdx.y * cap + (live ? iL : 0);
    uint32_t key = live ? best[o] : 0xFFFFFFFFu;
    const int bestDist = (int)(key >> 16);
    const bool refine = live && bestDist < 75; /* thOrbDist = (TH_HIGH+TH_LOW)/2, frame.cpp:828 */
    vslam_kp kL;
    float scaleduL = 0, scaledvL = 0, scaleduR0 = 0;
    bool inside = false;
    int oct = 0;
    if (refine) {
        kL = jb.kpsL[iL];
        oct = kL.octave;
        const float uR0 = jb.kpsR[key & 0xFFFF].x;
        const float sf = __fdiv_rn(1.0f, g.lv[oct].scale); /* mvInvScaleFactors[octave] = 1.0f/scale */
        scaleduL = roundf(__fmul_rn(kL.x, sf));
        scaledvL = roundf(__fmul_rn(kL.y, sf));
        scaleduR0 = roundf(__fmul_rn(uR0, sf));
        const float iniu = __fsub_rn(__fadd_rn(scaleduR0, 5.f), 5.f);
        const float endu = __fadd_rn(__fadd_rn(__fadd_rn(scaleduR0, 5.f), 5.f), 1.f);
        inside = !(iniu < 0 || endu >= (float)g.lv[oct].w);
    }
    /* The two patches go through LDS: the 16 lanes of a keypoint stage its left 11x11 patch (rows of 12 bytes) and the
     * right 11x21 strip (rows of 24 bytes) with unaligned dword loads -- 99 loads per keypoint instead of the 11 x 242
     * single-byte gathers of the first version -- and lane s then sums |(l - cL) - (r - cR_s)| = |l - (r + k_s)|,
     * k_s = cL - cR_s, over its shifted 11x11 window with v_sad_u16 on biased 16-bit pairs (+256 keeps r + k_s
     * non-negative): one instruction per two pixels.  Patches that touch the image border (reflected by the
     * reference's 19-px pyramid border) are staged pixel by pixel. */
    __shared__ __align__(4) uint8_t s_pl[16][11 * 12];
    __shared__ __align__(4) uint8_t s_pr[16][11 * 24];
    const bool work = refine && inside;
    if (work) {
        const LevelGeom lg = g.lv[oct];
        int pL, pR;
        const uint8_t* imL = level_base2(pyrL, strideL, srcL, lg, oct, jb.slotL, &pL);
        const uint8_t* imR = level_base2(pyrR, strideR, srcR, lg, oct, jb.slotR, &pR);
        const int cxL = (int)scaleduL, cy = (int)scaledvL, cxR0 = (int)scaleduR0;
        /* interior: every byte a dword row load touches lies inside the row (left 12 bytes from cxL-5, right 24 from cxR0-10) */
        const bool interior = cy - 5 >= 0 && cy + 5 < lg.h && cxL - 5 >= 0 && cxL + 7 <= lg.w && cxR0 - 10 >= 0 && cxR0 + 14 <= lg.w;
        if (interior) {
            for (int i = sub; i < 99; i += 16) { /* 33 left dwords, then 66 right dwords */
                if (i < 33) {
                    const int row = i / 3, dw = i - row * 3;
                    *(uint32_t*)&s_pl[grp][row * 12 + dw * 4] = *(const uint32_t*)(imL + mad24u((uint32_t)(cy - 5 + row), (uint32_t)pL, (uint32_t)(cxL - 5 + dw * 4)));
                } else {
                    const int j = i - 33, row = j / 6, dw = j - row * 6;
                    *(uint32_t*)&s_pr[grp][row * 24 + dw * 4] = *(const uint32_t*)(imR + mad24u((uint32_t)(cy - 5 + row), (uint32_t)pR, (uint32_t)(cxR0 - 10 + dw * 4)));
                }
            }
        } else {
            for (int i = sub; i < 11 * 11 + 11 * 21; i += 16) {
                if (i < 121) {
                    const int row = i / 11, c = i - row * 11;
                    s_pl[grp][row * 12 + c] = imL[(size_t)refl101(cy - 5 + row, lg.h) * pL + refl101(cxL - 5 + c, lg.w)];
                } else {
                    const int j = i - 121, row = j / 21, c = j - row * 21;
                    s_pr[grp][row * 24 + c] = imR[(size_t)refl101(cy - 5 + row, lg.h) * pR + refl101(cxR0 - 10 + c, lg.w)];
                }
            }
        }
    }
    __syncthreads();
    if (work && sub < 11) {
        const uint8_t* PLp = s_pl[grp];
        const uint8_t* PRp = s_pr[grp];
        const int k = (int)PLp[5 * 12 + 5] - (int)PRp[5 * 24 + sub + 5]; /* cL - cR of this shift */
        const uint32_t kb = (uint32_t)(k + 256) * 0x00010001u, lb = 256u * 0x00010001u;
        const uint32_t sh = (uint32_t)(sub & 3);
        const int rdw = sub >> 2; /* first right dword of the lane's window */
        uint32_t acc = 0;
#pragma unroll
        for (int row = 0; row < 11; row++) {
            const uint32_t* lw = (const uint32_t*)(PLp + row * 12);
            const uint32_t* rw = (const uint32_t*)(PRp + row * 24) + rdw;
            const uint32_t l0 = lw[0], l1 = lw[1], l2 = lw[2];
            const uint32_t a0 = rw[0], a1 = rw[1], a2 = rw[2], a3 = rw[3];
            const uint32_t r0 = __builtin_amdgcn_alignbyte(a1, a0, sh), r1 = __builtin_amdgcn_alignbyte(a2, a1, sh),
                           r2 = __builtin_amdgcn_alignbyte(a3, a2, sh);
            /* pixels 0..10 as u16 pairs: (0,1) (2,3) | (4,5) (6,7) | (8,9) (10,-); the eleventh pair repeats pixel 10 on
             * BOTH sides with the same offset, and its second half is made equal on both sides -> contributes |l - r - k| once */
#define SAD_PAIR(LW, RW, SEL)                                                                                  \
    acc = __builtin_amdgcn_sad_u16(__builtin_amdgcn_perm(0u, (LW), (SEL)) + lb, __builtin_amdgcn_perm(0u, (RW), (SEL)) + kb, acc);
            SAD_PAIR(l0, r0, 0x0c010c00u)
            SAD_PAIR(l0, r0, 0x0c030c02u)
            SAD_PAIR(l1, r1, 0x0c010c00u)
            SAD_PAIR(l1, r1, 0x0c030c02u)
            SAD_PAIR(l2, r2, 0x0c010c00u)
            {   /* pixel 10 alone: low half = the pixel, high half = 0 on the left and -k_bias-compensated on the right */
                const uint32_t lp = (__builtin_amdgcn_perm(0u, l2, 0x0c0c0c02u) + 256u);
                const uint32_t rp = (__builtin_amdgcn_perm(0u, r2, 0x0c0c0c02u) + (uint32_t)(k + 256));
                acc = __builtin_amdgcn_sad_u16(lp, rp, acc); /* high halves are both 0 */
            }
#undef SAD_PAIR
        }
        s_d[grp][sub] = (float)acc;
    }
    __syncthreads();
    if (!live || sub != 0) return;
    float ur = -1.f, dp = -1.f;
    int sd = -1;
    if (refine && inside) {
        int bestSad = 0x7FFFFFFF, bestinc = 0;
        for (int i = 0; i < 11; i++) {
            const float dist = s_d[grp][i];
            if (dist < (float)bestSad) {
                bestSad = (int)dist;
                bestinc = i - 5;
            }
        }
        if (bestinc != -5 && bestinc != 5) {
            const float dist1 = s_d[grp][5 + bestinc - 1], dist2 = s_d[grp][5 + bestinc],
                        dist3 = s_d[grp][5 + bestinc + 1];
            const float num = __fsub_rn(dist1, dist3);
            const float den = __fmul_rn(2.0f, __fsub_rn(__fadd_rn(dist1, dist3), __fmul_rn(2.0f, dist2)));
            const float deltaR = __fdiv_rn(num, den);
            if (!(deltaR < -1 || deltaR > 1)) {
                float bestuR = __fmul_rn(g.lv[oct].scale,
                                         __fadd_rn(__fadd_rn(scaleduR0, (float)bestinc), deltaR));
                float disparity = __fsub_rn(kL.x, bestuR);
                if (disparity >= 0 && disparity < maxD) {
                    if (disparity <= 0) {
                        disparity = 0.01f;
                        bestuR = (float)((double)kL.x - 0.01);
                    }
                    dp = __fdiv_rn(mbf, disparity);
                    ur = bestuR;
                    sd = bestSad;
                }
            }
        }
    }
    uRight[o] = ur;
    depth[o] = dp;
    sad[o] = sd;
}

/* ------------------------------------------------------------------------------------------------
 * S3  outlier cut (frame.cpp:983-996): median of the accepted SADs (element size/2 of the ascending
 *     list) by a two-level byte histogram, then reject every match with SAD >= 1.5*1.4*median.
 *     One workgroup per stereo pair.
 * ---------------------------------------------------------------------------------------------- */
__global__ void __launch_bounds__(256)
k_stereo_median_cut(StereoJobs jobs, float* uRight, float* depth, const int32_t* sad, int cap, int prio) {
    wave_prio_raise(prio);
    /* frame.cpp:983-996: median of the accepted SADs (element size/2 of the sorted list), drop SAD >= 1.5 * 1.4 * median.
     * SADs are < 65536 (frame.cpp:919-949: 120 pixels x 255 + bias), so the median is found by two 256-bin histograms --
     * high byte, then the low byte inside the selected bin -- and the bin holding rank k is found by a PREFIX SUM over
     * the bins (thread = bin), not by one thread walking them (round 2: 2 x 256 dependent LDS reads by thread 0, most of
     * the kernel's 16 us; and a 1024-thread workgroup that waited for a whole CU under load: 43 us). */
    __shared__ int s_hist[256];
    __shared__ int s_wsum[4];
    __shared__ int s_sel, s_rank, s_total;
    const StereoJob jb = jobs.job[blockIdx.x];
    const size_t base = (size_t)blockIdx.x * cap;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nl = JNL(jb);
    /* bin b holds rank k  <=>  prefix(b) <= k < prefix(b) + hist[b]: exactly one thread finds it */
    auto select = [&](int k) {
        const int h = s_hist[tid];
        int inc = h;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        if (lane == 63) s_wsum[wv] = inc;
        __syncthreads();
        int pre = inc - h;
        for (int w = 0; w < wv; w++) pre += s_wsum[w];
        if (h > 0 && pre <= k && k < pre + h) {
            s_sel = tid;
            s_rank = k - pre;
        }
        __syncthreads();
    };
    s_hist[tid] = 0;
    if (tid == 0) s_total = 0;
    __syncthreads();
    int mine = 0;
    for (int i = tid; i < nl; i += 256) {
        const int s = sad[base + i];
        if (s >= 0) {
            atomicAdd(&s_hist[(s >> 8) & 255], 1);
            mine++;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if (lane == 0 && mine) atomicAdd(&s_total, mine);
    __syncthreads();
    const int total = s_total;
    if (total == 0) return; /* the reference indexes an empty vector here (UB); nothing to cut */
    select(total / 2);
    const int hi = s_sel, rank = s_rank;
    __syncthreads();
    s_hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < nl; i += 256) {
        const int s = sad[base + i];
        if (s >= 0 && ((s >> 8) & 255) == hi) atomicAdd(&s_hist[s & 255], 1);
    }
    __syncthreads();
    select(rank);
    const float median = (float)((hi << 8) | s_sel);
    const float thDist = __fmul_rn(1.5f * 1.4f, median);
    for (int i = tid; i < nl; i += 256) {
        const int s = sad[base + i];
        if (s >= 0 && !((float)s < thDist)) {
            uRight[base + i] = -1.f;
            depth[base + i] = -1.f;
        }
    }
}

/* gather 32-byte descriptor rows by index (octave-0 subsets for SearchForInitialization) */
__global__ void k_gather_rows32(const uint8_t* __restrict__ src, const int32_t* __restrict__ idx, int n,
                                uint8_t* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * 2) return;
    ((uint4*)dst)[i] = ((const uint4*)src)[(size_t)idx[i >> 1] * 2 + (i & 1)];
}

/* ------------------------------------------------------------------------------------------------ */
size_t vk_stereo_rows_bytes(int njobs, int nrows, int max_band, int cap) {
    return (size_t)njobs * ((((size_t)nrows + 1) * 4 + 15) / 16 * 16 + (((size_t)cap * max_band * 2 + 15) & ~(size_t)15) + (size_t)cap * 8);
}

void vk_stereo(hipStream_t st, const StereoJobs& jobs, int njobs, int maxNL, int maxNR, const PyramidGeom& g,
               const uint8_t* pyrL, size_t strideL, const BatchSrc& srcL, const uint8_t* pyrR, size_t strideR,
               const BatchSrc& srcR, float mbf, float maxD, uint32_t* best, float* uRight, float* depth,
               int32_t* sad, int cap, int max_band, uint8_t* rows_scratch, int prio) {
    if (njobs <= 0 || maxNL <= 0) return;
    const int capR = maxNR; /* capacity of the right context's slots */
    const int nrows = g.lv[0].h;
    /* scratch: row_start[njobs][nrows + 1] | items[njobs][cap * max_band] | rattr[njobs][cap] */
    uint32_t* row_start = (uint32_t*)rows_scratch;
    size_t off = (size_t)njobs * ((((size_t)nrows + 1) * 4 + 15) / 16 * 16);
    row_start = (uint32_t*)rows_scratch;
    uint16_t* items = (uint16_t*)(rows_scratch + off);
    off += (size_t)njobs * (((size_t)capR * max_band * 2 + 15) & ~(size_t)15);
    float2* rattr = (float2*)(rows_scratch + off);
    hipLaunchKernelGGL(k_stereo_rows, dim3(njobs), dim3(SROWS_T), ((size_t)nrows + 1) * 4, st, jobs, g, nrows, max_band,
                       row_start, items, rattr, capR, prio);
    hipLaunchKernelGGL(k_stereo_best, dim3((maxNL + 3) / 4, njobs), dim3(256), 0, st, jobs, maxD, nrows, max_band, row_start,
                       items, rattr, best, cap, capR, prio);
    hipLaunchKernelGGL(k_stereo_refine, dim3((maxNL + 15) / 16, njobs), dim3(256), 0, st, jobs, g, pyrL, strideL,
                       srcL, pyrR, strideR, srcR, mbf, maxD, best, uRight, depth, sad, cap, prio);
    hipLaunchKernelGGL(k_stereo_median_cut, dim3(njobs), dim3(256), 0, st, jobs, uRight, depth, sad, cap, prio);
}

void vk_gather_rows32(hipStream_t st, const uint8_t* src, const int32_t* idx, int n, uint8_t* dst) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_gather_rows32, dim3((2 * n + 255) / 256), dim3(256), 0, st, src, idx, n, dst);
}

/* ------------------------------------------------------------------------------------------------
 * batched forms for SearchForInitialization: gather the octave-0 descriptor rows of every pair, then one
 * launch computes all dense distance matrices (grid.z = pair).
 * ---------------------------------------------------------------------------------------------- */
__global__ void k_gather_rows32_batch(MatJobs jobs, const int32_t* __restrict__ idx, uint8_t* __restrict__ tmp) {
    const MatJob jb = jobs.job[blockIdx.y >> 1];
    const int side = blockIdx.y & 1;
    const int n = side ? jb.nc : jb.nr;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * 2) return;
    const uint8_t* src = side ? jb.desc2 : jb.desc1;
    const int32_t* id = idx + (side ? jb.idx_off2 : jb.idx_off1);
    uint4* dst = (uint4*)(tmp + (size_t)(side ? jb.t_off : jb.q_off) * 32);
    dst[i] = ((const uint4*)src)[(size_t)id[i >> 1] * 2 + (i & 1)];
}

__global__ void __launch_bounds__(256)
k_hamming_matrix_batch(MatJobs jobs, const uint8_t* __restrict__ tmp, uint8_t* __restrict__ out) {
    __shared__ uint4 s_t[256 * 2];
    const MatJob jb = jobs.job[blockIdx.z];
    const int nq = jb.nr, nt = jb.nc;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int qi = blockIdx.x * 64 + lane;
    const int t0 = blockIdx.y * 256;
    if (blockIdx.x * 64 >= nq || t0 >= nt) return; /* block-uniform */
    const uint4* q = (const uint4*)(tmp + (size_t)jb.q_off * 32);
    const uint4* t = (const uint4*)(tmp + (size_t)jb.t_off * 32);
    for (int i = tid; i < 512; i += 256) {
        const int ti = t0 + (i >> 1);
        s_t[i] = ti < nt ? t[(size_t)ti * 2 + (i & 1)] : make_uint4(0, 0, 0, 0);
    }
    uint4 qa = make_uint4(0, 0, 0, 0), qb = qa;
    if (qi < nq) {
        qa = q[(size_t)qi * 2];
        qb = q[(size_t)qi * 2 + 1];
    }
    __syncthreads();
    if (qi >= nq) return;
    uint8_t* o = out + jb.out_off + (size_t)qi * nt;
    for (int j = wv * 64; j < wv * 64 + 64; j++) {
        const int ti = t0 + j;
        if (ti >= nt) break;
        const uint4 ta = s_t[2 * j], tb = s_t[2 * j + 1];
        const int d = __popc(qa.x ^ ta.x) + __popc(qa.y ^ ta.y) + __popc(qa.z ^ ta.z) + __popc(qa.w ^ ta.w) +
                      __popc(qb.x ^ tb.x) + __popc(qb.y ^ tb.y) + __popc(qb.z ^ tb.z) + __popc(qb.w ^ tb.w);
        o[ti] = (uint8_t)min(d, 255);
    }
}

void vk_hamming_matrix_batch(hipStream_t st, const MatJobs& jobs, int njobs, int maxr, int maxc, const int32_t* idx,
                             uint8_t* tmp, uint8_t* out) {
    if (njobs <= 0 || maxr <= 0 || maxc <= 0) return;
    const int mx = maxr > maxc ? maxr : maxc;
    hipLaunchKernelGGL(k_gather_rows32_batch, dim3((2 * mx + 255) / 256, 2 * njobs), dim3(256), 0, st, jobs, idx, tmp);
    hipLaunchKernelGGL(k_hamming_matrix_batch, dim3((maxr + 63) / 64, (maxc + 255) / 256, njobs), dim3(256), 0, st,
                       jobs, tmp, out);
}

/* ------------------------------------------------------------------------------------------------
 * pack result slots for the RCCL exchange (ring shift, vi_slam_amd/dist.py): per slot  int32 n, mono, cap, 0 | vslam_kp[cap] | desc[cap][32].
 * Counts are read from HBM, so packing can be enqueued before the host knows them.
 * ---------------------------------------------------------------------------------------------- */
__global__ void __launch_bounds__(256)
k_pack_slots(const vslam_kp* __restrict__ kps, const uint8_t* __restrict__ desc, const int32_t* __restrict__ counts,
             int cap, int first, uint8_t* __restrict__ dst, size_t slot_bytes, int prio) {
    wave_prio_raise(prio);
    const int slot = first + blockIdx.y;
    const int n = min(counts[slot * 4], cap);
    uint8_t* d = dst + (size_t)blockIdx.y * slot_bytes;
    const int tid = blockIdx.x * 256 + threadIdx.x, nth = gridDim.x * 256;
    if (tid == 0) {
        int32_t* h = (int32_t*)d;
        h[0] = n;
        h[1] = counts[slot * 4 + 1];
        h[2] = cap;
        h[3] = 0;
    }
    const uint32_t* ks = (const uint32_t*)(kps + (size_t)slot * cap);
    uint32_t* kd = (uint32_t*)(d + 16);
    for (int i = tid; i < n * 7; i += nth) kd[i] = ks[i];
    const uint4* ds = (const uint4*)(desc + (size_t)slot * cap * 32);
    uint4* dd = (uint4*)(d + 16 + (size_t)cap * sizeof(vslam_kp));
    for (int i = tid; i < n * 2; i += nth) dd[i] = ds[i];
}

void vk_pack_slots(hipStream_t st, const vslam_kp* kps, const uint8_t* desc, const int32_t* counts, int cap, int first,
                   int nslots, uint8_t* dst, size_t slot_bytes, int prio) {
    if (nslots <= 0) return;
    hipLaunchKernelGGL(k_pack_slots, dim3(8, nslots), dim3(256), 0, st, kps, desc, counts, cap, first, dst, slot_bytes, prio);
}

/* --------------------------------------------------------------------------------------------------
 * Result delivery without the runtime's copy path: a kernel stores up to four device ranges straight into
 * pinned, device-visible host memory (hipHostMalloc).  One ~4 us launch replaces several hipMemcpyAsync
 * calls, each of which costs ~60 us of host time on this stack -- the host thread, not the GPU, was the
 * limit of the batched pipeline.  Ranges are dword-granular; uint4 when both ends are 16-byte aligned.
 * Also used with src == nullptr to zero device ranges (replaces hipMemsetAsync).
 * ---------------------------------------------------------------------------------------------- */
__global__ void __launch_bounds__(256) k_copy_ranges(CopyRanges R) {
    const int tid = blockIdx.x * 256 + threadIdx.x, nth = gridDim.x * 256;
    for (int r = 0; r < R.n; r++) {
        const size_t bytes = R.bytes[r];
        uint8_t* dst = (uint8_t*)R.dst[r];
        const uint8_t* src = (const uint8_t*)R.src[r];
        if ((((uintptr_t)dst | (uintptr_t)src) & 15) == 0) {
            const size_t n16 = bytes >> 4;
            if (src) {
                size_t i = tid;
                for (; i + 7 * (size_t)nth < n16; i += 8 * (size_t)nth) { /* eight 16-byte loads in flight per lane */
                    uint4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) v[u] = ((const uint4*)src)[i + (size_t)u * nth];
#pragma unroll
                    for (int u = 0; u < 8; u++) ((uint4*)dst)[i + (size_t)u * nth] = v[u];
                }
                for (; i < n16; i += nth) ((uint4*)dst)[i] = ((const uint4*)src)[i];
            } else
                for (size_t i = tid; i < n16; i += nth) ((uint4*)dst)[i] = make_uint4(0, 0, 0, 0);
            const size_t done = n16 << 4, tail = (bytes - done) >> 2;
            for (size_t i = tid; i < tail; i += nth)
                ((uint32_t*)(dst + done))[i] = src ? ((const uint32_t*)(src + done))[i] : 0u;
        } else {
            const size_t n4 = bytes >> 2;
            for (size_t i = tid; i < n4; i += nth) ((uint32_t*)dst)[i] = src ? ((const uint32_t*)src)[i] : 0u;
        }
    }
}

int vk_copy_ranges(hipStream_t st, const CopyRanges& R, const vslam_tuning& T) {
    size_t total = 0;
    for (int r = 0; r < R.n; r++) total += R.bytes[r];
    if (!total) return 0;
    /* Large transfers between HBM and pinned host memory are handed to the runtime (one hipMemcpyAsync per range on the
     * same stream; in a kernel trace they appear as __amd_rocclr_copyBuffer, ~27 us per megabyte-sized copy).  With the
     * keypoints and descriptors of a 32-frame step (2 MB) copied by the kernel below, the mono workload ran 117-124 k
     * frames/s whatever its shape (8, 16, 32 or 256 workgroups, one or eight loads in flight per lane), through the runtime
     * 131 k, without the copy 132 k.  Small ones (a single image's results, match lists: latency, captured graphs) stay
     * one ~5-us kernel, cheaper to enqueue.  VSLAM_D2H=kernel|sdma forces one route. */
    const int mode = T.d2h_route == 1 ? 1 : T.d2h_route == 2 ? 2 : 0; /* 1 copy kernel, 2 runtime copy, 0 by size */
    if (mode == 2 || (mode == 0 && total >= (256u << 10))) {
        int ops = 0;
        for (int r = 0; r < R.n; r++)
            if (R.bytes[r]) (void)hipMemcpyAsync(R.dst[r], R.src[r], R.bytes[r], hipMemcpyDefault, st), ops++;
        return ops;
    }
    /* a transfer to or from host memory is bound by the link, not by the GPU: a few workgroups with several loads in
     * flight per lane keep it busy without parking waves on every CU (VSLAM_COPY_WGS overrides the cap of 16) */
    const int cap = std::max(1, tune_or(T.copy_wgs, 16));
    const int blocks = (int)std::min<size_t>((size_t)cap, (total / 16 + 255) / 256 + 1);
    hipLaunchKernelGGL(k_copy_ranges, dim3(blocks), dim3(256), 0, st, R);
    return 1;
}

/* Host images -> level 0 of the slots' pyramids, ONE launch per batch, sized for PCIe and not for the GPU: a handful of
 * workgroups (PULL_WG_PER_IMG per image) each keep PULL_DEPTH x 16 bytes per lane in flight -- ~1 MB in flight in
 * all, several times the bandwidth-delay product of the link -- instead of thousands of short-lived workgroups whose
 * waves would sit in every CU's wave slots waiting for the bus while the other contexts' kernels queue behind them.
 * A wave copies whole rows: lane = 16-byte chunk, reading the caller's pinned memory (or the context's pinned staging)
 * and writing 16-byte aligned chunks of the 128-byte-pitched level-0 buffer.  Source rows may start at any address
 * (KITTI rows are 1241 bytes): the loads are unaligned dwordx4; only the last partial chunk of the image's LAST row is
 * read bytewise, so nothing past the caller's buffer is touched. */
#define PULL_WG_PER_IMG 2
template <int PULL_DEPTH>
__global__ void __launch_bounds__(256)
k_pull_images(BatchSrc src, uint8_t* pyr, size_t slot_stride, uint32_t off0, int dpitch, int w, int h) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slot = blockIdx.y;
    const uint8_t* simg = src.l0[slot];
    const size_t spitch = src.pitch0[slot];
    uint8_t* dimg = pyr + (size_t)slot * slot_stride + off0;
    const int nch = (w + 15) >> 4; /* 16-byte chunks per row */
    const int nitems = h * nch;    /* (row, chunk) items of the image, dealt to the waves in blocks of 64 x PULL_DEPTH */
    const int total_waves = (int)gridDim.x * 4, wid = blockIdx.x * 4 + wave;
    for (int i0 = wid * 64 * PULL_DEPTH; i0 < nitems; i0 += total_waves * 64 * PULL_DEPTH) {
        uint4 v[PULL_DEPTH];
#pragma unroll
        for (int u = 0; u < PULL_DEPTH; u++) {
            const int i = i0 + u * 64 + lane;
            v[u] = make_uint4(0, 0, 0, 0);
            if (i < nitems) {
                const int y = i / nch, ch = i - y * nch;
                const uint8_t* p = simg + (size_t)y * spitch + 16 * ch;
                if (16 * ch + 16 <= w || y < h - 1) v[u] = *(const uint4*)p; /* may run into the next row: harmless */
                else {
                    uint32_t t[4] = {0, 0, 0, 0};
                    for (int k = 0; 16 * ch + k < w; k++) t[k >> 2] |= (uint32_t)p[k] << (8 * (k & 3));
                    v[u] = make_uint4(t[0], t[1], t[2], t[3]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < PULL_DEPTH; u++) {
            const int i = i0 + u * 64 + lane;
            if (i < nitems) {
                const int y = i / nch, ch = i - y * nch;
                *(uint4*)(dimg + (size_t)y * dpitch + 16 * ch) = v[u]; /* the padding up to the pitch may be written */
            }
        }
    }
}

void vk_pull_images(hipStream_t st, const BatchSrc& src, uint8_t* pyr, size_t slot_stride, uint32_t off0, int dpitch,
                    int w, int h, int nimg, int from_host, const vslam_tuning& T) {
    /* from_host: the rows come over PCIe.  Host reads stay in the L2's miss queues for microseconds; more of them in
     * flight than the link needs (bandwidth x latency ~ 150 KB) only delays the HBM requests of the kernels that run
     * beside the pull.  From device memory (the SDMA staging buffer) the deep variant simply finishes sooner. */
    const int depth_host = tune_or(T.pull_depth, 0); /* default: shallow beside other work (batches), deep for one or two images */
    /* a batch: 2 workgroups per image; one or two images (a synchronous single-frame call): spread each over 32 */
    const dim3 grid(nimg <= 2 ? 32 : PULL_WG_PER_IMG, nimg);
    const int depth = !from_host ? 8 : depth_host ? depth_host : nimg <= 2 ? 8 : 1;
    if (depth >= 8) hipLaunchKernelGGL(k_pull_images<8>, grid, dim3(256), 0, st, src, pyr, slot_stride, off0, dpitch, w, h);
    else if (depth >= 4) hipLaunchKernelGGL(k_pull_images<4>, grid, dim3(256), 0, st, src, pyr, slot_stride, off0, dpitch, w, h);
    else if (depth >= 2) hipLaunchKernelGGL(k_pull_images<2>, grid, dim3(256), 0, st, src, pyr, slot_stride, off0, dpitch, w, h);
    else hipLaunchKernelGGL(k_pull_images<1>, grid, dim3(256), 0, st, src, pyr, slot_stride, off0, dpitch, w, h);
}

/* zero the (total, overflow) header of every slot's candidate buffer and the quadtree's 16-byte error word */
__global__ void k_reset_headers(uint8_t* d_cand, size_t stride, int nimg, int32_t* d_err) {
    const int t = threadIdx.x;
    if (t < nimg) *(uint2*)(d_cand + (size_t)t * stride) = make_uint2(0u, 0u);
    if (d_err && t < 4) d_err[t] = 0;
}

void vk_reset_headers(hipStream_t st, uint8_t* d_cand, size_t cand_stride_bytes, int nimg, int32_t* d_err) {
    hipLaunchKernelGGL(k_reset_headers, dim3(1), dim3(64), 0, st, d_cand, cand_stride_bytes, nimg, d_err);
}

/* --------------------------------------------------------------------------------------------------
 * MapPoint::ComputeDistinctiveDescriptors (mappoint.cpp:322-390) for many MapPoints at once: one workgroup per
 * MapPoint, its N observed descriptors in LDS, one thread per row of the (never materialised) N x N distance
 * matrix.  The row median sorted[int(0.5*(N-1))] is found by bisection on the value (count of entries <= v),
 * the winner by an LDS atomicMin on median << 16 | row ("first wins").
 * ---------------------------------------------------------------------------------------------- */
__global__ void __launch_bounds__(256)
k_distinctive(const uint8_t* __restrict__ desc, const int32_t* __restrict__ offsets, int32_t* __restrict__ best) {
    extern __shared__ __align__(16) uint8_t dsm[];
    uint4* sd = (uint4*)dsm;
    __shared__ uint32_t s_best;
    const int set = blockIdx.x, tid = threadIdx.x;
    const int o = offsets[set], N = offsets[set + 1] - o;
    if (N <= 0) {
        if (tid == 0) best[set] = -1;
        return;
    }
    for (int i = tid; i < 2 * N; i += 256) sd[i] = ((const uint4*)desc)[(size_t)o * 2 + i];
    if (tid == 0) s_best = 0xFFFFFFFFu;
    __syncthreads();
    const int m = (int)(0.5 * (double)(N - 1)); /* vDists[0.5*(N-1)] */
    for (int i = tid; i < N; i += 256) {
        const uint4 a = sd[2 * i], b = sd[2 * i + 1];
        int lo = 0, hi = 256; /* smallest v with #(d <= v) >= m + 1 */
        while (lo < hi) {
            const int v = (lo + hi) >> 1;
            int cnt = 0;
            for (int j = 0; j < N; j++) {
                const uint4 c = sd[2 * j], d = sd[2 * j + 1];
                const int dist = __popc(a.x ^ c.x) + __popc(a.y ^ c.y) + __popc(a.z ^ c.z) + __popc(a.w ^ c.w) +
                                 __popc(b.x ^ d.x) + __popc(b.y ^ d.y) + __popc(b.z ^ d.z) + __popc(b.w ^ d.w);
                cnt += dist <= v ? 1 : 0;
            }
            if (cnt >= m + 1) hi = v;
            else lo = v + 1;
        }
        atomicMin(&s_best, ((uint32_t)lo << 16) | (uint32_t)i);
    }
    __syncthreads();
    if (tid == 0) best[set] = (int32_t)(s_best & 0xFFFF);
}

void vk_distinctive(hipStream_t st, const uint8_t* desc, const int32_t* offsets, int nsets, int maxN, int32_t* best) {
    if (nsets <= 0) return;
    hipLaunchKernelGGL(k_distinctive, dim3(nsets), dim3(256), (size_t)std::max(maxN, 1) * 32, st, desc, offsets, best);
}

int vk_distinctive_set_max_lds(size_t bytes) {
    return (int)hipFuncSetAttribute((const void*)k_distinctive, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
