/* vslam_fe.hip -- the C ABI of include/vslam_fe.h: context, memory layout in HBM, launch sequence.
 *
 * Per batch of image slots the extractor issues, on ONE stream:
 *     [H2D level 0 | zero-copy]  ->  7 x k_resize_level  ->  k_fast_cells  ->  D2H candidates (event)
 *     ->  k_blur7 (overlaps the host quadtree)  ->  H2D selected keypoints  ->  k_orient_describe
 *     ->  D2H keypoints + descriptors
 * The quadtree distribution (FExtractor::DistributeOctTree) is sequential by construction and runs on
 * the host, one task per (slot, level), on a small worker pool.
 */
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include "vslam_ctx.h"

#include <algorithm>
#include <chrono>
#include <mutex>
#include <cstdio>
#include <cstdlib>

void vslam_host_prof_report();

std::string& vslam_err() {
    static thread_local std::string e;
    return e;
}
extern "C" const char* vslam_last_error(void) { return vslam_err().c_str(); }

static std::mutex g_gate_mu; /* the gate links between contexts (set / destroy may come from different threads) */

static void gate_unlink(vslam_fe* fe) { /* g_gate_mu held */
    if (fe->fast_gate) {
        auto& w = fe->fast_gate->gate_waiters;
        w.erase(std::remove(w.begin(), w.end(), fe), w.end());
        fe->fast_gate = nullptr;
    }
}

static void free_ctx(vslam_fe* fe) {
    if (!fe) return;
    if (fe->stream) hipStreamSynchronize(fe->stream);
    delete fe->pool;
    hipFree(fe->d_pyr);
    hipFree(fe->d_blur);
    for (int l = 0; l < VSLAM_MAX_LEVELS; l++) {
        hipFree(fe->d_xtab[l]);
        hipFree(fe->d_xa[l]);
        hipFree(fe->d_ytab[l]);
        hipFree(fe->d_yb[l]);
        hipFree(fe->d_qbase[l]);
        hipFree(fe->d_quads[l]);
    }
    for (auto& g : fe->pyr_groups) hipFree(g.d_tiles);
    hipFree(fe->d_cells);
    hipFree(fe->d_bands);
    hipFree(fe->d_oct_parts);
    hipFree(fe->d_oct_cnt);
    hipFree(fe->d_band_classes);
    hipFree(fe->d_cand);
    if (fe->h_cand) hipHostFree(fe->h_cand);
    hipFree(fe->d_blur_tasks);
    hipFree(fe->d_sel);
    if (fe->h_sel) hipHostFree(fe->h_sel);
    hipFree(fe->d_res); /* d_counts, d_kps and d_desc are views into it; h_res likewise */
    if (fe->h_res) hipHostFree(fe->h_res);
    hipFree(fe->d_pattern);
    if (fe->h_top2) hipHostFree(fe->h_top2);
    hipFree(fe->d_part);
    hipFree(fe->d_idx2);
    hipFree(fe->d_dist2);
    hipFree(fe->d_dmat);
    hipFree(fe->d_tmp_desc[0]);
    hipFree(fe->d_tmp_desc[1]);
    hipFree(fe->d_stereo);
    hipFree(fe->d_pts[0]);
    hipFree(fe->d_pts[1]);
    hipFree(fe->d_nid);
    hipFree(fe->d_oct_sorted);
    hipFree(fe->d_oct_lut);
    hipFree(fe->d_sel_xyr);
    hipFree(fe->d_oct_redo);
    hipFree(fe->d_sel_cnt);
    if (fe->h_stereo) hipHostFree(fe->h_stereo);
    if (!fe->init_in_block) hipFree(fe->d_init);
    hipFree(fe->d_init_scratch);
    hipFree(fe->d_proj);
    hipFree(fe->d_sbp);
    if (fe->h_sbp) hipHostFree(fe->h_sbp);
    hipFree(fe->d_x3dw);
    hipFree(fe->d_mpflags);
    if (fe->h_proj) hipHostFree(fe->h_proj);
    if (fe->h_img) hipHostFree(fe->h_img);
    hipFree(fe->d_stage);
    if (fe->graph_exec) hipGraphExecDestroy(fe->graph_exec);
    hipFree(fe->d_bow);
    if (fe->h_bow) hipHostFree(fe->h_bow);
    hipFree(fe->d_init_fb);
    if (fe->h_init && !fe->init_in_block) hipHostFree(fe->h_init);
    if (fe->ev_cand) hipEventDestroy(fe->ev_cand);
    if (fe->ev_x) hipEventDestroy(fe->ev_x);
    for (int i = 0; i < 4; i++)
        if (fe->ev_user[i]) hipEventDestroy(fe->ev_user[i]);
    {   /* a destroyed context gates nobody: its waiters run un-gated from now on */
        std::lock_guard<std::mutex> lk(g_gate_mu);
        gate_unlink(fe);
        for (vslam_fe* w : fe->gate_waiters) w->fast_gate = nullptr;
        fe->gate_waiters.clear();
    }
    if (fe->ev_fast) hipEventDestroy(fe->ev_fast);
    for (int i = 0; i < 10; i++)
        if (fe->ev_prof[i]) hipEventDestroy(fe->ev_prof[i]);
    if (fe->stream) hipStreamDestroy(fe->stream);
    delete fe;
}

extern "C" void vslam_fe_destroy(vslam_fe* fe) {
    if (fe) hipSetDevice(fe->p.device);
    vslam_host_prof_report();
    free_ctx(fe);
}

template <typename T>
static int upload(T** dst, const void* src, size_t bytes) {
    HIPCHK(hipMalloc((void**)dst, bytes ? bytes : 4));
    if (bytes) HIPCHK(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    return VSLAM_OK;
}

static int create_impl(const vslam_fe_params* pp, vslam_fe* fe) {
    const vslam_fe_params& p = fe->p;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || p.device >= ndev) {
        g_err = "no usable HIP device (this library has no CPU fallback)";
        return VSLAM_ERR_NO_DEVICE;
    }
    HIPCHK(hipSetDevice(p.device));
    (void)pp;
    vslam::build_tables(p.nfeatures, p.scale_factor, p.nlevels, fe->tab);
    fe->B = p.max_batch;
    fe->cap = (p.nfeatures + 4 * p.nlevels + 8 + 3) & ~3; /* multiple of 4: packed descriptors stay 16-B aligned */
    static const int32_t def_taps[7] = {18, 34, 48, 56, 48, 34, 18};
    bool zero = true;
    for (int i = 0; i < 7; i++) zero = zero && p.gauss_taps[i] == 0;
    for (int i = 0; i < 7; i++) fe->taps[i] = zero ? def_taps[i] : p.gauss_taps[i];

    /* geometry: level sizes, pitches, offsets (level 0 staging included at offset 0) */
    memset(&fe->geom, 0, sizeof(fe->geom));
    fe->geom.nlevels = p.nlevels;
    size_t off = 0;
    for (int l = 0; l < p.nlevels; l++) {
        int lw, lh;
        vslam::level_size(fe->tab, p.width, p.height, l, &lw, &lh);
        LevelGeom& g = fe->geom.lv[l];
        g.w = lw;
        g.h = lh;
        g.pitch = (lw + 127) & ~127;
        g.off = (uint32_t)off;
        g.scale = fe->tab.scale[l];
        off += (size_t)g.pitch * lh;
        off = (off + 255) & ~(size_t)255;
        if (lw < 40 || lh < 40 || lw > 4095 + 32 || lh > 4095 + 32) {
            g_err = "image too small/large for this level count";
            return VSLAM_ERR_UNSUPPORTED;
        }
        /* DistributeOctTree needs nIni = round(W/H) >= 1 (fextractor.cpp:534) */
        if ((int)std::round((float)(lw - 32) / (float)(lh - 32)) < 1) {
            g_err = "portrait aspect ratio: the reference's DistributeOctTree has nIni == 0 (undefined)";
            return VSLAM_ERR_UNSUPPORTED;
        }
    }
    fe->slot_stride = off;
    {
        /* exact bound of what DistributeOctTree can return: a level ends with at most N_l + 2 nodes, except that
         * the first pass splits all nIni initial nodes unconditionally (up to 4*nIni).  Only matters for very
         * small nfeatures; the documented formula stays the floor so common configurations keep their layout. */
        int bound = 0;
        for (int l = 0; l < p.nlevels; l++) {
            const int W = fe->geom.lv[l].w - 2 * VSLAM_FAST_BORDER, H = fe->geom.lv[l].h - 2 * VSLAM_FAST_BORDER;
            const int nIni = (int)std::round((float)W / (float)H);
            bound += std::max(fe->tab.quota[l] + 3, 4 * nIni);
        }
        fe->cap = std::max(fe->cap, (bound + 3) & ~3);
    }
    HIPCHK(hipMalloc((void**)&fe->d_pyr, fe->slot_stride * fe->B));
    HIPCHK(hipMalloc((void**)&fe->d_blur, fe->slot_stride * fe->B));
    HIPCHK(hipMemset(fe->d_pyr, 0, fe->slot_stride * fe->B));
    HIPCHK(hipMemset(fe->d_blur, 0, fe->slot_stride * fe->B));

    std::vector<vslam::PyrLevelTables> pyr_tabs(p.nlevels);
    for (int l = 1; l < p.nlevels; l++) {
        vslam::ResizeTables& r = pyr_tabs[l].r;
        const LevelGeom &s = fe->geom.lv[l - 1], &d = fe->geom.lv[l];
        pyr_tabs[l].sw = s.w; pyr_tabs[l].sh = s.h; pyr_tabs[l].dw = d.w; pyr_tabs[l].dh = d.h;
        vslam::build_resize_tables(s.w, s.h, d.w, d.h, r);
        int rc;
        if ((rc = upload(&fe->d_xtab[l], r.xtab.data(), r.xtab.size() * 2))) return rc;
        if ((rc = upload(&fe->d_xa[l], r.xa.data(), r.xa.size() * 2))) return rc;
        if ((rc = upload(&fe->d_ytab[l], r.ytab.data(), r.ytab.size() * 2))) return rc;
        if ((rc = upload(&fe->d_yb[l], r.yb.data(), r.yb.size() * 2))) return rc;
        std::vector<uint16_t>& qbase = pyr_tabs[l].qbase;
        std::vector<uint32_t>& quads = pyr_tabs[l].quads;
        /* four outputs whose eight taps do not fit one 8-byte source window (scale factors >~ 1.6): that level uses
         * the generic one-pixel-per-thread kernel */
        if (vslam::build_resize_quads(r, s.w, d.w, qbase, quads)) {
            if ((rc = upload(&fe->d_qbase[l], qbase.data(), qbase.size() * 2))) return rc;
            if ((rc = upload(&fe->d_quads[l], quads.data(), quads.size() * 4))) return rc;
        } else {
            qbase.clear(); /* marks "no quad table" for the fused-pyramid planner */
            quads.clear();
        }
    }

    /* fused pyramid plan: levels 1-3 from level 0, then groups of four (8 levels = 2 launches instead of 7).  Needs the
     * quad table on every level; VSLAM_PYRAMID=levels keeps one launch per level (A/B runs). */
    {
        bool fused = fe->tune.pyramid_per_level != 1 && p.nlevels > 1;
        for (int l = 1; l < p.nlevels && fused; l++) fused = fe->d_quads[l] != nullptr;
        static_assert(sizeof(PyrTileDev) == sizeof(vslam::PyrTileLevel), "planner and kernel share the tile record");
        for (int l0 = 0; fused && l0 + 1 < p.nlevels;) {
            const int nl = std::min(l0 == 0 ? 3 : VSLAM_PYR_GROUP_LEVELS, p.nlevels - 1 - l0);
            std::vector<const vslam::PyrLevelTables*> gt;
            for (int j = 1; j <= nl; j++) gt.push_back(&pyr_tabs[l0 + j]);
            vslam::PyrGroupPlan plan;
            /* tile height: tall tiles (48 rows: less halo, fewer prologues) for batches, where the pyramid competes for
             * issue slots; short ones (16) for contexts of one or two images, where a frame's latency counts and more
             * workgroups per image finish sooner (batch-1 latency through the C ABI: 0.135 ms with 12-16 rows, 0.137 with
             * 20-28, 0.142 with 36, 0.146 with 48) */
            /* tile height: 16 rows for one or two images (latency), 48 for batches -- least halo, which is what counts in a
             * VALU-bound pipeline -- and 40 above a megapixel, where the step is the sum of the wide kernels' single-context
             * times and the pyramid alone is 7 % shorter with smaller tiles (profiles/r04_pyramid_tile_sweep.txt: 1080p +1.8 %) */
            const int pyr_rows = fe->tune.pyr_rows >= 0 ? fe->tune.pyr_rows
                                 : fe->B <= 2 ? 16 : (size_t)p.width * p.height > 1000000 ? 40 : 48;
#ifndef VSLAM_PYR_LDS_KB
#define VSLAM_PYR_LDS_KB 64 /* LDS a pyramid tile's cascade may take (32 / 48 / 96 measured) */
#endif
            if (!vslam::build_pyramid_group(gt, l0, VSLAM_PYR_LDS_KB * 1024, plan, pyr_rows)) {
                fused = false;
                break;
            }
            vslam_fe::PyrGroupCtx g;
            memset(&g.dev, 0, sizeof(g.dev));
            g.dev.l0 = l0;
            g.dev.nl = nl;
            g.dev.ntiles = plan.ntx * plan.nty;
            g.dev.readable_w0 = p.width;
            g.lds_bytes = plan.lds_bytes;
            g.d_tiles = nullptr;
            int rc;
            if ((rc = upload(&g.d_tiles, plan.tiles.data(), plan.tiles.size() * sizeof(PyrTileDev)))) return rc;
            g.dev.tiles = g.d_tiles;
            for (int j = 0; j <= nl; j++) g.dev.lg[j] = fe->geom.lv[l0 + j];
            for (int j = 1; j <= nl; j++) {
                g.dev.qbase[j - 1] = fe->d_qbase[l0 + j];
                g.dev.quads[j - 1] = fe->d_quads[l0 + j];
                g.dev.ytab[j - 1] = fe->d_ytab[l0 + j];
                g.dev.yb[j - 1] = fe->d_yb[l0 + j];
            }
            fe->pyr_groups.push_back(g);
            l0 += nl;
        }
        if (!fused) {
            for (auto& g : fe->pyr_groups) hipFree(g.d_tiles);
            fe->pyr_groups.clear();
        }
    }

    /* FAST cells */
    for (int l = 0; l < p.nlevels; l++) {
        fe->level_cell_first[l] = (int)fe->cells.size();
        vslam::build_cells(l, fe->geom.lv[l].w, fe->geom.lv[l].h, fe->cells);
    }
    fe->level_cell_first[p.nlevels] = (int)fe->cells.size();
    if (fe->cells.empty()) { /* (w-32)/30 or (h-32)/30 is 0 on every level: the reference divides by it (fextractor.cpp:772-778) */
        g_err = "image too small: no 30-px FAST cell fits on any pyramid level";
        return VSLAM_ERR_UNSUPPORTED;
    }
    int maxw = 8, maxh = 8;
    size_t cand_total = 0;
    std::vector<CellDesc> dc(fe->cells.size());
    for (size_t i = 0; i < fe->cells.size(); i++) {
        const vslam::HostCell& c = fe->cells[i];
        dc[i].level = c.level; dc[i].x0 = c.x0; dc[i].y0 = c.y0; dc[i].x1 = c.x1; dc[i].y1 = c.y1; dc[i].pad = 0;
        dc[i].base = (uint32_t)cand_total;
        cand_total += (size_t)((c.x1 - c.x0 - 6 + 1) / 2) * (size_t)((c.y1 - c.y0 - 6 + 1) / 2);
        maxw = std::max(maxw, c.x1 - c.x0);
        maxh = std::max(maxh, c.y1 - c.y0);
    }
    fe->tile_pitch = (maxw + 3) & ~3;
    fe->tile_rows = maxh;
    fe->max_px = (maxw - 6) * (maxh - 6);
    if (fe->max_px > 8192) {
        g_err = "FAST cell larger than 8192 px";
        return VSLAM_ERR_UNSUPPORTED;
    }
    {
        int rc;
        if ((rc = upload(&fe->d_cells, dc.data(), dc.size() * sizeof(CellDesc)))) return rc;
    }
    {   /* bands of k_fast_bands: up to four cells of a cell row whose interiors are together at most 128 px wide */
        std::vector<vslam::HostBand> hb;
        const int per = std::min(4, std::max(1, tune_or(fe->tune.fast_band_cells, 4)));
        vslam::build_bands(fe->cells, per, 128, hb);
        int mwh = 0, miw = 0, mc = 0;
        std::vector<BandDesc> db(hb.size());
        /* column tables, one 272-byte record per (cell pitch, interior width) class: cellbit[136] = 1 << (cell of interior
         * column x), 0 past the interior; cellfl[136] = bit 0 first / bit 1 last column of its cell */
        std::vector<std::pair<int, int>> cls;
        std::vector<uint8_t> tab;
        for (size_t i = 0; i < hb.size(); i++) {
            const vslam::HostBand& b = hb[i];
            mwh = std::max(mwh, (int)b.wh);
            miw = std::max(miw, (int)b.ww - 6);
            mc = std::max(mc, (int)b.ncell);
            const int iw = (int)b.ww - 6, wc = std::max(1, (int)b.wcell);
            size_t ci = 0;
            while (ci < cls.size() && cls[ci] != std::make_pair(wc, iw)) ci++;
            if (ci == cls.size()) {
                cls.push_back(std::make_pair(wc, iw));
                tab.resize(tab.size() + 272, 0);
                uint8_t* cb = tab.data() + ci * 272;
                for (int x = 0; x < 136 && x < iw; x++) {
                    const int c = x / wc, cend = std::min((c + 1) * wc, iw) - 1;
                    cb[x] = (uint8_t)(1u << std::min(c, 7));
                    cb[136 + x] = (uint8_t)((x == c * wc ? 1 : 0) | (x == cend ? 2 : 0));
                }
            }
            db[i].cell0 = b.cell0;
            db[i].lnw = (uint32_t)b.level | ((uint32_t)b.ncell << 4) | ((uint32_t)b.wcell << 8) | ((uint32_t)ci << 16);
            db[i].xy = (uint32_t)b.x0 | ((uint32_t)b.y0 << 16);
            db[i].wh = (uint32_t)b.ww | ((uint32_t)b.wh << 16);
        }
        bool ok = !hb.empty() && vk_fast_bands_check(mwh, miw, mc) == 0 && p.nlevels <= 16 && cls.size() < 65536;
        for (const vslam::HostBand& b : hb) ok = ok && b.wcell < 256;
        if (ok) {
            int rc;
            if ((rc = upload(&fe->d_band_classes, tab.data(), tab.size()))) return rc;
            if ((rc = upload(&fe->d_bands, db.data(), db.size() * sizeof(BandDesc)))) return rc;
            fe->nbands = (int)db.size();
            fe->band_max_wh = mwh;
            fe->band_max_iw = miw;
        }
    }
    const int ncells = (int)fe->cells.size();
    /* every cell owns a fixed segment sized by the exact upper bound of its NMS survivors */
    fe->cand_cap = (int)std::max<size_t>(cand_total, 16);
    fe->cand_stride = (8 + (size_t)ncells * sizeof(CellOut) + (size_t)fe->cand_cap * 4 + 255) & ~(size_t)255;
    HIPCHK(hipMalloc((void**)&fe->d_cand, fe->cand_stride * fe->B));
    HIPCHK((hipError_t)vslam_pinned_alloc((void**)&fe->h_cand, fe->cand_stride * fe->B));

    {
        int rc;
        if ((rc = upload(&fe->d_pattern, VSLAM_ORB_PATTERN, 1024))) return rc;
    }
    vk_upload_disc(fe->tab.umax);
    {
        /* marching-rows blur: one wave task per (level, row chunk, 248-column strip) */
        if (fe->tune.blur_rows >= 0) fe->blur_rows = std::min(512, std::max(8, (int)fe->tune.blur_rows));
        else if (fe->B <= 2) fe->blur_rows = 8; /* one or two images: more, shorter tasks finish sooner (batch-1 latency -5 us) */
        std::vector<uint32_t> tasks;
        for (int l = 0; l < p.nlevels; l++) {
            const int br = fe->blur_rows;
            const int ns = std::max(1, (fe->geom.lv[l].w + 247) / 248), nc = (fe->geom.lv[l].h + br - 1) / br;
            for (int c = 0; c < nc; c++)
                for (int s = 0; s < ns; s++) tasks.push_back(((uint32_t)l << 24) | ((uint32_t)c << 12) | (uint32_t)s);
        }
        fe->n_blur_tasks = (int)tasks.size();
        int rc;
        if ((rc = upload(&fe->d_blur_tasks, tasks.data(), tasks.size() * 4))) return rc;
        /* limits of k_fast_cells_v3 (LDS pitch 72, 2 keep words per interior row) and
         * k_blur7_v2 (8-byte row windows); cell windows are at most 59 + 6 px on a side and levels at least 40 px wide,
         * so these never trigger */
        if (maxw > 66 || maxh > 134 || fe->geom.lv[p.nlevels - 1].w < 8) {
            g_err = "FAST cell window larger than 66 x 134 px";
            return VSLAM_ERR_UNSUPPORTED;
        }
    }

    const size_t nk = (size_t)fe->B * fe->cap;
    HIPCHK(hipMalloc((void**)&fe->d_sel, nk * sizeof(SelKp)));
    HIPCHK((hipError_t)vslam_pinned_alloc((void**)&fe->h_sel, nk * sizeof(SelKp)));
    /* ONE result block per context -- counts | keypoints | descriptors, back to back -- and one pinned mirror of it: a
     * full batch leaves the device in a single transfer (vslam_fe.hip: enqueue_extract_plain) */
    fe->res_counts_bytes = (((size_t)(fe->B * 4 + 4) * 4) + 255) & ~(size_t)255;
    fe->res_feat_bytes = fe->res_counts_bytes + nk * sizeof(vslam_kp) + nk * 32; /* multiple of 16: cap % 4 == 0 */
    /* ... followed by the outputs of the device SearchForInitialization for up to B pairs (vnMatches12 | vbPrevMatched |
     * nmatches), so that extraction and matcher results of a step can leave in ONE transfer (want_host = 2) */
    fe->res_init_bytes = (nk * 12 + (size_t)fe->B * 16 + 255) & ~(size_t)255; /* = init_scratch's need for B pairs */
    fe->res_bytes = fe->res_feat_bytes + fe->res_init_bytes;
    HIPCHK(hipMalloc((void**)&fe->d_res, fe->res_bytes));
    HIPCHK((hipError_t)vslam_pinned_alloc((void**)&fe->h_res, fe->res_bytes));
    HIPCHK(hipMemset(fe->d_res, 0, fe->res_bytes));
    memset(fe->h_res, 0, fe->res_bytes);
    fe->d_counts = (int32_t*)fe->d_res;
    fe->h_counts = (int32_t*)fe->h_res;
    fe->d_kps = (vslam_kp*)(fe->d_res + fe->res_counts_bytes);
    fe->h_kps = (vslam_kp*)(fe->h_res + fe->res_counts_bytes);
    fe->d_desc = fe->d_res + fe->res_counts_bytes + nk * sizeof(vslam_kp);
    fe->h_desc = fe->h_res + fe->res_counts_bytes + nk * sizeof(vslam_kp);
    fe->d_init = fe->d_res + fe->res_feat_bytes; /* grows into an allocation of its own for more than B pairs (vslam_match.hip) */
    fe->h_init = fe->h_res + fe->res_feat_bytes;
    fe->init_bytes = fe->h_init_bytes = fe->res_init_bytes;
    fe->init_in_block = true;

    /* GPU quadtree distribution (k_octree): per-level parameters, key ping-pong arrays, result lists */
    {
        OctParams& O = fe->oct;
        memset(&O, 0, sizeof(O));
        int maxNodes = 16, selOff = 0;
        bool ok = !(p.flags & VSLAM_FLAG_HOST_OCTREE);
        for (int l = 0; l < p.nlevels; l++) {
            const int W = fe->geom.lv[l].w - 2 * VSLAM_FAST_BORDER, H = fe->geom.lv[l].h - 2 * VSLAM_FAST_BORDER;
            const int nIni = (int)std::round((float)W / (float)H);
            O.N[l] = fe->tab.quota[l];
            O.H[l] = H;
            O.nIni[l] = nIni;
            O.hX[l] = (float)W / nIni;
            O.cellFirst[l] = fe->level_cell_first[l];
            O.selOff[l] = selOff;
            const int cap_l = std::max(O.N[l] + 3, 4 * nIni) + 1;
            selOff += cap_l;
            maxNodes = std::max(maxNodes, cap_l);
            /* k_octree_v2 parks a level's cell offsets (one u32 per cell) in the second node array (4 u32 per node) */
            maxNodes = std::max(maxNodes, (fe->level_cell_first[l + 1] - fe->level_cell_first[l] + 1 + 3) / 4); /* + sentinel */
            if (nIni > 64) ok = false;
        }
        O.cellFirst[p.nlevels] = fe->level_cell_first[p.nlevels];
        O.selStride = selOff;
        O.maxNodes = (maxNodes + 15) & ~15;
        std::vector<uint32_t> lut;
        std::vector<OctPart> parts((size_t)VSLAM_MAX_LEVELS * VSLAM_OCT_MAX_PARTS);
        int maxPartCells = 1, cntWords = 0, maxcells_all = 0;
        {
            /* k_octree_v4's fine grid: one level deeper than the depth at which a full quadtree has N nodes (nIni * 4^d
             * at depth d), so that the split passes, which stop at N nodes, mostly stay above it; keys that cluster
             * below it are handled exactly by the kernel's in-cell path, so the depth only decides speed.  Both arrays
             * of the largest level must fit LDS next to the node arrays (budget below), at most 16384 cells (the cell
             * index travels in 16 bits), and no cell may be able to hold 65536 keys (the rank does too).
             * vslam_tuning.oct_fine_depth forces a depth where it is admissible (tests: deep splits everywhere). */
            const int fd = fe->tune.oct_fine_depth; /* -1: by the level's quota */
            const size_t nb = (vk_octree_lds_bytes(O.maxNodes) + 15) & ~(size_t)15;
            /* LDS a quadtree workgroup may take in all (default 128 KB of the CU's 160) */
            const size_t budget = (size_t)std::min(150, std::max(16, tune_or(fe->tune.oct_lds_budget_kb, 128))) * 1024;
            int maxcells = 0;
            for (int l = 0; l < p.nlevels; l++) {
                const int W = fe->geom.lv[l].w - 2 * VSLAM_FAST_BORDER, H = O.H[l];
                int D = 1;
                while ((O.nIni[l] << (2 * D)) < O.N[l]) D++;
                D += 1;
                if (fd >= 0) D = fd;
                D = std::max(1, std::min(D, 11));
                auto cell_keys = [&](int d) { /* strict 3x3 maxima a cell of depth d can hold: every other pixel of every other row */
                    const long long cw = (W / O.nIni[l] >> d) + 2, ch = (H >> d) + 2;
                    return ((cw + 1) / 2) * ((ch + 1) / 2);
                };
                while (D < 11 && cell_keys(D) >= 65535) D++;
                while (D > 1 && (((long long)O.nIni[l] << (2 * D)) > 16384 ||
                                 nb + 2 * (((size_t)O.nIni[l] << (2 * D)) + 1) * 4 + 16 > budget) && cell_keys(D - 1) < 65535)
                    D--;
                O.fineD[l] = D;
                maxcells = std::max(maxcells, O.nIni[l] << (2 * D));
                std::vector<uint32_t> xs, ys;
                vslam::build_oct_lut(W, H, D, xs, ys);
                O.lutOff[l] = (int32_t)lut.size();
                O.lutW[l] = (int32_t)xs.size();
                lut.insert(lut.end(), xs.begin(), xs.end());
                lut.insert(lut.end(), ys.begin(), ys.end());
                /* k_oct_count's parts of the level: rows of leaves = the first kb y decisions of the leaf index; as many
                 * as leave a part at least ~48 FAST cells (up to eight) */
                const int cf = fe->level_cell_first[l], cl = fe->level_cell_first[l + 1], ncl = cl - cf;
                int kb = 0;
                while (kb < 3 && kb < D && (ncl >> (kb + 1)) >= 48) kb++;
                O.partBits[l] = kb;
                O.fineCntOff[l] = cntWords;
                cntWords += ((O.nIni[l] << (2 * D)) + 3) & ~3;
                auto part_of = [&](uint32_t code) {
                    int pb = 0;
                    for (int t = 0; t < kb; t++) pb = (pb << 1) | (int)((code >> (2 * D - 1 - 2 * t)) & 1u);
                    return pb;
                };
                for (int pi = 0; pi < (1 << kb); pi++) {
                    OctPart& op = parts[(size_t)l * VSLAM_OCT_MAX_PARTS + pi];
                    op.ylo = op.yhi = 0;
                    bool any = false;
                    for (int y = 0; y <= H; y++)
                        if (part_of(ys[y]) == pi) {
                            if (!any) op.ylo = y;
                            op.yhi = y + 1;
                            any = true;
                        }
                    /* the FAST cells whose interior rows (border-relative: y0 + 3 - 16 .. y1 - 3 - 16) meet [ylo, yhi);
                     * cells are listed cell row by cell row, so they are one contiguous range */
                    op.ca = op.cb = cf;
                    bool first = true;
                    for (int c = cf; c < cl && any; c++) {
                        const int cy0 = (int)fe->cells[c].y0 + 3 - VSLAM_FAST_BORDER, cy1 = (int)fe->cells[c].y1 - 3 - VSLAM_FAST_BORDER;
                        if (cy1 > op.ylo && cy0 < op.yhi) {
                            if (first) op.ca = c;
                            op.cb = c + 1;
                            first = false;
                        }
                    }
                    maxPartCells = std::max(maxPartCells, op.cb - op.ca);
                }
                maxcells_all = std::max(maxcells_all, O.nIni[l] << (2 * D));
            }
            O.fineLdsOff = (int32_t)nb;
            O.fineLdsBytes = (int32_t)(2 * ((size_t)maxcells + 1) * 4 + 16);
            if (nb + (size_t)O.fineLdsBytes > 160 * 1024) ok = false; /* cannot happen with the limits above */
        }
        O.ptsCap = fe->cand_cap;
        O.dbg = nullptr;
        if (fe->tune.oct_debug == 1) {
            HIPCHK(hipMalloc(&O.dbg, 64 * 8));
            HIPCHK(hipMemset(O.dbg, 0, 64 * 8));
        }
        O.maxIter = tune_or(fe->tune.oct_max_iter, 64);
        if (vk_octree_lds_bytes(O.maxNodes) > 150 * 1024) ok = false; /* list does not fit LDS: host quadtree */
        fe->dev_octree = ok;
        if (ok) {
            if (vk_octree_set_max_lds((size_t)O.fineLdsOff + (size_t)O.fineLdsBytes + 16) != 0) {
                g_err = "hipFuncSetAttribute(k_octree, max dynamic LDS) failed";
                return VSLAM_ERR_HIP;
            }
            const size_t np = (size_t)fe->B * fe->cand_cap;
            HIPCHK(hipMalloc((void**)&fe->d_pts[0], np * 4));
            HIPCHK(hipMalloc((void**)&fe->d_pts[1], np * 4));
            HIPCHK(hipMalloc((void**)&fe->d_sel_xyr, (size_t)fe->B * O.selStride * 4));
            HIPCHK(hipMalloc((void**)&fe->d_sel_cnt, (size_t)fe->B * VSLAM_MAX_LEVELS * 4));
            if (fe->tune.octree_walk_kernel != 1) { /* 1: the walk-per-pass kernel only (A/B runs) */
                HIPCHK(hipMalloc((void**)&fe->d_oct_sorted, np * 8));
                HIPCHK(hipMalloc((void**)&fe->d_oct_redo, (size_t)fe->B * VSLAM_MAX_LEVELS * 4));
                HIPCHK(hipMemset(fe->d_oct_redo, 0, (size_t)fe->B * VSLAM_MAX_LEVELS * 4));
                int rc;
                if ((rc = upload(&fe->d_oct_lut, lut.data(), lut.size() * 4))) return rc;
                O.lut = fe->d_oct_lut;
                /* vslam_tuning.oct_precount = 1: walk 1 as k_oct_count, a level's keys over up to eight workgroups.  Built and
                 * measured in round 4 (profiles/r04_oct_precount_ab.txt), bit-exact, and NOT the default: the stage is 4 % shorter
                 * for 32 KITTI frames (78 vs 81 us) but 9 % longer at 1080p (253 vs 232 us) and 11 % longer for one or two
                 * frames (52 vs 47 us; 0.129 vs 0.126 ms per frame end to end): a part's walk is a chain of dependent round
                 * trips per batch of keys, not a matter of keys per CU -- eight parts of 256 threads still make 21 of them where
                 * the 512-thread workgroup made 48 -- and what they save is spent on a launch boundary, on every part summing
                 * the level's cell counts again, and on walk 2 reading keys another CU wrote.  Its three prefix sums travel in
                 * 21-bit fields: candidate buffers of 2^21 entries and more keep the walk inside in any case */
                fe->oct_maxcells = maxcells_all;
                if (tune_or(fe->tune.oct_precount, 0) == 1 && fe->cand_cap < (1 << 21)) {
                    if ((rc = upload(&fe->d_oct_parts, parts.data(), parts.size() * sizeof(OctPart)))) return rc;
                    HIPCHK(hipMalloc((void**)&fe->d_oct_cnt, (size_t)fe->B * cntWords * 4));
                    HIPCHK(hipMemset(fe->d_oct_cnt, 0, (size_t)fe->B * cntWords * 4));
                    if (vk_oct_count_set_max_lds(vk_oct_count_lds(maxcells_all, maxPartCells)) != 0) {
                        g_err = "hipFuncSetAttribute(k_oct_count, max dynamic LDS) failed";
                        return VSLAM_ERR_HIP;
                    }
                    O.parts = fe->d_oct_parts;
                    O.fineCnt = fe->d_oct_cnt;
                    O.fineCntStride = cntWords;
                    O.maxPartCells = maxPartCells;
                }
            } else {
                HIPCHK(hipMalloc((void**)&fe->d_nid, np * 2));
            }
        }
    }

    {
        /* The runtime multiplexes HIP streams onto a fixed number of hardware queues (GPU_MAX_HW_QUEUES), one pool per
         * stream priority, least-used queue first.  In a process that already holds dozens of streams (torch, four RCCL
         * communicators) two contexts ended up on ONE hardware queue and their passes ran one behind the other (mono
         * 157 k -> 97 k frames/s, the kernel trace shows the shared queue id).  Streams of a priority of their own draw
         * from a pool nobody else uses: with stream_priority 2 (1 = low) the context's stream is created with a priority of
         * its own.  The library default is 0 -- the default pool -- because a high-priority stream also pre-empts the
         * host application's own default-priority work; a pipelined caller opts in (bench.py does).  Measured: collective path at world size 1 97 k -> 143 k (high) / 141 k (low),
         * the plain path unchanged at 157 k. */
        const int pr = tune_or(fe->tune.stream_priority, 0);
        int lo = 0, hi = 0;
        if (pr != 0 && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi)
            HIPCHK(hipStreamCreateWithPriority(&fe->stream, hipStreamNonBlocking, pr == 1 ? lo : hi));
        else
            HIPCHK(hipStreamCreateWithFlags(&fe->stream, hipStreamNonBlocking));
    }
    HIPCHK(hipEventCreateWithFlags(&fe->ev_cand, hipEventDisableTiming));
    fe->use_graph = fe->tune.graphs != 0; /* 0: never replay captured graphs */
    /* threads per quadtree problem (vk_octree): 1024 for one or two images, else 256, 512 for frames above a megapixel */
    fe->oct_threads = fe->tune.oct_threads == 256 || fe->tune.oct_threads == 512 || fe->tune.oct_threads == 1024
                          ? fe->tune.oct_threads
                          : fe->B <= 2 ? 1024 : (size_t)p.width * p.height > 1000000 ? 512 : 256;
    fe->sel_level.resize((size_t)fe->B * p.nlevels);
    fe->cand_level.resize((size_t)fe->B * p.nlevels);
    unsigned hw = std::thread::hardware_concurrency();
    int nthreads = (int)std::min<unsigned>(hw ? hw : 4, 16) - 1;
    nthreads = std::min(nthreads, fe->B * p.nlevels - 1);
    fe->pool = new WorkerPool(std::max(nthreads, 0));
    HIPCHK(hipDeviceSynchronize());
    return VSLAM_OK;
}

extern "C" int vslam_fe_create(const vslam_fe_params* p, vslam_fe** out) {
    if (!p || !out) return VSLAM_ERR_INVALID;
    *out = nullptr;
    if (p->width <= 0 || p->height <= 0 || p->nfeatures <= 0 || p->nlevels < 1 ||
        p->nlevels > VSLAM_MAX_LEVELS || p->scale_factor <= 1.0f || p->max_batch < 1 ||
        p->max_batch > VSLAM_MAX_BATCH || p->min_th_fast < 1 || p->ini_th_fast < p->min_th_fast ||
        p->ini_th_fast > 254 || p->nfeatures > 60000) {
        g_err = "invalid parameters";
        return VSLAM_ERR_INVALID;
    }
    vslam_fe* fe = new vslam_fe();
    fe->p = *p;
    fe->tune = vslam_resolve_tuning(p->tuning);
    fe->p.tuning = nullptr; /* the caller's struct need not outlive the call */
    int rc = create_impl(p, fe);
    if (rc != VSLAM_OK) {
        std::string keep = g_err;
        free_ctx(fe);
        g_err = keep;
        return rc;
    }
    *out = fe;
    return VSLAM_OK;
}

extern "C" int vslam_fe_set_tuning(vslam_fe* fe, const vslam_tuning* t) {
    if (!fe || !t) return VSLAM_ERR_INVALID;
    vslam_apply_tuning(fe->tune, t);
    fe->use_graph = fe->use_graph && fe->tune.graphs != 0;
    /* a captured graph froze the launch shapes and transport routes it was captured with, and its key does not cover the
     * tuning fields: drop it, so that the next host-image pass captures again under the new switches (otherwise an A/B
     * run through this call would compare a configuration with itself) */
    if (fe->graph_exec) {
        if (fe->stream) hipStreamSynchronize(fe->stream);
        hipGraphExecDestroy(fe->graph_exec);
        fe->graph_exec = nullptr;
        fe->graph_key = 0;
    }
    return VSLAM_OK;
}

extern "C" int vslam_fe_tables(const vslam_fe* fe, float* scale, float* inv_scale, float* sigma2,
                               float* inv_sigma2, int32_t* quota) {
    if (!fe) return VSLAM_ERR_INVALID;
    for (int i = 0; i < fe->p.nlevels; i++) {
        if (scale) scale[i] = fe->tab.scale[i];
        if (inv_scale) inv_scale[i] = fe->tab.inv_scale[i];
        if (sigma2) sigma2[i] = fe->tab.sigma2[i];
        if (inv_sigma2) inv_sigma2[i] = fe->tab.inv_sigma2[i];
        if (quota) quota[i] = fe->tab.quota[i];
    }
    return fe->p.nlevels;
}

extern "C" int vslam_fe_octree_stats(const vslam_fe* fe, unsigned long long* problems, unsigned long long* split_below_grid,
                                     uint32_t* last_level_masks) {
    if (!fe) return VSLAM_ERR_INVALID;
    if (problems) *problems = fe->oct_problems;
    if (split_below_grid) *split_below_grid = fe->oct_deep;
    if (last_level_masks)
        for (int s = 0; s < fe->B; s++) last_level_masks[s] = s < fe->last_nimg ? fe->oct_last_mask[s] : 0u;
    return VSLAM_OK;
}

extern "C" int vslam_fe_delivery_stats(const vslam_fe* fe, unsigned long long* transfers, unsigned long long* bytes) {
    if (!fe) return VSLAM_ERR_INVALID;
    if (transfers) *transfers = fe->n_deliveries;
    if (bytes) *bytes = fe->n_delivery_bytes;
    return VSLAM_OK;
}

extern "C" int vslam_dbg_octree_stamps(vslam_fe* fe, unsigned long long* out64) {
    if (!fe || !fe->oct.dbg) return VSLAM_ERR_INVALID;
    HIPCHK(hipMemcpy(out64, fe->oct.dbg, 64 * 8, hipMemcpyDeviceToHost));
    return VSLAM_OK;
}

extern "C" void* vslam_fe_stream(vslam_fe* fe) { return fe ? (void*)fe->stream : nullptr; }

/* GPU-side ordering between two contexts: everything enqueued on `waiter` after this call runs after
 * everything enqueued on `signal` so far (hipEventRecord + hipStreamWaitEvent; no host synchronisation). */
extern "C" int vslam_fe_wait_for(vslam_fe* waiter, vslam_fe* signal) {
    if (!waiter || !signal || waiter->p.device != signal->p.device) return VSLAM_ERR_INVALID;
    if (waiter == signal) return VSLAM_OK;
    HIPCHK(hipSetDevice(waiter->p.device));
    if (!signal->ev_x) HIPCHK(hipEventCreateWithFlags(&signal->ev_x, hipEventDisableTiming));
    HIPCHK(hipEventRecord(signal->ev_x, signal->stream));
    HIPCHK(hipStreamWaitEvent(waiter->stream, signal->ev_x, 0));
    return VSLAM_OK;
}

/* Finer-grained form: four user events per context.  _record marks a point of fe's stream, _wait makes
 * waiter's stream wait for the last recorded instance of signal's event idx (a never-recorded event does
 * not block). */
extern "C" int vslam_fe_event_record(vslam_fe* fe, int idx) {
    if (!fe || idx < 0 || idx >= 4) return VSLAM_ERR_INVALID;
    HIPCHK(hipSetDevice(fe->p.device));
    if (!fe->ev_user[idx]) HIPCHK(hipEventCreateWithFlags(&fe->ev_user[idx], hipEventDisableTiming));
    HIPCHK(hipEventRecord(fe->ev_user[idx], fe->stream));
    return VSLAM_OK;
}

extern "C" int vslam_fe_event_wait(vslam_fe* waiter, vslam_fe* signal, int idx) {
    if (!waiter || !signal || idx < 0 || idx >= 4 || waiter->p.device != signal->p.device) return VSLAM_ERR_INVALID;
    if (!signal->ev_user[idx] || waiter == signal) return VSLAM_OK;
    HIPCHK(hipSetDevice(waiter->p.device));
    HIPCHK(hipStreamWaitEvent(waiter->stream, signal->ev_user[idx], 0));
    return VSLAM_OK;
}

/* Pacing of the one kernel that takes a CU's whole LDS: with a gate set, the FAST launch of every pass of `fe` waits (GPU side)
 * for the FAST launch of `signal`'s latest pass to have finished, so that a pipelined caller who keeps several contexts in
 * flight never has two FAST launches resident at once (bench.py --fast-chain).  signal == NULL removes the gate.  Passes
 * with a gate are not captured into graphs (an event of another stream cannot be waited for inside a capture). */
extern "C" int vslam_fe_set_fast_gate(vslam_fe* fe, vslam_fe* signal) {
    if (!fe || (signal && (signal == fe || signal->p.device != fe->p.device))) return VSLAM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(g_gate_mu);
    gate_unlink(fe);
    if (signal) {
        fe->fast_gate = signal;
        signal->gate_waiters.push_back(fe);
        HIPCHK(hipSetDevice(signal->p.device));
        if (!signal->ev_fast) HIPCHK(hipEventCreateWithFlags(&signal->ev_fast, hipEventDisableTiming));
        signal->fast_gated_by_someone = true;
        fe->use_graph = false;
        signal->use_graph = false;
    }
    return VSLAM_OK;
}

extern "C" int vslam_fe_set_profiling(vslam_fe* fe, int on) {
    if (!fe) return VSLAM_ERR_INVALID;
    HIPCHK(hipSetDevice(fe->p.device));
    if (on && !fe->ev_prof[0])
        for (int i = 0; i < 10; i++) HIPCHK(hipEventCreate(&fe->ev_prof[i]));
    fe->profiling = on != 0;
    for (int i = 0; i < 5; i++) fe->prof_ms[i] = 0;
    fe->prof_batches = fe->prof_images = 0;
    return VSLAM_OK;
}

extern "C" int vslam_fe_get_profile(vslam_fe* fe, double stage_ms[5], long* batches, long* images) {
    if (!fe || !stage_ms) return VSLAM_ERR_INVALID;
    for (int i = 0; i < 5; i++) stage_ms[i] = fe->prof_ms[i];
    if (batches) *batches = fe->prof_batches;
    if (images) *images = fe->prof_images;
    return VSLAM_OK;
}

static int pack_range(vslam_fe* fe, int first, int nslots, void* dev_dst, size_t slot_bytes, bool sync);

extern "C" int vslam_fe_pack_slot_range(vslam_fe* fe, int first, int nslots, void* dev_dst, size_t slot_bytes) {
    return pack_range(fe, first, nslots, dev_dst, slot_bytes, true);
}

/* same, but only enqueued on the context's stream (ordered after the extraction that produced the slots) */
extern "C" int vslam_fe_pack_slot_range_async(vslam_fe* fe, int first, int nslots, void* dev_dst,
                                              size_t slot_bytes) {
    return pack_range(fe, first, nslots, dev_dst, slot_bytes, false);
}

static int pack_range(vslam_fe* fe, int first, int nslots, void* dev_dst, size_t slot_bytes, bool sync) {
    if (!fe || first < 0 || nslots < 0 || first + nslots > fe->B || !dev_dst ||
        slot_bytes < 16 + (size_t)fe->cap * 60 || (slot_bytes & 15) || ((uintptr_t)dev_dst & 15)) {
        g_err = "invalid arguments (slot_bytes and dev_dst must be 16-byte aligned)";
        return VSLAM_ERR_INVALID;
    }
    HIPCHK(hipSetDevice(fe->p.device));
    /* one kernel; the keypoint counts are read from HBM, so this may be enqueued before the host knows them */
    vk_pack_slots(fe->stream, fe->d_kps, fe->d_desc, fe->d_counts, fe->cap, first, nslots, (uint8_t*)dev_dst,
                  slot_bytes, wave_prio_on(fe->tune, 8));
    HIPCHK(hipGetLastError());
    if (sync) HIPCHK(vslam_stream_wait(fe->stream));
    return VSLAM_OK;
}

extern "C" int vslam_fe_pack_slots(vslam_fe* fe, int nslots, void* dev_dst, size_t slot_bytes) {
    return vslam_fe_pack_slot_range(fe, 0, nslots, dev_dst, slot_bytes);
}

extern "C" int vslam_fe_level_size(const vslam_fe* fe, int level, int* w, int* h) {
    if (!fe || level < 0 || level >= fe->p.nlevels) return VSLAM_ERR_INVALID;
    if (w) *w = fe->geom.lv[level].w;
    if (h) *h = fe->geom.lv[level].h;
    return VSLAM_OK;
}

extern "C" int vslam_fe_level_copy(vslam_fe* fe, int slot, int level, int blurred, uint8_t* dst,
                                   size_t dst_pitch) {
    if (!fe || slot < 0 || slot >= fe->B || level < 0 || level >= fe->p.nlevels || !dst) return VSLAM_ERR_INVALID;
    const LevelGeom& g = fe->geom.lv[level];
    if (dst_pitch < (size_t)g.w) return VSLAM_ERR_INVALID;
    HIPCHK(hipSetDevice(fe->p.device));
    const uint8_t* s;
    size_t spitch;
    if (!blurred && level == 0) {
        s = fe->src.l0[slot];
        spitch = fe->src.pitch0[slot];
        if (!s) return VSLAM_ERR_INVALID;
    } else {
        s = (blurred ? fe->d_blur : fe->d_pyr) + (size_t)slot * fe->slot_stride + g.off;
        spitch = g.pitch;
    }
    HIPCHK(hipMemcpy2DAsync(dst, dst_pitch, s, spitch, g.w, g.h, hipMemcpyDeviceToHost, fe->stream));
    HIPCHK(vslam_stream_wait(fe->stream));
    return VSLAM_OK;
}

/* ------------------------------------------------------------------ extraction */
static void decode_candidates(vslam_fe* fe, int s, int l) {
    const int ncells = (int)fe->cells.size();
    const size_t hdr_bytes = 8 + (size_t)ncells * sizeof(CellOut);
    const uint8_t* base = fe->h_cand + (size_t)s * fe->cand_stride;
    const CellOut* co = (const CellOut*)(base + 8);
    const uint32_t* cand = (const uint32_t*)(base + hdr_bytes);
    std::vector<vslam::Cand>& cl = fe->cand_level[(size_t)s * fe->p.nlevels + l];
    cl.clear();
    for (int c = fe->level_cell_first[l]; c < fe->level_cell_first[l + 1]; c++) {
        const uint32_t* q = cand + co[c].base;
        for (uint32_t k = 0; k < co[c].count; k++) {
            vslam::Cand cd;
            cd.x = (int16_t)(q[k] & 0xFFF);
            cd.y = (int16_t)((q[k] >> 12) & 0xFFF);
            cd.response = (uint8_t)(q[k] >> 24);
            cl.push_back(cd);
        }
    }
}

/* steps shared by both quadtree placements: level 0, pyramid, FAST */
/* host images: rows into pinned staging (a copy kernel pulls them into HBM afterwards).  hipMemcpy2DAsync from
 * pageable memory took 2.8 ms per KITTI frame on this stack -- 90 % of a single-frame call. */
/* memcpy with streaming stores (dst 16-byte aligned): the staged image is read next by the GPU over PCIe, not by this
 * core -- lines left dirty in the core's cache have to be snooped out for every read the device makes */
static void copy_streaming(uint8_t* dst, const uint8_t* src, size_t n) {
#if defined(__SSE2__)
    size_t i = 0;
    for (; i + 64 <= n; i += 64) {
        const __m128i a = _mm_loadu_si128((const __m128i*)(src + i)), b = _mm_loadu_si128((const __m128i*)(src + i + 16));
        const __m128i c = _mm_loadu_si128((const __m128i*)(src + i + 32)), d = _mm_loadu_si128((const __m128i*)(src + i + 48));
        _mm_stream_si128((__m128i*)(dst + i), a);
        _mm_stream_si128((__m128i*)(dst + i + 16), b);
        _mm_stream_si128((__m128i*)(dst + i + 32), c);
        _mm_stream_si128((__m128i*)(dst + i + 48), d);
    }
    if (i < n) memcpy(dst + i, src + i, n - i);
    _mm_sfence();
#else
    memcpy(dst, src, n);
#endif
}

static int stage_host_images(vslam_fe* fe, int nimg, const uint8_t* const* imgs, size_t pitch) {
    const vslam_fe_params& p = fe->p;
    const size_t lp = fe->geom.lv[0].pitch, img_bytes = lp * (size_t)p.height;
    if (!fe->h_img) HIPCHK((hipError_t)vslam_pinned_alloc((void**)&fe->h_img, img_bytes * fe->B));
    for (int s = 0; s < nimg; s++)
        if (!imgs[s]) {
            g_err = "null image";
            return VSLAM_ERR_INVALID;
        }
    /* one task per image on the context's worker pool: a single thread copies ~18 GB/s, which bounded the
     * host-image batch path at 0.85 ms per 32 KITTI frames */
    /* dense rows (pitch == width, what cv::Mat::isContinuous() images are) stay dense in the staging: ONE memcpy per
     * image instead of one per row (a 1241 x 376 frame: 12 instead of 27 us) and 3 % fewer bytes over the link; the
     * pull kernel and the DMA route take any source pitch */
    fe->h_img_pitch = pitch == (size_t)p.width ? (size_t)p.width : lp;
    fe->pool->parallel_for(nimg, [&](int s) {
        uint8_t* hs = fe->h_img + img_bytes * s;
        if (pitch == (size_t)p.width) copy_streaming(hs, imgs[s], (size_t)p.width * p.height);
        else if (pitch == lp) memcpy(hs, imgs[s], img_bytes - (lp - p.width));
        else
            for (int y = 0; y < p.height; y++) memcpy(hs + (size_t)y * lp, imgs[s] + (size_t)y * pitch, p.width);
    });
    return VSLAM_OK;
}

/* VSLAM_H2D = sdma | pull forces one transport; default: the DMA engines for batches (they run beside the other
 * contexts' kernels without disturbing them), the pull kernel for one or two images (a synchronous single-frame call
 * has nothing to overlap with, and the hand-over between the DMA engine and the compute queue costs ~30 us) */
static bool h2d_uses_sdma(const vslam_fe* fe, int nimg) {
    const int mode = fe->tune.h2d_route; /* 1 pull, 2 sdma, else by batch size */
    return mode == 2 || (mode != 1 && nimg > 2);
}

/* the device staging buffer of the sdma transport, allocated OUTSIDE stream capture (hipMalloc is not capturable) */
static int ensure_stage(vslam_fe* fe, size_t spitch, int nimg) {
    if (!h2d_uses_sdma(fe, nimg)) return VSLAM_OK;
    const size_t one = (size_t)(fe->p.height - 1) * spitch + fe->p.width;
    const size_t stride = (one + 255) & ~(size_t)255;
    if (fe->d_stage_bytes >= stride * fe->B + 256) return VSLAM_OK;
    HIPCHK(vslam_stream_wait(fe->stream)); /* an earlier pass may still read the old buffer */
    return vslam_ensure((void**)&fe->d_stage, &fe->d_stage_bytes, stride * fe->B + 256);
}

/* Host rows -> level 0 of slots 0..nimg-1, enqueued on fe's stream.  `where` = VSLAM_IMGS_PINNED (the caller's pinned
 * images) or VSLAM_IMGS_HOST (rows already copied into the context's pinned staging by stage_host_images).
 * Two transports (VSLAM_H2D = sdma | pull, default sdma):
 *   sdma: hipMemcpyAsync (the DMA engines: no CU, no L2 miss-queue entries held for microseconds) copies every image
 *         as one linear block into a device staging buffer -- ONE call when the images are equally spaced in memory, as
 *         the buffers of a capture ring are -- and a kernel re-pitches from HBM into the 128-byte-pitched level 0.
 *         Measured beside the other contexts' kernels: 102 k frames/s against 64-72 k with the pull kernel, whose host
 *         reads (2-3 us each) sit in the L2's queues in front of everybody's HBM requests (describe 122 -> 274 us).
 *   pull: one kernel reads the host rows over PCIe itself (55 GB/s alone on the GPU; kept for A/B runs). */
static bool hip_stream_capturing(hipStream_t s) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
}

static int upload_host_rows(vslam_fe* fe, int nimg, const uint8_t* const* imgs, size_t pitch, int where) {
    const vslam_fe_params& p = fe->p;
    hipStream_t st = fe->stream;
    const size_t lp = fe->geom.lv[0].pitch, img_bytes = lp * (size_t)p.height;
    const bool use_sdma = h2d_uses_sdma(fe, nimg);
    BatchSrc hs;
    for (int s = 0; s < nimg; s++) {
        if (where == VSLAM_IMGS_PINNED && !imgs[s]) {
            g_err = "null image";
            return VSLAM_ERR_INVALID;
        }
        hs.l0[s] = where == VSLAM_IMGS_PINNED ? imgs[s] : fe->h_img + img_bytes * s;
        hs.pitch0[s] = where == VSLAM_IMGS_PINNED ? (uint32_t)pitch : (uint32_t)fe->h_img_pitch;
    }
    int from_host = 1;
    if (use_sdma) {
        const size_t spitch = hs.pitch0[0];
        const size_t one = (size_t)(p.height - 1) * spitch + p.width; /* bytes of one image, first to last pixel */
        const size_t stride = (one + 255) & ~(size_t)255;
        if (fe->d_stage_bytes < stride * fe->B + 256) { /* ensure_stage() runs before any capture; cannot happen */
            g_err = "internal: device staging buffer not allocated";
            return VSLAM_ERR_HIP;
        }
        hipStream_t cs = st;
        bool even = nimg > 1; /* equally spaced sources: one copy */
        const ptrdiff_t d = nimg > 1 ? hs.l0[1] - hs.l0[0] : 0;
        for (int s = 2; s < nimg && even; s++) even = hs.l0[s] - hs.l0[s - 1] == d;
        if (even && d > 0 && (size_t)d >= one && (size_t)d * (nimg - 1) + one <= fe->d_stage_bytes) {
            /* vslam_tuning.stage_split_event = k (0..3): the upload goes as TWO transfers and the context's user event k
             * (vslam_fe_event_wait) is recorded between them -- a pipelined caller that chains the uploads of its contexts
             * on that event keeps a second transfer queued behind the running one, so the link does not idle for the
             * hand-over between two chained uploads */
            const int ek = fe->tune.stage_split_event;
            const int h1 = nimg / 2;
            if (ek >= 0 && ek < 4 && h1 >= 1 && !hip_stream_capturing(cs)) {
                HIPCHK(hipMemcpyAsync(fe->d_stage, hs.l0[0], (size_t)d * h1, hipMemcpyHostToDevice, cs));
                if (!fe->ev_user[ek]) HIPCHK(hipEventCreateWithFlags(&fe->ev_user[ek], hipEventDisableTiming));
                HIPCHK(hipEventRecord(fe->ev_user[ek], cs));
                HIPCHK(hipMemcpyAsync(fe->d_stage + (size_t)d * h1, hs.l0[0] + (size_t)d * h1, (size_t)d * (nimg - 1 - h1) + one,
                                      hipMemcpyHostToDevice, cs));
            } else
                HIPCHK(hipMemcpyAsync(fe->d_stage, hs.l0[0], (size_t)d * (nimg - 1) + one, hipMemcpyHostToDevice, cs));
            for (int s = 0; s < nimg; s++) hs.l0[s] = fe->d_stage + (size_t)d * s;
        } else {
            for (int s = 0; s < nimg; s++) {
                HIPCHK(hipMemcpyAsync(fe->d_stage + stride * s, hs.l0[s], one, hipMemcpyHostToDevice, cs));
                hs.l0[s] = fe->d_stage + stride * s;
            }
        }
        from_host = 0;
    }
    vk_pull_images(st, hs, fe->d_pyr, fe->slot_stride, fe->geom.lv[0].off, (int)lp, p.width, p.height, nimg, from_host, fe->tune);
    return VSLAM_OK;
}

static int enqueue_front(vslam_fe* fe, int nimg, const uint8_t* const* imgs, size_t pitch, int on_device) {
    const vslam_fe_params& p = fe->p;
    const int L = p.nlevels;
    hipStream_t st = fe->stream;
    if (on_device == VSLAM_IMGS_DEVICE) {
        for (int s = 0; s < nimg; s++) {
            if (!imgs[s]) {
                g_err = "null image";
                return VSLAM_ERR_INVALID;
            }
            fe->src.l0[s] = imgs[s];
            fe->src.pitch0[s] = (uint32_t)pitch;
        }
    } else if (on_device == VSLAM_IMGS_STAGED) {
        for (int s = 0; s < nimg; s++) { /* vslam_fe_stage_images_async put them there */
            fe->src.l0[s] = fe->d_pyr + (size_t)s * fe->slot_stride + fe->geom.lv[0].off;
            fe->src.pitch0[s] = (uint32_t)fe->geom.lv[0].pitch;
        }
    } else {
        int rc = upload_host_rows(fe, nimg, imgs, pitch, on_device);
        if (rc) return rc;
        for (int s = 0; s < nimg; s++) {
            fe->src.l0[s] = fe->d_pyr + (size_t)s * fe->slot_stride + fe->geom.lv[0].off;
            fe->src.pitch0[s] = (uint32_t)fe->geom.lv[0].pitch;
        }
    }
    fe->last_nimg = nimg;
    fe->cand_on_host = false;
    /* per-slot candidate header (total, overflow) and the device-quadtree error word: one tiny kernel instead
     * of two runtime memsets */
    int32_t* d_errw = fe->dev_octree ? fe->d_counts + (size_t)fe->B * 4 : nullptr;
    if (fe->pyr_groups.empty()) vk_reset_headers(st, fe->d_cand, fe->cand_stride, nimg, d_errw); /* else: the first pyramid launch does it */
    const bool prof = fe->profiling;
    if (prof) HIPCHK(hipEventRecord(fe->ev_prof[0], st));
    bool first_group = true;
    for (const auto& g : fe->pyr_groups) {
        vk_pyramid_group(st, fe->d_pyr, fe->slot_stride, fe->src, g.dev, g.lds_bytes, nimg, fe->tune,
                         first_group ? fe->d_cand : nullptr, fe->cand_stride, first_group ? d_errw : nullptr);
        first_group = false;
    }
    for (int l = 1; l < L && fe->pyr_groups.empty(); l++) {
        if (fe->d_quads[l])
            vk_resize_level_v2(st, fe->d_pyr, fe->slot_stride, fe->src, fe->geom.lv[l - 1], fe->geom.lv[l], l - 1,
                               fe->d_qbase[l], fe->d_quads[l], fe->d_ytab[l], fe->d_yb[l], nimg);
        else
            vk_resize_level(st, fe->d_pyr, fe->slot_stride, fe->src, fe->geom.lv[l - 1], fe->geom.lv[l], l - 1,
                            fe->d_xtab[l], fe->d_xa[l], fe->d_ytab[l], fe->d_yb[l], nimg);
    }
    if (prof) HIPCHK(hipEventRecord(fe->ev_prof[1], st));
    if (fe->fast_gate && fe->fast_gate->ev_fast) HIPCHK(hipStreamWaitEvent(st, fe->fast_gate->ev_fast, 0));
    /* vslam_tuning.fast_kernel: 4 = one workgroup per band of cells (default for batches: 17 % fewer instructions, every
     * image byte fetched once), 3 = one per cell (default for contexts of one or two images, where the launch is a frame's
     * latency: 15 instead of 21 us, four times as many and shorter workgroups) */
    if (fe->nbands > 0 && tune_or(fe->tune.fast_kernel, fe->B <= 2 ? 3 : 4) != 3)
        vk_fast_bands(st, fe->d_pyr, fe->slot_stride, fe->src, fe->geom, fe->d_bands, fe->nbands, fe->d_band_classes, fe->d_cells,
                      (int)fe->cells.size(), fe->d_cand, fe->cand_stride, p.ini_th_fast, p.min_th_fast, fe->band_max_wh, fe->band_max_iw, nimg,
                      fe->tune);
    else
        vk_fast_cells_v3(st, fe->d_pyr, fe->slot_stride, fe->src, fe->geom, fe->d_cells, (int)fe->cells.size(), fe->d_cand,
                         fe->cand_stride, p.ini_th_fast, p.min_th_fast, fe->tile_rows, fe->tile_pitch, fe->max_px, nimg, fe->tune);
    if (fe->fast_gated_by_someone && fe->ev_fast) HIPCHK(hipEventRecord(fe->ev_fast, st));
    if (prof) HIPCHK(hipEventRecord(fe->ev_prof[2], st));
    return VSLAM_OK;
}

static int fetch_candidates(vslam_fe* fe, int nimg, bool everything) {
    /* header + cell table + a first chunk speculatively; the rest only if a slot needs it */
    hipStream_t st = fe->stream;
    const size_t hdr_bytes = 8 + fe->cells.size() * sizeof(CellOut);
    (void)everything; /* fixed per-cell segments: the used entries are scattered, copy the whole region */
    const size_t first_cands = (size_t)fe->cand_cap;
    HIPCHK(hipMemcpy2DAsync(fe->h_cand, fe->cand_stride, fe->d_cand, fe->cand_stride, hdr_bytes + first_cands * 4,
                            nimg, hipMemcpyDeviceToHost, st));
    HIPCHK(hipEventRecord(fe->ev_cand, st));
    return VSLAM_OK;
}

static int wait_candidates(vslam_fe* fe, int nimg) {
    hipStream_t st = fe->stream;
    HIPCHK(hipEventSynchronize(fe->ev_cand));
    for (int s = 0; s < nimg; s++) {
        const uint32_t* hdr = (const uint32_t*)(fe->h_cand + (size_t)s * fe->cand_stride);
        if (hdr[1]) {
            g_err = "FAST candidate buffer overflow";
            return VSLAM_ERR_CAPACITY;
        }
    }
    (void)st;
    fe->cand_on_host = true;
    return VSLAM_OK;
}

static void enqueue_blur(vslam_fe* fe, int nimg) {
    vk_blur7_v2(fe->stream, fe->d_pyr, fe->slot_stride, fe->src, fe->geom, fe->d_blur, fe->d_blur_tasks,
                fe->n_blur_tasks, fe->taps, fe->blur_rows, nimg);
}

/* quadtree on the host (fallback when the node list does not fit LDS, or VSLAM_FLAG_HOST_OCTREE) */
static int enqueue_back_host(vslam_fe* fe, int nimg, int lap0, int lap1) {
    const vslam_fe_params& p = fe->p;
    const int L = p.nlevels;
    hipStream_t st = fe->stream;
    const bool prof = fe->profiling;
    int rc = fetch_candidates(fe, nimg, false);
    if (rc) return rc;
    if (prof) HIPCHK(hipEventRecord(fe->ev_prof[3], st));
    enqueue_blur(fe, nimg); /* runs while the host distributes */
    if (prof) HIPCHK(hipEventRecord(fe->ev_prof[4], st));
    HIPCHK(hipGetLastError());
    if ((rc = wait_candidates(fe, nimg))) return rc;
    std::atomic<int> bad(0);
    fe->pool->parallel_for(nimg * L, [&](int task) {
        const int s = task / L, l = task % L;
        decode_candidates(fe, s, l);
        const std::vector<vslam::Cand>& cl = fe->cand_level[(size_t)s * L + l];
        const LevelGeom& g = fe->geom.lv[l];
        if (!vslam::distribute_octree(cl.data(), (int)cl.size(), g.w - 2 * VSLAM_FAST_BORDER,
                                      g.h - 2 * VSLAM_FAST_BORDER, fe->tab.quota[l],
                                      fe->sel_level[(size_t)s * L + l]))
            bad++;
    });
    if (bad.load()) {
        g_err = "DistributeOctTree: nIni == 0";
        return VSLAM_ERR_UNSUPPORTED;
    }
    /* output order (fextractor.cpp:1071-1129): level-major, lapping-area keypoints from the tail */
    int nsel = 0;
    for (int s = 0; s < nimg; s++) {
        int nk = 0;
        for (int l = 0; l < L; l++) nk += (int)fe->sel_level[(size_t)s * L + l].size();
        if (nk > fe->cap) {
            g_err = "internal keypoint capacity exceeded";
            return VSLAM_ERR_CAPACITY;
        }
        int monoIndex = 0, stereoIndex = nk - 1;
        for (int l = 0; l < L; l++) {
            const float scale = fe->tab.scale[l];
            for (const vslam::Cand& c : fe->sel_level[(size_t)s * L + l]) {
                const int lx = c.x + VSLAM_FAST_BORDER, ly = c.y + VSLAM_FAST_BORDER;
                float px = (float)lx;
                if (l != 0) px = px * scale;
                SelKp k;
                k.x = (uint16_t)lx;
                k.y = (uint16_t)ly;
                k.level = (uint8_t)l;
                k.slot = (uint8_t)s;
                k.response = c.response;
                k.pad = 0;
                k.out = (px >= (float)lap0 && px <= (float)lap1) ? (uint32_t)stereoIndex-- : (uint32_t)monoIndex++;
                fe->h_sel[nsel++] = k;
            }
        }
        fe->n_out[s] = nk;
        fe->mono_out[s] = monoIndex;
        fe->h_counts[s * 4] = nk;
        fe->h_counts[s * 4 + 1] = monoIndex;
    }
    /* device-side consumers (stereo matcher) read the counts from HBM */
    HIPCHK(hipMemcpyAsync(fe->d_counts, fe->h_counts, (size_t)nimg * 16, hipMemcpyHostToDevice, st));
    if (nsel) {
        HIPCHK(hipMemcpyAsync(fe->d_sel, fe->h_sel, (size_t)nsel * sizeof(SelKp), hipMemcpyHostToDevice, st));
        if (prof) HIPCHK(hipEventRecord(fe->ev_prof[5], st));
        vk_orient_describe(st, fe->d_pyr, fe->d_blur, fe->slot_stride, fe->src, fe->geom, fe->d_sel, nsel,
                           fe->d_pattern, fe->d_kps, fe->d_desc, fe->cap, (p.flags & VSLAM_FLAG_ATAN_FMA) ? 1 : 0);
        if (prof) HIPCHK(hipEventRecord(fe->ev_prof[6], st));
    } else if (prof) {
        HIPCHK(hipEventRecord(fe->ev_prof[5], st));
        HIPCHK(hipEventRecord(fe->ev_prof[6], st));
    }
    if (prof) {
        HIPCHK(hipEventRecord(fe->ev_prof[7], st));
        HIPCHK(hipEventRecord(fe->ev_prof[8], st));
    }
    HIPCHK(hipGetLastError());
    return VSLAM_OK;
}

/* quadtree on the device: nothing returns to the host until the results do */
static int enqueue_back_dev(vslam_fe* fe, int nimg, int lap0, int lap1) {
    const vslam_fe_params& p = fe->p;
    hipStream_t st = fe->stream;
    const bool prof = fe->profiling;
    int32_t* d_err = fe->d_counts + (size_t)fe->B * 4;
    if (prof) HIPCHK(hipEventRecord(fe->ev_prof[3], st));
    enqueue_blur(fe, nimg);
    if (prof) HIPCHK(hipEventRecord(fe->ev_prof[4], st));
    if (prof) HIPCHK(hipEventRecord(fe->ev_prof[7], st));
    if (fe->oct.parts)
        vk_oct_count(st, fe->d_cand, fe->cand_stride, (int)fe->cells.size(), fe->oct, fe->d_pts[0], fe->d_pts[1], (size_t)fe->cand_cap,
                     p.nlevels, nimg, fe->oct_maxcells);
    vk_octree(st, fe->d_cand, fe->cand_stride, (int)fe->cells.size(), fe->oct, fe->d_pts[0], fe->d_pts[1],
              fe->d_nid, fe->d_oct_sorted, (size_t)fe->cand_cap, fe->d_sel_xyr, fe->d_sel_cnt, d_err, p.nlevels, nimg,
              fe->d_oct_redo, fe->tune.oct_regkeys, fe->oct_threads, wave_prio_on(fe->tune, 1));
    vk_assign_out(st, fe->oct, fe->geom, fe->d_sel_xyr, fe->d_sel_cnt, lap0, lap1, fe->d_sel, fe->d_counts, fe->cap,
                  d_err, nimg, fe->d_oct_redo, wave_prio_on(fe->tune, 1));
    if (prof) HIPCHK(hipEventRecord(fe->ev_prof[8], st));
    if (prof) HIPCHK(hipEventRecord(fe->ev_prof[5], st));
    vk_orient_describe_dev(st, fe->d_pyr, fe->d_blur, fe->slot_stride, fe->src, fe->geom, fe->d_sel, fe->d_counts,
                           fe->d_pattern, fe->d_kps, fe->d_desc, fe->cap, (p.flags & VSLAM_FLAG_ATAN_FMA) ? 1 : 0,
                           nimg, wave_prio_on(fe->tune, 2), fe->tune.desc_kpw >= 0 ? fe->tune.desc_kpw : fe->desc_kpw_hint);
    if (prof) HIPCHK(hipEventRecord(fe->ev_prof[6], st));
    HIPCHK(hipGetLastError());
    return VSLAM_OK; /* counts travel to the host with the results (vslam_enqueue_extract) */
}

/* VSLAM_HOST_PROF=1: host-side wall time per API phase, printed by vslam_fe_destroy (diagnostics only) */
static double g_hp[6];
static long g_hp_n;
static const bool g_hp_on = vslam_process_tuning().host_prof == 1; /* process-wide */
static inline double hp_now() {
    return g_hp_on ? std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch())
                         .count()
                   : 0.0;
}
void vslam_host_prof_report() {
    if (!g_hp_on || !g_hp_n) return;
    fprintf(stderr, "[vslam host prof] per call (us): front %.1f back %.1f d2h-enqueue %.1f sync %.1f deliver %.1f (n=%ld)\n",
            g_hp[0] / g_hp_n, g_hp[1] / g_hp_n, g_hp[2] / g_hp_n, g_hp[3] / g_hp_n, g_hp[4] / g_hp_n, g_hp_n);
    g_hp_n = 0;
    for (double& v : g_hp) v = 0;
}

static int enqueue_extract_plain(vslam_fe* fe, int nimg, const uint8_t* const* imgs, size_t pitch, int on_device,
                                 int lap0, int lap1, int want_host);

/* want_host: 0 results stay in HBM (counts still reach the host), 1 delivered by this pass, 2 delivery DEFERRED to the
 * device SearchForInitialization that follows on this context (one transfer for both); if none follows, the wait delivers */
static int enqueue_extract_impl(vslam_fe* fe, int nimg, const uint8_t* const* imgs, size_t pitch, int on_device, int lap0,
                                int lap1, int want_host);

int vslam_enqueue_extract(vslam_fe* fe, int nimg, const uint8_t* const* imgs, size_t pitch, int on_device,
                          int lap0, int lap1, int want_host) {
    const int rc = enqueue_extract_impl(fe, nimg, imgs, pitch, on_device, lap0, lap1, want_host);
    /* set on every path, a replayed graph included (the captured pass of want_host = 2 holds no copy) */
    fe->deliver_deferred = rc == VSLAM_OK && want_host == 2 && fe->dev_octree && nimg == fe->B && fe->res_init_bytes != 0;
    return rc;
}

static int enqueue_extract_impl(vslam_fe* fe, int nimg, const uint8_t* const* imgs, size_t pitch, int on_device, int lap0,
                                int lap1, int want_host) {
    HIPCHK(hipSetDevice(fe->p.device));
    if (on_device == VSLAM_IMGS_HOST) {
        int rc = stage_host_images(fe, nimg, imgs, pitch);
        if (rc) return rc;
    }
    if (on_device == VSLAM_IMGS_HOST || on_device == VSLAM_IMGS_PINNED) {
        int rc = ensure_stage(fe, on_device == VSLAM_IMGS_PINNED ? pitch : (size_t)fe->geom.lv[0].pitch, nimg);
        if (rc) return rc;
    }
    /* Host-image passes of one shape replay a captured HIP graph: every kernel argument of such a pass is fixed
     * (staging, pyramid and result buffers belong to the context), so the ~20 launches become one hipGraphLaunch. */
    const bool graphable = fe->use_graph && on_device != VSLAM_IMGS_DEVICE && fe->dev_octree && !fe->profiling;
    if (!graphable) return enqueue_extract_plain(fe, nimg, imgs, pitch, on_device, lap0, lap1, want_host);
    long long key = ((long long)nimg << 48) ^ ((long long)(uint16_t)lap0 << 32) ^ ((long long)(uint16_t)lap1 << 16) ^
                    (long long)(want_host & 3) ^ ((long long)(lap0 >> 16) << 40) ^ ((long long)(lap1 >> 16) << 24) ^
                    ((long long)on_device << 56); /* host / pinned / staged passes capture different launches */
    if (on_device == VSLAM_IMGS_HOST && fe->h_img_pitch != (size_t)fe->geom.lv[0].pitch) key ^= 1ll << 59; /* dense staging rows */
    if (on_device == VSLAM_IMGS_PINNED) {
        /* the caller's pointers are kernel arguments of the captured pull: a different set of images is a different
         * graph (a capture-card ring of a few buffers per context hits the cache every time) */
        unsigned long long hsh = 0x9E3779B97F4A7C15ull ^ (unsigned long long)pitch;
        for (int s = 0; s < nimg; s++) hsh = (hsh ^ (unsigned long long)(uintptr_t)imgs[s]) * 0x100000001B3ull;
        key ^= (long long)(hsh | 2ull);
    }
    const bool same_imgs = on_device != VSLAM_IMGS_PINNED ||
                           (fe->graph_pitch == pitch && !memcmp(fe->graph_imgs, imgs, (size_t)nimg * sizeof(imgs[0])));
    if (fe->graph_exec && fe->graph_key == key && fe->graph_lap0 == lap0 && fe->graph_lap1 == lap1 && same_imgs) {
        fe->last_nimg = nimg;
        fe->cand_on_host = false;
        /* what enqueue_front records on a plain pass: later NON-captured consumers (stereo refinement,
         * vslam_fe_level_copy level 0) read fe->src, which a device-image pass in between may have pointed at the
         * caller's (by now possibly freed) images */
        for (int s = 0; s < nimg; s++) {
            fe->src.l0[s] = fe->d_pyr + (size_t)s * fe->slot_stride + fe->geom.lv[0].off;
            fe->src.pitch0[s] = (uint32_t)fe->geom.lv[0].pitch;
        }
        HIPCHK(hipGraphLaunch(fe->graph_exec, fe->stream));
        fe->n_deliveries += fe->graph_deliveries;
        fe->n_delivery_bytes += fe->graph_delivery_bytes;
        return VSLAM_OK;
    }
    if (fe->graph_exec) {
        hipGraphExecDestroy(fe->graph_exec);
        fe->graph_exec = nullptr;
    }
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(fe->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        fe->use_graph = false; /* capture unavailable: plain launches from now on */
        return enqueue_extract_plain(fe, nimg, imgs, pitch, on_device, lap0, lap1, want_host);
    }
    const unsigned long long del0 = fe->n_deliveries, delb0 = fe->n_delivery_bytes;
    int rc = enqueue_extract_plain(fe, nimg, imgs, pitch, on_device, lap0, lap1, want_host);
    const hipError_t ce = hipStreamEndCapture(fe->stream, &graph);
    fe->graph_deliveries = fe->n_deliveries - del0; /* what one replay of this graph sends to the host */
    fe->graph_delivery_bytes = fe->n_delivery_bytes - delb0;
    fe->n_deliveries = del0; /* the capture itself ran nothing */
    fe->n_delivery_bytes = delb0;
    if (rc != VSLAM_OK || ce != hipSuccess || !graph ||
        hipGraphInstantiate(&fe->graph_exec, graph, nullptr, nullptr, 0) != hipSuccess) {
        if (graph) hipGraphDestroy(graph);
        fe->graph_exec = nullptr;
        fe->use_graph = false;
        if (rc != VSLAM_OK) return rc;
        return enqueue_extract_plain(fe, nimg, imgs, pitch, on_device, lap0, lap1, want_host);
    }
    hipGraphDestroy(graph);
    fe->graph_key = key;
    fe->graph_pitch = pitch;
    /* only a pinned pass bakes the caller's pointers into the graph; STAGED passes may come with imgs == NULL */
    if (on_device == VSLAM_IMGS_PINNED) memcpy(fe->graph_imgs, imgs, (size_t)nimg * sizeof(imgs[0]));
    else memset(fe->graph_imgs, 0, sizeof(fe->graph_imgs));
    fe->graph_lap0 = lap0;
    fe->graph_lap1 = lap1;
    HIPCHK(hipGraphLaunch(fe->graph_exec, fe->stream));
    fe->n_deliveries += fe->graph_deliveries;
    fe->n_delivery_bytes += fe->graph_delivery_bytes;
    return VSLAM_OK;
}

static int enqueue_extract_plain(vslam_fe* fe, int nimg, const uint8_t* const* imgs, size_t pitch, int on_device,
                                 int lap0, int lap1, int want_host) {
    const double t0 = hp_now();
    int rc = enqueue_front(fe, nimg, imgs, pitch, on_device);
    if (rc) return rc;
    const double t1 = hp_now();
    rc = fe->dev_octree ? enqueue_back_dev(fe, nimg, lap0, lap1) : enqueue_back_host(fe, nimg, lap0, lap1);
    if (rc) return rc;
    const double t2 = hp_now();
    g_hp[0] += t1 - t0;
    g_hp[1] += t2 - t1;
    g_hp_n++;
    struct D2hTimer {
        double t;
        ~D2hTimer() { g_hp[2] += hp_now() - t; }
    } d2h_timer{t2};
    hipStream_t st = fe->stream;
    /* results go to pinned host memory by a kernel (whole blocks: the host does not know the counts yet) */
    CopyRanges R;
    memset(&R, 0, sizeof(R));
    if (want_host == 2 && fe->dev_octree && nimg == fe->B && fe->res_init_bytes != 0)
        return VSLAM_OK; /* deferred: vslam_search_init_dev_async (or the wait) sends the block; vslam_enqueue_extract sets the flag */
    if (want_host && fe->dev_octree && nimg == fe->B) {
        /* a full batch: counts, keypoints and descriptors are one contiguous block -> ONE transfer */
        R.dst[0] = fe->h_res;
        R.src[0] = fe->d_res;
        R.bytes[0] = fe->res_feat_bytes;
        R.n = 1;
        vslam_count_delivery(fe, vk_copy_ranges(st, R, fe->tune), R);
        HIPCHK(hipGetLastError());
        return VSLAM_OK;
    }
    if (fe->dev_octree) {
        R.dst[R.n] = fe->h_counts;
        R.src[R.n] = fe->d_counts;
        R.bytes[R.n++] = (size_t)(fe->B * 4 + 4) * 4;
    }
    if (want_host) {
        R.dst[R.n] = fe->h_kps;
        R.src[R.n] = fe->d_kps;
        R.bytes[R.n++] = (size_t)nimg * fe->cap * sizeof(vslam_kp);
        R.dst[R.n] = fe->h_desc;
        R.src[R.n] = fe->d_desc;
        R.bytes[R.n++] = (size_t)nimg * fe->cap * 32;
    }
    vslam_count_delivery(fe, vk_copy_ranges(st, R, fe->tune), R);
    HIPCHK(hipGetLastError());
    return VSLAM_OK;
}

int vslam_finish_extract(vslam_fe* fe, int nimg) {
    hipStream_t st = fe->stream;
    if (fe->deliver_deferred) { /* a deferred delivery that no matcher call picked up */
        fe->deliver_deferred = false;
        CopyRanges R;
        memset(&R, 0, sizeof(R));
        R.dst[0] = fe->h_res;
        R.src[0] = fe->d_res;
        R.bytes[0] = fe->res_feat_bytes;
        R.n = 1;
        vslam_count_delivery(fe, vk_copy_ranges(st, R, fe->tune), R);
        HIPCHK(hipGetLastError());
    }
    const double t_sync = hp_now();
    HIPCHK(vslam_stream_wait(st));
    g_hp[3] += hp_now() - t_sync;
    if (fe->dev_octree) {
        const int32_t* err = fe->h_counts + (size_t)fe->B * 4;
        if (err[0] & 1) {
            g_err = "FAST candidate buffer overflow (or > 65535 candidates on one level)";
            return VSLAM_ERR_CAPACITY;
        }
        if (err[0] & 2) {
            g_err = "internal keypoint capacity exceeded";
            return VSLAM_ERR_CAPACITY;
        }
        for (int s = 0; s < nimg; s++) {
            fe->n_out[s] = fe->h_counts[s * 4];
            fe->mono_out[s] = fe->h_counts[s * 4 + 1];
            fe->oct_last_mask[s] = (uint32_t)fe->h_counts[s * 4 + 2];
            fe->oct_deep += (unsigned)__builtin_popcount(fe->oct_last_mask[s]);
        }
        fe->oct_problems += (unsigned long long)nimg * fe->p.nlevels;
    }
    if (fe->profiling) {
        float ms;
        static const int span[5][2] = {{0, 1}, {1, 2}, {3, 4}, {5, 6}, {7, 8}};
        for (int i = 0; i < 5; i++) {
            HIPCHK(hipEventElapsedTime(&ms, fe->ev_prof[span[i][0]], fe->ev_prof[span[i][1]]));
            fe->prof_ms[i] += ms;
        }
        fe->prof_batches++;
        fe->prof_images += nimg;
    }
    return VSLAM_OK;
}

int vslam_deliver(vslam_fe* fe, int nimg, vslam_kp* const* kps, uint8_t* const* desc, int cap, int* n,
                  int* mono_index) {
    struct DeliverTimer {
        double t;
        ~DeliverTimer() { g_hp[4] += hp_now() - t; }
    } deliver_timer{hp_now()};
    for (int s = 0; s < nimg; s++) {
        if (kps && desc) {
            if (fe->n_out[s] > cap) {
                g_err = "caller keypoint capacity too small";
                return VSLAM_ERR_CAPACITY;
            }
            if (fe->n_out[s]) {
                memcpy(kps[s], fe->h_kps + (size_t)s * fe->cap, (size_t)fe->n_out[s] * sizeof(vslam_kp));
                memcpy(desc[s], fe->h_desc + (size_t)s * fe->cap * 32, (size_t)fe->n_out[s] * 32);
            }
        }
        if (n) n[s] = fe->n_out[s];
        if (mono_index) mono_index[s] = fe->mono_out[s];
    }
    return VSLAM_OK;
}

extern "C" int vslam_fe_extract_batch(vslam_fe* fe, int nimg, const uint8_t* const* imgs, size_t pitch,
                                      int imgs_on_device, int lap0, int lap1, vslam_kp* const* kps,
                                      uint8_t* const* desc, int cap, int* n, int* mono_index) {
    if (!fe || (!imgs && imgs_on_device != VSLAM_IMGS_STAGED) || nimg < 1 || nimg > fe->B ||
        (pitch < (size_t)fe->p.width && imgs_on_device != VSLAM_IMGS_STAGED) || imgs_on_device < 0 ||
        imgs_on_device > VSLAM_IMGS_STAGED) {
        g_err = "invalid arguments";
        return VSLAM_ERR_INVALID;
    }
    int rc = vslam_enqueue_extract(fe, nimg, imgs, pitch, imgs_on_device, lap0, lap1, kps && desc);
    if (rc == VSLAM_OK) rc = vslam_finish_extract(fe, nimg);
    if (rc != VSLAM_OK) {
        hipStreamSynchronize(fe->stream);
        return rc;
    }
    return vslam_deliver(fe, nimg, kps, desc, cap, n, mono_index);
}

/* split form: enqueue everything (no host synchronisation in the device-quadtree path), collect later */
extern "C" int vslam_fe_extract_batch_async(vslam_fe* fe, int nimg, const uint8_t* const* imgs, size_t pitch,
                                            int imgs_on_device, int lap0, int lap1, int want_host) {
    if (!fe || (!imgs && imgs_on_device != VSLAM_IMGS_STAGED) || nimg < 1 || nimg > fe->B ||
        (pitch < (size_t)fe->p.width && imgs_on_device != VSLAM_IMGS_STAGED) || imgs_on_device < 0 ||
        imgs_on_device > VSLAM_IMGS_STAGED) {
        g_err = "invalid arguments";
        return VSLAM_ERR_INVALID;
    }
    int rc = vslam_enqueue_extract(fe, nimg, imgs, pitch, imgs_on_device, lap0, lap1, want_host < 0 ? 0 : want_host > 2 ? 1 : want_host);
    if (rc != VSLAM_OK) hipStreamSynchronize(fe->stream);
    return rc;
}

extern "C" int vslam_fe_extract_wait(vslam_fe* fe, vslam_kp* const* kps, uint8_t* const* desc, int cap, int* n,
                                     int* mono_index) {
    if (!fe || fe->last_nimg < 1) {
        g_err = "nothing enqueued";
        return VSLAM_ERR_INVALID;
    }
    int rc = vslam_finish_extract(fe, fe->last_nimg);
    if (rc != VSLAM_OK) return rc;
    return vslam_deliver(fe, fe->last_nimg, kps, desc, cap, n, mono_index);
}

extern "C" int vslam_fe_extract(vslam_fe* fe, const uint8_t* img, size_t pitch, int lap0, int lap1,
                                vslam_kp* kps, uint8_t* desc, int cap, int* n, int* mono_index) {
    if (!fe || !img || !kps || !desc || !n) {
        g_err = "invalid arguments";
        if (mono_index) *mono_index = -1; /* FExtractor::compute returns -1 on an empty image */
        return VSLAM_ERR_INVALID;
    }
    const uint8_t* imgs[1] = {img};
    vslam_kp* k[1] = {kps};
    uint8_t* d[1] = {desc};
    return vslam_fe_extract_batch(fe, 1, imgs, pitch, 0, lap0, lap1, k, d, cap, n, mono_index);
}

extern "C" int vslam_fe_stage_images_async(vslam_fe* fe, int nimg, const uint8_t* const* imgs, size_t pitch,
                                           int where) {
    if (!fe || !imgs || nimg < 1 || nimg > fe->B || pitch < (size_t)fe->p.width ||
        (where != VSLAM_IMGS_PINNED && where != VSLAM_IMGS_HOST)) {
        g_err = "invalid arguments";
        return VSLAM_ERR_INVALID;
    }
    HIPCHK(hipSetDevice(fe->p.device));
    if (where == VSLAM_IMGS_HOST) {
        HIPCHK(vslam_stream_wait(fe->stream)); /* the pinned staging may still be read by the previous pull */
        int rc = stage_host_images(fe, nimg, imgs, pitch);
        if (rc) return rc;
    }
    int rc2 = ensure_stage(fe, where == VSLAM_IMGS_PINNED ? pitch : (size_t)fe->geom.lv[0].pitch, nimg);
    if (rc2) return rc2;
    /* (Tried in round 2 and removed: the DMA copy on an upload stream of its own, issued passes ahead of its use.  One
     * upload stream serialises copies that otherwise overlap on several DMA engines -- 77-86 k against 85-103 k mono
     * frames/s -- and the host-input rate is bound by the link either way.) */
    rc2 = upload_host_rows(fe, nimg, imgs, pitch, where);
    if (rc2) return rc2;
    HIPCHK(hipGetLastError());
    return VSLAM_OK;
}

extern "C" int vslam_fe_candidates(vslam_fe* fe, int slot, int level, vslam_kp* out, int cap) {
    if (!fe || slot < 0 || slot >= fe->B || level < 0 || level >= fe->p.nlevels || slot >= fe->last_nimg)
        return VSLAM_ERR_INVALID;
    if (!fe->cand_on_host) { /* device-quadtree path: the candidates never left HBM; fetch them now */
        HIPCHK(hipSetDevice(fe->p.device));
        int rc = fetch_candidates(fe, fe->last_nimg, true);
        if (rc == VSLAM_OK) rc = wait_candidates(fe, fe->last_nimg);
        if (rc) return rc;
    }
    if (fe->dev_octree || true) decode_candidates(fe, slot, level);
    const std::vector<vslam::Cand>& cl = fe->cand_level[(size_t)slot * fe->p.nlevels + level];
    for (int i = 0; i < (int)cl.size() && i < cap && out; i++) {
        out[i].x = (float)cl[i].x;
        out[i].y = (float)cl[i].y;
        out[i].size = 7.f;
        out[i].angle = -1.f;
        out[i].response = (float)cl[i].response;
        out[i].octave = 0;
        out[i].class_id = -1;
    }
    return (int)cl.size();
}

extern "C" int vslam_fe_slot_count_ptr(vslam_fe* fe, int slot, const int32_t** dev_n) {
    if (!fe || slot < 0 || slot >= fe->B || !dev_n) return VSLAM_ERR_INVALID;
    *dev_n = fe->d_counts + slot * 4;
    return VSLAM_OK;
}

extern "C" int vslam_fe_slot_buffers(vslam_fe* fe, int slot, const vslam_kp** dev_kps, const uint8_t** dev_desc,
                                     int* n) {
    if (!fe || slot < 0 || slot >= fe->B) return VSLAM_ERR_INVALID;
    if (dev_kps) *dev_kps = fe->d_kps + (size_t)slot * fe->cap;
    if (dev_desc) *dev_desc = fe->d_desc + (size_t)slot * fe->cap * 32;
    if (n) *n = fe->n_out[slot];
    return VSLAM_OK;
}

extern "C" int vslam_fe_capacity(const vslam_fe* fe) { return fe ? fe->cap : VSLAM_ERR_INVALID; }


/* ------------------------------------------------------------------ pinned host memory on the GPU's NUMA node
 * hipHostMalloc takes its pages from the node the calling thread happens to run on; across the socket boundary the
 * link delivered noticeably less (host-input rate of the mono workload: 79-84 k frames/s on some runs, 96 k on others).
 * The allocating thread is moved onto the CPUs next to the device for the duration of the allocation and its first
 * touch (sysfs: /sys/bus/pci/devices/<bus id>/local_cpulist), then back.  VSLAM_NUMA=0 disables. */
#include <sched.h>
static bool device_cpuset(cpu_set_t* out) {
    static std::mutex mu; /* contexts may be created from several threads */
    std::lock_guard<std::mutex> lk(mu);
    /* one entry per device (the current one): a process with contexts on several GPUs gets each device's own CPUs */
    struct PerDevice {
        int state = 0; /* 0 unknown, 1 unavailable, 2 cached */
        cpu_set_t set;
    };
    static PerDevice per_device[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    int& state = per_device[dev].state;
    cpu_set_t& cached = per_device[dev].set;
    if (state == 0) {
        state = 1;
        const bool off = vslam_process_tuning().numa == 0; /* process-wide */
        char bus[64] = {0};
        if (!off && hipDeviceGetPCIBusId(bus, (int)sizeof(bus), dev) == hipSuccess) {
            for (char* c = bus; *c; c++) *c = (char)tolower(*c);
            char path[160];
            snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/local_cpulist", bus);
            if (FILE* fp = fopen(path, "r")) {
                char line[1024] = {0};
                if (fgets(line, sizeof(line), fp)) {
                    CPU_ZERO(&cached);
                    int n = 0;
                    char* save = nullptr;
                    for (char* tok = strtok_r(line, ",\n", &save); tok; tok = strtok_r(nullptr, ",\n", &save)) {
                        int a = 0, b = 0;
                        const int k = sscanf(tok, "%d-%d", &a, &b);
                        if (k == 1) b = a;
                        if (k >= 1)
                            for (int c = a; c <= b && c < CPU_SETSIZE; c++) {
                                CPU_SET(c, &cached);
                                n++;
                            }
                    }
                    if (n > 0) state = 2;
                }
                fclose(fp);
            }
        }
    }
    if (state == 2) *out = cached;
    return state == 2;
}
int vslam_pinned_alloc(void** p, size_t bytes) {
    cpu_set_t near, saved;
    const bool move = device_cpuset(&near) && sched_getaffinity(0, sizeof(saved), &saved) == 0;
    bool moved = false;
    if (move) {
        cpu_set_t both;
        CPU_AND(&both, &near, &saved); /* never leave the set the process is allowed on */
        if (CPU_COUNT(&both) > 0) moved = sched_setaffinity(0, sizeof(both), &both) == 0;
    }
    const hipError_t rc = hipHostMalloc(p, bytes, hipHostMallocDefault);
    if (rc == hipSuccess && moved) memset(*p, 0, bytes); /* first touch while we are on the right node */
    if (moved) sched_setaffinity(0, sizeof(saved), &saved);
    return (int)rc;
}

extern "C" int vslam_host_alloc(size_t bytes, void** out) {
    if (!out || !bytes) return VSLAM_ERR_INVALID;
    *out = nullptr;
    HIPCHK((hipError_t)vslam_pinned_alloc(out, bytes));
    return VSLAM_OK;
}
extern "C" void vslam_host_free(void* p) {
    if (p) hipHostFree(p);
}

extern "C" int vslam_fe_slot_host_views(vslam_fe* fe, int slot, const vslam_kp** host_kps,
                                        const uint8_t** host_desc) {
    if (!fe || slot < 0 || slot >= fe->B) return VSLAM_ERR_INVALID;
    if (host_kps) *host_kps = fe->h_kps + (size_t)slot * fe->cap;
    if (host_desc) *host_desc = fe->h_desc + (size_t)slot * fe->cap * 32;
    return VSLAM_OK;
}

/* ------------------------------------------------------------------ matcher */
int vslam_ensure(void** p, size_t* have, size_t want) {
    if (*have >= want) return VSLAM_OK;
    if (*p) HIPCHK(hipFree(*p));
    *p = nullptr;
    *have = 0;
    HIPCHK(hipMalloc(p, want));
    *have = want;
    return VSLAM_OK;
}

int vslam_ensure_pinned(uint8_t** p, size_t* have, size_t want) {
    if (*have >= want) return VSLAM_OK;
    if (*p) HIPCHK(hipHostFree(*p));
    *p = nullptr;
    *have = 0;
    HIPCHK((hipError_t)vslam_pinned_alloc((void**)p, want));
    *have = want;
    return VSLAM_OK;
}

extern "C" int vslam_hamming_top2_batch(vslam_fe* fe, int nprob, const uint8_t* const* dev_q, const int32_t* nq,
                                        const uint8_t* const* dev_t, const int32_t* nt, int32_t* const* idx2,
                                        int32_t* const* dist2);
/* one problem = a batch of one (k_hamming_top2_batch splits the train set over as many workgroups as fill the GPU) */
extern "C" int vslam_hamming_top2(vslam_fe* fe, const uint8_t* dev_q, int nq, const uint8_t* dev_t, int nt,
                                  int32_t* idx2, int32_t* dist2) {
    if (!fe || nq < 0 || nt < 0 || nt > 65535 || (nq && (!dev_q || !idx2 || !dist2)) || (nt && !dev_t)) {
        g_err = "invalid arguments";
        return VSLAM_ERR_INVALID;
    }
    if (nq == 0) return VSLAM_OK;
    const int32_t nq1 = nq, nt1 = nt;
    return vslam_hamming_top2_batch(fe, 1, &dev_q, &nq1, &dev_t, &nt1, &idx2, &dist2);
}

extern "C" int vslam_hamming_top2_batch(vslam_fe* fe, int nprob, const uint8_t* const* dev_q, const int32_t* nq,
                                        const uint8_t* const* dev_t, const int32_t* nt, int32_t* const* idx2,
                                        int32_t* const* dist2) {
    if (!fe || nprob < 0 || nprob > VSLAM_MAX_TOP2_JOBS || (nprob && (!dev_q || !nq || !dev_t || !nt || !idx2 || !dist2))) {
        g_err = "invalid arguments";
        return VSLAM_ERR_INVALID;
    }
    Top2Jobs J;
    memset(&J, 0, sizeof(J));
    size_t rows = 0;
    int max_nq = 0, max_nt = 0;
    for (int p = 0; p < nprob; p++) {
        if (nq[p] < 0 || nt[p] < 0 || nt[p] > 65535 || (nq[p] && (!dev_q[p] || !idx2[p] || !dist2[p])) || (nt[p] && !dev_t[p])) {
            g_err = "invalid arguments";
            return VSLAM_ERR_INVALID;
        }
        J.job[p].q = (const uint32_t*)dev_q[p];
        J.job[p].t = (const uint32_t*)dev_t[p];
        J.job[p].nq = nq[p];
        J.job[p].nt = nt[p];
        J.job[p].row0 = (uint32_t)rows;
        rows += (size_t)nq[p];
        max_nq = std::max(max_nq, nq[p]);
        max_nt = std::max(max_nt, nt[p]);
    }
    if (rows == 0) return VSLAM_OK;
    HIPCHK(hipSetDevice(fe->p.device));
    const int nsplit = vk_hamming_top2_batch_split(nprob, max_nq, max_nt);
    int rc;
    if ((rc = vslam_ensure((void**)&fe->d_part, &fe->part_bytes, rows * nsplit * 8))) return rc;
    if (fe->top2_cap < rows) {
        hipFree(fe->d_idx2);
        hipFree(fe->d_dist2);
        fe->d_idx2 = fe->d_dist2 = nullptr;
        fe->top2_cap = 0;
        HIPCHK(hipMalloc((void**)&fe->d_idx2, rows * 8));
        HIPCHK(hipMalloc((void**)&fe->d_dist2, rows * 8));
        fe->top2_cap = rows;
    }
    if ((rc = vslam_ensure_pinned(&fe->h_top2, &fe->h_top2_bytes, rows * 16))) return rc;
    /* a split that lies past a problem's last tile, or a problem without train descriptors, writes nothing: "missing" */
    HIPCHK(hipMemsetAsync(fe->d_part, 0xFF, rows * nsplit * 8, fe->stream));
    vk_hamming_top2_batch(fe->stream, J, nprob, max_nq, (int)rows, nsplit, fe->d_part, fe->d_idx2, fe->d_dist2);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(fe->h_top2, fe->d_idx2, rows * 8, hipMemcpyDeviceToHost, fe->stream));
    HIPCHK(hipMemcpyAsync(fe->h_top2 + rows * 8, fe->d_dist2, rows * 8, hipMemcpyDeviceToHost, fe->stream));
    HIPCHK(vslam_stream_wait(fe->stream));
    for (int p = 0; p < nprob; p++) {
        if (!nq[p]) continue;
        memcpy(idx2[p], fe->h_top2 + (size_t)J.job[p].row0 * 8, (size_t)nq[p] * 8);
        memcpy(dist2[p], fe->h_top2 + rows * 8 + (size_t)J.job[p].row0 * 8, (size_t)nq[p] * 8);
    }
    return VSLAM_OK;
}

/* enqueue-only form for profilers and pipelines: results stay in the context's device arrays (rows back to back) */
extern "C" int vslam_hamming_top2_batch_dev_async(vslam_fe* fe, int nprob, const uint8_t* const* dev_q, const int32_t* nq,
                                                  const uint8_t* const* dev_t, const int32_t* nt) {
    if (!fe || nprob <= 0 || nprob > VSLAM_MAX_TOP2_JOBS || !dev_q || !nq || !dev_t || !nt) {
        g_err = "invalid arguments";
        return VSLAM_ERR_INVALID;
    }
    Top2Jobs J;
    memset(&J, 0, sizeof(J));
    size_t rows = 0;
    int max_nq = 0, max_nt = 0;
    for (int p = 0; p < nprob; p++) {
        if (nq[p] < 0 || nt[p] < 0 || nt[p] > 65535 || (nq[p] && !dev_q[p]) || (nt[p] && !dev_t[p])) {
            g_err = "invalid arguments";
            return VSLAM_ERR_INVALID;
        }
        J.job[p].q = (const uint32_t*)dev_q[p];
        J.job[p].t = (const uint32_t*)dev_t[p];
        J.job[p].nq = nq[p];
        J.job[p].nt = nt[p];
        J.job[p].row0 = (uint32_t)rows;
        rows += (size_t)nq[p];
        max_nq = std::max(max_nq, nq[p]);
        max_nt = std::max(max_nt, nt[p]);
    }
    if (rows == 0) return VSLAM_OK;
    HIPCHK(hipSetDevice(fe->p.device));
    const int nsplit = vk_hamming_top2_batch_split(nprob, max_nq, max_nt);
    if (fe->part_bytes < rows * nsplit * 8 || fe->top2_cap < rows) { /* sized by a previous vslam_hamming_top2_batch */
        g_err = "vslam_hamming_top2_batch_dev_async: call vslam_hamming_top2_batch with these sizes first (it allocates)";
        return VSLAM_ERR_INVALID;
    }
    HIPCHK(hipMemsetAsync(fe->d_part, 0xFF, rows * nsplit * 8, fe->stream));
    vk_hamming_top2_batch(fe->stream, J, nprob, max_nq, (int)rows, nsplit, fe->d_part, fe->d_idx2, fe->d_dist2);
    HIPCHK(hipGetLastError());
    return VSLAM_OK;
}

/* Frame::ComputeStereoFishEyeMatches (frame.cpp:1149-1174), the descriptor half: BFmatcher.knnMatch(left rows
 * [mono_left, n_left), right rows [mono_right, n_right), k = 2) + Lowe's ratio test.  What follows in the reference --
 * KannalaBrandt8::TriangulateMatches per surviving pair (:1176-1188) -- is the camera model's and stays with the caller. */
extern "C" int vslam_stereo_fisheye_candidates(vslam_fe* fe, const uint8_t* dev_desc_left, int n_left, int mono_left,
                                               const uint8_t* dev_desc_right, int n_right, int mono_right,
                                               int32_t* left_to_right, int32_t* best_dist, int32_t* second_dist,
                                               int* n_candidates) {
    if (!fe || !left_to_right || n_left < 0 || n_right < 0 || mono_left < 0 || mono_right < 0 || mono_left > n_left ||
        mono_right > n_right || (n_left && !dev_desc_left) || (n_right && !dev_desc_right)) {
        g_err = "invalid arguments";
        return VSLAM_ERR_INVALID;
    }
    for (int i = 0; i < n_left; i++) {
        left_to_right[i] = -1;
        if (best_dist) best_dist[i] = -1;
        if (second_dist) second_dist[i] = -1;
    }
    if (n_candidates) *n_candidates = 0;
    const int nq = n_left - mono_left, nt = n_right - mono_right;
    if (nq == 0) return VSLAM_OK;
    std::vector<int32_t> idx2((size_t)nq * 2), dist2((size_t)nq * 2);
    int rc = vslam_hamming_top2(fe, dev_desc_left + (size_t)mono_left * 32, nq, dev_desc_right + (size_t)mono_right * 32, nt,
                                idx2.data(), dist2.data());
    if (rc != VSLAM_OK) return rc;
    int nc = 0;
    for (int q = 0; q < nq; q++) {
        if (idx2[2 * q] < 0 || idx2[2 * q + 1] < 0) continue; /* (*it).size() >= 2 */
        /* DMatch::distance is a float; `distance * 0.7` is float x double: the comparison happens in double */
        const float d0 = (float)dist2[2 * q], d1 = (float)dist2[2 * q + 1];
        if (!((double)d0 < (double)d1 * 0.7)) continue;
        left_to_right[q + mono_left] = idx2[2 * q] + mono_right;
        if (best_dist) best_dist[q + mono_left] = dist2[2 * q];
        if (second_dist) second_dist[q + mono_left] = dist2[2 * q + 1];
        nc++;
    }
    if (n_candidates) *n_candidates = nc; /* the reference's descMatches */
    return VSLAM_OK;
}

extern "C" int vslam_hamming_matrix(vslam_fe* fe, const uint8_t* dev_q, int nq, const uint8_t* dev_t, int nt,
                                    uint8_t* out) {
    if (!fe || nq < 0 || nt < 0 || (nq && nt && (!dev_q || !dev_t || !out))) {
        g_err = "invalid arguments";
        return VSLAM_ERR_INVALID;
    }
    if (nq == 0 || nt == 0) return VSLAM_OK;
    HIPCHK(hipSetDevice(fe->p.device));
    int rc;
    if ((rc = vslam_ensure((void**)&fe->d_dmat, &fe->dmat_bytes, (size_t)nq * nt))) return rc;
    vk_hamming_matrix(fe->stream, dev_q, nq, dev_t, nt, fe->d_dmat);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, fe->d_dmat, (size_t)nq * nt, hipMemcpyDeviceToHost, fe->stream));
    HIPCHK(vslam_stream_wait(fe->stream));
    return VSLAM_OK;
}

