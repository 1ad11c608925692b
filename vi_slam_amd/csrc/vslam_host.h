/* vslam_host.h -- host-side logic of the front-end that involves no GPU call: constructor tables,
 * resize coefficient tables, the FAST cell list, the sequential quadtree distribution and the
 * order-dependent matcher replays.  Plain C++17; built into libvslam_fe.so (with the kernels) and into
 * libvslam_host.so (g++, no HIP) so the CPU test-suite can exercise it without a GPU.
 */
#ifndef VSLAM_HOST_H
#define VSLAM_HOST_H

#include <cstdint>
#include <vector>

#include "../../include/vslam_fe.h"

/* EDGE_THRESHOLD-3: the FAST cell grid starts 16 px inside the level (fextractor.cpp:764-767) */
#define VSLAM_FAST_BORDER 16

namespace vslam {

/* FExtractor constructor tables (fextractor.cpp:401-461). */
struct ExtractorTables {
    int nlevels = 0;
    std::vector<float> scale, inv_scale, sigma2, inv_sigma2;
    std::vector<int> quota; /* mnFeaturesPerLevel */
    int umax[16];
    std::vector<int8_t> disc_u, disc_v; /* flattened radius-15 disc of IC_Angle (fextractor.cpp:75-92) */
};
void build_tables(int nfeatures, float scaleFactor, int nlevels, ExtractorTables& t);

/* level sizes (fextractor.cpp:1139-1140) */
void level_size(const ExtractorTables& t, int w, int h, int level, int* lw, int* lh);

/* cv::resize INTER_LINEAR 8u coefficient tables for one (src -> dst) level pair, in the layout
 * k_resize_level reads: xtab = {sx, min(sx+1, sw-1)} per dx, xa = {alpha0, alpha1}; ytab = clipped
 * {sy, sy+1}, yb = {beta0, beta1}. */
struct ResizeTables {
    std::vector<uint16_t> xtab, ytab;
    std::vector<int16_t> xa, yb;
};
void build_resize_tables(int sw, int sh, int dw, int dh, ResizeTables& r);
/* Quad form of the column table for k_resize_level_v2 (see ResizeQuad in vslam_device.h): qbase[q], and per
 * quad 4 selectors + 4 packed coefficient pairs (8 uint32).  Returns false when some quad's taps do not fit an
 * 8-byte window (scale factor > 2) or the source is narrower than 8 px -- the caller keeps the v1 kernel. */
bool build_resize_quads(const ResizeTables& r, int sw, int dw, std::vector<uint16_t>& qbase,
                        std::vector<uint32_t>& quads);

/* ---- fused pyramid (k_pyramid_group): one workgroup computes a spatial tile of SEVERAL consecutive levels in LDS.
 * Level l is resized from level l-1 (a cascade, fextractor.cpp:1148), so a tile of level l needs a slightly larger
 * tile of level l-1: every (tile, level) gets a COMPUTE range (own range + what the deeper levels of the same tile
 * need, recomputed instead of exchanged between workgroups) and a STORE range (a partition of the level: every
 * pixel of a level is stored by exactly one tile).  Columns are counted in quads (4 output pixels, the unit of
 * k_resize_level_v2's column table). */
struct PyrTileLevel {     /* entry j of a tile: j = 0 is the group's source level (staged from HBM), j >= 1 computed */
    int16_t c0, nc;       /* first column (multiple of 4) and number of columns (multiple of 4) held in LDS */
    int16_t r0, nr;       /* first row, number of rows held in LDS */
    int16_t sq0, sq1;     /* quads stored to HBM: [sq0, sq1) */
    int16_t sr0, sr1;     /* rows stored to HBM: [sr0, sr1) */
    uint32_t lds_off;     /* byte offset of this level's tile in the workgroup's LDS (multiple of 16) */
    uint32_t pitch;       /* LDS row pitch in bytes: nc + 8 (the 8-byte tap window of the last quad may overrun nc)
                             rounded up to a multiple of 16 (rows are staged with 16-byte stores) */
    uint32_t rt_off;      /* j >= 1: byte offset of the level's row table in LDS (nr entries of 8 bytes) */
};
struct PyrGroupPlan {
    int l0 = 0, nl = 0;        /* source level, number of computed levels (l0+1 .. l0+nl) */
    int ntx = 0, nty = 0;      /* tiles per image */
    size_t lds_bytes = 0;      /* dynamic LDS per workgroup */
    std::vector<PyrTileLevel> tiles; /* [ntx*nty][nl+1] */
};
struct PyrLevelTables {   /* tables of destination level l (source l-1), as the per-level kernels use them */
    int sw, sh, dw, dh;
    ResizeTables r;
    std::vector<uint16_t> qbase;
    std::vector<uint32_t> quads;
};
/* Plan for levels l0+1 .. l0+nl (tabs[j-1] = tables of level l0+j).  Returns false if a tile would need more than 64
 * quads per row (lane = quad) or more LDS than max_lds -- the caller then keeps the per-level launches. */
bool build_pyramid_group(const std::vector<const PyrLevelTables*>& tabs, int l0, size_t max_lds, PyrGroupPlan& plan,
                         int rows_override = -1 /* vslam_tuning.pyr_rows */);
/* CPU emulation of k_pyramid_group with the SAME plan and indexing (LDS tiles as arrays); used by the CPU tests to
 * validate a plan before it ever runs on the GPU.  levels[j] = image of level l0+j (j = 0 input, j >= 1 output,
 * pitch = stride[j]).  Returns 0, or a negative code if an access leaves a tile (the GPU would read garbage). */
int emulate_pyramid_group(const PyrGroupPlan& plan, const std::vector<const PyrLevelTables*>& tabs,
                          const uint8_t* src, int sstride, int readable_w, std::vector<uint8_t*>& dst,
                          const std::vector<int>& dstride);

/* FAST cell grid of one level (fextractor.cpp:764-797).  Cells are listed in the reference's visiting
 * order (row-major, skipped cells omitted). */
struct HostCell {
    uint16_t level, x0, y0, x1, y1;
};
void build_cells(int level, int lw, int lh, std::vector<HostCell>& out);

/* Bands of k_fast_bands: up to max_cells CONSECUTIVE cells of one cell row (same level, same y0) whose interiors together
 * are at most max_width px wide; the cells of a row are visited left to right with a constant pitch `wcell` (the last
 * one may be clipped, fextractor.cpp:791-795), so cell k of a band has the interior columns [k * wcell, min((k + 1) * wcell,
 * ww - 6)) of the band's interior.  Bands are listed in cell order (cell0 ascending) and cover every cell exactly once. */
struct HostBand {
    uint32_t cell0;                  /* index of the band's first cell in the cell list */
    uint16_t level, ncell, wcell;
    uint16_t x0, y0, ww, wh;         /* shared window: [x0, x0 + ww) x [y0, y0 + wh) in level coordinates */
};
void build_bands(const std::vector<HostCell>& cells, int max_cells, int max_width, std::vector<HostBand>& out);

/* A FAST candidate / selected keypoint in level coordinates relative to the 16-px border. */
struct Cand {
    int16_t x, y;
    uint8_t response;
};

/* FExtractor::DistributeOctTree (fextractor.cpp:530-754) on integer candidates.
 * W = maxBorderX - minBorderX, H = maxBorderY - minBorderY, N = mnFeaturesPerLevel[level].
 * Result order = the reference's lNodes order (front to back).  Tie-break of equal-size nodes in the
 * "largest first" phase: the node created later is split first (the reference sorts by heap address,
 * which is not reproducible).  Returns false when nIni < 1 (reference UB). */
bool distribute_octree(const Cand* cands, int n, int W, int H, int N, std::vector<Cand>& out);

/* The implicit quadtree under DistributeOctTree's initial nodes (fextractor.cpp:534-561 for the roots, DivideNode
 * :472-517 for the halving): a key's root b = min((int)(x / hX), nIni - 1) and its quadrant at every depth depend on x and y
 * SEPARATELY (midpoints mx = x0 + ceil((x1 - x0) / 2), kp.x < mx; the same in y), so the path to depth D is
 *     path = xs[x] | ys[y],   xs[x] = b << 2D | (x's D decisions on the even bits), ys[y] = (y's on the odd bits),
 * most significant decision first -- two table look-ups instead of D dependent halvings per key (k_octree_v4).
 * W = maxBorderX - minBorderX, H = maxBorderY - minBorderY; xs gets W + 1 entries, ys H + 1.
 * oct_key_path: the same path by walking the halvings, any depth (the kernel uses it below the tables' depth). */
void build_oct_lut(int W, int H, int D, std::vector<uint32_t>& xs, std::vector<uint32_t>& ys);
uint32_t oct_key_path(int x, int y, int W, int H, int depth);

/* FMatcher::ComputeThreeMaxima (fmatcher.cpp:2813-2854) on bin sizes. */
void compute_three_maxima(const int* histo_sizes, int L, int& ind1, int& ind2, int& ind3);

/* Frame::AssignFeaturesToGrid + GetFeaturesInArea (frame.cpp:386-414, 678-756), undistorted image. */
struct FrameGrid {
    static const int COLS = 64, ROWS = 48; /* frame.h:42-43 */
    float minX, maxX, minY, maxY, invW, invH;
    std::vector<int> cell_start; /* COLS*ROWS+1, cells indexed ix*ROWS+iy */
    std::vector<int> cell_items;
    const vslam_kp* kps;
    int n;
    void build(const vslam_kp* k, int n_, int imgW, int imgH);
    void query(float x, float y, float r, int minLevel, int maxLevel, std::vector<int>& out) const;
};

/* FMatcher::SearchForInitialization (fmatcher.cpp:983-1098) sequential replay.  dist(i1,i2) is read
 * from a dense matrix over the octave-0 keypoints of both frames: row r of `dmat` belongs to l0_1[r],
 * column c to l0_2[c] (map2[i2] = column or -1). */
int search_for_initialization_replay(const vslam_kp* kps1, int n1, const vslam_kp* kps2, int n2,
                                     const uint8_t* dmat, const int* row_of_i1, const int* col_of_i2,
                                     int ncols, int imgW, int imgH, float* prevMatched, int32_t* matches12,
                                     int windowSize, float nnratio, bool checkOri);

} // namespace vslam
#endif
