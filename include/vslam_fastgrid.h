/* vslam_fastgrid.h -- C ABI of the grid FAST detector behind vi_slam::geometry::FAST::detect.
 *
 * Replaces, on MI355X (gfx950), what the reference reaches through
 *   vi_slam::geometry::FAST::detect(const cv::Mat&, vector<cv::KeyPoint>&)   include/vi_slam/geometry/fast_cuda.h:19-25,
 *                                                                            src/geometry/fast_cuda.cpp:70-132
 * namely vilib::FASTGPU (thirdparty/vilib/visual_lib/include/vilib/feature_detection/fast/fast_gpu.h:44-61) over
 * vilib::DetectorBaseGPU (.../feature_detection/detector_base_gpu.h) on a vilib::Frame's half-sampled pyramid
 * (.../preprocess/pyramid_gpu.cu).  Same arguments, same feature-grid buffers, same results -- including which of
 * several equal maxima of a cell is reported, which the reference decides by the launch geometry of its CUDA kernel
 * (tie_rule 0 reproduces that; tie_rule 1 is the raster order of the reference's CPU detector rosten::FASTCPU<true>).
 * Error codes and vslam_last_error() are those of vslam_fe.h.
 */
#ifndef VSLAM_FASTGRID_H
#define VSLAM_FASTGRID_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* vilib::fast_score (fast_common.h:39-43) */
#define VSLAM_FG_SUM_OF_ABS_DIFF_ALL 0
#define VSLAM_FG_SUM_OF_ABS_DIFF_ON_ARC 1
#define VSLAM_FG_MAX_THRESHOLD 2

/* The constructor arguments of vilib::FASTGPU (fast_gpu.cpp:52-62) in their order, then ours. */
typedef struct vslam_fg_params {
    int32_t image_width, image_height;
    int32_t cell_size_width, cell_size_height; /* 32 or 64; the reference asserts width % 32 == 0 */
    int32_t min_level, max_level;              /* levels min_level <= l < max_level are searched; max_level <= 8 */
    int32_t horizontal_border, vertical_border; /* raised to 3 (Bresenham circle) as the reference does */
    float threshold;                           /* FAST_EPSILON */
    int32_t min_arc_length;                    /* 9..12 */
    int32_t score;                             /* VSLAM_FG_* */
    int32_t tie_rule;                          /* 0: FASTGPU (CUDA launch order), 1: raster (FASTCPU<true>) */
    int32_t device;
    int32_t max_batch;                         /* images per vslam_fg_detect_batch call, 1..64 */
} vslam_fg_params;

typedef struct vslam_fg vslam_fg;

int vslam_fg_create(const vslam_fg_params* p, vslam_fg** out);
void vslam_fg_destroy(vslam_fg* fg);
/* DetectorBase::getCellCountHorizontal / getCellCountVertical (detector_base.h:79-80) */
int vslam_fg_grid(const vslam_fg* fg, int* n_cols, int* n_rows);

/* vilib::Frame(image, 0, levels) + FASTGPU::detect(frame->pyramid_) + DetectorBaseGPU::copyGridToHost
 * (common/frame.cpp:49-57, fast_gpu.cpp:97-131, detector_base_gpu.cpp:118-125).  The three outputs are the
 * h_pos_ / h_score_ / h_level_ arrays of the reference's feature grid: n_cols*n_rows cells, row-major;
 * pos = (x, y) on level 0; a cell holds a corner iff score > 0 (processGrid's test, detector_base_gpu.cpp:207).
 * Cells without one carry pos (0, 0) and level -1 (the reference leaves stale values there and never reads them). */
int vslam_fg_detect(vslam_fg* fg, const uint8_t* img_host, size_t pitch, float* pos, float* score, int32_t* level);

/* n images of the same size in one pass; imgs[i] are host pointers, or device pointers when on_device != 0.
 * Outputs hold n consecutive grids. */
int vslam_fg_detect_batch(vslam_fg* fg, int n, const uint8_t* const* imgs, size_t pitch, int on_device, float* pos,
                          float* score, int32_t* level);

/* A pyramid level of image `slot` of the last call (vilib::Subframe: width >> l, height >> l). */
int vslam_fg_level_copy(vslam_fg* fg, int slot, int level, uint8_t* dst, size_t dst_pitch, int* w, int* h);
/* DetectorBaseGPU::copyResponseTo (detector_base_gpu.cpp:127-141): the corner response of a searched level,
 * (width >> l) * (height >> l) floats, recomputed for image `slot` of the last call. */
int vslam_fg_response_copy(vslam_fg* fg, int slot, int level, float* dst);

#ifdef __cplusplus
}
#endif
#endif
