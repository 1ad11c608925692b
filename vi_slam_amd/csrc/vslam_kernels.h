/* vslam_kernels.h -- launch wrappers implemented in the kernel files (vslam_image_kernels.hip, vslam_kernels.hip,
 * vslam_octree_kernel.hip, vslam_match_kernels.hip, vslam_init_kernel.hip). */
#ifndef VSLAM_KERNELS_H
#define VSLAM_KERNELS_H

#include "vslam_tuning.h"
#include "vslam_device.h"

void vk_upload_disc(const int umax[16]); /* IC_Angle disc -> packed dword weights of the descriptor kernel */

void vk_resize_level(hipStream_t st, uint8_t* pyr, size_t slot_stride, const BatchSrc& src,
                     const LevelGeom& sg, const LevelGeom& dg, int src_level, const uint16_t* xtab,
                     const int16_t* xa, const uint16_t* ytab, const int16_t* yb, int nslots);

void vk_orient_describe(hipStream_t st, const uint8_t* pyr, const uint8_t* blur, size_t slot_stride,
                        const BatchSrc& src, const PyramidGeom& g, const SelKp* sel, int nsel,
                        const int8_t* pattern, vslam_kp* kps, uint8_t* desc, int cap, int atan_fma);

void vk_hamming_matrix(hipStream_t st, const uint8_t* q, int nq, const uint8_t* t, int nt, uint8_t* out);
/* nprob problems in one launch: part holds nrows * nsplit * 2 words (nrows = queries of all problems back to back,
 * Top2Job::row0), idx2 / dist2 nrows * 2; nsplit from vk_hamming_top2_batch_split */
int vk_hamming_top2_batch_split(int nprob, int max_nq, int max_nt);
void vk_hamming_top2_batch(hipStream_t st, const Top2Jobs& jobs, int nprob, int max_nq, int nrows, int nsplit, uint32_t* part,
                           int32_t* idx2, int32_t* dist2);

void vk_blur7_v2(hipStream_t st, const uint8_t* pyr, size_t slot_stride, const BatchSrc& src, const PyramidGeom& g,
                 uint8_t* blur, const uint32_t* tasks, int ntasks, const int32_t taps[7], int rows_per_task,
                 int nslots);
/* T: the context's resolved switches (vslam_tuning.h); the launchers read their A/B knobs from it, never from the environment */
void vk_pyramid_group(hipStream_t st, uint8_t* pyr, size_t slot_stride, const BatchSrc& src, const PyrGroupDev& G,
                      size_t lds_bytes, int nslots, const vslam_tuning& T, uint8_t* reset_cand = nullptr, size_t cand_stride = 0,
                      int32_t* reset_err = nullptr);
void vk_fast_cells_v3(hipStream_t st, const uint8_t* pyr, size_t slot_stride, const BatchSrc& src,
                      const PyramidGeom& g, const CellDesc* cells, int ncells, uint8_t* cand_region,
                      size_t cand_stride, int iniTh, int minTh, int tile_rows, int max_window_w, int max_px, int nslots,
                      const vslam_tuning& T);
/* k_fast_bands: one workgroup per band of cells (vslam::build_bands); max_wh = tallest band window.  _check: the
 * kernel's limits (0 = a band list it can run) */
int vk_fast_bands_check(int max_wh, int max_iw, int max_cells_per_band);
void vk_fast_bands(hipStream_t st, const uint8_t* pyr, size_t slot_stride, const BatchSrc& src, const PyramidGeom& g,
                   const BandDesc* bands, int nbands, const uint8_t* classes, const CellDesc* cells, int ncells, uint8_t* cand_region,
                   size_t cand_stride, int iniTh, int minTh, int max_wh, int max_iw, int nslots, const vslam_tuning& T);
void vk_resize_level_v2(hipStream_t st, uint8_t* pyr, size_t slot_stride, const BatchSrc& src, const LevelGeom& sg,
                        const LevelGeom& dg, int src_level, const uint16_t* qbase, const ResizeQuad* quads,
                        const uint16_t* ytab, const int16_t* yb, int nslots);
size_t vk_octree_lds_bytes(int maxNodes);
int vk_octree_set_max_lds(size_t bytes);
void vk_octree(hipStream_t st, const uint8_t* cand_region, size_t cand_stride, int ncells, const OctParams& P,
               uint32_t* keys_a, uint32_t* aux_a, uint16_t* nid_a, void* sorted_a, size_t pts_stride,
               uint32_t* sel_xyr, int32_t* sel_cnt, int32_t* err_flag, int nlevels, int nslots, int32_t* deep_flags,
               int regkeys, int threads = 1024, int prio = 0);
/* k_oct_count: walk 1 of the quadtree as a launch of its own (OctParams::parts); LDS = counters of the largest level + the
 * cell offsets of the largest part */
size_t vk_oct_count_lds(int maxcells, int maxPartCells);
int vk_oct_count_set_max_lds(size_t bytes);
void vk_oct_count(hipStream_t st, const uint8_t* cand_region, size_t cand_stride, int ncells, const OctParams& P, uint32_t* keys_a,
                  uint32_t* aux_a, size_t pts_stride, int nlevels, int nslots, int maxcells);
void vk_assign_out(hipStream_t st, const OctParams& P, const PyramidGeom& g, uint32_t* sel_xyr, int32_t* sel_cnt, int lap0,
                   int lap1, SelKp* sel, int32_t* slot_counts, int cap, int32_t* err_flag, int nslots,
                   const int32_t* deep_flags, int prio = 0);
void vk_orient_describe_dev(hipStream_t st, const uint8_t* pyr, const uint8_t* blur, size_t slot_stride,
                            const BatchSrc& src, const PyramidGeom& g, const SelKp* sel,
                            const int32_t* slot_counts, const int8_t* pattern, vslam_kp* kps, uint8_t* desc,
                            int cap, int atan_fma, int nslots, int prio = 0, int kpw_override = -1);

void vk_dbg_sincos(hipStream_t st, const float* x, int n, float* s, float* c);
void vk_dbg_logf(hipStream_t st, const float* x, int n, float* y);
void vk_dbg_atan2(hipStream_t st, const float* y, const float* x, int n, int fma, float* a);

void vk_stereo(hipStream_t st, const StereoJobs& jobs, int njobs, int maxNL, int maxNR, const PyramidGeom& g,
               const uint8_t* pyrL, size_t strideL, const BatchSrc& srcL, const uint8_t* pyrR, size_t strideR,
               const BatchSrc& srcR, float mbf, float maxD, uint32_t* best, float* uRight, float* depth,
               int32_t* sad, int cap, int max_band, uint8_t* rows_scratch, int prio = 0);
/* bytes of rows_scratch: row table, bucket items and {uR, octave} records of njobs stereo pairs */
size_t vk_stereo_rows_bytes(int njobs, int nrows, int max_band, int cap);
void vk_hamming_matrix_batch(hipStream_t st, const MatJobs& jobs, int njobs, int maxr, int maxc, const int32_t* idx,
                             uint8_t* tmp, uint8_t* out);
size_t vk_search_init_lds(int cap, int max_c2);
size_t vk_search_init_scratch_bytes(int npairs, int max_c2, int M);
int vk_search_init_set_max_lds(size_t bytes);
/* k_si_topm + k_si_replay; scratch = vk_search_init_scratch_bytes(); fallbacks (nullable) counts full re-scans */
void vk_search_init(hipStream_t st, const InitJobs& jobs, int npairs, int cap, int imgW, int imgH, int window,
                    float nnratio, int checkOri, int32_t* matches_out, float* prev_out, int32_t* nmatch_out,
                    int max_c2, int M, uint8_t* scratch, int* fallbacks, const vslam_tuning& T);
/* FMatcher::SearchByProjection(CurrentFrame, LastFrame): k_sbp_rank + k_sbp_replay over up to
 * VSLAM_MAX_SBP_JOBS problems; maxLast / maxCur size the grid and the LDS (capacities) */
size_t vk_sbp_rank_lds(int nCur);
size_t vk_sbp_replay_lds(int nCur, int nLast);
size_t vk_sbp_scratch_bytes(int nLast, int M);
size_t vk_sbp_proj_bytes(int nLast);
int vk_sbp_set_max_lds(size_t bytes);
void vk_search_by_projection(hipStream_t st, const SbpJobs& JS, int njobs, int maxLast, int maxCur, int* fallbacks,
                             int forceSeq);
int vk_distinctive_set_max_lds(size_t bytes);
void vk_distinctive(hipStream_t st, const uint8_t* desc, const int32_t* offsets, int nsets, int maxN, int32_t* best);
void vk_unproject_stereo(hipStream_t st, const UnprojJobs& U, int njobs);
void vk_fuse_search(hipStream_t st, const FuseArgsDev& A);
void vk_reset_headers(hipStream_t st, uint8_t* d_cand, size_t cand_stride_bytes, int nimg, int32_t* d_err);
/* device -> pinned host (or device) range copies / zero fills in one launch; see k_copy_ranges */
/* returns the number of copy operations put on the stream (runtime copies or one kernel launch) */
int vk_copy_ranges(hipStream_t st, const CopyRanges& R, const vslam_tuning& T = vslam_process_tuning());
/* host (pinned) images -> level 0 of the slots, one launch; src.l0 / src.pitch0 describe the host rows */
void vk_pull_images(hipStream_t st, const BatchSrc& src, uint8_t* pyr, size_t slot_stride, uint32_t off0, int dpitch,
                    int w, int h, int nimg, int from_host, const vslam_tuning& T);
void vk_pack_slots(hipStream_t st, const vslam_kp* kps, const uint8_t* desc, const int32_t* counts, int cap, int first,
                   int nslots, uint8_t* dst, size_t slot_bytes, int prio = 0);
void vk_gather_rows32(hipStream_t st, const uint8_t* src, const int32_t* idx, int n, uint8_t* dst);

#endif
