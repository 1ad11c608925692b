/* vslam_tuning.h -- the ONE place where behaviour switches are resolved (include/vslam_fe.h: vslam_tuning).
 * vslam_process_tuning(): the environment's values, read once per process under std::call_once, table-driven (the only
 * getenv of the library).  vslam_resolve_tuning(user): caller's field, else the process default, else -1; the use site
 * applies its built-in default to -1 (the comments explaining a default stay next to the code that uses it). */
#ifndef VSLAM_TUNING_H
#define VSLAM_TUNING_H
#include "../../include/vslam_fe.h"

const vslam_tuning& vslam_process_tuning();
vslam_tuning vslam_resolve_tuning(const vslam_tuning* user);
void vslam_apply_tuning(vslam_tuning& t, const vslam_tuning* user); /* fields of *user that are >= 0 overwrite t */
/* v if the caller or the environment set it (>= 0), else the built-in default */
static inline int tune_or(int v, int dflt) { return v >= 0 ? v : dflt; }
/* vslam_tuning.wave_prio is a bit mask of kernel classes (1 quadtree + output order, 2 descriptors, 4 matchers, 8 packing) */
static inline int wave_prio_on(const vslam_tuning& t, int cls) { return t.wave_prio > 0 && (t.wave_prio & cls) ? 1 : 0; }
#endif
