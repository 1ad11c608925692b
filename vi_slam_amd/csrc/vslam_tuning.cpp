/* vslam_tuning.cpp -- process defaults of the behaviour switches (vslam_tuning.h): the library's only getenv. */
#include "vslam_tuning.h"

#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace {
struct EnvField {
    const char* env;
    size_t off;
    const char* word1; /* VALUE that means 1 (e.g. VSLAM_PYRAMID=levels), or nullptr: numeric only */
    const char* word2; /* VALUE that means 2 */
};
#define TF(name, field, w1, w2) {name, offsetof(vslam_tuning, field), w1, w2}
const EnvField kEnv[] = {
    TF("VSLAM_PYRAMID", pyramid_per_level, "levels", nullptr),
    TF("VSLAM_PYR_ROWS", pyr_rows, nullptr, nullptr),
    TF("VSLAM_PYR_NT", pyr_threads, nullptr, nullptr),
    TF("VSLAM_BLUR_ROWS", blur_rows, nullptr, nullptr),
    TF("VSLAM_FAST_NT", fast_threads, nullptr, nullptr),
    TF("VSLAM_FAST_PITCH", fast_pitch, nullptr, nullptr),
    TF("VSLAM_FAST_LDS_PAD", fast_lds_pad, nullptr, nullptr),
    TF("VSLAM_OCTREE", octree_walk_kernel, "v2", nullptr),
    TF("VSLAM_OCT_FINE_D", oct_fine_depth, nullptr, nullptr),
    TF("VSLAM_OCT_LDS_BUDGET_KB", oct_lds_budget_kb, nullptr, nullptr),
    TF("VSLAM_OCT_REGKEYS", oct_regkeys, nullptr, nullptr),
    TF("VSLAM_OCT_MAXITER", oct_max_iter, nullptr, nullptr),
    TF("VSLAM_OCT_DBG", oct_debug, nullptr, nullptr),
    TF("VSLAM_GRAPH", graphs, nullptr, nullptr),
    TF("VSLAM_H2D", h2d_route, "pull", "sdma"),
    TF("VSLAM_D2H", d2h_route, "kernel", "sdma"),
    TF("VSLAM_COPY_WGS", copy_wgs, nullptr, nullptr),
    TF("VSLAM_PULL_DEPTH", pull_depth, nullptr, nullptr),
    TF("VSLAM_INIT_TOPM", init_topm, nullptr, nullptr),
    TF("VSLAM_INIT_MATCH", init_match_host, "host", nullptr),
    TF("VSLAM_SBP_TOPM", sbp_topm, nullptr, nullptr),
    TF("VSLAM_SBP_MODE", sbp_sequential, "seq", nullptr),
    TF("VSLAM_SI_QPB", si_queries_per_block, nullptr, nullptr),
    TF("VSLAM_FG_NT", fg_threads, nullptr, nullptr),
    TF("VSLAM_WAIT", wait_spin, "spin", nullptr),
    TF("VSLAM_NUMA", numa, nullptr, nullptr),
    TF("VSLAM_HOST_PROF", host_prof, nullptr, nullptr),
    TF("VSLAM_STREAM_PRIORITY", stream_priority, nullptr, nullptr),
    TF("VSLAM_STAGE_SPLIT_EVENT", stage_split_event, nullptr, nullptr),
    TF("VSLAM_OCT_THREADS", oct_threads, nullptr, nullptr),
    TF("VSLAM_FAST_KERNEL", fast_kernel, nullptr, nullptr),
    TF("VSLAM_FAST_BAND_CELLS", fast_band_cells, nullptr, nullptr),
    TF("VSLAM_WAVE_PRIO", wave_prio, nullptr, nullptr),
    TF("VSLAM_OCT_PRECOUNT", oct_precount, nullptr, nullptr),
    TF("VSLAM_DESC_KPW", desc_kpw, nullptr, nullptr),
};
#undef TF
vslam_tuning g_process;
std::once_flag g_once;

int32_t& field(vslam_tuning& t, size_t off) { return *reinterpret_cast<int32_t*>(reinterpret_cast<char*>(&t) + off); }
}  // namespace

extern "C" void vslam_tuning_init(vslam_tuning* t) {
    if (!t) return;
    int32_t* w = reinterpret_cast<int32_t*>(t);
    for (size_t i = 0; i < sizeof(vslam_tuning) / sizeof(int32_t); i++) w[i] = -1;
}

const vslam_tuning& vslam_process_tuning() {
    std::call_once(g_once, [] {
        vslam_tuning_init(&g_process);
        for (const EnvField& e : kEnv) {
            const char* v = getenv(e.env);
            if (!v || !*v) continue;
            int32_t x;
            if (e.word1 && !strcmp(v, e.word1)) x = 1;
            else if (e.word2 && !strcmp(v, e.word2)) x = 2;
            else if ((*v >= '0' && *v <= '9') || *v == '-') x = (int32_t)atoi(v);
            else continue; /* an unknown word: as if unset */
            if (x >= 0) field(g_process, e.off) = x;
        }
    });
    return g_process;
}

void vslam_apply_tuning(vslam_tuning& t, const vslam_tuning* user) {
    if (!user) return;
    for (const EnvField& e : kEnv) {
        const int32_t u = *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(user) + e.off);
        if (u >= 0) field(t, e.off) = u;
    }
}

vslam_tuning vslam_resolve_tuning(const vslam_tuning* user) {
    vslam_tuning t = vslam_process_tuning();
    vslam_apply_tuning(t, user);
    return t;
}
