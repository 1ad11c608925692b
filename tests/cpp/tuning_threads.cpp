// Two threads resolve and apply switches concurrently, as two Frame-constructor threads creating two extractor contexts
// do (frame.cpp:107-108): the process defaults are read from the environment exactly once (std::call_once), every
// context gets its own resolved copy.  Built with -fsanitize=thread by tests/test_tuning.py; exit code 0 and no
// ThreadSanitizer report = pass.  Prints the resolved values for the test to compare.
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "../../vi_slam_amd/csrc/vslam_tuning.h"

int main() {
    setenv("VSLAM_OCT_FINE_D", "3", 1);
    setenv("VSLAM_H2D", "sdma", 1);
    setenv("VSLAM_PYRAMID", "levels", 1);
    setenv("VSLAM_D2H", "bogus", 1); /* unknown word: as if unset */
    std::vector<vslam_tuning> got(8);
    std::vector<std::thread> th;
    for (int i = 0; i < 8; i++)
        th.emplace_back([&got, i] {
            vslam_tuning u;
            vslam_tuning_init(&u);
            u.init_topm = i; /* per-context override */
            if (i & 1) u.oct_fine_depth = 5;
            for (int r = 0; r < 1000; r++) got[i] = vslam_resolve_tuning(&u);
        });
    for (auto& t : th) t.join();
    setenv("VSLAM_OCT_FINE_D", "9", 1); /* too late: already resolved once */
    const vslam_tuning late = vslam_resolve_tuning(nullptr);
    for (int i = 0; i < 8; i++)
        printf("ctx %d: oct_fine_depth %d h2d_route %d pyramid_per_level %d d2h_route %d init_topm %d fast_threads %d\n", i,
               got[i].oct_fine_depth, got[i].h2d_route, got[i].pyramid_per_level, got[i].d2h_route, got[i].init_topm, got[i].fast_threads);
    printf("late: oct_fine_depth %d\n", late.oct_fine_depth);
    return 0;
}
