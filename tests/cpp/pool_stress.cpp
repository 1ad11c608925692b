// pool_stress.cpp -- ThreadSanitizer stress of the context's worker pool (vi_slam_amd/csrc/vslam_pool.h).
// Alternates small and large parallel_for calls whose lambdas live on the caller's stack: a worker that is
// late leaving job k must never claim an index of job k+1, no index may run twice, and parallel_for must not
// return while a task still runs.  Built and run by tests/test_host_logic.py with -fsanitize=thread.
#include <cstdio>
#include <vector>

#include "../../vi_slam_amd/csrc/vslam_pool.h"

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 4000;
    WorkerPool pool(7);
    long bad = 0;
    for (int r = 0; r < rounds; r++) {
        const int n = (r & 1) ? 2 + (r % 5) : 200 + (r % 57); // small, then large (stage_host_images -> nimg*L tasks)
        std::vector<int> hits(n, 0); // on this stack frame: a stale task would write into a dead vector
        pool.parallel_for(n, [&](int i) { hits[i]++; });
        for (int i = 0; i < n; i++) bad += hits[i] != 1;
    }
    if (bad) {
        printf("FAIL: %ld indices ran zero or several times\n", bad);
        return 1;
    }
    printf("pool_stress ok: %d rounds\n", rounds);
    return 0;
}
