// Batch-1 latency through the C ABI from a C++ caller (no Python in the loop): vslam_fe_extract = FExtractor::compute,
// pageable host image in -> keypoints + descriptors in the caller's arrays.  Median of 7 windows of 100 calls.
//   g++ -O2 -I include tools/latency_c.cpp -o /tmp/latency_c -L vi_slam_amd -lvslam_fe -Wl,-rpath,$PWD/vi_slam_amd -Wl,-rpath,/opt/rocm/lib
//   /tmp/latency_c frame.raw 1241 376 1000
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "vslam_fe.h"

int main(int argc, char** argv) {
    if (argc < 5) return 2;
    const int W = atoi(argv[2]), H = atoi(argv[3]), NF = atoi(argv[4]);
    std::vector<uint8_t> img((size_t)W * H);
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(img.data(), 1, img.size(), f) != img.size()) return 3;
    fclose(f);
    vslam_fe_params p = {};
    p.width = W; p.height = H; p.nfeatures = NF; p.scale_factor = 1.2f; p.nlevels = 8; p.ini_th_fast = 20; p.min_th_fast = 7;
    p.device = 0; p.max_batch = 1;
    vslam_fe* fe = nullptr;
    if (vslam_fe_create(&p, &fe) != VSLAM_OK) { fprintf(stderr, "%s\n", vslam_last_error()); return 4; }
    const int cap = vslam_fe_capacity(fe);
    std::vector<vslam_kp> kps(cap);
    std::vector<uint8_t> desc((size_t)cap * 32);
    int n = 0, mono = 0;
    for (int i = 0; i < 30; i++)
        if (vslam_fe_extract(fe, img.data(), W, 0, 1000, kps.data(), desc.data(), cap, &n, &mono) != VSLAM_OK) return 5;
    std::vector<double> ms;
    for (int w = 0; w < 7; w++) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 100; i++) vslam_fe_extract(fe, img.data(), W, 0, 1000, kps.data(), desc.data(), cap, &n, &mono);
        ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / 100);
    }
    std::sort(ms.begin(), ms.end());
    printf("{\"c_abi_vslam_fe_extract_pageable_n%d_ms_median_min_max\": [%.4f, %.4f, %.4f], \"keypoints\": %d}\n", NF, ms[3], ms[0], ms[6], n);
    vslam_fe_destroy(fe);
    return 0;
}
