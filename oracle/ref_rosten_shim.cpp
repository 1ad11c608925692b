/* oracle/ref_rosten_shim.cpp -- TEST INFRASTRUCTURE ONLY.
 * C entry point over the REFERENCE's Rosten FAST (compiled from /root/reference by oracle/Makefile into
 * oracle/_ref/libref_rosten.so).  fast9_detect_nonmax<true> == cv::FAST(TYPE_9_16, nms=true) on the same
 * (cell) image: detect on rows/cols [3,n-3), score = max threshold, suppress if any 8-neighbour >=.
 * Interface: thirdparty/vilib/visual_lib/src/feature_detection/fast/rosten/fast.h:10-39. */
#include <cstdlib>

#include "fast.h"

extern "C" int ref_fast9_detect_nonmax(const unsigned char* im, int xsize, int ysize, int stride, int b,
                                       int* xys_out, int cap) {
    int n = 0;
    vilib::rosten::xys* r = vilib::rosten::fast9_detect_nonmax<true>(im, xsize, ysize, stride, b, &n);
    for (int i = 0; i < n && i < cap; i++) {
        xys_out[3 * i] = r[i].x;
        xys_out[3 * i + 1] = r[i].y;
        xys_out[3 * i + 2] = r[i].s;
    }
    if (r) free(r);
    return n;
}

/* rosten::fastN_detect_nonmax<use_new_score> for N = 9..12 -- the functions rosten::FASTCPU dispatches to
 * (fast_cpu.cpp:68-92): new_score = 0 -> SUM_OF_ABS_DIFF_ON_ARC, 1 -> MAX_THRESHOLD. */
extern "C" int ref_fast_detect_nonmax(const unsigned char* im, int xsize, int ysize, int stride, int b, int arc,
                                      int new_score, int* xys_out, int cap) {
    using namespace vilib::rosten;
    int n = 0;
    xys* r = nullptr;
    switch (arc * 2 + (new_score ? 1 : 0)) {
        case 18: r = fast9_detect_nonmax<false>(im, xsize, ysize, stride, b, &n); break;
        case 19: r = fast9_detect_nonmax<true>(im, xsize, ysize, stride, b, &n); break;
        case 20: r = fast10_detect_nonmax<false>(im, xsize, ysize, stride, b, &n); break;
        case 21: r = fast10_detect_nonmax<true>(im, xsize, ysize, stride, b, &n); break;
        case 22: r = fast11_detect_nonmax<false>(im, xsize, ysize, stride, b, &n); break;
        case 23: r = fast11_detect_nonmax<true>(im, xsize, ysize, stride, b, &n); break;
        case 24: r = fast12_detect_nonmax<false>(im, xsize, ysize, stride, b, &n); break;
        case 25: r = fast12_detect_nonmax<true>(im, xsize, ysize, stride, b, &n); break;
        default: return -1;
    }
    for (int i = 0; i < n && i < cap; i++) {
        xys_out[3 * i] = r[i].x;
        xys_out[3 * i + 1] = r[i].y;
        xys_out[3 * i + 2] = r[i].s;
    }
    if (r) free(r);
    return n;
}
