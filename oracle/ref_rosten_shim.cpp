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
