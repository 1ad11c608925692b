/* oracle/ref_quicklz_shim.c -- TEST INFRASTRUCTURE ONLY.
 * C entry points over the REFERENCE's vendored QuickLZ (thirdparty/DBoW3/DBoW3/src/quicklz.c, compiled from
 * /root/reference by oracle/Makefile into oracle/_ref/libref_quicklz.so) with the state objects allocated here, so
 * that the tests can write DBoW3's compressed vocabulary stream exactly as Vocabulary::toStream does
 * (Vocabulary.cpp:1343-1362: 10000-byte chunks, one qlz_compress per chunk with one state) and can run the
 * reference's decoder next to the product's (vslam_voc_file.cpp).  Interface: quicklz.h:135-141. */
#include <stdlib.h>
#include <string.h>

#include "quicklz.h"

/* dst must hold n + 400 bytes; returns the packet size */
size_t ref_qlz_compress(const void* src, size_t n, char* dst) {
    qlz_state_compress* st = (qlz_state_compress*)calloc(1, sizeof(qlz_state_compress));
    size_t r = qlz_compress(src, dst, n, st);
    free(st);
    return r;
}

/* dst must hold ref_qlz_size_decompressed(src) bytes (+ 3: the decoder writes 4 bytes per literal run) */
size_t ref_qlz_decompress(const char* src, void* dst) {
    qlz_state_decompress* st = (qlz_state_decompress*)calloc(1, sizeof(qlz_state_decompress));
    size_t r = qlz_decompress(src, dst, st);
    free(st);
    return r;
}

size_t ref_qlz_size_decompressed(const char* src) { return qlz_size_decompressed(src); }
size_t ref_qlz_size_compressed(const char* src) { return qlz_size_compressed(src); }
