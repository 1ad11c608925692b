// dump_opencv_primitives.cpp -- for a maintainer who HAS OpenCV 4.2 (the reference's dependency, README.md:28-30;
// it is neither vendored in the reference nor installed in this project's image, so the oracle's restatement of
// cv::resize / cv::GaussianBlur / cv::fastAtan2 / cv::FAST / cvRound cannot be pinned here -- SURVEY.md 8c).
//
//   1. python tests/golden/export_opencv_inputs.py            (writes tests/golden/opencv/in_*.gray from the fixtures)
//   2. g++ -O2 -std=c++14 tools/dump_opencv_primitives.cpp -o dump_opencv_primitives `pkg-config --cflags --libs opencv4`
//   3. ./dump_opencv_primitives tests/golden/opencv            (writes tests/golden/opencv/out_*.bin)
//   4. python -m pytest tests/test_opencv_primitives.py        (compares the oracle -- both settings of every knob --
//                                                               with what OpenCV actually computed; skipped without out_*)
//
// Every call below is the reference's own call with the reference's own arguments:
//   cv::resize(src, dst, sz, 0, 0, INTER_LINEAR)              src/geometry/fextractor.cpp:1148 (level sizes :1139-1140)
//   cv::GaussianBlur(img, img, Size(7,7), 2, 2, BORDER_REFLECT_101)            :1086
//   cv::FAST(cell, kps, th, true)                             :800-806 (th = 20, then 7)
//   cv::fastAtan2((float)m_01, (float)m_10)                   :94
//   cvRound(float)                                            :72,106,110-111
// Output format (little endian): magic "VSLD", u32 kind, u32 n_dims, u32 dims[n_dims], payload.
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <opencv2/core.hpp>
#include <opencv2/features2d.hpp>
#include <opencv2/imgproc.hpp>

static cv::Mat read_gray(const std::string& path, int w, int h) {
    cv::Mat m(h, w, CV_8UC1);
    FILE* f = fopen(path.c_str(), "rb");
    if (!f || fread(m.data, 1, (size_t)w * h, f) != (size_t)w * h) {
        fprintf(stderr, "cannot read %s\n", path.c_str());
        exit(1);
    }
    fclose(f);
    return m;
}

static void write_blob(const std::string& path, uint32_t kind, const std::vector<uint32_t>& dims, const void* data,
                       size_t bytes) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) {
        fprintf(stderr, "cannot write %s\n", path.c_str());
        exit(1);
    }
    fwrite("VSLD", 1, 4, f);
    const uint32_t nd = (uint32_t)dims.size();
    fwrite(&kind, 4, 1, f);
    fwrite(&nd, 4, 1, f);
    fwrite(dims.data(), 4, nd, f);
    fwrite(data, 1, bytes, f);
    fclose(f);
}

int main(int argc, char** argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: %s <tests/golden/opencv>\n", argv[0]);
        return 2;
    }
    const std::string dir = argv[1];
    struct In { const char* name; int w, h; } inputs[] = {{"hut", 320, 240}, {"lenna", 256, 192}};
    const int nlevels = 8;
    const float scaleFactor = 1.2f;
    std::vector<float> invScale(nlevels, 1.f);
    {   // FExtractor::FExtractor, fextractor.cpp:406-422 (float products, scaleFactor held as double)
        std::vector<float> sc(nlevels, 1.f);
        const double sf = scaleFactor;
        for (int i = 1; i < nlevels; i++) sc[i] = (float)(sc[i - 1] * sf);
        for (int i = 0; i < nlevels; i++) invScale[i] = 1.0f / sc[i];
    }
    for (const In& in : inputs) {
        const cv::Mat img = read_gray(dir + "/in_" + in.name + "_" + std::to_string(in.w) + "x" + std::to_string(in.h) + ".gray",
                                      in.w, in.h);
        std::vector<cv::Mat> pyr(nlevels);
        pyr[0] = img;
        for (int l = 1; l < nlevels; l++) {  // ComputePyramid without the (unused) border, :1135-1160
            const cv::Size sz(cvRound((float)img.cols * invScale[l]), cvRound((float)img.rows * invScale[l]));
            cv::resize(pyr[l - 1], pyr[l], sz, 0, 0, cv::INTER_LINEAR);
        }
        for (int l = 0; l < nlevels; l++) {
            const std::string tag = std::string(in.name) + "_l" + std::to_string(l);
            write_blob(dir + "/out_resize_" + tag + ".bin", 1, {(uint32_t)pyr[l].rows, (uint32_t)pyr[l].cols}, pyr[l].data,
                       (size_t)pyr[l].rows * pyr[l].cols);
            cv::Mat b = pyr[l].clone();
            cv::GaussianBlur(b, b, cv::Size(7, 7), 2, 2, cv::BORDER_REFLECT_101);
            write_blob(dir + "/out_blur_" + tag + ".bin", 2, {(uint32_t)b.rows, (uint32_t)b.cols}, b.data, (size_t)b.rows * b.cols);
        }
        for (int th : {20, 7}) {  // cv::FAST on the whole crop = one "cell" (x, y, response as int32 triples, raster order)
            std::vector<cv::KeyPoint> kps;
            cv::FAST(img, kps, th, true);
            std::vector<int32_t> t;
            for (const cv::KeyPoint& k : kps) {
                t.push_back((int32_t)k.pt.x);
                t.push_back((int32_t)k.pt.y);
                t.push_back((int32_t)k.response);
            }
            write_blob(dir + "/out_fast_" + in.name + "_th" + std::to_string(th) + ".bin", 3, {(uint32_t)kps.size(), 3}, t.data(),
                       t.size() * 4);
        }
    }
    {   // fastAtan2 on integer moment pairs (what IC_Angle feeds it) and cvRound on halves
        std::vector<float> yx, out;
        uint64_t s = 20250215;
        for (int i = 0; i < 200000; i++) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            const int m01 = (int)((s >> 33) % 400001) - 200000;
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            const int m10 = (int)((s >> 33) % 400001) - 200000;
            yx.push_back((float)m01);
            yx.push_back((float)m10);
            out.push_back(cv::fastAtan2((float)m01, (float)m10));
        }
        write_blob(dir + "/out_atan2_in.bin", 4, {(uint32_t)out.size(), 2}, yx.data(), yx.size() * 4);
        write_blob(dir + "/out_atan2.bin", 5, {(uint32_t)out.size()}, out.data(), out.size() * 4);
        std::vector<float> rin;
        std::vector<int32_t> rout;
        for (int i = -2000; i <= 2000; i++) {
            rin.push_back(i * 0.25f);
            rout.push_back(cvRound(i * 0.25f));
        }
        write_blob(dir + "/out_cvround_in.bin", 6, {(uint32_t)rin.size()}, rin.data(), rin.size() * 4);
        write_blob(dir + "/out_cvround.bin", 7, {(uint32_t)rout.size()}, rout.data(), rout.size() * 4);
    }
    printf("wrote %s/out_*.bin (OpenCV %s)\n", dir.c_str(), CV_VERSION);
    return 0;
}
