/* vslam_voc_file.cpp -- reader for DBoW3 vocabulary files, the host half of vslam_voc_load.
 *
 * Replaces DBoW3::Vocabulary::load(const std::string&) (thirdparty/DBoW3/DBoW3/src/Vocabulary.cpp:1084-1112), which
 * core::System calls once at start-up (src/core/system.cpp:76).  The formats it accepts, in the order it tries them:
 *   1. the binary stream of Vocabulary::toStream / fromStream (:1292-1366, :1447-1521): u64 magic 88877711233, bool
 *      compressed, u32 node count, then (optionally in QuickLZ level-1 chunks of 10000 bytes, :1343-1362) k, L, scoring,
 *      weighting, the nodes as (id, parent, weight, descriptor) and the words as (word id, node id).  This is what
 *      Vocabulary::save writes by default, whatever the file is called (tools/createVoc/createVoc.cpp:57 saves
 *      "vocabulary.txt" compressed);
 *   2. the text form of load_fromtxt (:1372-1446) when the name contains ".txt": "k L scoring weighting", then one line
 *      per node "parent isLeaf d0 .. d31 weight";
 *   3. cv::FileStorage YAML/XML (:1523-1613): not read here (it is OpenCV's persistence format); VSLAM_ERR_UNSUPPORTED.
 * The result is the flat node table vslam_voc_create takes: children in the order the file attaches them (that order
 * decides ties in Vocabulary::transform), 32-byte descriptors, double weights (float precision for the text form, as in
 * the reference, which parses every token as float), word ids.
 *
 * The QuickLZ decoder below is written against the stream layout of QuickLZ 1.5.0 at compression level 1 with no
 * streaming buffer -- the settings of the reference's vendored copy (thirdparty/DBoW3/DBoW3/src/quicklz.h:25,31) -- with
 * every read and write bounds-checked (the vendored decoder is compiled without QLZ_MEMORY_SAFE).  No GPU code here:
 * the file is part of libvslam_fe.so and of the GPU-free libvslam_host.so the CPU tests load. */
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/vslam_fe.h"

namespace {

thread_local std::string t_err;

int fail(int code, const std::string& what) {
    t_err = what;
    return code;
}

/* bounds-checked little-endian reader over a byte range */
struct Cursor {
    const uint8_t* p;
    size_t left;
    bool take(void* dst, size_t n) {
        if (n > left) return false;
        memcpy(dst, p, n);
        p += n;
        left -= n;
        return true;
    }
    template <typename T>
    bool get(T* v) {
        return take(v, sizeof(T));
    }
};

inline uint32_t le(const uint8_t* s, int n) {
    uint32_t v = 0;
    for (int i = 0; i < n; i++) v |= (uint32_t)s[i] << (8 * i);
    return v;
}

/* One QuickLZ packet (header + payload) from `in`, appended to `out`.
 * Header: flags (bit 0 payload is compressed, bit 1 sizes are 4 bytes instead of 1, bits 2-3 level), packet size,
 * decoded size.  Level-1 payload: 32-bit control words (31 flags under a sentinel bit, consumed from bit 0); flag 1 = a
 * match of 2 bytes (12-bit table slot, 4-bit length - 2) or 3 bytes (length in the third), flag 0 = literals; the table
 * slot of a position is a hash of its three bytes and always names the latest hashed position, positions inside a
 * match are not hashed, and the last 10 bytes are always literals. */
bool qlz_packet(Cursor& in, std::vector<uint8_t>& out) {
    if (in.left < 3) return false;
    const uint8_t flags = in.p[0];
    const int n = (flags & 2) ? 4 : 1;
    const size_t header = 2 * (size_t)n + 1;
    if (in.left < header) return false;
    const size_t psize = le(in.p + 1, n), dsize = le(in.p + 1 + n, n);
    if (psize < header || psize > in.left) return false;
    if (dsize > (psize - header) * 128 + 16) return false; /* a match of <= 255 bytes costs >= 2: no payload expands that far */
    const uint8_t* src = in.p + header;
    const uint8_t* const end = in.p + psize;
    in.p += psize;
    in.left -= psize;
    const size_t base = out.size();
    out.resize(base + dsize);
    uint8_t* const dst = out.data() + base;
    if (!(flags & 1)) {
        if ((size_t)(end - src) < dsize) return false;
        memcpy(dst, src, dsize);
        return true;
    }
    if (((flags >> 2) & 3) != 1) return false; /* another compression level: another match encoding */

    std::vector<int32_t> slot(4096, -1);
    auto hash_at = [&](long pos) {
        const uint32_t v = le(dst + pos, 3);
        return ((v >> 12) ^ v) & 4095u;
    };
    const long size = (long)dsize;
    long d = 0, hashed = -1;
    uint32_t cword = 1;
    for (;;) {
        if (cword == 1) {
            if (end - src < 4) return false;
            cword = le(src, 4);
            src += 4;
        }
        if (cword & 1) {
            cword >>= 1;
            if (end - src < 2) return false;
            const uint32_t f = le(src, 2);
            long len;
            if (f & 15) {
                len = (f & 15) + 2;
                src += 2;
            } else {
                if (end - src < 3) return false;
                len = src[2];
                src += 3;
            }
            const long from = slot[(f >> 4) & 4095];
            if (from < 0 || from > d - 3 || len < 3 || len > size - d - 4) return false;
            for (long i = 0; i < len; i++) dst[d + i] = dst[from + i]; /* forward, byte by byte: ranges may overlap */
            while (hashed < d) {
                hashed++;
                slot[hash_at(hashed)] = (int32_t)hashed;
            }
            d += len;
            hashed = d - 1;
        } else if (d < size - 11) {
            static const int run[16] = {4, 0, 1, 0, 2, 0, 1, 0, 3, 0, 1, 0, 2, 0, 1, 0};
            const int k = run[cword & 15];
            if (end - src < k) return false;
            memcpy(dst + d, src, k);
            cword >>= k;
            d += k;
            src += k;
            while (hashed < d - 3) {
                hashed++;
                slot[hash_at(hashed)] = (int32_t)hashed;
            }
        } else {
            while (d < size) {
                if (cword == 1) {
                    if (end - src < 4) return false;
                    src += 4;
                    cword = 1u << 31;
                }
                if (end - src < 1) return false;
                dst[d++] = *src++;
                cword >>= 1;
            }
            return true;
        }
    }
}

}  // namespace

struct vslam_voc_file {
    int k = 0, L = 0, scoring = 0, weighting = 0, format = 0, n_words = 0;
    std::vector<int32_t> child_start, child_count, child_ids, word_id;
    std::vector<uint8_t> desc;
    std::vector<double> weight;
};

namespace {

/* (parent, child) pairs in the order the file attaches them -> child_start / child_count / child_ids */
int link_children(vslam_voc_file& v, const std::vector<uint32_t>& parent, const std::vector<uint32_t>& order) {
    const size_t n = parent.size();
    v.child_start.assign(n, 0);
    v.child_count.assign(n, 0);
    for (uint32_t c : order) v.child_count[parent[c]]++;
    int32_t pos = 0;
    for (size_t i = 0; i < n; i++) {
        v.child_start[i] = pos;
        pos += v.child_count[i];
    }
    v.child_ids.assign(order.size(), 0);
    std::vector<int32_t> fill(v.child_start);
    for (uint32_t c : order) v.child_ids[fill[parent[c]]++] = (int32_t)c;
    return VSLAM_OK;
}

int parse_binary(const std::vector<uint8_t>& file, vslam_voc_file& v) {
    Cursor in{file.data(), file.size()};
    uint64_t sig = 0;
    uint8_t compressed = 0;
    uint32_t nnodes = 0;
    if (!in.get(&sig) || !in.get(&compressed) || !in.get(&nnodes)) return fail(VSLAM_ERR_INVALID, "vocabulary file: truncated header");
    if (nnodes < 2) return fail(VSLAM_ERR_INVALID, "vocabulary file: no nodes");
    std::vector<uint8_t> plain;
    Cursor body = in;
    if (compressed) {
        uint32_t nchunks = 0;
        if (!in.get(&nchunks)) return fail(VSLAM_ERR_INVALID, "vocabulary file: truncated header");
        plain.reserve((size_t)nchunks * 10000);
        for (uint32_t c = 0; c < nchunks; c++)
            if (!qlz_packet(in, plain))
                return fail(VSLAM_ERR_INVALID, "vocabulary file: QuickLZ chunk " + std::to_string(c) + " of " +
                                                   std::to_string(nchunks) + " is damaged or not level-1 QuickLZ 1.5");
        body = Cursor{plain.data(), plain.size()};
    }
    int32_t k, L, scoring, weighting;
    if (!body.get(&k) || !body.get(&L) || !body.get(&scoring) || !body.get(&weighting))
        return fail(VSLAM_ERR_INVALID, "vocabulary file: truncated parameters");
    if (L < 1 || scoring < 0 || scoring > 5 || weighting < 0 || weighting > 3)
        return fail(VSLAM_ERR_INVALID, "vocabulary file: parameters out of range");
    v.k = k, v.L = L, v.scoring = scoring, v.weighting = weighting;
    v.format = compressed ? 2 : 1;
    v.desc.assign((size_t)nnodes * 32, 0);
    v.weight.assign(nnodes, 0.0);
    v.word_id.assign(nnodes, 0);
    std::vector<uint32_t> parent(nnodes, 0), order;
    std::vector<uint8_t> seen(nnodes, 0);
    order.reserve(nnodes - 1);
    for (uint32_t i = 1; i < nnodes; i++) {
        uint32_t nid, pid;
        double w;
        int32_t cols, rows, type;
        if (!body.get(&nid) || !body.get(&pid) || !body.get(&w) || !body.get(&cols) || !body.get(&rows) || !body.get(&type))
            return fail(VSLAM_ERR_INVALID, "vocabulary file: truncated node table");
        if (nid == 0 || nid >= nnodes || pid >= nnodes || seen[nid])
            return fail(VSLAM_ERR_INVALID, "vocabulary file: node id out of range or repeated");
        if (type != 0 || rows != 1 || cols != 32)
            return fail(VSLAM_ERR_UNSUPPORTED, "vocabulary file: descriptors are not 1 x 32 CV_8U (ORB)");
        if (!body.take(&v.desc[(size_t)nid * 32], 32)) return fail(VSLAM_ERR_INVALID, "vocabulary file: truncated node table");
        seen[nid] = 1;
        parent[nid] = pid;
        v.weight[nid] = w;
        order.push_back(nid);
    }
    uint32_t nwords = 0;
    if (!body.get(&nwords) || nwords > nnodes) return fail(VSLAM_ERR_INVALID, "vocabulary file: truncated word table");
    for (uint32_t i = 0; i < nwords; i++) {
        uint32_t wid, nid;
        if (!body.get(&wid) || !body.get(&nid)) return fail(VSLAM_ERR_INVALID, "vocabulary file: truncated word table");
        if (wid >= nwords || nid >= nnodes) return fail(VSLAM_ERR_INVALID, "vocabulary file: word table out of range");
        v.word_id[nid] = (int32_t)wid;
    }
    v.n_words = (int)nwords;
    return link_children(v, parent, order);
}

int parse_text(const std::vector<uint8_t>& file, vslam_voc_file& v) {
    const char* s = (const char*)file.data();
    const char* const e = s + file.size();
    auto line_end = [&](const char* p) {
        while (p < e && *p != '\n') p++;
        return p;
    };
    /* std::string copies keep strtol / strtof inside the line */
    const char* le0 = line_end(s);
    std::string first(s, le0);
    int k = -1, L = -1, n1 = -1, n2 = -1;
    if (sscanf(first.c_str(), "%d %d %d %d", &k, &L, &n1, &n2) != 4 || k < 0 || k > 20 || L < 1 || L > 10 || n1 < 0 ||
        n1 > 5 || n2 < 0 || n2 > 3)
        return fail(VSLAM_ERR_INVALID, "vocabulary file: not a DBoW3 text vocabulary (first line must be 'k L scoring weighting')");
    v.k = k, v.L = L, v.scoring = n1, v.weighting = n2, v.format = 3;
    std::vector<uint32_t> parent(1, 0), order;
    v.desc.assign(32, 0);
    v.weight.assign(1, 0.0);
    v.word_id.assign(1, 0);
    int nwords = 0;
    std::vector<float> vals;
    for (s = le0 < e ? le0 + 1 : e; s < e;) {
        const char* le1 = line_end(s);
        std::string line(s, le1);
        s = le1 < e ? le1 + 1 : e;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) break; /* the reference stops at the first empty line */
        const char* p = line.c_str();
        char* q = nullptr;
        const long pid = strtol(p, &q, 10);
        if (q == p) return fail(VSLAM_ERR_INVALID, "vocabulary file: malformed node line");
        p = q;
        const long leaf = strtol(p, &q, 10);
        if (q == p) return fail(VSLAM_ERR_INVALID, "vocabulary file: malformed node line");
        p = q;
        vals.clear();
        for (;;) {
            const float f = strtof(p, &q);
            if (q == p) break;
            vals.push_back(f);
            p = q;
        }
        const uint32_t nid = (uint32_t)parent.size();
        if (pid < 0 || (uint32_t)pid >= nid) return fail(VSLAM_ERR_INVALID, "vocabulary file: parent id does not precede its node");
        if (vals.size() != 33) return fail(VSLAM_ERR_UNSUPPORTED, "vocabulary file: descriptors are not 32 bytes (ORB)");
        parent.push_back((uint32_t)pid);
        order.push_back(nid);
        v.weight.push_back((double)vals[32]);
        for (int i = 0; i < 32; i++) {
            if (!(vals[i] >= 0.f && vals[i] < 256.f)) return fail(VSLAM_ERR_INVALID, "vocabulary file: descriptor byte out of range");
            v.desc.push_back((uint8_t)vals[i]);
        }
        v.word_id.push_back(leaf > 0 ? nwords : 0);
        if (leaf > 0) nwords++;
    }
    if (parent.size() < 2) return fail(VSLAM_ERR_INVALID, "vocabulary file: no nodes");
    v.n_words = nwords;
    return link_children(v, parent, order);
}

}  // namespace

/* test hook: one QuickLZ packet -> dst; returns the decoded size, or -1 (damaged / not level 1 / dst too small) */
extern "C" long vslam_dbg_qlz_decode(const uint8_t* packet, size_t n, uint8_t* dst, size_t cap, size_t* used) {
    if (!packet || (!dst && cap)) return -1;
    Cursor in{packet, n};
    std::vector<uint8_t> out;
    if (!qlz_packet(in, out) || out.size() > cap) return -1;
    if (!out.empty()) memcpy(dst, out.data(), out.size());
    if (used) *used = n - in.left;
    return (long)out.size();
}

extern "C" const char* vslam_voc_file_last_error(void) { return t_err.c_str(); }

extern "C" int vslam_voc_file_open(const char* path, vslam_voc_file** out) {
    if (!path || !out) return fail(VSLAM_ERR_INVALID, "invalid arguments");
    *out = nullptr;
    FILE* f = fopen(path, "rb");
    if (!f) return fail(VSLAM_ERR_INVALID, std::string("vocabulary file: cannot open ") + path);
    std::vector<uint8_t> file;
    uint8_t buf[1 << 16];
    size_t got;
    while ((got = fread(buf, 1, sizeof(buf), f)) > 0) file.insert(file.end(), buf, buf + got);
    fclose(f);
    vslam_voc_file* v = new vslam_voc_file();
    uint64_t sig = 0;
    if (file.size() >= 8) memcpy(&sig, file.data(), 8);
    int rc;
    if (sig == 88877711233ull)
        rc = parse_binary(file, *v);
    else if (std::string(path).find(".txt") != std::string::npos)
        rc = parse_text(file, *v);
    else
        rc = fail(VSLAM_ERR_UNSUPPORTED, "vocabulary file: neither DBoW3's binary stream nor a .txt vocabulary; a cv::FileStorage "
                                         "YAML/XML vocabulary must be re-saved with Vocabulary::save(\"name.dbow3\")");
    if (rc != VSLAM_OK) {
        delete v;
        return rc;
    }
    *out = v;
    return VSLAM_OK;
}

extern "C" void vslam_voc_file_close(vslam_voc_file* v) { delete v; }

extern "C" int vslam_voc_file_info(const vslam_voc_file* v, int* branching, int* depth_levels, int* scoring, int* weighting,
                                   int* norm, int* n_nodes, int* n_words, int* n_child_ids, int* format) {
    if (!v) return fail(VSLAM_ERR_INVALID, "invalid arguments");
    if (branching) *branching = v->k;
    if (depth_levels) *depth_levels = v->L;
    if (scoring) *scoring = v->scoring;
    if (weighting) *weighting = v->weighting;
    /* GeneralScoring::mustNormalize (ScoringObject.h:72-88): L1_NORM, CHI_SQUARE, KL, BHATTACHARYYA -> L1; L2_NORM -> L2;
     * DOT_PRODUCT -> none */
    if (norm) *norm = v->scoring == 1 ? 2 : v->scoring == 5 ? 0 : 1;
    if (n_nodes) *n_nodes = (int)v->child_start.size();
    if (n_words) *n_words = v->n_words;
    if (n_child_ids) *n_child_ids = (int)v->child_ids.size();
    if (format) *format = v->format;
    return VSLAM_OK;
}

extern "C" int vslam_voc_file_arrays(const vslam_voc_file* v, const int32_t** child_start, const int32_t** child_count,
                                     const int32_t** child_ids, const uint8_t** node_desc, const double** node_weight,
                                     const int32_t** node_word_id) {
    if (!v) return fail(VSLAM_ERR_INVALID, "invalid arguments");
    if (child_start) *child_start = v->child_start.data();
    if (child_count) *child_count = v->child_count.data();
    if (child_ids) *child_ids = v->child_ids.data();
    if (node_desc) *node_desc = v->desc.data();
    if (node_weight) *node_weight = v->weight.data();
    if (node_word_id) *node_word_id = v->word_id.data();
    return VSLAM_OK;
}
