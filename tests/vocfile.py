"""TEST INFRASTRUCTURE: writers for DBoW3 vocabulary files, so that the product's reader (vslam_voc_file.cpp) has
something to read -- the reference's ORBvoc blob is not in its tree and DBoW3 itself needs OpenCV to build.

* write_binary follows Vocabulary::toStream (thirdparty/DBoW3/DBoW3/src/Vocabulary.cpp:1292-1366) byte for byte: magic,
  `compressed`, node count, then k / L / scoring / weighting, the nodes in the order of its stack walk (children of the
  last pushed inner node first) as (id, parent, weight, DescManip::toStream = cols, rows, type, bytes), then the words;
  compressed = the same bytes in 10000-byte chunks through the reference's own qlz_compress (oracle/_ref/libref_quicklz.so).
* write_text writes the lines Vocabulary::load_fromtxt (:1372-1446) reads: "k L scoring weighting", then per node
  "parent isLeaf d0 .. d31 weight" in node-id order.
The vocabulary is a dict in vi_slam_amd.synth.make_vocabulary's layout."""
import ctypes as C
import os
import struct

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_QLZ = os.path.join(ROOT, "oracle", "_ref", "libref_quicklz.so")
MAGIC = 88877711233
CHUNK = 10000


def ref_quicklz():
    """ctypes handle of the reference's QuickLZ (None when oracle/_ref has not been built)."""
    if not os.path.exists(REF_QLZ):
        return None
    L = C.CDLL(REF_QLZ)
    L.ref_qlz_compress.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
    L.ref_qlz_compress.restype = C.c_size_t
    L.ref_qlz_decompress.argtypes = [C.c_char_p, C.c_char_p]
    L.ref_qlz_decompress.restype = C.c_size_t
    L.ref_qlz_size_decompressed.argtypes = [C.c_char_p]
    L.ref_qlz_size_decompressed.restype = C.c_size_t
    return L


def ref_compress(data, L=None):
    L = L or ref_quicklz()
    dst = C.create_string_buffer(len(data) + 400)
    n = L.ref_qlz_compress(bytes(data), len(data), dst)
    return dst.raw[:n]


def ref_decompress(packet, L=None):
    L = L or ref_quicklz()
    n = L.ref_qlz_size_decompressed(bytes(packet))
    dst = C.create_string_buffer(n + 16)
    got = L.ref_qlz_decompress(bytes(packet), dst)
    return dst.raw[:got]


def children(voc, node):
    s, c = int(voc["child_start"][node]), int(voc["child_count"][node])
    return [int(x) for x in voc["child_ids"][s:s + c]]


def words(voc):
    """node id of every word, by word id (DBoW3's m_words)"""
    leaves = np.flatnonzero(np.asarray(voc["child_count"]) == 0)
    leaves = leaves[leaves != 0]
    return leaves[np.argsort(np.asarray(voc["word_id"])[leaves], kind="stable")]


def stream_body(voc, scoring=0):
    out = [struct.pack("<iiii", int(voc["k"]), int(voc["L"]), int(scoring), int(voc.get("weighting", 0)))]
    stack = [0]
    while stack:
        pid = stack.pop()
        for c in children(voc, pid):
            out.append(struct.pack("<IId", c, pid, float(voc["weight"][c])))
            out.append(struct.pack("<iii", 32, 1, 0))
            out.append(np.asarray(voc["desc"][c], np.uint8).tobytes())
            if voc["child_count"][c] > 0:
                stack.append(c)
    w = words(voc)
    out.append(struct.pack("<I", len(w)))
    for wid, nid in enumerate(w):
        out.append(struct.pack("<II", wid, int(nid)))
    return b"".join(out)


def binary_bytes(voc, compressed=False, scoring=0, compress=None):
    body = stream_body(voc, scoring)
    head = struct.pack("<Q?I", MAGIC, bool(compressed), len(voc["child_start"]))
    if not compressed:
        return head + body
    compress = compress or ref_compress
    chunks = [body[i:i + CHUNK] for i in range(0, len(body), CHUNK)]
    return head + struct.pack("<I", len(chunks)) + b"".join(compress(c) for c in chunks)


def write_binary(path, voc, compressed=False, scoring=0):
    with open(path, "wb") as f:
        f.write(binary_bytes(voc, compressed, scoring))


def write_text(path, voc, scoring=0):
    """Node ids must be 1..n in file order, parents first (make_vocabulary's breadth-first numbering is)."""
    n = len(voc["child_start"])
    parent = np.zeros(n, np.int64)
    for p in range(n):
        for c in children(voc, p):
            parent[c] = p
    with open(path, "w") as f:
        f.write("%d %d %d %d\n" % (voc["k"], voc["L"], scoring, voc.get("weighting", 0)))
        for i in range(1, n):
            leaf = 1 if voc["child_count"][i] == 0 else 0
            f.write("%d %d %s %.17g\n" % (parent[i], leaf, " ".join(str(int(b)) for b in voc["desc"][i]), voc["weight"][i]))
