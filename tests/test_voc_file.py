"""DBoW3 vocabulary files (Vocabulary::load, thirdparty/DBoW3/DBoW3/src/Vocabulary.cpp:1084-1112, :1372-1521): the
product's reader (vslam_voc_file.cpp, through the GPU-free libvslam_host.so) against files written by tests/vocfile.py
the way DBoW3 writes them, against a committed file whose chunks the REFERENCE's QuickLZ compressed, and its QuickLZ
decoder against packets of that compressor (committed, plus fresh ones where oracle/_ref is built)."""
import ctypes as C
import os

import numpy as np
import pytest

import vi_slam_amd as V
from vi_slam_amd import synth

import vocfile

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def H():
    return V.bind_voc_file(C.CDLL(V.HOST_LIB_PATH))


def same_tree(got, voc, weight=None):
    for k in ("child_count", "child_ids"):
        assert np.array_equal(got[k], voc[k]), k
    assert np.array_equal(got["desc"][1:], voc["desc"][1:]) and not got["desc"][0].any()   # the root is not in the file
    inner = voc["child_count"] > 0
    assert np.array_equal(got["child_start"][inner], voc["child_start"][inner])
    assert np.array_equal(got["weight"], voc["weight"] if weight is None else weight)
    leaves = np.flatnonzero(voc["child_count"] == 0)
    assert np.array_equal(got["word_id"][leaves], voc["word_id"][leaves])
    assert got["k"] == voc["k"] and got["L"] == voc["L"] and got["weighting"] == voc["weighting"]
    assert got["n_words"] == len(leaves)


@pytest.mark.parametrize("k,L,weighting", [(10, 3, 0), (4, 5, 1), (7, 2, 3), (2, 1, 2)])
def test_plain_binary_stream(tmp_path, H, k, L, weighting):
    voc = synth.make_vocabulary(k, L, seed=k + L, weighting=weighting)
    p = str(tmp_path / "voc.dbow3")
    vocfile.write_binary(p, voc)
    got = V.read_vocabulary_file(p, H)
    same_tree(got, voc)
    assert got["format"] == "dbow3-binary" and got["norm"] == 1


def test_committed_file_compressed_by_the_reference_quicklz(H):
    """tests/golden/voc_k8_L3_quicklz.dbow3 (make_voc_golden.py): 585 nodes, four 10000-byte chunks, the last partial"""
    got = V.read_vocabulary_file(os.path.join(GOLD, "voc_k8_L3_quicklz.dbow3"), H)
    same_tree(got, synth.make_vocabulary(8, 3, seed=5))
    assert got["format"] == "dbow3-binary-quicklz"


@pytest.mark.skipif(vocfile.ref_quicklz() is None, reason="oracle/_ref/libref_quicklz.so not built (needs the reference tree)")
@pytest.mark.parametrize("k,L", [(10, 4), (3, 2), (6, 3)])
def test_compressed_stream_written_with_the_reference_compressor(tmp_path, H, k, L):
    """(10, 4): 11 111 nodes, 67 chunks; (3, 2): one packet of 788 bytes"""
    voc = synth.make_vocabulary(k, L, seed=3 * k + L)
    # a tree with many equal descriptors compresses into long matches
    voc["desc"][len(voc["desc"]) // 2:] = voc["desc"][1]
    p = str(tmp_path / "vocabulary.txt")  # createVoc.cpp:57 calls its compressed stream "vocabulary.txt"
    vocfile.write_binary(p, voc, compressed=True)
    same_tree(V.read_vocabulary_file(p, H), voc)
    assert os.path.getsize(p) < len(vocfile.binary_bytes(voc))


def test_text_form_keeps_float_weights(tmp_path, H):
    voc = synth.make_vocabulary(5, 3, seed=8, weighting=2)
    p = str(tmp_path / "ORBvoc.txt")
    vocfile.write_text(p, voc, scoring=1)
    got = V.read_vocabulary_file(p, H)
    same_tree(got, voc, weight=voc["weight"].astype(np.float32).astype(np.float64))
    assert got["format"] == "dbow3-text" and got["scoring"] == 1 and got["norm"] == 2
    # the reference stops at the first empty line and accepts CR LF
    txt = open(p).read().split("\n")
    open(p, "w").write("\r\n".join(txt[:40]) + "\r\n\r\n" + "\n".join(txt[40:]))
    cut = V.read_vocabulary_file(p, H)
    assert len(cut["child_start"]) == 40 and np.array_equal(cut["desc"][1:], voc["desc"][1:40])


@pytest.mark.parametrize("scoring,norm", [(0, 1), (1, 2), (2, 1), (3, 1), (4, 1), (5, 0)])
def test_scoring_type_gives_the_norm(tmp_path, H, scoring, norm):
    """createScoringObject (Vocabulary.cpp:50-84) + mustNormalize (ScoringObject.h:72-88)"""
    voc = synth.make_vocabulary(3, 2, seed=1)
    p = str(tmp_path / "v.dbow3")
    vocfile.write_binary(p, voc, scoring=scoring)
    got = V.read_vocabulary_file(p, H)
    assert got["scoring"] == scoring and got["norm"] == norm


def test_errors(tmp_path, H):
    voc = synth.make_vocabulary(3, 2, seed=1)
    with pytest.raises(V.VslamError) as e:
        V.read_vocabulary_file(str(tmp_path / "missing.dbow3"), H)
    assert e.value.code == V.ERR_INVALID and "cannot open" in str(e.value)
    y = tmp_path / "voc.yml"
    y.write_text("%YAML:1.0\nvocabulary:\n   k: 10\n")
    with pytest.raises(V.VslamError) as e:
        V.read_vocabulary_file(str(y), H)
    assert e.value.code == V.ERR_UNSUPPORTED
    t = tmp_path / "bad.txt"
    t.write_text("hello world\n")
    with pytest.raises(V.VslamError) as e:
        V.read_vocabulary_file(str(t), H)
    assert e.value.code == V.ERR_INVALID
    t.write_text("3 2 0 0\n5 0 " + " ".join(["1"] * 32) + " 0.5\n")  # parent 5 does not exist yet
    with pytest.raises(V.VslamError) as e:
        V.read_vocabulary_file(str(t), H)
    assert e.value.code == V.ERR_INVALID
    t.write_text("3 2 0 0\n0 1 " + " ".join(["1"] * 64) + " 0.5\n")  # 64-byte descriptors
    with pytest.raises(V.VslamError) as e:
        V.read_vocabulary_file(str(t), H)
    assert e.value.code == V.ERR_UNSUPPORTED
    raw = bytearray(vocfile.binary_bytes(voc))
    raw[13 + 16 + 16] = 64  # cols of the first node's descriptor
    b = tmp_path / "cols.dbow3"
    b.write_bytes(bytes(raw))
    with pytest.raises(V.VslamError) as e:
        V.read_vocabulary_file(str(b), H)
    assert e.value.code == V.ERR_UNSUPPORTED


def test_every_truncation_and_random_damage_is_an_error_not_a_crash(tmp_path, H):
    voc = synth.make_vocabulary(3, 2, seed=1)
    plain = vocfile.binary_bytes(voc)
    packed = open(os.path.join(GOLD, "voc_k8_L3_quicklz.dbow3"), "rb").read()
    p = tmp_path / "t.dbow3"
    for n in list(range(0, 120)) + list(range(120, len(plain), 37)):
        p.write_bytes(plain[:n])
        with pytest.raises(V.VslamError):
            V.read_vocabulary_file(str(p), H)
    for n in range(14, len(packed) - 1, 211):
        p.write_bytes(packed[:n])
        with pytest.raises(V.VslamError):
            V.read_vocabulary_file(str(p), H)
    rng = np.random.default_rng(0)
    want = synth.make_vocabulary(8, 3, seed=5)
    for _ in range(300):
        b = bytearray(packed)
        for i in rng.integers(17, len(b), int(rng.integers(1, 4))):
            b[i] ^= 1 << int(rng.integers(0, 8))
        p.write_bytes(bytes(b))
        try:
            got = V.read_vocabulary_file(str(p), H)   # a flipped descriptor or weight bit still parses
        except V.VslamError as e:
            assert e.code in (V.ERR_INVALID, V.ERR_UNSUPPORTED)
            continue
        assert len(got["child_start"]) == len(want["child_start"])


def _packets():
    z = np.load(os.path.join(GOLD, "quicklz_packets.npz"))
    cut = lambda a, o: [a[o[i]:o[i + 1]].tobytes() for i in range(len(o) - 1)]
    return cut(z["plain"], z["plain_off"]), cut(z["packets"], z["packets_off"])


def test_quicklz_decoder_on_the_reference_compressor_packets(H):
    plain, packets = _packets()
    assert len(packets) == 80
    kinds = set()
    for d, pk in zip(plain, packets):
        out, used = V.qlz_decode(pk + b"trailing", H)
        assert out == d and used == len(pk)
        kinds.add((pk[0] & 1, pk[0] & 2))
    assert kinds == {(0, 0), (1, 0), (0, 2), (1, 2)}  # stored / compressed x short / long header
    for pk in packets:
        assert ((pk[0] >> 2) & 3) == 1   # level 1, the reference's setting (quicklz.h:25)
        for n in (0, 1, 2, len(pk) // 2, len(pk) - 1):
            with pytest.raises(ValueError):
                V.qlz_decode(pk[:n], H)
    rng = np.random.default_rng(1)
    for pk in packets:          # damage must end in an error or in bounded output, never out of bounds
        for _ in range(20):
            b = bytearray(pk)
            b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
            try:
                V.qlz_decode(bytes(b), H)
            except ValueError:
                pass
    other_level = bytearray(packets[4])
    assert other_level[0] & 1
    other_level[0] = (other_level[0] & ~0x0c) | (3 << 2)
    with pytest.raises(ValueError):
        V.qlz_decode(bytes(other_level), H)


@pytest.mark.skipif(vocfile.ref_quicklz() is None, reason="oracle/_ref/libref_quicklz.so not built (needs the reference tree)")
def test_quicklz_decoder_equals_the_reference_decoder_on_fresh_packets(H):
    R = vocfile.ref_quicklz()
    rng = np.random.default_rng(7)
    for trial in range(400):
        n = int(rng.integers(1, 12000))
        if trial % 3 == 0:
            d = rng.integers(0, 1 + int(rng.integers(1, 256)), n, dtype=np.uint8).tobytes()
        elif trial % 3 == 1:
            period = int(rng.integers(1, 300))
            d = (rng.integers(0, 256, period, dtype=np.uint8).tobytes() * (n // period + 1))[:n]
        else:
            base = rng.integers(0, 256, 256, dtype=np.uint8).tobytes()
            d = b"".join(base[int(rng.integers(0, 250)):][:int(rng.integers(1, 300))] for _ in range(n // 20 + 1))[:n]
        pk = vocfile.ref_compress(d, R)
        out, used = V.qlz_decode(pk, H)
        assert out == vocfile.ref_decompress(pk, R) == d and used == len(pk)
