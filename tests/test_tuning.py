"""Behaviour switches (include/vslam_fe.h: vslam_tuning): resolved once per context, the environment read once per
process, nothing cached in unsynchronised statics.  CPU only: the resolver is plain C++."""
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "vi_slam_amd", "csrc")


def test_library_reads_the_environment_in_one_place_only():
    n = 0
    for f in os.listdir(CSRC):
        if f.endswith((".hip", ".cpp", ".h")):
            txt = open(os.path.join(CSRC, f)).read()
            calls = len(re.findall(r"\bgetenv\s*\(", txt))
            assert calls == 0 or f == "vslam_tuning.cpp", (f, calls)
            n += calls
    assert n == 1
    # no function-local "static int x = -1" switch caches left in the launchers
    for f in os.listdir(CSRC):
        if f.endswith(".hip"):
            assert not re.search(r"static int \w+ = -1;", open(os.path.join(CSRC, f)).read()), f


def test_resolver_two_threads_under_thread_sanitizer(tmp_path):
    exe = str(tmp_path / "tuning_threads")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread", "-o", exe,
                           os.path.join(HERE, "cpp", "tuning_threads.cpp"), os.path.join(CSRC, "vslam_tuning.cpp")])
    env = dict(os.environ)
    for k in list(env):
        if k.startswith("VSLAM_"):
            del env[k]
    out = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "ThreadSanitizer" not in out.stderr, out.stderr[-3000:]
    lines = out.stdout.strip().splitlines()
    for i in range(8):
        want_d = 5 if i & 1 else 3  # the caller's field wins over the environment's default
        assert lines[i] == ("ctx %d: oct_fine_depth %d h2d_route 2 pyramid_per_level 1 d2h_route -1 init_topm %d fast_threads -1"
                            % (i, want_d, i)), lines[i]
    assert lines[8] == "late: oct_fine_depth 3"  # the environment is read once per process


def test_python_mirror_matches_the_header():
    import vi_slam_amd as V
    hdr = open(os.path.join(ROOT, "include", "vslam_fe.h")).read()
    body = hdr[hdr.index("typedef struct vslam_tuning {"):hdr.index("} vslam_tuning;")]
    fields = re.findall(r"int32_t (\w+);", body)
    assert fields == V.TUNING_FIELDS
    assert "int32_t reserved[1];" in body
