"""GPU (-m gpu): round-2 additions -- pinned-image input, context re-use across input modes (graph replay),
matcher overflow reporting, the in-library RCCL exchange at world size 1, and the BASELINE configs that round 1
left without a parity test (stereo / init matcher at 1920x1080 N=4000, mono N=2000 = the YAML value)."""
import ctypes as C
import os

import numpy as np
import pytest

import vi_slam_amd as V
from oracle import orbo
from vi_slam_amd import synth

pytestmark = pytest.mark.gpu

BF, FX = 386.1448, 718.856


def _same_feats(res, ref, tag=""):
    k, d = res[0], res[1]
    ko, do = ref[0], ref[1]
    assert len(k) == len(ko), tag
    for f in k.dtype.names:
        assert np.array_equal(k[f], ko[f]), (tag, f)
    assert np.array_equal(d, do), tag


def _stereo_ref(L, R, nf):
    eL, eR = orbo.Extractor(nf), orbo.Extractor(nf)
    kL, dL, _ = eL.compute(L)
    kR, dR, _ = eR.compute(R)
    u, dep, _, _ = orbo.stereo(eL, eR, kL, dL, kR, dR, BF, FX)
    return (kL, dL), (kR, dR), u, dep


@pytest.mark.parametrize("pitch", [1241, 1243, 1280])
def test_pinned_host_images_equal_oracle(pitch):
    """VSLAM_IMGS_PINNED: the pass pulls the caller's pinned rows over PCIe itself (k_pull_images); rows of 1241
    bytes start at odd addresses (unaligned 16-byte loads, bytewise row tail)."""
    fe = V.FExtractor(2000, 1.2, 8, 20, 7, 1241, 376, max_batch=4)
    pin = V.PinnedImages(4, 376, 1241, pitch)
    try:
        imgs = [synth.make_frame(1241, 376, step=s, right=bool(s & 1)) for s in range(4)]
        for rep in range(2):  # second round: same pointers -> the captured graph is replayed; new content must be seen
            for s in range(4):
                pin.array[s][:] = imgs[(s + rep) % 4]
            fe.compute_batch_async(pin.ptrs, pitch, (0, 0), where=V.IMGS_PINNED)
            res = fe.wait(copy=True)
            e = orbo.Extractor(2000)
            for s in range(4):
                _same_feats(res[s], e.compute(imgs[(s + rep) % 4]), "pinned %d/%d" % (rep, s))
                assert np.array_equal(fe.mvImagePyramid(0, slot=s), imgs[(s + rep) % 4])
        # pinned stereo frames through the one-enqueue frame path
        fe.frame_stereo_async(pin.ptrs, pitch, BF, FX, where=V.IMGS_PINNED)
        feats, st = fe.frame_stereo_wait()
        for j in range(2):
            (kL, dL), (kR, dR), wu, wd = _stereo_ref(np.array(pin.array[2 * j]), np.array(pin.array[2 * j + 1]), 2000)
            _same_feats(feats[2 * j], (kL, dL))
            _same_feats(feats[2 * j + 1], (kR, dR))
            assert np.array_equal(st[j][0], wu) and np.array_equal(st[j][1], wd)
    finally:
        pin.close()
        fe.close()


def test_host_then_device_then_host_keeps_level0_sources():
    """a host-image pass replays a captured graph; a device-image pass in between must not leave the
    context's level-0 pointers on the caller's (then freed) device images when ComputeStereoMatches reads level 0."""
    import torch
    fe = V.FExtractor(1000, 1.2, 8, 20, 7, 640, 360, max_batch=2)
    try:
        L, R = synth.make_stereo_pair(640, 360, seed=11)
        L2, R2 = synth.make_stereo_pair(640, 360, seed=12)
        fe.compute_batch([L, R])          # host pass: captures the graph
        fe.compute_batch([L, R])          # host pass: replays it
        dev = torch.zeros((2, 360, 640), dtype=torch.uint8, device="cuda")
        dev[0] = torch.from_numpy(L2).cuda()
        dev[1] = torch.from_numpy(R2).cuda()
        torch.cuda.synchronize()
        fe.compute_batch(None, device_ptrs=[dev[0].data_ptr(), dev[1].data_ptr()], pitch=640)
        u, d = V.ComputeStereoMatches(fe, 0, fe, 1, 40.0, 435.2)
        eL, eR = orbo.Extractor(1000), orbo.Extractor(1000)
        kL, dL, _ = eL.compute(L2)
        kR, dR, _ = eR.compute(R2)
        wu, wd, _, _ = orbo.stereo(eL, eR, kL, dL, kR, dR, 40.0, 435.2)
        assert np.array_equal(u, wu) and np.array_equal(d, wd)
        dev.fill_(0)                      # the caller recycles its device images ...
        del dev
        torch.cuda.synchronize()
        res = fe.compute_batch([L, R])    # ... host pass again: graph replay
        u, d = V.ComputeStereoMatches(fe, 0, fe, 1, 40.0, 435.2)
        kL, dL, _ = eL.compute(L)
        kR, dR, _ = eR.compute(R)
        wu, wd, _, _ = orbo.stereo(eL, eR, kL, dL, kR, dR, 40.0, 435.2)
        _same_feats(res[0], (kL, dL))
        assert np.array_equal(u, wu) and np.array_equal(d, wd) and (wu >= 0).sum() > 50
        assert np.array_equal(fe.mvImagePyramid(0, slot=0), L)
    finally:
        fe.close()


def test_search_init_more_octave0_keypoints_than_the_context_quota():
    """keypoints of ANOTHER extractor configuration (2 levels: most points are octave 0) handed to a
    matcher context sized for 8 levels.  Host-keypoint entry: falls back to the host replay, result == oracle.
    Device-pointer entry: reports VSLAM_ERR_CAPACITY instead of silently truncating."""
    a, b = synth.make_frame(1241, 376, seed=7, step=0), synth.make_frame(1241, 376, seed=7, step=1)
    src = V.FExtractor(900, 1.2, 2, 20, 7, 1241, 376, max_batch=2)
    dst = V.FExtractor(1000, 1.2, 8, 20, 7, 1241, 376, max_batch=2)
    try:
        (k1, d1, _), (k2, d2, _) = src.compute_batch([a, b], (0, 1000))
        assert (k1["octave"] == 0).sum() > dst.features_per_level()[0] + 8 and len(k1) <= dst.cap
        _, pd1, _ = src.slot_buffers(0)
        _, pd2, _ = src.slot_buffers(1)
        m = V.FMatcher(dst, 0.9, True)
        nm, m12, pm = m.SearchForInitialization(k1, pd1, k2, pd2, np.stack([k1["x"], k1["y"]], 1), 100)
        wn, wm, wp = orbo.search_for_initialization(k1, d1, k2, d2, 1241, 376, window=100, nnratio=0.9)
        assert nm == wn and np.array_equal(m12, wm) and np.array_equal(pm, wp) and nm > 50
        p, c = src.slot_dev_ptrs(0), src.slot_dev_ptrs(1)
        m.search_init_dev_async([(p[0], p[1], p[2], c[0], c[1], c[2], 0)], 100)
        with pytest.raises(V.VslamError) as ei:
            m.search_init_dev_wait([len(k1)])
        assert ei.value.code == V.ERR_CAPACITY
    finally:
        src.close()
        dst.close()


def test_fast_gate_chains_contexts_without_changing_results():
    """vslam_fe_set_fast_gate: three contexts in a ring, each one's FAST launch waiting for the previous context's latest one;
    passes enqueued round-robin without host waits complete (no cycle in time) and equal the oracle"""
    import torch
    W, H, NF, B = 640, 360, 500, 4
    ctxs = [V.FExtractor(NF, 1.2, 8, 20, 7, W, H, max_batch=B) for _ in range(3)]
    try:
        for k, c in enumerate(ctxs):
            c.set_fast_gate(ctxs[(k - 1) % 3])
        frames = [synth.make_frame(W, H, seed=11, step=s) for s in range(B)]
        pitch = (W + 127) & ~127
        dev = torch.zeros((B, H, pitch), dtype=torch.uint8, device="cuda")
        for s in range(B):
            dev[s, :, :W] = torch.from_numpy(frames[s]).cuda()
        ptrs = [dev[s].data_ptr() for s in range(B)]
        torch.cuda.synchronize()
        e = orbo.Extractor(NF)
        want = [e.compute(f) for f in frames]
        nchecked = 0
        for t in range(9 + 2):  # three rounds over the ring; a pass is collected two enqueues later
            if t < 9:
                ctxs[t % 3].compute_batch_async(ptrs, pitch, (0, 0), to_host=True)
            if t >= 2:
                out = ctxs[(t - 2) % 3].wait()
                for s in range(B):
                    kk, dd, _ = out[s]
                    ok, od, _ = want[s]
                    assert len(kk) == len(ok) and all(np.array_equal(kk[f], ok[f]) for f in kk.dtype.names)
                    assert np.array_equal(dd, od)
                nchecked += 1
        assert nchecked == 9
        ctxs[0].set_fast_gate(None)
        out = ctxs[0].compute_batch(frames)
        assert len(out[0][0]) == len(want[0][0])
        # destroying a context that gates another one removes the gate: context 2 (gated by 1) keeps working
        ctxs[1].close()
        out = ctxs[2].compute_batch(frames)
        assert all(np.array_equal(out[s][1], want[s][1]) for s in range(B))
    finally:
        for c in ctxs:
            c.close()


def test_bench_verify_exchange_world1_collective_path():
    """bench.py --force-collective (RCCL, world size 1: the ring shift is a self-send through the same ncclSend/ncclRecv
    group) with the GPU-side proof of the exchange, for both exchange forms: the arrived slot equals the slot this GPU makes
    itself from the left neighbour's frame, and the matcher output agrees"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--force-collective", "--workload",
                        "kitti00_mono_1241x376_n1000", "--inputs", "device", "--no-cpu-baseline", "--steps", "4", "--warmup", "1",
                        "--min-seconds", "0.05"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(line) == 1, r.stdout[-2000:]
    d = json.loads(line[0])
    assert d["exchange_verified"] is True, d["exchange"]
    ver = d["exchange"]["verify"]
    assert set(ver) == {"ring", "allgather"}
    for mode, v in ver.items():
        assert v["verified"] and v["keypoints_compared_rank0"] > 500 and v["matches_rank0"] > 20, (mode, v)
    assert d["exchange"]["transport"] == "rccl"


def test_rccl_ring_and_allgather_world1_on_context_stream():
    """vslam_comm / vslam_exchange_ring at world size 1: ncclSend/ncclRecv to self inside a group, enqueued on the
    extractor's stream right behind k_pack_slots; the matcher then reads the RECEIVED buffer."""
    import torch
    fe = V.FExtractor(1000, 1.2, 8, 20, 7, 1241, 376, max_batch=4)
    comm = None
    try:
        comm = V.Comm(0, 0, 1, V.Comm.unique_id())
        frames = [synth.make_frame(1241, 376, step=s) for s in range(4)]
        res = fe.compute_batch(frames, (0, 1000))
        sb = fe.slot_bytes
        send = torch.zeros(4 * sb, dtype=torch.uint8, device="cuda")
        recv = torch.zeros(4 * sb, dtype=torch.uint8, device="cuda")
        allg = torch.zeros(4 * sb, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        fe.pack_slots(4, send.data_ptr(), sb, sync=False)
        comm.ring(fe, send.data_ptr(), recv.data_ptr(), 4 * sb)
        comm.allgather(fe, send.data_ptr(), allg.data_ptr(), 4 * sb)
        # frame s-1 (from the received buffer) -> frame s (own slot), all on fe's stream, no host sync in between
        jobs = []
        for s in range(1, 4):
            base = recv.data_ptr() + (s - 1) * sb
            c = fe.slot_dev_ptrs(s)
            jobs.append((base + 16, base + 16 + fe.cap * 28, base, c[0], c[1], c[2], 0))
        m = V.FMatcher(fe, 0.9, True)
        m.search_init_dev_async(jobs, 100)
        out = m.search_init_dev_wait([len(res[s - 1][0]) for s in range(1, 4)])
        for j, s in enumerate(range(1, 4)):
            wn, wm, _ = orbo.search_for_initialization(res[s - 1][0], res[s - 1][1], res[s][0], res[s][1], 1241, 376,
                                                       window=100, nnratio=0.9)
            assert out[j][0] == wn and np.array_equal(out[j][1], wm) and wn > 50
        torch.cuda.synchronize()
        assert torch.equal(send, recv) and torch.equal(send, allg)
        assert int(send[:4].cpu().numpy().view(np.int32)[0]) == len(res[0][0])
    finally:
        if comm:
            comm.close()
        fe.close()


def test_config5_stereo_1080p_n4000_matches_oracle():
    """BASELINE configs[4]: 1920x1080, 4000 features -- extraction AND ComputeStereoMatches at full size."""
    L, R = synth.make_stereo_pair(1920, 1080, seed=21)
    fe = V.FExtractor(4000, 1.2, 8, 20, 7, 1920, 1080, max_batch=2)
    try:
        res = fe.compute_batch([L, R])
        u, d = V.ComputeStereoMatches(fe, 0, fe, 1, BF, FX)
        (kL, dL), (kR, dR), wu, wd = _stereo_ref(L, R, 4000)
        _same_feats(res[0], (kL, dL), "L")
        _same_feats(res[1], (kR, dR), "R")
        assert np.array_equal(u, wu) and np.array_equal(d, wd) and (wu >= 0).sum() > 800
    finally:
        fe.close()


def test_config5_search_for_initialization_1080p_n4000():
    a, b = synth.make_frame(1920, 1080, seed=22, step=0), synth.make_frame(1920, 1080, seed=22, step=1)
    fe = V.FExtractor(4000, 1.2, 8, 20, 7, 1920, 1080, max_batch=2)
    try:
        (k1, d1, _), (k2, d2, _) = fe.compute_batch([a, b], (0, 1000))
        p, c = fe.slot_dev_ptrs(0), fe.slot_dev_ptrs(1)
        m = V.FMatcher(fe, 0.9, True)
        m.search_init_dev_async([(p[0], p[1], p[2], c[0], c[1], c[2], 0)], 100)
        out = m.search_init_dev_wait([len(k1)], want_prev=True)
        wn, wm, wp = orbo.search_for_initialization(k1, d1, k2, d2, 1920, 1080, window=100, nnratio=0.9)
        assert out[0][0] == wn and np.array_equal(out[0][1], wm) and np.array_equal(out[0][2], wp) and wn > 100
    finally:
        fe.close()


def test_kitti_mono_yaml_feature_count_n2000_pipeline():
    """config/KITTI00-Mono.yaml:20 says ORBextractor.nFeatures: 2000 (BASELINE configs[1] says 1000): the mono path
    (lapping area {0,1000}, frame.cpp:289) + SearchForInitialization at the YAML value, a chain of 4 frames."""
    frames = [synth.make_frame(1241, 376, seed=31, step=s) for s in range(4)]
    fe = V.FExtractor(2000, 1.2, 8, 20, 7, 1241, 376, max_batch=4)
    try:
        res = fe.compute_batch(frames, (0, 1000))
        e = orbo.Extractor(2000)
        for s in range(4):
            ko, do, mo = e.compute(frames[s], lap=(0, 1000))
            _same_feats(res[s], (ko, do), "frame %d" % s)
            assert res[s][2] == mo
        jobs = []
        for s in range(1, 4):
            p, c = fe.slot_dev_ptrs(s - 1), fe.slot_dev_ptrs(s)
            jobs.append((p[0], p[1], p[2], c[0], c[1], c[2], 0))
        m = V.FMatcher(fe, 0.9, True)
        m.search_init_dev_async(jobs, 100)
        out = m.search_init_dev_wait([len(res[s - 1][0]) for s in range(1, 4)])
        for j, s in enumerate(range(1, 4)):
            wn, wm, _ = orbo.search_for_initialization(res[s - 1][0], res[s - 1][1], res[s][0], res[s][1], 1241, 376,
                                                       window=100, nnratio=0.9)
            assert out[j][0] == wn and np.array_equal(out[j][1], wm) and wn > 100
    finally:
        fe.close()


def test_staged_upload_then_extract_equals_pinned_path():
    """vslam_fe_stage_images_async + VSLAM_IMGS_STAGED: the upload split from the extraction (bench.py enqueues it ahead
    of the cross-context waits); alternating with plain pinned passes on one context must not confuse the graph cache."""
    fe = V.FExtractor(1000, 1.2, 8, 20, 7, 1241, 376, max_batch=2)
    pin = V.PinnedImages(2, 376, 1241, 1241)
    try:
        e = orbo.Extractor(1000)
        for rep in range(3):
            imgs = [synth.make_frame(1241, 376, seed=40 + rep, step=s) for s in range(2)]
            for s in range(2):
                pin.array[s][:] = imgs[s]
            if rep == 1:
                fe.compute_batch_async(pin.ptrs, 1241, (0, 1000), where=V.IMGS_PINNED)
            else:
                fe.stage_images_async(pin.ptrs, 1241, V.IMGS_PINNED)
                fe.compute_batch_async(pin.ptrs, 1241, (0, 1000), where=V.IMGS_STAGED)
            res = fe.wait(copy=True)
            for s in range(2):
                ko, do, mo = e.compute(imgs[s], lap=(0, 1000))
                _same_feats(res[s], (ko, do), "staged %d/%d" % (rep, s))
    finally:
        pin.close()
        fe.close()


@pytest.mark.parametrize("env", [{}, {"oct_fine_depth": 1}, {"oct_fine_depth": 3}, {"octree_walk_kernel": 1}])
@pytest.mark.parametrize("cfg", [(1241, 376, 1000), (640, 480, 3000), (1920, 1080, 4000)])
def test_quadtree_fine_grid_depths_and_the_walk_kernel(env, cfg):
    """k_octree_v4 counts keys once into a fine grid, sorts them by fine cell and never walks them per pass; nodes finer
    than the grid (forced here with a depth of 1 or 3: almost every node) are resolved inside the kernel from the sorted
    keys of their cell.  The walk-per-pass kernel (k_octree_v2) stays selectable.  Every variant must give the oracle's
    keypoints."""
    w, h, nf = cfg
    imgs = [synth.make_frame(w, h, seed=50 + nf, step=s) for s in range(2)]
    fe = V.FExtractor(nf, 1.2, 8, 20, 7, w, h, max_batch=2, tuning=env)
    try:
        res = fe.compute_batch(imgs)
        e = orbo.Extractor(nf)
        for s in range(2):
            ko, do, _ = e.compute(imgs[s])
            _same_feats(res[s], (ko, do), "%s %s slot %d" % (env, cfg, s))
    finally:
        fe.close()


@pytest.mark.parametrize("env", [{"oct_fine_depth": 1}, {"oct_fine_depth": 3}])
def test_quadtree_shallow_grid_with_keys_in_global_memory(env):
    """batches of more than two images use the k_octree_v4 instantiation that re-reads its keys between the two walks
    (fewer VGPRs); with a shallow grid most nodes are split below it"""
    imgs = [synth.make_frame(1241, 376, seed=61, step=s) for s in range(4)]
    fe = V.FExtractor(1000, 1.2, 8, 20, 7, 1241, 376, max_batch=4, tuning=env)
    try:
        res = fe.compute_batch(imgs)
        e = orbo.Extractor(1000)
        for s in range(4):
            ko, do, _ = e.compute(imgs[s])
            _same_feats(res[s], (ko, do), "%s slot %d" % (env, s))
        prob, deep, masks = fe.octree_stats()
        assert prob == 4 * 8
        if env["oct_fine_depth"] == 1:  # 4 roots x 4 leaves against 217 wanted nodes: level 0 of every slot splits below the grid
            assert deep >= 4 and all(m & 1 for m in masks[:4])
    finally:
        fe.close()


def _digest(extra_env):
    import os
    import subprocess
    import sys
    env = dict(os.environ)
    env.update(extra_env)
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, os.path.join(here, "tools", "env_variant_check.py")], env=env, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("DIGEST ")]
    assert lines, out.stdout[-1000:]
    return lines[-1]


def test_switches_do_not_change_results():
    """Every A/B switch must deliver exactly what the default does -- extraction from device and from staged pinned
    images, and the device-resident init matcher.  Per-context switches go through vslam_fe_params.tuning in THIS process
    (several contexts with different settings side by side: nothing is cached process-wide any more); the environment
    defaults and the process-wide switches (wait mode, NUMA placement) get one child process each."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("env_variant_check", os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools",
                                                                                  "env_variant_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ref = mod.digest()
    for tuning in ({"oct_regkeys": 1}, {"oct_regkeys": 0}, {"si_queries_per_block": 8}, {"si_queries_per_block": 32},
                   {"fast_lds_pad": 4096}, {"h2d_route": 1}, {"h2d_route": 2}, {"oct_lds_budget_kb": 48},
                   {"oct_fine_depth": 2, "oct_regkeys": 1}, {"oct_fine_depth": 1, "oct_regkeys": 0}, {"d2h_route": 1}, {"d2h_route": 2}, {"graphs": 0},
                   {"pyramid_per_level": 1}, {"fast_threads": 256}, {"pyr_threads": 512}, {"blur_rows": 16}, {"init_topm": 16}, {"init_topm": 3},
                   {"d2h_route": 1, "copy_wgs": 4}, {"fast_threads": 64}, {"oct_threads": 1024}, {"oct_threads": 512}, {"oct_threads": 256},
                   {"oct_threads": 256, "oct_regkeys": 1}, {"oct_threads": 512, "oct_fine_depth": 1}):
        assert mod.digest(tuning) == ref, tuning
    for env in ({"VSLAM_WAIT": "spin", "VSLAM_NUMA": "0"}, {"VSLAM_D2H": "kernel", "VSLAM_OCT_REGKEYS": "1", "VSLAM_PYRAMID": "levels"}):
        assert _digest(env).split()[-1] == ref, env


def test_large_context_still_runs_after_a_small_one_was_created():
    """the maximum dynamic LDS of the quadtree kernels is an attribute of the FUNCTION, shared by every context of
    the process; creating a small context after a large one must not lower it (vk_octree_set_max_lds only ever raises)."""
    big = V.FExtractor(4000, 1.2, 8, 20, 7, 1920, 1080, max_batch=1)
    small = V.FExtractor(100, 1.2, 2, 20, 7, 128, 128, max_batch=1)
    try:
        im = synth.make_frame(1920, 1080, seed=71)
        k, d, _ = big.compute(im)
        ko, do, _ = orbo.Extractor(4000).compute(im)
        _same_feats((k, d), (ko, do), "1080p context after a 128x128 one")
        ks, ds, _ = small.compute(synth.make_frame(128, 128, seed=72))
        kso, dso, _ = orbo.Extractor(100, nlevels=2).compute(synth.make_frame(128, 128, seed=72))
        _same_feats((ks, ds), (kso, dso), "128x128 context")
    finally:
        small.close()
        big.close()


@pytest.mark.parametrize("seed,nf", [(1, 300), (2, 1500), (3, 6000)])
def test_quadtree_on_clustered_dot_images(seed, nf):
    """Adversarial candidate sets for the quadtree: isolated bright dots are FAST corners, so an image of dots IS a chosen
    key set -- tight clusters (dots two and three pixels apart: nodes must be split to depth 8-9, far below the kernel's
    fine grid), long runs on one row and one column, and a sparse background, with quotas below and above the number of
    candidates.  Every level must equal the oracle; the statistics must report the sub-grid splits."""
    rng = np.random.default_rng(seed)
    w, h = 800, 600
    img = np.full((h, w), 40, np.uint8)
    for _ in range(60):  # clusters: 3x3 .. 6x6 dots with a pitch of 2 or 3 px
        cx, cy = int(rng.integers(40, w - 60)), int(rng.integers(40, h - 60))
        pitch, k = int(rng.integers(2, 4)), int(rng.integers(3, 7))
        for a in range(k):
            for b in range(k):
                img[cy + pitch * a, cx + pitch * b] = int(rng.integers(200, 256))
    for x in range(30, w - 30, 2):  # a row and a column of dots: all keys of a node share a coordinate
        img[300, x] = 250
    for y in range(30, h - 30, 3):
        img[y, 401] = 251
    ys, xs = rng.integers(30, h - 30, 400), rng.integers(30, w - 30, 400)
    img[ys, xs] = rng.integers(180, 256, 400).astype(np.uint8)
    fe = V.FExtractor(nf, 1.2, 8, 20, 7, w, h, max_batch=3)
    try:
        res = fe.compute_batch([img, img[:, ::-1].copy(), img[::-1].copy()])
        prob, deep, masks = fe.octree_stats()
        e = orbo.Extractor(nf)
        for s, im in enumerate((img, img[:, ::-1].copy(), img[::-1].copy())):
            ko, do, _ = e.compute(im)
            _same_feats(res[s], (ko, do), "dots seed %d N=%d slot %d" % (seed, nf, s))
            if s == 0:
                assert len(e.candidates(0)) > 1000
        assert prob == 24
        if nf == 6000:  # quotas above the candidate counts: every node is split down to single keys, clusters far below the grid
            assert deep >= 3, (prob, deep, masks[:3])
    finally:
        fe.close()
