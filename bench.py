#!/usr/bin/env python3
"""bench.py -- frames/sec of the ORB extract + match front-end on N MI355X (one process per GPU).

    python bench.py --gpus 1 --steps K --warmup W            # N=1, the driver's default call
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one batch of synthetic frames per rank: `--batch` frames go
through extraction (pyramid, FAST+NMS per cell, quadtree, orientation, blur, rBRIEF) and each frame is
matched against its predecessor in video order (mono: FMatcher::SearchForInitialization, window 100;
stereo: Frame::ComputeStereoMatches L<->R).  Frames are dealt round-robin over ranks; the only collective
is one ring shift of packed result slots per step (mono workload, N>1; every predecessor lives on rank-1).  Input frames are resident in
HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# Several extractor contexts (HIP streams) are kept in flight; with the runtime's default of 4 hardware
# queues two of them end up behind each other on one queue (measured: 29k -> 38k frames/s with 8).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md

WORKLOADS = {
    # BASELINE.json configs[1]: KITTI-00 mono, 8-level pyramid, 1000 features/frame
    "kitti00_mono_1241x376_n1000": dict(w=1241, h=376, nf=1000, stereo=False),
    # configs[2]: KITTI-00 stereo (YAML: 2000 features), L<->R Hamming match
    "kitti00_stereo_1241x376_n2000": dict(w=1241, h=376, nf=2000, stereo=True),
    # configs[4]: synthetic 1920x1080 stream, 4000 features/frame
    "synthetic_stereo_1920x1080_n4000": dict(w=1920, h=1080, nf=4000, stereo=True),
    # SURVEY.md 8(f) rank 1, the per-frame tracking front-end of TrackWithMotionModel: stereo frame construction +
    # UnprojectStereo + SearchByProjection(frame, previous frame); every rank follows its own sequence
    "kitti00_stereo_track_1241x376_n2000": dict(w=1241, h=376, nf=2000, stereo=True, track=True),
}
BF, FX = 386.1448, 718.856  # config/KITTI00-Stereo.yaml Camera.bf, Camera.fx
FY, CX, CY = 718.856, 607.1928, 185.2157
TRACK_Z = 12.0  # the synthetic scene moves (+3,+1) px per frame; as a camera translation at this depth


def level_pixels(fe):
    return [fe.level_size(l)[0] * fe.level_size(l)[1] for l in range(fe.nlevels)]


def algorithmic_bytes(fe, nf):
    """SURVEY.md 8(d) per-image algorithmic bytes of each kernel stage."""
    px = level_pixels(fe)
    P = sum(px)
    return dict(pyramid=(P - px[-1]) + (P - px[0]), fast=P, blur=2 * P, describe=2 * P + 60 * nf, total_px=P)


def pmc_traffic(kernel, cfg, batch):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc summary (tools/collect_pmc.sh,
    separate FETCH_SIZE / WRITE_SIZE passes).  gfx950's FETCH_SIZE counts 64 B per 128-B request, i.e. half of
    a coalesced stream (MI355X_MICROARCH.md 'HBM'): calibrated here on the FAST kernel, whose unique input is
    the pyramid pixels (1.44 MB per image) and whose raw FETCH_SIZE reads half of that -> the factor 2 is applied.
    Only valid for the geometry/batch the profile was taken with; otherwise None."""
    name = "r01g_pmc_traffic_kitti_b%d.json" % batch
    path = os.path.join(ROOT, "profiles", name)
    if not (os.path.exists(path) and cfg["w"] == 1241 and cfg["h"] == 376):
        return None
    if kernel.startswith("k_orient_describe") and cfg["nf"] != 1000:
        return None  # the profile was taken with 1000 features; only this kernel's traffic depends on that
    k = json.load(open(path))["kernels"].get(kernel.split("(")[0])
    if not k or "fetch_bytes_raw" not in k or "write_bytes" not in k:
        return None
    return {"bytes": 2 * k["fetch_bytes_raw"] + k["write_bytes"], "fetch_size_raw_bytes": k["fetch_bytes_raw"],
            "write_size_bytes": k["write_bytes"], "source": "profiles/" + name}


def cpu_baseline(cfg, seconds=12.0):
    """The oracle (a port of the reference's CPU path, oracle/) timed on this box's host cores: the same
    workload on a bounded sample of frames, reference-faithful threading (1 thread per image; the two
    images of a stereo frame on 2 threads, frame.cpp:107-108)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import orbo
    from vi_slam_amd import synth
    w, h, nf, stereo = cfg["w"], cfg["h"], cfg["nf"], cfg["stereo"]
    nsample = 6
    t_end = time.time() + seconds
    frames = 0
    t0 = time.time()
    if stereo:
        pairs = [synth.make_stereo_pair(w, h, step=s) for s in range(nsample)]
        eL, eR = orbo.Extractor(nf), orbo.Extractor(nf)
        pool = ThreadPoolExecutor(2)
        prev_track = None
        t0 = time.time()
        while time.time() < t_end:
            L, R = pairs[frames % nsample]
            fl = pool.submit(eL.compute, L)
            fr = pool.submit(eR.compute, R)
            (kL, dL, _), (kR, dR, _) = fl.result(), fr.result()
            uR, depth = orbo.stereo(eL, eR, kL, dL, kR, dR, BF, FX)[:2]
            if cfg.get("track"):
                T0 = np.hstack([np.eye(3), np.zeros((3, 1))]).astype(np.float32)
                x3, has = orbo.unproject_stereo(kL, depth, T0, CX, CY, 1.0 / FX, 1.0 / FY)
                if prev_track is not None:
                    pk, pd, px, pf = prev_track
                    Tcw = np.hstack([np.eye(3), np.array([[3.0 / FX * TRACK_Z], [1.0 / FY * TRACK_Z], [0.0]])])
                    orbo.search_by_projection_frame(Tcw, T0, (FX, FY, CX, CY, BF, BF / FX), 15, pk, pf * 3, px, pd, kL, dL,
                                                    uR, eL.tables()["scale"], w, h)
                prev_track = (kL.copy(), dL.copy(), x3, has)
            frames += 1
        cores = 2
        sample = "%d synthetic stereo frames %dx%d, %d features: 2 threads extract L/R + ComputeStereoMatches%s" % (
            frames, w, h, nf, " + UnprojectStereo + SearchByProjection(prev frame)" if cfg.get("track") else "")
    else:
        imgs = [synth.make_frame(w, h, step=s) for s in range(nsample)]
        e = orbo.Extractor(nf)
        prev = None
        t0 = time.time()
        while time.time() < t_end:
            k, d, _ = e.compute(imgs[frames % nsample], lap=(0, 1000))
            if prev is not None:
                orbo.search_for_initialization(prev[0], prev[1], k, d, w, h, window=100, nnratio=0.9)
            prev = (k, d)
            frames += 1
        cores = 1
        sample = "%d synthetic mono frames %dx%d, %d features: extract + SearchForInitialization(prev)" % (
            frames, w, h, nf)
    dt = time.time() - t0
    out = {"value": frames / dt, "unit": "frames/s", "cores": cores, "kind": "port", "sample": sample}
    out["all_cores"] = cpu_baseline_all_cores(cfg)
    return out


def cpu_baseline_all_cores(cfg, seconds=8.0):
    """SURVEY.md 8(d) variant (ii): every host core of the box, frame-parallel (one oracle extractor per thread,
    each thread following its own frame sequence; ctypes releases the GIL).  Reported beside the
    reference-faithful figure, never as the headline."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import orbo
    from vi_slam_amd import synth
    w, h, nf, stereo = cfg["w"], cfg["h"], cfg["nf"], cfg["stereo"]
    ncores = max(1, min(os.cpu_count() or 1, 64))
    nsample = 3
    if stereo:
        data = [synth.make_stereo_pair(w, h, step=s) for s in range(nsample)]
    else:
        data = [synth.make_frame(w, h, step=s) for s in range(nsample)]
    t_end = time.time() + seconds

    def worker(_):
        e, e2 = orbo.Extractor(nf), orbo.Extractor(nf)
        prev, n = None, 0
        while time.time() < t_end:
            if stereo:
                L, R = data[n % nsample]
                kL, dL, _ = e.compute(L)
                kR, dR, _ = e2.compute(R)
                orbo.stereo(e, e2, kL, dL, kR, dR, BF, FX)
            else:
                k, d, _ = e.compute(data[n % nsample], lap=(0, 1000))
                if prev is not None:
                    orbo.search_for_initialization(prev[0], prev[1], k, d, w, h, window=100, nnratio=0.9)
                prev = (k, d)
            n += 1
        return n

    t0 = time.time()
    with ThreadPoolExecutor(ncores) as pool:
        total = sum(pool.map(worker, range(ncores)))
    dt = time.time() - t0
    return {"value": total / dt, "unit": "frames/s", "cores": ncores,
            "sample": "%d frames on %d threads, one sequence per thread" % (total, ncores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="kitti00_mono_1241x376_n1000", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=32, help="frames (mono) or images (stereo: L,R,L,R..) per rank per step")
    ap.add_argument("--inflight", type=int, default=4, help="extractor contexts (HIP streams) in flight per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo (slots staged through host memory) only exists to rehearse the N>1 path on one GPU")
    ap.add_argument("--same-gpu", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--exchange", default="shift", choices=["shift", "allgather"],
                    help="N>1 exchange step: ring shift (all_to_all_single, one non-empty split) or all-gather")
    ap.add_argument("--force-collective", action="store_true",
                    help="rehearsal: take the N>1 code path (pack + ring shift on the extractor stream) at any world size")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import vi_slam_amd as V
    from vi_slam_amd import dist as vd
    from vi_slam_amd import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world),
                  file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    if args.same_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1 or args.force_collective:
        if world == 1:  # --force-collective without a launcher: a one-rank group
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    multi = world > 1 or (args.force_collective and dist.is_initialized())
    cfg = WORKLOADS[args.workload]
    w, h, nf, stereo = cfg["w"], cfg["h"], cfg["nf"], cfg["stereo"]
    track = bool(cfg.get("track"))
    B = args.batch
    if stereo and B % 2:
        B += 1
    if track:
        B = min(B, 32)  # one SearchByProjection pass takes up to 16 frame pairs
    NCTX = max(3, args.inflight)  # extractor contexts (streams) kept in flight per GPU
    ctxs = [V.FExtractor(nf, 1.2, 8, 20, 7, w, h, device=local_rank, max_batch=B) for _ in range(NCTX)]
    fe = ctxs[0]
    matchers = [V.FMatcher(c, 0.9, True) for c in ctxs]
    lap = (0, 0) if stereo else (0, 1000)  # frame.cpp:107-108 vs :289

    # ---- synthetic frames, resident in HBM before the timed region
    pitch = (w + 127) & ~127
    dev_frames = torch.zeros((B, h, pitch), dtype=torch.uint8, device="cuda")
    for s in range(B):
        if track:  # one contiguous sequence per rank
            fr = synth.make_frame(w, h, seed=20250215 + rank, step=s // 2, right=bool(s & 1))
        elif stereo:
            fr = synth.make_frame(w, h, step=(s // 2) * world + rank, right=bool(s & 1))
        else:
            fr = synth.make_frame(w, h, step=vd.global_frame(rank, s, world))
        dev_frames[s, :, :w] = torch.from_numpy(fr).cuda()
    import ctypes
    ptrs = (ctypes.c_void_p * B)(*[dev_frames[s].data_ptr() for s in range(B)])
    slot_bytes = fe.slot_bytes
    desc_off = 16 + fe.cap * 28
    packed = [torch.zeros(B * slot_bytes, dtype=torch.uint8, device="cuda") for _ in range(NCTX if multi else 0)]
    # packed slots of the LEFT neighbour (every predecessor lives there, vi_slam_amd/dist.py), one buffer per context
    from_left = [torch.zeros(B * slot_bytes, dtype=torch.uint8, device="cuda") for _ in range(NCTX if multi else 0)]
    ring_in, ring_out = [0] * world, [0] * world
    ring_in[(rank + 1) % world] = B * slot_bytes
    ring_out[(rank - 1) % world] = B * slot_bytes
    # Safety net: should this RCCL build reject the uneven all_to_all_single, the exchange falls back to an all-gather
    # (world times the volume, same result); the left neighbour's block is then read out of the gathered buffer.
    xchg = {"mode": args.exchange, "gathered": None}

    def left_view(k):
        """Tensor holding the left neighbour's packed slots of context k."""
        if xchg["mode"] == "shift":
            return from_left[k]
        lo = ((rank - 1) % world) * B * slot_bytes
        return xchg["gathered"][k][lo:lo + B * slot_bytes]

    def exchange(k):
        if xchg["mode"] == "shift":
            try:
                dist.all_to_all_single(from_left[k], packed[k], output_split_sizes=ring_out, input_split_sizes=ring_in)
                return
            except RuntimeError as e:  # pragma: no cover - depends on the collective library
                if job_cache:
                    raise
                print("bench.py: all_to_all_single failed (%s); falling back to all_gather" % str(e).splitlines()[0],
                      file=sys.stderr)
                xchg["mode"] = "allgather"
        if xchg["gathered"] is None:
            xchg["gathered"] = [torch.zeros(world * B * slot_bytes, dtype=torch.uint8, device="cuda") for _ in range(NCTX)]
        dist.all_gather_into_tensor(xchg["gathered"][k], packed[k])
    ext_streams = [torch.cuda.ExternalStream(c.stream()) for c in ctxs] if multi else []
    state = {"matches": 0}
    torch.cuda.synchronize()

    def slot_ptrs_in(buf, s):
        """(kps, desc, count) device addresses of slot s inside a packed buffer."""
        base = buf.data_ptr() + s * slot_bytes
        return base + 16, base + desc_off, base

    job_cache = {}
    track_Twc = [np.hstack([np.eye(3), np.zeros((3, 1))]).astype(np.float32)] * (B // 2)
    track_cam = (CX, CY, float(np.float32(1.0) / np.float32(FX)), float(np.float32(1.0) / np.float32(FY)))
    track_Tcw = np.hstack([np.eye(3), np.array([[3.0 / FX * TRACK_Z], [1.0 / FY * TRACK_Z], [0.0]])]).astype(np.float32)

    def enqueue(t):
        """Enqueue step t completely -- extraction, (N>1) pack + ring shift, matcher -- without waiting for
        anything on the host: every pointer is a fixed device address and the counts stay in HBM."""
        c, k = ctxs[t % NCTX], t % NCTX
        nxt, prv = ctxs[(t + 1) % NCTX], ctxs[(t - 1) % NCTX]
        if track:
            npairs = B // 2
            c.event_wait(nxt, 1)  # nxt's matcher (step t-NCTX+1) read our last frame: it must finish first
            c.frame_stereo_async(ptrs, pitch, BF, FX)
            c.stereo_points_async(track_Twc, track_cam)  # UnprojectStereo of every left keypoint, own camera = world
            c.event_record(0)
            ck = (k, t == 0)
            if ck not in job_cache:
                jobs = []
                for j in range(npairs):
                    if j == 0 and t == 0:
                        continue
                    lc, lj = (c, j - 1) if j else (prv, npairs - 1)
                    lk, ld, ln = lc.slot_dev_ptrs(2 * lj)
                    x3, fl, _, _ = lc.stereo_points_buffers(lj, with_stereo=False)
                    ckp, cd, cn = c.slot_dev_ptrs(2 * j)
                    ur = c.stereo_points_buffers(j)[2]
                    jobs.append(dict(Tcw=track_Tcw, cam=(FX, FY, CX, CY, BF), th=15, forward=0, backward=0, img=(w, h),
                                     last_kps=lk, n_last=ln, last_flags=fl, last_x3dw=x3, mp_desc=ld, cur_kps=ckp,
                                     cur_desc=cd, n_cur=cn, cur_u_right=ur))
                job_cache[ck] = (V.FMatcher.make_sbp_jobs(jobs, True), len(jobs))
            arr, njobs = job_cache[ck]
            if t > 0:
                c.event_wait(prv, 0)
            matchers[k].search_by_projection_dev_async(arr)
            c.event_record(1)
            state.setdefault("njobs", {})[t] = njobs
            return
        if stereo:
            c.frame_stereo_async(ptrs, pitch, BF, FX)
            return
        c.event_wait(nxt, 1)  # nxt's matcher (step t-NCTX+1) read our last results: it must finish first
        c.compute_batch_async(ptrs, pitch, lap)
        if multi:
            if args.dist_backend == "nccl":
                c.pack_slots(B, packed[k].data_ptr(), slot_bytes, sync=False)  # one kernel on c's stream
                with torch.cuda.stream(ext_streams[k]):  # the ring shift is ordered on c's own stream
                    exchange(k)
            else:  # gloo rehearsal: staged through the host, fully synchronous
                c.pack_slots(B, packed[k].data_ptr(), slot_bytes, sync=True)
                vd.shift_slots(packed[k], from_left[k])
                torch.cuda.synchronize()
        c.event_record(0)  # step t's results (own, and the left neighbour's) are complete
        # the device addresses are fixed per context, so the job array is built once (t == 0 has no predecessor
        # for slot 0 and is built separately)
        ck = (k, t == 0)
        if ck not in job_cache:
            jobs = []
            uses_prev_step = False
            for s in range(B):
                pr, ps, prev_step = vd.predecessor(rank, s, world, B)
                if prev_step:
                    if t == 0:
                        continue
                    p = slot_ptrs_in(left_view((t - 1) % NCTX), B - 1) if multi else prv.slot_dev_ptrs(B - 1)
                    uses_prev_step = True
                elif not multi:
                    p = c.slot_dev_ptrs(ps)
                else:
                    p = slot_ptrs_in(left_view(k), ps)  # pr == (rank - 1) % world always
                q = c.slot_dev_ptrs(s)
                jobs.append((p[0], p[1], p[2], q[0], q[1], q[2], 0))
            job_cache[ck] = (V.FMatcher.make_init_jobs(jobs) if jobs else None, len(jobs), uses_prev_step)
        arr, njobs, uses_prev_step = job_cache[ck]
        if uses_prev_step:
            c.event_wait(prv, 0)  # the previous step's results (not its matcher)
        if njobs:
            matchers[k].search_init_dev_async(arr, 100, (w, h))
        c.event_record(1)  # matcher(t) complete
        state.setdefault("njobs", {})[t] = njobs

    def collect(t):
        """One host wait per step delivers keypoints, descriptors and (mono) the matches."""
        c = ctxs[t % NCTX]
        t_a = time.perf_counter()
        if stereo:
            feats, st = c.frame_stereo_wait()
            state["matches"] = sum(int((u >= 0).sum()) for u, _ in st)
            t_b = time.perf_counter()
            if track:
                nj = state["njobs"].pop(t)
                ncur = [len(feats[2 * j][0]) for j in range(B // 2 - nj, B // 2)]
                out = matchers[t % NCTX].search_by_projection_dev_wait(ncur)
                state["track_matches"] = sum(o[0] for o in out)
        else:
            res = c.wait()
            t_b = time.perf_counter()
            nj = state["njobs"].pop(t)
            if nj:
                out = matchers[t % NCTX].search_init_dev_wait([fe.cap] * nj)
                state["matches"] = sum(o[0] for o in out)
        hs = state.setdefault("host_s", [0.0, 0.0])
        hs[0] += t_b - t_a
        hs[1] += time.perf_counter() - t_b

    def run(nsteps):
        for t in range(nsteps + NCTX - 1):
            if t < nsteps:
                t_e = time.perf_counter()
                enqueue(t)
                state["enq_s"] = state.get("enq_s", 0.0) + time.perf_counter() - t_e
            if 0 <= t - (NCTX - 1) < nsteps:
                collect(t - (NCTX - 1))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run(args.warmup)
    state.pop("host_s", None)
    state.pop("enq_s", None)
    for c in ctxs:
        c.set_profiling(True)
    barrier()
    t0 = time.perf_counter()
    run(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    prof = {}
    for c in ctxs:
        for k, v in c.get_profile().items():
            prof[k] = prof.get(k, 0) + v
        c.set_profiling(False)
    # After the timed region: the same extraction pass with NOTHING else on the GPU (one context, no matcher), so the
    # dominant kernel's duration is also known without the other streams' workgroups inside its begin-to-end span.
    alone = {}
    if rank == 0:
        torch.cuda.synchronize()
        c0 = ctxs[0]
        c0.set_profiling(True)
        for _ in range(10):
            if stereo:
                c0.frame_stereo_async(ptrs, pitch, BF, FX)
                c0.frame_stereo_wait()
            else:
                c0.compute_batch_async(ptrs, pitch, lap)
                c0.wait()
        alone = c0.get_profile()
        c0.set_profiling(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    frames_per_step = (B // 2 if stereo else B) * world
    value = frames_per_step * args.steps / dt

    if rank == 0:
        ab = algorithmic_bytes(fe, nf)
        nb = max(prof["batches"], 1)
        stage_ms = {"pyramid": prof["pyramid_ms"] / nb, "fast": prof["fast_ms"] / nb, "blur": prof["blur_ms"] / nb,
                    "describe": prof["describe_ms"] / nb, "octree": prof["octree_ms"] / nb}
        # dominant streaming KERNEL: per launch (the pyramid stage is 7 launches; the quadtree moves no image bytes)
        per_launch = {k: (v / 7.0 if k == "pyramid" else v) for k, v in stage_ms.items() if k != "octree"}
        dom = max(per_launch, key=per_launch.get)
        kernel = {"pyramid": "k_resize_level_v2(x7)", "fast": "k_fast_cells_v3", "blur": "k_blur7_v2",
                  "describe": "k_orient_describe_dev"}[dom]
        tr = pmc_traffic(kernel, cfg, B)
        bytes_per_launch = ab[dom] * B / (7.0 if dom == "pyramid" else 1.0)
        achieved = bytes_per_launch / (per_launch[dom] * 1e-3) / 1e9 if per_launch[dom] > 0 else 0.0
        pipe_bytes = ab["pyramid"] + ab["fast"] + ab["blur"] + ab["describe"]
        images_per_s = B * args.steps / dt  # this rank's images (stereo: two per frame)
        uncontended = None
        if alone.get("batches"):
            a_ms = alone[{"pyramid": "pyramid_ms", "fast": "fast_ms", "blur": "blur_ms", "describe": "describe_ms"}[dom]]
            a_ms = a_ms / alone["batches"] / (7.0 if dom == "pyramid" else 1.0)
            if a_ms > 0:
                a_gbs = bytes_per_launch / (a_ms * 1e-3) / 1e9
                uncontended = {"note": "same kernel, same batch, nothing else on the GPU (10 launches after the timed region)",
                               "avg_launch_ms": a_ms, "achieved": a_gbs, "frac": a_gbs / HBM_PEAK_GBS}
        out = {
            "metric": "frames/sec ORB extract+match",
            "value": value,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": args.workload, "frames_per_step_per_gpu": B // 2 if stereo else B,
                       "images_per_step_per_gpu": B, "nfeatures": nf, "nlevels": 8, "scale_factor": 1.2,
                       "match": ("ComputeStereoMatches L<->R + UnprojectStereo + SearchByProjection(frame, previous frame), th 15, on device"
                                 if track else "ComputeStereoMatches L<->R") if stereo else "SearchForInitialization(prev frame), window 100, on device",
                       "sharding": "frames round-robin over ranks; one ring shift of result slots per step (all_to_all_single, on the extractor stream)"
                       if not stereo else ("one independent stereo sequence per rank, no collective" if track else
                                           "stereo frames independent per rank, no collective"),
                       "contexts_in_flight": NCTX, "matches_last_step_rank0": state["matches"],
                       "track_matches_last_step_rank0": state.get("track_matches"),
                       "host_ms_per_step": {k: v / args.steps * 1e3 for k, v in
                                            zip(("enqueue", "wait_step", "fetch_matches"),
                                                [state.get("enq_s", 0.0)] + state.get("host_s", [0, 0]))}},
            "roofline": {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": tr["bytes"] if tr else None,
                         "traffic_detail": tr,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "avg_launch_ms": per_launch[dom], "stage_ms_per_batch": stage_ms,
                         "uncontended": uncontended,
                         # all extraction stages together, algorithmic bytes per image x images/s
                         "pipeline": {"algorithmic_bytes_per_image": pipe_bytes,
                                      "achieved": pipe_bytes * images_per_s / 1e9,
                                      "frac": pipe_bytes * images_per_s / 1e9 / HBM_PEAK_GBS}},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(out))
    for c in ctxs:
        c.close()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
