#!/usr/bin/env python3
"""bench.py -- frames/sec of the ORB extract + match front-end on N MI355X (one process per GPU).

    python bench.py --gpus 1 --steps K --warmup W            # N=1, the driver's default call
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one batch of synthetic frames per rank: `--batch` frames go
through extraction (pyramid, FAST+NMS per cell, quadtree, orientation, blur, rBRIEF) and each frame is
matched against its predecessor in video order (mono: FMatcher::SearchForInitialization, window 100;
stereo: Frame::ComputeStereoMatches L<->R).  Frames are dealt round-robin over ranks; the only collective
is one all-gather of packed result slots per step (mono workload, N>1).  Input frames are resident in
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

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md

WORKLOADS = {
    # BASELINE.json configs[1]: KITTI-00 mono, 8-level pyramid, 1000 features/frame
    "kitti00_mono_1241x376_n1000": dict(w=1241, h=376, nf=1000, stereo=False),
    # configs[2]: KITTI-00 stereo (YAML: 2000 features), L<->R Hamming match
    "kitti00_stereo_1241x376_n2000": dict(w=1241, h=376, nf=2000, stereo=True),
    # configs[4]: synthetic 1920x1080 stream, 4000 features/frame
    "synthetic_stereo_1920x1080_n4000": dict(w=1920, h=1080, nf=4000, stereo=True),
}
BF, FX = 386.1448, 718.856  # config/KITTI00-Stereo.yaml Camera.bf, Camera.fx


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
    a coalesced stream (MI355X_MICROARCH.md 'HBM'): calibrated here on k_fast_cells_v2, whose unique input is
    the 23.1 MB of pyramid pixels and whose raw FETCH_SIZE reads 11.1 MB -> the factor 2 is applied.
    Only valid for the geometry/batch the profile was taken with; otherwise None."""
    path = os.path.join(ROOT, "profiles", "r01c_pmc_traffic_kitti_n2000_b16.json")
    if not (os.path.exists(path) and cfg["w"] == 1241 and cfg["h"] == 376 and batch == 16):
        return None
    k = json.load(open(path))["kernels"].get(kernel.split("(")[0])
    if not k or "fetch_bytes_raw" not in k or "write_bytes" not in k:
        return None
    return {"bytes": 2 * k["fetch_bytes_raw"] + k["write_bytes"], "fetch_size_raw_bytes": k["fetch_bytes_raw"],
            "write_size_bytes": k["write_bytes"], "source": "profiles/r01c_pmc_traffic_kitti_n2000_b16.json"}


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
        t0 = time.time()
        while time.time() < t_end:
            L, R = pairs[frames % nsample]
            fl = pool.submit(eL.compute, L)
            fr = pool.submit(eR.compute, R)
            (kL, dL, _), (kR, dR, _) = fl.result(), fr.result()
            orbo.stereo(eL, eR, kL, dL, kR, dR, BF, FX)
            frames += 1
        cores = 2
        sample = "%d synthetic stereo frames %dx%d, %d features: 2 threads extract L/R + ComputeStereoMatches" % (
            frames, w, h, nf)
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
    return {"value": frames / dt, "unit": "frames/s", "cores": cores, "kind": "port", "sample": sample}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="kitti00_mono_1241x376_n1000", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=16, help="frames (mono) or images (stereo: L,R,L,R..) per rank per step")
    ap.add_argument("--inflight", type=int, default=4, help="extractor contexts (HIP streams) in flight per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    cfg = WORKLOADS[args.workload]
    w, h, nf, stereo = cfg["w"], cfg["h"], cfg["nf"], cfg["stereo"]
    B = args.batch
    if stereo and B % 2:
        B += 1
    NCTX = max(3, args.inflight)  # extractor contexts (streams) kept in flight per GPU
    ctxs = [V.FExtractor(nf, 1.2, 8, 20, 7, w, h, device=local_rank, max_batch=B) for _ in range(NCTX)]
    fe = ctxs[0]
    matchers = [V.FMatcher(c, 0.9, True) for c in ctxs]
    lap = (0, 0) if stereo else (0, 1000)  # frame.cpp:107-108 vs :289

    # ---- synthetic frames, resident in HBM before the timed region
    pitch = (w + 127) & ~127
    dev_frames = torch.zeros((B, h, pitch), dtype=torch.uint8, device="cuda")
    for s in range(B):
        if stereo:
            fr = synth.make_frame(w, h, step=(s // 2) * world + rank, right=bool(s & 1))
        else:
            fr = synth.make_frame(w, h, step=vd.global_frame(rank, s, world))
        dev_frames[s, :, :w] = torch.from_numpy(fr).cuda()
    ptrs = [dev_frames[s].data_ptr() for s in range(B)]
    slot_bytes = fe.slot_bytes
    desc_off = 16 + fe.cap * 28
    packed = torch.zeros(B * slot_bytes, dtype=torch.uint8, device="cuda")
    gathered = torch.zeros(world * B * slot_bytes, dtype=torch.uint8, device="cuda")
    carry = torch.zeros(slot_bytes, dtype=torch.uint8, device="cuda")
    state = {"carry_kps": None, "matches": 0}
    torch.cuda.synchronize()

    def slot_host_kps(view):
        hdr = view[:16].cpu().numpy().view(np.int32)
        n = int(hdr[0])
        return view[16:16 + n * 28].cpu().numpy().view(V.KP_DTYPE)

    def enqueue(t):
        c = ctxs[t % NCTX]
        if stereo:
            c.frame_stereo_async(ptrs, pitch, BF, FX)
        else:
            c.compute_batch_async(ptrs, pitch, lap)

    def mid(t):
        """Step t's extraction is done: results to the host; (mono) enqueue the frame-to-frame matcher for
        every frame of the step -- on the GPU, fed with device pointers only."""
        c = ctxs[t % NCTX]
        if stereo:
            feats, st = c.frame_stereo_wait()
            state["matches"] = sum(int((u >= 0).sum()) for u, _ in st)
            return
        t_a = time.perf_counter()
        res = c.wait()
        t_b = time.perf_counter()
        if world > 1:
            c.pack_slots(B, packed.data_ptr(), slot_bytes)
            vd.exchange_slots(packed, gathered)
            torch.cuda.synchronize()
        jobs, n1 = [], []
        for s in range(B):
            pr, ps, prev_step = vd.predecessor(rank, s, world, B)
            if prev_step:
                if not state.get("have_carry"):
                    continue
                base = carry.data_ptr()
                prev_ptrs = (base + 16, base + desc_off, base)
                n_prev = state["carry_n"]
            elif world == 1:
                prev_ptrs = c.slot_dev_ptrs(ps)
                n_prev = len(res[ps][0])
            else:
                base = vd.slot_view(gathered, pr, ps, B, slot_bytes).data_ptr()
                prev_ptrs = (base + 16, base + desc_off, base)
                n_prev = fe.cap  # count lives in the packed header; deliver up to capacity
            cur = c.slot_dev_ptrs(s)
            jobs.append((prev_ptrs[0], prev_ptrs[1], prev_ptrs[2], cur[0], cur[1], cur[2], 0))
            n1.append(n_prev)
        if jobs:
            if world == 1 and t > 0:
                c.wait_for(ctxs[(t - 1) % NCTX])  # the previous step's carry copy (its stream) precedes our matcher
            matchers[t % NCTX].search_init_dev_async(jobs, 100, (w, h))
        state["pending_n1"] = state.get("pending_n1", {})
        state["pending_n1"][t] = n1
        # the last frame of this step precedes the first frame of the next one: keep it in `carry`
        if world == 1:
            c.pack_slots(1, carry.data_ptr(), slot_bytes, first=B - 1, sync=False)
            state["carry_n"] = len(res[B - 1][0])
        else:
            carry.copy_(vd.slot_view(gathered, world - 1, B - 1, B, slot_bytes))
            torch.cuda.synchronize()
            state["carry_n"] = fe.cap
        state["have_carry"] = True
        t_c = time.perf_counter()
        hs = state.setdefault("host_s", [0.0, 0.0, 0.0])
        hs[0] += t_b - t_a
        hs[1] += t_c - t_b

    def fin(t):
        """Collect step t's matches (synchronises that context's stream, which also orders its carry copy
        before the next step's matcher reads it)."""
        if stereo:
            return
        t_a = time.perf_counter()
        n1 = state["pending_n1"].pop(t)
        if n1:
            out = matchers[t % NCTX].search_init_dev_wait(n1)
            state["matches"] = sum(o[0] for o in out)
        state.setdefault("host_s", [0.0, 0.0, 0.0])[2] += time.perf_counter() - t_a

    def run(nsteps):
        # enqueue(t) .. mid(t - d1) .. fin(t - d2): a context is busy from enqueue to fin, NCTX = d2 + 1 of them
        d1, d2 = (NCTX - 1, NCTX - 1) if stereo else (2, NCTX - 1)
        for t in range(nsteps + d2):
            if t < nsteps:
                enqueue(t)
            if 0 <= t - d2 < nsteps:
                fin(t - d2)
            if 0 <= t - d1 < nsteps:
                mid(t - d1)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run(args.warmup)
    state.pop("host_s", None)
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
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    frames_per_step = (B // 2 if stereo else B) * world
    value = frames_per_step * args.steps / dt

    if rank == 0:
        ab = algorithmic_bytes(fe, nf)
        nb = max(prof["batches"], 1)
        stage_ms = {"pyramid": prof["pyramid_ms"] / nb, "fast": prof["fast_ms"] / nb, "blur": prof["blur_ms"] / nb,
                    "describe": prof["describe_ms"] / nb, "octree": prof["octree_ms"] / nb}
        streaming = {k: v for k, v in stage_ms.items() if k != "octree"}  # the quadtree moves no image bytes
        dom = max(streaming, key=streaming.get)
        kernel = {"pyramid": "k_resize_level(x7)", "fast": "k_fast_cells_v2", "blur": "k_blur7_v2",
                  "describe": "k_orient_describe_dev"}[dom]
        tr = pmc_traffic(kernel, cfg, B)
        bytes_per_launch = ab[dom] * B
        achieved = bytes_per_launch / (stage_ms[dom] * 1e-3) / 1e9 if stage_ms[dom] > 0 else 0.0
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
                       "match": "ComputeStereoMatches L<->R" if stereo else "SearchForInitialization(prev frame), window 100, on device",
                       "sharding": "frames round-robin over ranks; one all-gather of result slots per step"
                       if not stereo else "stereo frames independent per rank, no collective",
                       "contexts_in_flight": NCTX, "matches_last_step_rank0": state["matches"],
                       "host_ms_per_step": {k: v / args.steps * 1e3 for k, v in
                                            zip(("wait_extract", "enqueue_match", "wait_match"),
                                                state.get("host_s", [0, 0, 0]))}},
            "roofline": {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": tr["bytes"] if tr else None,
                         "traffic_detail": tr,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "avg_launch_ms": stage_ms[dom], "stage_ms_per_batch": stage_ms},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(out))
    for c in ctxs:
        c.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
