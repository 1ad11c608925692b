import sys; import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import vi_slam_amd as V
from oracle import orbo
W, H = 752, 480
yy, xx = np.mgrid[0:H, 0:W + 64]
rng = np.random.default_rng(11)
base = {
    "checker16": (((xx // 16 + yy // 16) & 1) * 255).astype(np.uint8),
    "binary_noise": (rng.integers(0, 2, xx.shape) * 255).astype(np.uint8),
    "dots": np.where(((xx % 7) == 3) & ((yy % 7) == 3), 255, 30).astype(np.uint8),
}
fx = fy = 500.0; cx, cy = W / 2, H / 2
bad = 0
fe = V.FExtractor(1500, 1.2, 8, 20, 7, W, H, max_batch=2)
for name, b in base.items():
    A = np.ascontiguousarray(b[:, 32:32 + W]); B = np.ascontiguousarray(b[:, 35:35 + W])
    res = fe.compute_batch([A, B]); res = [(k.copy(), d.copy(), m) for k, d, m in res]
    k0, d0 = res[0][0], res[0][1]; k1, d1 = res[1][0], res[1][1]
    z = np.full(len(k0), 9.0, np.float32)
    X = np.stack([(k0["x"] - cx) / fx * z, (k0["y"] - cy) / fy * z, z], 1).astype(np.float32)
    ck, cd, _ = fe.slot_dev_ptrs(1)
    T0 = np.hstack([np.eye(3), np.zeros((3, 1))]).astype(np.float32)
    Tcw = np.hstack([np.eye(3), np.array([[-3.0 / fx * 9.0], [0.0], [0.0]])]).astype(np.float32)
    for th in (7, 30):
        for fl in (3, 1):
            flags = np.full(len(k0), fl, np.uint8)
            m = V.FMatcher(fe, 0.9, True)
            nm, mc, _ = m.SearchByProjection(Tcw, T0, (fx, fy, cx, cy, 40.0, 0.08), th, k0, flags, X, d0, ck, cd, len(k1), None, True, (W, H))
            wn, wm, _ = orbo.search_by_projection_frame(Tcw, T0, (fx, fy, cx, cy, 40.0, 0.08), th, k0, flags, X, d0, k1, d1,
                                                        np.full(len(k1), -1, np.float32), fe.GetScaleFactors(), W, H, mono=True)
            ok = nm == wn and np.array_equal(mc, wm)
            # local map variant on the same data
            mps = np.zeros(len(k0), V.MP_TRACK_DTYPE)
            mps["proj_x"], mps["proj_y"] = k0["x"] - 3.0, k0["y"]; mps["proj_xr"] = mps["proj_x"] - 4.0
            mps["view_cos"] = 0.9; mps["level"] = k0["octave"]; mps["flags"] = fl
            m2 = V.FMatcher(fe, 0.8, True)
            n2, mc2 = m2.SearchByProjectionMapPoints(mps, d0, ck, cd, len(k1), None, float(th) / 7.0, None, (W, H))
            w2, wm2 = orbo.search_by_projection_mappoints(mps, d0, k1, d1, np.full(len(k1), -1, np.float32), fe.GetScaleFactors(), W, H, float(th) / 7.0, 0.8)
            ok2 = n2 == w2 and np.array_equal(mc2, wm2)
            print(name.ljust(13), th, fl, "frame", nm, "OK" if ok else "MISMATCH", "fb=%d" % m.search_init_fallbacks(), "| mappoints", n2, "OK" if ok2 else "MISMATCH", "fb=%d" % m2.search_init_fallbacks())
            bad += (not ok) + (not ok2)
fe.close()
print("bad", bad)
