"""Host-side mirror of the grid FAST detector behind vi_slam::geometry::FAST::detect (src/geometry/fast_cuda.cpp:70-132):
vilib::FASTGPU (thirdparty/vilib/visual_lib/include/vilib/feature_detection/fast/fast_gpu.h:44-61) over the C ABI of
include/vslam_fastgrid.h.  Same constructor arguments, same feature grid."""
import ctypes as C

import numpy as np

from . import _check, _p, lib

SUM_OF_ABS_DIFF_ALL, SUM_OF_ABS_DIFF_ON_ARC, MAX_THRESHOLD = 0, 1, 2  # vilib::fast_score


class _FgParams(C.Structure):  # vslam_fg_params
    _fields_ = [("image_width", C.c_int32), ("image_height", C.c_int32), ("cell_size_width", C.c_int32),
                ("cell_size_height", C.c_int32), ("min_level", C.c_int32), ("max_level", C.c_int32),
                ("horizontal_border", C.c_int32), ("vertical_border", C.c_int32), ("threshold", C.c_float),
                ("min_arc_length", C.c_int32), ("score", C.c_int32), ("tie_rule", C.c_int32), ("device", C.c_int32),
                ("max_batch", C.c_int32)]


_bound = False


def _bind():
    global _bound
    L = lib()
    if not _bound:
        vp, i = C.c_void_p, C.c_int
        L.vslam_fg_create.argtypes = [C.POINTER(_FgParams), C.POINTER(vp)]
        L.vslam_fg_destroy.argtypes = [vp]
        L.vslam_fg_destroy.restype = None
        L.vslam_fg_grid.argtypes = [vp, vp, vp]
        L.vslam_fg_detect.argtypes = [vp, vp, C.c_size_t, vp, vp, vp]
        L.vslam_fg_detect_batch.argtypes = [vp, i, vp, C.c_size_t, i, vp, vp, vp]
        L.vslam_fg_level_copy.argtypes = [vp, i, i, vp, C.c_size_t, vp, vp]
        L.vslam_fg_response_copy.argtypes = [vp, i, i, vp]
        _bound = True
    return L


class FASTGPU:
    """vilib::FASTGPU(image_width, image_height, cell_size_width, cell_size_height, min_level, max_level,
    horizontal_border, vertical_border, threshold, min_arc_length, score) (fast_gpu.cpp:52-92)."""

    def __init__(self, image_width, image_height, cell_size_width=32, cell_size_height=32, min_level=0, max_level=1,
                 horizontal_border=0, vertical_border=0, threshold=10.0, min_arc_length=10, score=SUM_OF_ABS_DIFF_ON_ARC,
                 tie_rule=0, device=0, max_batch=1):
        self.L = _bind()
        P = _FgParams(image_width, image_height, cell_size_width, cell_size_height, min_level, max_level,
                      horizontal_border, vertical_border, threshold, min_arc_length, score, tie_rule, device, max_batch)
        h = C.c_void_p()
        _check(self.L.vslam_fg_create(C.byref(P), C.byref(h)))
        self._h = h
        self.width, self.height, self.max_level, self.min_level = image_width, image_height, max_level, min_level
        nc, nr = C.c_int(), C.c_int()
        _check(self.L.vslam_fg_grid(self._h, C.byref(nc), C.byref(nr)))
        self.n_cols, self.n_rows = nc.value, nr.value  # getCellCountHorizontal / getCellCountVertical
        self.cells = self.n_cols * self.n_rows

    def close(self):
        if self._h:
            self.L.vslam_fg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def detect(self, image):
        """Frame(image) + FASTGPU::detect + copyGridToHost -> (pos[cells, 2], score[cells], level[cells])."""
        image = np.ascontiguousarray(image, np.uint8)
        assert image.shape == (self.height, self.width)
        pos = np.zeros((self.cells, 2), np.float32)
        sc = np.zeros(self.cells, np.float32)
        lv = np.zeros(self.cells, np.int32)
        _check(self.L.vslam_fg_detect(self._h, _p(image), image.strides[0], _p(pos), _p(sc), _p(lv)))
        return pos, sc, lv

    def detect_batch(self, images=None, dev_ptrs=None, pitch=None):
        """Several images per pass: host arrays, or device addresses (dev_ptrs, pitch)."""
        if dev_ptrs is None:
            imgs = [np.ascontiguousarray(im, np.uint8) for im in images]
            n, pitch = len(imgs), imgs[0].strides[0]
            ptrs = (C.c_void_p * n)(*[im.ctypes.data for im in imgs])
            on_dev = 0
        else:
            n = len(dev_ptrs)
            ptrs = (C.c_void_p * n)(*dev_ptrs)
            on_dev = 1
        pos = np.zeros((n, self.cells, 2), np.float32)
        sc = np.zeros((n, self.cells), np.float32)
        lv = np.zeros((n, self.cells), np.int32)
        _check(self.L.vslam_fg_detect_batch(self._h, n, ptrs, pitch, on_dev, _p(pos), _p(sc), _p(lv)))
        return pos, sc, lv

    def getPoints(self, pos, score, level):
        """DetectorBaseGPU::processGrid (detector_base_gpu.cpp:204-218): the occupied cells as (x, y, score, level)."""
        occ = np.nonzero(score > 0)[0]
        return [(float(pos[i, 0]), float(pos[i, 1]), float(score[i]), int(level[i])) for i in occ]

    def level(self, slot, level):
        w, h = C.c_int(), C.c_int()
        _check(self.L.vslam_fg_level_copy(self._h, slot, level, None, 0, C.byref(w), C.byref(h)))
        out = np.zeros((h.value, w.value), np.uint8)
        _check(self.L.vslam_fg_level_copy(self._h, slot, level, _p(out), w.value, None, None))
        return out

    def response(self, slot, level):
        """DetectorBaseGPU::copyResponseTo (detector_base_gpu.cpp:127-141)."""
        out = np.zeros((self.height >> level, self.width >> level), np.float32)
        _check(self.L.vslam_fg_response_copy(self._h, slot, level, _p(out)))
        return out
