"""vi_slam_amd -- MI355X-native ORB front-end (extract + Hamming match) behind vi_slam's FExtractor /
FMatcher / Frame interface.

This package is a thin ctypes mirror of the C ABI in include/vslam_fe.h (libvslam_fe.so, hand-written
HIP for gfx950).  There is NO CPU fallback: constructing an extractor without the built library or
without a GPU raises.  The reference-side names are kept (FExtractor.compute, GetScaleFactors,
FMatcher.DescriptorDistance / SearchForInitialization, Frame.ComputeStereoMatches) so tests read like
the reference's call sites (src/datastructures/frame.cpp:107-127,289; src/core/tracking.cpp:2323-2324).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VSLAM_FE_LIB") or os.path.join(_HERE, "libvslam_fe.so")  # VSLAM_FE_LIB: a diagnostic build
HOST_LIB_PATH = os.path.join(_HERE, "libvslam_host.so")

VSLAM_OK = 0
ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_CAPACITY, ERR_UNSUPPORTED, ERR_COMM = -1, -2, -3, -4, -5, -6
IMGS_HOST, IMGS_DEVICE, IMGS_PINNED, IMGS_STAGED = 0, 1, 2, 3  # where the input images live (vslam_fe.h VSLAM_IMGS_*)
COMM_ID_BYTES = 128
FLAG_ATAN_FMA = 1
FLAG_HOST_OCTREE = 2
MAX_BATCH = 64

#: numpy view of vslam_kp == cv::KeyPoint (28 bytes)
KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])

# include/vslam_fe.h entry points: every one of these must be exported by libvslam_fe.so
ABI_SYMBOLS = [
    "vslam_fe_create", "vslam_fe_destroy", "vslam_last_error", "vslam_fe_tables", "vslam_fe_extract",
    "vslam_fe_extract_batch", "vslam_fe_level_size", "vslam_fe_level_copy", "vslam_fe_candidates",
    "vslam_fe_slot_buffers", "vslam_fe_slot_host_views", "vslam_fe_capacity", "vslam_fe_stream", "vslam_hamming_top2", "vslam_hamming_matrix",
    "vslam_stereo_match", "vslam_stereo_match_batch", "vslam_search_for_initialization",
    "vslam_dbg_sincos", "vslam_dbg_fast_atan2", "vslam_fe_pack_slots", "vslam_fe_set_profiling",
    "vslam_fe_get_profile", "vslam_fe_extract_batch_async", "vslam_fe_extract_wait",
    "vslam_search_for_initialization_batch", "vslam_frame_stereo_batch_async", "vslam_frame_stereo_wait",
    "vslam_fe_pack_slot_range", "vslam_dbg_octree_stamps", "vslam_dbg_search_init_fallbacks", "vslam_search_init_dev_async",
    "vslam_search_init_dev_wait", "vslam_fe_slot_count_ptr", "vslam_fe_pack_slot_range_async", "vslam_fe_wait_for", "vslam_fe_event_record",
    "vslam_fe_event_wait", "vslam_projection_direction", "vslam_search_by_projection_frame",
    "vslam_search_by_projection_dev_async", "vslam_search_by_projection_dev_wait", "vslam_stereo_points_dev_async",
    "vslam_stereo_points_buffers", "vslam_search_by_projection_mappoints", "vslam_distinctive_descriptors", "vslam_voc_create", "vslam_voc_destroy",
    "vslam_voc_info", "vslam_bow_transform", "vslam_bow_transform_slots_async", "vslam_bow_transform_slots_wait",
    "vslam_bow_assemble", "vslam_search_by_bow", "vslam_search_by_bow_keyframes",
    "vslam_search_for_triangulation", "vslam_fuse_search", "vslam_dbg_logf", "vslam_search_by_projection_keyframe",
    "vslam_search_by_projection_sim3",
    "vslam_comm_unique_id", "vslam_comm_create", "vslam_comm_destroy", "vslam_comm_rank", "vslam_comm_world",
    "vslam_exchange_ring", "vslam_exchange_allgather", "vslam_host_alloc", "vslam_host_free",
    "vslam_fe_stage_images_async", "vslam_fe_octree_stats", "vslam_fe_delivery_stats", "vslam_dbg_search_init_replay_stats", "vslam_tuning_init", "vslam_fe_set_tuning", "vslam_stereo_fisheye_candidates", "vslam_hamming_top2_batch", "vslam_hamming_top2_batch_dev_async", "vslam_fe_set_fast_gate",
    "vslam_voc_load", "vslam_voc_file_open", "vslam_voc_file_close", "vslam_voc_file_info", "vslam_voc_file_arrays",
    "vslam_voc_file_last_error", "vslam_dbg_qlz_decode",
]


#: vslam_mp_track: per-MapPoint tracking record (mappoint.h:73-81 after Frame::isInFrustum)
MP_TRACK_DTYPE = np.dtype([("proj_x", "<f4"), ("proj_y", "<f4"), ("proj_xr", "<f4"), ("view_cos", "<f4"),
                           ("level", "<i4"), ("flags", "<u4")])


#: vslam_fuse_point: one candidate MapPoint of FMatcher::Fuse
FUSE_POINT_DTYPE = np.dtype([("pos", "<f4", 3), ("normal", "<f4", 3), ("min_distance", "<f4"),
                             ("max_distance", "<f4"), ("valid", "<i4")])


class _TriParams(C.Structure):  # vslam_tri_params
    _fields_ = [("F12", C.c_float * 9), ("ep_x", C.c_float), ("ep_y", C.c_float), ("only_stereo", C.c_int32),
                ("coarse", C.c_int32), ("check_orientation", C.c_int32)]


class _FuseParams(C.Structure):  # vslam_fuse_params
    _fields_ = [("Rcw", C.c_float * 9), ("tcw", C.c_float * 3), ("Ow", C.c_float * 3), ("fx", C.c_float),
                ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("bf", C.c_float), ("th", C.c_float),
                ("log_scale_factor", C.c_float), ("img_w", C.c_int32), ("img_h", C.c_int32), ("sim3", C.c_int32),
                ("gemm_float", C.c_int32), ("Rb", C.c_float * 9), ("tb", C.c_float * 3)]


class _ProjParams(C.Structure):  # vslam_proj_params
    _fields_ = [("Tcw", C.c_float * 12), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("mbf", C.c_float), ("th", C.c_float), ("forward", C.c_int32), ("backward", C.c_int32),
                ("check_orientation", C.c_int32), ("img_w", C.c_int32), ("img_h", C.c_int32),
                ("gemm_float", C.c_int32)]


class _SbpJob(C.Structure):  # vslam_sbp_job
    _fields_ = [("p", _ProjParams)] + [(n, C.c_void_p) for n in (
        "dev_last_kps", "dev_n_last", "dev_last_flags", "dev_last_x3dw", "dev_mp_desc", "dev_cur_kps", "dev_cur_desc",
        "dev_n_cur", "dev_cur_u_right", "dev_cur_occupied")]


class VslamError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("vslam error %d: %s" % (code, msg))
        self.code = code


#: vslam_tuning field names, in declaration order (include/vslam_fe.h); every field -1 = library default
TUNING_FIELDS = ["pyramid_per_level", "pyr_rows", "pyr_threads", "blur_rows", "fast_threads", "fast_pitch", "fast_lds_pad",
                 "octree_walk_kernel", "oct_fine_depth", "oct_lds_budget_kb", "oct_regkeys", "oct_max_iter",
                 "oct_debug", "graphs", "h2d_route", "d2h_route", "copy_wgs", "pull_depth",
                 "init_topm", "init_match_host", "sbp_topm", "sbp_sequential", "si_queries_per_block", "fg_threads",
                 "wait_spin", "numa", "host_prof", "stream_priority", "stage_split_event",
                 "oct_threads", "fast_kernel", "fast_band_cells", "wave_prio", "oct_precount", "desc_kpw"]


class _Tuning(C.Structure):  # vslam_tuning
    _fields_ = [(f, C.c_int32) for f in TUNING_FIELDS] + [("reserved", C.c_int32 * 1)]


def make_tuning(**kw):
    """vslam_tuning with every field at its default (-1) except the given ones, e.g. make_tuning(oct_fine_depth=1)"""
    t = _Tuning()
    lib().vslam_tuning_init(C.byref(t))
    for k, v in kw.items():
        if k not in TUNING_FIELDS:
            raise KeyError("unknown tuning field %r" % k)
        setattr(t, k, int(v))
    return t


class _Params(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("nfeatures", C.c_int32),
                ("scale_factor", C.c_float), ("nlevels", C.c_int32), ("ini_th_fast", C.c_int32),
                ("min_th_fast", C.c_int32), ("device", C.c_int32), ("max_batch", C.c_int32),
                ("flags", C.c_uint32), ("gauss_taps", C.c_int32 * 7), ("tuning", C.POINTER(_Tuning))]


class _InitJob(C.Structure):  # vslam_init_job
    _fields_ = [("dev_kps1", C.c_void_p), ("dev_desc1", C.c_void_p), ("dev_n1", C.c_void_p),
                ("dev_kps2", C.c_void_p), ("dev_desc2", C.c_void_p), ("dev_n2", C.c_void_p),
                ("dev_prev_matched", C.c_void_p)]


def bind_voc_file(L):
    """ctypes signatures of the vocabulary-file reader (in libvslam_fe.so and in the GPU-free libvslam_host.so)"""
    vp = C.c_void_p
    L.vslam_voc_file_open.argtypes = [C.c_char_p, vp]
    L.vslam_voc_file_close.argtypes = [vp]
    L.vslam_voc_file_close.restype = None
    L.vslam_voc_file_info.argtypes = [vp] + [vp] * 9
    L.vslam_voc_file_arrays.argtypes = [vp] + [vp] * 6
    L.vslam_voc_file_last_error.restype = C.c_char_p
    L.vslam_dbg_qlz_decode.argtypes = [C.c_char_p, C.c_size_t, vp, C.c_size_t, vp]
    L.vslam_dbg_qlz_decode.restype = C.c_long
    return L


def qlz_decode(packet, library=None, cap=None):
    """vslam_dbg_qlz_decode: one QuickLZ level-1 packet -> (bytes, packet size); raises ValueError on a damaged packet."""
    L = library if library is not None else lib()
    packet = bytes(packet)
    n = len(packet)
    if cap is None:
        cap = 16 + 128 * n
    dst = C.create_string_buffer(max(cap, 1))
    used = C.c_size_t(0)
    got = L.vslam_dbg_qlz_decode(packet, n, dst, cap, C.byref(used))
    if got < 0:
        raise ValueError("damaged QuickLZ packet")
    return dst.raw[:got], used.value


_lib = None


def lib():
    """Load libvslam_fe.so (the HIP product library).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(make -C vi_slam_amd/csrc); there is no CPU fallback" % LIB_PATH)
        try:
            # PyTorch bundles its own libamdhip64; if it is going to be used in this process it must be the
            # first HIP runtime loaded, or torch.cuda later reports "No HIP GPUs are available".
            import torch  # noqa: F401
        except Exception:
            pass
        L = C.CDLL(LIB_PATH)
        vp, i, f = C.c_void_p, C.c_int, C.c_float
        L.vslam_fe_create.argtypes = [C.POINTER(_Params), C.POINTER(vp)]
        L.vslam_fe_destroy.argtypes = [vp]
        L.vslam_fe_destroy.restype = None
        L.vslam_last_error.restype = C.c_char_p
        L.vslam_fe_tables.argtypes = [vp, vp, vp, vp, vp, vp]
        L.vslam_fe_extract.argtypes = [vp, vp, C.c_size_t, i, i, vp, vp, i, vp, vp]
        L.vslam_fe_extract_batch.argtypes = [vp, i, vp, C.c_size_t, i, i, i, vp, vp, i, vp, vp]
        L.vslam_fe_level_size.argtypes = [vp, i, vp, vp]
        L.vslam_fe_level_copy.argtypes = [vp, i, i, i, vp, C.c_size_t]
        L.vslam_fe_candidates.argtypes = [vp, i, i, vp, i]
        L.vslam_fe_slot_buffers.argtypes = [vp, i, vp, vp, vp]
        L.vslam_fe_slot_host_views.argtypes = [vp, i, vp, vp]
        L.vslam_fe_capacity.argtypes = [vp]
        L.vslam_projection_direction.argtypes = [vp, vp, C.c_float, i, i, vp, vp]
        L.vslam_search_by_projection_mappoints.argtypes = [vp, vp, vp, i, vp, vp, i, vp, vp, i, i, C.c_float, C.c_float,
                                                           vp, vp]
        L.vslam_distinctive_descriptors.argtypes = [vp, vp, vp, i, vp]
        L.vslam_voc_create.argtypes = [i, i, i, i, i, vp, vp, vp, i, vp, vp, vp, vp]
        L.vslam_voc_destroy.argtypes = [vp]
        L.vslam_voc_load.argtypes = [i, C.c_char_p, vp]
        bind_voc_file(L)
        L.vslam_voc_destroy.restype = None
        L.vslam_voc_info.argtypes = [vp, vp, vp, vp, vp]
        L.vslam_bow_transform.argtypes = [vp, vp, vp, i, i, vp, vp, vp]
        L.vslam_bow_transform_slots_async.argtypes = [vp, vp, i, i, i]
        L.vslam_bow_transform_slots_wait.argtypes = [vp, vp, vp, vp, vp]
        L.vslam_search_by_bow.argtypes = [vp, vp, vp, vp, i, vp, vp, vp, i, vp, vp, i, vp, vp, vp, i, C.c_float, i, vp, vp]
        L.vslam_search_by_bow_keyframes.argtypes = [vp, vp, vp, vp, i, vp, vp, vp, i, vp, vp, vp, i, vp, vp, vp, i,
                                                    C.c_float, i, vp, vp]
        L.vslam_bow_assemble.argtypes = [i, i, vp, vp, vp, i, vp, vp, vp, vp, vp, vp, vp]
        L.vslam_search_for_triangulation.argtypes = [vp, vp] + [vp, vp, vp, vp, i, vp, vp, vp, i] * 2 + [vp, vp]
        L.vslam_fuse_search.argtypes = [vp, vp, vp, vp, i, vp, vp, i, vp, vp, vp]
        L.vslam_dbg_logf.argtypes = [vp, vp, i, vp]
        L.vslam_search_by_projection_sim3.argtypes = [vp, vp, vp, C.c_float, C.c_float, i, vp, vp, vp, vp, vp, vp, i, vp, vp, i,
                                                      vp, vp, vp]
        L.vslam_search_by_projection_keyframe.argtypes = [vp, vp, vp, C.c_float, i, vp, i, vp, vp, vp, vp, vp, vp, vp, i, vp,
                                                          vp, vp]
        L.vslam_search_by_projection_dev_async.argtypes = [vp, i, vp]
        L.vslam_search_by_projection_dev_wait.argtypes = [vp, vp, vp, vp]
        L.vslam_stereo_points_dev_async.argtypes = [vp, i, vp, C.c_float, C.c_float, C.c_float, C.c_float, i, i]
        L.vslam_stereo_points_buffers.argtypes = [vp, i, vp, vp, vp, vp]
        L.vslam_search_by_projection_frame.argtypes = [vp, vp, vp, i, vp, vp, vp, vp, vp, i, vp, vp, vp, vp]
        L.vslam_fe_stream.argtypes = [vp]
        L.vslam_fe_stream.restype = vp
        L.vslam_hamming_top2.argtypes = [vp, vp, i, vp, i, vp, vp]
        L.vslam_hamming_matrix.argtypes = [vp, vp, i, vp, i, vp]
        L.vslam_stereo_match.argtypes = [vp, i, vp, i, f, f, vp, vp]
        L.vslam_stereo_match_batch.argtypes = [vp, vp, i, vp, vp, f, f, vp, vp]
        L.vslam_search_for_initialization.argtypes = [vp, vp, vp, i, vp, vp, i, i, i, vp, vp, i, f, i, vp]
        L.vslam_search_init_dev_async.argtypes = [vp, i, vp, i, i, i, f, i]
        L.vslam_search_init_dev_wait.argtypes = [vp, vp, vp, vp, vp]
        L.vslam_fe_slot_count_ptr.argtypes = [vp, i, vp]
        L.vslam_fe_pack_slots.argtypes = [vp, i, vp, C.c_size_t]
        L.vslam_fe_pack_slot_range.argtypes = [vp, i, i, vp, C.c_size_t]
        L.vslam_fe_pack_slot_range_async.argtypes = [vp, i, i, vp, C.c_size_t]
        L.vslam_fe_wait_for.argtypes = [vp, vp]
        L.vslam_fe_event_record.argtypes = [vp, i]
        L.vslam_fe_event_wait.argtypes = [vp, vp, i]
        L.vslam_fe_set_profiling.argtypes = [vp, i]
        L.vslam_fe_get_profile.argtypes = [vp, vp, vp, vp]
        L.vslam_fe_extract_batch_async.argtypes = [vp, i, vp, C.c_size_t, i, i, i, i]
        L.vslam_fe_extract_wait.argtypes = [vp, vp, vp, i, vp, vp]
        L.vslam_frame_stereo_batch_async.argtypes = [vp, i, vp, C.c_size_t, i, f, f, i]
        L.vslam_frame_stereo_wait.argtypes = [vp, vp, vp, i, vp, vp, vp]
        L.vslam_search_for_initialization_batch.argtypes = [vp, i, vp, vp, vp, vp, vp, vp, i, i, vp, vp, i, f, i,
                                                            vp]
        L.vslam_dbg_sincos.argtypes = [vp, vp, i, vp, vp]
        L.vslam_dbg_fast_atan2.argtypes = [vp, vp, vp, i, i, vp]
        L.vslam_comm_unique_id.argtypes = [vp]
        L.vslam_comm_create.argtypes = [i, i, i, vp, C.POINTER(vp)]
        L.vslam_comm_destroy.argtypes = [vp]
        L.vslam_comm_destroy.restype = None
        L.vslam_comm_rank.argtypes = [vp]
        L.vslam_comm_world.argtypes = [vp]
        L.vslam_exchange_ring.argtypes = [vp, vp, vp, vp, C.c_size_t]
        L.vslam_exchange_allgather.argtypes = [vp, vp, vp, vp, C.c_size_t]
        L.vslam_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
        L.vslam_host_free.argtypes = [vp]
        L.vslam_host_free.restype = None
        L.vslam_fe_stage_images_async.argtypes = [vp, i, vp, C.c_size_t, i]
        _lib = L
    return _lib


def _check(rc):
    if rc != VSLAM_OK:
        raise VslamError(rc, lib().vslam_last_error().decode())


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class PinnedImages:
    """`count` images of `height` x `width` bytes in ONE pinned host allocation (vslam_host_alloc = hipHostMalloc), rows
    `pitch` bytes apart: what a capture driver's DMA ring looks like.  `.array[i]` is a numpy view to fill, `.ptrs` the
    ctypes pointer table for compute_batch_async(..., where=IMGS_PINNED)."""

    def __init__(self, count, height, width, pitch=None):
        self.pitch = pitch or width
        self.count, self.height, self.width = count, height, width
        nbytes = count * height * self.pitch
        h = C.c_void_p()
        _check(lib().vslam_host_alloc(nbytes, C.byref(h)))
        self._h = h
        self._flat = np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(h.value))
        self.array = self._flat.reshape(count, height, self.pitch)[:, :, :width]
        self.ptrs = (C.c_void_p * count)(*[h.value + i * height * self.pitch for i in range(count)])

    def close(self):
        if getattr(self, "_h", None):
            self.array = self._flat = None
            lib().vslam_host_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm:
    """vslam_comm: one RCCL communicator per process/GPU for the exchange step (vslam_exchange_ring / _allgather).
    `unique_id()` on rank 0 -> distribute its 128 bytes by any channel -> `Comm(device, rank, world, id)` everywhere."""

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * COMM_ID_BYTES)()
        _check(lib().vslam_comm_unique_id(buf))
        return bytes(buf)

    def __init__(self, device, rank, world, uid):
        h = C.c_void_p()
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(uid)
        _check(lib().vslam_comm_create(device, rank, world, buf, C.byref(h)))
        self._h, self.rank, self.world = h, rank, world

    def ring(self, fe, dev_send, dev_recv, nbytes):
        """enqueue on fe's stream: nbytes of dev_send -> rank+1, dev_recv <- rank-1"""
        _check(lib().vslam_exchange_ring(fe._h, self._h, dev_send, dev_recv, nbytes))

    def allgather(self, fe, dev_send, dev_recv_all, nbytes_per_rank):
        _check(lib().vslam_exchange_allgather(fe._h, self._h, dev_send, dev_recv_all, nbytes_per_rank))

    def close(self):
        if getattr(self, "_h", None):
            lib().vslam_comm_destroy(self._h)
            self._h = None


class FExtractor:
    """vi_slam::geometry::FExtractor (include/vi_slam/geometry/fextractor.h:26-91) on one MI355X.

    Unlike the reference the image size is fixed at construction (pyramids live in HBM) and a context
    owns `max_batch` image slots so several frames go through each kernel launch.
    """

    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, width, height, device=0,
                 max_batch=1, flags=0, gauss_taps=None, tuning=None):
        """tuning: dict of vslam_tuning fields (TUNING_FIELDS) that override the library defaults for THIS context"""
        L = lib()
        tn = make_tuning(**tuning) if tuning else None
        p = _Params(width, height, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device, max_batch,
                    flags, (C.c_int32 * 7)(*(gauss_taps or [0] * 7)), C.pointer(tn) if tn is not None else None)
        h = C.c_void_p()
        _check(L.vslam_fe_create(C.byref(p), C.byref(h)))
        self._h = h
        self.nfeatures, self.nlevels, self.width, self.height = nfeatures, nlevels, width, height
        self.scaleFactor = scaleFactor
        self.max_batch = max_batch
        self.cap = L.vslam_fe_capacity(h)  # slot capacity: nfeatures + 4*nlevels + 8 (multiple of 4) or the exact bound
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            lib().vslam_fe_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- getters (fextractor.h:42-62)
    def _tables(self):
        n = self.nlevels
        sf, isf, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        q = np.zeros(n, np.int32)
        lib().vslam_fe_tables(self._h, _p(sf), _p(isf), _p(s2), _p(is2), _p(q))
        return sf, isf, s2, is2, q

    def GetLevels(self):
        return self.nlevels

    def GetScaleFactor(self):
        return self.scaleFactor

    def GetScaleFactors(self):
        return self._tables()[0]

    def GetInverseScaleFactors(self):
        return self._tables()[1]

    def GetScaleSigmaSquares(self):
        return self._tables()[2]

    def GetInverseScaleSigmaSquares(self):
        return self._tables()[3]

    def features_per_level(self):
        return self._tables()[4]

    # ---- compute (fextractor.h:38-40)
    def compute(self, image, vLappingArea=(0, 0)):
        """FExtractor::compute.  Returns (keypoints[KP_DTYPE], descriptors[N,32] u8, monoIndex)."""
        image = np.ascontiguousarray(image, dtype=np.uint8)
        if image.ndim != 2 or image.shape != (self.height, self.width):
            raise VslamError(ERR_INVALID, "image must be %dx%d CV_8UC1" % (self.width, self.height))
        kps = np.zeros(self.cap, KP_DTYPE)
        desc = np.zeros((self.cap, 32), np.uint8)
        n, mono = C.c_int(0), C.c_int(0)
        _check(lib().vslam_fe_extract(self._h, _p(image), image.strides[0], vLappingArea[0], vLappingArea[1],
                                      _p(kps), _p(desc), self.cap, C.byref(n), C.byref(mono)))
        return kps[:n.value].copy(), desc[:n.value].copy(), mono.value

    def compute_batch(self, images, vLappingArea=(0, 0), device_ptrs=None, pitch=None, to_host=True):
        """Batched compute: `images` is a list of HxW uint8 arrays (host), or pass `device_ptrs`
        (list of int device addresses, rows `pitch` bytes apart) for zero-copy HBM-resident input.
        Returns a list of (keypoints, descriptors, monoIndex); with to_host=False only counts."""
        if device_ptrs is not None:
            nimg = len(device_ptrs)
            ptrs = (C.c_void_p * nimg)(*device_ptrs)
            on_dev = 1
            keep = None
        else:
            keep = [np.ascontiguousarray(im, dtype=np.uint8) for im in images]
            for im in keep:
                if im.shape != (self.height, self.width):
                    raise VslamError(ERR_INVALID, "image must be %dx%d CV_8UC1" % (self.width, self.height))
            nimg = len(keep)
            ptrs = (C.c_void_p * nimg)(*[im.ctypes.data for im in keep])
            pitch = self.width
            on_dev = 0
        n = (C.c_int * nimg)()
        mono = (C.c_int * nimg)()
        if to_host:
            kps = np.zeros((nimg, self.cap), KP_DTYPE)
            desc = np.zeros((nimg, self.cap, 32), np.uint8)
            kp_ptrs = (C.c_void_p * nimg)(*[kps[i].ctypes.data for i in range(nimg)])
            d_ptrs = (C.c_void_p * nimg)(*[desc[i].ctypes.data for i in range(nimg)])
        else:
            kp_ptrs = d_ptrs = None
        _check(lib().vslam_fe_extract_batch(self._h, nimg, ptrs, pitch, on_dev, vLappingArea[0], vLappingArea[1],
                                            kp_ptrs, d_ptrs, self.cap, n, mono))
        if not to_host:
            return [(n[i], mono[i]) for i in range(nimg)]
        return [(kps[i, :n[i]].copy(), desc[i, :n[i]].copy(), mono[i]) for i in range(nimg)]

    def stage_images_async(self, ptrs, pitch, where=IMGS_PINNED):
        """Upload only (vslam_fe_stage_images_async): pull the host images into level 0 of the slots on this context's
        stream; follow with compute_batch_async / frame_stereo_async(..., where=IMGS_STAGED)."""
        nimg = len(ptrs)
        p = ptrs if isinstance(ptrs, C.Array) else (C.c_void_p * nimg)(*ptrs)
        self._staged_n = nimg
        _check(lib().vslam_fe_stage_images_async(self._h, nimg, p, pitch, where))

    def compute_batch_async(self, device_ptrs, pitch, vLappingArea=(0, 0), to_host=True, where=IMGS_DEVICE):
        """Enqueue one batched pass and return immediately (vslam_fe_extract_batch_async); collect with wait().
        `device_ptrs`: image addresses, HBM-resident (where=IMGS_DEVICE, zero copy) or pinned host memory
        (where=IMGS_PINNED, e.g. PinnedImages.ptrs: pulled over PCIe by the pass itself)."""
        nimg = len(device_ptrs)
        ptrs = device_ptrs if isinstance(device_ptrs, C.Array) else (C.c_void_p * nimg)(*device_ptrs)
        self._pending = (nimg, bool(to_host))
        # to_host="with_matcher": want_host = 2, the delivery rides with the SearchForInitialization that follows
        _check(lib().vslam_fe_extract_batch_async(self._h, nimg, ptrs, pitch, where, vLappingArea[0], vLappingArea[1],
                                                  2 if to_host == "with_matcher" else int(bool(to_host))))

    def wait(self, copy=False):
        """Block until the enqueued pass is done.  Returns a list of (keypoints, descriptors, monoIndex);
        the arrays are views into per-context staging that the next wait() overwrites unless copy=True."""
        nimg, to_host = self._pending
        n = (C.c_int * nimg)()
        mono = (C.c_int * nimg)()
        if not to_host:
            _check(lib().vslam_fe_extract_wait(self._h, None, None, 0, n, mono))
            return [(n[i], mono[i]) for i in range(nimg)]
        self._host_views()
        _check(lib().vslam_fe_extract_wait(self._h, None, None, 0, n, mono))
        if copy:
            return [(self._out_kps[i, :n[i]].copy(), self._out_desc[i, :n[i]].copy(), mono[i]) for i in range(nimg)]
        return [(self._out_kps[i, :n[i]], self._out_desc[i, :n[i]], mono[i]) for i in range(nimg)]

    def _host_views(self):
        """numpy views of the context's pinned result staging (vslam_fe_slot_host_views): the result kernel
        writes there, so reading in place needs no further copy."""
        if getattr(self, "_out_kps", None) is None:
            hk, hd = C.c_void_p(), C.c_void_p()
            _check(lib().vslam_fe_slot_host_views(self._h, 0, C.byref(hk), C.byref(hd)))
            nk = self.max_batch * self.cap
            self._out_kps = np.ctypeslib.as_array((C.c_uint8 * (nk * 28)).from_address(hk.value)).view(KP_DTYPE) \
                .reshape(self.max_batch, self.cap)
            self._out_desc = np.ctypeslib.as_array((C.c_uint8 * (nk * 32)).from_address(hd.value)) \
                .reshape(self.max_batch, self.cap, 32)

    # ---- Frame::Frame(stereo) hot section (frame.cpp:102-132), several frames per enqueue
    def frame_stereo_async(self, device_ptrs, pitch, bf, fx, to_host=True, where=IMGS_DEVICE):
        """device_ptrs = [L0, R0, L1, R1, ...] images, HBM-resident or (where=IMGS_PINNED) in pinned host memory.
        Enqueues extraction of all images and ComputeStereoMatches of every (L,R) pair; returns immediately.
        Collect with frame_stereo_wait()."""
        nimg = len(device_ptrs)
        ptrs = device_ptrs if isinstance(device_ptrs, C.Array) else (C.c_void_p * nimg)(*device_ptrs)
        self._pending = (nimg, to_host)
        _check(lib().vslam_frame_stereo_batch_async(self._h, nimg // 2, ptrs, pitch, where, bf, fx, int(to_host)))

    def frame_stereo_wait(self):
        """-> (list of (keypoints, descriptors) per image, list of (mvuRight, mvDepth) per stereo frame);
        views into per-context staging, overwritten by the next wait."""
        nimg, to_host = self._pending
        npairs = nimg // 2
        n = (C.c_int * nimg)()
        self._host_views()
        if getattr(self, "_out_u", None) is None:
            self._out_u = np.zeros((self.max_batch, self.cap), np.float32)
            self._out_dep = np.zeros((self.max_batch, self.cap), np.float32)
            self._out_u_ptrs = (C.c_void_p * self.max_batch)(*[self._out_u[i].ctypes.data
                                                               for i in range(self.max_batch)])
            self._out_dep_ptrs = (C.c_void_p * self.max_batch)(*[self._out_dep[i].ctypes.data
                                                                 for i in range(self.max_batch)])
        _check(lib().vslam_frame_stereo_wait(self._h, None, None, self.cap, n, self._out_u_ptrs, self._out_dep_ptrs))
        feats = [(self._out_kps[i, :n[i]], self._out_desc[i, :n[i]]) for i in range(nimg)] if to_host else \
            [(n[i], None) for i in range(nimg)]
        stereo = [(self._out_u[j, :n[2 * j]], self._out_dep[j, :n[2 * j]]) for j in range(npairs)]
        return feats, stereo

    # ---- mvImagePyramid (fextractor.h:64)
    def level_size(self, level):
        w, h = C.c_int(), C.c_int()
        _check(lib().vslam_fe_level_size(self._h, level, C.byref(w), C.byref(h)))
        return w.value, h.value

    def mvImagePyramid(self, level, slot=0, blurred=False):
        w, h = self.level_size(level)
        out = np.zeros((h, w), np.uint8)
        _check(lib().vslam_fe_level_copy(self._h, slot, level, int(blurred), _p(out), w))
        return out

    def candidates(self, level, slot=0):
        """vToDistributeKeys of the last compute (fextractor.cpp:769-817) for stage-wise parity tests."""
        n = lib().vslam_fe_candidates(self._h, slot, level, None, 0)
        if n < 0:
            _check(n)
        out = np.zeros(max(n, 1), KP_DTYPE)
        lib().vslam_fe_candidates(self._h, slot, level, _p(out), n)
        return out[:n]

    def slot_buffers(self, slot=0):
        """(device address of vslam_kp[n], device address of descriptors, n) of the last compute."""
        k, d, n = C.c_void_p(), C.c_void_p(), C.c_int()
        _check(lib().vslam_fe_slot_buffers(self._h, slot, C.byref(k), C.byref(d), C.byref(n)))
        return k.value, d.value, n.value

    def slot_dev_ptrs(self, slot):
        """(device address of the slot's vslam_kp array, of its descriptors, of its int32 keypoint count):
        fixed for the life of the context, so they can be used before the results exist."""
        if getattr(self, "_slot_ptrs", None) is None:
            self._slot_ptrs = []
            for s in range(self.max_batch):
                k, d, n = C.c_void_p(), C.c_void_p(), C.c_int()
                _check(lib().vslam_fe_slot_buffers(self._h, s, C.byref(k), C.byref(d), C.byref(n)))
                c = C.c_void_p()
                _check(lib().vslam_fe_slot_count_ptr(self._h, s, C.byref(c)))
                self._slot_ptrs.append((k.value, d.value, c.value))
        return self._slot_ptrs[slot]

    def wait_for(self, other):
        """GPU-side: work enqueued on this context from now on runs after everything enqueued on `other`."""
        _check(lib().vslam_fe_wait_for(self._h, other._h))

    def event_record(self, idx):
        _check(lib().vslam_fe_event_record(self._h, idx))

    def set_fast_gate(self, other):
        """this context's FAST launches wait (GPU side) for `other`'s latest FAST launch; None removes the gate"""
        L = lib()
        L.vslam_fe_set_fast_gate.argtypes = [C.c_void_p, C.c_void_p]
        _check(L.vslam_fe_set_fast_gate(self._h, other._h if other is not None else None))

    def event_wait(self, other, idx):
        """GPU-side: this context's later work waits for `other`'s last recorded event idx."""
        _check(lib().vslam_fe_event_wait(self._h, other._h, idx))

    def stream(self):
        return lib().vslam_fe_stream(self._h)

    @property
    def slot_bytes(self):
        """Bytes of one packed result slot (see vslam_fe_pack_slots), rounded to 256."""
        return (16 + self.cap * 60 + 255) & ~255

    def pack_slots(self, nslots, dev_dst, slot_bytes=None, first=0, sync=True):
        fn = lib().vslam_fe_pack_slot_range if sync else lib().vslam_fe_pack_slot_range_async
        _check(fn(self._h, first, nslots, dev_dst, slot_bytes or self.slot_bytes))

    # ---- stereo points of the last stereo enqueue (Frame::UnprojectStereo, frame.cpp:1023-1037)
    def stereo_points_async(self, Twc, cam, observations=True, gemm_float=False):
        """Twc: list of 3x4 [mRwc | mOw], one per stereo pair of the last stereo enqueue; cam = (cx, cy, invfx,
        invfy).  Fills the per-pair device arrays returned by stereo_points_buffers()."""
        T = np.ascontiguousarray(np.asarray(Twc, np.float32).reshape(len(Twc), -1)[:, :12])
        _check(lib().vslam_stereo_points_dev_async(self._h, len(Twc), _p(T), C.c_float(cam[0]), C.c_float(cam[1]),
                                                   C.c_float(cam[2]), C.c_float(cam[3]), int(observations),
                                                   int(gemm_float)))

    def stereo_points_buffers(self, pair, with_stereo=True):
        """-> device addresses (x3Dw [cap x 3 f32], flags [cap u8], mvuRight [cap f32], mvDepth [cap f32])."""
        x, f, u, d = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        _check(lib().vslam_stereo_points_buffers(self._h, pair, C.byref(x), C.byref(f),
                                                 C.byref(u) if with_stereo else None,
                                                 C.byref(d) if with_stereo else None))
        return x.value, f.value, u.value, d.value

    def set_tuning(self, **kw):
        """change per-call switches of this context (vslam_fe_set_tuning), e.g. set_tuning(sbp_sequential=1)"""
        t = make_tuning(**kw)
        lib().vslam_fe_set_tuning.argtypes = [C.c_void_p, C.c_void_p]
        _check(lib().vslam_fe_set_tuning(self._h, C.byref(t)))

    def octree_stats(self):
        """(problems, split_below_grid, last_level_masks): (slot, level) quadtree problems distributed on the device so far,
        on how many of them nodes were split below the kernel's fine grid, and the level bit masks of the last pass's slots"""
        a, b = C.c_ulonglong(), C.c_ulonglong()
        m = (C.c_uint32 * MAX_BATCH)()
        lib().vslam_fe_octree_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _check(lib().vslam_fe_octree_stats(self._h, C.byref(a), C.byref(b), m))
        return a.value, b.value, [int(v) for v in m]

    def delivery_stats(self):
        """(transfers, bytes): copy operations the extraction / SearchForInitialization paths sent to the host so far"""
        a, b = C.c_ulonglong(), C.c_ulonglong()
        lib().vslam_fe_delivery_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _check(lib().vslam_fe_delivery_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_profiling(self, on=True):
        _check(lib().vslam_fe_set_profiling(self._h, int(on)))

    def get_profile(self):
        ms = (C.c_double * 5)()
        b, im = C.c_long(), C.c_long()
        _check(lib().vslam_fe_get_profile(self._h, ms, C.byref(b), C.byref(im)))
        return dict(pyramid_ms=ms[0], fast_ms=ms[1], blur_ms=ms[2], describe_ms=ms[3], octree_ms=ms[4],
                    batches=b.value, images=im.value)


class FMatcher:
    """vi_slam::geometry::FMatcher (include/vi_slam/geometry/fmatcher.h:70-147), hot-path subset."""

    TH_HIGH, TH_LOW, HISTO_LENGTH = 100, 50, 30  # fmatcher.cpp:313-315

    def __init__(self, extractor, nnratio=0.6, checkOri=True):
        self.fe = extractor
        self.mfNNratio = nnratio
        self.mbCheckOrientation = checkOri

    def hamming_top2(self, dev_q, nq, dev_t, nt):
        """All-pairs DescriptorDistance (fmatcher.cpp:2859-2875) -> two nearest per query."""
        idx = np.zeros((max(nq, 1), 2), np.int32)
        dist = np.zeros((max(nq, 1), 2), np.int32)
        _check(lib().vslam_hamming_top2(self.fe._h, dev_q, nq, dev_t, nt, _p(idx), _p(dist)))
        return idx[:nq], dist[:nq]

    def hamming_top2_batch(self, problems):
        """problems: list of (dev_q, nq, dev_t, nt) -- independent brute-force matches in ONE launch (vslam_hamming_top2_batch)
        -> list of (idx2[nq, 2], dist2[nq, 2])"""
        n = len(problems)
        idx = [np.zeros((max(p[1], 1), 2), np.int32) for p in problems]
        dist = [np.zeros((max(p[1], 1), 2), np.int32) for p in problems]
        vp, ip = C.c_void_p * n, C.c_int32 * n
        L = lib()
        L.vslam_hamming_top2_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _check(L.vslam_hamming_top2_batch(self.fe._h, n, vp(*[p[0] for p in problems]), ip(*[p[1] for p in problems]),
                                          vp(*[p[2] for p in problems]), ip(*[p[3] for p in problems]),
                                          vp(*[a.ctypes.data for a in idx]), vp(*[a.ctypes.data for a in dist])))
        return [(idx[i][:problems[i][1]], dist[i][:problems[i][1]]) for i in range(n)]

    def hamming_top2_batch_async(self, problems):
        """enqueue the same launch again without waiting or copying (profilers; sizes as in a previous hamming_top2_batch)"""
        n = len(problems)
        vp, ip = C.c_void_p * n, C.c_int32 * n
        L = lib()
        L.vslam_hamming_top2_batch_dev_async.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _check(L.vslam_hamming_top2_batch_dev_async(self.fe._h, n, vp(*[p[0] for p in problems]), ip(*[p[1] for p in problems]),
                                                    vp(*[p[2] for p in problems]), ip(*[p[3] for p in problems])))

    def ComputeStereoFishEyeCandidates(self, dev_desc_left, n_left, mono_left, dev_desc_right, n_right, mono_right):
        """Frame::ComputeStereoFishEyeMatches (frame.cpp:1149-1174) up to the ratio test: -> (left_to_right[n_left],
        best_dist, second_dist, descMatches)"""
        l2r = np.zeros(max(n_left, 1), np.int32)
        d0 = np.zeros(max(n_left, 1), np.int32)
        d1 = np.zeros(max(n_left, 1), np.int32)
        nc = C.c_int()
        L = lib()
        L.vslam_stereo_fisheye_candidates.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _check(L.vslam_stereo_fisheye_candidates(self.fe._h, dev_desc_left, n_left, mono_left, dev_desc_right, n_right,
                                                 mono_right, _p(l2r), _p(d0), _p(d1), C.byref(nc)))
        return l2r[:n_left], d0[:n_left], d1[:n_left], nc.value

    def hamming_matrix(self, dev_q, nq, dev_t, nt):
        out = np.zeros((max(nq, 1), max(nt, 1)), np.uint8)
        _check(lib().vslam_hamming_matrix(self.fe._h, dev_q, nq, dev_t, nt, _p(out)))
        return out[:nq, :nt]

    def SearchForInitialization(self, kps1, dev_desc1, kps2, dev_desc2, vbPrevMatched, windowSize=10,
                                img_size=None):
        """FMatcher::SearchForInitialization (fmatcher.cpp:983-1098).

        kps1/kps2: host keypoint arrays; dev_desc1/2: device addresses of their descriptors.
        Returns (nmatches, vnMatches12, updated vbPrevMatched)."""
        kps1 = np.ascontiguousarray(kps1, KP_DTYPE)
        kps2 = np.ascontiguousarray(kps2, KP_DTYPE)
        pm = np.ascontiguousarray(vbPrevMatched, np.float32).copy()
        m = np.full(max(len(kps1), 1), -1, np.int32)
        nm = C.c_int(0)
        w, h = img_size or (self.fe.width, self.fe.height)
        _check(lib().vslam_search_for_initialization(self.fe._h, _p(kps1), dev_desc1, len(kps1), _p(kps2),
                                                     dev_desc2, len(kps2), w, h, _p(pm), _p(m), windowSize,
                                                     self.mfNNratio, int(self.mbCheckOrientation),
                                                     C.byref(nm)))
        return nm.value, m[:len(kps1)], pm

    # ---- device-resident form: nothing but device pointers go in, one kernel, results later
    def search_init_dev_async(self, jobs, windowSize=10, img_size=None):
        """jobs: list of (dev_kps1, dev_desc1, dev_n1, dev_kps2, dev_desc2, dev_n2, dev_prev_or_0) device
        addresses.  Enqueues the whole matcher for all pairs on the extractor's stream and returns."""
        n = len(jobs)
        arr = jobs if isinstance(jobs, C.Array) else self.make_init_jobs(jobs)
        w, h = img_size or (self.fe.width, self.fe.height)
        self._init_n = n
        _check(lib().vslam_search_init_dev_async(self.fe._h, n, arr, w, h, windowSize, self.mfNNratio,
                                                 int(self.mbCheckOrientation)))

    @staticmethod
    def make_init_jobs(jobs):
        """ctypes job array for search_init_dev_async (build once when the device addresses are fixed)."""
        arr = (_InitJob * len(jobs))()
        for j, t in enumerate(jobs):
            arr[j] = _InitJob(*[C.c_void_p(v or None) for v in t])
        return arr

    def search_init_dev_wait(self, n1, want_prev=False):
        """-> list of (nmatches, vnMatches12[n1[j]], vbPrevMatched or None)."""
        n = self._init_n
        cap = self.fe.cap
        if getattr(self, "_m_buf", None) is None or self._m_buf.shape[0] < n:
            self._m_buf = np.zeros((MAX_BATCH, cap), np.int32)
            self._p_buf = np.zeros((MAX_BATCH, cap, 2), np.float32)
            self._m_ptrs = (C.c_void_p * MAX_BATCH)(*[self._m_buf[i].ctypes.data for i in range(MAX_BATCH)])
            self._p_ptrs = (C.c_void_p * MAX_BATCH)(*[self._p_buf[i].ctypes.data for i in range(MAX_BATCH)])
        nm = (C.c_int * n)()
        _check(lib().vslam_search_init_dev_wait(self.fe._h, (C.c_int * n)(*n1), self._m_ptrs,
                                                self._p_ptrs if want_prev else None, nm))
        return [(nm[j], self._m_buf[j, :n1[j]], self._p_buf[j, :n1[j]] if want_prev else None) for j in range(n)]

    # ---- tracking matcher of TrackWithMotionModel (tracking.cpp:2728)
    def SearchByProjection(self, Tcw, Tlw, cam, th, last_kps, last_flags, last_x3dw, mp_desc, dev_cur_kps,
                           dev_cur_desc, n_cur, cur_u_right=None, bMono=False, img_size=None, occupied=None,
                           gemm_float=False):
        """FMatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) (fmatcher.cpp:2471-2687, pinhole).
        Tcw/Tlw: 3x4 poses [R|t] of the current / last frame; cam = (fx, fy, cx, cy, mbf, mb).
        last_flags: bit0 = has a non-outlier MapPoint, bit1 = that MapPoint has observations.
        -> (nmatches, match_cur[n_cur]) with match_cur[i2] = last-frame index or -1."""
        Tcw = np.ascontiguousarray(np.asarray(Tcw, np.float32).reshape(-1)[:12])
        Tlw = np.ascontiguousarray(np.asarray(Tlw, np.float32).reshape(-1)[:12])
        fx, fy, cx, cy, mbf, mb = [float(v) for v in cam]
        fwd, bwd = C.c_int(), C.c_int()
        _check(lib().vslam_projection_direction(_p(Tcw), _p(Tlw), C.c_float(mb), int(bMono), int(gemm_float),
                                                C.byref(fwd), C.byref(bwd)))
        w, h = img_size or (self.fe.width, self.fe.height)
        P = _ProjParams()
        for i in range(12):
            P.Tcw[i] = float(Tcw[i])
        P.fx, P.fy, P.cx, P.cy, P.mbf, P.th = fx, fy, cx, cy, mbf, float(th)
        P.forward, P.backward = fwd.value, bwd.value
        P.check_orientation = int(self.mbCheckOrientation)
        P.img_w, P.img_h, P.gemm_float = w, h, int(gemm_float)
        last_kps = np.ascontiguousarray(last_kps, KP_DTYPE)
        fl = np.ascontiguousarray(last_flags, np.uint8)
        xw = np.ascontiguousarray(last_x3dw, np.float32)
        md = np.ascontiguousarray(mp_desc, np.uint8)
        ur = None if cur_u_right is None else np.ascontiguousarray(cur_u_right, np.float32)
        oc = None if occupied is None else np.ascontiguousarray(occupied, np.uint8)
        m = np.full(max(n_cur, 1), -1, np.int32)
        nm = C.c_int(0)
        _check(lib().vslam_search_by_projection_frame(
            self.fe._h, C.byref(P), _p(last_kps), len(last_kps), _p(fl), _p(xw), _p(md), C.c_void_p(dev_cur_kps),
            C.c_void_p(dev_cur_desc), n_cur, _p(ur) if ur is not None else None, _p(oc) if oc is not None else None,
            _p(m), C.byref(nm)))
        return nm.value, m[:n_cur], (bool(fwd.value), bool(bwd.value))

    def SearchByProjectionMapPoints(self, mps, mp_desc, dev_cur_kps, dev_cur_desc, n_cur, cur_u_right=None, th=1.0,
                                    occupied=None, img_size=None):
        """FMatcher::SearchByProjection(F, vpMapPoints, th, ...) (fmatcher.cpp:321-411, pinhole): mps is a
        MP_TRACK_DTYPE array (what Frame::isInFrustum left in each MapPoint).  -> (nmatches, match_cur[n_cur])."""
        mps = np.ascontiguousarray(mps, MP_TRACK_DTYPE)
        md = np.ascontiguousarray(mp_desc, np.uint8)
        ur = None if cur_u_right is None else np.ascontiguousarray(cur_u_right, np.float32)
        oc = None if occupied is None else np.ascontiguousarray(occupied, np.uint8)
        w, h = img_size or (self.fe.width, self.fe.height)
        m = np.full(max(n_cur, 1), -1, np.int32)
        nm = C.c_int(0)
        _check(lib().vslam_search_by_projection_mappoints(
            self.fe._h, _p(mps), _p(md), len(mps), C.c_void_p(dev_cur_kps), C.c_void_p(dev_cur_desc), n_cur,
            _p(ur) if ur is not None else None, _p(oc) if oc is not None else None, w, h, C.c_float(th),
            C.c_float(self.mfNNratio), _p(m), C.byref(nm)))
        return nm.value, m[:n_cur]

    @staticmethod
    def make_sbp_jobs(jobs, check_orientation=True):
        """jobs: list of dicts with keys Tcw, cam=(fx,fy,cx,cy,mbf), th, forward, backward, img=(w,h) and the device
        addresses last_kps, n_last, last_flags, last_x3dw, mp_desc, cur_kps, cur_desc, n_cur, cur_u_right (or 0),
        cur_occupied (or 0).  -> ctypes array for search_by_projection_dev_async."""
        arr = (_SbpJob * len(jobs))()
        for j, d in enumerate(jobs):
            P = arr[j].p
            T = np.asarray(d["Tcw"], np.float32).reshape(-1)
            for i in range(12):
                P.Tcw[i] = float(T[i])
            P.fx, P.fy, P.cx, P.cy, P.mbf = [float(v) for v in d["cam"][:5]]
            P.th = float(d["th"])
            P.forward, P.backward = int(d.get("forward", 0)), int(d.get("backward", 0))
            P.check_orientation = int(check_orientation)
            P.img_w, P.img_h = d["img"]
            P.gemm_float = int(d.get("gemm_float", 0))
            for k in ("last_kps", "n_last", "last_flags", "last_x3dw", "mp_desc", "cur_kps", "cur_desc", "n_cur",
                      "cur_u_right", "cur_occupied"):
                setattr(arr[j], "dev_" + k, d.get(k) or None)
        return arr

    def search_by_projection_dev_async(self, jobs):
        arr = jobs if isinstance(jobs, C.Array) else self.make_sbp_jobs(jobs, self.mbCheckOrientation)
        self._sbp_n = len(arr)
        _check(lib().vslam_search_by_projection_dev_async(self.fe._h, len(arr), arr))

    def search_by_projection_dev_wait(self, n_cur):
        """-> list of (nmatches, match_cur[n_cur[j]])."""
        n = self._sbp_n
        cap = self.fe.cap
        if getattr(self, "_sbp_buf", None) is None:
            self._sbp_buf = np.zeros((16, cap), np.int32)
            self._sbp_ptrs = (C.c_void_p * 16)(*[self._sbp_buf[i].ctypes.data for i in range(16)])
        nm = (C.c_int * n)()
        _check(lib().vslam_search_by_projection_dev_wait(self.fe._h, (C.c_int * n)(*n_cur), self._sbp_ptrs, nm))
        return [(nm[j], self._sbp_buf[j, :n_cur[j]]) for j in range(n)]

    def SearchByBoW(self, kf_kps, dev_kf_desc, kf_flags, kf_fv, f_kps, dev_f_desc, f_fv):
        """FMatcher::SearchByBoW(pKF, F, vpMapPointMatches) (fmatcher.cpp:546-748, pinhole).  *_fv: dicts with
        fv_nodes / fv_off / fv_feat.  -> (nmatches, match_f[nF] = KeyFrame feature index or -1)."""
        kf_kps = np.ascontiguousarray(kf_kps, KP_DTYPE)
        f_kps = np.ascontiguousarray(f_kps, KP_DTYPE)
        kfl = np.ascontiguousarray(kf_flags, np.uint8)
        a = [np.ascontiguousarray(kf_fv[k], np.int32) for k in ("fv_nodes", "fv_off", "fv_feat")]
        b = [np.ascontiguousarray(f_fv[k], np.int32) for k in ("fv_nodes", "fv_off", "fv_feat")]
        m = np.full(max(len(f_kps), 1), -1, np.int32)
        nm = C.c_int(0)
        _check(lib().vslam_search_by_bow(self.fe._h, _p(kf_kps), C.c_void_p(dev_kf_desc), _p(kfl), len(kf_kps), _p(a[0]),
                                         _p(a[1]), _p(a[2]), len(a[0]), _p(f_kps), C.c_void_p(dev_f_desc), len(f_kps),
                                         _p(b[0]), _p(b[1]), _p(b[2]), len(b[0]), C.c_float(self.mfNNratio),
                                         int(self.mbCheckOrientation), _p(m), C.byref(nm)))
        return nm.value, m[:len(f_kps)]

    def SearchByBoWKeyFrames(self, kps1, dev_desc1, flags1, fv1, kps2, dev_desc2, flags2, fv2):
        """FMatcher::SearchByBoW(pKF1, pKF2, vpMatches12) (fmatcher.cpp:1100-1240) -> (nmatches, match12[n1])."""
        kps1 = np.ascontiguousarray(kps1, KP_DTYPE)
        kps2 = np.ascontiguousarray(kps2, KP_DTYPE)
        f1 = np.ascontiguousarray(flags1, np.uint8)
        f2 = np.ascontiguousarray(flags2, np.uint8)
        a = [np.ascontiguousarray(fv1[k], np.int32) for k in ("fv_nodes", "fv_off", "fv_feat")]
        b = [np.ascontiguousarray(fv2[k], np.int32) for k in ("fv_nodes", "fv_off", "fv_feat")]
        m = np.full(max(len(kps1), 1), -1, np.int32)
        nm = C.c_int(0)
        _check(lib().vslam_search_by_bow_keyframes(self.fe._h, _p(kps1), C.c_void_p(dev_desc1), _p(f1), len(kps1), _p(a[0]),
                                                   _p(a[1]), _p(a[2]), len(a[0]), _p(kps2), C.c_void_p(dev_desc2), _p(f2),
                                                   len(kps2), _p(b[0]), _p(b[1]), _p(b[2]), len(b[0]),
                                                   C.c_float(self.mfNNratio), int(self.mbCheckOrientation), _p(m),
                                                   C.byref(nm)))
        return nm.value, m[:len(kps1)]

    def SearchByProjectionKeyFrame(self, Tcw, Ow, cam, th, ORBdist, log_scale_factor, kf_kps, mp_flags, mp_x3dw,
                                   mp_min_dist, mp_max_dist, mp_desc, dev_cur_kps, dev_cur_desc, n_cur, occupied=None,
                                   img_size=None, gemm_float=False):
        """FMatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (fmatcher.cpp:2689-2811), the
        relocalisation matcher.  cam = (fx, fy, cx, cy).  -> (nmatches, match_cur[n_cur] = pKF keypoint index or -1)."""
        kk = np.ascontiguousarray(kf_kps, KP_DTYPE)
        fl = np.ascontiguousarray(mp_flags, np.uint8)
        x = np.ascontiguousarray(mp_x3dw, np.float32)
        mn = np.ascontiguousarray(mp_min_dist, np.float32)
        mx = np.ascontiguousarray(mp_max_dist, np.float32)
        md = np.ascontiguousarray(mp_desc, np.uint8)
        oc = None if occupied is None else np.ascontiguousarray(occupied, np.uint8)
        O = np.ascontiguousarray(Ow, np.float32).reshape(3)
        w, h = img_size or (self.fe.width, self.fe.height)
        P = _ProjParams()
        T = np.asarray(Tcw, np.float32).reshape(-1)
        for k in range(12):
            P.Tcw[k] = float(T[k])
        P.fx, P.fy, P.cx, P.cy = [float(v) for v in cam[:4]]
        P.mbf, P.th = 0.0, float(th)
        P.forward = P.backward = 0
        P.check_orientation, P.img_w, P.img_h, P.gemm_float = int(self.mbCheckOrientation), int(w), int(h), int(gemm_float)
        m = np.full(max(n_cur, 1), -1, np.int32)
        nm = C.c_int(0)
        _check(lib().vslam_search_by_projection_keyframe(
            self.fe._h, C.byref(P), _p(O), C.c_float(log_scale_factor), int(ORBdist), _p(kk), len(kk), _p(fl), _p(x), _p(mn),
            _p(mx), _p(md), C.c_void_p(dev_cur_kps), C.c_void_p(dev_cur_desc), n_cur, _p(oc) if oc is not None else None,
            _p(m), C.byref(nm)))
        return nm.value, m[:n_cur]

    def SearchByProjectionSim3(self, Tcw, Ow, cam, th, ratioHamming, log_scale_factor, mp_flags, mp_x3dw, mp_normals,
                               mp_min_dist, mp_max_dist, mp_desc, dev_kf_kps, dev_kf_desc, n_kf, matched=None,
                               img_size=None, proj_variant=0, gemm_float=False):
        """FMatcher::SearchByProjection(pKF, Scw, vpPoints, [vpPointsKFs,] vpMatched, [vpMatchedKF,] th, ratioHamming)
        (fmatcher.cpp:750-863; proj_variant=1: :865-981), the loop-closing matchers.  Tcw = Rcw | tcw after the
        decomposition of Scw.  -> (nmatches, match_kf[n_kf] = candidate index or -1)."""
        fl = np.ascontiguousarray(mp_flags, np.uint8)
        x = np.ascontiguousarray(mp_x3dw, np.float32)
        nr = np.ascontiguousarray(mp_normals, np.float32)
        mn = np.ascontiguousarray(mp_min_dist, np.float32)
        mx = np.ascontiguousarray(mp_max_dist, np.float32)
        md = np.ascontiguousarray(mp_desc, np.uint8)
        mt = None if matched is None else np.ascontiguousarray(matched, np.uint8)
        O = np.ascontiguousarray(Ow, np.float32).reshape(3)
        w, h = img_size or (self.fe.width, self.fe.height)
        P = _ProjParams()
        T = np.asarray(Tcw, np.float32).reshape(-1)
        for k in range(12):
            P.Tcw[k] = float(T[k])
        P.fx, P.fy, P.cx, P.cy = [float(v) for v in cam[:4]]
        P.mbf, P.th = 0.0, float(int(th))
        P.forward = P.backward = P.check_orientation = 0
        P.img_w, P.img_h, P.gemm_float = int(w), int(h), int(gemm_float)
        m = np.full(max(n_kf, 1), -1, np.int32)
        nm = C.c_int(0)
        _check(lib().vslam_search_by_projection_sim3(
            self.fe._h, C.byref(P), _p(O), C.c_float(log_scale_factor), C.c_float(ratioHamming), int(proj_variant), _p(fl),
            _p(x), _p(nr), _p(mn), _p(mx), _p(md), len(fl), C.c_void_p(dev_kf_kps), C.c_void_p(dev_kf_desc), n_kf,
            _p(mt) if mt is not None else None, _p(m), C.byref(nm)))
        return nm.value, m[:n_kf]

    def SearchForTriangulation(self, kps1, dev_desc1, has_mp1, u_right1, fv1, kps2, dev_desc2, has_mp2, u_right2, fv2,
                               F12, ep, bOnlyStereo=False, bCoarse=False):
        """FMatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo, bCoarse)
        (fmatcher.cpp:1242-1482 / :1484-1725, pinhole).  F12 as Pinhole::epipolarConstrain builds it, ep the epipole of
        pKF1's centre in pKF2.  -> (nmatches, vMatchedPairs as an (n, 2) array, match12[n1])."""
        kps1 = np.ascontiguousarray(kps1, KP_DTYPE)
        kps2 = np.ascontiguousarray(kps2, KP_DTYPE)
        f1 = np.ascontiguousarray(has_mp1, np.uint8)
        f2 = np.ascontiguousarray(has_mp2, np.uint8)
        u1 = np.ascontiguousarray(u_right1, np.float32)
        u2 = np.ascontiguousarray(u_right2, np.float32)
        a = [np.ascontiguousarray(fv1[k], np.int32) for k in ("fv_nodes", "fv_off", "fv_feat")]
        b = [np.ascontiguousarray(fv2[k], np.int32) for k in ("fv_nodes", "fv_off", "fv_feat")]
        P = _TriParams()
        P.F12[:] = [float(v) for v in np.asarray(F12, np.float32).reshape(9)]
        P.ep_x, P.ep_y = float(ep[0]), float(ep[1])
        P.only_stereo, P.coarse, P.check_orientation = int(bOnlyStereo), int(bCoarse), int(self.mbCheckOrientation)
        m = np.full(max(len(kps1), 1), -1, np.int32)
        nm = C.c_int(0)
        _check(lib().vslam_search_for_triangulation(
            self.fe._h, C.byref(P), _p(kps1), C.c_void_p(dev_desc1), _p(f1), _p(u1), len(kps1), _p(a[0]), _p(a[1]),
            _p(a[2]), len(a[0]), _p(kps2), C.c_void_p(dev_desc2), _p(f2), _p(u2), len(kps2), _p(b[0]), _p(b[1]),
            _p(b[2]), len(b[0]), _p(m), C.byref(nm)))
        m = m[:len(kps1)]
        i1 = np.nonzero(m >= 0)[0]
        return nm.value, np.stack([i1, m[i1]], 1).astype(np.int64), m

    def FuseSearch(self, points, mp_desc, dev_kf_kps, dev_kf_desc, n_kf, kf_u_right, Rcw, tcw, Ow, cam, th,
                   log_scale_factor, img_size=None, sim3=False, gemm_float=False, Rb=None, tb=None):
        """The search half of FMatcher::Fuse (fmatcher.cpp:1918-2119; sim3=True: the Scw overload :2121-2243): points is
        a FUSE_POINT_DTYPE array, cam = (fx, fy, cx, cy, bf).  -> (best_idx[n], best_dist[n]); the caller applies
        best_dist <= TH_LOW and the map updates in order."""
        pts = np.ascontiguousarray(points, FUSE_POINT_DTYPE)
        md = np.ascontiguousarray(mp_desc, np.uint8)
        ur = None if kf_u_right is None else np.ascontiguousarray(kf_u_right, np.float32)
        w, h = img_size or (self.fe.width, self.fe.height)
        P = _FuseParams()
        P.Rcw[:] = [float(v) for v in np.asarray(Rcw, np.float32).reshape(9)]
        P.tcw[:] = [float(v) for v in np.asarray(tcw, np.float32).reshape(3)]
        P.Ow[:] = [float(v) for v in np.asarray(Ow, np.float32).reshape(3)]
        P.fx, P.fy, P.cx, P.cy, P.bf = [float(v) for v in cam]
        P.th, P.log_scale_factor, P.img_w, P.img_h = float(th), float(log_scale_factor), int(w), int(h)
        P.sim3, P.gemm_float = int(sim3), int(gemm_float)
        if Rb is not None:
            P.Rb[:] = [float(v) for v in np.asarray(Rb, np.float32).reshape(9)]
            P.tb[:] = [float(v) for v in np.asarray(tb, np.float32).reshape(3)]
        bi = np.full(max(len(pts), 1), -1, np.int32)
        bd = np.full(max(len(pts), 1), 256, np.int32)
        _check(lib().vslam_fuse_search(self.fe._h, C.byref(P), _p(pts), _p(md), len(pts), C.c_void_p(dev_kf_kps),
                                       C.c_void_p(dev_kf_desc), n_kf, _p(ur) if ur is not None else None, _p(bi),
                                       _p(bd)))
        return bi[:len(pts)], bd[:len(pts)]

    def SearchBySim3(self, pts1, desc1, dev_kps1, dev_desc1, n1, R1w, t1w, pts2, desc2, dev_kps2, dev_desc2, n2, R2w, t2w,
                     s12, R12, t12, th, cam, log_scale_factor, img_size=None, gemm_float=False):
        """FMatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th) (fmatcher.cpp:2245-2469).  pts1 / pts2:
        FUSE_POINT_DTYPE records per keypoint of pKF1 / pKF2 (valid = has a MapPoint, not bad, not matched yet).
        -> (nFound, match12[n1] = idx2 or -1)."""
        R12 = np.asarray(R12, np.float32).reshape(3, 3)
        t12 = np.asarray(t12, np.float32).reshape(3)
        sR12 = (np.float32(s12) * R12).astype(np.float32)
        sR21 = ((1.0 / s12) * R12.T).astype(np.float32)  # cv::Mat algebra on the caller's side, :2262-2264
        t21 = (-(sR21.astype(np.float64) @ t12.astype(np.float64))).astype(np.float32)
        z3, camb = np.zeros(3, np.float32), tuple(cam[:4]) + (0.0,)
        b1, d1 = self.FuseSearch(pts1, desc1, dev_kps2, dev_desc2, n2, None, R1w, t1w, z3, camb, th, log_scale_factor,
                                 img_size, 2, gemm_float, sR21, t21)
        b2, d2 = self.FuseSearch(pts2, desc2, dev_kps1, dev_desc1, n1, None, R2w, t2w, z3, camb, th, log_scale_factor,
                                 img_size, 2, gemm_float, sR12, t12)
        vn1 = np.where((b1 >= 0) & (d1 <= 100), b1, -1)
        vn2 = np.where((b2 >= 0) & (d2 <= 100), b2, -1)
        m12 = np.full(len(vn1), -1, np.int32)
        ok = vn1 >= 0
        idx = np.nonzero(ok)[0]
        agree = vn2[vn1[idx]] == idx
        m12[idx[agree]] = vn1[idx[agree]]
        return int(agree.sum()), m12, (sR21, t21, sR12)

    def search_init_fallbacks(self):
        """Diagnostics: queries whose whole window had to be re-scanned since the last call (read-and-reset)."""
        c = C.c_int()
        _check(lib().vslam_dbg_search_init_fallbacks(self.fe._h, C.byref(c)))
        return c.value

    def search_init_replay_stats(self):
        """Diagnostics: (rounds, queries, pairs) of the replay wave since the last call (read-and-reset)."""
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        _check(lib().vslam_dbg_search_init_replay_stats(self.fe._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def SearchForInitializationBatch(self, pairs, windowSize=10, img_size=None):
        """Several independent SearchForInitialization problems in one pass of the kernels.
        pairs: list of (kps1, dev_desc1, kps2, dev_desc2, vbPrevMatched).  Returns a list of
        (nmatches, vnMatches12, vbPrevMatched)."""
        npairs = len(pairs)
        k1 = [np.ascontiguousarray(p[0], KP_DTYPE) for p in pairs]
        k2 = [np.ascontiguousarray(p[2], KP_DTYPE) for p in pairs]
        pm = [np.ascontiguousarray(p[4], np.float32).copy().reshape(-1, 2) for p in pairs]
        for j in range(npairs):  # ctypes needs a valid address even for empty frames
            if len(k1[j]) == 0:
                pm[j] = np.zeros((1, 2), np.float32)
        m = [np.full(max(len(k), 1), -1, np.int32) for k in k1]
        vpa = C.c_void_p * npairs
        ia = C.c_int * npairs
        nm = ia()
        w, h = img_size or (self.fe.width, self.fe.height)
        dummy = np.zeros(1, KP_DTYPE)
        _check(lib().vslam_search_for_initialization_batch(
            self.fe._h, npairs, vpa(*[(k if len(k) else dummy).ctypes.data for k in k1]),
            vpa(*[(p[1] or dummy.ctypes.data) for p in pairs]), ia(*[len(k) for k in k1]),
            vpa(*[(k if len(k) else dummy).ctypes.data for k in k2]),
            vpa(*[(p[3] or dummy.ctypes.data) for p in pairs]), ia(*[len(k) for k in k2]), w, h,
            vpa(*[a.ctypes.data for a in pm]), vpa(*[a.ctypes.data for a in m]), windowSize, self.mfNNratio,
            int(self.mbCheckOrientation), nm))
        return [(nm[j], m[j][:len(k1[j])], pm[j][:len(k1[j])]) for j in range(npairs)]


def bow_assemble(weighting, norm, word, weight, nid):
    """Host half of DBoW3::Vocabulary::transform (vslam_bow_assemble, GPU-free): per-feature (word, weight, node)
    -> dict(bow_ids, bow_vals, fv_nodes, fv_off, fv_feat)."""
    n = len(word)
    word = np.ascontiguousarray(word, np.int32)
    weight = np.ascontiguousarray(weight, np.float64)
    nid = np.ascontiguousarray(nid, np.int32)
    bi, bv = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.float64)
    fn, fo, ff = np.zeros(max(n, 1), np.int32), np.zeros(n + 2, np.int32), np.zeros(max(n, 1), np.int32)
    nb, nf = C.c_int(), C.c_int()
    rc = lib().vslam_bow_assemble(weighting, norm, _p(word), _p(weight), _p(nid), n, _p(bi), _p(bv), C.byref(nb), _p(fn),
                                  _p(fo), _p(ff), C.byref(nf))
    if rc:
        raise VslamError(rc, "vslam_bow_assemble: invalid arguments")
    return dict(bow_ids=bi[:nb.value], bow_vals=bv[:nb.value], fv_nodes=fn[:nf.value], fv_off=fo[:nf.value + 1],
                fv_feat=ff[:fo[nf.value]])


VOC_FORMATS = {1: "dbow3-binary", 2: "dbow3-binary-quicklz", 3: "dbow3-text"}


def read_vocabulary_file(path, library=None):
    """DBoW3::Vocabulary::load's parsing half (vslam_voc_file_*; no GPU needed): the flat node table of a vocabulary
    file as a dict in vi_slam_amd.synth.make_vocabulary's layout, plus scoring / n_words / format."""
    L = library if library is not None else lib()
    h = C.c_void_p()
    rc = L.vslam_voc_file_open(os.fsencode(path), C.byref(h))
    if rc != VSLAM_OK:
        raise VslamError(rc, (L.vslam_voc_file_last_error() or b"").decode())
    try:
        iv = [C.c_int() for _ in range(9)]
        L.vslam_voc_file_info(h, *[C.byref(x) for x in iv])
        k, depth, scoring, weighting, norm, n_nodes, n_words, n_child, fmt = (x.value for x in iv)
        ptrs = [C.c_void_p() for _ in range(6)]
        L.vslam_voc_file_arrays(h, *[C.byref(x) for x in ptrs])

        def arr(ptr, n, ctype, dtype):
            if n == 0:
                return np.zeros(0, dtype)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), (n,)).astype(dtype, copy=True)

        return dict(k=k, L=depth, scoring=scoring, weighting=weighting, norm=norm, n_words=n_words,
                    format=VOC_FORMATS.get(fmt, str(fmt)),
                    child_start=arr(ptrs[0], n_nodes, C.c_int32, np.int32),
                    child_count=arr(ptrs[1], n_nodes, C.c_int32, np.int32),
                    child_ids=arr(ptrs[2], n_child, C.c_int32, np.int32),
                    desc=arr(ptrs[3], n_nodes * 32, C.c_uint8, np.uint8).reshape(n_nodes, 32),
                    weight=arr(ptrs[4], n_nodes, C.c_double, np.float64),
                    word_id=arr(ptrs[5], n_nodes, C.c_int32, np.int32))
    finally:
        L.vslam_voc_file_close(h)


class Vocabulary:
    """DBoW3::Vocabulary on the device (flat node arrays, see vi_slam_amd.synth.make_vocabulary for the layout)."""

    def __init__(self, voc, device=0):
        self.weighting, self.norm, self.L = int(voc.get("weighting", 0)), int(voc.get("norm", 1)), int(voc["L"])
        cs = np.ascontiguousarray(voc["child_start"], np.int32)
        cc = np.ascontiguousarray(voc["child_count"], np.int32)
        ci = np.ascontiguousarray(voc["child_ids"], np.int32)
        nd = np.ascontiguousarray(voc["desc"], np.uint8)
        nw = np.ascontiguousarray(voc["weight"], np.float64)
        wi = np.ascontiguousarray(voc["word_id"], np.int32)
        h = C.c_void_p()
        _check(lib().vslam_voc_create(device, self.L, self.weighting, self.norm, len(cs), _p(cs), _p(cc), _p(ci), len(ci),
                                      _p(nd), _p(nw), _p(wi), C.byref(h)))
        self._h = h

    @classmethod
    def load(cls, path, device=0):
        """DBoW3::Vocabulary::load(filename) (Vocabulary.cpp:1084-1112): binary (plain or QuickLZ) or .txt vocabulary."""
        self = cls.__new__(cls)
        h = C.c_void_p()
        _check(lib().vslam_voc_load(device, os.fsencode(path), C.byref(h)))
        self._h = h
        iv = [C.c_int() for _ in range(4)]
        lib().vslam_voc_info(h, *[C.byref(x) for x in iv])
        self.L, self.weighting, self.norm, self.n_nodes = (x.value for x in iv)
        return self

    def close(self):
        if getattr(self, "_h", None):
            lib().vslam_voc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def assemble(self, word, weight, nid):
        """Host half of Vocabulary::transform -> dict(bow_ids, bow_vals, fv_nodes, fv_off, fv_feat)."""
        return bow_assemble(self.weighting, self.norm, word, weight, nid)

    def transform(self, fe, dev_desc, n, levelsup=4):
        """Frame::ComputeBoW for n device-resident descriptors -> per-feature (word, weight, nid) + assembled vectors."""
        w, wt, nd = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.float64), np.zeros(max(n, 1), np.int32)
        _check(lib().vslam_bow_transform(fe._h, self._h, C.c_void_p(dev_desc), n, levelsup, _p(w), _p(wt), _p(nd)))
        out = self.assemble(w[:n], wt[:n], nd[:n])
        out.update(word=w[:n], weight=wt[:n], nid=nd[:n])
        return out

    def transform_slots_async(self, fe, first_slot, nslots, levelsup=4):
        self._pending = (fe, nslots)
        _check(lib().vslam_bow_transform_slots_async(fe._h, self._h, first_slot, nslots, levelsup))

    def transform_slots_wait(self, n, assemble=True):
        fe, ns = self._pending
        cap = fe.cap
        if getattr(self, "_bufs", None) is None or self._bufs[0].shape != (MAX_BATCH, cap):
            self._bufs = (np.zeros((MAX_BATCH, cap), np.int32), np.zeros((MAX_BATCH, cap), np.float64),
                          np.zeros((MAX_BATCH, cap), np.int32))
            self._ptrs = [(C.c_void_p * MAX_BATCH)(*[b[i].ctypes.data for i in range(MAX_BATCH)]) for b in self._bufs]
        _check(lib().vslam_bow_transform_slots_wait(fe._h, (C.c_int * ns)(*n), self._ptrs[0], self._ptrs[1],
                                                    self._ptrs[2]))
        res = []
        for j in range(ns):
            w, wt, nd = self._bufs[0][j, :n[j]], self._bufs[1][j, :n[j]], self._bufs[2][j, :n[j]]
            out = self.assemble(w, wt, nd) if assemble else {}
            out.update(word=w, weight=wt, nid=nd)
            res.append(out)
        return res


def ComputeDistinctiveDescriptors(fe, desc, offsets):
    """MapPoint::ComputeDistinctiveDescriptors (mappoint.cpp:322-390) for many MapPoints: set s owns
    desc[offsets[s]:offsets[s+1]] -> index inside the set of its representative descriptor (-1 if empty)."""
    desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    off = np.ascontiguousarray(offsets, np.int32)
    best = np.full(max(len(off) - 1, 1), -1, np.int32)
    _check(lib().vslam_distinctive_descriptors(fe._h, _p(desc) if len(desc) else None, _p(off), len(off) - 1, _p(best)))
    return best[:len(off) - 1]


def ComputeStereoMatches(feL, slotL, feR, slotR, bf, fx):
    """Frame::ComputeStereoMatches (frame.cpp:823-997) -> (mvuRight, mvDepth) for the left keypoints."""
    _, _, n = feL.slot_buffers(slotL)
    u = np.full(max(n, 1), -1, np.float32)
    d = np.full(max(n, 1), -1, np.float32)
    _check(lib().vslam_stereo_match(feL._h, slotL, feR._h, slotR, bf, fx, _p(u), _p(d)))
    return u[:n], d[:n]


def ComputeStereoMatchesBatch(feL, slotsL, feR, slotsR, bf, fx):
    """ComputeStereoMatches for several pairs in one pass of the kernels -> list of (mvuRight, mvDepth)."""
    npairs = len(slotsL)
    ns = [feL.slot_buffers(s)[2] for s in slotsL]
    us = [np.full(max(n, 1), -1, np.float32) for n in ns]
    ds = [np.full(max(n, 1), -1, np.float32) for n in ns]
    sl = (C.c_int * npairs)(*slotsL)
    sr = (C.c_int * npairs)(*slotsR)
    up = (C.c_void_p * npairs)(*[u.ctypes.data for u in us])
    dp = (C.c_void_p * npairs)(*[d.ctypes.data for d in ds])
    _check(lib().vslam_stereo_match_batch(feL._h, feR._h, npairs, sl, sr, bf, fx, up, dp))
    return [(us[j][:ns[j]], ds[j][:ns[j]]) for j in range(npairs)]


def dbg_sincos(fe, x):
    x = np.ascontiguousarray(x, np.float32)
    s, c = np.zeros_like(x), np.zeros_like(x)
    _check(lib().vslam_dbg_sincos(fe._h, _p(x), len(x), _p(s), _p(c)))
    return s, c


def dbg_logf(fe, x):
    x = np.ascontiguousarray(x, np.float32)
    y = np.zeros_like(x)
    _check(lib().vslam_dbg_logf(fe._h, _p(x), len(x), _p(y)))
    return y


def dbg_fast_atan2(fe, y, x, fma=0):
    y = np.ascontiguousarray(y, np.float32)
    x = np.ascontiguousarray(x, np.float32)
    a = np.zeros_like(x)
    _check(lib().vslam_dbg_fast_atan2(fe._h, _p(y), _p(x), len(x), fma, _p(a)))
    return a
