"""Host-side mirror of the reference front end on top of the C ABI (include/vo_mi355x.h).

The three functions of the reference boundary keep their names and argument meaning
(reference: include/visual_odometry/camera_utility.hpp):

    opt_DetectFeatures(img1, img2)              :742   -> KpAsPoint2f_Match
    optRelativePose(kP, K, R, t)                :633   -> (R, t, inlier, success)
    triangPoints(inl1, inl2, R, t, K)           :288   -> 3xM float64

numpy arrays stand in for cv::Mat / std::vector<Point2f>.  Everything below calls libvo_mi355x.so;
there is no Python or CPU implementation behind it.
"""
from __future__ import annotations

import ctypes as C
import dataclasses

import numpy as np

from . import _abi
from ._abi import DM_DTYPE, KP_DTYPE, VoPairOut, VoParams, default_params

u8p = C.POINTER(C.c_uint8)
f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)
i32p = C.POINTER(C.c_int32)


class VoError(RuntimeError):
    def __init__(self, status: int, lib=None):
        self.status = status
        msg = f"vo status {status}"
        if lib is not None:
            lib.vo_strerror.restype = C.c_char_p
            lib.vo_last_hip_error.restype = C.c_char_p
            msg += ": " + lib.vo_strerror(status).decode()
            hip = lib.vo_last_hip_error().decode()
            if hip:
                msg += " [" + hip + "]"
        super().__init__(msg)


@dataclasses.dataclass
class KpAsPoint2f_Match:
    """struct KpAsPoint2f_Match, camera_utility.hpp:73-78"""
    Kpoints1: np.ndarray  # [n,2] float32
    Kpoints2: np.ndarray  # [n,2] float32
    match: np.ndarray  # DM_DTYPE [n]


def _p(a, t):
    return a.ctypes.data_as(t)


def _img(img):
    img = np.asarray(img)
    if img.dtype != np.uint8 or img.ndim != 2:
        raise ValueError("expected a CV_8UC1 image (2-D uint8)")
    if img.strides[1] != 1:
        img = np.ascontiguousarray(img)
    return img


class Context:
    """One vo_ctx: one GPU, one host thread."""

    def __init__(self, params: VoParams | None = None, device: int = 0, lib_path: str | None = None):
        self.lib = _abi.load_library(lib_path)
        self._h = C.c_void_p()
        self.params = params if params is not None else default_params()
        st = self.lib.vo_ctx_create(C.byref(self.params), int(device), C.byref(self._h))
        if st != 0:
            raise VoError(st, self.lib)

    def cached_batch(self, key, make):
        """A batch that lives as long as the context (the replay reuses its sequence / pair batches from call to call:
        creating and destroying one costs ~4 ms each, hipFree synchronises the device)."""
        cache = self.__dict__.setdefault("_batch_cache", {})
        if key in cache:
            cache[key] = cache.pop(key)  # most recently used last
            return cache[key]
        same = [k for k in cache if k[0] == key[0]]
        for k in same[: max(0, len(same) - 5)]:  # at most six batches of a kind stay alive (the replay uses two at a time)
            cache.pop(k).close()
        cache[key] = make()
        return cache[key]

    def close(self):
        for b in list(self.__dict__.get("_batch_cache", {}).values()):
            b.close()
        self.__dict__.get("_batch_cache", {}).clear()
        if self._h:
            self.lib.vo_ctx_destroy.argtypes = [C.c_void_p]
            self.lib.vo_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, st, ok=(0,)):
        if st not in ok:
            raise VoError(st, self.lib)
        return st

    # ---- inspection
    def integral(self, img):
        img = _img(img)
        h, w = img.shape
        out = np.empty((h + 1, w + 1), np.int32)
        self._chk(self.lib.vo_integral(self._h, _p(img, u8p), h, w, C.c_size_t(img.strides[0]), _p(out, i32p)))
        return out

    def hessian_layer(self, img, octave, layer):
        img = _img(img)
        h, w = img.shape
        det = np.empty((h >> octave, w >> octave), np.float32)
        tr = np.empty_like(det)
        self._chk(self.lib.vo_hessian_layer(self._h, _p(img, u8p), h, w, C.c_size_t(img.strides[0]), octave, layer, _p(det, f32p), _p(tr, f32p)))
        return det, tr

    def hessian_product_det(self, img):
        """det planes of all 20 layers as the product's fused Hessian + NMS kernels hold them: list of 20 arrays."""
        img = _img(img)
        h, w = img.shape
        sizes = [((h >> o), (w >> o)) for o in range(4) for _ in range(5)]
        flat = np.empty(sum(a * b for a, b in sizes), np.float32)
        self._chk(self.lib.vo_hessian_product_det(self._h, _p(img, u8p), h, w, C.c_size_t(img.strides[0]), _p(flat, f32p), C.c_size_t(flat.size)))
        out, off = [], 0
        for a, b in sizes:
            out.append(flat[off : off + a * b].reshape(a, b))
            off += a * b
        return out

    # ---- a1-a6
    def detect_describe(self, img, describe=True, cap=None):
        img = _img(img)
        h, w = img.shape
        cap = cap or self.params.kp_capacity
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, self.descriptor_length), np.float32) if describe else None
        n = C.c_int(0)
        self._chk(self.lib.vo_detect_describe(self._h, _p(img, u8p), h, w, C.c_size_t(img.strides[0]), kps.ctypes.data_as(C.c_void_p),
                                              _p(desc, f32p) if describe else None, cap, C.byref(n)))
        return kps[: n.value].copy(), (desc[: n.value].copy() if describe else None)

    @property
    def descriptor_length(self) -> int:
        """64 floats, or 128 with vo_params.extended (SURF's `extended`, camera_utility.hpp:37)"""
        return 128 if self.params.extended else 64

    # ---- a7
    def match_knn2(self, a, b):
        a = np.ascontiguousarray(a, np.float32)
        b = np.ascontiguousarray(b, np.float32)
        if a.ndim != 2 or b.ndim != 2 or a.shape[1] != self.descriptor_length or b.shape[1] != self.descriptor_length:
            raise ValueError(f"descriptors must be (n, {self.descriptor_length}) float32")
        idx = np.zeros((len(a), 2), np.int32)
        dist = np.zeros((len(a), 2), np.float32)
        self._chk(self.lib.vo_match_knn2(self._h, _p(a, f32p), len(a), _p(b, f32p), len(b), _p(idx, i32p), _p(dist, f32p)))
        return idx, dist

    # ---- opt_DetectFeatures
    def opt_DetectFeatures(self, img1, img2, cap=None) -> KpAsPoint2f_Match:
        img1, img2 = _img(img1), _img(img2)
        if img1.shape != img2.shape:
            raise ValueError("frames of a pair must have the same size")
        h, w = img1.shape
        cap = cap or self.params.match_capacity
        p1 = np.zeros((cap, 2), np.float32)
        p2 = np.zeros((cap, 2), np.float32)
        m = np.zeros(cap, DM_DTYPE)
        n = C.c_int(0)
        self._chk(self.lib.vo_detect_match(self._h, _p(img1, u8p), C.c_size_t(img1.strides[0]), _p(img2, u8p), C.c_size_t(img2.strides[0]),
                                           h, w, _p(p1, f32p), _p(p2, f32p), m.ctypes.data_as(C.c_void_p), cap, C.byref(n)))
        k = n.value
        return KpAsPoint2f_Match(p1[:k].copy(), p2[:k].copy(), m[:k].copy())

    # ---- a10 pieces
    def ransac_subsets(self, seed, count, n_tuples):
        out = np.zeros((n_tuples, 5), np.int32)
        self._chk(self.lib.vo_ransac_subsets(C.c_uint64(seed), count, n_tuples, _p(out, i32p)))
        return out

    def five_point(self, x1, x2):
        x1 = np.ascontiguousarray(x1, np.float64).reshape(-1, 5, 2)
        x2 = np.ascontiguousarray(x2, np.float64).reshape(-1, 5, 2)
        nh = len(x1)
        E = np.zeros((nh, 10, 9), np.float64)
        nm = np.zeros(nh, np.int32)
        self._chk(self.lib.vo_five_point(self._h, _p(x1, f64p), _p(x2, f64p), nh, _p(E, f64p), _p(nm, i32p)))
        return E, nm

    def sampson_masks(self, p1, p2, K, thr_px, E):
        p1 = np.ascontiguousarray(p1, np.float32)
        p2 = np.ascontiguousarray(p2, np.float32)
        K = np.ascontiguousarray(K, np.float64)
        E = np.ascontiguousarray(E, np.float64).reshape(-1, 9)
        n, nm = len(p1), len(E)
        masks = np.zeros((nm, n), np.uint8)
        counts = np.zeros(nm, np.int32)
        self._chk(self.lib.vo_sampson_masks(self._h, _p(p1, f32p), _p(p2, f32p), n, _p(K, f64p), C.c_double(thr_px), _p(E, f64p), nm, _p(masks, u8p), _p(counts, i32p)))
        return masks, counts

    def essential_ransac(self, p1, p2, K, prob=0.99, thr_px=2.0, max_iters=1000, seed=2**64 - 1):
        """findEssentialMat(p1, p2, K, RANSAC, prob, thr, mask): returns (status, E, mask, n_inl, iters)."""
        p1 = np.ascontiguousarray(p1, np.float32)
        p2 = np.ascontiguousarray(p2, np.float32)
        K = np.ascontiguousarray(K, np.float64)
        n = len(p1)
        E = np.zeros(9, np.float64)
        mask = np.zeros(max(n, 1), np.uint8)
        ninl, iters = C.c_int(0), C.c_int(0)
        st = self.lib.vo_essential_ransac(self._h, _p(p1, f32p), _p(p2, f32p), n, _p(K, f64p), C.c_double(prob), C.c_double(thr_px), max_iters,
                                          C.c_uint64(seed), _p(E, f64p), _p(mask, u8p), C.byref(ninl), C.byref(iters))
        self._chk(st, ok=(0, _abi.VO_ERR_TOO_FEW_POINTS, _abi.VO_ERR_NO_MODEL))
        return st, E.reshape(3, 3), mask[:n], ninl.value, iters.value

    def recover_pose(self, E, p1, p2, K, dist_thresh=50.0):
        E = np.ascontiguousarray(E, np.float64)
        p1 = np.ascontiguousarray(p1, np.float32)
        p2 = np.ascontiguousarray(p2, np.float32)
        K = np.ascontiguousarray(K, np.float64)
        R, t, ng = np.zeros(9), np.zeros(3), C.c_int(0)
        self._chk(self.lib.vo_recover_pose(self._h, _p(E, f64p), _p(p1, f32p), _p(p2, f32p), len(p1), _p(K, f64p), C.c_double(dist_thresh),
                                           _p(R, f64p), _p(t, f64p), C.byref(ng)))
        return R.reshape(3, 3), t, ng.value

    # ---- optRelativePose
    def optRelativePose(self, kP, K, R=None, t=None):
        """Returns (status, R, t, inlier: KpAsPoint2f_Match, success).  R, t are the in/out pose of the reference:
        they come back unchanged unless success."""
        p1 = np.ascontiguousarray(kP.Kpoints1, np.float32)
        p2 = np.ascontiguousarray(kP.Kpoints2, np.float32)
        K = np.ascontiguousarray(K, np.float64)
        n = len(p1)
        R = np.eye(3).ravel().copy() if R is None else np.array(R, np.float64).ravel().copy()
        t = np.zeros(3) if t is None else np.array(t, np.float64).ravel().copy()
        i1 = np.zeros((max(n, 1), 2), np.float32)
        i2 = np.zeros((max(n, 1), 2), np.float32)
        ninl, ok = C.c_int(0), C.c_int(0)
        st = self.lib.vo_relative_pose(self._h, _p(p1, f32p), _p(p2, f32p), n, _p(K, f64p), _p(R, f64p), _p(t, f64p), _p(i1, f32p), _p(i2, f32p),
                                       C.byref(ninl), C.byref(ok))
        self._chk(st, ok=(0, _abi.VO_ERR_TOO_FEW_POINTS, _abi.VO_ERR_NO_MODEL))
        k = ninl.value
        m = np.zeros(k, DM_DTYPE)
        m["queryIdx"] = np.arange(k)
        m["trainIdx"] = np.arange(k)
        return st, R.reshape(3, 3), t, KpAsPoint2f_Match(i1[:k].copy(), i2[:k].copy(), m), bool(ok.value)

    # ---- a13 / triangPoints
    def triangulate(self, P0, P1, p1, p2):
        P0 = np.ascontiguousarray(P0, np.float64)
        P1 = np.ascontiguousarray(P1, np.float64)
        p1 = np.ascontiguousarray(p1, np.float32)
        p2 = np.ascontiguousarray(p2, np.float32)
        n = len(p1)
        X4 = np.zeros((4, max(n, 1)), np.float32)
        self._chk(self.lib.vo_triangulate(self._h, _p(P0, f64p), _p(P1, f64p), _p(p1, f32p), _p(p2, f32p), n, _p(X4, f32p)))
        return X4.reshape(-1)[: 4 * n].reshape(4, n) if n else X4[:, :0]

    def triangPoints(self, inl1, inl2, R, t, K):
        p1 = np.ascontiguousarray(inl1, np.float32)
        p2 = np.ascontiguousarray(inl2, np.float32)
        R = np.ascontiguousarray(R, np.float64)
        t = np.ascontiguousarray(t, np.float64)
        K = np.ascontiguousarray(K, np.float64)
        n = len(p1)
        cap = max(n, 1)
        X = np.zeros((3, cap), np.float64)
        m = C.c_int(0)
        self._chk(self.lib.vo_triang_points(self._h, _p(p1, f32p), _p(p2, f32p), n, _p(R, f64p), _p(t, f64p), _p(K, f64p), _p(X, f64p), cap, C.byref(m)))
        return X[:, : m.value].copy()

    def triangPoints_noalloc(self, inl1, inl2, R, t, K):
        """vo_triang_points_noalloc: triangPoints for callers that cannot allocate device memory (static device scratch,
        at most VO_NOALLOC_MAX_POINTS points); same result as triangPoints."""
        p1 = np.ascontiguousarray(inl1, np.float32)
        p2 = np.ascontiguousarray(inl2, np.float32)
        R = np.ascontiguousarray(R, np.float64)
        t = np.ascontiguousarray(t, np.float64)
        K = np.ascontiguousarray(K, np.float64)
        n = len(p1)
        cap = max(n, 1)
        X = np.zeros((3, cap), np.float64)
        m = C.c_int(0)
        self._chk(self.lib.vo_triang_points_noalloc(self._h, _p(p1, f32p), _p(p2, f32p), n, _p(R, f64p), _p(t, f64p), _p(K, f64p), _p(X, f64p), cap,
                                                    C.byref(m)))
        return X[:, : m.value].copy()

    # ---- profiling
    def debug_set(self, key, value):
        """Test hooks (vo_debug_set): key 0 = force the sequential descriptor-window walk."""
        self._chk(self.lib.vo_debug_set(self._h, int(key), int(value)))

    def profile_enable(self, on=True):
        self._chk(self.lib.vo_profile_enable(self._h, int(on)))

    def profile_reset(self):
        self._chk(self.lib.vo_profile_reset(self._h))

    def profile(self):
        self.lib.vo_stage_name.restype = C.c_char_p
        out = {}
        for s in range(self.lib.vo_stage_count()):
            ms, n = C.c_double(0), C.c_int64(0)
            self._chk(self.lib.vo_profile_get(self._h, s, C.byref(ms), C.byref(n)))
            out[self.lib.vo_stage_name(s).decode()] = (ms.value, n.value)
        return out


class Batch:
    """Device-resident batch of frame pairs (vo_batch_*)."""

    def __init__(self, ctx: Context, n_pairs: int, rows: int, cols: int, _create=True):
        self.ctx, self.n_pairs, self.rows, self.cols = ctx, n_pairs, rows, cols
        self._h = C.c_void_p()
        if _create:
            ctx._chk(ctx.lib.vo_batch_create(ctx._h, n_pairs, rows, cols, C.byref(self._h)))

    def close(self):
        if self._h:
            self.ctx.lib.vo_batch_destroy.argtypes = [C.c_void_p]
            self.ctx.lib.vo_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_pair(self, i, img1, img2, K):
        img1, img2 = _img(img1), _img(img2)
        K = np.ascontiguousarray(K, np.float64)
        self.ctx._chk(self.ctx.lib.vo_batch_set_pair(self._h, i, _p(img1, u8p), C.c_size_t(img1.strides[0]), _p(img2, u8p),
                                                     C.c_size_t(img2.strides[0]), _p(K, f64p)))

    def set_k(self, i, K):
        """K of a pair whose frames arrive through Prep.to_batch (vo_batch_set_k)."""
        K = np.ascontiguousarray(K, np.float64)
        self.ctx._chk(self.ctx.lib.vo_batch_set_k(self._h, i, _p(K, f64p)))

    def run(self):
        self.ctx._chk(self.ctx.lib.vo_batch_run(self._h))

    def sync(self):
        self.ctx._chk(self.ctx.lib.vo_batch_sync(self._h))

    def get(self, i, age: int = 0) -> VoPairOut:
        """Record of pair i of the latest run (age 0) or of the run before it (age 1, vo_batch_get_run)."""
        out = VoPairOut()
        self.ctx._chk(self.ctx.lib.vo_batch_get_run(self._h, int(age), i, C.byref(out)))
        return out

    def get_records(self, age: int = 0):
        """Records of ALL pairs of the latest run (age 0) or of the run before it (age 1) in one call (vo_batch_get_run_all)."""
        outs = (VoPairOut * self.n_pairs)()
        self.ctx._chk(self.ctx.lib.vo_batch_get_run_all(self._h, int(age), outs))
        return list(outs)

    def get_points(self, i, cap=None):
        cap = cap or self.ctx.params.match_capacity
        p1 = np.zeros((cap, 2), np.float32)
        p2 = np.zeros((cap, 2), np.float32)
        m = np.zeros(cap, DM_DTYPE)
        mask = np.zeros(cap, np.uint8)
        X = np.zeros((3, cap), np.float64)
        self.ctx._chk(self.ctx.lib.vo_batch_get_points(self._h, i, _p(p1, f32p), _p(p2, f32p), m.ctypes.data_as(C.c_void_p), _p(mask, u8p), cap,
                                                       _p(X, f64p), cap))
        o = self.get(i)
        nm = max(o.n_match, 0)
        return p1[:nm], p2[:nm], m[:nm], mask[:nm], X[:, : max(o.n_world, 0)]


    def get_all(self):
        """(records, points) of EVERY pair of the latest run: one wait and one download (vo_batch_get_points_all).  points[i] =
        (p1, p2, matches, mask, Xk) as get_points(i) returns them - views into shared arrays."""
        outs = self.get_records()
        cap = max(1, max(o.n_match for o in outs))
        capw = max(1, max(o.n_world for o in outs))
        P = self.n_pairs
        p1 = np.zeros((P, cap, 2), np.float32)
        p2 = np.zeros((P, cap, 2), np.float32)
        m = np.zeros((P, cap), DM_DTYPE)
        mask = np.zeros((P, cap), np.uint8)
        X = np.zeros((P, 3, capw), np.float64)
        self.ctx._chk(self.ctx.lib.vo_batch_get_points_all(self._h, _p(p1, f32p), _p(p2, f32p), m.ctypes.data_as(C.c_void_p), _p(mask, u8p), cap,
                                                           _p(X, f64p), capw))
        pts = []
        for i, o in enumerate(outs):
            nm, nw = min(max(o.n_match, 0), cap), max(o.n_world, 0)
            pts.append((p1[i, :nm], p2[i, :nm], m[i, :nm], mask[i, :nm], X[i, :, :nw]))
        return outs, pts


class Sequence(Batch):
    """Device-resident run of consecutive frames (vo_seq_*): every frame is detected and described once, pair p is
    (frame p, frame p + 1) unless set_pair re-pairs it.  run / sync / get / get_points as for Batch."""

    def __init__(self, ctx: Context, n_frames: int, rows: int, cols: int):
        super().__init__(ctx, n_frames - 1, rows, cols, _create=False)
        self.n_frames = n_frames
        ctx._chk(ctx.lib.vo_seq_create(ctx._h, n_frames, rows, cols, C.byref(self._h)))

    def set_frame(self, f, img):
        img = _img(img)
        self.ctx._chk(self.ctx.lib.vo_seq_set_frame(self._h, f, _p(img, u8p), C.c_size_t(img.strides[0])))

    def set_pair(self, i, frame_a, frame_b, K):  # noqa: D102  (different signature from Batch.set_pair on purpose)
        K = np.ascontiguousarray(K, np.float64)
        self.ctx._chk(self.ctx.lib.vo_seq_set_pair(self._h, i, int(frame_a), int(frame_b), _p(K, f64p)))
