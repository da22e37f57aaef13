"""ctypes binding of include/celeste_mi355x.h (the C-ABI drop-in boundary).

`load_library()` fails loudly when the HIP shared library has not been built:
there is no CPU fallback anywhere in this package.
"""
import ctypes as C
import os
from typing import List, Optional, Sequence

import numpy as np

P = 44
STAMP = 51
COEF = 53

OK, ERR_INVALID_ARG, ERR_NONFINITE_INPUT, ERR_NONFINITE_RESULT, ERR_HIP, ERR_NO_DEVICE, ERR_ALLOC, ERR_ABORTED = range(8)
FLAG_GRAD, FLAG_HESS, FLAG_KL, FLAG_FP32, FLAG_SPLIT, FLAG_PACKED_HESS = 1, 2, 4, 8, 16, 32
HP = 990   # doubles of a packed Hessian (upper triangle by columns)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libceleste_mi355x.so")

c_double_p = C.POINTER(C.c_double)
c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)
c_int64_p = C.POINTER(C.c_int64)
c_uint8_p = C.POINTER(C.c_uint8)


class ImageT(C.Structure):
    _fields_ = [("H", C.c_int32), ("W", C.c_int32), ("band", C.c_int32), ("reserved", C.c_int32),
                ("pixels", c_float_p), ("sky", c_float_p), ("nelec_per_nmgy", c_float_p)]


class PatchT(C.Structure):
    _fields_ = [("off_h", C.c_int32), ("off_w", C.c_int32), ("H2", C.c_int32), ("W2", C.c_int32),
                ("bitmap", c_uint8_p), ("wcs_jacobian", C.c_double * 4), ("world_center", C.c_double * 2),
                ("pixel_center", C.c_double * 2), ("psf", c_double_p), ("stamp", C.c_int32),
                ("reserved", C.c_int32)]


class PriorT(C.Structure):
    _fields_ = [("is_star", C.c_double * 2), ("flux_mean", C.c_double * 2), ("flux_var", C.c_double * 2),
                ("k", (C.c_double * 8) * 2), ("color_mean", ((C.c_double * 4) * 8) * 2),
                ("color_cov", ((C.c_double * 16) * 8) * 2), ("gal_radius_px_mean", C.c_double),
                ("gal_radius_px_var", C.c_double)]


class ProblemT(C.Structure):
    _fields_ = [("n_images", C.c_int32), ("n_sources", C.c_int32), ("psf_K", C.c_int32), ("n_stamps", C.c_int32),
                ("images", C.POINTER(ImageT)), ("patches", C.POINTER(PatchT)), ("stamps", c_double_p),
                ("nbr_offsets", c_int64_p), ("nbr_index", c_int32_p), ("prior", C.POINTER(PriorT)),
                ("n_patch_entries", C.c_int64), ("patch_source", c_int32_p), ("patch_image", c_int32_p)]


class WorkStatsT(C.Structure):
    _fields_ = [("n_targets", C.c_int64), ("active_pixel_visits", C.c_int64), ("patch_rows", C.c_int64),
                ("neighbor_links", C.c_int64), ("algorithmic_bytes", C.c_int64), ("record_bytes", C.c_int64),
                ("record_tiles", C.c_int64)]


class OptimConfigT(C.Structure):
    """celeste_optim_config_t: ElboConfig defaults (ElboMaximize.jl:43-49, 95-108)"""
    _fields_ = [("loc_width", C.c_double), ("loc_scale", C.c_double), ("max_iters", C.c_int32),
                ("include_kl", C.c_int32), ("xtol_abs", C.c_double), ("ftol_rel", C.c_double), ("gtol", C.c_double),
                ("initial_delta", C.c_double), ("delta_hat", C.c_double), ("tr_secular_iters", C.c_int32),
                ("reserved", C.c_int32)]


class GroupInfoT(C.Structure):
    """celeste_group_info_t"""
    _fields_ = [("n_members", C.c_int32), ("n_devices", C.c_int32), ("exchange", C.c_int32), ("rccl_ranks", C.c_int32),
                ("devices", C.c_int32 * 16)]


EXCHANGE_RCCL, EXCHANGE_PEER_COPY = 1, 2


class CelesteError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__("celeste_mi355x status %d: %s" % (status, msg))
        self.status = status


# every symbol include/celeste_mi355x.h declares
EXPORTED_SYMBOLS = [
    "celeste_version", "celeste_strerror", "celeste_ctx_create", "celeste_ctx_destroy", "celeste_elbo_eval",
    "celeste_elbo_eval_batch", "celeste_elbo_eval_multi", "celeste_elbo_eval_batch_device", "celeste_ctx_enable_timing",
    "celeste_ctx_last_kernel_ms", "celeste_ctx_last_record_sum_ms", "celeste_ctx_work_stats", "celeste_spline_prefilter",
    "celeste_ctx_spline_coefficients", "celeste_psf_raster",
    "celeste_maximize_batch", "celeste_render_expected", "celeste_optim_stats", "celeste_tr_solve_batch",
    "celeste_images_create", "celeste_images_destroy", "celeste_ctx_create_on",
    "celeste_host_alloc", "celeste_host_free", "celeste_host_register", "celeste_host_unregister",
    "celeste_maximize_batch_device", "celeste_joint_infer",
    "celeste_group_create", "celeste_group_destroy", "celeste_group_info", "celeste_group_elbo_eval_batch",
    "celeste_group_sweep_plan", "celeste_group_sweep", "celeste_group_sweep_wait", "celeste_group_sweep_results",
    "celeste_group_shard_sizes", "celeste_group_enable_timing", "celeste_group_last_sweep_ms", "celeste_group_last_kernel_ms",
    "celeste_group_maximize_batch", "celeste_group_joint_infer", "celeste_group_collectives",
]
ABI_VERSION = 220   # CELESTE_ABI_VERSION of include/celeste_mi355x.h these structs were written against

_lib = None


def load_library(path: Optional[str] = None) -> C.CDLL:
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get("CELESTE_MI355X_LIB") or LIB_PATH
    if not os.path.exists(path):
        raise ImportError(
            "HIP extension %s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % path)
    # PyTorch ships its own HIP/HSA runtime libraries.  Whichever HIP runtime is loaded first serves the whole
    # process, and torch cannot initialise its device after a different one: load torch's first so that tensors,
    # streams and this library share one runtime whatever the import order of the caller.
    try:
        import torch  # noqa: F401
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    except RuntimeError:
        # a forked worker of a process that already initialised its device (synthetic.gen_images with workers > 1):
        # only the host-side entry points (celeste_spline_prefilter) are used there
        pass
    lib = C.CDLL(path)
    vp = C.c_void_p
    lib.celeste_version.restype = C.c_int
    if lib.celeste_version() // 100 != ABI_VERSION // 100:
        raise ImportError("%s has ABI version %d, this binding was written against %d: the struct layouts "
                          "(celeste_optim_config_t ...) may differ -- rebuild the library" % (path, lib.celeste_version(), ABI_VERSION))
    lib.celeste_strerror.restype = C.c_char_p
    lib.celeste_strerror.argtypes = [C.c_int]
    lib.celeste_ctx_create.argtypes = [C.POINTER(ProblemT), C.c_int, C.POINTER(vp)]
    lib.celeste_ctx_destroy.argtypes = [vp]
    lib.celeste_ctx_destroy.restype = None
    lib.celeste_elbo_eval.argtypes = [vp, c_double_p, C.c_int32, C.c_uint32, c_double_p, c_double_p, c_double_p,
                                      c_int64_p, c_int64_p]
    lib.celeste_elbo_eval_batch.argtypes = [vp, c_double_p, C.c_int32, c_int32_p, C.c_uint32, c_double_p,
                                            c_double_p, c_double_p, c_int64_p, c_int32_p]
    lib.celeste_elbo_eval_batch_device.argtypes = [vp, vp, C.c_int32, vp, C.c_uint32, vp, vp, vp, vp, vp, vp]
    lib.celeste_ctx_enable_timing.argtypes = [vp, C.c_int]
    lib.celeste_ctx_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.celeste_elbo_eval_multi.argtypes = [vp, c_double_p, C.c_int32, c_int32_p, C.c_uint32, C.POINTER(C.c_double),
                                            c_double_p, c_double_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.celeste_ctx_last_record_sum_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.celeste_ctx_work_stats.argtypes = [vp, C.c_int32, c_int32_p, C.POINTER(WorkStatsT)]
    lib.celeste_spline_prefilter.argtypes = [c_double_p, c_double_p]
    lib.celeste_ctx_spline_coefficients.argtypes = [vp, C.c_int32, c_double_p]
    lib.celeste_psf_raster.argtypes = [C.c_int, c_double_p, C.c_int32, c_double_p, C.c_int32, c_double_p,
                                       C.c_int32, c_double_p]
    lib.celeste_maximize_batch.argtypes = [vp, c_double_p, c_double_p, c_double_p, C.c_int32, c_int32_p,
                                           C.POINTER(OptimConfigT), c_int32_p, c_int32_p, c_double_p, c_int32_p]
    lib.celeste_maximize_batch_device.argtypes = [vp, vp, vp, vp, C.c_int32, vp, C.POINTER(OptimConfigT), vp, vp, vp, vp, vp]
    lib.celeste_joint_infer.argtypes = [vp, c_double_p, C.c_int32, c_int64_p, c_int32_p, c_double_p,
                                        C.POINTER(OptimConfigT), c_int32_p, c_int32_p, c_double_p, c_int32_p]
    lib.celeste_optim_stats.argtypes = [C.c_int, C.POINTER(C.c_uint64)]
    lib.celeste_tr_solve_batch.argtypes = [C.c_int, C.c_int32, c_double_p, c_double_p, c_double_p, C.c_int32, C.c_int32,
                                           c_double_p, c_double_p, c_int32_p, c_int32_p]
    lib.celeste_render_expected.argtypes = [vp, c_double_p, C.c_int32, c_double_p]
    lib.celeste_images_create.argtypes = [C.c_int32, C.POINTER(ImageT), C.c_int, C.POINTER(vp)]
    lib.celeste_images_destroy.argtypes = [vp]
    lib.celeste_images_destroy.restype = None
    lib.celeste_ctx_create_on.argtypes = [vp, C.POINTER(ProblemT), C.POINTER(vp)]
    lib.celeste_host_alloc.argtypes = [C.c_size_t]
    lib.celeste_host_alloc.restype = vp
    lib.celeste_host_free.argtypes = [vp]
    lib.celeste_host_free.restype = None
    lib.celeste_host_register.argtypes = [vp, C.c_size_t]
    lib.celeste_host_unregister.argtypes = [vp]
    c_float_p_ = C.POINTER(C.c_float)
    lib.celeste_group_create.argtypes = [C.POINTER(ProblemT), C.c_int32, c_int32_p, C.POINTER(vp)]
    lib.celeste_group_destroy.argtypes = [vp]
    lib.celeste_group_destroy.restype = None
    lib.celeste_group_info.argtypes = [vp, C.POINTER(GroupInfoT)]
    lib.celeste_group_collectives.argtypes = [vp, c_int64_p, c_int32_p]
    lib.celeste_group_elbo_eval_batch.argtypes = [vp, c_double_p, C.c_int32, c_int32_p, C.c_uint32, c_double_p, c_double_p,
                                                  c_double_p, c_int64_p, c_int32_p]
    lib.celeste_group_sweep_plan.argtypes = [vp, c_double_p, C.c_int32, c_int32_p, C.c_uint32]
    lib.celeste_group_sweep.argtypes = [vp]
    lib.celeste_group_sweep_wait.argtypes = [vp]
    lib.celeste_group_sweep_results.argtypes = [vp, c_double_p, c_double_p, c_double_p, c_int64_p, c_int32_p]
    lib.celeste_group_shard_sizes.argtypes = [vp, c_int32_p, c_int64_p]
    lib.celeste_group_enable_timing.argtypes = [vp, C.c_int]
    lib.celeste_group_last_sweep_ms.argtypes = [vp, c_float_p_, c_float_p_]
    lib.celeste_group_last_kernel_ms.argtypes = [vp, C.c_int32, c_float_p_]
    lib.celeste_group_maximize_batch.argtypes = [vp, c_double_p, c_double_p, c_double_p, C.c_int32, c_int32_p,
                                                 C.POINTER(OptimConfigT), c_int32_p, c_int32_p, c_double_p, c_int32_p]
    lib.celeste_group_joint_infer.argtypes = [vp, c_double_p, C.c_int32, C.c_int32, c_int64_p, c_int64_p, c_int32_p, c_double_p,
                                              C.POINTER(OptimConfigT), c_int32_p, c_int32_p, c_double_p, c_int32_p, c_int64_p]
    _lib = lib
    return lib


def check(status: int, lib=None):
    if status != OK:
        lib = lib or load_library()
        raise CelesteError(status, lib.celeste_strerror(status).decode())


def _dp(a: np.ndarray):
    return a.ctypes.data_as(c_double_p)


def marshal_image_structs(images, keep: list):
    """Model.Image list -> celeste_image_t array (column-major float32 planes; `keep` holds the buffers alive)."""
    c_images = (ImageT * len(images))()
    for n, im in enumerate(images):
        pix = np.asfortranarray(im.pixels, dtype=np.float32)
        sky = np.asfortranarray(im.sky, dtype=np.float32)
        iota = np.ascontiguousarray(im.nelec_per_nmgy, dtype=np.float32)
        assert sky.shape == pix.shape and iota.shape == (pix.shape[0],)
        keep += [pix, sky, iota]
        ci = c_images[n]
        ci.H, ci.W, ci.band = pix.shape[0], pix.shape[1], int(im.b)
        ci.pixels = pix.ctypes.data_as(c_float_p)
        ci.sky = sky.ctypes.data_as(c_float_p)
        ci.nelec_per_nmgy = iota.ctypes.data_as(c_float_p)
    return c_images


class ImageSet:
    """celeste_images_t: the image planes of a box in HBM, shared by every context created on it -- the reference's
    per-source `ElboArgs(images, patches[[t; neighbors], :], [1])` all borrow the same `images`
    (process_source, ParallelRun.jl:468-488).  Reference counted by the library: `close()` drops this object's
    reference; the planes are released when the last context on the set is closed too."""

    def __init__(self, images, device: int = 0):
        self.lib = load_library()
        self.images = images
        self.device = device
        keep: List[object] = []
        c_images = marshal_image_structs(images, keep)
        h = C.c_void_p()
        check(self.lib.celeste_images_create(len(images), c_images, device, C.byref(h)), self.lib)
        self.handle = h   # the host planes are not needed after the upload

    def close(self):
        if getattr(self, "handle", None):
            self.lib.celeste_images_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- page-locked host arrays for the host-pointer entry points ----------------------------------------------
class _PinnedBlock:
    """One celeste_host_alloc block; returns to the pool (or is freed) when the last array on it dies."""
    __slots__ = ("ptr", "nbytes", "__weakref__")

    def __init__(self, ptr, nbytes):
        self.ptr, self.nbytes = ptr, nbytes

    def __del__(self):
        try:
            _pinned_release(self.ptr, self.nbytes)
        except Exception:
            pass


_pinned_pool = {}          # nbytes -> [ptr, ...]
_pinned_pool_bytes = [0]
PINNED_POOL_LIMIT = 1 << 30


def _pinned_release(ptr, nbytes):
    if _lib is None:
        return
    if _pinned_pool_bytes[0] + nbytes <= PINNED_POOL_LIMIT:
        _pinned_pool.setdefault(nbytes, []).append(ptr)
        _pinned_pool_bytes[0] += nbytes
    else:
        _lib.celeste_host_free(ptr)


def pinned_empty(shape, dtype=np.float64) -> np.ndarray:
    """np.empty in page-locked memory (celeste_host_alloc): the library DMAs results straight into such arrays.
    Blocks are recycled through a size-keyed pool, because page-locking is slow (the array may be kept as long as
    needed: its block goes back to the pool when the array is garbage collected)."""
    lib = load_library()
    shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list)) else (shape,)))
    dt = np.dtype(dtype)
    nbytes = max(int(np.prod(shape)) * dt.itemsize, 1)
    nbytes = (nbytes + 4095) // 4096 * 4096
    free = _pinned_pool.get(nbytes)
    if free:
        ptr = free.pop()
        _pinned_pool_bytes[0] -= nbytes
    else:
        ptr = lib.celeste_host_alloc(nbytes)
        if not ptr:   # no device / out of lockable memory: plain pageable memory works too (staged by the library)
            return np.empty(shape, dtype=dt)
    block = _PinnedBlock(ptr, nbytes)
    buf = (C.c_char * nbytes).from_address(ptr)
    buf._celeste_block = block   # ties the block's lifetime to the ctypes buffer numpy keeps as its base
    return np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)


NF = 41   # free parameters of the optimiser (CELESTE_NF)


def tr_solve_batch(H, g, delta, solver=0, secular_iters=0, device=0):
    """celeste_tr_solve_batch: the optimiser's trust-region sub-problem on its own (Optim.jl's solve_tr_subproblem!).
    H [n, 41, 41] symmetric, g [n, 41], delta [n] -> (p [n, 41], model value [n], interior [n], fell_back [n])."""
    H = np.ascontiguousarray(H, dtype=np.float64).reshape(-1, NF, NF)
    n = H.shape[0]
    g = np.ascontiguousarray(g, dtype=np.float64).reshape(n, NF)
    delta = np.ascontiguousarray(np.broadcast_to(np.asarray(delta, dtype=np.float64), (n,)))
    p = np.zeros((n, NF)); m = np.zeros(n); interior = np.zeros(n, dtype=np.int32); fell = np.zeros(n, dtype=np.int32)
    dp = lambda a: a.ctypes.data_as(c_double_p)
    lib = load_library()
    st = lib.celeste_tr_solve_batch(int(device), n, dp(H), dp(g), dp(delta), int(solver), int(secular_iters), dp(p), dp(m),
                                    interior.ctypes.data_as(c_int32_p), fell.ctypes.data_as(c_int32_p))
    if st != 0:
        raise CelesteError(st, lib.celeste_strerror(st).decode())
    return p, m, interior, fell


def unpack_hessian(hp: np.ndarray) -> np.ndarray:
    """[..., 990] packed upper triangles (CELESTE_FLAG_PACKED_HESS) -> [..., 44, 44] symmetric matrices."""
    hp = np.asarray(hp)
    iu = np.triu_indices(P)            # row-major (i <= j) pairs ...
    order = np.argsort(iu[1] * (iu[1] + 1) // 2 + iu[0], kind="stable")   # ... -> position j (j + 1) / 2 + i
    out = np.zeros(hp.shape[:-1] + (P, P))
    out[..., iu[0][order], iu[1][order]] = hp
    out[..., iu[1][order], iu[0][order]] = hp
    return out


def prior_struct(prior: dict) -> PriorT:
    """dict with the layout of tests/golden/priors.json -> celeste_prior_t"""
    p = PriorT()
    for i in range(2):
        p.is_star[i] = prior["is_star"][i]
        p.flux_mean[i] = prior["flux_mean"][i]
        p.flux_var[i] = prior["flux_var"][i]
        for d in range(8):
            p.k[i][d] = prior["k"][i][d]
            for c in range(4):
                p.color_mean[i][d][c] = prior["color_mean"][i][d][c]
            for c in range(16):
                p.color_cov[i][d][c] = prior["color_cov"][i][d][c]
    p.gal_radius_px_mean = prior["gal_radius_px_mean"]
    p.gal_radius_px_var = prior["gal_radius_px_var"]
    return p


class Problem:
    """Marshals images / patches / neighbour lists into a celeste_problem_t.

    Keeps every numpy buffer alive for as long as the struct is in use.  Images are
    stored column-major (h fastest) as the reference stores them.
    """

    def __init__(self, images, patches, neighbors: Optional[Sequence[Sequence[int]]] = None, psf_K: int = 2,
                 prior: Optional[dict] = None, marshal_images: bool = True):
        """marshal_images=False: the planes already live on the device behind a shared image handle (`ImageSet`);
        the struct's `images` stays NULL and only patches / stamps / neighbours are marshalled."""
        from .model import PatchRow
        self.images = images
        self.patches = patches
        N, S = len(images), len(patches)
        self._keep: List[object] = []
        self.c_images = marshal_image_structs(images, self._keep) if marshal_images else None
        # shared stamp table: deduplicate by content of the raw stamp (object identity first: a constant PSF map hands
        # the same array to every patch of an image)
        stamps: List[np.ndarray] = []
        stamp_key, stamp_obj = {}, {}

        def stamp_index(raw):
            k = stamp_obj.get(id(raw))
            if k is None:
                st = np.ascontiguousarray(np.asfortranarray(raw, dtype=np.float64).T)  # column-major bytes
                key = st.tobytes()
                if key not in stamp_key:
                    stamp_key[key] = len(stamps)
                    stamps.append(st.reshape(-1))
                k = stamp_obj[id(raw)] = stamp_key[key]
                self._keep.append(raw)   # a reference, so that id(raw) stays unique while the table is built
            return k

        def fill(cp, p, n):
            cp.off_h, cp.off_w = int(p.bitmap_offset[0]), int(p.bitmap_offset[1])
            bm = np.asfortranarray(p.active_pixel_bitmap, dtype=np.uint8)
            cp.H2, cp.W2 = bm.shape[0], bm.shape[1]
            sub = images[n].pixels[cp.off_h:cp.off_h + cp.H2, cp.off_w:cp.off_w + cp.W2]
            if bm.size and not np.array_equal(bm.astype(bool), ~np.isnan(sub)):
                self._keep.append(bm)
                cp.bitmap = bm.ctypes.data_as(c_uint8_p)  # explicit bitmap only when it differs from !isnan
            J = np.asarray(p.wcs_jacobian, dtype=np.float64)
            cp.wcs_jacobian[0], cp.wcs_jacobian[1], cp.wcs_jacobian[2], cp.wcs_jacobian[3] = \
                J[0, 0], J[1, 0], J[0, 1], J[1, 1]
            cp.world_center[0], cp.world_center[1] = float(p.world_center[0]), float(p.world_center[1])
            cp.pixel_center[0], cp.pixel_center[1] = float(p.pixel_center[0]), float(p.pixel_center[1])
            psf = np.ascontiguousarray(p.psf, dtype=np.float64)
            assert psf.shape == (psf_K, 6)
            self._keep.append(psf)
            cp.psf = _dp(psf)
            cp.stamp = stamp_index(p.stamp)

        # rows of model.PatchRow (many-image problems) -> the sparse patch list of celeste_problem_t
        self.sparse = S > 0 and all(isinstance(row, PatchRow) for row in patches)
        if self.sparse:
            pairs = [(s, n, p) for s in range(S) for n, p in patches[s].nonempty()]
            self.c_patches = (PatchT * max(len(pairs), 1))()
            self.patch_source = np.array([s for s, _, _ in pairs], dtype=np.int32)
            self.patch_image = np.array([n for _, n, _ in pairs], dtype=np.int32)
            for k, (s, n, p) in enumerate(pairs):
                fill(self.c_patches[k], p, n)
            if not pairs:
                raise ValueError("no source overlaps any image")
        else:
            self.c_patches = (PatchT * (S * N))()
            for s in range(S):
                assert len(patches[s]) == N
                for n in range(N):
                    fill(self.c_patches[s * N + n], patches[s][n], n)
        self.stamps = np.ascontiguousarray(np.stack(stamps))
        if neighbors is None:
            neighbors = [[] for _ in range(S)]
        off = np.zeros(S + 1, dtype=np.int64)
        for s in range(S):
            off[s + 1] = off[s] + len(neighbors[s])
        idx = np.array([j for row in neighbors for j in row], dtype=np.int32)
        if idx.size == 0:
            idx = np.zeros(1, dtype=np.int32)
        self.nbr_off, self.nbr_idx = off, idx
        self.neighbors = [list(r) for r in neighbors]
        self.c_prior = prior_struct(prior) if prior is not None else None
        self.c = ProblemT()
        self.c.n_images, self.c.n_sources, self.c.psf_K, self.c.n_stamps = N, S, psf_K, len(stamps)
        if self.c_images is not None:
            self.c.images = self.c_images
        self.c.patches = self.c_patches
        self.c.stamps = _dp(self.stamps)
        self.c.nbr_offsets = off.ctypes.data_as(c_int64_p)
        self.c.nbr_index = idx.ctypes.data_as(c_int32_p)
        self.c.prior = C.pointer(self.c_prior) if self.c_prior is not None else None
        if self.sparse:
            self.c.n_patch_entries = len(self.patch_source)
            self.c.patch_source = self.patch_source.ctypes.data_as(c_int32_p)
            self.c.patch_image = self.patch_image.ctypes.data_as(c_int32_p)
        self.n_images, self.n_sources = N, S


PATCH_DTYPE = np.dtype([("off_h", "<i4"), ("off_w", "<i4"), ("H2", "<i4"), ("W2", "<i4"), ("bitmap", "<u8"),
                        ("wcs_jacobian", "<f8", 4), ("world_center", "<f8", 2), ("pixel_center", "<f8", 2), ("psf", "<u8"),
                        ("stamp", "<i4"), ("reserved", "<i4")])   # celeste_patch_t, field for field


def problem_from_table(images, table, neighbors: Optional[Sequence[Sequence[int]]] = None, psf_K: int = 2,
                       prior: Optional[dict] = None, marshal_images: bool = True) -> "Problem":
    """A celeste_problem_t straight from a model.PatchTable (the geometry of get_sky_patches as arrays): the same
    struct Problem(images, get_sky_patches(...), neighbor_map(...)) builds, without one Python object and two dozen
    ctypes assignments per (source, image) pair.  Bitmaps are never explicit here (a patch of get_sky_patches masks
    exactly the NaN pixels, which is the library's default).  Dense tables become the dense patch array, the others the
    sparse patch list."""
    assert PATCH_DTYPE.itemsize == C.sizeof(PatchT)
    pb = Problem.__new__(Problem)
    N, S = len(images), table.n_sources
    assert table.n_images == N
    pb.images, pb.patches = images, None
    pb._keep = []
    pb.c_images = marshal_image_structs(images, pb._keep) if marshal_images else None
    E = len(table.source)
    if E == 0:
        raise ValueError("no source overlaps any image")
    arr = np.zeros(E, dtype=PATCH_DTYPE)
    arr["off_h"] = table.box[:, 0] - 1; arr["off_w"] = table.box[:, 2] - 1
    arr["H2"] = table.H2; arr["W2"] = table.W2
    arr["world_center"] = table.world_center; arr["pixel_center"] = table.pixel_center
    # per-image constants; the raw stamp at the patch centre (one per image for a constant PSF map), deduplicated by
    # content like Problem does
    stamps, stamp_key = [], {}

    def stamp_index(raw):
        st = np.ascontiguousarray(np.asfortranarray(raw, dtype=np.float64).T)
        key = st.tobytes()
        if key not in stamp_key:
            stamp_key[key] = len(stamps)
            stamps.append(st.reshape(-1))
        return stamp_key[key]
    from .model import ConstantPSFMap
    for n, im in enumerate(images):
        e = np.flatnonzero(table.image == n)
        J = np.asarray(im.wcs_jacobian, dtype=np.float64)
        arr["wcs_jacobian"][e] = [J[0, 0], J[1, 0], J[0, 1], J[1, 1]]
        psf = np.ascontiguousarray(im.psf, dtype=np.float64)
        assert psf.shape == (psf_K, 6)
        pb._keep.append(psf)
        arr["psf"][e] = psf.ctypes.data
        if isinstance(im.psfmap, ConstantPSFMap):
            arr["stamp"][e] = stamp_index(im.psfmap.stamp)
        else:
            for k in e.tolist():
                arr["stamp"][k] = stamp_index(im.psfmap(table.pixel_center[k, 0], table.pixel_center[k, 1]))
    pb._keep.append(arr)
    pb.c_patches = (PatchT * E).from_buffer(arr)
    pb.sparse = not table.dense
    if pb.sparse:
        pb.patch_source = np.ascontiguousarray(table.source, dtype=np.int32)
        pb.patch_image = np.ascontiguousarray(table.image, dtype=np.int32)
    else:
        assert E == S * N
    pb.stamps = np.ascontiguousarray(np.stack(stamps))
    if neighbors is None:
        neighbors = [[] for _ in range(S)]
    off = np.zeros(S + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(r) for r in neighbors])
    idx = np.array([j for row in neighbors for j in row], dtype=np.int32)
    if idx.size == 0:
        idx = np.zeros(1, dtype=np.int32)
    pb.nbr_off, pb.nbr_idx = off, idx
    pb.neighbors = [list(r) for r in neighbors]
    pb.c_prior = prior_struct(prior) if prior is not None else None
    pb.c = ProblemT()
    pb.c.n_images, pb.c.n_sources, pb.c.psf_K, pb.c.n_stamps = N, S, psf_K, len(stamps)
    if pb.c_images is not None:
        pb.c.images = pb.c_images
    pb.c.patches = pb.c_patches
    pb.c.stamps = _dp(pb.stamps)
    pb.c.nbr_offsets = off.ctypes.data_as(c_int64_p)
    pb.c.nbr_index = idx.ctypes.data_as(c_int32_p)
    pb.c.prior = C.pointer(pb.c_prior) if pb.c_prior is not None else None
    if pb.sparse:
        pb.c.n_patch_entries = E
        pb.c.patch_source = pb.patch_source.ctypes.data_as(c_int32_p)
        pb.c.patch_image = pb.patch_image.ctypes.data_as(c_int32_p)
    pb.n_images, pb.n_sources = N, S
    return pb


def spline_prefilter(stamp: np.ndarray) -> np.ndarray:
    """ImagePatch ctor arithmetic (imaged_sources.jl:97-107) -> 53 x 53 B-spline coefficients."""
    lib = load_library()
    st = np.ascontiguousarray(np.asarray(stamp, dtype=np.float64).T).reshape(-1)  # column-major
    out = np.empty(COEF * COEF)
    check(lib.celeste_spline_prefilter(_dp(st), _dp(out)), lib)
    return out.reshape(COEF, COEF).T.copy()  # [h, w]
