"""Host-side mirror of the reference's ELBO interface, backed by the HIP engine.

  reference (src/deterministic_vi/)                      here
  ------------------------------------------------------  --------------------------------
  ElboArgs(images, patches, active_sources; psf_K,        ElboArgs(...)            elbo_args.jl:165-211
           include_kl)
  SensitiveFloat (v, d[P x S], h[PS x PS])                SensitiveFloat           SensitiveFloats.jl:23-47
  elbo(ea, vp) / elbo_likelihood(ea, vp)                  elbo / elbo_likelihood   elbo_objective.jl:400-492
  (per-box sweep of process_source)                       FieldContext.eval_batch  ParallelRun.jl:468-498

Same names, argument meaning and error behaviour: non-finite parameters or results raise
AssertionError like the reference's @assert (elbo_objective.jl:487, elbo_args.jl:145-149).
Sa = 1 is the production configuration (batched, tuned); Sa > 1 goes through celeste_elbo_eval_multi.
"""
import time
import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import cabi
from .cabi import FLAG_GRAD, FLAG_HESS, FLAG_KL, P


@dataclass
class SensitiveFloat:
    """Value, gradient (P x 1) and Hessian (P x P) of the ELBO for one active source."""
    v: float
    d: Optional[np.ndarray]
    h: Optional[np.ndarray]
    active_pixel_counter: int = 0
    inactive_pixel_counter: int = 0

    @property
    def has_gradient(self):
        return self.d is not None

    @property
    def has_hessian(self):
        return self.h is not None


class FieldContext:
    """Device-resident problem: images, all patches, neighbour graph (celeste_ctx_t)."""

    def __init__(self, images, patches, neighbors=None, psf_K: int = 2, prior: Optional[dict] = None,
                 device: int = 0, image_set: Optional["cabi.ImageSet"] = None, problem: Optional["cabi.Problem"] = None):
        """image_set: a shared image handle (cabi.ImageSet over the same `images`); the context then costs a patch-table
        upload instead of a copy of every plane -- the per-source ElboArgs of ParallelRun.jl:468-488.
        problem: an already marshalled celeste_problem_t (from_catalog)."""
        self.lib = cabi.load_library()
        self.problem = problem if problem is not None else cabi.Problem(images, patches, neighbors, psf_K=psf_K, prior=prior,
                                                                        marshal_images=image_set is None)
        self.S, self.N = self.problem.n_sources, self.problem.n_images
        h = C.c_void_p()
        t0 = time.perf_counter()
        if image_set is None:
            self.device = device
            cabi.check(self.lib.celeste_ctx_create(C.byref(self.problem.c), device, C.byref(h)), self.lib)
        else:
            assert len(image_set.images) == self.N
            self.device = image_set.device
            cabi.check(self.lib.celeste_ctx_create_on(image_set.handle, C.byref(self.problem.c), C.byref(h)), self.lib)
        self.create_ms = (time.perf_counter() - t0) * 1e3   # the C call alone: uploads, stamp conditioning + prefilter, tables
        self.handle = h

    @classmethod
    def from_catalog(cls, images, catalog, psf_K: int = 2, prior: Optional[dict] = None, device: int = 0,
                     image_set: Optional["cabi.ImageSet"] = None, sparse: Optional[bool] = None):
        """The context of ParallelRun's box inference for a catalog: get_sky_patches + find_neighbors
        (imaged_sources.jl:165-182, 232-244) through model.patch_table -- the same patches and neighbour lists as
        FieldContext(images, get_sky_patches(images, catalog), neighbor_map(patches)), an order of magnitude less host
        time (no per-patch objects).  sparse (default: when there are more than 8 images) selects the sparse patch
        list.  The table stays available as `ctx.table` (costs(), neighbors)."""
        from . import model
        if sparse is None:
            sparse = len(images) > 8
        table = model.patch_table(images, catalog, sparse=sparse)
        neighbors = table.neighbors()
        problem = cabi.problem_from_table(images, table, neighbors, psf_K=psf_K, prior=prior,
                                          marshal_images=image_set is None)
        ctx = cls(images, None, neighbors, psf_K=psf_K, prior=prior, device=device, image_set=image_set, problem=problem)
        ctx.table = table
        return ctx

    def close(self):
        if getattr(self, "handle", None):
            self.lib.celeste_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- host-pointer API ---------------------------------------------------------------
    def eval_batch(self, vp, targets: Sequence[int], flags: int = FLAG_GRAD | FLAG_HESS | FLAG_KL,
                   raise_on_error: bool = True, pinned: bool = True):
        """vp: S x 44 array (row s = source s).  Returns (v[n], d[n,44], h[n,44,44], counters[n,2], status[n]);
        with FLAG_PACKED_HESS h is [n, 990] (cabi.unpack_hessian).  pinned: large outputs are allocated in page-locked
        memory (cabi.pinned_empty), which the library fills by DMA while later parts of the batch still compute."""
        vp = np.ascontiguousarray(np.asarray(vp, dtype=np.float64).reshape(self.S, P))
        tg = np.ascontiguousarray(np.asarray(targets, dtype=np.int32))
        n = tg.size
        big = pinned and n >= 64
        v = np.zeros(n)
        d = np.zeros((n, P)) if flags & (FLAG_GRAD | FLAG_HESS) else None
        h = None
        if flags & FLAG_HESS:
            hshape = (n, cabi.HP) if flags & cabi.FLAG_PACKED_HESS else (n, P, P)
            h = cabi.pinned_empty(hshape) if big else np.zeros(hshape)
        cnt = np.zeros((n, 2), dtype=np.int64)
        status = np.zeros(n, dtype=np.int32)
        dp = cabi.c_double_p
        st = self.lib.celeste_elbo_eval_batch(
            self.handle, vp.ctypes.data_as(dp), n, tg.ctypes.data_as(cabi.c_int32_p), flags,
            v.ctypes.data_as(dp), d.ctypes.data_as(dp) if d is not None else None,
            h.ctypes.data_as(dp) if h is not None else None, cnt.ctypes.data_as(cabi.c_int64_p),
            status.ctypes.data_as(cabi.c_int32_p))
        if st in (cabi.ERR_NONFINITE_INPUT, cabi.ERR_NONFINITE_RESULT):
            if raise_on_error:
                raise AssertionError(self.lib.celeste_strerror(st).decode())
            # targets that failed carry no result (the pooled page-locked blocks are not cleared between calls)
            bad = status != 0
            if d is not None:
                d[bad] = np.nan
            if h is not None:
                h[bad] = np.nan
        else:
            cabi.check(st, self.lib)
        # h is symmetric, so column-major == row-major
        return v, d, h, cnt, status

    def eval_multi(self, vp, active: Sequence[int], flags: int = FLAG_GRAD | FLAG_HESS | FLAG_KL):
        """elbo() with several active sources: returns (v, d[44, Sa], h[44 Sa, 44 Sa], counters[2]); d / h are None
        when not requested.  Column a of d and block a of h belong to active[a] (SensitiveFloats.jl:29-31)."""
        vp = np.ascontiguousarray(np.asarray(vp, dtype=np.float64).reshape(self.S, P))
        act = np.ascontiguousarray(np.asarray(active, dtype=np.int32))
        sa = act.size
        want_h = bool(flags & FLAG_HESS)
        want_d = want_h or bool(flags & FLAG_GRAD)
        v = C.c_double()
        d = np.zeros(sa * P) if want_d else None
        h = np.zeros((sa * P, sa * P)) if want_h else None
        na = C.c_int64(); ni = C.c_int64()
        dp = cabi.c_double_p
        st = self.lib.celeste_elbo_eval_multi(self.handle, vp.ctypes.data_as(dp), sa, act.ctypes.data_as(cabi.c_int32_p),
                                              flags, C.byref(v), d.ctypes.data_as(dp) if want_d else None,
                                              h.ctypes.data_as(dp) if want_h else None, C.byref(na), C.byref(ni))
        if st in (cabi.ERR_NONFINITE_INPUT, cabi.ERR_NONFINITE_RESULT):
            raise AssertionError(self.lib.celeste_strerror(st).decode())
        cabi.check(st, self.lib)
        return (v.value, d.reshape(sa, P).T.copy() if want_d else None, h, np.array([na.value, ni.value]))

    # -- device-pointer API (torch tensors or raw pointers) -----------------------------------
    def eval_batch_device(self, d_vp: int, n_targets: int, d_targets: int, flags: int, d_v: int, d_d: int,
                          d_h: int, d_counters: int, d_status: int, stream: int = 0):
        cabi.check(self.lib.celeste_elbo_eval_batch_device(self.handle, d_vp, n_targets, d_targets, flags, d_v, d_d,
                                                           d_h, d_counters, d_status, stream), self.lib)

    def maximize_batch(self, vp, targets: Sequence[int], cfg: Optional["ElboConfig"] = None, include_kl: bool = True,
                       vp_neighbors=None, pos_centers=None, raise_on_error: bool = True):
        """maximize! for every target (ElboMaximize.jl:228-242), neighbours frozen at `vp_neighbors` (default: the
        input vp); `pos_centers` [n,2] pins the position boxes (default: the current positions).
        Returns (vp_new[S,44], iterations[n], f_evals[n], elbo[n], status[n]); vp is not modified.
        A target whose ELBO turns non-finite keeps its input row and gets a non-zero status; the other targets are
        optimised normally.  raise_on_error=True turns such a status into the reference's AssertionError (as a direct
        maximize! call would raise); the node-level loops pass False and skip the source like the reference's
        try/catch does (ParallelRun.jl:389-396, 582-597)."""
        cfg = cfg or ElboConfig()
        vp = np.ascontiguousarray(np.asarray(vp, dtype=np.float64).reshape(self.S, P)).copy()
        tg = np.ascontiguousarray(np.asarray(targets, dtype=np.int32))
        n = tg.size
        its = np.zeros(n, dtype=np.int32); evals = np.zeros(n, dtype=np.int32)
        el = np.zeros(n); status = np.zeros(n, dtype=np.int32)
        ccfg = cfg.to_c(include_kl)
        nb = None if vp_neighbors is None else np.ascontiguousarray(
            np.asarray(vp_neighbors, dtype=np.float64).reshape(self.S, P))
        pc = None if pos_centers is None else np.ascontiguousarray(np.asarray(pos_centers, dtype=np.float64).reshape(n, 2))
        st = self.lib.celeste_maximize_batch(self.handle, vp.ctypes.data_as(cabi.c_double_p),
                                             nb.ctypes.data_as(cabi.c_double_p) if nb is not None else None,
                                             pc.ctypes.data_as(cabi.c_double_p) if pc is not None else None, n,
                                             tg.ctypes.data_as(cabi.c_int32_p), C.byref(ccfg),
                                             its.ctypes.data_as(cabi.c_int32_p), evals.ctypes.data_as(cabi.c_int32_p),
                                             el.ctypes.data_as(cabi.c_double_p), status.ctypes.data_as(cabi.c_int32_p))
        if st in (cabi.ERR_NONFINITE_INPUT, cabi.ERR_NONFINITE_RESULT):
            if raise_on_error:
                raise AssertionError(self.lib.celeste_strerror(st).decode())
        else:
            cabi.check(st, self.lib)
        return vp, its, evals, el, status

    def maximize_batch_device(self, d_vp: int, n_targets: int, d_targets: int, cfg: Optional["ElboConfig"] = None,
                              include_kl: bool = True, d_vp_neighbors: int = 0, d_pos_centers: int = 0, d_iterations: int = 0,
                              d_f_evals: int = 0, d_elbo: int = 0, d_status: int = 0, stream: int = 0):
        """celeste_maximize_batch_device: maximize! of the targets against the parameter table at device pointer `d_vp`
        (n_sources x 44 doubles, optimised in place for the targets); every other argument a device pointer (0 = NULL) as
        in include/celeste_mi355x.h.  Asynchronous on `stream` for batches of up to 1024 targets."""
        ccfg = (cfg or ElboConfig()).to_c(include_kl)
        cabi.check(self.lib.celeste_maximize_batch_device(self.handle, d_vp, d_vp_neighbors or None, d_pos_centers or None,
                                                          n_targets, d_targets, C.byref(ccfg), d_iterations or None,
                                                          d_f_evals or None, d_elbo or None, d_status or None, stream or None),
                   self.lib)

    def joint_infer(self, vp, layers: Sequence[Sequence[int]], cfg: Optional["ElboConfig"] = None, include_kl: bool = True,
                    pos_centers=None):
        """celeste_joint_infer: the layers (lists of mutually non-neighbouring sources) are optimised one after another
        against ONE device-resident parameter table.  pos_centers: one [len(layer), 2] array per layer (or None).
        Returns (vp_new[S,44], iterations, f_evals, elbo, status) with the per-entry outputs concatenated over the layers
        in order; vp is not modified.  Sources that fail keep the row they had before their layer (status != 0)."""
        cfg = cfg or ElboConfig()
        vp = np.ascontiguousarray(np.asarray(vp, dtype=np.float64).reshape(self.S, P)).copy()
        off = np.zeros(len(layers) + 1, dtype=np.int64)
        for l, layer in enumerate(layers):
            off[l + 1] = off[l] + len(layer)
        total = int(off[-1])
        tg = np.ascontiguousarray(np.concatenate([np.asarray(l, dtype=np.int32) for l in layers]) if total else np.zeros(0, np.int32))
        pc = None
        if pos_centers is not None:
            pc = np.ascontiguousarray(np.concatenate([np.asarray(c, dtype=np.float64).reshape(-1, 2) for c in pos_centers]))
            assert pc.shape == (total, 2)
        its = np.zeros(total, dtype=np.int32); evals = np.zeros(total, dtype=np.int32)
        el = np.zeros(total); status = np.zeros(total, dtype=np.int32)
        ccfg = cfg.to_c(include_kl)
        st = self.lib.celeste_joint_infer(self.handle, vp.ctypes.data_as(cabi.c_double_p), len(layers),
                                          off.ctypes.data_as(cabi.c_int64_p), tg.ctypes.data_as(cabi.c_int32_p),
                                          pc.ctypes.data_as(cabi.c_double_p) if pc is not None else None, C.byref(ccfg),
                                          its.ctypes.data_as(cabi.c_int32_p), evals.ctypes.data_as(cabi.c_int32_p),
                                          el.ctypes.data_as(cabi.c_double_p), status.ctypes.data_as(cabi.c_int32_p))
        if st not in (cabi.ERR_NONFINITE_INPUT, cabi.ERR_NONFINITE_RESULT):
            cabi.check(st, self.lib)
        return vp, its, evals, el, status

    def render_expected(self, vp, image: int) -> np.ndarray:
        """sum_s E[G_s] in nanomaggies on image `image` (bin/write_celeste_expectation.jl:112-156); H x W array."""
        vp = np.ascontiguousarray(np.asarray(vp, dtype=np.float64).reshape(self.S, P))
        img = self.problem.images[image]
        H, W = img.pixels.shape
        out = np.zeros(H * W)
        cabi.check(self.lib.celeste_render_expected(self.handle, vp.ctypes.data_as(cabi.c_double_p), image,
                                                    out.ctypes.data_as(cabi.c_double_p)), self.lib)
        return out.reshape(W, H).T.copy()

    def enable_timing(self, on: bool = True):
        cabi.check(self.lib.celeste_ctx_enable_timing(self.handle, 1 if on else 0), self.lib)

    def spline_coefficients(self, stamp: int) -> np.ndarray:
        """The 53 x 53 B-spline coefficients [h, w] the context holds for stamp `stamp` (conditioned + prefiltered on the
        device at creation: spline_prefilter_kernel), as cabi.spline_prefilter returns them for one stamp on the host."""
        out = np.empty(cabi.COEF * cabi.COEF)
        cabi.check(self.lib.celeste_ctx_spline_coefficients(self.handle, int(stamp), out.ctypes.data_as(cabi.c_double_p)), self.lib)
        return out.reshape(cabi.COEF, cabi.COEF).T.copy()

    def last_kernel_ms(self):
        ms = (C.c_float * 3)()
        cabi.check(self.lib.celeste_ctx_last_kernel_ms(self.handle, ms), self.lib)
        return [float(x) for x in ms]

    def last_record_sum_ms(self) -> float:
        """FLAG_SPLIT launches: duration of the streaming per-patch record sum."""
        ms = C.c_float()
        cabi.check(self.lib.celeste_ctx_last_record_sum_ms(self.handle, C.byref(ms)), self.lib)
        return float(ms.value)

    def work_stats(self, targets: Sequence[int]) -> dict:
        tg = np.ascontiguousarray(np.asarray(targets, dtype=np.int32))
        ws = cabi.WorkStatsT()
        cabi.check(self.lib.celeste_ctx_work_stats(self.handle, tg.size, tg.ctypes.data_as(cabi.c_int32_p),
                                                   C.byref(ws)), self.lib)
        return {k: int(getattr(ws, k)) for k, _ in cabi.WorkStatsT._fields_}


@dataclass
class ElboConfig:
    """ElboMaximize.ElboConfig (ElboMaximize.jl:27-61) -- the knobs maximize! exposes."""
    loc_width: float = 1e-4
    loc_scale: float = 1.0
    max_iters: int = 50
    xtol_abs: float = 1e-7
    ftol_rel: float = 1e-6
    gtol: float = 1e-8
    initial_delta: float = 1.0
    delta_hat: float = 1e9
    tr_secular_iters: int = 0   # 0: multiplier iterations run to convergence; 5: Optim.jl's cap (celeste_optim_config_t)

    def to_c(self, include_kl: bool) -> "cabi.OptimConfigT":
        return cabi.OptimConfigT(self.loc_width, self.loc_scale, self.max_iters, int(include_kl), self.xtol_abs,
                                 self.ftol_rel, self.gtol, self.initial_delta, self.delta_hat, self.tr_secular_iters, 0)


class ElboArgs:
    """elbo_args.jl:165-211.  `patches` is S x N (list of rows); `active_sources` 0-based.

    The local source list is exactly the reference's: source 0..S-1 all contribute to any
    pixel they cover; only the active source carries derivatives.
    """

    def __init__(self, images, patches, active_sources: Sequence[int], psf_K: int = 2, include_kl: bool = True,
                 prior: Optional[dict] = None, device: int = 0):
        self.S = len(patches)
        self.Sa = len(active_sources)
        self.N = len(images)
        assert all(len(row) == self.N for row in patches)
        assert psf_K > 0
        assert self.Sa >= 1 and len(set(active_sources)) == self.Sa
        self.psf_K = psf_K
        self.images = images
        self.patches = patches
        self.active_sources = list(active_sources)
        self.include_kl = include_kl
        nbrs = [[] for _ in range(self.S)]
        for a in self.active_sources:   # every other local source is a neighbour of every active source
            nbrs[a] = [s for s in range(self.S) if s != a]
        self._ctx = FieldContext(images, patches, nbrs, psf_K=psf_K, prior=prior, device=device)


def _eval(ea: ElboArgs, vp, flags: int) -> SensitiveFloat:
    vp = np.asarray(vp, dtype=np.float64).reshape(ea.S, P)
    assert np.all(np.isfinite(vp)), "vp contains NaNs or Infs"
    if ea.Sa > 1:   # d is P x Sa, h is (P Sa) x (P Sa), as in SensitiveFloats.jl:29-31
        v, d, h, cnt = ea._ctx.eval_multi(vp, ea.active_sources, flags)
        return SensitiveFloat(float(v), d, h, int(cnt[0]), int(cnt[1]))
    v, d, h, cnt, _ = ea._ctx.eval_batch(vp, ea.active_sources, flags)
    return SensitiveFloat(float(v[0]), None if d is None else d[0].copy(), None if h is None else h[0].copy(),
                          int(cnt[0, 0]), int(cnt[0, 1]))


def elbo_likelihood(ea: ElboArgs, vp, calculate_gradient: bool = True, calculate_hessian: bool = True):
    """elbo_objective.jl:400-474"""
    flags = (FLAG_GRAD if calculate_gradient else 0) | (FLAG_HESS if calculate_gradient and calculate_hessian else 0)
    return _eval(ea, vp, flags)


def elbo(ea: ElboArgs, vp, calculate_gradient: bool = True, calculate_hessian: bool = True):
    """elbo_objective.jl:482-492"""
    flags = (FLAG_GRAD if calculate_gradient else 0) | (FLAG_HESS if calculate_gradient and calculate_hessian else 0)
    if ea.include_kl:
        flags |= FLAG_KL
    return _eval(ea, vp, flags)


def maximize(ea: ElboArgs, vp, cfg: Optional[ElboConfig] = None):
    """ElboMaximize.maximize!(ea, vp, cfg) (ElboMaximize.jl:228-242): optimises the active source in place.
    Returns (f_evals, max_value, vp) like the reference returns (f_calls, min_value, ...)."""
    vp_arr = np.asarray(vp, dtype=np.float64).reshape(ea.S, P)
    if ea.Sa != 1:
        raise NotImplementedError("maximize! on the device optimises one active source (ParallelRun.jl:482)")
    new, its, evals, el, _ = ea._ctx.maximize_batch(vp_arr, ea.active_sources, cfg, include_kl=ea.include_kl)
    a = ea.active_sources[0]
    vp_arr[a] = new[a]
    return int(evals[0]), float(el[0]), vp_arr
