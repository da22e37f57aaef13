"""One process, N devices: the ctypes face of celeste_group_* (include/celeste_mi355x.h).

  reference (src/ParallelRun.jl)                                here
  ------------------------------------------------------------  ----------------------------------------
  one_node_single_infer: N workers drain one source list        FieldGroup.maximize_batch
      inside one process (:546-607)
  one_node_joint_infer: Cyclades batches, the connected          FieldGroup.joint_infer
      components of a batch drained by N workers (:135-196, :302-397)
  process_source -> elbo() per target (:468-498)                FieldGroup.eval_batch / plan + sweep + results

All sharding, worker threads, streams and the RCCL catalog gather live inside libceleste_mi355x.so; this module only
marshals arguments -- the same calls shim/CelesteMI355X.jl makes.  (`parallel.py` is the torch.distributed variant:
one process per GPU.)
"""
import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from . import cabi
from .cabi import FLAG_GRAD, FLAG_HESS, FLAG_KL, P
from .elbo import ElboConfig


def cyclades_schedule(targets: Sequence[int], neighbors: List[List[int]], batch_size: int = 400,
                      rng: Optional[np.random.Generator] = None):
    """The batches of partition_cyclades_dynamic (partition.jl:173-236) in the layout celeste_group_joint_infer takes:
    (batch_offsets[n_batches + 1], comp_offsets[n_components + 1], comp_targets[n_entries]) -- source ids, components in
    the order infer.joint_layers walks them."""
    from .partition import partition_cyclades_dynamic
    targets = list(targets)
    tset = set(targets)
    nmap = {t: [n for n in neighbors[t] if n in tset] for t in targets}
    batches = partition_cyclades_dynamic(targets, nmap, batch_size=batch_size, rng=rng or np.random.default_rng(42))
    b_off, c_off, flat = [0], [0], []
    for comps in batches:
        for comp in comps:
            flat.extend(targets[i] for i in comp)
            c_off.append(len(flat))
        b_off.append(len(c_off) - 1)
    return (np.asarray(b_off, dtype=np.int64), np.asarray(c_off, dtype=np.int64), np.asarray(flat, dtype=np.int32))


def schedule_layers(batch_offsets, comp_offsets, comp_targets, n_sweeps: int):
    """The same schedule flattened into celeste_joint_infer's layers (layer j of a batch = the j-th sources of its
    components), together with, for every layer entry, its index into comp_targets -- the one-device reference of
    FieldGroup.joint_infer."""
    layers, entries = [], []
    for _ in range(n_sweeps):
        for b in range(len(batch_offsets) - 1):
            comps = [(int(comp_offsets[k]), int(comp_offsets[k + 1])) for k in range(int(batch_offsets[b]), int(batch_offsets[b + 1]))]
            depth = max((hi - lo for lo, hi in comps), default=0)
            for j in range(depth):
                idx = [lo + j for lo, hi in comps if hi - lo > j]
                layers.append([int(comp_targets[e]) for e in idx])
                entries.append(idx)
    return layers, entries


class FieldGroup:
    """celeste_group_t: the field replicated on `devices`, targets sharded by cost, results exchanged over RCCL."""

    def __init__(self, images, patches, neighbors=None, psf_K: int = 2, prior: Optional[dict] = None,
                 devices: Sequence[int] = (0,), problem: Optional["cabi.Problem"] = None):
        self.lib = cabi.load_library()
        self.problem = problem if problem is not None else cabi.Problem(images, patches, neighbors, psf_K=psf_K, prior=prior)
        self.S, self.N = self.problem.n_sources, self.problem.n_images
        dev = np.ascontiguousarray(np.asarray(list(devices), dtype=np.int32))
        h = C.c_void_p()
        cabi.check(self.lib.celeste_group_create(C.byref(self.problem.c), dev.size, dev.ctypes.data_as(cabi.c_int32_p),
                                                 C.byref(h)), self.lib)
        self.handle = h
        self.n_members = int(dev.size)
        self._n_planned = 0
        self._flags = 0

    @classmethod
    def from_catalog(cls, images, catalog, psf_K: int = 2, prior: Optional[dict] = None, devices: Sequence[int] = (0,),
                     sparse: Optional[bool] = None):
        from . import model
        if sparse is None:
            sparse = len(images) > 8
        table = model.patch_table(images, catalog, sparse=sparse)
        neighbors = table.neighbors()
        problem = cabi.problem_from_table(images, table, neighbors, psf_K=psf_K, prior=prior)
        g = cls(images, None, neighbors, psf_K=psf_K, prior=prior, devices=devices, problem=problem)
        g.table = table
        return g

    def close(self):
        if getattr(self, "handle", None):
            self.lib.celeste_group_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self) -> dict:
        gi = cabi.GroupInfoT()
        cabi.check(self.lib.celeste_group_info(self.handle, C.byref(gi)), self.lib)
        return {"n_members": gi.n_members, "n_devices": gi.n_devices,
                "exchange": {cabi.EXCHANGE_RCCL: "rccl", cabi.EXCHANGE_PEER_COPY: "peer_copy"}.get(gi.exchange, str(gi.exchange)),
                "rccl_ranks": gi.rccl_ranks, "devices": [int(gi.devices[k]) for k in range(gi.n_members)]}

    def collectives(self):
        """celeste_group_collectives: ([collectives enqueued per member], aborted)"""
        enq = np.zeros(self.n_members, dtype=np.int64)
        ab = C.c_int32(0)
        cabi.check(self.lib.celeste_group_collectives(self.handle, enq.ctypes.data_as(cabi.c_int64_p), C.byref(ab)), self.lib)
        return enq.tolist(), bool(ab.value)

    # -- elbo() over the members ------------------------------------------------------------------------------------
    def _outputs(self, n: int, flags: int, pinned: bool = True):
        v = np.zeros(n)
        d = np.zeros((n, P)) if flags & (FLAG_GRAD | FLAG_HESS) else None
        h = None
        if flags & FLAG_HESS:
            hshape = (n, cabi.HP) if flags & cabi.FLAG_PACKED_HESS else (n, P, P)
            # (page-locked, as FieldContext.eval_batch allocates it: the members' DMA lands Hessians in it directly)
            h = cabi.pinned_empty(hshape) if pinned and n >= 64 else np.zeros(hshape)
        return v, d, h, np.zeros((n, 2), dtype=np.int64), np.zeros(n, dtype=np.int32)

    def _finish(self, st, outs, raise_on_error):
        v, d, h, cnt, status = outs
        if st in (cabi.ERR_NONFINITE_INPUT, cabi.ERR_NONFINITE_RESULT):
            if raise_on_error:
                raise AssertionError(self.lib.celeste_strerror(st).decode())
            bad = status != 0
            if d is not None:
                d[bad] = np.nan
            if h is not None:
                h[bad] = np.nan
        else:
            cabi.check(st, self.lib)
        return v, d, h, cnt, status

    def eval_batch(self, vp, targets: Sequence[int], flags: int = FLAG_GRAD | FLAG_HESS | FLAG_KL, raise_on_error: bool = True):
        """celeste_group_elbo_eval_batch: (v[n], d[n,44], h[n,44,44], counters[n,2], status[n]) in the order of `targets`."""
        vp = np.ascontiguousarray(np.asarray(vp, dtype=np.float64).reshape(self.S, P))
        tg = np.ascontiguousarray(np.asarray(targets, dtype=np.int32))
        outs = self._outputs(tg.size, flags)
        v, d, h, cnt, status = outs
        dp = cabi.c_double_p
        st = self.lib.celeste_group_elbo_eval_batch(
            self.handle, vp.ctypes.data_as(dp), tg.size, tg.ctypes.data_as(cabi.c_int32_p), flags, v.ctypes.data_as(dp),
            d.ctypes.data_as(dp) if d is not None else None, h.ctypes.data_as(dp) if h is not None else None,
            cnt.ctypes.data_as(cabi.c_int64_p), status.ctypes.data_as(cabi.c_int32_p))
        return self._finish(st, outs, raise_on_error)

    def plan(self, vp, targets: Sequence[int], flags: int = FLAG_GRAD | FLAG_HESS | FLAG_KL):
        vp = np.ascontiguousarray(np.asarray(vp, dtype=np.float64).reshape(self.S, P))
        tg = np.ascontiguousarray(np.asarray(targets, dtype=np.int32))
        cabi.check(self.lib.celeste_group_sweep_plan(self.handle, vp.ctypes.data_as(cabi.c_double_p), tg.size,
                                                     tg.ctypes.data_as(cabi.c_int32_p), flags), self.lib)
        self._n_planned, self._flags = int(tg.size), flags

    def sweep(self):
        cabi.check(self.lib.celeste_group_sweep(self.handle), self.lib)

    def wait(self):
        cabi.check(self.lib.celeste_group_sweep_wait(self.handle), self.lib)

    def results(self, raise_on_error: bool = True, hessians: bool = True):
        outs = list(self._outputs(self._n_planned, self._flags))
        if not hessians:
            outs[2] = None       # (they stay on the members that own the targets)
        v, d, h, cnt, status = outs
        dp = cabi.c_double_p
        st = self.lib.celeste_group_sweep_results(self.handle, v.ctypes.data_as(dp), d.ctypes.data_as(dp) if d is not None else None,
                                                  h.ctypes.data_as(dp) if h is not None else None,
                                                  cnt.ctypes.data_as(cabi.c_int64_p), status.ctypes.data_as(cabi.c_int32_p))
        return self._finish(st, outs, raise_on_error)

    def shard_sizes(self):
        sizes = np.zeros(self.n_members, dtype=np.int32)
        costs = np.zeros(self.n_members, dtype=np.int64)
        cabi.check(self.lib.celeste_group_shard_sizes(self.handle, sizes.ctypes.data_as(cabi.c_int32_p),
                                                      costs.ctypes.data_as(cabi.c_int64_p)), self.lib)
        return sizes.tolist(), costs.tolist()

    def enable_timing(self, on: bool = True):
        cabi.check(self.lib.celeste_group_enable_timing(self.handle, 1 if on else 0), self.lib)

    def last_sweep_ms(self):
        a = (C.c_float * self.n_members)()
        b = (C.c_float * self.n_members)()
        cabi.check(self.lib.celeste_group_last_sweep_ms(self.handle, a, b), self.lib)
        return [float(x) for x in a], [float(x) for x in b]

    def last_kernel_ms(self, member: int = 0):
        ms = (C.c_float * 3)()
        cabi.check(self.lib.celeste_group_last_kernel_ms(self.handle, member, ms), self.lib)
        return [float(x) for x in ms]

    # -- maximize! / joint inference over the members ------------------------------------------------------------------
    def maximize_batch(self, vp, targets: Sequence[int], cfg: Optional[ElboConfig] = None, include_kl: bool = True,
                       vp_neighbors=None, pos_centers=None, raise_on_error: bool = True):
        """celeste_group_maximize_batch; returns (vp_new[S,44], iterations[n], f_evals[n], elbo[n], status[n])."""
        cfg = cfg or ElboConfig()
        vp = np.ascontiguousarray(np.asarray(vp, dtype=np.float64).reshape(self.S, P)).copy()
        tg = np.ascontiguousarray(np.asarray(targets, dtype=np.int32))
        n = tg.size
        its = np.zeros(n, dtype=np.int32); evals = np.zeros(n, dtype=np.int32)
        el = np.zeros(n); status = np.zeros(n, dtype=np.int32)
        ccfg = cfg.to_c(include_kl)
        nb = None if vp_neighbors is None else np.ascontiguousarray(np.asarray(vp_neighbors, dtype=np.float64).reshape(self.S, P))
        pc = None if pos_centers is None else np.ascontiguousarray(np.asarray(pos_centers, dtype=np.float64).reshape(n, 2))
        dp = cabi.c_double_p
        st = self.lib.celeste_group_maximize_batch(self.handle, vp.ctypes.data_as(dp), nb.ctypes.data_as(dp) if nb is not None else None,
                                                   pc.ctypes.data_as(dp) if pc is not None else None, n,
                                                   tg.ctypes.data_as(cabi.c_int32_p), C.byref(ccfg), its.ctypes.data_as(cabi.c_int32_p),
                                                   evals.ctypes.data_as(cabi.c_int32_p), el.ctypes.data_as(dp),
                                                   status.ctypes.data_as(cabi.c_int32_p))
        if st in (cabi.ERR_NONFINITE_INPUT, cabi.ERR_NONFINITE_RESULT):
            if raise_on_error:
                raise AssertionError(self.lib.celeste_strerror(st).decode())
        else:
            cabi.check(st, self.lib)
        return vp, its, evals, el, status

    def joint_infer(self, vp, batch_offsets, comp_offsets, comp_targets, n_sweeps: int = 3, cfg: Optional[ElboConfig] = None,
                    include_kl: bool = True, pos_centers=None):
        """celeste_group_joint_infer on a Cyclades schedule (cyclades_schedule).  pos_centers: [n_entries, 2] or None.
        Returns (vp_new, iterations, f_evals, elbo, status, n_exchanges); per-entry outputs are [n_sweeps, n_entries]."""
        cfg = cfg or ElboConfig()
        vp = np.ascontiguousarray(np.asarray(vp, dtype=np.float64).reshape(self.S, P)).copy()
        b_off = np.ascontiguousarray(batch_offsets, dtype=np.int64)
        c_off = np.ascontiguousarray(comp_offsets, dtype=np.int64)
        tg = np.ascontiguousarray(comp_targets, dtype=np.int32)
        E = int(tg.size)
        pc = None if pos_centers is None else np.ascontiguousarray(np.asarray(pos_centers, dtype=np.float64).reshape(E, 2))
        its = np.zeros((n_sweeps, E), dtype=np.int32); evals = np.zeros((n_sweeps, E), dtype=np.int32)
        el = np.zeros((n_sweeps, E)); status = np.zeros((n_sweeps, E), dtype=np.int32)
        nx = C.c_int64(0)
        ccfg = cfg.to_c(include_kl)
        dp = cabi.c_double_p
        st = self.lib.celeste_group_joint_infer(self.handle, vp.ctypes.data_as(dp), n_sweeps, b_off.size - 1,
                                                b_off.ctypes.data_as(cabi.c_int64_p), c_off.ctypes.data_as(cabi.c_int64_p),
                                                tg.ctypes.data_as(cabi.c_int32_p), pc.ctypes.data_as(dp) if pc is not None else None,
                                                C.byref(ccfg), its.ctypes.data_as(cabi.c_int32_p), evals.ctypes.data_as(cabi.c_int32_p),
                                                el.ctypes.data_as(dp), status.ctypes.data_as(cabi.c_int32_p), C.byref(nx))
        if st not in (cabi.ERR_NONFINITE_INPUT, cabi.ERR_NONFINITE_RESULT):
            cabi.check(st, self.lib)
        return vp, its, evals, el, status, int(nx.value)
