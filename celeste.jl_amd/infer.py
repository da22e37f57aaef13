"""Node-level inference loops on top of the device optimiser.

  reference (src/ParallelRun.jl)                              here
  ----------------------------------------------------------  -----------------------------------------
  one_node_single_infer  (:546-607) -> process_source (:468)  one_node_single_infer
  one_node_joint_infer   (:135-196): setup_vecs, Cyclades     one_node_joint_infer
  infer_box / _infer_box (:610-672), OptimizedSource, bad_sky  infer_box, OptimizedSource, bad_sky
      batches, num_joint_vi_iters sweeps, process_sources_kernel!
      (:372-397) = sequential maximize! inside a connected component

Single inference: every target starts from generic_init_source and sees its neighbours frozen at
catalog_init_source (DeterministicVI.init_sources, DeterministicVI.jl:94-103).
Joint inference: all targets share one parameter table (setup_vecs: generic_init_source for targets,
catalog_init_source for the rest); inside a Cyclades batch the connected components are independent, and the
sources of one component are optimised one after another.  On the GPU the j-th sources of all components of a
batch form one launch ("layer"): no two of them are neighbours, so optimising them simultaneously is exactly
the reference's schedule.
"""
import logging
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence

import numpy as np

from .elbo import ElboConfig, FieldContext
from .params import catalog_init_source, generic_init_source, init_source_table  # noqa: F401
from .parallel import sharded_maximize
from .partition import color_classes, partition_cyclades_dynamic

NUM_JOINT_VI_ITERS = 3   # Config.num_joint_vi_iters (src/config.jl:17-25)
log = logging.getLogger("celeste_jl_amd.infer")


def default_infer_config() -> ElboConfig:
    """The optimiser settings of the node-level loops when the caller gives none: ElboConfig's defaults with the
    trust-region multiplier iteration capped at 5 steps, as Optim.jl's solve_tr_subproblem! does (converged or not) --
    the reference's production behaviour (ElboMaximize.jl:228-242 -> Optim.optimize).  ElboConfig() itself keeps
    tr_secular_iters = 0 (run to convergence, <= 20 steps), the setting the optimiser tests pin against a 60-digit
    solution of the sub-problem; the two differ only on boundary steps whose multiplier needs more than 5 Newton steps."""
    return ElboConfig(tr_secular_iters=5)


def _report_failures(targets, status, where: str, failed: Optional[set]):
    """A source whose ELBO turned non-finite is logged and skipped -- the reference's production / multi-thread
    behaviour (Log.exception in process_sources_kernel! and one_node_single_infer, ParallelRun.jl:389-396, 582-597):
    its row stays where it was, every other source of the batch keeps its result."""
    status = np.asarray(status)
    for k in np.flatnonzero(status):
        t = targets[int(k)]
        log.warning("%s: source %d skipped (status %d)", where, int(t), int(status[k]))
        if failed is not None:
            failed.add(int(t))


def one_node_single_infer(ctx: FieldContext, catalog, target_sources: Sequence[int],
                          cfg: Optional[ElboConfig] = None, failed: Optional[set] = None) -> np.ndarray:
    """Returns the optimised parameters, one row per target (OptimizedSource.vs).  Targets that failed are logged,
    keep their initial row and are added to `failed` (the reference drops them from its result list)."""
    vp_nbr = init_source_table(catalog)
    vp = init_source_table(catalog, target_sources)
    cfg = cfg or default_infer_config()
    new, _, _, _, st = ctx.maximize_batch(vp, list(target_sources), cfg, vp_neighbors=vp_nbr, raise_on_error=False)
    _report_failures(target_sources, st, "one_node_single_infer", failed)
    return new[list(target_sources)]


def joint_infer_sweeps(maximize_layer: Callable, vp: np.ndarray, targets: Sequence[int], neighbors: List[List[int]],
                       batch_size: int = 400, n_iters: int = NUM_JOINT_VI_ITERS,
                       rng: Optional[np.random.Generator] = None, rank: int = 0, world: int = 1,
                       costs: Optional[Sequence[float]] = None, all_gather: Optional[Callable] = None,
                       schedule: str = "cyclades", device: Optional[int] = None) -> np.ndarray:
    """The joint-inference schedule, independent of who optimises a layer.

    schedule = "cyclades": the reference's randomly drawn batches, connected components processed source by source
    (ParallelRun.jl:302-397).  schedule = "coloring": the colour classes of a greedy colouring of the neighbour graph,
    one launch per colour and sweep -- the same conflict-freedom with far fewer, larger launches (2.3x faster on
    the 2000-source bench field).  The sweep order differs, as it does between two Cyclades seeds; with most sources in
    the first colour class a sweep is closer to a Jacobi step, so crowded scenes may want more sweeps.

    maximize_layer(vp, layer, pos_centers) -> [len(layer), 44] optimises the sources of `layer` (no two of them are
    neighbours) against the shared table `vp` and returns their new rows.  With world > 1 every rank holds the
    whole table, optimises its cost-balanced shard of each layer and the updated rows are all-gathered (352 B per
    target and layer: SURVEY.md 8(e)) -- the only exchange of the path.  vp is updated in place and returned."""
    targets = list(targets)
    centers = {t: vp[t, 0:2].copy() for t in targets}        # boxes stay at the initial positions
    for layer in joint_layers(targets, neighbors, batch_size, n_iters, rng, schedule):

        def run(local):
            pc = np.stack([centers[t] for t in local])
            return maximize_layer(vp, list(local), pc)
        if world == 1:
            vp[layer] = run(layer)
        else:
            lc = [1.0 if costs is None else costs[t] for t in layer]
            vp[layer] = sharded_maximize(run, layer, lc, rank, world, all_gather, device)
    return vp


def joint_layers(targets: Sequence[int], neighbors: List[List[int]], batch_size: int = 400,
                 n_iters: int = NUM_JOINT_VI_ITERS, rng: Optional[np.random.Generator] = None,
                 schedule: str = "cyclades") -> List[List[int]]:
    """The joint-inference schedule as a flat list of layers (lists of source ids no two of which are neighbours), in the
    order they are optimised: for every sweep, for every batch, the j-th sources of all its connected components
    (j = 0, 1, ...).  This is what celeste_joint_infer consumes."""
    targets = list(targets)
    tset = set(targets)
    nmap = {t: [n for n in neighbors[t] if n in tset] for t in targets}
    if schedule == "coloring":
        batches = [[[i] for i in cls] for cls in color_classes(targets, nmap)]   # one layer per colour
    else:
        assert schedule == "cyclades"
        batches = partition_cyclades_dynamic(targets, nmap, batch_size=batch_size,
                                             rng=rng or np.random.default_rng(42))   # srand(42), ParallelRun.jl:143
    layers = []
    for _ in range(n_iters):
        for components in batches:
            depth = max(len(c) for c in components)
            for j in range(depth):
                layers.append([targets[c[j]] for c in components if len(c) > j])
    return layers


def one_node_joint_infer(ctx: FieldContext, catalog, target_sources: Sequence[int], neighbors: List[List[int]],
                         cfg: Optional[ElboConfig] = None, batch_size: int = 400, n_iters: int = NUM_JOINT_VI_ITERS,
                         rng: Optional[np.random.Generator] = None, rank: int = 0, world: int = 1,
                         costs: Optional[Sequence[float]] = None, all_gather: Optional[Callable] = None,
                         schedule: str = "cyclades", failed: Optional[set] = None) -> np.ndarray:
    """Cyclades-batched joint inference; returns the optimised parameters, one row per target.  rank / world > 1:
    one process per GPU, images replicated (every rank builds the same FieldContext), layers sharded.
    A source that fails in some layer keeps the row it had before that layer (logged, added to `failed`)."""
    targets = list(target_sources)
    vp = init_source_table(catalog, targets)
    cfg = cfg or default_infer_config()
    if world == 1:
        # the whole schedule in one call: the parameter table stays in HBM across all layers (celeste_joint_infer)
        layers = joint_layers(targets, neighbors, batch_size, n_iters, rng, schedule)
        centers = [vp[layer, 0:2].copy() for layer in layers]      # boxes stay at the initial positions
        vp, _, _, _, st = ctx.joint_infer(vp, layers, cfg, pos_centers=centers)
        _report_failures([t for layer in layers for t in layer], st, "one_node_joint_infer", failed)
        return vp[targets]
    if all_gather is None:
        # one process per GPU: every rank optimises its shard of each layer against its own device-resident table and
        # the optimised rows (+ status) are all-gathered on the device after every layer (RCCL; gloo: through the host)
        from .parallel import DeviceJointInfer
        dj = DeviceJointInfer(ctx, vp, rank, world, cfg)
        centers = {t: vp[t, 0:2].copy() for t in targets}
        for layer in joint_layers(targets, neighbors, batch_size, n_iters, rng, schedule):
            lc = [1.0 if costs is None else costs[t] for t in layer]
            st = dj.layer(layer, lc, np.stack([centers[t] for t in layer]))
            _report_failures(layer, st, "one_node_joint_infer", failed)
        return dj.table()[targets]

    def maximize_layer(table, layer, pc):
        new, _, _, _, st = ctx.maximize_batch(table, layer, cfg, pos_centers=pc, raise_on_error=False)
        _report_failures(layer, st, "one_node_joint_infer", failed)
        return new[layer]
    vp = joint_infer_sweeps(maximize_layer, vp, targets, neighbors, batch_size, n_iters, rng, rank, world, costs,
                            all_gather, schedule, device=ctx.device)
    return vp[targets]


# ---- box-level driver (ParallelRun.infer_box, ParallelRun.jl:610-672) ------------------------------------------------

@dataclass
class BoundingBox:
    """src/dataset.jl:1-13"""
    ramin: float
    ramax: float
    decmin: float
    decmax: float

    def __post_init__(self):
        assert self.ramax > self.ramin, "ramax must be greater than ramin"
        assert self.decmax > self.decmin, "decmax must be greater than decmin"

    def contains(self, pos) -> bool:
        return self.ramin < pos[0] < self.ramax and self.decmin < pos[1] < self.decmax


@dataclass
class OptimizedSource:
    """ParallelRun.jl:425-430"""
    init_ra: float
    init_dec: float
    vs: np.ndarray
    is_sky_bad: bool
    failed: bool = False   # the optimiser reported a non-finite ELBO for this source; vs is its last good row


def bad_sky(ce, images) -> bool:
    """ParallelRun.jl:437-460: is the claimed sky of the i-band image 5 photons per pixel below the median of the
    pixels in a 50-pixel box around the object?"""
    from .model import box_around_point, clamp_box, julia_round
    img = next((im for im in images if im.b == 4), None)
    if img is None:
        return False
    pc = img.world_to_pix(ce.pos)
    h = max(1, min(julia_round(pc[0]), img.H))
    w = max(1, min(julia_round(pc[1]), img.W))
    claimed_sky = float(img.sky[h - 1, w - 1]) * float(img.nelec_per_nmgy[h - 1])
    (h0, h1), (w0, w1) = clamp_box(box_around_point(img, ce.pos, 50.0), (img.H, img.W))
    px = img.pixels[h0 - 1:h1, w0 - 1:w1]
    px = px[~np.isnan(px)]
    if px.size == 0:
        return False
    # median(px) by selection (what np.median does, without its per-call overhead: 2000 sources took 0.3 s)
    k = px.size // 2
    if px.size % 2:
        med = float(np.partition(px, k)[k])
    else:
        part = np.partition(px, [k - 1, k])
        med = float((part[k - 1] + part[k]) * px.dtype.type(0.5))
    return (claimed_sky + 5) < med


def bad_sky_flags(entries, images, device=None, force_torch: bool = False) -> List[bool]:
    """bad_sky for many catalog entries at once: the 101 x 101 boxes of all sources are gathered and sorted on the GPU
    (torch: device memory and a sort -- plumbing), the median taken from the sorted rows exactly as np.median takes it.
    Used when torch's CUDA runtime is already up in this process (its first use costs ~0.8 s, more than the per-entry
    loop takes for a whole field); otherwise, and for a handful of sources, the per-entry function is used."""
    entries = list(entries)
    img = next((im for im in images if im.b == 4), None)
    if img is None or not entries:
        return [False] * len(entries)
    try:
        import torch
        use = force_torch or (torch.cuda.is_available() and torch.cuda.is_initialized() and len(entries) >= 64)
    except ImportError:       # pragma: no cover
        use = False
    if not use:
        return [bad_sky(ce, images) for ce in entries]
    dev = torch.device("cuda", device or 0) if torch.cuda.is_available() else torch.device("cpu")
    H, W = img.H, img.W
    pos = np.array([ce.pos for ce in entries], dtype=np.float64).reshape(-1, 2)
    pc = (pos - img.wcs_world0) @ np.asarray(img.wcs_jacobian).T + img.wcs_pix0          # world_to_pix, row by row
    hc = np.clip(np.rint(pc[:, 0]).astype(np.int64), 1, H)
    wc = np.clip(np.rint(pc[:, 1]).astype(np.int64), 1, W)
    claimed = img.sky[hc - 1, wc - 1].astype(np.float64) * np.asarray(img.nelec_per_nmgy)[hc - 1].astype(np.float64)
    # clamp_box(box_around_point(img, pos, 50), (H, W)): inclusive 1-based ranges, possibly empty
    h0 = np.clip(np.rint(pc[:, 0] - 50.0).astype(np.int64), 1, H + 1); h1 = np.clip(np.rint(pc[:, 0] + 50.0).astype(np.int64), 0, H)
    w0 = np.clip(np.rint(pc[:, 1] - 50.0).astype(np.int64), 1, W + 1); w1 = np.clip(np.rint(pc[:, 1] + 50.0).astype(np.int64), 0, W)
    px = torch.from_numpy(np.ascontiguousarray(img.pixels, dtype=np.float32)).to(dev)
    out = np.zeros(len(entries), dtype=bool)
    B = 102                                                                              # a box has at most 102 rows / columns
    off = torch.arange(B, device=dev)
    for lo in range(0, len(entries), 4096):
        sl = slice(lo, min(len(entries), lo + 4096))
        th0 = torch.from_numpy(h0[sl]).to(dev); th1 = torch.from_numpy(h1[sl]).to(dev)
        tw0 = torch.from_numpy(w0[sl]).to(dev); tw1 = torch.from_numpy(w1[sl]).to(dev)
        rows = th0[:, None] + off[None, :]; cols = tw0[:, None] + off[None, :]           # 1-based
        okr = rows <= th1[:, None]; okc = cols <= tw1[:, None]
        vals = px[(rows.clamp(1, H) - 1)[:, :, None], (cols.clamp(1, W) - 1)[:, None, :]]
        vals = torch.where(okr[:, :, None] & okc[:, None, :], vals, torch.full_like(vals, float("nan"))).reshape(vals.shape[0], -1)
        n = (~torch.isnan(vals)).sum(dim=1)
        srt = torch.sort(vals, dim=1).values                                             # NaNs last
        k = torch.div(n, 2, rounding_mode="floor")
        hi = srt.gather(1, k.clamp(max=srt.shape[1] - 1)[:, None])[:, 0]
        lo_ = srt.gather(1, (k - 1).clamp(min=0)[:, None])[:, 0]
        med = torch.where(n % 2 == 1, hi, (lo_ + hi) * 0.5)                              # float32, like np.median
        res = (torch.from_numpy(claimed[sl] + 5.0).to(dev) < med.double()) & (n > 0)
        out[sl] = res.cpu().numpy()
    return [bool(x) for x in out]


def group_single_infer(group, catalog, target_sources: Sequence[int], cfg: Optional[ElboConfig] = None,
                       failed: Optional[set] = None) -> np.ndarray:
    """one_node_single_infer (ParallelRun.jl:546-607) over the devices of a `group.FieldGroup`: the N workers of the
    reference's loop are the group's members (celeste_group_maximize_batch)."""
    vp_nbr = init_source_table(catalog)
    vp = init_source_table(catalog, target_sources)
    new, _, _, _, st = group.maximize_batch(vp, list(target_sources), cfg or default_infer_config(), vp_neighbors=vp_nbr,
                                            raise_on_error=False)
    _report_failures(target_sources, st, "one_node_single_infer", failed)
    return new[list(target_sources)]


def group_joint_infer(group, catalog, target_sources: Sequence[int], neighbors: List[List[int]],
                      cfg: Optional[ElboConfig] = None, batch_size: int = 400, n_iters: int = NUM_JOINT_VI_ITERS,
                      rng: Optional[np.random.Generator] = None, failed: Optional[set] = None) -> np.ndarray:
    """one_node_joint_infer (ParallelRun.jl:135-196) over the devices of a `group.FieldGroup`: the connected components of
    every Cyclades batch are sharded over the members, and rows are exchanged in front of every batch in which a member reads a
    row another member wrote since the last exchange -- at most once per batch, once in all for a group of one
    (celeste_group_joint_infer).  Same table as one_node_joint_infer on one device, bit for bit.
    batch_size is the reference's (Config: 400 sources per Cyclades batch, sized for CPU threads); several devices want batches
    of thousands -- between two exchanges a member's launch ends on its slowest component."""
    from .group import cyclades_schedule
    targets = list(target_sources)
    vp = init_source_table(catalog, targets)
    b_off, c_off, flat = cyclades_schedule(targets, neighbors, batch_size=batch_size, rng=rng)
    pos = vp[flat, 0:2].copy()                                   # boxes stay at the initial positions
    vp, _, _, _, st, _ = group.joint_infer(vp, b_off, c_off, flat, n_iters, cfg or default_infer_config(), pos_centers=pos)
    for sweep in st:
        _report_failures([int(t) for t in flat], sweep, "one_node_joint_infer", failed)
    return vp[targets]


def infer_box(images, box: BoundingBox, catalog, method: str = "joint_vi", cfg: Optional[ElboConfig] = None,
              n_iters: int = NUM_JOINT_VI_ITERS, device: int = 0, schedule: str = "cyclades",
              devices: Optional[Sequence[int]] = None) -> List[OptimizedSource]:
    """infer_box / _infer_box (ParallelRun.jl:610-672) for a given catalog: patches for every catalog entry, targets
    = entries strictly inside the box, neighbours may lie outside it, then joint or single variational inference
    on the device.  (Source detection and MCMC are out of scope: `catalog` is required, method in {joint_vi, single_vi}.)
    devices: HIP ordinals of a device group (celeste_group_*: one process, the reference's N workers = N devices, RCCL
    inside the library); None = the one `device`."""
    targets = [i for i, ce in enumerate(catalog) if box.contains(ce.pos)]
    if not targets:
        return []
    if devices is not None:
        from .group import FieldGroup
        if schedule != "cyclades":
            raise ValueError("a device group runs the reference's Cyclades schedule")
        group = FieldGroup.from_catalog(images, catalog, devices=list(devices), sparse=len(images) > 5)
        failed = set()
        try:
            if method == "joint_vi":
                vs = group_joint_infer(group, catalog, targets, group.problem.neighbors, cfg, n_iters=n_iters, failed=failed)
            elif method == "single_vi":
                vs = group_single_infer(group, catalog, targets, cfg, failed=failed)
            else:
                raise ValueError("unknown method: %s" % method)
        finally:
            group.close()
        flags = bad_sky_flags([catalog[t] for t in targets], images, devices[0])
        return [OptimizedSource(float(catalog[t].pos[0]), float(catalog[t].pos[1]), vs[k].copy(), flags[k], t in failed)
                for k, t in enumerate(targets)]
    # patches and neighbour lists as arrays (model.patch_table: the geometry of get_sky_patches / find_neighbors
    # without a Python object per patch); several fields: sources see a few images each -> sparse patch list
    ctx = FieldContext.from_catalog(images, catalog, device=device, sparse=len(images) > 5)
    neighbors = ctx.problem.neighbors
    failed: set = set()
    try:
        if method == "joint_vi":
            vs = one_node_joint_infer(ctx, catalog, targets, neighbors, cfg, n_iters=n_iters, schedule=schedule,
                                      failed=failed)
        elif method == "single_vi":
            vs = one_node_single_infer(ctx, catalog, targets, cfg, failed=failed)
        else:
            raise ValueError("unknown method: %s" % method)
    finally:
        ctx.close()
    flags = bad_sky_flags([catalog[t] for t in targets], images, device)
    return [OptimizedSource(float(catalog[t].pos[0]), float(catalog[t].pos[1]), vs[k].copy(), flags[k], t in failed)
            for k, t in enumerate(targets)]
