"""Node-level inference loops on top of the device optimiser.

  reference (src/ParallelRun.jl)                              here
  ----------------------------------------------------------  -----------------------------------------
  one_node_single_infer  (:546-607) -> process_source (:468)  one_node_single_infer
  one_node_joint_infer   (:135-196): setup_vecs, Cyclades     one_node_joint_infer
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
from typing import List, Optional, Sequence

import numpy as np

from .elbo import ElboConfig, FieldContext
from .params import catalog_init_source, generic_init_source
from .partition import partition_cyclades_dynamic

NUM_JOINT_VI_ITERS = 3   # Config.num_joint_vi_iters (src/config.jl:17-25)


def one_node_single_infer(ctx: FieldContext, catalog, target_sources: Sequence[int],
                          cfg: Optional[ElboConfig] = None) -> np.ndarray:
    """Returns the optimised parameters, one row per target (OptimizedSource.vs)."""
    vp_nbr = np.stack([catalog_init_source(ce) for ce in catalog])
    vp = vp_nbr.copy()
    for t in target_sources:
        vp[t] = generic_init_source(catalog[t].pos)
    new, _, _, _, st = ctx.maximize_batch(vp, list(target_sources), cfg, vp_neighbors=vp_nbr)
    return new[list(target_sources)]


def one_node_joint_infer(ctx: FieldContext, catalog, target_sources: Sequence[int], neighbors: List[List[int]],
                         cfg: Optional[ElboConfig] = None, batch_size: int = 400, n_iters: int = NUM_JOINT_VI_ITERS,
                         rng: Optional[np.random.Generator] = None) -> np.ndarray:
    """Cyclades-batched joint inference; returns the optimised parameters, one row per target."""
    targets = list(target_sources)
    tset = set(targets)
    vp = np.stack([catalog_init_source(ce) for ce in catalog])
    for t in targets:
        vp[t] = generic_init_source(catalog[t].pos)
    centers = {t: vp[t, 0:2].copy() for t in targets}        # boxes stay at the initial positions
    nmap = {t: [n for n in neighbors[t] if n in tset] for t in targets}
    batches = partition_cyclades_dynamic(targets, nmap, batch_size=batch_size,
                                         rng=rng or np.random.default_rng(42))   # srand(42), ParallelRun.jl:143
    for _ in range(n_iters):
        for components in batches:
            depth = max(len(c) for c in components)
            for j in range(depth):
                layer = [targets[c[j]] for c in components if len(c) > j]
                pc = np.stack([centers[t] for t in layer])
                vp, _, _, _, st = ctx.maximize_batch(vp, layer, cfg, pos_centers=pc)
    return vp[targets]
