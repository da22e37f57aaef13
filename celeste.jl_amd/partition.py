"""Source-partition loop of the reference, re-cast as the multi-GPU sharder.

  reference                                               here
  ------------------------------------------------------  ------------------------------------
  estimate_time (sum of active pixels)                    estimate_time           ParallelRun.jl:45-47
  load_balance_across_threads (greedy least-loaded)       load_balance            ParallelRun.jl:49-56
  partition_equally                                       partition_equally       partition.jl:250-273
  partition_cyclades_dynamic (shuffle + union-find CCs)   partition_cyclades_dynamic  partition.jl:173-236
  one_node_single_infer work queue                        shard_targets (static, cost balanced)

Sources are independent units: a rank evaluates its shard against replicated
images; no data-path collective, only the final gather of the per-source results.
"""
from typing import Dict, List, Sequence

import numpy as np


def estimate_time(patch_row) -> int:
    """ParallelRun.jl:45-47: sum over images of active pixels of one source's patches."""
    from .model import row_entries
    return int(sum(int(p.active_pixel_bitmap.sum()) for _, p in row_entries(patch_row)))


def load_balance(n_parts: int, times: Sequence[float]) -> List[float]:
    """ParallelRun.jl:49-56: greedy assignment in the given order, returns the per-part totals."""
    ts = [0.0] * n_parts
    for t in times:
        ts[int(np.argmin(ts))] += t
    return ts


def shard_targets(costs: Sequence[float], n_parts: int) -> List[List[int]]:
    """Longest-processing-time-first greedy sharding of target indices 0..len(costs)-1.

    Deterministic (ties broken by index) so that every rank computes the same partition without
    communicating.  Shards keep ascending target order, which keeps patch reads spatially coherent
    when the catalog is spatially sorted.
    """
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0.0] * n_parts
    shards: List[List[int]] = [[] for _ in range(n_parts)]
    for i in order:
        k = min(range(n_parts), key=lambda p: (loads[p], p))
        shards[k].append(i)
        loads[k] += costs[i]
    return [sorted(s) for s in shards]


def partition_equally(n_threads: int, n_sources: int) -> List[List[List[int]]]:
    """partition.jl:250-273 (0-based indices): [thread][batch=0][sources]."""
    per = n_sources // n_threads
    out = []
    for t in range(n_threads):
        start, end = t * per, (t + 1) * per
        if t == n_threads - 1:
            end = n_sources
        out.append([list(range(start, end))])
    return out


def _find(i: int, tree: List[int]) -> int:
    root = i
    while tree[root] != root:
        root = tree[root]
    while tree[i] != root:
        tree[i], i = root, tree[i]
    return root


def partition_cyclades_dynamic(target_sources: Sequence[int], neighbor_map: Dict[int, Sequence[int]],
                               batch_size: int = 60, rng: np.random.Generator = None) -> List[List[List[int]]]:
    """partition.jl:173-236: shuffle the sources, cut into batches of `batch_size`, and return for each
    batch the connected components (under the neighbour graph restricted to the batch) as lists of
    *indices into target_sources*.  Components of one batch never conflict, so they may be evaluated
    concurrently (one GPU launch per batch); batches are processed in order.
    """
    rng = rng or np.random.default_rng(42)
    n = len(target_sources)
    src_to_idx = {s: i for i, s in enumerate(target_sources)}
    sources = list(neighbor_map.keys())
    assert sorted(sources) == sorted(target_sources), "neighbor_map keys must be the target sources"
    rng.shuffle(sources)
    batches: List[List[List[int]]] = []
    for start in range(0, n, batch_size):
        batch = sources[start:start + batch_size]
        local = {s: i for i, s in enumerate(batch)}
        tree = list(range(len(batch)))
        for i, s in enumerate(batch):
            target = _find(i, tree)
            for nb in neighbor_map[s]:
                if nb in local:
                    tree[_find(local[nb], tree)] = target
        comps: Dict[int, List[int]] = {}
        for i, s in enumerate(batch):
            comps.setdefault(_find(i, tree), []).append(src_to_idx[s])
        batches.append(list(comps.values()))
    assert sum(len(c) for b in batches for c in b) == n
    return batches


def color_classes(target_sources: Sequence[int], neighbor_map: Dict[int, Sequence[int]]) -> List[List[int]]:
    """Greedy colouring of the neighbour graph (largest degree first): every class is an independent set, so its
    sources can be optimised in one launch, and processing the classes in order is one sequential sweep over the
    sources -- the same serialisability guarantee Cyclades batches give (partition.jl:37-73), with as many launches
    per sweep as there are colours (max degree + 1 at most) instead of the longest connected component of a batch.
    Returns lists of *indices into target_sources*."""
    idx = {s: i for i, s in enumerate(target_sources)}
    order = sorted(target_sources, key=lambda s: -len(neighbor_map[s]))
    color: Dict[int, int] = {}
    for s in order:
        used = {color[n] for n in neighbor_map[s] if n in color}
        c = 0
        while c in used:
            c += 1
        color[s] = c
    classes: List[List[int]] = [[] for _ in range(max(color.values()) + 1)] if color else []
    for s in target_sources:
        classes[color[s]].append(idx[s])
    return classes
