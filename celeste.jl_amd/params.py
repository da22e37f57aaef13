"""Parameter index tables and source initialisers.

Mirrors src/model/param_set.jl:76-107 (CanonicalParams, 0-based here),
src/model/light_source_model.jl:11-20 (CatalogEntry) and
src/DeterministicVI.jl:39-91 (generic_init_source / catalog_init_source).
"""
from dataclasses import dataclass
from typing import List
import math
import numpy as np

NUM_BANDS = 5
NUM_SOURCE_TYPES = 2
NUM_COLOR_COMPONENTS = 8
P = 44


class _Ids:
    """ids (0-based).  Matrices are indexed [row, type] like the reference."""
    pos = np.array([0, 1])
    gal_frac_dev = 2
    gal_axis_ratio = 3
    gal_angle = 4
    gal_radius_px = 5
    flux_loc = np.array([6, 7])
    flux_scale = np.array([8, 9])
    color_mean = np.arange(10, 18).reshape(2, 4).T      # [c, i]
    color_var = np.arange(18, 26).reshape(2, 4).T       # [c, i]
    is_star = np.array([26, 27])
    k = np.arange(28, 44).reshape(2, 8).T               # [d, i]


ids = _Ids()


def ids_names() -> List[str]:
    names = [""] * P
    for name in ("pos", "flux_loc", "flux_scale", "is_star"):
        for j, i in enumerate(getattr(ids, name)):
            names[i] = "%s_%d" % (name, j + 1)
    for name in ("gal_frac_dev", "gal_axis_ratio", "gal_angle", "gal_radius_px"):
        names[getattr(ids, name)] = name
    for name in ("color_mean", "color_var", "k"):
        m = getattr(ids, name)
        for r in range(m.shape[0]):
            for c in range(m.shape[1]):
                names[m[r, c]] = "%s_%d_%d" % (name, r + 1, c + 1)
    return names


@dataclass
class CatalogEntry:
    """light_source_model.jl:11-20"""
    pos: np.ndarray
    is_star: bool
    star_fluxes: np.ndarray
    gal_fluxes: np.ndarray
    gal_frac_dev: float
    gal_axis_ratio: float
    gal_angle: float
    gal_radius_px: float


def generic_init_source(init_pos) -> np.ndarray:
    """DeterministicVI.jl:39-53"""
    ret = np.empty(P)
    ret[ids.is_star] = 0.5
    ret[ids.pos] = init_pos
    ret[ids.flux_loc] = math.log(2.0)
    ret[ids.flux_scale] = 1e-3
    ret[ids.gal_frac_dev] = 0.5
    ret[ids.gal_axis_ratio] = 0.5
    ret[ids.gal_angle] = 0.0
    ret[ids.gal_radius_px] = 1.0
    ret[ids.k] = 1.0 / NUM_COLOR_COMPONENTS
    ret[ids.color_mean] = 0.0
    ret[ids.color_var] = 1e-2
    return ret


def _jmax(a: float, b: float) -> float:
    """Julia's max: NaN if either argument is (Python's max(0.1, nan) is 0.1, np.maximum agrees with Julia)"""
    return float(np.maximum(a, b))


def _jmin(a: float, b: float) -> float:
    return float(np.minimum(a, b))


def _get_color(c2: float, c1: float) -> float:
    if c2 > 0 and c1 > 0:
        return _jmin(_jmax(math.log(c2 / c1), -9.0), 9.0)
    if c2 > 0 and c1 <= 0:
        return 3.0
    if c2 <= 0 and c1 > 0:
        return -3.0
    return 0.0


def catalog_init_source(ce: CatalogEntry, max_gal_radius_px=math.inf) -> np.ndarray:
    """DeterministicVI.jl:59-91"""
    ret = generic_init_source(ce.pos)
    ret[ids.is_star[0]] = 0.8 if ce.is_star else 0.2
    ret[ids.is_star[1]] = 0.2 if ce.is_star else 0.8
    ret[ids.flux_loc[0]] = math.log(_jmax(0.1, ce.star_fluxes[2]))
    ret[ids.flux_loc[1]] = math.log(_jmax(0.1, ce.gal_fluxes[2]))
    ret[ids.color_mean[:, 0]] = [_get_color(ce.star_fluxes[c + 1], ce.star_fluxes[c]) for c in range(4)]
    ret[ids.color_mean[:, 1]] = [_get_color(ce.gal_fluxes[c + 1], ce.gal_fluxes[c]) for c in range(4)]
    ret[ids.gal_frac_dev] = _jmin(_jmax(ce.gal_frac_dev, 0.015), 0.985)
    ret[ids.gal_axis_ratio] = 0.8 if ce.is_star else _jmin(_jmax(ce.gal_axis_ratio, 0.015), 0.985)
    ret[ids.gal_angle] = ce.gal_angle
    ret[ids.gal_radius_px] = 0.2 if ce.is_star else _jmin(max_gal_radius_px, _jmax(ce.gal_radius_px, 0.2))
    return ret


def init_sources(target_sources, catalog) -> List[np.ndarray]:
    """DeterministicVI.jl:94-103 (target_sources 0-based)"""
    ret = [catalog_init_source(ce) for ce in catalog]
    for s in target_sources:
        ret[s][:] = generic_init_source(catalog[s].pos)
    return ret


def init_source_table(catalog, target_sources=(), max_gal_radius_px=math.inf) -> np.ndarray:
    """init_sources (DeterministicVI.jl:94-103) as ONE [S, 44] array: catalog_init_source for every entry, then
    generic_init_source for the targets -- the table ParallelRun.setup_vecs builds (ParallelRun.jl:96-132).  Same values as
    the per-entry functions, bit for bit (the logarithms go through math.log like theirs); 30 000 entries take 0.1 s
    instead of 0.8 s."""
    S = len(catalog)
    ret = np.empty((S, P))
    if S == 0:
        return ret
    ret[:] = generic_init_source((0.0, 0.0))
    ret[:, ids.pos] = np.array([ce.pos for ce in catalog], dtype=np.float64).reshape(S, 2)
    star = np.fromiter((bool(ce.is_star) for ce in catalog), dtype=bool, count=S)
    ret[:, ids.is_star[0]] = np.where(star, 0.8, 0.2)
    ret[:, ids.is_star[1]] = np.where(star, 0.2, 0.8)
    sf = np.array([ce.star_fluxes for ce in catalog], dtype=np.float64).reshape(S, -1)
    gf = np.array([ce.gal_fluxes for ce in catalog], dtype=np.float64).reshape(S, -1)
    ret[:, ids.flux_loc[0]] = [math.log(x) for x in np.maximum(0.1, sf[:, 2]).tolist()]
    ret[:, ids.flux_loc[1]] = [math.log(x) for x in np.maximum(0.1, gf[:, 2]).tolist()]

    def colors(f, c):       # _get_color(f[c + 1], f[c]) for every row; the logarithm through math.log like the scalar function
        c2, c1 = f[:, c + 1], f[:, c]
        both = (c2 > 0) & (c1 > 0)
        out = np.where((c2 > 0) & (c1 <= 0), 3.0, np.where((c2 <= 0) & (c1 > 0), -3.0, 0.0))
        if both.any():
            out[both] = np.clip([math.log(r) for r in (c2[both] / c1[both]).tolist()], -9.0, 9.0)
        return out
    for c in range(4):
        ret[:, ids.color_mean[c, 0]] = colors(sf, c)
        ret[:, ids.color_mean[c, 1]] = colors(gf, c)
    ret[:, ids.gal_frac_dev] = np.clip(np.array([ce.gal_frac_dev for ce in catalog], dtype=np.float64), 0.015, 0.985)
    ab = np.clip(np.array([ce.gal_axis_ratio for ce in catalog], dtype=np.float64), 0.015, 0.985)
    ret[:, ids.gal_axis_ratio] = np.where(star, 0.8, ab)
    ret[:, ids.gal_angle] = [ce.gal_angle for ce in catalog]
    rad = np.minimum(max_gal_radius_px, np.maximum(np.array([ce.gal_radius_px for ce in catalog], dtype=np.float64), 0.2))
    ret[:, ids.gal_radius_px] = np.where(star, 0.2, rad)
    tg = np.asarray(list(target_sources), dtype=np.int64)
    if tg.size:
        pos = ret[tg][:, ids.pos].copy()
        ret[tg] = generic_init_source((0.0, 0.0))
        ret[tg[:, None], np.asarray(ids.pos)[None, :]] = pos
    return ret


def perturb_params(vp) -> None:
    """test/SampleData.jl:127-141: move parameters away from the truth."""
    for vs in vp:
        vs[ids.is_star] = [0.4, 0.6]
        vs[ids.pos[0]] += .8
        vs[ids.pos[1]] -= .7
        vs[ids.flux_loc] -= math.log(10)
        vs[ids.flux_scale] *= 25.
        vs[ids.gal_frac_dev] += 0.05
        vs[ids.gal_axis_ratio] += 0.05
        vs[ids.gal_angle] += math.pi / 10
        vs[ids.gal_radius_px] *= 1.2
        vs[ids.color_mean] += 0.5
        vs[ids.color_var] = 1e-1
