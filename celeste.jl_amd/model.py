"""Host-side data model: images, PSF components, image patches, neighbours.

This is input preparation for the ELBO hot path (it runs once per inference,
never per elbo() call) and mirrors, name for name, the reference's
  src/model/image_model.jl:6-38        Image
  src/model/psf_model.jl:17-75         PsfComponent, get_psf_width, render_psf, ConstantPSFMap
  src/model/imaged_sources.jl:10-244   boxes, ImagePatch, box_from_catalog, get_sky_patches,
                                       choose_patch_radius, find_neighbors
  src/model/wcs_utils.jl:14-18         linear_world_to_pix
Only linear (affine) WCS is supported: the synthetic configurations use the
identity WCS of test/SampleData.jl:30-34.
"""
import collections.abc
from dataclasses import dataclass, field
from typing import Dict, List, Sequence, Tuple
import math
import numpy as np

from .params import CatalogEntry

STAMP = 51


def julia_round(x: float) -> int:
    """round(Int, x) with ties to even (RoundNearest), imaged_sources.jl:126-129."""
    return int(np.rint(x))


def make_psf(alpha: Sequence[float], xi: Sequence[Sequence[float]], tau: Sequence[np.ndarray]) -> np.ndarray:
    """K x 6 array {alphaBar, xiBar1, xiBar2, tauBar11, tauBar12, tauBar22} (psf_model.jl:17-29)."""
    out = np.zeros((len(alpha), 6))
    for k in range(len(alpha)):
        t = np.asarray(tau[k], dtype=float)
        out[k] = [alpha[k], xi[k][0], xi[k][1], t[0, 0], t[0, 1], t[1, 1]]
    return out


def get_psf_width(psf: np.ndarray, width_scale: float = 1.0) -> float:
    """psf_model.jl:32-52"""
    alpha_norm = psf[:, 0].sum()
    cov = np.zeros((2, 2))
    for a, x1, x2, t11, t12, t22 in psf:
        xi = np.array([x1, x2])
        cov += a * (np.outer(xi, xi) + np.array([[t11, t12], [t12, t22]])) / alpha_norm
    return width_scale * math.sqrt(np.linalg.eigvalsh(cov)[-1]) * alpha_norm


def render_psf(psf: np.ndarray, dims: Tuple[int, int] = (STAMP, STAMP)) -> np.ndarray:
    """psf_model.jl:61-75: stamp[i, j] = sum_k alphaBar_k N((i, j) - center; xiBar_k, tauBar_k)."""
    c0, c1 = (dims[0] + 1) / 2, (dims[1] + 1) / 2
    ii = np.arange(1, dims[0] + 1)[:, None] - c0
    jj = np.arange(1, dims[1] + 1)[None, :] - c1
    stamp = np.zeros(dims)
    for a, x1, x2, t11, t12, t22 in psf:
        det = t11 * t22 - t12 * t12
        d0, d1 = ii - x1, jj - x2
        q = (t22 * d0 * d0 - 2 * t12 * d0 * d1 + t11 * d1 * d1) / det
        stamp += a * np.exp(-0.5 * q) / (2 * math.pi * math.sqrt(det))
    return stamp


@dataclass
class ConstantPSFMap:
    """psf_model.jl:89-92"""
    stamp: np.ndarray

    def __post_init__(self):
        self.stamp = np.array(self.stamp, dtype=np.float64)
        self.stamp.setflags(write=False)   # shared by every patch of the image

    def __call__(self, x, y):
        return self.stamp


@dataclass
class SDSSPSFMap:
    """src/SDSSIO.jl:239-299: spatially variable PSF = eigen-images (columns of rrows, each a flattened
    rnrow x rncol stamp, column-major) weighted by polynomials
    w_k(x, y) = sum_ij cmat[i, j, k] (RCS (x - 1))^i (RCS (y - 1))^j, RCS = 0.001, x / y one-based pixel coordinates."""
    rrows: np.ndarray   # (rnrow * rncol) x nk
    rnrow: int
    rncol: int
    cmat: np.ndarray    # ni x nj x nk

    def __post_init__(self):
        self.rrows = np.asarray(self.rrows, dtype=np.float64)
        self.cmat = np.asarray(self.cmat, dtype=np.float64)
        assert self.rrows.shape[0] == self.rnrow * self.rncol
        assert self.rrows.shape[1] == self.cmat.shape[2]

    def __call__(self, x, y):
        RCS = 0.001
        px = (RCS * (x - 1.0)) ** np.arange(self.cmat.shape[0])
        py = (RCS * (y - 1.0)) ** np.arange(self.cmat.shape[1])
        w = np.einsum("ijk,i,j->k", self.cmat, px, py)
        return (self.rrows @ w).reshape(self.rncol, self.rnrow).T.copy()   # column-major stamp


@dataclass
class SDSSBackground:
    """src/SDSSIO.jl:56-99: the sky plane of an SDSS frame, bilinear interpolation of a small sky image at the
    per-row / per-column coordinates sky_x / sky_y (constant extrapolation), times the per-row calibration.
    All arithmetic in Float32 like the reference (including its weight convention: the fractional offset
    xw0 = sky_x - floor(sky_x) multiplies the *lower* sample).  `materialize()` gives the H x W float32 plane
    that celeste_image_t.sky expects."""
    sky_small: np.ndarray
    sky_x: np.ndarray
    sky_y: np.ndarray
    calibration: np.ndarray

    def __post_init__(self):
        self.sky_small = np.asarray(self.sky_small, dtype=np.float32)
        self.sky_x = np.asarray(self.sky_x, dtype=np.float32)
        self.sky_y = np.asarray(self.sky_y, dtype=np.float32)
        self.calibration = np.asarray(self.calibration, dtype=np.float32)
        assert self.calibration.shape == self.sky_x.shape

    @property
    def shape(self):
        return (self.sky_x.size, self.sky_y.size)

    def materialize(self) -> np.ndarray:
        nx, ny = self.sky_small.shape
        one = np.float32(1.0)
        x0 = np.floor(self.sky_x).astype(np.int64); xw0 = (self.sky_x - x0.astype(np.float32)).astype(np.float32)
        y0 = np.floor(self.sky_y).astype(np.int64); yw0 = (self.sky_y - y0.astype(np.float32)).astype(np.float32)
        xw1 = (one - xw0).astype(np.float32); yw1 = (one - yw0).astype(np.float32)
        x1 = np.clip(x0 + 1, 1, nx) - 1; x0 = np.clip(x0, 1, nx) - 1
        y1 = np.clip(y0 + 1, 1, ny) - 1; y0 = np.clip(y0, 1, ny) - 1
        s = self.sky_small
        dns = ((xw0[:, None] * yw0[None, :]).astype(np.float32) * s[np.ix_(x0, y0)]
               + (xw1[:, None] * yw0[None, :]).astype(np.float32) * s[np.ix_(x1, y0)]
               + (xw0[:, None] * yw1[None, :]).astype(np.float32) * s[np.ix_(x0, y1)]
               + (xw1[:, None] * yw1[None, :]).astype(np.float32) * s[np.ix_(x1, y1)]).astype(np.float32)
        return (dns * self.calibration[:, None]).astype(np.float32)

    def __getitem__(self, ij):
        i, j = ij   # 1-based like the reference
        return SDSSBackground(self.sky_small, self.sky_x[i - 1:i], self.sky_y[j - 1:j],
                              self.calibration[i - 1:i]).materialize()[0, 0]


@dataclass
class Image:
    """image_model.jl:6-38.  pixels/sky are H x W (first index = row h), float32."""
    pixels: np.ndarray
    b: int                       # band 1..5
    psf: np.ndarray              # K x 6
    sky: np.ndarray              # H x W float32, nmgy
    nelec_per_nmgy: np.ndarray   # H float32
    psfmap: object               # ConstantPSFMap or SDSSPSFMap: (x, y) -> 51 x 51 raw stamp
    wcs_jacobian: np.ndarray = field(default_factory=lambda: np.eye(2))  # affine WCS: pix = J (world - w0) + p0
    wcs_world0: np.ndarray = field(default_factory=lambda: np.zeros(2))
    wcs_pix0: np.ndarray = field(default_factory=lambda: np.zeros(2))

    @property
    def H(self):
        return self.pixels.shape[0]

    @property
    def W(self):
        return self.pixels.shape[1]

    def world_to_pix(self, world):
        return self.wcs_jacobian @ (np.asarray(world, float) - self.wcs_world0) + self.wcs_pix0

    def pix_to_world(self, pix):
        return np.linalg.solve(self.wcs_jacobian, np.asarray(pix, float) - self.wcs_pix0) + self.wcs_world0


Box = Tuple[Tuple[int, int], Tuple[int, int]]  # ((first, last), (first, last)), inclusive, 1-based


def clamp_box(box: Box, dims: Tuple[int, int]) -> Box:
    """imaged_sources.jl:10-14 (empty ranges allowed)."""
    def cl(v, lo, hi):
        return min(max(v, lo), hi)
    return ((cl(box[0][0], 1, dims[0] + 1), cl(box[0][1], 0, dims[0])),
            (cl(box[1][0], 1, dims[1] + 1), cl(box[1][1], 0, dims[1])))


def _ranges_overlap(r1, r2) -> bool:
    return r1[0] <= r2[1] and r2[0] <= r1[1]


def boxes_overlap(b1: Box, b2: Box) -> bool:
    """imaged_sources.jl:36-40"""
    return _ranges_overlap(b1[0], b2[0]) and _ranges_overlap(b1[1], b2[1])


@dataclass
class ImagePatch:
    """imaged_sources.jl:60-117"""
    box: Box
    world_center: np.ndarray
    psf: np.ndarray
    stamp: np.ndarray              # raw psfmap(pixel_center) output; conditioned + prefiltered downstream
    wcs_jacobian: np.ndarray
    pixel_center: np.ndarray
    bitmap_offset: Tuple[int, int]
    active_pixel_bitmap: np.ndarray  # H2 x W2 bool
    stamp_id: int = -1               # index into a shared stamp table (set by the context builder)

    @classmethod
    def from_box(cls, img: Image, box: Box, stamp_id: int = -1) -> "ImagePatch":
        box = clamp_box(box, (img.H, img.W))
        pixel_center = np.array([(box[0][0] + box[0][1]) / 2, (box[1][0] + box[1][1]) / 2])
        world_center = img.pix_to_world(pixel_center)
        off = (box[0][0] - 1, box[1][0] - 1)
        h2 = max(box[0][1] - box[0][0] + 1, 0)
        w2 = max(box[1][1] - box[1][0] + 1, 0)
        sub = img.pixels[off[0]:off[0] + h2, off[1]:off[1] + w2]
        bitmap = ~np.isnan(sub)
        return cls(box, world_center, img.psf, img.psfmap(pixel_center[0], pixel_center[1]),
                   img.wcs_jacobian.copy(), pixel_center, off, bitmap, stamp_id)


def empty_patch(img: Image) -> ImagePatch:
    """The patch of a source that lies off the image: clamp_box leaves an empty range (imaged_sources.jl:10-14), so
    it covers no pixel and overlaps nothing.  One shared object per image."""
    p = getattr(img, "_empty_patch", None)
    if p is None:
        p = ImagePatch(((1, 0), (1, 0)), img.pix_to_world(np.array([0.5, 0.5])), img.psf, img.psfmap(0.5, 0.5),
                       img.wcs_jacobian.copy(), np.array([0.5, 0.5]), (0, 0), np.zeros((0, 0), dtype=bool))
        img._empty_patch = p
    return p


class PatchRow(collections.abc.Sequence):
    """patches[s, :] of a many-image problem (overlapping fields): an ImagePatch is stored only for the images the
    source overlaps; every other index yields the image's `empty_patch`.  Behaves like the dense row (len, index,
    iteration) so that callers written against patches[s][n] keep working; `cabi.Problem` hands rows of this type
    to the library as the sparse patch list of celeste_problem_t."""
    __slots__ = ("images", "entries")

    def __init__(self, images: Sequence[Image], entries: Dict[int, ImagePatch]):
        self.images = images
        self.entries = dict(sorted(entries.items()))

    def __len__(self):
        return len(self.images)

    def __getitem__(self, n):
        if isinstance(n, slice):
            return [self[i] for i in range(*n.indices(len(self)))]
        if n < 0:
            n += len(self)
        if not 0 <= n < len(self):
            raise IndexError(n)
        p = self.entries.get(n)
        return p if p is not None else empty_patch(self.images[n])

    def nonempty(self):
        """(image index, patch) pairs in image order"""
        return self.entries.items()


def row_entries(row):
    """(image index, patch) for the non-empty patches of a dense or sparse row"""
    if isinstance(row, PatchRow):
        return row.nonempty()
    return [(n, p) for n, p in enumerate(row) if p.active_pixel_bitmap.size > 0]


def box_around_point(img: Image, world_center, pixel_radius: float) -> Box:
    """imaged_sources.jl:120-136"""
    pc = img.world_to_pix(world_center)
    return ((julia_round(pc[0] - pixel_radius), julia_round(pc[0] + pixel_radius)),
            (julia_round(pc[1] - pixel_radius), julia_round(pc[1] + pixel_radius)))


def choose_patch_radius(ce: CatalogEntry, img: Image, width_scale=1.0, max_radius=25, _cache=None) -> float:
    """imaged_sources.jl:197-223.  _cache (a dict, optional): the image's PSF width and sky level, which do not
    depend on the source, are computed once per image by callers that loop over a catalog."""
    key = (id(img), width_scale)
    if _cache is not None and key in _cache:
        psf_width, epsilon = _cache[key]
    else:
        psf_width = get_psf_width(img.psf, width_scale=width_scale)
        epsilon = float(img.sky[img.H // 2 - 1, img.W // 2 - 1])
        if _cache is not None:
            _cache[key] = (psf_width, epsilon)
    obj_width = 0.0 if ce.is_star else width_scale * ce.gal_radius_px / 0.67
    obj_width += psf_width
    flux = ce.star_fluxes[img.b - 1] if ce.is_star else ce.gal_fluxes[img.b - 1]
    if not flux > 0.:
        raise AssertionError("flux > 0")
    pdf_90 = math.exp(-0.5 * 1.64 ** 2) / (math.sqrt(2 * math.pi) * obj_width)
    pdf_target = min(pdf_90, epsilon / (20 * flux))
    rhs = math.log(pdf_target) + 0.5 * math.log(2 * math.pi) + math.log(obj_width)
    radius_req = math.sqrt(-2 * obj_width ** 2 * rhs)
    return min(radius_req, max_radius)


def box_from_catalog(img: Image, ce: CatalogEntry, width_scale=1.0, max_radius=25, _cache=None) -> Box:
    """imaged_sources.jl:147-160"""
    r = choose_patch_radius(ce, img, width_scale=width_scale, max_radius=max_radius, _cache=_cache)
    return box_around_point(img, ce.pos, r)


def get_sky_patches(images: Sequence[Image], catalog: Sequence[CatalogEntry],
                    radius_override_pix: float = math.nan, sparse: bool = False):
    """imaged_sources.jl:165-182.  Returns patches[s][n]; with sparse=True every row is a `PatchRow` that only
    stores the patches that cover at least one pixel (same boxes: a source is tried against an image whenever it
    lies within max_radius + 1 pixels of it)."""
    cache = {}

    def patch(img, ce):
        if math.isnan(radius_override_pix):
            box = box_from_catalog(img, ce, width_scale=1.2, _cache=cache)
        else:
            box = box_around_point(img, ce.pos, radius_override_pix)
        return ImagePatch.from_box(img, box)
    if not sparse:
        return [[patch(img, ce) for img in images] for ce in catalog]
    reach = (25.0 if math.isnan(radius_override_pix) else radius_override_pix) + 1.0
    pos = np.array([ce.pos for ce in catalog], dtype=float).reshape(-1, 2)
    entries = [dict() for _ in catalog]
    for n, img in enumerate(images):
        pc = (pos - img.wcs_world0) @ img.wcs_jacobian.T + img.wcs_pix0
        near = np.flatnonzero((pc[:, 0] > -reach) & (pc[:, 0] < img.H + 1 + reach) &
                              (pc[:, 1] > -reach) & (pc[:, 1] < img.W + 1 + reach))
        for s in near:
            p = patch(img, catalog[s])
            if p.active_pixel_bitmap.size > 0:
                entries[s][n] = p
    return [PatchRow(images, e) for e in entries]


@dataclass
class PatchTable:
    """The geometry of get_sky_patches for a whole catalog as flat arrays -- what the device library needs of the
    patches, without an ImagePatch object (51 x 51 stamp reference, bitmap array, ...) per (source, image) pair.
    Entries are ordered by (source, image); `source` / `image` index them.  dense: one entry for EVERY pair (empty
    boxes included), else only the pairs whose box covers at least one pixel (the sparse patch list)."""
    n_sources: int
    n_images: int
    dense: bool
    source: np.ndarray        # [E] int32
    image: np.ndarray         # [E] int32
    box: np.ndarray           # [E, 4] int64: first row, last row, first column, last column (1-based, inclusive, clamped)
    pixel_center: np.ndarray  # [E, 2]
    world_center: np.ndarray  # [E, 2]
    active_pixels: np.ndarray  # [E] int64: pixels of the box that are not NaN (ParallelRun.jl:45-47's cost)

    @property
    def H2(self):
        return np.maximum(self.box[:, 1] - self.box[:, 0] + 1, 0)

    @property
    def W2(self):
        return np.maximum(self.box[:, 3] - self.box[:, 2] + 1, 0)

    def costs(self) -> np.ndarray:
        """estimate_time of every source"""
        return np.bincount(self.source, weights=self.active_pixels, minlength=self.n_sources).astype(np.int64)

    def neighbors(self) -> List[List[int]]:
        """find_neighbors for every source (imaged_sources.jl:232-244): sources whose boxes overlap in some image;
        ascending, empty boxes overlap nothing"""
        ok = (self.H2 > 0) & (self.W2 > 0)
        pairs = []
        for n in range(self.n_images):
            e = np.flatnonzero(ok & (self.image == n))
            if e.size < 2:
                continue
            b = self.box[e]
            # sweep over the boxes sorted by first row: b can only overlap a (a before b) if it starts before a ends
            o = np.argsort(b[:, 0], kind="stable")
            b = b[o]; src = self.source[e[o]].astype(np.int64)
            end = np.searchsorted(b[:, 0], b[:, 1], side="right")
            cnt = np.maximum(end - (np.arange(e.size) + 1), 0)
            ia = np.repeat(np.arange(e.size), cnt)
            ib = np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt) + ia + 1
            hit = (b[ia, 2] <= b[ib, 3]) & (b[ib, 2] <= b[ia, 3])
            si, sj = src[ia[hit]], src[ib[hit]]
            pairs.append(si * self.n_sources + sj); pairs.append(sj * self.n_sources + si)
        out = [[] for _ in range(self.n_sources)]
        if pairs:
            u = np.unique(np.concatenate(pairs))       # sorted by (source, neighbour)
            a, b = u // self.n_sources, u % self.n_sources
            cut = np.searchsorted(a, np.arange(self.n_sources + 1))
            bl = b.tolist()
            out = [bl[cut[s]:cut[s + 1]] for s in range(self.n_sources)]
        return out


def patch_table(images: Sequence[Image], catalog: Sequence[CatalogEntry], radius_override_pix: float = math.nan,
                sparse: bool = False) -> PatchTable:
    """get_sky_patches (imaged_sources.jl:165-182) without the per-patch objects: the same boxes (box_from_catalog /
    box_around_point / clamp_box, called as the object path calls them), centres and active-pixel counts, as arrays.
    sparse=True keeps only the pairs that cover a pixel, tried for the same sources as get_sky_patches(sparse=True)."""
    S, N = len(catalog), len(images)
    src, img_i, boxes = [], [], []
    cache = {}
    reach = (25.0 if math.isnan(radius_override_pix) else radius_override_pix) + 1.0
    pos = np.array([ce.pos for ce in catalog], dtype=float).reshape(-1, 2)
    for n, img in enumerate(images):
        if sparse:
            pc = (pos - img.wcs_world0) @ img.wcs_jacobian.T + img.wcs_pix0
            cand = np.flatnonzero((pc[:, 0] > -reach) & (pc[:, 0] < img.H + 1 + reach) &
                                  (pc[:, 1] > -reach) & (pc[:, 1] < img.W + 1 + reach)).tolist()
        else:
            cand = range(S)
        dims = (img.H, img.W)
        for s in cand:
            ce = catalog[s]
            if math.isnan(radius_override_pix):
                box = box_from_catalog(img, ce, width_scale=1.2, _cache=cache)
            else:
                box = box_around_point(img, ce.pos, radius_override_pix)
            box = clamp_box(box, dims)
            if sparse and (box[0][1] < box[0][0] or box[1][1] < box[1][0]):
                continue
            src.append(s); img_i.append(n); boxes.append((box[0][0], box[0][1], box[1][0], box[1][1]))
    source = np.array(src, dtype=np.int32); image = np.array(img_i, dtype=np.int32)
    box = np.array(boxes, dtype=np.int64).reshape(-1, 4)
    order = np.lexsort((image, source))
    source, image, box = source[order], image[order], box[order]
    pixel_center = np.stack([(box[:, 0] + box[:, 1]) / 2, (box[:, 2] + box[:, 3]) / 2], axis=1)
    world_center = np.zeros_like(pixel_center)
    active = np.zeros(len(source), dtype=np.int64)
    h2 = np.maximum(box[:, 1] - box[:, 0] + 1, 0); w2 = np.maximum(box[:, 3] - box[:, 2] + 1, 0)
    for n, img in enumerate(images):
        e = np.flatnonzero(image == n)
        if e.size == 0:
            continue
        # pix_to_world, one LAPACK solve per centre exactly as Image.pix_to_world does it
        rhs = (pixel_center[e] - img.wcs_pix0)[:, :, None]
        world_center[e] = np.linalg.solve(np.broadcast_to(img.wcs_jacobian, (e.size, 2, 2)), rhs)[:, :, 0] + img.wcs_world0
        nan = np.isnan(img.pixels)
        if not nan.any():
            active[e] = h2[e] * w2[e]
        else:   # non-NaN pixels of every box from one summed-area table
            sat = np.zeros((img.H + 1, img.W + 1), dtype=np.int64)
            sat[1:, 1:] = np.cumsum(np.cumsum(~nan, axis=0, dtype=np.int64), axis=1)
            r0, r1 = box[e, 0] - 1, np.maximum(box[e, 1], box[e, 0] - 1)
            c0, c1 = box[e, 2] - 1, np.maximum(box[e, 3], box[e, 2] - 1)
            active[e] = np.where((h2[e] > 0) & (w2[e] > 0), sat[r1, c1] - sat[r0, c1] - sat[r1, c0] + sat[r0, c0], 0)
    return PatchTable(S, N, not sparse, source, image, box, pixel_center, world_center, active)


def find_neighbors(patches: List[List[ImagePatch]], target: int) -> List[int]:
    """imaged_sources.jl:232-244"""
    out = []
    for i in range(len(patches)):
        if i == target:
            continue
        for j in range(len(patches[i])):
            if boxes_overlap(patches[target][j].box, patches[i][j].box):
                out.append(i)
                break
    return out


def neighbor_map(patches) -> List[List[int]]:
    """find_neighbors for every source, via a sort-based sweep over the non-empty patches of each image instead of
    the O(S^2 N) scan (same result, ascending order like the reference's loop; empty boxes overlap nothing)."""
    S = len(patches)
    if S == 0:
        return []
    N = len(patches[0])
    per_image = [[] for _ in range(N)]
    for s in range(S):
        for n, p in row_entries(patches[s]):
            per_image[n].append((s, p.box))
    nbrs = [set() for _ in range(S)]
    for n in range(N):
        if not per_image[n]:
            continue
        src = np.array([s for s, _ in per_image[n]])
        lo_h = np.array([b[0][0] for _, b in per_image[n]])
        hi_h = np.array([b[0][1] for _, b in per_image[n]])
        lo_w = np.array([b[1][0] for _, b in per_image[n]])
        hi_w = np.array([b[1][1] for _, b in per_image[n]])
        order = np.argsort(lo_h, kind="stable")
        for a_i, a in enumerate(order):
            for b in order[a_i + 1:]:
                if lo_h[b] > hi_h[a]:
                    break
                if lo_w[a] <= hi_w[b] and lo_w[b] <= hi_w[a] and lo_h[a] <= hi_h[b]:
                    nbrs[src[a]].add(int(src[b]))
                    nbrs[src[b]].add(int(src[a]))
    return [sorted(x) for x in nbrs]


def linear_world_to_pix(wcs_jacobian, world_center, pixel_center, world):
    """wcs_utils.jl:14-18"""
    return wcs_jacobian @ (np.asarray(world, float) - world_center) + pixel_center
