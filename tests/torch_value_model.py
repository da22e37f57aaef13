"""An independently written, value-only, differentiable restatement of elbo() in torch (fp64).

Test infrastructure.  It restates the *model* (SURVEY.md 7.1), not the reference's derivative
code: torch.autograd then supplies gradients and Hessians, which is the reference's own
verification idea (test/test_elbo.jl:223-301: manual derivatives == ForwardDiff of the same code).
Vectorised over the target's patch pixels; sources = [target] + neighbours.
"""
import math

import numpy as np
import torch

from celeste_jl_amd import cabi
from celeste_jl_amd.synthetic import galaxy_prototypes, load_prior

DT = torch.float64


def _bw(f):
    o = 1.0 - f
    return [o ** 3 / 6, 2.0 / 3 - f * f + f ** 3 / 2, 2.0 / 3 - o * o + o ** 3 / 2, f ** 3 / 6]


def star_density(coef, xh, xw):
    ix = torch.clamp(torch.floor(xh.detach()), 1, 50).long()
    iy = torch.clamp(torch.floor(xw.detach()), 1, 50).long()
    wx, wy = _bw(xh - ix), _bw(xw - iy)
    y = 0
    for a in range(4):
        for b in range(4):
            y = y + coef[ix - 1 + a, iy - 1 + b] * wx[a] * wy[b]
    return torch.where(y < 0, 1e-3 * torch.exp(torch.clamp(y, max=0.0)), 1e-3 * (y + 1))


def galaxy_density(psf, m, dev, ratio, angle, radius, hh, ww):
    eta, nu = galaxy_prototypes()
    cp, sp = torch.cos(angle), torch.sin(angle)
    ab = ratio ** 2 - 1
    s2 = radius ** 2
    x11 = s2 * (1 + ab * sp * sp); x22 = s2 * (1 + ab * cp * cp); x12 = -s2 * cp * sp * ab
    out = 0
    for i in range(2):
        th = dev if i == 0 else 1.0 - dev
        for j in range(8 if i == 0 else 6):
            for a, xi1, xi2, t11, t12, t22 in psf:
                s11, s12, s22 = t11 + nu[i, j] * x11, t12 + nu[i, j] * x12, t22 + nu[i, j] * x22
                det = s11 * s22 - s12 * s12
                d1 = hh - (xi1 + m[0]); d2 = ww - (xi2 + m[1])
                q = (s22 * d1 * d1 - 2 * s12 * d1 * d2 + s11 * d2 * d2) / det
                out = out + th * a * eta[i, j] / (2 * math.pi * torch.sqrt(det)) * torch.exp(-0.5 * q)
    return out


def brightness(vs, i, b):
    """E[l_b | a = i], E[l_b^2 | a = i]; b 0-based"""
    r, v = vs[6 + i], vs[8 + i]
    cm, cv = vs[10 + 4 * i:14 + 4 * i], vs[18 + 4 * i:22 + 4 * i]
    l = r + v / 2; ll = 2 * r + 2 * v
    if b >= 3: l = l + cm[2] + cv[2] / 2; ll = ll + 2 * cm[2] + 2 * cv[2]
    if b >= 4: l = l + cm[3] + cv[3] / 2; ll = ll + 2 * cm[3] + 2 * cv[3]
    if b <= 1: l = l - cm[1] + cv[1] / 2; ll = ll - 2 * cm[1] + 2 * cv[1]
    if b <= 0: l = l - cm[0] + cv[0] / 2; ll = ll - 2 * cm[0] + 2 * cv[0]
    return torch.exp(l), torch.exp(ll)


def neg_kl(vs, prior):
    a = vs[26:28]
    out = -(a * (torch.log(a) - torch.log(torch.tensor(prior["is_star"], dtype=DT)))).sum()
    for i in range(2):
        k = vs[28 + 8 * i:36 + 8 * i]
        pk = torch.tensor(prior["k"][i], dtype=DT)
        out = out - a[i] * (k * (torch.log(k) - torch.log(pk))).sum()
        mu2, var2 = prior["flux_mean"][i], prior["flux_var"][i]
        r, v = vs[6 + i], vs[8 + i]
        out = out - a[i] * 0.5 * (math.log(var2) - torch.log(v) + (v + (r - mu2) ** 2) / var2 - 1)
        cm, cv = vs[10 + 4 * i:14 + 4 * i], vs[18 + 4 * i:22 + 4 * i]
        for d in range(8):
            S2 = torch.tensor(np.asarray(prior["color_cov"][i][d]).reshape(4, 4), dtype=DT)
            m2 = torch.tensor(prior["color_mean"][i][d], dtype=DT)
            inv = torch.linalg.inv(S2)
            diff = m2 - cm
            kl = (torch.diagonal(inv) * cv).sum() - 4 + diff @ inv @ diff + torch.logdet(S2) - torch.log(cv).sum()
            out = out - a[i] * k[d] * 0.5 * kl
    x = vs[5]
    out = out - 0.5 * (math.log(2 * math.pi) + math.log(prior["gal_radius_px_var"]) +
                       (x - prior["gal_radius_px_mean"]) ** 2 / prior["gal_radius_px_var"])
    return out


def make_value_fn(images, patches, neighbors, vp_all, target, include_kl=True, prior=None):
    """Returns f(theta44) -> scalar tensor for the target, neighbours frozen at vp_all."""
    prior = prior or load_prior()
    N = len(images)
    src = [target] + list(neighbors[target])
    vp_all = torch.tensor(np.asarray(vp_all), dtype=DT)
    per_image = []
    for n in range(N):
        img = images[n]
        pa = patches[target][n]
        H2, W2 = pa.active_pixel_bitmap.shape
        if H2 == 0 or W2 == 0:
            per_image.append(None)
            continue
        h0, w0 = pa.bitmap_offset
        hh = torch.arange(h0 + 1, h0 + H2 + 1, dtype=DT)[:, None].expand(H2, W2)
        ww = torch.arange(w0 + 1, w0 + W2 + 1, dtype=DT)[None, :].expand(H2, W2)
        x = torch.tensor(img.pixels[h0:h0 + H2, w0:w0 + W2].astype(np.float64))
        visit = torch.tensor(pa.active_pixel_bitmap.copy()) & ~torch.isnan(x)
        sky = torch.tensor(img.sky[h0:h0 + H2, w0:w0 + W2].astype(np.float64))
        iota32 = img.nelec_per_nmgy[h0:h0 + H2]
        iota = torch.tensor(iota32.astype(np.float64))[:, None].expand(H2, W2)
        log_iota = torch.tensor(np.log(iota32.astype(np.float64)).astype(np.float32).astype(np.float64))[:, None].expand(H2, W2)
        covers = []
        for s in src:
            p = patches[s][n]
            ph2 = hh - p.bitmap_offset[0]; pw2 = ww - p.bitmap_offset[1]
            PH2, PW2 = p.active_pixel_bitmap.shape
            inb = (ph2 >= 1) & (ph2 <= PH2) & (pw2 >= 1) & (pw2 < PW2)
            bm = torch.zeros(H2, W2, dtype=torch.bool)
            if PH2 > 0 and PW2 > 0:
                pi = torch.clamp(ph2.long() - 1, 0, PH2 - 1); pj = torch.clamp(pw2.long() - 1, 0, PW2 - 1)
                bm = torch.tensor(p.active_pixel_bitmap.copy())[pi, pj]
            covers.append(inb & bm)
        coefs = [torch.tensor(cabi.spline_prefilter(patches[s][n].stamp), dtype=DT) for s in src]
        per_image.append((img.b - 1, hh, ww, torch.nan_to_num(x), visit, sky, iota, log_iota, covers, coefs))

    def f(theta):
        total = 0
        for n in range(N):
            if per_image[n] is None:
                continue
            b, hh, ww, x, visit, sky, iota, log_iota, covers, coefs = per_image[n]
            E = sky.clone(); V = torch.zeros_like(sky)
            for q, s in enumerate(src):
                vs = theta if q == 0 else vp_all[s]
                p = patches[s][n]
                J = torch.tensor(np.asarray(p.wcs_jacobian), dtype=DT)
                m = J @ (vs[0:2] - torch.tensor(p.world_center, dtype=DT)) + torch.tensor(p.pixel_center, dtype=DT)
                f0 = star_density(coefs[q], hh - m[0] + 26, ww - m[1] + 26)
                f1 = galaxy_density(p.psf, m, vs[2], vs[3], vs[4], vs[5], hh, ww)
                Es = 0; E2s = 0
                for i, fi in enumerate((f0, f1)):
                    El, Ell = brightness(vs, i, b)
                    Es = Es + vs[26 + i] * El * fi
                    E2s = E2s + vs[26 + i] * Ell * fi * fi
                cov = covers[q].to(DT)
                E = E + cov * Es
                V = V + cov * (E2s - Es * Es)
            term = x * (log_iota + torch.log(E) - V / (2 * E * E)) - iota * E - torch.lgamma(x + 1)
            total = total + (term * visit.to(DT)).sum()
        if include_kl:
            total = total + neg_kl(theta, prior)
        return total

    return f, vp_all[target].clone()


def value_grad_hess(images, patches, neighbors, vp_all, target, include_kl=True):
    f, theta0 = make_value_fn(images, patches, neighbors, vp_all, target, include_kl)
    theta = theta0.clone().requires_grad_(True)
    v = f(theta)
    g, = torch.autograd.grad(v, theta, create_graph=True)
    H = torch.stack([torch.autograd.grad(g[i], theta, retain_graph=True)[0] for i in range(44)])
    return v.item(), g.detach().numpy(), H.detach().numpy()
