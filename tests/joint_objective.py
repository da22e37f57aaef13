"""Value of elbo_likelihood with several active sources (the score of test/test_infer.jl:9-29): every active,
non-NaN pixel of the union of the active sources' patches is visited once; every source that covers it
(strict last-column rule) contributes.  numpy/torch, value only -- test infrastructure."""
import numpy as np
import torch

import torch_value_model as tvm
from celeste_jl_amd import cabi

DT = torch.float64


def expected_planes(images, patches, vp):
    """(E - sky) per image: sum over sources of E[G_s] on their own patches (value-only add_pixel_term!)"""
    out = []
    joint_objective(images, patches, vp, set(), planes=out)
    return out


def joint_objective(images, patches, vp, active, planes=None):
    with torch.no_grad():
        return float(joint_objective_tensor(images, patches, torch.tensor(np.asarray(vp), dtype=DT), active, planes))


def joint_value_grad_hess(images, patches, vp, active, include_kl=True):
    """elbo() with the active sources `active` (in that order): value, d [Sa, 44], h [44 Sa, 44 Sa] by autograd"""
    vp = torch.tensor(np.asarray(vp), dtype=DT)
    active = list(active)
    theta0 = torch.cat([vp[a] for a in active]).clone().requires_grad_(True)

    def f(theta):
        rows = [vp[s] for s in range(vp.shape[0])]
        for k, a in enumerate(active):
            rows[a] = theta[44 * k:44 * (k + 1)]
        val = joint_objective_tensor(images, patches, torch.stack(rows), set(active))
        if include_kl:
            prior = tvm.load_prior()
            for k in range(len(active)):
                val = val + tvm.neg_kl(theta[44 * k:44 * (k + 1)], prior)
        return val
    v = f(theta0)
    g, = torch.autograd.grad(v, theta0, create_graph=True)
    H = torch.stack([torch.autograd.grad(g[i], theta0, retain_graph=True)[0] for i in range(theta0.numel())])
    return v.item(), g.detach().numpy().reshape(len(active), 44), H.detach().numpy()


def joint_objective_tensor(images, patches, vp, active, planes=None):
    total = 0.0
    S = len(patches)
    if True:
        for n, img in enumerate(images):
            H, W = img.pixels.shape
            E = torch.tensor(img.sky.astype(np.float64))
            V = torch.zeros(H, W, dtype=DT)
            visit = torch.zeros(H, W, dtype=torch.bool)
            for s in range(S):
                p = patches[s][n]
                H2, W2 = p.active_pixel_bitmap.shape
                if H2 == 0 or W2 == 0:
                    continue
                h0, w0 = p.bitmap_offset
                bm = torch.tensor(p.active_pixel_bitmap.copy())
                if s in active:
                    visit[h0:h0 + H2, w0:w0 + W2] |= bm
                if W2 < 2:
                    continue
                hh = torch.arange(h0 + 1, h0 + H2 + 1, dtype=DT)[:, None].expand(H2, W2 - 1)
                ww = torch.arange(w0 + 1, w0 + W2, dtype=DT)[None, :].expand(H2, W2 - 1)
                vs = vp[s]
                J = torch.tensor(np.asarray(p.wcs_jacobian), dtype=DT)
                m = J @ (vs[0:2] - torch.tensor(p.world_center, dtype=DT)) + torch.tensor(p.pixel_center, dtype=DT)
                coef = torch.tensor(cabi.spline_prefilter(p.stamp), dtype=DT)
                f0 = tvm.star_density(coef, hh - m[0] + 26, ww - m[1] + 26)
                f1 = tvm.galaxy_density(p.psf, m, vs[2], vs[3], vs[4], vs[5], hh, ww)
                Es = 0; E2s = 0
                for i, fi in enumerate((f0, f1)):
                    El, Ell = tvm.brightness(vs, i, img.b - 1)
                    Es = Es + vs[26 + i] * El * fi
                    E2s = E2s + vs[26 + i] * Ell * fi * fi
                cov = bm[:, :W2 - 1].to(DT)
                pad = (w0, W - (w0 + W2 - 1), h0, H - (h0 + H2))
                E = E + torch.nn.functional.pad(cov * Es, pad)
                V = V + torch.nn.functional.pad(cov * (E2s - Es * Es), pad)
            if planes is not None:
                planes.append((E - torch.tensor(img.sky.astype(np.float64))).detach().numpy())
            x = torch.tensor(img.pixels.astype(np.float64))
            visit &= ~torch.isnan(x)
            x = torch.nan_to_num(x)
            iota32 = img.nelec_per_nmgy
            iota = torch.tensor(iota32.astype(np.float64))[:, None]
            log_iota = torch.tensor(np.log(iota32.astype(np.float64)).astype(np.float32).astype(np.float64))[:, None]
            term = x * (log_iota + torch.log(E) - V / (2 * E * E)) - iota * E - torch.lgamma(x + 1)
            total = total + (term * visit.to(DT)).sum()
    return total
