"""CPU restatement of the device SLIC (focused-attention-vit_amd/csrc/slic.hip) -- TEST INFRASTRUCTURE ONLY.

Only tests/ may import this module; the product never does.  PARITY UNPINNED with respect to the reference's own
segmentation (restated from the published algorithm and scikit-image >= 0.19's parametrisation of it, whose first
step -- the per-image min-max rescale to [0, 1] -- is included): the reference calls skimage.segmentation.slic (models/sppp.py:64-66), scikit-image is not pinned by
the reference and not importable in this image, and the reference holds no fixture of a label map.  This file restates
the PUBLISHED algorithm (Achanta et al. 2012 as scikit-image parametrises it: gaussian pre-smoothing, CIELAB, seeds on
skimage.util.regular_grid, 2*step search windows, distance spatial^2/step^2 + (dLab/compactness)^2, max_num_iter
rounds, connectivity enforcement with min_size_factor 0.5) in the same integer arithmetic as the kernels, so the
device result can be checked bit for bit from the quantised features on (stage 2, stage 3), and stage 1 (float:
blur + colour conversion) to +-1 quantisation step.
"""
import math

import numpy as np


def regular_grid_2d(H, W, n_segments):
    """skimage.util.regular_grid on (1, H, W), reduced to the two image axes (see kernels.slic_grid)."""
    if H * W <= n_segments:
        return list(range(H)), list(range(W)), 1
    s = math.sqrt(H * W / float(n_segments))
    sy = sx = s
    if min(H, W) < s:
        if H <= W:
            sy, sx = float(H), W / float(n_segments)
        else:
            sx, sy = float(W), H / float(n_segments)
    ys = list(range(int(sy // 2), H, max(1, int(round(sy)))))
    xs = list(range(int(sx // 2), W, max(1, int(round(sx)))))
    return ys, xs, max(max(1, int(round(sy))), max(1, int(round(sx))))


def features(img, sigma, rescale=True):
    """img float [3,H,W] -> int16 [H*W,3]: round(16 * Lab(gaussian(rescale(img)))) (float64 here; the device works
    in fp32).  rescale: scikit-image >= 0.19's first step, `image -= image.min(); image /= (max - min)` over the whole
    array (all channels), skipped for a constant image (skimage/segmentation/slic_superpixels.py, "Rescale image to
    [0, 1] to make choice of compactness insensitive to input image scale")."""
    img = np.array(img, dtype=np.float64)
    _, H, W = img.shape
    if rescale:
        lo, hi = img.min(), img.max()
        img -= lo
        if hi != lo:
            img /= (hi - lo)
    r = int(4.0 * sigma + 0.5) if sigma > 0 else 0
    if r > 0:
        k = np.exp(-0.5 * (np.arange(-r, r + 1) ** 2) / (sigma * sigma))
        k /= k.sum()

        def refl(i, n):
            while i < 0 or i >= n:
                i = -i - 1 if i < 0 else 2 * n - 1 - i
            return i
        iy = np.array([[refl(y + d, H) for d in range(-r, r + 1)] for y in range(H)])
        ix = np.array([[refl(x + d, W) for d in range(-r, r + 1)] for x in range(W)])
        tmp = np.einsum("cykw,k->cyw", img[:, iy, :], k)
        img = np.einsum("cyxk,k->cyx", tmp[:, :, ix], k)
    lin = np.where(img > 0.04045, np.abs((img + 0.055) / 1.055) ** 2.4, img / 12.92)
    M = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]])
    xyz = np.einsum("ij,jyx->iyx", M, lin) / np.array([0.95047, 1.0, 1.08883])[:, None, None]
    f = np.where(xyz > 0.008856, np.cbrt(xyz), 7.787 * xyz + 16.0 / 116.0)
    lab = np.stack([116.0 * f[1] - 16.0, 500.0 * (f[0] - f[1]), 200.0 * (f[1] - f[2])], -1)
    return np.clip(np.rint(lab * 16.0), -8191, 8191).astype(np.int16).reshape(H * W, 3)


def _tdiv(a, b):
    """C integer division (truncation toward zero)."""
    q = abs(int(a)) // int(b)
    return q if a >= 0 else -q


def cluster(feat, H, W, ys, xs, step, coef, iters):
    """feat int [H*W,3] -> uint8 labels [H*W] (integer k-means, first minimum wins, truncated integer means)."""
    q = feat.astype(np.int64)
    yy, xx = np.divmod(np.arange(H * W, dtype=np.int64), W)
    cen = []
    for cy in ys:
        for cx in xs:
            c = q[cy * W + cx]
            cen.append([cy * 16, cx * 16, int(c[0]), int(c[1]), int(c[2])])
    cen = np.array(cen, dtype=np.int64)
    K = len(cen)
    lab = np.zeros(H * W, dtype=np.int64)
    for _ in range(iters):
        best = np.full(H * W, np.iinfo(np.int64).max, dtype=np.int64)
        new = np.full(H * W, 255, dtype=np.int64)
        for k in range(K):
            cy, cx = cen[k, 0] >> 4, cen[k, 1] >> 4
            win = (yy >= cy - 2 * step) & (yy <= cy + 2 * step) & (xx >= cx - 2 * step) & (xx <= cx + 2 * step)
            dy, dx = 16 * yy - cen[k, 0], 16 * xx - cen[k, 1]
            dq = q - cen[k, 2:5]
            d = dy * dy + dx * dx + coef * (dq * dq).sum(1)
            take = win & (d < best)
            best[take] = d[take]
            new[take] = k
        new = np.where(new == 255, lab, new)
        changed = bool((new != lab).any())
        lab = new
        for k in range(K):
            m = lab == k
            n = int(m.sum())
            if n:
                cen[k] = [_tdiv((16 * yy[m]).sum(), n), _tdiv((16 * xx[m]).sum(), n), _tdiv(q[m, 0].sum(), n),
                          _tdiv(q[m, 1].sum(), n), _tdiv(q[m, 2].sum(), n)]
        if not changed:
            break
    return lab.astype(np.uint8)


def connect(lab, H, W, min_size):
    """uint8 cluster map [H*W] -> (int64 labels [H*W], n_regions): 4-connected components in raster order of their
    first pixel; small ones take the label of an already labelled neighbour of that pixel (x+1, x-1, y+1, y-1, the
    last found wins; 0 if none), the rest are numbered consecutively."""
    lab2 = lab.reshape(H, W)
    comp = -np.ones((H, W), dtype=np.int64)
    roots, sizes = [], []
    for y0 in range(H):
        for x0 in range(W):
            if comp[y0, x0] >= 0:
                continue
            cid = len(roots)
            stack = [(y0, x0)]
            comp[y0, x0] = cid
            n = 0
            while stack:
                y, x = stack.pop()
                n += 1
                for yy, xx in ((y, x + 1), (y, x - 1), (y + 1, x), (y - 1, x)):
                    if 0 <= yy < H and 0 <= xx < W and comp[yy, xx] < 0 and lab2[yy, xx] == lab2[y0, x0]:
                        comp[yy, xx] = cid
                        stack.append((yy, xx))
            roots.append((y0, x0))
            sizes.append(n)
    final, nxt = [], 0
    for cid, ((y, x), n) in enumerate(zip(roots, sizes)):
        if n >= min_size:
            final.append(nxt)
            nxt += 1
            continue
        adj = 0
        for yy, xx in ((y, x + 1), (y, x - 1), (y + 1, x), (y - 1, x)):
            if 0 <= yy < H and 0 <= xx < W and comp[yy, xx] < cid:
                adj = final[comp[yy, xx]]
        final.append(adj)
    return np.asarray(final, dtype=np.int64)[comp].reshape(H * W), nxt
