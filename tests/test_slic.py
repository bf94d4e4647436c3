"""Device SLIC (reference models/sppp.py:44-74 calls skimage.segmentation.slic per image on the host).

PARITY UNPINNED with respect to scikit-image (third-party, unpinned by the reference, absent from the image; the
reference holds no label-map fixture).  The parity definition used instead, all checked here:
  * stage by stage against the CPU restatement oracle/slic_oracle.py: the float stage (blur + CIELAB) to +-1
    quantisation step, the integer stages (k-means, connectivity) BIT-EXACT from the device's own quantised features;
  * ground truth: an image made of 16 flat colour cells is segmented into exactly those cells;
  * properties at the full 224x224 size: labels 0..n-1 without gaps, every region 4-connected, no region below
    min_size, deterministic, a larger compactness gives spatially tighter regions;
  * the SPPP model consumes the device label maps without a host hop.
"""
import numpy as np
import pytest
import torch

from oracle import slic_oracle as SO

DEV = "cuda"


def _cell_image(H, W, gy, gx, seed, noise=0.0):
    rs = np.random.RandomState(seed)
    cols = rs.uniform(0.05, 0.95, size=(gy * gx, 3))
    yy, xx = np.mgrid[0:H, 0:W]
    cell = (yy * gy // H) * gx + (xx * gx // W)
    img = cols[cell].transpose(2, 0, 1).astype(np.float32)
    if noise:
        img += rs.normal(0, noise, img.shape).astype(np.float32)
    return img, cell


def _smooth_image(H, W, seed):
    """Soft colour gradients: a 6 x 6 random colour field upsampled bicubically."""
    g = torch.Generator().manual_seed(seed)
    low = torch.rand(1, 3, 6, 6, generator=g)
    return torch.nn.functional.interpolate(low, size=(H, W), mode="bicubic", align_corners=False)[0].clamp(0, 1).numpy()


def test_oracle_recovers_flat_colour_cells():
    """CPU: the restated algorithm on a 4x4 board of flat colours (seeds fall inside the cells) returns the board."""
    H = W = 64
    img, cell = _cell_image(H, W, 4, 4, seed=1)
    ys, xs, step = SO.regular_grid_2d(H, W, 16)
    assert (ys, xs, step) == ([8, 24, 40, 56], [8, 24, 40, 56], 16)
    feat = SO.features(img, sigma=0.0)
    lab = SO.cluster(feat, H, W, ys, xs, step, coef=int(round((step / 0.1) ** 2)), iters=10)
    out, n = SO.connect(lab, H, W, min_size=int(0.5 * H * W / 16))
    assert n == 16
    np.testing.assert_array_equal(out.reshape(H, W), cell)


def test_oracle_connectivity_merges_small_islands():
    H, W = 8, 10
    lab = np.zeros((H, W), dtype=np.uint8)
    lab[:, 5:] = 1
    lab[2, 2] = 1            # an island of label 1 inside region 0: must take region 0's new label
    lab[6, 8] = 0            # an island of label 0 inside region 1
    out, n = SO.connect(lab.reshape(-1), H, W, min_size=4)
    out = out.reshape(H, W)
    assert n == 2 and out[2, 2] == out[0, 0] == 0 and out[6, 8] == out[0, 9] == 1


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,nseg,sigma,compactness,seed", [(64, 64, 16, 1.0, 0.1, 0), (48, 80, 12, 1.0, 0.1, 1),
                                                              (96, 96, 9, 0.0, 1.0, 2), (224, 224, 16, 1.0, 0.1, 3),
                                                              (40, 40, 4, 2.0, 10.0, 4)])
def test_device_slic_stages_match_the_oracle(favit, H, W, nseg, sigma, compactness, seed):
    K = favit.kernels
    imgs = np.stack([_smooth_image(H, W, seed * 10 + i) for i in range(3)])
    out, feat, lab, nreg = K.slic(torch.from_numpy(imgs).to(DEV), n_segments=nseg, compactness=compactness, sigma=sigma,
                                  stages=True)
    torch.cuda.synchronize()
    ys, xs, step = SO.regular_grid_2d(H, W, nseg)
    assert K.slic_grid(H, W, nseg) == (ys, xs, step)
    coef = int(round((step / compactness) ** 2))
    min_size = int(0.5 * (H * W / float(len(ys) * len(xs))))
    feat, lab, out, nreg = feat.cpu().numpy(), lab.cpu().numpy(), out.cpu().numpy(), nreg.cpu().numpy()
    for b in range(imgs.shape[0]):
        ref_f = SO.features(imgs[b], sigma)
        diff = np.abs(feat[b, :, :3].astype(np.int64) - ref_f.astype(np.int64))
        assert diff.max() <= 1 and (diff == 0).mean() > 0.97, (diff.max(), (diff == 0).mean())       # fp32 vs fp64
        assert (feat[b, :, 3] == 0).all()
        ref_l = SO.cluster(feat[b, :, :3], H, W, ys, xs, step, coef, iters=10)                       # integer: bit-exact
        np.testing.assert_array_equal(lab[b].reshape(-1), ref_l)
        ref_o, ref_n = SO.connect(lab[b].reshape(-1), H, W, min_size)
        np.testing.assert_array_equal(out[b].reshape(-1), ref_o)
        assert int(nreg[b]) == ref_n


@pytest.mark.gpu
@pytest.mark.parametrize("rescale", [True, False])
def test_device_slic_on_mean_std_normalised_inputs(favit, rescale):
    """The reference's models hand SLIC their mean/std-NORMALISED input (models/sppp_mhla.py:278): values of about
    -2.1 .. 2.6.  scikit-image >= 0.19 rescales every image to [0, 1] by its own min / max first (rescale=True, the
    default here); < 0.19 does not (rescale=False: negative values go through sRGB -> Lab as they are).  Both forms:
    float stage within +-1 quantisation step of the fp64 restatement, integer stages bit-exact; and with the rescale
    a global affine change of the input (a * x + b, a > 0) leaves the features unchanged."""
    K = favit.kernels
    H = W = 96
    mean = np.array([0.485, 0.456, 0.406], dtype=np.float32)[:, None, None]
    std = np.array([0.229, 0.224, 0.225], dtype=np.float32)[:, None, None]
    imgs = np.stack([(_smooth_image(H, W, 40 + i) - mean) / std for i in range(3)]).astype(np.float32)
    assert imgs.min() < -1.0 and imgs.max() > 1.5
    out, feat, lab, nreg = K.slic(torch.from_numpy(imgs).to(DEV), n_segments=9, compactness=1.0, sigma=1.0, stages=True,
                                  rescale=rescale)
    ys, xs, step = SO.regular_grid_2d(H, W, 9)
    coef = int(round((step / 1.0) ** 2))
    min_size = int(0.5 * (H * W / float(len(ys) * len(xs))))
    feat, lab, out = feat.cpu().numpy(), lab.cpu().numpy(), out.cpu().numpy()
    for b in range(3):
        ref_f = SO.features(imgs[b], 1.0, rescale=rescale)
        diff = np.abs(feat[b, :, :3].astype(np.int64) - ref_f.astype(np.int64))
        assert diff.max() <= 1 and (diff == 0).mean() > 0.95, (rescale, diff.max(), (diff == 0).mean())
        ref_l = SO.cluster(feat[b, :, :3], H, W, ys, xs, step, coef, iters=10)
        np.testing.assert_array_equal(lab[b].reshape(-1), ref_l)
        ref_o, _ = SO.connect(lab[b].reshape(-1), H, W, min_size)
        np.testing.assert_array_equal(out[b].reshape(-1), ref_o)
    if rescale:
        _, feat2, _, _ = K.slic(torch.from_numpy(3.5 * imgs - 0.7).to(DEV), n_segments=9, compactness=1.0, sigma=1.0,
                                stages=True)
        d = np.abs(feat2.cpu().numpy().astype(np.int64) - feat.astype(np.int64))
        assert d.max() <= 1 and (d == 0).mean() > 0.98, (d.max(), (d == 0).mean())
        # the segmentation model uses the rescaled form by default and exposes the opt-out
        seg = favit.models.sppp.SuperpixelSegmentation(num_segments=9, compactness=1.0)
        assert seg.rescale_input is True
        a = seg.segment(torch.from_numpy(imgs).to(DEV))
        np.testing.assert_array_equal(a.cpu().numpy(), out)
    # a constant image (max == min) is not divided by zero: one flat Lab value, seed-grid Voronoi squares
    flat = torch.full((1, 3, 64, 64), -0.3, device=DEV)
    seg = K.slic(flat, n_segments=16, rescale=rescale)[0]
    assert int(seg.max()) == 15 and torch.isfinite(seg.float()).all()


@pytest.mark.gpu
def test_device_slic_rescale_of_a_normalised_image_equals_its_unit_range_original(favit):
    """What the min-max rescale is for: an image that spans [0, 1], normalised with ONE mean / std for all channels
    (the reference's generic datasets: (x - 0.5) / 0.5), segments with rescale=True exactly as its original does with
    rescale=False (the scikit-image < 0.19 form on [0, 1] data): same features up to one quantisation step, hence the
    same integer stages downstream.  (With per-channel ImageNet statistics the normalisation is not a global affine
    map and no such identity exists: that case has no parity target at all, INTEGRATION.md.)"""
    K = favit.kernels
    H = W = 96
    img = _smooth_image(H, W, 7).astype(np.float32)
    img = (img - img.min()) / (img.max() - img.min())                 # spans [0, 1] exactly
    norm = (img - 0.5) / 0.5
    _, f0, l0, _ = K.slic(torch.from_numpy(img[None]).to(DEV), n_segments=9, compactness=1.0, stages=True, rescale=False)
    _, f1, l1, _ = K.slic(torch.from_numpy(norm[None]).to(DEV), n_segments=9, compactness=1.0, stages=True, rescale=True)
    d = (f0.cpu().numpy().astype(np.int64) - f1.cpu().numpy().astype(np.int64))
    assert np.abs(d).max() <= 1 and (d == 0).mean() > 0.98
    assert (l0 == l1).float().mean().item() > 0.995                    # (a one-step feature difference can move a border pixel)


@pytest.mark.gpu
def test_device_slic_ground_truth_and_properties(favit):
    from scipy import ndimage
    K = favit.kernels
    # ground truth: 16 flat colour cells at the model's image size, no pre-smoothing (with sigma > 0 the blended
    # border pixels form a connected network of their own at compactness 0.1 -- colour-dominated SLIC)
    img, cell = _cell_image(224, 224, 4, 4, seed=5)
    seg = K.slic(torch.from_numpy(img[None]).to(DEV), sigma=0.0)[0].cpu().numpy()
    np.testing.assert_array_equal(seg, cell)
    # properties on soft random images (compactness 10: scikit-image's suggested range for CIELAB; at the
    # reference's 0.1 a smooth image fragments into components that are all below min_size)
    imgs = torch.from_numpy(np.stack([_smooth_image(224, 224, 100 + i) for i in range(4)])).to(DEV)
    assert torch.equal(K.slic(imgs), K.slic(imgs)), "deterministic at the default parameters"
    a, _, _, nreg = K.slic(imgs, compactness=10.0, stages=True)
    b = K.slic(imgs, compactness=10.0)
    assert torch.equal(a, b), "deterministic"
    min_size = int(0.5 * 224 * 224 / 16)
    for i in range(4):
        s = a[i].cpu().numpy()
        n = int(nreg[i].item())
        assert n >= 2 and sorted(np.unique(s).tolist()) == list(range(n)), "labels 0..n-1 without gaps"
        for l in range(n):
            _, ncomp = ndimage.label(s == l)
            assert ncomp == 1, f"region {l} of image {i} is not 4-connected"
            assert (s == l).sum() >= min_size

    def spread(seg):                                               # mean squared distance to the region centroid
        seg = seg.cpu().numpy()
        yy, xx = np.mgrid[0:224, 0:224]
        tot = 0.0
        for l in np.unique(seg):
            m = seg == l
            tot += ((yy[m] - yy[m].mean()) ** 2 + (xx[m] - xx[m].mean()) ** 2).sum()
        return tot / seg.size
    loose = K.slic(imgs[:1], compactness=3.0)[0]
    tight = K.slic(imgs[:1], compactness=100.0)[0]
    assert spread(tight) < spread(loose)
    # compactness -> infinity is the seed grid's Voronoi partition: 16 squares of 56 x 56 (borders within a pixel:
    # equidistant pixels go to the lower-numbered centre)
    grid = K.slic(imgs[:1], compactness=1e4)[0].cpu().numpy()
    yy, xx = np.mgrid[0:224, 0:224]
    assert (grid == (yy // 56) * 4 + xx // 56).mean() > 0.97


@pytest.mark.gpu
def test_sppp_model_segments_on_the_device(favit):
    """SPPPViTMHLA without installed label maps: segment() runs the device SLIC (no skimage, no host hop) and the
    result equals a forward with the same maps installed explicitly."""
    favit.set_compute_dtype("fp32")
    torch.manual_seed(0)
    m = favit.models.sppp_mhla.SPPPViTMHLA(img_size=224, patch_size=16, num_classes=10, embed_dim=64, depth=2, num_heads=4,
                                          num_superpixels=16, pooling_type="mean", window_size=7, use_mhla=True).to(DEV).eval()
    imgs = torch.from_numpy(np.stack([_cell_image(224, 224, 4, 4, seed=20 + i)[0] for i in range(2)])).to(DEV)
    m.segmentation.sigma = 0.0          # flat colour cells: exactly 16 superpixels per image (see the ground-truth test)
    with torch.no_grad():
        y = m(imgs)
        maps = m.segmentation.segment(imgs)
        assert maps.is_cuda and maps.dtype == torch.int64 and tuple(maps.shape) == (2, 224, 224)
        m.segmentation.set_label_maps(maps)
        y2 = m(imgs)
    assert torch.isfinite(y).all() and torch.equal(y, y2)
