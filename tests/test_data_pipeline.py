"""Device input pipeline (reference utils/data_utils.py:21-81, utils/metrics.py:152-308).

torchvision is not importable here; its Resize / RandomResizedCrop on PIL images are Pillow's
Image.resize(BILINEAR), and Pillow IS importable, so Pillow is the checker: the device resize must reproduce its
bytes EXACTLY (8-bit two-pass resampling with fixed-point coefficients), and ToTensor / Normalize follow in fp32.
Random augmentation parameters have no parity target (torch's RNG stream inside torchvision is not reproduced):
they are passed explicitly here and only their distributions are checked.
"""
import numpy as np
import pytest
import torch
from PIL import Image

DEV = "cuda"


def _pil_pipeline(img_u8, top, left, ch, cw, pad, rh, rw, oy, ox, S, flip_src, flip_out, mean, std):
    """crop (of the zero-padded image) -> flip -> PIL bilinear resize -> window -> flip -> ToTensor -> Normalize"""
    H, W, C = img_u8.shape
    padded = np.zeros((H + 2 * pad, W + 2 * pad, C), dtype=np.uint8)
    padded[pad:pad + H, pad:pad + W] = img_u8
    crop = padded[top:top + ch, left:left + cw]
    if flip_src:
        crop = crop[:, ::-1]
    res = np.asarray(Image.fromarray(np.ascontiguousarray(crop)).resize((rw, rh), Image.BILINEAR))
    win = res[oy:oy + S, ox:ox + S]
    if flip_out:
        win = win[:, ::-1]
    t = torch.from_numpy(np.ascontiguousarray(win)).permute(2, 0, 1).float().div(255)
    m, s = torch.tensor(mean).view(-1, 1, 1), torch.tensor(std).view(-1, 1, 1)
    return np.ascontiguousarray(win), (t - m) / s


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,S,case", [(32, 32, 224, "cifar_train"), (32, 32, 224, "resize"), (32, 32, 32, "cifar_train"),
                                        (180, 240, 224, "imagenet_test"), (375, 500, 224, "imagenet_test"),
                                        (300, 400, 224, "imagenet_train"), (64, 48, 96, "imagenet_train"),
                                        (500, 333, 64, "imagenet_test")])
def test_device_transform_is_bit_identical_to_pillow(favit, H, W, S, case):
    D = favit.data
    rs = np.random.RandomState(H * 7 + W + S)
    B = 5
    imgs = rs.randint(0, 256, size=(B, H, W, 3), dtype=np.uint8)
    imgs[0, :, :, :] = (np.add.outer(np.arange(H), np.arange(W))[:, :, None] * 3 % 256).astype(np.uint8)   # smooth ramp
    kind = {"cifar_train": "cifar10_train", "resize": "resize", "imagenet_test": "imagenet_test", "imagenet_train": "imagenet_train"}[case]
    tf = D.DeviceTransform(kind, S, D.IMAGENET_MEAN, D.IMAGENET_STD, seed=3)
    prm = tf.params(B, H, W)
    out, u8 = tf(torch.from_numpy(imgs).to(DEV), params=prm, want_bytes=True)
    out, u8 = out.cpu(), u8.cpu().numpy()
    for b in range(B):
        top, left, ch, cw, pad, rh, rw, oy, ox, fs, fo, _ = [int(v) for v in prm[b]]
        ref_u8, ref_f = _pil_pipeline(imgs[b], top, left, ch, cw, pad, rh, rw, oy, ox, S, fs, fo, D.IMAGENET_MEAN, D.IMAGENET_STD)
        np.testing.assert_array_equal(u8[b], ref_u8)                       # byte work: bit-exact
        assert torch.equal(out[b], ref_f), (out[b] - ref_f).abs().max()    # same fp32 operations as torch


def test_transform_parameter_distributions(favit):
    """CPU: the random parameters follow torchvision's rules (no GPU needed to draw them)."""
    D = favit.data
    tf = D.DeviceTransform("cifar10_train", 224, D.CIFAR10_MEAN, D.CIFAR10_STD, seed=1)
    p = tf.params(4000, 32, 32)
    assert p[:, 0].min() == 0 and p[:, 0].max() == 8 and p[:, 1].min() == 0 and p[:, 1].max() == 8    # RandomCrop(32, padding=4)
    assert (p[:, 2] == 32).all() and (p[:, 4] == 4).all() and (p[:, 5] == 224).all()
    assert abs(p[:, 9].mean() - 0.5) < 0.03 and (p[:, 10] == 0).all()
    tf = D.DeviceTransform("imagenet_train", 224, D.IMAGENET_MEAN, D.IMAGENET_STD, seed=2)
    p = tf.params(4000, 375, 500)
    area = p[:, 2] * p[:, 3] / (375.0 * 500.0)
    ratio = p[:, 3] / p[:, 2].astype(np.float64)
    assert area.min() >= 0.07 and area.max() <= 1.0 and 0.74 <= ratio.min() and ratio.max() <= 1.34    # scale (0.08,1), ratio (3/4,4/3)
    assert (p[:, 0] + p[:, 2] <= 375).all() and (p[:, 1] + p[:, 3] <= 500).all()
    assert abs(p[:, 10].mean() - 0.5) < 0.03
    tf = D.DeviceTransform("imagenet_test", 224, D.IMAGENET_MEAN, D.IMAGENET_STD)
    p = tf.params(1, 375, 500)[0]
    assert (p[5], p[6]) == (255, 340) and (p[7], p[8]) == (16, 58)           # Resize(255) -> CenterCrop(224)
    assert set(D.get_transforms("cifar10", 224)) == {"train", "test"}


@pytest.mark.gpu
def test_device_loader_matches_direct_transform_and_overlaps(favit):
    D = favit.data
    rs = np.random.RandomState(0)
    batches = [(rs.randint(0, 256, size=(16, 32, 32, 3), dtype=np.uint8), rs.randint(0, 10, size=16)) for _ in range(5)]
    tf = D.DeviceTransform("resize", 64, D.CIFAR10_MEAN, D.CIFAR10_STD)
    loader = D.DeviceLoader(batches, tf)
    assert len(loader) == 5
    seen = 0
    for (x, y), (hi, hl) in zip(loader, batches):
        assert x.is_cuda and tuple(x.shape) == (16, 3, 64, 64) and y.dtype == torch.int64
        ref = tf(torch.from_numpy(hi).to(DEV))
        assert torch.equal(x, ref) and torch.equal(y.cpu(), torch.from_numpy(hl))
        seen += 1
    assert seen == 5


@pytest.mark.gpu
def test_cu_masked_stream(favit):
    """streams.cu_masked_stream: the stream carries the requested CU mask (read back through hipExtStreamGetCUMask),
    kernels launched on it give the results of the default stream (device SLIC: bit-exact integers), and asking for
    every CU returns an ordinary stream."""
    import ctypes as C
    S = favit.streams
    total = torch.cuda.get_device_properties(0).multi_processor_count
    s = S.cu_masked_stream(64)
    words = (total + 31) // 32
    back = (C.c_uint32 * words)()
    hip = S._hip()
    hip.hipExtStreamGetCUMask.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    assert hip.hipExtStreamGetCUMask(C.c_void_p(s.cuda_stream), words, back) == 0
    assert sum(bin(w).count("1") for w in back) == 64
    g = torch.Generator(device=DEV).manual_seed(11)
    low = torch.rand(4, 3, 8, 8, device=DEV, generator=g)
    img = torch.nn.functional.interpolate(low, size=(64, 64), mode="bicubic").contiguous()
    want = favit.kernels.slic(img, n_segments=16, compactness=10.0)
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        got = favit.kernels.slic(img, n_segments=16, compactness=10.0)
    s.synchronize()
    assert torch.equal(got, want)
    assert not isinstance(S.cu_masked_stream(total), torch.cuda.ExternalStream)
    with pytest.raises(ValueError):
        S.cu_masked_stream(0)


@pytest.mark.gpu
@pytest.mark.parametrize("on_compute_stream", [False, True])
def test_device_loader_prefetches_label_maps_for_the_sppp_models(favit, on_compute_stream):
    """DeviceLoader(segmenter=model.segmentation): the device SLIC of batch k+1 runs on the loader's preparation stream
    under the consumer's work on batch k; when a batch is yielded its label maps are installed, i.e. the model's
    segment() call (reference: inside forward, models/sppp_mhla.py:278) returns exactly what segmenting the yielded
    images directly gives.  The consumer on the default stream (the CU-masked preparation stream then runs in turn with
    it) and on ``loader.compute_stream`` (where the two overlap)."""
    import contextlib
    D = favit.data
    rs = np.random.RandomState(3)
    low = rs.randint(0, 256, size=(4, 8, 7, 7, 3), dtype=np.uint8)                  # blocky images: clear regions
    batches = [(np.kron(low[i], np.ones((1, 8, 8, 1), dtype=np.uint8)), rs.randint(0, 10, size=8)) for i in range(4)]
    tf = D.DeviceTransform("resize", 64, D.CIFAR10_MEAN, D.CIFAR10_STD)
    seg = favit.models.sppp.SuperpixelSegmentation(num_segments=16, compactness=10.0)
    loader = D.DeviceLoader(batches, tf, segmenter=seg)
    n = 0
    torch.cuda.synchronize()
    with (torch.cuda.stream(loader.compute_stream) if on_compute_stream else contextlib.nullcontext()):
        for (x, y), (hi, hl) in zip(loader, batches):
            ref_x = tf(torch.from_numpy(hi).to(DEV))
            assert torch.equal(x, ref_x) and torch.equal(y.cpu(), torch.from_numpy(hl))
            got = seg.segment(x)                                         # what the model's forward would receive
            # (busy work on the consumer stream while the loader's next batch is being segmented on its own stream)
            _ = (x @ x.transpose(-1, -2)).sum()
            want = seg.segment_device(x)
            assert got.dtype == torch.int64 and tuple(got.shape) == (8, 64, 64)
            assert torch.equal(got, want)
            n += 1
    torch.cuda.synchronize()
    assert n == 4
    seg.set_label_maps(None)


@pytest.mark.gpu
def test_device_loader_label_maps_reach_a_captured_step(favit):
    """DeviceLoader(segmenter=...) + train.GraphedStep: the replayed kernels read the label-map tensor that was installed
    at capture, so once a step is captured the loader COPIES each batch's maps into it (update_label_maps).  The loss of
    the replayed step on (images, maps) of every yielded batch equals the eager model's on the same images and maps."""
    D = favit.data
    favit.set_compute_dtype("bf16")
    try:
        torch.manual_seed(5)
        mk = lambda: favit.models.sppp_mhla.SPPPViTMHLA(img_size=64, patch_size=8, num_classes=10, embed_dim=64, depth=2, num_heads=4,
                                                        num_superpixels=4, pooling_type="mean", window_size=3, use_mhla=True).to(DEV).train()
        m, ref = mk(), mk()
        ref.load_state_dict(m.state_dict())
        # 2 x 2 blocks of flat colour: SLIC finds exactly the four quadrants (4 superpixel tokens per image), whatever the colours
        rs = np.random.RandomState(9)
        def batch():
            q = rs.randint(30, 226, size=(8, 2, 2, 3)).astype(np.uint8)
            return np.kron(q, np.ones((1, 32, 32, 1), dtype=np.uint8)), rs.randint(0, 10, size=8)
        batches = [batch() for _ in range(4)]
        tf = D.DeviceTransform("resize", 64, (0.5, 0.5, 0.5), (0.5, 0.5, 0.5))
        for mm in (m, ref):
            mm.segmentation.compactness = 10.0
            mm.assume_num_tokens = 4
        opt = favit.train.FusedAdamW(favit.train.param_groups(m, lr=0.0), lr=0.0, weight_decay=0.0, distributed=False)
        ropt = favit.train.FusedAdamW(favit.train.param_groups(ref, lr=0.0), lr=0.0, weight_decay=0.0, distributed=False)
        x0 = tf(torch.from_numpy(batches[0][0]).to(DEV))
        y0 = torch.from_numpy(batches[0][1]).to(DEV)
        m.segmentation.set_label_maps(m.segmentation.segment_device(x0))
        step = favit.train.GraphedStep(m, opt, x0, y0)
        assert m.segmentation._captured
        installed = m.segmentation._maps
        losses = []
        for x, y in D.DeviceLoader(batches, tf, segmenter=m.segmentation):
            assert m.segmentation._maps is installed                   # updated in place, never re-bound
            got = step(x, y).item()
            ref.segmentation.set_label_maps(ref.segmentation.segment_device(x))
            want = favit.train.train_step(ref, x, y, ropt).item()
            assert abs(got - want) < 2e-3 * max(1.0, abs(want)), (got, want)
            losses.append(got)
        assert len(set(round(v, 4) for v in losses)) > 1              # the batches (and their maps) really differ
    finally:
        favit.set_compute_dtype("fp32")
        favit.functional.clear_lp_mirrors()


@pytest.mark.gpu
def test_harness_epoch_loop_and_measurements(favit, tmp_path):
    """fit / evaluate / measure_* on a tiny model and a synthetic uint8 dataset: the loss goes down, the result row
    has the reference's columns (experiments/mhla_pretrained.py:486-525), timers return positive device times."""
    D, Hh = favit.data, favit.harness
    favit.set_compute_dtype("fp32")
    torch.manual_seed(0)
    rs = np.random.RandomState(1)
    protos = rs.randint(0, 256, size=(4, 32, 32, 3))
    def make(n):
        y = rs.randint(0, 4, size=n)
        x = np.clip(protos[y] + rs.randint(-20, 21, size=(n, 32, 32, 3)), 0, 255).astype(np.uint8)
        return x, y
    train = [make(32) for _ in range(6)]
    test = [make(32) for _ in range(2)]
    tfs = D.get_transforms("cifar10", 32, seed=0)
    m = favit.models.vit_mhla.VisionTransformerMHLA(img_size=32, patch_size=4, num_classes=4, embed_dim=64, depth=2,
                                                    num_heads=4, use_mhla=True).to(DEV)
    opt = favit.train.FusedAdamW(favit.train.param_groups(m, lr=3e-3), lr=3e-3, weight_decay=0.0, distributed=False)
    res = Hh.fit(m, D.DeviceLoader(train, tfs["test"]), D.DeviceLoader(test, tfs["test"]), opt, epochs=6, log=lambda s: None)
    h = res["history"]
    assert h["train_loss"][-1] < 0.5 * h["train_loss"][0] and res["final_val_acc"] > 70.0
    ev = Hh.evaluate(m, D.DeviceLoader(test, tfs["test"]))
    assert set(ev) == {"test_loss", "test_acc", "avg_inference_time", "avg_inference_time_per_image"} and ev["avg_inference_time"] > 0
    x = torch.randn(8, 3, 32, 32, device=DEV)
    y = torch.randint(0, 4, (8,), device=DEV)
    ti = Hh.measure_inference_time(m, x, num_iterations=5, warm_up=2)
    tt = Hh.measure_training_time(m, x, y, favit.train.cross_entropy, opt, num_iterations=3)
    mem = Hh.measure_memory_usage(m, x, backward=True)
    assert ti["fps"] > 0 and tt["iterations_per_second"] > 0 and mem["gpu_memory_peak_mb"] > 0
    row = {"model": "ViT + MHLA", "avg_epoch_time": res["avg_epoch_time"], "total_training_time": res["total_training_time"],
           "final_val_acc": res["final_val_acc"], "final_val_loss": res["final_val_loss"], "test_acc": ev["test_acc"],
           "test_loss": ev["test_loss"], "avg_inference_time_per_image": ev["avg_inference_time_per_image"],
           "peak_gpu_memory_mb": res["peak_gpu_memory_mb"]}
    path = tmp_path / "results" / "exp.csv"
    Hh.save_results_csv(str(path), row)
    lines = path.read_text().strip().splitlines()
    assert lines[0].split(",") == list(row.keys()) and len(lines) == 2
    favit.functional.clear_lp_mirrors()


def test_transform_refuses_downscales_beyond_the_kernels_tap_window(favit):
    """More than 64 filter taps per output pixel (a crop downscaled > ~31x) used to be truncated silently; the
    host-side parameter check refuses it (no GPU needed: the check runs before any launch)."""
    T = favit.data.DeviceTransform
    ok = np.zeros((2, 12), dtype=np.int32)
    ok[:, 2], ok[:, 3], ok[:, 5], ok[:, 6] = 500, 375, 224, 224
    T.check_params(ok)
    bad = ok.copy()
    bad[1, 2], bad[1, 5] = 8000, 224                       # 35.7x along the vertical axis -> 74 taps
    with pytest.raises(ValueError, match="filter taps"):
        T.check_params(bad)
    tf = T("resize", 32, (0.5, 0.5, 0.5), (0.5, 0.5, 0.5))
    with pytest.raises(ValueError, match="filter taps"):
        tf.params(1, 2048, 2048)                           # 64x downscale
