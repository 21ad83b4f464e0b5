"""CPU oracle for the image front end (TEST INFRASTRUCTURE ONLY -- never imported by the product).

Restates what the reference's OpenCV build does in `minigpt4_preprocess_image` (/root/reference/minigpt4.cpp:2597-2651):

    m = PillowResize::resize(m, 224x224, INTERPOLATION_BICUBIC)       (:2620)   u8 HWC -> u8 HWC
    m.convertTo(m, CV_32FC3, 1.0f / 255.0f)                           (:2624)
    m = (m - mean) / std                                              (:2625)   CLIP mean / std (:2621-2622)
    HWC -> CHW, reported as width = 1, height = 150528, channels = 1  (:2627-2640)

PillowResize (zurutech/pillow-resize, fetched by the reference's CMake when OpenCV is enabled; not on this machine) is a C++ port of
Pillow's `ImagingResample` for 8-bit images (libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc,
ImagingResampleVertical_8bpc).  That published algorithm is restated here in numpy -- and, unlike the rest of this repo's oracle, it IS pinned:
Pillow itself is importable in the build container, `tests/golden/make_image_goldens.py` stores `Image.resize((224, 224), BICUBIC)` outputs,
and `tests/test_cpu_image.py` requires bit-equality of this restatement with them.

The float tail follows OpenCV's semantics for a CV_32F matrix and double scalars: the working type of add / divide with a scalar is float,
i.e. `(float(u8) * float(1/255) - float(mean)) / float(std)` with one rounding per operation  [UPSTREAM-RECALL: OpenCV is not on this machine].
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
IMAGE_RESIZE = 224
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def bicubic_filter(x: float) -> float:
    """Pillow Resample.c bicubic_filter (a = -0.5, support 2.0)."""
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size: int, out_size: int, support_base: float = 2.0):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the full box (in0 = 0, in1 = in_size).
    Returns (ksize, bounds[out_size][2] = (xmin, count), kk int32 [out_size][ksize])."""
    scale = float(in_size) / float(out_size)
    filterscale = max(scale, 1.0)
    support = support_base * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)       # C (int) cast: truncation toward zero
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [bicubic_filter((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _clip8(acc: np.ndarray) -> np.ndarray:
    return np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)   # arithmetic shift == C's on the int32 accumulator


def resample_horizontal(img: np.ndarray, out_w: int) -> np.ndarray:
    h, w, c = img.shape
    if out_w == w:
        return img.copy()                                            # Pillow skips a pass that does not change the size
    _, bounds, kk = precompute_coeffs(w, out_w)
    out = np.empty((h, out_w, c), np.uint8)
    src = img.astype(np.int64)
    for xx in range(out_w):
        x0, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(src[:, x0:x0 + n, :], kk[xx, :n].astype(np.int64), axes=([1], [0]))
        out[:, xx, :] = _clip8(acc)
    return out


def resample_vertical(img: np.ndarray, out_h: int) -> np.ndarray:
    h, w, c = img.shape
    if out_h == h:
        return img.copy()
    _, bounds, kk = precompute_coeffs(h, out_h)
    out = np.empty((out_h, w, c), np.uint8)
    src = img.astype(np.int64)
    for yy in range(out_h):
        y0, n = int(bounds[yy, 0]), int(bounds[yy, 1])
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(kk[yy, :n].astype(np.int64), src[y0:y0 + n], axes=([0], [0]))
        out[yy] = _clip8(acc)
    return out


def pillow_resize_bicubic(img_u8_hwc: np.ndarray, out_w: int = IMAGE_RESIZE, out_h: int = IMAGE_RESIZE) -> np.ndarray:
    """Image.resize((out_w, out_h), BICUBIC) for an 8-bit HWC image: horizontal pass, then vertical pass (Resample.c ImagingResampleInner)."""
    assert img_u8_hwc.dtype == np.uint8 and img_u8_hwc.ndim == 3
    return resample_vertical(resample_horizontal(img_u8_hwc, out_w), out_h)


def normalize_chw(img_u8_hwc: np.ndarray) -> np.ndarray:
    """convertTo(CV_32FC3, 1/255) ; (m - mean) / std ; HWC -> CHW (reference :2624-2634)."""
    x = img_u8_hwc.astype(np.float32) * np.float32(1.0 / 255.0)
    mean = np.asarray(CLIP_MEAN, np.float64).astype(np.float32)
    std = np.asarray(CLIP_STD, np.float64).astype(np.float32)
    x = (x - mean) / std
    return np.ascontiguousarray(x.transpose(2, 0, 1)).astype(np.float32)


def preprocess(img_u8_hwc: np.ndarray) -> np.ndarray:
    """u8 HWC RGB of any size -> f32 [3][224][224], the whole of minigpt4_preprocess_image."""
    return normalize_chw(pillow_resize_bicubic(img_u8_hwc))
