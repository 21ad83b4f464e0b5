#!/usr/bin/env python3
"""Generates the image-front-end fixtures of tests/golden/ -- run in the BUILD container (needs Pillow; the GPU box never runs this).

    python tests/golden/make_image_goldens.py

What pins what:
  * decoders (csrc/imageio.cpp, the cv::imread half of minigpt4_image_load_from_file): every file under tests/golden/images/ is decoded by
    Pillow (libpng / libjpeg-turbo -- the same codec libraries OpenCV wraps) -> `decoded/<name>` arrays in image_goldens.npz;
  * resize (oracle/refimage.py and the HIP kernels): Pillow's own Image.resize((224, 224), BICUBIC) of seeded inputs -> sha256 of the bytes
    (+ the full array for the reference's llama.png);
  * 16-bit PNGs follow libpng's png_set_strip_16 (high byte), which is what OpenCV asks for; Pillow widens 16-bit grey to mode I, so the expectation
    for those two files is computed here from the raw samples.
Files written by `tiny_jpeg` cover what Pillow's encoder cannot produce (4:4:0 / odd sampling-factor mixes, restart intervals with several
components); they are valid baseline JPEGs decoded by Pillow for the expectation like every other file.
"""
import hashlib
import io
import os
import struct
import sys
import zlib

import numpy as np
from PIL import Image, ImageOps

HERE = os.path.dirname(os.path.abspath(__file__))
IMG_DIR = os.path.join(HERE, "images")
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))


def smooth(h, w, seed):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    a = np.stack([128 + 100 * np.sin(x / 7.0 + y / 13.0), 128 + 90 * np.cos(x / 5.0 - y / 9.0), (x * 255.0 / max(w - 1, 1) + y * 3) % 256], -1) + rng.normal(0, 10, (h, w, 3))
    return np.clip(a, 0, 255).astype(np.uint8)


# ----------------------------------------------------------------------------------------------------------------- tiny baseline JPEG writer
ZZ = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37,
      44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]
QL = [16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64,
      81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99]
QC = [17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99] + [99] * 32
DC_L = ([0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0], list(range(12)))
DC_C = ([0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0], list(range(12)))
AC_L = ([0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d],
        [0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52,
         0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45,
         0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83,
         0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6,
         0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8,
         0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa])
AC_C = ([0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77],
        [0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33,
         0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44,
         0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a,
         0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4,
         0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7,
         0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa])


def _codes(spec):
    bits, vals = spec
    out, code, k = {}, 0, 0
    for ln in range(1, 17):
        for _ in range(bits[ln - 1]):
            out[vals[k]] = (code, ln)
            code += 1
            k += 1
        code <<= 1
    return out


def tiny_jpeg(rgb, sampling, restart=0, jfif=True, comp_ids=(1, 2, 3), adobe_transform=None):
    """Baseline JPEG of an RGB array; sampling = ((hY, vY), (hCb, vCb), (hCr, vCr)); restart = MCUs per restart interval (0: none).
    adobe_transform=0 stores the RGB planes untransformed behind an Adobe APP14 marker."""
    h, w, _ = rgb.shape
    f = rgb.astype(np.float64)
    if adobe_transform == 0:
        planes = [f[..., 0], f[..., 1], f[..., 2]]
    else:
        planes = [0.299 * f[..., 0] + 0.587 * f[..., 1] + 0.114 * f[..., 2], 128 - 0.168736 * f[..., 0] - 0.331264 * f[..., 1] + 0.5 * f[..., 2],
                  128 + 0.5 * f[..., 0] - 0.418688 * f[..., 1] - 0.081312 * f[..., 2]]
    hmax, vmax = max(s[0] for s in sampling), max(s[1] for s in sampling)
    mx, my = -(-w // (8 * hmax)), -(-h // (8 * vmax))
    n = np.arange(8)
    C = np.sqrt(2.0 / 8) * np.cos((2 * n[None, :] + 1) * n[:, None] * np.pi / 16)
    C[0, :] = np.sqrt(1.0 / 8)
    qts = [np.array(QL, np.float64).reshape(8, 8), np.array(QC, np.float64).reshape(8, 8)]
    comp_blocks = []
    for ci, (hs, vs) in enumerate(sampling):
        fx, fy = hmax // hs, vmax // vs
        p = np.pad(planes[ci], ((0, my * 8 * vmax - h), (0, mx * 8 * hmax - w)), mode="edge")
        p = p.reshape(p.shape[0] // fy, fy, p.shape[1] // fx, fx).mean(axis=(1, 3)) - 128.0
        q = qts[0 if ci == 0 else 1]
        bh, bw = p.shape[0] // 8, p.shape[1] // 8
        blocks = np.zeros((bh, bw, 64), np.int64)
        for by in range(bh):
            for bx in range(bw):
                d = C @ p[by * 8:by * 8 + 8, bx * 8:bx * 8 + 8] @ C.T
                blocks[by, bx] = np.rint(d / q).astype(np.int64).reshape(64)[ZZ]
        comp_blocks.append(blocks)
    dcs, acs = [_codes(DC_L), _codes(DC_C)], [_codes(AC_L), _codes(AC_C)]
    out = bytearray()
    acc, nb = 0, 0

    def put(code, ln):
        nonlocal acc, nb
        acc = (acc << ln) | (code & ((1 << ln) - 1))
        nb += ln
        while nb >= 8:
            b = (acc >> (nb - 8)) & 0xFF
            out.append(b)
            if b == 0xFF:
                out.append(0)
            nb -= 8

    def flush():
        nonlocal acc, nb
        if nb:
            put((1 << (8 - nb)) - 1, 8 - nb)
        acc, nb = 0, 0

    def mag(v):
        a = abs(int(v))
        s = a.bit_length()
        return s, (v if v >= 0 else v + (1 << s) - 1)

    pred = [0, 0, 0]
    count = 0
    rst = 0
    for y in range(my):
        for x in range(mx):
            if restart and count and count % restart == 0:
                flush()
                out.extend([0xFF, 0xD0 + (rst & 7)])
                rst += 1
                pred = [0, 0, 0]
            for ci, (hs, vs) in enumerate(sampling):
                t = 0 if ci == 0 else 1
                for by in range(vs):
                    for bx in range(hs):
                        blk = comp_blocks[ci][y * vs + by, x * hs + bx]
                        s, bits = mag(blk[0] - pred[ci])
                        pred[ci] = int(blk[0])
                        put(*dcs[t][s])
                        if s:
                            put(bits, s)
                        run = 0
                        for k in range(1, 64):
                            v = int(blk[k])
                            if v == 0:
                                run += 1
                                continue
                            while run > 15:
                                put(*acs[t][0xF0])
                                run -= 16
                            s, bits = mag(v)
                            put(*acs[t][(run << 4) | s])
                            put(bits, s)
                            run = 0
                        if run:
                            put(*acs[t][0x00])
            count += 1
    flush()
    scan = bytes(out)

    def seg(marker, payload):
        return bytes([0xFF, marker]) + struct.pack(">H", len(payload) + 2) + payload
    f = bytearray(b"\xFF\xD8")
    if jfif:
        f += seg(0xE0, b"JFIF\0\x01\x01\x00\x00\x01\x00\x01\x00\x00")
    if adobe_transform is not None:
        f += seg(0xEE, b"Adobe\0\x64\0\0\0\0" + bytes([adobe_transform]))
    for i, q in enumerate((QL, QC)):
        f += seg(0xDB, bytes([i]) + bytes(int(q[z]) for z in ZZ))
    f += seg(0xC0, struct.pack(">BHHB", 8, h, w, 3) + b"".join(bytes([comp_ids[i], (sampling[i][0] << 4) | sampling[i][1], 0 if i == 0 else 1]) for i in range(3)))
    for cls, idx, spec in ((0, 0, DC_L), (1, 0, AC_L), (0, 1, DC_C), (1, 1, AC_C)):
        f += seg(0xC4, bytes([(cls << 4) | idx]) + bytes(spec[0]) + bytes(spec[1]))
    if restart:
        f += seg(0xDD, struct.pack(">H", restart))
    f += seg(0xDA, bytes([3]) + b"".join(bytes([comp_ids[i], 0x00 if i == 0 else 0x11]) for i in range(3)) + bytes([0, 63, 0]))
    f += scan + b"\xFF\xD9"
    return bytes(f)


# ----------------------------------------------------------------------------------------------------------------- raw PNG writer (16-bit, interlace)
def raw_png(samples, color_type, bit_depth, palette=None):
    """samples: [h][w][channels] integer array of raw sample values; no interlace (Pillow writes interlaced files itself)."""
    h, w, ch = samples.shape
    rows = bytearray()
    for y in range(h):
        rows.append(0)
        if bit_depth == 16:
            rows += samples[y].astype(">u2").tobytes()
        elif bit_depth == 8:
            rows += samples[y].astype(np.uint8).tobytes()
        else:
            per = 8 // bit_depth
            vals = samples[y].reshape(-1)
            vals = np.concatenate([vals, np.zeros((-len(vals)) % per, vals.dtype)])
            packed = np.zeros(len(vals) // per, np.uint8)
            for i in range(per):
                packed |= (vals[i::per].astype(np.uint8) << (bit_depth * (per - 1 - i))).astype(np.uint8)
            rows += packed.tobytes()

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, bit_depth, color_type, 0, 0, 0))
    if palette is not None:
        out += chunk(b"PLTE", bytes(palette))
    co = zlib.compressobj(9)
    z = co.compress(bytes(rows)) + co.flush()
    for i in range(0, len(z), 4000):      # several IDAT chunks
        out += chunk(b"IDAT", z[i:i + 4000])
    return out + chunk(b"IEND", b"")


def main():
    os.makedirs(IMG_DIR, exist_ok=True)
    files = {}          # name -> bytes
    expect = {}         # name -> decoded RGB array (what cv::imread + BGR2RGB yields)

    def add(name, data, want=None):
        files[name] = data
        if want is None:
            want = np.asarray(ImageOps.exif_transpose(Image.open(io.BytesIO(data))).convert("RGB"))
        expect[name] = np.ascontiguousarray(want, dtype=np.uint8)

    def pil_bytes(im, fmt, **kw):
        b = io.BytesIO()
        im.save(b, fmt, **kw)
        return b.getvalue()

    # PNG: every colour type / bit depth Pillow can write, interlaced + non-interlaced; stored/fixed/dynamic deflate blocks
    src = smooth(37, 53, 1)
    rgb = Image.fromarray(src, "RGB")
    alpha = Image.fromarray((src[..., 0] // 2 + 60).astype(np.uint8))
    variants = {"rgb8": rgb, "rgba8": rgb.copy(), "l8": rgb.convert("L"), "la8": rgb.convert("L"), "p8": rgb.quantize(200), "p4": rgb.quantize(13), "p2": rgb.quantize(4),
                "p1": rgb.quantize(2), "l1": rgb.convert("1")}
    variants["rgba8"].putalpha(alpha)
    la = variants["la8"].convert("LA")
    la.putalpha(alpha)
    variants["la8"] = la
    for name, im in variants.items():
        add(f"png_{name}.png", pil_bytes(im, "PNG"))
    add("png_rgb8_level0.png", pil_bytes(rgb, "PNG", compress_level=0))       # stored blocks
    add("png_rgb8_level1.png", pil_bytes(rgb, "PNG", compress_level=1))
    add("png_1x1.png", pil_bytes(Image.fromarray(smooth(1, 1, 2), "RGB"), "PNG"))
    add("png_3x300.png", pil_bytes(Image.fromarray(smooth(3, 300, 3), "RGB"), "PNG"))
    add("png_noise.png", pil_bytes(Image.fromarray(np.random.default_rng(5).integers(0, 256, (40, 40, 3), dtype=np.uint8), "RGB"), "PNG"))
    # grey 2/4 bit, 16-bit grey / rgb / rgba, tRNS-less palette: written raw (Pillow cannot), decoded by Pillow where it keeps 8 bits
    g = smooth(19, 23, 6)[..., :1]
    add("png_l2.png", raw_png(g >> 6, 0, 2))
    add("png_l4.png", raw_png(g >> 4, 0, 4))
    s16 = (smooth(11, 13, 7).astype(np.uint32) * 257 + np.random.default_rng(8).integers(0, 200, (11, 13, 3))).clip(0, 65535)
    add("png_rgb16.png", raw_png(s16, 2, 16), want=(s16 >> 8).astype(np.uint8))                       # libpng strip_16: high byte
    add("png_l16.png", raw_png(s16[..., :1], 0, 16), want=np.repeat((s16[..., :1] >> 8).astype(np.uint8), 3, axis=2))
    rgba16 = np.concatenate([s16, np.full((11, 13, 1), 30000)], axis=2)
    add("png_rgba16.png", raw_png(rgba16, 6, 16), want=(s16 >> 8).astype(np.uint8))
    # Adam7: Pillow cannot write interlaced PNGs -> a pass-by-pass writer on top of raw_png's pieces
    add("png_adam7_rgb8.png", adam7_png(smooth(21, 17, 9), 2, 8))
    add("png_adam7_l1.png", adam7_png((smooth(10, 9, 10)[..., :1] >> 7), 0, 1))
    add("png_adam7_1x1.png", adam7_png(smooth(1, 1, 11), 2, 8))

    # JPEG via Pillow: subsampling x progressive x sizes (incl. widths where libjpeg switches off fancy upsampling), grey, restart markers, EXIF orientations
    for ss, tag in ((0, "444"), (1, "422"), (2, "420"), ("4:1:1", "411")):
        for (h, w) in ((33, 47), (17, 2), (9, 5), (64, 64)):
            for prog in (False, True):
                add(f"jpg_{tag}_{h}x{w}_{'prog' if prog else 'base'}.jpg", pil_bytes(Image.fromarray(smooth(h, w, 20 + h + w)), "JPEG", quality=85, subsampling=ss, progressive=prog))
    add("jpg_grey.jpg", pil_bytes(Image.fromarray(smooth(30, 41, 30)[..., 0], "L"), "JPEG", quality=75))
    add("jpg_grey_prog.jpg", pil_bytes(Image.fromarray(smooth(30, 41, 31)[..., 0], "L"), "JPEG", quality=75, progressive=True))
    add("jpg_q10.jpg", pil_bytes(Image.fromarray(smooth(40, 40, 32)), "JPEG", quality=10))
    add("jpg_q100.jpg", pil_bytes(Image.fromarray(np.random.default_rng(33).integers(0, 256, (24, 24, 3), dtype=np.uint8)), "JPEG", quality=100, subsampling=2))
    add("jpg_optimized.jpg", pil_bytes(Image.fromarray(smooth(35, 35, 34)), "JPEG", quality=80, optimize=True))
    try:
        add("jpg_restart_rows.jpg", pil_bytes(Image.fromarray(smooth(50, 70, 35)), "JPEG", quality=80, restart_marker_rows=1))
        add("jpg_restart_blocks_prog.jpg", pil_bytes(Image.fromarray(smooth(50, 70, 36)), "JPEG", quality=80, restart_marker_blocks=3, progressive=True))
    except TypeError:
        pass
    for o in range(2, 9):
        ex = Image.Exif()
        ex[0x0112] = o
        add(f"jpg_exif{o}.jpg", pil_bytes(Image.fromarray(smooth(20, 31, 40 + o)), "JPEG", quality=90, exif=ex))
    # JPEG via tiny_jpeg: sampling-factor mixes Pillow's encoder cannot produce
    tj = smooth(45, 38, 50)
    add("jpg_tiny_440.jpg", tiny_jpeg(tj, ((1, 2), (1, 1), (1, 1))))                      # h1v2 fancy upsampling
    add("jpg_tiny_440_narrow.jpg", tiny_jpeg(smooth(21, 2, 51), ((1, 2), (1, 1), (1, 1))))
    add("jpg_tiny_420_rst.jpg", tiny_jpeg(tj, ((2, 2), (1, 1), (1, 1)), restart=2))
    add("jpg_tiny_mixed.jpg", tiny_jpeg(tj, ((2, 2), (2, 1), (1, 2))))                    # Cb: h1v2, Cr: h2v1
    add("jpg_tiny_41.jpg", tiny_jpeg(tj, ((4, 1), (1, 1), (2, 1))))                       # 4:1 box + h2v1
    add("jpg_tiny_14.jpg", tiny_jpeg(tj, ((1, 4), (1, 1), (1, 2))))
    add("jpg_tiny_nojfif_ids.jpg", tiny_jpeg(tj, ((1, 1), (1, 1), (1, 1)), jfif=False))
    add("jpg_tiny_adobe_rgb.jpg", tiny_jpeg(tj, ((1, 1), (1, 1), (1, 1)), jfif=False, adobe_transform=0))
    add("jpg_tiny_rgb_ids.jpg", tiny_jpeg_rgb_ids(tj))
    # BMP / PNM
    add("bmp_24.bmp", pil_bytes(Image.fromarray(smooth(13, 11, 60)), "BMP"))
    add("bmp_8.bmp", pil_bytes(Image.fromarray(smooth(13, 11, 61)).quantize(50), "BMP"))
    add("ppm_p6.ppm", pil_bytes(Image.fromarray(smooth(9, 14, 62)), "PPM"))
    add("pgm_p5.pgm", pil_bytes(Image.fromarray(smooth(9, 14, 63)[..., 0], "L"), "PPM"))

    for name, data in files.items():
        with open(os.path.join(IMG_DIR, name), "wb") as fh:
            fh.write(data)

    # resize goldens: Pillow itself
    import refimage as R
    resize = {}
    cases = {"up_37x53": smooth(37, 53, 70), "down_500x300": smooth(300, 500, 71), "down_big_1000x777": smooth(777, 1000, 72), "same_w_224x500": smooth(500, 224, 73),
             "same_h_448x224": smooth(224, 448, 74), "same_224": smooth(224, 224, 75), "tiny_1x1": smooth(1, 1, 76), "tiny_3x5": smooth(5, 3, 77), "odd_225x223": smooth(223, 225, 78),
             "noise_640x480": np.random.default_rng(79).integers(0, 256, (480, 640, 3), dtype=np.uint8)}
    for name, a in cases.items():
        want = np.asarray(Image.fromarray(a, "RGB").resize((224, 224), Image.BICUBIC))
        assert np.array_equal(want, R.pillow_resize_bicubic(a)), name      # the oracle must already agree when the goldens are made
        resize[name] = hashlib.sha256(want.tobytes()).hexdigest()
    llama = np.asarray(Image.open("/root/reference/minigpt4/images/llama.png").convert("RGB"))
    llama_small = np.asarray(Image.fromarray(llama).resize((187, 140), Image.BICUBIC))   # a real photo at fixture size (the repo does not carry the reference's PNG)
    add("png_llama_small.png", pil_bytes(Image.fromarray(llama_small), "PNG", optimize=True))
    with open(os.path.join(IMG_DIR, "png_llama_small.png"), "wb") as fh:
        fh.write(files["png_llama_small.png"])
    llama_224 = np.asarray(Image.fromarray(llama_small).resize((224, 224), Image.BICUBIC))
    np.savez_compressed(os.path.join(HERE, "image_goldens.npz"), llama_224=llama_224,
                        resize_names=np.array(list(resize.keys())), resize_sha256=np.array(list(resize.values())),
                        resize_seeds=np.array([70, 71, 72, 73, 74, 75, 76, 77, 78, 79]), resize_shapes=np.array([cases[k].shape[:2] for k in resize]),
                        **{"decoded/" + k: v for k, v in expect.items()})
    print(f"wrote {len(files)} image files ({sum(len(v) for v in files.values()) / 1024:.0f} KiB) and image_goldens.npz "
          f"({os.path.getsize(os.path.join(HERE, 'image_goldens.npz')) / 1024:.0f} KiB)")


def adam7_png(samples, color_type, bit_depth):
    """Adam7-interlaced PNG (raw sample values [h][w][ch])."""
    h, w, ch = samples.shape
    XS, YS, XD, YD = (0, 4, 0, 2, 0, 1, 0), (0, 0, 4, 0, 2, 0, 1), (8, 8, 4, 4, 2, 2, 1), (8, 8, 8, 4, 4, 2, 2)
    rows = bytearray()
    for p in range(7):
        sub = samples[YS[p]::YD[p], XS[p]::XD[p]]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        for y in range(sub.shape[0]):
            rows.append(0)
            if bit_depth == 8:
                rows += sub[y].astype(np.uint8).tobytes()
            else:
                per = 8 // bit_depth
                vals = sub[y].reshape(-1)
                vals = np.concatenate([vals, np.zeros((-len(vals)) % per, vals.dtype)])
                packed = np.zeros(len(vals) // per, np.uint8)
                for i in range(per):
                    packed |= (vals[i::per].astype(np.uint8) << (bit_depth * (per - 1 - i))).astype(np.uint8)
                rows += packed.tobytes()

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, bit_depth, color_type, 0, 0, 1)) + chunk(b"IDAT", zlib.compress(bytes(rows), 6)) +
            chunk(b"IEND", b""))


def tiny_jpeg_rgb_ids(rgb):
    """RGB planes stored untransformed, identified only by component ids 'R','G','B' (no JFIF, no Adobe marker)."""
    data = bytearray(tiny_jpeg(rgb, ((1, 1), (1, 1), (1, 1)), jfif=False, comp_ids=(ord("R"), ord("G"), ord("B")), adobe_transform=0))
    i = data.find(b"\xFF\xEE")
    ln = struct.unpack(">H", data[i + 2:i + 4])[0]
    del data[i:i + 2 + ln]
    return bytes(data)


if __name__ == "__main__":
    main()
