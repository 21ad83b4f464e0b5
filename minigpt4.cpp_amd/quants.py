"""numpy block (de)quantisers for the ggml tensor types carried by the two model files.

Host-side tooling: used by the synthetic-model generator (`modelgen.py`), by `bench.py`
to build the BASELINE-shaped weight files, and by the tests as an independent float64
view of a weight file.  Nothing here runs on the product's compute path.

Only the *dequantisation* side is normative (it has to agree with ggml's block layouts,
SURVEY.md section 2.5; the reference reaches them through `ggml_mul_mat`,
/root/reference/minigpt4.cpp:1022, and `llama_eval`, :2373).  The quantisers are simple
min/max variants -- any encoder that emits valid blocks is acceptable for synthetic
weights.

Type ids: `GGML_*` is ggml's own numbering (used inside the GGJT LLM file);
`MiniGPT4DataType` numbering (vision file + C API, /root/reference/minigpt4.h:30-48)
is mapped in `MG4_TO_GGML`.
"""
from __future__ import annotations

import numpy as np

# ggml_type numbering (LLM file)
GGML_F32, GGML_F16, GGML_Q4_0, GGML_Q4_1 = 0, 1, 2, 3
GGML_Q5_0, GGML_Q5_1, GGML_Q8_0, GGML_Q8_1 = 6, 7, 8, 9
GGML_Q2_K, GGML_Q3_K, GGML_Q4_K, GGML_Q5_K, GGML_Q6_K, GGML_Q8_K = 10, 11, 12, 13, 14, 15
GGML_I32 = 18  # not stored in LLM files; used internally for the vision file's I32/L64 tensors
GGML_I64 = 19

# MiniGPT4DataType numbering (vision file / C API)
MG4_F16, MG4_F32, MG4_I32, MG4_L64 = 0, 1, 2, 3
MG4_Q4_0, MG4_Q4_1, MG4_Q5_0, MG4_Q5_1, MG4_Q8_0, MG4_Q8_1 = 4, 5, 6, 7, 8, 9
MG4_Q2_K, MG4_Q3_K, MG4_Q4_K, MG4_Q5_K, MG4_Q6_K, MG4_Q8_K = 10, 11, 12, 13, 14, 15

MG4_TO_GGML = {
    MG4_F16: GGML_F16, MG4_F32: GGML_F32, MG4_I32: GGML_I32, MG4_L64: GGML_I64,
    MG4_Q4_0: GGML_Q4_0, MG4_Q4_1: GGML_Q4_1, MG4_Q5_0: GGML_Q5_0, MG4_Q5_1: GGML_Q5_1,
    MG4_Q8_0: GGML_Q8_0, MG4_Q8_1: GGML_Q8_1, MG4_Q2_K: GGML_Q2_K, MG4_Q3_K: GGML_Q3_K,
    MG4_Q4_K: GGML_Q4_K, MG4_Q5_K: GGML_Q5_K, MG4_Q6_K: GGML_Q6_K, MG4_Q8_K: GGML_Q8_K,
}
GGML_TO_MG4 = {v: k for k, v in MG4_TO_GGML.items()}

TYPE_NAMES = {
    GGML_F32: "f32", GGML_F16: "f16", GGML_Q4_0: "q4_0", GGML_Q4_1: "q4_1", GGML_Q5_0: "q5_0",
    GGML_Q5_1: "q5_1", GGML_Q8_0: "q8_0", GGML_Q4_K: "q4_k", GGML_Q5_K: "q5_k", GGML_Q6_K: "q6_k",
    GGML_Q2_K: "q2_k", GGML_Q3_K: "q3_k", GGML_I32: "i32", GGML_I64: "i64",
}
NAME_TO_TYPE = {v: k for k, v in TYPE_NAMES.items()}

# (elements per block, bytes per block)
BLOCK = {
    GGML_F32: (1, 4), GGML_F16: (1, 2), GGML_I32: (1, 4), GGML_I64: (1, 8),
    GGML_Q4_0: (32, 18), GGML_Q4_1: (32, 20), GGML_Q5_0: (32, 22), GGML_Q5_1: (32, 24),
    GGML_Q8_0: (32, 34), GGML_Q8_1: (32, 40),
    GGML_Q2_K: (256, 84), GGML_Q3_K: (256, 110), GGML_Q4_K: (256, 144), GGML_Q5_K: (256, 176),
    GGML_Q6_K: (256, 210), GGML_Q8_K: (256, 292),
}


def nbytes(gtype: int, n_elements: int) -> int:
    e, b = BLOCK[gtype]
    assert n_elements % e == 0, (gtype, n_elements)
    return n_elements // e * b


def _f16(x) -> np.ndarray:
    return np.asarray(x, dtype=np.float32).astype(np.float16)


# --------------------------------------------------------------------------- 32-wide blocks
def quantize_q4_0(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, np.float32).reshape(-1, 32)
    idx = np.argmax(np.abs(x), axis=1)
    mx = x[np.arange(x.shape[0]), idx]
    d = (mx / -8.0).astype(np.float32)
    dh = _f16(d)
    inv = np.where(d != 0, 1.0 / np.where(d != 0, d, 1), 0).astype(np.float32)
    q = np.minimum(15, (x * inv[:, None] + 8.5).astype(np.int32)).clip(0, 15).astype(np.uint8)
    out = np.zeros((x.shape[0], 18), np.uint8)
    out[:, 0:2] = dh.view(np.uint8).reshape(-1, 2)
    out[:, 2:] = q[:, :16] | (q[:, 16:] << 4)
    return out.reshape(-1)


def dequantize_q4_0(buf: np.ndarray, n: int) -> np.ndarray:
    b = np.frombuffer(buf, np.uint8, n // 32 * 18).reshape(-1, 18)
    d = b[:, 0:2].copy().view(np.float16).astype(np.float64)
    qs = b[:, 2:]
    q = np.concatenate([qs & 15, qs >> 4], axis=1).astype(np.float64) - 8.0
    return (q * d).reshape(-1)


def quantize_q4_1(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, np.float32).reshape(-1, 32)
    mn, mx = x.min(1), x.max(1)
    d = ((mx - mn) / 15.0).astype(np.float32)
    inv = np.where(d != 0, 1.0 / np.where(d != 0, d, 1), 0).astype(np.float32)
    q = np.minimum(15, ((x - mn[:, None]) * inv[:, None] + 0.5).astype(np.int32)).clip(0, 15).astype(np.uint8)
    out = np.zeros((x.shape[0], 20), np.uint8)
    out[:, 0:2] = _f16(d).view(np.uint8).reshape(-1, 2)
    out[:, 2:4] = _f16(mn).view(np.uint8).reshape(-1, 2)
    out[:, 4:] = q[:, :16] | (q[:, 16:] << 4)
    return out.reshape(-1)


def dequantize_q4_1(buf: np.ndarray, n: int) -> np.ndarray:
    b = np.frombuffer(buf, np.uint8, n // 32 * 20).reshape(-1, 20)
    d = b[:, 0:2].copy().view(np.float16).astype(np.float64)
    m = b[:, 2:4].copy().view(np.float16).astype(np.float64)
    qs = b[:, 4:]
    q = np.concatenate([qs & 15, qs >> 4], axis=1).astype(np.float64)
    return (q * d + m).reshape(-1)


def _pack_qh32(hi: np.ndarray) -> np.ndarray:
    """hi: [nb, 32] of 0/1 -> [nb, 4] bytes, bit j of the little-endian u32 = element j."""
    w = (hi.astype(np.uint32) << np.arange(32, dtype=np.uint32)[None, :]).sum(1).astype(np.uint32)
    return w.view(np.uint8).reshape(-1, 4)


def _unpack_qh32(b4: np.ndarray) -> np.ndarray:
    w = b4.copy().view(np.uint32).reshape(-1)
    return ((w[:, None] >> np.arange(32, dtype=np.uint32)[None, :]) & 1).astype(np.uint8)


def quantize_q5_0(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, np.float32).reshape(-1, 32)
    idx = np.argmax(np.abs(x), axis=1)
    mx = x[np.arange(x.shape[0]), idx]
    d = (mx / -16.0).astype(np.float32)
    inv = np.where(d != 0, 1.0 / np.where(d != 0, d, 1), 0).astype(np.float32)
    q = np.minimum(31, (x * inv[:, None] + 16.5).astype(np.int32)).clip(0, 31).astype(np.uint8)
    out = np.zeros((x.shape[0], 22), np.uint8)
    out[:, 0:2] = _f16(d).view(np.uint8).reshape(-1, 2)
    out[:, 2:6] = _pack_qh32(q >> 4)
    lo = q & 15
    out[:, 6:] = lo[:, :16] | (lo[:, 16:] << 4)
    return out.reshape(-1)


def dequantize_q5_0(buf: np.ndarray, n: int) -> np.ndarray:
    b = np.frombuffer(buf, np.uint8, n // 32 * 22).reshape(-1, 22)
    d = b[:, 0:2].copy().view(np.float16).astype(np.float64)
    hi = _unpack_qh32(b[:, 2:6])
    qs = b[:, 6:]
    lo = np.concatenate([qs & 15, qs >> 4], axis=1)
    q = (lo | (hi << 4)).astype(np.float64) - 16.0
    return (q * d).reshape(-1)


def quantize_q5_1(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, np.float32).reshape(-1, 32)
    mn, mx = x.min(1), x.max(1)
    d = ((mx - mn) / 31.0).astype(np.float32)
    inv = np.where(d != 0, 1.0 / np.where(d != 0, d, 1), 0).astype(np.float32)
    q = np.minimum(31, ((x - mn[:, None]) * inv[:, None] + 0.5).astype(np.int32)).clip(0, 31).astype(np.uint8)
    out = np.zeros((x.shape[0], 24), np.uint8)
    out[:, 0:2] = _f16(d).view(np.uint8).reshape(-1, 2)
    out[:, 2:4] = _f16(mn).view(np.uint8).reshape(-1, 2)
    out[:, 4:8] = _pack_qh32(q >> 4)
    lo = q & 15
    out[:, 8:] = lo[:, :16] | (lo[:, 16:] << 4)
    return out.reshape(-1)


def dequantize_q5_1(buf: np.ndarray, n: int) -> np.ndarray:
    b = np.frombuffer(buf, np.uint8, n // 32 * 24).reshape(-1, 24)
    d = b[:, 0:2].copy().view(np.float16).astype(np.float64)
    m = b[:, 2:4].copy().view(np.float16).astype(np.float64)
    hi = _unpack_qh32(b[:, 4:8])
    qs = b[:, 8:]
    lo = np.concatenate([qs & 15, qs >> 4], axis=1)
    q = (lo | (hi << 4)).astype(np.float64)
    return (q * d + m).reshape(-1)


def quantize_q8_0(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, np.float32).reshape(-1, 32)
    amax = np.abs(x).max(1)
    d = (amax / 127.0).astype(np.float32)
    inv = np.where(d != 0, 1.0 / np.where(d != 0, d, 1), 0).astype(np.float32)
    q = np.rint(x * inv[:, None]).clip(-127, 127).astype(np.int8)
    out = np.zeros((x.shape[0], 34), np.uint8)
    out[:, 0:2] = _f16(d).view(np.uint8).reshape(-1, 2)
    out[:, 2:] = q.view(np.uint8)
    return out.reshape(-1)


def dequantize_q8_0(buf: np.ndarray, n: int) -> np.ndarray:
    b = np.frombuffer(buf, np.uint8, n // 32 * 34).reshape(-1, 34)
    d = b[:, 0:2].copy().view(np.float16).astype(np.float64)
    q = b[:, 2:].copy().view(np.int8).astype(np.float64)
    return (q * d).reshape(-1)


# --------------------------------------------------------------------------- k-quants (256-wide)
def _pack_scales_k4(sc: np.ndarray, mn: np.ndarray) -> np.ndarray:
    """sc, mn: [nb, 8] of 6-bit values -> [nb, 12] bytes (inverse of get_scale_min_k4)."""
    sc = sc.astype(np.uint8)
    mn = mn.astype(np.uint8)
    out = np.zeros((sc.shape[0], 12), np.uint8)
    for j in range(4):
        out[:, j] = (sc[:, j] & 63) | ((sc[:, j + 4] >> 4) << 6)
        out[:, j + 4] = (mn[:, j] & 63) | ((mn[:, j + 4] >> 4) << 6)
        out[:, j + 8] = (sc[:, j + 4] & 15) | ((mn[:, j + 4] & 15) << 4)
    return out


def _unpack_scales_k4(s12: np.ndarray):
    sc = np.zeros((s12.shape[0], 8), np.uint8)
    mn = np.zeros((s12.shape[0], 8), np.uint8)
    for j in range(4):
        sc[:, j] = s12[:, j] & 63
        mn[:, j] = s12[:, j + 4] & 63
    for j in range(4, 8):
        sc[:, j] = (s12[:, j + 4] & 15) | ((s12[:, j - 4] >> 6) << 4)
        mn[:, j] = (s12[:, j + 4] >> 4) | ((s12[:, j] >> 6) << 4)
    return sc, mn


def _kquant_scales(x: np.ndarray, qmax: int):
    """x: [nb, 8, 32] -> (d f16, dmin f16, sc u8[nb,8], mn u8[nb,8], q u8[nb,8,32])."""
    mn_ = np.minimum(x.min(2), 0.0)
    mx_ = np.maximum(x.max(2), mn_)
    scale = (mx_ - mn_) / float(qmax)
    d = (scale.max(1) / 63.0).astype(np.float32)
    dmin = ((-mn_).max(1) / 63.0).astype(np.float32)
    dh, dminh = _f16(d), _f16(dmin)
    df, dminf = dh.astype(np.float32), dminh.astype(np.float32)
    inv_d = np.where(df > 0, 1.0 / np.where(df > 0, df, 1), 0)
    inv_dm = np.where(dminf > 0, 1.0 / np.where(dminf > 0, dminf, 1), 0)
    sc = np.rint(scale * inv_d[:, None]).clip(0, 63).astype(np.uint8)
    mn = np.rint(-mn_ * inv_dm[:, None]).clip(0, 63).astype(np.uint8)
    eff = df[:, None] * sc
    inv_eff = np.where(eff > 0, 1.0 / np.where(eff > 0, eff, 1), 0)
    q = np.rint((x + (dminf[:, None] * mn)[:, :, None]) * inv_eff[:, :, None]).clip(0, qmax).astype(np.uint8)
    return dh, dminh, sc, mn, q


def quantize_q4_k(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, np.float32).reshape(-1, 8, 32)
    dh, dminh, sc, mn, q = _kquant_scales(x, 15)
    nb = x.shape[0]
    out = np.zeros((nb, 144), np.uint8)
    out[:, 0:2] = dh.view(np.uint8).reshape(-1, 2)
    out[:, 2:4] = dminh.view(np.uint8).reshape(-1, 2)
    out[:, 4:16] = _pack_scales_k4(sc, mn)
    q = q.reshape(nb, 4, 2, 32)
    out[:, 16:] = (q[:, :, 0, :] | (q[:, :, 1, :] << 4)).reshape(nb, 128)
    return out.reshape(-1)


def dequantize_q4_k(buf: np.ndarray, n: int) -> np.ndarray:
    b = np.frombuffer(buf, np.uint8, n // 256 * 144).reshape(-1, 144)
    nb = b.shape[0]
    d = b[:, 0:2].copy().view(np.float16).astype(np.float64)
    dmin = b[:, 2:4].copy().view(np.float16).astype(np.float64)
    sc, mn = _unpack_scales_k4(b[:, 4:16])
    qs = b[:, 16:].reshape(nb, 4, 32)
    q = np.stack([qs & 15, qs >> 4], axis=2).reshape(nb, 8, 32).astype(np.float64)
    y = (d * sc)[:, :, None] * q - (dmin * mn)[:, :, None]
    return y.reshape(-1)


def quantize_q5_k(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, np.float32).reshape(-1, 8, 32)
    dh, dminh, sc, mn, q = _kquant_scales(x, 31)
    nb = x.shape[0]
    out = np.zeros((nb, 176), np.uint8)
    out[:, 0:2] = dh.view(np.uint8).reshape(-1, 2)
    out[:, 2:4] = dminh.view(np.uint8).reshape(-1, 2)
    out[:, 4:16] = _pack_scales_k4(sc, mn)
    hi = (q >> 4).astype(np.uint8)                      # [nb, 8, 32]; sub-block s -> bit s of qh[l]
    qh = np.zeros((nb, 32), np.uint8)
    for s in range(8):
        qh |= hi[:, s, :] << s
    out[:, 16:48] = qh
    lo = (q & 15).reshape(nb, 4, 2, 32)
    out[:, 48:] = (lo[:, :, 0, :] | (lo[:, :, 1, :] << 4)).reshape(nb, 128)
    return out.reshape(-1)


def dequantize_q5_k(buf: np.ndarray, n: int) -> np.ndarray:
    b = np.frombuffer(buf, np.uint8, n // 256 * 176).reshape(-1, 176)
    nb = b.shape[0]
    d = b[:, 0:2].copy().view(np.float16).astype(np.float64)
    dmin = b[:, 2:4].copy().view(np.float16).astype(np.float64)
    sc, mn = _unpack_scales_k4(b[:, 4:16])
    qh = b[:, 16:48]
    qs = b[:, 48:].reshape(nb, 4, 32)
    lo = np.stack([qs & 15, qs >> 4], axis=2).reshape(nb, 8, 32)
    hi = np.stack([(qh >> s) & 1 for s in range(8)], axis=1)   # [nb, 8, 32]
    q = (lo | (hi << 4)).astype(np.float64)
    y = (d * sc)[:, :, None] * q - (dmin * mn)[:, :, None]
    return y.reshape(-1)


def quantize_q6_k(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, np.float32).reshape(-1, 16, 16)
    nb = x.shape[0]
    idx = np.argmax(np.abs(x), axis=2)
    mx = np.take_along_axis(x, idx[:, :, None], axis=2)[:, :, 0]
    s = mx / -32.0                                       # per 16-wide sub-block scale (signed)
    d = (np.abs(s).max(1) / 127.0).astype(np.float32)
    dh = _f16(d)
    df = dh.astype(np.float32)
    inv_d = np.where(df > 0, 1.0 / np.where(df > 0, df, 1), 0)
    sc = np.rint(s * inv_d[:, None]).clip(-128, 127).astype(np.int8)
    eff = df[:, None] * sc.astype(np.float32)
    inv_eff = np.where(eff != 0, 1.0 / np.where(eff != 0, eff, 1), 0)
    q = (np.rint(x * inv_eff[:, :, None]).clip(-32, 31) + 32).astype(np.uint8).reshape(nb, 256)
    out = np.zeros((nb, 210), np.uint8)
    ql = np.zeros((nb, 128), np.uint8)
    qh = np.zeros((nb, 64), np.uint8)
    for n_ in range(2):
        w = q[:, 128 * n_:128 * n_ + 128].reshape(nb, 4, 32)   # [nb, a, l] -> y[128n + 32a + l]
        ql[:, 64 * n_:64 * n_ + 32] = (w[:, 0] & 15) | ((w[:, 2] & 15) << 4)
        ql[:, 64 * n_ + 32:64 * n_ + 64] = (w[:, 1] & 15) | ((w[:, 3] & 15) << 4)
        qh[:, 32 * n_:32 * n_ + 32] = (w[:, 0] >> 4) | ((w[:, 1] >> 4) << 2) | ((w[:, 2] >> 4) << 4) | ((w[:, 3] >> 4) << 6)
    out[:, 0:128] = ql
    out[:, 128:192] = qh
    out[:, 192:208] = sc.view(np.uint8)
    out[:, 208:210] = dh.view(np.uint8).reshape(-1, 2)
    return out.reshape(-1)


def dequantize_q6_k(buf: np.ndarray, n: int) -> np.ndarray:
    b = np.frombuffer(buf, np.uint8, n // 256 * 210).reshape(-1, 210)
    nb = b.shape[0]
    ql, qh = b[:, 0:128], b[:, 128:192]
    sc = b[:, 192:208].copy().view(np.int8).astype(np.float64)
    d = b[:, 208:210].copy().view(np.float16).astype(np.float64)
    y = np.zeros((nb, 256), np.float64)
    for n_ in range(2):
        l0 = ql[:, 64 * n_:64 * n_ + 32]
        l1 = ql[:, 64 * n_ + 32:64 * n_ + 64]
        h = qh[:, 32 * n_:32 * n_ + 32]
        w = [
            (l0 & 15) | (((h >> 0) & 3) << 4),
            (l1 & 15) | (((h >> 2) & 3) << 4),
            (l0 >> 4) | (((h >> 4) & 3) << 4),
            (l1 >> 4) | (((h >> 6) & 3) << 4),
        ]
        for a in range(4):
            y[:, 128 * n_ + 32 * a:128 * n_ + 32 * a + 32] = w[a].astype(np.float64) - 32.0
    y = y.reshape(nb, 16, 16) * (d * sc)[:, :, None]
    return y.reshape(-1)


# --------------------------------------------------------------------------- Q2_K (84 bytes: scales[16] (4-bit scale | 4-bit min << 4), qs[64], d fp16, dmin fp16)
def _q2k_values(b: np.ndarray) -> np.ndarray:
    """[nb, 84] -> 2-bit values [nb, 256] in element order: y[128 n + 32 j + l] = (qs[32 n + l] >> 2 j) & 3."""
    qs = b[:, 16:80].astype(np.int32)
    v = np.zeros((b.shape[0], 256), np.int32)
    for n_ in range(2):
        for j in range(4):
            v[:, 128 * n_ + 32 * j:128 * n_ + 32 * j + 32] = (qs[:, 32 * n_:32 * n_ + 32] >> (2 * j)) & 3
    return v


def dequantize_q2_k(buf: np.ndarray, n: int) -> np.ndarray:
    b = np.frombuffer(buf, np.uint8, n // 256 * 84).reshape(-1, 84)
    sc = (b[:, 0:16] & 15).astype(np.float64)
    mn = (b[:, 0:16] >> 4).astype(np.float64)
    d = b[:, 80:82].copy().view(np.float16).astype(np.float64)
    dmin = b[:, 82:84].copy().view(np.float16).astype(np.float64)
    y = _q2k_values(b).reshape(-1, 16, 16).astype(np.float64) * (d * sc)[:, :, None] - (dmin * mn)[:, :, None]
    return y.reshape(-1)


def quantize_q2_k(x: np.ndarray) -> np.ndarray:
    """Valid Q2_K blocks (per sub-block min / range mapped onto 4-bit scale and min against the largest of the super-block); not ggml's make_qkx1_quants search."""
    x = np.asarray(x, np.float32).reshape(-1, 16, 16)
    nb = x.shape[0]
    lo = np.minimum(x.min(axis=2), 0.0)                   # min <= 0 as in ggml (the offset is subtracted)
    hi = x.max(axis=2)
    scale = (hi - lo) / 3.0                               # per sub-block step
    mn = -lo
    d = _f16(scale.max(axis=1) / 15.0)
    dmin = _f16(mn.max(axis=1) / 15.0)
    df, mf = d.astype(np.float32), dmin.astype(np.float32)
    ls = np.where(df[:, None] != 0, np.rint(scale / np.where(df != 0, df, 1)[:, None]), 0).clip(0, 15).astype(np.int32)
    lm = np.where(mf[:, None] != 0, np.rint(mn / np.where(mf != 0, mf, 1)[:, None]), 0).clip(0, 15).astype(np.int32)
    eff, off = df[:, None] * ls, mf[:, None] * lm
    q = np.where(eff[:, :, None] != 0, np.rint((x + off[:, :, None]) / np.where(eff != 0, eff, 1)[:, :, None]), 0).clip(0, 3).astype(np.int32).reshape(nb, 256)
    out = np.zeros((nb, 84), np.uint8)
    qs = np.zeros((nb, 64), np.int32)
    for n_ in range(2):
        for j in range(4):
            qs[:, 32 * n_:32 * n_ + 32] |= q[:, 128 * n_ + 32 * j:128 * n_ + 32 * j + 32] << (2 * j)
    out[:, 0:16] = (ls | (lm << 4)).astype(np.uint8)
    out[:, 16:80] = qs.astype(np.uint8)
    out[:, 80:82] = d.view(np.uint8).reshape(-1, 2)
    out[:, 82:84] = dmin.view(np.uint8).reshape(-1, 2)
    return out.reshape(-1)


# --------------------------------------------------------------------------- Q3_K (110 bytes: hmask[32], qs[64], scales[12] (16 x 6 bit), d fp16)
def _q3k_unpack_scales(sb: np.ndarray) -> np.ndarray:
    """[nb, 12] uint8 -> [nb, 16] int (0..63), ggml's kmask1/kmask2 shuffle (k_quants.c dequantize_row_q3_K)."""
    sb = sb.astype(np.int32)
    out = np.zeros((sb.shape[0], 16), np.int32)
    for j in range(4):
        hi = sb[:, 8 + j]
        out[:, j] = (sb[:, j] & 15) | (((hi >> 0) & 3) << 4)
        out[:, 4 + j] = (sb[:, 4 + j] & 15) | (((hi >> 2) & 3) << 4)
        out[:, 8 + j] = (sb[:, j] >> 4) | (((hi >> 4) & 3) << 4)
        out[:, 12 + j] = (sb[:, 4 + j] >> 4) | (((hi >> 6) & 3) << 4)
    return out


def _q3k_values(b: np.ndarray) -> np.ndarray:
    """[nb, 110] -> signed 3-bit values [nb, 256] in element order (q - 4 when the hmask bit is clear)."""
    nb = b.shape[0]
    hm, qs = b[:, 0:32].astype(np.int32), b[:, 32:96].astype(np.int32)
    v = np.zeros((nb, 256), np.int32)
    for n_ in range(2):
        for j in range(4):
            lo = (qs[:, 32 * n_:32 * n_ + 32] >> (2 * j)) & 3
            hb = (hm >> (4 * n_ + j)) & 1
            v[:, 128 * n_ + 32 * j:128 * n_ + 32 * j + 32] = lo - np.where(hb != 0, 0, 4)
    return v


def dequantize_q3_k(buf: np.ndarray, n: int) -> np.ndarray:
    b = np.frombuffer(buf, np.uint8, n // 256 * 110).reshape(-1, 110)
    sc = _q3k_unpack_scales(b[:, 96:108]).astype(np.float64) - 32.0
    d = b[:, 108:110].copy().view(np.float16).astype(np.float64)
    y = _q3k_values(b).reshape(-1, 16, 16).astype(np.float64) * (d * sc)[:, :, None]
    return y.reshape(-1)


def quantize_q3_k(x: np.ndarray) -> np.ndarray:
    """Valid Q3_K blocks (sub-block scale from the signed maximum, 6-bit scales against the largest one); not ggml's rmse search."""
    x = np.asarray(x, np.float32).reshape(-1, 16, 16)
    nb = x.shape[0]
    idx = np.argmax(np.abs(x), axis=2)
    mx = np.take_along_axis(x, idx[:, :, None], axis=2)[:, :, 0]
    s = mx / -4.0                                        # signed per-sub-block scale: the extreme value maps to -4
    sidx = np.argmax(np.abs(s), axis=1)
    smax = s[np.arange(nb), sidx]
    d = (smax / -32.0).astype(np.float32)                # the largest scale maps to -32
    dh = _f16(d)
    df = dh.astype(np.float32)
    inv_d = np.where(df != 0, 1.0 / np.where(df != 0, df, 1), 0)
    l6 = np.rint(s * inv_d[:, None]).clip(-32, 31).astype(np.int32)       # scale - 32
    eff = df[:, None] * l6.astype(np.float32)
    inv_eff = np.where(eff != 0, 1.0 / np.where(eff != 0, eff, 1), 0)
    q = (np.rint(x * inv_eff[:, :, None]).clip(-4, 3) + 4).astype(np.int32).reshape(nb, 256)   # 0..7
    out = np.zeros((nb, 110), np.uint8)
    hm = np.zeros((nb, 32), np.int32)
    qs = np.zeros((nb, 64), np.int32)
    for n_ in range(2):
        for j in range(4):
            w = q[:, 128 * n_ + 32 * j:128 * n_ + 32 * j + 32]
            hm |= (w >> 2) << (4 * n_ + j)               # bit set <=> value >= 0 (nothing subtracted)
            qs[:, 32 * n_:32 * n_ + 32] |= (w & 3) << (2 * j)
    L = l6 + 32
    sb = np.zeros((nb, 12), np.int32)
    for j in range(16):
        lo, hi = L[:, j] & 15, L[:, j] >> 4
        if j < 8:
            sb[:, j] |= lo
        else:
            sb[:, j - 8] |= lo << 4
        sb[:, 8 + j % 4] |= hi << (2 * (j // 4))
    out[:, 0:32] = hm.astype(np.uint8)
    out[:, 32:96] = qs.astype(np.uint8)
    out[:, 96:108] = sb.astype(np.uint8)
    out[:, 108:110] = dh.view(np.uint8).reshape(-1, 2)
    return out.reshape(-1)


def q3_k_to_q6_k(buf: np.ndarray, n: int) -> np.ndarray:
    """Lossless re-encoding of Q3_K blocks as Q6_K blocks: both are d * scale_16 * q with 16 sub-blocks of 16, so q6 = q3 (in [-4, 3]),
    int8 scale = scale6 - 32, same d -- every dequantised value and every integer block dot product is unchanged.  Numpy twin of the product's
    load-time conversion (csrc/quantize.cpp: q3k_to_q6k)."""
    b = np.frombuffer(buf, np.uint8, n // 256 * 110).reshape(-1, 110)
    nb = b.shape[0]
    q = (_q3k_values(b) + 32).astype(np.uint8)           # 6-bit field, 28..35
    sc = (_q3k_unpack_scales(b[:, 96:108]) - 32).astype(np.int8)
    out = np.zeros((nb, 210), np.uint8)
    for n_ in range(2):
        w = q[:, 128 * n_:128 * n_ + 128].reshape(nb, 4, 32)
        out[:, 64 * n_:64 * n_ + 32] = (w[:, 0] & 15) | ((w[:, 2] & 15) << 4)
        out[:, 64 * n_ + 32:64 * n_ + 64] = (w[:, 1] & 15) | ((w[:, 3] & 15) << 4)
        out[:, 128 + 32 * n_:128 + 32 * n_ + 32] = (w[:, 0] >> 4) | ((w[:, 1] >> 4) << 2) | ((w[:, 2] >> 4) << 4) | ((w[:, 3] >> 4) << 6)
    out[:, 192:208] = sc.view(np.uint8)
    out[:, 208:210] = b[:, 108:110]
    return out.reshape(-1)


# --------------------------------------------------------------------------- dispatch
def quantize(gtype: int, x: np.ndarray) -> np.ndarray:
    """float array -> raw bytes (uint8) in ggml block layout, row-major over the flattened array."""
    x = np.ascontiguousarray(x)
    if gtype == GGML_F32:
        return x.astype(np.float32).reshape(-1).view(np.uint8)
    if gtype == GGML_F16:
        return x.astype(np.float16).reshape(-1).view(np.uint8)
    fn = {
        GGML_Q4_0: quantize_q4_0, GGML_Q4_1: quantize_q4_1, GGML_Q5_0: quantize_q5_0,
        GGML_Q5_1: quantize_q5_1, GGML_Q8_0: quantize_q8_0, GGML_Q4_K: quantize_q4_k,
        GGML_Q5_K: quantize_q5_k, GGML_Q6_K: quantize_q6_k, GGML_Q3_K: quantize_q3_k, GGML_Q2_K: quantize_q2_k,
    }[gtype]
    return fn(x)


def dequantize(gtype: int, buf, n: int) -> np.ndarray:
    """raw bytes -> float64[n]."""
    if gtype == GGML_F32:
        return np.frombuffer(buf, np.float32, n).astype(np.float64)
    if gtype == GGML_F16:
        return np.frombuffer(buf, np.float16, n).astype(np.float64)
    fn = {
        GGML_Q4_0: dequantize_q4_0, GGML_Q4_1: dequantize_q4_1, GGML_Q5_0: dequantize_q5_0,
        GGML_Q5_1: dequantize_q5_1, GGML_Q8_0: dequantize_q8_0, GGML_Q4_K: dequantize_q4_k,
        GGML_Q5_K: dequantize_q5_k, GGML_Q6_K: dequantize_q6_k, GGML_Q3_K: dequantize_q3_k, GGML_Q2_K: dequantize_q2_k,
    }[gtype]
    return fn(np.frombuffer(buf, np.uint8), n)
