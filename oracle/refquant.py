"""CPU oracle for minigpt4_quantize_model's block quantisers (TEST INFRASTRUCTURE ONLY -- never imported by the product).

The reference quantises with `ggml_quantize_chunk` (/root/reference/minigpt4.cpp:2932), i.e. ggml's *reference* row quantisers of llama.cpp
master-31cfbb1 (ggml.c `quantize_row_q{4_0,4_1,5_0,5_1,8_0}_reference`, k_quants.c `quantize_row_q{4,5,6}_K_reference` with the `make_qkx1_quants` /
`make_qx_quants` searches).  Those sources are not on this machine: the algorithms are restated here from their published form, in scalar
numpy float32 arithmetic, independently of the C++ restatement in csrc/quantize.cpp -- PARITY UNPINNED: the two restatements must agree byte for byte
(tests/test_cpu_quantize.py), the block layouts are those of the pinned-by-layout dequantisers in minigpt4.cpp_amd/quants.py, but no golden vector of ggml
itself exists here.

Every arithmetic step is a float32 operation in ggml's order (C `float` expressions without contraction).
"""
import numpy as np

F = np.float32


def _h(x) -> np.float32:          # GGML_FP32_TO_FP16 then back
    return F(np.float16(F(x)))


def _hbytes(x) -> bytes:
    return np.float16(F(x)).tobytes()


def nearest_int(f) -> int:
    v = F(F(f) + F(12582912.0))
    i = int(np.frombuffer(v.tobytes(), np.int32)[0])
    return (i & 0x007FFFFF) - 0x00400000


def _i8(x) -> int:                # (int8_t)(float): truncation toward zero, then wrap to 8 bits (the values that occur here fit)
    return int(np.int8(int(x)))


def q4_0(x):
    amax, mx = F(0), F(0)
    for v in x:
        if amax < abs(v):
            amax, mx = abs(v), v
    d = F(mx / F(-8))
    idv = F(F(1) / d) if d != 0 else F(0)
    out = bytearray(_hbytes(d))
    for j in range(16):
        x0, x1 = F(x[j] * idv), F(x[16 + j] * idv)
        a, b = min(15, _i8(F(x0 + F(8.5)))), min(15, _i8(F(x1 + F(8.5))))
        out.append((a & 0xFF) | ((b & 0xFF) << 4) & 0xFF)
    return bytes(out)


def q4_1(x):
    mn, mx = F(np.finfo(np.float32).max), F(-np.finfo(np.float32).max)
    for v in x:
        mn, mx = min(mn, v), max(mx, v)
    d = F(F(mx - mn) / F(15))
    idv = F(F(1) / d) if d != 0 else F(0)
    out = bytearray(_hbytes(d) + _hbytes(mn))
    for j in range(16):
        x0, x1 = F(F(x[j] - mn) * idv), F(F(x[16 + j] - mn) * idv)
        a, b = min(15, _i8(F(x0 + F(0.5)))), min(15, _i8(F(x1 + F(0.5))))
        out.append((a | (b << 4)) & 0xFF)
    return bytes(out)


def q5_0(x):
    amax, mx = F(0), F(0)
    for v in x:
        if amax < abs(v):
            amax, mx = abs(v), v
    d = F(mx / F(-16))
    idv = F(F(1) / d) if d != 0 else F(0)
    qs, qh = bytearray(), 0
    for j in range(16):
        x0, x1 = F(x[j] * idv), F(x[16 + j] * idv)
        a, b = min(31, _i8(F(x0 + F(16.5)))) & 0xFF, min(31, _i8(F(x1 + F(16.5)))) & 0xFF
        qs.append((a & 0x0F) | ((b & 0x0F) << 4))
        qh |= ((a & 0x10) >> 4) << j
        qh |= ((b & 0x10) >> 4) << (j + 16)
    return _hbytes(d) + int(qh).to_bytes(4, "little") + bytes(qs)


def q5_1(x):
    mn, mx = F(np.finfo(np.float32).max), F(-np.finfo(np.float32).max)
    for v in x:
        mn, mx = min(mn, v), max(mx, v)
    d = F(F(mx - mn) / F(31))
    idv = F(F(1) / d) if d != 0 else F(0)
    qs, qh = bytearray(), 0
    for j in range(16):
        x0, x1 = F(F(x[j] - mn) * idv), F(F(x[16 + j] - mn) * idv)
        a, b = int(F(x0 + F(0.5))) & 0xFF, int(F(x1 + F(0.5))) & 0xFF
        qs.append((a & 0x0F) | ((b & 0x0F) << 4))
        qh |= ((a & 0x10) >> 4) << j
        qh |= ((b & 0x10) >> 4) << (j + 16)
    return _hbytes(d) + _hbytes(mn) + int(qh).to_bytes(4, "little") + bytes(qs)


def _roundf(v) -> int:            # C roundf: half away from zero
    v = float(v)
    return int(np.floor(v + 0.5)) if v >= 0 else -int(np.floor(-v + 0.5))


def q8_0(x):
    amax = F(0)
    for v in x:
        amax = max(amax, abs(v))
    d = F(amax / F(127))
    idv = F(F(1) / d) if d != 0 else F(0)
    return _hbytes(d) + bytes((_roundf(F(v * idv)) & 0xFF) for v in x)


def make_qkx1_quants(n, nmax, x, ntry):
    mn, mx = x[0], x[0]
    for v in x[1:]:
        mn, mx = min(mn, v), max(mx, v)
    L = [0] * n
    if mx == mn:
        return F(0), L, F(0)
    if mn > 0:
        mn = F(0)
    iscale = F(F(nmax) / F(mx - mn))
    scale = F(F(1) / iscale)
    for _ in range(ntry):
        sumlx, suml2, did_change = F(0), 0, False
        for i in range(n):
            l = max(0, min(nmax, nearest_int(F(iscale * F(x[i] - mn)))))
            if l != L[i]:
                L[i], did_change = l, True
            sumlx = F(sumlx + F(F(x[i] - mn) * F(l)))
            suml2 += l * l
        scale = F(sumlx / F(suml2))
        s = F(0)
        for i in range(n):
            s = F(s + F(x[i] - F(scale * F(L[i]))))
        mn = F(s / F(n))
        if mn > 0:
            mn = F(0)
        iscale = F(F(1) / scale)
        if not did_change:
            break
    return scale, L, F(-mn)


def _scale_min_k4(j, q):
    if j < 4:
        return q[j] & 63, q[j + 4] & 63
    return (q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4), (q[j + 4] >> 4) | ((q[j] >> 6) << 4)


def _q45_levels(x, nmax):
    scales, mins, L = [], [], []
    for j in range(8):
        s, l, m = make_qkx1_quants(32, nmax, x[32 * j:32 * j + 32], 5)
        scales.append(s)
        mins.append(m)
        L += l
    max_scale, max_min = max([F(0)] + scales), max([F(0)] + mins)
    inv_scale = F(F(63) / max_scale) if max_scale > 0 else F(0)
    inv_min = F(F(63) / max_min) if max_min > 0 else F(0)
    sc = [0] * 12
    for j in range(8):
        ls, lm = min(63, nearest_int(F(inv_scale * scales[j]))), min(63, nearest_int(F(inv_min * mins[j])))
        if j < 4:
            sc[j], sc[j + 4] = ls, lm
        else:
            sc[j + 4] = (ls & 0xF) | ((lm & 0xF) << 4)
            sc[j - 4] |= (ls >> 4) << 6
            sc[j] |= (lm >> 4) << 6
    d16, m16 = _hbytes(F(max_scale / F(63))), _hbytes(F(max_min / F(63)))
    dd, dmin = _h(F(max_scale / F(63))), _h(F(max_min / F(63)))
    for j in range(8):
        s, m = _scale_min_k4(j, sc)
        d = F(dd * F(s))
        if d == 0:
            continue
        dm = F(dmin * F(m))
        for ii in range(32):
            L[32 * j + ii] = max(0, min(nmax, nearest_int(F(F(x[32 * j + ii] + dm) / d))))
    return d16 + m16 + bytes(sc), L


def q2_k(x):
    """quantize_row_q2_K_reference: {scales[16] (scale | min << 4), qs[64], d, dmin}."""
    scales, mins, L = [], [], []
    for j in range(16):
        s, l, m = make_qkx1_quants(16, 3, x[16 * j:16 * j + 16], 5)
        scales.append(s); mins.append(m); L += l
    max_scale, max_min = F(0), F(0)
    for j in range(16):
        if scales[j] > max_scale:
            max_scale = scales[j]
        if mins[j] > max_min:
            max_min = mins[j]
    ysc = [0] * 16
    if max_scale > 0:
        iscale = F(F(15) / max_scale)
        for j in range(16):
            ysc[j] = nearest_int(F(iscale * scales[j])) & 0xFF
        d_b = _hbytes(F(max_scale / F(15)))
    else:
        d_b = _hbytes(F(0))
    if max_min > 0:
        iscale = F(F(15) / max_min)
        for j in range(16):
            ysc[j] = (ysc[j] | (nearest_int(F(iscale * mins[j])) << 4)) & 0xFF
        m_b = _hbytes(F(max_min / F(15)))
    else:
        m_b = _hbytes(F(0))
    dd, dmin = _h(F(max_scale / F(15))) if max_scale > 0 else F(0), _h(F(max_min / F(15))) if max_min > 0 else F(0)
    for j in range(16):
        d = F(dd * F(ysc[j] & 0xF))
        if d == 0:
            continue
        dm = F(dmin * F(ysc[j] >> 4))
        for ii in range(16):
            L[16 * j + ii] = max(0, min(3, nearest_int(F(F(x[16 * j + ii] + dm) / d))))
    qs = [0] * 64
    for j in range(0, 256, 128):
        for l in range(32):
            qs[j // 4 + l] = L[j + l] | (L[j + l + 32] << 2) | (L[j + l + 64] << 4) | (L[j + l + 96] << 6)
    return bytes(ysc) + bytes(qs) + d_b + m_b


def make_q3_quants(n, nmax, x):
    """make_q3_quants with do_rmse = true: returns (scale, levels offset by nmax)."""
    mx, amax = F(0), F(0)
    for v in x:
        if abs(v) > amax:
            amax, mx = abs(v), v
    if amax == 0:
        return F(0), [0] * n
    iscale = F(F(-nmax) / mx)
    clamp = lambda l: max(-nmax, min(nmax - 1, l))          # noqa: E731
    L, sumlx, suml2 = [0] * n, F(0), F(0)
    for i in range(n):
        l = clamp(nearest_int(F(iscale * x[i])))
        L[i] = l
        w = F(x[i] * x[i])
        sumlx = F(sumlx + F(F(w * x[i]) * F(l)))
        suml2 = F(suml2 + F(F(w * F(l)) * F(l)))
    for _ in range(5):
        n_changed = 0
        for i in range(n):
            w = F(x[i] * x[i])
            slx = F(sumlx - F(F(w * x[i]) * F(L[i])))
            if slx > 0:
                sl2 = F(suml2 - F(F(w * F(L[i])) * F(L[i])))
                new_l = clamp(nearest_int(F(F(x[i] * sl2) / slx)))
                if new_l != L[i]:
                    slx = F(slx + F(F(w * x[i]) * F(new_l)))
                    sl2 = F(sl2 + F(F(w * F(new_l)) * F(new_l)))
                    if sl2 > 0 and F(F(slx * slx) * suml2) > F(F(sumlx * sumlx) * sl2):
                        L[i] = new_l
                        sumlx, suml2 = slx, sl2
                        n_changed += 1
        if not n_changed:
            break
    return F(sumlx / suml2), [l + nmax for l in L]


def q3_k(x):
    """quantize_row_q3_K_reference: {hmask[32], qs[64], scales[12] (sixteen 6-bit values, offset 32), d}."""
    scales, L = [], []
    max_scale, amax = F(0), F(0)
    for j in range(16):
        s, l = make_q3_quants(16, 4, x[16 * j:16 * j + 16])
        scales.append(s); L += l
        if abs(s) > amax:
            amax, max_scale = abs(s), s
    ysc = [0] * 12
    if max_scale != 0:
        iscale = F(F(-32) / max_scale)
        for j in range(16):
            l = max(-32, min(31, nearest_int(F(iscale * scales[j])))) + 32
            if j < 8:
                ysc[j] = l & 0xF
            else:
                ysc[j - 8] |= (l & 0xF) << 4
            ysc[j % 4 + 8] |= (l >> 4) << (2 * (j // 4))
        d_b, dd = _hbytes(F(F(1) / iscale)), _h(F(F(1) / iscale))
    else:
        d_b, dd = _hbytes(F(0)), F(0)
    for j in range(16):
        sc = (ysc[j] & 0xF) if j < 8 else (ysc[j - 8] >> 4)
        sc = (sc | (((ysc[8 + j % 4] >> (2 * (j // 4))) & 3) << 4)) - 32
        d = F(dd * F(sc))
        if d == 0:
            continue
        for ii in range(16):
            L[16 * j + ii] = max(-4, min(3, nearest_int(F(x[16 * j + ii] / d)))) + 4
    hmask = [0] * 32
    for j in range(256):
        if L[j] > 3:
            hmask[j % 32] |= 1 << (j // 32)
            L[j] -= 4
    qs = [0] * 64
    for j in range(0, 256, 128):
        for l in range(32):
            qs[j // 4 + l] = L[j + l] | (L[j + l + 32] << 2) | (L[j + l + 64] << 4) | (L[j + l + 96] << 6)
    return bytes(hmask) + bytes(qs) + bytes(ysc) + d_b


def q4_k(x):
    head, L = _q45_levels(x, 15)
    qs = bytearray()
    for j in range(0, 256, 64):
        for l in range(32):
            qs.append(L[j + l] | (L[j + l + 32] << 4))
    return head + bytes(qs)


def q5_k(x):
    head, L = _q45_levels(x, 31)
    qh, ql = [0] * 32, bytearray()
    m1, m2 = 1, 2
    for n in range(0, 256, 64):
        for j in range(32):
            l1, l2 = L[n + j], L[n + j + 32]
            if l1 > 15:
                l1 -= 16
                qh[j] |= m1
            if l2 > 15:
                l2 -= 16
                qh[j] |= m2
            ql.append(l1 | (l2 << 4))
        m1, m2 = (m1 << 2) & 0xFF, (m2 << 2) & 0xFF
    return head + bytes(qh) + bytes(ql)


def make_qx_quants(n, nmax, x, rmse_type):
    mx, amax = F(0), F(0)
    for v in x:
        if abs(v) > amax:
            amax, mx = abs(v), v
    if amax == 0:
        return F(0), [0] * n
    iscale = F(F(-nmax) / mx)
    clamp = lambda l: max(-nmax, min(nmax - 1, l))          # noqa: E731
    if rmse_type == 0:
        return F(F(1) / iscale), [nmax + clamp(nearest_int(F(iscale * v))) for v in x]
    wt = rmse_type % 2
    w = [F(v * v) if wt == 1 else F(1) for v in x]
    L, sumlx, suml2 = [0] * n, F(0), F(0)
    for i in range(n):
        l = clamp(nearest_int(F(iscale * x[i])))
        L[i] = l + nmax
        sumlx = F(sumlx + F(F(w[i] * x[i]) * F(l)))
        suml2 = F(suml2 + F(F(w[i] * F(l)) * F(l)))
    scale = F(sumlx / suml2)
    best = F(scale * sumlx)
    for _ in range(3):
        iscale = F(F(1) / scale)
        slx, sl2, changed = F(0), F(0), False
        for i in range(n):
            l = clamp(nearest_int(F(iscale * x[i])))
            if l + nmax != L[i]:
                changed = True
            slx = F(slx + F(F(w[i] * x[i]) * F(l)))
            sl2 = F(sl2 + F(F(w[i] * F(l)) * F(l)))
        if not changed or sl2 == 0 or F(slx * slx) <= F(best * sl2):
            break
        for i in range(n):
            L[i] = nmax + clamp(nearest_int(F(iscale * x[i])))
        sumlx, suml2 = slx, sl2
        scale = F(sumlx / suml2)
        best = F(scale * sumlx)
    for _ in range(5):
        n_changed = 0
        for i in range(n):
            l = L[i] - nmax
            slx = F(sumlx - F(F(w[i] * x[i]) * F(l)))
            if slx > 0:
                sl2 = F(suml2 - F(F(w[i] * F(l)) * F(l)))
                new_l = clamp(nearest_int(F(F(x[i] * sl2) / slx)))
                if new_l != l:
                    slx = F(slx + F(F(w[i] * x[i]) * F(new_l)))
                    sl2 = F(sl2 + F(F(w[i] * F(new_l)) * F(new_l)))
                    if sl2 > 0 and F(F(slx * slx) * suml2) > F(F(sumlx * sumlx) * sl2):
                        L[i] = nmax + new_l
                        sumlx, suml2 = slx, sl2
                        scale = F(sumlx / suml2)
                        best = F(scale * sumlx)
                        n_changed += 1
        if not n_changed:
            break
    return scale, L


def q6_k(x):
    L, scales = [], []
    max_scale, max_abs = F(0), F(0)
    for ib in range(16):
        s, l = make_qx_quants(16, 32, x[16 * ib:16 * ib + 16], 1)
        scales.append(s)
        L += l
        if abs(s) > max_abs:
            max_abs, max_scale = abs(s), s
    with np.errstate(divide="ignore", invalid="ignore"):
        iscale = F(F(-128) / max_scale)
        d16 = _hbytes(F(F(1) / iscale))
        dd = _h(F(F(1) / iscale))
        sc = [min(127, nearest_int(F(iscale * s))) if np.isfinite(F(iscale * s)) else 0 for s in scales]
    for j in range(16):
        d = F(dd * F(sc[j]))
        if d == 0:
            continue
        for ii in range(16):
            L[16 * j + ii] = max(-32, min(31, nearest_int(F(x[16 * j + ii] / d)))) + 32
    ql, qh = bytearray(128), bytearray(64)
    for half in range(2):
        j = 128 * half
        for l in range(32):
            q1, q2, q3, q4 = L[j + l] & 0xF, L[j + l + 32] & 0xF, L[j + l + 64] & 0xF, L[j + l + 96] & 0xF
            ql[64 * half + l] = q1 | (q3 << 4)
            ql[64 * half + l + 32] = q2 | (q4 << 4)
            qh[32 * half + l] = (L[j + l] >> 4) | ((L[j + l + 32] >> 4) << 2) | ((L[j + l + 64] >> 4) << 4) | ((L[j + l + 96] >> 4) << 6)
    return bytes(ql) + bytes(qh) + bytes((s & 0xFF) for s in sc) + d16


BLOCK_FN = {2: (32, q4_0), 3: (32, q4_1), 6: (32, q5_0), 7: (32, q5_1), 8: (32, q8_0), 10: (256, q2_k), 11: (256, q3_k), 12: (256, q4_k), 13: (256, q5_k), 14: (256, q6_k)}


def quantize_chunk(ggml_type: int, x: np.ndarray) -> np.ndarray:
    blk, fn = BLOCK_FN[ggml_type]
    x = np.ascontiguousarray(x, np.float32).reshape(-1)
    assert x.size % blk == 0
    return np.frombuffer(b"".join(fn([F(v) for v in x[i:i + blk]]) for i in range(0, x.size, blk)), np.uint8).copy()
