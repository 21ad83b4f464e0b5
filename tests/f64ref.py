"""Independent float64 numpy evaluation of the two models (tolerance reference for the oracle).

No quantised-activation tricks, no fp16 rounding points: plain math on the dequantised weights.
The oracle (ggml numerics) must agree with this to within the activation-quantisation noise.
"""
from __future__ import annotations

import numpy as np


def _rms(x, w, eps=1e-6):
    return x / np.sqrt((x * x).mean(-1, keepdims=True) + eps) * w


def _ln(x, w, b, eps=1e-5):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * w + b


def _softmax(s):
    s = s - s.max(-1, keepdims=True)
    e = np.exp(s)
    return e / e.sum(-1, keepdims=True)


def _gelu(x):
    return 0.5 * x * (1.0 + np.tanh(0.7978845608028654 * x * (1.0 + 0.044715 * x * x)))


def _rope(x, pos0, n_head):
    # x: [N, E]; interleaved pairs, theta_i = pos * 10000^(-2i/hd)
    N, E = x.shape
    hd = E // n_head
    x = x.reshape(N, n_head, hd // 2, 2).copy()
    inv = 10000.0 ** (-2.0 * np.arange(hd // 2) / hd)
    ang = (pos0 + np.arange(N))[:, None] * inv[None, :]
    c, s = np.cos(ang)[:, None, :], np.sin(ang)[:, None, :]
    x0, x1 = x[..., 0].copy(), x[..., 1].copy()
    x[..., 0] = x0 * c - x1 * s
    x[..., 1] = x0 * s + x1 * c
    return x.reshape(N, E)


class LlamaF64:
    def __init__(self, llm_file):
        self.f = llm_file
        hp = llm_file.hparams
        self.E, self.H, self.L, self.V = hp["n_embd"], hp["n_head"], hp["n_layer"], hp["n_vocab"]
        self.w = {k: llm_file.f64(k) for k in llm_file.tensors}
        self.k = [np.zeros((0, self.E)) for _ in range(self.L)]
        self.v = [np.zeros((0, self.E)) for _ in range(self.L)]

    def eval(self, tokens=None, embd=None):
        w = self.w
        x = w["tok_embeddings.weight"][np.asarray(tokens)] if tokens is not None else np.asarray(embd, np.float64)
        N = x.shape[0]
        hd = self.E // self.H
        for il in range(self.L):
            p = f"layers.{il}."
            n_past = self.k[il].shape[0]
            cur = _rms(x, w[p + "attention_norm.weight"])
            q = _rope(cur @ w[p + "attention.wq.weight"].T, n_past, self.H)
            k = _rope(cur @ w[p + "attention.wk.weight"].T, n_past, self.H)
            v = cur @ w[p + "attention.wv.weight"].T
            self.k[il] = np.concatenate([self.k[il], k])
            self.v[il] = np.concatenate([self.v[il], v])
            T = self.k[il].shape[0]
            qh = q.reshape(N, self.H, hd).transpose(1, 0, 2)
            kh = self.k[il].reshape(T, self.H, hd).transpose(1, 0, 2)
            vh = self.v[il].reshape(T, self.H, hd).transpose(1, 0, 2)
            s = qh @ kh.transpose(0, 2, 1) / np.sqrt(hd)
            mask = np.arange(T)[None, :] > (n_past + np.arange(N))[:, None]
            s = np.where(mask[None], -np.inf, s)
            a = (_softmax(s) @ vh).transpose(1, 0, 2).reshape(N, self.E)
            x = x + a @ w[p + "attention.wo.weight"].T
            cur = _rms(x, w[p + "ffn_norm.weight"])
            g = cur @ w[p + "feed_forward.w1.weight"].T
            u = cur @ w[p + "feed_forward.w3.weight"].T
            x = x + ((g / (1.0 + np.exp(-g))) * u) @ w[p + "feed_forward.w2.weight"].T
        x = _rms(x, w["norm.weight"])
        return x @ w["output.weight"].T  # [N, V]


def vision_f64(vf, image_chw, stage=0):
    g = lambda m, n: vf.f64(m, n)
    ve = "visual_encoder"
    D = vf.models[ve]["pos_embed"].ne[0]
    img = np.asarray(image_chw, np.float64).reshape(3, 16, 14, 16, 14)
    patches = img.transpose(1, 3, 0, 2, 4).reshape(256, 588)
    wpe = g(ve, "patch_embed.proj.weight").reshape(D, 588)
    pe = patches @ wpe.T + g(ve, "patch_embed.proj.bias")
    x = np.concatenate([g(ve, "cls_token").reshape(1, D), pe]) + g(ve, "pos_embed").reshape(257, D)
    if stage == 1:
        return x
    heads = D // 88
    i = 0
    while f"blocks.{i}.norm1.weight" in vf.models[ve]:
        p = f"blocks.{i}."
        cur = _ln(x, g(ve, p + "norm1.weight"), g(ve, p + "norm1.bias"))
        bias = np.concatenate([g(ve, p + "attn.q_bias"), np.zeros(D), g(ve, p + "attn.v_bias")])
        qkv = cur @ g(ve, p + "attn.qkv.weight").T + bias
        q, k, v = [qkv[:, j * D:(j + 1) * D].reshape(257, heads, 88).transpose(1, 0, 2) for j in range(3)]
        a = _softmax((q / np.sqrt(88.0)) @ k.transpose(0, 2, 1)) @ v
        a = a.transpose(1, 0, 2).reshape(257, D)
        x = x + a @ g(ve, p + "attn.proj.weight").T + g(ve, p + "attn.proj.bias")
        cur = _ln(x, g(ve, p + "norm2.weight"), g(ve, p + "norm2.bias"))
        h = _gelu(cur @ g(ve, p + "mlp.fc1.weight").T + g(ve, p + "mlp.fc1.bias"))
        x = x + h @ g(ve, p + "mlp.fc2.weight").T + g(ve, p + "mlp.fc2.bias")
        i += 1
    img_e = _ln(x, g("ln_vision", "weight"), g("ln_vision", "bias"))
    if stage == 2:
        return img_e
    qf = "Qformer"
    hs = _ln(g("query_tokens", "weight").reshape(-1, 768), g(qf, "bert.embeddings.LayerNorm.weight"),
             g(qf, "bert.embeddings.LayerNorm.bias"))

    def attn(prefix, hidden, enc):
        src = hidden if enc is None else enc
        Qm = hidden @ g(qf, prefix + "self.query.weight").T + g(qf, prefix + "self.query.bias")
        K = src @ g(qf, prefix + "self.key.weight").T + g(qf, prefix + "self.key.bias")
        V = src @ g(qf, prefix + "self.value.weight").T + g(qf, prefix + "self.value.bias")
        sp = lambda t: t.reshape(t.shape[0], 12, 64).transpose(1, 0, 2)
        c = (_softmax(sp(Qm) @ sp(K).transpose(0, 2, 1) / 8.0) @ sp(V)).transpose(1, 0, 2).reshape(-1, 768)
        d = c @ g(qf, prefix + "output.dense.weight").T + g(qf, prefix + "output.dense.bias") + hidden
        return _ln(d, g(qf, prefix + "output.LayerNorm.weight"), g(qf, prefix + "output.LayerNorm.bias"))

    l = 0
    while f"bert.encoder.layer.{l}.attention.self.query.weight" in vf.models[qf]:
        p = f"bert.encoder.layer.{l}."
        a = attn(p + "attention.", hs, None)
        if (p + "crossattention.self.query.weight") in vf.models[qf]:
            a = attn(p + "crossattention.", a, img_e)
        im = _gelu(a @ g(qf, p + "intermediate_query.dense.weight").T + g(qf, p + "intermediate_query.dense.bias"))
        o = im @ g(qf, p + "output_query.dense.weight").T + g(qf, p + "output_query.dense.bias") + a
        hs = _ln(o, g(qf, p + "output_query.LayerNorm.weight"), g(qf, p + "output_query.LayerNorm.bias"))
        l += 1
    if stage == 3:
        return hs
    return hs @ g("llama_proj", "weight").T + g("llama_proj", "bias")
