"""Oracle (test infrastructure): CPU restatement of the ADT network.

Functional fp32 torch-CPU code with every step written out (no
``nn.Transformer*`` modules), following the reference:

  * ``ADTModel.forward`` / ``_loss_fn``          model.py:228-258
  * ``Encoder.forward``                          model.py:129-135 (layers :118-127)
  * ``Decoder.forward`` (additive -1e4 masks)    model.py:170-190
  * ``TokenEmbedding_plain`` / ``PositionalEncoding``  model.py:42-65
  * ``ADTModel.sample`` (greedy)                 model.py:260-324
  * ``ADTTrainer.compute_loss`` teacher forcing  train.py:56-70

``nn.TransformerEncoderLayer`` / ``nn.TransformerDecoderLayer`` are used by the
reference with ``norm_first=False`` (post-norm), ``activation="gelu"`` (exact
erf), LayerNorm eps 1e-5, ``batch_first=True`` and no final norm inside
``nn.TransformerEncoder/Decoder``; ``nn.MultiheadAttention`` packs q/k/v in
``in_proj_weight[3d, d]`` and adds float ``attn_mask`` and ``key_padding_mask``
together before the softmax.

Pinned by tests/golden/adt_tiny.npz (exact tensors from the reference's own
modules) and tests/golden/adt_full_stats.npz (full-size statistics).

``bf16=True`` rounds every GEMM / attention operand to bfloat16 (fp32
accumulate), which is what the HIP path computes and what the reference does
under ``bf16`` autocast on a GPU; it is used to check the kernels with a tight
tolerance, separately from the precision question.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import logmel as o_logmel

State = Dict[str, torch.Tensor]


# ----------------------------------------------------------------------------- helpers
def _r(x: torch.Tensor, bf16: bool) -> torch.Tensor:
    return x.bfloat16().float() if bf16 else x


def linear(x, w, b=None, bf16=False):
    y = _r(x, bf16) @ _r(w, bf16).t()
    return y if b is None else y + b


def layer_norm(x, w, b, eps=1e-5):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def gelu(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def positional_encoding(d: int, maxlen: int = 2048) -> torch.Tensor:
    """model.py:55-62."""
    den = torch.exp(-torch.arange(0, d, 2) * math.log(10000) / d)
    pos = torch.arange(0, maxlen).reshape(maxlen, 1)
    pe = torch.zeros((maxlen, d))
    pe[:, 0::2] = torch.sin(pos * den)
    pe[:, 1::2] = torch.cos(pos * den)
    return pe.unsqueeze(0)


def causal_mask(T: int) -> torch.Tensor:
    """utils/utils.py:28-33 -- True = masked."""
    return torch.triu(torch.ones(T, T, dtype=torch.bool), diagonal=1)


def key_padding_mask(lengths, T: int) -> torch.Tensor:
    """utils/utils.py:36-43 -- True where position >= length."""
    lengths = torch.as_tensor(lengths)
    return torch.arange(T).unsqueeze(0) >= lengths.unsqueeze(1)


def mha(x_q, x_kv, in_w, in_b, out_w, out_b, nhead, add_mask=None, bf16=False, pdrop=None):
    """nn.MultiheadAttention forward (batch_first).  ``add_mask`` is an additive
    float mask broadcastable to [B, H, Tq, Tk]."""
    B, Tq, d = x_q.shape
    Tk = x_kv.shape[1]
    dh = d // nhead
    q = linear(x_q, in_w[:d], in_b[:d], bf16)
    k = linear(x_kv, in_w[d:2 * d], in_b[d:2 * d], bf16)
    v = linear(x_kv, in_w[2 * d:], in_b[2 * d:], bf16)
    q = q.view(B, Tq, nhead, dh).transpose(1, 2)
    k = k.view(B, Tk, nhead, dh).transpose(1, 2)
    v = v.view(B, Tk, nhead, dh).transpose(1, 2)
    s = (_r(q, bf16) @ _r(k, bf16).transpose(-1, -2)) / math.sqrt(dh)
    if add_mask is not None:
        s = s + add_mask
    p = torch.softmax(s, dim=-1)
    if pdrop is not None:                      # dropout on the attention probabilities (nn.MultiheadAttention dropout)
        p = p * pdrop(p.shape)
    o = _r(p, bf16) @ _r(v, bf16)
    o = o.transpose(1, 2).reshape(B, Tq, d)
    return linear(o, out_w, out_b, bf16)


# ----------------------------------------------------------------------------- network
def _nodrop(site):
    return None


def _apply(x, d):
    return x if d is None else x * d(x.shape)


def encoder(state: State, x: torch.Tensor, nhead: int, n_layers: int, bf16=False, drop=_nodrop) -> torch.Tensor:
    """model.py:129-135: dense (no bias) -> +PE -> dropout -> layers -> LN -> dropout.
    ``drop(site)`` returns None or a function shape -> scale tensor (mask / (1-p)); the sites are the
    dropout modules of the reference (Encoder.dropout_layer, and dropout / dropout1 / dropout2 + the
    attention dropout inside nn.TransformerEncoderLayer)."""
    p = "encoder."
    x = linear(x, state[p + "dense_layer.weight"], None, bf16)
    x = x + state[p + "positional_encoding.pos_embedding"][:, : x.size(1), :]
    x = _apply(x, drop("enc.pe"))
    for i in range(n_layers):
        q = f"{p}encoder.layers.{i}."
        s_ = q[:-1]
        a = mha(x, x, state[q + "self_attn.in_proj_weight"], state[q + "self_attn.in_proj_bias"],
                state[q + "self_attn.out_proj.weight"], state[q + "self_attn.out_proj.bias"], nhead, None, bf16, drop(s_ + ".attn"))
        x = layer_norm(x + _apply(a, drop(s_ + ".drop1")), state[q + "norm1.weight"], state[q + "norm1.bias"])
        h = _apply(gelu(linear(x, state[q + "linear1.weight"], state[q + "linear1.bias"], bf16)), drop(s_ + ".ffn"))
        h = linear(h, state[q + "linear2.weight"], state[q + "linear2.bias"], bf16)
        x = layer_norm(x + _apply(h, drop(s_ + ".drop2")), state[q + "norm2.weight"], state[q + "norm2.bias"])
    return _apply(layer_norm(x, state[p + "layer_norm.weight"], state[p + "layer_norm.bias"]), drop("enc.final"))


def decoder(state: State, tgt: torch.Tensor, memory: torch.Tensor, nhead: int, n_layers: int,
            tgt_mask: Optional[torch.Tensor], tgt_padding_mask: Optional[torch.Tensor], bf16=False, drop=_nodrop) -> torch.Tensor:
    """model.py:170-190.  Bool masks become additive 0 / -1e4 floats (:173-181);
    PyTorch adds the causal and the key-padding mask."""
    p = "decoder."
    d = state[p + "tgt_tok_emb.embedding.weight"].shape[1]
    x = state[p + "tgt_tok_emb.embedding.weight"][tgt.long()] * math.sqrt(d)
    x = x + state[p + "positional_encoding.pos_embedding"][:, : x.size(1), :]
    x = _apply(x, drop("dec.emb"))
    add = None
    if tgt_mask is not None:
        add = torch.zeros(tgt_mask.shape).masked_fill(tgt_mask, -1e4)[None, None]
    if tgt_padding_mask is not None:
        kp = torch.zeros(tgt_padding_mask.shape).masked_fill(tgt_padding_mask, -1e4)[:, None, None, :]
        add = kp if add is None else add + kp
    for i in range(n_layers):
        q = f"{p}decoder.layers.{i}."
        s_ = q[:-1]
        a = mha(x, x, state[q + "self_attn.in_proj_weight"], state[q + "self_attn.in_proj_bias"],
                state[q + "self_attn.out_proj.weight"], state[q + "self_attn.out_proj.bias"], nhead, add, bf16, drop(s_ + ".sattn"))
        x = layer_norm(x + _apply(a, drop(s_ + ".drop1")), state[q + "norm1.weight"], state[q + "norm1.bias"])
        c = mha(x, memory, state[q + "multihead_attn.in_proj_weight"], state[q + "multihead_attn.in_proj_bias"],
                state[q + "multihead_attn.out_proj.weight"], state[q + "multihead_attn.out_proj.bias"], nhead, None, bf16,
                drop(s_ + ".cattn"))
        x = layer_norm(x + _apply(c, drop(s_ + ".drop2")), state[q + "norm2.weight"], state[q + "norm2.bias"])
        h = _apply(gelu(linear(x, state[q + "linear1.weight"], state[q + "linear1.bias"], bf16)), drop(s_ + ".ffn"))
        h = linear(h, state[q + "linear2.weight"], state[q + "linear2.bias"], bf16)
        x = layer_norm(x + _apply(h, drop(s_ + ".drop3")), state[q + "norm3.weight"], state[q + "norm3.bias"])
    return linear(x, state[p + "generator.weight"], state[p + "generator.bias"], bf16)


def loss_fn(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """model.py:228-238: fp32, nan_to_num, CE with ignore_index=1 (mean over kept)."""
    lg = torch.nan_to_num(logits.float(), nan=0.0, posinf=1e4, neginf=-1e4)
    lg = lg.reshape(-1, lg.shape[-1])
    y = labels.long().reshape(-1)
    lse = torch.logsumexp(lg, dim=-1)
    nll = lse - lg.gather(1, y[:, None]).squeeze(1)
    keep = y != 1
    return (nll * keep).sum() / keep.sum()


def n_layers_of(state: State, prefix: str) -> int:
    idx = {int(k[len(prefix):].split(".")[0]) for k in state if k.startswith(prefix)}
    return max(idx) + 1


def forward(state: State, cfg: dict, src: torch.Tensor, tgt: torch.Tensor,
            tgt_padding_mask: Optional[torch.Tensor], labels: torch.Tensor, bf16=False, drop=_nodrop) -> dict:
    """``ADTModel.forward`` (model.py:240-258) with ``tgt_mask=None`` as
    ``ADTTrainer.compute_loss`` calls it (train.py:64-70).  Returns every
    intermediate the parity tests compare."""
    nhead = cfg["nhead"]
    mel = o_logmel.logmel(src, cfg["sample_rate"], cfg["win_length"], cfg["time_res"], cfg["n_mels"])
    x = linear(mel, state["project_to_mel.weight"], state["project_to_mel.bias"], bf16)
    memory = encoder(state, x, nhead, n_layers_of(state, "encoder.encoder.layers."), bf16, drop)
    cm = causal_mask(tgt.size(1))
    logits = decoder(state, tgt, memory, nhead, n_layers_of(state, "decoder.decoder.layers."), cm,
                     tgt_padding_mask, bf16, drop)
    return {"logmel": mel, "memory": memory, "logits": logits, "loss": loss_fn(logits, labels)}


def compute_loss(state: State, cfg: dict, batch: dict, bf16=False, drop=_nodrop) -> dict:
    """``ADTTrainer.compute_loss`` (train.py:40-78): teacher-forcing shift and
    padding mask from ``token_lengths``."""
    tokens = torch.as_tensor(batch["tokens"])
    tgt_in, labels = tokens[:, :-1], tokens[:, 1:]
    pad = key_padding_mask(torch.as_tensor(batch["token_lengths"]), tgt_in.size(1))
    return forward(state, cfg, torch.as_tensor(batch["wavs"]), tgt_in, pad, labels, bf16, drop)


def greedy_sample(state: State, cfg: dict, src: torch.Tensor, max_length: int = 1000,
                  start_token: int = 2, end_token: int = 3, bf16=False) -> torch.Tensor:
    """``ADTModel.sample`` (model.py:260-324): encoder once, full decoder over the
    prefix each step, argmax of the last position, finished rows forced to EOS."""
    nhead = cfg["nhead"]
    mel = o_logmel.logmel(src, cfg["sample_rate"], cfg["win_length"], cfg["time_res"], cfg["n_mels"])
    x = linear(mel, state["project_to_mel.weight"], state["project_to_mel.bias"], bf16)
    memory = encoder(state, x, nhead, n_layers_of(state, "encoder.encoder.layers."), bf16)
    nd = n_layers_of(state, "decoder.decoder.layers.")
    B = src.shape[0]
    gen = torch.full((B, 1), start_token, dtype=torch.long)
    finished = torch.zeros(B, dtype=torch.bool)
    for _ in range(max_length - 1):
        logits = decoder(state, gen, memory, nhead, nd, causal_mask(gen.shape[1]), None, bf16)
        nxt = torch.argmax(logits[:, -1, :], dim=-1)
        nxt = torch.where(finished, torch.full_like(nxt, end_token), nxt)
        gen = torch.cat([gen, nxt[:, None]], dim=1)
        finished = finished | (nxt == end_token)
        if bool(finished.all()):
            break
    return gen


def teacher_forced_logits(state: State, cfg: dict, src: torch.Tensor, tokens: torch.Tensor, bf16=False) -> torch.Tensor:
    """Decoder logits ``[B, n, V]`` for a given token prefix (causal mask, no padding mask) -- what ``ADTModel.sample``
    (model.py:300-322) evaluates at each step; lets a test judge a greedy decode produced elsewhere position by position."""
    nhead = cfg["nhead"]
    mel = o_logmel.logmel(src, cfg["sample_rate"], cfg["win_length"], cfg["time_res"], cfg["n_mels"])
    x = linear(mel, state["project_to_mel.weight"], state["project_to_mel.bias"], bf16)
    memory = encoder(state, x, nhead, n_layers_of(state, "encoder.encoder.layers."), bf16)
    return decoder(state, tokens, memory, nhead, n_layers_of(state, "decoder.decoder.layers."), causal_mask(tokens.shape[1]), None, bf16)


# ----------------------------------------------------------------------------- seeded weights
def seeded_state(template: State, seed: int) -> State:
    """Portable random init: numpy PCG64 keyed by ``seed``, walked over the
    state dict in key order.  Matrices ~ N(0, 0.7/sqrt(fan_in)), biases
    ~ N(0, 0.02), LayerNorm weight 1 + N(0, 0.05); constant buffers
    (PE tables, Hann window, mel filterbank) are left as they are."""
    rng = np.random.default_rng(seed)
    out = {}
    for k, v in template.items():
        if "pos_embedding" in k or k.startswith("compute_spectrogram."):
            out[k] = v.clone()
            continue
        shape = tuple(v.shape)
        if v.ndim >= 2:
            std = 0.7 / math.sqrt(shape[-1])
            if "embedding" in k:
                std = 0.7 / math.sqrt(shape[-1])
            a = rng.standard_normal(shape, dtype=np.float32) * np.float32(std)
        elif "norm" in k and k.endswith("weight"):
            a = np.float32(1.0) + rng.standard_normal(shape, dtype=np.float32) * np.float32(0.05)
        else:
            a = rng.standard_normal(shape, dtype=np.float32) * np.float32(0.02)
        out[k] = torch.from_numpy(a.astype(np.float32))
    return out
