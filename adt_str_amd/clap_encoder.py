"""CLAP audio tower on the gfx950 kernels: drop-in for the reference's ``ClapWrapper`` audio path
(``modules/clap_encoder.py:8-54``): ``get_audio_features(list of [1, L] clips @ 48 kHz) -> [B, 512]`` unit vectors.

The reference delegates to ``transformers``: ``ClapProcessor`` (float64 numpy features on the host),
``ClapModel.audio_model`` (HTSAT = Swin transformer over a 256 x 256 folded mel image) and
``ClapModel.audio_projection``.  Here the features come from ``adt_clap_logmel_db_f32`` (K9) and the encoder runs on
``adt_htsat_*`` / ``adt_window_attn_fwd`` / ``adt_patch_merge_ln`` (K10, K11), ``adt_gemm_bf16`` and
``adt_layernorm_fwd``, with bf16 GEMM/attention operands and fp32 accumulation / residual stream.  Weights are taken
from a ``transformers.ClapModel`` state dict (same keys), so a pretrained ``laion/clap-htsat-fused`` checkpoint loads
unchanged.  The text tower is not part of the hot path (never called by the curation pipeline).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
from torch import nn

from . import _ffi
from . import kernels as K
from .clap_frontend import N_FRAMES, N_MELS, ClapLogMel

F32, BF16 = torch.float32, torch.bfloat16
WINDOW = 8


ROWBLOCK_CHANNELS = (96, 192)        # stages whose layers run entirely on the fused row-block kernels (csrc/htsat_fused.hip)
ROWBLOCK_PARTIAL = (384,)            # LN + q|k|v, attention output + residual and LN + fc1 + GELU fused; fc2 stays a plain GEMM


def _frags_rows(w: torch.Tensor) -> torch.Tensor:
    """W [N, K] (N % 32 == 0, K % 16 == 0) -> bf16 [N/32, K/16, 512]: per output tile n and k-step s the 1 KiB MFMA A fragment, lane
    (r, h) = W[32n + r][16s + 8h + j] at position (32h + r) * 8 + j (include/adt_hip.h, K15)."""
    N, K = w.shape
    return w.to(BF16).view(N // 32, 32, K // 16, 2, 8).permute(0, 2, 3, 1, 4).reshape(N // 32, K // 16, 512).contiguous()


def pack_rowblock_weights(mode: int, w1: torch.Tensor, w2: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The weight stream ``adt_htsat_rowblock`` consumes, one fragment after the other (1-D bf16).  ``mode`` 0 / 1: ``w1`` [N, C], tile by
    tile.  ``mode`` 2 (``w1`` = fc1 [4C, C], ``w2`` = fc2 [C, 4C]): the software-pipelined MLP kernel consumes STEPS k = 0 .. 4C/32 + 1, each
    the C/16 fragments of fc1(hidden tile k) interleaved with the C/16 fragments (k-step s2 = 0, 1, channel tile ct) of fc2(hidden tile
    k - 2) -- fragment 2s = fc1, fragment 2(s2 CT + ct) + 1 = fc2, zeros where the tile index is out of range; the hidden units of an
    fc2 k-step come in accumulator order 8 (j>>2) + 4h + (j&3)."""
    if mode not in (2, 3):
        return _frags_rows(w1).reshape(-1)
    C = w1.shape[1]
    NT, CT, KS = w1.shape[0] // 32, C // 32, C // 16
    f1 = _frags_rows(w1)                                                            # [NT, KS, 512]
    f2 = w2.to(BF16).view(CT, 32, NT, 2, 2, 2, 4).permute(2, 3, 0, 5, 1, 4, 6).reshape(NT, 2 * CT, 512)   # [n, (s2, ct), (h, r, jhi, jlo)]
    if mode == 3:                                       # A/B arm: the MLP phase by phase, per hidden tile [fc1 fragments | fc2 fragments]
        return torch.cat([f1, f2], dim=1).reshape(-1).contiguous()
    steps = torch.zeros((NT + 2, 2 * KS, 512), dtype=BF16, device=w1.device)
    steps[:NT, 0::2] = f1
    steps[2:, 1::2] = f2
    return steps.reshape(-1).contiguous()


def pack_attn_block_weights(wqkv: torch.Tensor, bqkv: torch.Tensor, wo: torch.Tensor, heads: int):
    """(weight stream, padded q|k|v bias [heads, 3, 32]) of ``adt_htsat_attn_block``: per head [Wq_h | Wk_h | Wv_h | Wo_h] fragments, every head's 24
    units / inputs padded to 32 with zeros."""
    C = wo.shape[0]
    CT = C // 32
    chunks, bias = [], torch.zeros((heads, 3, 32), dtype=F32, device=wqkv.device)
    for hd in range(heads):
        for which in range(3):
            w = torch.zeros((32, C), dtype=F32, device=wqkv.device)
            w[:24] = wqkv[which * C + 24 * hd: which * C + 24 * hd + 24]
            bias[hd, which, :24] = bqkv[which * C + 24 * hd: which * C + 24 * hd + 24]
            chunks.append(_frags_rows(w).reshape(-1))
        wo_h = torch.zeros((C, 32), dtype=F32, device=wo.device)
        wo_h[:, :24] = wo[:, 24 * hd: 24 * hd + 24]
        # [ct, r, n = 1, s2, jhi, h, jlo] -> [s2, ct, h, r, jhi, jlo]: the inputs of a k-step in accumulator order
        chunks.append(wo_h.to(BF16).view(CT, 32, 1, 2, 2, 2, 4).permute(2, 3, 0, 5, 1, 4, 6).reshape(-1))
    return torch.cat(chunks).contiguous(), bias.contiguous()


def rowblock(mode: int, x: torch.Tensor, wpk: torch.Tensor, n_tiles: int, bias1: torch.Tensor, *, a16: Optional[torch.Tensor] = None,
             ln=None, eps: float = 1e-5, bias2: Optional[torch.Tensor] = None, out16: Optional[torch.Tensor] = None) -> None:
    """One fused half-layer on x [M, C] fp32 (see ``adt_htsat_rowblock``)."""
    M, C = x.shape
    assert x.dtype == F32 and x.is_contiguous() and wpk.dtype == BF16
    _ffi.call("adt_htsat_rowblock", mode, _ffi.dptr(x), M, C, _ffi.dptr(a16) if a16 is not None else None,
              a16.stride(0) if a16 is not None else 0, _ffi.dptr(ln[0]) if ln else None, _ffi.dptr(ln[1]) if ln else None, eps,
              _ffi.dptr(wpk), n_tiles, _ffi.dptr(bias1), _ffi.dptr(bias2) if bias2 is not None else None,
              _ffi.dptr(out16) if out16 is not None else None, out16.stride(0) if out16 is not None else 0, _ffi.current_stream())


def window_bias_layout(bias: torch.Tensor) -> torch.Tensor:
    """[..., 64 (query), 64 (key)] -> the lane-linear order ``adt_window_attn_fwd`` reads: [..., qt 2, kt 2, g 4, h 2, r 32, e 4] with
    query = 32 qt + r and key = 32 kt + 8 g + 4 h + e (one contiguous KiB per wave-load of a (qt, kt, g) piece), **in log2 units**:
    the kernels run their softmax as 2^(scores * scale * log2 e + bias * log2 e - max), so the table is multiplied by log2 e here
    instead of once per score in the kernel."""
    lead = bias.shape[:-2]
    v = (bias.float() * math.log2(math.e)).reshape(*lead, 2, 32, 2, 4, 2, 4)      # [..., qt, r, kt, g, h, e]
    n = len(lead)
    return v.permute(*range(n), n, n + 2, n + 3, n + 4, n + 1, n + 5).contiguous().reshape(*lead, 64, 64)


def window_bias_layout_bf16(bias: torch.Tensor) -> torch.Tensor:
    """The same table as bf16 for ``adt_htsat_layer_block`` at C = 192 (its bias staging takes half the LDS): [..., qt 2, kt 2, gp 2, h 2, r 32,
    gi 2, e 4] with key = 32 kt + 8 (2 gp + gi) + 4 h + e -- one 16-byte piece per lane (h, r) and (qt, kt, gp), in log2 units like the fp32 form."""
    lead = bias.shape[:-2]
    v = (bias.float() * math.log2(math.e)).reshape(*lead, 2, 32, 2, 2, 2, 2, 4)      # [..., qt, r, kt, gp, gi, h, e]
    n = len(lead)
    return v.permute(*range(n), n, n + 2, n + 3, n + 5, n + 1, n + 4, n + 6).contiguous().reshape(*lead, 64, 64).to(BF16)


def _shift_mask(R: int, shift: int) -> torch.Tensor:
    """[nW, 64, 64] additive 0 / -100 mask of a shifted-window layer (ClapAudioLayer.get_attn_mask)."""
    idx = torch.arange(R)
    region = (idx >= R - WINDOW).long() + (idx >= R - shift).long()
    img = region[:, None] * 3 + region[None, :]                                   # [R, R]
    win = img.view(R // WINDOW, WINDOW, R // WINDOW, WINDOW).permute(0, 2, 1, 3).reshape(-1, WINDOW * WINDOW).float()
    diff = win.unsqueeze(1) - win.unsqueeze(2)
    return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))


class HtsatEncoder:
    """Forward-only HTSAT + projection head.  ``sd``: state dict of a ``transformers.ClapModel``."""

    def __init__(self, sd: Dict[str, torch.Tensor], audio_config, device="cuda"):
        self.dev = torch.device(device)
        c = audio_config
        if c.window_size != WINDOW or c.patch_size != 4 or tuple(np.atleast_1d(c.patch_stride).tolist()) not in ((4,), (4, 4)):
            raise NotImplementedError("HTSAT kernels are built for window 8 and 4x4/stride-4 patches")
        self.depths, self.heads = list(c.depths), list(c.num_attention_heads)
        self.C0, self.spec, self.n_mels, self.eps = c.patch_embeds_hidden_size, c.spec_size, c.num_mel_bins, c.layer_norm_eps
        if any(self.C0 * 2 ** s != self.heads[s] * 24 for s in range(len(self.depths))):
            raise NotImplementedError("window attention kernel is built for head_dim 24")
        self.enable_fusion = bool(c.enable_fusion)
        p = "audio_model.audio_encoder."
        g = lambda k: sd[k].detach().to(self.dev, F32).contiguous()
        bn_scale = g(p + "batch_norm.weight") / torch.sqrt(g(p + "batch_norm.running_var") + 1e-5)
        self.bn_scale, self.bn_shift = bn_scale.contiguous(), (g(p + "batch_norm.bias") - g(p + "batch_norm.running_mean") * bn_scale).contiguous()
        self.pe_w = g(p + "patch_embed.proj.weight").reshape(self.C0, 16).contiguous()
        self.pe_b, self.pe_g, self.pe_beta = g(p + "patch_embed.proj.bias"), g(p + "patch_embed.norm.weight"), g(p + "patch_embed.norm.bias")
        self.aff = None
        if self.enable_fusion:
            if c.fusion_type != "aff_2d":
                raise NotImplementedError("only the aff_2d fusion of laion/clap-htsat-fused is built")
            self.aff = self._fold_aff(g, p + "patch_embed.")
        b16 = lambda t: t.to(BF16).contiguous()
        self.stages = []
        R = self.spec // 4
        for s, depth in enumerate(self.depths):
            C, nh = self.C0 * 2 ** s, self.heads[s]
            layers = []
            for i in range(depth):
                q = f"{p}layers.{s}.blocks.{i}."
                a = q + "attention.self."
                table, index = g(a + "relative_position_bias_table"), sd[a + "relative_position_index"].to(self.dev).long()
                bias = table[index.view(-1)].view(64, 64, nh).permute(2, 0, 1).contiguous()             # [nh, 64, 64]
                shift = 0 if (i % 2 == 0 or R <= WINDOW) else WINDOW // 2
                if shift:
                    bias = (bias.unsqueeze(0) + _shift_mask(R, shift).to(self.dev).unsqueeze(1)).contiguous()   # [nW, nh, 64, 64]
                n_bias = bias.shape[0] if shift else 1
                bias16 = window_bias_layout_bf16(bias) if C == 192 else None
                bias = window_bias_layout(bias)
                layers.append(dict(
                    shift=shift, bias=bias, bias16=bias16, n_bias=n_bias,
                    ln1=(g(q + "layernorm_before.weight"), g(q + "layernorm_before.bias")),
                    wqkv=b16(torch.cat([g(a + "query.weight"), g(a + "key.weight"), g(a + "value.weight")], 0)),
                    bqkv=torch.cat([g(a + "query.bias"), g(a + "key.bias"), g(a + "value.bias")], 0).contiguous(),
                    wo=b16(g(q + "attention.output.dense.weight")), bo=g(q + "attention.output.dense.bias"),
                    ln2=(g(q + "layernorm_after.weight"), g(q + "layernorm_after.bias")),
                    w1=b16(g(q + "intermediate.dense.weight")), b1=g(q + "intermediate.dense.bias"),
                    w2=b16(g(q + "output.dense.weight")), b2=g(q + "output.dense.bias")))
                if C in ROWBLOCK_CHANNELS or C in ROWBLOCK_PARTIAL:     # the same weights as the fragment streams of the fused kernels
                    L = layers[-1]
                    L["qkv_pk"] = pack_rowblock_weights(0, L["wqkv"].float()).to(self.dev)
                    L["wo_pk"] = pack_rowblock_weights(1, L["wo"].float()).to(self.dev)
                if C in ROWBLOCK_PARTIAL:
                    L["fc1_pk"] = pack_rowblock_weights(0, L["w1"].float()).to(self.dev)
                    L["mlp_pk"] = pack_rowblock_weights(2, L["w1"].float(), L["w2"].float()).to(self.dev)
                if (C, nh) in ((96, 4), (192, 8), (384, 16)):      # the whole attention half in one launch (C = 192 / 384: round 6, ADT_HTSAT_ATTN_BIG)
                    L["attn_pk"], L["attn_qkvb"] = (t.to(self.dev) for t in pack_attn_block_weights(L["wqkv"].float(), L["bqkv"], L["wo"].float(), nh))
                if C in ROWBLOCK_CHANNELS:
                    L = layers[-1]
                    L["mlp_pk"] = pack_rowblock_weights(2, L["w1"].float(), L["w2"].float()).to(self.dev)
                    L["mlp_pk3"] = pack_rowblock_weights(3, L["w1"].float(), L["w2"].float()).to(self.dev)
                L = layers[-1]
                if "attn_pk" in L and "mlp_pk" in L:
                    # the one-launch halves with the LayerNorm's affine part folded into what follows it (W' = W diag(gamma), b' = b + W beta, from
                    # the fp32 weights, rounded to bf16 once): the kernels then normalise only -- 4 C/16 loads of gamma / beta per token row cost a
                    # wave alone on its SIMD ~10 k cycles of issue per workgroup (profiles/r06/clap_residual_ab.txt).  ADT_HTSAT_FOLD_LN=0: in-kernel affine.
                    wqkv32 = torch.cat([g(a + "query.weight"), g(a + "key.weight"), g(a + "value.weight")], 0)
                    g1, be1 = L["ln1"]
                    L["attn_pkf"], L["attn_qkvbf"] = (t.to(self.dev) for t in pack_attn_block_weights(wqkv32 * g1[None, :], L["bqkv"] + (wqkv32 * be1[None, :]).sum(1),
                                                                                                          L["wo"].float(), nh))
                    w1_32 = g(q + "intermediate.dense.weight")
                    g2, be2 = L["ln2"]
                    L["mlp_pkf"] = pack_rowblock_weights(2, w1_32 * g2[None, :], L["w2"].float()).to(self.dev)
                    L["b1f"] = (L["b1"] + (w1_32 * be2[None, :]).sum(1)).contiguous()           # (element-wise: no BLAS in the product)
            merge = None
            if s < len(self.depths) - 1:
                d = f"{p}layers.{s}.downsample."
                merge = dict(norm=(g(d + "norm.weight"), g(d + "norm.bias")), w=b16(g(d + "reduction.weight")))
                if C == 96:                                # gather + LayerNorm(384) + reduction in one launch (adt_htsat_merge_rowblock)
                    merge["pk"] = pack_rowblock_weights(0, merge["w"].float()).to(self.dev)
                    merge["zero_bias"] = torch.zeros(merge["w"].shape[0], dtype=F32, device=self.dev)
            self.stages.append(dict(C=C, nh=nh, R=R, layers=layers, merge=merge))
            R //= 2
        self.final_ln = (g(p + "norm.weight"), g(p + "norm.bias"))
        self.proj = (b16(g("audio_projection.linear1.weight")), g("audio_projection.linear1.bias"),
                     b16(g("audio_projection.linear2.weight")), g("audio_projection.linear2.bias"))

    def _fold_aff(self, g, pe: str):
        """Weights of mel_conv2d + ClapAudioAFFBlock with the eval-mode BatchNorms folded into the 1x1 convolutions."""
        def conv_bn(prefix, conv, bn):
            w, b = g(f"{prefix}{conv}.weight").flatten(1), g(f"{prefix}{conv}.bias")
            s = g(f"{prefix}{bn}.weight") / torch.sqrt(g(f"{prefix}{bn}.running_var") + 1e-5)
            return (w * s[:, None]).contiguous(), ((b - g(f"{prefix}{bn}.running_mean")) * s + g(f"{prefix}{bn}.bias")).contiguous()
        la, ga = pe + "fusion_model.local_att.", pe + "fusion_model.global_att."
        t = dict(conv_w=g(pe + "mel_conv2d.weight").reshape(self.C0, 48).contiguous(), conv_b=g(pe + "mel_conv2d.bias"))
        t["local_w1"], t["local_b1"] = conv_bn(la, "0", "1")
        t["local_w2"], t["local_b2"] = conv_bn(la, "3", "4")
        t["global_w1"], t["global_b1"] = conv_bn(ga, "1", "2")
        t["global_w2"], t["global_b2"] = conv_bn(ga, "4", "5")
        t["inter"] = t["local_w1"].shape[0]
        return t

    def _fusion_tokens(self, img_global: torch.Tensor, img_local: torch.Tensor, out_rows: torch.Tensor, st) -> None:
        """AFF branch for one clip: img_global [side, side], img_local [3, side, side] -> out_rows [(side/4)^2, C0] (in place)."""
        a = self.aff
        w = _ffi.AffWeights(**{k: _ffi.dptr(v) for k, v in dict(
            proj_w=self.pe_w, proj_b=self.pe_b, conv_w=a["conv_w"], conv_b=a["conv_b"], local_w1=a["local_w1"], local_b1=a["local_b1"],
            local_w2=a["local_w2"], local_b2=a["local_b2"], global_w1=a["global_w1"], global_b1=a["global_b1"], global_w2=a["global_w2"],
            global_b2=a["global_b2"], ln_gamma=self.pe_g, ln_beta=self.pe_beta).items()})
        nbytes = _ffi.load().adt_htsat_fusion_embed_workspace_bytes(self.spec, self.C0)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=self.dev)
        import ctypes
        _ffi.call("adt_htsat_fusion_embed", _ffi.dptr(img_global), _ffi.dptr(img_local), self.spec, ctypes.byref(w), self.eps, self.C0,
                  a["inter"], _ffi.dptr(ws), nbytes, _ffi.dptr(out_rows), st)

    @torch.no_grad()
    def forward(self, mel: torch.Tensor, is_longer: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """mel [B, 1001, 64] (the four fusion channels are identical: clips of at most 10 s) or [B, 4, 1001, 64] fp32,
        is_longer [B] / [B, 1] bool (items routed through the AFF fusion patch embedding, as the feature extractor marks them)
        -> {"pooled": [B, 768], "embedding": [B, 512]} fp32."""
        mel = mel.to(self.dev, F32).contiguous()
        four = mel.dim() == 4
        B, T, M = mel.shape[0], mel.shape[-2], mel.shape[-1]
        long_idx = [] if is_longer is None else torch.as_tensor(is_longer).reshape(-1).nonzero().reshape(-1).tolist()
        if long_idx and self.aff is None:
            raise ValueError("is_longer items need a fusion model (audio_config.enable_fusion)")
        st = _ffi.current_stream()
        side = self.spec
        img = torch.empty((B, side, side), dtype=F32, device=self.dev)
        _ffi.call("adt_htsat_front_f32", _ffi.dptr(mel), mel.stride(0), B, T, M, side * (side // M), side, _ffi.dptr(self.bn_scale),
                  _ffi.dptr(self.bn_shift), _ffi.dptr(img), st)
        n_tok = (side // 4) ** 2
        x = torch.empty((B * n_tok, self.C0), dtype=F32, device=self.dev)
        _ffi.call("adt_htsat_patch_embed", _ffi.dptr(img), B, side, _ffi.dptr(self.pe_w), _ffi.dptr(self.pe_b), _ffi.dptr(self.pe_g),
                  _ffi.dptr(self.pe_beta), self.eps, self.C0, _ffi.dptr(x), None, st)
        for b in long_idx:
            if four:
                loc = torch.empty((3, side, side), dtype=F32, device=self.dev)
                _ffi.call("adt_htsat_front_f32", _ffi.dptr(mel[b, 1:]), T * M, 3, T, M, side * (side // M), side, _ffi.dptr(self.bn_scale),
                          _ffi.dptr(self.bn_shift), _ffi.dptr(loc), st)
            else:
                loc = img[b].unsqueeze(0).expand(3, side, side).contiguous()
            self._fusion_tokens(img[b], loc, x[b * n_tok:(b + 1) * n_tok], st)
        import os
        fused = os.environ.get("ADT_HTSAT_FUSED", "1") != "0"
        for S in self.stages:
            C, nh, R = S["C"], S["nh"], S["R"]
            for L in S["layers"]:
                big = os.environ.get("ADT_HTSAT_ATTN_BIG", "1")            # "1" (default since round 6): C = 192 and 384 too; "384" / "192": that stage only; "0": three launches
                attn_one = C == 96 or (C in (192, 384) and (big == "1" or big == str(C)))
                if fused and "attn_pk" in L and attn_one and os.environ.get("ADT_HTSAT_ATTN", "1") != "0":
                    # the attention half in one launch: only the residual stream touches HBM
                    fold = "attn_pkf" in L and os.environ.get("ADT_HTSAT_FOLD_LN", "1") != "0"
                    if fold and ((C == 384 and os.environ.get("ADT_HTSAT_LAYER384", "1") != "0" and os.environ.get("ADT_HTSAT_MLP384", "1") != "0")
                                 or (C == 192 and os.environ.get("ADT_HTSAT_LAYER192", "1") != "0")
                                 or (C == 96 and os.environ.get("ADT_HTSAT_LAYER96", "1") != "0")):
                        # the whole layer in one launch: the rows stay in the accumulators between the halves (ADT_HTSAT_LAYER384=0: two launches)
                        _ffi.call("adt_htsat_layer_block", _ffi.dptr(x), B, R, C, nh, L["shift"], self.eps, _ffi.dptr(L["attn_pkf"]), _ffi.dptr(L["attn_qkvbf"]),
                                  _ffi.dptr(L["bo"]), _ffi.dptr(L["bias"]), L["n_bias"], 1.0 / math.sqrt(24.0), _ffi.dptr(L["mlp_pkf"]), C // 8,
                                  _ffi.dptr(L["b1f"]), _ffi.dptr(L["b2"]), _ffi.dptr(L["bias16"]) if L["bias16"] is not None else None, st)
                        continue
                    if fold:
                        _ffi.call("adt_htsat_attn_block", _ffi.dptr(x), B, R, C, nh, L["shift"], None, None, self.eps,
                                  _ffi.dptr(L["attn_pkf"]), _ffi.dptr(L["attn_qkvbf"]), _ffi.dptr(L["bo"]), _ffi.dptr(L["bias"]), L["n_bias"],
                                  1.0 / math.sqrt(24.0), st)
                    else:
                        _ffi.call("adt_htsat_attn_block", _ffi.dptr(x), B, R, C, nh, L["shift"], _ffi.dptr(L["ln1"][0]), _ffi.dptr(L["ln1"][1]), self.eps,
                                  _ffi.dptr(L["attn_pk"]), _ffi.dptr(L["attn_qkvb"]), _ffi.dptr(L["bo"]), _ffi.dptr(L["bias"]), L["n_bias"],
                                  1.0 / math.sqrt(24.0), st)
                    if fold and (C != 384 or os.environ.get("ADT_HTSAT_MLP384", "1") != "0"):
                        rowblock(2, x, L["mlp_pkf"], C // 8, L["b1f"], ln=None, eps=self.eps, bias2=L["b2"])
                    elif C != 384:
                        rowblock(2, x, L["mlp_pk"], C // 8, L["b1"], ln=L["ln2"], eps=self.eps, bias2=L["b2"])
                    elif os.environ.get("ADT_HTSAT_MLP384", "1") != "0":
                        rowblock(2, x, L["mlp_pk"], C // 8, L["b1"], ln=L["ln2"], eps=self.eps, bias2=L["b2"])
                    else:
                        h = torch.empty((x.shape[0], 4 * C), dtype=BF16, device=self.dev)
                        rowblock(4, x, L["fc1_pk"], 4 * C // 32, L["b1"], ln=L["ln2"], eps=self.eps, out16=h)
                        K.gemm(h, L["w2"], bias=L["b2"], residual=x, out=x)
                    continue
                if fused and "mlp_pk" in L and "fc1_pk" not in L:
                    # bandwidth-bound stages: three launches per layer, the residual stream is read and written once per half
                    qkv = torch.empty((x.shape[0], 3 * C), dtype=BF16, device=self.dev)
                    rowblock(0, x, L["qkv_pk"], 3 * C // 32, L["bqkv"], ln=L["ln1"], eps=self.eps, out16=qkv)
                    ctx = torch.empty((x.shape[0], C), dtype=BF16, device=self.dev)
                    _ffi.call("adt_window_attn_fwd", _ffi.dptr(qkv), qkv.stride(0), _ffi.dptr(ctx), C, _ffi.dptr(L["bias"]), L["n_bias"], B, R, C,
                              nh, L["shift"], 1.0 / math.sqrt(24.0), st)
                    rowblock(1, x, L["wo_pk"], C // 32, L["bo"], a16=ctx)
                    if os.environ.get("ADT_HTSAT_MLP") == "3":
                        rowblock(3, x, L["mlp_pk3"], C // 8, L["b1"], ln=L["ln2"], eps=self.eps, bias2=L["b2"])
                    else:
                        rowblock(2, x, L["mlp_pk"], C // 8, L["b1"], ln=L["ln2"], eps=self.eps, bias2=L["b2"])
                    continue
                if fused and "fc1_pk" in L:
                    # C = 384: the same three fusions, except that fc2 (K = 4C) runs as a plain GEMM with the residual in its epilogue
                    qkv = torch.empty((x.shape[0], 3 * C), dtype=BF16, device=self.dev)
                    rowblock(0, x, L["qkv_pk"], 3 * C // 32, L["bqkv"], ln=L["ln1"], eps=self.eps, out16=qkv)
                    ctx = torch.empty((x.shape[0], C), dtype=BF16, device=self.dev)
                    _ffi.call("adt_window_attn_fwd", _ffi.dptr(qkv), qkv.stride(0), _ffi.dptr(ctx), C, _ffi.dptr(L["bias"]), L["n_bias"], B, R, C,
                              nh, L["shift"], 1.0 / math.sqrt(24.0), st)
                    rowblock(1, x, L["wo_pk"], C // 32, L["bo"], a16=ctx)
                    if os.environ.get("ADT_HTSAT_MLP384", "1") != "0":
                        # the whole MLP in one launch (round 6: one workgroup per CU on 484 registers): the hidden activation never exists --
                        # 806 MB less HBM traffic per layer at 512 clips, 433 vs 494 us alone, +0.5 % embeds/s inside the tower
                        # (profiles/r06/clap_mlp384_ab.txt; a lone wave per SIMD is bound by its own instruction issue where the two launches
                        # are bound by HBM)
                        rowblock(2, x, L["mlp_pk"], C // 8, L["b1"], ln=L["ln2"], eps=self.eps, bias2=L["b2"])
                        continue
                    h = torch.empty((x.shape[0], 4 * C), dtype=BF16, device=self.dev)
                    rowblock(4, x, L["fc1_pk"], 4 * C // 32, L["b1"], ln=L["ln2"], eps=self.eps, out16=h)
                    K.gemm(h, L["w2"], bias=L["b2"], residual=x, out=x)
                    continue
                _, xn, _, _ = K.layernorm_fwd(x, *L["ln1"], eps=self.eps, want32=False)
                qkv = K.gemm(xn, L["wqkv"], bias=L["bqkv"])
                ctx = torch.empty((x.shape[0], C), dtype=BF16, device=self.dev)
                _ffi.call("adt_window_attn_fwd", _ffi.dptr(qkv), qkv.stride(0), _ffi.dptr(ctx), C, _ffi.dptr(L["bias"]), L["n_bias"], B, R, C,
                          nh, L["shift"], 1.0 / math.sqrt(24.0), st)
                K.gemm(ctx, L["wo"], bias=L["bo"], residual=x, out=x)
                _, xn, _, _ = K.layernorm_fwd(x, *L["ln2"], eps=self.eps, want32=False)
                h = K.gemm(xn, L["w1"], bias=L["b1"], act=1)
                K.gemm(h, L["w2"], bias=L["b2"], residual=x, out=x)
            if S["merge"] is not None and "pk" in S["merge"] and fused and os.environ.get("ADT_HTSAT_MERGE", "1") != "0":
                mg = S["merge"]
                xm = torch.empty((x.shape[0] // 4, mg["w"].shape[0]), dtype=F32, device=self.dev)
                _ffi.call("adt_htsat_merge_rowblock", _ffi.dptr(x), B, R, C, _ffi.dptr(mg["norm"][0]), _ffi.dptr(mg["norm"][1]), 1e-5,
                          _ffi.dptr(mg["pk"]), mg["w"].shape[0] // 32, _ffi.dptr(mg["zero_bias"]), _ffi.dptr(xm), xm.stride(0), st)
                x = xm
            elif S["merge"] is not None:
                m16 = torch.empty((x.shape[0] // 4, 4 * C), dtype=BF16, device=self.dev)
                _ffi.call("adt_patch_merge_ln", _ffi.dptr(x), B, R, C, _ffi.dptr(S["merge"]["norm"][0]), _ffi.dptr(S["merge"]["norm"][1]),
                          1e-5, _ffi.dptr(m16), st)
                x = K.gemm(m16, S["merge"]["w"], out_dtype=F32)
        Cf, Tf = x.shape[1], x.shape[0] // B
        pooled = torch.empty((B, Cf), dtype=F32, device=self.dev)
        pooled16 = torch.empty((B, Cf), dtype=BF16, device=self.dev)
        # final LayerNorm + mean over the clip's tokens in one pass (the normalised rows are never written)
        _ffi.call("adt_ln_mean_tokens", _ffi.dptr(x), B, Tf, Cf, _ffi.dptr(self.final_ln[0]), _ffi.dptr(self.final_ln[1]), 1e-5, _ffi.dptr(pooled),
                  _ffi.dptr(pooled16), st)
        w1, b1, w2, b2 = self.proj
        p1 = K.gemm(pooled16, w1, bias=b1, act=2)
        p2 = K.gemm(p1, w2, bias=b2, out_dtype=F32)
        emb = torch.empty_like(p2)
        _ffi.call("adt_l2_normalize", _ffi.dptr(p2), B, p2.shape[1], _ffi.dptr(emb), st)
        return {"pooled": pooled, "embedding": emb}


class ClapWrapper(nn.Module):
    """``ClapWrapper(model_name, device, sample_rate)`` of the reference (clap_encoder.py:9-19), audio path only.

    ``clap_model``: an already constructed ``transformers.ClapModel`` (offline / random weights); otherwise
    ``ClapModel.from_pretrained(model_name)`` is used exactly like the reference does."""

    def __init__(self, model_name: str, device, sample_rate: int, clap_model=None, **kwargs):
        super().__init__()
        if clap_model is None:
            from transformers import ClapModel
            clap_model = ClapModel.from_pretrained(model_name)
        if sample_rate != 48000:
            raise ValueError("the CLAP audio tower runs at 48 kHz (ClapFeatureExtractor.sampling_rate)")
        self.device, self.sample_rate, self.config = torch.device(device), sample_rate, clap_model.config
        self.encoder = HtsatEncoder(clap_model.state_dict(), clap_model.config.audio_config, self.device)
        self.features = ClapLogMel(self.device)

    @torch.no_grad()
    def get_audio_features(self, audios: Sequence[torch.Tensor], is_longer: Optional[torch.Tensor] = None) -> torch.Tensor:
        """list of [1, L] (or [L]) 48 kHz clips -> [B, 512] L2-normalised embeddings (clap_encoder.py:21-24, 30-54).

        ``is_longer``: the flags the reference's ClapProcessor would hand to the model.  Its feature extractor marks every clip
        longer than 10 s (those carry three random crops + a shrunk mel, ``np.random.choice``) and, when a batch has none, ONE
        RANDOM clip (feature_extraction_clap.py:347-350, ``np.random.randint``); those draws are part of the reference's result, so
        seed ``numpy`` / pass the recorded flags to reproduce it.  Default (None): the same rules with the same ``numpy`` draws in the
        same order."""
        flat = [a.reshape(-1) for a in audios]
        if all(a.numel() <= 10 * self.sample_rate for a in flat):
            mel, auto_longer = self.features.mel(flat), torch.zeros(len(flat), dtype=torch.bool)      # one mel per clip: no 4-channel copy
        else:
            mel, auto_longer = self.features.features(flat)       # [B, 4, 1001, 64]: long clips carry three crops + a shrunk mel
        if is_longer is None:
            is_longer = auto_longer
            if len(flat) and self.encoder.enable_fusion and not bool(is_longer.any()):
                is_longer = is_longer.clone()
                is_longer[np.random.randint(0, len(flat))] = True                                    # feature_extraction_clap.py:347-350
        return self.encoder.forward(mel, is_longer)["embedding"]

    @torch.no_grad()
    def _get_audio_features(self, input_features: Optional[torch.Tensor] = None, is_longer: Optional[torch.Tensor] = None,
                            attention_mask=None, output_attentions=None, output_hidden_states=None, return_dict=None) -> torch.Tensor:
        """``input_features [B, 4, 1001, 64]`` + ``is_longer [B, 1]`` (what ClapProcessor returns) -> [B, 512]
        (clap_encoder.py:30-54; the attention / hidden-state outputs of the reference signature are never read by its callers)."""
        if output_attentions or output_hidden_states:
            raise NotImplementedError("attention maps / hidden states are not produced by the fused kernels")
        return self.encoder.forward(input_features, is_longer)["embedding"]

    def get_text_features(self, text):
        raise NotImplementedError("the text tower is outside the MI355X hot path (the curation pipeline never calls it)")


def random_init_clap_model(seed: int = 0):
    """A ``transformers.ClapModel`` with the architecture of ``laion/clap-htsat-fused`` (fused HTSAT audio tower, ``aff_2d``) and
    random weights, with non-trivial BatchNorm statistics so the eval-mode affine is exercised.  For benchmarks and end-to-end
    drivers on machines without the pretrained checkpoint (there is no network here); pass it as ``ClapWrapper(clap_model=...)``."""
    import torch
    from transformers import ClapConfig, ClapModel
    torch.manual_seed(seed)
    model = ClapModel(ClapConfig(audio_config={"enable_fusion": True, "fusion_type": "aff_2d"})).eval()
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0.0, 0.5)
                m.running_var.uniform_(0.5, 2.0)
                m.weight.normal_(1.0, 0.1)
                m.bias.normal_(0.0, 0.1)
    return model
