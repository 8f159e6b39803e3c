"""ADT network on the gfx950 kernels: drop-in for the reference's ``ADTModel``
(``model.py:193-324``), ``Encoder`` (``:100-135``), ``Decoder`` (``:138-190``),
``TokenEmbedding_plain`` (``:42-49``) and ``PositionalEncoding`` (``:52-65``).

The module tree (and therefore every state-dict key and the default
initialisation) is the reference's: ``nn.TransformerEncoder/Decoder`` objects are
instantiated as *parameter containers* only.  All arithmetic -- forward,
backward, loss -- is done by ``_Engine`` below through the C ABI
(``libadt_hip.so``): bf16 MFMA GEMMs with fused bias/GELU/residual epilogues,
flash-style attention, fp32 LayerNorm / cross-entropy.  fp32 master weights,
bf16 operand copies (refreshed when a parameter changes), fp32 gradients; this
is what ``bf16`` autocast does in the reference (``train.py:233-234``).

``forward(src, tgt, tgt_mask, tgt_padding_mask, labels) -> loss`` keeps the
reference signature and plugs into autograd through one ``autograd.Function``
whose backward is the hand-written backward pass, so HF ``Trainer`` /
``accelerate`` work unchanged.  ``loss_and_grads`` is the same computation
without autograd, writing gradients into one flat fp32 buffer (what
``adt_str_amd.trainer`` and ``bench.py`` use, with a single bucketed all-reduce).
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional

import torch
import torch.nn as nn
from transformers import PretrainedConfig, PreTrainedModel

from . import kernels as K
from .frontend import ComputeMelSpectrogram

F32, BF16 = torch.float32, torch.bfloat16
PRECISIONS = ("bf16", "fp32", "bf16x3")


class ADTModelConfig(PretrainedConfig):
    """Same fields and defaults as the reference's ``ADTModelConfig`` (config.py:81-119)."""
    model_type = "adt_model"

    def __init__(self, input_sec: float = 0.0, time_res: float = 0.0, win_length: int = 0, sample_rate: int = 0,
                 enc_layers: int = 0, dec_layers: int = 0, nhead: int = 0, d_query: int = 0, dropout: float = 0.0,
                 tgt_vocab_size: int = 0, enc_lr: float = 0.0, dec_lr: float = 0.0, plain: bool = False,
                 n_mels: int = 0, **kwargs):
        super().__init__(**kwargs)
        self.input_sec, self.time_res, self.win_length, self.sample_rate = input_sec, time_res, win_length, sample_rate
        self.enc_layers, self.dec_layers, self.nhead, self.d_query = enc_layers, dec_layers, nhead, d_query
        self.dropout, self.tgt_vocab_size, self.enc_lr, self.dec_lr = dropout, tgt_vocab_size, enc_lr, dec_lr
        self.plain, self.n_mels = plain, n_mels


# ----------------------------------------------------------------------------- parameter containers
class TokenEmbedding_plain(nn.Module):
    def __init__(self, vocab_size, emb_size):
        super().__init__()
        self.embedding = nn.Embedding(vocab_size, emb_size)
        self.emb_size = emb_size


class PositionalEncoding(nn.Module):
    """Sinusoidal table ``pos_embedding[1, maxlen, d]`` (model.py:55-62), persistent buffer."""

    def __init__(self, emb_size: int, maxlen: int = 2048):
        super().__init__()
        freq = torch.exp(-torch.arange(0, emb_size, 2) * math.log(10000) / emb_size)
        pos = torch.arange(0, maxlen).reshape(maxlen, 1)
        table = torch.zeros((maxlen, emb_size))
        table[:, 0::2] = torch.sin(pos * freq)
        table[:, 1::2] = torch.cos(pos * freq)
        self.register_buffer("pos_embedding", table.unsqueeze(0))


def _no_call(self, *a, **k):
    raise RuntimeError("container module: the ADT network runs through adt_str_amd's HIP engine, not nn.Module.forward")


class Encoder(nn.Module):
    def __init__(self, enc_layers, d_query, nhead, ffn_hid_dim, dropout):
        super().__init__()
        self.num_features = d_query * nhead
        self.dense_layer = nn.Linear(self.num_features, self.num_features, bias=False)
        self.positional_encoding = PositionalEncoding(self.num_features)
        self.layer_norm = nn.LayerNorm(self.num_features, elementwise_affine=True)
        self.dropout_layer = nn.Dropout(p=dropout)
        layer = nn.TransformerEncoderLayer(d_model=self.num_features, nhead=nhead, dim_feedforward=ffn_hid_dim,
                                           dropout=dropout, activation="gelu", batch_first=True, norm_first=False)
        self.encoder = nn.TransformerEncoder(layer, num_layers=enc_layers, enable_nested_tensor=False)

    forward = _no_call


class Decoder(nn.Module):
    def __init__(self, dec_layers, d_query, nhead, ffn_hid_dim, tgt_vocab_size, dropout, plain=False):
        super().__init__()
        if not plain:
            raise NotImplementedError("only plain token embedding is supported (model.py:42-49; configs use plain: true)")
        self.num_features = d_query * nhead
        self.tgt_tok_emb = TokenEmbedding_plain(tgt_vocab_size, self.num_features)
        self.positional_encoding = PositionalEncoding(self.num_features)
        self.dropout_layer = nn.Dropout(p=dropout)
        self.generator = nn.Linear(self.num_features, tgt_vocab_size)
        layer = nn.TransformerDecoderLayer(d_model=self.num_features, nhead=nhead, dim_feedforward=ffn_hid_dim,
                                           dropout=dropout, activation="gelu", batch_first=True, norm_first=False)
        self.decoder = nn.TransformerDecoder(layer, num_layers=dec_layers)

    forward = _no_call


# ----------------------------------------------------------------------------- engine
class _Lin:
    """A linear layer's fp32 parameters plus its bf16 operands W [N,K] and W^T [K,N]."""

    def __init__(self, name: str, weight: nn.Parameter, bias: Optional[nn.Parameter]):
        self.name, self.weight, self.bias = name, weight, bias
        self.w16 = self.wt16 = None

    def refresh(self):
        self.w16, self.wt16 = K.cast_bf16(self.weight.data.contiguous(), True, True)

    @property
    def b(self):
        return None if self.bias is None else self.bias.data


class _Engine:
    """``precision``: ``"bf16"`` (default; bf16 GEMM / attention operands, fp32 accumulate = the reference's bf16 autocast),
    ``"fp32"`` (the exact parity arm: fp32 activations end to end on the f32-input MFMA kernels of ``csrc/precise.hip``) or
    ``"bf16x3"`` (the fast parity arm: the same fp32 activations and the same kernels, every product taken as three bf16 MFMAs on
    hi / lo splits of the fp32 operands -- ~1e-5 relative per product instead of 6e-8, on the 16 x faster matrix pipe).  Both parity
    arms meet BASELINE's "logits within 1e-3 rel-tol of the CPU reference".  Default from ``ADT_PRECISION``."""

    def __init__(self, model: "ADTModel", precision: Optional[str] = None):
        self.m = model
        cfg = model.config
        precision = (precision or os.environ.get("ADT_PRECISION", "bf16")).lower()
        if precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {PRECISIONS}, not {precision!r}")
        self.precision = precision
        self.fp32 = precision != "bf16"                   # fp32 activations / operands (both parity arms)
        self.products = "bf16x3" if precision == "bf16x3" else "f32"     # how the fp32-operand kernels multiply (kernels.f32_products)
        self.adt = F32 if self.fp32 else BF16            # dtype of GEMM / attention operands and stored activations
        self.H, self.d, self.dh = cfg.nhead, cfg.nhead * cfg.d_query, cfg.d_query
        if cfg.d_query != 128 and not (self.fp32 and cfg.d_query in (16, 32, 64)):
            raise NotImplementedError("the bf16 attention kernels are built for head_dim (d_query) = 128 "
                                      "(the fp32 parity path also takes 16, 32 and 64)")
        self.scale = 1.0 / math.sqrt(cfg.d_query)
        self.V = cfg.tgt_vocab_size
        if self.V % 8 or cfg.n_mels % 8:
            raise NotImplementedError("tgt_vocab_size and n_mels must be multiples of 8 (16-byte bf16 rows for the GEMMs)")
        self.lins: Dict[str, _Lin] = {}
        named = dict(model.named_parameters())

        def lin(prefix, wname="weight", bname="bias"):
            w = named[f"{prefix}.{wname}" if wname == "weight" else f"{prefix}.{wname}"]
            b = named.get(f"{prefix}.{bname}")
            self.lins[prefix + "." + wname] = _Lin(prefix + "." + wname, w, b)
            return self.lins[prefix + "." + wname]

        self.proj = lin("project_to_mel")
        self.dense = lin("encoder.dense_layer")
        self.gen = lin("decoder.generator")
        self.enc, self.dec = [], []
        for i in range(cfg.enc_layers):
            p = f"encoder.encoder.layers.{i}"
            self.enc.append(dict(p=p, sa=lin(p + ".self_attn", "in_proj_weight", "in_proj_bias"), sa_o=lin(p + ".self_attn.out_proj"),
                                 l1=lin(p + ".linear1"), l2=lin(p + ".linear2")))
        for i in range(cfg.dec_layers):
            p = f"decoder.decoder.layers.{i}"
            self.dec.append(dict(p=p, sa=lin(p + ".self_attn", "in_proj_weight", "in_proj_bias"), sa_o=lin(p + ".self_attn.out_proj"),
                                 ca=lin(p + ".multihead_attn", "in_proj_weight", "in_proj_bias"),
                                 ca_o=lin(p + ".multihead_attn.out_proj"), l1=lin(p + ".linear1"), l2=lin(p + ".linear2")))
        self.named = named
        self._versions = None
        self._cast_table = None
        self.gflat: Optional[torch.Tensor] = None
        self.G: Dict[str, torch.Tensor] = {}
        self._rq, self._rq_arena = None, None
        self.grad_ready_hook = None      # callable(lo, hi): flat gradient range [lo, hi) is final (used to overlap all-reduce)
        self.drop_p = 0.0                # active dropout probability of the current pass (0 in eval)
        self.drop_seed = 0               # changes every training step; masks are regenerated from it in the backward
        self.drop_base = 0               # what seed_dropout() started the counter from (rank-specific); drop_seed - drop_base = passes drawn
        self._sites: Dict[str, int] = {}
        self.generation = 0              # bumped by every pass that writes the gradient buffers (see _ADTLossFn.backward)
        self.hf_reducer = None           # HF Trainer + DDP: GradReducer driven by this engine's backward (trainer.install_engine_reduction)
        self.hf_sync = lambda: True      # ... and whether the wrapper wants this pass reduced (False inside DDP.no_sync())
        self.reduced_generation = -1     # generation whose gradients were averaged by hf_reducer (DDP's comm hook then passes them through)
        self._wg_pending = []            # deferred decoder weight-gradient products (see _wgrad)

    def seed_dropout(self, seed: int, rank: int = 0):
        """Start the per-step dropout counter from a value derived from (experiment seed, data-parallel rank): ranks draw
        different masks (as nn.Dropout does under DDP), a different seed gives a different run, and a checkpoint that stores
        ``drop_seed`` resumes the sequence instead of replaying it from step 1."""
        self.drop_seed = self.drop_base = K.mix32(K.mix32(int(seed)) ^ ((int(rank) + 1) * 0x9E3779B9 & 0xFFFFFFFF)) & 0x3FFFFFFF

    def D(self, site: str):
        """(p, key) of a named dropout site for the current step, or None when dropout is off."""
        if self.drop_p <= 0.0:
            return None
        idx = self._sites.setdefault(site, len(self._sites) + 1)
        return K.drop_site(self.drop_p, self.drop_seed, idx)

    # ---- parameter plumbing -------------------------------------------------------
    def P(self, name):
        return self.named[name].data

    def refresh_weights(self, force=False):
        """Re-derive the bf16 operands when any parameter was modified in place (optimizer step, load_state_dict)."""
        K.set_f32_products(self.products)                 # (every pass of the engine starts here or in _backward)
        if self.fp32:                                     # the fp32 masters are the operands: nothing to derive
            for l in self.lins.values():
                if not l.weight.data.is_contiguous():
                    raise RuntimeError(f"{l.name}: the fp32 path reads the master weight in place; it must be contiguous")
                l.w16, l.wt16 = l.weight.data, None
            if self.products == "bf16x3":
                # the split-bf16 arm's large products run on the persistent bf16 kernels over [hi | lo] planes: every weight (and its
                # transpose, the data gradients' operand) is split once per parameter version, not per GEMM
                ver = tuple(p._version for p in self.named.values()) + tuple(l.weight.data.data_ptr() for l in self.lins.values())
                if force or ver != self._versions or K._x3_owner != id(self):
                    K.x3_register_weights([l.weight.data for l in self.lins.values()])
                    K._x3_owner = id(self)
                    self._versions = ver
            return
        ver = tuple(p._version for p in self.named.values()) + (str(next(iter(self.named.values())).device),)
        if force or ver != self._versions:
            lins = list(self.lins.values())
            ws = [l.weight.data for l in lins]
            if any(not w.is_contiguous() for w in ws):
                for l in lins:
                    l.refresh()
            else:                                         # one launch for every weight (pointers are stable between steps)
                if self._cast_table is None or not self._cast_table.matches(ws):
                    self._cast_table = K.CastTable(ws)
                    for l, y, yt in zip(lins, self._cast_table.y, self._cast_table.y_t):
                        l.w16, l.wt16 = y, yt
                self._cast_table.run()
            self._kv_cat = None                           # concatenated cross-attention K|V operands follow the weights
            self._versions = tuple(p._version for p in self.named.values()) + (ver[-1],)

    def _kv_operands(self):
        """The cross-attention K | V projection weights of ALL decoder layers as one operand: (W [layers*2d, d] bf16, bias [layers*2d] fp32,
        W^T arranged [d, layers*2d] bf16 for the memory gradient).  Built once per weight refresh (the bf16 copies only change there), not per
        step / micro-batch."""
        if getattr(self, "_kv_cat", None) is None:
            d = self.d
            self._kv_cat = (torch.cat([L["ca"].w16[d:] for L in self.dec]), torch.cat([L["ca"].b[d:] for L in self.dec]),
                            torch.cat([L["ca"].wt16[:, d:] for L in self.dec], dim=1))
        return self._kv_cat

    def grad_buffers(self):
        """One flat fp32 gradient buffer with a view per parameter, in parameter order."""
        dev = next(iter(self.named.values())).device
        n = sum(p.numel() for p in self.named.values())
        if self.gflat is None or self.gflat.device != dev or self.gflat.numel() != n:
            self.gflat = torch.zeros(n, dtype=F32, device=dev)
            off = 0
            self.G = {}
            for name, p in self.named.items():
                self.G[name] = self.gflat[off:off + p.numel()].view_as(p)
                off += p.numel()
        return self.gflat, self.G

    # ---- precision plumbing (the two modes share every line of the forward / backward below) --------------------
    def _operand(self, x32):
        """fp32 activation -> GEMM operand (a bf16 copy, or the tensor itself on the fp32 path)."""
        return x32 if self.fp32 else K.cast_bf16(x32)[0]

    def _gemm_dual(self, a, w, **kw):
        """GEMM whose result is needed as the fp32 residual stream AND as the next GEMM's operand -> (x32, x16)."""
        if self.fp32:
            x32 = K.gemm(a, w, **kw)
            return x32, x32
        x16 = torch.empty((a.shape[0], w.shape[0]), dtype=BF16, device=a.device)
        return K.gemm(a, w, out_dtype=F32, aux_bf16_out=x16, **kw), x16

    def _dgrad(self, dy, lin: _Lin, lo=None, hi=None, **kw):
        """Data gradient ``dy @ W[lo:hi]`` (bf16: NT GEMM against the transposed copy; fp32: NN GEMM against the master)."""
        if self.fp32:
            return K.gemm(dy, lin.w16[lo:hi], b_kn=True, **kw)
        return K.gemm(dy, lin.wt16[:, lo:hi], **kw)

    def _wgrad(self, dy, x, out, defer=False):
        """Weight gradient ``out = dy.T @ x``.  ``defer``: queue it for one grouped launch (`_flush_wgrads`) -- the decoder's
        products are too small to fill the chip one at a time (K = B * T rows)."""
        ok = (defer and not self.fp32 and not os.environ.get("ADT_NO_GROUPED_WGRAD") and dy.shape[0] % 64 == 0 and dy.shape[1] % 8 == 0 and x.shape[1] % 8 == 0 and
              dy.stride(0) % 8 == 0 and x.stride(0) % 8 == 0 and out.stride(0) % 4 == 0 and
              dy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0 and out.data_ptr() % 16 == 0)
        if ok:
            self._wg_pending.append((dy, x, out))
        else:
            K.gemm(dy, x, trans=True, out=out)

    def _flush_wgrads(self):
        if self._wg_pending:
            K.gemm_tn_grouped(self._wg_pending)
            self._wg_pending = []

    def _ln(self, x, g, b, want32=True, drop=None):
        if self.fp32:
            y32, _, mean, rstd = K.layernorm_fwd(x, g, b, want32=True, want16=False, drop=drop)
            return (y32 if want32 else None), y32, mean, rstd
        return K.layernorm_fwd(x, g, b, want32=want32, drop=drop)

    def _lean_ln(self) -> bool:
        """LayerNorms inside the layer stacks write only their bf16 output; the fp32 residual is rebuilt by the consuming GEMM's
        epilogue (bf16 path; ADT_NO_RES_LN=1 keeps the fp32 outputs, for A/B runs)."""
        return not self.fp32 and not os.environ.get("ADT_NO_RES_LN")

    def _ln_bwd(self, *a, **kw):
        return K.layernorm_bwd(*a, branch_dtype=self.adt, **kw)

    def _embed(self, tokens, emb, pe, scale, drop=None):
        if self.fp32:
            y32, _ = K.embed_pe_fwd(tokens, emb, pe, scale, want16=False, drop=drop)
            return y32, y32
        return K.embed_pe_fwd(tokens, emb, pe, scale, drop=drop)

    # ---- forward pieces -------------------------------------------------------------
    def _encoder_fwd(self, src, save: Optional[list]):
        m, d, H = self.m, self.d, self.H
        mel = m.compute_spectrogram(src)                                   # K1: [B, S, n_mels] fp32
        B, S, n_mels = mel.shape
        M = B * S
        mel16 = self._operand(mel.view(M, n_mels))
        x0 = K.gemm(mel16, self.proj.w16, bias=self.proj.b)                # project_to_mel (model.py:249)
        pe = self.m.encoder.positional_encoding.pos_embedding[0]
        x32, x16 = self._gemm_dual(x0, self.dense.w16, residual=pe, res_row_mod=S,
                                   drop=self.D("enc.pe"), drop_after_residual=True)   # dense + PE + dropout (model.py:130-132)
        if save is not None:
            save.append(dict(mel16=mel16, x0=x0, B=B, S=S))
        # The residual stream between two LayerNorms is read exactly once, by the epilogue of the next residual GEMM: that epilogue
        # rebuilds LayerNorm(y) from the saved pre-LayerNorm tensor and its row statistics (adt_gemm_epilogue.res_ln_*), so the
        # LayerNorms write only their bf16 output -- 4 bytes per element less HBM traffic each.  `res` is what the next residual
        # GEMM adds: {"residual": x32} or {"residual": y, "residual_ln": (mean, rstd, gamma, beta)}.
        lean = self._lean_ln()
        res = dict(residual=x32)
        for li, L in enumerate(self.enc):
            p = L["p"]
            qkv = K.gemm(x16, L["sa"].w16, bias=L["sa"].b)
            attn, lse = K.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], B, H, S, S, self.scale, drop=self.D(p + ".attn"), head_dim=self.dh,
                                   save_bits=save is not None)
            y1 = K.gemm(attn, L["sa_o"].w16, bias=L["sa_o"].b, out_dtype=F32, drop=self.D(p + ".drop1"), **res)
            g1, b1 = self.P(p + ".norm1.weight"), self.P(p + ".norm1.bias")
            x1_32, x1_16, mean1, rstd1 = self._ln(y1, g1, b1, want32=not lean)
            res = dict(residual=y1, residual_ln=(mean1, rstd1, g1, b1)) if lean else dict(residual=x1_32)
            u = torch.empty((M, L["l1"].w16.shape[0]), dtype=self.adt, device=src.device)
            h = K.gemm(x1_16, L["l1"].w16, bias=L["l1"].b, act=1, act_grad_out=u, drop=self.D(p + ".ffn"))
            y2 = K.gemm(h, L["l2"].w16, bias=L["l2"].b, out_dtype=F32, drop=self.D(p + ".drop2"), **res)
            g2, b2 = self.P(p + ".norm2.weight"), self.P(p + ".norm2.bias")
            last = li + 1 == len(self.enc)                                 # the final LayerNorm reads this one's fp32 output
            x2_32, x2_16, mean2, rstd2 = self._ln(y2, g2, b2, want32=last or not lean)
            res = dict(residual=y2, residual_ln=(mean2, rstd2, g2, b2)) if lean and not last else dict(residual=x2_32)
            if save is not None:
                save.append(dict(x16=x16, qkv=qkv, attn=attn, lse=lse, y1=y1, mean1=mean1, rstd1=rstd1, x1_16=x1_16, u=u, h=h,
                                 y2=y2, mean2=mean2, rstd2=rstd2))
            x32, x16 = x2_32, x2_16
        _, mem16, meanf, rstdf = self._ln(x32, self.P("encoder.layer_norm.weight"), self.P("encoder.layer_norm.bias"),
                                                 want32=False, drop=self.D("enc.final"))      # LN + dropout (model.py:134)
        if save is not None:
            save.append(dict(x32=x32, mean=meanf, rstd=rstdf))
        return mem16, B, S

    def _decoder_fwd(self, tgt, mem16, B, S, key_len, save: Optional[list]):
        d, H = self.d, self.H
        tgt = tgt.long().contiguous()
        T = tgt.shape[1]
        Md = B * T
        emb = self.P("decoder.tgt_tok_emb.embedding.weight")
        pe = self.m.decoder.positional_encoding.pos_embedding[0]
        x32, x16 = self._embed(tgt, emb, pe, math.sqrt(d), drop=self.D("dec.emb"))      # model.py:171-172
        dev = tgt.device
        # The cross-attention K | V projections of the memory for ALL decoder layers in one GEMM (N = layers * 2d): the layers'
        # slices of one [B*S, layers*2d] buffer.  What this buys is in the backward: the memory's gradient becomes ONE product
        # with K = layers * 2d instead of a chain of GEMMs that each re-read and re-write the fp32 [B*S, d] accumulator.
        kv_all = None
        if not self.fp32 and len(self.dec) > 1 and not os.environ.get("ADT_NO_KV_BATCH"):
            wkv, bkv, _ = self._kv_operands()
            kv_all = K.gemm(mem16, wkv, bias=bkv)
        lean = self._lean_ln()
        res = dict(residual=x32)
        for li, L in enumerate(self.dec):
            p = L["p"]
            qkv = K.gemm(x16, L["sa"].w16, bias=L["sa"].b)
            sa, lse_s = K.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], B, H, T, T, self.scale, causal=True, key_len=key_len,
                                   drop=self.D(p + ".sattn"), head_dim=self.dh, save_bits=save is not None)
            y1 = K.gemm(sa, L["sa_o"].w16, bias=L["sa_o"].b, out_dtype=F32, drop=self.D(p + ".drop1"), **res)
            g1, b1 = self.P(p + ".norm1.weight"), self.P(p + ".norm1.bias")
            x1_32, x1_16, mean1, rstd1 = self._ln(y1, g1, b1, want32=not lean)
            res = dict(residual=y1, residual_ln=(mean1, rstd1, g1, b1)) if lean else dict(residual=x1_32)
            ca_w, ca_b = L["ca"].w16, L["ca"].b
            qc = K.gemm(x1_16, ca_w[:d], bias=ca_b[:d])
            kvc = kv_all[:, 2 * d * li:2 * d * (li + 1)] if kv_all is not None else K.gemm(mem16, ca_w[d:], bias=ca_b[d:])
            ca, lse_c = K.attn_fwd(qc, kvc[:, :d], kvc[:, d:], B, H, T, S, self.scale, drop=self.D(p + ".cattn"), head_dim=self.dh,
                                   save_bits=save is not None)
            y2 = K.gemm(ca, L["ca_o"].w16, bias=L["ca_o"].b, out_dtype=F32, drop=self.D(p + ".drop2"), **res)
            g2, b2 = self.P(p + ".norm2.weight"), self.P(p + ".norm2.bias")
            x2_32, x2_16, mean2, rstd2 = self._ln(y2, g2, b2, want32=not lean)
            res = dict(residual=y2, residual_ln=(mean2, rstd2, g2, b2)) if lean else dict(residual=x2_32)
            u = torch.empty((Md, L["l1"].w16.shape[0]), dtype=self.adt, device=dev)
            h = K.gemm(x2_16, L["l1"].w16, bias=L["l1"].b, act=1, act_grad_out=u, drop=self.D(p + ".ffn"))
            y3 = K.gemm(h, L["l2"].w16, bias=L["l2"].b, out_dtype=F32, drop=self.D(p + ".drop3"), **res)
            g3, b3 = self.P(p + ".norm3.weight"), self.P(p + ".norm3.bias")
            x3_32, x3_16, mean3, rstd3 = self._ln(y3, g3, b3, want32=not lean)
            res = dict(residual=y3, residual_ln=(mean3, rstd3, g3, b3)) if lean else dict(residual=x3_32)
            if save is not None:
                save.append(dict(x16=x16, qkv=qkv, sa=sa, lse_s=lse_s, y1=y1, mean1=mean1, rstd1=rstd1, x1_16=x1_16, qc=qc, kvc=kvc,
                                 ca=ca, lse_c=lse_c, y2=y2, mean2=mean2, rstd2=rstd2, x2_16=x2_16, u=u, h=h, y3=y3, mean3=mean3,
                                 rstd3=rstd3))
            x32, x16 = x3_32, x3_16
        logits = K.gemm(x16, self.gen.w16, bias=self.gen.b, out_dtype=F32)  # generator (model.py:190), fp32 for the loss
        if save is not None:
            save.append(dict(xo16=x16, T=T, tgt=tgt, kv_batched=kv_all is not None))
        return logits

    @staticmethod
    def key_len_from_mask(tgt_padding_mask, B, T, device):
        """The kernels take per-sequence valid lengths; the reference passes the bool mask
        ``arange(T) >= length`` (utils/utils.py:36-43), i.e. padding is always a suffix."""
        if tgt_padding_mask is None:
            return None
        if tgt_padding_mask.dtype != torch.bool:
            raise NotImplementedError("float key-padding masks are not supported; pass the bool mask of create_mask_plain")
        return (T - tgt_padding_mask.to(device).sum(dim=1)).to(torch.int32).contiguous()

    # ---- loss + gradients ---------------------------------------------------------------
    def loss_and_grads(self, src, tgt, tgt_padding_mask, labels, want_grads=True, return_logits=False):
        """Full training step arithmetic.  Gradients land in ``self.G`` (views of ``self.gflat``)."""
        self.refresh_weights()
        self.drop_p = float(self.m.config.dropout) if self.m.training else 0.0
        if self.drop_p > 0.0:
            self.drop_seed += 1
        dev = src.device
        B, T = tgt.shape
        key_len = self.key_len_from_mask(tgt_padding_mask, B, T, dev)
        enc_save: Optional[list] = [] if want_grads else None
        dec_save: Optional[list] = [] if want_grads else None
        mem16, B, S = self._encoder_fwd(src, enc_save)
        logits = self._decoder_fwd(tgt, mem16, B, S, key_len, dec_save)
        loss, dlogits = K.cross_entropy(logits, labels.long().reshape(-1), ignore_index=1, want_grad=want_grads, grad_dtype=self.adt)
        out = {"loss": loss[0]}
        if return_logits:
            out["logits"] = logits.view(B, T, -1)
            out["memory"] = mem16
        if not want_grads:
            return out
        self.generation += 1
        if self.fp32 or os.environ.get("ADT_NO_REDUCE_QUEUE"):
            self._rq = None
            self._backward(dlogits, mem16, B, S, key_len, enc_save, dec_save)
        else:
            # the ~55 small second-stage reductions of the bias / LayerNorm gradients: one launch per layer instead
            if self._rq_arena is None or self._rq_arena.device != dev:
                self._rq_arena = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
            with K.reduce_queue(self._rq_arena) as self._rq:
                self._backward(dlogits, mem16, B, S, key_len, enc_save, dec_save)
            self._rq = None
        return out

    def _backward(self, dlogits, mem16, B, S, key_len, enc_save, dec_save):
        K.set_f32_products(self.products)
        d, H = self.d, self.H
        _, G = self.grad_buffers()
        tail = dec_save[-1]
        T, tgt, xo16 = tail["T"], tail["tgt"], tail["xo16"]
        # generator
        self._wgrad(dlogits, xo16, G["decoder.generator.weight"], defer=True)
        K.colsum(dlogits, out=G["decoder.generator.bias"])
        dx32 = self._dgrad(dlogits, self.gen, out_dtype=F32)
        dmem32 = None
        dkv_all = torch.empty((mem16.shape[0], 2 * d * len(self.dec)), dtype=mem16.dtype, device=mem16.device) if tail["kv_batched"] else None
        for li in range(len(self.dec) - 1, -1, -1):
            L, s = self.dec[li], dec_save[li]
            p = L["p"]
            dy3_32, dy3_16 = self._ln_bwd(dx32, s["y3"], self.P(p + ".norm3.weight"), s["mean3"], s["rstd3"], G[p + ".norm3.weight"],
                                             G[p + ".norm3.bias"], G[p + ".linear2.bias"], dx16_drop=self.D(p + ".drop3"))
            du = self._dgrad(dy3_16, L["l2"], act_grad=s["u"], colsum_out=G[p + ".linear1.bias"])
            self._wgrad(dy3_16, s["h"], G[p + ".linear2.weight"], defer=True)
            self._wgrad(du, s["x2_16"], G[p + ".linear1.weight"], defer=True)
            dx2_32 = self._dgrad(du, L["l1"], residual=dy3_32, out_dtype=F32)
            dy2_32, dy2_16 = self._ln_bwd(dx2_32, s["y2"], self.P(p + ".norm2.weight"), s["mean2"], s["rstd2"], G[p + ".norm2.weight"],
                                             G[p + ".norm2.bias"], G[p + ".multihead_attn.out_proj.bias"], dx16_drop=self.D(p + ".drop2"))
            dca = self._dgrad(dy2_16, L["ca_o"])
            self._wgrad(dy2_16, s["ca"], G[p + ".multihead_attn.out_proj.weight"], defer=True)
            dqc = torch.empty_like(s["qc"])
            dkvc = dkv_all[:, 2 * d * li:2 * d * (li + 1)] if dkv_all is not None else torch.empty_like(s["kvc"])
            gw, gb = G[p + ".multihead_attn.in_proj_weight"], G[p + ".multihead_attn.in_proj_bias"]
            K.attn_bwd(s["qc"], s["kvc"][:, :d], s["kvc"][:, d:], s["ca"], dca, s["lse_c"], dqc, dkvc[:, :d], dkvc[:, d:], B, H, T, S,
                       self.scale, drop=self.D(p + ".cattn"), bias_grad=gb, head_dim=self.dh)
            self._wgrad(dqc, s["x1_16"], gw[:d], defer=True)
            K.gemm(dkvc, mem16, trans=True, out=gw[d:])
            if dkv_all is not None:
                pass                                                   # one K = layers * 2d product after the loop
            elif dmem32 is None:
                dmem32 = self._dgrad(dkvc, L["ca"], d, None, out_dtype=F32)
            else:
                self._dgrad(dkvc, L["ca"], d, None, residual=dmem32, out=dmem32)
            dx1_32 = self._dgrad(dqc, L["ca"], None, d, residual=dy2_32, out_dtype=F32)
            dy1_32, dy1_16 = self._ln_bwd(dx1_32, s["y1"], self.P(p + ".norm1.weight"), s["mean1"], s["rstd1"], G[p + ".norm1.weight"],
                                             G[p + ".norm1.bias"], G[p + ".self_attn.out_proj.bias"], dx16_drop=self.D(p + ".drop1"))
            dsa = self._dgrad(dy1_16, L["sa_o"])
            self._wgrad(dy1_16, s["sa"], G[p + ".self_attn.out_proj.weight"], defer=True)
            dqkv = torch.empty_like(s["qkv"])
            q = s["qkv"]
            K.attn_bwd(q[:, :d], q[:, d:2 * d], q[:, 2 * d:], s["sa"], dsa, s["lse_s"], dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:],
                       B, H, T, T, self.scale, causal=True, key_len=key_len, drop=self.D(p + ".sattn"),
                       bias_grad=G[p + ".self_attn.in_proj_bias"], head_dim=self.dh)
            self._wgrad(dqkv, s["x16"], G[p + ".self_attn.in_proj_weight"], defer=True)
            dx32 = self._dgrad(dqkv, L["sa"], residual=dy1_32, out_dtype=F32)
            self._flush_reductions()
        if dkv_all is not None:
            dmem32 = K.gemm(dkv_all, self._kv_operands()[2], out_dtype=F32)
        K.embed_bwd(tgt, dx32, math.sqrt(d), G["decoder.tgt_tok_emb.embedding.weight"], drop=self.D("dec.emb"), f32=self.fp32)
        self._flush_wgrads()
        self._ready("decoder.")
        # encoder
        fin = enc_save[-1]
        dx32, _ = self._ln_bwd(dmem32, fin["x32"], self.P("encoder.layer_norm.weight"), fin["mean"], fin["rstd"],
                                  G["encoder.layer_norm.weight"], G["encoder.layer_norm.bias"], None, want16=False,
                                  dy_drop=self.D("enc.final"))
        dx16 = None
        for li in range(len(self.enc) - 1, -1, -1):
            L, s = self.enc[li], enc_save[li + 1]
            p = L["p"]
            dy2_32, dy2_16 = self._ln_bwd(dx32, s["y2"], self.P(p + ".norm2.weight"), s["mean2"], s["rstd2"], G[p + ".norm2.weight"],
                                             G[p + ".norm2.bias"], G[p + ".linear2.bias"], dx16_drop=self.D(p + ".drop2"))
            du = self._dgrad(dy2_16, L["l2"], act_grad=s["u"], colsum_out=G[p + ".linear1.bias"])
            K.gemm(dy2_16, s["h"], trans=True, out=G[p + ".linear2.weight"])
            K.gemm(du, s["x1_16"], trans=True, out=G[p + ".linear1.weight"])
            dx1_32 = self._dgrad(du, L["l1"], residual=dy2_32, out_dtype=F32)
            dy1_32, dy1_16 = self._ln_bwd(dx1_32, s["y1"], self.P(p + ".norm1.weight"), s["mean1"], s["rstd1"], G[p + ".norm1.weight"],
                                             G[p + ".norm1.bias"], G[p + ".self_attn.out_proj.bias"], dx16_drop=self.D(p + ".drop1"))
            dattn = self._dgrad(dy1_16, L["sa_o"])
            K.gemm(dy1_16, s["attn"], trans=True, out=G[p + ".self_attn.out_proj.weight"])
            dqkv = torch.empty_like(s["qkv"])
            q = s["qkv"]
            K.attn_bwd(q[:, :d], q[:, d:2 * d], q[:, 2 * d:], s["attn"], dattn, s["lse"], dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:],
                       B, H, S, S, self.scale, drop=self.D(p + ".attn"), bias_grad=G[p + ".self_attn.in_proj_bias"], head_dim=self.dh)
            K.gemm(dqkv, s["x16"], trans=True, out=G[p + ".self_attn.in_proj_weight"])
            if li == 0 and not self.fp32:
                dx16 = torch.empty((dqkv.shape[0], d), dtype=BF16, device=dqkv.device)
            dx32 = self._dgrad(dqkv, L["sa"], residual=dy1_32, out_dtype=F32, aux_bf16_out=dx16,
                               drop=self.D("enc.pe") if li == 0 else None, drop_after_residual=True)   # layer 0: back through the PE dropout
            if li == 0 and self.fp32:
                dx16 = dx32
            self._ready(p + ".")
        head = enc_save[0]
        K.gemm(dx16, head["x0"], trans=True, out=G["encoder.dense_layer.weight"])
        dx0 = self._dgrad(dx16, self.dense)
        K.gemm(dx0, head["mel16"], trans=True, out=G["project_to_mel.weight"])
        K.colsum(dx0, out=G["project_to_mel.bias"])
        self._ready("encoder.dense_layer.", "encoder.layer_norm.", "project_to_mel.")

    def _flush_reductions(self):
        if self._rq is not None:
            self._rq.flush()

    def _ready(self, *prefixes):
        self._flush_reductions()            # the segment's queued bias / LayerNorm gradient reductions, before anyone reads it
        if self.grad_ready_hook is None:
            return
        for pre in prefixes:
            lo, hi = self.flat_range(pre)
            self.grad_ready_hook(lo, hi)

    def flat_range(self, prefix: str):
        """[lo, hi) of the flat parameter / gradient buffers covered by parameters whose name starts with ``prefix``."""
        off, lo, hi = 0, None, None
        for name, p in self.named.items():
            if name.startswith(prefix):
                lo = off if lo is None else lo
                hi = off + p.numel()
            off += p.numel()
        if lo is None:
            raise KeyError(prefix)
        return lo, hi

    # ---- inference --------------------------------------------------------------------------
    @torch.no_grad()
    def encode(self, src):
        self.refresh_weights()
        return self._encoder_fwd(src, None)

    @torch.no_grad()
    def decode_logits(self, tgt, mem16, B, S, key_len=None):
        K.set_f32_products(self.products)
        return self._decoder_fwd(tgt, mem16, B, S, key_len, None).view(B, tgt.shape[1], -1)

    @torch.no_grad()
    def greedy_decode_cached(self, mem16, B, S, max_length: int, start_token: int, end_token: int, sync_every: int = 16,
                             use_graph: bool = True):
        """KV-cached greedy decoding (hot-path row f1): the same arithmetic as running ``_decoder_fwd`` over the whole
        prefix at every step (model.py:300-322), but each step pushes ONE position through the decoder.

        Per layer the packed in_proj output of every position decoded so far stays in a ``[B, max_length, 3d]`` cache
        (the attention kernel reads K/V from it in place; positions not written yet are zeros and sit behind the
        key-padding length, i.e. get the reference's additive -1e4), and the cross-attention K/V of the encoder memory
        are projected once instead of once per step.  A step is ~50 small launches, so it is launch-bound: all of its
        state (position, token, lengths, finished flags) lives in device buffers updated in place, which makes every
        step the same work list -- captured once as a HIP graph and replayed (``use_graph``; measured on MI355X at B = 8:
        0.32-0.34 ms/step replayed = the GPU time of the ~70 small kernels, 0.68 ms/step eager (the host's launch rate),
        0.89 ms/step for the full-prefix recompute at 256 tokens; round 2, before the skinny-M GEMM and the single-query
        attention kernel: 0.68-0.83 ms/step replayed).  The host
        looks at the "all rows finished" flag every ``sync_every`` steps (the reference syncs every step).
        Returns ``[B, n]`` tokens, n as the reference would stop."""
        K.set_f32_products(self.products)
        d, H, dev = self.d, self.H, mem16.device
        Tmax = int(max_length)
        emb = self.P("decoder.tgt_tok_emb.embedding.weight")
        pe = self.m.decoder.positional_encoding.pos_embedding[0]
        if Tmax > pe.shape[0]:
            raise ValueError(f"max_length {Tmax} exceeds the positional table ({pe.shape[0]})")
        caches = [torch.zeros((B, Tmax, 3 * d), dtype=self.adt, device=dev) for _ in self.dec]
        kvcs = [K.gemm(mem16, L["ca"].w16[d:], bias=L["ca"].b[d:]) for L in self.dec]
        gen = torch.full((B, Tmax), end_token, dtype=torch.long, device=dev)
        gen[:, 0] = start_token
        st = dict(tok=gen[:, :1].clone(), t=torch.zeros(1, dtype=torch.long, device=dev),
                  klen=torch.ones(B, dtype=torch.int32, device=dev), finished=torch.zeros(B, dtype=torch.bool, device=dev),
                  done_at=torch.full((1,), Tmax, dtype=torch.long, device=dev))
        scale_e = math.sqrt(d)

        # LayerNorm -> projection pairs in one launch (adt_ln_gemm_bf16): every LayerNorm output of the step feeds exactly one GEMM as
        # its operand and one later GEMM as the residual, so the step's twelve LayerNorm launches fold into their consumers
        fuse_ln = not self.fp32 and B <= 64 and d % 128 == 0 and d <= 1024 and not os.environ.get("ADT_NO_LN_GEMM")

        def step():
            t = st["t"]
            x32, x16 = self._embed(st["tok"], emb, pe.index_select(0, t), scale_e)         # PE row t
            qkv, y3 = None, None
            for li, (L, cache, kvc) in enumerate(zip(self.dec, caches, kvcs)):
                p = L["p"]
                if qkv is None:
                    qkv = K.gemm(x16, L["sa"].w16, bias=L["sa"].b)                         # [B, 3d] of position t
                cache.index_copy_(1, t, qkv.unsqueeze(1))
                flat = cache.view(B * Tmax, 3 * d)
                sa, _ = K.attn_fwd(qkv[:, :d], flat[:, d:2 * d], flat[:, 2 * d:], B, H, 1, Tmax, self.scale, key_len=st["klen"], head_dim=self.dh)
                y1 = K.gemm(sa, L["sa_o"].w16, bias=L["sa_o"].b, residual=x32, out_dtype=F32)
                if fuse_ln:
                    qc, x1_32 = K.ln_gemm(y1, self.P(p + ".norm1.weight"), self.P(p + ".norm1.bias"), L["ca"].w16[:d], bias=L["ca"].b[:d])
                else:
                    x1_32, x1_16, _, _ = self._ln(y1, self.P(p + ".norm1.weight"), self.P(p + ".norm1.bias"))
                    qc = K.gemm(x1_16, L["ca"].w16[:d], bias=L["ca"].b[:d])
                ca, _ = K.attn_fwd(qc, kvc[:, :d], kvc[:, d:], B, H, 1, S, self.scale, head_dim=self.dh)
                y2 = K.gemm(ca, L["ca_o"].w16, bias=L["ca_o"].b, residual=x1_32, out_dtype=F32)
                if fuse_ln:
                    h, x2_32 = K.ln_gemm(y2, self.P(p + ".norm2.weight"), self.P(p + ".norm2.bias"), L["l1"].w16, bias=L["l1"].b, act=1)
                else:
                    x2_32, x2_16, _, _ = self._ln(y2, self.P(p + ".norm2.weight"), self.P(p + ".norm2.bias"))
                    h = K.gemm(x2_16, L["l1"].w16, bias=L["l1"].b, act=1)
                y3 = K.gemm(h, L["l2"].w16, bias=L["l2"].b, residual=x2_32, out_dtype=F32)
                n3w, n3b = self.P(p + ".norm3.weight"), self.P(p + ".norm3.bias")
                if not fuse_ln:
                    x32, x16, _, _ = self._ln(y3, n3w, n3b)
                    qkv = None
                elif li + 1 < len(self.dec):                                               # norm3 -> the next layer's in-projection
                    nxt_l = self.dec[li + 1]
                    qkv, x32 = K.ln_gemm(y3, n3w, n3b, nxt_l["sa"].w16, bias=nxt_l["sa"].b)
            if fuse_ln:                                                                    # the last norm3 -> generator (model.py:190)
                logits, _ = K.ln_gemm(y3, n3w, n3b, self.gen.w16, bias=self.gen.b, out_dtype=F32, want_x32=False)
            else:
                logits = K.gemm(x16, self.gen.w16, bias=self.gen.b, out_dtype=F32)
            # argmax + finished / end-token logic + the step's counters (t, klen, tok, the column of `gen`): one launch
            K.greedy_step(logits, st["finished"], gen, t, st["tok"], st["klen"], st["done_at"], end_token)

        n_steps = Tmax - 1
        done = 0
        graph = None
        if use_graph and n_steps >= 24:               # capture costs ~10 ms; a replayed step saves ~0.35 ms of host launch time
            cur = torch.cuda.current_stream()
            side = torch.cuda.Stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                step()                                                                     # step 0, eager: one-time host setup happens here
            cur.wait_stream(side)
            done = 1
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                step()
        while done < n_steps:
            if graph is not None:
                graph.replay()
            else:
                step()
            done += 1
            if done % sync_every == 0 and int(st["done_at"].item()) < Tmax:
                break
        n = min(Tmax, int(st["done_at"].item()))
        return gen[:, :n].clone()


class _ADTLossFn(torch.autograd.Function):
    """Bridges the hand-written backward into autograd: forward runs forward+backward of the
    engine (gradients are a by-product of the fused cross-entropy kernel), backward hands the
    parameter gradients to autograd scaled by the incoming gradient."""

    @staticmethod
    def forward(ctx, engine, src, tgt, pad_mask, labels, *params):
        # Under DistributedDataParallel (the reference's ``accelerate launch train.py <yaml>``) the engine's own backward pass drives the
        # gradient all-reduce: ``trainer.install_engine_reduction`` hooks a GradReducer to the segments as they finish, so the collectives
        # run under the remaining backward kernels instead of after all of them (DDP only sees the gradients when autograd hands them over,
        # all 132 at once).  What autograd / DDP then receive is already the mean over the ranks; DDP's comm hook passes it through.
        red = engine.hf_reducer
        sync = red is not None and engine.hf_sync()
        if red is not None:
            red.enabled = sync
        out = engine.loss_and_grads(src, tgt, pad_mask, labels, want_grads=True)
        if sync:
            red.finish()
            engine.reduced_generation = engine.generation
        ctx.engine, ctx.generation = engine, engine.generation
        return out["loss"].clone()

    @staticmethod
    def backward(ctx, g):
        eng = ctx.engine
        if ctx.generation != eng.generation:
            raise RuntimeError("the engine's gradient buffers were overwritten by a later forward pass before this loss was "
                               "back-propagated (two forwards, then backward): call backward() right after each forward, as "
                               "HF Trainer.training_step does")
        # ONE pass over the flat gradient buffer (276 MB at setting-1) into a fresh buffer, then a view per parameter -- not 132
        # multiplies / allocations.  The copy cannot be skipped: autograd may keep what it is handed as ``p.grad`` (or add the
        # next micro-batch into it under accumulation) while the engine's own buffer is overwritten by the next forward pass.
        scaled = eng.gflat * g.reshape(()).to(eng.gflat.dtype)
        grads, off = [], 0
        for p in eng.named.values():
            grads.append(scaled[off:off + p.numel()].view(p.shape))
            off += p.numel()
        return (None, None, None, None, None) + tuple(grads)


class ADTModel(PreTrainedModel):
    config_class = ADTModelConfig

    def __init__(self, config: ADTModelConfig) -> None:
        super().__init__(config)
        self.config = config
        ffn = int(config.d_query * config.nhead * 4)
        self.encoder = Encoder(config.enc_layers, config.d_query, config.nhead, ffn, config.dropout)
        self.decoder = Decoder(config.dec_layers, config.d_query, config.nhead, ffn, config.tgt_vocab_size, config.dropout,
                               plain=config.plain)
        self.compute_spectrogram = ComputeMelSpectrogram(config.sample_rate, config.win_length, config.time_res, config.n_mels)
        self.project_to_mel = nn.Linear(config.n_mels, int(config.d_query * config.nhead))
        self._engine_obj: Optional[_Engine] = None
        self._precision: Optional[str] = None          # None: ADT_PRECISION or "bf16"
        self._drop_seed_args = None                    # (seed, rank) re-applied whenever the engine is rebuilt (.to(device))

    @property
    def engine(self) -> _Engine:
        if self._engine_obj is None:
            self._engine_obj = _Engine(self, self._precision)
            if self._drop_seed_args is not None:
                self._engine_obj.seed_dropout(*self._drop_seed_args)
        return self._engine_obj

    def seed_dropout(self, seed: int, rank: int = 0) -> "ADTModel":
        """Key the dropout masks by (experiment seed, data-parallel rank); survives ``.to(device)`` (which rebuilds the engine)."""
        self._drop_seed_args = (int(seed), int(rank))
        if self._engine_obj is not None:
            self._engine_obj.seed_dropout(seed, rank)
        return self

    def set_precision(self, precision: str) -> "ADTModel":
        """``"bf16"`` (throughput default), ``"fp32"`` or ``"bf16x3"`` (the parity arms, see ``_Engine``)."""
        if precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {PRECISIONS}, not {precision!r}")
        self._precision, self._engine_obj = precision, None
        return self

    def _apply(self, fn, *a, **k):                 # .to(device) / .float(): parameters are replaced -> rebuild the engine
        r = super()._apply(fn, *a, **k)
        self._engine_obj = None
        return r

    def forward(self, src, tgt, tgt_mask, tgt_padding_mask, labels):
        """Scalar CE loss (model.py:240-258).  ``tgt_mask`` must be None or the causal mask."""
        if tgt_mask is not None and tgt_mask.shape != (tgt.shape[1], tgt.shape[1]):
            raise ValueError("tgt_mask must be the [T, T] causal mask (or None)")
        eng = self.engine
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return _ADTLossFn.apply(eng, src, tgt, tgt_padding_mask, labels, *eng.named.values())
        return eng.loss_and_grads(src, tgt, tgt_padding_mask, labels, want_grads=False)["loss"]

    @torch.no_grad()
    def sample(self, src, src_mask=None, tgt_mask=None, max_length: int = 1000, start_token: int = 2, end_token: int = 3,
               use_cache: bool = True):
        """Greedy decoding (model.py:260-324): encoder once, argmax of the last position, finished rows pinned to
        ``end_token``, stop when every row is finished.  ``use_cache=True`` (default) decodes one position per step
        against per-layer K/V caches; ``use_cache=False`` runs the full decoder over the prefix at every step exactly
        like the reference (kept as the parity arm)."""
        if not self.config.plain:
            raise NotImplementedError("Non-plain mode is not implemented")
        self.eval()
        eng = self.engine
        mem16, B, S = eng.encode(src)
        if use_cache:
            return eng.greedy_decode_cached(mem16, B, S, max_length, start_token, end_token)
        gen = torch.full((B, 1), start_token, dtype=torch.long, device=src.device)
        finished = torch.zeros(B, dtype=torch.bool, device=src.device)
        for _ in range(max_length - 1):
            logits = eng.decode_logits(gen, mem16, B, S)
            nxt = torch.argmax(logits[:, -1, :], dim=-1)
            nxt = torch.where(finished, torch.full_like(nxt, end_token), nxt)
            gen = torch.cat([gen, nxt.unsqueeze(1)], dim=1)
            finished = finished | (nxt == end_token)
            if bool(torch.all(finished)):
                break
        return gen
