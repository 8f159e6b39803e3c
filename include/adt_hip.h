/* adt_hip.h -- C ABI of libadt_hip.so: the MI355X (gfx950) hot path of ADT_STR.
 *
 * The reference (pier-maker92/ADT_STR) is pure Python and has no FFI layer; the
 * calls below replace the third-party native code its Python reaches (torchaudio
 * STFT/mel, torch.nn.Transformer*, the per-note Python mixer loop).  Each entry
 * cites the reference call site it stands behind.  The Python host
 * (adt_str_amd/) binds these with ctypes; INTEGRATION.md shows the stub a
 * reference maintainer would add.
 *
 * Conventions
 *   - return 0 on success, a negative ADT_E* code on failure; the message is
 *     available from adt_last_error() (thread-local).  Never throws or aborts.
 *   - every pointer is a DEVICE pointer on the current HIP device unless the
 *     parameter name starts with h_; all buffers are owned by the caller
 *     (PyTorch).  Kernels never allocate; scratch is passed in.
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream).  All
 *     work is asynchronous on it; no device synchronisation inside.
 *   - tensors are dense row-major; leading dimensions are explicit where a
 *     caller may pass a view.
 */
#ifndef ADT_HIP_H
#define ADT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADT_OK       0
#define ADT_EINVAL  -1   /* bad argument (null pointer, negative size)        */
#define ADT_ESHAPE  -2   /* shape the kernels do not support                  */
#define ADT_EHIP    -3   /* HIP runtime error (launch failed, no device ...)  */

/* Dropout site: element (row, col) of a tensor with last dimension L has index row * even(L) + col; elements 2m and 2m + 1
 * share hash(m, key) and each is kept iff its 16-bit half >= round(p * 65536), then scaled by 1/(1-p)
 * (adt_str_amd/csrc/dropout.h; oracle/dropout.py mirrors it).  Replaces nn.Dropout / the SDPA dropout of the reference
 * (model.py:116,132,134,156,172 and inside nn.Transformer*Layer); masks are never stored, the
 * backward entry points regenerate them from the same (p, key).  p == 0 or a null pointer: off. */
typedef struct adt_dropout { float p; uint32_t key; } adt_dropout;

/* ABI version (18): bumped whenever a signature below changes or entries are added. */
int adt_version(void);

/* Message of the last failing call on this thread ("" if none). */
const char* adt_last_error(void);

/* Testing aid: occupy n_wg workgroup slots with lds_bytes of LDS each for about `micros` microseconds on `stream` -- the
 * stand-in for a collective's kernel (RCCL under data parallelism) in single-GPU tests: the persistent GEMMs take tiles from
 * work counters, so CUs held by another stream's kernel delay no tile, and a workgroup that cannot be placed at first lands
 * on a CU one of its siblings has left (tests/test_gemm_gpu.py, tools/exp_occupy.py). */
int adt_debug_occupy(int32_t n_wg, int32_t lds_bytes, int32_t micros, void* stream);

/* ---------------------------------------------------------------------------
 * K1  fused STFT -> power -> mel -> log -> clamp -> scale -> trim
 *
 * Replaces ComputeMelSpectrogram.forward, reference model.py:81-97
 * (torchaudio.transforms.MelSpectrogram built at model.py:71-78, then
 * log(x+1e-10) :91, clamp(-23,12) :92, (x+23)/35 :93, permute+trim :95-97).
 *
 *   wave      [n_clips, ld_wave] fp32, n_samples valid per row
 *   window    [n_fft] fp32                      (state dict: ...spectrogram.window)
 *   mel_meta  [n_mels][4] int32 = {first_bin, n_bins, offset into mel_w, 0}:
 *             the banded (CSR) form of the filterbank fb[n_fft/2+1, n_mels]
 *             (state dict: ...mel_scale.fb); mel_w holds the non-zero weights.
 *             Precondition (device memory, not checked by the entry point): first_bin in
 *             [0, n_fft/2], 0 <= n_bins <= 127, first_bin + n_bins <= n_fft/2 + 1,
 *             offset + n_bins <= mel_nnz.  A band that violates it is clamped into range by
 *             the kernel (no out-of-range access; that band's output is then undefined).
 *   out       [n_clips, n_out, n_mels] fp32, fully overwritten:
 *             out[b,f,j] = (clamp(log(mel[b, frame_lo+f, j] + log_eps), lo, hi) - lo) / (hi - lo)
 *
 * Frame t covers reflect-padded samples [t*hop - n_fft/2, t*hop + n_fft/2)
 * (center=True).  Supported: n_fft == 2048, n_mels <= 128 and a multiple of 4,
 * n_samples > n_fft/2.
 */
int adt_logmel_f32(const float* wave, int64_t n_clips, int64_t n_samples, int64_t ld_wave,
                   int32_t n_fft, int32_t hop, int32_t frame_lo, int32_t n_out,
                   const float* window, const int32_t* mel_meta, const float* mel_w,
                   int32_t n_mels, int32_t mel_nnz,
                   float log_eps, float clamp_lo, float clamp_hi,
                   float* out, void* stream);

/* ---------------------------------------------------------------------------
 * K2  batched one-shot drum mixer
 *
 * Replaces the per-note Python loop of SynthDrum.__call__ (reference
 * modules/synthetiser.py:255-292), drum_rendering (:214-239) and
 * VolumeMixer.instrument_mixer / _normalize_audio (:142-156) for a whole batch
 * of clips.  All random draws (timbre choice :192-202, mixup :217) and the
 * velocity->volume law (:204-212) are evaluated by the host and arrive as
 * explicit per-note fields, so the call is a pure function of its inputs.
 *
 *   bank, bank_off   flat one-shot bank: shot i = bank[bank_off[i] .. bank_off[i+1])
 *   notes            n_notes records, clip by clip (clip c owns
 *                    notes[clip_note_off[c] .. clip_note_off[c+1])); inside a clip
 *                    they are grouped by track in track order (track = order of
 *                    first appearance of the pitch, synthetiser.py:146-147) and
 *                    keep note order inside a track
 *   clip_len[c]      rendered length W_c of clip c (synthetiser.py:262-263,243);
 *                    samples >= W_c are written as 0 (collate padding)
 *   clip_gain[c]     volume of the clip's maximum velocity (synthetiser.py:266,292)
 *   out              [n_clips, ld_out] fp32, columns [0, width) fully overwritten:
 *                    wav = sum_tracks gain_t * sum_{notes of t} vol * o / max|o|,
 *                    o = main*(1-mixup) + mixup*sub (zero-padded to equal length),
 *                    out = wav / max|wav| * clip_gain; a clip without notes is all 0
 *   ws               scratch of adt_mix_workspace_bytes(n_notes, n_clips) bytes
 */
typedef struct adt_note {
  int32_t start;            /* first output sample: int(onset * sample_rate)   */
  int32_t main_shot;        /* index into bank_off                              */
  int32_t sub_shot;
  int32_t track;            /* 0-based track id inside the clip                 */
  float   one_minus_mixup;  /* fp32(1 - mixup)                                  */
  float   mixup;            /* fp32(mixup)                                      */
  float   vol;              /* velocity -> volume                               */
  float   track_gain;       /* per-class volume of the note's track             */
} adt_note;

size_t adt_mix_workspace_bytes(int64_t n_notes, int64_t n_clips);

int adt_mix_render_f32(const float* bank, const int64_t* bank_off, int64_t n_shots,
                       const adt_note* notes, int64_t n_notes, const int32_t* clip_note_off,
                       const int32_t* clip_len, const float* clip_gain, int64_t n_clips, int64_t width,
                       float* out, int64_t ld_out, void* ws, size_t ws_bytes, void* stream);

/* K14: the same with the reference's optional FX chain (VolumeMixer._add_fx, synthetiser.py:121-137,154-155) applied to the
 * un-normalised mix of the clips whose fx[c].flags != 0, in the reference's order Reverb -> Compressor -> Limiter, before the
 * peak normalisation.  The host draws which effects and their parameters (BoardChain, synthetiser.py:30-87); the processing
 * restates what pedalboard's JUCE effects do to a mono signal (juce::Reverb::processMono, juce::dsp::Compressor,
 * juce::dsp::Limiter with pedalboard's default 100 ms release).  fx: DEVICE array of n_clips records, or null (= adt_mix_render_f32).
 * sample_rate must be in [12544, 96000]. */
typedef struct adt_fx_params {
  int32_t flags;                /* bit 0 reverb, bit 1 compressor, bit 2 limiter */
  float room_size, damping, wet_level, dry_level, width;
  float c_threshold_db, c_ratio, c_attack_ms, c_release_ms;
  float l_threshold_db, l_release_ms;
} adt_fx_params;
int adt_mix_render_fx_f32(const float* bank, const int64_t* bank_off, int64_t n_shots,
                          const adt_note* notes, int64_t n_notes, const int32_t* clip_note_off,
                          const int32_t* clip_len, const float* clip_gain, int64_t n_clips, int64_t width,
                          const adt_fx_params* fx, int32_t sample_rate,
                          float* out, int64_t ld_out, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------
 * K3/K5  bf16 MFMA GEMM with fused epilogue (fp32 accumulate)
 *
 * Replaces the GEMMs of every nn.Linear on the path -- project_to_mel
 * (reference model.py:224), Encoder.dense_layer (:111), Decoder.generator (:157),
 * and in_proj / out_proj / linear1 / linear2 inside nn.TransformerEncoderLayer /
 * nn.TransformerDecoderLayer (:118-127, :159-168) -- forward and backward.
 *
 *   trans = 0:  C[M,N] = A[M,K] . B[N,K]^T        (y = x W^T;  dx = dy . (W^T)^T)
 *   trans = 1:  C[M,N] = A[K,M]^T . B[K,N]        (dW = dy^T x; K = number of rows)
 * A, B are bf16 row-major with leading dimensions lda/ldb (elements), rows
 * 16-byte aligned and a multiple of 8 elements long.
 *
 * Epilogue, applied in this order to z = alpha * acc:
 *   + bias[col]                          (fp32, may be null)
 *   * gelu'(gelu_grad_of[row,col])       (bf16 pre-activation u; dgrad through GELU)
 *   pre_act_out[row,col] = bf16(z)       (saved pre-activation, may be null)
 *   act == 1: z = gelu_erf(z)            (exact erf GELU, activation="gelu"); act == 2: z = max(z, 0) (ReLU)
 *   act_grad_mode == 1 changes what is saved for / read by the backward of "dropout(gelu(z))": the forward call (act == 1,
 *   pre_act_out given) stores bf16(gelu'(z) * keep) -- the derivative times the dropout factor of this very epilogue --
 *   instead of z, and GELU sees the fp32 z; the backward call multiplies by gelu_grad_of[row,col] as stored (no erf, no
 *   mask hashing: pass no dropout there).  One extra bf16 rounding on the factor, well inside the bf16 path's tolerance.
 *   dropout (drop.p > 0, drop_after_residual == 0), element index row*N + col
 *   + residual[row % res_row_mod, col]   (fp32; res_row_mod == 0: plain row) --
 *                                         residual stream, or the sinusoidal PE
 *                                         table with res_row_mod = frames per clip
 *   dropout (drop.p > 0, drop_after_residual != 0)
 *   C = out_fp32 ? z : bf16(z);  aux_bf16_out[row,col] = bf16(z) as well when given
 *   (an fp32 residual-stream output plus the bf16 operand of the next GEMM in one pass)
 * trans = 1 may split K across workgroups; partial fp32 slabs go to `ws`
 * (adt_gemm_workspace_bytes) and are summed in slab order (reproducible).
 * colsum_out (fp32 [N], may be null; trans = 0 and a bf16 C only): also receives the column sums of C as stored --
 * the bias gradient of the layer whose output gradient this GEMM produces (dgrad through GELU) -- summed in a fixed
 * order inside the epilogue, so C is not read again; `ws` must then hold adt_gemm_colsum_workspace_bytes(M, N).
 * Large problems run on persistent kernels that take tiles from device work counters owned by the library: one
 * 512-byte allocation per (device, stream), made on the first large call on that stream (the only allocation this
 * library ever makes; it synchronises the device once).  The counters are reset by the kernels themselves (the last
 * ticket of a launch zeroes its counter), so no host state mirrors them.  While `stream` is being captured into a
 * HIP graph the persistent kernels are not used (the tiled kernels are: no allocation, no shared counters).  As for
 * any stream-ordered API, calls that target the same stream must not race each other from different host threads.
 * M <= 64 with trans = 0 and K a multiple of 128 (the per-token projections of the KV-cached decode, reference
 * model.py:260-324) runs on a skinny kernel: one workgroup per 16 output columns, K split over its four waves and summed
 * in a fixed order; same epilogue, same dropout mask.
 */
typedef struct adt_gemm_epilogue {
  const float* bias;
  const void*  gelu_grad_of;  int64_t ld_gelu_grad;
  void*        pre_act_out;   int64_t ld_pre_act;
  const void*  residual;      int64_t ld_res;      int32_t res_row_mod;
  int32_t      act;
  float        alpha;
  int32_t      out_fp32;
  void*        aux_bf16_out;  int64_t ld_aux;
  adt_dropout  drop;          int32_t drop_after_residual;
  float*       colsum_out;
  int32_t      act_grad_mode;
  /* residual = LayerNorm(y) rebuilt on the fly: with res_ln_mean set, `residual` points at the PRE-LayerNorm tensor y and the
   * epilogue adds (y - res_ln_mean[row]) * res_ln_rstd[row] * res_ln_gamma[col] + res_ln_beta[col] -- the arithmetic of
   * adt_layernorm_fwd -- so the LayerNorm that feeds this residual never has to write its fp32 output (it is read exactly once,
   * here: 4 bytes per element of HBM traffic less per LayerNorm).  res_row_mod must be 0.  bf16 path only. */
  const float* res_ln_mean;
  const float* res_ln_rstd;
  const float* res_ln_gamma;
  const float* res_ln_beta;
  /* adt_gemm_bf16x3 only: gelu_grad_of / pre_act_out are fp32 arrays (ld_* in fp32 elements) -- the fp32-activation parity arm keeps its
   * saved GELU factor in fp32.  adt_gemm_bf16 rejects it unless the output is fp32 and the shape takes a persistent kernel. */
  int32_t      side_fp32;
} adt_gemm_epilogue;

size_t adt_gemm_workspace_bytes(int32_t trans, int64_t M, int64_t N, int64_t K);
size_t adt_gemm_colsum_workspace_bytes(int64_t M, int64_t N);

int adt_gemm_bf16(int32_t trans, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda,
                  const void* B, int64_t ldb, void* C, int64_t ldc, const adt_gemm_epilogue* ep,
                  void* ws, size_t ws_bytes, void* stream);

/* C = LayerNorm(y) @ B^T (+ the epilogue of adt_gemm_bf16) for a handful of rows: the decode step's LayerNorm -> projection pairs
 * (reference model.py:159-168 inside the sampler loop :260-324) in one launch.  y fp32 [M, K] (1 <= M <= 64, K a multiple of
 * 128 up to 1024), gamma / beta fp32 [K], B bf16 [N, K]; the operand is LN(y) rounded to bf16, exactly what adt_layernorm_fwd
 * hands to adt_gemm_bf16 (two-pass statistics in fp32).  x32 (optional, fp32 [M, ldx]) receives LN(y) itself -- the residual
 * a later GEMM adds.  Other shapes: ADT_ESHAPE (call adt_layernorm_fwd + adt_gemm_bf16). */
int adt_ln_gemm_bf16(int64_t M, int64_t N, int64_t K, const float* y, int64_t ldy, const float* gamma, const float* beta, float eps,
                     const void* B, int64_t ldb, void* C, int64_t ldc, const adt_gemm_epilogue* ep, float* x32, int64_t ldx, void* stream);

/* Grouped weight gradients: C_i[M_i, N_i] (fp32) = A_i[K_i, M_i]^T B_i[K_i, N_i] (bf16) for up to 32 independent items in ONE
 * launch, whole-K tiles, no workspace.  Replaces the autograd weight-gradient GEMMs of the reference's decoder layers
 * (nn.Linear / MultiheadAttention backward inside model.py:159-189): 25 products with K = B * T rows, each too small to fill
 * the chip without a K split and a reduction launch of its own.  Results equal adt_gemm_bf16(trans = 1) up to the summation
 * order over K (one chain per output element here; slab sums there).  M_i, N_i multiples of 8, K_i multiples of 64,
 * pointers 16-byte aligned, lda / ldb multiples of 8 elements, ldc of 4; items is host memory. */
typedef struct adt_gemm_tn_item {
  const void* A; int64_t lda;      /* bf16 [K, M] */
  const void* B; int64_t ldb;      /* bf16 [K, N] */
  float*      C; int64_t ldc;      /* fp32 [M, N] */
  int64_t     M, N, K;
} adt_gemm_tn_item;
int adt_gemm_bf16_tn_grouped(const adt_gemm_tn_item* items, int32_t n, void* stream);

/* ---------------------------------------------------------------------------
 * K4  multi-head attention forward / backward (flash-style, MFMA, head_dim 128)
 *
 * Replaces nn.MultiheadAttention's scaled-dot-product core inside the reference's
 * transformer layers: encoder self-attention (reference model.py:118-127,133; no mask),
 * decoder self-attention with the ADDITIVE causal and key-padding masks of
 * model.py:173-181 (0 / mask_value = -1e4, the two add up), and cross-attention
 * onto the encoder memory (:182-189; no mask).
 *
 * q, k, v, o (and dout, dq, dk, dv) are bf16 matrices of B*q_len or B*k_len rows;
 * row r of batch b is b*len + r, head h occupies columns [h*128, h*128+128), and
 * ldq/ldk/ldv/ldo are the row strides in elements -- so a packed in_proj output
 * [B*S, 3*d] is consumed in place (q at column 0, k at d, v at 2d; ld = 3d).
 * dq/dk/dv use the strides of q/k/v.
 *   o   = softmax(q k^T * scale + mask) v
 *   lse = log-sum-exp of each score row, fp32 [B, heads, q_len] (kept for backward)
 *   key_len (optional, int32 [B]): keys >= key_len[b] get mask_value added
 *   causal != 0: keys > query index get mask_value added
 * The backward recomputes the probabilities from lse (two kernels: dq; dk+dv),
 * without float atomics.  ws: adt_attn_bwd_workspace_bytes.
 * q_len == 1 without dropout or a causal mask (one step of the KV-cached greedy decode: the new position over the cache,
 * or over the encoder memory) runs on a single-query kernel that splits the keys over the waves of one workgroup per
 * (batch, head) and visits only the keys below key_len when mask_value <= -1000 (the masked keys' weights are exactly 0
 * in fp32 then); the probabilities are rounded to bf16 for the P V product, as in the tiled kernel.
 */
typedef struct adt_attn_desc {
  int32_t batch, heads, q_len, k_len, head_dim;
  int32_t causal;
  int64_t ldq, ldk, ldv, ldo;
  float   scale;
  float   mask_value;
  const int32_t* key_len;
  adt_dropout drop;       /* dropout on the attention probabilities; row = (b*heads+h)*q_len+q, col = k (see adt_dropout) */
  /* backward only, all three or none (may be null): fp32 [heads*128] bias gradient of the in-projection = column sums of
   * dq (taken in the dQ kernel's epilogue, fixed order), dk (identically zero: the rows of dS sum to zero, so exact zeros are
   * written instead of the rounding noise a sum over the stored dk would give) and dv (one pass over its columns) */
  float* dq_colsum; float* dk_colsum; float* dv_colsum;
  /* dropout only, may be null: adt_attn_keep_bits_bytes(d) bytes of device memory, 16-byte aligned.  adt_attn_fwd leaves every keep
   * decision it took there as a bit (its compare results, one 32-bit word per key and 32-query slice); adt_attn_bwd given the SAME
   * buffer (and the same drop) reads them back instead of hashing every mask again and takes the one-kernel backward.  The bits are
   * a cache of the mask function of adt_dropout, never a different mask: with or without them the gradients are those of the same
   * forward.  Ignored by the fp32-operand entry points. */
  void* keep_bits;
  /* fp32-operand entry points only (adt_attn_fwd_f32 / adt_attn_bwd_f32; must be 0 or 1, the bf16 entry points ignore it): 0 = exact
   * f32-input MFMA products, 1 = split-bf16 products (every fp32 operand as bf16 hi + lo, three bf16 MFMAs per product: "bf16x3") */
  int32_t f32_products;
} adt_attn_desc;

int adt_attn_fwd(const adt_attn_desc* d, const void* q, const void* k, const void* v, void* o, float* lse, void* stream);
size_t adt_attn_keep_bits_bytes(const adt_attn_desc* d);
size_t adt_attn_bwd_workspace_bytes(const adt_attn_desc* d);
int adt_attn_bwd(const adt_attn_desc* d, const void* q, const void* k, const void* v, const void* o, const void* dout,
                 const float* lse, void* dq, void* dk, void* dv, void* ws, size_t ws_bytes, void* stream);
/* The one-kernel backward hands dQ tiles between the key-block workgroups of a head through flags in device memory; a wave whose
 * partner never shows up within ~5 s gives up (instead of hanging the GPU) and its dQ tile is incomplete.  Every give-up is counted in a
 * pinned host word of the current device: the next adt_attn_bwd call on the device fails with ADT_EHIP, and this function returns the
 * count (>= 0; clear != 0 also resets it; -1: no device) without synchronising -- adt_str_amd/trainer.py checks it at the end of every
 * training step, so a step never silently trains on an incomplete gradient. */
int32_t adt_attn_bwd_giveups(int32_t clear);

/* ---------------------------------------------------------------------------
 * K6  LayerNorm forward / backward  (nn.LayerNorm(d), eps 1e-5, fp32 statistics)
 *
 * Replaces norm1/norm2/norm3 of the post-norm transformer layers and
 * Encoder.layer_norm (reference model.py:113-115,118-127,159-168).
 *   fwd: y = (x - mean) * rstd * gamma + beta, written as fp32 (y32) and/or
 *        bf16 (y16, the next GEMM's operand); mean/rstd [M] are kept for backward.
 *   bwd: dx from dy, the saved input x and statistics; dx as fp32 and/or bf16;
 *        dgamma, dbeta and dxsum[D] = column sums of dx (the bias gradient of the
 *        linear layer that produced x's branch); any of the three may be null.
 *   Dropout: out_drop masks the forward output (element index row*D + col); in the backward dy_drop
 *   re-applies that mask to dy, and dx16_drop masks the bf16 dx and dxsum (the gradient of a branch
 *   whose output was dropped before the residual add) while dx32 stays the residual-stream gradient.
 */
int adt_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps,
                      float* y32, void* y16, int64_t ldy, float* mean, float* rstd, int64_t M, int64_t D,
                      const adt_dropout* out_drop, void* stream);
size_t adt_layernorm_bwd_workspace_bytes(int64_t M, int64_t D);
int adt_layernorm_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma,
                      const float* mean, const float* rstd, float* dx32, void* dx16, int64_t lddx,
                      float* dgamma, float* dbeta, float* dxsum, int64_t M, int64_t D,
                      const adt_dropout* dy_drop, const adt_dropout* dx16_drop, void* ws,
                      size_t ws_bytes, void* stream);

/* Column sums of a bf16 [M,N] matrix -> fp32 [N] (bias gradients of in_proj, linear1,
 * generator, project_to_mel).  Fixed-order two-stage reduction. */
size_t adt_colsum_workspace_bytes(int64_t M, int64_t N);
int adt_colsum_bf16(const void* x, int64_t ld, int64_t M, int64_t N, float* out, void* ws, size_t ws_bytes, void* stream);

/* Deferred second-stage reductions.  The bias / LayerNorm-parameter gradients above all end in the same small reduction
 * (column sums over per-workgroup partial rows); a training step launches ~55 of them, a few microseconds each.  Between
 * adt_reduce_queue_begin and _end (per calling thread, for launches on `stream`), adt_layernorm_bwd*, adt_colsum_bf16,
 * adt_gemm_bf16 with colsum_out (bf16 NT path) and adt_attn_bwd with column-sum outputs write their partials into slices of
 * `arena` and QUEUE that reduction; adt_reduce_queue_flush launches everything queued as one kernel (same summation order:
 * results are bitwise those of the immediate launches) and rewinds the arena.  The queued outputs (dgamma, dbeta, dxsum,
 * colsum_out, the attention bias gradients) are undefined until the flush.  A producer that finds the arena full reduces
 * immediately, as without a queue.  _end flushes (discard = 0) or drops the queued entries (discard != 0: error unwinding),
 * and closes the queue; it is a no-op without an open queue.  The host side flushes once per gradient segment, before the
 * segment is handed to the all-reduce (adt_str_amd/network.py: _ready). */
int adt_reduce_queue_begin(void* arena, size_t arena_bytes, void* stream);
int adt_reduce_queue_flush(void);
int adt_reduce_queue_end(int discard);

/* The tail of one greedy-decode step (reference model.py:300-322: argmax of the generator's logits, finished rows keep emitting
 * the end token, stop once every row has finished), on device state so that the whole step replays as a HIP graph:
 *   nxt[b]  = finished[b] ? end_token : argmax_c logits[b][c]     (first index on ties; NaN counts as the maximum: torch.argmax)
 *   gen[b][*t + 1] = nxt[b];  finished[b] |= nxt[b] == end_token;  tok[b] = nxt[b];  klen[b] += 1
 *   if every row has finished and *done_at == max_length:  *done_at = *t + 2   (the number of columns the reference returns)
 *   *t += 1
 * logits fp32 [B, V] (row stride ld), finished uint8 [B], gen int64 [B, ld_gen >= max_length], t / done_at int64 scalars,
 * tok int64 [B], klen int32 [B].  The caller keeps *t + 1 < max_length. */
int adt_greedy_step(const float* logits, int64_t ld, int64_t B, int64_t V, uint8_t* finished, int64_t* gen, int64_t ld_gen,
                    int64_t* t, int64_t* tok, int32_t* klen, int64_t* done_at, int64_t end_token, int64_t max_length, void* stream);

/* ---------------------------------------------------------------------------
 * K7  token embedding * sqrt(d) + positional encoding
 *
 * Replaces TokenEmbedding_plain.forward and PositionalEncoding.forward
 * (reference model.py:42-65) as used by Decoder.forward (:171).
 *   fwd: y[row] = table[tokens[row]] * scale + pe[row % T]   (rows = B*T, row-major)
 *   bwd: dtable[tokens[row]] += scale * dy[row]              (dtable pre-zeroed by the caller)
 *        adt_embed_bwd does it with fp32 atomic adds (summation order not fixed from run to run; any vocab / D).
 *        adt_embed_bwd_operands instead writes the two bf16 operands of dtable = onehot^T . dy16 -- onehot [n_rows, vocab]
 *        (1.0 at the row's token) and dy16 = bf16(scale * keep * dy) -- for adt_gemm_bf16(trans = 1, M = vocab, N = D,
 *        K = n_rows): fixed summation order and MFMA rate, the form the training step uses (vocab % 8 == 0, D % 8 == 0).
 */
int adt_embed_pe_fwd(const int64_t* tokens, const float* table, const float* pe, float scale, float* y32, void* y16,
                     int64_t n_rows, int64_t T, int64_t D, int64_t vocab, const adt_dropout* drop, void* stream);
int adt_embed_bwd(const int64_t* tokens, const float* dy, float scale, float* dtable, int64_t n_rows, int64_t D,
                  int64_t vocab, const adt_dropout* drop, void* stream);
int adt_embed_bwd_operands(const int64_t* tokens, const float* dy, float scale, void* onehot, int64_t ld_onehot, void* dy16,
                           int64_t n_rows, int64_t D, int64_t vocab, const adt_dropout* drop, void* stream);

/* ---------------------------------------------------------------------------
 * K8  cross-entropy forward + backward
 *
 * Replaces ADTModel._loss_fn (reference model.py:228-238): fp32 logits,
 * nan_to_num(nan=0, +-inf=+-1e4), F.cross_entropy(ignore_index, mean over kept rows).
 *   loss[0]  scalar fp32
 *   dlogits  (optional) bf16 [M, ldd]: d loss / d logits
 */
size_t adt_cross_entropy_workspace_bytes(int64_t M);
int adt_cross_entropy(const float* logits, int64_t ld, const int64_t* labels, int64_t ignore_index, int64_t M, int64_t V,
                      float* loss, void* dlogits, int64_t ldd, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------
 * fp32-operand parity path (adt_str_amd/csrc/precise.hip)
 *
 * BASELINE's parity target -- "logits within 1e-3 rel-tol of the CPU reference" -- is a statement about the reference's
 * fp32 CPU path (ADTModel.forward, reference model.py:240-258, run without autocast).  The entries below are the fp32
 * twins of K3-K8: every operand and activation is fp32 and every contraction runs on v_mfma_f32_32x32x2_f32 (exact f32
 * products, f32 accumulate).  The engine selects them with precision="fp32" (adt_str_amd/network.py); they are the
 * parity arm, never the throughput default (the f32 MFMA rate is 1/16 of the bf16 rate).
 *
 *   adt_gemm_f32   layout bit 0: A is stored [K, M] (else [M, K]); bit 1: B is stored [K, N] (else [N, K]);
 *                  0 = y = x W^T, 2 = dx = dy W, 3 = dW = dy^T x.  C[M, N] fp32.
 *                  bit 2 (value 4): split-bf16 products ("bf16x3", round 6) -- every fp32 operand is split, while its tile is staged,
 *                  into hi = bf16(x) and lo = bf16(x - hi), and a product is a_lo b_hi + a_hi b_lo + a_hi b_hi: three
 *                  v_mfma_f32_32x32x16_bf16 into one fp32 accumulator, ~4e-6 of the output's magnitude against 1e-6 for the exact
 *                  f32 MFMA and 2e-3 for bf16 operands, on the 16 x faster matrix pipe.  Same layouts, epilogue and fp32 tensors.
 *                  ws (adt_gemm_f32_workspace_bytes, may be null): products whose output has too few 128 x 128 tiles to fill the
 *                  chip and a plain epilogue (the weight gradients) are split along K through fp32 slabs summed in slab order.  Same epilogue struct and order as
 *                  adt_gemm_bf16 with gelu_grad_of / pre_act_out read and written as fp32; aux_bf16_out and colsum_out
 *                  must be null (adt_colsum_f32 takes the bias gradients).  The contiguous extent of each operand and
 *                  lda / ldb are multiples of 4 floats, operands 16-byte aligned.
 *   adt_attn_fwd_f32 / adt_attn_bwd_f32   adt_attn_fwd / adt_attn_bwd on fp32 q, k, v, o (same descriptor, masks and
 *                  dropout indices; row strides multiples of 4); the dq/dk/dv_colsum fields must be null.
 *                  adt_attn_desc.f32_products = 1: split-bf16 products as above (head_dim 16 ... 128).
 *   adt_colsum_f32        column sums of an fp32 [M, N] matrix, fixed order.
 *   adt_layernorm_bwd_f32 adt_layernorm_bwd with the branch gradient (dx16 there) written as fp32.
 *   adt_cross_entropy_f32 adt_cross_entropy with fp32 dlogits and libm exp / log.
 *   adt_embed_bwd_operands_f32  fp32 one-hot / scaled-gradient operands of dtable = onehot^T . dy (adt_gemm_f32 layout 3).
 */
size_t adt_gemm_f32_workspace_bytes(int32_t layout, int64_t M, int64_t N, int64_t K);
int adt_gemm_f32(int32_t layout, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                 float* C, int64_t ldc, const adt_gemm_epilogue* ep, void* ws, size_t ws_bytes, void* stream);
int adt_attn_fwd_f32(const adt_attn_desc* d, const float* q, const float* k, const float* v, float* o, float* lse, void* stream);
size_t adt_attn_bwd_f32_workspace_bytes(const adt_attn_desc* d);
int adt_attn_bwd_f32(const adt_attn_desc* d, const float* q, const float* k, const float* v, const float* o, const float* dout,
                     const float* lse, float* dq, float* dk, float* dv, void* ws, size_t ws_bytes, void* stream);
size_t adt_colsum_f32_workspace_bytes(int64_t M, int64_t N);
int adt_colsum_f32(const float* x, int64_t ld, int64_t M, int64_t N, float* out, void* ws, size_t ws_bytes, void* stream);
int adt_layernorm_bwd_f32(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma,
                          const float* mean, const float* rstd, float* dx32, float* dx_branch, int64_t lddx,
                          float* dgamma, float* dbeta, float* dxsum, int64_t M, int64_t D,
                          const adt_dropout* dy_drop, const adt_dropout* branch_drop, void* ws,
                          size_t ws_bytes, void* stream);
int adt_cross_entropy_f32(const float* logits, int64_t ld, const int64_t* labels, int64_t ignore_index, int64_t M, int64_t V,
                          float* loss, float* dlogits, int64_t ldd, void* ws, size_t ws_bytes, void* stream);
int adt_embed_bwd_operands_f32(const int64_t* tokens, const float* dy, float scale, float* onehot, int64_t ld_onehot, float* dy_scaled,
                               int64_t n_rows, int64_t D, int64_t vocab, const adt_dropout* drop, void* stream);

/* Split-bf16 products on the persistent bf16 kernels (the fast form of the "bf16x3" parity arm, round 6): the reference's fp32 CPU
 * arithmetic of every nn.Linear (model.py:111,118-127,157,159-168,224) within ~1e-5 relative per product, at the speed of the bf16
 * matrix pipe.  An fp32 operand is split ONCE per tensor into two bf16 planes hi = bf16(x), lo = bf16(x - hi), stored side by side in
 * one row-major buffer ([rows, ..hi.., ..lo..]; adt_split_bf16x2: planes[r, c] and planes[r, lo_off + c]; transpose != 0 writes the
 * planes of x^T: planes[c, r] and planes[c, lo_off + r]), and
 *   adt_gemm_bf16x3(trans = 0):  C[M, N] = A[M, K] . B[N, K]^T     A2 = planes of A (K wide), B2 = planes of B (K wide)
 *   adt_gemm_bf16x3(trans = 1):  C[M, N] = A[K, M]^T . B[K, N]     A2 = planes of A (M wide), B2 = planes of B (N wide)
 * runs a_lo b_hi + a_hi b_lo + a_hi b_hi as ONE bf16 GEMM over three segments of K (small terms first), fp32 accumulate, fp32 output,
 * the epilogue menu of adt_gemm_bf16 (ep->out_fp32 must be set; side_fp32 makes gelu_grad_of / pre_act_out fp32 arrays).
 * a_lo / b_lo: element offset of the lo plane from the hi plane (>= the plane width, a multiple of 8) -- a column slice of a weight's
 * planes keeps the full plane distance.  Only shapes that take a persistent 256 x 256 kernel (adt_gemm_bf16x3_supported; K a multiple
 * of 64): everything else ADT_ESHAPE -- call adt_gemm_f32 with layout bit 4 there.  ws as adt_gemm_bf16x3_workspace_bytes (trans = 1). */
int adt_split_bf16x2(const float* x, int64_t ldx, int64_t rows, int64_t cols, void* planes, int64_t ldp, int64_t lo_off, int32_t transpose,
                     void* stream);
int32_t adt_gemm_bf16x3_supported(int32_t trans, int64_t M, int64_t N, int64_t K);
size_t adt_gemm_bf16x3_workspace_bytes(int32_t trans, int64_t M, int64_t N, int64_t K);
int adt_gemm_bf16x3(int32_t trans, int64_t M, int64_t N, int64_t K, const void* A2, int64_t lda, int64_t a_lo,
                    const void* B2, int64_t ldb, int64_t b_lo, float* C, int64_t ldc, const adt_gemm_epilogue* ep,
                    void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------
 * Parameter plumbing for the bf16 compute path: fp32 master weight [rows, cols]
 * -> bf16 copy y (same layout) and/or y_t [cols, rows] (the NT operand of dgrad).
 */
int adt_cast_bf16(const float* x, void* y, void* y_t, int64_t rows, int64_t cols, void* stream);

/* The same for a whole table of matrices in one launch (the per-step refresh of every weight's bf16 operands):
 * items_dev is a DEVICE array of n_items records; max_tiles = max over items of ceil(rows/64) * ceil(cols/64).
 * y and/or y_t may be null per item. */
typedef struct adt_cast_item {
  const float* x;       /* fp32 [rows, cols], contiguous */
  void* y;              /* bf16 [rows, cols] or null     */
  void* y_t;            /* bf16 [cols, rows] or null     */
  int32_t rows, cols;
} adt_cast_item;
int adt_cast_bf16_batched(const adt_cast_item* items_dev, int32_t n_items, int32_t max_tiles, void* stream);

/* ---------------------------------------------------------------------------
 * Optimizer step on flat fp32 buffers: torch.nn.utils.clip_grad_norm_(max_norm)
 * followed by torch.optim.AdamW (reference train.py:219-249, HF Trainer defaults).
 *   adt_grad_norm: norm_and_clip[0] = ||g||_2, [1] = min(1, max_norm/(norm+1e-6))
 *                  (1 when max_norm <= 0); both stay on the device.
 *   adt_adamw_step: p, m, v updated in place with g * norm_and_clip[1]
 *                  (norm_and_clip may be null); p_bf16 (optional) = bf16(p).
 *                  nodecay_ranges (DEVICE int64 [n_nodecay][2], sorted, disjoint, bounds multiples of 4; may be null):
 *                  flat ranges [lo, hi) that get no weight decay -- biases and LayerNorm weights, the parameters HF
 *                  Trainer.get_decay_parameter_names leaves out of the decayed group (the reference trains through it).
 */
size_t adt_grad_norm_workspace_bytes(void);
int adt_grad_norm(const float* g, int64_t n, float max_norm, float* norm_and_clip, void* ws, size_t ws_bytes, void* stream);
int adt_adamw_step(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int64_t step, const float* norm_and_clip,
                   const int64_t* nodecay_ranges, int32_t n_nodecay, void* stream);

/* ---------------------------------------------------------------------------
 * K9  CLAP log-mel (dB) feature extractor
 *
 * Replaces ClapProcessor(audio=..., sampling_rate=48000) as the reference calls it
 * (modules/clap_encoder.py:22-23; transformers ClapFeatureExtractor, float64 numpy per clip):
 * "repeatpad" each clip to target_len samples (tile floor(target/n) times, then zeros),
 * STFT n_fft 1024 / hop, periodic Hann, center=True reflect, power, banded mel filterbank
 * (same CSR form as K1: mel_meta [n_mels][4], mel_w), 10*log10(max(mel, amin)).
 *   waves    concatenated fp32 clips; clip c = waves[offsets[c] .. offsets[c+1]), each 1..target_len samples
 *   out      [n_clips, n_frames, n_mels] fp32 (the reference stacks this 4x for the fusion model)
 * Supported: n_fft == 1024, n_mels <= 64 and a multiple of 4.
 */
int adt_clap_logmel_db_f32(const float* waves, const int64_t* offsets, int64_t n_clips, int32_t target_len, int32_t n_fft,
                           int32_t hop, int32_t n_frames, const float* window, const int32_t* mel_meta, const float* mel_w,
                           int32_t n_mels, int32_t mel_nnz, float amin, float* out, void* stream);
/* The same over clips that live in device memory one by one (modules/clap_encoder.py:22 hands the extractor a LIST of clips): clip_ptrs is a
 * device array of n_clips pointers, clip c = clip_ptrs[c][0 .. offsets[c+1] - offsets[c]) -- no concatenation pass. */
int adt_clap_logmel_db_ptrs_f32(const float* const* clip_ptrs, const int64_t* offsets, int64_t n_clips, int32_t target_len, int32_t n_fft,
                                int32_t hop, int32_t n_frames, const float* window, const int32_t* mel_meta, const float* mel_w,
                                int32_t n_mels, int32_t mel_nnz, float amin, float* out, void* stream);

/* Bilinear resize of one fp32 image [H_in, W_in] -> [H_out, W_out]: torch.nn.functional.interpolate(mode="bilinear",
 * align_corners=False) as ClapFeatureExtractor's fusion truncation calls it to shrink the whole mel of a clip longer than 10 s to
 * [1001, 64] (transformers feature_extraction_clap.py: _random_mel_fusion; reached from clap_encoder.py:22-23). */
int adt_bilinear_resize_f32(const float* in, int64_t H_in, int64_t W_in, int64_t ld_in, float* out, int64_t H_out, int64_t W_out,
                            int64_t ld_out, void* stream);

/* ---------------------------------------------------------------------------
 * K10/K11  HTSAT (audio Swin) specific kernels, forward only
 *
 * Replace the non-GEMM parts of transformers' ClapAudioEncoder.forward, which the reference runs through
 * ClapWrapper._get_audio_features (modules/clap_encoder.py:45-49): the dense layers use adt_gemm_bf16 and
 * adt_layernorm_fwd.
 *   adt_htsat_front_f32    mel [B, in_frames, n_mels] fp32 -> image [B, img_side, img_side] fp32:
 *                          eval BatchNorm2d per mel bin (y = x * bn_scale[f] + bn_shift[f]), bicubic resize of the
 *                          time axis to out_frames (align_corners, A = -0.75), freq-stacking fold (reshape_mel2img)
 *   adt_htsat_patch_embed  4x4/stride-4 conv, 1 -> C channels (w [C,16], bias [C]) + LayerNorm -> tokens [B*(side/4)^2, C]
 *   adt_window_attn_fwd    8x8-window attention, head_dim 24: qkv [B*R*R, >= 3C] bf16 (q | k | v, head h at column
 *                          24h) -> ctx [B*R*R, C] bf16; cyclic shift `shift` and the window partition are index math;
 *                          bias fp32 = relative position bias (+ the -100 shifted-window mask of each window when
 *                          n_bias_windows == (R/8)^2) of query q, key k, stored lane-linear for the kernel's accumulator
 *                          layout: [n_bias_windows][heads][qt 2][kt 2][g 4][h 2][r 32][e 4] with q = 32 qt + r,
 *                          k = 32 kt + 8 g + 4 h + e, and IN LOG2 UNITS: every entry multiplied by log2(e), because the
 *                          kernel evaluates softmax as 2^(q.k * scale * log2 e + bias - max) (one multiply per score less);
 *                          adt_str_amd/clap_encoder.py:window_bias_layout builds it
 *   adt_patch_merge_ln     Swin patch merging gather (2x2 -> 4C, order (0,0),(1,0),(0,1),(1,1)) + LayerNorm -> bf16
 *   adt_mean_tokens        mean over the T tokens of each clip;  adt_l2_normalize  rows / ||row||
 */
int adt_htsat_front_f32(const float* mel, int64_t ld_clip, int64_t B, int32_t in_frames, int32_t n_mels, int32_t out_frames,
                        int32_t img_side, const float* bn_scale, const float* bn_shift, float* img, void* stream);
int adt_htsat_patch_embed(const float* img, int64_t B, int32_t img_side, const float* w, const float* bias, const float* gamma,
                          const float* beta, float eps, int32_t C, float* out32, void* out16, void* stream);
/* AFF fusion branch of ClapAudioPatchEmbed.forward (transformers modeling_clap.py, `is_longer` items; reached from
 * clap_encoder.py:45-49 because the feature extractor marks at least one clip per batch) for ONE clip:
 * img_global [side, side] (mel channel 0 after adt_htsat_front_f32), img_local [3, side, side] (channels 1..3) ->
 * tokens out32 [(side/4)^2, C] after proj / mel_conv2d (4 x 12, stride 4 x 12) / ClapAudioAFFBlock / LayerNorm.
 * The 1x1 convolutions of local_att / global_att carry their eval-mode BatchNorm folded in (w' = w * s, b' = (b - mean) * s + beta). */
struct adt_aff_weights {
  const float *proj_w, *proj_b;        /* [C,16], [C] */
  const float *conv_w, *conv_b;        /* [C,48], [C] */
  const float *local_w1, *local_b1, *local_w2, *local_b2;     /* [I,C], [I], [C,I], [C] */
  const float *global_w1, *global_b1, *global_w2, *global_b2; /* [I,C], [I], [C,I], [C] */
  const float *ln_gamma, *ln_beta;     /* [C] */
};
size_t adt_htsat_fusion_embed_workspace_bytes(int32_t img_side, int32_t C);
int adt_htsat_fusion_embed(const float* img_global, const float* img_local, int32_t img_side, const struct adt_aff_weights* w, float eps,
                           int32_t C, int32_t inter, void* ws, size_t ws_bytes, float* out32, void* stream);
int adt_window_attn_fwd(const void* qkv, int64_t ld_qkv, void* ctx, int64_t ld_ctx, const float* bias, int32_t n_bias_windows,
                        int64_t B, int32_t R, int32_t C, int32_t heads, int32_t shift, float scale, void* stream);
int adt_patch_merge_ln(const float* x, int64_t B, int32_t R, int32_t C, const float* gamma, const float* beta, float eps,
                       void* out_bf16, void* stream);
int adt_mean_tokens(const float* x, int64_t B, int32_t T, int32_t C, float* out32, void* out16, void* stream);
/* LayerNorm of every token row, then the mean over each clip's T tokens, in one pass (ClapAudioEncoder.norm + the average-pool head,
 * modeling_clap.py; x [B * T, D] fp32, D % 4 == 0, D <= 1024): out32 [B, D] fp32 and / or out16 [B, D] bf16. */
int adt_ln_mean_tokens(const float* x, int64_t B, int32_t T, int32_t D, const float* gamma, const float* beta, float eps, float* out32,
                       void* out16, void* stream);

/* K15  fused row-block kernels for the bandwidth-bound Swin stages (C = 96, 192; modes 0, 1, 4 also C = 384): one launch per half of a
 * ClapAudioLayer (transformers modeling_clap.py ClapAudioLayer.forward, reached from modules/clap_encoder.py:45-49).
 * x [M, C] fp32 is the residual stream (tokens in any order: the kernels are row-wise).
 *   mode 0  LN + GEMM:        out16[M, 32 n_tiles] (bf16, row stride ldo) = LayerNorm(x) W^T + bias1        (layernorm_before + q|k|v)
 *   mode 1  GEMM + residual:  x += a16[M, C] (bf16, row stride lda) W^T + bias1                             (attention.output.dense)
 *   mode 2  MLP:              x += gelu(LayerNorm(x) W1^T + bias1) W2^T + bias2, W1 [4C, C], W2 [C, 4C]     (layernorm_after + MLP)
 *   mode 4  LN + GEMM + GELU: out16 = gelu(LayerNorm(x) W^T + bias1)                                        (layernorm_after + intermediate)
 *   (mode 3: mode 2 without its software pipeline, the A/B arm; GELU = the erf form to half a bf16 ulp, csrc/gelu.h)
 * w_packed: the weights as the stream of 1 KiB MFMA fragments the kernel consumes (adt_str_amd/clap_encoder.py:pack_rowblock_weights):
 *   product over C, output tile n, k-step s:  64 lanes x 8 bf16, lane (r, h) = W[32n + r][16s + 8h + j], j = 0..7;
 *   modes 0 / 1: tiles n = 0 .. n_tiles-1, k-steps s = 0 .. C/16-1 each;  mode 2: per hidden tile n its C/16 fragments of W1, then
 *   for s2 = 0, 1 and channel tile ct = 0 .. C/32-1 the fragment lane (r, h) = W2[32ct + r][32n + 16 s2 + 8 (j>>2) + 4h + (j&3)];
 *   n_tiles must be a multiple of adt_htsat_rowblock_chunk_tiles(mode, C) (the LDS-DMA chunk).
 * ln_gamma = ln_beta = NULL (modes 0, 2, 4 and adt_htsat_attn_block): plain normalisation (x - mean) * rstd -- the caller has folded the affine
 *   part into what follows, W' = W diag(gamma), bias' = bias + W beta (exact in real arithmetic; the fused tower does, it saves the kernels
 *   4 C/16 loads per token row). */
int adt_htsat_rowblock_chunk_tiles(int32_t mode, int32_t C);
/* The attention half of a ClapAudioLayer in ONE launch (C = 96, 4 heads): x += out_proj(window_attention(q|k|v(LayerNorm(x)))) in place
 * (modeling_clap.py ClapAudioLayer.forward up to the first residual; shift / window partition are index math as in adt_window_attn_fwd).
 *   w_packed: per head one 24 KiB chunk of MFMA fragments (see adt_htsat_rowblock): the 32-unit tiles Wq_h, Wk_h, Wv_h (rows = the head's 24
 *   output units + 8 zero rows; 6 k-step fragments each), then Wo's slice for the head's 24 (+ 8 zero) inputs as fragments (s2, ct) with the
 *   inputs of a k-step in accumulator order 8 (j>>2) + 4h + (j&3)  (adt_str_amd/clap_encoder.py:pack_attn_block_weights);
 *   qkv_bias [heads][3][32] (units 24..31 zero), out_bias [C], rel_bias / n_bias_windows as adt_window_attn_fwd. */
int adt_htsat_attn_block(float* x, int64_t B, int32_t R, int32_t C, int32_t heads, int32_t shift, const float* ln_gamma,
                         const float* ln_beta, float eps, const void* w_packed, const float* qkv_bias, const float* out_bias,
                         const float* rel_bias, int32_t n_bias_windows, float scale, void* stream);
int adt_htsat_rowblock(int32_t mode, float* x, int64_t M, int32_t C, const void* a16, int64_t lda, const float* ln_gamma,
                       const float* ln_beta, float eps, const void* w_packed, int32_t n_tiles, const float* bias1,
                       const float* bias2, void* out16, int64_t ldo, void* stream);
/* One whole ClapAudioLayer in ONE launch (C = 96 / 192 / 384 with 4 / 8 / 16 heads -- the first three stages; modeling_clap.py ClapAudioLayer.forward): adt_htsat_attn_block
 * followed by adt_htsat_rowblock mode 2 on the rows still in registers, so the residual stream is read once and written once per layer.
 * Both LayerNorms folded into the weights by the caller (see ln_gamma = NULL above): attn_w_packed / qkv_bias as adt_htsat_attn_block with
 * W' = Wq|k|v diag(gamma1), mlp_w_packed / fc1_bias as mode 2 with W1' = W1 diag(gamma2); n_tiles = C / 8.
 * rel_bias_bf16 (C = 192 only, else ignored / NULL): the same bias table as bf16 in the order [.., query tile 2, key tile 2, group pair 2,
 * lane 64, group 2, e 4] (clap_encoder.py:window_bias_layout_bf16) -- that stage runs two workgroups per CU and stages the bias in half the LDS. */
int adt_htsat_layer_block(float* x, int64_t B, int32_t R, int32_t C, int32_t heads, int32_t shift, float eps, const void* attn_w_packed,
                          const float* qkv_bias, const float* out_bias, const float* rel_bias, int32_t n_bias_windows, float scale,
                          const void* mlp_w_packed, int32_t n_tiles, const float* fc1_bias, const float* fc2_bias, const void* rel_bias_bf16,
                          void* stream);
/* Patch merging between stage 0 and stage 1 in ONE launch (ClapAudioPatchMerging.forward, modeling_clap.py: 2x2 gather in the order
 * (0,0) (1,0) (0,1) (1,1) -> LayerNorm(4 C_src) -> reduction Linear without bias), replacing adt_patch_merge_ln + adt_gemm_bf16 there:
 *   x [B * R * R, C_src] fp32 (C_src = 96), out32 [B * (R/2)^2, ldo] fp32 = LayerNorm(gather(x)) W^T + bias,
 *   w_packed = W [32 n_tiles, 4 C_src] as the mode-0 fragment stream of adt_htsat_rowblock, bias [32 n_tiles] (zeros for the reference). */
int adt_htsat_merge_rowblock(const float* x, int64_t B, int32_t R, int32_t C_src, const float* ln_gamma, const float* ln_beta, float eps,
                             const void* w_packed, int32_t n_tiles, const float* bias, float* out32, int64_t ldo, void* stream);
int adt_l2_normalize(const float* x, int64_t n_rows, int32_t D, float* out, void* stream);

/* ---------------------------------------------------------------------------
 * K12  CLAP curation: cosine similarity to the class means + per-sample best class
 *
 * Replaces the similarity loop and the greedy "first occurrence wins" assignment of the
 * reference's curation driver (data_modules/augment_data_with_CLAP.py:139-151,182-193).
 *   emb [N, ld] fp32 sample embeddings, refs [C, D] fp32 class-mean embeddings
 *   cos(n, c) = <x_n, r_c> / sqrt(max(|x_n|^2 |r_c|^2, eps^2))      (F.cosine_similarity, eps 1e-8)
 *   best_class[n] = lowest c with the maximum cos(n, c); best_score[n] = that maximum
 *   scores (optional) [N, C] = every cos(n, c)
 */
int adt_cosine_argmax_f32(const float* emb, int64_t ld, const float* refs, int64_t N, int64_t D, int64_t C, float eps,
                          int32_t* best_class, float* best_score, float* scores, void* stream);

/* ---------------------------------------------------------------------------
 * K13  Polyphase sinc resampler
 *
 * Replaces torchaudio.transforms.Resample(orig_freq, new_freq)(waveform) with default arguments
 * (utils/audio_utils.py:18-20, inference.py:89-90, data_modules/augment_data_with_CLAP.py:56-59).
 *   in  [B, L_in] fp32 (row stride ld_in) -> out [B, L_out] fp32, L_out <= ceil(new * L_in / orig)
 *   orig, neu   the two rates divided by their gcd
 *   bank        [neu][K] fp32, K = 2 * width + orig: torchaudio's sinc_interp_hann kernel bank
 *   tap_range   [neu][2] int32: first / one-past-last tap of each phase that is not an exact zero
 * out[b][m * neu + p] = sum_k bank[p][k] * in_padded[b][m * orig + k], in_padded = in shifted by `width` zeros.
 */
int adt_resample_f32(const float* in, int64_t B, int64_t L_in, int64_t ld_in, const float* bank, const int32_t* tap_range,
                     int32_t K, int32_t width, int32_t orig, int32_t neu, float* out, int64_t L_out, int64_t ld_out, void* stream);

/* ---------------------------------------------------------------------------
 * H1  Curation host I/O: batched WAV decode and batched file copies (host threads, no GPU work)
 *
 * Replaces, for a whole batch of files per call, what the reference does one file at a time on the thread that drives the
 * GPU: torchaudio.load(path) -> waveform.mean(dim=0, keepdim=True) -> waveform / waveform.abs().max()
 * (data_modules/augment_data_with_CLAP.py:51-68, convert_augmented_to_hdf5.py:97-103) and shutil.copy2 per chosen file
 * (augment_data_with_CLAP.py:184-196).  Decoding is adt_str_amd/audio_io.py:read_wav to the bit: RIFF chunk walk (last
 * "fmt " / "data" chunk wins, WAVE_FORMAT_EXTENSIBLE sub-format, a data chunk cut short by the end of the file is taken as
 * far as it goes), 8 / 16 / 24 / 32-bit PCM as sample / 2^(bits-1) and 32-bit float, all in float32.
 *
 *   adt_wav_probe_batch   info[i] <- header of paths[i]; info[i].status = ADT_OK, or ADT_EINVAL (unreadable / not RIFF-WAVE /
 *                         no fmt or data chunk) or ADT_ESHAPE (an encoding outside the list above).  The call itself
 *                         returns ADT_OK: one bad file does not fail the batch.
 *   adt_wav_decode_batch  for every file with status ADT_OK: out[offsets[i] .. offsets[i] + info[i].frames) <- the mean
 *                         over channels (sum in channel order, then / channels); peaks[i] (optional) <- max |x| before
 *                         normalisation (NaN if any sample is); with ADT_WAV_NORMALIZE every sample is then divided by that
 *                         peak (a silent file becomes NaNs, as in the reference -- callers that skip silent files read peaks).
 *                         `out` is host memory (pinned, when a host-to-device copy follows); a file that changed since the
 *                         probe gets its status overwritten.
 *   adt_copy_files        dst[i] <- src[i]: contents, permission bits and access / modification times (shutil.copy2 without
 *                         extended attributes); status[i] = ADT_OK or ADT_EINVAL.  Destination directories must exist; every
 *                         destination should appear once.
 * `threads`: size of the pool the call spins up (files are handed out dynamically); <= 1 runs inline. */
typedef struct adt_wav_info {
  int64_t frames;        /* samples per channel */
  int64_t data_offset;   /* byte offset of the data chunk's payload */
  int64_t data_bytes;
  int32_t sample_rate;
  int32_t channels;
  int32_t format;        /* 1 = integer PCM, 3 = IEEE float (after resolving WAVE_FORMAT_EXTENSIBLE) */
  int32_t bits;
  int32_t status;
  int32_t reserved;
} adt_wav_info;
#define ADT_WAV_NORMALIZE 1
int adt_wav_probe_batch(const char* const* paths, int32_t n, int32_t threads, adt_wav_info* info);
int adt_wav_decode_batch(const char* const* paths, int32_t n, int32_t threads, adt_wav_info* info, const int64_t* offsets,
                         int32_t flags, float* out, float* peaks);
int adt_copy_files(const char* const* src, const char* const* dst, int32_t n, int32_t threads, int32_t* status);

#ifdef __cplusplus
}
#endif
#endif /* ADT_HIP_H */
