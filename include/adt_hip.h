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

/* ABI version: bumped whenever a signature below changes. */
int adt_version(void);

/* Message of the last failing call on this thread ("" if none). */
const char* adt_last_error(void);

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
 *             (state dict: ...mel_scale.fb); mel_w holds the non-zero weights
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

#ifdef __cplusplus
}
#endif
#endif /* ADT_HIP_H */
