// Host emulation of K9's phase bodies (adt_str_amd/csrc/fft1024_phases.h); see logmel2_emu.cpp.
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../adt_str_amd/csrc/fft1024_phases.h"

using namespace adt;

extern "C" int emu_clap_logmel(const float* waves, const int64_t* offsets, long n_clips, int target, int hop, int n_frames,
                               const float* window, const int32_t* mel_meta, const float* mel_w, int n_mels, float amin, float* out) {
  std::vector<cf> tw(1024);                 // W_1024^j, the whole circle (second generation: pass 1 into L1, K1's passes 2 / 3)
  for (int j = 0; j < 1024; ++j) {
    const double a = M_PI * j / 512.0;
    tw[j] = cf{static_cast<float>(std::cos(a)), static_cast<float>(-std::sin(a))};
  }
  std::vector<cf> buf(kL2Buf);
  const int pairs = (n_frames + 1) / 2;
  for (long b = 0; b < n_clips; ++b) {
    const float* clip = waves + offsets[b];
    const int n = static_cast<int>(offsets[b + 1] - offsets[b]);
    for (int p = 0; p < pairs; ++p) {
      const int f0 = 2 * p;
      const bool has1 = f0 + 1 < n_frames;
      const int base0 = f0 * hop - kN1k / 2, base1 = base0 + hop;
      const bool interior = base0 >= 0 && base1 + kN1k <= target;
      for (int lane = 0; lane < 64; ++lane) {
        float win16[16];
        for (int n1 = 0; n1 < 16; ++n1) win16[n1] = window[lane + 64 * n1];
        if (interior) p1k_pass1_l1<true>(lane, clip, n, target, base0, base1, has1, win16, tw.data(), buf.data());
        else p1k_pass1_l1<false>(lane, clip, n, target, base0, base1, has1, win16, tw.data(), buf.data());
      }
      std::vector<cf> z(64 * 2 * 8);
      for (int it = 0; it < 2; ++it)
        for (int lane = 0; lane < 64; ++lane) l2_pass2_load(lane, it, buf.data(), &z[(lane * 2 + it) * 8]);
      for (int it = 0; it < 2; ++it)
        for (int lane = 0; lane < 64; ++lane) l2_pass2_store(lane, it, &z[(lane * 2 + it) * 8], tw.data(), buf.data());
      for (int it = 0; it < 2; ++it)
        for (int lane = 0; lane < 64; ++lane) l2_pass3_load(lane, it, buf.data(), &z[(lane * 2 + it) * 8]);
      for (int it = 0; it < 2; ++it)
        for (int lane = 0; lane < 64; ++lane) l2_pass3_store(lane, it, &z[(lane * 2 + it) * 8], buf.data());
      std::vector<cf> snap(buf);
      for (int lane = 0; lane < 64; ++lane) {
        std::vector<cf> tmp(snap);
        p1k_untangle(lane, tmp.data());
        for (int i = 0; i < 8; ++i) buf[lane + 64 * i] = tmp[lane + 64 * i];
        if (lane == 0) buf[512] = tmp[512];
      }
      for (int j = 0; j < n_mels; ++j) {
        cf acc = {0.f, 0.f};
        for (int s = 0; s < 4; ++s) {
          cf part = mel_partial(s, mel_meta[4 * j], mel_meta[4 * j + 1], mel_meta[4 * j + 2], mel_w, buf.data());
          acc.x += part.x; acc.y += part.y;
        }
        out[(b * n_frames + f0) * n_mels + j] = to_db(acc.x, amin);
        if (has1) out[(b * n_frames + f0 + 1) * n_mels + j] = to_db(acc.y, amin);
      }
    }
  }
  return 0;
}
