// Host emulation of K1's phase bodies (adt_str_amd/csrc/logmel_phases.h): runs every
// phase for lanes 0..63 in sequence on a heap "LDS" buffer.  Test scaffolding only --
// it checks the FFT index maps, twiddles, untangling and the banded mel reduction on a
// machine without a GPU.  Built by tests/test_logmel_emu.py with g++.
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../adt_str_amd/csrc/logmel_phases.h"

using namespace adt;

extern "C" int emu_logmel(const float* wave, long n_clips, int n_samples, long ld_wave, int hop, int frame_lo,
                          int n_out, const float* window, const int32_t* mel_meta, const float* mel_w, int n_mels,
                          float eps, float lo, float hi, float* out) {
  std::vector<cf> tw(1024);
  for (int j = 0; j < 1024; ++j) {
    const double a = M_PI * j / 1024.0;
    tw[j] = cf{static_cast<float>(std::cos(a)), static_cast<float>(-std::sin(a))};
  }
  std::vector<cf> buf(kBufElems);
  const int pairs = (n_out + 1) / 2;
  for (long b = 0; b < n_clips; ++b) {
    const float* clip = wave + b * ld_wave;
    for (int p = 0; p < pairs; ++p) {
      const int f0 = 2 * p;
      const bool has1 = f0 + 1 < n_out;
      const int base0 = (frame_lo + f0) * hop - kNfft / 2, base1 = base0 + hop;
      const bool interior = base0 >= 0 && base1 + kNfft <= n_samples;
      for (int it = 0; it < 2; ++it)
        for (int lane = 0; lane < 64; ++lane) {
          float win16[16];
          for (int n1 = 0; n1 < 16; ++n1) win16[n1] = window[lane + 64 * it + 128 * n1];
          if (interior) pass1<true>(lane, it, clip, n_samples, base0, base1, has1, win16, tw.data(), buf.data());
          else pass1<false>(lane, it, clip, n_samples, base0, base1, has1, win16, tw.data(), buf.data());
        }
      // pass 2 is in place per lane (each lane reads and writes the same 16 slots)
      for (int it = 0; it < 2; ++it)
        for (int lane = 0; lane < 64; ++lane) pass2(lane, it, tw.data(), buf.data());
      std::vector<cf> z(64 * 4 * 8);
      for (int it = 0; it < 4; ++it)
        for (int lane = 0; lane < 64; ++lane) pass3_load(lane, it, buf.data(), &z[(lane * 4 + it) * 8]);
      for (int it = 0; it < 4; ++it)
        for (int lane = 0; lane < 64; ++lane) pass3_store(lane, it, &z[(lane * 4 + it) * 8], buf.data());
      // untangle: all reads before all writes, as the lock-step wave does
      std::vector<cf> snapshot(buf);
      for (int lane = 0; lane < 64; ++lane) {
        std::vector<cf> tmp(snapshot);
        untangle(lane, tmp.data());
        for (int i = 0; i < 16; ++i) buf[lane + 64 * i] = tmp[lane + 64 * i];
        if (lane == 0) buf[1024] = tmp[1024];
      }
      for (int g = 0; g < 16; ++g)
        for (int i = 0; i < 8; ++i) {
          const int j = g + 16 * i;
          if (j >= n_mels) continue;
          cf acc = {0.f, 0.f};
          for (int s = 0; s < 4; ++s) {
            cf part = mel_partial(s, mel_meta[4 * j], mel_meta[4 * j + 1], mel_meta[4 * j + 2], mel_w, buf.data());
            acc.x += part.x; acc.y += part.y;
          }
          out[(b * n_out + f0) * n_mels + j] = post(acc.x, eps, lo, hi);
          if (has1) out[(b * n_out + f0 + 1) * n_mels + j] = post(acc.y, eps, lo, hi);
        }
    }
  }
  return 0;
}
