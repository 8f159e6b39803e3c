// Host emulation of the second-generation K1 phase bodies (adt_str_amd/csrc/logmel2_phases.h): every phase runs for lanes
// 0..63 in sequence on a heap "LDS" buffer, with the kernel's own ordering of loads and stores (all loads of a phase before its
// stores where the kernel puts a wave barrier between them).  Test scaffolding only -- it checks the even / odd packing, the
// three radix passes and their LDS layouts, the untangling and the banded mel reduction on a machine without a GPU.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../adt_str_amd/csrc/logmel2_phases.h"

using namespace adt;

extern "C" int emu_logmel2(const float* wave, long n_clips, int n_samples, long ld_wave, int hop, int frame_lo,
                           int n_out, const float* window, const int32_t* mel_meta, const float* mel_w, int n_mels,
                           float eps, float lo, float hi, float* out) {
  std::vector<cf> tw(1024), t2k(520);
  for (int j = 0; j < 1024; ++j) {
    const double a = M_PI * j / 512.0;
    tw[j] = cf{static_cast<float>(std::cos(a)), static_cast<float>(-std::sin(a))};
  }
  for (int j = 0; j <= 512; ++j) {
    const double a = M_PI * j / 1024.0;
    t2k[j] = cf{static_cast<float>(std::cos(a)), static_cast<float>(-std::sin(a))};
  }
  std::vector<cf> win2(1024);
  for (int m = 0; m < 1024; ++m) win2[m] = cf{window[2 * m], window[2 * m + 1]};
  std::vector<cf> buf(kL2Buf + 256);
  for (long b = 0; b < n_clips; ++b) {
    const float* clip = wave + b * ld_wave;
    for (int f = 0; f < n_out; ++f) {
      const int base = (frame_lo + f) * hop - 1024;
      const bool interior = base >= 0 && base + 2048 <= n_samples;
      for (int lane = 0; lane < 64; ++lane) {
        if (interior && (f & 1)) l2_pass1<true, false>(lane, clip, n_samples, base, win2.data(), tw.data(), buf.data());
        else if (interior) {
          if (base & 1) l2_pass1<true, false>(lane, clip, n_samples, base, win2.data(), tw.data(), buf.data());
          else l2_pass1<true, true>(lane, clip, n_samples, base, win2.data(), tw.data(), buf.data());     // (unaligned 8-byte reads are fine on the host)
        } else l2_pass1<false, false>(lane, clip, n_samples, base, win2.data(), tw.data(), buf.data());
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
      std::vector<float> pk(64 * 8), pnk(64 * 8), p512(64);
      for (int lane = 0; lane < 64; ++lane) l2_untangle_load(lane, t2k.data(), buf.data(), &pk[lane * 8], &pnk[lane * 8], p512[lane]);
      float* pw = reinterpret_cast<float*>(buf.data());
      for (int lane = 0; lane < 64; ++lane) l2_untangle_store(lane, &pk[lane * 8], &pnk[lane * 8], p512[lane], pw);
      for (int g = 0; g < 16; ++g)
        for (int i = 0; i < 8; ++i) {
          const int j = g + 16 * i;
          if (j >= n_mels) continue;
          float acc = 0.f;
          for (int s = 0; s < 4; ++s) acc += l2_mel_partial(s, mel_meta[4 * j], mel_meta[4 * j + 1], mel_meta[4 * j + 2], mel_w, pw);
          out[(b * n_out + f) * n_mels + j] = post_fast(acc, eps, lo, hi);
        }
    }
  }
  return 0;
}

// Byte addresses (inside the wave's buffer) of every wave-wide LDS access of the phases, for the bank-conflict check in
// tests/test_logmel_emu.py: fills addr[access][lane] and kind[access] (0 = 8-byte read, 1 = 8-byte write, 2 = 16-byte read,
// 3 = 4-byte write, 4 = 4-byte read), returns the number of accesses.
extern "C" int emu_logmel2_accesses(int* addr, int* kind, int max_acc) {
  int n = 0;
  auto put = [&](int k, int (*f)(int lane, int a, int b), int a, int b) {
    if (n >= max_acc) return;
    for (int lane = 0; lane < 64; ++lane) addr[n * 64 + lane] = f(lane, a, b);
    kind[n++] = k;
  };
  for (int k1 = 0; k1 < 16; ++k1) put(1, [](int l, int k1_, int) { return 8 * l1_index(l >> 3, k1_, l & 7); }, k1, 0);                    // pass 1 stores
  for (int n1 = 0; n1 < 16; ++n1) put(0, [](int l, int n1_, int) { return 8 * (l + 64 * n1_); }, n1, 0);                                   // window reads
  for (int it = 0; it < 2; ++it)
    for (int n2 = 0; n2 < 8; ++n2) put(0, [](int l, int it_, int n2_) { const int c = l + 64 * it_; return 8 * l1_index(n2_, c >> 3, c & 7); }, it, n2);   // pass 2 loads
  for (int it = 0; it < 2; ++it)
    for (int k2 = 0; k2 < 8; ++k2) put(1, [](int l, int it_, int k2_) { const int c = l + 64 * it_; return 8 * l2_index(k2_, c >> 3, c & 7); }, it, k2);   // pass 2 stores
  for (int it = 0; it < 2; ++it)
    for (int j = 0; j < 4; ++j)
      put(2, [](int l, int it_, int j_) { const int k1 = l & 15, k2 = (l >> 4) + 4 * it_; return 8 * (k2 * 128 + 8 * k1 + 2 * ((j_ + (k1 >> 2)) & 3)); }, it, j);   // pass 3 loads
  for (int it = 0; it < 2; ++it)
    for (int k3 = 0; k3 < 8; ++k3) put(1, [](int l, int it_, int k3_) { return 8 * ((l & 15) + 16 * ((l >> 4) + 4 * it_) + 128 * k3_); }, it, k3);       // pass 3 stores
  for (int i = 0; i < 8; ++i) {
    put(0, [](int l, int i_, int) { return 8 * (l + 64 * i_); }, i, 0);                                                                    // untangle: Z[k], W^k
    put(0, [](int l, int i_, int) { return 8 * ((1024 - (l + 64 * i_)) & 1023); }, i, 0);                                                  //           Z[1024 - k]
    put(3, [](int l, int i_, int) { return 4 * (l + 64 * i_); }, i, 0);                                                                    //           power stores
    put(3, [](int l, int i_, int) { return 4 * (1024 - (l + 64 * i_)); }, i, 0);
  }
  return n;
}
