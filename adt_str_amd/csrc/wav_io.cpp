// wav_io.cpp -- host side of the curation path: batched RIFF/WAVE decode and batched file copies on a thread pool.
//
// The reference reads one file at a time on the thread that also drives the GPU (torchaudio.load -> mean over channels ->
// x / max|x|: data_modules/augment_data_with_CLAP.py:51-68) and copies the chosen files one at a time with shutil.copy2
// (:184-196); convert_augmented_to_hdf5.py:97-103 loads every curated file the same way once more.  At 33k embeddings/s on
// the GPU those per-file Python calls are what the pipeline waits for, so the decode of a whole batch (and the copy of a
// whole plan) is one call here.  No device code: this file is plain C++ linked into libadt_hip.so.
//
// Decoding follows adt_str_amd/audio_io.py:read_wav to the bit (that function is what the tests compare with): chunk walk
// with the LAST "fmt " / "data" chunk winning, WAVE_FORMAT_EXTENSIBLE sub-format, 8/16/24/32-bit PCM and 32-bit float,
// sample / 2^(bits-1), a data chunk cut short by the end of the file is taken as far as it goes; float32 throughout.
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

#include "adt_common.h"

namespace {

using adt::set_error;

template <typename F>
void parallel_for(int n, int threads, F&& body) {
  if (threads < 1) threads = 1;
  if (threads > n) threads = n;
  if (threads <= 1) {
    for (int i = 0; i < n; ++i) body(i);
    return;
  }
  std::atomic<int> next{0};
  std::vector<std::thread> pool;
  pool.reserve(threads);
  for (int t = 0; t < threads; ++t)
    pool.emplace_back([&] {
      for (int i = next.fetch_add(1, std::memory_order_relaxed); i < n; i = next.fetch_add(1, std::memory_order_relaxed)) body(i);
    });
  for (auto& th : pool) th.join();
}

struct Fd {
  int fd;
  explicit Fd(int f) : fd(f) {}
  ~Fd() { if (fd >= 0) close(fd); }
  Fd(const Fd&) = delete;
  Fd& operator=(const Fd&) = delete;
};

bool pread_full(int fd, void* buf, size_t n, off_t off) {
  char* p = static_cast<char*>(buf);
  while (n > 0) {
    const ssize_t r = pread(fd, p, n, off);
    if (r < 0) { if (errno == EINTR) continue; return false; }
    if (r == 0) return false;
    p += r; off += r; n -= static_cast<size_t>(r);
  }
  return true;
}

uint32_t le32(const unsigned char* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | (static_cast<uint32_t>(p[3]) << 24); }
uint16_t le16(const unsigned char* p) { return static_cast<uint16_t>(p[0] | (p[1] << 8)); }

// Fills `w` from the file's chunk list; returns ADT_OK or the error code recorded in w->status.
int probe_one(const char* path, adt_wav_info* w) {
  std::memset(w, 0, sizeof(*w));
  w->status = ADT_EINVAL;
  Fd f(open(path, O_RDONLY | O_CLOEXEC));
  if (f.fd < 0) return w->status;
  struct stat st;
  if (fstat(f.fd, &st) != 0) return w->status;
  const int64_t size = st.st_size;
  unsigned char head[12];
  if (size < 12 || !pread_full(f.fd, head, 12, 0) || std::memcmp(head, "RIFF", 4) != 0 || std::memcmp(head + 8, "WAVE", 4) != 0) return w->status;
  bool have_fmt = false, have_data = false;
  int64_t pos = 12;
  while (pos + 8 <= size) {
    unsigned char ch[8];
    if (!pread_full(f.fd, ch, 8, pos)) return w->status;
    const int64_t csize = le32(ch + 4);
    const int64_t avail = size - (pos + 8) < csize ? size - (pos + 8) : csize;       // a chunk the file cuts short
    if (std::memcmp(ch, "fmt ", 4) == 0) {
      unsigned char b[26];
      if (avail < 16) return w->status;                                               // (struct.error in read_wav)
      const int64_t take = avail < 26 ? avail : 26;
      if (!pread_full(f.fd, b, static_cast<size_t>(take), pos + 8)) return w->status;
      w->format = le16(b);
      w->channels = le16(b + 2);
      w->sample_rate = static_cast<int32_t>(le32(b + 4));
      w->bits = le16(b + 14);
      if (w->format == 0xFFFE && take >= 26) w->format = le16(b + 24);
      have_fmt = true;
    } else if (std::memcmp(ch, "data", 4) == 0) {
      w->data_offset = pos + 8;
      w->data_bytes = avail;
      have_data = true;
    }
    pos += 8 + csize + (csize & 1);
  }
  if (!have_fmt || !have_data) return w->status;
  const bool known = (w->format == 3 && w->bits == 32) || (w->format == 1 && (w->bits == 8 || w->bits == 16 || w->bits == 24 || w->bits == 32));
  if (!known || w->channels <= 0) { w->status = ADT_ESHAPE; return w->status; }
  const int bps = w->bits / 8;
  int64_t samples;
  if (w->bits == 24) samples = w->data_bytes / 3;                      // read_wav drops a trailing partial sample for 24-bit only
  else if (w->data_bytes % bps != 0) { w->status = ADT_ESHAPE; return w->status; }      // (np.frombuffer refuses it)
  else samples = w->data_bytes / bps;
  w->frames = samples / w->channels;
  w->status = ADT_OK;
  return ADT_OK;
}

inline float sample_at(const unsigned char* p, int format, int bits) {
  if (format == 3) { float v; std::memcpy(&v, p, 4); return v; }
  switch (bits) {
    case 16: return static_cast<float>(static_cast<int16_t>(le16(p))) / 32768.0f;
    case 32: return static_cast<float>(static_cast<int32_t>(le32(p))) / 2147483648.0f;
    case 24: {
      int32_t v = p[0] | (p[1] << 8) | (p[2] << 16);
      v = (v ^ 0x800000) - 0x800000;
      return static_cast<float>(v) / 8388608.0f;
    }
    default: return (static_cast<float>(p[0]) - 128.0f) / 128.0f;
  }
}

// Decodes file i into out[0 .. frames): channel mean (sequential sum over channels, then / channels, in float32: numpy's
// mean(axis=0) of the [channels, frames] array) and, when asked, x / max|x| (0 / 0 = NaN for a silent file, like the reference).
int decode_one(const char* path, const adt_wav_info& w, int flags, float* out, float* peak_out, std::vector<unsigned char>& scratch) {
  if (w.status != ADT_OK) return w.status;
  if (w.frames == 0) { if (peak_out) *peak_out = 0.f; return ADT_OK; }
  Fd f(open(path, O_RDONLY | O_CLOEXEC));
  if (f.fd < 0) return ADT_EINVAL;
  const int bps = w.bits / 8, ch = w.channels;
  const size_t need = static_cast<size_t>(w.frames) * ch * bps;
  if (scratch.size() < need) scratch.resize(need);
  if (!pread_full(f.fd, scratch.data(), need, static_cast<off_t>(w.data_offset))) return ADT_EINVAL;
  const unsigned char* p = scratch.data();
  const float inv_count = static_cast<float>(ch);
  float peak = 0.f;
  bool any_nan = false;
  if (ch == 1 && w.format == 1 && w.bits == 16) {                       // the common case, kept tight
    for (int64_t i = 0; i < w.frames; ++i) {
      const float v = static_cast<float>(static_cast<int16_t>(le16(p + 2 * i))) / 32768.0f;
      out[i] = v;
      const float a = std::fabs(v);
      peak = a > peak ? a : peak;
    }
  } else {
    for (int64_t i = 0; i < w.frames; ++i) {
      float s = sample_at(p + (static_cast<size_t>(i) * ch) * bps, w.format, w.bits);
      for (int c = 1; c < ch; ++c) s += sample_at(p + (static_cast<size_t>(i) * ch + c) * bps, w.format, w.bits);
      const float v = s / inv_count;
      out[i] = v;
      const float a = std::fabs(v);
      if (a != a) any_nan = true;
      peak = a > peak ? a : peak;
    }
  }
  if (any_nan) peak = NAN;                                              // torch.max / np.max propagate NaN
  if (peak_out) *peak_out = peak;
  if (flags & ADT_WAV_NORMALIZE)
    for (int64_t i = 0; i < w.frames; ++i) out[i] = out[i] / peak;
  return ADT_OK;
}

// shutil.copy2: contents, permission bits, access / modification times (extended attributes are not carried over).
int copy_one(const char* src, const char* dst) {
  Fd in(open(src, O_RDONLY | O_CLOEXEC));
  if (in.fd < 0) return ADT_EINVAL;
  struct stat st;
  if (fstat(in.fd, &st) != 0 || !S_ISREG(st.st_mode)) return ADT_EINVAL;
  struct stat sd;
  if (stat(dst, &sd) == 0 && sd.st_dev == st.st_dev && sd.st_ino == st.st_ino) return ADT_EINVAL;   // same file (shutil.SameFileError): never truncate the source
  Fd out(open(dst, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666));
  if (out.fd < 0) return ADT_EINVAL;
  int64_t left = st.st_size;
  bool fallback = false;
  while (left > 0) {
    const ssize_t r = copy_file_range(in.fd, nullptr, out.fd, nullptr, static_cast<size_t>(left), 0);
    if (r < 0) { if (errno == EINTR) continue; fallback = true; break; }
    if (r == 0) break;
    left -= r;
  }
  if (!fallback && left > 0) return ADT_EINVAL;                         // the source shrank under the copy: a short file is not a copy
  if (fallback) {                                                       // file systems without copy_file_range (cross-device on old kernels)
    if (lseek(in.fd, 0, SEEK_SET) < 0 || lseek(out.fd, 0, SEEK_SET) < 0 || ftruncate(out.fd, 0) != 0) return ADT_EINVAL;
    std::vector<char> buf(1 << 20);
    for (;;) {
      const ssize_t r = read(in.fd, buf.data(), buf.size());
      if (r < 0) { if (errno == EINTR) continue; return ADT_EINVAL; }
      if (r == 0) break;
      ssize_t done = 0;
      while (done < r) {
        const ssize_t wv = write(out.fd, buf.data() + done, static_cast<size_t>(r - done));
        if (wv < 0) { if (errno == EINTR) continue; return ADT_EINVAL; }
        done += wv;
      }
    }
  }
  const struct timespec times[2] = {st.st_atim, st.st_mtim};
  if (fchmod(out.fd, st.st_mode & 07777) != 0 || futimens(out.fd, times) != 0) return ADT_EINVAL;   // copy2 raises on these too
  return ADT_OK;
}

}  // namespace

extern "C" int adt_wav_probe_batch(const char* const* paths, int32_t n, int32_t threads, adt_wav_info* info) {
  if (n < 0 || (n > 0 && (!paths || !info))) return set_error(ADT_EINVAL, "adt_wav_probe_batch: null pointer");
  parallel_for(n, threads, [&](int i) {
    try { probe_one(paths[i], &info[i]); } catch (...) { info[i].status = ADT_EINVAL; }      // an exception in a worker thread would end the process
  });
  return ADT_OK;
}

extern "C" int adt_wav_decode_batch(const char* const* paths, int32_t n, int32_t threads, adt_wav_info* info, const int64_t* offsets,
                                    int32_t flags, float* out, float* peaks) {
  if (n < 0 || (n > 0 && (!paths || !info || !offsets || !out))) return set_error(ADT_EINVAL, "adt_wav_decode_batch: null pointer");
  for (int i = 0; i < n; ++i)
    if (info[i].status == ADT_OK && (offsets[i] < 0 || info[i].frames < 0)) return set_error(ADT_EINVAL, "adt_wav_decode_batch: negative offset");
  parallel_for(n, threads, [&](int i) {
    thread_local std::vector<unsigned char> scratch;
    if (info[i].status != ADT_OK) { if (peaks) peaks[i] = 0.f; return; }
    int rc;
    try {
      rc = decode_one(paths[i], info[i], flags, out + offsets[i], peaks ? &peaks[i] : nullptr, scratch);
    } catch (...) {                                                     // std::bad_alloc from the scratch buffer of a huge data chunk: this file fails, not the run
      rc = ADT_EINVAL;
      if (peaks) peaks[i] = 0.f;
    }
    if (rc != ADT_OK) info[i].status = rc;                              // the file changed or vanished since the probe
  });
  return ADT_OK;
}

extern "C" int adt_copy_files(const char* const* src, const char* const* dst, int32_t n, int32_t threads, int32_t* status) {
  if (n < 0 || (n > 0 && (!src || !dst || !status))) return set_error(ADT_EINVAL, "adt_copy_files: null pointer");
  parallel_for(n, threads, [&](int i) {
    try { status[i] = copy_one(src[i], dst[i]); } catch (...) { status[i] = ADT_EINVAL; }
  });
  return ADT_OK;
}
