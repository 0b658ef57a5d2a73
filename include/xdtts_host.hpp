// xdtts_host.hpp -- header-only C++ mirror of the reference's Rust surface for the hot path, over
// the C ABI of include/xdtts.h.  (The reference's host language is Rust; this image has no Rust
// toolchain, so the host-side mirror is C++.  INTEGRATION.md holds the equivalent Rust shim.)
//
//   xdtts::Tacotron2::load(path)            -- Tacotron2::load,  src/tacotron2/mod.rs:242
//   xdtts::Tacotron2::infer(units)          -- Tacotron2::infer, src/tacotron2/mod.rs:398
//   xdtts::create_mel_filter_bank(...)      -- griffin_lim::mel, src/tacotron2/mod.rs:453
//   xdtts::GriffinLim(basis, noverlap, power, iter, momentum) / infer(mel)
//                                           -- src/tacotron2/mod.rs:456, src/lib.rs:141
//   xdtts::create_griffin_lim()             -- src/tacotron2/mod.rs:441-458
// Errors become xdtts::Error (the shim's anyhow::Error); nothing here computes on the CPU.
#pragma once
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "xdtts.h"

namespace xdtts {

struct Error : std::runtime_error {
  xdtts_status status;
  Error(xdtts_status s, const char *what) : std::runtime_error(what), status(s) {}
};
inline void check(xdtts_status s) {
  if (s != XDTTS_OK) throw Error(s, xdtts_last_error());
}

// Array2<f32> stand-in: row-major (rows x cols)
struct Array2 {
  size_t rows = 0, cols = 0;
  std::vector<float> data;
  float &operator()(size_t r, size_t c) { return data[r * cols + c]; }
  float operator()(size_t r, size_t c) const { return data[r * cols + c]; }
};

// A unit is carried as its token text ("IH0", " ", ".", "a"), i.e. Unit::from_str's input.
struct Unit {
  std::string token;
  bool is_character = false;  // Unit::Character(c) built directly (grapheme input)
};

class Tacotron2 {
 public:
  // (device_id: XDTTS_DEVICE_DEFAULT = the process's default GPU, environment variable XDTTS_DEVICE -- the reference's load(path) takes none)
  static Tacotron2 load(const std::string &path, int device_id = XDTTS_DEVICE_DEFAULT) {
    xdtts_tacotron2 *h = nullptr;
    check(xdtts_tacotron2_load(path.c_str(), device_id, &h));
    return Tacotron2(h);
  }
  static Tacotron2 synthetic(uint32_t seed = 20240327u, float rec_scale = 1.0f, int device_id = XDTTS_DEVICE_DEFAULT) {
    xdtts_tacotron2 *h = nullptr;
    check(xdtts_tacotron2_load_synthetic(seed, rec_scale, device_id, &h));
    return Tacotron2(h);
  }
  Tacotron2(Tacotron2 &&o) noexcept : h_(std::exchange(o.h_, nullptr)) {}
  Tacotron2 &operator=(Tacotron2 &&o) noexcept {
    std::swap(h_, o.h_);
    return *this;
  }
  Tacotron2(const Tacotron2 &) = delete;
  Tacotron2 &operator=(const Tacotron2 &) = delete;
  ~Tacotron2() { xdtts_tacotron2_free(h_); }

  // src/tacotron2/mod.rs:398-437: find_splits(units, 100); units with no id are dropped (:403-406);
  // chunks are inferred independently and concatenated on the time axis.
  Array2 infer(const std::vector<Unit> &units, const xdtts_infer_opts *opts = nullptr) const {
    std::vector<int64_t> ids;
    for (const Unit &u : units) {
      const int64_t id = xdtts_unit_id(u.token.c_str(), u.is_character ? 1 : 0);
      if (id >= 0) ids.push_back(id);
    }
    xdtts_infer_opts o;
    xdtts_infer_opts_default(&o);
    if (opts) o = *opts;
    std::vector<size_t> splits(ids.size() + 2);
    size_t n_splits = 0;
    check(xdtts_find_splits(ids.data(), ids.size(), (size_t)o.max_chunk, splits.data(), splits.size(), &n_splits));
    float *mel = nullptr;
    size_t frames = 0;
    check(xdtts_tacotron2_infer_ids(h_, ids.data(), ids.size(), splits.data(), n_splits, &o, &mel, &frames));
    Array2 out;
    out.rows = 80;
    out.cols = frames;
    out.data.assign(mel, mel + 80 * frames);
    xdtts_free(mel);
    return out;
  }
  xdtts_tacotron2 *raw() const { return h_; }

 private:
  explicit Tacotron2(xdtts_tacotron2 *h) : h_(h) {}
  xdtts_tacotron2 *h_ = nullptr;
};

// create_mel_filter_bank(sample_rate, n_fft, n_mels, fmin, fmax: Option<f32>); NaN = None
inline Array2 create_mel_filter_bank(float sample_rate, size_t n_fft, size_t n_mels, float fmin, float fmax = NAN) {
  Array2 b;
  b.rows = n_mels;
  b.cols = n_fft / 2 + 1;
  b.data.resize(b.rows * b.cols);
  check(xdtts_mel_filter_bank(sample_rate, n_fft, n_mels, fmin, fmax, b.data.data()));
  return b;
}

class GriffinLim {
 public:
  GriffinLim(const Array2 &mel_basis, size_t noverlap, float power, size_t iter, float momentum, int device_id = XDTTS_DEVICE_DEFAULT) {
    check(xdtts_griffinlim_new(mel_basis.data.data(), mel_basis.rows, mel_basis.cols, noverlap, power, iter, momentum,
                               device_id, &g_));
  }
  GriffinLim(GriffinLim &&o) noexcept : g_(std::exchange(o.g_, nullptr)) {}
  GriffinLim(const GriffinLim &) = delete;
  GriffinLim &operator=(const GriffinLim &) = delete;
  ~GriffinLim() { xdtts_griffinlim_free(g_); }

  std::vector<float> infer(const Array2 &mel) const {
    float *audio = nullptr;
    size_t n = 0;
    check(xdtts_griffinlim_infer(g_, mel.data.data(), mel.rows, mel.cols, &audio, &n));
    std::vector<float> out(audio, audio + n);
    xdtts_free(audio);
    return out;
  }
  // The conventions of the crate's mel->linear step as switches (xdtts_griffinlim_opts)
  void set_opts(const xdtts_griffinlim_opts &o) { check(xdtts_griffinlim_set_opts(g_, &o)); }
  xdtts_griffinlim_opts opts() const {
    xdtts_griffinlim_opts o;
    check(xdtts_griffinlim_get_opts(g_, &o));
    return o;
  }
  xdtts_griffinlim *raw() const { return g_; }

 private:
  xdtts_griffinlim *g_ = nullptr;
};

// src/tacotron2/mod.rs:441-458
inline GriffinLim create_griffin_lim(int device_id = XDTTS_DEVICE_DEFAULT) {
  const Array2 mel_basis = create_mel_filter_bank(22050.0f, 1024, 80, 0.0f, 8000.0f);
  return GriffinLim(mel_basis, 1024 - 256, 1.7f, 30, 0.99f, device_id);
}

// XdTts::infer (src/lib.rs:110-159) for several utterances in one call (xdtts_synthesize_batch): units -> ids
// (units with no id are dropped, mod.rs:403-406), find_splits per utterance (mod.rs:399,412-414), all chunks in one
// lock-step batch, the mel kept in HBM between mel-gen and vocoder.  Returns (mel, audio) per utterance.
inline std::vector<std::pair<Array2, std::vector<float>>> infer_many(const Tacotron2 &model, const GriffinLim &vocoder,
                                                                       const std::vector<std::vector<Unit>> &texts,
                                                                       const xdtts_infer_opts *opts = nullptr) {
  xdtts_infer_opts o;
  xdtts_infer_opts_default(&o);
  if (opts) o = *opts;
  const size_t T = (size_t)o.max_chunk;
  std::vector<int64_t> ids;      // [B][T], zero-padded
  std::vector<int32_t> lens, utt_chunks;
  for (const std::vector<Unit> &units : texts) {
    std::vector<int64_t> u;
    for (const Unit &x : units) {
      const int64_t id = xdtts_unit_id(x.token.c_str(), x.is_character ? 1 : 0);
      if (id >= 0) u.push_back(id);
    }
    std::vector<size_t> splits(u.size() + 2);
    size_t n_splits = 0;
    check(xdtts_find_splits(u.data(), u.size(), T, splits.data(), splits.size(), &n_splits));
    splits.resize(n_splits);
    if (splits.empty() || splits.back() != u.size()) splits.push_back(u.size());  // the trailing split (mod.rs:412-414)
    int32_t n = 0;
    size_t a = 0;
    for (size_t e : splits) {
      if (e <= a) continue;
      if (e - a > T) throw std::runtime_error("xdtts: a chunk exceeds the encoder window");
      ids.resize(ids.size() + T, 0);
      std::copy(u.begin() + (long)a, u.begin() + (long)e, ids.end() - (long)T);
      lens.push_back((int32_t)(e - a));
      a = e;
      ++n;
    }
    utt_chunks.push_back(n);
  }
  const size_t n_utt = texts.size();
  std::vector<float *> mels(n_utt, nullptr), audios(n_utt, nullptr);
  std::vector<size_t> nf(n_utt, 0), ns(n_utt, 0);
  check(xdtts_synthesize_batch(model.raw(), vocoder.raw(), ids.data(), lens.data(), (int32_t)lens.size(), (int32_t)T, utt_chunks.data(),
                               (int32_t)n_utt, &o, nullptr, mels.data(), nf.data(), audios.data(), ns.data()));
  std::vector<std::pair<Array2, std::vector<float>>> out(n_utt);
  for (size_t i = 0; i < n_utt; ++i) {
    out[i].first.rows = 80;
    out[i].first.cols = nf[i];
    out[i].first.data.assign(mels[i], mels[i] + 80 * nf[i]);
    out[i].second.assign(audios[i], audios[i] + ns[i]);
    xdtts_free(mels[i]);
    xdtts_free(audios[i]);
  }
  return out;
}

// XdTts::infer for a sequence of sentences, each decoded alone as the reference does (a loop over src/lib.rs:122-141), the vocoder
// of one overlapped with the encoder of the next (xdtts_synthesize_sequence): what `app` would call for a text of several sentences.
inline std::vector<std::pair<Array2, std::vector<float>>> infer_sequence(const Tacotron2 &model, const GriffinLim &vocoder,
                                                                           const std::vector<std::vector<Unit>> &texts,
                                                                           const xdtts_infer_opts *opts = nullptr) {
  xdtts_infer_opts o;
  xdtts_infer_opts_default(&o);
  if (opts) o = *opts;
  const size_t n_utt = texts.size();
  std::vector<std::vector<int64_t>> ids(n_utt);
  std::vector<std::vector<size_t>> splits(n_utt);
  std::vector<const int64_t *> ids_p(n_utt);
  std::vector<const size_t *> sp_p(n_utt);
  std::vector<size_t> n_ids(n_utt), n_sp(n_utt);
  for (size_t u = 0; u < n_utt; ++u) {
    for (const Unit &x : texts[u]) {
      const int64_t id = xdtts_unit_id(x.token.c_str(), x.is_character ? 1 : 0);
      if (id >= 0) ids[u].push_back(id);  // units with no id are dropped, mod.rs:403-406
    }
    splits[u].resize(ids[u].size() + 2);
    size_t n = 0;
    check(xdtts_find_splits(ids[u].data(), ids[u].size(), (size_t)o.max_chunk, splits[u].data(), splits[u].size(), &n));
    splits[u].resize(n);
    ids_p[u] = ids[u].data();
    n_ids[u] = ids[u].size();
    sp_p[u] = splits[u].data();
    n_sp[u] = n;
  }
  std::vector<float *> mels(n_utt, nullptr), audios(n_utt, nullptr);
  std::vector<size_t> nf(n_utt, 0), ns(n_utt, 0);
  check(xdtts_synthesize_sequence(model.raw(), vocoder.raw(), ids_p.data(), n_ids.data(), sp_p.data(), n_sp.data(), (int32_t)n_utt, &o, mels.data(),
                                  nf.data(), audios.data(), ns.data()));
  std::vector<std::pair<Array2, std::vector<float>>> out(n_utt);
  for (size_t i = 0; i < n_utt; ++i) {
    out[i].first.rows = 80;
    out[i].first.cols = nf[i];
    out[i].first.data.assign(mels[i], mels[i] + 80 * nf[i]);
    out[i].second.assign(audios[i], audios[i] + ns[i]);
    xdtts_free(mels[i]);
    xdtts_free(audios[i]);
  }
  return out;
}

// XdTts::infer's output stage (src/lib.rs:145-157): RTF, `(sample * i16::MAX as f32) as i16`,
// mono 22050 Hz 16-bit WAV (WAV_SPEC, src/lib.rs:25-30).
inline std::vector<int16_t> to_i16(const std::vector<float> &audio) {
  std::vector<int16_t> pcm(audio.size());
  check(xdtts_audio_to_i16(audio.data(), audio.size(), pcm.data()));
  return pcm;
}
inline void write_wav(const std::string &path, const std::vector<float> &audio) {
  const std::vector<int16_t> pcm = to_i16(audio);
  check(xdtts_wav_write(path.c_str(), pcm.data(), pcm.size(), XDTTS_SAMPLE_RATE));
}
inline void write_npy(const std::string &path, const Array2 &mel) {  // src/lib.rs:128-141
  check(xdtts_npy_write_f32(path.c_str(), mel.data.data(), mel.rows, mel.cols));
}

}  // namespace xdtts
