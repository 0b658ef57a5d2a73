// host.cpp -- host-side front of Tacotron2::infer that stays on the CPU side of the FFI:
// the Unit -> id table (generate_id_list, src/tacotron2/mod.rs:90-122), the id lookup
// (best_match_for_unit, src/phonemes.rs:627-660) and the chunker (find_splits,
// src/phonemes.rs:681-753 with split_score :663-671).  Integer/string work, microseconds; kept in
// C++ because the reference's host language (Rust) has no toolchain in this image.  Also the
// output stage of src/lib.rs (f32 -> i16, WAV, .npy dump, SSML-break silence, RTF).
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "common.h"

namespace {

const char *const PHONES[84] = {
    "AA", "AA0", "AA1", "AA2", "AE", "AE0", "AE1", "AE2", "AH", "AH0", "AH1", "AH2", "AO", "AO0",
    "AO1", "AO2", "AW", "AW0", "AW1", "AW2", "AY", "AY0", "AY1", "AY2", "B",  "CH",  "D",  "DH",
    "EH", "EH0", "EH1", "EH2", "ER", "ER0", "ER1", "ER2", "EY", "EY0", "EY1", "EY2", "F",  "G",
    "HH", "IH",  "IH0", "IH1", "IH2", "IY", "IY0", "IY1", "IY2", "JH", "K",  "L",   "M",  "N",
    "NG", "OW",  "OW0", "OW1", "OW2", "OY", "OY0", "OY1", "OY2", "P",  "R",  "S",   "SH", "T",
    "TH", "UH",  "UH0", "UH1", "UH2", "UW", "UW0", "UW1", "UW2", "V",  "W",  "Y",   "Z",  "ZH"};

// id order of mod.rs:101-120: pad, 10 punctuation marks, space, A-Z, a-z, 84 ARPAbet symbols
const std::vector<std::string> &symbols() {
  static const std::vector<std::string> s = [] {
    std::vector<std::string> v = {"<PAD>", "-", "!", "'", "(", ")", ",", ".", ":", ";", "?", " "};
    for (char c = 'A'; c <= 'Z'; ++c) v.emplace_back(1, c);
    for (char c = 'a'; c <= 'z'; ++c) v.emplace_back(1, c);
    for (const char *p : PHONES) v.emplace_back(p);
    return v;
  }();
  return s;
}

constexpr int FIRST_CHAR = 12, FIRST_PHONE = 64;

int find_in(const std::string &tok, int lo, int hi) {
  const auto &s = symbols();
  for (int i = lo; i < hi; ++i)
    if (s[i] == tok) return i;
  return -1;
}

// split_score, src/phonemes.rs:663-671
int split_score(int64_t id) {
  switch (id) {
    case 7:   // '.'
    case 10:  // '?'
    case 2:   // '!'
    case 0:   // padding
      return 3;
    case 6:  // ','
    case 9:  // ';'
      return 2;
    case 11:  // space
      return 1;
    default:
      return 0;
  }
}

}  // namespace

extern "C" {

int32_t xdtts_symbol_count(void) { return (int32_t)symbols().size(); }

const char *xdtts_symbol_token(int32_t id) {
  return id >= 0 && id < xdtts_symbol_count() ? symbols()[(size_t)id].c_str() : nullptr;
}

// Unit::from_str (src/phonemes.rs:450-487) followed by best_match_for_unit (:627-660).
// as_character != 0 builds Unit::Character(token[0]) directly, as convert_to_units does for
// grapheme input.  Returns -1 where the reference finds no id and silently drops the unit
// (src/tacotron2/mod.rs:403-406).
int64_t xdtts_unit_id(const char *token, int32_t as_character) {
  if (!token || !*token) return -1;
  std::string raw(token);
  if (as_character) {
    if (raw == " ") return 11;
    return raw.size() == 1 ? find_in(raw, FIRST_CHAR, FIRST_PHONE) : -1;
  }
  size_t a = 0, b = raw.size();
  while (a < b && std::isspace((unsigned char)raw[a])) ++a;
  while (b > a && std::isspace((unsigned char)raw[b - 1])) --b;
  const std::string t = raw.substr(a, b - a);
  if (t.empty()) return 11;  // "" if !s.is_empty() => Unit::Space
  if (t == "<PAD>") return 0;
  if (t == "<UNK>") return -1;
  const int punct = find_in(t, 1, 11);
  if (punct >= 0) return punct;
  // ARPAbet is tried before the single-character fallback (phonemes.rs:469-482)
  int id = find_in(t, FIRST_PHONE, xdtts_symbol_count());
  if (id >= 0) return id;
  if (t.size() >= 2) {
    // a known phone with a stress/auxiliary mark the table lacks: best_match keeps the first
    // entry of that phone (the un-marked one)
    size_t cut = t.size();
    while (cut > 0 && !std::isalpha((unsigned char)t[cut - 1])) --cut;
    if (cut > 0 && cut < t.size()) {
      id = find_in(t.substr(0, cut), FIRST_PHONE, xdtts_symbol_count());
      if (id >= 0) return id;
    }
  }
  if (t.size() == 1) return find_in(t, FIRST_CHAR, FIRST_PHONE);
  return -1;
}

int32_t xdtts_split_score(int64_t id) { return split_score(id); }

// find_splits(units, max_size), src/phonemes.rs:681-753, on the id sequence.  Writes up to `cap`
// split indices (ascending) to out and the count to *n_out.
xdtts_status xdtts_find_splits(const int64_t *ids, size_t n, size_t max_size, size_t *out, size_t cap,
                               size_t *n_out) {
  if ((!ids && n) || !n_out || (!out && cap)) {
    xdtts::set_last_error("find_splits: null argument");
    return XDTTS_ERR_BAD_ARG;
  }
  std::vector<std::pair<size_t, int>> marks;  // positions where a split is allowed
  for (size_t i = 0; i < n; ++i) {
    const int s = split_score(ids[i]);
    if (s > 0) marks.emplace_back(i, s);
  }
  std::vector<size_t> results{0};
  for (const auto &m : marks)
    if (m.second > 2) results.push_back(m.first);
  int threshold = 1;
  bool scan = true;
  std::vector<size_t> fresh;
  while (scan) {
    scan = false;
    size_t last_ref = n;
    for (size_t r = results.size(); r-- > 0;) {
      const size_t index = results[r];
      if (last_ref - index > max_size) {
        scan = true;
        for (const auto &m : marks)
          if (m.first < last_ref && m.first > index + 1 && m.second > threshold) fresh.push_back(m.first);
      }
      last_ref = index;
    }
    if (scan) {
      results.insert(results.end(), fresh.begin(), fresh.end());
      fresh.clear();
      std::sort(results.begin(), results.end());
    }
    if (threshold > 0)
      --threshold;
    else
      scan = false;
  }
  // merge neighbours that were broken up smaller than needed
  std::vector<size_t> merged;
  size_t running = 0, last_insert = 0;
  for (size_t i : results) {
    if ((i - last_insert) + running > max_size) {
      merged.push_back(last_insert);
      running = i - last_insert;
    } else {
      running += i - last_insert;
    }
    last_insert = i;
  }
  if (running + (n - last_insert) > max_size && !results.empty()) merged.push_back(results.back());
  merged.erase(std::unique(merged.begin(), merged.end()), merged.end());
  *n_out = merged.size();
  if (merged.size() > cap) {
    xdtts::set_last_error("find_splits: output buffer too small");
    return XDTTS_ERR_BAD_ARG;
  }
  std::copy(merged.begin(), merged.end(), out);
  return XDTTS_OK;
}

// ---- output stage: src/lib.rs:25-30 (WAV_SPEC), :128-141 (.npy dump), :145-157 (RTF, i16), :162-176 ----

// `(*sample * i16::MAX as f32) as i16`: Rust's `as` truncates toward zero, saturates, NaN -> 0.
xdtts_status xdtts_audio_to_i16(const float *audio, size_t n, int16_t *pcm) {
  if ((!audio || !pcm) && n) {
    xdtts::set_last_error("audio_to_i16: null argument");
    return XDTTS_ERR_BAD_ARG;
  }
  for (size_t i = 0; i < n; ++i) {
    const float v = audio[i] * 32767.0f;
    int16_t q;
    if (v != v) q = 0;
    else if (v >= 32767.0f) q = 32767;
    else if (v <= -32768.0f) q = -32768;
    else q = (int16_t)v;  // C++ float -> int conversion truncates toward zero, like Rust's in range
    pcm[i] = q;
  }
  return XDTTS_OK;
}

size_t xdtts_silence_samples(double seconds, uint32_t sample_rate) {
  // `(sample_rate as f32 * duration.as_secs_f32()).round() as u32`, src/lib.rs:166 -- the product
  // and the rounding are f32 operations (volatile: no contraction or wider evaluation)
  volatile float sr = (float)sample_rate, secs = (float)seconds;
  volatile float prod = sr * secs;
  const float v = std::round(prod);
  return v > 0.f ? (size_t)v : 0;
}

size_t xdtts_silence_samples_duration(uint64_t secs, uint32_t nanos, uint32_t sample_rate) {
  // Duration::as_secs_f32 (Rust std): `(secs as f32) + (nanos as f32) / (NANOS_PER_SEC as f32)`,
  // every operation in f32; then src/lib.rs:166 as above.
  volatile float fs = (float)secs, fn = (float)nanos;
  volatile float frac = fn / 1000000000.0f;
  volatile float dur = fs + frac;
  volatile float sr = (float)sample_rate;
  volatile float prod = sr * dur;
  const float v = std::round(prod);
  return v > 0.f ? (size_t)v : 0;
}

static bool put(FILE *f, const void *p, size_t n) { return std::fwrite(p, 1, n, f) == n; }

xdtts_status xdtts_wav_write(const char *path, const int16_t *pcm, size_t n, uint32_t sample_rate) {
  if (!path || (!pcm && n) || n > 0x7fffffffu / 2) {
    xdtts::set_last_error("wav_write: bad argument");
    return XDTTS_ERR_BAD_ARG;
  }
  FILE *f = std::fopen(path, "wb");
  if (!f) {
    xdtts::set_last_error("wav_write: cannot open output file");
    return XDTTS_ERR_IO;
  }
  const uint32_t data = (uint32_t)(n * 2), riff = 36 + data, fmt_len = 16, byte_rate = sample_rate * 2;
  const uint16_t pcm_tag = 1, channels = 1, block = 2, bits = 16;
  bool ok = put(f, "RIFF", 4) && put(f, &riff, 4) && put(f, "WAVEfmt ", 8) && put(f, &fmt_len, 4) && put(f, &pcm_tag, 2) &&
            put(f, &channels, 2) && put(f, &sample_rate, 4) && put(f, &byte_rate, 4) && put(f, &block, 2) &&
            put(f, &bits, 2) && put(f, "data", 4) && put(f, &data, 4) && (n == 0 || put(f, pcm, n * 2));
  ok = (std::fclose(f) == 0) && ok;
  if (!ok) {
    xdtts::set_last_error("wav_write: short write");
    return XDTTS_ERR_IO;
  }
  return XDTTS_OK;
}

xdtts_status xdtts_npy_write_f32(const char *path, const float *data, size_t rows, size_t cols) {
  if (!path || (!data && rows * cols)) {
    xdtts::set_last_error("npy_write: null argument");
    return XDTTS_ERR_BAD_ARG;
  }
  char dict[128];
  int len = std::snprintf(dict, sizeof dict, "{'descr': '<f4', 'fortran_order': False, 'shape': (%zu, %zu), }", rows, cols);
  std::string hdr(dict, (size_t)len);
  while ((10 + hdr.size() + 1) % 64 != 0) hdr.push_back(' ');  // magic(6) + version(2) + len(2) + dict + '\n'
  hdr.push_back('\n');
  const uint16_t hlen = (uint16_t)hdr.size();
  FILE *f = std::fopen(path, "wb");
  if (!f) {
    xdtts::set_last_error("npy_write: cannot open output file");
    return XDTTS_ERR_IO;
  }
  bool ok = put(f, "\x93NUMPY\x01\x00", 8) && put(f, &hlen, 2) && put(f, hdr.data(), hdr.size()) &&
            (rows * cols == 0 || put(f, data, rows * cols * sizeof(float)));
  ok = (std::fclose(f) == 0) && ok;
  if (!ok) {
    xdtts::set_last_error("npy_write: short write");
    return XDTTS_ERR_IO;
  }
  return XDTTS_OK;
}

double xdtts_real_time_factor(double compute_seconds, size_t n_samples) {
  return n_samples ? compute_seconds / ((double)n_samples / (double)XDTTS_SAMPLE_RATE) : 0.0;
}

}  // extern "C"
