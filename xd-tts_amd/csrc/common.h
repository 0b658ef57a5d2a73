// common.h -- shared declarations of libxdtts_hip.so (MI355X / gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/xdtts.h"

namespace xdtts {

// ---- model constants: src/tacotron2/mod.rs:205-208 + NVIDIA Tacotron2 defaults ------------
constexpr int N_SYMBOLS = 148;  // mod.rs:90-122
constexpr int EMB = 512;        // encoder_embedding_dim, mod.rs:207
constexpr int ENC_CONVS = 3;
constexpr int ENC_K = 5;
constexpr int ENC_H = 256;      // BiLSTM hidden per direction
constexpr int N_MEL = 80;       // mod.rs:208
constexpr int PRENET = 256;
constexpr int ATT_RNN = 1024;   // mod.rs:205
constexpr int DEC_RNN = 1024;   // mod.rs:206
constexpr int ATT_DIM = 128;
constexpr int LOC_F = 32;
constexpr int LOC_K = 31;
constexpr int POST_CONVS = 5;
constexpr int POST_CH = 512;
constexpr int POST_K = 5;
constexpr int T_MAX = 512;      // attention kernel LDS budget
constexpr int ATT_IN = PRENET + EMB;              // 768
constexpr int ATT_COLS = ATT_IN + ATT_RNN;        // 1792 packed columns [W_ih | W_hh]
constexpr int DEC_IN = ATT_RNN + EMB;             // 1536
constexpr int DEC_COLS = DEC_IN + DEC_RNN;        // 2560
constexpr int PROJ_IN = DEC_RNN + EMB;            // 1536

// ---- error plumbing -----------------------------------------------------------------------
struct Error : std::runtime_error {
  xdtts_status code;
  Error(xdtts_status c, const std::string &m) : std::runtime_error(m), code(c) {}
};
void set_last_error(const char *msg);
[[noreturn]] void fail(xdtts_status code, const char *fmt, ...);

#define HIP_CHECK(expr)                                                                       \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      ::xdtts::fail(e_ == hipErrorOutOfMemory ? XDTTS_ERR_OOM : XDTTS_ERR_HIP, "%s: %s (%s:%d)", \
                    #expr, hipGetErrorString(e_), __FILE__, __LINE__);                        \
  } while (0)

// ---- kernels whose workgroups wait for one another -----------------------------------------------
// The persistent decoder, the cooperative encoder BiLSTM and the persistent Griffin-Lim exchange data
// between workgroups inside a launch, so their whole grid must be co-resident.  They are launched with
// hipLaunchCooperativeKernel: the runtime then checks the grid against the device's occupancy AT LAUNCH
// and refuses it (hipErrorCooperativeLaunchTooLarge) instead of letting resident workgroups spin for
// ones that were never scheduled.  Costs ~15-20 us of host time per launch (MI355X_MICROARCH.md,
// coop-launch); XDTTS_COOP=0 switches back to plain launches (residency is identical, only the check
// is lost).  Every spin stays bounded either way: a grid kept off the chip by ANOTHER process is
// caught by the timeout path, not by this check.
// `coop_default` = what the call site does when XDTTS_COOP is unset: the decoder and the encoder BiLSTM
// launch cooperatively (their launches last milliseconds); the persistent Griffin-Lim does not -- the
// cooperative launch measured +17 us on a 200-350 us vocoder call (5 %), its grid is checked against the
// occupancy query when the handle is created, and a grid that still is not resident is caught by the
// bounded spins like everywhere else.  XDTTS_COOP=1 / 0 forces either form for all three.
// A cooperative launch is what VALIDATES a grid (kernel, block size, LDS, grid size <= what the runtime's occupancy says the
// device can hold at once); its residency is the same as a plain launch's.  So the check is paid once: the first launch of
// a given (kernel, block, LDS) on a device goes through hipLaunchCooperativeKernel, and once a grid of N workgroups has been
// accepted every later launch of <= N workgroups of the same configuration is a plain one (-17 us of host time each: three
// such launches per utterance).  XDTTS_COOP=1 keeps every launch cooperative, XDTTS_COOP=0 makes every launch plain.
struct CoopValidated {
  const void *fn;
  unsigned threads, blocks;
  size_t lds;
  int device;
};
inline bool coop_validated(const void *fn, unsigned threads, size_t lds, unsigned blocks, bool record) {
  static std::vector<CoopValidated> seen;
  static std::mutex mu;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(mu);
  for (CoopValidated &v : seen)
    if (v.fn == fn && v.threads == threads && v.lds == lds && v.device == dev) {
      if (record && blocks > v.blocks) v.blocks = blocks;
      return blocks <= v.blocks;
    }
  if (record) seen.push_back(CoopValidated{fn, threads, blocks, lds, dev});
  return false;
}
template <class... Args>
inline hipError_t launch_coresident(bool coop_default, const void *fn, dim3 grid, dim3 block, size_t lds, hipStream_t s, Args... args) {
  void *argv[] = {(void *)&args...};
  static const int forced = [] {
    const char *e = getenv("XDTTS_COOP");
    return !e ? -1 : (e[0] == '0' ? 0 : 1);
  }();
  const unsigned blocks = grid.x * grid.y * grid.z, threads = block.x * block.y * block.z;
  bool coop = forced < 0 ? coop_default : forced == 1;
  if (coop && forced < 0 && coop_validated(fn, threads, lds, blocks, false)) coop = false;  // this grid has been accepted before
  if (!coop) return hipLaunchKernel(fn, grid, block, argv, lds, s);
  const hipError_t e = hipLaunchCooperativeKernel(fn, grid, block, argv, (unsigned)lds, s);
  if (e == hipSuccess) (void)coop_validated(fn, threads, lds, blocks, true);
  return e;
}

// A cooperative launch the runtime refused (hipErrorCooperativeLaunchTooLarge: CU masking, fewer CUs than the grid needs, a
// register budget that no longer admits one block per CU): not an error of the request -- the caller switches the handle
// to its launch-per-stage / single-workgroup engine and runs the request there.
struct CoopRefused {};
#define COOP_CHECK(expr)                                      \
  do {                                                        \
    const hipError_t ce_ = (expr);                            \
    if (ce_ == hipErrorCooperativeLaunchTooLarge) {           \
      (void)hipGetLastError(); /* clear the sticky error */   \
      throw ::xdtts::CoopRefused();                           \
    }                                                         \
    HIP_CHECK(ce_);                                           \
  } while (0)

// ---- counter-based RNG (specification shared with oracle/, implemented independently) ----
__host__ __device__ inline uint32_t mix32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}
__host__ __device__ inline uint32_t rng_u32(uint32_t seed, uint32_t stream, uint32_t idx) {
  return mix32(mix32(mix32(seed ^ 0x9E3779B9U) + stream) + idx);
}
__host__ __device__ inline float rng_uniform(uint32_t seed, uint32_t stream, uint32_t idx) {
  return (float)(rng_u32(seed, stream, idx) >> 8) * (1.0f / 16777216.0f);
}

// ---- device memory helper -------------------------------------------------------------------
template <class T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  void alloc(size_t count) {
    if (count <= n) return;
    release();
    HIP_CHECK(hipMalloc((void **)&p, count * sizeof(T)));
    n = count;
  }
  void upload(const T *src, size_t count, hipStream_t s) {
    alloc(count);
    HIP_CHECK(hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, s));
  }
};

}  // namespace xdtts
