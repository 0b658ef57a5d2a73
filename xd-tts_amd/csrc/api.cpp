// api.cpp -- the extern "C" boundary of libxdtts_hip.so (declared in include/xdtts.h) and the
// host-side orchestration of the HIP kernels.  No CPU compute path exists here: without a HIP
// device every entry point fails with XDTTS_ERR_NO_DEVICE.
#include <algorithm>
#include <cmath>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>

#include <fcntl.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cerrno>
#include <cstring>

#include "kernels.h"

namespace xdtts {

static thread_local std::string g_last_error;
void set_last_error(const char *msg) { g_last_error = msg ? msg : ""; }
void fail(xdtts_status code, const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  throw Error(code, buf);
}

template <class F>
static xdtts_status guard(F &&f) {
  try {
    f();
    return XDTTS_OK;
  } catch (const Error &e) {
    set_last_error(e.what());
    return e.code;
  } catch (const std::bad_alloc &) {
    set_last_error("host allocation failed");
    return XDTTS_ERR_OOM;
  } catch (const std::exception &e) {
    set_last_error(e.what());
    return XDTTS_ERR_HIP;
  } catch (const CoopRefused &) {  // (not a std::exception; the engines catch it where they can fall back)
    set_last_error("cooperative launch refused by the runtime: the grid does not fit this device");
    return XDTTS_ERR_HIP;
  } catch (...) {  // nothing unwinds through the C ABI
    set_last_error("unexpected exception");
    return XDTTS_ERR_HIP;
  }
}

// XDTTS_DEVICE_DEFAULT (-1) as a device_id = "the process's default GPU": the value of the environment variable XDTTS_DEVICE (read at
// every handle creation), 0 without it.  It is how a host that keeps the reference's constructor signatures -- Tacotron2::load(path),
// GriffinLim::new(..) take no device (src/lib.rs:40-58) -- is spread over the 8 GPUs of a node: one process per GPU, XDTTS_DEVICE = its
// rank; the shim's load_on / new_on pass an explicit id instead.
static int default_device() {
  const char *e = getenv("XDTTS_DEVICE");
  if (!e || !*e) return 0;
  char *end = nullptr;
  const long v = std::strtol(e, &end, 10);
  if (end == e || *end != 0 || v < 0 || v > 1023) fail(XDTTS_ERR_BAD_ARG, "XDTTS_DEVICE=\"%s\" is not a device index", e);
  return (int)v;
}
static int select_device(int device_id) {  // returns the device actually selected
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    fail(XDTTS_ERR_NO_DEVICE, "no HIP device visible: libxdtts_hip has no CPU fallback");
  if (device_id == XDTTS_DEVICE_DEFAULT) device_id = default_device();
  if (device_id < 0 || device_id >= n) fail(XDTTS_ERR_BAD_ARG, "device_id %d out of range (0..%d)", device_id, n - 1);
  HIP_CHECK(hipSetDevice(device_id));
  return device_id;
}

// Pinned host buffers handed to the caller.  hipHostMalloc/hipHostFree cost hundreds of
// microseconds (page pinning), comparable to a whole vocoder run, so released buffers are kept in
// a small size-classed pool and reused by later calls.
namespace {
struct PinnedPool {
  std::mutex mu;
  std::map<void *, size_t> live;                     // buffer -> capacity (bytes)
  std::multimap<size_t, void *> spare;               // capacity -> buffer
  size_t spare_bytes = 0;
  static constexpr size_t MAX_SPARE = 256u << 20;
  static size_t size_class(size_t bytes) {
    size_t c = 4096;
    while (c < bytes) c <<= 1;
    return c;
  }
  float *get(size_t bytes) {
    const size_t cap = size_class(bytes);
    {
      std::lock_guard<std::mutex> lk(mu);
      auto it = spare.find(cap);
      if (it != spare.end()) {
        void *p = it->second;
        spare.erase(it);
        spare_bytes -= cap;
        live[p] = cap;
        return (float *)p;
      }
    }
    void *p = nullptr;
    HIP_CHECK(hipHostMalloc(&p, cap, hipHostMallocDefault));
    std::lock_guard<std::mutex> lk(mu);
    live[p] = cap;
    return (float *)p;
  }
  // A slab handed out in pieces (the mels of a batch: one copy from the device, no repacking on the host): every piece
  // is released on its own (xdtts_free), the slab goes back to the pool with the last one.
  std::map<void *, void *> part_of;  // piece -> slab
  std::map<void *, int> pieces;      // slab -> pieces outstanding
  void add_pieces(void *slab, const std::vector<float *> &cut) {  // all or nothing, under one lock
    std::lock_guard<std::mutex> lk(mu);
    size_t done = 0;
    try {
      for (; done < cut.size(); ++done) part_of[cut[done]] = slab;
      pieces[slab] = (int)cut.size();
    } catch (...) {
      for (size_t i = 0; i < done; ++i) part_of.erase(cut[i]);
      throw;
    }
  }
  void put(void *p) {
    std::unique_lock<std::mutex> lk(mu);
    auto pt = part_of.find(p);
    if (pt != part_of.end()) {
      void *slab = pt->second;
      part_of.erase(pt);
      if (--pieces[slab] > 0) return;
      pieces.erase(slab);
      p = slab;
    }
    if (pieces.count(p)) return;  // a second xdtts_free of a slab's first piece while others are still out: not the slab's turn
    auto it = live.find(p);
    if (it == live.end()) return;  // not ours, or already released (a double xdtts_free): nothing to do --
                                   // freeing it here could hand a buffer in `spare` back to the runtime
    const size_t cap = it->second;
    live.erase(it);
    if (spare_bytes + cap <= MAX_SPARE) {
      spare.emplace(cap, p);
      spare_bytes += cap;
      return;
    }
    lk.unlock();
    (void)hipHostFree(p);
  }
};
PinnedPool &pinned_pool() {
  static PinnedPool *pool = new PinnedPool();  // intentionally leaked: outlives static destruction order
  return *pool;
}
}  // namespace

static float *pinned_alloc(size_t n_floats) { return pinned_pool().get(std::max<size_t>(n_floats, 1) * sizeof(float)); }

// Owns a pinned output buffer until the call has succeeded: an exception on the way (a failed
// copy, a later stage that throws) returns it to the pool instead of leaving it in `live` forever.
struct PinnedGuard {
  float *p = nullptr;
  PinnedGuard() = default;
  explicit PinnedGuard(size_t n_floats) : p(pinned_alloc(n_floats)) {}
  PinnedGuard(PinnedGuard &&o) noexcept : p(o.p) { o.p = nullptr; }
  PinnedGuard(const PinnedGuard &) = delete;
  PinnedGuard &operator=(const PinnedGuard &) = delete;
  PinnedGuard &operator=(PinnedGuard &&o) noexcept {
    if (this != &o) {
      if (p) pinned_pool().put(p);
      p = o.p;
      o.p = nullptr;
    }
    return *this;
  }
  ~PinnedGuard() {
    if (p) pinned_pool().put(p);
  }
  float *release() {
    float *r = p;
    p = nullptr;
    return r;
  }
};

// A pinned slab whose pieces go to the caller one by one (PinnedPool::add_piece).  Until hand_over() the slab is the
// guard's: an exception on the way returns it whole.
struct PinnedSlab {
  float *base = nullptr;
  std::vector<float *> cut;
  explicit PinnedSlab(size_t n_floats) : base(pinned_alloc(n_floats)) {}
  PinnedSlab(const PinnedSlab &) = delete;
  PinnedSlab &operator=(const PinnedSlab &) = delete;
  ~PinnedSlab() {
    if (base) pinned_pool().put(base);
  }
  float *piece(size_t offset_floats) {  // (distinct offsets: a piece is identified by its address)
    cut.push_back(base + offset_floats);
    return cut.back();
  }
  void hand_over() {  // from here on every piece is the caller's; the slab follows the last one
    if (cut.empty()) return;
    pinned_pool().add_pieces(base, cut);
    base = nullptr;
  }
};

struct Events {
  hipEvent_t e[4] = {nullptr, nullptr, nullptr, nullptr};
  void create() {
    for (auto &x : e) HIP_CHECK(hipEventCreate(&x));
  }
  ~Events() {
    for (auto &x : e)
      if (x) (void)hipEventDestroy(x);
  }
};

}  // namespace xdtts

#include "edge_floor.hip"

using namespace xdtts;

// The cooperative encoder BiLSTM and the persistent decoder need their whole grid co-resident, so
// two of them from different handles must never be in flight together (each could hold CUs the
// other waits for).  Every call that launches one holds this lock from enqueue to completion.
// Between PROCESSES that share a GPU the same rule holds; XDTTS_CHIP_LOCK_DIR=<dir> (read once) adds an flock on
// <dir>/xdtts_chip_<pci bus id>.lock to the lock, so that co-resident launches of different processes take turns instead of
// timing out into the fallback engines (bench.py's two-ranks-on-one-GPU test mode uses it; so can a multi-worker server).
class ChipLock {
  std::recursive_mutex m;
  int depth = 0, fd = -2;  // fd -2: not looked at yet (in this process), -1: no file lock
  const int device;
  pid_t owner = 0;         // the process that opened fd: a forked child inherits the open file DESCRIPTION, on which parent and
                           // child would both "hold" the flock -- it opens its own
  void open_file() {
    if (fd >= 0 && owner != getpid()) ::close(fd);   // (the inherited descriptor; the parent's stays open in the parent)
    fd = -1;
    owner = getpid();
    const char *dir = getenv("XDTTS_CHIP_LOCK_DIR");
    if (!dir || !*dir) return;
    char bus[64] = "unknown";
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) std::snprintf(bus, sizeof bus, "dev%d", device);
    for (char *c = bus; *c; ++c)
      if (*c == ':' || *c == '/') *c = '_';
    const std::string path = std::string(dir) + "/xdtts_chip_" + bus + ".lock";
    fd = ::open(path.c_str(), O_CREAT | O_RDWR | O_CLOEXEC, 0666);
    if (fd >= 0) (void)::fchmod(fd, 0666);  // (the creator's umask must not lock a second user out; fails harmlessly for a non-owner)
    if (fd < 0) fd = ::open(path.c_str(), O_RDONLY | O_CLOEXEC);  // another user's 0644 file, or fs.protected_regular in a sticky directory: flock needs no write access
    if (fd < 0) {
      // the caller asked for cross-process serialisation and cannot have it: an error, not a warning (without the lock two
      // processes time each other's co-resident launches out into the fallback engines)
      fd = -2;
      fail(XDTTS_ERR_IO, "XDTTS_CHIP_LOCK_DIR: cannot open %s (%s)", path.c_str(), std::strerror(errno));
    }
  }

 public:
  explicit ChipLock(int d) : device(d) {}
  void lock() {
    m.lock();
    if (depth == 0) {
      try {
        if (fd == -2 || owner != getpid()) open_file();
      } catch (...) {
        m.unlock();
        throw;
      }
      if (fd >= 0)
        while (::flock(fd, LOCK_EX) != 0 && errno == EINTR) {
        }
    }
    ++depth;
  }
  void unlock() {
    if (--depth == 0 && fd >= 0) (void)::flock(fd, LOCK_UN);
    m.unlock();
  }
};
static ChipLock &chip_mutex(int device) {
  static std::mutex g;
  static std::map<int, std::unique_ptr<ChipLock>> locks;  // one per GPU of this process, keyed by the device id itself
  std::lock_guard<std::mutex> l(g);
  std::unique_ptr<ChipLock> &p = locks[device];
  if (!p) p.reset(new ChipLock(device));
  return *p;
}

// ================================================================================================
// Tacotron2 handle
// ================================================================================================
struct xdtts_tacotron2 {
  int device = 0;
  hipStream_t stream = nullptr;
  mutable std::mutex mu;
  std::vector<float> blob;  // canonical weights (host), for save/get_tensor
  DeviceWeights w;
  Events ev;
  float last_ms[4] = {0, 0, 0, 0};
  int last_steps = 0;

  // workspaces (grown on demand)
  // an input array on the device: either its own allocation (upload) or a view into the request's one staged block (infer_batch_device)
  template <class T>
  struct DevSlot {
    T *p = nullptr;
    DevBuf<T> own;
    void upload(const T *src, size_t count, hipStream_t s) {
      own.upload(src, count, s);
      p = own.p;
    }
  };
  DevSlot<int64_t> ids;
  DevSlot<int> n_valid, limits;
  DevBuf<unsigned char> in_blk;       // ids | lens | step caps | dropout-stream order of one request: ONE host-to-device copy
  unsigned char *in_host = nullptr;   // pinned staging of the same
  size_t in_host_bytes = 0;
  std::vector<int> lim_on_dev;        // the step caps limits.p holds (run_decoder uploads them only when they differ)
  // One control block on the device, mirrored by host_ctl: [0..1] ctl (step counter, spare), [2] encoder error word, [3] decoder
  // error word, [4 ..] frames per chunk -- so that what a decode hands back to the host is ONE copy (it was three of 4-5 us each)
  struct IntRef {
    int *p = nullptr;
  };
  DevBuf<int> ctlblk;
  IntRef ctl, enc_err, dec_err, nframes;
  DevBuf<float> xpadA, xpadB, xproj, memory, pmem;
  const float *xpad_zero[2] = {nullptr, nullptr};  // the allocations and layout whose padding rows are known to be zero
  int xpad_B = 0, xpad_T = 0;
  DevBuf<unsigned long long> enc_exchange;
  DevBuf<unsigned long long> dec_exchange;  // granule buffers of the persistent decoder
  DevBuf<float> ctx_fold;                   // [B][CTXF_ROWS][CTXF_LD] context-fold table of the persistent decoder (kernels.h)
  DevBuf<unsigned long long> att_exchange;
  DevBuf<float> att_part;  // early partial pre-activations of the attention LSTM (DecoderBufs::att_part)
  DevBuf<float> dec_part;  // two-launch form: early partial of the decoder LSTM's h_dec columns (DecoderBufs::dec_part)
  DevBuf<unsigned> h_ring;  // two-launch form: h_att per step as a write-once ring (DecoderBufs::hring)
  DevBuf<unsigned> h_stage;  // ... its per-XCD copies and their counters (DecoderBufs::hstage, hcnt)
  DevBuf<unsigned long long> tail_exchange;  // two-launch form: h_dec and mel granules (DecoderBufs::hdg, melg)
  // batched mode: energies, softmax and context in one launch (XDTTS_ATT_FUSED=0: the two-kernel form; also after
  // an exchange of that launch timed out)
  static int att_fused_default() {
    const char *e = getenv("XDTTS_ATT_FUSED");
    return e ? atoi(e) : 2;
  }
  int n_cu = 0;
  bool att_demoted = false;
  int att_demoted_calls = 0;
  int att_fused = att_fused_default();  // 2: with the attention LSTM in the same launch, 1: attention alone, 0: two kernels
  // XDTTS_NO_EARLY (read when a handle is created): the attention launch multiplies its whole K instead of adding the early
  // partial of the previous decoder-LSTM launch (second form of the same arithmetic for the agreement test; results agree to 1e-5)
  bool early_partial = getenv("XDTTS_NO_EARLY") == nullptr;
  // XDTTS_NO_CTXFOLD (read when a handle is created): the persistent kernel folds the context columns into the encoder memory
  // itself, in every launch, instead of reading the table one GEMM per request makes (tests compare the two forms)
  bool ctx_fold_table = getenv("XDTTS_NO_CTXFOLD") == nullptr;
  // XDTTS_NO_SKEW (read when a handle is created): pairs of chunks run the persistent kernel's lock-step loop instead of the skewed one
  bool pair_skew = getenv("XDTTS_NO_SKEW") == nullptr;
  // XDTTS_P8=0 (read when a handle is created): 3..8 chunks go to the engines that served them before decoder_persistent8.hip
  bool p8_wanted = [] { const char *p = getenv("XDTTS_P8"); return !(p && p[0] == '0'); }();
  bool two_launch = getenv("XDTTS_NO_TAIL") == nullptr;  // (XDTTS_NO_TAIL: keep the prenet launch; read when a handle is created)
  int persist_state = -1;                   // -1 unknown, 0 unavailable on this device / demoted, 1 usable
  bool persist_probe_ok = false;            // the device can host the persistent grid (occupancy probe)
  bool coop_ok = true;                      // cooperative encoder BiLSTM usable (cleared after a timed-out exchange)
  bool coop_refused = false;                // the runtime refused its cooperative launch: never probed again
  bool persist_refused = false;             // the same for the persistent decoder's grid (engine_reset does not undo it)
  int enc_demoted_calls = 0;                // encoder calls since its demotion (own re-probe counter)
  int coop_group = 16;                      // chunks per cooperative BiLSTM launch: 8 workgroups of 1024 threads per
                                            // chunk must be co-resident, one per CU (set from the CU count in init)
  DevBuf<float> att_h, att_c, dec_h, dec_c, aw, awc, ctx, x, loc, e_part, pmel, frames, gates;
  DevBuf<float> frag;       // batched mode: MFMA-operand copies of x, ctx, att_h[2], dec_h[2]
  DevBuf<float> pmem_t;     // batched mode: processed_memory as [B][32][T][4]
  DevSlot<int> item_perm;    // batched mode: dropout-stream index of the (length-sorted) chunks
  DevBuf<float> dec_in_dev; // parity hook: decoder_input of xdtts_tacotron2_decoder_step
  DevBuf<unsigned char> drop_dev;  // dropout_mode 2: the caller's keep masks
  DevBuf<float> state_stage;       // parity hook: the seven state tensors in the caller's layout
  int demoted_calls = 0;    // decoder calls since a demotion (the fast engines are probed again after PROBE_AFTER)
  static constexpr int PROBE_AFTER = 64;
  DevBuf<float> pp0, ppA, ppB, mel_dev;
  std::vector<long> pp_sig;  // layout (items, frames, allocations) whose padding is known to be zero in pp0 / ppA / ppB
  std::function<void()> before_decoder;  // enqueued between the encoder and the frame loop of infer_batch_device (or empty)
  std::function<void()> while_decoding;  // host work for the time the frame loop runs: called once everything of the decode is enqueued, before the host waits for it (or empty)
  int *host_ctl = nullptr;  // pinned mirror of ctlblk: [0..1] ctl, [HOST_ENC_ERR] / [HOST_DEC_ERR] the engines' error words, [HOST_NF ..] nframes
  static constexpr int HOST_ENC_ERR = 2, HOST_DEC_ERR = 3, HOST_NF = 4, CTL_INTS = HOST_NF + 4096;

  // cached hipGraph of GRAPH_STEPS decoder steps for the current (B, T, buffers)
  static constexpr int GRAPH_STEPS = 20;
  hipGraphExec_t graph = nullptr;
  DecoderBufs graph_key{};

  hipEvent_t fetched = nullptr;  // behind the copies that bring error word and frame counts back (run_decoder)
  ~xdtts_tacotron2() {
    if (fetched) (void)hipEventDestroy(fetched);
    if (graph) (void)hipGraphExecDestroy(graph);
    if (host_ctl) (void)hipHostFree(host_ctl);
    if (in_host) (void)hipHostFree(in_host);
    if (stream) (void)hipStreamDestroy(stream);
  }

  void init(int dev) {
    device = dev;
    select_device(dev);
    HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    ev.create();
    HIP_CHECK(hipEventCreateWithFlags(&fetched, hipEventDisableTiming));
    HIP_CHECK(hipHostMalloc((void **)&host_ctl, sizeof(int) * CTL_INTS, hipHostMallocDefault));
    int cus = 0;
    HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    coop_group = cus / 8 < 1 ? 1 : cus / 8;
    n_cu = cus;
    ctlblk.alloc(CTL_INTS);
    HIP_CHECK(hipMemsetAsync(ctlblk.p, 0, sizeof(int) * CTL_INTS, stream));
    ctl.p = ctlblk.p;
    enc_err.p = ctlblk.p + HOST_ENC_ERR;
    dec_err.p = ctlblk.p + HOST_DEC_ERR;
    nframes.p = ctlblk.p + HOST_NF;
    w.upload(blob, stream);
  }

  // ---- encoder.onnx (mod.rs:379): ids [B][T] on device -> memory, pmem -------------------------
  // every dense contraction of the handle goes through here: split-K where it pays (gemm.hip: gemm_splitk_plan), the slices'
  // meeting place and the tiles' arrival counters owned by the handle (one stream: launches never overlap)
  DevBuf<float> gemm_ws;
  DevBuf<unsigned> gemm_cnt;
  void run_gemm(GemmArgs &g) {
    size_t wsf = 0, tiles = 0;
    int tile = 0;
    const int sk = gemm_splitk_plan(g, &wsf, &tiles, &tile);
    if (sk > 1) {
      if (tiles > gemm_cnt.n) {
        gemm_cnt.alloc(std::max<size_t>(tiles, 4096));
        HIP_CHECK(hipMemsetAsync(gemm_cnt.p, 0, gemm_cnt.n * sizeof(unsigned), stream));
      }
      gemm_ws.alloc(wsf);
      g.splitk = sk;
      g.tile = tile;
      g.ws = gemm_ws.p;
      g.cnt = gemm_cnt.p;
    }
    launch_gemm_nt(g, stream);
  }

  void run_encoder(int B, int T) {
    const int pad = (ENC_K - 1) / 2, TP = T + 2 * pad;
    const size_t padded = (size_t)B * TP * EMB;
    xpadA.alloc(padded);
    xpadB.alloc(padded);
    xproj.alloc((size_t)2 * B * T * 4 * ENC_H);
    memory.alloc((size_t)B * T * EMB);
    pmem.alloc((size_t)B * T * ATT_DIM);
    // only the padding rows must read as zero, and nothing ever writes them (the embedding and the convolutions store rows
    // pad .. pad + T - 1 of every chunk): the fills are needed when the layout (or the allocation) changes, not per request
    if (xpad_zero[0] != xpadA.p || xpad_zero[1] != xpadB.p || xpad_B != B || xpad_T != T) {
      HIP_CHECK(hipMemsetAsync(xpadA.p, 0, padded * sizeof(float), stream));
      HIP_CHECK(hipMemsetAsync(xpadB.p, 0, padded * sizeof(float), stream));
      xpad_zero[0] = xpadA.p;
      xpad_zero[1] = xpadB.p;
      xpad_B = B;
      xpad_T = T;
    }
    launch_embed(ids.p, w.emb.p, xpadA.p, B, T, pad, stream);
    float *src = xpadA.p, *dst = xpadB.p;
    for (int i = 0; i < ENC_CONVS; ++i) {
      GemmArgs g{};
      g.A = src;
      g.lda = EMB;
      g.strideA = (long)TP * EMB;
      g.W = w.enc_conv[i].w.p;
      g.bias = w.enc_conv[i].b.p;
      g.C = dst + (size_t)pad * EMB;
      g.ldc = EMB;
      g.strideC = (long)TP * EMB;
      g.M = T;
      g.N = EMB;
      g.K = ENC_K * EMB;
      g.batch = B;
      g.act = 1;
      run_gemm(g);
      std::swap(src, dst);
    }
    for (int d = 0; d < 2; ++d) {  // BiLSTM input projections for all T at once
      GemmArgs g{};
      g.A = src + (size_t)pad * EMB;
      g.lda = EMB;
      g.strideA = (long)TP * EMB;
      g.W = w.enc_wih[d].p;
      g.bias = w.enc_bias[d].p;
      g.C = xproj.p + (size_t)d * B * T * 4 * ENC_H;
      g.ldc = 4 * ENC_H;
      g.strideC = (long)T * 4 * ENC_H;
      g.M = T;
      g.N = 4 * ENC_H;
      g.K = EMB;
      g.batch = B;
      run_gemm(g);
    }
    // a demoted encoder probes the cooperative recurrence again by itself (its own counter: the decoder's re-probe does not
    // depend on it, and a batched decode never passes through use_persistent)
    if (!coop_ok && !coop_refused && ++enc_demoted_calls >= PROBE_AFTER) {
      enc_demoted_calls = 0;
      coop_ok = true;
    }
    bool coop_ran = false;
    if (coop_ok) {
      // groups of four workgroups per direction, at most coop_group of them per launch; from coop_group + 1 chunks on a group
      // takes two chunks (52 chunks on 256 CUs: one launch of 26 two-chunk groups; it was 26 + 26 one-chunk groups)
      const int slots = B > coop_group ? (B + 1) / 2 : B;  // groups needed
      const int launches = (slots + coop_group - 1) / coop_group, group = B > coop_group ? coop_group : (B + launches - 1) / launches;
      enc_exchange.alloc(bilstm_coop_exchange_words(2 * group));
      try {
        launch_bilstm_coop(xproj.p, w.enc_whhT[0].p, w.enc_whhT[1].p, memory.p, enc_exchange.p, enc_err.p, B, T, group,
                           stream);
        coop_ran = true;
      } catch (const CoopRefused &) {
        // the runtime refused the cooperative grid (CU masking, a smaller part): the single-workgroup recurrence serves
        // this handle from now on -- the refusal is a property of the device, not a transient
        coop_ok = false;
        coop_refused = true;
        std::fprintf(stderr, "libxdtts_hip: cooperative encoder BiLSTM launch refused by the runtime; this handle uses the "
                             "single-workgroup recurrence\n");
      }
    }
    if (!coop_ran) launch_bilstm(xproj.p, w.enc_whhT[0].p, w.enc_whhT[1].p, memory.p, B, T, stream);
    GemmArgs g{};  // processed_memory = memory_layer(memory)
    g.A = memory.p;
    g.lda = EMB;
    g.strideA = (long)T * EMB;
    g.W = w.mem_w.p;
    g.C = pmem.p;
    g.ldc = ATT_DIM;
    g.strideC = (long)T * ATT_DIM;
    g.M = T;
    g.N = ATT_DIM;
    g.K = EMB;
    g.batch = B;
    run_gemm(g);
  }

  // dropout_mode 2 (SURVEY 8(b) "explicit(mask ptr)"): the caller's keep bytes [B][steps][2][256] go to the device; every
  // chunk's step limit must lie inside them.  Any other mode value than 0 / 1 / 2 is refused here too.
  void upload_dropout_masks(const xdtts_infer_opts &o, int B, const int *lim) {
    if (o.dropout_mode < 0 || o.dropout_mode > 2) fail(XDTTS_ERR_BAD_ARG, "dropout_mode %d out of range (0 off, 1 seeded, 2 explicit)", o.dropout_mode);
    if (o.dropout_mode != 2) return;
    if (!o.dropout_masks || o.dropout_mask_steps <= 0) fail(XDTTS_ERR_BAD_ARG, "dropout_mode 2 needs dropout_masks and dropout_mask_steps");
    for (int b = 0; b < B; ++b)
      if (lim[b] > o.dropout_mask_steps)
        fail(XDTTS_ERR_BAD_ARG, "chunk %d may run %d steps, the dropout masks cover %d", b, lim[b], o.dropout_mask_steps);
    drop_dev.upload(o.dropout_masks, (size_t)B * o.dropout_mask_steps * 2 * PRENET, stream);
  }

  // force_batched: -1 = by batch size (the MFMA kernels from BATCH_MFMA_MIN chunks), 0 / 1 = the parity hook's choice
  DecoderBufs decoder_bufs(int B, int T, const float *mem, const float *pm, const xdtts_infer_opts &o, int force_batched = -1) {
    const bool batched = force_batched < 0 ? B >= BATCH_MFMA_MIN : force_batched != 0;
    const int ms = o.max_steps;
    att_h.alloc((size_t)2 * B * ATT_RNN);
    att_c.alloc((size_t)((B + 15) / 16 * 16) * ATT_RNN);  // (batched mode: [256][Bpad][4])
    dec_h.alloc((size_t)2 * B * DEC_RNN);
    dec_c.alloc((size_t)((B + 15) / 16 * 16) * DEC_RNN);
    aw.alloc((size_t)B * T);
    awc.alloc((size_t)B * T);
    ctx.alloc((size_t)B * EMB);
    x.alloc((size_t)B * PRENET);
    loc.alloc((size_t)B * T * ATT_DIM);
    e_part.alloc((size_t)B * (ATT_DIM / 4) * T);
    pmel.alloc(decoder_pmel_floats(B));
    frames.alloc((size_t)B * ms * N_MEL);
    gates.alloc((size_t)B * ms);
    DecoderBufs d{};
    d.B = B;
    d.T = T;
    d.memory = mem;
    d.pmem = pm;
    d.n_valid = n_valid.p;
    d.att_h[0] = att_h.p;
    d.att_h[1] = att_h.p + (size_t)B * ATT_RNN;
    d.att_c = att_c.p;
    d.dec_h[0] = dec_h.p;
    d.dec_h[1] = dec_h.p + (size_t)B * DEC_RNN;
    d.dec_c = dec_c.p;
    d.aw = aw.p;
    d.awc = awc.p;
    d.ctx = ctx.p;
    d.x = x.p;
    d.loc = loc.p;
    d.e_part = e_part.p;
    d.pmel = pmel.p;
    d.frames = frames.p;
    d.gates = gates.p;
    d.nframes = nframes.p;
    d.ctl = ctl.p;
    d.max_steps = ms;
    d.use_gate = o.fixed_steps > 0 ? 0 : 1;
    d.gate_threshold = o.gate_threshold;
    {  // gate_fires (device_utils.h): where the verdict needs no sigmoid
      const double t = (double)o.gate_threshold;
      d.gate_lo = -INFINITY;  // (an empty band on either side = always the reference's arithmetic)
      d.gate_hi = INFINITY;
      if (t > 1e-3 && t < 1.0 - 1e-3) {
        const double L = std::log(t / (1.0 - t)), w = 1e-3 * (1.0 + std::fabs(L));
        d.gate_lo = (float)(L - w);
        d.gate_hi = (float)(L + w);
      }
    }
    d.dropout_mode = o.dropout_mode;
    d.dropout_seed = o.dropout_seed;
    d.item_base = o.item_base;
    if (o.dropout_mode == 2) {  // the caller's keep masks, uploaded by upload_dropout_masks()
      d.drop_masks = drop_dev.p;
      d.drop_steps = o.dropout_mask_steps;
    }
    if (batched) {  // MFMA B-operand copies [K/4][Bpad][4] of the vectors the LSTM GEMMs consume
      const int Bpad = (B + 15) / 16 * 16;
      frag.alloc((size_t)Bpad * (PRENET + EMB + 2 * ATT_RNN + 2 * DEC_RNN) + (size_t)B * T);
      d.Bpad = Bpad;
      d.xf = frag.p;
      d.ctxf = d.xf + (size_t)Bpad * PRENET;
      d.att_hf[0] = d.ctxf + (size_t)Bpad * EMB;
      d.att_hf[1] = d.att_hf[0] + (size_t)Bpad * ATT_RNN;
      d.dec_hf[0] = d.att_hf[1] + (size_t)Bpad * ATT_RNN;
      d.dec_hf[1] = d.dec_hf[0] + (size_t)Bpad * DEC_RNN;
      d.awc2 = d.dec_hf[1] + (size_t)Bpad * DEC_RNN;
      pmem_t.alloc((size_t)B * T * ATT_DIM);
      launch_dimgroup_transpose(pm, pmem_t.p, B, T, stream);
      d.pmem_t = pmem_t.p;
      // a timed-out exchange demoted the handle to separate kernels; the cause may be transient: try again after PROBE_AFTER batches
      if (att_demoted && ++att_demoted_calls >= PROBE_AFTER) {
        att_demoted = false;
        att_fused = att_fused_default();
      }
      if (att_fused > 0 && T <= T_MAX) {  // one-launch attention: partial energies cross as tagged granules
        const size_t ne = (size_t)B * ATT_EXCHANGE_BLOCKS * T;
        att_exchange.alloc(ne + (size_t)B * ATT_RNN);
        d.ep_g = att_exchange.p;
        d.att_err = dec_err.p;
        // ... and the attention LSTM in the same launch: its 256 blocks of 512 threads must be resident together, one per CU
        if (att_fused > 1 && B <= 64 && n_cu >= ATT_RNN / 4) {
          d.hg = att_exchange.p + ne;
          if (early_partial) {  // early partial of the attention-LSTM GEMM, computed by extra blocks of the decoder-LSTM launch (kernels.h)
            att_part.alloc((size_t)(ATT_RNN / 4) * 4 * 64 * 4);
            d.att_part = att_part.p;
            if (two_launch && T <= PERSIST_T_MAX) {  // ... and the prenet as the tail of that launch (h_dec / mel cross as granules)
              tail_exchange.alloc((size_t)B * (DEC_RNN + 96));
              d.hdg = tail_exchange.p;
              d.melg = d.hdg + (size_t)B * DEC_RNN;
              static const bool no_dh = getenv("XDTTS_NO_DHEARLY") != nullptr;  // developer comparison aid
              // (built, parity-green, measured, NOT the default -- profiles/r06_config3_hfirst_rejected.txt: the decoder-LSTM launch loses 1.5 us
              // per configs[2] iteration, the attention launch gains 4.5: eight more k-steps per wave ahead of the attention cell are eight more
              // operand round trips at the one or two k-steps of look-ahead a 128-register wave has, not work hidden in the wait for x)
              static const bool want_hf = getenv("XDTTS_HFIRST") != nullptr && getenv("XDTTS_HFIRST")[0] == '1';
              d.att_hfirst = want_hf ? 1 : 0;
              if (!no_dh) {
                dec_part.alloc((size_t)(DEC_RNN / 4) * 4 * 64 * 4);
                d.dec_part = dec_part.p;
                // ... and h_att(s) as a write-once ring too, so that the decoder LSTM's h_att columns are multiplied inside the attention
                // launch (kernels.h: DecoderBufs::hring): one slab of 4 kB per chunk slot and step -- 262 MB at 64 slots x 1000 steps;
                // a request capped beyond 1 GiB of ring (or with no room for it) keeps the 1536-column pass
                // Built, parity-green, measured and NOT the default (profiles/r06_config3_hring_rejected.txt): the decoder-LSTM launch shrinks by
                // 3.3 us per iteration at configs[2] and the attention launch grows by 3.6-4.4 -- its extra blocks read 48-64 MB of fresh
                // cross-XCD data per step inside the launch at the ~6.5 TB/s every in-launch bulk edge on this chip has shown (round 3),
                // 8-10 us for what a grid boundary delivers in 2.  XDTTS_HRING=1 enables it.
                static const bool want_hr = getenv("XDTTS_HRING") != nullptr && getenv("XDTTS_HRING")[0] == '1';
                const size_t ring_words = (size_t)ms * ATT_RNN * Bpad;
                static const bool want_relay = want_hr && getenv("XDTTS_HRING")[1] == 'r';  // XDTTS_HRING=1r: with the per-XCD relay
                if (want_hr && ring_words * sizeof(unsigned) <= ((size_t)1 << 30)) try {
                  h_ring.alloc(ring_words);
                  d.hring = h_ring.p;
                  d.hring_steps = ms;
                  if (want_relay) {
                    h_stage.alloc((size_t)16 * ATT_RNN * Bpad + (size_t)8 * ms * 64);
                    d.hstage = h_stage.p;
                    d.hcnt = h_stage.p + (size_t)16 * ATT_RNN * Bpad;
                  }
                } catch (const Error &e) {
                  if (e.code != XDTTS_ERR_OOM) throw;
                  (void)hipGetLastError();
                }
              }
            }
          }
        }
        if (const char *sp = getenv("XDTTS_ATT_SPINS")) d.att_spins = atoi(sp);  // test hooks for the
        if (const char *ft = getenv("XDTTS_ATT_FAULT")) d.att_fault = atoi(ft);  // lost-block path
        if (const char *ft = getenv("XDTTS_TAIL_FAULT")) d.tail_fault = atoi(ft);  // (two-launch form: a block whose h_dec never arrives)
        if (const char *sl = getenv("XDTTS_ATT_SLOW")) d.att_slow = atoi(sl);      // straggler block
      }
    }
    return d;
  }

  // Replays (building on first use) a hipGraph holding GRAPH_STEPS decoder steps.  The kernels
  // read the step index from device memory, so one graph serves every position of the loop.
  void replay_steps(const DecoderBufs &d) {
    static const bool no_graph = getenv("XDTTS_NO_GRAPH") != nullptr;  // developer comparison aid
    if (no_graph) {
      launch_decoder_steps(d, w, GRAPH_STEPS, stream);
      return;
    }
    if (!graph || std::memcmp(&graph_key, &d, sizeof d) != 0) {
      if (graph) {
        (void)hipGraphExecDestroy(graph);
        graph = nullptr;
      }
      hipGraph_t g = nullptr;
      HIP_CHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
      try {
        launch_decoder_steps(d, w, GRAPH_STEPS, stream);
      } catch (...) {
        (void)hipStreamEndCapture(stream, &g);
        if (g) (void)hipGraphDestroy(g);
        throw;
      }
      HIP_CHECK(hipStreamEndCapture(stream, &g));
      hipError_t e = hipGraphInstantiate(&graph, g, nullptr, nullptr, 0);
      (void)hipGraphDestroy(g);
      HIP_CHECK(e);
      graph_key = d;
    }
    HIP_CHECK(hipGraphLaunch(graph, stream));
  }

  // run_decoder frame loop (mod.rs:302-342) for B chunks; limits = per-chunk step caps (host).
  // Returns the number of lock-step iterations executed; host_ctl[2+b] = frames of chunk b.
  // Small lock-step batches run the persistent weight-stationary kernel (decoder_persistent.hip)
  // when its 256-workgroup grid can be co-resident; XDTTS_DECODER=launch forces the
  // launch-per-stage path (developer comparison aid).
  // 3..16 chunks: the persistent MFMA engine (decoder_persistent8.hip: 4 / 8 chunk slots, decoder_persistent16.hip: 16), one launch
  // for the whole loop.  It shares the persistent
  // engine's fate: a timed-out exchange or a refused launch demotes both (persist_state), the request runs again on the
  // launch-per-stage engine.
  int p8_state = -1;  // -1 unknown, 0 off (XDTTS_P8=0, the device cannot host the grid, or an exchange timed out), 1 usable
  bool p8_probe_ok = false;  // wanted, and the device can host the grid (occupancy probe): a demotion may be transient
  int p8_demoted_calls = 0;  // small-batch requests since a timed-out exchange demoted the engine (own re-probe counter)
  bool p8_refused = false;  // the runtime refused the cooperative launch: a property of the device (engine_reset does not undo it)
  bool small_batch_engine(int B, int T, int max_steps) {
    if (B < 3 || B > P8_B_MAX || T > PERSIST_T_MAX || max_steps > P8_STEPS_MAX) return false;
    const char *e = getenv("XDTTS_DECODER");
    if (e && std::string(e) == "launch") return false;
    if (p8_state < 0) {
      p8_probe_ok = p8_wanted && decoder_p8_supported(device, 8, PERSIST_T_MAX) && decoder_p8_supported(device, P8_B_MAX, PERSIST_T_MAX);  // (the 8- and the 16-slot kernel)
      p8_state = p8_probe_ok ? 1 : 0;
    }
    // like the persistent engine: the cause of a timed-out exchange (another process holding CUs) may be transient
    if (p8_state == 0 && p8_probe_ok && !p8_refused && ++p8_demoted_calls >= PROBE_AFTER) {
      p8_demoted_calls = 0;
      p8_state = 1;
      if (persist_state == 0 && persist_probe_ok) persist_state = 1, demoted_calls = 0;  // (they were demoted together)
    }
    return p8_state == 1 && persist_state != 0;
  }
  bool use_persistent(const DecoderBufs &d) {
    if (d.xf || d.B > 2 * PERSIST_B_MAX || d.T > PERSIST_T_MAX) return false;  // 3..4 chunks: two launches of <= 2
    const char *e = getenv("XDTTS_DECODER");
    if (e && std::string(e) == "launch") return false;
    if (persist_state < 0) {
      persist_state = decoder_persistent_supported(device, PERSIST_B_MAX, PERSIST_T_MAX) ? 1 : 0;
      persist_probe_ok = persist_state == 1;
    }
    // a timed-out exchange demotes the handle; the cause (another process holding CUs) may be transient, so
    // the persistent engine gets another try every PROBE_AFTER calls (xdtts_tacotron2_engine_reset: at once)
    if (persist_state == 0 && persist_probe_ok && ++demoted_calls >= PROBE_AFTER) {
      demoted_calls = 0;
      persist_state = 1;
    }
    return persist_state == 1;
  }

  // `after` (may be empty): work that only needs the frame COUNTS of a gate-less decode -- known beforehand: every chunk
  // runs to its cap -- enqueued behind the persistent launches and ahead of the sync that fetches error word and counts,
  // so the stream does not idle for that round trip (~90 us of the 7.7 ms headline utterance).  `*after_ran` tells the
  // caller whether what `after` enqueued stands: not when the engine faulted and the request was decoded again.
  int run_decoder(const DecoderBufs &d, const std::vector<int> &lim, const std::function<void()> &after = {},
                  bool *after_ran = nullptr) {
    if (after_ran) *after_ran = false;
    if (lim_on_dev != lim || !limits.p) {
      limits.upload(lim.data(), lim.size(), stream);
      lim_on_dev = lim;
    }
    launch_decoder_init(d, limits.p, stream);
    launch_decoder_prologue(d, w, stream);
    const int max_lim = *std::max_element(lim.begin(), lim.end());
    int launched = 0;
    const bool spec = after && !d.use_gate;
    bool spec_ran = false;
    // step counter and frame counts to the host; the last fetch of a decode brings the engines' error word along and,
    // for a gate-less decode, lets `after` enqueue its work before the host waits (for the copies only)
    auto fetch = [&](bool last = false) {
      (void)last;  // (the error words ride along every time: they sit between the step counter and the frame counts)
      HIP_CHECK(hipMemcpyAsync(host_ctl, ctlblk.p, sizeof(int) * (size_t)(HOST_NF + d.B), hipMemcpyDeviceToHost, stream));
      if (last && spec) {
        HIP_CHECK(hipEventRecord(fetched, stream));
        after();
        spec_ran = true;
        if (while_decoding) while_decoding();
        HIP_CHECK(hipEventSynchronize(fetched));
      } else {
        if (last && while_decoding) while_decoding();
        HIP_CHECK(hipStreamSynchronize(stream));
      }
    };
    auto finish = [&]() {  // frame counts are on the host: lock-step iterations; does what `after` enqueued stand?
      int steps = 0;
      bool as_planned = true;
      for (int b = 0; b < d.B; ++b) {
        steps = std::max(steps, host_ctl[HOST_NF + b]);
        as_planned = as_planned && host_ctl[HOST_NF + b] == lim[b];
      }
      if (after_ran) *after_ran = spec_ran && as_planned;
      return steps;
    };
    // (its exchange holds one slab per step, 11.3 kB per chunk slot: a request capped at more than 16384 steps -- 190 s of speech --
    // takes the other engines rather than gigabytes of ring)
    bool p8_go = !d.xf && small_batch_engine(d.B, d.T, max_lim);
    if (p8_go) try {
      dec_exchange.alloc(p8_exchange_words(d.B, max_lim));   // 11.3 kB per chunk slot and step: 90 MB at 8 slots x 1000 steps
    } catch (const Error &e) {
      if (e.code != XDTTS_ERR_OOM) throw;
      p8_go = false;      // no room for the ring: this request takes the other engines (whose exchange is 63 kB per chunk)
      (void)hipGetLastError();
    }
    if (p8_go) try {
      std::lock_guard<ChipLock> lk(chip_mutex(device));
      P8Bufs g8 = p8_bufs(dec_exchange.p, dec_err.p, d.B, max_lim);
      if (const char *sp = getenv("XDTTS_PERSIST_SPINS")) g8.spins = atoi(sp);  // test hooks for the lost-workgroup path
      if (const char *ft = getenv("XDTTS_PERSIST_FAULT")) g8.fault = atoi(ft);
      launch_p8_seed(d, g8, limits.p, stream);
      launch_decoder_p8(d, w, g8, max_lim, stream);
      fetch(true);
      if (!host_ctl[HOST_DEC_ERR]) return finish();
      spec_ran = false;
      HIP_CHECK(hipMemsetAsync(dec_err.p, 0, sizeof(int), stream));
      // the 256-workgroup grid was not co-resident: the pair-persistent engine, which needs the same, would spend a second
      // 2^21-spin time-out finding that out -- both are demoted, both are probed again after PROBE_AFTER requests.  The
      // request runs again on the launch-per-stage engine (row-major state, any B <= 8).
      p8_state = 0;
      p8_demoted_calls = 0;
      if (persist_state != 0) {
        if (persist_state < 0) persist_probe_ok = decoder_persistent_supported(device, PERSIST_B_MAX, PERSIST_T_MAX);
        persist_state = 0;
        demoted_calls = 0;
      }
      std::fprintf(stderr, "libxdtts_hip: persistent MFMA decoder exchange timed out (grid not co-resident); this handle now "
                           "uses the launch-per-stage decoder (probed again after %d calls)\n", PROBE_AFTER);
      launch_decoder_init(d, limits.p, stream);
    } catch (const CoopRefused &) {
      p8_state = 0;
      p8_refused = true;
      std::fprintf(stderr, "libxdtts_hip: persistent MFMA decoder launch refused by the runtime; this handle decodes small batches "
                           "with its other engines\n");
      HIP_CHECK(hipStreamSynchronize(stream));
      HIP_CHECK(hipMemsetAsync(dec_err.p, 0, sizeof(int), stream));
      launch_decoder_init(d, limits.p, stream);
    }
    if (use_persistent(d)) try {
      // one launch for the whole loop: the stop rule runs on the device and the kernel ends by itself.
      // Its grid must own the chip, so persistent launches of different handles never overlap.
      std::lock_guard<ChipLock> lk(chip_mutex(device));
      // Chunks are independent, so 3 or 4 of them run as two launches of <= 2 over views of the
      // state arrays (measured: 2 x 15.4 us per step-pair against 37 us per step of the launch path).
      // A 2-chunk launch ends when its first chunk stops and the other is continued by the 1-chunk
      // kernel (~1 us per step faster): the state crosses through the kernel's write-back, x(s)
      // stays in the exchange.
      // The context columns of every LSTM / projection row against the encoder memory, once per request (a GEMM of
      // 0.85 GFLOP per chunk) instead of a 33 us fold loop per chunk in each of the request's launches.
      const float *fold = nullptr;
      if (ctx_fold_table && w.ctx_w.p) {
        ctx_fold.alloc((size_t)d.B * CTXF_ROWS * CTXF_LD);
        GemmArgs fg{};
        fg.A = d.memory;
        fg.lda = EMB;
        fg.strideA = (long)d.T * EMB;
        fg.W = w.ctx_w.p;
        fg.C = ctx_fold.p;
        fg.ldc = CTXF_LD;
        fg.strideC = (long)CTXF_ROWS * CTXF_LD;
        fg.M = d.T;
        fg.N = CTXF_ROWS;
        fg.K = EMB;
        fg.batch = d.B;
        fg.transpose_out = 1;
        launch_gemm_nt(fg, stream);
        fold = ctx_fold.p;
      }
      auto view = [&](int b0, int n) {
        DecoderBufs v = d;
        v.B = n;
        v.ctx_fold = fold ? fold + (size_t)b0 * CTXF_ROWS * CTXF_LD : nullptr;
        v.memory += (size_t)b0 * d.T * EMB;
        v.pmem += (size_t)b0 * d.T * ATT_DIM;
        v.n_valid += b0;
        for (int i = 0; i < 2; ++i) {
          v.att_h[i] = d.att_h[0] + (size_t)b0 * ATT_RNN;  // the persistent kernel keeps h in slot 0 only
          v.dec_h[i] = d.dec_h[0] + (size_t)b0 * DEC_RNN;
        }
        v.att_c += (size_t)b0 * ATT_RNN;
        v.dec_c += (size_t)b0 * DEC_RNN;
        v.aw += (size_t)b0 * d.T;
        v.awc += (size_t)b0 * d.T;
        v.ctx += (size_t)b0 * EMB;
        v.frames += (size_t)b0 * d.max_steps * N_MEL;
        v.gates += (size_t)b0 * d.max_steps;
        v.nframes += b0;
        v.item_base += (uint32_t)b0;
        if (v.drop_masks) v.drop_masks += (size_t)b0 * d.drop_steps * 2 * PRENET;
        return v;
      };
      const char *no_shrink = getenv("XDTTS_NO_SHRINK");  // developer comparison aid
      for (int b0 = 0; b0 < d.B; b0 += PERSIST_B_MAX) {
        const int n = std::min(PERSIST_B_MAX, d.B - b0);
        const DecoderBufs v = view(b0, n);
        int sub_lim = 0;
        for (int b = 0; b < n; ++b) sub_lim = std::max(sub_lim, lim[b0 + b]);
        dec_exchange.alloc(persist_granule_words(n));
        PersistBufs g = persist_bufs(dec_exchange.p, dec_err.p, n);
        if (const char *lz = getenv("XDTTS_LAZY_POLL")) g.lazy = atoi(lz);  // developer tuning knobs
        if (const char *fp = getenv("XDTTS_FIRST_POLL")) g.first = atoi(fp);
        if (const char *pf = getenv("XDTTS_PFIRST")) g.pfirst = atoi(pf);
        if (const char *xf = getenv("XDTTS_XFIRST")) g.xfirst = atoi(xf);
        if (const char *ef = getenv("XDTTS_EFIRST")) g.efirst = atoi(ef);
        if (!pair_skew) g.skew = 0;
        if (const char *xl = getenv("XDTTS_XLAZY")) g.xlazy = atoi(xl);
        if (const char *cl = getenv("XDTTS_CLAZY")) g.clazy = atoi(cl);
        if (const char *sp = getenv("XDTTS_PERSIST_SPINS")) g.spins = atoi(sp);  // test hooks for the
        if (const char *ft = getenv("XDTTS_PERSIST_FAULT")) g.fault = atoi(ft);  // lost-workgroup path
        if (const char *sl = getenv("XDTTS_PERSIST_SLOW")) g.slow = atoi(sl);    // straggler workgroup
        g.shrink = (n == 2 && !no_shrink) ? 1 : 0;
        if (g.shrink && !d.use_gate && lim[b0] == lim[b0 + 1]) {
          // gate-less pair with equal caps: the host knows that neither chunk outlives the other, so no continuation launches
          // (each would do the full weight / LDS set-up and write-back for zero steps)
          g.shrink = 0;
          g.both_run = 1;
        }
#ifdef XDTTS_PERSIST_PROFILE
        static DevBuf<unsigned long long> prof;
        prof.alloc(256 * 24);
        g.prof = prof.p;
#endif
        if (b0 > 0) HIP_CHECK(hipMemsetAsync(d.ctl, 0, sizeof(int), stream));  // step counter of the new launch
        launch_persist_seed(v, g, limits.p + b0, stream);
        if (g.shrink && !d.use_gate && lim[b0] != lim[b0 + 1]) {
          // without the gate the host knows which chunk outlives the other: no round trip in between
          const int first = std::min(lim[b0], lim[b0 + 1]), r = lim[b0] > lim[b0 + 1] ? 0 : 1;
          g.shrink = 0;
          g.both_run = 1;  // (neither chunk stops inside these `first` steps: the skewed loop may run them)
          launch_decoder_persistent(v, w, g, first, stream);
          launch_decoder_persistent(view(b0 + r, 1), w, persist_view(g, r), lim[b0 + r] - first, stream);
        } else {
          launch_decoder_persistent(v, w, g, sub_lim, stream);
          if (g.shrink) {
            // Which chunk, if any, is still running is on the device (ctl[0] = steps executed, nframes[b] = its end): rather than
            // ask (a stream sync and two copies, ~0.1 ms of an 6.7 ms utterance) the continuation of BOTH chunks is enqueued --
            // the 1-chunk kernel returns at once for a chunk that has stopped (at most one survives the other; a survivor that
            // runs first leaves ctl[0] at its own end, which is past the other's), and stops by itself at the chunk's cap.
            for (int r = 0; r < n; ++r) {
              PersistBufs g1 = persist_view(g, r);
              g1.shrink = 0;
              launch_decoder_persistent(view(b0 + r, 1), w, g1, lim[b0 + r], stream);
            }
          }
        }
#ifdef XDTTS_PERSIST_PROFILE
        if (const char *path = getenv("XDTTS_PERSIST_PROFILE")) {
          std::vector<unsigned long long> hp(256 * 24);
          HIP_CHECK(hipMemcpyAsync(hp.data(), prof.p, hp.size() * 8, hipMemcpyDeviceToHost, stream));
          HIP_CHECK(hipStreamSynchronize(stream));
          if (FILE *f = fopen(path, "w")) {
            for (int c = 0; c < 256; ++c) {
              for (int i = 0; i < 24; ++i) fprintf(f, "%llu ", hp[c * 24 + i]);
              fprintf(f, "\n");
            }
            fclose(f);
          }
        }
#endif
      }
      fetch(true);
      if (!host_ctl[HOST_DEC_ERR]) return finish();
      spec_ran = false;  // (what `after` enqueued ran on a failed decode: it is enqueued again below)
      // A bounded spin ran out: the 256-workgroup grid was not co-resident (CUs masked or held by
      // another process).  Not silent, not fatal: say so, switch this handle to the launch-per-stage
      // engine for good, and decode this request again from the initial state.
      HIP_CHECK(hipMemsetAsync(dec_err.p, 0, sizeof(int), stream));
      persist_state = 0;
      demoted_calls = 0;
      std::fprintf(stderr, "libxdtts_hip: persistent decoder exchange timed out (grid not co-resident); "
                           "this handle now uses the launch-per-stage decoder (probed again after %d calls)\n", PROBE_AFTER);
      launch_decoder_init(d, limits.p, stream);
    } catch (const CoopRefused &) {
      // the runtime refused the cooperative grid: this device cannot host the persistent engine (not a transient, so no
      // re-probe); decode the request on the launch-per-stage engine
      persist_state = 0;
      persist_probe_ok = false;
      persist_refused = true;
      std::fprintf(stderr, "libxdtts_hip: persistent decoder launch refused by the runtime; this handle uses the "
                           "launch-per-stage decoder\n");
      HIP_CHECK(hipStreamSynchronize(stream));
      HIP_CHECK(hipMemsetAsync(dec_err.p, 0, sizeof(int), stream));
      launch_decoder_init(d, limits.p, stream);
    }
    // the launch that holds the attention LSTM and the attention needs its 256 blocks resident together: like the
    // persistent engine's, such launches of different handles never overlap
    std::unique_lock<ChipLock> chip;
    if (d.hg) chip = std::unique_lock<ChipLock>(chip_mutex(device));
    if (d.hring)  // the slabs of the steps this request can reach: "not yet written"
      HIP_CHECK(hipMemsetAsync(d.hring, 0xff, (size_t)std::min(max_lim, d.hring_steps) * ATT_RNN * d.Bpad * sizeof(unsigned), stream));
    if (d.hcnt) {
      HIP_CHECK(hipMemsetAsync(d.hstage, 0xff, (size_t)16 * ATT_RNN * d.Bpad * sizeof(unsigned), stream));
      HIP_CHECK(hipMemsetAsync(d.hcnt, 0, (size_t)d.hring_steps * 8 * 64 * sizeof(unsigned), stream));
    }
    if (!d.use_gate) {  // deterministic work: every chunk runs to its cap
      while (launched + GRAPH_STEPS <= max_lim) {
        replay_steps(d);
        launched += GRAPH_STEPS;
      }
      if (launched < max_lim) {
        // the last < GRAPH_STEPS steps as plain launches of exactly that many (even) steps: a whole graph would run up to
        // GRAPH_STEPS - 1 steps with no chunk active, three launches of ~3.5 us each (the 647-iteration batch of configs[2]: 13)
        int r = max_lim - launched;
        r += r & 1;
        launch_decoder_steps(d, w, r, stream);
        launched += r;
      }
      launch_decoder_flush(d, w, stream);
      fetch(true);
    } else {
      const int check_every = 3 * GRAPH_STEPS;
      for (;;) {
        for (int i = 0; i < check_every && launched < max_lim; i += GRAPH_STEPS) {
          replay_steps(d);
          launched += GRAPH_STEPS;
        }
        fetch();
        int need = 0;
        for (int b = 0; b < d.B; ++b) need = std::max(need, host_ctl[HOST_NF + b]);
        if (host_ctl[0] >= need || launched >= max_lim) break;
      }
      // the projection of step s is completed by the first kernel of step s+1: finish the last one
      launch_decoder_flush(d, w, stream);
      fetch(true);
    }
    if (d.ep_g) {
      if (host_ctl[HOST_DEC_ERR]) {  // a block of the one-launch attention never saw its neighbours' energies: not silent, not fatal
        HIP_CHECK(hipMemsetAsync(dec_err.p, 0, sizeof(int), stream));
        att_fused = 0;
        att_demoted = true;
        att_demoted_calls = 0;
        std::fprintf(stderr, "libxdtts_hip: batched attention exchange timed out; this handle now uses the "
                             "separate attention kernels (probed again after %d batches)\n", PROBE_AFTER);
        DecoderBufs d2 = d;
        d2.ep_g = nullptr;
        d2.hg = nullptr;
        d2.att_part = nullptr;
        d2.hdg = d2.melg = nullptr;
        d2.dec_part = nullptr;
        d2.hring = nullptr;
        d2.hstage = d2.hcnt = nullptr;
        d2.att_hfirst = 0;
        return run_decoder(d2, lim);  // (no `after`: the caller enqueues its work behind this decode)
      }
    }
    return finish();
  }

  // postnet.onnx (mod.rs:345-355) for one chunk: frames_dev [F][80] (row stride 80) ->
  // out[m * ldc + t] for m < 80, t < F  (the (80 x F) Array2 layout), residual included.
  // postnet.onnx (mod.rs:345-355) for n <= GEMM_RAGGED_MAX chunks in one launch per layer: chunk i has
  // F[i] frames at frames_dev + i * frame_stride ([F][80], row stride 80) and its (80 x F[i]) result
  // goes to out + col_off[i] with row stride ldc (the (80 x F_total) Array2 layout), residual included.
  // dense_items: chunk i's result is a dense (80 x F[i]) matrix of its own at out + col_off[i] (ldc unused).
  void run_postnet(const float *frames_dev, size_t frame_stride, const int *F, const long *col_off, int n, float *out,
                   long ldc, bool dense_items = false) {
    const int pad = (POST_K - 1) / 2;
    int Fmax = 0;
    for (int i = 0; i < n; ++i) Fmax = std::max(Fmax, F[i]);
    const size_t FP = (size_t)Fmax + 2 * pad, slot = FP * POST_CH, slot0 = FP * N_MEL;
    pp0.alloc(slot0 * n);
    ppA.alloc(slot * n);
    ppB.alloc(slot * n);
    // What must read as zero -- the padding rows and, in a ragged group, the rows between a chunk's end and the longest chunk's
    // -- is never written (the copy and the convolutions store rows pad .. pad + F[z] - 1 of item z), so the three fills are
    // due when the group's layout or an allocation changes, not per request (the 80-channel input has its own buffer for that:
    // as a second view of ppB it left 80-channel rows where the 512-channel layout has its padding)
    {
      std::vector<long> sig{(long)n, (long)Fmax, (long)(size_t)pp0.p, (long)(size_t)ppA.p, (long)(size_t)ppB.p};
      for (int z = 0; z < n; ++z) sig.push_back(F[z]);
      if (sig != pp_sig) {
        HIP_CHECK(hipMemsetAsync(pp0.p, 0, slot0 * n * sizeof(float), stream));
        HIP_CHECK(hipMemsetAsync(ppA.p, 0, slot * n * sizeof(float), stream));
        HIP_CHECK(hipMemsetAsync(ppB.p, 0, slot * n * sizeof(float), stream));
        pp_sig = sig;
      }
    }
    // layer 0 input: the frames themselves in zero-padded [FP][80] buffers
    launch_copy_rows(frames_dev, frame_stride, pp0.p + (size_t)pad * N_MEL, slot0, F, n, N_MEL, stream);
    float *src = pp0.p, *dst = ppA.p;
    for (int i = 0; i < POST_CONVS; ++i) {
      const ConvGemm &c = w.post_conv[i];
      const bool last = i == POST_CONVS - 1;
      GemmArgs g{};
      g.A = src;
      g.lda = c.ci;
      g.strideA = (long)(i == 0 ? slot0 : slot);
      g.W = c.w.p;
      g.bias = c.b.p;
      g.M = Fmax;
      g.N = c.co;
      g.K = c.k * c.ci;
      g.batch = n;
      g.ragged = 1;
      for (int z = 0; z < n; ++z) {
        g.Mz[z] = F[z];
        g.Cz[z] = last ? col_off[z] : 0;
      }
      if (!last) {
        g.C = dst + (size_t)pad * c.co;
        g.ldc = c.co;
        g.strideC = (long)slot;
        g.act = 2;
      } else {
        g.C = out;
        g.ldc = ldc;
        g.transpose_out = 1;
        g.ldc_rows = dense_items ? 1 : 0;
        g.R = frames_dev;
        g.ldr = N_MEL;
        g.strideR = (long)frame_stride;
      }
      run_gemm(g);
      if (i == 0) src = ppB.p;  // (layers 1.. ping-pong between the two 512-channel buffers)
      std::swap(src, dst);
    }
  }

  // infer_chunk x B (mod.rs:361-393).  ids_host [B][T] already zero-padded.  Leaves the final mel
  // of chunk b at mel_dev + col_off[b] with row stride F_total; returns per-chunk frame counts.
  // per_chunk: mel_dev receives one dense (80 x F_b) matrix per chunk instead, back to back in the caller's order.
  std::vector<int> infer_batch_device(const int64_t *ids_host, const int *lens, int B, int T,
                                      const xdtts_infer_opts &o, const int *fixed_per_item, int *F_total, bool per_chunk = false) {
    if (B <= 0 || B > 4096) fail(XDTTS_ERR_BAD_ARG, "batch %d out of range", B);
    if (T <= 0 || T > T_MAX) fail(XDTTS_ERR_BAD_ARG, "window %d out of range (1..%d)", T, T_MAX);
    if (o.max_steps <= 0 || o.max_steps > 100000) fail(XDTTS_ERR_BAD_ARG, "max_steps %d out of range", o.max_steps);
    for (int b = 0; b < B; ++b) {
      if (lens[b] <= 0) fail(XDTTS_ERR_BAD_ARG, "chunk %d is empty", b);
      if (lens[b] > T) fail(XDTTS_ERR_TOO_LONG, "chunk %d has %d ids, window is %d", b, lens[b], T);
      for (int t = 0; t < T; ++t) {
        const int64_t id = ids_host[(size_t)b * T + t];
        if (id < 0 || id >= N_SYMBOLS) fail(XDTTS_ERR_BAD_ARG, "id %lld out of range (0..%d)", (long long)id, N_SYMBOLS - 1);
      }
    }
    HIP_CHECK(hipSetDevice(device));
    HIP_CHECK(hipEventRecord(ev.e[0], stream));
    // Step caps per chunk, then (batched mode) the lock-step order: longest first, so that the chunks
    // still running always fill a prefix of the 16-chunk MFMA tiles and finished tiles are skipped.
    // order[j] = caller's index of the chunk decoded in slot j; everything below works on slots.
    std::vector<int> lim0(B), order(B);
    for (int b = 0; b < B; ++b) {
      int l = o.max_steps;
      if (fixed_per_item) l = fixed_per_item[b];
      else if (o.fixed_steps > 0) l = o.fixed_steps;
      else if (o.fixed_frames_per_id > 0.f) l = (int)std::lround((double)o.fixed_frames_per_id * lens[b]);
      lim0[b] = std::min(std::max(l, 1), o.max_steps);
      order[b] = b;
    }
    upload_dropout_masks(o, B, lim0.data());  // (caller's chunk order: a sorted batch finds its masks through item_perm)
    const bool gate_off = fixed_per_item || o.fixed_steps > 0 || o.fixed_frames_per_id > 0.f;
    // (3..8 chunks on the persistent MFMA engine keep the row-major state of the small-batch engines and the caller's order)
    const bool batched_mode = B >= BATCH_MFMA_MIN && !small_batch_engine(B, T, *std::max_element(lim0.begin(), lim0.end()));
    if (batched_mode)
      std::stable_sort(order.begin(), order.end(), [&](int a, int c) {
        return gate_off ? lim0[a] > lim0[c] : lens[a] > lens[c];  // with the gate on, length is the proxy for duration
      });
    std::vector<int64_t> ids_sorted((size_t)B * T);
    std::vector<int> lens_sorted(B), lim(B);
    for (int j = 0; j < B; ++j) {
      std::copy(ids_host + (size_t)order[j] * T, ids_host + (size_t)(order[j] + 1) * T, ids_sorted.begin() + (size_t)j * T);
      lens_sorted[j] = lens[order[j]];
      lim[j] = lim0[order[j]];
    }
    ids_host = ids_sorted.data();
    lens = lens_sorted.data();
    {  // ids, lengths, step caps and (batched mode) the dropout-stream order: one pinned block, one copy (it was four of 4-5 us
       // each, with the host's enqueue time in front of every one of them at the start of a request)
      const size_t b_ids = sizeof(int64_t) * (size_t)B * T, b_int = sizeof(int) * (size_t)B, need = b_ids + 3 * b_int;
      if (need > in_host_bytes) {
        if (in_host) (void)hipHostFree(in_host);
        in_host = nullptr;
        in_host_bytes = 0;
        HIP_CHECK(hipHostMalloc((void **)&in_host, need, hipHostMallocDefault));
        in_host_bytes = need;
      }
      std::memcpy(in_host, ids_host, b_ids);
      std::memcpy(in_host + b_ids, lens, b_int);
      std::memcpy(in_host + b_ids + b_int, lim.data(), b_int);
      std::memcpy(in_host + b_ids + 2 * b_int, order.data(), b_int);
      in_blk.alloc(need);
      HIP_CHECK(hipMemcpyAsync(in_blk.p, in_host, need, hipMemcpyHostToDevice, stream));  // (run_decoder's final wait is behind it)
      ids.p = reinterpret_cast<int64_t *>(in_blk.p);
      n_valid.p = reinterpret_cast<int *>(in_blk.p + b_ids);
      limits.p = reinterpret_cast<int *>(in_blk.p + b_ids + b_int);
      item_perm.p = reinterpret_cast<int *>(in_blk.p + b_ids + 2 * b_int);
      lim_on_dev = lim;
    }
    std::lock_guard<ChipLock> chip(chip_mutex(device));  // released after run_decoder's final wait
    run_encoder(B, T);
    HIP_CHECK(hipEventRecord(ev.e[1], stream));
    // (the cooperative BiLSTM's error word comes back with the decoder's own final fetch: same block, same copy)
    if (batched_mode) w.ensure_batched_layout(blob, stream);
    if (before_decoder) before_decoder();  // (xdtts_synthesize_sequence: the frame loop waits for the previous utterance's vocoder)
    DecoderBufs d = decoder_bufs(B, T, memory.p, pmem.p, o, batched_mode ? 1 : 0);
    if (batched_mode) d.item_perm = item_perm.p;
    if (fixed_per_item || o.fixed_frames_per_id > 0.f) d.use_gate = 0;
    // everything behind the decoder: frame counts -> column offsets -> post-net.  A gate-less decode on the persistent
    // engine enqueues it BEFORE the sync that fetches the counts (they are the caps), see run_decoder.
    std::vector<int> Fs(B), F(B);  // frames per slot / per caller index
    int total = 0;
    auto postnet_all = [&](const int *frames_per_slot) {
      HIP_CHECK(hipEventRecord(ev.e[2], stream));
      total = 0;
      for (int j = 0; j < B; ++j) {
        Fs[j] = frames_per_slot[j];
        F[order[j]] = Fs[j];
        total += Fs[j];
      }
      // (In a sequence the vocoder of the PREVIOUS utterance may still be reading mel_dev on its own stream when this runs for the next
      // one.  DevBuf::alloc only ever grows: the buffer is kept unless this utterance is longer than every one before it, and then the
      // hipFree inside it synchronises the whole device before the old buffer goes -- correct, at the price of that one overlap.  The
      // post-net's own writes into a kept buffer are ordered behind the vocoder by the event the frame loop waits for, before_decoder.)
      mel_dev.alloc((size_t)N_MEL * total);
      std::vector<long> col0(B), col(B);  // the final mel keeps the caller's chunk order on the time axis (mod.rs:430)
      long off = 0;
      for (int b = 0; b < B; ++b) {
        col0[b] = off;
        off += F[b];
      }
      for (int j = 0; j < B; ++j) col[j] = (per_chunk ? N_MEL : 1) * col0[order[j]];
      for (int b = 0; b < B; b += GEMM_RAGGED_MAX) {
        const int n = std::min(GEMM_RAGGED_MAX, B - b);
        run_postnet(d.frames + (size_t)b * d.max_steps * N_MEL, (size_t)d.max_steps * N_MEL, Fs.data() + b, col.data() + b, n,
                    mel_dev.p, total, per_chunk);
      }
      HIP_CHECK(hipEventRecord(ev.e[3], stream));
    };
    bool postnet_done = false;
    last_steps = run_decoder(d, lim, [&] { postnet_all(lim.data()); }, &postnet_done);  // (the decoder has finished; the post-net may be running)
    if (host_ctl[HOST_ENC_ERR] != 0) {
      postnet_done = false;
      HIP_CHECK(hipMemsetAsync(enc_err.p, 0, sizeof(int), stream));
      // the 4-CU cooperative BiLSTM needs its workgroups co-resident too: same policy as the decoder --
      // say so, use the single-workgroup recurrence from now on, and run the request again
      coop_ok = false;
      std::fprintf(stderr, "libxdtts_hip: encoder BiLSTM exchange timed out (grid not co-resident); "
                           "this handle now uses the single-workgroup recurrence\n");
      run_encoder(B, T);
      // batched mode attends over the [B][32][T][4] transpose of processed_memory, written by decoder_bufs() from the
      // timed-out encoder's output: redo it from the fresh one
      if (d.pmem_t) launch_dimgroup_transpose(pmem.p, pmem_t.p, B, T, stream);
      last_steps = run_decoder(d, lim);
    }
    if (!postnet_done) postnet_all(host_ctl + HOST_NF);
    *F_total = total;
    return F;
  }

  // the cooperative BiLSTM bounds its spins; a timeout there must not pass silently
  bool encoder_exchange_failed() {
    int e = 0;
    HIP_CHECK(hipMemcpyAsync(&e, enc_err.p, sizeof(int), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    if (e) HIP_CHECK(hipMemsetAsync(enc_err.p, 0, sizeof(int), stream));
    return e != 0;
  }
  void check_encoder_exchange() {
    if (encoder_exchange_failed()) {
      coop_ok = false;
      fail(XDTTS_ERR_HIP, "encoder BiLSTM hidden-state exchange timed out (retry uses the single-workgroup recurrence)");
    }
  }

  void finish_timings() {
    HIP_CHECK(hipStreamSynchronize(stream));
    for (int i = 0; i < 3; ++i) HIP_CHECK(hipEventElapsedTime(&last_ms[i], ev.e[i], ev.e[i + 1]));
    HIP_CHECK(hipEventElapsedTime(&last_ms[3], ev.e[0], ev.e[3]));
  }
};

static void chunks_from_splits(const int64_t *ids, size_t n, const size_t *splits, size_t n_splits, int T,
                               std::vector<int64_t> &padded, std::vector<int> &lens) {
  if (!ids || n == 0) fail(XDTTS_ERR_BAD_ARG, "empty id sequence");
  std::vector<size_t> ends;
  if (splits && n_splits) ends.assign(splits, splits + n_splits);
  if (ends.empty() || ends.back() != n) ends.push_back(n);  // mod.rs:412-414
  size_t start = 0;
  for (size_t e : ends) {
    if (e < start || e > n) fail(XDTTS_ERR_BAD_ARG, "splits must be ascending offsets into ids");
    if (e == start) continue;
    const size_t len = e - start;
    if ((int)len > T) fail(XDTTS_ERR_TOO_LONG, "chunk of %zu ids exceeds the %d-id window", len, T);  // mod.rs:363
    lens.push_back((int)len);
    const size_t base = padded.size();
    padded.resize(base + (size_t)T, 0);  // pad id 0 = Unit::Padding, mod.rs:369-371
    std::copy(ids + start, ids + e, padded.begin() + (long)base);
    start = e;
  }
}

// ================================================================================================
// Griffin-Lim handle
// ================================================================================================
struct xdtts_griffinlim {
  int device = 0;
  hipStream_t stream = nullptr;
  mutable std::mutex mu;
  int n_mels = 0, nb = 0, n_fft = 0, hop = 0, iters = 0;
  float power = 1.f, momentum = 0.99f;
  uint32_t seed = 0;
  Events ev;
  float last_ms[3] = {0, 0, 0};
  DevBuf<float> pinv, win, S, melT, mel_in, frames, wss_inv, audio, phase0;
  // mel->linear options (xdtts_griffinlim_opts) and the NNLS refinement's operands
  xdtts_griffinlim_opts gopts = [] { xdtts_griffinlim_opts o; xdtts_griffinlim_opts_default(&o); return o; }();  // one source for the defaults
  static constexpr int NBP = 528;  // bins padded to the GEMM's K granule
  DevBuf<float> basis_p, basisT_p, nnls_x, nnls_r, norm_parts;  // norm_parts: GLN_SCRATCH per utterance
  DevBuf<int2> norm_tab;  // (first sample, samples) per utterance of a vocoder batch
  float nnls_step = 0.f;   // 1 / lambda_max(A A^T)
  hipGraphExec_t graph = nullptr;  // n_iter x (istft, stft) + final ISTFT for the cached (buffers, F, iterations)
  GlBufs graph_key{};
  int graph_iters = -1;
  float graph_alpha = 0.f;
  const float *graph_audio = nullptr;
  DevBuf<float2> tw, ang, ang2, tprev, tprev2;  // tprev2: final rebuilt spectrum of the parity hook
  // persistent engine (griffinlim.hip: k_gl_persistent)
  DevBuf<unsigned long long> xch;  // neighbour-overlap granules
  DevBuf<GlSeg> segs;              // vocoder batch: per-workgroup segment table
  DevBuf<int> frame_local;         // vocoder batch: row -> frame index inside its utterance
  DevBuf<int> gl_err;
  int *host_err = nullptr;         // pinned
  unsigned epoch = 0;              // tag base; tags are never reused while xch lives
  int persist_state = -1;          // -1 unknown, 0 unavailable / demoted, 1 usable
  int n_cu = 0;
  int per_cu4 = 1;                 // co-resident workgroups of the 4-frame shape per CU (vocoder batch)
  bool last_persistent = false;    // the last run_iterations used the persistent engine
  int demoted_calls = 0;           // calls since a demotion (the engine is probed again after PROBE_AFTER)
  static constexpr int PROBE_AFTER = 64;
  // vocoder batch: the audio of a finished launch goes to the host while the next launches run
  hipStream_t copy_stream = nullptr;
  std::vector<hipEvent_t> copy_ev;
  hipEvent_t launch_done(size_t k) {
    if (!copy_stream) HIP_CHECK(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
    while (copy_ev.size() <= k) {
      hipEvent_t e = nullptr;
      HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      copy_ev.push_back(e);
    }
    return copy_ev[k];
  }

  ~xdtts_griffinlim() {
    for (hipEvent_t e : copy_ev) (void)hipEventDestroy(e);
    if (copy_stream) (void)hipStreamDestroy(copy_stream);
    if (host_err) (void)hipHostFree(host_err);
    if (graph) (void)hipGraphExecDestroy(graph);
    if (stream) (void)hipStreamDestroy(stream);
  }

  GlBufs bufs(int F) {
    S.alloc((size_t)F * nb);
    ang.alloc((size_t)F * nb);
    ang2.alloc((size_t)F * nb);
    tprev.alloc((size_t)F * nb);
    frames.alloc((size_t)F * n_fft);
    wss_inv.alloc((size_t)std::max(1, hop * (F - 1)));
    audio.alloc((size_t)std::max(1, hop * (F - 1)));
    GlBufs g{};
    g.F = F;
    g.n_fft = n_fft;
    g.hop = hop;
    g.nb = nb;
    g.S = S.p;
    g.ang = ang.p;
    g.ang2 = ang2.p;
    g.tprev = tprev.p;
    g.frames = frames.p;
    g.wss_inv = wss_inv.p;
    g.tw = tw.p;
    g.win = win.p;
    return g;
  }

  // step 1 of GriffinLim::infer: mel (device, n_mels x F) -> S [F][nb], with the convention switches of
  // xdtts_griffinlim_opts: de-compression, optional projected-gradient NNLS refinement, exponent.
  //   x0 = max(pinv m, 0);  x <- max(x - (1/L) A^T (A x - m), 0)  nnls_iters times, batched over the
  //   frames as two MFMA GEMMs per step (residual [F][80], then the update of X [F][NBP]).
  void mel_to_linear(const float *mel_dev_ptr, int F) {
    melT.alloc((size_t)F * n_mels);
    launch_gl_exp_transpose(mel_dev_ptr, melT.p, n_mels, F, gopts.mel_decompress, stream);
    const float ex = gopts.power_mode == 0 ? 1.0f / power : (gopts.power_mode == 1 ? power : 1.0f);
    GemmArgs a{};
    a.A = melT.p;
    a.lda = n_mels;
    a.W = pinv.p;
    a.M = F;
    a.N = nb;
    a.K = n_mels;
    a.batch = 1;
    if (gopts.nnls_iters <= 0) {
      a.C = S.p;
      a.ldc = nb;
      a.act = ex == 1.0f ? 1 : 3;
      a.p = ex;
      launch_gemm_nt(a, stream);
      return;
    }
    nnls_x.alloc((size_t)F * NBP);
    nnls_r.alloc((size_t)F * n_mels);
    HIP_CHECK(hipMemsetAsync(nnls_x.p, 0, (size_t)F * NBP * sizeof(float), stream));  // padding columns stay 0
    a.C = nnls_x.p;
    a.ldc = NBP;
    a.act = 1;
    launch_gemm_nt(a, stream);
    for (int it = 0; it < gopts.nnls_iters; ++it) {
      GemmArgs r{};  // R = X A^T - m
      r.A = nnls_x.p;
      r.lda = NBP;
      r.W = basis_p.p;  // [n_mels][NBP]
      r.C = nnls_r.p;
      r.ldc = n_mels;
      r.M = F;
      r.N = n_mels;
      r.K = NBP;
      r.batch = 1;
      r.R = melT.p;
      r.ldr = n_mels;
      r.beta = -1.0f;
      launch_gemm_nt(r, stream);
      GemmArgs u{};  // X = max(X - (1/L) R A, 0)
      u.A = nnls_r.p;
      u.lda = n_mels;
      u.W = basisT_p.p;  // [NBP][n_mels]
      u.C = nnls_x.p;
      u.ldc = NBP;
      u.M = F;
      u.N = NBP;
      u.K = n_mels;
      u.batch = 1;
      u.alpha = -nnls_step;
      u.R = nnls_x.p;
      u.ldr = NBP;
      u.r_before_act = 1;
      u.act = 1;
      launch_gemm_nt(u, stream);
    }
    launch_gl_pow_rows(nnls_x.p, NBP, S.p, nb, F, ex, stream);
  }

  // Once per API call (never inside a retry attempt): a demoted handle counts the call and, after PROBE_AFTER of them,
  // gives the persistent engine another try -- the cause of a timed-out exchange may have been transient.
  void probe_tick() {
    if (persist_state == 0 && n_cu > 0 && ++demoted_calls >= PROBE_AFTER) {
      demoted_calls = 0;
      persist_state = 1;
    }
  }
  bool persistent_usable() {
    const char *e = getenv("XDTTS_GL");
    if (e && std::string(e) == "launch") return false;  // developer comparison aid: launch-per-iteration engine
    if (persist_state < 0) persist_state = gl_persistent_supported(device, &n_cu, &per_cu4) ? 1 : 0;
    return persist_state == 1;
  }

  // The iteration engine on the current state (ang, tprev, S in place): n_iter iterations and, when
  // audio_out is given, the final ISTFT into it.  Returns the buffer holding the final angles
  // (*tprev_fin: the final rebuilt spectrum).  Small-to-medium frame counts run as ONE persistent
  // launch; the caller holds the chip lock until the stream has drained and then asks
  // persistent_failed().
  const float2 *run_iterations(const GlBufs &g, int n_iter, float alpha, float *audio_out, bool want_state = false,
                               const float2 **tprev_fin = nullptr, bool gen_phase = false) {
    int TF = 0, nblk = 0;
    last_persistent = false;
    err_fetched = false;
    if (tprev_fin) *tprev_fin = g.tprev;
    if (persistent_usable() && gl_persistent_plan(g.F, n_cu, &TF, &nblk)) {
      const size_t words = gl_persistent_xch_words(nblk);
      if (words > xch.n || epoch > 0x7fff0000u - (unsigned)n_iter) {  // fresh (or wrapped) tags: clear every granule
        xch.alloc(words);
        HIP_CHECK(hipMemsetAsync(xch.p, 0, xch.n * sizeof(unsigned long long), stream));
        epoch = 0;
      }
      if (!gl_err.p) {
        gl_err.alloc(1);
        HIP_CHECK(hipMemsetAsync(gl_err.p, 0, sizeof(int), stream));
        HIP_CHECK(hipHostMalloc((void **)&host_err, sizeof(int), hipHostMallocDefault));
      }
      GlPersist p{};
      p.xch = xch.p;
      p.err = gl_err.p;
      p.epoch = epoch;
      p.nblk = nblk;
      p.TF = TF;
      p.poll_delay = 6;  // first poll 6 x 128 clocks after the publish (tools/gl_poll_sweep.py, F = 1000: 4.31-4.36 us per iteration at 5..7, 4.39 behind the overlap-add, 4.45-4.53 at 2 or 10..12)
      if (const char *sp = getenv("XDTTS_GL_SPINS")) p.spins = atoi(sp);  // test hook
      if (const char *sl = getenv("XDTTS_GL_SLOW")) p.slow = atoi(sl);    // test hook: straggler workgroup
      if (const char *pd = getenv("XDTTS_GL_POLL_DELAY")) p.poll_delay = atoi(pd);  // developer sweep
      p.gen_phase = gen_phase ? 1 : 0;
      p.seed = seed;
      epoch += (unsigned)n_iter + 2u;
      if (want_state) {
        p.ang_out = g.ang2;
        tprev2.alloc((size_t)g.F * g.nb);
        p.tprev_out = tprev2.p;
        if (tprev_fin) *tprev_fin = p.tprev_out;
      }
#ifdef XDTTS_GL_PROFILE
      static DevBuf<unsigned long long> prof;
      prof.alloc(256 * 12);
      p.prof = prof.p;
#endif
      launch_gl_persistent(g, p, g.ang, g.tprev, n_iter, alpha, audio_out, stream);
#ifdef XDTTS_GL_PROFILE
      if (const char *path = getenv("XDTTS_GL_PROFILE")) {
        std::vector<unsigned long long> hp((size_t)nblk * 12);
        HIP_CHECK(hipMemcpyAsync(hp.data(), prof.p, hp.size() * 8, hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        if (FILE *f = fopen(path, "w")) {
          fprintf(f, "%d %d\n", nblk, n_iter);
          for (int c = 0; c < nblk; ++c) {
            for (int i = 0; i < 12; ++i) fprintf(f, "%llu ", hp[(size_t)c * 12 + i]);
            fprintf(f, "\n");
          }
          fclose(f);
        }
      }
#endif
      last_persistent = true;
      return want_state ? g.ang2 : g.ang;
    }
    const float2 *fin = launch_gl_iterate(g, n_iter, alpha, stream);
    if (audio_out) launch_gl_final(g, fin, audio_out, stream);
    return fin;
  }

  // After the stream has drained: did a bounded spin of the persistent launch run out (grid not
  // co-resident)?  If so the handle is demoted to the launch-per-iteration engine (and probes the
  // persistent one again after PROBE_AFTER calls); the input state is intact, the caller re-runs.
  // (fetch_error_word() ahead of a sync the caller needs anyway saves persistent_failed() its own round trip)
  bool err_fetched = false;
  void fetch_error_word() {
    if (!last_persistent) return;
    HIP_CHECK(hipMemcpyAsync(host_err, gl_err.p, sizeof(int), hipMemcpyDeviceToHost, stream));
    err_fetched = true;
  }
  bool persistent_failed() {
    if (!last_persistent) return false;
    if (!err_fetched) {
      HIP_CHECK(hipMemcpyAsync(host_err, gl_err.p, sizeof(int), hipMemcpyDeviceToHost, stream));
      HIP_CHECK(hipStreamSynchronize(stream));
    }
    err_fetched = false;
    if (!*host_err) return false;
    HIP_CHECK(hipMemsetAsync(gl_err.p, 0, sizeof(int), stream));
    HIP_CHECK(hipMemsetAsync(xch.p, 0, xch.n * sizeof(unsigned long long), stream));
    epoch = 0;
    persist_state = 0;
    demoted_calls = 0;
    std::fprintf(stderr, "libxdtts_hip: persistent Griffin-Lim exchange timed out (grid not co-resident); "
                         "this handle uses the launch-per-iteration engine for the next %d calls\n", PROBE_AFTER);
    return true;
  }

  // phase init + iterations + final ISTFT; S already in place.  Result in audio (device).
  void iterate(const GlBufs &g, const float *phase0_dev, int n_iter) {
    const float alpha = momentum / (1.0f + momentum);
    int TF = 0, nblk = 0;
    if (persistent_usable() && gl_persistent_plan(g.F, n_cu, &TF, &nblk)) {
      // one launch: nothing to capture; with the seeded stream the kernel draws the phase itself (no
      // phase-init launch, no window-sum table: the kernel keeps its own)
      if (phase0_dev) launch_gl_phase_init(g, seed, phase0_dev, stream);
      run_iterations(g, n_iter, alpha, audio.p, false, nullptr, phase0_dev == nullptr);
      return;
    }
    launch_gl_phase_init(g, seed, phase0_dev, stream);
    launch_gl_prepare(g, stream);
    last_persistent = false;
    // the launch-per-iteration loop is launch-bound: replay it as one hipGraph
    if (!graph || std::memcmp(&graph_key, &g, sizeof g) != 0 || graph_iters != n_iter || graph_alpha != alpha ||
        graph_audio != audio.p) {
      if (graph) {
        (void)hipGraphExecDestroy(graph);
        graph = nullptr;
      }
      hipGraph_t gr = nullptr;
      HIP_CHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
      try {
        launch_gl_iterations(g, n_iter, alpha, audio.p, stream);
      } catch (...) {
        (void)hipStreamEndCapture(stream, &gr);
        if (gr) (void)hipGraphDestroy(gr);
        throw;
      }
      HIP_CHECK(hipStreamEndCapture(stream, &gr));
      hipError_t e = hipGraphInstantiate(&graph, gr, nullptr, nullptr, 0);
      (void)hipGraphDestroy(gr);
      HIP_CHECK(e);
      graph_key = g;
      graph_iters = n_iter;
      graph_alpha = alpha;
      graph_audio = audio.p;
    }
    HIP_CHECK(hipGraphLaunch(graph, stream));
  }

  void finish_timings() {
    HIP_CHECK(hipStreamSynchronize(stream));
    HIP_CHECK(hipEventElapsedTime(&last_ms[0], ev.e[0], ev.e[1]));
    HIP_CHECK(hipEventElapsedTime(&last_ms[1], ev.e[1], ev.e[2]));
    HIP_CHECK(hipEventElapsedTime(&last_ms[2], ev.e[0], ev.e[2]));
  }
};

// Slaney-scale helpers for create_mel_filter_bank (librosa.filters.mel, htk=False, norm="slaney")
static double hz_to_mel(double f) {
  const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp;
  const double logstep = std::log(6.4) / 27.0;
  return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp;
}
static double mel_to_hz(double m) {
  const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp;
  const double logstep = std::log(6.4) / 27.0;
  return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m;
}

// pinv(A) = A^T (A A^T)^-1 for the full-row-rank mel basis, fp64 Cholesky.  librosa's nnls starts
// from lstsq(A, M) clipped at 0 and its L-BFGS-B refinement stops at iteration 0 for this
// objective scaling, so clip(pinv M, 0) is the inversion the vocoder performs.
static void host_pinv(const float *basis, int n, int nbins, std::vector<float> &out) {
  std::vector<double> G((size_t)n * n, 0.0), z(n);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = 0;
      for (int b = 0; b < nbins; ++b) s += (double)basis[(size_t)i * nbins + b] * basis[(size_t)j * nbins + b];
      G[(size_t)i * n + j] = G[(size_t)j * n + i] = s;
    }
  for (int j = 0; j < n; ++j) {
    double d = G[(size_t)j * n + j];
    for (int k = 0; k < j; ++k) d -= G[(size_t)j * n + k] * G[(size_t)j * n + k];
    if (!(d > 0)) fail(XDTTS_ERR_BAD_ARG, "mel basis is rank deficient (filter %d)", j);
    d = std::sqrt(d);
    G[(size_t)j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = G[(size_t)i * n + j];
      for (int k = 0; k < j; ++k) s -= G[(size_t)i * n + k] * G[(size_t)j * n + k];
      G[(size_t)i * n + j] = s / d;
    }
  }
  out.resize((size_t)nbins * n);
  for (int b = 0; b < nbins; ++b) {
    for (int i = 0; i < n; ++i) {
      double s = basis[(size_t)i * nbins + b];
      for (int k = 0; k < i; ++k) s -= G[(size_t)i * n + k] * z[k];
      z[i] = s / G[(size_t)i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
      double s = z[i];
      for (int k = i + 1; k < n; ++k) s -= G[(size_t)k * n + i] * z[k];
      z[i] = s / G[(size_t)i * n + i];
    }
    for (int i = 0; i < n; ++i) out[(size_t)b * n + i] = (float)z[i];
  }
}

// lambda_max(A A^T) by power iteration (double): the Lipschitz constant of the NNLS gradient
static double host_lipschitz(const float *basis, int n, int nbins) {
  std::vector<double> G((size_t)n * n, 0.0), v(n, 1.0 / std::sqrt((double)n)), w(n);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = 0;
      for (int b = 0; b < nbins; ++b) s += (double)basis[(size_t)i * nbins + b] * basis[(size_t)j * nbins + b];
      G[(size_t)i * n + j] = G[(size_t)j * n + i] = s;
    }
  double lam = 0;
  for (int it = 0; it < 1000; ++it) {
    double nrm = 0;
    for (int i = 0; i < n; ++i) {
      double a = 0;
      for (int j = 0; j < n; ++j) a += G[(size_t)i * n + j] * v[j];
      w[i] = a;
      nrm += a * a;
    }
    nrm = std::sqrt(nrm);
    if (!(nrm > 0)) fail(XDTTS_ERR_BAD_ARG, "mel basis is all zero");
    for (int i = 0; i < n; ++i) v[i] = w[i] / nrm;
    const bool done = std::fabs(nrm - lam) <= 1e-13 * nrm;
    lam = nrm;
    if (done) break;
  }
  return lam;
}

// ================================================================================================
// extern "C"
// ================================================================================================
extern "C" {

void xdtts_infer_opts_default(xdtts_infer_opts *o) {
  if (!o) return;
  o->gate_threshold = 0.6f;  // src/tacotron2/mod.rs:279
  o->max_steps = 1000;       // src/tacotron2/mod.rs:280
  o->fixed_steps = 0;
  o->dropout_mode = 1;
  o->dropout_seed = 0;
  o->max_chunk = 100;  // src/tacotron2/mod.rs:363,369-371,399
  o->item_base = 0;
  o->fixed_frames_per_id = 0.f;
  o->dropout_masks = nullptr;
  o->dropout_mask_steps = 0;
}

const char *xdtts_last_error(void) { return g_last_error.c_str(); }

int32_t xdtts_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

void xdtts_free(void *p) {
  if (p) pinned_pool().put(p);
}

int32_t xdtts_tensor_count(void) { return (int32_t)tensor_table().size(); }
const char *xdtts_tensor_name(int32_t i) {
  return i >= 0 && i < xdtts_tensor_count() ? tensor_table()[i].name : nullptr;
}
int32_t xdtts_tensor_ndim(int32_t i) { return i >= 0 && i < xdtts_tensor_count() ? tensor_table()[i].ndim : 0; }
int32_t xdtts_tensor_dim(int32_t i, int32_t d) {
  return i >= 0 && i < xdtts_tensor_count() && d >= 0 && d < 3 ? tensor_table()[i].dims[d] : 0;
}
size_t xdtts_tensor_offset(int32_t i) { return i >= 0 && i < xdtts_tensor_count() ? tensor_table()[i].offset : 0; }
size_t xdtts_tensor_total(void) { return tensor_total(); }

static xdtts_status make_handle(std::vector<float> &&blob, int32_t device_id, xdtts_tacotron2 **out) {
  return guard([&] {
    if (!out) fail(XDTTS_ERR_BAD_ARG, "out handle pointer is null");
    *out = nullptr;
    auto h = std::make_unique<xdtts_tacotron2>();
    h->blob = std::move(blob);
    h->init(device_id);
    *out = h.release();
  });
}

xdtts_status xdtts_tacotron2_load(const char *dir, int32_t device_id, xdtts_tacotron2 **out) {
  std::vector<float> blob;
  xdtts_status st = guard([&] {
    if (!dir) fail(XDTTS_ERR_BAD_ARG, "dir is null");
    device_id = select_device(device_id);
    load_model_dir(dir, blob);
  });
  if (st != XDTTS_OK) return st;
  return make_handle(std::move(blob), device_id, out);
}

int32_t xdtts_default_device(void) {
  int32_t v = 0;
  return guard([&] { v = default_device(); }) == XDTTS_OK ? v : -1;
}

xdtts_status xdtts_model_dir_read(const char *dir, float *blob, size_t n_floats) {
  return guard([&] {
    if (!dir || !blob) fail(XDTTS_ERR_BAD_ARG, "null argument");
    if (n_floats != tensor_total()) fail(XDTTS_ERR_BAD_ARG, "blob has %zu floats, expected %zu", n_floats, tensor_total());
    std::vector<float> v;
    load_model_dir(dir, v);
    std::memcpy(blob, v.data(), v.size() * sizeof(float));
  });
}

xdtts_status xdtts_model_dir_describe(const char *dir, char *buf, size_t cap, size_t *needed) {
  return guard([&] {
    if (!dir || (!buf && cap)) fail(XDTTS_ERR_BAD_ARG, "null argument");
    const std::string d = describe_onnx_dir(dir);
    if (needed) *needed = d.size() + 1;
    if (cap) {
      const size_t n = std::min(cap - 1, d.size());
      std::memcpy(buf, d.data(), n);
      buf[n] = 0;
    }
  });
}

xdtts_status xdtts_tacotron2_load_synthetic(uint32_t seed, float rec_scale, int32_t device_id,
                                            xdtts_tacotron2 **out) {
  std::vector<float> blob;
  xdtts_status st = guard([&] {
    device_id = select_device(device_id);
    synthetic_blob(seed, rec_scale, blob);
  });
  if (st != XDTTS_OK) return st;
  return make_handle(std::move(blob), device_id, out);
}

xdtts_status xdtts_tacotron2_load_blob(const float *blob, size_t n_floats, int32_t device_id,
                                       xdtts_tacotron2 **out) {
  std::vector<float> v;
  xdtts_status st = guard([&] {
    if (!blob) fail(XDTTS_ERR_BAD_ARG, "blob is null");
    if (n_floats != tensor_total()) fail(XDTTS_ERR_BAD_ARG, "blob has %zu floats, expected %zu", n_floats, tensor_total());
    device_id = select_device(device_id);
    v.assign(blob, blob + n_floats);
  });
  if (st != XDTTS_OK) return st;
  return make_handle(std::move(v), device_id, out);
}

xdtts_status xdtts_tacotron2_save(const xdtts_tacotron2 *h, const char *dir) {
  return guard([&] {
    if (!h || !dir) fail(XDTTS_ERR_BAD_ARG, "null argument");
    save_container(dir, h->blob);
  });
}

xdtts_status xdtts_tacotron2_get_tensor(const xdtts_tacotron2 *h, int32_t i, float *out) {
  return guard([&] {
    if (!h || !out || i < 0 || i >= xdtts_tensor_count()) fail(XDTTS_ERR_BAD_ARG, "bad tensor request");
    const TensorInfo &t = tensor_table()[i];
    std::memcpy(out, h->blob.data() + t.offset, t.numel * sizeof(float));
  });
}

void xdtts_tacotron2_free(xdtts_tacotron2 *h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  delete h;
}

xdtts_status xdtts_tacotron2_sync(xdtts_tacotron2 *h) {
  return guard([&] {
    if (!h) fail(XDTTS_ERR_BAD_ARG, "null handle");
    HIP_CHECK(hipStreamSynchronize(h->stream));
  });
}

static xdtts_infer_opts resolve_opts(const xdtts_infer_opts *opts) {
  xdtts_infer_opts o;
  xdtts_infer_opts_default(&o);
  if (opts) o = *opts;
  if (o.max_chunk <= 0) o.max_chunk = 100;
  return o;
}

xdtts_status xdtts_tacotron2_infer_ids(xdtts_tacotron2 *h, const int64_t *ids, size_t n, const size_t *splits,
                                       size_t n_splits, const xdtts_infer_opts *opts, float **mel,
                                       size_t *n_frames) {
  return guard([&] {
    if (!h || !mel || !n_frames) fail(XDTTS_ERR_BAD_ARG, "null argument");
    *mel = nullptr;
    *n_frames = 0;
    std::lock_guard<std::mutex> lk(h->mu);
    const xdtts_infer_opts o = resolve_opts(opts);
    std::vector<int64_t> padded;
    std::vector<int> lens;
    chunks_from_splits(ids, n, splits, n_splits, o.max_chunk, padded, lens);
    int total = 0;
    h->infer_batch_device(padded.data(), lens.data(), (int)lens.size(), o.max_chunk, o, nullptr, &total);
    PinnedGuard host((size_t)N_MEL * total);
    HIP_CHECK(hipMemcpyAsync(host.p, h->mel_dev.p, (size_t)N_MEL * total * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    h->finish_timings();
    *mel = host.release();
    *n_frames = (size_t)total;
  });
}

xdtts_status xdtts_tacotron2_infer_batch(xdtts_tacotron2 *h, const int64_t *ids, const int32_t *lens, int32_t B,
                                         int32_t t_stride, const xdtts_infer_opts *opts,
                                         const int32_t *fixed_steps_per_item, float **mels, size_t *n_frames) {
  return guard([&] {
    if (!h || !ids || !lens || !mels || !n_frames) fail(XDTTS_ERR_BAD_ARG, "null argument");
    std::lock_guard<std::mutex> lk(h->mu);
    const xdtts_infer_opts o = resolve_opts(opts);
    if (B <= 0) fail(XDTTS_ERR_BAD_ARG, "batch %d out of range", B);
    for (int b = 0; b < B; ++b) mels[b] = nullptr;
    const int T = o.max_chunk;
    if (t_stride <= 0) fail(XDTTS_ERR_BAD_ARG, "t_stride must be positive");
    std::vector<int64_t> padded((size_t)B * T, 0);
    for (int b = 0; b < B; ++b) {
      if (lens[b] > T) fail(XDTTS_ERR_TOO_LONG, "chunk %d has %d ids, window is %d", b, lens[b], T);
      if (lens[b] > t_stride || lens[b] <= 0) fail(XDTTS_ERR_BAD_ARG, "chunk %d: bad length %d", b, lens[b]);
      std::copy(ids + (size_t)b * t_stride, ids + (size_t)b * t_stride + lens[b], padded.begin() + (size_t)b * T);
    }
    int total = 0;
    // the post-net leaves one dense (80 x F_b) matrix per chunk, back to back: ONE copy into a pinned slab whose pieces
    // are the buffers the caller receives (the (80 x F_total) layout took 4 160 strided row copies on the host for the
    // 52-chunk batch -- as long as the post-net on one thread, 0.3 ms on four; 52 pitched copies from the device 0.9 ms)
    std::vector<int> F = h->infer_batch_device(padded.data(), lens, B, T, o, fixed_steps_per_item, &total, true);
    PinnedSlab slab((size_t)N_MEL * total);
    struct Drain {  // the slab does not go back to the pool with the copy in flight
      hipStream_t s;
      ~Drain() { (void)hipStreamSynchronize(s); }
    } drain{h->stream};
    HIP_CHECK(hipMemcpyAsync(slab.base, h->mel_dev.p, (size_t)N_MEL * total * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    h->finish_timings();
    size_t off = 0;
    for (int b = 0; b < B; ++b) {
      mels[b] = slab.piece(off);
      n_frames[b] = (size_t)F[b];
      off += (size_t)N_MEL * F[b];
    }
    slab.hand_over();
  });
}

xdtts_status xdtts_tacotron2_encoder(xdtts_tacotron2 *h, const int64_t *ids, int32_t T, float *memory,
                                     float *processed_memory) {
  return guard([&] {
    if (!h || !ids || !memory || !processed_memory) fail(XDTTS_ERR_BAD_ARG, "null argument");
    if (T <= 0 || T > T_MAX) fail(XDTTS_ERR_BAD_ARG, "T %d out of range", T);
    for (int t = 0; t < T; ++t)
      if (ids[t] < 0 || ids[t] >= N_SYMBOLS) fail(XDTTS_ERR_BAD_ARG, "id out of range");
    std::lock_guard<std::mutex> lk(h->mu);
    HIP_CHECK(hipSetDevice(h->device));
    h->ids.upload(ids, T, h->stream);
    std::lock_guard<ChipLock> chip(chip_mutex(h->device));
    h->run_encoder(1, T);
    HIP_CHECK(hipMemcpyAsync(memory, h->memory.p, (size_t)T * EMB * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_CHECK(hipMemcpyAsync(processed_memory, h->pmem.p, (size_t)T * ATT_DIM * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_CHECK(hipStreamSynchronize(h->stream));
    h->check_encoder_exchange();
  });
}

xdtts_status xdtts_tacotron2_decoder(xdtts_tacotron2 *h, const float *memory, const float *processed_memory,
                                     int32_t T, int32_t n_valid, const xdtts_infer_opts *opts, float *frames,
                                     float *gates, size_t *n_frames) {
  return guard([&] {
    if (!h || !memory || !processed_memory || !frames || !n_frames) fail(XDTTS_ERR_BAD_ARG, "null argument");
    if (T <= 0 || T > T_MAX || n_valid <= 0 || n_valid > T) fail(XDTTS_ERR_BAD_ARG, "bad T/n_valid");
    std::lock_guard<std::mutex> lk(h->mu);
    HIP_CHECK(hipSetDevice(h->device));
    const xdtts_infer_opts o = resolve_opts(opts);
    h->memory.upload(memory, (size_t)T * EMB, h->stream);
    h->pmem.upload(processed_memory, (size_t)T * ATT_DIM, h->stream);
    int nv = n_valid;
    h->n_valid.upload(&nv, 1, h->stream);
    HIP_CHECK(hipStreamSynchronize(h->stream));
    std::vector<int> lim(1, std::min(o.fixed_steps > 0 ? o.fixed_steps : o.max_steps, o.max_steps));
    h->upload_dropout_masks(o, 1, lim.data());
    DecoderBufs d = h->decoder_bufs(1, T, h->memory.p, h->pmem.p, o);
    HIP_CHECK(hipEventRecord(h->ev.e[0], h->stream));
    HIP_CHECK(hipEventRecord(h->ev.e[1], h->stream));
    h->last_steps = h->run_decoder(d, lim);
    HIP_CHECK(hipEventRecord(h->ev.e[2], h->stream));
    HIP_CHECK(hipEventRecord(h->ev.e[3], h->stream));
    const int F = h->host_ctl[xdtts_tacotron2::HOST_NF];
    HIP_CHECK(hipMemcpyAsync(frames, d.frames, (size_t)F * N_MEL * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    if (gates) HIP_CHECK(hipMemcpyAsync(gates, d.gates, (size_t)F * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    h->finish_timings();
    *n_frames = (size_t)F;
  });
}

// Parity hook: n_steps consecutive decoder_iter.onnx calls (mod.rs:304; each call's outputs fed back as mod.rs:328-341 does)
// for B chunks from caller-held state, through the frame-loop engine the caller names.
xdtts_status xdtts_tacotron2_decoder_steps(xdtts_tacotron2 *h, int32_t engine, int32_t B, const float *memory, const float *processed_memory,
                                           int32_t T, const int32_t *n_valid, const xdtts_infer_opts *opts, uint32_t step0, int32_t n_steps,
                                           const float *decoder_input, float *attention_hidden, float *attention_cell, float *decoder_hidden,
                                           float *decoder_cell, float *attention_weights, float *attention_weights_cum, float *attention_context,
                                           float *decoder_output, float *gate_prediction) {
  return guard([&] {
    if (!h || !memory || !processed_memory || !n_valid || !decoder_input || !attention_hidden || !attention_cell || !decoder_hidden ||
        !decoder_cell || !attention_weights || !attention_weights_cum || !attention_context || !decoder_output || !gate_prediction)
      fail(XDTTS_ERR_BAD_ARG, "null argument");
    if (engine < 0 || engine > 3)
      fail(XDTTS_ERR_BAD_ARG, "engine %d out of range (0 launch-per-stage, 1 persistent, 2 batched MFMA, 3 persistent MFMA)", engine);
    if (engine == 3 && (B > P8_B_MAX || T > PERSIST_T_MAX))
      fail(XDTTS_ERR_BAD_ARG, "the persistent MFMA engine takes at most %d chunks of at most %d encoder steps", P8_B_MAX, PERSIST_T_MAX);
    if (B <= 0 || B > 64) fail(XDTTS_ERR_BAD_ARG, "batch %d out of range (1..64)", B);
    if (T <= 0 || T > T_MAX) fail(XDTTS_ERR_BAD_ARG, "T %d out of range", T);
    if (n_steps <= 0 || n_steps > 100000 || (uint64_t)step0 + (uint64_t)n_steps > (1u << 30)) fail(XDTTS_ERR_BAD_ARG, "bad step range");
    for (int b = 0; b < B; ++b)
      if (n_valid[b] <= 0 || n_valid[b] > T) fail(XDTTS_ERR_BAD_ARG, "chunk %d: bad n_valid %d", b, n_valid[b]);
    if (engine == 1 && (B > PERSIST_B_MAX || T > PERSIST_T_MAX))
      fail(XDTTS_ERR_BAD_ARG, "the persistent engine takes at most %d chunks of at most %d encoder steps", PERSIST_B_MAX, PERSIST_T_MAX);
    if (engine == 1) {
      // The persistent engine folds the context columns of its weights into the encoder memory and works from the attention
      // WEIGHTS (decoder_persistent.hip), so the incoming context must be the one those weights give -- true for every state
      // the graph itself produced (out_attention_context = out_attention_weights . memory, fed back at mod.rs:332-339).
      for (int b = 0; b < B; ++b)
        for (int j = 0; j < EMB; ++j) {
          double c = 0;
          for (int t = 0; t < T; ++t) c += (double)attention_weights[(size_t)b * T + t] * memory[((size_t)b * T + t) * EMB + j];
          if (std::fabs(c - attention_context[(size_t)b * EMB + j]) > 1e-4 + 1e-4 * std::fabs(c))
            fail(XDTTS_ERR_BAD_ARG, "engine 1 needs attention_context = attention_weights . memory (chunk %d, column %d: %g vs %g)", b, j,
                 (double)attention_context[(size_t)b * EMB + j], c);
        }
    }
    std::lock_guard<std::mutex> lk(h->mu);
    HIP_CHECK(hipSetDevice(h->device));
    xdtts_infer_opts o = resolve_opts(opts);
    const int end = (int)step0 + n_steps;
    o.max_steps = end + 1;
    hipStream_t st = h->stream;
    h->memory.upload(memory, (size_t)B * T * EMB, st);
    h->pmem.upload(processed_memory, (size_t)B * T * ATT_DIM, st);
    h->n_valid.upload(n_valid, B, st);
    std::vector<int> lim((size_t)B, end);
    h->upload_dropout_masks(o, B, lim.data());
    if (engine == 2) h->w.ensure_batched_layout(h->blob, st);
    DecoderBufs d = h->decoder_bufs(B, T, h->memory.p, h->pmem.p, o, engine == 2 ? 1 : 0);
    d.use_gate = 0;  // the caller applies the stop rule to gate_prediction (mod.rs:319)
    h->limits.upload(lim.data(), lim.size(), st);
    h->lim_on_dev = lim;
    launch_decoder_init(d, h->limits.p, st);
    h->dec_in_dev.upload(decoder_input, (size_t)B * N_MEL, st);
    // staging of the seven state tensors in the caller's row-major layout
    const size_t nh = (size_t)B * ATT_RNN, nt = (size_t)B * T, nc = (size_t)B * EMB;
    h->state_stage.alloc(4 * nh + 2 * nt + nc);
    float *s_ah = h->state_stage.p, *s_ac = s_ah + nh, *s_dh = s_ac + nh, *s_dc = s_dh + nh, *s_aw = s_dc + nh, *s_awc = s_aw + nt, *s_ctx = s_awc + nt;
    auto up = [&](float *dst, const float *src, size_t n) { HIP_CHECK(hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyHostToDevice, st)); };
    auto dd = [&](float *dst, const float *src, size_t n) { HIP_CHECK(hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, st)); };
    up(s_ah, attention_hidden, nh);
    up(s_ac, attention_cell, nh);
    up(s_dh, decoder_hidden, nh);
    up(s_dc, decoder_cell, nh);
    up(s_aw, attention_weights, nt);
    up(s_awc, attention_weights_cum, nt);
    up(s_ctx, attention_context, nc);
    const int s0 = (int)step0;
    HIP_CHECK(hipMemcpyAsync(d.ctl, &s0, sizeof(int), hipMemcpyHostToDevice, st));
    HIP_CHECK(hipStreamSynchronize(st));  // the sources above are caller memory / locals
    dd(d.aw, s_aw, nt);
    dd(d.awc, s_awc, nt);
    dd(d.ctx, s_ctx, nc);
    if (engine == 2) {  // the batched kernels keep h, c and the context in MFMA-operand order
      launch_frag_convert(s_ah, d.att_hf[0], B, d.Bpad, ATT_RNN, 0, st);
      launch_frag_convert(s_dh, d.dec_hf[0], B, d.Bpad, DEC_RNN, 0, st);
      launch_frag_convert(s_ac, d.att_c, B, d.Bpad, ATT_RNN, 0, st);
      launch_frag_convert(s_dc, d.dec_c, B, d.Bpad, DEC_RNN, 0, st);
      launch_frag_convert(s_ctx, d.ctxf, B, d.Bpad, EMB, 0, st);
    } else {
      dd(d.att_h[0], s_ah, nh);
      dd(d.att_c, s_ac, nh);
      dd(d.dec_h[0], s_dh, nh);
      dd(d.dec_c, s_dc, nh);
    }
    d.dec_in = h->dec_in_dev.p;
    int fin = 0;  // ping-pong half that holds the final hidden states
    if (engine == 1) {
      if (!decoder_persistent_supported(h->device, PERSIST_B_MAX, PERSIST_T_MAX))
        fail(XDTTS_ERR_HIP, "persistent engine not available on this device (its 256-workgroup grid cannot be co-resident)");
      std::lock_guard<ChipLock> chip(chip_mutex(h->device));
      launch_decoder_prenet(d, h->w, st);  // x(step0) = prenet(decoder_input): the persistent kernel's own prenet produces x(s + 1)
      d.dec_in = nullptr;
      h->dec_exchange.alloc(persist_granule_words(B));
      PersistBufs g = persist_bufs(h->dec_exchange.p, h->dec_err.p, B);
      launch_persist_seed_at(d, g, h->limits.p, s0, st);
      try {
        launch_decoder_persistent(d, h->w, g, n_steps, st);
      } catch (const CoopRefused &) {
        (void)hipStreamSynchronize(st);
        fail(XDTTS_ERR_HIP, "persistent engine not available on this device (cooperative launch refused)");
      }
      int e = 0;
      HIP_CHECK(hipMemcpyAsync(&e, h->dec_err.p, sizeof(int), hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipStreamSynchronize(st));
      if (e) {
        HIP_CHECK(hipMemsetAsync(h->dec_err.p, 0, sizeof(int), st));
        fail(XDTTS_ERR_HIP, "persistent decoder exchange timed out (grid not co-resident)");
      }
    } else if (engine == 3) {
      if (!decoder_p8_supported(h->device, B, PERSIST_T_MAX))
        fail(XDTTS_ERR_HIP, "persistent MFMA engine not available on this device (its 256-workgroup grid cannot be co-resident)");
      std::lock_guard<ChipLock> chip(chip_mutex(h->device));
      launch_decoder_prenet(d, h->w, st);  // x(step0) = prenet(decoder_input)
      d.dec_in = nullptr;
      h->dec_exchange.alloc(p8_exchange_words(B, n_steps));
      P8Bufs g = p8_bufs(h->dec_exchange.p, h->dec_err.p, B, n_steps);
      launch_p8_seed_at(d, g, h->limits.p, s0, st);
      try {
        launch_decoder_p8(d, h->w, g, n_steps, st);
      } catch (const CoopRefused &) {
        (void)hipStreamSynchronize(st);
        fail(XDTTS_ERR_HIP, "persistent MFMA engine not available on this device (cooperative launch refused)");
      }
      int e = 0;
      HIP_CHECK(hipMemcpyAsync(&e, h->dec_err.p, sizeof(int), hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipStreamSynchronize(st));
      if (e) {
        HIP_CHECK(hipMemsetAsync(h->dec_err.p, 0, sizeof(int), st));
        fail(XDTTS_ERR_HIP, "persistent MFMA decoder exchange timed out (grid not co-resident)");
      }
    } else {
      std::unique_lock<ChipLock> chip;
      if (d.hg) chip = std::unique_lock<ChipLock>(chip_mutex(h->device));
      if (engine == 0) launch_decoder_location(d, h->w, st);  // (the batched prenet launch computes them itself)
      if (d.hring) HIP_CHECK(hipMemsetAsync(d.hring, 0xff, (size_t)d.hring_steps * ATT_RNN * d.Bpad * sizeof(unsigned), st));
      if (d.hcnt) {
        HIP_CHECK(hipMemsetAsync(d.hstage, 0xff, (size_t)16 * ATT_RNN * d.Bpad * sizeof(unsigned), st));
        HIP_CHECK(hipMemsetAsync(d.hcnt, 0, (size_t)d.hring_steps * 8 * 64 * sizeof(unsigned), st));
      }
      launch_decoder_early(d, h->w, 0, st);  // (batched engine: the first attention-LSTM pass's early partial, from the imported state)
      launch_decoder_prologue(d, h->w, st);  // (two-launch form: x and location features of the first step; d.dec_in = decoder_input)
      for (int i = 0; i < n_steps; ++i) {
        launch_decoder_step_at(d, h->w, i, st);
        d.dec_in = nullptr;  // from the second step on the loop feeds itself
      }
      launch_decoder_advance(d, n_steps, st);
      launch_decoder_flush(d, h->w, st);  // decoder_output and gate_prediction of the last step
      fin = n_steps & 1;
      if (d.ep_g) {
        int e = 0;
        HIP_CHECK(hipMemcpyAsync(&e, h->dec_err.p, sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        if (e) {
          HIP_CHECK(hipMemsetAsync(h->dec_err.p, 0, sizeof(int), st));
          fail(XDTTS_ERR_HIP, "batched attention exchange timed out (grid not co-resident)");
        }
      }
    }
    if (engine == 2) {
      launch_frag_convert(s_ah, d.att_hf[fin], B, d.Bpad, ATT_RNN, 1, st);
      launch_frag_convert(s_dh, d.dec_hf[fin], B, d.Bpad, DEC_RNN, 1, st);
      launch_frag_convert(s_ac, d.att_c, B, d.Bpad, ATT_RNN, 1, st);
      launch_frag_convert(s_dc, d.dec_c, B, d.Bpad, DEC_RNN, 1, st);
    }
    auto down = [&](float *dst, const float *src, size_t n) { HIP_CHECK(hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToHost, st)); };
    down(attention_hidden, engine == 2 ? s_ah : d.att_h[fin], nh);
    down(attention_cell, engine == 2 ? s_ac : d.att_c, nh);
    down(decoder_hidden, engine == 2 ? s_dh : d.dec_h[fin], nh);
    down(decoder_cell, engine == 2 ? s_dc : d.dec_c, nh);
    down(attention_weights, d.aw, nt);
    down(attention_weights_cum, engine == 2 && (n_steps & 1) ? d.awc2 : d.awc, nt);  // (batched: ping-pong by step parity)
    down(attention_context, d.ctx, nc);
    for (int b = 0; b < B; ++b) {
      down(decoder_output + (size_t)b * n_steps * N_MEL, d.frames + ((size_t)b * d.max_steps + step0) * N_MEL, (size_t)n_steps * N_MEL);
      down(gate_prediction + (size_t)b * n_steps, d.gates + (size_t)b * d.max_steps + step0, (size_t)n_steps);
    }
    HIP_CHECK(hipStreamSynchronize(st));
  });
}

// Parity hook: ONE decoder_iter.onnx call (mod.rs:304) from caller-held state, on the launch-per-stage kernels.
xdtts_status xdtts_tacotron2_decoder_step(xdtts_tacotron2 *h, const float *memory, const float *processed_memory, int32_t T,
                                          int32_t n_valid, const xdtts_infer_opts *opts, uint32_t step, const float *decoder_input,
                                          float *attention_hidden, float *attention_cell, float *decoder_hidden, float *decoder_cell,
                                          float *attention_weights, float *attention_weights_cum, float *attention_context,
                                          float *decoder_output, float *gate_prediction) {
  return xdtts_tacotron2_decoder_steps(h, 0, 1, memory, processed_memory, T, &n_valid, opts, step, 1, decoder_input, attention_hidden,
                                       attention_cell, decoder_hidden, decoder_cell, attention_weights, attention_weights_cum,
                                       attention_context, decoder_output, gate_prediction);
}

// Which engines this handle currently uses (1 = the persistent / cooperative one, 0 = demoted to the
// launch-per-stage / single-workgroup one after a timed-out exchange, -1 = not probed yet).
xdtts_status xdtts_tacotron2_engine_state(const xdtts_tacotron2 *h, int32_t *decoder_persistent, int32_t *encoder_cooperative,
                                          int32_t *batched_attention) {
  return guard([&] {
    if (!h) fail(XDTTS_ERR_BAD_ARG, "null handle");
    std::lock_guard<std::mutex> lk(h->mu);
    if (decoder_persistent) *decoder_persistent = h->persist_state;
    if (encoder_cooperative) *encoder_cooperative = h->coop_ok ? 1 : 0;
    if (batched_attention) *batched_attention = h->att_fused;
  });
}

xdtts_status xdtts_tacotron2_small_batch_engine_state(const xdtts_tacotron2 *h, int32_t *state) {
  return guard([&] {
    if (!h || !state) fail(XDTTS_ERR_BAD_ARG, "null argument");
    std::lock_guard<std::mutex> lk(h->mu);
    *state = h->p8_state;
  });
}

// Puts a demoted handle back on the fast engines (they are probed again on the next call).
xdtts_status xdtts_tacotron2_engine_reset(xdtts_tacotron2 *h) {
  return guard([&] {
    if (!h) fail(XDTTS_ERR_BAD_ARG, "null handle");
    std::lock_guard<std::mutex> lk(h->mu);
    // (a launch the runtime REFUSED is a property of the device, not a transient: those engines stay off)
    if (!h->persist_refused) h->persist_state = -1;
    if (!h->p8_refused) h->p8_state = -1;
    if (!h->coop_refused) h->coop_ok = true;
    h->demoted_calls = 0;
    h->p8_demoted_calls = 0;
    h->enc_demoted_calls = 0;
    h->att_fused = xdtts_tacotron2::att_fused_default();
    h->att_demoted = false;
  });
}

// Measurement aid (bench.py's roofline.latency_floor_us): the five dependent all-gather exchanges of one persistent-decoder
// step with no arithmetic between them, timed on THIS device now (csrc/edge_floor.hip).
xdtts_status xdtts_edge_floor_us(int32_t device_id, int32_t steps, int32_t T, int32_t tuned, double *us_per_step) {
  return guard([&] {
    if (!us_per_step || steps < 1 || steps > 1000000 || T < 1 || T > 128) fail(XDTTS_ERR_BAD_ARG, "bad argument");
    device_id = select_device(device_id);
    std::lock_guard<ChipLock> chip(chip_mutex(device_id));  // its grid must be co-resident, like the engine's
    const double us = xdtts_edge_floor::measure(device_id, steps, T, xdtts_edge_floor::kernel_delays(tuned ? 1 : 0), 5, false);
    if (us < 0) fail(XDTTS_ERR_HIP, "edge-floor skeleton: grid not co-resident on this device, or an exchange failed");
    *us_per_step = us;
  });
}

xdtts_status xdtts_tacotron2_postnet(xdtts_tacotron2 *h, const float *frames, int32_t F, float *mel_out) {
  return guard([&] {
    if (!h || !frames || !mel_out || F <= 0) fail(XDTTS_ERR_BAD_ARG, "bad argument");
    std::lock_guard<std::mutex> lk(h->mu);
    HIP_CHECK(hipSetDevice(h->device));
    h->frames.upload(frames, (size_t)F * N_MEL, h->stream);
    HIP_CHECK(hipStreamSynchronize(h->stream));
    h->mel_dev.alloc((size_t)N_MEL * F);
    const long zero = 0;
    h->run_postnet(h->frames.p, 0, &F, &zero, 1, h->mel_dev.p, F);
    HIP_CHECK(hipMemcpyAsync(mel_out, h->mel_dev.p, (size_t)N_MEL * F * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_CHECK(hipStreamSynchronize(h->stream));
  });
}

xdtts_status xdtts_tacotron2_last_timings(const xdtts_tacotron2 *h, float ms[4], int32_t *steps) {
  return guard([&] {
    if (!h || !ms) fail(XDTTS_ERR_BAD_ARG, "null argument");
    for (int i = 0; i < 4; ++i) ms[i] = h->last_ms[i];
    if (steps) *steps = h->last_steps;
  });
}

// ---- Griffin-Lim ---------------------------------------------------------------------------------

xdtts_status xdtts_mel_filter_bank(float sample_rate, size_t n_fft, size_t n_mels, float fmin, float fmax_or_nan,
                                   float *out) {
  return guard([&] {
    if (!out || n_fft < 2 || n_mels == 0 || !(sample_rate > 0)) fail(XDTTS_ERR_BAD_ARG, "bad filter bank request");
    const double sr = sample_rate;
    const double fmax = std::isnan(fmax_or_nan) ? sr / 2.0 : (double)fmax_or_nan;  // Option<f32>::None
    const int nb = (int)(n_fft / 2 + 1), nm = (int)n_mels;
    std::vector<double> mel_f(nm + 2);
    const double m0 = hz_to_mel(fmin), m1 = hz_to_mel(fmax);
    for (int i = 0; i < nm + 2; ++i) mel_f[i] = mel_to_hz(m0 + (m1 - m0) * i / (nm + 1));
    for (int i = 0; i < nm; ++i) {
      const double enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
      for (int b = 0; b < nb; ++b) {
        const double f = (sr / 2.0) * b / (nb - 1);
        const double lower = (f - mel_f[i]) / (mel_f[i + 1] - mel_f[i]);
        const double upper = (mel_f[i + 2] - f) / (mel_f[i + 2] - mel_f[i + 1]);
        const double wv = std::min(lower, upper);
        out[(size_t)i * nb + b] = (float)((wv > 0 ? wv : 0) * enorm);
      }
    }
  });
}

xdtts_status xdtts_griffinlim_new(const float *mel_basis, size_t n_mels, size_t n_bins, size_t noverlap, float power,
                                  size_t iters, float momentum, int32_t device_id, xdtts_griffinlim **out) {
  return guard([&] {
    if (!out) fail(XDTTS_ERR_BAD_ARG, "out handle pointer is null");
    *out = nullptr;
    if (!mel_basis || n_mels == 0 || n_bins < 2) fail(XDTTS_ERR_BAD_ARG, "bad mel basis");
    const size_t n_fft = 2 * (n_bins - 1);
    if (n_fft != 1024) fail(XDTTS_ERR_BAD_ARG, "n_fft %zu unsupported: the framed-FFT kernel is built for 1024", n_fft);
    if (noverlap >= n_fft) fail(XDTTS_ERR_BAD_ARG, "noverlap %zu must be < n_fft %zu", noverlap, n_fft);
    if (n_fft - noverlap != n_fft / 4)
      fail(XDTTS_ERR_BAD_ARG, "hop %zu unsupported: the framed-FFT kernels are built for hop = n_fft/4 = 256 (mod.rs:456)", n_fft - noverlap);
    if (n_mels % 16 != 0) fail(XDTTS_ERR_BAD_ARG, "n_mels %zu must be a multiple of 16", n_mels);
    if (!(power > 0) || momentum < 0) fail(XDTTS_ERR_BAD_ARG, "bad power/momentum");
    device_id = select_device(device_id);
    auto g = std::make_unique<xdtts_griffinlim>();
    g->device = device_id;
    g->n_mels = (int)n_mels;
    g->nb = (int)n_bins;
    g->n_fft = (int)n_fft;
    g->hop = (int)(n_fft - noverlap);
    g->iters = (int)iters;
    g->power = power;
    g->momentum = momentum;
    HIP_CHECK(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
    g->ev.create();
    std::vector<float> pinv;
    host_pinv(mel_basis, (int)n_mels, (int)n_bins, pinv);
    g->pinv.upload(pinv.data(), pinv.size(), g->stream);
    {  // NNLS refinement operands: the basis and its transpose, bins zero-padded to NBP, and the step 1/L
      const int NBP = xdtts_griffinlim::NBP, nm = (int)n_mels, nbi = (int)n_bins;
      std::vector<float> bp((size_t)nm * NBP, 0.f), bt((size_t)NBP * nm, 0.f);
      for (int i = 0; i < nm; ++i)
        for (int b = 0; b < nbi; ++b) bp[(size_t)i * NBP + b] = bt[(size_t)b * nm + i] = mel_basis[(size_t)i * nbi + b];
      g->basis_p.upload(bp.data(), bp.size(), g->stream);
      g->basisT_p.upload(bt.data(), bt.size(), g->stream);
      g->nnls_step = (float)(1.0 / host_lipschitz(mel_basis, nm, nbi));
      g->norm_parts.alloc(GLN_SCRATCH);
    }
    std::vector<float2> tw(n_fft);
    std::vector<float> win(n_fft);
    const double PI = 3.14159265358979323846;
    for (size_t k = 0; k < n_fft; ++k) {
      tw[k] = make_float2((float)std::cos(2.0 * PI * k / n_fft), (float)(-std::sin(2.0 * PI * k / n_fft)));
      win[k] = (float)(0.5 - 0.5 * std::cos(2.0 * PI * k / n_fft));  // periodic hann
    }
    g->tw.upload(tw.data(), tw.size(), g->stream);
    g->win.upload(win.data(), win.size(), g->stream);
    HIP_CHECK(hipStreamSynchronize(g->stream));
    *out = g.release();
  });
}

void xdtts_griffinlim_opts_default(xdtts_griffinlim_opts *o) {
  if (!o) return;
  o->nnls_iters = 0;
  o->power_mode = 0;
  o->mel_decompress = 0;
  o->output_normalise = 3;  // rms, never past +-1: the level of the reference's own WAV_SPEC files
  o->batch_shape = 0;
  o->rms_target = 0.1f;
}

xdtts_status xdtts_griffinlim_set_opts(xdtts_griffinlim *g, const xdtts_griffinlim_opts *o) {
  return guard([&] {
    if (!g || !o) fail(XDTTS_ERR_BAD_ARG, "null argument");
    if (o->nnls_iters < 0 || o->nnls_iters > 100000 || o->power_mode < 0 || o->power_mode > 2 || o->mel_decompress < 0 ||
        o->mel_decompress > 2 || o->output_normalise < 0 || o->output_normalise > 3 || !(o->rms_target > 0.f) || !(o->rms_target <= 1e6f) || (o->batch_shape != 0 && o->batch_shape != 4))
      fail(XDTTS_ERR_BAD_ARG, "griffin-lim option out of range");
    std::lock_guard<std::mutex> lk(g->mu);
    g->gopts = *o;
  });
}

xdtts_status xdtts_griffinlim_get_opts(const xdtts_griffinlim *g, xdtts_griffinlim_opts *o) {
  return guard([&] {
    if (!g || !o) fail(XDTTS_ERR_BAD_ARG, "null argument");
    std::lock_guard<std::mutex> lk(g->mu);
    *o = g->gopts;
  });
}

xdtts_status xdtts_griffinlim_set_seed(xdtts_griffinlim *g, uint32_t seed) {
  return guard([&] {
    if (!g) fail(XDTTS_ERR_BAD_ARG, "null handle");
    std::lock_guard<std::mutex> lk(g->mu);
    g->seed = seed;
  });
}

// Phase init + iterations + final ISTFT on the S in place, then the audio to a pinned host buffer.
// The persistent engine needs its grid co-resident: the chip lock is held from the launch until the
// stream has drained; a timed-out exchange demotes the handle and the request runs again on the
// launch-per-iteration engine (S and the phase seed are intact).
static void gl_iterate_and_fetch(xdtts_griffinlim *g, const GlBufs &b, const float *phase0_dev, int iters, float **audio,
                                 size_t *n_samples, bool normalise) {
  const size_t N = (size_t)g->hop * (size_t)(b.F - 1);
  std::lock_guard<ChipLock> chip(chip_mutex(g->device));
  g->probe_tick();
  for (int attempt = 0; attempt < 2; ++attempt) {
    g->iterate(b, phase0_dev, iters);
    if (normalise) launch_gl_output_normalise(g->audio.p, nullptr, 1, 0, (int)N, g->gopts.output_normalise, g->gopts.rms_target, g->norm_parts.p, g->stream);
    HIP_CHECK(hipEventRecord(g->ev.e[2], g->stream));
    PinnedGuard host(N);
    HIP_CHECK(hipMemcpyAsync(host.p, g->audio.p, N * sizeof(float), hipMemcpyDeviceToHost, g->stream));
    g->fetch_error_word();
    g->finish_timings();  // (drains the stream)
    if (g->persistent_failed()) {
      HIP_CHECK(hipEventRecord(g->ev.e[1], g->stream));  // time the run that counts
      continue;
    }
    *audio = host.release();
    *n_samples = N;
    return;
  }
  fail(XDTTS_ERR_HIP, "Griffin-Lim: the fallback engine reported an exchange failure");
}

// The two halves of gl_run_from_device_mel for a caller that overlaps the vocoder with other work (xdtts_synthesize_sequence):
// everything enqueued on g->stream, nothing waited for; then the wait, the engine's error word and -- after a timed-out
// exchange -- the request again on the fallback engine (S is intact until the next enqueue).  Caller holds g->mu and the chip lock.
static void gl_enqueue_from_device_mel(xdtts_griffinlim *g, const float *mel_dev_ptr, int F, PinnedGuard &host) {
  GlBufs b = g->bufs(F);
  const size_t N = (size_t)g->hop * (size_t)(F - 1);
  HIP_CHECK(hipEventRecord(g->ev.e[0], g->stream));
  g->mel_to_linear(mel_dev_ptr, F);
  HIP_CHECK(hipEventRecord(g->ev.e[1], g->stream));
  g->probe_tick();
  g->iterate(b, nullptr, g->iters);
  launch_gl_output_normalise(g->audio.p, nullptr, 1, 0, (int)N, g->gopts.output_normalise, g->gopts.rms_target, g->norm_parts.p, g->stream);
  HIP_CHECK(hipEventRecord(g->ev.e[2], g->stream));
  if (!host.p) host = PinnedGuard(N);  // (the sequence hands in one it took from the pool while the frame loop ran)
  HIP_CHECK(hipMemcpyAsync(host.p, g->audio.p, N * sizeof(float), hipMemcpyDeviceToHost, g->stream));
  g->fetch_error_word();
}
static void gl_collect(xdtts_griffinlim *g, int F, PinnedGuard &host, float **audio, size_t *n_samples) {
  g->finish_timings();  // (drains the stream)
  if (g->persistent_failed()) {
    host = PinnedGuard();
    gl_iterate_and_fetch(g, g->bufs(F), nullptr, g->iters, audio, n_samples, true);
    return;
  }
  *audio = host.release();
  *n_samples = (size_t)g->hop * (size_t)(F - 1);
}

static void gl_run_from_device_mel(xdtts_griffinlim *g, const float *mel_dev_ptr, int F, float **audio, size_t *n_samples) {
  GlBufs b = g->bufs(F);
  HIP_CHECK(hipEventRecord(g->ev.e[0], g->stream));
  g->mel_to_linear(mel_dev_ptr, F);
  HIP_CHECK(hipEventRecord(g->ev.e[1], g->stream));
  gl_iterate_and_fetch(g, b, nullptr, g->iters, audio, n_samples, true);  // GriffinLim::infer: G1..G6
}

xdtts_status xdtts_griffinlim_infer(xdtts_griffinlim *g, const float *mel, size_t n_mels, size_t n_frames,
                                    float **audio, size_t *n_samples) {
  return guard([&] {
    if (!g || !mel || !audio || !n_samples) fail(XDTTS_ERR_BAD_ARG, "null argument");
    *audio = nullptr;
    *n_samples = 0;
    if ((int)n_mels != g->n_mels) fail(XDTTS_ERR_BAD_ARG, "mel has %zu rows, basis has %d", n_mels, g->n_mels);
    if (n_frames < 2) fail(XDTTS_ERR_BAD_ARG, "need at least 2 frames, got %zu", n_frames);
    std::lock_guard<std::mutex> lk(g->mu);
    HIP_CHECK(hipSetDevice(g->device));
    g->mel_in.upload(mel, n_mels * n_frames, g->stream);
    HIP_CHECK(hipStreamSynchronize(g->stream));
    gl_run_from_device_mel(g, g->mel_in.p, (int)n_frames, audio, n_samples);
  });
}

// The vocoder half of a batch from a mel that is already in HBM (on g's device): [n_mels][sum Fu], utterance u at columns
// fbase[u] .. fbase[u] + Fu[u].  mel -> linear is one GEMM over all frames, and the persistent kernel takes as many
// utterances per launch as fit one workgroup per CU (a workgroup never spans two utterances and exchanges overlaps only
// inside its own).  Caller holds g->mu.  The reads of the mel are enqueued on g->stream: the caller orders them behind
// the mel's producer (a stream sync or an event wait on g->stream).
static void gl_batch_from_device(xdtts_griffinlim *g, const float *mel_dev_all, const std::vector<int> &Fu, float **audios,
                                 size_t *n_samples) {
  {
    const int n_utt = (int)Fu.size();
    std::vector<int> fbase(n_utt), abase(n_utt);
    size_t Ftot = 0, Ntot = 0;
    for (int u = 0; u < n_utt; ++u) {
      fbase[u] = (int)Ftot;
      abase[u] = (int)Ntot;
      Ftot += (size_t)Fu[u];
      Ntot += (size_t)g->hop * (size_t)(Fu[u] - 1);
      if (Ftot > (1u << 24)) fail(XDTTS_ERR_BAD_ARG, "batch too large");
    }
    HIP_CHECK(hipSetDevice(g->device));
    hipStream_t st = g->stream;
    std::vector<int> fl(Ftot);
    for (int u = 0; u < n_utt; ++u)
      for (int f = 0; f < Fu[u]; ++f) fl[(size_t)fbase[u] + f] = f;
    g->frame_local.upload(fl.data(), fl.size(), st);
    GlBufs all = g->bufs((int)Ftot);
    g->audio.alloc(std::max<size_t>(Ntot, 1));
    HIP_CHECK(hipStreamSynchronize(st));  // the host vector above
    const float alpha = g->momentum / (1.0f + g->momentum);
    std::lock_guard<ChipLock> chip(chip_mutex(g->device));
    g->probe_tick();
    for (int attempt = 0;; ++attempt) {
      HIP_CHECK(hipEventRecord(g->ev.e[0], st));
      g->mel_to_linear(mel_dev_all, (int)Ftot);
      HIP_CHECK(hipEventRecord(g->ev.e[1], st));
      launch_gl_phase_init_batch(all, g->seed, g->frame_local.p, st);
      // pack consecutive utterances into persistent launches of <= one workgroup per CU.  A workgroup owns up to
      // 4 frames (one wave each) or up to 8 (two waves per SIMD): an iteration of the 8-frame shape takes 6.8 us
      // against 5.35 us (tools/gl_tf_sweep.py), so it wins as soon as it saves launches.  Two 4-frame workgroups
      // per CU (k_gl_persistent<4, 2>: the state in LDS, 256 registers) take 7.1 us for the same eight frames
      // (tools/vocoder_batch.py) and keep the 4-frame split, i.e. the single call's audio bit for bit.
      const bool pers = g->persistent_usable();
      std::vector<GlSeg> segs;
      struct Launch { int seg0, nblk; std::vector<int> utts; };
      std::vector<Launch> launches;
      std::vector<char> batched(n_utt, 0);
      auto pack = [&](int tf, int wg, bool build) {  // returns the relative cost: launches x time per iteration of the shape
        // first-fit decreasing over launches of n_cu workgroups (which launch an utterance rides in does not
        // change its audio: its own split into workgroups depends on its frame count alone)
        std::vector<std::pair<int, int>> items;  // (workgroups, utterance)
        int n_alone = 0;
        for (int u = 0; u < n_utt; ++u) {
          const int nb = (Fu[u] + tf - 1) / tf;
          if (Fu[u] < 16 || nb > g->n_cu || Fu[u] / nb < 3) {  // on its own below
            n_alone += Fu[u] >= 16;  // (a launch of the 5..8-frame shape; the tiny ones cost next to nothing)
            continue;
          }
          items.emplace_back(nb, u);
        }
        std::stable_sort(items.begin(), items.end(), [](const std::pair<int, int> &a, const std::pair<int, int> &b) { return a.first > b.first; });
        std::vector<int> room;                 // free workgroups of each launch
        std::vector<std::vector<int>> riders;  // its utterances
        for (const auto &it : items) {
          size_t k = 0;
          while (k < room.size() && room[k] < it.first) ++k;
          if (k == room.size()) {
            room.push_back(g->n_cu * wg);
            riders.emplace_back();
          }
          room[k] -= it.first;
          riders[k].push_back(it.second);
        }
        if (build)
          for (size_t k = 0; k < riders.size(); ++k) {
            const int seg0 = (int)segs.size();
            for (int u : riders[k]) {
              const int nb = (Fu[u] + tf - 1) / tf;
              for (int b = 0; b < nb; ++b) {
                GlSeg sg{};
                sg.fbase = fbase[u];
                sg.F = Fu[u];
                sg.f0 = (int)(((long long)b * Fu[u]) / nb);
                sg.n_own = (int)(((long long)(b + 1) * Fu[u]) / nb) - sg.f0;
                sg.first = b == 0;
                sg.last = b + 1 == nb;
                sg.abase = abase[u];
                segs.push_back(sg);
              }
              batched[u] = 1;
            }
            launches.push_back({seg0, (int)segs.size() - seg0, riders[k]});
          }
        // us per iteration of one launch of each shape (tools/vocoder_shapes.py, round 4 with the 16-byte exchange granules:
        // 4.5-5.2 / 5.8-6.3 / 5.9-6.3; round 3: 5.35 / 7.1 / 6.8) -- at equal cost the 4-frame shape, whose audio is the single call's
        return (tf <= 4 ? (wg > 1 ? 6.0 : 4.85) : 6.1) * (double)riders.size() + 6.1 * n_alone;
      };
      int TF = 4, WG = 1;
      if (pers) {
        // (the 4-frame shape splits an utterance the way its own call does, one or two workgroups per CU: batch_shape 4)
        if (g->per_cu4 >= 2 && pack(4, 2, false) < pack(4, 1, false)) WG = 2;
        if (g->gopts.batch_shape == 0 && pack(GLP_TF_MAX, 1, false) < pack(4, WG, false)) TF = GLP_TF_MAX, WG = 1;
        if (const char *fs = getenv("XDTTS_GL_BATCH_FORCE")) {  // developer: "8" = 8-frame workgroups, "41" / "42" = 4-frame, one / two per CU
          const int v = atoi(fs);
          if (v == 8) TF = GLP_TF_MAX, WG = 1;
          if (v == 41) TF = 4, WG = 1;
          if (v == 42 && g->per_cu4 >= 2) TF = 4, WG = 2;
        }
        pack(TF, WG, true);
      }
      bool used_persistent = false;
      std::vector<PinnedGuard> out;  // each utterance straight into the buffer the caller receives
      out.reserve((size_t)n_utt);
      for (int u = 0; u < n_utt; ++u) out.emplace_back((size_t)g->hop * (size_t)(Fu[u] - 1));
      struct Drain {  // no buffer of `out` goes back to the pool while a copy into it may be in flight
        hipStream_t &s;
        ~Drain() {
          if (s) (void)hipStreamSynchronize(s);
        }
      } drain{g->copy_stream};
      // Output normalisation (G6): the utterances of one fetch are consecutive rows of `tab` (launch by launch, then the
      // ones that run alone), so each fetch is preceded by ONE two-launch normalisation of exactly its utterances.
      const int norm_mode = g->gopts.output_normalise;
      std::vector<int> tab_pos((size_t)n_utt, 0);
      if (norm_mode) {
        std::vector<int2> tab;
        tab.reserve((size_t)n_utt);
        auto add = [&](int u) {
          tab_pos[(size_t)u] = (int)tab.size();
          tab.push_back(make_int2(abase[u], g->hop * (Fu[u] - 1)));
        };
        for (const Launch &L : launches)
          for (int u : L.utts) add(u);
        for (int u = 0; u < n_utt; ++u)
          if (!batched[u]) add(u);
        g->norm_tab.upload(tab.data(), tab.size(), st);
        g->norm_parts.alloc((size_t)GLN_SCRATCH * (size_t)n_utt);
        HIP_CHECK(hipStreamSynchronize(st));
      }
      size_t n_ev = 0;
      auto fetch_audio = [&](const std::vector<int> &utts) {  // after the work just enqueued on `st`
        if (norm_mode && !utts.empty()) {
          int n_max = 0;
          for (int u : utts) n_max = std::max(n_max, g->hop * (Fu[u] - 1));
          const int r0 = tab_pos[(size_t)utts[0]];
          launch_gl_output_normalise(g->audio.p, g->norm_tab.p + r0, (int)utts.size(), 0, n_max, norm_mode, g->gopts.rms_target,
                                     g->norm_parts.p + (size_t)r0 * GLN_SCRATCH, st);
        }
        hipEvent_t e = g->launch_done(n_ev++);
        HIP_CHECK(hipEventRecord(e, st));
        HIP_CHECK(hipStreamWaitEvent(g->copy_stream, e, 0));
        for (int u : utts)
          HIP_CHECK(hipMemcpyAsync(out[(size_t)u].p, g->audio.p + abase[u], sizeof(float) * (size_t)g->hop * (size_t)(Fu[u] - 1),
                                   hipMemcpyDeviceToHost, g->copy_stream));
      };
      if (!segs.empty()) {
        g->segs.upload(segs.data(), segs.size(), st);
        HIP_CHECK(hipStreamSynchronize(st));
        const size_t words = gl_persistent_xch_words(g->n_cu * std::max(1, std::min(g->per_cu4, 2)));
        unsigned need = 0;
        for (size_t i = 0; i < launches.size(); ++i) need += (unsigned)g->iters + 2u;
        if (words > g->xch.n || g->epoch > 0x7fff0000u - need) {
          g->xch.alloc(words);
          HIP_CHECK(hipMemsetAsync(g->xch.p, 0, g->xch.n * sizeof(unsigned long long), st));
          g->epoch = 0;
        }
        if (!g->gl_err.p) {
          g->gl_err.alloc(1);
          HIP_CHECK(hipMemsetAsync(g->gl_err.p, 0, sizeof(int), st));
          HIP_CHECK(hipHostMalloc((void **)&g->host_err, sizeof(int), hipHostMallocDefault));
        }
        for (const Launch &L : launches) {
          GlPersist p{};
          p.segs = g->segs.p + L.seg0;
          p.xch = g->xch.p;
          p.err = g->gl_err.p;
          p.epoch = g->epoch;
          p.nblk = L.nblk;
          p.TF = TF;
          p.per_cu = WG;
          p.poll_delay = 6;
          g->epoch += (unsigned)g->iters + 2u;
          launch_gl_persistent(all, p, all.ang, all.tprev, g->iters, alpha, g->audio.p, st);
          fetch_audio(L.utts);
        }
        used_persistent = true;
      }
      for (int u = 0; u < n_utt; ++u) {  // the rest one by one (tiny / very long utterances, or a demoted handle)
        if (batched[u]) continue;
        GlBufs v = all;
        v.F = Fu[u];
        v.S = all.S + (size_t)fbase[u] * g->nb;
        v.ang = all.ang + (size_t)fbase[u] * g->nb;
        v.ang2 = all.ang2 + (size_t)fbase[u] * g->nb;
        v.tprev = all.tprev + (size_t)fbase[u] * g->nb;
        v.frames = all.frames + (size_t)fbase[u] * g->n_fft;
        v.wss_inv = all.wss_inv + abase[u];
        launch_gl_prepare(v, st);
        g->run_iterations(v, g->iters, alpha, g->audio.p + abase[u]);  // the engine the single-utterance call uses
        used_persistent = used_persistent || g->last_persistent;
        fetch_audio(std::vector<int>(1, u));
      }
      HIP_CHECK(hipEventRecord(g->ev.e[2], st));
      g->finish_timings();
      HIP_CHECK(hipStreamSynchronize(g->copy_stream));
      g->last_persistent = used_persistent;
      if (g->persistent_failed()) {
        if (attempt) fail(XDTTS_ERR_HIP, "Griffin-Lim batch: exchange failure on the fallback engine");
        continue;  // demoted: everything runs one by one on the launch-per-iteration kernels
      }
      for (int u = 0; u < n_utt; ++u) {
        audios[u] = out[(size_t)u].release();
        n_samples[u] = (size_t)g->hop * (size_t)(Fu[u] - 1);
      }
      return;
    }
  }
}

// GriffinLim::infer for several utterances at once (the vocoder half of a batch, BASELINE.json configs[3]):
// the utterances' frames are concatenated, mel -> linear is one GEMM over all of them, and the persistent
// kernel takes as many utterances per launch as fit one workgroup per CU.  Every utterance's audio is bit-identical to what
// xdtts_griffinlim_infer returns for it alone.
xdtts_status xdtts_griffinlim_infer_batch(xdtts_griffinlim *g, const float *const *mels, size_t n_mels, const size_t *n_frames,
                                          int32_t n_utt, float **audios, size_t *n_samples) {
  return guard([&] {
    if (!g || !mels || !n_frames || !audios || !n_samples || n_utt <= 0) fail(XDTTS_ERR_BAD_ARG, "bad argument");
    if ((int)n_mels != g->n_mels) fail(XDTTS_ERR_BAD_ARG, "mel has %zu rows, basis has %d", n_mels, g->n_mels);
    std::vector<int> fbase(n_utt), Fu(n_utt);
    size_t Ftot = 0;
    for (int u = 0; u < n_utt; ++u) {
      audios[u] = nullptr;
      n_samples[u] = 0;
      if (!mels[u] || n_frames[u] < 2) fail(XDTTS_ERR_BAD_ARG, "utterance %d: need at least 2 frames", u);
      fbase[u] = (int)Ftot;
      Fu[u] = (int)n_frames[u];
      Ftot += n_frames[u];
      if (Ftot > (1u << 24)) fail(XDTTS_ERR_BAD_ARG, "batch too large");
    }
    std::lock_guard<std::mutex> lk(g->mu);
    HIP_CHECK(hipSetDevice(g->device));
    // mel of all utterances side by side: [n_mels][Ftot], staged in pinned memory (one fast upload)
    PinnedGuard mel_all((size_t)n_mels * Ftot);
    for (int u = 0; u < n_utt; ++u)
      for (size_t m = 0; m < n_mels; ++m)
        std::memcpy(mel_all.p + m * Ftot + fbase[u], mels[u] + m * n_frames[u], sizeof(float) * n_frames[u]);
    g->mel_in.upload(mel_all.p, (size_t)n_mels * Ftot, g->stream);
    HIP_CHECK(hipStreamSynchronize(g->stream));  // the staging buffer goes back to the pool
    gl_batch_from_device(g, g->mel_in.p, Fu, audios, n_samples);
  });
}

xdtts_status xdtts_griffinlim_mel_to_linear(xdtts_griffinlim *g, const float *mel, size_t n_mels, size_t n_frames,
                                            float *S_out) {
  return guard([&] {
    if (!g || !mel || !S_out || n_frames == 0) fail(XDTTS_ERR_BAD_ARG, "bad argument");
    if ((int)n_mels != g->n_mels) fail(XDTTS_ERR_BAD_ARG, "mel has %zu rows, basis has %d", n_mels, g->n_mels);
    std::lock_guard<std::mutex> lk(g->mu);
    HIP_CHECK(hipSetDevice(g->device));
    const int F = (int)n_frames;
    g->mel_in.upload(mel, n_mels * n_frames, g->stream);
    HIP_CHECK(hipStreamSynchronize(g->stream));
    g->bufs(F);
    g->mel_to_linear(g->mel_in.p, F);
    // S is [F][nb] on the device; the boundary layout is the crate's (nb x F)
    g->frames.alloc((size_t)F * g->n_fft);
    launch_transpose(g->S.p, g->frames.p, F, g->nb, g->stream);
    HIP_CHECK(hipMemcpyAsync(S_out, g->frames.p, (size_t)F * g->nb * sizeof(float), hipMemcpyDeviceToHost, g->stream));
    HIP_CHECK(hipStreamSynchronize(g->stream));
  });
}

xdtts_status xdtts_griffinlim_infer_linear(xdtts_griffinlim *g, const float *S, const float *phase0, size_t n_frames,
                                           size_t iters, float **audio, size_t *n_samples) {
  return guard([&] {
    if (!g || !S || !audio || !n_samples) fail(XDTTS_ERR_BAD_ARG, "null argument");
    *audio = nullptr;
    *n_samples = 0;
    if (n_frames < 2) fail(XDTTS_ERR_BAD_ARG, "need at least 2 frames, got %zu", n_frames);
    std::lock_guard<std::mutex> lk(g->mu);
    HIP_CHECK(hipSetDevice(g->device));
    const int F = (int)n_frames;
    GlBufs b = g->bufs(F);
    // boundary layout (nb x F) -> device layout [F][nb]
    g->frames.upload(S, (size_t)F * g->nb, g->stream);
    launch_transpose(g->frames.p, g->S.p, g->nb, F, g->stream);
    const float *p0 = nullptr;
    if (phase0) {
      g->phase0.upload(phase0, (size_t)F * g->nb * 2, g->stream);
      p0 = g->phase0.p;
    }
    HIP_CHECK(hipStreamSynchronize(g->stream));
    HIP_CHECK(hipEventRecord(g->ev.e[0], g->stream));
    HIP_CHECK(hipEventRecord(g->ev.e[1], g->stream));
    gl_iterate_and_fetch(g, b, p0, iters ? (int)iters : g->iters, audio, n_samples, false);  // G2..G5 + final ISTFT only
  });
}

// Parity hook: `n_iter` Griffin-Lim iterations (no final ISTFT) from a caller-held state.
xdtts_status xdtts_griffinlim_step(xdtts_griffinlim *g, const float *S, float *angles, float *rebuilt, size_t n_frames,
                                   size_t n_iter) {
  return guard([&] {
    if (!g || !S || !angles || !rebuilt) fail(XDTTS_ERR_BAD_ARG, "null argument");
    if (n_frames < 2) fail(XDTTS_ERR_BAD_ARG, "need at least 2 frames, got %zu", n_frames);
    if (n_iter == 0) n_iter = 1;
    std::lock_guard<std::mutex> lk(g->mu);
    HIP_CHECK(hipSetDevice(g->device));
    const int F = (int)n_frames;
    const size_t ne = (size_t)F * g->nb;
    GlBufs b = g->bufs(F);
    g->frames.upload(S, ne, g->stream);
    launch_transpose(g->frames.p, g->S.p, g->nb, F, g->stream);
    g->phase0.alloc(ne * 4);  // staging: angles then rebuilt, (nb x F x 2) each
    HIP_CHECK(hipMemcpyAsync(g->phase0.p, angles, ne * 2 * sizeof(float), hipMemcpyHostToDevice, g->stream));
    HIP_CHECK(hipMemcpyAsync(g->phase0.p + ne * 2, rebuilt, ne * 2 * sizeof(float), hipMemcpyHostToDevice, g->stream));
    launch_gl_state_import(b, g->phase0.p, g->phase0.p + ne * 2, g->stream);
    launch_gl_prepare(b, g->stream);
    const float alpha = g->momentum / (1.0f + g->momentum);
    std::lock_guard<ChipLock> chip(chip_mutex(g->device));
    g->probe_tick();
    for (int attempt = 0;; ++attempt) {
      const float2 *tp = nullptr;
      const float2 *fin = g->run_iterations(b, (int)n_iter, alpha, nullptr, true, &tp);
      if (!g->last_persistent) {  // the launch engine updates the state in place
        launch_gl_state_export(b, fin, b.tprev, g->phase0.p, g->phase0.p + ne * 2, g->stream);
        break;
      }
      HIP_CHECK(hipStreamSynchronize(g->stream));
      if (!g->persistent_failed()) {
        launch_gl_state_export(b, fin, tp, g->phase0.p, g->phase0.p + ne * 2, g->stream);
        break;
      }
      if (attempt) fail(XDTTS_ERR_HIP, "Griffin-Lim step: exchange failure");
    }
    HIP_CHECK(hipMemcpyAsync(angles, g->phase0.p, ne * 2 * sizeof(float), hipMemcpyDeviceToHost, g->stream));
    HIP_CHECK(hipMemcpyAsync(rebuilt, g->phase0.p + ne * 2, ne * 2 * sizeof(float), hipMemcpyDeviceToHost, g->stream));
    HIP_CHECK(hipStreamSynchronize(g->stream));
  });
}

xdtts_status xdtts_griffinlim_last_timings(const xdtts_griffinlim *g, float ms[3]) {
  return guard([&] {
    if (!g || !ms) fail(XDTTS_ERR_BAD_ARG, "null argument");
    for (int i = 0; i < 3; ++i) ms[i] = g->last_ms[i];
  });
}

void xdtts_griffinlim_free(xdtts_griffinlim *g) {
  if (!g) return;
  (void)hipSetDevice(g->device);
  if (g->stream) (void)hipStreamSynchronize(g->stream);
  delete g;
}

// ---- XdTts::infer (src/lib.rs:110-159) --------------------------------------------------------------

xdtts_status xdtts_synthesize_ids(xdtts_tacotron2 *h, xdtts_griffinlim *g, const int64_t *ids, size_t n,
                                  const size_t *splits, size_t n_splits, const xdtts_infer_opts *opts, float **mel,
                                  size_t *n_frames, float **audio, size_t *n_samples) {
  return guard([&] {
    if (!h || !g || !mel || !n_frames || !audio || !n_samples) fail(XDTTS_ERR_BAD_ARG, "null argument");
    if (h->device != g->device) fail(XDTTS_ERR_BAD_ARG, "tacotron2 and griffin-lim handles live on different devices");
    *mel = nullptr;
    *audio = nullptr;
    *n_frames = *n_samples = 0;
    std::lock_guard<std::mutex> lk(h->mu);
    std::lock_guard<std::mutex> lk2(g->mu);
    const xdtts_infer_opts o = resolve_opts(opts);
    std::vector<int64_t> padded;
    std::vector<int> lens;
    chunks_from_splits(ids, n, splits, n_splits, o.max_chunk, padded, lens);
    int total = 0;
    h->infer_batch_device(padded.data(), lens.data(), (int)lens.size(), o.max_chunk, o, nullptr, &total);
    if (total < 2) fail(XDTTS_ERR_BAD_ARG, "mel has %d frame(s); the vocoder needs at least 2", total);
    PinnedGuard mel_host((size_t)N_MEL * total);
    // the vocoder stream reads the mel behind the post-net (event 3 of infer_batch_device); the mel's copy to the host
    // follows on the mel-gen stream and overlaps the vocoder
    HIP_CHECK(hipStreamWaitEvent(g->stream, h->ev.e[3], 0));
    HIP_CHECK(hipMemcpyAsync(mel_host.p, h->mel_dev.p, (size_t)N_MEL * total * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    struct Drain {  // the pinned buffer does not go back to the pool with the copy in flight
      hipStream_t s;
      ~Drain() { (void)hipStreamSynchronize(s); }
    } drain{h->stream};
    gl_run_from_device_mel(g, h->mel_dev.p, total, audio, n_samples);
    h->finish_timings();  // (stream sync: the mel has landed)
    *mel = mel_host.release();
    *n_frames = (size_t)total;
  });
}

// XdTts::infer for a SEQUENCE of utterances, one after the other as the reference runs them (src/lib.rs:110-159: each utterance
// decoded alone, batch 1) -- but software-pipelined across the two halves: the frame loop owns every CU (weights in the register
// files), so nothing can run beside it; what can overlap is utterance u's vocoder (mel -> linear, Griffin-Lim, normalise: its own
// stream) with utterance u + 1's ENCODER (embedding, three convolutions, BiLSTM on 16 CUs, memory layer).  The frame loop of
// u + 1 is ordered behind the vocoder of u by an event (two grids that each want the chip co-resident never meet), and the host
// collects u's audio while u + 1 decodes.  Same bits as xdtts_synthesize_ids called once per utterance.
xdtts_status xdtts_synthesize_sequence(xdtts_tacotron2 *h, xdtts_griffinlim *g, const int64_t *const *ids, const size_t *n_ids,
                                       const size_t *const *splits, const size_t *n_splits, int32_t n_utt, const xdtts_infer_opts *opts,
                                       float **mels, size_t *n_frames, float **audios, size_t *n_samples) {
  return guard([&] {
    if (!h || !g || !ids || !n_ids || !n_frames || !audios || !n_samples || n_utt <= 0) fail(XDTTS_ERR_BAD_ARG, "null argument / no utterance");
    if (h->device != g->device) fail(XDTTS_ERR_BAD_ARG, "tacotron2 and griffin-lim handles live on different devices");
    for (int u = 0; u < n_utt; ++u) {
      audios[u] = nullptr;
      n_frames[u] = n_samples[u] = 0;
      if (mels) mels[u] = nullptr;
      if (!ids[u] || n_ids[u] == 0) fail(XDTTS_ERR_BAD_ARG, "utterance %d is empty", u);
    }
    std::lock_guard<std::mutex> lk(h->mu);
    std::lock_guard<std::mutex> lk2(g->mu);
    std::lock_guard<ChipLock> chip(chip_mutex(h->device));  // the whole sequence: co-resident launches of two streams are in flight
    const xdtts_infer_opts o = resolve_opts(opts);
    struct Hook {  // (the hook never outlives this call, whatever throws)
      xdtts_tacotron2 *h;
      ~Hook() {
        h->before_decoder = nullptr;
        h->while_decoding = nullptr;
      }
    } unhook{h};
    std::vector<PinnedGuard> mel_host(n_utt), audio_host(n_utt);
    std::vector<int> total(n_utt, 0);
    std::vector<float *> audio_out(n_utt, nullptr);
    // timing events per utterance (the post-net of u ends while the host is already enqueuing u + 1): utterance u records into a
    // fresh set, read when everything has drained; xdtts_*_last_timings then report the SUMS over the sequence
    std::vector<Events> per(n_utt);
    for (Events &e : per) e.create();
    // Declared BEHIND the pinned buffers and the event sets, i.e. destroyed BEFORE them: whatever throws (utterance u's chunking
    // fails while utterance u - 1's mel copy, vocoder and audio copy are still in flight), both streams drain first and only then
    // do the buffers go back to the shared pool and the events get destroyed.
    struct Drain {
      hipStream_t a, b;
      ~Drain() {
        (void)hipStreamSynchronize(a);
        (void)hipStreamSynchronize(b);
      }
    } drain{h->stream, g->stream};
    float gsum[3] = {0.f, 0.f, 0.f};
    int steps_sum = 0;
    auto add_gl = [&]() {
      for (int i = 0; i < 3; ++i) gsum[i] += g->last_ms[i];
    };
    auto release_all = [&]() {
      for (int u = 0; u < n_utt; ++u)
        if (audio_out[u]) pinned_pool().put(audio_out[u]);
    };
    try {
      for (int u = 0; u < n_utt; ++u) {
        std::vector<int64_t> padded;
        std::vector<int> lens;
        chunks_from_splits(ids[u], n_ids[u], splits ? splits[u] : nullptr, (splits && n_splits) ? n_splits[u] : 0, o.max_chunk, padded, lens);
        if (u > 0) h->before_decoder = [&] { HIP_CHECK(hipStreamWaitEvent(h->stream, g->ev.e[2], 0)); };  // vocoder of u - 1 done (its audio copy is behind it on g->stream)
        // The pinned output buffers of utterance u are taken from the pool WHILE its frame loop runs (the host has 5.6 ms to wait there), not
        // between the vocoder's enqueue and the next encoder's: a pool miss is a hipHostMalloc of 0.8 MB -- 0.2-0.4 ms on some boxes -- and in
        // that place it made the next encoder start when the vocoder had finished instead of beside it (round 6: the sequence headline's two
        // modes, 6.15-6.25 / 6.5-6.6 ms per utterance; the kernel timeline of tools/sequence_timeline.sh shows k_embed behind k_gl_persistent
        // in the slow utterances).  Gate-less decodes only: the frame count is then known beforehand.
        long predicted = 0;
        if (o.fixed_steps > 0 || o.fixed_frames_per_id > 0.f)
          for (int len : lens) {
            const long l = o.fixed_steps > 0 ? o.fixed_steps : std::lround((double)o.fixed_frames_per_id * len);
            predicted += std::min<long>(std::max<long>(l, 1), o.max_steps);
          }
        h->while_decoding = nullptr;
        if (predicted >= 2)
          h->while_decoding = [&, u, predicted] {
            if (!mel_host[u].p) mel_host[u] = PinnedGuard((size_t)N_MEL * predicted);
            if (!audio_host[u].p) audio_host[u] = PinnedGuard((size_t)g->hop * (size_t)(predicted - 1));
          };
        for (int i = 0; i < 4; ++i) std::swap(h->ev.e[i], per[u].e[i]);  // (per[u] now holds what the handle had: utterance u - 1's set, or its own)
        h->infer_batch_device(padded.data(), lens.data(), (int)lens.size(), o.max_chunk, o, nullptr, &total[u]);
        h->before_decoder = nullptr;
        h->while_decoding = nullptr;
        if (predicted != total[u]) mel_host[u] = PinnedGuard(), audio_host[u] = PinnedGuard();  // (the stop rule decided otherwise: sized below)
        steps_sum += h->last_steps;
        if (total[u] < 2) fail(XDTTS_ERR_BAD_ARG, "utterance %d: mel has %d frame(s); the vocoder needs at least 2", u, total[u]);
        // the frame loop of u has drained, and it waited for the vocoder of u - 1: collect that audio now
        if (u > 0) {
          gl_collect(g, total[u - 1], audio_host[u - 1], &audio_out[u - 1], &n_samples[u - 1]);
          add_gl();
        }
        if (!mel_host[u].p) mel_host[u] = PinnedGuard((size_t)N_MEL * total[u]);
        HIP_CHECK(hipStreamWaitEvent(g->stream, h->ev.e[3], 0));  // the vocoder reads the mel behind the post-net
        HIP_CHECK(hipMemcpyAsync(mel_host[u].p, h->mel_dev.p, (size_t)N_MEL * total[u] * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        gl_enqueue_from_device_mel(g, h->mel_dev.p, total[u], audio_host[u]);
      }
      gl_collect(g, total[n_utt - 1], audio_host[n_utt - 1], &audio_out[n_utt - 1], &n_samples[n_utt - 1]);
      add_gl();
      h->finish_timings();  // (stream sync: every mel has landed; last_ms = the last utterance's phases)
      float hsum[4] = {h->last_ms[0], h->last_ms[1], h->last_ms[2], h->last_ms[3]};
      for (int u = 1; u < n_utt; ++u) {  // utterance u - 1's events sit in per[u]
        float ms = 0.f;
        for (int i = 0; i < 3; ++i) {
          HIP_CHECK(hipEventElapsedTime(&ms, per[u].e[i], per[u].e[i + 1]));
          hsum[i] += ms;
        }
        HIP_CHECK(hipEventElapsedTime(&ms, per[u].e[0], per[u].e[3]));
        hsum[3] += ms;
      }
      for (int i = 0; i < 4; ++i) h->last_ms[i] = hsum[i];
      for (int i = 0; i < 3; ++i) g->last_ms[i] = gsum[i];
      h->last_steps = steps_sum;
    } catch (...) {
      (void)hipStreamSynchronize(h->stream);  // (nothing in flight writes into a buffer that is handed back below)
      (void)hipStreamSynchronize(g->stream);
      release_all();
      throw;
    }
    for (int u = 0; u < n_utt; ++u) {
      audios[u] = audio_out[u];
      n_frames[u] = (size_t)total[u];
      if (mels) mels[u] = mel_host[u].release();
    }
  });
}

// XdTts::infer for several utterances in one call (BASELINE.json configs[3]; the author's "batched / parallel
// sentences" note, src/phonemes.rs:677-680): all chunks through one lock-step mel-gen batch (chunks are independent,
// src/tacotron2/mod.rs:422-434), the post-net writes every utterance's chunks side by side on the time axis
// (mod.rs:430), and the vocoder batch reads that mel where it lies in HBM -- no copy to the host and back, no
// re-staging between the two halves.  The per-utterance mels leave for the host while the vocoder runs.
xdtts_status xdtts_synthesize_batch(xdtts_tacotron2 *h, xdtts_griffinlim *g, const int64_t *ids, const int32_t *lens, int32_t B,
                                    int32_t t_stride, const int32_t *utt_chunks, int32_t n_utt, const xdtts_infer_opts *opts,
                                    const int32_t *fixed_steps_per_item, float **mels, size_t *n_frames, float **audios,
                                    size_t *n_samples) {
  return guard([&] {
    if (!h || !g || !ids || !lens || !utt_chunks || !n_frames || !audios || !n_samples) fail(XDTTS_ERR_BAD_ARG, "null argument");
    if (h->device != g->device) fail(XDTTS_ERR_BAD_ARG, "tacotron2 and griffin-lim handles live on different devices");
    if (B <= 0 || n_utt <= 0 || t_stride <= 0) fail(XDTTS_ERR_BAD_ARG, "batch %d / utterances %d / stride %d out of range", B, n_utt, t_stride);
    long nchunks = 0;
    for (int u = 0; u < n_utt; ++u) {
      if (utt_chunks[u] <= 0) fail(XDTTS_ERR_BAD_ARG, "utterance %d has no chunk", u);
      nchunks += utt_chunks[u];
      audios[u] = nullptr;
      n_frames[u] = n_samples[u] = 0;
      if (mels) mels[u] = nullptr;
    }
    if (nchunks != B) fail(XDTTS_ERR_BAD_ARG, "utt_chunks sum to %ld, the batch has %d chunks", nchunks, B);
    std::lock_guard<std::mutex> lk(h->mu);
    std::lock_guard<std::mutex> lk2(g->mu);
    const xdtts_infer_opts o = resolve_opts(opts);
    const int T = o.max_chunk;
    std::vector<int64_t> padded((size_t)B * T, 0);
    for (int b = 0; b < B; ++b) {
      if (lens[b] > T) fail(XDTTS_ERR_TOO_LONG, "chunk %d has %d ids, window is %d", b, lens[b], T);
      if (lens[b] > t_stride || lens[b] <= 0) fail(XDTTS_ERR_BAD_ARG, "chunk %d: bad length %d", b, lens[b]);
      std::copy(ids + (size_t)b * t_stride, ids + (size_t)b * t_stride + lens[b], padded.begin() + (size_t)b * T);
    }
    int total = 0;
    const std::vector<int> F = h->infer_batch_device(padded.data(), lens, B, T, o, fixed_steps_per_item, &total);
    std::vector<int> Fu(n_utt, 0), col0(n_utt, 0);
    for (int u = 0, b = 0, off = 0; u < n_utt; ++u) {
      col0[u] = off;
      for (int k = 0; k < utt_chunks[u]; ++k) Fu[u] += F[b++];
      off += Fu[u];
      if (Fu[u] < 2) fail(XDTTS_ERR_BAD_ARG, "utterance %d has %d mel frame(s); the vocoder needs at least 2", u, Fu[u]);
    }
    // the vocoder stream reads the mel behind the post-net (event 3 of infer_batch_device); the host copies of the
    // mel follow on the mel-gen stream and overlap the vocoder
    HIP_CHECK(hipStreamWaitEvent(g->stream, h->ev.e[3], 0));
    std::vector<PinnedGuard> mel_out;
    struct Drain {  // no mel buffer goes back to the pool while a copy into it may be in flight
      hipStream_t s;
      ~Drain() { (void)hipStreamSynchronize(s); }
    } drain{h->stream};
    if (mels) {
      mel_out.reserve((size_t)n_utt);
      for (int u = 0; u < n_utt; ++u) {
        mel_out.emplace_back((size_t)N_MEL * Fu[u]);
        HIP_CHECK(hipMemcpy2DAsync(mel_out[(size_t)u].p, sizeof(float) * (size_t)Fu[u], h->mel_dev.p + col0[u], sizeof(float) * (size_t)total,
                                   sizeof(float) * (size_t)Fu[u], N_MEL, hipMemcpyDeviceToHost, h->stream));
      }
    }
    gl_batch_from_device(g, h->mel_dev.p, Fu, audios, n_samples);
    try {
      h->finish_timings();  // (stream sync: the mel copies have landed)
    } catch (...) {  // the caller gets either every buffer of the call or none
      for (int u = 0; u < n_utt; ++u) {
        if (audios[u]) pinned_pool().put(audios[u]);
        audios[u] = nullptr;
        n_samples[u] = 0;
      }
      throw;
    }
    for (int u = 0; u < n_utt; ++u) {
      n_frames[u] = (size_t)Fu[u];
      if (mels) mels[u] = mel_out[(size_t)u].release();
    }
  });
}

}  // extern "C"
