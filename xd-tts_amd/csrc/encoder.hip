// encoder.hip -- encoder BiLSTM recurrence (inside encoder.onnx, src/tacotron2/mod.rs:379) spread
// over four CUs per (direction, chunk).
//
// The recurrence is T = 100 dependent steps of a 1024 x 256 mat-vec.  One CU cannot hold the 1 MB
// of W_hh on chip, so the single-block kernel (gemm.hip: k_bilstm) re-reads most of it through its
// L2 port every step (~8 us per step).  Here block k of 4 owns hidden units [64k, 64k+64) -- all
// four gates, so the cell update stays local -- and keeps its 256 x 256 slice of W_hh in registers
// for the whole sequence (64 floats per thread); per step the only traffic is the exchange of
// the 256-entry hidden state between the four blocks, done with data-tagged 8-byte granules
// ({step+1, value}, one relaxed agent-scope store per value; readers re-read until the tag
// matches -- MI355X_MICROARCH.md hand-off recipe R2): no flag, no fence, placement-independent.
// Two granule slots per value (step parity) make overwriting impossible: a block can only write
// step s+2 after every peer consumed its step-s value.  Every spin is bounded; a timeout raises
// an error word that the host turns into XDTTS_ERR_HIP.
#include "device_utils.h"
#include "kernels.h"

namespace xdtts {

namespace {

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;


constexpr int CO_BLOCKS = 4, CO_UNITS = ENC_H / CO_BLOCKS;  // 64 hidden units per block
constexpr unsigned SPIN_LIMIT = 1u << 22;
constexpr int ENC_FIRST_POLL = 2;  // delay before a gather thread's first poll, x 512 clocks (0 / 1 / 2 / 3 / 4: encoder 0.30 / 0.29 / 0.276 / 0.295 / 0.317 ms of the headline utterance)

// NC chunks per group of four blocks: the blocks' W_hh slices are the same for every chunk, so a batch that needs more than
// one launch of one-chunk groups (52 chunks: 26 + 26 on 256 CUs) runs as ONE launch of two-chunk groups instead -- a step is
// the exchange latency plus 64 FMAs per thread and chunk, so the second chunk costs ~10 % of a step, not a second launch
// (52-chunk encoder BiLSTM 0.39 -> 0.22 ms).  Cell updates of chunk j run on threads [64 j, 64 j + 64), its polls on
// threads [256 + 256 j, 512 + 256 j).
template <int NC>
__global__ __launch_bounds__(1024) void k_bilstm_coop(const float *__restrict__ xproj,
                                                      const float *__restrict__ whhT_f,
                                                      const float *__restrict__ whhT_b, float *memory, u64 *exchange,
                                                      int *err, int B, int T, int b0, int Btot, unsigned spin_limit, int fault, int first) {
  // (spin_limit / fault: test hooks -- poll limit, and a block (linear index + 1) that never publishes)
  // this launch runs chunks b0 .. b0+B-1 of a batch of Btot (xproj is [2][Btot][T][4H], memory [Btot][T][EMB])
  const int k = blockIdx.x, dir = blockIdx.y, tid = threadIdx.x;
  const int bz = NC * blockIdx.z;             // first chunk of this group (within the launch)
  const int nc = B - bz < NC ? B - bz : NC;   // chunks of this group (the last group of an odd batch has one)
  const int rl = tid & 255, q = tid >> 8;     // local gate row, column quarter
  const int g = rl >> 6, ul = rl & 63;
  const int row = g * ENC_H + CO_UNITS * k + ul;  // PyTorch gate order i,f,g,o
  const float *whhT = dir ? whhT_b : whhT_f;      // [256 cols][1024 rows]
  float w[64];
#pragma unroll
  for (int j = 0; j < 64; ++j) w[j] = whhT[(size_t)(64 * q + j) * (4 * ENC_H) + row];
  __shared__ float h[NC][ENC_H], part[NC][4][256];
  __shared__ int dead;
  const float *xp[NC];
  gu64 *ex[NC];
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const int b = bz + (j < nc ? j : 0);
    xp[j] = xproj + ((size_t)dir * Btot + b0 + b) * T * (4 * ENC_H);
    ex[j] = (gu64 *)(exchange + ((size_t)dir * B + b) * 2 * ENC_H);
  }
  float c = 0.f;  // cell state of (chunk tid / 64, unit tid % 64) on the cell-update threads
  for (int i = tid; i < NC * ENC_H; i += 1024) h[i / ENC_H][i % ENC_H] = 0.f;
  if (tid == 0) dead = 0;
  __syncthreads();
  float xnext[NC];  // the input projection runs one step ahead of its use
#pragma unroll
  for (int j = 0; j < NC; ++j) xnext[j] = (q == 0 && j < nc) ? xp[j][(size_t)(dir ? T - 1 : 0) * (4 * ENC_H) + row] : 0.f;
  for (int s = 0; s < T; ++s) {
    const int t = dir ? T - 1 - s : s;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      if (j >= nc) break;  // (block-uniform)
      const float xin = xnext[j];
      if (q == 0 && s + 1 < T) xnext[j] = xp[j][(size_t)(dir ? T - 2 - s : s + 1) * (4 * ENC_H) + row];
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
      for (int i = 0; i < 64; i += 4) {
        a0 = fmaf(w[i + 0], h[j][64 * q + i + 0], a0);
        a1 = fmaf(w[i + 1], h[j][64 * q + i + 1], a1);
        a2 = fmaf(w[i + 2], h[j][64 * q + i + 2], a2);
        a3 = fmaf(w[i + 3], h[j][64 * q + i + 3], a3);
      }
      part[j][q][rl] = ((a0 + a1) + (a2 + a3)) + xin;  // (xin: the input projection, held by the q = 0 quarter, 0 elsewhere)
    }
    __syncthreads();
    if (tid < NC * CO_UNITS) {
      // the four column quarters of the unit's four gate rows meet here (one barrier, no staging pass); hardware exp2 / rcp
      // forms as in the decoder engines
      const int j = tid / CO_UNITS, uu = tid % CO_UNITS;
      if (j < nc) {
        float gs[4];
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) gs[gi] = (part[j][0][64 * gi + uu] + part[j][1][64 * gi + uu]) + (part[j][2][64 * gi + uu] + part[j][3][64 * gi + uu]);
        const float ig = fast_sigmoid(gs[0]), fg = fast_sigmoid(gs[1]);
        const float gg = fast_tanh(gs[2]), og = fast_sigmoid(gs[3]);
        c = fmaf(fg, c, ig * gg);
        const float hn = og * fast_tanh(c);
        const int u = CO_UNITS * k + uu;
        h[j][u] = hn;
        memory[((size_t)(b0 + bz + j) * T + t) * EMB + dir * ENC_H + u] = hn;
        const bool lost = fault && (int)(blockIdx.x + CO_BLOCKS * (blockIdx.y + 2 * blockIdx.z)) == fault - 1;
        if (s + 1 < T && !lost)  // publish: tag = step + 1 (never 0), one naturally aligned 8-byte store
          __hip_atomic_store(ex[j] + (s & 1) * ENC_H + u, ((u64)(unsigned)(s + 1) << 32) | (u64)__float_as_uint(hn),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else if (tid >= 256 && tid < 256 + NC * ENC_H && s + 1 < T) {
      // gather the peers' units (one unit of one chunk per thread; the cell-update threads are
      // busy publishing): re-read the granule until its tag is this step's
      const int j = (tid - 256) / ENC_H, u = (tid - 256) % ENC_H;
      if (j < nc && (u >> 6) != k) {
        u64 v = 0;
        unsigned spins = 0;
        const bool skip = dead != 0;
        // the peers' cell updates and the stores' way through the fabric take ~0.6 us: a poll issued before that
        // only costs a round trip (x 512 clocks; the decoder's first-poll delay)
        for (int i = 0; i < first; ++i) __builtin_amdgcn_s_sleep(8);
        while (!skip) {
          v = __hip_atomic_load(ex[j] + (s & 1) * ENC_H + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((unsigned)(v >> 32) == (unsigned)(s + 1)) break;
          if (++spins > spin_limit) {
            dead = 1;
            atomicExch(err, 1);
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
        h[j][u] = __uint_as_float((unsigned)v);
      }
    }
    __syncthreads();
  }
}

}  // namespace

size_t bilstm_coop_exchange_words(int B) { return (size_t)2 * B * 2 * ENC_H; }

void launch_bilstm_coop(const float *xproj, const float *whhT_fwd, const float *whhT_bwd, float *memory,
                        unsigned long long *exchange, int *err, int B, int T, int group, hipStream_t s) {
  unsigned spins = SPIN_LIMIT;
  int fault = 0;
  if (const char *e = getenv("XDTTS_ENC_SPINS")) spins = (unsigned)atoi(e);  // test hooks for the lost-workgroup path
  if (const char *e = getenv("XDTTS_ENC_FAULT")) fault = atoi(e);
  int first = ENC_FIRST_POLL;
  if (const char *e = getenv("XDTTS_ENC_FIRST")) first = atoi(e);  // developer tuning knob
  // 8 workgroups per group of chunks must be co-resident, so a large batch runs as launches of at most `group` groups each
  // (sized to the CU count by the caller); a batch of more than `group` chunks puts two chunks on a group.  The exchange
  // buffer (bilstm_coop_exchange_words(2 * group) then) is reused between launches.
  const int nc = B > group ? 2 : 1, per_launch = nc * group;
  for (int b0 = 0; b0 < B; b0 += per_launch) {
    const int n = B - b0 < per_launch ? B - b0 : per_launch;
    // tags must start at 0 for every launch
    HIP_CHECK(hipMemsetAsync(exchange, 0, bilstm_coop_exchange_words(n) * sizeof(unsigned long long), s));
    const void *fn = nc == 2 ? reinterpret_cast<const void *>(k_bilstm_coop<2>) : reinterpret_cast<const void *>(k_bilstm_coop<1>);
    COOP_CHECK(launch_coresident(true, fn, dim3(CO_BLOCKS, 2, (n + nc - 1) / nc), dim3(1024), 0, s, xproj, whhT_fwd, whhT_bwd, memory, exchange,
                                 err, n, T, b0, B, spins, fault, first));
  }
}

}  // namespace xdtts
