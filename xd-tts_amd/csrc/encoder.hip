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

__global__ __launch_bounds__(1024) void k_bilstm_coop(const float *__restrict__ xproj,
                                                      const float *__restrict__ whhT_f,
                                                      const float *__restrict__ whhT_b, float *memory, u64 *exchange,
                                                      int *err, int B, int T, int b0, int Btot) {
  // this launch runs chunks b0 .. b0+B-1 of a batch of Btot (xproj is [2][Btot][T][4H], memory [Btot][T][EMB])
  const int k = blockIdx.x, dir = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  const int rl = tid & 255, q = tid >> 8;     // local gate row, column quarter
  const int g = rl >> 6, ul = rl & 63;
  const int row = g * ENC_H + CO_UNITS * k + ul;  // PyTorch gate order i,f,g,o
  const float *whhT = dir ? whhT_b : whhT_f;      // [256 cols][1024 rows]
  float w[64];
#pragma unroll
  for (int j = 0; j < 64; ++j) w[j] = whhT[(size_t)(64 * q + j) * (4 * ENC_H) + row];
  __shared__ float h[ENC_H], part[4][256];
  __shared__ int dead;
  const float *xp = xproj + ((size_t)dir * Btot + b0 + b) * T * (4 * ENC_H);
  gu64 *ex = (gu64 *)(exchange + ((size_t)dir * B + b) * 2 * ENC_H);
  float c = 0.f;
  if (tid < ENC_H) h[tid] = 0.f;
  if (tid == 0) dead = 0;
  __syncthreads();
  float xnext = q == 0 ? xp[(size_t)(dir ? T - 1 : 0) * (4 * ENC_H) + row] : 0.f;  // the input projection runs one step ahead of its use
  for (int s = 0; s < T; ++s) {
    const int t = dir ? T - 1 - s : s;
    const float xin = xnext;
    if (q == 0 && s + 1 < T) xnext = xp[(size_t)(dir ? T - 2 - s : s + 1) * (4 * ENC_H) + row];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int j = 0; j < 64; j += 4) {
      a0 = fmaf(w[j + 0], h[64 * q + j + 0], a0);
      a1 = fmaf(w[j + 1], h[64 * q + j + 1], a1);
      a2 = fmaf(w[j + 2], h[64 * q + j + 2], a2);
      a3 = fmaf(w[j + 3], h[64 * q + j + 3], a3);
    }
    part[q][rl] = ((a0 + a1) + (a2 + a3)) + xin;  // (xin: the input projection, held by the q = 0 quarter, 0 elsewhere)
    __syncthreads();
    if (tid < CO_UNITS) {
      // the four column quarters of the unit's four gate rows meet here (one barrier, no staging pass); hardware exp2 / rcp
      // forms as in the decoder engines
      float gs[4];
#pragma unroll
      for (int gi = 0; gi < 4; ++gi) gs[gi] = (part[0][64 * gi + tid] + part[1][64 * gi + tid]) + (part[2][64 * gi + tid] + part[3][64 * gi + tid]);
      const float ig = fast_sigmoid(gs[0]), fg = fast_sigmoid(gs[1]);
      const float gg = fast_tanh(gs[2]), og = fast_sigmoid(gs[3]);
      c = fmaf(fg, c, ig * gg);
      const float hn = og * fast_tanh(c);
      const int u = CO_UNITS * k + tid;
      h[u] = hn;
      memory[((size_t)(b0 + b) * T + t) * EMB + dir * ENC_H + u] = hn;
      if (s + 1 < T)  // publish: tag = step + 1 (never 0), one naturally aligned 8-byte store
        __hip_atomic_store(ex + (s & 1) * ENC_H + u, ((u64)(unsigned)(s + 1) << 32) | (u64)__float_as_uint(hn),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (tid >= 256 && tid < 256 + ENC_H && ((tid - 256) >> 6) != k && s + 1 < T) {
      // gather the peers' units (threads 256..511, one unit each; the cell-update threads are
      // busy publishing): re-read the granule until its tag is this step's
      const int u = tid - 256;
      u64 v = 0;
      unsigned spins = 0;
      const bool skip = dead != 0;
      while (!skip) {
        v = __hip_atomic_load(ex + (s & 1) * ENC_H + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(v >> 32) == (unsigned)(s + 1)) break;
        if (++spins > SPIN_LIMIT) {
          dead = 1;
          atomicExch(err, 1);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      h[u] = __uint_as_float((unsigned)v);
    }
    __syncthreads();
  }
}

}  // namespace

size_t bilstm_coop_exchange_words(int B) { return (size_t)2 * B * 2 * ENC_H; }

void launch_bilstm_coop(const float *xproj, const float *whhT_fwd, const float *whhT_bwd, float *memory,
                        unsigned long long *exchange, int *err, int B, int T, int group, hipStream_t s) {
  // 8 workgroups per chunk must be co-resident, so a large batch runs as launches of at most
  // `group` chunks each (sized to the CU count by the caller); the exchange buffer is reused
  for (int b0 = 0; b0 < B; b0 += group) {
    const int n = B - b0 < group ? B - b0 : group;
    // tags must start at 0 for every launch
    HIP_CHECK(hipMemsetAsync(exchange, 0, bilstm_coop_exchange_words(n) * sizeof(unsigned long long), s));
    HIP_CHECK(launch_coresident(true, reinterpret_cast<const void *>(k_bilstm_coop), dim3(CO_BLOCKS, 2, n), dim3(1024), 0, s, xproj,
                                whhT_fwd, whhT_bwd, memory, exchange, err, n, T, b0, B));
  }
}

}  // namespace xdtts
