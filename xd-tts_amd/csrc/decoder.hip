// decoder.hip -- the Tacotron2 autoregressive decoder step as HIP kernels for gfx950.
//
// Replaces the per-frame `self.decoder.run(inputs)` of the reference (src/tacotron2/mod.rs:304,
// graph decoder_iter.onnx) including the host-side stop test (mod.rs:319-324), which runs on the
// device here so the host is out of the loop.  B independent chunks advance in lock-step; a chunk
// is active at step s iff s < nframes[b].
//
// Per step, six kernels in stream order (each a grid-wide dependency of the next):
//   k_prenet  -> k_lstm<ATT> -> k_query -> k_attention -> k_lstm<DEC> -> k_project
// The two LSTM GEMVs stream 71.3 MB of fp32 weights per step and are the HBM-bound part; rows
// are packed [unit][gate][cols] so each wave reads one contiguous 4-row slab with 16-byte
// lane-consecutive loads (1 KiB per wave instruction) and owns a hidden unit end-to-end, which
// fuses the cell update into the GEMV.  Wavefront = 64 everywhere.
#include "kernels.h"

namespace xdtts {

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float dot4(float4 a, float4 b, float acc) {
  acc = fmaf(a.x, b.x, acc);
  acc = fmaf(a.y, b.y, acc);
  acc = fmaf(a.z, b.z, acc);
  acc = fmaf(a.w, b.w, acc);
  return acc;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// src/tacotron2/mod.rs:126-133: the two-branch sigmoid applied to the gate logit on the host
__device__ __forceinline__ float gate_sigmoid(float x) {
  if (x >= 0.0f) return 1.0f / (1.0f + expf(-x));
  const float e = expf(x);
  return e / (1.0f + e);
}
__device__ __forceinline__ bool any_active(const DecoderBufs &d, int step) {
  bool a = false;
  for (int b = 0; b < d.B; ++b) a |= step < d.nframes[b];
  return a;
}

// DecoderState::new (mod.rs:202-233): all recurrent state zero.
__global__ void k_decoder_init(DecoderBufs d, const int *limits) {
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < ATT_RNN; i += blockDim.x) {
    d.att_h[0][b * ATT_RNN + i] = 0.f;
    d.att_h[1][b * ATT_RNN + i] = 0.f;
    d.att_c[b * ATT_RNN + i] = 0.f;
    d.dec_h[0][b * DEC_RNN + i] = 0.f;
    d.dec_h[1][b * DEC_RNN + i] = 0.f;
    d.dec_c[b * DEC_RNN + i] = 0.f;
  }
  for (int i = threadIdx.x; i < d.T; i += blockDim.x) {
    d.aw[b * d.T + i] = 0.f;
    d.awc[b * d.T + i] = 0.f;
  }
  for (int i = threadIdx.x; i < EMB; i += blockDim.x) d.ctx[b * EMB + i] = 0.f;
  if (threadIdx.x == 0) {
    d.nframes[b] = limits[b];
    if (b == 0) {
      d.ctl[0] = 0;
      d.ctl[1] = 0;
    }
  }
}

// D1 prenet: x = relu(W1 relu(W0 mel_prev) * m0 * 2) * m1 * 2, no bias, Bernoulli(0.5) masks from
// the counter RNG (the exported graph keeps this dropout on at inference).  One block per chunk;
// weights are stored transposed so lanes read consecutive addresses.
__global__ __launch_bounds__(1024) void k_prenet(DecoderBufs d, const float *__restrict__ W0T,
                                                 const float *__restrict__ W1T) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const int step = d.ctl[0];
  if (step >= d.nframes[b]) return;
  __shared__ float mel[N_MEL], part[4][PRENET], x1[PRENET];
  if (tid < N_MEL)
    mel[tid] = step == 0 ? 0.f : d.frames[((size_t)b * d.max_steps + (step - 1)) * N_MEL + tid];
  __syncthreads();
  const int p = tid >> 8, j = tid & 255;
  const uint32_t item = d.item_base + (uint32_t)b;
  float acc = 0.f;
#pragma unroll 4
  for (int i = p * (N_MEL / 4); i < (p + 1) * (N_MEL / 4); ++i) acc = fmaf(W0T[i * PRENET + j], mel[i], acc);
  part[p][j] = acc;
  __syncthreads();
  if (tid < PRENET) {
    float v = (part[0][j] + part[1][j]) + (part[2][j] + part[3][j]);
    v = fmaxf(v, 0.f);
    if (d.dropout_mode)
      v = (rng_u32(d.dropout_seed, 0x1000u + 2u * item, (uint32_t)step * 256u + (uint32_t)j) >> 31) ? 0.f : 2.f * v;
    x1[j] = v;
  }
  __syncthreads();
  acc = 0.f;
#pragma unroll 8
  for (int i = p * (PRENET / 4); i < (p + 1) * (PRENET / 4); ++i) acc = fmaf(W1T[i * PRENET + j], x1[i], acc);
  part[p][j] = acc;
  __syncthreads();
  if (tid < PRENET) {
    float v = (part[0][j] + part[1][j]) + (part[2][j] + part[3][j]);
    v = fmaxf(v, 0.f);
    if (d.dropout_mode)
      v = (rng_u32(d.dropout_seed, 0x1001u + 2u * item, (uint32_t)step * 256u + (uint32_t)j) >> 31) ? 0.f : 2.f * v;
    d.x[b * PRENET + j] = v;
  }
}

// D2 / D4: LSTM cell as a weight-streaming GEMV with the cell update fused.  One wave owns one
// hidden unit: its four gate rows (i,f,g,o) are contiguous in the packed layout.  The rows are
// pulled into registers once (NCOLS/64 floats per lane per row) and reused for every chunk of
// the batch, so HBM sees each weight once per step regardless of B.
//   KIND 0: attention_rnn, input [prenet x (256) ; ctx_prev (512)] , hidden att_h   -> 1792 cols
//   KIND 1: decoder_rnn,   input [att_h_new (1024) ; ctx (512)]    , hidden dec_h   -> 2560 cols
template <int NCOLS, int KIND>
__global__ __launch_bounds__(256) void k_lstm(DecoderBufs d, const float4 *__restrict__ Wp,
                                              const float *__restrict__ bias) {
  constexpr int NCH = NCOLS / 256;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int unit = blockIdx.x * 4 + wave;
  const int step = d.ctl[0], cur = step & 1;
  if (!any_active(d, step)) return;
  float4 w[4][NCH];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int k = 0; k < NCH; ++k) w[g][k] = Wp[((size_t)(unit * 4 + g) * NCOLS) / 4 + lane + 64 * k];
  const float4 bz = *reinterpret_cast<const float4 *>(bias + unit * 4);
  for (int b = 0; b < d.B; ++b) {
    if (step >= d.nframes[b]) continue;
    const float *seg0, *seg1, *seg2;
    float *h_out, *c;
    if (KIND == 0) {
      seg0 = d.x + b * PRENET;
      seg1 = d.ctx + b * EMB;
      seg2 = d.att_h[cur] + b * ATT_RNN;
      h_out = d.att_h[cur ^ 1] + b * ATT_RNN;
      c = d.att_c + b * ATT_RNN;
    } else {
      seg0 = d.att_h[cur ^ 1] + b * ATT_RNN;
      seg1 = d.ctx + b * EMB;
      seg2 = d.dec_h[cur] + b * DEC_RNN;
      h_out = d.dec_h[cur ^ 1] + b * DEC_RNN;
      c = d.dec_c + b * DEC_RNN;
    }
    constexpr int N0 = KIND == 0 ? PRENET : ATT_RNN;  // first segment length
    constexpr int N1 = EMB;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int col = 256 * k;  // wave-uniform segment choice: boundaries are multiples of 256
      const float *src = col < N0 ? seg0 + col : (col < N0 + N1 ? seg1 + (col - N0) : seg2 + (col - N0 - N1));
      const float4 xv = *reinterpret_cast<const float4 *>(src + 4 * lane);
      a0 = dot4(w[0][k], xv, a0);
      a1 = dot4(w[1][k], xv, a1);
      a2 = dot4(w[2][k], xv, a2);
      a3 = dot4(w[3][k], xv, a3);
    }
    a0 = wave_sum(a0);
    a1 = wave_sum(a1);
    a2 = wave_sum(a2);
    a3 = wave_sum(a3);
    if (lane == 0) {
      const float ig = sigmoidf_(a0 + bz.x), fg = sigmoidf_(a1 + bz.y);
      const float gg = tanhf(a2 + bz.z), og = sigmoidf_(a3 + bz.w);
      const float cn = fmaf(fg, c[unit], ig * gg);
      c[unit] = cn;
      h_out[unit] = og * tanhf(cn);
    }
  }
}

// D3a: processed query q = W_q att_h_new (128 x 1024, no bias); one wave per row.
__global__ __launch_bounds__(256) void k_query(DecoderBufs d, const float4 *__restrict__ Wq) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int step = d.ctl[0], cur = step & 1;
  if (!any_active(d, step)) return;
  float4 w[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) w[k] = Wq[(size_t)row * (ATT_RNN / 4) + lane + 64 * k];
  for (int b = 0; b < d.B; ++b) {
    if (step >= d.nframes[b]) continue;
    const float *h = d.att_h[cur ^ 1] + b * ATT_RNN;
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) a = dot4(w[k], *reinterpret_cast<const float4 *>(h + 256 * k + 4 * lane), a);
    a = wave_sum(a);
    if (lane == 0) d.q[b * ATT_DIM + row] = a;
  }
}

// D3b: location-sensitive attention for one chunk per block (16 waves):
//   loc = Dense32->128(Conv1d(2->32,k=31,pad=15)([w_prev ; w_cum]))
//   e_t = v . tanh(q + loc_t + processed_memory_t), -inf where t >= n_valid   (mask, mod.rs:219-220)
//   w = softmax_t(e); ctx = sum_t w_t memory_t; w_cum += w
// memory/processed_memory rows are read with lane-consecutive addresses; the weights, the
// energies and the context partials live in LDS.
__global__ __launch_bounds__(1024) void k_attention(DecoderBufs d, const float *__restrict__ v_w,
                                                    const float *__restrict__ loc_conv,
                                                    const float *__restrict__ loc_denseT) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int step = d.ctl[0];
  if (step >= d.nframes[b]) return;
  const int T = d.T, PADK = (LOC_K - 1) / 2, TP = T + 2 * PADK;
  float *s_aw = smem;                     // [TP] zero-padded previous weights
  float *s_awc = s_aw + TP;               // [TP] zero-padded cumulative weights
  float *s_q = s_awc + TP;                // [128]
  float *s_v = s_q + ATT_DIM;             // [128]
  float *s_cw = s_v + ATT_DIM;            // [32*2*31]
  float *s_wd = s_cw + LOC_F * 2 * LOC_K; // [32][128]
  float *s_lc = s_wd + LOC_F * ATT_DIM;   // [T][33]
  float *s_e = s_lc + T * (LOC_F + 1);    // [T]
  float *s_part = s_e + T;                // [512]
  for (int i = tid; i < TP; i += 1024) {
    const int t = i - PADK;
    const bool in = t >= 0 && t < T;
    s_aw[i] = in ? d.aw[b * T + t] : 0.f;
    s_awc[i] = in ? d.awc[b * T + t] : 0.f;
  }
  if (tid < ATT_DIM) {
    s_q[tid] = d.q[b * ATT_DIM + tid];
    s_v[tid] = v_w[tid];
  }
  for (int i = tid; i < LOC_F * 2 * LOC_K; i += 1024) s_cw[i] = loc_conv[i];
  for (int i = tid; i < LOC_F * ATT_DIM; i += 1024) s_wd[i] = loc_denseT[i];
  __syncthreads();
  // location conv: (t, f) outputs; channel 0 = previous weights, channel 1 = cumulative
  for (int o = tid; o < T * LOC_F; o += 1024) {
    const int t = o / LOC_F, f = o % LOC_F;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < LOC_K; ++k) acc = fmaf(s_cw[(f * 2 + 0) * LOC_K + k], s_aw[t + k], acc);
#pragma unroll
    for (int k = 0; k < LOC_K; ++k) acc = fmaf(s_cw[(f * 2 + 1) * LOC_K + k], s_awc[t + k], acc);
    s_lc[t * (LOC_F + 1) + f] = acc;
  }
  __syncthreads();
  // energies: one wave per time step, lane covers attention dims lane and lane+64
  const int nv = d.n_valid[b];
  const float *pm = d.pmem + (size_t)b * T * ATT_DIM;
  for (int t = wave; t < T; t += 16) {
    float l0 = 0.f, l1 = 0.f;
#pragma unroll
    for (int f = 0; f < LOC_F; ++f) {
      const float c = s_lc[t * (LOC_F + 1) + f];
      l0 = fmaf(s_wd[f * ATT_DIM + lane], c, l0);
      l1 = fmaf(s_wd[f * ATT_DIM + lane + 64], c, l1);
    }
    float e = s_v[lane] * tanhf(s_q[lane] + l0 + pm[t * ATT_DIM + lane]) +
              s_v[lane + 64] * tanhf(s_q[lane + 64] + l1 + pm[t * ATT_DIM + lane + 64]);
    e = wave_sum(e);
    if (lane == 0) s_e[t] = t >= nv ? -INFINITY : e;
  }
  __syncthreads();
  // softmax over t by wave 0
  if (wave == 0) {
    float m = -INFINITY;
    for (int t = lane; t < T; t += 64) m = fmaxf(m, s_e[t]);
    m = wave_max(m);
    float sum = 0.f;
    for (int t = lane; t < T; t += 64) {
      const float ex = expf(s_e[t] - m);
      s_e[t] = ex;
      sum += ex;
    }
    sum = wave_sum(sum);
    for (int t = lane; t < T; t += 64) s_e[t] = s_e[t] / sum;
  }
  __syncthreads();
  for (int t = tid; t < T; t += 1024) {
    const float wv = s_e[t];
    d.aw[b * T + t] = wv;
    d.awc[b * T + t] = s_awc[t + PADK] + wv;
  }
  // context: thread (half, c) accumulates its half of the time axis for column c
  const int c = tid & 511, half = tid >> 9;
  const float *mem = d.memory + (size_t)b * T * EMB;
  const int t0 = half ? (T + 1) / 2 : 0, t1 = half ? T : (T + 1) / 2;
  float acc = 0.f;
#pragma unroll 4
  for (int t = t0; t < t1; ++t) acc = fmaf(s_e[t], mem[(size_t)t * EMB + c], acc);
  if (half) s_part[c] = acc;
  __syncthreads();
  if (!half) d.ctx[b * EMB + c] = acc + s_part[c];
}

// D5 + D6: mel = W_p [dec_h ; ctx] + b_p (80 rows), gate = W_g [dec_h ; ctx] + b_g (row 80), and
// the stop rule of mod.rs:319-324 (sigmoid(gate) > threshold, the tripping frame is kept) applied
// on the device: the gate wave lowers nframes[b] to step+1.  The last block to finish advances
// the step counter (all blocks have read it by then).
__global__ __launch_bounds__(256) void k_project(DecoderBufs d, const float4 *__restrict__ Wp,
                                                 const float *__restrict__ bias) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int step = d.ctl[0], cur = step & 1;
  if (row <= N_MEL && any_active(d, step)) {
    float4 w[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) w[k] = Wp[(size_t)row * (PROJ_IN / 4) + lane + 64 * k];
    const float bz = bias[row];
    for (int b = 0; b < d.B; ++b) {
      if (step >= d.nframes[b]) continue;
      const float *h = d.dec_h[cur ^ 1] + b * DEC_RNN, *cx = d.ctx + b * EMB;
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) a = dot4(w[k], *reinterpret_cast<const float4 *>(h + 256 * k + 4 * lane), a);
#pragma unroll
      for (int k = 0; k < 2; ++k) a = dot4(w[4 + k], *reinterpret_cast<const float4 *>(cx + 256 * k + 4 * lane), a);
      a = wave_sum(a) + bz;
      if (lane == 0) {
        if (row < N_MEL) {
          d.frames[((size_t)b * d.max_steps + step) * N_MEL + row] = a;
        } else {
          d.gates[(size_t)b * d.max_steps + step] = a;
          if (d.use_gate && gate_sigmoid(a) > d.gate_threshold) d.nframes[b] = step + 1;
        }
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    const int t = atomicAdd(&d.ctl[1], 1);
    if (t == (int)gridDim.x - 1) {
      d.ctl[1] = 0;
      d.ctl[0] = step + 1;
    }
  }
}

}  // namespace

size_t attention_lds_bytes(int T) {
  const int TP = T + (LOC_K - 1);
  return sizeof(float) * (size_t)(2 * TP + 2 * ATT_DIM + LOC_F * 2 * LOC_K + LOC_F * ATT_DIM + T * (LOC_F + 1) + T + EMB);
}

void launch_decoder_init(const DecoderBufs &d, const int *limits_dev, hipStream_t s) {
  hipLaunchKernelGGL(k_decoder_init, dim3(d.B), dim3(256), 0, s, d, limits_dev);
  HIP_CHECK(hipGetLastError());
}

void launch_decoder_steps(const DecoderBufs &d, const DeviceWeights &w, int nsteps, hipStream_t s) {
  const size_t lds = attention_lds_bytes(d.T);
  for (int i = 0; i < nsteps; ++i) {
    hipLaunchKernelGGL(k_prenet, dim3(d.B), dim3(1024), 0, s, d, w.pre0T.p, w.pre1T.p);
    hipLaunchKernelGGL((k_lstm<ATT_COLS, 0>), dim3(ATT_RNN / 4), dim3(256), 0, s, d,
                       reinterpret_cast<const float4 *>(w.att_w.p), w.att_b.p);
    hipLaunchKernelGGL(k_query, dim3(ATT_DIM / 4), dim3(256), 0, s, d, reinterpret_cast<const float4 *>(w.q_w.p));
    hipLaunchKernelGGL(k_attention, dim3(d.B), dim3(1024), lds, s, d, w.v_w.p, w.loc_conv.p, w.loc_denseT.p);
    hipLaunchKernelGGL((k_lstm<DEC_COLS, 1>), dim3(DEC_RNN / 4), dim3(256), 0, s, d,
                       reinterpret_cast<const float4 *>(w.dec_w.p), w.dec_b.p);
    hipLaunchKernelGGL(k_project, dim3((N_MEL + 1 + 3) / 4), dim3(256), 0, s, d,
                       reinterpret_cast<const float4 *>(w.proj_w.p), w.proj_b.p);
  }
  HIP_CHECK(hipGetLastError());
}

}  // namespace xdtts
