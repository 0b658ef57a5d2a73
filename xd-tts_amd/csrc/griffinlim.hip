// griffinlim.hip -- Griffin-Lim phase recovery as HIP kernels for gfx950.
//
// Replaces GriffinLim::infer of the un-vendored `griffin-lim` crate (called at src/lib.rs:141,
// parameters fixed at src/tacotron2/mod.rs:453-456: n_fft 1024, hop 256, 30 iterations, momentum
// 0.99).  One iteration is  inverse = ISTFT(S*angles); rebuilt = STFT(inverse);
// angles = normalise(rebuilt - m/(1+m) * previous rebuilt).
//
// Framed FFTs: one 64-lane wavefront per frame.  A 1024-point real transform is done as a
// 512-point complex Stockham FFT, radix 8 x 3 passes, 8 points per lane held in registers, with
// LDS used only for the two inter-pass exchanges (conflict-free: lanes touch consecutive float2).
// The real<->complex split/merge twiddle step is fused into the first/last pass, the synthesis
// and analysis windows and the magnitude projection are fused into the loads/stores, so one
// iteration touches S, angles and the previous rebuilt spectrum exactly once each.  An iteration is
// ONE launch (k_gl_fused): the windowed ISTFT frames of a tile (+ halo) live in LDS and the
// overlap-add is folded into the STFT's gather.
#include <algorithm>
#include <utility>

#include "kernels.h"

namespace xdtts {

namespace {

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }  // * (-i)

// forward 8-point DFT, natural-order output
__device__ __forceinline__ void fft8(float2 (&v)[8]) {
  const float h = 0.70710678118654752440f;
  float2 b0 = cadd(v[0], v[4]), b4 = csub(v[0], v[4]);
  float2 b1 = cadd(v[1], v[5]), b5 = csub(v[1], v[5]);
  float2 b2 = cadd(v[2], v[6]), b6 = csub(v[2], v[6]);
  float2 b3 = cadd(v[3], v[7]), b7 = csub(v[3], v[7]);
  b5 = make_float2(h * (b5.x + b5.y), h * (b5.y - b5.x));   // * (1-i)/sqrt2
  b6 = mul_mi(b6);                                           // * (-i)
  b7 = make_float2(h * (b7.y - b7.x), -h * (b7.x + b7.y));  // * (-1-i)/sqrt2
  float2 d0 = cadd(b0, b2), d1 = csub(b0, b2), d2 = cadd(b1, b3), d3 = mul_mi(csub(b1, b3));
  v[0] = cadd(d0, d2);
  v[4] = csub(d0, d2);
  v[2] = cadd(d1, d3);
  v[6] = csub(d1, d3);
  d0 = cadd(b4, b6);
  d1 = csub(b4, b6);
  d2 = cadd(b5, b7);
  d3 = mul_mi(csub(b5, b7));
  v[1] = cadd(d0, d2);
  v[5] = csub(d0, d2);
  v[3] = cadd(d1, d3);
  v[7] = csub(d1, d3);
}

// Per-lane twiddles of the two twiddled passes, pulled from the table once at kernel entry (in
// the same memory round trip as the frame's data) so the FFT itself never waits on global memory.
struct Twiddles {
  float2 p8[7];   // pass Ns = 8 : e^{-2 pi i r k / 64},  k = lane & 7  -> table index r*k*16
  float2 p64[7];  // pass Ns = 64: e^{-2 pi i r k / 512}, k = lane      -> table index r*k*2
};
__device__ __forceinline__ Twiddles load_twiddles(const float2 *__restrict__ tw, int lane) {
  Twiddles t;
  const int k = lane & 7;
#pragma unroll
  for (int r = 1; r < 8; ++r) {
    t.p8[r - 1] = tw[r * k * 16];
    t.p64[r - 1] = tw[r * lane * 2];
  }
  return t;
}

// 512-point forward complex FFT of one wave.  In: v[r] = x[lane + 64 r].  Runs Stockham passes
// Ns = 1 and 8 through `buf` (512 float2 of LDS owned by this wave) and the twiddle + butterfly
// of pass Ns = 64; on return v[r] = X[lane + 64 r] (natural order), nothing left in LDS.
// buf is private to the calling wave, so the exchanges only need wave-level ordering.
// Orders a wave's LDS stores before its later LDS loads of other lanes' addresses.  DS operations
// of one wave execute in order, so only the compiler has to be held back; waves stay decoupled.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void fft512(float2 (&v)[8], float2 *buf, const Twiddles &t, int lane) {
  // pass Ns = 1: no twiddles; out[8 j + r]
  fft8(v);
#pragma unroll
  for (int r = 0; r < 8; ++r) buf[8 * lane + r] = v[r];
  wave_lds_sync();
  // pass Ns = 8
  {
    const int k = lane & 7;
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = buf[lane + 64 * r];
#pragma unroll
    for (int r = 1; r < 8; ++r) v[r] = cmul(v[r], t.p8[r - 1]);
    fft8(v);
    wave_lds_sync();
    const int j0 = (lane >> 3) * 64 + k;
#pragma unroll
    for (int r = 0; r < 8; ++r) buf[j0 + 8 * r] = v[r];
  }
  wave_lds_sync();
  // pass Ns = 64; out[j + 64 r]
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = buf[lane + 64 * r];
#pragma unroll
  for (int r = 1; r < 8; ++r) v[r] = cmul(v[r], t.p64[r - 1]);
  fft8(v);
}

// The same transform with the exchange buffer addressed through phys(i) = i + (i >> 4) (544 float2
// per wave): the stride-8 stores of pass 1 (float2 slots 8 lane + r: two distinct 16-slot bank
// positions per 16-lane group, an 8-way conflict) and the stride-64 lane groups of pass 2 become
// conflict-free; all offsets stay compile-time per r.  Used where one wave per SIMD runs the
// transform and every LDS cycle is exposed latency (the persistent kernel).
constexpr int FFT_SWZ_F2 = 544;  // (the transform itself: fft512sp below)

// ---- the same arithmetic on (re, im) register PAIRS ----------------------------------------------------------------------
// The persistent kernel runs ONE wave per SIMD, so its iteration is a chain of single-wave VALU issues (4 clocks each): the
// instruction count is the time.  Left to itself the compiler pairs scalars of DIFFERENT complex numbers into v_pk_* operations
// and pays for it with register shuffles (288 v_mov in 2 500 instructions).  Here a complex number is one 64-bit register pair
// and every complex operation is one or two packed instructions whose swaps, conjugations and multiplications by -i are operand
// modifiers (op_sel / neg_lo / neg_hi).  Each operation is the IEEE operation the float2 forms above spell out (negations and
// the factors 1/2 are exact); what differs from a build of those forms is only where the compiler contracts a*b+c.  Measured in
// one session on one box: 4.87 -> 4.61 us per iteration at F = 1000, 251 VGPRs and no AGPR copies instead of 256 + 51.
typedef float f2 __attribute__((ext_vector_type(2)));
#define XD_PK2(name, op, mods)                                            \
  __device__ __forceinline__ f2 name(f2 a, f2 b) {                        \
    f2 r;                                                                 \
    asm(op " %0, %1, %2 " mods : "=v"(r) : "v"(a), "v"(b));               \
    return r;                                                             \
  }
XD_PK2(pk_add_mi, "v_pk_add_f32", "op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")    // a + (-i) b = (a.x + b.y, a.y - b.x)
XD_PK2(pk_sub_mi, "v_pk_add_f32", "op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]")    // a - (-i) b = (a.x - b.y, a.y + b.x)
XD_PK2(pk_add_conj, "v_pk_add_f32", "neg_hi:[0,1]")                               // a + conj b = (a.x + b.x, a.y - b.y)
XD_PK2(pk_sub_conj, "v_pk_add_f32", "neg_lo:[0,1]")                               // a - conj b = (a.x - b.x, a.y + b.y)
XD_PK2(pk_conj_sub, "v_pk_add_f32", "neg_lo:[0,1] neg_hi:[1,0]")                  // conj(a - b) = (a.x - b.x, b.y - a.y)
XD_PK2(pk_conj_sub_mi, "v_pk_add_f32", "op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[1,1]")  // conj(a - (-i) b) = (a.x - b.y, -a.y - b.x)
XD_PK2(pk_odd, "v_pk_add_f32", "op_sel:[1,1] op_sel_hi:[0,0] neg_hi:[1,0]")       // (a.y + b.y, b.x - a.x)
XD_PK2(pk_mul_conj, "v_pk_mul_f32", "neg_hi:[0,1]")                               // (a.x b.x, -a.y b.y)
__device__ __forceinline__ f2 pk_rot7(f2 a) {  // (a.y - a.x, -a.x - a.y) = a (-1 - i)
  f2 r;
  asm("v_pk_add_f32 %0, %1, %1 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[1,1]" : "=v"(r) : "v"(a));
  return r;
}
__device__ __forceinline__ f2 pk_cmul(f2 a, f2 b) {  // a b:  (fma(a.x, b.x, -(a.y b.y)), fma(a.x, b.y, a.y b.x)) -- cmul()
  f2 r;  // (one asm statement: between two of them the hazard recogniser puts a wait state it cannot rule out)
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\tv_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]"
      : "=&v"(r)
      : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ f2 pk_cmul_conj(f2 a, f2 b) {  // a conj(b) -- cmul(a, cconj(b))
  f2 r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]\n\tv_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1] neg_hi:[0,1,0]"
      : "=&v"(r)
      : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ f2 as_f2(float2 v) { return (f2){v.x, v.y}; }

// fft8 on register pairs: 28 packed instructions (16 + 8 butterflies, 4 for the two odd eighth-turn rotations; the
// multiplications by -i are folded into the butterflies that consume them)
__device__ __forceinline__ void fft8p(f2 (&v)[8]) {
  const float h = 0.70710678118654752440f;
  const f2 b0 = v[0] + v[4], b4 = v[0] - v[4];
  const f2 b1 = v[1] + v[5];
  f2 b5 = v[1] - v[5];
  const f2 b2 = v[2] + v[6], b6 = v[2] - v[6];  // (b6 enters as -i b6 below)
  const f2 b3 = v[3] + v[7];
  f2 b7 = v[3] - v[7];
  b5 = pk_add_mi(b5, b5) * h;  // (b5.x + b5.y, b5.y - b5.x) h = b5 (1 - i) / sqrt 2
  b7 = pk_rot7(b7) * h;        // b7 (-1 - i) / sqrt 2
  f2 d0 = b0 + b2, d1 = b0 - b2, d2 = b1 + b3, t = b1 - b3;
  v[0] = d0 + d2;
  v[4] = d0 - d2;
  v[2] = pk_add_mi(d1, t);
  v[6] = pk_sub_mi(d1, t);
  d0 = pk_add_mi(b4, b6);
  d1 = pk_sub_mi(b4, b6);
  d2 = b5 + b7;
  t = b5 - b7;
  v[1] = d0 + d2;
  v[5] = d0 - d2;
  v[3] = pk_add_mi(d1, t);
  v[7] = pk_sub_mi(d1, t);
}
struct TwiddlesP {
  f2 p8[7], p64[7];
};
__device__ __forceinline__ TwiddlesP pack_twiddles(const Twiddles &t) {
  TwiddlesP q;
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    q.p8[r] = as_f2(t.p8[r]);
    q.p64[r] = as_f2(t.p64[r]);
  }
  return q;
}
// fft512 on register pairs, the exchange buffer addressed through phys(i) = i + (i >> 4) as described above
__device__ __forceinline__ void fft512sp(f2 (&v)[8], f2 *buf, const TwiddlesP &t, int lane) {
  fft8p(v);
  {
    f2 *w = buf + 8 * lane + (lane >> 1);
#pragma unroll
    for (int r = 0; r < 8; ++r) w[r] = v[r];
  }
  wave_lds_sync();
  const f2 *rd = buf + lane + (lane >> 4);
  {
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = rd[68 * r];
#pragma unroll
    for (int r = 1; r < 8; ++r) v[r] = pk_cmul(v[r], t.p8[r - 1]);
    fft8p(v);
    wave_lds_sync();
    const int j0 = (lane >> 3) * 64 + (lane & 7);
    f2 *w = buf + j0 + (j0 >> 4);
#pragma unroll
    for (int r = 0; r < 8; ++r) w[8 * r + (r >> 1)] = v[r];
  }
  wave_lds_sync();
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = rd[68 * r];
#pragma unroll
  for (int r = 1; r < 8; ++r) v[r] = pk_cmul(v[r], t.p64[r - 1]);
  fft8p(v);
}

constexpr int FRAMES_PER_BLOCK = 4;  // one wave per frame

// Spectrum of frame f -> the packed 512-point input of the inverse transform, in registers:
// X = S * angles (513 bins); Hermitian merge E/O; Z = E + i O; returns conj(Z)[lane + 64 r]
// (inverse FFT = conj(FFT(conj Z)) / 512).
__device__ __forceinline__ void istft_load(const GlBufs &g, const float2 *__restrict__ ang, int f, int lane,
                                           float2 (&v)[8]) {
  const float *S = g.S + (size_t)f * g.nb;
  const float2 *A = ang + (size_t)f * g.nb;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int k = lane + 64 * r, kc = 512 - k;
    float2 xk = A[k], xc = A[kc];
    const float sk = S[k], sc = S[kc];
    xk = make_float2(xk.x * sk, xk.y * sk);
    xc = make_float2(xc.x * sc, xc.y * sc);
    if (k == 0) {  // irfft ignores the imaginary part of DC and Nyquist
      xk.y = 0.f;
      xc.y = 0.f;
    }
    const float2 e = make_float2(0.5f * (xk.x + xc.x), 0.5f * (xk.y - xc.y));   // (X[k] + conj X[512-k]) / 2
    const float2 d = make_float2(0.5f * (xk.x - xc.x), 0.5f * (xk.y + xc.y));   // (X[k] - conj X[512-k]) / 2
    const float2 o = cmul(d, cconj(g.tw[k]));                                   // * e^{+2 pi i k / 1024}
    v[r] = make_float2(e.x - o.y, -(e.y + o.x));
  }
}

// ISTFT, per-frame half: irfft(1024) of S * angles -> synthesis window -> frames[f][1024].
__global__ __launch_bounds__(256) void k_istft_frames(GlBufs g, const float2 *__restrict__ ang) {
  __shared__ float2 lds[FRAMES_PER_BLOCK][512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = blockIdx.x * FRAMES_PER_BLOCK + wave;
  const bool ok = fr < g.F;
  const int f = ok ? fr : g.F - 1;
  const Twiddles tws = load_twiddles(g.tw, lane);
  float2 v[8];
  istft_load(g, ang, f, lane, v);
  fft512(v, lds[wave], tws, lane);
  if (!ok) return;
  float2 *out = reinterpret_cast<float2 *>(g.frames + (size_t)f * g.n_fft);
  const float2 *win = reinterpret_cast<const float2 *>(g.win);
  const float sc = 1.0f / 512.0f;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int m = lane + 64 * r;
    const float2 w = win[m];
    out[m] = make_float2(v[r].x * sc * w.x, -v[r].y * sc * w.y);  // x[2m] = Re z, x[2m+1] = Im z
  }
}

// One sample of the ISTFT output (overlap-add of the windowed frames, window-sum-square
// normalised, centre-trimmed) gathered straight from the per-frame buffer: at hop = n_fft/4 four
// frames cover a sample.  wss_inv is precomputed per F.
constexpr int NFFT = 1024, HOP = 256;  // the reference's vocoder geometry (mod.rs:453-456); checked at GriffinLim::new

__device__ __forceinline__ float ola_sample(const GlBufs &g, int n) {
  // hop = n_fft/4: sample q of the un-trimmed signal lies in frames jb-3..jb at offsets
  // (q mod hop) + k*hop; frames outside [0, F) contribute nothing.  Branch-free: four independent
  // loads so a lane's 16 samples keep 64 loads in flight.  Shifts, not divisions.
  const int q = n + NFFT / 2;
  const int jb = q >> 8, r = q & (HOP - 1);
  float v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int j = jb - k;
    const bool in = j >= 0 && j < g.F;
    v[k] = g.frames[(size_t)(in ? j : 0) * NFFT + r + k * HOP];
    v[k] = in ? v[k] : 0.f;
  }
  // ascending frame order and a true division, like the CPU path, so rounding matches it
  return (((v[3] + v[2]) + v[1]) + v[0]) / g.wss_inv[n];
}

// numpy "reflect" padding index: mirror without repeating the edge sample; one fold for normal
// sizes, a few for signals shorter than the pad
__device__ __forceinline__ int reflect_index(int p, int N) {
  while (p < 0 || p >= N) p = p < 0 ? -p : 2 * (N - 1) - p;
  return p;
}

// the same for |overhang| < N (signals longer than the pad): branch-free
__device__ __forceinline__ int reflect_once(int p, int N) {
  p = p < 0 ? -p : p;
  return p >= N ? 2 * (N - 1) - p : p;
}

// window sum-of-squares divisor per output sample (1 where the sum is below tiny, like librosa)
__global__ void k_wss_inv(GlBufs g) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= g.hop * (g.F - 1)) return;
  const int q = n + NFFT / 2;
  const int jb = q >> 8, r = q & (HOP - 1);
  float wss = 0.f;
#pragma unroll
  for (int k = 3; k >= 0; --k) {
    const int j = jb - k;
    const float w = g.win[r + k * HOP];
    wss = (j >= 0 && j < g.F) ? wss + w * w : wss;
  }
  g.wss_inv[n] = wss > 1.17549435e-38f ? wss : 1.0f;  // the divisor (1 where the sum is below tiny)
}

// Packed 512-point spectrum Z (in LDS) of a real 1024-sample frame -> its 513 bins X, then the
// Griffin-Lim phase update:  a = X - alpha * tprev;  tprev = X;  angles = a / (|a| + 1e-16).
// The caller fetches tprev early (tprev_load) so its latency hides behind the transforms.
__device__ __forceinline__ void tprev_load(const GlBufs &g, int f, int lane, float2 (&pv)[9]) {
  const float2 *tp = g.tprev + (size_t)f * g.nb;
#pragma unroll
  for (int r = 0; r < 8; ++r) pv[r] = tp[lane + 64 * r];
  pv[8] = tp[512];
}

__device__ __forceinline__ void stft_update_store(const GlBufs &g, float2 *__restrict__ ang_out, int f, int lane,
                                                  const float2 *buf, float alpha, const float2 (&pvs)[9]) {
  float2 *ang = ang_out + (size_t)f * g.nb;
  float2 *tp = g.tprev + (size_t)f * g.nb;
#pragma unroll
  for (int r = 0; r <= 8; ++r) {
    const int k = lane + 64 * r;
    if (r == 8 && lane != 0) break;
    const float2 zk = buf[k & 511], zc = buf[(512 - k) & 511];
    const float2 e = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y - zc.y));  // (Z[k] + conj Z[512-k]) / 2
    const float2 o = make_float2(0.5f * (zk.y + zc.y), 0.5f * (zc.x - zk.x));  // (Z[k] - conj Z[512-k]) / (2i)
    const float2 twk = k == 512 ? make_float2(-1.f, 0.f) : g.tw[k];
    const float2 x = cadd(e, cmul(twk, o));
    const float2 pv = pvs[r];
    tp[k] = x;
    const float2 a = make_float2(fmaf(-alpha, pv.x, x.x), fmaf(-alpha, pv.y, x.y));
    const float mag = sqrtf(fmaf(a.x, a.x, a.y * a.y)) + 1e-16f;
    ang[k] = make_float2(a.x / mag, a.y / mag);
  }
}

// STFT + phase update, one wave per frame: the frame's 1024 samples are gathered from the
// overlap-added ISTFT frames (reflect padding at the ends), analysis window, rfft(1024), then
//   a = rebuilt - alpha * tprev;  tprev = rebuilt;  angles = a / (|a| + 1e-16).
__global__ __launch_bounds__(256) void k_stft_update(GlBufs g, float alpha) {
  __shared__ float2 lds[FRAMES_PER_BLOCK][512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = blockIdx.x * FRAMES_PER_BLOCK + wave;
  const bool ok = fr < g.F;
  const int f = ok ? fr : g.F - 1;
  const int N = g.hop * (g.F - 1);
  const float2 *win = reinterpret_cast<const float2 *>(g.win);
  const Twiddles tws = load_twiddles(g.tw, lane);
  float2 v[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int m = lane + 64 * r;
    const int base = f * HOP + 2 * m - NFFT / 2;
    const int p0 = reflect_index(base, N), p1 = reflect_index(base + 1, N);
    const float2 w = win[m];
    v[r] = make_float2(ola_sample(g, p0) * w.x, ola_sample(g, p1) * w.y);
  }
  float2 *buf = lds[wave];
  fft512(v, buf, tws, lane);
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 8; ++r) buf[lane + 64 * r] = v[r];
  __syncthreads();
  if (!ok) return;
  float2 pv[9];
  tprev_load(g, f, lane, pv);
  stft_update_store(g, g.ang, f, lane, buf, alpha, pv);
}

// One whole Griffin-Lim iteration in ONE launch.  A block owns TF consecutive frames; it inverse-
// transforms those plus a halo of three frames either side (the frames whose windows overlap its
// samples at hop = n_fft/4) into LDS, overlap-adds them there, and forward-transforms its own TF
// frames from LDS -- the ISTFT frame buffer never leaves the CU and an iteration costs one launch
// boundary instead of two.  The halo is recomputed by the neighbouring block too ((TF+6)/TF more
// inverse FFTs), and because neighbours read this block's angles while it writes new ones the
// angles ping-pong between two buffers.  Rounding is identical to the two-kernel path (same FFT,
// ascending-frame sums, true division by the window sum-square).
template <int TF>
__global__ __launch_bounds__(64 * (TF + 6)) void k_gl_fused(GlBufs g, const float2 *__restrict__ ang_in,
                                                            float2 *__restrict__ ang_out, float alpha) {
  constexpr int NSLOT = TF + 6;  // = waves per block: one inverse transform each
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *fb = smem;                                                   // [NSLOT][1024] windowed frames
  float2 *scratch = reinterpret_cast<float2 *>(fb + NSLOT * NFFT);    // [NSLOT waves][512] FFT exchange
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int f0 = blockIdx.x * TF, N = HOP * (g.F - 1);
  const Twiddles tws = load_twiddles(g.tw, lane);
  const float2 *win = reinterpret_cast<const float2 *>(g.win);
  float2 *buf = scratch + wave * 512;
  // waves 0..TF-1 also own frame f0+wave in phase 2: fetch what that needs from HBM now
  const int f2 = f0 + wave;
  const bool own = wave < TF && f2 < g.F;
  float2 pv[9];
  float wss[16];
  if (own) {
    tprev_load(g, f2, lane, pv);
#pragma unroll
    for (int i = 0; i < 16; ++i)
      wss[i] = g.wss_inv[reflect_index(f2 * HOP + 2 * (lane + 64 * (i >> 1)) + (i & 1) - NFFT / 2, N)];
  }
  // phase 1: wave w inverse-transforms frame f0-3+w (zeros outside [0, F))
  {
    const int f = f0 - 3 + wave;
    const bool ok = f >= 0 && f < g.F;
    float2 *out = reinterpret_cast<float2 *>(fb + wave * NFFT);
    if (ok) {
      float2 v[8];
      istft_load(g, ang_in, f, lane, v);
      fft512(v, buf, tws, lane);
      const float sc = 1.0f / 512.0f;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int m = lane + 64 * r;
        const float2 w = win[m];
        out[m] = make_float2(v[r].x * sc * w.x, -v[r].y * sc * w.y);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 8; ++r) out[lane + 64 * r] = make_float2(0.f, 0.f);
    }
  }
  __syncthreads();
  // phase 2: waves 0..TF-1 forward-transform the block's own frames, samples gathered from LDS
  if (!own) return;
  const int f = f2;
  float2 v[8];
  if (f >= 2 && f <= g.F - 3) {
    // no reflection: sample pair (2m, 2m+1), m = lane + 64 r, sits in frames f+(r>>1)-k, k = 0..3,
    // at float2 offset lane + 64 (r&1) + 128 k -- all compile-time offsets from one base
    const float2 *fb2 = reinterpret_cast<const float2 *>(fb) + (wave + 3) * (NFFT / 2) + lane;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      float2 t[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) t[k] = fb2[((r >> 1) - k) * (NFFT / 2) + 64 * (r & 1) + 128 * k];
      const float y0 = (((t[3].x + t[2].x) + t[1].x) + t[0].x) / wss[2 * r];
      const float y1 = (((t[3].y + t[2].y) + t[1].y) + t[0].y) / wss[2 * r + 1];
      const float2 w = win[lane + 64 * r];
      v[r] = make_float2(y0 * w.x, y1 * w.y);
    }
  } else {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int m = lane + 64 * r;
      const int base = f * HOP + 2 * m - NFFT / 2;
      float y[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int p = reflect_index(base + h, N);
        const int q = p + NFFT / 2, jb = q >> 8, rr = q & (HOP - 1);
        float t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          int slot = jb - k - (f0 - 3);
          slot = slot < 0 ? 0 : (slot > NSLOT - 1 ? NSLOT - 1 : slot);  // in range by construction
          t[k] = fb[slot * NFFT + rr + k * HOP];
        }
        y[h] = (((t[3] + t[2]) + t[1]) + t[0]) / wss[2 * r + h];
      }
      const float2 w = win[m];
      v[r] = make_float2(y[0] * w.x, y[1] * w.y);
    }
  }
  fft512(v, buf, tws, lane);
  wave_lds_sync();
#pragma unroll
  for (int r = 0; r < 8; ++r) buf[lane + 64 * r] = v[r];
  wave_lds_sync();
  stft_update_store(g, ang_out, f, lane, buf, alpha, pv);
}

// ================================================================================================
// Persistent Griffin-Lim: ALL iterations and the final ISTFT in ONE launch.
//
// The per-iteration launch above is bound by latency, not bytes: a frame is one wave's dependent
// inverse-FFT -> overlap-add -> FFT chain, the halo makes every block redo 2.5x the inverse
// transforms, S / angles / tprev make a round trip through HBM per iteration and every iteration
// pays a launch boundary.  Here a workgroup owns 3..TF consecutive frames for the whole call
// (wave w = frame f0 + w) and keeps their S, angles and previous rebuilt spectrum in LDS; what
// crosses workgroups per iteration is only the overlap of the windowed time frames with the two
// neighbours' sample ranges: 768 pre-summed samples each way (at hop = n_fft/4 a sample is covered
// by four frames, so a block's range [256 f0, 256 (f0 + n) + 768) takes three frames' tails from
// the left neighbour and three frames' heads from the right one).  They travel as data-tagged
// 16-byte granules {three samples, tag = epoch + iteration + 1} -- one sc1 store each, the
// reader re-reads until the tag matches (MI355X_MICROARCH.md hand-off recipe R2, the decoder's
// scheme): no flags, no fences, no grid barrier, placement-independent; two slots by iteration
// parity (a slot is rewritten two iterations later, after its reader has published the iteration
// in between, which the writer had to wait for).  Per iteration and frame there is exactly one
// inverse and one forward transform, and the window-sum division happens once per sample.
//
// Rounding: identical to the launch-per-iteration kernels except for the overlap-add order of a
// sample whose frames span two workgroups: the left neighbour's (ascending) partial sum, then the
// own frames ascending, then the right neighbour's partial sum.
// ================================================================================================
typedef unsigned long long u64;
constexpr int GLP_HALO = 768;
constexpr int FBS = 2 * FFT_SWZ_F2;  // floats per frame row of fb: 1024 samples, 1088 as swizzled FFT scratch
constexpr unsigned GLP_SPIN_LIMIT = 1u << 20;

__device__ __forceinline__ int glp_fstart(int b, int F, int nblk) { return (int)(((long long)b * F) / nblk); }

__device__ __forceinline__ bool glp_give_up(unsigned &spins, int *err, unsigned limit) {
  if (++spins > limit || ((spins & 63u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
    atomicExch(err, 1);
    return true;
  }
  __builtin_amdgcn_s_sleep(1);
  return false;
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void put16(u64 *base, unsigned byte_off, u32x4 v) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 0x7fffffff, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)byte_off, 0, 16);  // sc1
}
__device__ __forceinline__ u32x4 get16(const u64 *base, unsigned byte_off) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 0x7fffffff, 0x00020000);
  return __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 16);  // sc1
}

// W = waves per workgroup the instantiation is compiled for (its register budget): 4 -> one wave per
// SIMD with the whole 512-register file, 8 -> two per SIMD.
// Developer build (-DXDTTS_GL_PROFILE): thread 0 of every workgroup accumulates the 100 MHz wall clock
// between phase markers into p.prof[workgroup][12] (tools/gl_profile.py).
#ifdef XDTTS_GL_PROFILE
#define GLP_MARK(i)                     \
  do {                                  \
    if (tid == 0) {                     \
      const u64 now_ = wall_clock64();  \
      prof_acc[i] += now_ - prof_last;  \
      prof_last = now_;                 \
    }                                   \
  } while (0)
#else
#define GLP_MARK(i) do { } while (0)
#endif

// PC: workgroups per CU the launch is sized for (2: the vocoder batch's shape of two 4-frame workgroups, 256 registers each)
template <int W, int PC = 1>
__global__ __launch_bounds__(64 * W, PC) void k_gl_persistent(GlBufs g, GlPersist p, const float2 *__restrict__ ang_in,
                                                         const float2 *__restrict__ tprev_in, int n_iter, float alpha,
                                                         float *__restrict__ audio) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int TF = p.TF, nthr = 64 * TF;
  // (wave index in an SGPR: "own frame?" and every per-wave base address are scalar, branches instead of exec masks)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), b = blockIdx.x;
  // this workgroup's frames: an even split of one utterance, or its row of the segment table (batch)
  int F = g.F, f0 = glp_fstart(b, g.F, p.nblk), nb_own = glp_fstart(b + 1, g.F, p.nblk) - f0, fbase = 0, abase = 0;
  bool seg_first = b == 0, seg_last = b + 1 == p.nblk;
  if (p.segs) {
    const GlSeg sg = p.segs[b];
    F = sg.F;
    f0 = sg.f0;
    nb_own = sg.n_own;
    fbase = sg.fbase;
    abase = sg.abase;
    seg_first = sg.first != 0;
    seg_last = sg.last != 0;
  }
  const int N = HOP * (F - 1);
  const int range = (nb_own + 3) * HOP, Q0 = f0 * HOP;
  float *sS = smem;                                             // [TF][516]  magnitudes
  float2 *sA = reinterpret_cast<float2 *>(sS + TF * 516);       // [TF][513]  S * (unit-modulus phase estimate): the spectrum the next ISTFT inverts
  float2 *sP = sA + TF * 513;                                   // [TF][513]  previous rebuilt spectrum
  float *fb = reinterpret_cast<float *>(sP + TF * 513);         // [TF][1024] windowed time frames (= each wave's FFT scratch)
  float *yb = fb + TF * FBS;                                   // [(TF+3) 256] overlap-added, normalised signal of the block's range
  float *ws = yb + (TF + 3) * HOP;                              // [(TF+3) 256] 1 / window sum-square
  int *s_err = reinterpret_cast<int *>(ws + (TF + 3) * HOP);    // [1] error word as seen by thread 0 at this iteration
  const bool own = wave < nb_own;
  const int f = f0 + wave;
  const unsigned limit = p.spins > 0 ? (unsigned)p.spins : GLP_SPIN_LIMIT;

  // ---- per-lane constants, once: FFT twiddles, the split/merge twiddles e^{-2 pi i k / 1024} and the
  // window at this lane's 8 packed positions ----
  const TwiddlesP tws = pack_twiddles(load_twiddles(g.tw, lane));
  f2 twk[8], wn[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    twk[r] = as_f2(g.tw[lane + 64 * r]);
    // HALF the window: the 1/2 of the real <-> complex split (e, o below) rides on it -- into the inverse transform's output
    // scale and ahead of the forward transform -- exactly (powers of two), so 24 multiplications per iteration disappear
    wn[r] = as_f2(reinterpret_cast<const float2 *>(g.win)[lane + 64 * r]) * 0.5f;
  }
  // W == 4 (one wave per SIMD, the whole register file): the previous spectrum and the magnitudes of the
  // nine bins a lane updates (pairs k / 512-k for k = lane + 64 r, and k = 256 on lane 0) stay in REGISTERS
  // for the whole call -- 27 LDS accesses less per iteration in the phase-update chain.
  constexpr bool REGSTATE = W == 4 && PC == 1;
  // The edges of the range cross as 16-byte granules {three samples, tag}: one sc1 store and one sc1 load per side for each of the
  // workgroup's first 256 threads (16-byte sc1 accesses are not torn on gfx950: MI355X_MICROARCH.md, hand-off recipe R2).
  f2 rP[9];
  float rS[9];
  if (REGSTATE && own) {
    const float *S = g.S + (size_t)(fbase + f) * g.nb;
    const float2 *P = tprev_in + (size_t)(fbase + f) * g.nb;
    const f2 zero2 = (f2){0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = lane + 64 * r;
      rP[2 * r] = p.gen_phase ? zero2 : as_f2(P[k]);
      rP[2 * r + 1] = p.gen_phase ? zero2 : as_f2(P[512 - k]);
      rS[2 * r] = S[k];
      rS[2 * r + 1] = S[512 - k];
    }
    rP[8] = p.gen_phase ? zero2 : as_f2(P[256]);
    rS[8] = S[256];
  }
  // ---- state of this block's frames into LDS ----
  if (own) {
    const float *S = g.S + (size_t)(fbase + f) * g.nb;
    const float2 *A = ang_in + (size_t)(fbase + f) * g.nb, *P = tprev_in + (size_t)(fbase + f) * g.nb;
    // (all of a lane's loads in flight together: as a loop this was nine dependent round trips to memory per call)
    float sk9[9];
    float2 ak9[9], pk9[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int k = lane + 64 * i;
      sk9[i] = k < 513 ? S[k] : 0.f;
      ak9[i] = pk9[i] = make_float2(0.f, 0.f);
      if (!p.gen_phase && k < 513) {
        ak9[i] = A[k];
        pk9[i] = P[k];
      }
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int k = lane + 64 * i;
      if (k < 513) {
        const float sk = sk9[i];
        float2 ak = ak9[i];
        if (p.gen_phase) {  // the seeded stream of k_phase_init: u keyed (seed, frame * 513 + bin), frame inside the utterance
          const float u = rng_uniform(p.seed, 0x47u, (uint32_t)(f * g.nb + k));
          float sn, cs;
          sincospif(2.0f * u, &sn, &cs);
          ak = make_float2(cs, sn);
        }
        sS[wave * 516 + k] = sk;
        sA[wave * 513 + k] = make_float2(ak.x * sk, ak.y * sk);
        sP[wave * 513 + k] = pk9[i];
      }
    }
  }
  {
    // a thread's samples j = tid + m nthr all sit at the same offset r of their hop (Q0 and nthr are multiples of 256): its
    // four window values once, not per sample
    const int r = tid & (HOP - 1);
    float w4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) w4[k] = g.win[r + k * HOP];
    for (int j = tid; j < range; j += nthr) {
      const int jb = (Q0 + j) >> 8;
      float wss = 0.f;
#pragma unroll
      for (int k = 3; k >= 0; --k) {
        const int fr = jb - k;
        wss = (fr >= 0 && fr < F) ? fmaf(w4[k], w4[k], wss) : wss;  // (one rounding per term, as the launch-per-iteration kernels' window sum)
      }
      ws[j] = wss > 1.17549435e-38f ? 1.0f / wss : 1.0f;  // reciprocal of the divisor: one multiply per sample and iteration
    }
  }
  if (tid == 0) s_err[1] = 0;
  __syncthreads();

  u64 *inL = p.xch + (size_t)b * 4 * GLP_HALO, *outL = !seg_first ? p.xch + ((size_t)(b - 1) * 4 + 1) * GLP_HALO : nullptr;
  u64 *outR = !seg_last ? p.xch + (size_t)(b + 1) * 4 * GLP_HALO : nullptr;
  // slot layout per block: [parity][side: 0 = from the left neighbour, 1 = from the right neighbour][768]
  f2 *buf = reinterpret_cast<f2 *>(fb + wave * FBS);

#ifdef XDTTS_GL_PROFILE
  u64 prof_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  u64 prof_last = wall_clock64();
#endif
  for (int it = 0; it <= n_iter; ++it) {
    const unsigned want = p.epoch + (unsigned)it + 1u;
    const int par = it & 1;
    // test hook: a straggler workgroup -- before its transforms (even iterations) or between its publish and its polls (odd)
    const bool lag = p.slow && b == p.slow - 1;
    if (lag && !(it & 1))
      for (int i = 0; i < 2; ++i) __builtin_amdgcn_s_sleep(127);
    GLP_MARK(0);  // loop overhead
    // ---- A: inverse transform of the own frame: irfft(1024) of S * angles, synthesis window -> fb ----
    if (own) {
      const f2 *X = reinterpret_cast<const f2 *>(sA + wave * 513);
      f2 v[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int k = lane + 64 * r, kc = 512 - k;
        f2 xk = X[k], xc = X[kc];
        if (k == 0) {  // irfft ignores the imaginary part of DC and Nyquist
          xk.y = 0.f;
          xc.y = 0.f;
        }
        const f2 e = pk_add_conj(xk, xc);   // X[k] + conj X[512-k]   (the 1/2 is in wn)
        const f2 d = pk_sub_conj(xk, xc);   // X[k] - conj X[512-k]
        const f2 o = pk_cmul_conj(d, twk[r]);
        v[r] = pk_conj_sub_mi(e, o);               // conj(e + i o): the forward transform of the conjugate = the inverse's conjugate
      }
      fft512sp(v, buf, tws, lane);
      const float sc = 1.0f / 512.0f;
#pragma unroll
      for (int r = 0; r < 8; ++r) buf[lane + 64 * r] = pk_mul_conj(v[r] * sc, wn[r]);
    }
    GLP_MARK(1);  // A: inverse transform
    __syncthreads();
    GLP_MARK(2);  // barrier after A
    // ---- B0: the own frames' sums over the first and the last 768 samples of the range (own frames 0..2 / n-3..n-1,
    // ascending), straight from the frames and before anything else: they are what the neighbours lack -- so the granules
    // travel while the middle of the range is summed -- and what B2 adds the neighbours' sums to, kept in registers ----
    // Edge threads: the first 256 of the workgroup, thread t <-> samples t, t + 256, t + 512 of each edge (u = the 256-sample
    // third, hence which own frames reach it); they travel as ONE 16-byte granule {three samples, tag} per side and thread.
    constexpr int U = 3, ETHR = 256;
    const bool edge = tid < ETHR;
    float pl[U], pr[U];
    {
      const bool has_l = !seg_first, has_r = outR != nullptr;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = tid + u * ETHR;
        pl[u] = pr[u] = 0.f;
        if (edge) {
          float v = fb[k];
          if (u >= 1) v += fb[FBS + k - HOP];
          if (u >= 2) v += fb[2 * FBS + k - 2 * HOP];
          pl[u] = v;
          const float l1 = fb[(nb_own - 1) * FBS + HOP + k];
          v = l1;
          if (u == 0) v = (fb[(nb_own - 3) * FBS + 3 * HOP + k] + fb[(nb_own - 2) * FBS + 2 * HOP + k]) + l1;
          else if (u == 1) v = fb[(nb_own - 2) * FBS + 2 * HOP + k] + l1;
          pr[u] = v;
        }
      }
      if (edge) {
        if (has_l) put16(outL, (unsigned)(par * 2 * GLP_HALO) * 8u + 16u * (unsigned)tid, (u32x4){__float_as_uint(pl[0]), __float_as_uint(pl[1]), __float_as_uint(pl[2]), want});
        if (has_r) put16(outR, (unsigned)(par * 2 * GLP_HALO) * 8u + 16u * (unsigned)tid, (u32x4){__float_as_uint(pr[0]), __float_as_uint(pr[1]), __float_as_uint(pr[2]), want});
      }
    }
    GLP_MARK(10);  // B0: publish
    if (lag && (it & 1))
      for (int i = 0; i < 2; ++i) __builtin_amdgcn_s_sleep(127);
    // First poll of the neighbours' granules, issued NOW: the loads' round trip (~1 us) overlaps the own overlap-add
    // below; the neighbours run in lock-step with this workgroup, so their stores are usually on their way already.
    // What is not there yet is polled again in B2.
    u32x4 qv_l = (u32x4){0u, 0u, 0u, 0u}, qv_r = qv_l;
    auto early_poll = [&]() {
      const u64 *gl0 = inL + (size_t)par * 2 * GLP_HALO, *gr0 = gl0 + GLP_HALO;
      if (edge && !seg_first) qv_l = get16(gl0, 16u * (unsigned)tid);
      if (edge && outR != nullptr) qv_r = get16(gr0, 16u * (unsigned)tid);
    };
    // (the error word the workgroup acts on at the end of B2 is requested here, ahead of the overlap-add: read where it is
    // needed, its round trip -- 0.3 us -- sat between the last LDS store of B2 and the barrier of every iteration)
    int err_seen = 0;
    if (tid == 0) err_seen = __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (p.poll_delay >= 0) {
      for (int i = 0; i < p.poll_delay; ++i) __builtin_amdgcn_s_sleep(2);
      early_poll();
    }
    // ---- B1: overlap-add and normalisation of the samples no neighbour reaches, [768, 256 n): four own frames each -> yb.
    // float4 per thread; the 256-sample chunk index is wave-uniform.  (The edges of the range are B0's sums + B2.)
    {
      const float4 *fb4 = reinterpret_cast<const float4 *>(fb);
      float4 *yb4 = reinterpret_cast<float4 *>(yb);
      for (int q4 = 3 * (HOP / 4) + tid; q4 < nb_own * (HOP / 4); q4 += nthr) {
        const int c = q4 >> 6;
        float4 a = fb4[(c - 3) * (FBS / 4) + q4 - (HOP / 4) * (c - 3)];
#pragma unroll
        for (int i = 2; i >= 0; --i) {  // ascending frame order
          const float4 t = fb4[(c - i) * (FBS / 4) + q4 - (HOP / 4) * (c - i)];
          a.x += t.x;
          a.y += t.y;
          a.z += t.z;
          a.w += t.w;
        }
        const float4 w = reinterpret_cast<const float4 *>(ws)[q4];
        yb4[q4] = make_float4(a.x * w.x, a.y * w.y, a.z * w.z, a.w * w.w);
      }
    }
    if (p.poll_delay < 0) {
      for (int i = 0; i < -1 - p.poll_delay; ++i) __builtin_amdgcn_s_sleep(2);
      early_poll();
    }
    GLP_MARK(3);  // own overlap-add
    // ---- B2: take the neighbours' contributions to the first / last 768 samples, normalise ----
    {
      float hl[U], hr[U];
      const bool has_l = !seg_first, has_r = outR != nullptr;
      const u64 *gl_ = inL + (size_t)par * 2 * GLP_HALO, *gr_ = gl_ + GLP_HALO;
      unsigned spins = 0;
      GLP_MARK(4);  // middle samples
      {
        bool wl = edge && has_l, wr = edge && has_r;
        for (;;) {
          if (wl && qv_l.w == want) wl = false;
          if (wr && qv_r.w == want) wr = false;
          if (!(wl || wr) || glp_give_up(spins, p.err, limit)) break;
          if (wl) qv_l = get16(gl_, 16u * (unsigned)tid);
          if (wr) qv_r = get16(gr_, 16u * (unsigned)tid);
          asm volatile("" ::: "memory");
        }
        hl[0] = __uint_as_float(qv_l.x), hl[1] = __uint_as_float(qv_l.y), hl[2] = __uint_as_float(qv_l.z);
        hr[0] = __uint_as_float(qv_r.x), hr[1] = __uint_as_float(qv_r.y), hr[2] = __uint_as_float(qv_r.z);
        if (wl) hl[0] = hl[1] = hl[2] = 0.f;  // (timed out: the launch drains)
        if (wr) hr[0] = hr[1] = hr[2] = 0.f;
        if (wl || wr) s_err[1] = 1;  // a poll of this workgroup gave up (or saw the error word set)
      }
      GLP_MARK(5);  // wait for the neighbours' overlaps
      if (edge) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int k = tid + u * ETHR;
          // left neighbour's frames come first in ascending order, the right neighbour's last.  When the
          // block has 3 frames the two edges meet at j = 768 and never overlap (256 n >= 768).
          yb[k] = (has_l ? hl[u] + pl[u] : pl[u]) * ws[k];
          const int j = HOP * nb_own + k;
          yb[j] = (has_r ? pr[u] + hr[u] : pr[u]) * ws[j];
        }
      }
      if (tid == 0) s_err[0] = err_seen;  // what thread 0 saw before the overlap-add (or s_err[1]: a poll that gave up)
    }
    __syncthreads();
    GLP_MARK(6);  // finalise + barrier after B
    if (s_err[0] | s_err[1]) return;  // an exchange timed out somewhere: the whole launch drains, the host falls back
    if (it == n_iter) {
      // ---- final ISTFT: the block's share of the centre-trimmed signal ----
      if (audio) {
        const int lo = seg_first ? 0 : 384, hi = seg_last ? range : 384 + HOP * nb_own;
        for (int j = lo + tid; j < hi; j += nthr) {
          const int n = Q0 + j - NFFT / 2;
          if (n >= 0 && n < N) audio[abase + n] = yb[j];
        }
      }
      break;
    }
    // ---- C: forward transform of the own frame from the block's signal, phase update ----
    if (own) {
      f2 v[8];
      if (f >= 2 && f <= F - 3) {
        const f2 *y2 = reinterpret_cast<const f2 *>(yb + HOP * wave);
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = y2[lane + 64 * r] * wn[r];
      } else {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int base = f * HOP + 2 * (lane + 64 * r) - NFFT / 2;
          // (one fold suffices: this engine runs from 16 frames on, N >= 3840 > the 512-sample pad -- and the general
          // loop cost the two edge workgroups 1.1 us per iteration that every other workgroup then waited for)
          // (the 256-register instantiations keep the fold loop: the branch-free form lengthens live ranges there -- 2 -> 13
          // spills, the 32-utterance vocoder batch 8 % slower)
          const int i0 = REGSTATE ? reflect_once(base, N) : reflect_index(base, N), i1 = REGSTATE ? reflect_once(base + 1, N) : reflect_index(base + 1, N);
          const float y0 = yb[i0 + NFFT / 2 - Q0], y1 = yb[i1 + NFFT / 2 - Q0];
          v[r] = (f2){y0 * wn[r].x, y1 * wn[r].y};
        }
      }
      GLP_MARK(7);  // C: gather + window
      fft512sp(v, buf, tws, lane);
      GLP_MARK(8);  // C: forward FFT
      wave_lds_sync();
#pragma unroll
      for (int r = 0; r < 8; ++r) buf[lane + 64 * r] = v[r];
      wave_lds_sync();
      // Unpack the 512-point spectrum Z of the packed real frame into the 513 bins X and update the
      // phase: bins k and 512 - k share everything up to one complex product,
      //   X[k] = e + w o,  X[512-k] = conj(e - w o),  e = (Z[k] + conj Z[512-k]) / 2,
      //   o = (Z[k] - conj Z[512-k]) / 2i,  w = e^{-2 pi i k / 1024},
      // so a lane takes the pair (k, 512-k) for k = lane + 64 r < 256; k = 256 is its own partner
      // (X[256] = conj Z[256]) and goes to lane 0 afterwards.  a = X - alpha * previous X;
      // angles = a / (|a| + 1e-16) through v_sqrt_f32 / v_rcp_f32 (1 ulp each); the stored value is
      // S * angles, what the next inverse transform needs.
      // All LDS loads first, all stores last: the compiler cannot tell the state arrays apart (they are
      // carved from one buffer), so interleaved loads and stores would serialise into ~27 dependent LDS
      // round trips per iteration.
      f2 *X = reinterpret_cast<f2 *>(sA + wave * 513), *P = reinterpret_cast<f2 *>(sP + wave * 513);
      const float *S = sS + wave * 516;
      f2 *ang_g = (p.ang_out && it == n_iter - 1) ? reinterpret_cast<f2 *>(p.ang_out + (size_t)(fbase + f) * g.nb) : nullptr;  // parity hook
      f2 zk[4], zc[4], pv[9];
      float sk[9];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = lane + 64 * r;
        zk[r] = buf[k];
        zc[r] = buf[(512 - k) & 511];
        pv[2 * r] = REGSTATE ? rP[2 * r] : P[k];
        pv[2 * r + 1] = REGSTATE ? rP[2 * r + 1] : P[512 - k];
        sk[2 * r] = REGSTATE ? rS[2 * r] : S[k];
        sk[2 * r + 1] = REGSTATE ? rS[2 * r + 1] : S[512 - k];
      }
      const f2 z256 = buf[256];
      pv[8] = REGSTATE ? rP[8] : P[256];
      sk[8] = REGSTATE ? rS[8] : S[256];
      f2 xs[9], xo[9];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const f2 e = pk_add_conj(zk[r], zc[r]);  // (Z[k] + conj Z[512-k]) / 2   (Z is the transform of the half-windowed frame)
        const f2 o = pk_odd(zk[r], zc[r]);       // (Z[k] - conj Z[512-k]) / 2i
        const f2 t = pk_cmul(twk[r], o);
        xs[2 * r] = e + t;
        xs[2 * r + 1] = pk_conj_sub(e, t);
      }
      xs[8] = (f2){2.f * z256.x, -2.f * z256.y};  // X[256] = conj Z[256] of the full-window frame
      const f2 malpha = (f2){-alpha, -alpha};
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const f2 a = __builtin_elementwise_fma(malpha, pv[i], xs[i]);
        const float inv = __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(fmaf(a.x, a.x, a.y * a.y)) + 1e-16f);
        xo[i] = a * inv;  // the unit-modulus angle
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = lane + 64 * r;
        if (ang_g) {
          ang_g[k] = xo[2 * r];
          ang_g[512 - k] = xo[2 * r + 1];
        }
        if (REGSTATE) {
          rP[2 * r] = xs[2 * r];
          rP[2 * r + 1] = xs[2 * r + 1];
        } else {
          P[k] = xs[2 * r];
          P[512 - k] = xs[2 * r + 1];
        }
        X[k] = xo[2 * r] * sk[2 * r];
        X[512 - k] = xo[2 * r + 1] * sk[2 * r + 1];
      }
      if (REGSTATE) rP[8] = xs[8];
      if (lane == 0) {
        if (ang_g) ang_g[256] = xo[8];
        if (!REGSTATE) P[256] = xs[8];
        X[256] = xo[8] * sk[8];
      }
      wave_lds_sync();
    }
    GLP_MARK(9);  // C: unpack + phase update
  }
#ifdef XDTTS_GL_PROFILE
  if (p.prof && tid == 0)
    for (int i = 0; i < 12; ++i) p.prof[b * 12 + i] = prof_acc[i];
#endif
  // ---- state write-back (parity hook only; the angles were stored by the last update) ----
  if (p.tprev_out && own) {
    float2 *P = p.tprev_out + (size_t)(fbase + f) * g.nb;
    if (REGSTATE) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        P[lane + 64 * r] = make_float2(rP[2 * r].x, rP[2 * r].y);
        P[512 - (lane + 64 * r)] = make_float2(rP[2 * r + 1].x, rP[2 * r + 1].y);
      }
      if (lane == 0) P[256] = make_float2(rP[8].x, rP[8].y);
    } else {
      for (int k = lane; k < 513; k += 64) P[k] = sP[wave * 513 + k];
    }
  }
}

// Final ISTFT output: overlap-add + normalisation + centre trim for every sample.
__global__ void k_overlap_add(GlBufs g, float *y) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= g.hop * (g.F - 1)) return;
  y[n] = ola_sample(g, n);
}

// angles = exp(2 pi i u), u from the counter RNG keyed (seed, frame*nb + bin); or a caller-supplied
// phase0 [nb][F][2].  Also clears tprev (rebuilt = 0 before the first iteration).
__global__ void k_phase_init(GlBufs g, uint32_t seed, const float *phase0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= g.F * g.nb) return;
  const int f = i / g.nb, k = i % g.nb;
  float2 a;
  if (phase0) {
    a = make_float2(phase0[((size_t)k * g.F + f) * 2], phase0[((size_t)k * g.F + f) * 2 + 1]);
  } else {
    const float u = rng_uniform(seed, 0x47u, (uint32_t)i);
    float sn, cs;
    sincospif(2.0f * u, &sn, &cs);
    a = make_float2(cs, sn);
  }
  g.ang[i] = a;
  g.tprev[i] = make_float2(0.f, 0.f);
}

// batch form of k_phase_init: rows of several utterances; the seeded stream is indexed inside each utterance
__global__ void k_phase_init_batch(GlBufs g, uint32_t seed, const int *__restrict__ frame_local) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= g.F * g.nb) return;
  const int f = i / g.nb, k = i % g.nb;
  const float u = rng_uniform(seed, 0x47u, (uint32_t)(frame_local[f] * g.nb + k));
  float sn, cs;
  sincospif(2.0f * u, &sn, &cs);
  g.ang[i] = make_float2(cs, sn);
  g.tprev[i] = make_float2(0.f, 0.f);
}

// de-compress the mel (mode 0: exp of the natural-log mel Tacotron2 emits; 1: none; 2: 10^x) and
// transpose (80 x F) -> (F x 80) for the pinv GEMM
__global__ void k_exp_transpose(const float *mel, float *out, int n_mels, int F, int mode) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_mels * F) return;
  const int f = i / n_mels, m = i % n_mels;
  const float v = mel[(size_t)m * F + f];
  out[i] = mode == 0 ? expf(v) : (mode == 2 ? exp10f(v) : v);
}

__global__ void k_pow_rows(const float *in, int ld, float *out, int nb, int F, float p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nb * F) return;
  const int f = i / nb, k = i % nb;
  const float v = in[(size_t)f * ld + k];
  out[i] = p == 1.0f ? v : powf(fmaxf(v, 0.f), p);
}

// Output normalisation (G6; xdtts_griffinlim_opts.output_normalise): 1 = y / max|y|, 2 = y * target / rms(y),
// 3 = y * min(target / rms(y), 1 / max|y|) (mode 2 that never scales a sample past +-1, where src/lib.rs:155's cast would clip).
// Two launches per group of utterances, no atomics: GLN_PARTS blocks per utterance each reduce a fixed strided share
// (lane-strided accumulation, xor-shuffle tree, the four waves summed in order), then every scaling block re-reduces the
// utterance's GLN_PARTS partials in one fixed order -- the result does not depend on scheduling.  An utterance's scratch is
// two planes of GLN_PARTS floats: sums of squares (peaks in mode 1), and the peaks of mode 3.
__global__ void __launch_bounds__(256) k_norm_partials(const float *y, const int2 *tab, int2 single, int mode, float *parts) {
  const int2 t = tab ? tab[blockIdx.y] : single;  // (first sample, samples)
  const float *p = y + t.x;
  float acc = 0.f, pk = 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < t.y; i += GLN_PARTS * 256) {
    const float v = p[i];
    acc = mode == 1 ? fmaxf(acc, fabsf(v)) : fmaf(v, v, acc);
    pk = fmaxf(pk, fabsf(v));
  }
  for (int o = 32; o > 0; o >>= 1) {
    const float b = __shfl_xor(acc, o, 64);
    acc = mode == 1 ? fmaxf(acc, b) : acc + b;
    pk = fmaxf(pk, __shfl_xor(pk, o, 64));
  }
  __shared__ float w[4], wp[4];
  if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = acc, wp[threadIdx.x >> 6] = pk;
  __syncthreads();
  if (threadIdx.x == 0) {
    float *q = parts + (size_t)blockIdx.y * GLN_SCRATCH;
    q[blockIdx.x] = mode == 1 ? fmaxf(fmaxf(w[0], w[1]), fmaxf(w[2], w[3])) : ((w[0] + w[1]) + (w[2] + w[3]));
    q[GLN_PARTS + blockIdx.x] = fmaxf(fmaxf(wp[0], wp[1]), fmaxf(wp[2], wp[3]));
  }
}
__global__ void __launch_bounds__(256) k_norm_scale(float *y, const int2 *tab, int2 single, int mode, float target, const float *parts) {
  const int2 t = tab ? tab[blockIdx.y] : single;
  if (blockIdx.x * 1024 >= t.y) return;
  static_assert(GLN_PARTS == 64, "one lane per partial");
  const float *q = parts + (size_t)blockIdx.y * GLN_SCRATCH;
  float acc = q[threadIdx.x & 63], pk = q[GLN_PARTS + (threadIdx.x & 63)];
  for (int o = 32; o > 0; o >>= 1) {
    const float b = __shfl_xor(acc, o, 64);
    acc = mode == 1 ? fmaxf(acc, b) : acc + b;
    pk = fmaxf(pk, __shfl_xor(pk, o, 64));
  }
  float *p = y + t.x;
  if (mode == 1) {
    if (!(acc > 0.f)) return;
    for (int i = blockIdx.x * 1024 + threadIdx.x; i < min(t.y, (int)(blockIdx.x + 1) * 1024); i += 256) p[i] = p[i] / acc;
  } else {
    const float r = sqrtf(acc / (float)t.y);
    if (!(r > 0.f)) return;
    float sc = target / r;
    if (mode == 3 && sc * pk > 1.f) sc = 1.f / pk;
    for (int i = blockIdx.x * 1024 + threadIdx.x; i < min(t.y, (int)(blockIdx.x + 1) * 1024); i += 256) p[i] = p[i] * sc;
  }
}

// Parity hook (xdtts_griffinlim_step): the iteration state crosses the boundary in the crate's
// (n_bins x F x 2) layout; on the device it is [F][nb] float2.
__global__ void k_state_import(GlBufs g, const float *ang_in, const float *reb_in) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= g.F * g.nb) return;
  const int f = i / g.nb, k = i % g.nb;
  const size_t o = ((size_t)k * g.F + f) * 2;
  g.ang[i] = make_float2(ang_in[o], ang_in[o + 1]);
  g.tprev[i] = make_float2(reb_in[o], reb_in[o + 1]);
}
__global__ void k_state_export(GlBufs g, const float2 *ang, const float2 *tprev, float *ang_out, float *reb_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= g.F * g.nb) return;
  const int f = i / g.nb, k = i % g.nb;
  const size_t o = ((size_t)k * g.F + f) * 2;
  const float2 a = ang[i], r = tprev[i];
  ang_out[o] = a.x;
  ang_out[o + 1] = a.y;
  reb_out[o] = r.x;
  reb_out[o + 1] = r.y;
}

}  // namespace

void launch_gl_exp_transpose(const float *mel_80xF, float *out_Fx80, int n_mels, int F, int mode, hipStream_t s) {
  const int n = n_mels * F;
  hipLaunchKernelGGL(k_exp_transpose, dim3((n + 255) / 256), dim3(256), 0, s, mel_80xF, out_Fx80, n_mels, F, mode);
  HIP_CHECK(hipGetLastError());
}

void launch_gl_pow_rows(const float *in, int ld, float *out, int nb, int F, float p, hipStream_t s) {
  const int n = nb * F;
  hipLaunchKernelGGL(k_pow_rows, dim3((n + 255) / 256), dim3(256), 0, s, in, ld, out, nb, F, p);
  HIP_CHECK(hipGetLastError());
}

void launch_gl_output_normalise(float *y, const int2 *tab_dev, int n_utt, int first, int n_max, int mode, float target,
                                float *parts, hipStream_t s) {
  if (mode == 0 || n_utt <= 0 || n_max <= 0) return;
  const int2 single = make_int2(first, n_max);
  hipLaunchKernelGGL(k_norm_partials, dim3(GLN_PARTS, n_utt), dim3(256), 0, s, y, tab_dev, single, mode, parts);
  hipLaunchKernelGGL(k_norm_scale, dim3((n_max + 1023) / 1024, n_utt), dim3(256), 0, s, y, tab_dev, single, mode, target, parts);
  HIP_CHECK(hipGetLastError());
}

void launch_gl_phase_init(const GlBufs &g, uint32_t seed, const float *phase0_dev, hipStream_t s) {
  const int n = g.F * g.nb;
  hipLaunchKernelGGL(k_phase_init, dim3((n + 255) / 256), dim3(256), 0, s, g, seed, phase0_dev);
  HIP_CHECK(hipGetLastError());
}

void launch_gl_phase_init_batch(const GlBufs &g, uint32_t seed, const int *frame_local, hipStream_t s) {
  const int n = g.F * g.nb;
  hipLaunchKernelGGL(k_phase_init_batch, dim3((n + 255) / 256), dim3(256), 0, s, g, seed, frame_local);
  HIP_CHECK(hipGetLastError());
}

void launch_gl_prepare(const GlBufs &g, hipStream_t s) {
  const int N = g.hop * (g.F - 1);
  hipLaunchKernelGGL(k_wss_inv, dim3((N + 255) / 256), dim3(256), 0, s, g);
  HIP_CHECK(hipGetLastError());
}

void launch_gl_state_import(const GlBufs &g, const float *ang_in, const float *reb_in, hipStream_t s) {
  const int n = g.F * g.nb;
  hipLaunchKernelGGL(k_state_import, dim3((n + 255) / 256), dim3(256), 0, s, g, ang_in, reb_in);
  HIP_CHECK(hipGetLastError());
}

void launch_gl_state_export(const GlBufs &g, const float2 *ang, const float2 *tprev, float *ang_out, float *reb_out,
                            hipStream_t s) {
  const int n = g.F * g.nb;
  hipLaunchKernelGGL(k_state_export, dim3((n + 255) / 256), dim3(256), 0, s, g, ang, tprev, ang_out, reb_out);
  HIP_CHECK(hipGetLastError());
}

// ---- persistent engine: plan, resources, launch ----
bool gl_persistent_plan(int F, int n_cu, int *TF, int *nblk) {
  if (F < 16) return false;  // reflect padding folds more than once: two-kernel path
  int tf = std::max(4, (F + n_cu - 1) / n_cu);
  if (tf > GLP_TF_MAX) return false;
  const int nb = (F + tf - 1) / tf;
  if (nb > n_cu || F / nb < 3) return false;  // one workgroup per CU; every block needs >= 3 frames
  *TF = tf;
  *nblk = nb;
  return true;
}
size_t gl_persistent_lds_bytes(int TF) { return sizeof(float) * ((size_t)TF * (516 + 2 * 1026 + FBS) + 2 * (size_t)(TF + 3) * HOP + 4); }
size_t gl_persistent_xch_words(int nblk) { return (size_t)nblk * 4 * GLP_HALO; }

bool gl_persistent_supported(int device, int *n_cu, int *per_cu4) {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return false;
  const size_t lds4 = gl_persistent_lds_bytes(4), lds8 = gl_persistent_lds_bytes(GLP_TF_MAX);
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_gl_persistent<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4) !=
          hipSuccess ||
      hipFuncSetAttribute(reinterpret_cast<const void *>(k_gl_persistent<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8) !=
          hipSuccess)
    return false;
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_gl_persistent<4>, 256, lds4) != hipSuccess || per_cu < 1) return false;
  // two 4-frame workgroups per CU (73 of 160 KB of LDS each, the state in LDS instead of registers): the vocoder batch
  *per_cu4 = 1;
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_gl_persistent<4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4) ==
          hipSuccess &&
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_gl_persistent<4, 2>, 256, lds4) == hipSuccess && per_cu >= 2)
    *per_cu4 = 2;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_gl_persistent<8>, 512, lds8) != hipSuccess || per_cu < 1) return false;
  *n_cu = prop.multiProcessorCount;
  return true;
}

// All n_iter iterations and (audio != null) the final ISTFT in one launch.  The state is read from
// ang_in / tprev_in and left untouched, so a failed exchange (p.err set) can be retried on the
// launch-per-iteration path; p.ang_out / p.tprev_out (parity hook) receive the final state.
void launch_gl_persistent(const GlBufs &g, const GlPersist &p, const float2 *ang_in, const float2 *tprev_in, int n_iter,
                          float alpha, float *audio, hipStream_t s) {
  const void *fn = p.TF > 4        ? reinterpret_cast<const void *>(k_gl_persistent<8>)
                   : p.per_cu > 1 ? reinterpret_cast<const void *>(k_gl_persistent<4, 2>)
                                  : reinterpret_cast<const void *>(k_gl_persistent<4>);
  HIP_CHECK(launch_coresident(false, fn, dim3(p.nblk), dim3(64 * p.TF), gl_persistent_lds_bytes(p.TF), s, g, p, ang_in, tprev_in, n_iter, alpha,
                              audio));
}

// Enqueues n_iter iterations (no final ISTFT) and returns the buffer that holds the final angles.
// Frame counts of 16 and more run the fused one-launch iteration with the angles ping-ponging
// between g.ang and g.ang2; tiny inputs (whose reflect padding folds more than once) use the
// two-kernel iteration in place.
const float2 *launch_gl_iterate(const GlBufs &g, int n_iter, float alpha, hipStream_t s) {
  const int nblk = (g.F + FRAMES_PER_BLOCK - 1) / FRAMES_PER_BLOCK;
  const float2 *final_ang = g.ang;
  if (g.F >= 16) {
    constexpr int TF = 4;
    const size_t lds = sizeof(float) * (TF + 6) * (NFFT + 1024);
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(k_gl_fused<TF>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    HIP_CHECK(attr);
    const int grid = (g.F + TF - 1) / TF;
    float2 *a = g.ang, *b = g.ang2;
    for (int i = 0; i < n_iter; ++i) {
      hipLaunchKernelGGL(k_gl_fused<TF>, dim3(grid), dim3(64 * (TF + 6)), lds, s, g, a, b, alpha);
      std::swap(a, b);
    }
    final_ang = a;
  } else {
    for (int i = 0; i < n_iter; ++i) {
      hipLaunchKernelGGL(k_istft_frames, dim3(nblk), dim3(256), 0, s, g, g.ang);
      hipLaunchKernelGGL(k_stft_update, dim3(nblk), dim3(256), 0, s, g, alpha);
    }
  }
  HIP_CHECK(hipGetLastError());
  return final_ang;
}

// The final ISTFT of `ang` into `audio`.
void launch_gl_final(const GlBufs &g, const float2 *ang, float *audio, hipStream_t s) {
  const int nblk = (g.F + FRAMES_PER_BLOCK - 1) / FRAMES_PER_BLOCK;
  const int N = g.hop * (g.F - 1);
  hipLaunchKernelGGL(k_istft_frames, dim3(nblk), dim3(256), 0, s, g, ang);
  hipLaunchKernelGGL(k_overlap_add, dim3((N + 255) / 256), dim3(256), 0, s, g, audio);
  HIP_CHECK(hipGetLastError());
}

void launch_gl_iterations(const GlBufs &g, int n_iter, float alpha, float *audio, hipStream_t s) {
  launch_gl_final(g, launch_gl_iterate(g, n_iter, alpha, s), audio, s);
}

}  // namespace xdtts
