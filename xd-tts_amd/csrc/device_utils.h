// device_utils.h -- device-side helpers shared by the decoder kernels (gfx950, wave64).
#pragma once
#include "common.h"

namespace xdtts {
namespace {

// Wave-wide reductions on the DPP data path (VALU cross-lane moves, a few cycles each) instead of
// __shfl_xor, which lowers to ds_bpermute: six dependent LDS-crossbar round trips (~100 cycles
// each) per reduction were a visible part of every latency-bound kernel here.  Sequence (rocPRIM's
// wave64 pattern for gfx9): quad_perm swaps, row_ror:4, row_ror:8 leave each 16-lane row's total
// in all its lanes; row_bcast:15 adds row 0 into row 1 and row 2 into row 3; row_bcast:31 adds
// rows 0+1 into rows 2,3; lane 63 then holds the total and is broadcast with readlane.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move(float old, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_move<0xB1, 0xf>(0.f, v);   // quad_perm:[1,0,3,2]
  v += dpp_move<0x4E, 0xf>(0.f, v);   // quad_perm:[2,3,0,1]
  v += dpp_move<0x124, 0xf>(0.f, v);  // row_ror:4
  v += dpp_move<0x128, 0xf>(0.f, v);  // row_ror:8
  v += dpp_move<0x142, 0xa>(0.f, v);  // row_bcast:15 -> rows 1, 3
  v += dpp_move<0x143, 0xc>(0.f, v);  // row_bcast:31 -> rows 2, 3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_move<0xB1, 0xf>(v, v));
  v = fmaxf(v, dpp_move<0x4E, 0xf>(v, v));
  v = fmaxf(v, dpp_move<0x124, 0xf>(v, v));
  v = fmaxf(v, dpp_move<0x128, 0xf>(v, v));
  v = fmaxf(v, dpp_move<0x142, 0xa>(v, v));
  v = fmaxf(v, dpp_move<0x143, 0xc>(v, v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float dot4(float4 a, float4 b, float acc) {
  acc = fmaf(a.x, b.x, acc);
  acc = fmaf(a.y, b.y, acc);
  acc = fmaf(a.z, b.z, acc);
  acc = fmaf(a.w, b.w, acc);
  return acc;
}
// streamed-once weights: non-temporal so the two GEMV streams do not evict the ~2 MB of
// small-kernel weights and partial buffers from the 4 MB-per-XCD L2
__device__ __forceinline__ float4 ld_stream(const float4 *p) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
// Prenet dropout (D1; the exported decoder graph keeps it on at inference): is unit j of `layer` dropped at this step?
//   mode 1: the seeded counter stream, keyed (seed, dropout-stream index of the chunk, step, layer, unit)
//   mode 2: the caller's keep bytes [chunk][drop_steps][2][256] (chunk = index of the chunk within the call)
__device__ __forceinline__ bool prenet_dropped(int mode, uint32_t seed, uint32_t item, const unsigned char *masks, int drop_steps, int chunk,
                                               int step, int layer, int j) {
  if (mode == 2) {
    if (step >= drop_steps) return false;  // past the caller's rows: kept, as the oracle's prenet_keep (only speculative steps get here)
    return masks[(((size_t)chunk * drop_steps + step) * 2 + layer) * PRENET + j] == 0;
  }
  return (rng_u32(seed, 0x1000u + (uint32_t)layer + 2u * item, (uint32_t)step * 256u + (uint32_t)j) >> 31) != 0;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// src/tacotron2/mod.rs:126-133: the two-branch sigmoid applied to the gate logit on the host
__device__ __forceinline__ float gate_sigmoid(float x) {
  if (x >= 0.0f) return 1.0f / (1.0f + expf(-x));
  const float e = expf(x);
  return e / (1.0f + e);
}

// The stop rule (mod.rs:319-324): sigmoid(gate) > threshold.  The two-branch sigmoid above is ~100 dependent instructions (libm
// expf, a division) on the prenet role's critical path of EVERY step -- 0.4 us of a 9 us step -- although its verdict is plain
// for all but the logits next to logit(threshold): the host passes a band [lo, hi] around that point, 1e-3 (1 + |logit|) wide --
// a thousand times the f32 sigmoid's own uncertainty there -- and only a logit inside it takes the reference's arithmetic.
__device__ __forceinline__ bool gate_fires(float x, float lo, float hi, float threshold) {
  if (x < lo) return false;
  if (x > hi) return true;
  return gate_sigmoid(x) > threshold;  // (also NaN: both comparisons above are false)
}

// Hardware exp2/rcp forms for the few transcendental chains that sit on the per-step critical path
// (cell updates, energies, softmax).  v_exp_f32 / v_rcp_f32 are 1-ulp instructions; against libm's
// expf/tanhf the results move by a few 1e-7 absolute, far inside the 1e-4 parity bar, and each cell
// update loses ~120 dependent instructions.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + fast_exp(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(fast_exp(2.0f * x) + 1.0f); }

}  // namespace
}  // namespace xdtts
