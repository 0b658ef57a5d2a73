// p8_exchange.h -- the exchange primitives of the persistent MFMA decoders (decoder_persistent8.hip: 3..8 chunks, decoder_persistent16.hip:
// 9..16 chunks): {tag, value} granules for the narrow edges, write-once rings of plain values (a value is its own arrival flag)
// for the vectors every workgroup gathers, bounded polls that watch a global error word.  Included by both translation units.
#pragma once
#include "device_utils.h"
#include "kernels.h"

namespace xdtts {
namespace {

typedef unsigned long long u64;
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned ACT_BIT = 0x80000000u;

__device__ __forceinline__ void publish(u64 *slot, unsigned tag, float v) {
  __hip_atomic_store(slot, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 peek(const u64 *slot) { return __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
struct PollCtl {
  int *err;
  unsigned limit;
};
__device__ __forceinline__ bool give_up(unsigned &spins, const PollCtl &pc) {
  if (++spins > pc.limit || ((spins & 127u) == 0 && __hip_atomic_load(pc.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
    atomicExch(pc.err, 1);
    return true;
  }
  __builtin_amdgcn_s_sleep(1);
  return false;
}
// N granules at base[at(i)] (those of the bit mask `need`), all loads in flight together; every value is handed to
// sink(i, value, tag) the moment its tag matches -- nothing is kept in registers behind the loads themselves.  A timed-out slot
// is never delivered.  EVERY round issues all N loads (a granule that is not wanted, or has been delivered, is asked for again --
// or the first wanted one in its place): with the loads themselves under per-lane conditions, lanes were handed the value of
// ANOTHER granule of the same round now and then (four neighbouring lanes = one 32-byte sector at a time, caught by comparing
// the LDS copy with the granule it came from: the chunk-1 value in chunk 0's place).  That was a 512-thread build that spilled
// 1.3 kB per lane; the form alone does not misdeliver (tools/ubench_condload.hip: 0 wrong values in 2 x 3000 x 256 gathers of
// 4..32 granules per thread, conditional or not), so the culprit was probably the spill code around the divergent loads -- the
// kernel as it is has no scratch, and keeps the unconditional form.
__device__ __forceinline__ void nap(int n) {
#pragma unroll 1
  for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1);
}
template <int N, class At>
__device__ __forceinline__ void gather_issue(u64 (&v)[N], const u64 *base, unsigned need, At at) {
  const int first = need ? __ffs(need) - 1 : 0;
  unsigned zero = 0u;
  asm volatile("" : "+v"(zero));  // (opaque: the N addresses are formed next to their loads, not kept in 2 N registers across the loop)
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = peek(base + (zero + at(((need >> i) & 1u) ? i : first)));
}
// v: the first round's loads, issued by the caller some work ago (gather_issue)
template <int N, class At, class Sink>
__device__ __forceinline__ unsigned gather_from(u64 (&v)[N], const u64 *base, unsigned want, unsigned need, const PollCtl &pc, At at, Sink sink) {
  unsigned pending = N < 32 ? need & ((1u << (N & 31)) - 1u) : need, spins = 0;
  need = pending;
  while (pending) {
    if (spins) gather_issue<N>(v, base, need, at);
#pragma unroll
    for (int i = 0; i < N; ++i)
      if ((pending >> i) & 1u) {
        const unsigned t = (unsigned)(v[i] >> 32);
        if ((t & ~ACT_BIT) == want) {
          sink(i, __uint_as_float((unsigned)v[i]), t);
          pending &= ~(1u << i);
        }
      }
    if (pending && give_up(spins, pc)) return spins;
  }
  return spins;  // failed rounds
}
template <int N, class At, class Sink>
__device__ __forceinline__ unsigned gather(const u64 *base, unsigned want, unsigned need, const PollCtl &pc, At at, Sink sink) {
  u64 v[N];
  need = N < 32 ? need & ((1u << (N & 31)) - 1u) : need;
  if (need) gather_issue<N>(v, base, need, at);
  return gather_from<N>(v, base, want, need, pc, at, sink);
}
// The same for a write-once slab of plain values: N 16-byte loads at byte offsets at(i) of `slab`, a quad is delivered once none
// of its four words is the fill pattern (they are four dword stores of one producer, or one 16-byte store).  First round sc1,
// retries sc0 sc1.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned UNWRITTEN = 0xffffffffu;
__device__ __forceinline__ unsigned value_bits(float v) {  // what a producer stores: never the fill pattern
  const unsigned b = __float_as_uint(v);
  return b == UNWRITTEN ? 0x7fc00000u : b;
}
__device__ __forceinline__ void put(unsigned *slot, unsigned bits) { __hip_atomic_store(slot, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <int N, class At, class Sink>
__device__ __forceinline__ unsigned gather16(const unsigned *slab, unsigned need, const PollCtl &pc, At at, Sink sink) {
  unsigned pending = need & ((1u << N) - 1u), spins = 0;
  const int first = pending ? __ffs(pending) - 1 : 0;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)slab, 0, 0x7fffffff, 0x00020000);
  while (pending) {
    u32x4 v[N];
    unsigned zero = 0u;
    asm volatile("" : "+v"(zero));
    if (spins == 0) {
#pragma unroll
      for (int i = 0; i < N; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(zero + at(((need >> i) & 1u) ? i : first)), 0, 16);
    } else {
#pragma unroll
      for (int i = 0; i < N; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(zero + at(((need >> i) & 1u) ? i : first)), 0, 17);
    }
#pragma unroll
    for (int i = 0; i < N; ++i)
      if (((pending >> i) & 1u) && v[i].x != UNWRITTEN && v[i].y != UNWRITTEN && v[i].z != UNWRITTEN && v[i].w != UNWRITTEN) {
        sink(i, v[i]);
        pending &= ~(1u << i);
      }
    if (pending && give_up(spins, pc)) return spins;
    asm volatile("" ::: "memory");
  }
  return spins;  // failed rounds
}
__device__ __forceinline__ float4 as_f4(u32x4 v) { return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)); }
__device__ __forceinline__ float4 lds4(const float *p) { return *reinterpret_cast<const float4 *>(p); }


}  // namespace
}  // namespace xdtts
