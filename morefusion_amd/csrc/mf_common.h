// Shared device/host helpers for libmfhip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mfhip.h"

namespace mf {

void set_last_error(hipError_t e, const char *where);

inline int check_launch(const char *where) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_last_error(e, where);
    return -(int)e;
  }
  return 0;
}

#define MF_TRY(expr)                                  \
  do {                                                \
    hipError_t _e = (expr);                           \
    if (_e != hipSuccess) {                           \
      ::mf::set_last_error(_e, #expr);                \
      return -(int)_e;                                \
    }                                                 \
  } while (0)

// Opt a kernel in to more than 64 KB of dynamic LDS.  The attribute is per DEVICE (one process
// may drive several GPUs) and this may be called from any thread: api.hip keeps a mutex-guarded
// (device, function) set.
int allow_big_lds(const void *kernel, int bytes);

// Fill ``nbytes`` (a multiple of 4) at a 4-byte aligned address with the byte ``value``, by a KERNEL on ``stream``.
// The library never calls hipMemsetAsync: captured into a hipGraph it becomes a memset NODE, and on this stack
// (ROCm 7.0/7.2 runtime) the kernels behind such a node were seen to start before the fill had landed when the
// graph is replayed (tools/repro_graph_memset.hip) -- stale per-voxel chain heads then walk out of
// the point arrays.  A kernel node is ordered like every other kernel of the captured stream.
int fill_bytes(void *dst, int value, int64_t nbytes, hipStream_t stream);

constexpr int kWave = 64;

// Opaque use of a staged 16-byte register at a program point (MF_HOLD(r) after a block of MFMAs): without it the
// compiler hoists arithmetic on a prefetched value -- and with it the wait for the load -- to right behind the load,
// in front of the MFMAs the load is meant to overlap.  (tests/host_emul/mf_common.h: a no-op.)
#define MF_HOLD(r_) asm volatile("" : "+v"(r_.x), "+v"(r_.y), "+v"(r_.z), "+v"(r_.w))

// ---- buffer loads with a hardware range check (round 4: the masked operand chunks of csrc/gemm_bf16.hip) -----------
// A raw buffer resource over [p, p + kBufSpan): buf_load16(rs, off) is one buffer_load_dwordx4 at byte offset `off`;
// an offset >= kBufSpan -- kBufMasked -- is out of range and the hardware returns ZEROS without touching memory.
// A masked lane therefore costs one v_cndmask on its 32-bit offset: no pointer select, no AND on the loaded data.
// (Real offsets must stay below 2^31 bytes: the callers' tensors do.)
constexpr uint32_t kBufSpan = 0x80000000u, kBufMasked = 0x80000000u;
struct BufRsrc {
  __amdgpu_buffer_rsrc_t r;
};
__device__ __forceinline__ BufRsrc make_rsrc(const void *p) {
  BufRsrc b;
  b.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, kBufSpan, 0x00020000);
  return b;
}
__device__ __forceinline__ uint4 buf_load16(const BufRsrc &b, uint32_t byte_off) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(b.r, (int)byte_off, 0, 0);
  return make_uint4(v.x, v.y, v.z, v.w);
}

// ---- LDS-DMA and hand-counted waits (round 6: k_gemm_nt_bf16_pp of csrc/gemm_bf16.hip) ---------------------------
// glds16(rs, off, dst): ONE buffer_load_dwordx4 ... lds -- every lane's 16 bytes at byte offset `off` of the resource
// go straight into LDS at dst + 16 * lane (dst wave-uniform: it travels in M0), no VGPR in between; an out-of-range
// offset (kBufMasked) lands ZEROS.  The transfer counts on the wave's vmcnt like any load; nothing else orders it:
// a reader needs the issuing wave's wait_dma(), then a barrier it has passed.
// lds_read16_async<IMM>(addr): ds_read_b128 as inline asm.  The compiler waits vmcnt(0) in front of every LDS read
// it can see while an LDS-DMA is in flight (it cannot tell which bytes the DMA writes); it does not look into asm.
// The result is valid only behind wait_lds_reads() -- the caller's obligation, the compiler does not track it -- and
// its register must stay LIVE up to that wait (use every result behind it, e.g. MF_HOLD): the compiler takes the
// register as written when the asm statement ends and re-uses a dead one while the data is still on its way.
typedef uint32_t lds_addr_t;  // byte address inside the workgroup's LDS
__device__ __forceinline__ lds_addr_t lds_addr(const void *p) {
  return (lds_addr_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}
__device__ __forceinline__ void glds16(const BufRsrc &b, uint32_t byte_off, unsigned char *lds_wave_base) {
  typedef __attribute__((address_space(3))) void *lds_void;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(b.r, (lds_void)lds_wave_base, 16, (int)byte_off, 0, 0, 0);
}
template <int IMM>
__device__ __forceinline__ uint4 lds_read16_async(lds_addr_t addr) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM));
  return make_uint4(v.x, v.y, v.z, v.w);
}
// The same for an MFMA operand that comes out of gfx950's transposing read (lds_read_tr16_b64x2 below): the two
// ds_read_b64_tr_b16 at addr + IMM and addr + IMM + 2048 write the low and the high half of one 16-byte register.
template <int IMM>
__device__ __forceinline__ uint4 lds_read_tr16_x2_async(lds_addr_t addr) {
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  u32x2 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "n"(IMM));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(IMM + 2048));
  return make_uint4(lo.x, lo.y, hi.x, hi.y);
}
__device__ __forceinline__ void wait_lds_reads() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);  // (register-only instructions would be hoisted over the asm wait)
}
template <int N>
__device__ __forceinline__ void wait_dma() {  // until at most N of this wave's DMA requests (the latest N) are pending
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// s_barrier alone: no fence, no counter drained (__syncthreads() waits vmcnt(0) while an LDS-DMA is pending)
__device__ __forceinline__ void raw_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// ---- gfx950's transposing LDS read (round 4: the TN engine of csrc/gemm_bf16.hip) -------------------------------
// ds_read_b64_tr_b16: every lane gives the address of 4 contiguous 16-bit elements; within each group of 16 lanes the
// 16 x 4 elements are taken as a matrix [4 rows][16 columns] -- row r = lanes 4 r .. 4 r + 3 of the group, in lane
// order -- and lane c of the group receives column c: element r of its result = element (c & 3) of lane 4 r + (c >> 2).
// lds_read_tr16_b64x2(p, d): the reads at p and p + d as one MFMA operand (8 values: rows 0..3 of each block).
__device__ __forceinline__ uint4 lds_read_tr16_b64x2(const unsigned char *p, int second) {
  typedef short v4i16 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) v4i16 *lds_v4i16;
  const v4i16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16)(p));
  const v4i16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16)(p + second));
  const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
  return make_uint4(l2.x, l2.y, h2.x, h2.y);
}

// Touch every 64-byte line of the kernel-argument segment with ONE batch of scalar loads.  A kernel with a few hundred
// bytes of by-value arguments and high SGPR pressure re-reads its arguments piecemeal (s_load ... s_waitcnt pairs
// all through its prologue); the scalar cache is cold at kernel start, so each first touch of a line is a memory
// round trip on the critical path -- after this they hit.  (tests/host_emul/mf_common.h: a no-op.)
__device__ __forceinline__ void warm_kernargs(int bytes) {
  typedef const uint32_t __attribute__((address_space(4))) *kptr_t;
  kptr_t kp = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
  uint32_t acc = 0;
#pragma unroll
  for (int off = 0; off < 1024; off += 64)
    if (off < bytes) acc |= kp[off / 4];
  asm volatile("" ::"s"(acc));
}

// fp32 add to global memory as ONE hardware atomic (global_atomic_add_f32, no return value, device scope) -- plain
// atomicAdd(float*) compiles to a compare-and-swap loop without -munsafe-fp-atomics
__device__ __forceinline__ void atomic_add_f32(float *p, float v) { unsafeAtomicAdd(p, v); }

// dynamic LDS of the workgroup (tests/host_emul/mf_common.h gives the host-emulation form)
#define MF_DYN_LDS(type, name) extern __shared__ __attribute__((aligned(16))) type name[]

__device__ __forceinline__ int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- bf16 (round 4: csrc/gemm_bf16.hip and the bf16 point / voxel kernels) -----------------------------------
// Eight bf16 travel as one uint4 (16 bytes: one global / LDS access, one MFMA operand).
typedef float mf_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 mf_bf16x8_t __attribute__((ext_vector_type(8)));
// v_mfma_f32_32x32x16_bf16: D[i][j] += sum_k A[i][k] B[k][j]; lane l supplies A[l % 32][8 (l / 32) .. + 7] and
// B[8 (l / 32) .. + 7][l % 32] (element e of the uint4 = k offset e, low half-word first) and holds
// D[(e & 3) + 8 (e >> 2) + 4 (l / 32)][l % 32] in accumulator element e.
__device__ __forceinline__ mf_f32x16 mfma_bf16_32x32x16(uint4 a, uint4 b, mf_f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mf_bf16x8_t, a), __builtin_bit_cast(mf_bf16x8_t, b),
                                                 c, 0, 0, 0);
}
// float -> bf16, round to nearest even (NaN stays NaN): what torch's .to(torch.bfloat16) does
__device__ __forceinline__ uint32_t bf16_bits(float f) {
  return (uint32_t)__builtin_bit_cast(unsigned short, (__bf16)f);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) { return bf16_bits(lo) | (bf16_bits(hi) << 16); }
__device__ __forceinline__ float bf16_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// round((p - o) / pitch), CUDA round() == roundf(): half away from zero.
// Division stays a correctly-rounded IEEE divide (no reciprocal tricks): voxel
// indices must be bit-identical to the oracle.
__device__ __forceinline__ float voxel_coord(float p, float o, float pitch) {
  return (p - o) / pitch;
}

// ---- cross-lane reductions on DPP (one VALU op per step; __shfl_* compile to ds_bpermute,
// an LDS-crossbar round trip per step).  All lanes of the wave must be active.
// DPP controls: quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, row_half_mirror = 0x141,
// row_mirror = 0x140 -- a butterfly over each aligned group of 16 lanes (a DPP "row").
template <int CTRL>
__device__ __forceinline__ float dpp_zero(float v) {  // inactive / invalid source lanes read 0
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ float dpp_self(float v) {  // inactive / invalid source lanes read v
  return __int_as_float(
      __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xf, 0xf, false));
}

// Sum over each 16-lane row; every lane of the row receives the same bits
// (((v + v^1) + (..)^2) + half-mirror) + mirror: a fixed association).
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_zero<0xB1>(v);
  v += dpp_zero<0x4E>(v);
  v += dpp_zero<0x141>(v);
  v += dpp_zero<0x140>(v);
  return v;
}

__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_self<0xB1>(v));
  v = fmaxf(v, dpp_self<0x4E>(v));
  v = fmaxf(v, dpp_self<0x141>(v));
  v = fmaxf(v, dpp_self<0x140>(v));
  return v;
}

__device__ __forceinline__ float lane_value(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// Wave-wide sum / max, result in EVERY lane: (row0 + row1) + (row2 + row3).
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}

__device__ __forceinline__ float wave_max(float v) {
  v = row16_max(v);
  return fmaxf(fmaxf(lane_value(v, 0), lane_value(v, 16)), fmaxf(lane_value(v, 32), lane_value(v, 48)));
}

}  // namespace mf
