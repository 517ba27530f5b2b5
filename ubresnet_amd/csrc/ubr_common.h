// Common device/host helpers for the ubresnet_amd HIP kernels (gfx950 / CDNA4 only).
//
// Data model
//   * activations: NHWC ("pixel-major"), channels contiguous, element type T in {f32, bf16, f16};
//     every tensor argument carries explicit element strides so channel slices of a concat
//     buffer and stride-2 phase views are addressed in place.
//   * the unit of channel work is 16 bytes ("unit"): 4 f32 or 8 bf16/f16 channels.  One MFMA
//     K-step consumes 4 units (one per 16-lane quad of the wave):
//        bf16/f16 : 1 x v_mfma_f32_16x16x32_{bf16,f16}
//        f32      : 4 x v_mfma_f32_16x16x4_f32  (element j of every quad's unit), exact fp32
//     so the LDS images and fragment addressing are byte-identical for all three types.
//   * accumulation is always fp32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define UBR_WAVE 64

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;

struct bf16_t { __bf16 v; };
struct f16_t { _Float16 v; };

// ---------------------------------------------------------------------------------------------
// element traits
// ---------------------------------------------------------------------------------------------
template <typename T> struct ET;

template <> struct ET<float> {
  static constexpr int CPU = 4;   // channels per 16-byte unit
  static constexpr int ID = 0;
  static __device__ __forceinline__ void unpack(const uint4& u, float* f) {
    f[0] = __uint_as_float(u.x); f[1] = __uint_as_float(u.y); f[2] = __uint_as_float(u.z); f[3] = __uint_as_float(u.w);
  }
  static __device__ __forceinline__ uint4 pack(const float* f) {
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
  }
  static __device__ __forceinline__ float ld(const void* p, long i) { return ((const float*)p)[i]; }
  static __device__ __forceinline__ void st(void* p, long i, float v) { ((float*)p)[i] = v; }
};

template <> struct ET<bf16_t> {
  static constexpr int CPU = 8;
  static constexpr int ID = 1;
  static __device__ __forceinline__ void unpack(const uint4& u, float* f) {
    f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
    f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
    f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
    f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
  }
  static __device__ __forceinline__ uint32_t pk2(float a, float b) {
    // plain casts: hipcc emits v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN preserving)
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
    bf2 r; r[0] = (__bf16)a; r[1] = (__bf16)b;
    return __builtin_bit_cast(uint32_t, r);
  }
  static __device__ __forceinline__ uint4 pack(const float* f) {
    return make_uint4(pk2(f[0], f[1]), pk2(f[2], f[3]), pk2(f[4], f[5]), pk2(f[6], f[7]));
  }
  static __device__ __forceinline__ float ld(const void* p, long i) {
    return __uint_as_float(((uint32_t)((const uint16_t*)p)[i]) << 16);
  }
  static __device__ __forceinline__ void st(void* p, long i, float v) {
    __bf16 b = (__bf16)v; ((uint16_t*)p)[i] = __builtin_bit_cast(uint16_t, b);
  }
};

template <> struct ET<f16_t> {
  static constexpr int CPU = 8;
  static constexpr int ID = 2;
  static __device__ __forceinline__ void unpack(const uint4& u, float* f) {
    f16x8_t h = __builtin_bit_cast(f16x8_t, u);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)h[i];
  }
  static __device__ __forceinline__ uint4 pack(const float* f) {
    f16x8_t h;
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = (_Float16)f[i];
    return __builtin_bit_cast(uint4, h);
  }
  static __device__ __forceinline__ float ld(const void* p, long i) { return (float)((const _Float16*)p)[i]; }
  static __device__ __forceinline__ void st(void* p, long i, float v) { ((_Float16*)p)[i] = (_Float16)v; }
};

// BatchNorm + ReLU on load, `max((x - mean) * scale + shift, lo)` per channel (lo = 0 or -inf): the same IEEE operations as the
// scalar form -- a subtraction, a fused multiply-add, a max -- issued two channels at a time (v_pk_add_f32 / v_pk_fma_f32) and
// with v_max_f32 written out: for an operand it cannot prove canonical, fmaxf() costs a second v_max_f32 (x, x) in front of
// the real one.  This transform is most of the VALU work of every staging phase, and VALU issue slots are what the two streams
// of a training step compete for.
typedef float ubr_f2 __attribute__((ext_vector_type(2)));
typedef unsigned ubr_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float ubr_vmax(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// Wave priority of the compute stream's kernels (s_setprio: instruction arbitration among the waves of a SIMD).  The weight-gradient
// kernels of the side stream keep priority 0.
#ifndef UBR_MAIN_PRIO
#define UBR_MAIN_PRIO 3
#endif
__device__ __forceinline__ void ubr_main_prio() {
  if constexpr (UBR_MAIN_PRIO > 0) __builtin_amdgcn_s_setprio(UBR_MAIN_PRIO);
}

template <int N>
__device__ __forceinline__ void ubr_bnrelu(float* f, const float* sub, const float* sc, const float* sh, const float* lo) {
#pragma unroll
  for (int e = 0; e < N; e += 2) {
    ubr_f2 a = {f[e], f[e + 1]};
    a = a - ubr_f2{sub[e], sub[e + 1]};
    a = __builtin_elementwise_fma(a, ubr_f2{sc[e], sc[e + 1]}, ubr_f2{sh[e], sh[e + 1]});
    f[e] = ubr_vmax(a[0], lo[e]);
    f[e + 1] = ubr_vmax(a[1], lo[e + 1]);
  }
}

// One K-step (4 units across the 4 lane quads): acc[16 x 16] += A(16 x K) * B(K x 16).
// a = this lane's unit of the A operand (row = lane&15), b = of the B operand (col = lane&15).
// C/D map (all types): col = lane & 15, row = (lane >> 4) * 4 + reg.
template <typename T> __device__ __forceinline__ f32x4 mma_step(f32x4 acc, const uint4& a, const uint4& b);

template <> __device__ __forceinline__ f32x4 mma_step<float>(f32x4 acc, const uint4& a, const uint4& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
  return acc;
}
template <> __device__ __forceinline__ f32x4 mma_step<bf16_t>(f32x4 acc, const uint4& a, const uint4& b) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4 mma_step<f16_t>(f32x4 acc, const uint4& a, const uint4& b) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint4 ldg16(const void* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ void stg16(void* p, const uint4& v) { *reinterpret_cast<uint4*>(p) = v; }

// fp64 accumulators that many workgroups add into are striped over UBR_STAT_SLOTS copies
// (slot = blockIdx.x % UBR_STAT_SLOTS) and summed by the finalize kernels: thousands of same-address
// atomics serialise at the memory side (a 16-channel layer's statistics were 60 % of its conv time).
#define UBR_STAT_SLOTS 32

template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_quadrow_sum16(float v) {
  // sum over the 16 lanes that share lane>>4, on the DPP path (no LDS traffic): every lane ends with the total
  v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);   // row_half_mirror
  v += dpp_mov<0x140>(v);   // row_mirror
  return v;
}
__device__ __forceinline__ float wave_sum64(float v) {
  v = wave_quadrow_sum16(v);
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ double wave_sum64d(double v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
  return v;
}

static inline int ubr_cdiv(int a, int b) { return (a + b - 1) / b; }
