// HBM-bound kernels of the U-ResNet path: BatchNorm finalize/backward, BasicBlock tail
// (relu(bn2)+shortcut+relu, models/common_layers.py:47-56) forward/backward, MaxPool2d(3,s,1)
// forward/backward, channel sums.  NHWC, 16 bytes per lane per access; every thread keeps ONE
// 16-byte channel unit for the whole launch (grid size is made a multiple of the units per
// pixel), so per-channel constants live in registers and pixel indices need no division in the loop.
// Per-channel reductions: registers -> fp64 LDS atomics -> fp64 global atomics.
#include "ubr_common.h"
#include "ubr_host.h"
#include <float.h>
#include <string.h>

namespace {

static int gcd_i(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; }

// blocks such that blocks*256 is a multiple of CU and the grid covers the work reasonably
static int pick_blocks(int64_t npix, int CU, int max_blocks = 2048, int min_iters = 1) {
  const int mult = CU / gcd_i(256, CU);
  int64_t want = (npix * CU + 255) / 256;
  // reduction passes end with one global fp64 atomic per channel and sum in EVERY block: on the low-resolution layers a thread
  // per 16-byte unit meant 1024 blocks x 1024 atomics behind four pixels of work each; eight units per thread there
  // (bn_bwd reduce at 16x16x512: 13 -> 8 us)
  if (min_iters > 1) want = (want + min_iters - 1) / min_iters;
  if (want > max_blocks) want = max_blocks;
  if (want < 1) want = 1;
  int64_t b = ((want + mult - 1) / mult) * mult;
  return (int)b;
}

template <typename T> struct UnitIdx {
  int c;          // channel unit of this thread
  long p, pstep;  // first pixel and pixel step
  __device__ UnitIdx(int CU) {
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    const long Tn = (long)gridDim.x * 256;
    c = (int)(g % CU);
    p = g / CU;
    pstep = Tn / CU;
  }
};

template <typename T> __device__ __forceinline__ void ldunit(const void* base, long pix, long ps, int c, float* f) {
  constexpr int CPU = ET<T>::CPU;
  const uint4 v = ldg16((const char*)base + (pix * ps + (long)c * CPU) * (16 / CPU));
  ET<T>::unpack(v, f);
}
template <typename T> __device__ __forceinline__ void stunit(void* base, long pix, long ps, int c, const float* f) {
  constexpr int CPU = ET<T>::CPU;
  stg16((char*)base + (pix * ps + (long)c * CPU) * (16 / CPU), ET<T>::pack(f));
}
// One NHWC tensor walked by a thread at its fixed channel unit: a buffer resource over the tensor and a 32-bit byte offset
// that advances by a constant per iteration -- one add and one buffer instruction per access, where base + (pix * ps + c) * esz
// costs a 64-bit multiply-add per tensor and iteration.  These kernels are HBM-bound on their own, but their VALU / SALU issue
// slots are what the weight-gradient stream beside them runs on (profiles/r02_instruction_counts.txt: a quarter of all
// VALU instructions of a train step were block-tail / BatchNorm backward).  Tensors are < 2 GiB (checked by the host side).
template <typename T> struct UnitStream {
  __amdgpu_buffer_rsrc_t r;
  unsigned off, step;
  __device__ __forceinline__ UnitStream(const void* base, long npix, long ps, const UnitIdx<T>& ix) {
    constexpr unsigned ESZ = 16 / ET<T>::CPU;
    r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, base != nullptr ? (int)(npix * ps * ESZ) : 0, 0x00020000);
    off = (unsigned)ix.p * (unsigned)ps * ESZ + (unsigned)ix.c * 16u;
    step = (unsigned)ix.pstep * (unsigned)ps * ESZ;
  }
  __device__ __forceinline__ void ld(float* f) const {
    const ubr_u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
    ET<T>::unpack(make_uint4(v.x, v.y, v.z, v.w), f);
  }
  __device__ __forceinline__ void st(const float* f) const {
    const uint4 v = ET<T>::pack(f);
    __builtin_amdgcn_raw_buffer_store_b128(ubr_u4{v.x, v.y, v.z, v.w}, r, (int)off, 0, 0);
  }
  __device__ __forceinline__ void next() { off += step; }
};
template <int CPU> __device__ __forceinline__ void ldconst2(const float* a, int c, ubr_f2* f) {
#pragma unroll
  for (int e = 0; e < CPU; e += 2) f[e / 2] = ubr_f2{a[c * CPU + e], a[c * CPU + e + 1]};
}

template <int CPU> __device__ __forceinline__ void ldconst(const float* a, int c, float* f) {
#pragma unroll
  for (int e = 0; e < CPU; ++e) f[e] = a[c * CPU + e];
}

// flush per-thread channel partials: LDS fp64 atomics, then one global fp64 atomic per channel
template <int CPU, int NQ>
__device__ __forceinline__ void flush_sums(const float (*acc)[CPU], int c, int C, double* lds, double* const* outs) {
  // outs[qn] already point into this block's slot (callers add blockIdx.x % UBR_STAT_SLOTS)
  for (int i = threadIdx.x; i < NQ * C; i += 256) lds[i] = 0.0;
  __syncthreads();
#pragma unroll
  for (int qn = 0; qn < NQ; ++qn)
#pragma unroll
    for (int e = 0; e < CPU; ++e) atomicAdd(&lds[qn * C + c * CPU + e], (double)acc[qn][e]);
  __syncthreads();
  for (int i = threadIdx.x; i < NQ * C; i += 256) {
    const int qn = i / C, ch = i - qn * C;
    if (outs[qn] != nullptr) atomicAdd(&outs[qn][ch], lds[i]);
  }
}

// The same flush for power-of-two unit counts (every U-ResNet layer): lanes that hold the same channel unit are summed in
// registers first (butterfly over the lane bits above log2(CU)), each wave writes ONE partial per channel to its own LDS row,
// and the four rows are added in a fixed order -- no LDS atomics.  With CU = 2 (16 channels) the atomic form issued 32 fp64 LDS
// atomics per thread, each a 32-way same-address conflict inside the wave.
template <int CPU, int NQ>
__device__ __forceinline__ void flush_sums_pow2(float (*acc)[CPU], int c, int CU, int C, float* lds, double* const* outs) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int m = CU; m < 64; m <<= 1) {
#pragma unroll
    for (int qn = 0; qn < NQ; ++qn)
#pragma unroll
      for (int e = 0; e < CPU; ++e) acc[qn][e] += __shfl_xor(acc[qn][e], m, 64);
  }
  // rows: [wave][qn][C]; with CU > 64 a wave covers only 64 of the units: rows are zeroed first
  if (CU > 64) {
    for (int i = threadIdx.x; i < 4 * NQ * C; i += 256) lds[i] = 0.f;
    __syncthreads();
  }
  if (lane < CU) {
#pragma unroll
    for (int qn = 0; qn < NQ; ++qn)
#pragma unroll
      for (int e = 0; e < CPU; ++e) lds[(wave * NQ + qn) * C + c * CPU + e] = acc[qn][e];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NQ * C; i += 256) {
    const int qn = i / C, ch = i - qn * C;
    if (outs[qn] != nullptr) {
      const double v = ((double)lds[(0 * NQ + qn) * C + ch] + (double)lds[(1 * NQ + qn) * C + ch]) +
                       ((double)lds[(2 * NQ + qn) * C + ch] + (double)lds[(3 * NQ + qn) * C + ch]);
      atomicAdd(&outs[qn][ch], v);
    }
  }
}

#ifdef UBR_TUNE
static int g_tune_red_blocks = 0, g_tune_red_iters = 0, g_tune_flush = 1, g_tune_app_blocks = 0, g_tune_slots = 0;
extern "C" void ubr_tune_set(const char* key, int v) {
  if (!strcmp(key, "red_blocks")) g_tune_red_blocks = v;
  if (!strcmp(key, "red_iters")) g_tune_red_iters = v;
  if (!strcmp(key, "flush")) g_tune_flush = v;
  if (!strcmp(key, "app_blocks")) g_tune_app_blocks = v;
  if (!strcmp(key, "slots")) g_tune_slots = v;
}
#endif

// ------------------------------------------------------------------------------------------
// Streaming structure shared by the block-tail / BatchNorm kernels below.  A thread walks its channel unit down the pixels in
// trips of UNR pixels: every 16-byte load of a trip (UNR pixels x 2-4 tensors) is issued before the first value is unpacked,
// so a wave keeps 8-16 loads in flight instead of one or two.  (The first form of these kernels tested optional operands --
// second gradient, bypass branch, mask -- with wave-uniform branches INSIDE the loop; the compiler then waited for each load
// before the next branch, and the reduce passes ran at a third of HBM speed while their own apply passes, with more waves to
// hide it, reached 60 %.)  Optional operands are template parameters here; a pixel beyond the end loads from an out-of-range
// buffer offset (zeros) and its stores are dropped by the same range check, so trips need no remainder loop.
// ------------------------------------------------------------------------------------------
constexpr int kUnitOOR = (int)0x80000000;     // beyond any num_records: loads return zero, stores are dropped

template <typename T, int UNR> struct UnitTrip {
  __amdgpu_buffer_rsrc_t r;
  unsigned off, step;
  __device__ __forceinline__ UnitTrip(const void* base, long npix, long ps, const UnitIdx<T>& ix) {
    constexpr unsigned ESZ = 16 / ET<T>::CPU;
    r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, base != nullptr ? (int)(npix * ps * ESZ) : 0, 0x00020000);
    off = (unsigned)ix.p * (unsigned)ps * ESZ + (unsigned)ix.c * 16u;
    step = (unsigned)ix.pstep * (unsigned)ps * ESZ;
  }
  __device__ __forceinline__ void ld(ubr_u4 (&v)[UNR], const bool (&ok)[UNR]) const {
#pragma unroll
    for (int u = 0; u < UNR; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(r, ok[u] ? (int)(off + (unsigned)u * step) : kUnitOOR, 0, 0);
  }
  __device__ __forceinline__ void st(int u, const float* f, bool ok) const {
    const uint4 v = ET<T>::pack(f);
    __builtin_amdgcn_raw_buffer_store_b128(ubr_u4{v.x, v.y, v.z, v.w}, r, ok ? (int)(off + (unsigned)u * step) : kUnitOOR, 0, 0);
  }
  __device__ __forceinline__ void next() { off += (unsigned)UNR * step; }
};
// the ReLU bit mask of a block tail: one byte per (pixel, unit)
template <typename T, int UNR> struct MaskTrip {
  __amdgpu_buffer_rsrc_t r;
  unsigned off, step;
  __device__ __forceinline__ MaskTrip(const uint8_t* base, long npix, int CU, const UnitIdx<T>& ix) {
    r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(base), 0, base != nullptr ? (int)(npix * CU) : 0, 0x00020000);
    off = (unsigned)ix.p * (unsigned)CU + (unsigned)ix.c;
    step = (unsigned)ix.pstep * (unsigned)CU;
  }
  __device__ __forceinline__ void ld(unsigned (&m)[UNR], const bool (&ok)[UNR]) const {
#pragma unroll
    for (int u = 0; u < UNR; ++u) m[u] = __builtin_amdgcn_raw_buffer_load_b8(r, ok[u] ? (int)(off + (unsigned)u * step) : kUnitOOR, 0, 0);
  }
  __device__ __forceinline__ void st(int u, unsigned m, bool ok) const {
    __builtin_amdgcn_raw_buffer_store_b8((unsigned char)m, r, ok ? (int)(off + (unsigned)u * step) : kUnitOOR, 0, 0);
  }
  __device__ __forceinline__ void next() { off += (unsigned)UNR * step; }
};
template <typename T> __device__ __forceinline__ void unpack4(const ubr_u4& v, float* f) { ET<T>::unpack(make_uint4(v.x, v.y, v.z, v.w), f); }

// ------------------------------------------------------------------------------------------
// BasicBlock tail forward
// ------------------------------------------------------------------------------------------
// train-mode BatchNorm finalize of one site, fused into its consumer (ubr_block_tail_fwd_fin): the producing conv's striped
// fp64 sums in, the site's vectors out (written by workgroup 0; every workgroup computes its own copy into LDS)
struct BnFwdFin {
  const double* stats; const float *gamma, *beta;
  float *rmean, *rvar; long long* nbt;
  float momentum, eps;
  float *scale, *shift, *mean, *invstd;
};
struct TailF {
  long npix; int CU, C;
  const void *c2, *sc; void* out; long c2_ps, sc_ps, out_ps;
  const float *m2, *s2, *t2, *mb, *sb, *tb;
  uint8_t* relu_mask;
  BnFwdFin f2, fb;               // fused finalize (f2.stats != nullptr): m2/s2/t2 (mb/sb/tb) are not read
  double count; int nslots;
};
// -> kk[0..C) mean, kk[C..2C) scale, kk[2C..3C) shift.  Same arithmetic as bn_finalize_kernel.
__device__ __forceinline__ void fwd_fin_site(const BnFwdFin& f, int nslots, int C, double count, float* kk, float mom_cum) {
  for (int c = threadIdx.x; c < C; c += 256) {
    double s1 = 0.0, s2 = 0.0;
    if (nslots == UBR_RED_SLOTS) {
      double a[UBR_RED_SLOTS], b[UBR_RED_SLOTS];
#pragma unroll
      for (int sl = 0; sl < UBR_RED_SLOTS; ++sl) { a[sl] = f.stats[(size_t)sl * 2 * C + c]; b[sl] = f.stats[(size_t)sl * 2 * C + C + c]; }
#pragma unroll
      for (int sl = 0; sl < UBR_RED_SLOTS; ++sl) { s1 += a[sl]; s2 += b[sl]; }
    } else {
      for (int sl = 0; sl < nslots; ++sl) { s1 += f.stats[(size_t)sl * 2 * C + c]; s2 += f.stats[(size_t)sl * 2 * C + C + c]; }
    }
    const double m = s1 / count;
    double var = s2 / count - m * m;
    if (var < 0.0) var = 0.0;
    const double is = 1.0 / sqrt(var + (double)f.eps);
    const float sc = (float)((double)f.gamma[c] * is);
    kk[c] = (float)m; kk[C + c] = sc; kk[2 * C + c] = f.beta[c];
    if (blockIdx.x == 0) {
      f.scale[c] = sc; f.shift[c] = f.beta[c]; f.mean[c] = (float)m; f.invstd[c] = (float)is;
      if (f.rmean != nullptr) {
        const double momentum = f.momentum < 0.f ? (double)mom_cum : (double)f.momentum;
        const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        f.rmean[c] = (float)((1.0 - momentum) * (double)f.rmean[c] + momentum * m);
        f.rvar[c] = (float)((1.0 - momentum) * (double)f.rvar[c] + momentum * unb);
      }
    }
  }
}
template <typename T, bool BYP, bool MASK>
__global__ __launch_bounds__(256) void tail_fwd_kernel(const TailF k) {
  ubr_main_prio();
  constexpr int CPU = ET<T>::CPU;
  constexpr int H2 = CPU / 2;
  constexpr int UNR = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  UnitIdx<T> ix(k.CU);
  ubr_f2 a2[H2], b2[H2], m2[H2], ab[H2], bb[H2], mb[H2];
  if (k.f2.stats != nullptr) {
    float* kk = reinterpret_cast<float*>(smem);
    // (momentum = None: cumulative average; only workgroup 0 touches the counters)
    float cum2 = 0.f, cumb = 0.f;
    if (blockIdx.x == 0) {
      if (k.f2.nbt != nullptr) cum2 = (float)(1.0 / (double)(*k.f2.nbt + 1));
      if (BYP && k.fb.nbt != nullptr) cumb = (float)(1.0 / (double)(*k.fb.nbt + 1));
      __syncthreads();
      if (threadIdx.x == 0) {
        if (k.f2.nbt != nullptr) *k.f2.nbt += 1;
        if (BYP && k.fb.nbt != nullptr) *k.fb.nbt += 1;
      }
    }
    fwd_fin_site(k.f2, k.nslots, k.C, k.count, kk, cum2);
    if (BYP) fwd_fin_site(k.fb, k.nslots, k.C, k.count, kk + 3 * k.C, cumb);
    __syncthreads();
    ldconst2<CPU>(kk, ix.c, m2); ldconst2<CPU>(kk + k.C, ix.c, a2); ldconst2<CPU>(kk + 2 * k.C, ix.c, b2);
    if (BYP) { ldconst2<CPU>(kk + 3 * k.C, ix.c, mb); ldconst2<CPU>(kk + 4 * k.C, ix.c, ab); ldconst2<CPU>(kk + 5 * k.C, ix.c, bb); }
  } else {
    ldconst2<CPU>(k.s2, ix.c, a2); ldconst2<CPU>(k.t2, ix.c, b2); ldconst2<CPU>(k.m2, ix.c, m2);
    if (BYP) { ldconst2<CPU>(k.sb, ix.c, ab); ldconst2<CPU>(k.tb, ix.c, bb); ldconst2<CPU>(k.mb, ix.c, mb); }
  }
  UnitTrip<T, UNR> C2(k.c2, k.npix, k.c2_ps, ix), SC(k.sc, k.npix, k.sc_ps, ix), OUT(k.out, k.npix, k.out_ps, ix);
  MaskTrip<T, UNR> MK(k.relu_mask, k.npix, k.CU, ix);
  const float zero = 0.f;
  for (long p = ix.p; p < k.npix; p += UNR * ix.pstep) {
    bool ok[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) ok[u] = p + u * ix.pstep < k.npix;
    ubr_u4 rv[UNR], rs[UNR];
    C2.ld(rv, ok); SC.ld(rs, ok);
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      float v[CPU], s[CPU], o[CPU];
      unpack4<T>(rv[u], v); unpack4<T>(rs[u], s);
#pragma unroll
      for (int h = 0; h < H2; ++h) {
        // relu(bn2(c2)) + shortcut (through its own BatchNorm on a bypass block), relu: same operations as the scalar form
        const ubr_f2 bn = __builtin_elementwise_fma(ubr_f2{v[2 * h], v[2 * h + 1]} - m2[h], a2[h], b2[h]);
        const ubr_f2 r2 = {ubr_vmax(bn[0], zero), ubr_vmax(bn[1], zero)};
        ubr_f2 sh = {s[2 * h], s[2 * h + 1]};
        if (BYP) sh = __builtin_elementwise_fma(sh - mb[h], ab[h], bb[h]);
        const ubr_f2 t = r2 + sh;
        o[2 * h] = ubr_vmax(t[0], zero); o[2 * h + 1] = ubr_vmax(t[1], zero);
      }
      OUT.st(u, o, ok[u]);
      if (MASK) {
        // one bit per channel of this unit: "the STORED output is positive" (after rounding to the storage type, which is what the
        // backward pass used to test on the tensor itself).  The backward's two passes then read 1 byte per unit instead of 16.
        float r[CPU];
        ET<T>::unpack(ET<T>::pack(o), r);
        unsigned m = 0u;
#pragma unroll
        for (int e = 0; e < CPU; ++e) m |= (r[e] > 0.f ? 1u : 0u) << e;
        MK.st(u, m, ok[u]);
      }
    }
    C2.next(); SC.next(); OUT.next(); MK.next();
  }
}

// ------------------------------------------------------------------------------------------
// BasicBlock tail backward
// ------------------------------------------------------------------------------------------
struct TailB {
  long npix; int C, CU;
  const void *go, *go2, *out, *c2, *cb;
  long go_ps, go2_ps, out_ps, c2_ps, cb_ps;
  const float *s2, *t2, *m2, *i2, *k1_2, *k2_2;
  const float *sb, *mb, *ib, *k1_b, *k2_b;
  double *red2, *redb;
  void *g_c2, *g_sc; long g_c2_ps, g_sc_ps;
  const uint8_t* relu_mask;      // optional: bit e of byte [pixel][unit] = (out > 0) for channel e of the unit; replaces the read of `out`
  int flush;                     // reduce pass: 1 = register butterfly + per-wave LDS rows (flush_sums_pow2), 0 = LDS fp64 atomics
  int nslots;                    // reduce pass: stripes of the fp64 accumulators in use (slot = blockIdx.x % nslots)
  // apply pass, fused finalize (fin_red2 != nullptr): every workgroup sums the reduce pass's stripes itself (k1 = sum / count,
  // k2 likewise), workgroup 0 also writes dgamma / dbeta -- no ubr_bn_bwd_finalize launch between the two passes
  const double *fin_red2, *fin_redb; double count;
  float *dgamma2, *dbeta2, *dgamma_b, *dbeta_b;
};

// k1 / k2 of one BatchNorm site from the striped sums [slot][2 C] (sum g | sum g*xhat), into LDS floats kk[0..C) = k1, kk[C..2C) = k2
__device__ __forceinline__ void fin_site(const double* red, int nslots, int C, double count, float* kk, float* dgamma, float* dbeta) {
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    double a = 0.0;
    if (nslots == UBR_RED_SLOTS) {        // all stripe loads in flight at once (a runtime trip count makes them dependent round trips)
      double v[UBR_RED_SLOTS];
#pragma unroll
      for (int sl = 0; sl < UBR_RED_SLOTS; ++sl) v[sl] = red[(size_t)sl * 2 * C + i];
#pragma unroll
      for (int sl = 0; sl < UBR_RED_SLOTS; ++sl) a += v[sl];
    } else {
      for (int sl = 0; sl < nslots; ++sl) a += red[(size_t)sl * 2 * C + i];
    }
    kk[i] = (float)(a / count);
    if (blockIdx.x == 0) {
      if (i < C) { if (dbeta != nullptr) dbeta[i] = (float)a; }
      else if (dgamma != nullptr) dgamma[i - C] = (float)a;
    }
  }
}

template <typename T, bool APPLY, bool BYP, bool HAS_GO2, bool MASK>
__global__ __launch_bounds__(256) void tail_bwd_kernel(const TailB k) {
  ubr_main_prio();
  constexpr int CPU = ET<T>::CPU;
  constexpr int H2 = CPU / 2;
  constexpr int UNR = APPLY ? 2 : 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  UnitIdx<T> ix(k.CU);
  ubr_f2 s2[H2], t2[H2], m2[H2], i2[H2], mb[H2], ib[H2];
  ldconst2<CPU>(k.s2, ix.c, s2); ldconst2<CPU>(k.t2, ix.c, t2); ldconst2<CPU>(k.m2, ix.c, m2); ldconst2<CPU>(k.i2, ix.c, i2);
  if (BYP) { ldconst2<CPU>(k.mb, ix.c, mb); ldconst2<CPU>(k.ib, ix.c, ib); }
  ubr_f2 k12[H2], k22[H2], k1b[H2], k2b[H2], sb[H2];
  if (APPLY) {
    if (k.fin_red2 != nullptr) {
      float* kk = reinterpret_cast<float*>(smem);
      fin_site(k.fin_red2, k.nslots, k.C, k.count, kk, k.dgamma2, k.dbeta2);
      if (BYP) fin_site(k.fin_redb, k.nslots, k.C, k.count, kk + 2 * k.C, k.dgamma_b, k.dbeta_b);
      __syncthreads();
      ldconst2<CPU>(kk, ix.c, k12); ldconst2<CPU>(kk + k.C, ix.c, k22);
      if (BYP) { ldconst2<CPU>(kk + 2 * k.C, ix.c, k1b); ldconst2<CPU>(kk + 3 * k.C, ix.c, k2b); }
    } else {
      ldconst2<CPU>(k.k1_2, ix.c, k12); ldconst2<CPU>(k.k2_2, ix.c, k22);
      if (BYP) { ldconst2<CPU>(k.k1_b, ix.c, k1b); ldconst2<CPU>(k.k2_b, ix.c, k2b); }
    }
    if (BYP) ldconst2<CPU>(k.sb, ix.c, sb);
  }
  float acc[4][CPU];
#pragma unroll
  for (int qn = 0; qn < 4; ++qn)
#pragma unroll
    for (int e = 0; e < CPU; ++e) acc[qn][e] = 0.f;
  UnitTrip<T, UNR> GO(k.go, k.npix, k.go_ps, ix), GO2(k.go2, k.npix, k.go2_ps, ix), OUT(k.out, k.npix, k.out_ps, ix),
      C2(k.c2, k.npix, k.c2_ps, ix), CB(k.cb, k.npix, k.cb_ps, ix), GC2(k.g_c2, k.npix, k.g_c2_ps, ix), GSC(k.g_sc, k.npix, k.g_sc_ps, ix);
  MaskTrip<T, UNR> MK(k.relu_mask, k.npix, k.CU, ix);
  const bool write_sc = k.g_sc != nullptr;

  for (long p = ix.p; p < k.npix; p += UNR * ix.pstep) {
    bool ok[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) ok[u] = p + u * ix.pstep < k.npix;
    ubr_u4 rg[UNR], rg2[HAS_GO2 ? UNR : 1], ro[MASK ? 1 : UNR], rc2[UNR], rcb[BYP ? UNR : 1];
    unsigned rm[UNR];
    GO.ld(rg, ok);
    if constexpr (HAS_GO2) GO2.ld(rg2, ok);
    if constexpr (MASK) MK.ld(rm, ok); else OUT.ld(ro, ok);
    C2.ld(rc2, ok);
    if constexpr (BYP) CB.ld(rcb, ok);
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      float g[CPU], o[CPU], c2[CPU], cb[CPU];
      unpack4<T>(rg[u], g);
      if constexpr (HAS_GO2) {
        float g2[CPU];
        unpack4<T>(rg2[u], g2);
#pragma unroll
        for (int e = 0; e < CPU; ++e) g[e] += g2[e];
      }
      unsigned mbits = 0u;
      if constexpr (MASK) mbits = rm[u]; else unpack4<T>(ro[u], o);
      unpack4<T>(rc2[u], c2);
      if constexpr (BYP) unpack4<T>(rcb[u], cb);
      float r2[CPU], rs[CPU];
#pragma unroll
      for (int h = 0; h < H2; ++h) {
        // (the same operations, in the same order, as the scalar form: out > 0 gates the incoming gradient, relu(bn2) gates the
        // branch through conv2; xh = (c - mean) * invstd; apply: s * (gy - k1 - xh * k2))
        const ubr_f2 d2 = ubr_f2{c2[2 * h], c2[2 * h + 1]} - m2[h];
        const ubr_f2 bn = __builtin_elementwise_fma(d2, s2[h], t2[h]);
        const ubr_f2 xh2 = d2 * i2[h];
        const bool p0 = MASK ? ((mbits >> (2 * h)) & 1u) != 0u : o[2 * h] > 0.f;
        const bool p1 = MASK ? ((mbits >> (2 * h + 1)) & 1u) != 0u : o[2 * h + 1] > 0.f;
        const ubr_f2 gz = {p0 ? g[2 * h] : 0.f, p1 ? g[2 * h + 1] : 0.f};
        const ubr_f2 gy2 = {bn[0] > 0.f ? gz[0] : 0.f, bn[1] > 0.f ? gz[1] : 0.f};
        ubr_f2 xhb = {0.f, 0.f};
        if (BYP) xhb = (ubr_f2{cb[2 * h], cb[2 * h + 1]} - mb[h]) * ib[h];
        if (APPLY) {
          const ubr_f2 a = s2[h] * (gy2 - k12[h] - xh2 * k22[h]);
          r2[2 * h] = a[0]; r2[2 * h + 1] = a[1];
          ubr_f2 b = gz;
          if (BYP) b = sb[h] * (gz - k1b[h] - xhb * k2b[h]);
          rs[2 * h] = b[0]; rs[2 * h + 1] = b[1];
        } else {
          const ubr_f2 gx = gy2 * xh2;
          acc[0][2 * h] += gy2[0]; acc[0][2 * h + 1] += gy2[1];
          acc[1][2 * h] += gx[0]; acc[1][2 * h + 1] += gx[1];
          if (BYP) {
            const ubr_f2 gb = gz * xhb;
            acc[2][2 * h] += gz[0]; acc[2][2 * h + 1] += gz[1];
            acc[3][2 * h] += gb[0]; acc[3][2 * h + 1] += gb[1];
          }
        }
      }
      if (APPLY) { GC2.st(u, r2, ok[u]); if (write_sc) GSC.st(u, rs, ok[u]); }
    }
    GO.next(); GO2.next(); OUT.next(); C2.next(); CB.next(); MK.next();
    if (APPLY) { GC2.next(); GSC.next(); }
  }
  if (!APPLY) {
    const size_t so = (size_t)(blockIdx.x % k.nslots) * 2 * k.C;
    double* outs[4] = {k.red2 + so, k.red2 + so + k.C, BYP ? k.redb + so : nullptr, BYP ? k.redb + so + k.C : nullptr};
    if (k.flush) flush_sums_pow2<CPU, 4>(acc, ix.c, k.CU, k.C, reinterpret_cast<float*>(smem), outs);
    else flush_sums<CPU, 4>(acc, ix.c, k.C, reinterpret_cast<double*>(smem), outs);
  }
}

// ------------------------------------------------------------------------------------------
// BatchNorm(+ReLU) backward: a = max(bn(c), 0)
// ------------------------------------------------------------------------------------------
struct BnB {
  long npix; int C, CU;
  const void *ga, *ga2, *c; long ga_ps, ga2_ps, c_ps;
  const float *scale, *shift, *mean, *invstd, *k1, *k2;
  double* red; void* gc; long gc_ps;
  int flush, nslots;
  const double* fin_red; double count; float *dgamma, *dbeta;     // apply pass with the finalize fused (see TailB)
};
template <typename T, bool APPLY, bool HAS_GA2, bool RELU>
__global__ __launch_bounds__(256) void bn_bwd_kernel(const BnB k) {
  ubr_main_prio();
  constexpr int CPU = ET<T>::CPU;
  constexpr int H2 = CPU / 2;
  constexpr int UNR = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  UnitIdx<T> ix(k.CU);
  ubr_f2 sc[H2], sh[H2], mu[H2], is[H2], k1[H2], k2[H2];
  ldconst2<CPU>(k.scale, ix.c, sc); ldconst2<CPU>(k.shift, ix.c, sh); ldconst2<CPU>(k.mean, ix.c, mu); ldconst2<CPU>(k.invstd, ix.c, is);
  if (APPLY) {
    if (k.fin_red != nullptr) {
      float* kk = reinterpret_cast<float*>(smem);
      fin_site(k.fin_red, k.nslots, k.C, k.count, kk, k.dgamma, k.dbeta);
      __syncthreads();
      ldconst2<CPU>(kk, ix.c, k1); ldconst2<CPU>(kk + k.C, ix.c, k2);
    } else {
      ldconst2<CPU>(k.k1, ix.c, k1); ldconst2<CPU>(k.k2, ix.c, k2);
    }
  }
  float acc[2][CPU];
#pragma unroll
  for (int e = 0; e < CPU; ++e) { acc[0][e] = 0.f; acc[1][e] = 0.f; }
  UnitTrip<T, UNR> GA(k.ga, k.npix, k.ga_ps, ix), GA2(k.ga2, k.npix, k.ga2_ps, ix), CC(k.c, k.npix, k.c_ps, ix), GC(k.gc, k.npix, k.gc_ps, ix);
  for (long p = ix.p; p < k.npix; p += UNR * ix.pstep) {
    bool ok[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) ok[u] = p + u * ix.pstep < k.npix;
    ubr_u4 rg[UNR], rg2[HAS_GA2 ? UNR : 1], rc[UNR];
    GA.ld(rg, ok);
    if constexpr (HAS_GA2) GA2.ld(rg2, ok);
    CC.ld(rc, ok);
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      float g[CPU], c[CPU], r[CPU];
      unpack4<T>(rg[u], g);
      if constexpr (HAS_GA2) {
        float g2[CPU];
        unpack4<T>(rg2[u], g2);
#pragma unroll
        for (int e = 0; e < CPU; ++e) g[e] += g2[e];
      }
      unpack4<T>(rc[u], c);
#pragma unroll
      for (int h = 0; h < H2; ++h) {
        const ubr_f2 d = ubr_f2{c[2 * h], c[2 * h + 1]} - mu[h];
        const ubr_f2 bn = __builtin_elementwise_fma(d, sc[h], sh[h]);
        const ubr_f2 xh = d * is[h];
        const ubr_f2 gy = {(!RELU || bn[0] > 0.f) ? g[2 * h] : 0.f, (!RELU || bn[1] > 0.f) ? g[2 * h + 1] : 0.f};
        if (APPLY) {
          const ubr_f2 a = sc[h] * (gy - k1[h] - xh * k2[h]);
          r[2 * h] = a[0]; r[2 * h + 1] = a[1];
        } else {
          const ubr_f2 gx = gy * xh;
          acc[0][2 * h] += gy[0]; acc[0][2 * h + 1] += gy[1];
          acc[1][2 * h] += gx[0]; acc[1][2 * h + 1] += gx[1];
        }
      }
      if (APPLY) GC.st(u, r, ok[u]);
    }
    GA.next(); GA2.next(); CC.next();
    if (APPLY) GC.next();
  }
  if (!APPLY) {
    const size_t so = (size_t)(blockIdx.x % k.nslots) * 2 * k.C;
    double* outs[2] = {k.red + so, k.red + so + k.C};
    if (k.flush) flush_sums_pow2<CPU, 2>(acc, ix.c, k.CU, k.C, reinterpret_cast<float*>(smem), outs);
    else flush_sums<CPU, 2>(acc, ix.c, k.C, reinterpret_cast<double*>(smem), outs);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void channel_sum_kernel(long npix, int C, int CU, const void* g, long g_ps, double* red) {
  constexpr int CPU = ET<T>::CPU;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  UnitIdx<T> ix(CU);
  float acc[1][CPU];
#pragma unroll
  for (int e = 0; e < CPU; ++e) acc[0][e] = 0.f;
#pragma unroll 2
  for (long p = ix.p; p < npix; p += ix.pstep) {
    float v[CPU];
    ldunit<T>(g, p, g_ps, ix.c, v);
#pragma unroll
    for (int e = 0; e < CPU; ++e) acc[0][e] += v[e];
  }
  double* outs[1] = {red + (size_t)(blockIdx.x % UBR_STAT_SLOTS) * C};
  flush_sums<CPU, 1>(acc, ix.c, C, reinterpret_cast<double*>(smem), outs);
}

// ------------------------------------------------------------------------------------------
// BatchNorm finalize kernels (tiny)
// ------------------------------------------------------------------------------------------
__global__ void bn_finalize_kernel(const double* stats, double count, const float* gamma, const float* beta,
                                   float* rmean, float* rvar, long long* nbt, float momentum, float eps, int C,
                                   float* scale, float* shift, float* mean, float* invstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  // momentum < 0 = nn.BatchNorm2d(momentum=None): cumulative moving average, factor 1 / (batches tracked, this one included).
  // That form is launched as ONE workgroup, so every thread reads the counter before thread 0 advances it.
  const long long n_old = nbt != nullptr ? *nbt : 0;
  __syncthreads();
  if (c == 0 && nbt != nullptr) *nbt = n_old + 1;
  if (momentum < 0.f) momentum = (float)(1.0 / (double)(n_old + 1));
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  // (all slot loads in flight at once: the compiler's partial unroll made this eight dependent round trips to L2, ~4 of the
  // ~6 us these kernels take on the critical chain between a reduction pass and its consumer)
#pragma unroll
  for (int sl = 0; sl < UBR_STAT_SLOTS; ++sl) { s1 += stats[(size_t)sl * 2 * C + c]; s2 += stats[(size_t)sl * 2 * C + C + c]; }
  const double m = s1 / count;
  double var = s2 / count - m * m;
  if (var < 0.0) var = 0.0;
  const double is = 1.0 / sqrt(var + (double)eps);
  const float sc = (float)((double)gamma[c] * is);
  scale[c] = sc;
  shift[c] = beta[c];   // bn(x) = (x - mean)*scale + shift
  mean[c] = (float)m;
  invstd[c] = (float)is;
  if (rmean != nullptr) {
    const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
    rmean[c] = (float)((1.0 - (double)momentum) * (double)rmean[c] + (double)momentum * m);
    rvar[c] = (float)((1.0 - (double)momentum) * (double)rvar[c] + (double)momentum * unb);
  }
}
__global__ void bn_eval_affine_kernel(const float* gamma, const float* beta, const float* rmean, const float* rvar,
                                      float eps, int C, float* scale, float* shift, float* mean, float* invstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float is = 1.0f / sqrtf(rvar[c] + eps);
  scale[c] = gamma[c] * is;
  shift[c] = beta[c];
  mean[c] = rmean[c];
  invstd[c] = is;
}
__global__ void bn_bwd_finalize_kernel(const double* red, double count, int C, float* dgamma, float* dbeta, int accumulate,
                                       float* k1, float* k2) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double sg = 0.0, sgx = 0.0;
  // (all slot loads in flight at once: the compiler's partial unroll made this eight dependent round trips to L2, ~4 of the
  // ~6 us these kernels take on the critical chain between a reduction pass and its consumer)
#pragma unroll
  for (int sl = 0; sl < UBR_STAT_SLOTS; ++sl) { sg += red[(size_t)sl * 2 * C + c]; sgx += red[(size_t)sl * 2 * C + C + c]; }
  if (dgamma != nullptr) dgamma[c] = accumulate ? dgamma[c] + (float)sgx : (float)sgx;
  if (dbeta != nullptr) dbeta[c] = accumulate ? dbeta[c] + (float)sg : (float)sg;
  k1[c] = (float)(sg / count);
  k2[c] = (float)(sgx / count);
}
__global__ void cast_f64_kernel(const double* src, float* dst, int n, int stride, int slots, double scale, int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double a = 0.0;
  if (slots == UBR_STAT_SLOTS) {      // the striped accumulators: every slot load in flight at once (see bn_finalize_kernel)
#pragma unroll
    for (int sl = 0; sl < UBR_STAT_SLOTS; ++sl) a += src[(size_t)sl * stride + i];
  } else {
    for (int sl = 0; sl < slots; ++sl) a += src[(size_t)sl * stride + i];
  }
  const float v = (float)(a * scale);
  dst[i] = accumulate ? dst[i] + v : v;
}

// ------------------------------------------------------------------------------------------
// MaxPool2d(3, stride, padding=1)
// ------------------------------------------------------------------------------------------
struct PoolK {
  int N, H, W, OH, OW, C, CU, stride;
  const void* x; long x_ps;
  const float *sub, *scale, *shift, *lo;
  void* pooled; long p_ps;
  void* xcopy; long xc_ps;
  const void* gp; long gp_ps;
  const void* ge; long ge_ps;
  void* gx; long gx_ps;
  uint8_t* amax;          // [N][OH][OW][C] tap index (ky*3+kx) of the window maximum: written by forward, read by backward
};

// forward: one thread per pooled pixel and channel unit.  All nine window loads are issued before the first compare (buffer
// loads; a tap outside the image gets an out-of-range offset and is excluded from the maximum by its validity bit), where the
// first form walked the taps with a branch and a 64-bit address computation per tap and ran at 2 TB/s.
template <typename T, int S, bool XF, bool AMAX, bool XCOPY>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const PoolK k) {
  constexpr int CPU = ET<T>::CPU;
  constexpr unsigned ESZ = 16 / CPU;
  UnitIdx<T> ix(k.CU);
  float sc[XF ? CPU : 1], sh[XF ? CPU : 1], lo[XF ? CPU : 1], sb[XF ? CPU : 1];
  if constexpr (XF) { ldconst<CPU>(k.scale, ix.c, sc); ldconst<CPU>(k.shift, ix.c, sh); ldconst<CPU>(k.lo, ix.c, lo); ldconst<CPU>(k.sub, ix.c, sb); }
  const long npix = (long)k.N * k.OH * k.OW;
  const long npix_in = (long)k.N * k.H * k.W;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(k.x), 0, (int)(npix_in * k.x_ps * ESZ), 0x00020000);
  const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(k.pooled, 0, (int)(npix * k.p_ps * ESZ), 0x00020000);
  const __amdgpu_buffer_rsrc_t cr = __builtin_amdgcn_make_buffer_rsrc(k.xcopy, 0, XCOPY ? (int)(npix_in * k.xc_ps * ESZ) : 0, 0x00020000);
  const unsigned xps = (unsigned)k.x_ps * ESZ, cps = (unsigned)k.xc_ps * ESZ;
  for (long p = ix.p; p < npix; p += ix.pstep) {
    // (pixel counts are < 2^31: the host checks tensor sizes; 32-bit division is a third of the 64-bit sequence)
    const unsigned pu = (unsigned)p, ru = pu / (unsigned)k.OW;
    const int ox = (int)(pu - ru * (unsigned)k.OW);
    const int n = (int)(ru / (unsigned)k.OH);
    const int oy = (int)(ru - (unsigned)n * (unsigned)k.OH);
    ubr_u4 raw[9];
    bool ok[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int iy = oy * S - 1 + ky, jx = ox * S - 1 + kx;
        const bool in = (unsigned)iy < (unsigned)k.H && (unsigned)jx < (unsigned)k.W;
        const unsigned ip = ((unsigned)n * (unsigned)k.H + (unsigned)iy) * (unsigned)k.W + (unsigned)jx;
        ok[ky * 3 + kx] = in;
        raw[ky * 3 + kx] = __builtin_amdgcn_raw_buffer_load_b128(xr, in ? (int)(ip * xps + (unsigned)ix.c * 16u) : kUnitOOR, 0, 0);
      }
    float m[CPU];
    int am[CPU];
#pragma unroll
    for (int e = 0; e < CPU; ++e) { m[e] = -FLT_MAX; am[e] = -1; }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float v[CPU];
      unpack4<T>(raw[t], v);
      if constexpr (XF) ubr_bnrelu<CPU>(v, sb, sc, sh, lo);
      if constexpr (AMAX) {
#pragma unroll
        for (int e = 0; e < CPU; ++e)
          if (ok[t] && (am[e] < 0 || v[e] > m[e])) { m[e] = v[e]; am[e] = t; }   // strict >: first maximum wins (ATen)
      } else {
#pragma unroll
        for (int e = 0; e < CPU; ++e) m[e] = ok[t] ? fmaxf(m[e], v[e]) : m[e];
      }
      // each stride-2 window owns the 2x2 input pixels (2oy..2oy+1, 2ox..2ox+1) = taps ky,kx in {1,2}: always inside the image
      if constexpr (XCOPY) {
        if (t / 3 >= 1 && t % 3 >= 1) {
          const unsigned ip = ((unsigned)n * (unsigned)k.H + (unsigned)(oy * S - 1 + t / 3)) * (unsigned)k.W + (unsigned)(ox * S - 1 + t % 3);
          const uint4 pk = ET<T>::pack(v);
          __builtin_amdgcn_raw_buffer_store_b128(ubr_u4{pk.x, pk.y, pk.z, pk.w}, cr, (int)(ip * cps + (unsigned)ix.c * 16u), 0, 0);
        }
      }
    }
    {
      const uint4 pk = ET<T>::pack(m);
      __builtin_amdgcn_raw_buffer_store_b128(ubr_u4{pk.x, pk.y, pk.z, pk.w}, pr, (int)(pu * (unsigned)k.p_ps * ESZ + (unsigned)ix.c * 16u), 0, 0);
    }
    if constexpr (AMAX) {
      uint8_t* a = k.amax + p * k.C + (long)ix.c * CPU;
      if constexpr (CPU == 8) {
        *reinterpret_cast<uint2*>(a) = make_uint2((unsigned)am[0] | ((unsigned)am[1] << 8) | ((unsigned)am[2] << 16) | ((unsigned)am[3] << 24),
                                                  (unsigned)am[4] | ((unsigned)am[5] << 8) | ((unsigned)am[6] << 16) | ((unsigned)am[7] << 24));
      } else {
        *reinterpret_cast<unsigned*>(a) = (unsigned)am[0] | ((unsigned)am[1] << 8) | ((unsigned)am[2] << 16) | ((unsigned)am[3] << 24);
      }
    }
  }
}

// stride-2 backward from the saved arg-max: one thread per pooled pixel owns the 2x2 input pixels and looks at the
// (up to) four windows that overlap them -- four gradient units and four index words instead of 36 transformed loads.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_s2_amax_kernel(const PoolK k) {
  constexpr int CPU = ET<T>::CPU;
  UnitIdx<T> ix(k.CU);
  const long npix = (long)k.N * k.OH * k.OW;
#pragma unroll 2
  for (long p = ix.p; p < npix; p += ix.pstep) {
    // (pixel counts are < 2^31: the host checks tensor sizes; 32-bit division is a third of the 64-bit sequence)
    const unsigned pu = (unsigned)p, ru = pu / (unsigned)k.OW;
    const int ox = (int)(pu - ru * (unsigned)k.OW);
    const int n = (int)(ru / (unsigned)k.OH);
    const int oy = (int)(ru - (unsigned)n * (unsigned)k.OH);
    const long r = (long)ru;
    (void)r;
    float g[2][2][CPU];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        if (k.ge != nullptr) ldunit<T>(k.ge, ((long)n * k.H + 2 * oy + a) * k.W + 2 * ox + b, k.ge_ps, ix.c, g[a][b]);
        else {
#pragma unroll
          for (int e = 0; e < CPU; ++e) g[a][b][e] = 0.f;
        }
      }
#pragma unroll
    for (int wy = 0; wy < 2; ++wy) {
      if (oy + wy >= k.OH) continue;
#pragma unroll
      for (int wx = 0; wx < 2; ++wx) {
        if (ox + wx >= k.OW) continue;
        const long wp = ((long)n * k.OH + oy + wy) * k.OW + ox + wx;
        float gp[CPU];
        ldunit<T>(k.gp, wp, k.gp_ps, ix.c, gp);
        unsigned am[CPU];
        const uint8_t* ap = k.amax + wp * k.C + (long)ix.c * CPU;
        if constexpr (CPU == 8) {
          const uint2 w = *reinterpret_cast<const uint2*>(ap);
#pragma unroll
          for (int e = 0; e < 4; ++e) { am[e] = (w.x >> (8 * e)) & 0xffu; am[4 + e] = (w.y >> (8 * e)) & 0xffu; }
        } else {
          const unsigned w = *reinterpret_cast<const unsigned*>(ap);
#pragma unroll
          for (int e = 0; e < 4; ++e) am[e] = (w >> (8 * e)) & 0xffu;
        }
        // tap (ky,kx) of window (wy,wx) is owned pixel (a,b) = (2wy-1+ky, 2wx-1+kx)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const int ky = a + 1 - 2 * wy, kx = b + 1 - 2 * wx;
            if (ky < 0 || ky > 2 || kx < 0 || kx > 2) continue;
#pragma unroll
            for (int e = 0; e < CPU; ++e)
              if (am[e] == (unsigned)(ky * 3 + kx)) g[a][b][e] += gp[e];
          }
      }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) stunit<T>(k.gx, ((long)n * k.H + 2 * oy + a) * k.W + 2 * ox + b, k.gx_ps, ix.c, g[a][b]);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const PoolK k) {
  constexpr int CPU = ET<T>::CPU;
  UnitIdx<T> ix(k.CU);
  const bool xf = k.scale != nullptr;
  float sc[CPU], sh[CPU], lo[CPU], sb[CPU];
  if (xf) { ldconst<CPU>(k.scale, ix.c, sc); ldconst<CPU>(k.shift, ix.c, sh); ldconst<CPU>(k.lo, ix.c, lo); ldconst<CPU>(k.sub, ix.c, sb); }
  const long npix = (long)k.N * k.H * k.W;
  const int s = k.stride;
#pragma unroll 2
  for (long p = ix.p; p < npix; p += ix.pstep) {
    const int jx = (int)(p % k.W);
    const long r = p / k.W;
    const int iy = (int)(r % k.H);
    const int n = (int)(r / k.H);
    float g[CPU];
    if (k.ge != nullptr) ldunit<T>(k.ge, p, k.ge_ps, ix.c, g);
    else {
#pragma unroll
      for (int e = 0; e < CPU; ++e) g[e] = 0.f;
    }
    // windows (oy,ox) whose 3x3 footprint contains (iy,jx): oy*s-1 <= iy <= oy*s+1
    const int oy_lo = (iy - 1 + s - 1 >= 0) ? (iy - 1 + s - 1) / s : 0;   // ceil((iy-1)/s), clamped at 0
    const int oy_hi = min((iy + 1) / s, k.OH - 1);
    const int ox_lo = (jx - 1 + s - 1 >= 0) ? (jx - 1 + s - 1) / s : 0;
    const int ox_hi = min((jx + 1) / s, k.OW - 1);
    for (int oy = oy_lo; oy <= oy_hi; ++oy)
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        float best[CPU]; bool me[CPU];
#pragma unroll
        for (int e = 0; e < CPU; ++e) { best[e] = -FLT_MAX; me[e] = false; }
        bool first = true;
        for (int ky = 0; ky < 3; ++ky) {
          const int yy = oy * s - 1 + ky;
          if ((unsigned)yy >= (unsigned)k.H) continue;
          for (int kx = 0; kx < 3; ++kx) {
            const int xx = ox * s - 1 + kx;
            if ((unsigned)xx >= (unsigned)k.W) continue;
            float v[CPU];
            ldunit<T>(k.x, ((long)n * k.H + yy) * k.W + xx, k.x_ps, ix.c, v);
            if (xf) {
              ubr_bnrelu<CPU>(v, sb, sc, sh, lo);
            }
            const bool here = (yy == iy) && (xx == jx);
#pragma unroll
            for (int e = 0; e < CPU; ++e)
              if (first || v[e] > best[e]) { best[e] = v[e]; me[e] = here; }   // strict >: first maximum wins (ATen)
            first = false;
          }
        }
        float gp[CPU];
        ldunit<T>(k.gp, ((long)n * k.OH + oy) * k.OW + ox, k.gp_ps, ix.c, gp);
#pragma unroll
        for (int e = 0; e < CPU; ++e) if (me[e]) g[e] += gp[e];
      }
    stunit<T>(k.gx, p, k.gx_ps, ix.c, g);
  }
}

// stride-2 backward, one thread per pooled pixel: it owns the 2x2 input pixels (2oy..2oy+1, 2ox..2ox+1)
// and scans the (up to) four windows that overlap them -- 36 unit loads for 4 outputs instead of 36 per output.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_s2_kernel(const PoolK k) {
  constexpr int CPU = ET<T>::CPU;
  UnitIdx<T> ix(k.CU);
  const bool xf = k.scale != nullptr;
  float sc[CPU], sh[CPU], lo[CPU], sb[CPU];
  if (xf) { ldconst<CPU>(k.scale, ix.c, sc); ldconst<CPU>(k.shift, ix.c, sh); ldconst<CPU>(k.lo, ix.c, lo); ldconst<CPU>(k.sub, ix.c, sb); }
  const long npix = (long)k.N * k.OH * k.OW;
#pragma unroll 2
  for (long p = ix.p; p < npix; p += ix.pstep) {
    // (pixel counts are < 2^31: the host checks tensor sizes; 32-bit division is a third of the 64-bit sequence)
    const unsigned pu = (unsigned)p, ru = pu / (unsigned)k.OW;
    const int ox = (int)(pu - ru * (unsigned)k.OW);
    const int n = (int)(ru / (unsigned)k.OH);
    const int oy = (int)(ru - (unsigned)n * (unsigned)k.OH);
    const long r = (long)ru;
    (void)r;
    float g[2][2][CPU];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        if (k.ge != nullptr) ldunit<T>(k.ge, ((long)n * k.H + 2 * oy + a) * k.W + 2 * ox + b, k.ge_ps, ix.c, g[a][b]);
        else {
#pragma unroll
          for (int e = 0; e < CPU; ++e) g[a][b][e] = 0.f;
        }
      }
#pragma unroll
    for (int wy = 0; wy < 2; ++wy) {
      if (oy + wy >= k.OH) continue;
#pragma unroll
      for (int wx = 0; wx < 2; ++wx) {
        if (ox + wx >= k.OW) continue;
        float best[CPU]; int bidx[CPU];
#pragma unroll
        for (int e = 0; e < CPU; ++e) { best[e] = -FLT_MAX; bidx[e] = -1; }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int yy = 2 * (oy + wy) - 1 + ky;
          if ((unsigned)yy >= (unsigned)k.H) continue;
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int xx = 2 * (ox + wx) - 1 + kx;
            if ((unsigned)xx >= (unsigned)k.W) continue;
            float v[CPU];
            ldunit<T>(k.x, ((long)n * k.H + yy) * k.W + xx, k.x_ps, ix.c, v);
            if (xf) {
              ubr_bnrelu<CPU>(v, sb, sc, sh, lo);
            }
#pragma unroll
            for (int e = 0; e < CPU; ++e)
              if (bidx[e] < 0 || v[e] > best[e]) { best[e] = v[e]; bidx[e] = ky * 3 + kx; }   // strict >: first maximum wins (ATen)
          }
        }
        float gp[CPU];
        ldunit<T>(k.gp, ((long)n * k.OH + oy + wy) * k.OW + ox + wx, k.gp_ps, ix.c, gp);
        // tap (ky,kx) of window (wy,wx) is owned pixel (a,b) = (2wy-1+ky, 2wx-1+kx) when both are in {0,1}
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const int ky = a + 1 - 2 * wy, kx = b + 1 - 2 * wx;
            if (ky < 0 || ky > 2 || kx < 0 || kx > 2) continue;
#pragma unroll
            for (int e = 0; e < CPU; ++e) if (bidx[e] == ky * 3 + kx) g[a][b][e] += gp[e];
          }
      }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) stunit<T>(k.gx, ((long)n * k.H + 2 * oy + a) * k.W + 2 * ox + b, k.gx_ps, ix.c, g[a][b]);
  }
}

template <typename K> struct Dispatch3 {};

}  // namespace

#define UBR_DT_SWITCH(dtype, CALL)                    \
  switch (dtype) {                                    \
    case UBR_F32: { typedef float TT; CALL; } break;  \
    case UBR_BF16: { typedef bf16_t TT; CALL; } break;\
    default: { typedef f16_t TT; CALL; } break;       \
  }

// reduce passes: the register-butterfly flush needs a thread's lane-mates to hold the same unit (power-of-two unit counts below
// a wave) or one unit per lane (>= 64 units); its four per-wave rows must fit the LDS carve-out sized for the atomic form
static int red_flush_mode(int CU, int C, int NQ) {
  const bool pow2 = (CU & (CU - 1)) == 0;
  return ((CU >= 64 || pow2) && (size_t)16 * NQ * C <= 65536) ? 1 : 0;
}

static int check_nhwc(const char* who, int dtype, int64_t npix, int C, const void* p, int64_t ps) {
  UBR_CHECK(ubr_dtype_ok(dtype), "%s: bad dtype", who);
  const int esz = ubr_esize(dtype);
  UBR_CHECK(npix > 0 && C > 0 && C % ubr_cpu(dtype) == 0, "%s: bad extent npix=%ld C=%d", who, (long)npix, C);
  UBR_CHECK(p != nullptr && ubr_aligned16(p) && ps >= C && (ps * esz) % 16 == 0, "%s: tensor must be non-null, 16-byte aligned, pixel stride %ld >= C and 16-byte multiple", who, (long)ps);
  UBR_CHECK(npix * ps * esz < (int64_t)1 << 31, "%s: tensor of %ld pixels x %ld elements exceeds 2 GiB (32-bit buffer offsets)", who, (long)npix, (long)ps);
  return UBR_OK;
}
#define UBR_TRY(x) do { int rc__ = (x); if (rc__ != UBR_OK) return rc__; } while (0)

#define UBR_BOOL2(b0, b1, CALL)                                                          \
  do {                                                                                   \
    if (b0) { constexpr bool B0 = true; if (b1) { constexpr bool B1 = true; CALL; } else { constexpr bool B1 = false; CALL; } } \
    else { constexpr bool B0 = false; if (b1) { constexpr bool B1 = true; CALL; } else { constexpr bool B1 = false; CALL; } }   \
  } while (0)
#define UBR_BOOL3(b0, b1, b2, CALL)                                                      \
  do {                                                                                   \
    if (b2) { constexpr bool B2 = true; UBR_BOOL2(b0, b1, CALL); } else { constexpr bool B2 = false; UBR_BOOL2(b0, b1, CALL); } \
  } while (0)

static int tail_fwd_common(int dtype, int64_t npix, int C, const void* c2, int64_t c2_ps, const float* mean2, const float* scale2,
                           const float* shift2, const void* sc, int64_t sc_ps, const float* mean_b, const float* scale_b,
                           const float* shift_b, void* out, int64_t out_ps, uint8_t* relu_mask, void* stream,
                           const ubr_bn_fwd_fin* fin2 = nullptr, const ubr_bn_fwd_fin* finb = nullptr, double count = 0.0) {
  UBR_TRY(check_nhwc("ubr_block_tail_fwd(c2)", dtype, npix, C, c2, c2_ps));
  UBR_TRY(check_nhwc("ubr_block_tail_fwd(sc)", dtype, npix, C, sc, sc_ps));
  UBR_TRY(check_nhwc("ubr_block_tail_fwd(out)", dtype, npix, C, out, out_ps));
  TailF k{};
  k.npix = npix; k.CU = C / ubr_cpu(dtype); k.C = C;
  k.c2 = c2; k.sc = sc; k.out = out; k.c2_ps = c2_ps; k.sc_ps = sc_ps; k.out_ps = out_ps;
  k.relu_mask = relu_mask;
  bool byp;
  size_t lds = 0;
  if (fin2 != nullptr) {
    auto conv = [](const ubr_bn_fwd_fin* f, BnFwdFin* o) {
      o->stats = f->stats; o->gamma = f->gamma; o->beta = f->beta; o->rmean = f->running_mean; o->rvar = f->running_var;
      o->nbt = (long long*)f->num_batches_tracked; o->momentum = f->momentum; o->eps = f->eps;
      o->scale = f->scale; o->shift = f->shift; o->mean = f->mean; o->invstd = f->invstd;
    };
    auto ok = [](const ubr_bn_fwd_fin* f) {
      return f->stats && f->gamma && f->beta && f->scale && f->shift && f->mean && f->invstd && ((f->running_mean == nullptr) == (f->running_var == nullptr)) &&
             (f->momentum >= 0.f || f->num_batches_tracked != nullptr);
    };
    UBR_CHECK(ok(fin2) && (finb == nullptr || ok(finb)) && count >= 1.0 && (size_t)24 * C <= 65536, "ubr_block_tail_fwd_fin: bad arguments");
    conv(fin2, &k.f2);
    if (finb != nullptr) conv(finb, &k.fb);
    k.count = count; k.nslots = UBR_RED_SLOTS;
    byp = finb != nullptr;
    lds = (size_t)(byp ? 24 : 12) * C;
  } else {
    UBR_CHECK(mean2 && scale2 && shift2 && ((scale_b == nullptr) == (shift_b == nullptr)) && ((scale_b == nullptr) == (mean_b == nullptr)), "ubr_block_tail_fwd: bad affine pointers");
    k.m2 = mean2; k.s2 = scale2; k.t2 = shift2; k.mb = mean_b; k.sb = scale_b; k.tb = shift_b;
    byp = scale_b != nullptr;
  }
  const int blocks = pick_blocks(npix, k.CU, fin2 != nullptr ? 1024 : 2048, 4);
  const bool msk = relu_mask != nullptr;
  UBR_DT_SWITCH(dtype, UBR_BOOL2(byp, msk, ubr_launch((tail_fwd_kernel<TT, B0, B1>), dim3(blocks), dim3(256), lds, (hipStream_t)stream, k)));
  UBR_LAUNCH_CHECK("ubr_block_tail_fwd");
  return UBR_OK;
}
extern "C" int ubr_block_tail_fwd(int dtype, int64_t npix, int C, const void* c2, int64_t c2_ps, const float* mean2, const float* scale2,
                                  const float* shift2, const void* sc, int64_t sc_ps, const float* mean_b, const float* scale_b,
                                  const float* shift_b, void* out, int64_t out_ps, void* stream) {
  return tail_fwd_common(dtype, npix, C, c2, c2_ps, mean2, scale2, shift2, sc, sc_ps, mean_b, scale_b, shift_b, out, out_ps, nullptr, stream);
}
extern "C" int ubr_block_tail_fwd_masked(int dtype, int64_t npix, int C, const void* c2, int64_t c2_ps, const float* mean2, const float* scale2,
                                         const float* shift2, const void* sc, int64_t sc_ps, const float* mean_b, const float* scale_b,
                                         const float* shift_b, void* out, int64_t out_ps, uint8_t* relu_mask, void* stream) {
  UBR_CHECK(relu_mask != nullptr, "ubr_block_tail_fwd_masked: null mask");
  return tail_fwd_common(dtype, npix, C, c2, c2_ps, mean2, scale2, shift2, sc, sc_ps, mean_b, scale_b, shift_b, out, out_ps, relu_mask, stream);
}

struct TailFin { const double *red2, *redb; double count; float *dgamma2, *dbeta2, *dgamma_b, *dbeta_b; };

extern "C" int ubr_block_tail_fwd_fin(int dtype, int64_t npix, int C, const void* c2, int64_t c2_ps, const ubr_bn_fwd_fin* bn2,
                                      const void* sc, int64_t sc_ps, const ubr_bn_fwd_fin* bn_b, double count,
                                      void* out, int64_t out_ps, uint8_t* relu_mask, void* stream) {
  UBR_CHECK(bn2 != nullptr, "ubr_block_tail_fwd_fin: null bn2");
  return tail_fwd_common(dtype, npix, C, c2, c2_ps, nullptr, nullptr, nullptr, sc, sc_ps, nullptr, nullptr, nullptr, out, out_ps, relu_mask, stream, bn2, bn_b, count);
}

static int tail_bwd_common(bool apply, int dtype, int64_t npix, int C, const void* go, int64_t go_ps, const void* go2, int64_t go2_ps,
                           const void* out, int64_t out_ps, const void* c2, int64_t c2_ps,
                           const float* scale2, const float* shift2, const float* mean2, const float* invstd2,
                           const float* k1_2, const float* k2_2,
                           const void* cb, int64_t cb_ps, const float* scale_b, const float* mean_b, const float* invstd_b,
                           const float* k1_b, const float* k2_b, double* red2, double* red_b,
                           void* g_c2, int64_t g_c2_ps, void* g_sc, int64_t g_sc_ps, void* stream, const uint8_t* relu_mask = nullptr,
                           const TailFin* fin = nullptr) {
  const char* who = apply ? "ubr_block_tail_bwd_apply" : "ubr_block_tail_bwd_reduce";
  UBR_TRY(check_nhwc(who, dtype, npix, C, go, go_ps));
  if (go2) UBR_TRY(check_nhwc(who, dtype, npix, C, go2, go2_ps));
  if (relu_mask == nullptr) UBR_TRY(check_nhwc(who, dtype, npix, C, out, out_ps));
  UBR_TRY(check_nhwc(who, dtype, npix, C, c2, c2_ps));
  if (cb) UBR_TRY(check_nhwc(who, dtype, npix, C, cb, cb_ps));
  UBR_CHECK(scale2 && shift2 && mean2 && invstd2, "%s: null bn2 constants", who);
  if (cb) UBR_CHECK(mean_b && invstd_b, "%s: null bnpass constants", who);
  if (relu_mask) UBR_CHECK(npix * (C / ubr_cpu(dtype)) < (int64_t)1 << 31, "%s: mask exceeds 2 GiB", who);
  if (apply) {
    UBR_TRY(check_nhwc(who, dtype, npix, C, g_c2, g_c2_ps));
    if (g_sc != nullptr || cb != nullptr) UBR_TRY(check_nhwc(who, dtype, npix, C, g_sc, g_sc_ps));
    if (fin == nullptr) UBR_CHECK(k1_2 && k2_2 && (!cb || (k1_b && k2_b)), "%s: null backward constants", who);
    else UBR_CHECK(fin->red2 && (!cb || fin->redb) && fin->count >= 1.0 && (size_t)16 * C <= 65536, "%s: bad fused-finalize arguments", who);
    UBR_CHECK(!cb || scale_b, "%s: null bnpass scale", who);
  } else {
    UBR_CHECK(red2 && (!cb || red_b), "%s: null reduction buffer", who);
  }
  TailB k{};
  k.npix = npix; k.C = C; k.CU = C / ubr_cpu(dtype);
  k.go = go; k.go2 = go2; k.out = out; k.c2 = c2; k.cb = cb;
  k.go_ps = go_ps; k.go2_ps = go2_ps; k.out_ps = out_ps; k.c2_ps = c2_ps; k.cb_ps = cb_ps;
  k.s2 = scale2; k.t2 = shift2; k.m2 = mean2; k.i2 = invstd2; k.k1_2 = k1_2; k.k2_2 = k2_2;
  k.sb = scale_b; k.mb = mean_b; k.ib = invstd_b; k.k1_b = k1_b; k.k2_b = k2_b;
  k.red2 = red2; k.redb = red_b; k.g_c2 = g_c2; k.g_sc = g_sc; k.g_c2_ps = g_c2_ps; k.g_sc_ps = g_sc_ps;
  k.relu_mask = relu_mask;
  k.nslots = UBR_RED_SLOTS;
  if (fin != nullptr) {
    k.fin_red2 = fin->red2; k.fin_redb = fin->redb; k.count = fin->count;
    k.dgamma2 = fin->dgamma2; k.dbeta2 = fin->dbeta2; k.dgamma_b = fin->dgamma_b; k.dbeta_b = fin->dbeta_b;
  }
  int red_iters = 8, red_blocks = 512, app_blocks = 2048;
  k.flush = red_flush_mode(k.CU, C, 4);
#ifdef UBR_TUNE
  if (g_tune_red_iters) red_iters = g_tune_red_iters;
  if (g_tune_red_blocks) red_blocks = g_tune_red_blocks;
  if (g_tune_app_blocks) app_blocks = g_tune_app_blocks;
  if (g_tune_slots) k.nslots = g_tune_slots;
  if (!g_tune_flush) k.flush = 0;
#endif
  const int blocks = pick_blocks(npix, k.CU, apply ? app_blocks : red_blocks, apply ? 2 : red_iters);
  // reduce: fp64 atomics need 4 C doubles, the per-wave rows 4 x 4 C floats; apply with the finalize fused: 4 C floats
  const size_t lds = apply ? (fin != nullptr ? (size_t)16 * C : 0) : (size_t)(k.flush ? 16 : 8) * 4 * C;
  const bool byp = cb != nullptr, g2 = go2 != nullptr, msk = relu_mask != nullptr;
  if (apply) { UBR_DT_SWITCH(dtype, UBR_BOOL3(byp, g2, msk, ubr_launch((tail_bwd_kernel<TT, true, B0, B1, B2>), dim3(blocks), dim3(256), lds, (hipStream_t)stream, k))); }
  else { UBR_DT_SWITCH(dtype, UBR_BOOL3(byp, g2, msk, ubr_launch((tail_bwd_kernel<TT, false, B0, B1, B2>), dim3(blocks), dim3(256), lds, (hipStream_t)stream, k))); }
  UBR_LAUNCH_CHECK(who);
  return UBR_OK;
}

extern "C" int ubr_block_tail_bwd_reduce(int dtype, int64_t npix, int C, const void* go, int64_t go_ps, const void* go2, int64_t go2_ps,
                                         const void* out, int64_t out_ps, const void* c2, int64_t c2_ps,
                                         const float* scale2, const float* shift2, const float* mean2, const float* invstd2,
                                         const void* cb, int64_t cb_ps, const float* mean_b, const float* invstd_b,
                                         double* red2, double* red_b, void* stream) {
  return tail_bwd_common(false, dtype, npix, C, go, go_ps, go2, go2_ps, out, out_ps, c2, c2_ps, scale2, shift2, mean2, invstd2,
                         nullptr, nullptr, cb, cb_ps, nullptr, mean_b, invstd_b, nullptr, nullptr, red2, red_b,
                         nullptr, 0, nullptr, 0, stream);
}
extern "C" int ubr_block_tail_bwd_apply(int dtype, int64_t npix, int C, const void* go, int64_t go_ps, const void* go2, int64_t go2_ps,
                                        const void* out, int64_t out_ps, const void* c2, int64_t c2_ps,
                                        const float* scale2, const float* shift2, const float* mean2, const float* invstd2,
                                        const float* k1_2, const float* k2_2,
                                        const void* cb, int64_t cb_ps, const float* scale_b, const float* mean_b, const float* invstd_b,
                                        const float* k1_b, const float* k2_b,
                                        void* g_c2, int64_t g_c2_ps, void* g_sc, int64_t g_sc_ps, void* stream) {
  return tail_bwd_common(true, dtype, npix, C, go, go_ps, go2, go2_ps, out, out_ps, c2, c2_ps, scale2, shift2, mean2, invstd2,
                         k1_2, k2_2, cb, cb_ps, scale_b, mean_b, invstd_b, k1_b, k2_b, nullptr, nullptr,
                         g_c2, g_c2_ps, g_sc, g_sc_ps, stream);
}
// the same two passes reading the forward's ReLU bit mask (ubr_block_tail_fwd_masked) instead of the block output
extern "C" int ubr_block_tail_bwd_reduce_masked(int dtype, int64_t npix, int C, const void* go, int64_t go_ps, const void* go2, int64_t go2_ps,
                                                const uint8_t* relu_mask, const void* c2, int64_t c2_ps,
                                                const float* scale2, const float* shift2, const float* mean2, const float* invstd2,
                                                const void* cb, int64_t cb_ps, const float* mean_b, const float* invstd_b,
                                                double* red2, double* red_b, void* stream) {
  UBR_CHECK(relu_mask != nullptr, "ubr_block_tail_bwd_reduce_masked: null mask");
  return tail_bwd_common(false, dtype, npix, C, go, go_ps, go2, go2_ps, nullptr, 0, c2, c2_ps, scale2, shift2, mean2, invstd2,
                         nullptr, nullptr, cb, cb_ps, nullptr, mean_b, invstd_b, nullptr, nullptr, red2, red_b,
                         nullptr, 0, nullptr, 0, stream, relu_mask);
}
extern "C" int ubr_block_tail_bwd_apply_masked(int dtype, int64_t npix, int C, const void* go, int64_t go_ps, const void* go2, int64_t go2_ps,
                                               const uint8_t* relu_mask, const void* c2, int64_t c2_ps,
                                               const float* scale2, const float* shift2, const float* mean2, const float* invstd2,
                                               const float* k1_2, const float* k2_2,
                                               const void* cb, int64_t cb_ps, const float* scale_b, const float* mean_b, const float* invstd_b,
                                               const float* k1_b, const float* k2_b,
                                               void* g_c2, int64_t g_c2_ps, void* g_sc, int64_t g_sc_ps, void* stream) {
  UBR_CHECK(relu_mask != nullptr, "ubr_block_tail_bwd_apply_masked: null mask");
  return tail_bwd_common(true, dtype, npix, C, go, go_ps, go2, go2_ps, nullptr, 0, c2, c2_ps, scale2, shift2, mean2, invstd2,
                         k1_2, k2_2, cb, cb_ps, scale_b, mean_b, invstd_b, k1_b, k2_b, nullptr, nullptr,
                         g_c2, g_c2_ps, g_sc, g_sc_ps, stream, relu_mask);
}

struct BnFin { const double* red; double count; float *dgamma, *dbeta; };

static int bn_bwd_common(bool apply, int dtype, int64_t npix, int C, const void* ga, int64_t ga_ps, const void* ga2, int64_t ga2_ps,
                         const void* c, int64_t c_ps, const float* scale, const float* shift, const float* mean,
                         const float* invstd, int relu, const float* k1, const float* k2, double* red,
                         void* gc, int64_t gc_ps, void* stream, const BnFin* fin = nullptr) {
  const char* who = apply ? "ubr_bn_bwd_apply" : "ubr_bn_bwd_reduce";
  UBR_TRY(check_nhwc(who, dtype, npix, C, ga, ga_ps));
  if (ga2) UBR_TRY(check_nhwc(who, dtype, npix, C, ga2, ga2_ps));
  UBR_TRY(check_nhwc(who, dtype, npix, C, c, c_ps));
  UBR_CHECK(scale && shift && mean && invstd, "%s: null bn constants", who);
  if (apply) {
    UBR_TRY(check_nhwc(who, dtype, npix, C, gc, gc_ps));
    if (fin == nullptr) UBR_CHECK(k1 && k2, "%s: null k1/k2", who);
    else UBR_CHECK(fin->red && fin->count >= 1.0 && (size_t)8 * C <= 65536, "%s: bad fused-finalize arguments", who);
  } else UBR_CHECK(red != nullptr, "%s: null reduction buffer", who);
  BnB k{};
  k.npix = npix; k.C = C; k.CU = C / ubr_cpu(dtype);
  k.ga = ga; k.ga2 = ga2; k.c = c; k.ga_ps = ga_ps; k.ga2_ps = ga2_ps; k.c_ps = c_ps;
  k.scale = scale; k.shift = shift; k.mean = mean; k.invstd = invstd; k.k1 = k1; k.k2 = k2; k.red = red; k.gc = gc; k.gc_ps = gc_ps;
  k.nslots = UBR_RED_SLOTS;
  if (fin != nullptr) { k.fin_red = fin->red; k.count = fin->count; k.dgamma = fin->dgamma; k.dbeta = fin->dbeta; }
  int red_iters = 8, red_blocks = 512, app_blocks = 2048;
  k.flush = red_flush_mode(k.CU, C, 2);
#ifdef UBR_TUNE
  if (g_tune_red_iters) red_iters = g_tune_red_iters;
  if (g_tune_red_blocks) red_blocks = g_tune_red_blocks;
  if (g_tune_app_blocks) app_blocks = g_tune_app_blocks;
  if (g_tune_slots) k.nslots = g_tune_slots;
  if (!g_tune_flush) k.flush = 0;
#endif
  const int blocks = pick_blocks(npix, k.CU, apply ? app_blocks : red_blocks, apply ? 4 : red_iters);
  const size_t lds = apply ? (fin != nullptr ? (size_t)8 * C : 0) : (size_t)(k.flush ? 16 : 8) * 2 * C;
  const bool g2 = ga2 != nullptr, rl = relu != 0;
  if (apply) { UBR_DT_SWITCH(dtype, UBR_BOOL2(g2, rl, ubr_launch((bn_bwd_kernel<TT, true, B0, B1>), dim3(blocks), dim3(256), lds, (hipStream_t)stream, k))); }
  else { UBR_DT_SWITCH(dtype, UBR_BOOL2(g2, rl, ubr_launch((bn_bwd_kernel<TT, false, B0, B1>), dim3(blocks), dim3(256), lds, (hipStream_t)stream, k))); }
  UBR_LAUNCH_CHECK(who);
  return UBR_OK;
}
extern "C" int ubr_bn_bwd_reduce(int dtype, int64_t npix, int C, const void* ga, int64_t ga_ps, const void* ga2, int64_t ga2_ps,
                                 const void* c, int64_t c_ps, const float* scale, const float* shift, const float* mean,
                                 const float* invstd, int relu, double* red, void* stream) {
  return bn_bwd_common(false, dtype, npix, C, ga, ga_ps, ga2, ga2_ps, c, c_ps, scale, shift, mean, invstd, relu, nullptr, nullptr, red, nullptr, 0, stream);
}
extern "C" int ubr_bn_bwd_apply(int dtype, int64_t npix, int C, const void* ga, int64_t ga_ps, const void* ga2, int64_t ga2_ps,
                                const void* c, int64_t c_ps, const float* scale, const float* shift, const float* mean,
                                const float* invstd, int relu, const float* k1, const float* k2,
                                void* gc, int64_t gc_ps, void* stream) {
  return bn_bwd_common(true, dtype, npix, C, ga, ga_ps, ga2, ga2_ps, c, c_ps, scale, shift, mean, invstd, relu, k1, k2, nullptr, gc, gc_ps, stream);
}

// Apply passes with the finalize fused (no ubr_bn_bwd_finalize launch between reduce and apply): `red` is what the reduce pass
// accumulated; every workgroup forms k1 / k2 from it, workgroup 0 writes dgamma / dbeta.
extern "C" int ubr_bn_bwd_apply_fin(int dtype, int64_t npix, int C, const void* ga, int64_t ga_ps, const void* ga2, int64_t ga2_ps,
                                    const void* c, int64_t c_ps, const float* scale, const float* shift, const float* mean,
                                    const float* invstd, int relu, const double* red, double count, float* dgamma, float* dbeta,
                                    void* gc, int64_t gc_ps, void* stream) {
  const BnFin fin{red, count, dgamma, dbeta};
  return bn_bwd_common(true, dtype, npix, C, ga, ga_ps, ga2, ga2_ps, c, c_ps, scale, shift, mean, invstd, relu, nullptr, nullptr, nullptr, gc, gc_ps, stream, &fin);
}
extern "C" int ubr_block_tail_bwd_apply_fin(int dtype, int64_t npix, int C, const void* go, int64_t go_ps, const void* go2, int64_t go2_ps,
                                            const uint8_t* relu_mask, const void* c2, int64_t c2_ps,
                                            const float* scale2, const float* shift2, const float* mean2, const float* invstd2,
                                            const double* red2, float* dgamma2, float* dbeta2,
                                            const void* cb, int64_t cb_ps, const float* scale_b, const float* mean_b, const float* invstd_b,
                                            const double* red_b, float* dgamma_b, float* dbeta_b, double count,
                                            void* g_c2, int64_t g_c2_ps, void* g_sc, int64_t g_sc_ps, void* stream) {
  UBR_CHECK(relu_mask != nullptr, "ubr_block_tail_bwd_apply_fin: null mask");
  const TailFin fin{red2, red_b, count, dgamma2, dbeta2, dgamma_b, dbeta_b};
  return tail_bwd_common(true, dtype, npix, C, go, go_ps, go2, go2_ps, nullptr, 0, c2, c2_ps, scale2, shift2, mean2, invstd2,
                         nullptr, nullptr, cb, cb_ps, scale_b, mean_b, invstd_b, nullptr, nullptr, nullptr, nullptr,
                         g_c2, g_c2_ps, g_sc, g_sc_ps, stream, relu_mask, &fin);
}

extern "C" int ubr_channel_sum(int dtype, int64_t npix, int C, const void* g, int64_t g_ps, double* red, void* stream) {
  UBR_TRY(check_nhwc("ubr_channel_sum", dtype, npix, C, g, g_ps));
  UBR_CHECK(red != nullptr, "ubr_channel_sum: null output");
  const int CU = C / ubr_cpu(dtype);
  const int blocks = pick_blocks(npix, CU, 1024);
  UBR_DT_SWITCH(dtype, ubr_launch(channel_sum_kernel<TT>, dim3(blocks), dim3(256), (size_t)C * sizeof(double), (hipStream_t)stream,
                                          (long)npix, C, CU, g, (long)g_ps, red));
  UBR_LAUNCH_CHECK("ubr_channel_sum");
  return UBR_OK;
}

extern "C" int ubr_bn_finalize(const double* stats, double count, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, int64_t* num_batches_tracked,
                               float momentum, float eps, int C,
                               float* scale, float* shift, float* mean, float* invstd, void* stream) {
  UBR_CHECK(stats && gamma && beta && scale && shift && mean && invstd && C > 0 && count >= 1.0, "ubr_bn_finalize: bad arguments");
  UBR_CHECK((running_mean == nullptr) == (running_var == nullptr), "ubr_bn_finalize: running stats must come together");
  UBR_CHECK(momentum >= 0.f || (C <= 1024 && num_batches_tracked != nullptr), "ubr_bn_finalize: cumulative averaging (momentum < 0) needs C <= 1024 and the batch counter");
  const dim3 grid(momentum < 0.f ? 1 : ubr_cdiv(C, 128)), block(momentum < 0.f ? ubr_cdiv(C, 64) * 64 : 128);
  ubr_launch(bn_finalize_kernel, grid, block, 0, (hipStream_t)stream, stats, count, gamma, beta,
                     running_mean, running_var, (long long*)num_batches_tracked, momentum, eps, C, scale, shift, mean, invstd);
  UBR_LAUNCH_CHECK("ubr_bn_finalize");
  return UBR_OK;
}
extern "C" int ubr_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, float eps, int C, float* scale, float* shift,
                                  float* mean, float* invstd, void* stream) {
  UBR_CHECK(gamma && beta && running_mean && running_var && scale && shift && mean && invstd && C > 0, "ubr_bn_eval_affine: bad arguments");
  ubr_launch(bn_eval_affine_kernel, dim3(ubr_cdiv(C, 128)), dim3(128), 0, (hipStream_t)stream, gamma, beta, running_mean, running_var, eps, C, scale, shift, mean, invstd);
  UBR_LAUNCH_CHECK("ubr_bn_eval_affine");
  return UBR_OK;
}
extern "C" int ubr_bn_bwd_finalize(const double* red, double count, const float* scale, const float* invstd,
                                   int C, float* dgamma, float* dbeta, int accumulate, float* k1, float* k2, void* stream) {
  (void)scale; (void)invstd;
  UBR_CHECK(red && k1 && k2 && C > 0 && count >= 1.0, "ubr_bn_bwd_finalize: bad arguments");
  ubr_launch(bn_bwd_finalize_kernel, dim3(ubr_cdiv(C, 128)), dim3(128), 0, (hipStream_t)stream, red, count, C, dgamma, dbeta, accumulate, k1, k2);
  UBR_LAUNCH_CHECK("ubr_bn_bwd_finalize");
  return UBR_OK;
}
extern "C" int ubr_cast_f64_to_f32(const double* src, int stride, int slots, float* dst, int n, double scale, int accumulate, void* stream) {
  UBR_CHECK(src && dst && n > 0 && slots >= 1 && stride >= n, "ubr_cast_f64_to_f32: bad arguments");
  ubr_launch(cast_f64_kernel, dim3(ubr_cdiv(n, 128)), dim3(128), 0, (hipStream_t)stream, src, dst, n, stride, slots, scale, accumulate);
  UBR_LAUNCH_CHECK("ubr_cast_f64_to_f32");
  return UBR_OK;
}
extern "C" int ubr_zero(void* p, int64_t bytes, void* stream) {
  UBR_CHECK(p != nullptr && bytes > 0, "ubr_zero: bad arguments");
  hipError_t e = hipMemsetAsync(p, 0, (size_t)bytes, (hipStream_t)stream);
  if (e != hipSuccess) { ubr_set_error("ubr_zero: %s", hipGetErrorString(e)); return UBR_ELAUNCH; }
  ubr_tape* t = ubr_tape_current();
  if (t != nullptr && t->recording && t->paused == 0) {
    const size_t nb = (size_t)bytes;
    t->push((hipStream_t)stream, [p, nb](hipStream_t s) { (void)hipMemsetAsync(p, 0, nb, s); });
  }
  return UBR_OK;
}

static int pool_common(bool bwd, int dtype, int N, int H, int W, int C, int stride, const void* x, int64_t x_ps, ubr_chan_affine xf,
                       void* pooled, int64_t p_ps, void* xcopy, int64_t xc_ps,
                       const void* gp, int64_t gp_ps, const void* ge, int64_t ge_ps, void* gx, int64_t gx_ps, uint8_t* amax, void* stream) {
  const char* who = bwd ? "ubr_maxpool_bwd" : "ubr_maxpool_fwd";
  UBR_CHECK(N > 0 && H > 0 && W > 0 && (stride == 1 || stride == 2), "%s: bad extents", who);
  const int OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
  const int64_t npix_in = (int64_t)N * H * W, npix_out = (int64_t)N * OH * OW;
  UBR_TRY(check_nhwc(who, dtype, npix_in, C, x, x_ps));
  const bool hx = xf.scale != nullptr;
  UBR_CHECK(hx == (xf.shift != nullptr) && hx == (xf.lo != nullptr) && hx == (xf.sub != nullptr), "%s: xf needs sub, scale, shift and lo together", who);
  PoolK k{};
  k.N = N; k.H = H; k.W = W; k.OH = OH; k.OW = OW; k.C = C; k.CU = C / ubr_cpu(dtype); k.stride = stride;
  k.x = x; k.x_ps = x_ps; k.sub = xf.sub; k.scale = xf.scale; k.shift = xf.shift; k.lo = xf.lo;
  k.amax = amax;
  if (amax) UBR_CHECK(C % ubr_cpu(dtype) == 0 && (((uintptr_t)amax) & 7) == 0, "%s: argmax buffer must be 8-byte aligned", who);
  if (!bwd) {
    UBR_TRY(check_nhwc(who, dtype, npix_out, C, pooled, p_ps));
    if (xcopy) {
      UBR_CHECK(stride == 2 && H % 2 == 0 && W % 2 == 0, "%s: xcopy needs stride 2 and even H, W", who);
      UBR_TRY(check_nhwc(who, dtype, npix_in, C, xcopy, xc_ps));
    }
    k.pooled = pooled; k.p_ps = p_ps; k.xcopy = xcopy; k.xc_ps = xc_ps;
    const int blocks = pick_blocks(npix_out, k.CU, 4096);
    const bool am = amax != nullptr, xc = xcopy != nullptr;
    if (stride == 2) { UBR_DT_SWITCH(dtype, UBR_BOOL3(hx, am, xc, ubr_launch((maxpool_fwd_kernel<TT, 2, B0, B1, B2>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, k))); }
    else { UBR_DT_SWITCH(dtype, UBR_BOOL2(hx, am, ubr_launch((maxpool_fwd_kernel<TT, 1, B0, B1, false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, k))); }
  } else {
    UBR_TRY(check_nhwc(who, dtype, npix_out, C, gp, gp_ps));
    if (ge) UBR_TRY(check_nhwc(who, dtype, npix_in, C, ge, ge_ps));
    UBR_TRY(check_nhwc(who, dtype, npix_in, C, gx, gx_ps));
    k.gp = gp; k.gp_ps = gp_ps; k.ge = ge; k.ge_ps = ge_ps; k.gx = gx; k.gx_ps = gx_ps;
    if (amax && stride == 2 && H % 2 == 0 && W % 2 == 0) {
      const int blocks = pick_blocks(npix_out, k.CU);
      UBR_DT_SWITCH(dtype, ubr_launch(maxpool_bwd_s2_amax_kernel<TT>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, k));
    } else if (stride == 2 && H % 2 == 0 && W % 2 == 0) {
      const int blocks = pick_blocks(npix_out, k.CU);
      UBR_DT_SWITCH(dtype, ubr_launch(maxpool_bwd_s2_kernel<TT>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, k));
    } else {
      const int blocks = pick_blocks(npix_in, k.CU);
      UBR_DT_SWITCH(dtype, ubr_launch(maxpool_bwd_kernel<TT>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, k));
    }
  }
  UBR_LAUNCH_CHECK(who);
  return UBR_OK;
}
extern "C" int ubr_maxpool_fwd(int dtype, int N, int H, int W, int C, int stride, const void* x, int64_t x_ps,
                               ubr_chan_affine xf, void* pooled, int64_t p_ps, void* xcopy, int64_t xc_ps, uint8_t* argmax, void* stream) {
  return pool_common(false, dtype, N, H, W, C, stride, x, x_ps, xf, pooled, p_ps, xcopy, xc_ps, nullptr, 0, nullptr, 0, nullptr, 0, argmax, stream);
}
extern "C" int ubr_maxpool_bwd(int dtype, int N, int H, int W, int C, int stride, const void* x, int64_t x_ps,
                               ubr_chan_affine xf, const void* g_pooled, int64_t gp_ps, const void* g_extra, int64_t ge_ps,
                               void* gx, int64_t gx_ps, const uint8_t* argmax, void* stream) {
  return pool_common(true, dtype, N, H, W, C, stride, x, x_ps, xf, nullptr, 0, nullptr, 0, g_pooled, gp_ps, g_extra, ge_ps, gx, gx_ps,
                     const_cast<uint8_t*>(argmax), stream);
}
