// Implicit-GEMM direct convolution on MFMA for NHWC activations (gfx950).
//
// Replaces, for the U-ResNet path, every nn.Conv2d / nn.ConvTranspose2d forward and their data
// gradients (reference: models/common_layers.py:13-15,33,125; models/ub_uresnet.py:60,64;
// models/ASPP_ResNet.py:199-220,275).  No im2col: a workgroup stages the input halo of its
// output tile in LDS ONCE (applying the producer's BatchNorm+ReLU as a per-channel affine on the
// way in and zero-filling the padding), stages the weight slab for its output channels, and every
// tap reads shifted windows of that same LDS image.
//
// GEMM orientation: D[cout][pixel] = W[cout][k] * X[k][pixel], k = (tap, cin) in 16-byte units.
// The accumulator then holds 4 consecutive output channels of one pixel per lane, so the NHWC
// store is a packed 8/16-byte store and 16 lanes cover 16 consecutive pixels.
//
// Workgroup = 256 threads = 4 waves.  Tile = TH x TW output pixels (TW = 16*TWF) x TN = 16*NT
// output channels.  Wave w owns FW pixel fragments (16 px each) x NT channel fragments.
#include <stdlib.h>
#include <type_traits>
#include "ubr_common.h"
#include "ubr_host.h"

namespace {

struct ConvK {
  const char* x; long x_sn, x_sy, x_sx;        // byte strides
  const float *in_sub, *in_scale, *in_shift, *in_lo;
  const char* w;
  char* y; long y_sn, y_sy, y_sx;
  const char* ad; long a_sn, a_sy, a_sx;
  const float* bias;
  double* stats;
  int H, W;
  int CU, UPB, lgUPB, nblk;                     // cin units total, per block, log2, number of blocks
  int Cout, Cout_pad;
  int ntaps, S, iy0, ix0, OH, OW;
  int dymin, dxmin, HH, HW;
  int tiles_x, tiles_y;
  unsigned tx_magic, ty_magic;                   // ceil(2^32 / tiles): the tile decomposition is two scalar multiply-highs
  int nunits, steps;
  int pixb;                                      // LDS bytes per halo pixel
  unsigned rw, rw_magic;                         // items per halo row and ceil(2^32/rw)
  int x_sy32, x_sx32;                            // input row / pixel strides in bytes (per-image offsets fit 31 bits)
  int step_j, step_hy, step_goff, wrap_goff;     // halo walk: advance of (column item, row, byte offset) per 256 items
  int wl_off, halo_off, red_off, xfc_off;        // LDS carve offsets
  int epilogue, act, wide_store, dbg, wlinear, N;
  int buf_stride, ncot;                          // conv_pc_kernel: bytes between its two LDS buffers; cout tiles of the layer
  // training epilogues of the data-gradient convs (ubr_conv_desc.addend_mask / bnb_c)
  const uint8_t* ad_mask; int ad_mask_cu;        // ReLU bit mask gating the addend; units per pixel of the mask
  const char* bc; long bc_sn, bc_sy, bc_sx;      // saved activation c of the BatchNorm-backward sums (view of the output grid)
  const float *bmean, *bscale, *bshift, *binvstd;
  int nslots;                                    // stripes of `stats` in use
  int pair_store;                                // conv_igemm_kernel: 16-byte stores of fragment pairs
  int fast_epi;                                  // conv_igemm_kernel: per-image output / addend extents fit 31-bit offsets (buffer-addressed epilogue)
  // several output phases in one launch (blockIdx.z; ubr_conv_desc.nphase): each has its own tap range, output and addend base
  int ptap0[4], pnunits[4], psteps[4];
  long pyoff[4], paoff[4];                       // bytes
  int8_t dy[UBR_MAX_TAPS], dx[UBR_MAX_TAPS];
  uint8_t wt[UBR_MAX_TAPS];
  unsigned long long* stamps;                    // diagnostic build (-DUBR_CONV_STAMPS): per-workgroup phase cycle sums
};

template <typename T> __device__ __forceinline__ void store4(char* p, const float* v);
template <> __device__ __forceinline__ void store4<float>(char* p, const float* v) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void store4<bf16_t>(char* p, const float* v) {
  *reinterpret_cast<uint2*>(p) = make_uint2(ET<bf16_t>::pk2(v[0], v[1]), ET<bf16_t>::pk2(v[2], v[3]));
}
template <> __device__ __forceinline__ void store4<f16_t>(char* p, const float* v) {
  f16x4_t h; h[0] = (_Float16)v[0]; h[1] = (_Float16)v[1]; h[2] = (_Float16)v[2]; h[3] = (_Float16)v[3];
  *reinterpret_cast<uint2*>(p) = __builtin_bit_cast(uint2, h);
}
template <typename T> __device__ __forceinline__ void load4(const char* p, float* v);
template <> __device__ __forceinline__ void load4<float>(const char* p, float* v) {
  float4 f = *reinterpret_cast<const float4*>(p); v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
}
template <> __device__ __forceinline__ void load4<bf16_t>(const char* p, float* v) {
  uint2 u = *reinterpret_cast<const uint2*>(p);
  v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
  v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
}
template <> __device__ __forceinline__ void load4<f16_t>(const char* p, float* v) {
  f16x4_t h = __builtin_bit_cast(f16x4_t, *reinterpret_cast<const uint2*>(p));
  v[0] = (float)h[0]; v[1] = (float)h[1]; v[2] = (float)h[2]; v[3] = (float)h[3];
}

// v rounded to the storage type (what a later pass would read back from the stored tensor)
template <typename T> __device__ __forceinline__ void round4(const float* v, float* r);
template <> __device__ __forceinline__ void round4<float>(const float* v, float* r) { r[0] = v[0]; r[1] = v[1]; r[2] = v[2]; r[3] = v[3]; }
template <> __device__ __forceinline__ void round4<bf16_t>(const float* v, float* r) {
  // round-to-nearest-even on the bit pattern (integer ops; NaN kept as NaN): the result v_cvt_pk_bf16_f32 gives
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t u = __float_as_uint(v[i]);
    const uint32_t rne = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
    r[i] = __uint_as_float((u & 0x7fffffffu) > 0x7f800000u ? (u | 0x00400000u) & 0xffff0000u : rne);
  }
}
template <> __device__ __forceinline__ void round4<f16_t>(const float* v, float* r) {
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = (float)(_Float16)v[i];
}
// BatchNorm-backward sums of one 4-channel group: g_y = g*[bn(c) > 0], xhat = (c - mean)*invstd (same operations as bn_bwd_kernel)
__device__ __forceinline__ void bnb_accumulate(const float* g, const float* c, const float4& mu, const float4& sc, const float4& sh, const float4& is,
                                               float* s1, float* s2) {
  const float m[4] = {mu.x, mu.y, mu.z, mu.w}, a[4] = {sc.x, sc.y, sc.z, sc.w}, b[4] = {sh.x, sh.y, sh.z, sh.w}, iv[4] = {is.x, is.y, is.z, is.w};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float d = c[r] - m[r];
    const float bn = __builtin_fmaf(d, a[r], b[r]);
    const float xh = d * iv[r];
    const float gy = bn > 0.f ? g[r] : 0.f;
    s1[r] += gy;
    s2[r] += gy * xh;
  }
}

// LDS bytes per halo pixel.  A ds_read_b128 is served in four groups of 16 lanes that each pair 8 lanes of one quad with the
// COMPLEMENTARY 8 lanes of the next quad ({0-3,12-15 | 20-27}, {4-11 | 16-19,28-31}, ... -- MI355X_MICROARCH.md, LDS table),
// and in the MFMA loop neighbouring quads read neighbouring 16-byte channel units of the same 16 pixels.  Against that lane
// grouping (64 banks x 4 B) the natural "pixel stride = channels + 16 bytes of padding" layout of round 1 (48 B for 2-unit
// blocks, 80 B for 4-unit blocks) is a 2-way conflict on EVERY fragment read; 32 B (no padding at all) for 2 units and
// 96 B for 4 units are conflict-free (tools/lds_banks.py enumerates the candidates).  Measured: 2-unit layers at 32 B gain
// (7x7 16->16 @512^2: 204 -> 151 us); 4-unit layers at 96 B LOSE -- the thin 32-channel layers drop from three workgroups per
// CU to two (37 vs 32 us), the wide layers do not change -- so 4-unit blocks keep 80 B.
__host__ __device__ constexpr int conv_pixb(int upb) { return upb == 2 ? 32 : upb * 16 + 16; }

// conv_thin_kernel, 4-unit blocks: -DUBR_THIN_PIXB4=96 selects the conflict-free 96-byte pixel stride (with the halo region
// sized to the halo exactly, three workgroups still fit a CU).  Measured against the 80-byte stride, same box: 32 -> 32 channels
// at 256^2 43 us either way, train step 11.71 vs 11.69 ms -- the 2-way conflict on the pixel-fragment reads is not what bounds
// this kernel.  Default: 80 bytes (less LDS).
#ifndef UBR_THIN_PIXB4
#define UBR_THIN_PIXB4 80
#endif
__host__ __device__ constexpr int thin_pixb(int upb) { return upb == 4 ? UBR_THIN_PIXB4 : conv_pixb(upb); }

// register slots of the cin-block pipeline (PIPE): one block's halo and weight items per thread, sized for 3x3 taps
__host__ __device__ constexpr int conv_pipe_hslots(int fw, int twf) { return ((4 * fw / twf + 2) * (twf * 16 + 2) * 4 + 255) / 256; }
__host__ __device__ constexpr int conv_pipe_wslots(int nt) { return (36 * nt * 16 + 255) / 256; }

// Two pixel fragments' 4-channel groups (fp32) -> this lane's 16 output bytes after the quad exchange (see store_pair16).
// bf16: converts and swaps in ONE asm statement with early-clobber outputs; wait states by hand (hipcc pads nothing inside asm).
// (The wrong lanes 12-15 once seen after this block were the 16-byte store hazard described at buf_store16: a convert writing the
// data registers of the store issued just before it.)
template <typename T> __device__ __forceinline__ uint4 pack_swap_pair(const float* v0, const float* v1);
template <> __device__ __forceinline__ uint4 pack_swap_pair<bf16_t>(const float* v0, const float* v1) {
  uint2 X, Y;
  asm volatile("v_cvt_pk_bf16_f32 %0, %4, %5\n\tv_cvt_pk_bf16_f32 %1, %6, %7\n\tv_cvt_pk_bf16_f32 %2, %8, %9\n\t"
               "v_cvt_pk_bf16_f32 %3, %10, %11\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %2\n\tv_permlane16_swap_b32 %1, %3\n\ts_nop 3"
               : "=&v"(X.x), "=&v"(X.y), "=&v"(Y.x), "=&v"(Y.y)
               : "v"(v0[0]), "v"(v0[1]), "v"(v0[2]), "v"(v0[3]), "v"(v1[0]), "v"(v1[1]), "v"(v1[2]), "v"(v1[3]));
  return make_uint4(X.x, X.y, Y.x, Y.y);
}
template <> __device__ __forceinline__ uint4 pack_swap_pair<f16_t>(const float* v0, const float* v1) {
  f16x4_t a, b;
#pragma unroll
  for (int r = 0; r < 4; ++r) { a[r] = (_Float16)v0[r]; b[r] = (_Float16)v1[r]; }
  uint2 X = __builtin_bit_cast(uint2, a), Y = __builtin_bit_cast(uint2, b);
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\ts_nop 3" : "+v"(X.x), "+v"(Y.x), "+v"(X.y), "+v"(Y.y));
  return make_uint4(X.x, X.y, Y.x, Y.y);
}
template <> __device__ __forceinline__ uint4 pack_swap_pair<float>(const float* v0, const float* v1) { return make_uint4(0u, 0u, 0u, 0u); }

// 16-bit types: lane (q, l16) holds channels 4q..4q+3 of pixel l16 of fragment 0 (X) and of fragment 1 (Y), 8 bytes each.
// v_permlane16_swap exchanges the odd quads of X with the even quads of Y: afterwards an even quad holds channels
// 4q..4q+7 of its fragment-0 pixel and an odd quad channels 4(q-1)..4q+3 of its fragment-1 pixel -- 16 contiguous bytes.
// p0 / p1 address channel 4q of the lane's pixel in fragment 0 / 1.
template <typename T> __device__ __forceinline__ void store_frag_pair(char* p0, char* p1, bool ok0, bool ok1, const float* v0, const float* v1, int q) {
  if constexpr (std::is_same<T, float>::value) {
    if (ok0) store4<float>(p0, v0);
    if (ok1) store4<float>(p1, v1);
  } else {
    const uint4 v = pack_swap_pair<T>(v0, v1);
    const bool odd = (q & 1) != 0;
    char* p = odd ? (p1 - 8) : p0;          // an odd quad's 16 bytes start one quad (4 channels) lower
    if (odd ? ok1 : ok0) *reinterpret_cast<uint4*>(p) = v;
  }
}

typedef __attribute__((ext_vector_type(2))) unsigned ubr_u2;
__device__ __forceinline__ unsigned udiv_magic(unsigned n, unsigned magic) { return magic == 0u ? n : __umulhi(n, magic); }

// 16-byte buffer store with a scalar offset.  gfx950 reads a store's data registers over several cycles; a VALU write to them in the
// next instruction slot corrupts what the store sends (observed: the second dword of lanes 12-15 of each 16-lane row).  hipcc (ROCm 7.2)
// inserts the wait state for stores wider than 8 bytes only when soffset is NOT an SGPR (it assumes the scalar operand fetch hides the
// hazard -- it does not on this part: ISA of the fp32 epilogues, `buffer_store_dwordx4 v[96:99], v135, s[24:27], s90 offen` directly
// followed by `v_pk_add_f32 v[96:97], ...`).  So 16-byte stores fold the scalar offset into the vector one (one v_add) and leave
// soffset zero, which puts them under the compiler's own hazard handling; 8-byte stores are not affected.  (This is what the "stale
// lanes 12-15" failures of the training epilogues in round 3 were; they had first been put down to MFMA result latency and to a
// v_cvt_pk_bf16_f32 next to v_permlane16_swap.)
__device__ __forceinline__ void buf_store16(const ubr_u4& v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b128(v, r, voff + soff, 0, 0);
}

template <typename T, int FW, int NT, int TWF, bool PIPE>
__global__ __launch_bounds__(256, ((NT == 4 && FW <= 2 && !PIPE) ? 4 : 2)) void conv_igemm_kernel(const ConvK k) {
  ubr_main_prio();
  constexpr int TN = NT * 16;
  constexpr int F = 4 * FW;
  constexpr int TH = F / TWF;
  constexpr int TW = TWF * 16;
  constexpr int CPU = ET<T>::CPU;
  constexpr int ESZ = 16 / CPU;
  static_assert(F % TWF == 0, "tile shape");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* tbl = reinterpret_cast<int*>(smem);
  char* wl = smem + k.wl_off;
  char* halo = smem + k.halo_off;
  float* red = reinterpret_cast<float*>(smem + k.red_off);

#ifdef UBR_CONV_STAMPS
  const unsigned long long t_entry = __builtin_amdgcn_s_memtime();
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, l16 = lane & 15;
  // output phase of this workgroup (one phase unless the launch batches the phases of a transposed conv / stride-2 data gradient)
  const int ph = blockIdx.z;
  const int tap0 = k.ptap0[ph], p_nunits = k.pnunits[ph], p_steps = k.psteps[ph];
  char* const yb = k.y + k.pyoff[ph];
  const char* const adb = k.ad != nullptr ? k.ad + k.paoff[ph] : nullptr;
  const unsigned tile_l = blockIdx.x;
  const unsigned tile_r = udiv_magic(tile_l, k.tx_magic);
  const int tx = (int)(tile_l - tile_r * (unsigned)k.tiles_x);
  const int n = (int)udiv_magic(tile_r, k.ty_magic);
  const int ty = (int)(tile_r - (unsigned)n * (unsigned)k.tiles_y);
  const int n0 = blockIdx.y * TN;
  // The tap tables are built from per-tap argument bytes indexed by LANE, i.e. memory loads: requested here, ahead of the halo
  // loads, and consumed after those have been issued (in-kernel stamps, 16x128x128x64 -> 64: the two table loops cost 4-6 k of
  // the workgroup's 11.5 k prologue cycles, two dependent round trips behind the halo issue).
  // (unconditional loads with a clamped index: a load inside a branch is waited for where the branch rejoins)
  const int tb_tap = min(tap0 + (tid >> k.lgUPB), k.ntaps - 1);
  const int tb_dy = k.dy[tb_tap], tb_dx = k.dx[tb_tap], tb_wt = k.wt[tb_tap];
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int hy0 = oy0 * k.S + k.iy0 + k.dymin, hx0 = ox0 * k.S + k.ix0 + k.dxmin;

  // Cin-block pipeline: block 0's halo and the layer's BatchNorm constants are requested before anything else, so that their
  // round trips run under the table set-up, the barrier and the weight-slab requests instead of one after the other.
  constexpr int HS_P = PIPE ? conv_pipe_hslots(FW, TWF) : 1;
  constexpr int kOOR = (int)0x80000000;        // beyond any num_records: the load returns zeros
  ubr_u4 hv[HS_P];
  int hoff[HS_P];
  unsigned hok = 0u;
  float xfr[PIPE ? 8 : 1];
  const bool xf_early = PIPE && k.in_scale != nullptr && k.CU * ET<T>::CPU <= 512;
  if constexpr (PIPE) {
    const int c = tid & (k.UPB - 1);
    const int nit = k.HH * (int)k.rw;
    int hy = (int)__umulhi((unsigned)tid, k.rw_magic);
    int j = tid - hy * (int)k.rw;
#pragma unroll
    for (int u = 0; u < HS_P; ++u) {
      const int iy = hy0 + hy, ix = hx0 + (j >> k.lgUPB);
      const bool ok = (tid + u * 256 < nit) && (unsigned)iy < (unsigned)k.H && (unsigned)ix < (unsigned)k.W;
      hoff[u] = ok ? iy * k.x_sy32 + ix * k.x_sx32 + c * 16 : kOOR;
      hok |= ok ? (1u << u) : 0u;
      j += k.step_j; hy += k.step_hy;
      if (j >= (int)k.rw) { j -= (int)k.rw; hy += 1; }
    }
    const __amdgpu_buffer_rsrc_t xr0 = __builtin_amdgcn_make_buffer_rsrc((void*)(k.x + (long)n * k.x_sn), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int u = 0; u < HS_P; ++u) hv[u] = __builtin_amdgcn_raw_buffer_load_b128(xr0, hoff[u], 0, 0);
    if (xf_early) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ch = tid + i * 256;
        const bool in = ch < k.CU * ET<T>::CPU;
        xfr[4 * i + 0] = in ? k.in_sub[ch] : 0.f; xfr[4 * i + 1] = in ? k.in_scale[ch] : 0.f;
        xfr[4 * i + 2] = in ? k.in_shift[ch] : 0.f; xfr[4 * i + 3] = in ? k.in_lo[ch] : 0.f;
      }
    }
  }
#ifdef UBR_CONV_STAMPS
  const unsigned long long t_p1 = __builtin_amdgcn_s_memtime();
#endif
  // tbl: LDS offset of a K unit's tap window; wsrc: its weight-slab source index (16-byte items, cin block 0, channel 0; -1 = padding)
  int* wsrc = tbl + 4 * k.steps;
  if (tid < 4 * k.steps) {
    const int c = tid & (k.UPB - 1);
    const bool in = tid < p_nunits;
    tbl[tid] = in ? ((tb_dy - k.dymin) * k.HW + (tb_dx - k.dxmin)) * k.pixb + c * 16 : 0;
    wsrc[tid] = in ? (tb_wt * k.CU + c) * k.Cout_pad : -1;
  }
  for (int u = tid + 256; u < 4 * k.steps; u += 256) {       // (more than 256 K units: 64 taps x 4-unit blocks)
    int off = 0, v = -1;
    if (u < p_nunits) {
      const int tap = tap0 + (u >> k.lgUPB), c = u & (k.UPB - 1);
      off = ((k.dy[tap] - k.dymin) * k.HW + (k.dx[tap] - k.dxmin)) * k.pixb + c * 16;
      v = ((int)k.wt[tap] * k.CU + c) * k.Cout_pad;
    }
    tbl[u] = off;
    wsrc[u] = v;
  }
#ifdef UBR_CONV_STAMPS
  const unsigned long long t_p2 = __builtin_amdgcn_s_memtime();
#endif
  __syncthreads();
#ifdef UBR_CONV_STAMPS
  const unsigned long long t_p3 = __builtin_amdgcn_s_memtime();
#endif

  f32x4 acc[FW][NT];
#pragma unroll
  for (int i = 0; i < FW; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  int fragbase[FW];
#pragma unroll
  for (int i = 0; i < FW; ++i) {
    const int f = wave * FW + i;
    const int fr = f / TWF, fc = f % TWF;
    fragbase[i] = ((fr * k.S) * k.HW + fc * 16 * k.S) * k.pixb + l16 * k.S * k.pixb;
  }

  const char* xn = k.x + (long)n * k.x_sn;
  const int nitems = k.HH * (int)k.rw;
  const bool has_xf = k.in_scale != nullptr;

  if constexpr (PIPE) {
    // Cin-block pipeline (wide layers, >= 2 cin blocks): the global loads of block b+1 (halo and weight slab) are
    // issued into registers right before the MFMA phase of block b and written to LDS after it, so a block's load
    // round trip hides under the previous block's matrix work instead of being waited for at the top of each block.
    //
    // With two workgroups per CU a SIMD holds two waves, and everything that is not an MFMA is issue-bound (in-kernel
    // cycle stamps, 16x32x32x256 -> 256: of 58.7 k cycles per workgroup 23.7 k were the MFMA loop, 15.1 k the ISSUE of 15
    // guarded loads per block, 17.2 k the BatchNorm transform + LDS stores, 0.07 k waiting for data).  So the per-block
    // bookkeeping is done ONCE per workgroup: a slot's source offset (buffer load, out-of-range = zero padding), its
    // validity bit and its LDS address do not depend on the cin block; the block only moves a scalar offset.
    constexpr int HS = HS_P, WS = conv_pipe_wslots(NT);
    ubr_u4 wv[WS];
    const int c = tid & (k.UPB - 1);
    const int nw = 4 * p_steps * TN;
    int woff[WS];
#pragma unroll
    for (int u = 0; u < WS; ++u) {
      const int i = tid + u * 256;
      int v = kOOR;
      if (i < nw) {
        const int src = wsrc[i / TN];
        if (src >= 0) v = (src + n0 + (i % TN)) * 16;
      }
      woff[u] = v;
    }
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)xn, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)k.w, 0, 0x7fffffff, 0x00020000);
    // BatchNorm constants of every input channel, [unit][sub | scale | shift | lo][CPU] in LDS: a thread's block constants are
    // 4 * CPU consecutive floats
    float* xfc = reinterpret_cast<float*>(smem + k.xfc_off);
    auto load_blk = [&](int blk, bool halo_too = true) {
      const int sh = blk * k.UPB * 16, sw = blk * k.UPB * k.Cout_pad * 16;
      if (halo_too) {
#pragma unroll
        for (int u = 0; u < HS; ++u) hv[u] = __builtin_amdgcn_raw_buffer_load_b128(xr, hoff[u], sh, 0);
      }
#pragma unroll
      for (int u = 0; u < WS; ++u) wv[u] = __builtin_amdgcn_raw_buffer_load_b128(wr, woff[u], sw, 0);
    };
    load_blk(0, false);        // block 0's weight slab (its halo has been in flight since the top of the kernel)
    if (has_xf) {
      if (xf_early) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int ch = tid + i * 256;
          if (ch < k.CU * CPU) {
            float* o = xfc + (ch / CPU) * 4 * CPU + (ch % CPU);
            o[0] = xfr[4 * i + 0]; o[CPU] = xfr[4 * i + 1]; o[2 * CPU] = xfr[4 * i + 2]; o[3 * CPU] = xfr[4 * i + 3];
          }
        }
      } else {
        for (int ch = tid; ch < k.CU * CPU; ch += 256) {
          float* o = xfc + (ch / CPU) * 4 * CPU + (ch % CPU);
          o[0] = k.in_sub[ch]; o[CPU] = k.in_scale[ch]; o[2 * CPU] = k.in_shift[ch]; o[3 * CPU] = k.in_lo[ch];
        }
      }
    }
    // complete slots (every thread has an item) / the partial one: wave-uniform tests in front of the LDS writes
    const int hfull = nitems >> 8, hrem = nitems & 255, wfull = nw >> 8, wrem = nw & 255;
    char* const halo_w = halo + (tid >> k.lgUPB) * k.pixb + c * 16;
    const int halo_inc = (256 >> k.lgUPB) * k.pixb;
    char* const wl_w = wl + tid * 16;
    auto store_blk = [&](int blk) {
#pragma unroll
      for (int u = 0; u < WS; ++u)
        if (u < wfull || (u == wfull && tid < wrem)) *reinterpret_cast<ubr_u4*>(wl_w + u * 4096) = wv[u];
      float xsub[CPU], xsc[CPU], xsh[CPU], xlo[CPU];
      if (has_xf) {
        const float4* cp = reinterpret_cast<const float4*>(xfc + (blk * k.UPB + c) * 4 * CPU);
#pragma unroll
        for (int e = 0; e < CPU; e += 4) {
          const float4 a = cp[e / 4], b = cp[(CPU + e) / 4], cc = cp[(2 * CPU + e) / 4], d = cp[(3 * CPU + e) / 4];
          xsub[e] = a.x; xsub[e + 1] = a.y; xsub[e + 2] = a.z; xsub[e + 3] = a.w;
          xsc[e] = b.x; xsc[e + 1] = b.y; xsc[e + 2] = b.z; xsc[e + 3] = b.w;
          xsh[e] = cc.x; xsh[e + 1] = cc.y; xsh[e + 2] = cc.z; xsh[e + 3] = cc.w;
          xlo[e] = d.x; xlo[e + 1] = d.y; xlo[e + 2] = d.z; xlo[e + 3] = d.w;
        }
      }
#pragma unroll
      for (int u = 0; u < HS; ++u) {
        uint4 v = make_uint4(hv[u].x, hv[u].y, hv[u].z, hv[u].w);
        if (has_xf) {
          float f[CPU];
          ET<T>::unpack(v, f);
          ubr_bnrelu<CPU>(f, xsub, xsc, xsh, xlo);
          const uint4 t4 = ET<T>::pack(f);
          const bool ok = (hok >> u) & 1u;          // padding stays zero
          v.x = ok ? t4.x : 0u; v.y = ok ? t4.y : 0u; v.z = ok ? t4.z : 0u; v.w = ok ? t4.w : 0u;
        }
        if (u < hfull || (u == hfull && tid < hrem)) *reinterpret_cast<uint4*>(halo_w + u * halo_inc) = v;
      }
    };
#ifdef UBR_CONV_STAMPS
    unsigned long long tS = 0, tL = 0, tC = 0, tB = 0, tW = 0, t_ = __builtin_amdgcn_s_memtime();
    const unsigned long long t_begin = t_;
#define UBR_STAMP(acc_) do { unsigned long long n_ = __builtin_amdgcn_s_memtime(); acc_ += n_ - t_; t_ = n_; } while (0)
#else
#define UBR_STAMP(acc_) do { } while (0)
#endif
    UBR_STAMP(tL);
    for (int blk = 0; blk < k.nblk; ++blk) {
      __syncthreads();          // previous block's fragments fully read (block 0: the constants above are visible)
#ifdef UBR_CONV_STAMPS
      UBR_STAMP(tB);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      UBR_STAMP(tW);
#endif
      store_blk(blk);
      UBR_STAMP(tS);
      __syncthreads();
      UBR_STAMP(tB);
      if (blk + 1 < k.nblk) load_blk(blk + 1);
      UBR_STAMP(tL);
      for (int s = 0; s < p_steps; ++s) {
        const int off = tbl[4 * s + q];
        uint4 wf[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j)
          wf[j] = *reinterpret_cast<const uint4*>(wl + (((4 * s + q) * TN) + j * 16 + l16) * 16);
#pragma unroll
        for (int i = 0; i < FW; ++i) {
          const uint4 a = *reinterpret_cast<const uint4*>(halo + fragbase[i] + off);
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = mma_step<T>(acc[i][j], wf[j], a);
        }
      }
      UBR_STAMP(tC);
    }
#ifdef UBR_CONV_STAMPS
    if (k.stamps != nullptr && tid == 0) {
      unsigned long long* o = k.stamps + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16;
      o[0] = tS; o[1] = tL; o[2] = tC; o[3] = tB; o[6] = tW; o[4] = t_ - t_begin; o[5] = t_begin; o[7] = t_entry; o[8] = t_;
      o[10] = t_p1; o[14] = t_p2; o[15] = t_p3;
    }
#endif
  } else
  for (int blk = 0; blk < k.nblk; ++blk) {
    if (blk) __syncthreads();
    // ---- stage the input halo (transform + zero padding) ----
    // 256 % UPB == 0, so a thread's channel unit c is fixed (BatchNorm constants loaded once per cin block).
    // Item i = tid + 256 m maps to halo pixel i >> lgUPB (LDS is linear in i); its (row, column) and global
    // byte offset advance by constant steps with one conditional wrap -- no division or 64-bit multiply per item.
    {
      constexpr int SB = (NT <= 2 && TWF == 2) ? 6 : 4;   // thin tiles: the whole halo (<= 5.3 items per thread for 3x3) in ONE batch -- bytes in flight, not instructions, limit this phase
      const int c = tid & (k.UPB - 1);
      const int ch0 = (blk * k.UPB + c) * CPU;
      float xsub[CPU], xsc[CPU], xsh[CPU], xlo[CPU];
      if (has_xf) {
#pragma unroll
        for (int e = 0; e < CPU; ++e) { xsub[e] = k.in_sub[ch0 + e]; xsc[e] = k.in_scale[ch0 + e]; xsh[e] = k.in_shift[ch0 + e]; xlo[e] = k.in_lo[ch0 + e]; }
      }
      const char* xc = xn + (long)ch0 * ESZ;
      int hy = (int)__umulhi((unsigned)tid, k.rw_magic);
      int j = tid - hy * (int)k.rw;                               // position inside the halo row, in items
      int goff = (hy0 + hy) * k.x_sy32 + (hx0 + (j >> k.lgUPB)) * k.x_sx32;   // byte offset inside the image
      for (int ib = tid; ib < nitems; ib += 256 * SB) {
        uint4 v[SB];
        bool ok[SB];
#pragma unroll
        for (int u = 0; u < SB; ++u) {
          const int iy = hy0 + hy, ix = hx0 + (j >> k.lgUPB);
          ok[u] = (ib + u * 256 < nitems) && (unsigned)iy < (unsigned)k.H && (unsigned)ix < (unsigned)k.W;
          v[u] = make_uint4(0u, 0u, 0u, 0u);
          if (ok[u]) v[u] = ldg16(xc + goff);
          // advance by 256 items
          j += k.step_j; hy += k.step_hy; goff += k.step_goff;
          if (j >= (int)k.rw) { j -= (int)k.rw; hy += 1; goff += k.wrap_goff; }
        }
#pragma unroll
        for (int u = 0; u < SB; ++u) {
          const int i = ib + u * 256;
          if (has_xf && ok[u]) {
            float f[CPU];
            ET<T>::unpack(v[u], f);
            ubr_bnrelu<CPU>(f, xsub, xsc, xsh, xlo);
            v[u] = ET<T>::pack(f);
          }
          if (i < nitems) *reinterpret_cast<uint4*>(halo + (i >> k.lgUPB) * k.pixb + c * 16) = v[u];
        }
      }
    }
    // ---- stage the weight slab [unit][TN][16 B] (zero beyond nunits).  A unit's TN items are contiguous in the
    // packed image, so item i = (unit, nn) needs only a table lookup for the unit's source index; loads are
    // batched (WB in flight per thread). ----
    {
      constexpr int WB = 8;
      const int nw = 4 * p_steps * TN;
      const int boff = blk * k.UPB * k.Cout_pad + n0;
      for (int ib = tid; ib < nw; ib += 256 * WB) {
        uint4 v[WB];
#pragma unroll
        for (int u = 0; u < WB; ++u) {
          const int i = ib + u * 256;
          v[u] = make_uint4(0u, 0u, 0u, 0u);
          if (i < nw) {
            const int src = wsrc[i / TN];
            if (src >= 0) v[u] = ldg16(k.w + ((long)(src + boff + (i % TN))) * 16);
          }
        }
#pragma unroll
        for (int u = 0; u < WB; ++u) {
          const int i = ib + u * 256;
          if (i < nw) *reinterpret_cast<uint4*>(wl + (long)i * 16) = v[u];
        }
      }
    }
    __syncthreads();
    // ---- MFMA over (tap, cin-unit) ----
    for (int s = 0; s < p_steps; ++s) {
      const int off = tbl[4 * s + q];
      uint4 wf[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j)
        wf[j] = *reinterpret_cast<const uint4*>(wl + (((4 * s + q) * TN) + j * 16 + l16) * 16);
#pragma unroll
      for (int i = 0; i < FW; ++i) {
        const uint4 a = *reinterpret_cast<const uint4*>(halo + fragbase[i] + off);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = mma_step<T>(acc[i][j], wf[j], a);
      }
    }
  }

  // ---------------------------------- epilogue ----------------------------------
#ifdef UBR_CONV_STAMPS
  const unsigned long long te0 = __builtin_amdgcn_s_memtime();
#endif
  float bs[NT][4];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ch = n0 + j * 16 + 4 * q + r;
      bs[j][r] = (k.bias != nullptr && ch < k.Cout) ? k.bias[ch] : 0.f;
    }
  float s1[NT][4], s2[NT][4];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[j][r] = 0.f; s2[j][r] = 0.f; }

  // Buffer-addressed epilogue (plain output, no BatchNorm-backward sums): beside an MFMA-bound wave of the other resident workgroup
  // every VALU instruction of this phase costs ~10 cycles, and the general form below spends ~25 of its ~40 instructions per store
  // on 64-bit addresses and per-fragment tests (in-kernel stamps, 16x128x128x64 -> 64: 11 k of the workgroup's 47 k cycles were
  // "values + stores").  Here a store's address is one per-lane offset computed once plus a SCALAR fragment offset; lanes outside the
  // output (ragged tiles, padded channels) get an out-of-range offset, which drops the store / returns a zero addend; all addend and
  // mask loads are issued before the first value is finished.  Same arithmetic per value, in the same order.
  if (k.fast_epi && k.epilogue == 0 && k.bc == nullptr && !(FW % 2 == 0 && sizeof(T) == 2 && k.pair_store)) {
    constexpr int kOut = (int)0x80000000;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const bool full = (ox0 + TW <= k.OW) && (oy0 + TH <= k.OH) && (n0 + TN <= k.Cout);
    const int ysx = (int)k.y_sx, ysy = (int)k.y_sy, asx = (int)k.a_sx, asy = (int)k.a_sy;
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(yb + (long)n * k.y_sn + (long)oy0 * k.y_sy + (long)ox0 * k.y_sx + (long)n0 * ESZ), 0, 0x7fffffff, 0x00020000);
    const bool has_ad = adb != nullptr, has_mask = has_ad && k.ad_mask != nullptr;
    const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(has_ad ? adb + (long)n * k.a_sn + (long)oy0 * k.a_sy + (long)ox0 * k.a_sx + (long)n0 * ESZ : k.x), 0, has_ad ? 0x7fffffff : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t mr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(has_mask ? (const char*)k.ad_mask + (((long)n * k.OH + oy0) * k.OW + ox0) * k.ad_mask_cu + n0 / CPU : k.x), 0, has_mask ? 0x7fffffff : 0, 0x00020000);
    const int lane_y = l16 * ysx + 4 * q * ESZ, lane_a = l16 * asx + 4 * q * ESZ, lane_m = l16 * k.ad_mask_cu + (4 * q) / CPU;
    // per fragment: scalar offsets and (ragged tiles only) the lane's validity
    int so_y[FW], so_a[FW], so_m[FW];
    bool okp[FW];        // pixel inside the output
#pragma unroll
    for (int i = 0; i < FW; ++i) {
      const int f = wv * FW + i, fr = f / TWF, fc = f % TWF;
      so_y[i] = fr * ysy + fc * 16 * ysx; so_a[i] = fr * asy + fc * 16 * asx; so_m[i] = (fr * k.OW + fc * 16) * k.ad_mask_cu;
      okp[i] = full || ((oy0 + fr < k.OH) && (ox0 + fc * 16 + l16 < k.OW));
    }
    bool okc[NT];        // channel group inside Cout
#pragma unroll
    for (int j = 0; j < NT; ++j) okc[j] = full || (n0 + j * 16 + 4 * q < k.Cout);
    ubr_u4 adv[FW][NT];  // raw addend: 8 bytes (16-bit types) or 16
    unsigned admk[FW][NT];
    if (has_ad) {
#pragma unroll
      for (int i = 0; i < FW; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int vo = (okp[i] && okc[j]) ? lane_a : kOut;
          if constexpr (sizeof(T) == 2) {
            const auto t2 = __builtin_amdgcn_raw_buffer_load_b64(ar, vo, so_a[i] + j * 16 * ESZ, 0);
            adv[i][j] = ubr_u4{t2[0], t2[1], 0u, 0u};
          } else {
            adv[i][j] = __builtin_amdgcn_raw_buffer_load_b128(ar, vo, so_a[i] + j * 16 * ESZ, 0);
          }
          if (has_mask) admk[i][j] = __builtin_amdgcn_raw_buffer_load_b8(mr, (okp[i] && okc[j]) ? lane_m : kOut, so_m[i] + j * (16 / CPU), 0);
        }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
      for (int i = 0; i < FW; ++i) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] + bs[j][r];
        if (k.act & 1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        if (has_ad) {
          float a4[4];
          if constexpr (sizeof(T) == 2) {
            const uint2 raw = make_uint2(adv[i][j][0], adv[i][j][1]);
            load4<T>(reinterpret_cast<const char*>(&raw), a4);
          } else {
            a4[0] = __uint_as_float(adv[i][j][0]); a4[1] = __uint_as_float(adv[i][j][1]); a4[2] = __uint_as_float(adv[i][j][2]); a4[3] = __uint_as_float(adv[i][j][3]);
          }
          if (has_mask) {
            const unsigned mb = (admk[i][j] & 0xffu) >> ((4 * q) % CPU);
#pragma unroll
            for (int r = 0; r < 4; ++r) a4[r] = ((mb >> r) & 1u) ? a4[r] : 0.f;
          }
          if (okp[i] && okc[j]) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += a4[r];
          }
        }
        if (k.act & 2) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        if (k.stats != nullptr && okp[i]) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { s1[j][r] += v[r]; s2[j][r] += v[r] * v[r]; }
        }
        const int vo = (okp[i] && okc[j]) ? lane_y : kOut;
        if constexpr (sizeof(T) == 2) {
          uint2 pk;
          store4<T>(reinterpret_cast<char*>(&pk), v);
          __builtin_amdgcn_raw_buffer_store_b64(ubr_u2{pk.x, pk.y}, yr, vo, so_y[i] + j * 16 * ESZ, 0);
        } else {
          buf_store16(ubr_u4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, yr, vo, so_y[i] + j * 16 * ESZ);
        }
      }
    }
  } else
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int ch = n0 + j * 16 + 4 * q;
    // BatchNorm-backward sums: this group's per-channel constants (one set per cout fragment, outside the pixel loop)
    float4 bmu = make_float4(0.f, 0.f, 0.f, 0.f), bsc = bmu, bsh = bmu, bis = bmu;
    const bool bnb = k.bc != nullptr && ch < k.Cout;
    if (bnb) {
      bmu = *reinterpret_cast<const float4*>(k.bmean + ch); bsc = *reinterpret_cast<const float4*>(k.bscale + ch);
      bsh = *reinterpret_cast<const float4*>(k.bshift + ch); bis = *reinterpret_cast<const float4*>(k.binvstd + ch);
    }
    // one fragment's 4-channel group: bias, activations, (masked) addend, statistics
    auto finish = [&](int i, float* v, int& oy, int& ox) -> bool {
      const int f = wave * FW + i;
      oy = oy0 + f / TWF; ox = ox0 + (f % TWF) * 16 + l16;
      const bool valid = (oy < k.OH) && (ox < k.OW);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] + bs[j][r];
      if (k.act & 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
      }
      if (adb != nullptr && valid && ch < k.Cout) {
        float a4[4];
        load4<T>(adb + (long)n * k.a_sn + (long)oy * k.a_sy + (long)ox * k.a_sx + (long)ch * ESZ, a4);
        if (k.ad_mask != nullptr) {
          const unsigned mb = (unsigned)k.ad_mask[(((long)n * k.OH + oy) * k.OW + ox) * k.ad_mask_cu + ch / CPU] >> (ch % CPU);
#pragma unroll
          for (int r = 0; r < 4; ++r) a4[r] = ((mb >> r) & 1u) ? a4[r] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += a4[r];
      }
      if (k.act & 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
      }
      if (k.stats != nullptr && valid) {
        if (k.bc != nullptr) {
          if (bnb) {
            float g4[4], c4[4];
            round4<T>(v, g4);
            load4<T>(k.bc + (long)n * k.bc_sn + (long)oy * k.bc_sy + (long)ox * k.bc_sx + (long)ch * ESZ, c4);
            bnb_accumulate(g4, c4, bmu, bsc, bsh, bis, s1[j], s2[j]);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) { s1[j][r] += v[r]; s2[j][r] += v[r] * v[r]; }
        }
      }
      return valid;
    };
    // 16-bit outputs with whole, 16-byte aligned channel octets: two pixel fragments exchange half their channels
    // (v_permlane16_swap) and every lane stores 16 bytes -- half the store instructions of the 8-byte form, which split every
    // 128-byte line of the output into sixteen pieces
    constexpr bool PAIRS = (FW % 2 == 0) && sizeof(T) == 2;
    if (PAIRS && k.pair_store && k.epilogue == 0) {
#pragma unroll
      for (int i = 0; i < FW; i += 2) {
        float v0[4], v1[4];
        int oy_a, ox_a, oy_b, ox_b;
        const bool ok0 = finish(i, v0, oy_a, ox_a) && ch < k.Cout;
        const bool ok1 = finish(i + (FW > 1 ? 1 : 0), v1, oy_b, ox_b) && ch < k.Cout;
        store_frag_pair<T>(yb + (long)n * k.y_sn + (long)oy_a * k.y_sy + (long)ox_a * k.y_sx + (long)ch * ESZ,
                           yb + (long)n * k.y_sn + (long)oy_b * k.y_sy + (long)ox_b * k.y_sx + (long)ch * ESZ, ok0, ok1, v0, v1, q);
      }
    } else {
#pragma unroll
    for (int i = 0; i < FW; ++i) {
      float v[4];
      int oy, ox;
      const bool valid = finish(i, v, oy, ox);
      if (k.epilogue == 0) {
        if (valid && ch < k.Cout)
          store4<T>(yb + (long)n * k.y_sn + (long)oy * k.y_sy + (long)ox * k.y_sx + (long)ch * ESZ, v);
      } else if (j == 0) {
        // fused LogSoftmax over the first Cout (<=16) channels, fp32 NCHW output
        float m = -3.0e38f;
#pragma unroll
        for (int r = 0; r < 4; ++r) if (ch + r < k.Cout) m = fmaxf(m, v[r]);
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float e = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) if (ch + r < k.Cout) e += expf(v[r] - m);
        e += __shfl_xor(e, 16, 64);
        e += __shfl_xor(e, 32, 64);
        const float lse = m + logf(e);
        if (valid) {
          float* o = reinterpret_cast<float*>(k.y);
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (ch + r < k.Cout) o[(((long)n * k.Cout + ch + r) * k.OH + oy) * k.OW + ox] = v[r] - lse;
        }
      }
    }
    }
  }

#ifdef UBR_CONV_STAMPS
  const unsigned long long te1 = __builtin_amdgcn_s_memtime();
#endif
  if (k.stats != nullptr) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a = wave_quadrow_sum16(s1[j][r]);
        const float b = wave_quadrow_sum16(s2[j][r]);
        if (l16 == 0) {
          red[(wave * TN + j * 16 + 4 * q + r) * 2 + 0] = a;
          red[(wave * TN + j * 16 + 4 * q + r) * 2 + 1] = b;
        }
      }
    __syncthreads();
    if (tid < TN) {
      const int ch = n0 + tid;
      if (ch < k.Cout) {
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { a += (double)red[(w * TN + tid) * 2]; b += (double)red[(w * TN + tid) * 2 + 1]; }
        double* st = k.stats + (size_t)(blockIdx.x % k.nslots) * 2 * k.Cout;
        atomicAdd(&st[ch], a);
        atomicAdd(&st[k.Cout + ch], b);
      }
    }
  }
#ifdef UBR_CONV_STAMPS
  if (PIPE && k.stamps != nullptr) {
    const unsigned long long te2 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (tid == 0) {
      unsigned long long* o = k.stamps + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16;
      o[9] = __builtin_amdgcn_s_memtime(); o[11] = te0; o[12] = te1; o[13] = te2;
    }
  }
#endif
}

// ---------------------------------------------------------------------------------------------
// Thin layers (one cin block: Cin <= 32 for 16-bit types, stride 1): the full- and half-resolution convolutions that
// hold 72 % of the network's activation bytes and run on the HBM roofline, not the MFMA one (AI 105-124 flop/B).
// Same LDS images and MFMA addressing as conv_igemm_kernel; what changes is the ORDER of a workgroup's life, because
// for these tiles (36 KB of HBM traffic, 40-72 MFMAs per wave) latency chains were the cost, not instructions:
//   * the halo loads are issued at kernel entry, before any table is built -- the offset / weight-source tables are
//     computed while they fly;
//   * the weight slab is loaded into registers right behind them (it used to be a second, serialised round trip
//     after the halo had been written to LDS);
//   * the MFMA loop prefetches the next step's tap offset and weight fragments (no LDS read -> address -> LDS read
//     chain per step);
//   * 16-bit outputs are stored as one 16-byte store per lane (two pixel fragments exchange half their channels with
//     v_permlane16_swap) instead of two 8-byte stores: the epilogue was store-issue bound;
//   * BatchNorm+ReLU on load (XF) is one packed FMA + max per channel pair for 16-bit types,
//     x*s + (beta - mean*s): the fp32 cancellation error is far below a bf16 ulp (fp32 keeps the (x-mean)*s+beta form);
//   * <= 128 VGPRs: four workgroups per CU alone, and two beside the 240-VGPR weight-gradient kernels of the side stream
//     (the 165-VGPR kernel dropped to ONE workgroup per CU there).
// ---------------------------------------------------------------------------------------------
struct ThinK {
  const char* x; long x_sn; int x_sy, x_sx; unsigned x_bytes;
  const float *in_sub, *in_scale, *in_shift, *in_lo;
  const char* w;
  char* y; long y_sn; int y_sy, y_sx; unsigned y_bytes;
  const char* ad; long a_sn; int a_sy, a_sx; unsigned a_bytes;
  const float* bias; double* stats;
  int H, W, HW, nitems; unsigned hw_magic;
  int tiles_x, tiles_y, ntiles; unsigned tx_magic, ty_magic;
  int steps, nunits, hy_org, hx_org, dymin, dxmin;
  int Cout, Cout_pad, CU, OH, OW;
  int act, epilogue, wlinear, dbg;
  int wl_off, halo_off, red_off;
  int8_t dy[UBR_MAX_TAPS], dx[UBR_MAX_TAPS];
  uint8_t wt[UBR_MAX_TAPS];
  unsigned long long* stamps;                    // diagnostic build (-DUBR_CONV_STAMPS)
  // training epilogues of the data-gradient convs (EXT = 2: ReLU bit mask on the addend; EXT = 1: BatchNorm-backward sums)
  const uint8_t* ad_mask; int ad_mask_cu; unsigned mask_bytes;
  const char* bc; long bc_sn; int bc_sy, bc_sx; unsigned bc_bytes;
  const float *bmean, *bscale, *bshift, *binvstd;
  int nslots;
  // several output phases from ONE staged halo (transposed conv k4 s2 p1: the four phases share the 3x3 input neighbourhood):
  // phase ph walks K-steps [pstep0, pstep0 + psteps) of the tap table and stores at byte offset pyoff from the output view
  int nphase, pstep0[4], psteps[4], pyoff[4];
};

// exact unsigned division of small operands by a runtime divisor: magic = ceil(2^32 / d) for d >= 2 (valid while n * d < 2^32);
// d == 1 has no 32-bit magic and is encoded as 0

// ROW7 (7x7 layers over 16 channels, 16x32-pixel tile, FW = 8): a wave owns FOUR output rows of two 16-pixel fragments and
// walks the 4 + 6 input rows once per horizontal tap pair.  The halo fragment of input row r is the operand of output row
// r - dy for every vertical tap dy, so one LDS fragment read feeds up to four MFMAs and the seven weight fragments of a
// tap-pair column stay in registers: 27 fragment reads per 56 MFMAs instead of 70.  (Cout = 16 gives no reuse across
// cout tiles: in the generic tap loop every MFMA needs a fresh 1 KB fragment and the 7x7 layers were LDS-read bound,
// 151 us alone, ~190 us beside the weight-gradient stream.)  The horizontal taps are padded from 7 to 8 so that a K-step's two
// taps share a row (56 virtual taps, the host supplies their table; the pad tap has zero weights): +12 % MFMAs.
template <typename T, int FW, int NT, int TWF, int UPB, bool XF, bool LSM, bool ROW7 = false, int EXT = 0>
__global__ __launch_bounds__(256, ((ROW7 || (EXT == 1 && NT == 2)) ? 2 : 3)) void conv_thin_kernel(const ThinK k) {
  ubr_main_prio();
  static_assert(EXT == 0 || (!XF && !LSM), "the training epilogues belong to the data-gradient convs (no transform on load, NHWC output)");
  constexpr int TN = NT * 16;
  constexpr int TH = 4 * FW / TWF;
  constexpr int TW = TWF * 16;
  constexpr int CPU = ET<T>::CPU;
  constexpr int ESZ = 16 / CPU;
  constexpr int LG = UPB == 4 ? 2 : (UPB == 2 ? 1 : 0);
  constexpr int PIXB = thin_pixb(UPB);        // LDS bytes per halo pixel (conflict-free fragment reads: see conv_pixb / thin_pixb)
  constexpr int HS = ROW7 ? 7 : (UPB == 4 ? 6 : 5);        // halo register slots per thread (host: nitems <= 256*HS)
  constexpr int PPS = 256 / UPB;              // halo pixels covered by one slot of the whole workgroup
  constexpr int WB = 4;                       // weight items per thread and batch
  constexpr int FH = ROW7 ? 8 : 4;            // pixel fragments per accumulator group
  static_assert(TWF == 2 && FW % FH == 0, "tile shape");
  static_assert(!ROW7 || (FW == 8 && NT == 1 && UPB == 2), "ROW7 is the 16x32-pixel, 16-cout, 16-cin tile");
  constexpr bool WIDE = ESZ == 2;             // 16-bit outputs: one 16-byte store per lane and fragment pair

  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* tbl = reinterpret_cast<int*>(smem);
  char* wl = smem + k.wl_off;
  char* halo = smem + k.halo_off;
  float* red = reinterpret_cast<float*>(smem + k.red_off);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, l16 = lane & 15;
  const int n0 = blockIdx.y * TN;
  const int c = tid & (UPB - 1);
  const int nw = 4 * k.steps * TN;

  // ---- per-thread geometry, computed ONCE (the workgroup is persistent): every halo slot's position inside the halo and
  // its byte offset relative to the tile origin; per tile a slot then costs an add, a compare and a select ----
  int hxu[HS], relu[HS];
  int hyu[XF ? HS : 1];
#pragma unroll
  for (int u = 0; u < HS; ++u) {
    const int pix = (tid >> LG) + u * PPS;
    const int hy = (int)udiv_magic((unsigned)pix, k.hw_magic), hx = pix - hy * k.HW;
    const bool in = tid + u * 256 < k.nitems;
    hxu[u] = in ? hx : -(1 << 24);            // fails every column check: the slot loads zeros
    relu[u] = hy * k.x_sy + hx * k.x_sx + c * 16;
    if constexpr (XF) hyu[u] = hy;
  }
  char* const halo_w = halo + (tid >> LG) * PIXB + c * 16;

  ubr_u4 hv[HS];
  unsigned hok = 0u;
  auto load_halo = [&](int tile) {
    const unsigned t = (unsigned)__builtin_amdgcn_readfirstlane(tile);
    const unsigned r = udiv_magic(t, k.tx_magic), tx = t - r * (unsigned)k.tiles_x;
    const unsigned n = udiv_magic(r, k.ty_magic), ty = r - n * (unsigned)k.tiles_y;
    const int hy0 = (int)ty * TH + k.hy_org, hx0 = (int)tx * TW + k.hx_org;
    // rows outside the image fall outside the buffer range (negative offsets wrap to huge unsigned ones) and read as zero;
    // columns outside it would alias the neighbouring row, so they are checked and sent out of range explicitly
    __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)(k.x + (long)n * k.x_sn), 0, (int)k.x_bytes, 0x00020000);
    const int goff = hy0 * k.x_sy + hx0 * k.x_sx;
    hok = 0u;
#pragma unroll
    for (int u = 0; u < HS; ++u) {
      const bool ok = (unsigned)(hx0 + hxu[u]) < (unsigned)k.W;
      int vo = ok ? relu[u] + goff : -1;
      if (k.dbg & 1) vo = -1;
      hv[u] = __builtin_amdgcn_raw_buffer_load_b128(xr, vo, 0, 0);
      if constexpr (XF) hok |= (unsigned)(ok & ((unsigned)(hy0 + hyu[u]) < (unsigned)k.H)) << u;      // (branch-free: one select per slot)
    }
  };

  // ---- prologue: everything with a memory latency is put in flight first ----
  const int ntiles = k.ntiles;
  int tile = blockIdx.x;
  load_halo(tile);
  // weight slab.  For the natural tap order with one cout tile the slab IS the packed image (unit u = tap*UPB + c at
  // image item u*TN + nn): a linear copy with no table in front of it.
  uint4 wv[WB];
  if (k.wlinear) {
    const int nvalid = k.nunits * TN;
#pragma unroll
    for (int u = 0; u < WB; ++u) {
      const int i = tid + u * 256;
      wv[u] = make_uint4(0u, 0u, 0u, 0u);
      if (i < nvalid) wv[u] = ldg16(k.w + (long)i * 16);
    }
  }
  float bs[NT][4];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ch = n0 + j * 16 + 4 * q + r;
      bs[j][r] = (k.bias != nullptr && ch < k.Cout) ? k.bias[ch] : 0.f;
    }
  // BatchNorm+ReLU-on-load constants live in LDS ([4][UPB*CPU] floats behind the statistics scratch), not in 32 registers held
  // across the MFMA loop: the XF variants of the 32-channel tiles spilled.  (Same (x - mean)*scale + beta form as every other
  // kernel of the library: an activation re-formed on load by the forward conv, the weight gradient and the BatchNorm
  // backward must be the SAME bits, whichever kernel or tile computes it.)
  float* xfc = red + 4 * TN * 2;
  if constexpr (XF) {
    for (int i = tid; i < 4 * UPB * CPU; i += 256) {
      const int which = i / (UPB * CPU), ch = i - which * (UPB * CPU);
      const float* src = which == 0 ? k.in_sub : (which == 1 ? k.in_scale : (which == 2 ? k.in_shift : k.in_lo));
      xfc[i] = src[ch];
    }
  }
  if constexpr (EXT == 1) {
    // BatchNorm-backward sums: mean | scale | shift | invstd of this workgroup's TN output channels, [4][TN] floats
    for (int i = tid; i < 4 * TN; i += 256) {
      const int which = i / TN, ch = n0 + (i - which * TN);
      const float* src = which == 0 ? k.bmean : (which == 1 ? k.bscale : (which == 2 ? k.bshift : k.binvstd));
      xfc[i] = ch < k.Cout ? src[ch] : 0.f;
    }
  }
  int* wsrc = tbl + 4 * k.steps;
  for (int u = tid; u < 4 * k.steps; u += 256) {
    int off = 0, v = -1;
    if (u < k.nunits) {
      const int tap = u >> LG, cc = u & (UPB - 1);
      off = ((k.dy[tap] - k.dymin) * k.HW + (k.dx[tap] - k.dxmin)) * PIXB + cc * 16;
      v = k.wt[tap] == 255 ? -1 : ((int)k.wt[tap] * k.CU + cc) * k.Cout_pad;       // 255: padding tap (zero weights)
    }
    tbl[u] = off;
    wsrc[u] = v;
  }
  if (k.wlinear) {
#pragma unroll
    for (int u = 0; u < WB; ++u) {
      const int i = tid + u * 256;
      if (i < nw) *reinterpret_cast<uint4*>(wl + (long)i * 16) = wv[u];
    }
    const int nvalid = k.nunits * TN;
    for (int ib = tid + 256 * WB; ib < nw; ib += 256 * WB) {       // 49-tap layers only
#pragma unroll
      for (int u = 0; u < WB; ++u) {
        const int i = ib + u * 256;
        wv[u] = make_uint4(0u, 0u, 0u, 0u);
        if (i < nvalid) wv[u] = ldg16(k.w + (long)i * 16);
      }
#pragma unroll
      for (int u = 0; u < WB; ++u) {
        const int i = ib + u * 256;
        if (i < nw) *reinterpret_cast<uint4*>(wl + (long)i * 16) = wv[u];
      }
    }
  } else {
    __syncthreads();        // wsrc
    for (int ib = tid; ib < nw; ib += 256 * WB) {
#pragma unroll
      for (int u = 0; u < WB; ++u) {
        const int i = ib + u * 256;
        wv[u] = make_uint4(0u, 0u, 0u, 0u);
        if (i < nw) {
          const int src = wsrc[i / TN];
          if (src >= 0) wv[u] = ldg16(k.w + ((long)(src + n0 + (i % TN))) * 16);
        }
      }
#pragma unroll
      for (int u = 0; u < WB; ++u) {
        const int i = ib + u * 256;
        if (i < nw) *reinterpret_cast<uint4*>(wl + (long)i * 16) = wv[u];
      }
    }
  }

  if constexpr (XF || EXT == 1) __syncthreads();          // the BatchNorm constants in LDS are read before the tile loop's first barrier
  // fragment addressing: fragment f = wave*FW + g0 + i of the tile sits at row f/2, column half f%2
  const int fb0 = (((wave * FW) / TWF) * k.HW + l16) * PIXB;          // group 0, fragment 0 of this wave
  const int rowb = k.HW * PIXB;
  // output addressing (tiles are exact: the host only takes this kernel when OH % TH == 0 and OW % TW == 0)
  int vo_out[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    if constexpr (WIDE) vo_out[j] = ((wave * FW) / TWF) * k.y_sy + ((q & 1) * 16 + l16) * k.y_sx + (n0 + j * 16 + 8 * (q >> 1)) * ESZ;
    else vo_out[j] = ((wave * FW) / TWF) * k.y_sy + l16 * k.y_sx + (n0 + j * 16 + 4 * q) * ESZ;
  }
  int vo_ad[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) vo_ad[j] = ((wave * FW) / TWF) * k.a_sy + l16 * k.a_sx + (n0 + j * 16 + 4 * q) * ESZ;
  int vo_c[EXT == 1 ? NT : 1], vo_m[EXT == 2 ? NT : 1];
  if constexpr (EXT == 1) {
#pragma unroll
    for (int j = 0; j < NT; ++j) vo_c[j] = ((wave * FW) / TWF) * k.bc_sy + l16 * k.bc_sx + (n0 + j * 16 + 4 * q) * ESZ;
  }
  if constexpr (EXT == 2) {
#pragma unroll
    for (int j = 0; j < NT; ++j) vo_m[j] = (((wave * FW) / TWF) * k.OW + l16) * k.ad_mask_cu + (n0 + j * 16 + 4 * q) / CPU;
  }
  float s1[NT][4], s2[NT][4];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[j][r] = 0.f; s2[j][r] = 0.f; }

#ifdef UBR_CONV_STAMPS
  unsigned long long tS = 0, tL = 0, tC = 0, tE = 0, tB = 0, tW = 0, t_ = __builtin_amdgcn_s_memtime();
  const unsigned long long t_begin = t_;
#define UBR_TSTAMP(acc_) do { unsigned long long n_ = __builtin_amdgcn_s_memtime(); acc_ += n_ - t_; t_ = n_; } while (0)
#else
#define UBR_TSTAMP(acc_) do { } while (0)
#endif
  while (tile < ntiles) {
#ifdef UBR_CONV_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    UBR_TSTAMP(tW);
#endif
    // ---- halo of this tile: transform, write to LDS (slots beyond the halo hold zeros and land in the slack of the region) ----
    float xsub[XF ? CPU : 1], xsc[XF ? CPU : 1], xsh[XF ? CPU : 1], xlo[XF ? CPU : 1];
    if constexpr (XF) {
#pragma unroll
      for (int e = 0; e < CPU; ++e) {
        xsub[e] = xfc[0 * UPB * CPU + c * CPU + e]; xsc[e] = xfc[1 * UPB * CPU + c * CPU + e];
        xsh[e] = xfc[2 * UPB * CPU + c * CPU + e]; xlo[e] = xfc[3 * UPB * CPU + c * CPU + e];
      }
    }
#pragma unroll
    for (int u = 0; u < HS; ++u) {
      uint4 v = __builtin_bit_cast(uint4, hv[u]);
      if constexpr (XF) {
        if ((hok >> u) & 1u) {
          float f[CPU];
          ET<T>::unpack(v, f);
          ubr_bnrelu<CPU>(f, xsub, xsc, xsh, xlo);
          v = ET<T>::pack(f);
        }
      }
      if (tid + u * 256 < k.nitems) *reinterpret_cast<uint4*>(halo_w + u * (PPS * PIXB)) = v;     // (the region holds the halo exactly)
    }
    UBR_TSTAMP(tS);
    __syncthreads();
    UBR_TSTAMP(tB);
    const unsigned t = (unsigned)__builtin_amdgcn_readfirstlane(tile);
    const unsigned rr = udiv_magic(t, k.tx_magic), tx = t - rr * (unsigned)k.tiles_x;
    const unsigned n = udiv_magic(rr, k.ty_magic), ty = rr - n * (unsigned)k.tiles_y;
    const int oy0 = (int)ty * TH, ox0 = (int)tx * TW;
    const int next = tile + gridDim.x;
    if (next < ntiles) load_halo(next);       // in flight during the MFMAs and stores below
    __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)(k.y + (long)n * k.y_sn), 0, (int)k.y_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc((void*)(k.ad != nullptr ? k.ad + (long)n * k.a_sn : k.x), 0, k.ad != nullptr ? (int)k.a_bytes : 0, 0x00020000);
    const int ybase = oy0 * k.y_sy + ox0 * k.y_sx, abase = oy0 * k.a_sy + ox0 * k.a_sx;
    __amdgpu_buffer_rsrc_t cr = yr, mr = yr;
    int cbase = 0, mbase = 0;
    if constexpr (EXT == 1) {
      cr = __builtin_amdgcn_make_buffer_rsrc((void*)(k.bc + (long)n * k.bc_sn), 0, (int)k.bc_bytes, 0x00020000);
      cbase = oy0 * k.bc_sy + ox0 * k.bc_sx;
    }
    if constexpr (EXT == 2) {
      mr = __builtin_amdgcn_make_buffer_rsrc((void*)(k.ad_mask + (long)n * k.mask_bytes), 0, (int)k.mask_bytes, 0x00020000);
      mbase = (oy0 * k.OW + ox0) * k.ad_mask_cu;
    }
    UBR_TSTAMP(tL);

#pragma unroll 1
    for (int g0 = 0; g0 < FW; g0 += FH) {
      // addend (gradient accumulation of the data-gradient convs): in flight under the MFMA loop
      ubr_u4 adv4[FH][NT];     // fp32: 16 bytes per fragment
      ubr_u2 adv2[FH][NT];     // 16-bit types: 8 bytes
      if (k.ad != nullptr) {
#pragma unroll
        for (int i = 0; i < FH; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const int so = abase + ((g0 + i) / TWF) * k.a_sy;
            if constexpr (WIDE) adv2[i][j] = __builtin_amdgcn_raw_buffer_load_b64(ar, vo_ad[j] + (i % TWF) * 16 * k.a_sx, so, 0);
            else adv4[i][j] = __builtin_amdgcn_raw_buffer_load_b128(ar, vo_ad[j] + (i % TWF) * 16 * k.a_sx, so, 0);
          }
      }
      unsigned admk[EXT == 2 ? FH : 1][EXT == 2 ? NT : 1];      // ReLU bit-mask byte of this lane's channel unit
      ubr_u4 cv4[EXT == 1 ? FH : 1][EXT == 1 ? NT : 1];         // saved activation c of the BatchNorm-backward sums (fp32: 16 bytes)
      ubr_u2 cv2[EXT == 1 ? FH : 1][EXT == 1 ? NT : 1];         // (16-bit types: 8 bytes)
      if constexpr (EXT == 2) {
#pragma unroll
        for (int i = 0; i < FH; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            admk[i][j] = __builtin_amdgcn_raw_buffer_load_b8(mr, vo_m[j] + (i % TWF) * 16 * k.ad_mask_cu, mbase + ((g0 + i) / TWF) * k.OW * k.ad_mask_cu, 0);
      }
      if constexpr (EXT == 1) {
#pragma unroll
        for (int i = 0; i < FH; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const int so = cbase + ((g0 + i) / TWF) * k.bc_sy;
            if constexpr (WIDE) cv2[i][j] = __builtin_amdgcn_raw_buffer_load_b64(cr, vo_c[j] + (i % TWF) * 16 * k.bc_sx, so, 0);
            else cv4[i][j] = __builtin_amdgcn_raw_buffer_load_b128(cr, vo_c[j] + (i % TWF) * 16 * k.bc_sx, so, 0);
          }
      }
      // (phased launches: every phase of this fragment group from the same halo, one after the other)
      const int nph = (ROW7 || LSM || EXT != 0) ? 1 : k.nphase;
#pragma unroll 1
      for (int ph = 0; ph < nph; ++ph) {
      const int s_lo = nph > 1 ? k.pstep0[ph] : 0;
      const int s_cnt = nph > 1 ? k.psteps[ph] : k.steps;
      const int y_ph = nph > 1 ? k.pyoff[ph] : 0;
      f32x4 acc[FH][NT];
#pragma unroll
      for (int i = 0; i < FH; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      const char* hb = halo + fb0 + (g0 / TWF) * rowb;
      if constexpr (ROW7) {
        constexpr int ROWB = (TW + 6) * PIXB;          // halo row pitch (host: HW == TW + 6)
#pragma unroll 1
        for (int pp = 0; pp < 4; ++pp) {               // horizontal tap pairs (dx = 2pp, 2pp + 1; the last one holds the pad tap)
          const char* pb = hb + tbl[4 * pp + q];
          const char* pw = wl + ((4 * pp + q) * TN + l16) * 16;
          uint4 w7[7];
#pragma unroll
          for (int dy = 0; dy < 7; ++dy) w7[dy] = *reinterpret_cast<const uint4*>(pw + dy * (16 * TN * 16));   // step dy*4 + pp
#pragma unroll
          for (int ir = 0; ir < 10; ++ir) {            // input rows of the four output rows
            const uint4 b0 = *reinterpret_cast<const uint4*>(pb + ir * ROWB);
            const uint4 b1 = *reinterpret_cast<const uint4*>(pb + ir * ROWB + 16 * PIXB);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int dy = ir - r;
              if (dy >= 0 && dy <= 6) {
                acc[2 * r][0] = mma_step<T>(acc[2 * r][0], w7[dy], b0);
                acc[2 * r + 1][0] = mma_step<T>(acc[2 * r + 1][0], w7[dy], b1);
              }
            }
          }
        }
      } else {
      int off_n = tbl[4 * s_lo + q];
      uint4 wf_n[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) wf_n[j] = *reinterpret_cast<const uint4*>(wl + (((4 * s_lo + q) * TN) + j * 16 + l16) * 16);
      const int nsteps = s_lo + ((k.dbg & 2) ? 1 : s_cnt);
      for (int s = s_lo; s < nsteps; ++s) {
        const char* pa = hb + off_n;
        uint4 wf[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) wf[j] = wf_n[j];
        if (s + 1 < nsteps) {
          off_n = tbl[4 * (s + 1) + q];
#pragma unroll
          for (int j = 0; j < NT; ++j) wf_n[j] = *reinterpret_cast<const uint4*>(wl + (((4 * (s + 1) + q) * TN) + j * 16 + l16) * 16);
        }
        uint4 a[FH];
        a[0] = *reinterpret_cast<const uint4*>(pa);
        a[1] = *reinterpret_cast<const uint4*>(pa + 16 * PIXB);
        a[2] = *reinterpret_cast<const uint4*>(pa + rowb);
        a[3] = *reinterpret_cast<const uint4*>(pa + rowb + 16 * PIXB);
#pragma unroll
        for (int i = 0; i < FH; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = mma_step<T>(acc[i][j], wf[j], a[i]);
      }
      }   // !ROW7
      UBR_TSTAMP(tC);

      // ---- epilogue of this fragment group ----
#pragma unroll
      for (int i = 0; i < FH; i += 2) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          float v[2][4];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[h][r] = acc[i + h][j][r] + bs[j][r];     // (same order of additions as conv_igemm_kernel)
            if (k.act & 1) {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[h][r] = fmaxf(v[h][r], 0.f);
            }
            if (k.ad != nullptr) {
              float a4[4];
              if constexpr (WIDE) {
                const uint2 raw = make_uint2(adv2[i + h][j][0], adv2[i + h][j][1]);
                load4<T>(reinterpret_cast<const char*>(&raw), a4);
              } else {
                const uint4 raw = __builtin_bit_cast(uint4, adv4[i + h][j]);
                ET<T>::unpack(raw, a4);
              }
              if constexpr (EXT == 2) {
                // (a * bit + v as one fused multiply-add: exact for bit in {0, 1}, i.e. the same sum as the unmasked form)
                const unsigned mb = (admk[i + h][j] & 0xffu) >> ((n0 + j * 16 + 4 * q) % CPU);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[h][r] = __builtin_fmaf(a4[r], (float)((mb >> r) & 1u), v[h][r]);
              } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[h][r] += a4[r];
              }
            }
            if (k.act & 2) {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[h][r] = fmaxf(v[h][r], 0.f);
            }
            if constexpr (EXT == 1) {
              float g4[4], c4[4];
              round4<T>(v[h], g4);
              if constexpr (WIDE) {
                const uint2 raw = make_uint2(cv2[i + h][j][0], cv2[i + h][j][1]);
                load4<T>(reinterpret_cast<const char*>(&raw), c4);
              } else {
                const uint4 raw = __builtin_bit_cast(uint4, cv4[i + h][j]);
                ET<T>::unpack(raw, c4);
              }
              const float* cc = xfc + j * 16 + 4 * q;
              bnb_accumulate(g4, c4, *reinterpret_cast<const float4*>(cc), *reinterpret_cast<const float4*>(cc + TN),
                             *reinterpret_cast<const float4*>(cc + 2 * TN), *reinterpret_cast<const float4*>(cc + 3 * TN), s1[j], s2[j]);
            } else if (k.stats != nullptr) {
#pragma unroll
              for (int r = 0; r < 4; ++r) { s1[j][r] += v[h][r]; s2[j][r] += v[h][r] * v[h][r]; }
            }
          }
          const int so = ybase + ((g0 + i) / TWF) * k.y_sy + y_ph;
          if constexpr (!LSM) {
            if (k.dbg & 4) {
              if (v[0][0] == 1.2345e-30f && v[1][1] == 1.2345e-30f) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[0][0]), yr, vo_out[j], so, 0);
            } else if constexpr (WIDE) {
              const uint4 pk = pack_swap_pair<T>(v[0], v[1]);
              ubr_u4 o = {pk.x, pk.y, pk.z, pk.w};
              buf_store16(o, yr, vo_out[j], so);
            } else {
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                ubr_u4 o = {__float_as_uint(v[h][0]), __float_as_uint(v[h][1]), __float_as_uint(v[h][2]), __float_as_uint(v[h][3])};
                buf_store16(o, yr, vo_out[j] + h * 16 * k.y_sx, so);
              }
            }
          } else if (j == 0) {
            (void)so;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              // fused LogSoftmax over the first Cout (<=16) channels, fp32 NCHW output
              const int ch = 4 * q;
              const int f = wave * FW + g0 + i + h;
              const int oy = oy0 + f / TWF, ox = ox0 + (f % TWF) * 16 + l16;
              float m = -3.0e38f;
#pragma unroll
              for (int r = 0; r < 4; ++r) if (ch + r < k.Cout) m = fmaxf(m, v[h][r]);
              m = fmaxf(m, __shfl_xor(m, 16, 64));
              m = fmaxf(m, __shfl_xor(m, 32, 64));
              float e = 0.f;
#pragma unroll
              for (int r = 0; r < 4; ++r) if (ch + r < k.Cout) e += expf(v[h][r] - m);
              e += __shfl_xor(e, 16, 64);
              e += __shfl_xor(e, 32, 64);
              const float lse = m + logf(e);
              float* o = reinterpret_cast<float*>(k.y);
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (ch + r < k.Cout) o[(((long)n * k.Cout + ch + r) * k.OH + oy) * k.OW + ox] = v[h][r] - lse;
            }
          }
        }
      }
      }   // phases
      UBR_TSTAMP(tE);
    }   // fragment groups
    tile = next;
    if (tile < ntiles) __syncthreads();        // every wave is done with this tile's halo image before it is overwritten
    UBR_TSTAMP(tB);
  }
#ifdef UBR_CONV_STAMPS
  if (k.stamps != nullptr && tid == 0) {
    unsigned long long* o = k.stamps + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16;
    o[0] = tS; o[1] = tL; o[2] = tC; o[3] = tB; o[6] = tW; o[10] = tE; o[4] = t_ - t_begin; o[5] = t_begin; o[7] = t_begin; o[8] = t_; o[9] = t_;
  }
#endif

  if (k.stats != nullptr) {
    // (fp32 partial sums span all tiles of the workgroup: <= 16 tiles x 128 pixels per lane here, then fp64)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a = wave_quadrow_sum16(s1[j][r]);
        const float b = wave_quadrow_sum16(s2[j][r]);
        if (l16 == 0) {
          red[(wave * TN + j * 16 + 4 * q + r) * 2 + 0] = a;
          red[(wave * TN + j * 16 + 4 * q + r) * 2 + 1] = b;
        }
      }
    __syncthreads();
    if (tid < TN) {
      const int ch = n0 + tid;
      if (ch < k.Cout) {
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { a += (double)red[(w * TN + tid) * 2]; b += (double)red[(w * TN + tid) * 2 + 1]; }
        double* st = k.stats + (size_t)(blockIdx.x % k.nslots) * 2 * k.Cout;
        atomicAdd(&st[ch], a);
        atomicAdd(&st[k.Cout + ch], b);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Wide layers (Cout a multiple of 64, >= 1 cin block of 4 units, stride 1): the 128^2 ... 16^2 maps.  conv_igemm_kernel walks a
// cin block as  barrier - registers to LDS (BatchNorm transform) - barrier - next block's loads - MFMA loop,  one phase after the
// other in every wave; at 32^2 and below a layer has only 256 workgroups, so nothing else on the CU fills the staging phases and
// the matrix pipe idles ~60 % of a workgroup's life (in-kernel stamps), and at 128^2 (1024 small workgroups
// of two cin blocks) the exposed prologue / epilogue latencies of each workgroup dominate (53 us for 11 us of HBM traffic).
//
// conv_pc_kernel: ONE persistent workgroup of 8 waves per CU.
//   * waves 4-7 (producers, one per SIMD) stage: buffer loads two blocks ahead into two register sets, BatchNorm transform,
//     LDS writes into the buffer the consumers are NOT reading;
//   * waves 0-3 (consumers, one per SIMD) run the MFMA loop of conv_igemm_kernel's 64-cout tile (FW pixel fragments x 4 cout
//     fragments per wave) out of the other buffer, and a tile's epilogue (bias / activation / addend / statistics / stores);
//   * one s_barrier per cin block (two LDS buffers of halo + weight slab), and the block stream runs ACROSS tiles: a workgroup
//     walks (cout tile, pixel tile) units with a stride of the grid, so the next tile's first block is staged under the current
//     tile's last MFMA loop and epilogue.
// Arithmetic per output element is conv_igemm_kernel's (same order over cin blocks, taps and units): results do not depend on
// which kernel a launch gets.
// ---------------------------------------------------------------------------------------------
template <typename T, int FW, int TWF, int STEPS>
__global__ __launch_bounds__(512) void conv_pc_kernel(const ConvK k) {
  constexpr int NT = 4, TN = 64;
  constexpr int F = 4 * FW;
  constexpr int TH = F / TWF;
  constexpr int TW = TWF * 16;
  constexpr int CPU = ET<T>::CPU;
  constexpr int ESZ = 16 / CPU;
  constexpr int HS = conv_pipe_hslots(FW, TWF), WS = conv_pipe_wslots(NT);
  constexpr int kOOR = (int)0x80000000;        // beyond any num_records: the load returns zeros
  static_assert(F % TWF == 0, "tile shape");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* tbl = reinterpret_cast<int*>(smem);
  int* wsrc = tbl + 4 * k.steps;
  float* xfc = reinterpret_cast<float*>(smem + k.xfc_off);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, l16 = lane & 15;
  const bool producer = wave >= 4;
  const int ptid = tid & 255;
  const bool has_xf = k.in_scale != nullptr;

  for (int u = tid; u < 4 * k.steps; u += 512) {
    int off = 0, v = -1;
    if (u < k.nunits) {
      const int tap = u >> 2, cc = u & 3;
      off = ((k.dy[tap] - k.dymin) * k.HW + (k.dx[tap] - k.dxmin)) * k.pixb + cc * 16;
      v = ((int)k.wt[tap] * k.CU + cc) * k.Cout_pad;      // per-unit weight-slab source index (16-byte items, cin block 0, channel 0)
    }
    tbl[u] = off;
    wsrc[u] = v;
  }
  if (has_xf) {
    // BatchNorm constants of every input channel, [unit][sub | scale | shift | lo][CPU]
    for (int ch = tid; ch < k.CU * CPU; ch += 512) {
      float* o = xfc + (ch / CPU) * 4 * CPU + (ch % CPU);
      o[0] = k.in_sub[ch]; o[CPU] = k.in_scale[ch]; o[2 * CPU] = k.in_shift[ch]; o[3 * CPU] = k.in_lo[ch];
    }
  }
  __syncthreads();

  const int ntiles = k.tiles_x * k.tiles_y * k.N;
  const int total = ntiles * k.ncot;
  const int G = (int)gridDim.x;
  const int nmine = (int)blockIdx.x < total ? (total - (int)blockIdx.x + G - 1) / G : 0;
  const int niter = nmine * k.nblk;

  // per-channel statistics of the consumers (fp32 partials over this workgroup's tiles of one cout tile, then fp64)
  float s1[NT][4], s2[NT][4];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[j][r] = 0.f; s2[j][r] = 0.f; }

  if (producer) {
    // ------------------------------------------------ producers ------------------------------------------------
    const int c = ptid & 3;
    const int nit = k.HH * (int)k.rw;                 // halo items of one block
    const int nw = 4 * k.steps * TN;                  // weight items of one block
    // slot geometry that does not depend on the unit: (row, column) inside the halo, byte offset relative to the halo origin
    int h_rc[HS], h_rel[HS];
#pragma unroll
    for (int u = 0; u < HS; ++u) {
      const int it = ptid + u * 256;
      const int hy = (int)__umulhi((unsigned)it, k.rw_magic);
      const int hx = (it - hy * (int)k.rw) >> 2;
      h_rc[u] = it < nit ? ((hy << 16) | hx) : -1;
      h_rel[u] = hy * k.x_sy32 + hx * k.x_sx32 + c * 16;
    }
    int w_rel[WS];
#pragma unroll
    for (int u = 0; u < WS; ++u) {
      const int i = ptid + u * 256;
      int v = kOOR;
      if (i < nw) {
        const int src = wsrc[i / TN];
        if (src >= 0) v = (src + (i % TN)) * 16;
      }
      w_rel[u] = v;
    }
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)k.w, 0, 0x7fffffff, 0x00020000);
    const int x_img = k.H * k.x_sy32;                 // bytes of one image: rows above / below it read as zero
    const int hfull = nit >> 8, hrem = nit & 255, wfull = nw >> 8, wrem = nw & 255;
    const int halo_inc = 64 * k.pixb;                 // 256 items = 64 pixels
    char* const halo_w0 = smem + k.halo_off + (ptid >> 2) * k.pixb + c * 16;
    char* const wl_w0 = smem + k.wl_off + ptid * 16;

    // load cursor: the (unit, block) whose loads are issued next
    int lu = (int)blockIdx.x, lblk = 0;
    int hoff[HS], woff[WS];
    unsigned hok_unit = 0u;
    __amdgpu_buffer_rsrc_t xr = wr;
    auto issue = [&](ubr_u4 (&hv)[HS], ubr_u4 (&wv)[WS], unsigned& hokm) {
      if (lblk == 0) {
        const int u_ = __builtin_amdgcn_readfirstlane(lu);
        const int ct = u_ / ntiles;
        int t = u_ - ct * ntiles;
        const int tx = t % k.tiles_x; t /= k.tiles_x;
        const int ty = t % k.tiles_y;
        const int n = t / k.tiles_y;
        const int hy0 = ty * TH + k.iy0 + k.dymin, hx0 = tx * TW + k.ix0 + k.dxmin;
        xr = __builtin_amdgcn_make_buffer_rsrc((void*)(k.x + (long)n * k.x_sn), 0, x_img, 0x00020000);
        const int org = hy0 * k.x_sy32 + hx0 * k.x_sx32;     // negative above the image: the sum wraps out of range
        hok_unit = 0u;
#pragma unroll
        for (int u = 0; u < HS; ++u) {
          const int hy = h_rc[u] >> 16, hx = h_rc[u] & 0xffff;
          const bool okx = h_rc[u] >= 0 && (unsigned)(hx0 + hx) < (unsigned)k.W;
          hoff[u] = okx ? h_rel[u] + org : kOOR;
          hok_unit |= (okx && (unsigned)(hy0 + hy) < (unsigned)k.H) ? (1u << u) : 0u;
        }
        const int n0b = ct * TN * 16;
#pragma unroll
        for (int u = 0; u < WS; ++u) woff[u] = w_rel[u] == kOOR ? kOOR : w_rel[u] + n0b;
      }
      const int sh = lblk * 64, sw = lblk * 4 * k.Cout_pad * 16;
#pragma unroll
      for (int u = 0; u < HS; ++u) hv[u] = __builtin_amdgcn_raw_buffer_load_b128(xr, hoff[u], sh, 0);
#pragma unroll
      for (int u = 0; u < WS; ++u) wv[u] = __builtin_amdgcn_raw_buffer_load_b128(wr, woff[u], sw, 0);
      hokm = hok_unit;
      if (++lblk == k.nblk) { lblk = 0; lu += G; }
    };
    // stage cursor: cin block (inside its unit) of the register set that is written to LDS next
    int sblk = 0;
    auto stage = [&](const ubr_u4 (&hv)[HS], const ubr_u4 (&wv)[WS], unsigned hokm, int buf) {
      char* const wl_w = wl_w0 + buf * k.buf_stride;
      char* const halo_w = halo_w0 + buf * k.buf_stride;
#pragma unroll
      for (int u = 0; u < WS; ++u)
        if (u < wfull || (u == wfull && ptid < wrem)) *reinterpret_cast<ubr_u4*>(wl_w + u * 4096) = wv[u];
      float xsub[CPU], xsc[CPU], xsh[CPU], xlo[CPU];
      if (has_xf) {
        const float4* cp = reinterpret_cast<const float4*>(xfc + (sblk * 4 + c) * 4 * CPU);
#pragma unroll
        for (int e = 0; e < CPU; e += 4) {
          const float4 a = cp[e / 4], b = cp[(CPU + e) / 4], cc = cp[(2 * CPU + e) / 4], d = cp[(3 * CPU + e) / 4];
          xsub[e] = a.x; xsub[e + 1] = a.y; xsub[e + 2] = a.z; xsub[e + 3] = a.w;
          xsc[e] = b.x; xsc[e + 1] = b.y; xsc[e + 2] = b.z; xsc[e + 3] = b.w;
          xsh[e] = cc.x; xsh[e + 1] = cc.y; xsh[e + 2] = cc.z; xsh[e + 3] = cc.w;
          xlo[e] = d.x; xlo[e + 1] = d.y; xlo[e + 2] = d.z; xlo[e + 3] = d.w;
        }
      }
#pragma unroll
      for (int u = 0; u < HS; ++u) {
        uint4 v = make_uint4(hv[u].x, hv[u].y, hv[u].z, hv[u].w);
        if (has_xf) {
          float f[CPU];
          ET<T>::unpack(v, f);
          ubr_bnrelu<CPU>(f, xsub, xsc, xsh, xlo);
          const uint4 t4 = ET<T>::pack(f);
          const bool ok = (hokm >> u) & 1u;          // padding stays zero
          v.x = ok ? t4.x : 0u; v.y = ok ? t4.y : 0u; v.z = ok ? t4.z : 0u; v.w = ok ? t4.w : 0u;
        }
        if (u < hfull || (u == hfull && ptid < hrem)) *reinterpret_cast<uint4*>(halo_w + u * halo_inc) = v;
      }
      if (++sblk == k.nblk) sblk = 0;
    };

    ubr_u4 hvA[HS], wvA[WS], hvB[HS], wvB[WS];
    unsigned hokA = 0u, hokB = 0u;
#ifdef UBR_CONV_STAMPS
    unsigned long long tS = 0, tL = 0, tB = 0, tW = 0, t_ = __builtin_amdgcn_s_memtime();
    const unsigned long long t_begin = t_;
#define UBR_PSTAMP(acc_) do { unsigned long long n_ = __builtin_amdgcn_s_memtime(); acc_ += n_ - t_; t_ = n_; } while (0)
#define UBR_PWAIT(n_) do { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(n_) : "memory"); UBR_PSTAMP(tW); } while (0)
#else
#define UBR_PSTAMP(acc_) do { } while (0)
#define UBR_PWAIT(n_) do { } while (0)
#endif
    if (niter > 0) issue(hvA, wvA, hokA);
    if (niter > 1) issue(hvB, wvB, hokB);
    UBR_PSTAMP(tL);
    if (niter > 0) stage(hvA, wvA, hokA, 0);
    UBR_PSTAMP(tS);
    if (niter > 2) issue(hvA, wvA, hokA);
    UBR_PSTAMP(tL);
    __syncthreads();                                   // buffer 0 holds block 0
    UBR_PSTAMP(tB);
    for (int i = 0; i < niter; i += 2) {
      if (i + 1 < niter) {                             // consumers are in block i (buffer 0): block i+1 goes to buffer 1
        UBR_PWAIT(HS + WS);
        stage(hvB, wvB, hokB, 1);
        UBR_PSTAMP(tS);
        if (i + 3 < niter) issue(hvB, wvB, hokB);
        UBR_PSTAMP(tL);
      }
      __syncthreads();
      UBR_PSTAMP(tB);
      if (i + 1 >= niter) break;
      if (i + 2 < niter) {                             // consumers are in block i+1 (buffer 1): block i+2 goes to buffer 0
        UBR_PWAIT(HS + WS);
        stage(hvA, wvA, hokA, 0);
        UBR_PSTAMP(tS);
        if (i + 4 < niter) issue(hvA, wvA, hokA);
        UBR_PSTAMP(tL);
      }
      __syncthreads();
      UBR_PSTAMP(tB);
    }
#ifdef UBR_CONV_STAMPS
    if (k.stamps != nullptr && ptid == 0) {
      unsigned long long* o = k.stamps + (size_t)blockIdx.x * 16;
      o[8] = tS; o[9] = tL; o[10] = tB; o[11] = tW; o[12] = t_ - t_begin;
    }
#endif
  } else {
    // ------------------------------------------------ consumers ------------------------------------------------
    f32x4 acc[FW][NT];
    int fragbase[FW];
#pragma unroll
    for (int i = 0; i < FW; ++i) {
      const int f = wave * FW + i;
      const int fr = f / TWF, fc = f % TWF;
      fragbase[i] = (fr * k.HW + fc * 16) * k.pixb + l16 * k.pixb;
    }
    int cu = (int)blockIdx.x, cblk = 0;
    int stat_n0 = -1;
    // a wave's per-channel sums: after the quad-row reduction lane (q, l16 == 0) holds channels 4q..4q+3 of each cout fragment;
    // they are transposed through a wave-private LDS row (no barrier: a wave sees its own LDS writes after lgkmcnt(0)) so that
    // lane c owns channel c and the wave issues TWO fp64 atomic instructions (64 of them, one lane each, cost ~8 k cycles per flush)
    float* srow = reinterpret_cast<float*>(smem + k.red_off) + wave * 2 * TN;
    auto flush_stats = [&](int n0) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float a = wave_quadrow_sum16(s1[j][r]);
          const float b = wave_quadrow_sum16(s2[j][r]);
          if (l16 == 0) { srow[j * 16 + 4 * q + r] = a; srow[TN + j * 16 + 4 * q + r] = b; }
          s1[j][r] = 0.f; s2[j][r] = 0.f;
        }
      __builtin_amdgcn_s_waitcnt(0xc07f);              // lgkmcnt(0): this wave's LDS writes have landed
      __builtin_amdgcn_wave_barrier();
      const float a = srow[lane], b = srow[TN + lane];
      const int ch = n0 + lane;
      if (ch < k.Cout) {
        double* st = k.stats + (size_t)((blockIdx.x * 4 + wave) % k.nslots) * 2 * k.Cout;
        atomicAdd(&st[ch], (double)a);
        atomicAdd(&st[k.Cout + ch], (double)b);
      }
    };
    // this lane's tap offsets of every K-step (quad q reads cin unit q of the step's tap)
    int offq[STEPS > 0 ? STEPS : 1];
    if constexpr (STEPS > 0) {
#pragma unroll
      for (int s_ = 0; s_ < STEPS; ++s_) offq[s_] = tbl[4 * s_ + q];
    }
    __builtin_amdgcn_s_setprio(2);                     // the matrix pipe's wave wins issue arbitration against the producer on its SIMD
#ifdef UBR_CONV_STAMPS
    unsigned long long tC = 0, tE = 0, tB = 0, t_ = __builtin_amdgcn_s_memtime();
    const unsigned long long t_begin = t_;
#endif
    __syncthreads();                                   // buffer 0 holds block 0
    UBR_PSTAMP(tB);
    for (int i = 0; i < niter; ++i) {
      const char* wl = smem + k.wl_off + (i & 1) * k.buf_stride;
      const char* halo = smem + k.halo_off + (i & 1) * k.buf_stride;
      if (cblk == 0) {
#pragma unroll
        for (int a = 0; a < FW; ++a)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[a][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      // ---- MFMA over (tap, cin unit); the next step's fragments (weights and pixels) are read one step ahead ----
      if constexpr (STEPS > 0) {
        uint4 wfc[NT], ac[FW], wfn[NT], an[FW];
#pragma unroll
        for (int j = 0; j < NT; ++j) wfc[j] = *reinterpret_cast<const uint4*>(wl + ((q * TN) + j * 16 + l16) * 16);
#pragma unroll
        for (int i2 = 0; i2 < FW; ++i2) ac[i2] = *reinterpret_cast<const uint4*>(halo + fragbase[i2] + offq[0]);
#pragma unroll
        for (int s_ = 0; s_ < STEPS; ++s_) {
          if (s_ + 1 < STEPS) {
#pragma unroll
            for (int j = 0; j < NT; ++j) wfn[j] = *reinterpret_cast<const uint4*>(wl + (((4 * (s_ + 1) + q) * TN) + j * 16 + l16) * 16);
#pragma unroll
            for (int i2 = 0; i2 < FW; ++i2) an[i2] = *reinterpret_cast<const uint4*>(halo + fragbase[i2] + offq[s_ + 1]);
          }
          __builtin_amdgcn_sched_barrier(0);          // keep the next step's reads ahead of this step's MFMAs (the scheduler would sink them to their uses)
#pragma unroll
          for (int i2 = 0; i2 < FW; ++i2)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i2][j] = mma_step<T>(acc[i2][j], wfc[j], ac[i2]);
          __builtin_amdgcn_sched_barrier(0);
          if (s_ + 1 < STEPS) {
#pragma unroll
            for (int j = 0; j < NT; ++j) wfc[j] = wfn[j];
#pragma unroll
            for (int i2 = 0; i2 < FW; ++i2) ac[i2] = an[i2];
          }
        }
      } else {
      int off_n = tbl[q];
      uint4 wf_n[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) wf_n[j] = *reinterpret_cast<const uint4*>(wl + ((q * TN) + j * 16 + l16) * 16);
      for (int s = 0; s < k.steps; ++s) {
        const int off = off_n;
        uint4 wf[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) wf[j] = wf_n[j];
        if (s + 1 < k.steps) {
          off_n = tbl[4 * (s + 1) + q];
#pragma unroll
          for (int j = 0; j < NT; ++j) wf_n[j] = *reinterpret_cast<const uint4*>(wl + (((4 * (s + 1) + q) * TN) + j * 16 + l16) * 16);
        }
        uint4 a[FW];
#pragma unroll
        for (int i2 = 0; i2 < FW; ++i2) a[i2] = *reinterpret_cast<const uint4*>(halo + fragbase[i2] + off);
#pragma unroll
        for (int i2 = 0; i2 < FW; ++i2)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i2][j] = mma_step<T>(acc[i2][j], wf[j], a[i2]);
      }
      }
      UBR_PSTAMP(tC);
      if (++cblk == k.nblk) {
        // ---------------------------------- epilogue of this unit ----------------------------------
        cblk = 0;
        const int u_ = __builtin_amdgcn_readfirstlane(cu);
        cu += G;
        const int ct = u_ / ntiles;
        int t = u_ - ct * ntiles;
        const int tx = t % k.tiles_x; t /= k.tiles_x;
        const int ty = t % k.tiles_y;
        const int n = t / k.tiles_y;
        const int n0 = ct * TN, oy0 = ty * TH, ox0 = tx * TW;
        if (k.stats != nullptr && stat_n0 != n0) {
          if (stat_n0 >= 0) flush_stats(stat_n0);
          stat_n0 = n0;
        }
        float bs[NT][4];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int ch = n0 + j * 16 + 4 * q + r;
            bs[j][r] = (k.bias != nullptr && ch < k.Cout) ? k.bias[ch] : 0.f;
          }
        if (k.fast_epi) {
          // buffer-addressed form (see conv_igemm_kernel's epilogue): one per-lane offset, scalar fragment offsets, addend loads up front
          constexpr int kOut = (int)0x80000000;
          const int wv = __builtin_amdgcn_readfirstlane(wave);
          const bool full = (ox0 + TW <= k.OW) && (oy0 + TH <= k.OH) && (n0 + TN <= k.Cout);
          const int ysx = (int)k.y_sx, ysy = (int)k.y_sy, asx = (int)k.a_sx, asy = (int)k.a_sy;
          const bool has_ad = k.ad != nullptr;
          const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(
              (void*)(k.y + (long)n * k.y_sn + (long)oy0 * k.y_sy + (long)ox0 * k.y_sx + (long)n0 * ESZ), 0, 0x7fffffff, 0x00020000);
          const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(
              (void*)(has_ad ? k.ad + (long)n * k.a_sn + (long)oy0 * k.a_sy + (long)ox0 * k.a_sx + (long)n0 * ESZ : k.x), 0, has_ad ? 0x7fffffff : 0, 0x00020000);
          const int lane_y = l16 * ysx + 4 * q * ESZ, lane_a = l16 * asx + 4 * q * ESZ;
          int so_y[FW], so_a[FW];
          bool okp[FW], okc[NT];
#pragma unroll
          for (int i2 = 0; i2 < FW; ++i2) {
            const int f = wv * FW + i2, fr = f / TWF, fc = f % TWF;
            so_y[i2] = fr * ysy + fc * 16 * ysx; so_a[i2] = fr * asy + fc * 16 * asx;
            okp[i2] = full || ((oy0 + fr < k.OH) && (ox0 + fc * 16 + l16 < k.OW));
          }
#pragma unroll
          for (int j = 0; j < NT; ++j) okc[j] = full || (n0 + j * 16 + 4 * q < k.Cout);
          ubr_u4 adv[FW][NT];
          if (has_ad) {
#pragma unroll
            for (int i2 = 0; i2 < FW; ++i2)
#pragma unroll
              for (int j = 0; j < NT; ++j) {
                const int vo = (okp[i2] && okc[j]) ? lane_a : kOut;
                if constexpr (sizeof(T) == 2) {
                  const auto t2 = __builtin_amdgcn_raw_buffer_load_b64(ar, vo, so_a[i2] + j * 16 * ESZ, 0);
                  adv[i2][j] = ubr_u4{t2[0], t2[1], 0u, 0u};
                } else {
                  adv[i2][j] = __builtin_amdgcn_raw_buffer_load_b128(ar, vo, so_a[i2] + j * 16 * ESZ, 0);
                }
              }
          }
#pragma unroll
          for (int i2 = 0; i2 < FW; ++i2) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
              float v[4];
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = acc[i2][j][r] + bs[j][r];
              if (k.act & 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
              }
              if (has_ad && okp[i2] && okc[j]) {
                float a4[4];
                if constexpr (sizeof(T) == 2) {
                  const uint2 raw = make_uint2(adv[i2][j][0], adv[i2][j][1]);
                  load4<T>(reinterpret_cast<const char*>(&raw), a4);
                } else {
                  a4[0] = __uint_as_float(adv[i2][j][0]); a4[1] = __uint_as_float(adv[i2][j][1]); a4[2] = __uint_as_float(adv[i2][j][2]); a4[3] = __uint_as_float(adv[i2][j][3]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += a4[r];
              }
              if (k.act & 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
              }
              if (k.stats != nullptr && okp[i2]) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { s1[j][r] += v[r]; s2[j][r] += v[r] * v[r]; }
              }
              const int vo = (okp[i2] && okc[j]) ? lane_y : kOut;
              if constexpr (sizeof(T) == 2) {
                uint2 pk;
                store4<T>(reinterpret_cast<char*>(&pk), v);
                __builtin_amdgcn_raw_buffer_store_b64(ubr_u2{pk.x, pk.y}, yr, vo, so_y[i2] + j * 16 * ESZ, 0);
              } else {
                buf_store16(ubr_u4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, yr, vo, so_y[i2] + j * 16 * ESZ);
              }
            }
          }
        } else
#pragma unroll
        for (int i2 = 0; i2 < FW; ++i2) {
          const int f = wave * FW + i2;
          const int oy = oy0 + f / TWF, ox = ox0 + (f % TWF) * 16 + l16;
          const bool valid = (oy < k.OH) && (ox < k.OW);
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const int ch = n0 + j * 16 + 4 * q;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i2][j][r] + bs[j][r];
            if (k.act & 1) {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            if (k.ad != nullptr && valid && ch < k.Cout) {
              float a4[4];
              load4<T>(k.ad + (long)n * k.a_sn + (long)oy * k.a_sy + (long)ox * k.a_sx + (long)ch * ESZ, a4);
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] += a4[r];
            }
            if (k.act & 2) {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            if (k.stats != nullptr && valid) {
#pragma unroll
              for (int r = 0; r < 4; ++r) { s1[j][r] += v[r]; s2[j][r] += v[r] * v[r]; }
            }
            if (valid && ch < k.Cout)
              store4<T>(k.y + (long)n * k.y_sn + (long)oy * k.y_sy + (long)ox * k.y_sx + (long)ch * ESZ, v);
          }
        }
      }
      UBR_PSTAMP(tE);
      __syncthreads();
      UBR_PSTAMP(tB);
    }
    if (k.stats != nullptr && stat_n0 >= 0) flush_stats(stat_n0);
#ifdef UBR_CONV_STAMPS
    UBR_PSTAMP(tE);
    if (k.stamps != nullptr && tid == 0) {
      unsigned long long* o = k.stamps + (size_t)blockIdx.x * 16;
      o[0] = tC; o[1] = tE; o[2] = tB; o[3] = t_ - t_begin; o[4] = (unsigned long long)niter;
    }
#endif
  }
}

static thread_local int g_last_conv_cfg[4] = {0, 0, 0, 0};   // FW, NT, TWF, kind (0 generic, 1 cin-block pipeline, 2 thin, 3 wide) of this thread's last launch
static thread_local char g_last_conv_name[160] = "";         // kernel symbol of that launch, as rocprofv3 prints it
struct TileCfg { int FW, NT, TWF; };
// id -> config; keep in sync with the dispatch switch
static const TileCfg kCfgs[] = {
    {8, 1, 2},  // 0: 16x32 px, 16 ch   (thin, full resolution)
    {4, 1, 2},  // 1:  8x32 px, 16 ch   (7x7 head: smaller halo)
    {4, 2, 2},  // 2:  8x32 px, 32 ch
    {4, 4, 2},  // 3:  8x32 px, 64 ch
    {2, 4, 1},  // 4:  8x16 px, 64 ch
    {1, 4, 1},  // 5:  4x16 px, 64 ch
    {2, 2, 1},  // 6:  8x16 px, 32 ch
    {2, 1, 1},  // 7:  8x16 px, 16 ch
    {1, 2, 1},  // 8:  4x16 px, 32 ch
    {1, 1, 1},  // 9:  4x16 px, 16 ch
};
constexpr int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);

template <typename T, int FW, int NT, int TWF, bool PIPE>
int launch_one(const ConvK& k, dim3 grid, size_t lds, hipStream_t st) {
  auto fn = conv_igemm_kernel<T, FW, NT, TWF, PIPE>;
  if (lds > 64 * 1024) {
    static thread_local size_t maxset = 0;  // per instantiation
    if (lds > maxset) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) { ubr_set_error("ubr_conv: cannot raise LDS limit to %zu: %s", lds, hipGetErrorString(e)); return UBR_ELAUNCH; }
      maxset = lds;
    }
  }
  ubr_launch(fn, grid, dim3(256), lds, st, k);
  UBR_LAUNCH_CHECK("ubr_conv");
  return UBR_OK;
}

template <typename T, int FW, int NT, int TWF, int UPB, bool XF, bool LSM = false, bool ROW7 = false, int EXT = 0>
int launch_thin(const ThinK& k, dim3 grid, size_t lds, hipStream_t st) {
  auto fn = conv_thin_kernel<T, FW, NT, TWF, UPB, XF, LSM, ROW7, EXT>;
  if (lds > 64 * 1024) {
    static thread_local size_t maxset = 0;
    if (lds > maxset) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) { ubr_set_error("ubr_conv: cannot raise LDS limit to %zu: %s", lds, hipGetErrorString(e)); return UBR_ELAUNCH; }
      maxset = lds;
    }
  }
  ubr_launch(fn, grid, dim3(256), lds, st, k);
  UBR_LAUNCH_CHECK("ubr_conv");
  return UBR_OK;
}

static unsigned magic_u32(unsigned d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + d - 1) / d); }

// Thin layers (conv_thin_kernel): one cin block of 2 or 4 units, stride 1, exact tiling, halo within the register slots,
// 16-bit outputs storable as whole channel octets.  Returns 1 when the launch was taken.
template <typename T, int FW, int NT, int TWF>
int try_thin(const ConvK& c, dim3 grid, hipStream_t st, int* rc) {
  constexpr int TH = 4 * FW / TWF, TW = TWF * 16, TN = NT * 16;
  static const bool thin_on = [] { const char* e = getenv("UBR_CONV_THIN"); return !e || atoi(e) != 0; }();
  const int esz = 16 / ET<T>::CPU;
  if (!thin_on || c.nblk != 1 || c.S != 1 || (c.UPB != 2 && c.UPB != 4)) return 0;
  const int nph = (int)grid.z;
  int steps_tot = c.steps, nunits_tot = c.nunits;
  if (nph > 1) {
    // phases share one staged halo and one tap table: whole K-steps per phase, taps of the phases back to back, plain NHWC output
    if (nph > 4 || c.ad != nullptr || c.stats != nullptr || c.epilogue != 0 || c.bc != nullptr || c.ad_mask != nullptr) return 0;
    steps_tot = 0; nunits_tot = 0;
    for (int p = 0; p < nph; ++p) {
      if (c.pnunits[p] % 4 || c.ptap0[p] * c.UPB != nunits_tot || c.pyoff[p] < 0 || c.pyoff[p] >= (1L << 30)) return 0;
      steps_tot += c.psteps[p]; nunits_tot += c.pnunits[p];
    }
    if (nunits_tot != c.ntaps * c.UPB) return 0;
  }
  // ROW7: the full 7x7 tap set over 16 input channels on the 16x32-pixel tile (see conv_thin_kernel)
  int map7[7][7];
  bool row7 = false;
  if constexpr (FW == 8 && NT == 1 && sizeof(T) == 2) {
    static const bool row7_on = [] { const char* e = getenv("UBR_CONV_ROW7"); return !e || atoi(e) != 0; }();
    row7 = row7_on && c.ntaps == 49 && c.UPB == 2 && c.dymin == -3 && c.dxmin == -3 && c.HW == TW + 6 && c.HH == TH + 6;
    if (row7) {
      for (int a = 0; a < 7; ++a) for (int b = 0; b < 7; ++b) map7[a][b] = -1;
      for (int t = 0; t < 49 && row7; ++t) {
        const int a = c.dy[t] + 3, b = c.dx[t] + 3;
        if (a < 0 || a > 6 || b < 0 || b > 6 || map7[a][b] >= 0 || c.wt[t] == 255) row7 = false; else map7[a][b] = c.wt[t];
      }
    }
  }
  const int HSn = row7 ? 7 : (c.UPB == 4 ? 6 : 5);
  if (c.HH * (int)c.rw > 256 * HSn) return 0;
  if (c.OH % TH || c.OW % TW) return 0;
  if (c.epilogue == 0 && esz == 2 && !c.wide_store) return 0;
  if (c.epilogue == 0 && (c.Cout % 4 || c.Cout_pad != (int)grid.y * TN)) return 0;
  if (c.x_sy >= (1L << 24) || c.y_sy >= (1L << 24) || (long)c.H * c.x_sy >= (1L << 31) || (long)c.OH * c.y_sy >= (1L << 31)) return 0;
  if (c.ad != nullptr && (c.a_sy >= (1L << 24) || (long)c.OH * c.a_sy >= (1L << 31))) return 0;
  const int ntiles = c.tiles_x * c.tiles_y * c.N;
  if ((unsigned long long)ntiles * (unsigned)c.tiles_x >= (1ull << 32)) return 0;
  ThinK k{};
  k.x = c.x; k.x_sn = c.x_sn; k.x_sy = (int)c.x_sy; k.x_sx = (int)c.x_sx; k.x_bytes = (unsigned)((long)c.H * c.x_sy);
  k.in_sub = c.in_sub; k.in_scale = c.in_scale; k.in_shift = c.in_shift; k.in_lo = c.in_lo;
  k.w = c.w;
  k.y = c.y; k.y_sn = c.y_sn; k.y_sy = (int)c.y_sy; k.y_sx = (int)c.y_sx; k.y_bytes = (unsigned)((long)c.OH * c.y_sy);
  k.ad = c.ad; k.a_sn = c.a_sn; k.a_sy = (int)c.a_sy; k.a_sx = (int)c.a_sx; k.a_bytes = c.ad ? (unsigned)((long)c.OH * c.a_sy) : 0u;
  k.bias = c.bias; k.stats = c.stats;
  k.H = c.H; k.W = c.W; k.HW = c.HW; k.nitems = c.HH * (int)c.rw; k.hw_magic = magic_u32((unsigned)c.HW);
  k.tiles_x = c.tiles_x; k.tiles_y = c.tiles_y; k.ntiles = ntiles;
  k.tx_magic = magic_u32((unsigned)c.tiles_x); k.ty_magic = magic_u32((unsigned)c.tiles_y);
  k.steps = steps_tot; k.nunits = nunits_tot; k.hy_org = c.iy0 + c.dymin; k.hx_org = c.ix0 + c.dxmin; k.dymin = c.dymin; k.dxmin = c.dxmin;
  k.nphase = nph;
  for (int p = 0, s0 = 0; p < 4; ++p) {
    k.pstep0[p] = s0; k.psteps[p] = p < nph ? c.psteps[p] : 0; k.pyoff[p] = p < nph ? (int)c.pyoff[p] : 0;
    s0 += k.psteps[p];
  }
  k.Cout = c.Cout; k.Cout_pad = c.Cout_pad; k.CU = c.CU; k.OH = c.OH; k.OW = c.OW;
  k.act = c.act; k.epilogue = c.epilogue; k.wlinear = nph > 1 ? 0 : c.wlinear; k.dbg = c.dbg; k.stamps = c.stamps;
  for (int t = 0; t < c.ntaps; ++t) { k.dy[t] = c.dy[t]; k.dx[t] = c.dx[t]; k.wt[t] = c.wt[t]; }
  // training epilogues: BatchNorm-backward sums (EXT 1) or a ReLU bit mask on the addend (EXT 2); never both (ubr_conv checks)
  const int ext = c.bc != nullptr ? 1 : (c.ad_mask != nullptr ? 2 : 0);
  if (ext != 0 && (c.in_scale != nullptr || c.epilogue != 0)) return 0;
  k.nslots = c.nslots;
  if (ext == 1) {
    if (c.bc_sy >= (1L << 24) || (long)c.OH * c.bc_sy >= (1L << 31)) return 0;
    k.bc = c.bc; k.bc_sn = c.bc_sn; k.bc_sy = (int)c.bc_sy; k.bc_sx = (int)c.bc_sx; k.bc_bytes = (unsigned)((long)c.OH * c.bc_sy);
    k.bmean = c.bmean; k.bscale = c.bscale; k.bshift = c.bshift; k.binvstd = c.binvstd;
  }
  if (ext == 2) {
    if ((long)c.OH * c.OW * c.ad_mask_cu >= (1L << 31)) return 0;
    k.ad_mask = c.ad_mask; k.ad_mask_cu = c.ad_mask_cu; k.mask_bytes = (unsigned)((long)c.OH * c.OW * c.ad_mask_cu);
  }
  if (row7) {
    // 56 virtual taps, row-major with the horizontal taps padded to 8: K-step dy*4 + p holds taps (dy, 2p) and (dy, 2p+1)
    for (int a = 0; a < 7; ++a)
      for (int b = 0; b < 8; ++b) {
        const int t = a * 8 + b;
        k.dy[t] = (int8_t)(a - 3); k.dx[t] = (int8_t)((b < 7 ? b : 6) - 3); k.wt[t] = (uint8_t)(b < 7 ? map7[a][b] : 255);
      }
    k.nunits = 56 * 2; k.steps = 28; k.wlinear = 0;
  }
  // LDS: tap-offset + weight-source tables | weight slab | halo (HS slots of the whole workgroup) | statistics scratch
  const int pixb = thin_pixb(c.UPB);
  size_t off = ((size_t)2 * 16 * k.steps + 15) & ~(size_t)15;
  k.wl_off = (int)off; off += (size_t)4 * k.steps * TN * 16;
  // (sized to the halo exactly: slots that reach past it are not written)
  k.halo_off = (int)off; off += ((size_t)c.HH * c.HW * pixb + 15) & ~(size_t)15;
  k.red_off = (int)off; off += (size_t)4 * TN * 2 * sizeof(float) + (size_t)4 * 4 * 8 * sizeof(float);   // + BatchNorm-on-load constants
  const size_t lds = off;
  if (lds > 150 * 1024) return 0;
  const size_t lds_budget = 159 * 1024;
  // persistent grid: the workgroups that fit the chip at once (register- or LDS-limited), each walking tiles with a stride of the grid
  static const int wg_per_cu = [] { const char* e = getenv("UBR_CONV_THIN_WGS"); return e ? atoi(e) : 0; }();
  int per_cu = (int)(lds_budget / lds);
  const int reg_cap = (row7 || (ext == 1 && NT == 2)) ? 2 : (NT == 1 ? (ext == 1 ? 3 : 4) : 3);   // <= 128 VGPRs for the 16-cout tiles, <= 168 for the 32-cout ones, <= 256 for ROW7 and the 32-cout BatchNorm-backward epilogue
  if (per_cu > reg_cap) per_cu = reg_cap;
  if (wg_per_cu > 0) per_cu = wg_per_cu;
  if (per_cu < 1) per_cu = 1;
  dim3 g(grid.x, grid.y);
  const unsigned cap = 256u * (unsigned)per_cu / (grid.y ? grid.y : 1);
  if (g.x > cap && cap > 0) g.x = cap;
  g_last_conv_cfg[3] = 2;
  const bool xf = c.in_scale != nullptr;
  // (the full template argument list, as rocprofv3 prints the symbol)
  snprintf(g_last_conv_name, sizeof(g_last_conv_name), "conv_thin_kernel<%s, %d, %d, %d, %d, %s, %s, false, 0>", sizeof(T) == 4 ? "float" : (std::is_same<T, bf16_t>::value ? "bf16_t" : "f16_t"),
           FW, NT, TWF, c.UPB, xf ? "true" : "false", c.epilogue == 1 ? "true" : "false");
  if (ext != 0)
    snprintf(g_last_conv_name, sizeof(g_last_conv_name), "conv_thin_kernel<%s, %d, %d, %d, %d, false, false, %s, %d>", sizeof(T) == 4 ? "float" : (std::is_same<T, bf16_t>::value ? "bf16_t" : "f16_t"),
             FW, NT, TWF, c.UPB, row7 ? "true" : "false", ext);
  if constexpr (FW == 8 && NT == 1 && sizeof(T) == 2) {
    if (row7) {
      if (ext == 0)
      snprintf(g_last_conv_name, sizeof(g_last_conv_name), "conv_thin_kernel<%s, 8, 1, 2, 2, %s, %s, true, 0>", std::is_same<T, bf16_t>::value ? "bf16_t" : "f16_t",
               xf ? "true" : "false", c.epilogue == 1 ? "true" : "false");
      if (c.epilogue == 1) *rc = xf ? launch_thin<T, 8, 1, 2, 2, true, true, true>(k, g, lds, st) : launch_thin<T, 8, 1, 2, 2, false, true, true>(k, g, lds, st);
      else if (ext == 1) *rc = launch_thin<T, 8, 1, 2, 2, false, false, true, 1>(k, g, lds, st);
      else if (ext == 2) *rc = launch_thin<T, 8, 1, 2, 2, false, false, true, 2>(k, g, lds, st);
      else *rc = xf ? launch_thin<T, 8, 1, 2, 2, true, false, true>(k, g, lds, st) : launch_thin<T, 8, 1, 2, 2, false, false, true>(k, g, lds, st);
      return 1;
    }
  }
  if (c.epilogue == 1) {
    // conv11 + LogSoftmax (models/ub_uresnet.py:64,143): 7x7 over 16 channels, the 8x32-pixel tile
    if constexpr (FW == 4 && NT == 1) {
      if (c.UPB != 2) return 0;
      *rc = xf ? launch_thin<T, FW, NT, TWF, 2, true, true>(k, g, lds, st) : launch_thin<T, FW, NT, TWF, 2, false, true>(k, g, lds, st);
      return 1;
    } else {
      return 0;
    }
  }
  if (ext == 1) *rc = c.UPB == 2 ? launch_thin<T, FW, NT, TWF, 2, false, false, false, 1>(k, g, lds, st) : launch_thin<T, FW, NT, TWF, 4, false, false, false, 1>(k, g, lds, st);
  else if (ext == 2) *rc = c.UPB == 2 ? launch_thin<T, FW, NT, TWF, 2, false, false, false, 2>(k, g, lds, st) : launch_thin<T, FW, NT, TWF, 4, false, false, false, 2>(k, g, lds, st);
  else if (c.UPB == 2) *rc = xf ? launch_thin<T, FW, NT, TWF, 2, true>(k, g, lds, st) : launch_thin<T, FW, NT, TWF, 2, false>(k, g, lds, st);
  else *rc = xf ? launch_thin<T, FW, NT, TWF, 4, true>(k, g, lds, st) : launch_thin<T, FW, NT, TWF, 4, false>(k, g, lds, st);
  return 1;
}

// the cin-block pipeline exists for the 64-cout tiles, when a block's items fit its register slots
template <typename T, int FW, int NT, int TWF>
int launch_cfg(const ConvK& k, dim3 grid, size_t lds, hipStream_t st) {
  if constexpr (NT <= 2 && TWF == 2) {
    int rc = UBR_OK;
    if (try_thin<T, FW, NT, TWF>(k, grid, st, &rc)) return rc;
  }
  if constexpr (NT == 4) {
    static const bool enabled = [] { const char* e = getenv("UBR_CONV_PIPE"); return !e || atoi(e) != 0; }();
    if (enabled && k.nblk >= 2 && k.HH * (int)k.rw <= 256 * conv_pipe_hslots(FW, TWF) && 4 * k.steps * NT * 16 <= 256 * conv_pipe_wslots(NT))
      { g_last_conv_cfg[3] = 1; return launch_one<T, FW, NT, TWF, true>(k, grid, lds, st); }
  }
  g_last_conv_cfg[3] = 0;
  return launch_one<T, FW, NT, TWF, false>(k, grid, lds, st);
}

template <typename T, int FW, int TWF, int STEPS>
int launch_pc_steps(const ConvK& k, dim3 grid, size_t lds, hipStream_t st) {
  auto fn = conv_pc_kernel<T, FW, TWF, STEPS>;
  if (lds > 64 * 1024) {
    static thread_local size_t maxset = 0;
    if (lds > maxset) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) { ubr_set_error("ubr_conv: cannot raise LDS limit to %zu: %s", lds, hipGetErrorString(e)); return UBR_ELAUNCH; }
      maxset = lds;
    }
  }
  ubr_launch(fn, grid, dim3(512), lds, st, k);
  UBR_LAUNCH_CHECK("ubr_conv");
  return UBR_OK;
}

template <typename T, int FW, int TWF>
int launch_pc_one(const ConvK& k, dim3 grid, size_t lds, hipStream_t st) {
  // 3x3 and the 2x2 phases of the transposed convs get a fully unrolled K-step loop; other tap counts the generic one
  if (k.steps == 9) return launch_pc_steps<T, FW, TWF, 9>(k, grid, lds, st);
  if (k.steps == 4) return launch_pc_steps<T, FW, TWF, 4>(k, grid, lds, st);
  return launch_pc_steps<T, FW, TWF, 0>(k, grid, lds, st);
}

template <typename T>
int launch_pc(int pi, const ConvK& k, dim3 grid, size_t lds, hipStream_t st) {
  g_last_conv_cfg[3] = 3;
  switch (pi) {
    case 0: return launch_pc_one<T, 4, 2>(k, grid, lds, st);
    case 1: return launch_pc_one<T, 2, 2>(k, grid, lds, st);
    case 2: return launch_pc_one<T, 4, 1>(k, grid, lds, st);
    case 3: return launch_pc_one<T, 2, 1>(k, grid, lds, st);
  }
  ubr_set_error("ubr_conv: bad producer/consumer tile %d", pi);
  return UBR_EINVAL;
}

template <typename T>
int launch_T(int cfg, const ConvK& k, dim3 grid, size_t lds, hipStream_t st) {
  if (cfg >= 100) return launch_pc<T>(cfg - 100, k, grid, lds, st);
  switch (cfg) {
    case 0: return launch_cfg<T, 8, 1, 2>(k, grid, lds, st);
    case 1: return launch_cfg<T, 4, 1, 2>(k, grid, lds, st);
    case 2: return launch_cfg<T, 4, 2, 2>(k, grid, lds, st);
    case 3: return launch_cfg<T, 4, 4, 2>(k, grid, lds, st);
    case 4: return launch_cfg<T, 2, 4, 1>(k, grid, lds, st);
    case 5: return launch_cfg<T, 1, 4, 1>(k, grid, lds, st);
    case 6: return launch_cfg<T, 2, 2, 1>(k, grid, lds, st);
    case 7: return launch_cfg<T, 2, 1, 1>(k, grid, lds, st);
    case 8: return launch_cfg<T, 1, 2, 1>(k, grid, lds, st);
    case 9: return launch_cfg<T, 1, 1, 1>(k, grid, lds, st);
  }
  ubr_set_error("ubr_conv: bad tile config %d", cfg);
  return UBR_EINVAL;
}

struct Plan { int cfg; size_t lds; int UPB, steps, HH, HW, pixb, wl_off, halo_off, red_off, xfc_off, tiles_x, tiles_y, pc_buf; };

// taps a workgroup walks: all of them, or the largest phase of a phased launch
static int eff_ntaps(const ubr_conv_desc* d) {
  if (d->nphase <= 1) return d->ntaps;
  int m = 0;
  for (int p = 0; p < d->nphase && p < 4; ++p) m = d->phase_ntaps[p] > m ? d->phase_ntaps[p] : m;
  return m;
}

static bool plan_tile(const ubr_conv_desc* d, int cfg, int TH, int TW, int TN, int dymin, int dymax, int dxmin, int dxmax, Plan* p) {
  if (d->Cout_pad % TN) return false;
  if (d->epilogue == 1 && TN != 16) return false;
  const int cpu = ubr_cpu(d->dtype);
  const int CU = d->Cin / cpu;
  int UPB = (CU % 4 == 0) ? 4 : (CU % 2 == 0 ? 2 : 1);
  // Stride-2 convs (halo of 4x the output pixels) and 16-tap layers (the data gradients of the transposed convs: a
  // 65 KB weight slab per 4-unit cin block) overflow 80 KB of LDS with 4-unit blocks and run one workgroup per CU;
  // 2-unit blocks halve both images.  The rule depends on the LAYER only (never on the tile or the batch), so an
  // image's result does not depend on what it is batched with.
  if (UPB == 4 && (d->S == 2 || eff_ntaps(d) >= 16)) UPB = 2;
  const int nunits = eff_ntaps(d) * UPB;
  const int steps = (nunits + 3) / 4;
  const int HH = (TH - 1) * d->S + 1 + (dymax - dymin);
  const int HW = (TW - 1) * d->S + 1 + (dxmax - dxmin);
  const int pixb = conv_pixb(UPB);
  size_t off = ((size_t)2 * 16 * steps + 15) & ~(size_t)15;   // offset table + weight source table
  p->wl_off = (int)off; off += (size_t)4 * steps * TN * 16;
  p->halo_off = (int)off; off += (size_t)HH * HW * pixb;
  off = (off + 15) & ~(size_t)15;
  p->red_off = (int)off; off += (size_t)4 * TN * 2 * sizeof(float);
  p->xfc_off = (int)off;
  if (d->xf.scale != nullptr && CU / UPB >= 2) off += (size_t)d->Cin * 4 * sizeof(float);   // cin-block pipeline: the layer's BatchNorm constants
  p->cfg = cfg; p->lds = off; p->UPB = UPB; p->steps = steps; p->HH = HH; p->HW = HW; p->pixb = pixb;
  p->tiles_x = ubr_cdiv(d->OW, TW); p->tiles_y = ubr_cdiv(d->OH, TH);
  if ((size_t)HH * HW * UPB >= 60000) return false;   // exact-division bound of the magic multiply
  return off <= 160 * 1024;
}

static bool plan_for(const ubr_conv_desc* d, int cfg, int dymin, int dymax, int dxmin, int dxmax, Plan* p) {
  const TileCfg& c = kCfgs[cfg];
  return plan_tile(d, cfg, 4 * c.FW / c.TWF, c.TWF * 16, c.NT * 16, dymin, dymax, dxmin, dxmax, p);
}

// conv_pc_kernel tiles: {FW, TWF} (always 64 output channels); id = 100 + index
struct PcCfg { int FW, TWF; };
static const PcCfg kPc[] = {
    {4, 2},   // 100:  8x32 px
    {2, 2},   // 101:  4x32 px
    {4, 1},   // 102: 16x16 px
    {2, 1},   // 103:  8x16 px
};
constexpr int kNumPc = sizeof(kPc) / sizeof(kPc[0]);

static bool plan_pc(const ubr_conv_desc* d, int pi, int dymin, int dymax, int dxmin, int dxmax, Plan* p) {
  const PcCfg& c = kPc[pi];
  const int TH = 4 * c.FW / c.TWF, TW = c.TWF * 16, TN = 64;
  const int cpu = ubr_cpu(d->dtype);
  if (d->S != 1 || d->epilogue != 0 || d->ntaps > 9 || d->Cout_pad % TN || d->Cin % (4 * cpu)) return false;
  if (c.TWF == 2 && d->OW < 32) return false;
  const int steps = d->ntaps;                                   // 4-unit cin blocks: one K-step per tap
  const int HH = TH + (dymax - dymin), HW = TW + (dxmax - dxmin);
  // 96 bytes per halo pixel: conflict-free against the ds_read_b128 lane groups (conv_pixb: the 80-byte stride of the other kernels
  // is a 2-way conflict on every pixel-fragment read, and in this kernel the LDS array, shared by four reading and four writing
  // waves, is the busiest unit)
  static const int pixb = [] { const char* e = getenv("UBR_PC_PIXB"); return e ? atoi(e) : 96; }();
  if ((long)HH * HW * 4 > 256L * conv_pipe_hslots(c.FW, c.TWF)) return false;      // register slots of the producers
  if (4L * steps * TN > 256L * conv_pipe_wslots(4)) return false;
  size_t off = ((size_t)2 * 16 * steps + 15) & ~(size_t)15;     // tap-offset table + weight-source table
  p->wl_off = (int)off;
  const size_t wlb = (size_t)4 * steps * TN * 16;
  p->halo_off = (int)(off + wlb);
  const size_t buf = (wlb + (size_t)HH * HW * pixb + 15) & ~(size_t)15;
  off += 2 * buf;
  p->pc_buf = (int)buf;
  p->red_off = (int)off; off += (size_t)4 * 2 * TN * sizeof(float);        // one statistics row per consumer wave
  p->xfc_off = (int)off;
  if (d->xf.scale != nullptr) off += (size_t)d->Cin * 4 * sizeof(float);
  p->cfg = 100 + pi; p->lds = off; p->UPB = 4; p->steps = steps; p->HH = HH; p->HW = HW; p->pixb = pixb;
  p->tiles_x = ubr_cdiv(d->OW, TW); p->tiles_y = ubr_cdiv(d->OH, TH);
  if ((size_t)HH * HW * 4 >= 60000) return false;
  return off <= 160 * 1024;
}

}  // namespace

extern "C" int ubr_conv_last_kernel(char* buf, int n) {
  if (buf == nullptr || n <= 0) return UBR_EINVAL;
  snprintf(buf, (size_t)n, "%s", g_last_conv_name);
  return UBR_OK;
}

extern "C" void ubr_conv_last_config(int* fw, int* nt, int* twf, int* pipe) {
  if (fw) *fw = g_last_conv_cfg[0];
  if (nt) *nt = g_last_conv_cfg[1];
  if (twf) *twf = g_last_conv_cfg[2];
  if (pipe) *pipe = g_last_conv_cfg[3];
}

// host mirror of try_thin's conditions (a layer property plus exact tiling; never the batch size)
static bool thin_eligible(const ubr_conv_desc* d, int cfg, const Plan& p, bool wide_ok) {
  static const bool thin_on = [] { const char* e = getenv("UBR_CONV_THIN"); return !e || atoi(e) != 0; }();
  const TileCfg& c = kCfgs[cfg];
  if (!thin_on || c.TWF != 2 || c.NT > 2 || (c.FW % 4)) return false;
  const int cpu = ubr_cpu(d->dtype), esz = ubr_esize(d->dtype);
  const int CU = d->Cin / cpu, TH = 4 * c.FW / c.TWF, TW = c.TWF * 16;
  if (p.UPB != CU || d->S != 1 || (p.UPB != 2 && p.UPB != 4)) return false;
  if (d->OH % TH || d->OW % TW) return false;
  if (d->nphase > 1) {      // (try_thin's conditions for a phased launch)
    static const bool ph_on = [] { const char* e = getenv("UBR_THIN_PHASES"); return !e || atoi(e) != 0; }();
    if (!ph_on || d->addend.p != nullptr || d->stats != nullptr || d->epilogue != 0) return false;
    int t0 = 0;
    for (int q = 0; q < d->nphase; ++q) {
      if ((d->phase_ntaps[q] * p.UPB) % 4 || d->phase_tap0[q] != t0) return false;
      t0 += d->phase_ntaps[q];
    }
    if (t0 != d->ntaps) return false;
  }
  // the row-stationary 7x7 form (ROW7) of the 16x32-pixel tile: try_thin re-checks the tap set
  static const bool row7_on = [] { const char* e = getenv("UBR_CONV_ROW7"); return !e || atoi(e) != 0; }();
  const bool row7 = row7_on && cfg == 0 && esz == 2 && d->ntaps == 49 && p.UPB == 2 && p.HW == TW + 6 && p.HH == TH + 6;
  if ((long)p.HH * p.HW * p.UPB > 256L * (row7 ? 7 : (p.UPB == 4 ? 6 : 5))) return false;
  if (d->epilogue == 1) return (cfg == 1 || row7) && p.UPB == 2;
  if (esz == 2 && !wide_ok) return false;
  return d->Cout % 4 == 0;
}

extern "C" int ubr_conv(const ubr_conv_desc* d, void* stream) {
  UBR_CHECK(d != nullptr, "ubr_conv: null descriptor");
  UBR_CHECK(ubr_dtype_ok(d->dtype), "ubr_conv: bad dtype %d", d->dtype);
  const int cpu = ubr_cpu(d->dtype), esz = ubr_esize(d->dtype);
  UBR_CHECK(d->N > 0 && d->H > 0 && d->W > 0 && d->OH > 0 && d->OW > 0, "ubr_conv: empty extent");
  UBR_CHECK(d->Cin > 0 && d->Cin % cpu == 0, "ubr_conv: Cin=%d must be a positive multiple of %d", d->Cin, cpu);
  UBR_CHECK(d->Cout > 0 && d->Cout_pad % 16 == 0 && d->Cout <= d->Cout_pad && d->Cout_pad - d->Cout < 16,
            "ubr_conv: bad Cout=%d Cout_pad=%d", d->Cout, d->Cout_pad);
  UBR_CHECK(d->ntaps >= 1 && d->ntaps <= UBR_MAX_TAPS, "ubr_conv: ntaps=%d out of range", d->ntaps);
  UBR_CHECK(d->S == 1 || d->S == 2, "ubr_conv: S=%d unsupported", d->S);
  UBR_CHECK(d->x.p && d->w && d->y.p, "ubr_conv: null tensor");
  UBR_CHECK(ubr_aligned16(d->x.p) && ubr_aligned16(d->w), "ubr_conv: x/w must be 16-byte aligned");
  UBR_CHECK((d->x.sx * esz) % 16 == 0 && (d->x.sy * esz) % 16 == 0 && (d->x.sn * esz) % 16 == 0,
            "ubr_conv: input strides must keep 16-byte alignment");
  UBR_CHECK(d->x.sx >= d->Cin, "ubr_conv: x pixel stride %ld < Cin %d", (long)d->x.sx, d->Cin);
  const bool xf = d->xf.scale != nullptr;
  UBR_CHECK(xf == (d->xf.shift != nullptr) && xf == (d->xf.lo != nullptr) && xf == (d->xf.sub != nullptr), "ubr_conv: xf needs sub, scale, shift and lo together");
  if (d->epilogue == 0) {
    UBR_CHECK(d->Cout % 4 == 0, "ubr_conv: NHWC store needs Cout %% 4 == 0 (got %d)", d->Cout);
    UBR_CHECK((((uintptr_t)d->y.p) % (4 * esz)) == 0 && d->y.sx % 4 == 0 && d->y.sy % 4 == 0 && d->y.sn % 4 == 0,
              "ubr_conv: output view must be aligned to 4 elements");
    UBR_CHECK(d->y.sx >= d->Cout, "ubr_conv: y pixel stride %ld < Cout %d", (long)d->y.sx, d->Cout);
    if (d->addend.p)
      UBR_CHECK((((uintptr_t)d->addend.p) % (4 * esz)) == 0 && d->addend.sx % 4 == 0 && d->addend.sy % 4 == 0 && d->addend.sn % 4 == 0,
                "ubr_conv: addend view must be aligned to 4 elements");
  } else {
    UBR_CHECK(d->epilogue == 1 && d->Cout <= 16 && d->Cout_pad == 16, "ubr_conv: log-softmax epilogue needs Cout<=16");
    UBR_CHECK(d->addend.p == nullptr && d->stats == nullptr && d->act == 0, "ubr_conv: log-softmax epilogue takes no addend/stats/act");
  }
  UBR_CHECK((d->act & ~3) == 0, "ubr_conv: bad act flags %d", d->act);
  if (d->addend_mask != nullptr)
    UBR_CHECK(d->addend.p != nullptr && d->epilogue == 0 && d->Cout % cpu == 0 && d->bnb_c.p == nullptr, "ubr_conv: addend_mask needs an addend, NHWC output, whole channel units, and excludes bnb_c");
  if (d->bnb_c.p != nullptr) {
    UBR_CHECK(d->stats != nullptr && d->epilogue == 0 && d->act == 0 && d->Cout % 4 == 0 && d->bnb_mean && d->bnb_scale && d->bnb_shift && d->bnb_invstd,
              "ubr_conv: bnb_c needs stats, the four BatchNorm vectors, NHWC output and no activation");
    UBR_CHECK((((uintptr_t)d->bnb_c.p) % (4 * esz)) == 0 && d->bnb_c.sx % 4 == 0 && d->bnb_c.sy % 4 == 0 && d->bnb_c.sn % 4 == 0 && d->bnb_c.sx >= d->Cout &&
              ubr_aligned16(d->bnb_mean) && ubr_aligned16(d->bnb_scale) && ubr_aligned16(d->bnb_shift) && ubr_aligned16(d->bnb_invstd),
              "ubr_conv: bnb_c view must be aligned to 4 elements, the BatchNorm vectors to 16 bytes");
  }
  const bool train_epi = d->addend_mask != nullptr || d->bnb_c.p != nullptr;
  const int nphase = d->nphase > 1 ? d->nphase : 1;
  if (nphase > 1) {
    UBR_CHECK(nphase <= 4 && d->epilogue == 0 && !train_epi && d->stats == nullptr, "ubr_conv: a phased launch takes 2..4 phases, NHWC output, no statistics or training epilogue");
    int covered = 0;
    for (int p = 0; p < nphase; ++p) {
      UBR_CHECK(d->phase_ntaps[p] >= 1 && d->phase_tap0[p] == covered, "ubr_conv: phase tap ranges must tile [0, ntaps) in order");
      covered += d->phase_ntaps[p];
      UBR_CHECK(d->phase_yoff[p] % 4 == 0 && d->phase_aoff[p] % 4 == 0, "ubr_conv: phase offsets must keep 4-element alignment");
    }
    UBR_CHECK(covered == d->ntaps, "ubr_conv: phase tap ranges must tile [0, ntaps)");
  }
  int dymin = 127, dymax = -128, dxmin = 127, dxmax = -128;
  for (int t = 0; t < d->ntaps; ++t) {
    dymin = d->dy[t] < dymin ? d->dy[t] : dymin; dymax = d->dy[t] > dymax ? d->dy[t] : dymax;
    dxmin = d->dx[t] < dxmin ? d->dx[t] : dxmin; dxmax = d->dx[t] > dxmax ? d->dx[t] : dxmax;
  }
  UBR_CHECK(dymax - dymin <= 16 && dxmax - dxmin <= 16, "ubr_conv: tap extent too large");

  // 16-byte stores of channel octets need every octet whole and 16-byte aligned in the output view
  const bool wide_ok = d->epilogue == 0 && esz == 2 && d->Cout % 8 == 0 && (((uintptr_t)d->y.p) % 16) == 0 && (d->y.sx * esz) % 16 == 0 &&
                       (d->y.sy * esz) % 16 == 0 && (d->y.sn * esz) % 16 == 0;
  // ---- choose a tile configuration ----
  Plan best{}; bool have = false;
  if (d->tile_hint > 100) {        // 101.. = conv_pc_kernel tiles (tests)
    UBR_CHECK(d->tile_hint <= 100 + kNumPc && !train_epi, "ubr_conv: tile_hint %d out of range (or a training epilogue on conv_pc_kernel)", d->tile_hint);
    have = plan_pc(d, d->tile_hint - 101, dymin, dymax, dxmin, dxmax, &best);
    UBR_CHECK(have, "ubr_conv: producer/consumer tile_hint %d does not fit this shape", d->tile_hint);
  } else if (d->tile_hint > 0) {
    UBR_CHECK(d->tile_hint <= kNumCfgs, "ubr_conv: tile_hint %d out of range", d->tile_hint);
    have = plan_for(d, d->tile_hint - 1, dymin, dymax, dxmin, dxmax, &best);
    UBR_CHECK(have, "ubr_conv: tile_hint %d does not fit this shape", d->tile_hint);
  } else {
    // widest channel tile that divides Cout_pad; then the largest pixel tile that still yields
    // >= 512 workgroups (2 per CU), else the smallest tile.
    const int order_by_nt[3][4] = {{3, 4, 5, -1}, {2, 6, 8, -1}, {0, 1, 7, 9}};
    // conv_pc_kernel (Cout a multiple of 64, Cin a multiple of 4 units, stride 1, 4..9 taps) takes the layers where it beats
    // conv_igemm_kernel INSIDE a train step: at most two cin blocks (a workgroup of the block pipeline has no second block to
    // overlap anything with: 32 -> 64 at 256^2 95 vs 152 us, 64 -> 128 at 128^2 66 vs 89 us), and 16 x 16 maps (512 -> 512: 45 vs
    // 52 us).  On the 4- and 8-block layers at 64^2 / 32^2 it is level alone and SLOWER in the step (45 vs 36 us): its 128-147 KB of
    // LDS do not fit on a CU beside a weight-gradient workgroup of the side stream (79 KB), so it waits for free CUs.
    // The rule depends on the layer, never on the batch; both kernels compute the same bits.
    {
      static const int pc_mode = [] { const char* e = getenv("UBR_CONV_PC"); return e ? atoi(e) : 1; }();   // 0 off, 1 selective, 2 every eligible layer
      const int nblk_pc = d->Cin / (4 * cpu);
      const bool pick = pc_mode == 2 || (pc_mode == 1 && (nblk_pc <= 2 || (long)d->OH * d->OW <= 256) && !(d->Cin == d->Cout_pad && nblk_pc == 2));
      if (pick && !train_epi && nphase == 1 && d->Cout_pad % 64 == 0 && d->ntaps >= 4) {
        Plan cand{}; bool any = false;
        const int wide[] = {0, 1}, narrow[] = {2, 3};
        const int* ord = d->OW >= 32 ? wide : narrow;
        for (int i = 0; i < 2 && !have; ++i) {
          Plan p{};
          if (!plan_pc(d, ord[i], dymin, dymax, dxmin, dxmax, &p)) continue;
          const long units = (long)p.tiles_x * p.tiles_y * d->N * (d->Cout_pad / 64);
          cand = p; any = true;
          if (units >= 256) { best = p; have = true; }
        }
        if (!have && any) { best = cand; have = true; }
      }
    }
    // thin layers (one cin block, <= 32 output channels): the largest tile the persistent conv_thin_kernel can take wins outright
    if (d->Cout_pad <= 32) {
      const int g = d->Cout_pad == 32 ? 1 : 2;
      for (int i = 0; i < 4 && !have; ++i) {
        const int cfg = order_by_nt[g][i];
        if (cfg < 0 || kCfgs[cfg].TWF != 2) continue;
        Plan p{};
        if (!plan_for(d, cfg, dymin, dymax, dxmin, dxmax, &p)) continue;
        if (thin_eligible(d, cfg, p, wide_ok)) { best = p; have = true; }
      }
    }
    for (int g = 0; g < 3 && !have; ++g) {
      Plan cand{}; bool any = false;
      for (int i = 0; i < 4; ++i) {
        const int cfg = order_by_nt[g][i];
        if (cfg < 0) continue;
        Plan p{};
        if (!plan_for(d, cfg, dymin, dymax, dxmin, dxmax, &p)) continue;
        const TileCfg& c = kCfgs[cfg];
        if (c.TWF == 2 && d->OW < 32) continue;
        const long wgs = (long)p.tiles_x * p.tiles_y * d->N * (d->Cout_pad / (c.NT * 16));
        cand = p; any = true;
        if (wgs >= 512 && p.lds <= 80 * 1024) break;
      }
      if (any) { best = cand; have = true; }
    }
    UBR_CHECK(have, "ubr_conv: no tile configuration fits (Cout_pad=%d ntaps=%d)", d->Cout_pad, d->ntaps);
  }

  ConvK k{};
  k.x = (const char*)d->x.p; k.x_sn = d->x.sn * esz; k.x_sy = d->x.sy * esz; k.x_sx = d->x.sx * esz;
  k.in_sub = d->xf.sub; k.in_scale = d->xf.scale; k.in_shift = d->xf.shift; k.in_lo = d->xf.lo;
  k.w = (const char*)d->w;
  k.y = (char*)d->y.p; k.y_sn = d->y.sn * esz; k.y_sy = d->y.sy * esz; k.y_sx = d->y.sx * esz;
  k.ad = (const char*)d->addend.p; k.a_sn = d->addend.sn * esz; k.a_sy = d->addend.sy * esz; k.a_sx = d->addend.sx * esz;
  k.bias = d->bias; k.stats = d->stats;
  k.ad_mask = d->addend_mask; k.ad_mask_cu = d->Cout / cpu;
  k.bc = (const char*)d->bnb_c.p; k.bc_sn = d->bnb_c.sn * esz; k.bc_sy = d->bnb_c.sy * esz; k.bc_sx = d->bnb_c.sx * esz;
  k.bmean = d->bnb_mean; k.bscale = d->bnb_scale; k.bshift = d->bnb_shift; k.binvstd = d->bnb_invstd;
  UBR_CHECK(d->stats_slots == 0 || d->stats_slots == UBR_RED_SLOTS || d->stats_slots == UBR_STAT_SLOTS, "ubr_conv: stats_slots must be 0, %d or %d", UBR_RED_SLOTS, UBR_STAT_SLOTS);
  k.nslots = (d->bnb_c.p != nullptr || d->stats_slots == UBR_RED_SLOTS) ? UBR_RED_SLOTS : UBR_STAT_SLOTS;
  k.H = d->H; k.W = d->W;
  k.CU = d->Cin / cpu; k.UPB = best.UPB; k.lgUPB = ubr_ilog2(best.UPB); k.nblk = k.CU / best.UPB;
  k.Cout = d->Cout; k.Cout_pad = d->Cout_pad;
  k.ntaps = d->ntaps; k.S = d->S; k.iy0 = d->iy0; k.ix0 = d->ix0; k.OH = d->OH; k.OW = d->OW;
  k.dymin = dymin; k.dxmin = dxmin; k.HH = best.HH; k.HW = best.HW;
  k.tiles_x = best.tiles_x; k.tiles_y = best.tiles_y;
  k.tx_magic = magic_u32((unsigned)best.tiles_x); k.ty_magic = magic_u32((unsigned)best.tiles_y);
  k.nunits = eff_ntaps(d) * best.UPB; k.steps = best.steps; k.pixb = best.pixb;
  for (int p = 0; p < 4; ++p) {
    const bool on = p < nphase;
    const int nt = nphase > 1 ? (on ? d->phase_ntaps[p] : 0) : d->ntaps;
    k.ptap0[p] = nphase > 1 && on ? d->phase_tap0[p] : 0;
    k.pnunits[p] = nt * best.UPB;
    k.psteps[p] = (nt * best.UPB + 3) / 4;
    k.pyoff[p] = nphase > 1 && on ? d->phase_yoff[p] * (long)esz : 0;
    k.paoff[p] = nphase > 1 && on ? d->phase_aoff[p] * (long)esz : 0;
  }
  k.rw = (unsigned)(best.HW * best.UPB);
  k.rw_magic = (unsigned)((0x100000000ull + k.rw - 1) / k.rw);
  UBR_CHECK((long)d->H * k.x_sy < (1L << 31) && k.x_sx < (1L << 20), "ubr_conv: image too large for 32-bit offsets");
  k.x_sy32 = (int)k.x_sy; k.x_sx32 = (int)k.x_sx;
  k.step_j = (int)(256 % k.rw); k.step_hy = (int)(256 / k.rw);
  // moving step_j items right = (step_j / UPB) pixels (step_j is a multiple of UPB because rw and 256 are)
  k.step_goff = k.step_hy * k.x_sy32 + (k.step_j >> k.lgUPB) * k.x_sx32;
  k.wrap_goff = k.x_sy32 - best.HW * k.x_sx32;
  k.wl_off = best.wl_off; k.halo_off = best.halo_off; k.red_off = best.red_off; k.xfc_off = best.xfc_off;
  k.epilogue = d->epilogue; k.act = d->act; k.N = d->N;
  {
    bool nat = best.cfg < 100 && d->Cout_pad == kCfgs[best.cfg].NT * 16;
    for (int t = 0; t < d->ntaps && nat; ++t) nat = d->wt[t] == t;
    k.wlinear = nat ? 1 : 0;
  }
#ifdef UBR_CONV_STAMPS      // diagnostic builds only (tools/build_variant.sh): phase masks give wrong results, the stamp pointer is a raw device address
  { static const int dbg = [] { const char* e = getenv("UBR_CONV_DBG"); return e ? atoi(e) : 0; }(); k.dbg = dbg; }
  { static const char* sp = getenv("UBR_CONV_STAMP_PTR"); k.stamps = sp ? (unsigned long long*)strtoull(sp, nullptr, 0) : nullptr; }
#else
  k.dbg = 0; k.stamps = nullptr;
#endif
  k.wide_store = wide_ok ? 1 : 0;
  { static const bool pairs = [] { const char* e = getenv("UBR_CONV_PAIRS"); return e && atoi(e) != 0; }(); k.pair_store = (wide_ok && pairs) ? 1 : 0; }
  {
    static const bool fe = [] { const char* e = getenv("UBR_CONV_FAST_EPI"); return !e || atoi(e) != 0; }();
    // (offsets are relative to the tile origin: at most a tile's rows times the row pitch)
    const long lim = 1L << 26;
    k.fast_epi = (fe && k.y_sy < lim && (k.ad == nullptr || k.a_sy < lim) && (long)k.OW * k.ad_mask_cu < lim) ? 1 : 0;
  }
  for (int t = 0; t < d->ntaps; ++t) { k.dy[t] = d->dy[t]; k.dx[t] = d->dx[t]; k.wt[t] = d->wt[t]; }

  TileCfg c{};
  int tn;
  if (best.cfg >= 100) { const PcCfg& w = kPc[best.cfg - 100]; c = TileCfg{w.FW, 4, w.TWF}; tn = 64; }
  else { c = kCfgs[best.cfg]; tn = c.NT * 16; }
  g_last_conv_cfg[0] = c.FW; g_last_conv_cfg[1] = c.NT; g_last_conv_cfg[2] = c.TWF;
  dim3 grid((unsigned)(best.tiles_x * best.tiles_y * d->N), (unsigned)(d->Cout_pad / tn), (unsigned)nphase);
  if (best.cfg >= 100) {
    // persistent: one workgroup of 8 waves per CU walks the (cout tile, pixel tile) units
    static const int ncu = [] { hipDeviceProp_t pr; int dev = 0; return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256; }();
    k.ncot = d->Cout_pad / 64;
    k.buf_stride = best.pc_buf;
    const long units = (long)grid.x * k.ncot;
    UBR_CHECK(units < (1L << 30), "ubr_conv: too many tiles");
    grid = dim3((unsigned)(units < ncu ? units : ncu), 1);
  }
  hipStream_t st = (hipStream_t)stream;
  {
    static const bool dbg = [] { const char* e = getenv("UBR_CONV_DEBUG"); return e && atoi(e) != 0; }();
    if (dbg)
      fprintf(stderr, "ubr_conv plan: N%d %dx%d Cin%d Cout%d taps%d S%d | cfg%d (%d,%d,%d) UPB%d nblk%d lds %zu wgs %u x %u\n", d->N, d->OH, d->OW,
              d->Cin, d->Cout, d->ntaps, d->S, best.cfg, c.FW, c.NT, c.TWF, best.UPB, k.nblk, best.lds, grid.x, grid.y);
  }
  {
    const char* tn = d->dtype == UBR_F32 ? "float" : (d->dtype == UBR_BF16 ? "bf16_t" : "f16_t");
    g_last_conv_name[0] = 0;
    int rc;
    switch (d->dtype) {
      case UBR_F32: rc = launch_T<float>(best.cfg, k, grid, best.lds, st); break;
      case UBR_BF16: rc = launch_T<bf16_t>(best.cfg, k, grid, best.lds, st); break;
      default: rc = launch_T<f16_t>(best.cfg, k, grid, best.lds, st); break;
    }
    if (g_last_conv_name[0] == 0) {     // (the thin path names itself)
      if (best.cfg >= 100) { const PcCfg& w = kPc[best.cfg - 100]; snprintf(g_last_conv_name, sizeof(g_last_conv_name), "conv_pc_kernel<%s, %d, %d, %d>", tn, w.FW, w.TWF, (best.steps == 9 || best.steps == 4) ? best.steps : 0); }
      else snprintf(g_last_conv_name, sizeof(g_last_conv_name), "conv_igemm_kernel<%s, %d, %d, %d, %s>", tn, c.FW, c.NT, c.TWF, g_last_conv_cfg[3] == 1 ? "true" : "false");
    }
    return rc;
  }
}

// ---------------------------------------------------------------------------------------------
// weight packing
// ---------------------------------------------------------------------------------------------
namespace {
struct PackK {
  const float* src; char* dst;
  int M, Mpad, Kvalid, KU, ntaps;
  long sm, sk;
  int tapidx[UBR_MAX_TAPS];
};
template <typename T>
__global__ __launch_bounds__(256) void pack_kernel(const PackK k) {
  constexpr int CPU = ET<T>::CPU;
  const long total = (long)k.ntaps * k.KU * k.Mpad;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int m = (int)(i % k.Mpad);
    long r = i / k.Mpad;
    const int ku = (int)(r % k.KU);
    const int t = (int)(r / k.KU);
    float f[CPU];
#pragma unroll
    for (int e = 0; e < CPU; ++e) {
      const int kc = ku * CPU + e;
      f[e] = (m < k.M && kc < k.Kvalid) ? k.src[(long)m * k.sm + (long)kc * k.sk + k.tapidx[t]] : 0.f;
    }
    *reinterpret_cast<uint4*>(k.dst + i * 16) = ET<T>::pack(f);
  }
}
}  // namespace

namespace {
// All weight images of a network in one launch.  blockIdx.y = item; the workgroups of an item walk its tiles of
// 16 output rows (m) x KB 16-byte units (k) x all taps.  For the two dense layouts (every Conv2d / ConvTranspose2d
// weight in either orientation) a tile's source is a handful of CONTIGUOUS runs: they are read coalesced into LDS and
// transposed from there, and the packed items of a (tap, unit) pair are written as one contiguous 256-byte row.
// (The first version gathered 4-byte words at a stride of Cin*kh*kw floats: 6.7x over-fetch, 142 us per launch.)
//   mode A (sk == ntaps): src[m*sm + k*ntaps + t]  -> per m one run over (k, t)      (Conv2d forward, deconv dgrad)
//   mode B (sm == ntaps): src[k*sk + m*ntaps + t]  -> per k one run over (m, t)      (Conv2d dgrad, deconv forward)
//   otherwise: per-element gather (column-expanded stem, zero-padded head dgrad: tiny)
constexpr int kPackLdsFloats = 8192;
template <typename T>
__global__ __launch_bounds__(256) void pack_batched_kernel(const ubr_pack_item* items) {
  constexpr int CPU = ET<T>::CPU;
  __shared__ float lds[kPackLdsFloats + 256];
  const ubr_pack_item it = items[blockIdx.y];
  const int nt = it.ntaps;
  const bool modeA = it.tap_stride == 1 && it.sk == nt;
  const bool modeB = it.tap_stride == 1 && it.sm == nt && !modeA;
  if (!modeA && !modeB) {
    const long total = (long)nt * it.KU * it.Mpad;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
      const int m = (int)(i % it.Mpad);
      long r = i / it.Mpad;
      const int ku = (int)(r % it.KU);
      const int t = (int)(r / it.KU);
      const float sc = (it.oscale != nullptr && m < it.M) ? it.oscale[m] : 1.f;
      float f[CPU];
#pragma unroll
      for (int e = 0; e < CPU; ++e) {
        const int kc = ku * CPU + e;
        f[e] = (m < it.M && kc < it.Kvalid) ? it.src[(long)m * it.sm + (long)kc * it.sk + (long)t * it.tap_stride] * sc : 0.f;
      }
      *reinterpret_cast<uint4*>((char*)it.dst + i * 16) = ET<T>::pack(f);
    }
    return;
  }
  int KB = kPackLdsFloats / (16 * CPU * nt);
  if (KB > it.KU) KB = it.KU;
  if (KB > 16) KB = 16;
  if (KB < 1) KB = 1;
  const int mtiles = it.Mpad / 16, ktiles = (it.KU + KB - 1) / KB;
  for (int tile = blockIdx.x; tile < mtiles * ktiles; tile += gridDim.x) {
    const int m0 = (tile % mtiles) * 16, ku0 = (tile / mtiles) * KB;
    const int kb = min(KB, it.KU - ku0);                 // units in this tile
    const int k0 = ku0 * CPU;
    const int kn = max(0, min(kb * CPU, it.Kvalid - k0));   // valid k values
    const int mn = max(0, min(16, it.M - m0));            // valid rows
    __syncthreads();                                      // previous tile fully consumed
    int rowlen;
    if (modeA) {        // lds[m][k*nt + t]
      rowlen = (kb * CPU * nt) | 1;
      const int run = kn * nt;
      for (int i = threadIdx.x; i < mn * run; i += 256) {
        const int m = i / run, c = i - m * run;
        lds[m * rowlen + c] = it.src[(long)(m0 + m) * it.sm + (long)k0 * nt + c];
      }
    } else {            // lds[k][m*nt + t]
      rowlen = (16 * nt) | 1;
      const int run = mn * nt;
      for (int i = threadIdx.x; i < kn * run; i += 256) {
        const int kk = i / run, c = i - kk * run;
        lds[kk * rowlen + c] = it.src[(long)(k0 + kk) * it.sk + (long)m0 * nt + c];
      }
    }
    __syncthreads();
    for (int o = threadIdx.x; o < nt * kb * 16; o += 256) {
      const int m = o & 15;
      const int kul = (o >> 4) % kb, t = (o >> 4) / kb;
      const float sc = (it.oscale != nullptr && m < mn) ? it.oscale[m0 + m] : 1.f;
      float f[CPU];
#pragma unroll
      for (int e = 0; e < CPU; ++e) {
        const int kl = kul * CPU + e;
        float v = 0.f;
        if (m < mn && kl < kn) v = modeA ? lds[m * rowlen + kl * nt + t] : lds[kl * rowlen + m * nt + t];
        f[e] = v * sc;
      }
      *reinterpret_cast<uint4*>((char*)it.dst + (((long)t * it.KU + ku0 + kul) * it.Mpad + m0 + m) * 16) = ET<T>::pack(f);
    }
  }
}

// BatchNorm (eval) folded into the preceding convolution, all sites of a network in one launch:
//   scale[c] = gamma/sqrt(running_var+eps)   (multiplies the packed weights, ubr_pack_item.oscale)
//   bias[c]  = (conv_bias[c] - running_mean[c]) * scale[c] + beta[c]
__global__ void bn_fold_batched_kernel(const ubr_bn_fold_item* items) {
  const ubr_bn_fold_item it = items[blockIdx.y];
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < it.C; c += gridDim.x * blockDim.x) {
    const double s = (double)it.gamma[c] / sqrt((double)it.running_var[c] + (double)it.eps);
    const double b = it.conv_bias != nullptr ? (double)it.conv_bias[c] : 0.0;
    it.scale[c] = (float)s;
    it.bias[c] = (float)((b - (double)it.running_mean[c]) * s + (double)it.beta[c]);
  }
}
}  // namespace

extern "C" int ubr_bn_fold_batched(const ubr_bn_fold_item* items_dev, int nitems, void* stream) {
  UBR_CHECK(items_dev != nullptr && nitems >= 1 && nitems <= 65535, "ubr_bn_fold_batched: bad arguments");
  ubr_launch(bn_fold_batched_kernel, dim3(4, (unsigned)nitems), dim3(256), 0, (hipStream_t)stream, items_dev);
  UBR_LAUNCH_CHECK("ubr_bn_fold_batched");
  return UBR_OK;
}

extern "C" int ubr_pack_weights_batched(int dtype, const ubr_pack_item* items_dev, int nitems, void* stream) {
  UBR_CHECK(ubr_dtype_ok(dtype), "ubr_pack_weights_batched: bad dtype");
  UBR_CHECK(items_dev != nullptr && nitems >= 1 && nitems <= 65535, "ubr_pack_weights_batched: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  // blocks per image: the eight 512x512x9 images are most of the bytes and 48 blocks each left the launch under-parallel
  // (forward repack 60 -> 40 us at 128; 256 over-subscribes the small images' exits)
  dim3 grid(128, (unsigned)nitems);
  if (dtype == UBR_F32) ubr_launch(pack_batched_kernel<float>, grid, dim3(256), 0, st, items_dev);
  else if (dtype == UBR_BF16) ubr_launch(pack_batched_kernel<bf16_t>, grid, dim3(256), 0, st, items_dev);
  else ubr_launch(pack_batched_kernel<f16_t>, grid, dim3(256), 0, st, items_dev);
  UBR_LAUNCH_CHECK("ubr_pack_weights_batched");
  return UBR_OK;
}

extern "C" int ubr_pack_weights(int dtype, const float* src, void* dst, int M, int Mpad, int Kvalid, int Kpad,
                                int64_t sm, int64_t sk, int ntaps, const int32_t* tapidx_host, void* stream) {
  UBR_CHECK(ubr_dtype_ok(dtype), "ubr_pack_weights: bad dtype");
  const int cpu = ubr_cpu(dtype);
  UBR_CHECK(src && dst && tapidx_host, "ubr_pack_weights: null pointer");
  UBR_CHECK(M > 0 && Mpad >= M && Mpad % 16 == 0 && Kvalid > 0 && Kpad >= Kvalid && Kpad % cpu == 0,
            "ubr_pack_weights: bad extents M=%d Mpad=%d K=%d Kpad=%d", M, Mpad, Kvalid, Kpad);
  UBR_CHECK(ntaps >= 1 && ntaps <= UBR_MAX_TAPS, "ubr_pack_weights: ntaps out of range");
  UBR_CHECK(ubr_aligned16(dst), "ubr_pack_weights: dst must be 16-byte aligned");
  PackK k{};
  k.src = src; k.dst = (char*)dst; k.M = M; k.Mpad = Mpad; k.Kvalid = Kvalid; k.KU = Kpad / cpu; k.ntaps = ntaps;
  k.sm = sm; k.sk = sk;
  for (int t = 0; t < ntaps; ++t) k.tapidx[t] = tapidx_host[t];
  const long total = (long)ntaps * k.KU * Mpad;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == UBR_F32) ubr_launch(pack_kernel<float>, dim3(blocks), dim3(256), 0, st, k);
  else if (dtype == UBR_BF16) ubr_launch(pack_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, k);
  else ubr_launch(pack_kernel<f16_t>, dim3(blocks), dim3(256), 0, st, k);
  UBR_LAUNCH_CHECK("ubr_pack_weights");
  return UBR_OK;
}
