// Weight gradients on MFMA with the pixel axis as the GEMM K dimension (gfx950).
//
// Replaces the weight half of autograd's ConvolutionBackward for every conv / deconv of the
// path (reference layers: models/common_layers.py:13-15,33,125; models/ub_uresnet.py:60,64).
//   dW[t][co][ci] = sum_pixels g[p][co] * xform(x)[p + tap_t][ci]
// D[co][ci] += G^T[co][p] * X[p][ci]: both operands need "K contiguous per lane", i.e. the
// TRANSPOSE of the NHWC tiles held in LDS.  For 16-bit types that transpose is free:
// ds_read_b64_tr_b16 delivers a 4-pixel x 16-channel block column-major, and because every lane
// supplies its own row address a tap shift is just a different row -- no alignment constraint.
// fp32 (parity path) uses four scalar LDS reads per fragment and v_mfma_f32_16x16x4_f32.
//
// A workgroup owns (cout tile, cin tile, tap group), walks pixel tiles with a grid stride
// (split-K), keeps all its dW partials in registers, and finally writes ONE fp32 slab;
// ubr_wgrad_reduce sums slabs in fixed order -> bitwise reproducible, no float atomics.
#include <stdlib.h>
#include "ubr_common.h"
#include "ubr_host.h"
#ifndef UBR_WGRAD_TC9
#define UBR_WGRAD_TC9 1
#endif

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

struct WgK {
  const char* x; long x_sn, x_sy, x_sx;
  const float *in_sub, *in_scale, *in_shift, *in_lo;
  const char* g; long g_sn, g_sy, g_sx;
  float* slabs;
  int N, H, W, GH, GW, Cin, Cout_pad;
  int ntaps, S, iy0, ix0, dymin, dxmin, HH, HW;
  int TH, tiles_x, tiles_y, ntiles;
  int pixbG, pixbX, x_off;
  unsigned hw_magic;                             // ceil(2^32 / HW)
  int x_sy32, x_sx32, g_sy32, g_sx32;            // row / pixel strides in bytes (per-image offsets fit 31 bits)
  int n_cot, dbg;
  int rowreuse;                                  // 3x3 unit-stride tap grid, stride 1, full-height tile: compute_tile reads every halo row once
  unsigned long long* stamps;     // diagnostic build (UBR_WGRAD_STAMPS): per-workgroup phase cycle sums
  int8_t dy[UBR_MAX_TAPS], dx[UBR_MAX_TAPS];
};

// transposed fragment: this lane's 4 units-of-K for channel (lane&15); 16-bit types
__device__ __forceinline__ uint4 tr_frag16(const char* p0, int pstride4) {
  typedef __attribute__((address_space(3))) s16x4_t* lds_p;
  s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p0));
  s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p0 + pstride4));
  uint2 a = __builtin_bit_cast(uint2, lo), b = __builtin_bit_cast(uint2, hi);
  return make_uint4(a.x, a.y, b.x, b.y);
}
__device__ __forceinline__ uint4 sc_frag32(const char* p0, int pstride) {
  return make_uint4(*reinterpret_cast<const uint32_t*>(p0), *reinterpret_cast<const uint32_t*>(p0 + pstride),
                    *reinterpret_cast<const uint32_t*>(p0 + 2 * pstride), *reinterpret_cast<const uint32_t*>(p0 + 3 * pstride));
}

// NSPLIT = false: the 4 waves split the pixel rows (K) and their partial sums meet in LDS (thin layers).
// NSPLIT = true : the 4 waves split the cin fragments (N), every wave walks all rows: a 64 x 64-channel tile
//                 with 9 taps costs the same 144 accumulator registers per wave as a K-split 32 x 32 tile but
//                 re-stages G and X half as often per flop and needs no cross-wave reduction (C >= 64 layers).
// Input-halo staging slots (16-byte items) per thread.  The software-pipelined tile loop keeps one tile's loads in
// registers across the MFMA phase, so slots are sized for the common geometry of each variant (stride 1, dilation 1,
// 8-row tiles when waves split rows, 4-row tiles when they split channels, 7x7 for the 25-tap kernel) instead of the
// worst case; the planner shrinks the tile height for dilated / strided layers until their halo fits.
// BIGX variants (9-tap, Cout tile 16 only: the ASPP branches and the column-expanded stem) carry 12 slots.
// The wide 9-tap tile (N-split, MA = 2) walks 8-row pixel tiles: 10 halo rows for 8 output rows instead of 6 for 4 (the BatchNorm
// transform of the halo was 22 % of that kernel, in-kernel stamps), half as many barrier pairs and tile set-ups; its 11 + 4 slots
// replace the two 7 + 2 slot sets of the two-tiles-ahead prefetch (one 8-row MFMA phase covers a load round trip).
#ifndef UBR_WGRAD_TALL
#define UBR_WGRAD_TALL 1
#endif
__host__ __device__ constexpr bool wgrad_tall(bool nsplit, int tpg, int ma) { return UBR_WGRAD_TALL && nsplit && tpg == 9 && ma == 2; }
// K-split (thin-layer) kernels with at most two channel fragments: pixel tiles of UBR_WGRAD_KTH rows.  These layers are
// HBM-bound, one workgroup per CU, and what they keep in flight is two pixel tiles; -DUBR_WGRAD_KTH=16 doubles it with 16-row
// tiles.  Measured in the train step, same box: 11.80 vs 11.74 ms (slower: the longer-lived, larger workgroups cost the compute
// stream more than the weight-gradient stream gains).  Default 8.
#ifndef UBR_WGRAD_KTH
#define UBR_WGRAD_KTH 8
#endif
__host__ __device__ constexpr int wgrad_kth(bool nsplit, int tpg, int ma, int nb) { return (!nsplit && tpg != 25 && ma * nb <= 2) ? UBR_WGRAD_KTH : 8; }
__host__ __device__ constexpr int wgrad_xslots(bool nsplit, int tpg, int nb, int cpu, bool bigx, int ma = 0) {
  return (bigx ? 12 : (wgrad_tall(nsplit, tpg, ma) ? 11 : (nsplit ? 7 : (tpg == 25 ? 5 : (wgrad_kth(nsplit, tpg, ma, nb) == 16 ? (nb == 1 ? 5 : 10) : (nb == 1 ? 3 : 6)))))) * (8 / cpu);
}

// PC = true (the wide 3x3 tile, 16-bit types): 512 threads.  Waves 4-7 (producers) own the staging -- loads two tiles ahead into
// two register sets, the BatchNorm transform IN registers, the LDS writes; waves 0-3 (consumers) own the accumulators and do
// nothing but fragment reads and MFMAs.  With one wave per SIMD the staging of a tile (5.6 k cycles, 2 k of them the transform's
// VALU work) and its MFMA phase (2.5 k) ran one after the other; as two waves per SIMD the transform and the load issue of tile
// i+1 run under the MFMAs of tile i, and only the LDS writes (between the two barriers) stay exposed.
template <typename T, int MA, int NB, int TPG, bool NSPLIT, bool BIGX, bool PC = false>
__global__ __launch_bounds__(PC ? 512 : 256, (TPG == 25 ? 2 : 1)) void wgrad_kernel(const WgK k) {
  constexpr int NBW = NSPLIT ? NB / 4 : NB;     // cin fragments owned by one wave
  static_assert(!NSPLIT || NB % 4 == 0, "N-split needs a multiple of 4 cin fragments");
  constexpr int CPU = ET<T>::CPU;
  constexpr int ESZ = 16 / CPU;
  constexpr int TCO = MA * 16, TCI = NB * 16;
  constexpr int UG = TCO / CPU, UX = TCI / CPU;   // 16-byte units per pixel
  constexpr int PXS = 4 * CPU;                    // pixels per K-step (32 or 16)
  constexpr int KSR = 32 / PXS;                   // K-steps per 32-pixel tile row

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* gl = smem;
  char* xl = smem + k.x_off;

  const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3, q = lane >> 4, l16 = lane & 15;
  const int tid = threadIdx.x & 255;             // staging thread id (PC: of the producer waves) / wave = consumer wave id
  const bool producer = PC && threadIdx.x >= 256;
  const int cot = blockIdx.y % k.n_cot, cit = blockIdx.y / k.n_cot;
  const int co0 = cot * TCO, ci0 = cit * TCI;
  const int t0 = blockIdx.z * TPG;
  const bool has_xf = k.in_scale != nullptr;

  f32x4 acc[TPG][MA][NBW];
#pragma unroll
  for (int t = 0; t < TPG; ++t)
#pragma unroll
    for (int a = 0; a < MA; ++a)
#pragma unroll
      for (int b = 0; b < NBW; ++b) acc[t][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nb0 = NSPLIT ? wave * NBW : 0;

  // ---- staging: each thread keeps ONE channel unit (256 % units-per-pixel == 0), so the BatchNorm constants
  // of its channels sit in registers.  The tile loop is software-pipelined: the global loads of tile i+1 are
  // issued into registers (gv/xv) right before the MFMA phase of tile i and written to LDS after it. ----
  constexpr bool TALL = wgrad_tall(NSPLIT, TPG, MA);
  constexpr int GS = ((NSPLIT && !TALL) ? 4 : wgrad_kth(NSPLIT, TPG, MA, NB)) * 32 * UG / 256;       // G slots per thread (TH <= 4 / 8 / 16)
  constexpr int XS = wgrad_xslots(NSPLIT, TPG, NB, CPU, BIGX, MA);     // X slots per thread (host shrinks the tile until the halo fits)
  float xsub[CPU], xsc[CPU], xsh[CPU], xlo[CPU];
  if (has_xf) {
    const int ch0 = ci0 + (tid % UX) * CPU;
#pragma unroll
    for (int e = 0; e < CPU; ++e) { xsub[e] = k.in_sub[ch0 + e]; xsc[e] = k.in_scale[ch0 + e]; xsh[e] = k.in_shift[ch0 + e]; xlo[e] = k.in_lo[ch0 + e]; }
  }
  const int cg = tid % UG, pg0 = tid / UG;
  constexpr int PGS = 256 / UG;
  const int cx = tid % UX, px0 = tid / UX;
  constexpr int PXS_T = 256 / UX;
  const int npxG = k.TH * 32, npxX = k.HH * k.HW;
  // Tile loads run TWO tiles ahead (register sets A and B) where the kernel owns the whole register file (one wave per
  // SIMD, every variant but the 25-tap one): with one workgroup per CU nothing else hides a load, and one tile's MFMA phase
  // (72 MFMAs per wave, ~0.5 us) is shorter than a round trip to HBM under load -- the wide 9-tap kernel spent more time
  // waiting for the next tile than computing.
  constexpr bool DEEP = TPG != 25 && (!TALL || PC);
  u32x4_t gvA[GS], xvA[XS], gvB[DEEP ? GS : 1], xvB[DEEP ? XS : 1];
  unsigned xokA = 0u, xokB = 0u;

  // A thread's staging slots cover the same tile-relative pixels in every tile, so everything about a slot that does not
  // depend on the tile is computed ONCE: its byte offset relative to the tile origin (channel unit included; out of range
  // for slots beyond the tile), its row / column for the image-edge tests, its LDS address.  Per tile and slot that leaves one
  // add, one column compare and a select in front of a buffer load whose range check supplies the zero rows above and below
  // the image (the per-image resource covers exactly the image).  With one wave per SIMD every instruction of this phase is
  // serial issue time: the guarded-load form (two compares, a branch and 64-bit address arithmetic per slot) took 1.9 k of the
  // 7.8 k cycles a 256-channel tile costs, 1.2 k of them MFMA (in-kernel cycle stamps, tools/microbench.py kernelonly).
  constexpr int kOOR = (int)0x80000000;
  int g_col[GS], g_rel[GS];
#pragma unroll
  for (int u = 0; u < GS; ++u) {
    const int px = pg0 + u * PGS;
    g_col[u] = px & 31;
    g_rel[u] = px < npxG ? (px >> 5) * k.g_sy32 + (px & 31) * k.g_sx32 + (co0 + cg * CPU) * ESZ : kOOR;
  }
  int x_rc[XS], x_rel[XS];
#pragma unroll
  for (int u = 0; u < XS; ++u) {
    const int px = px0 + u * PXS_T;
    const int hy = (int)__umulhi((unsigned)px, k.hw_magic), hx = px - hy * k.HW;
    x_rc[u] = (hy << 16) | hx;
    x_rel[u] = px < npxX ? hy * k.x_sy32 + hx * k.x_sx32 + (ci0 + cx * CPU) * ESZ : kOOR;
  }
  const int g_img = k.GH * k.g_sy32, x_img = k.H * k.x_sy32;      // bytes of one image (rows beyond it read as zero)

  // Successive load_tile calls walk tiles blockIdx.x, +G, +2G, ...: the (column, row, image) decomposition is carried along
  // and advanced by the decomposition of the grid stride instead of being re-derived by three divisions per tile.
  int nx_tx, nx_ty, nx_n, st_tx, st_ty, st_n;
  {
    int tt = blockIdx.x;
    nx_tx = tt % k.tiles_x; tt /= k.tiles_x; nx_ty = tt % k.tiles_y; nx_n = tt / k.tiles_y;
    tt = gridDim.x;
    st_tx = tt % k.tiles_x; tt /= k.tiles_x; st_ty = tt % k.tiles_y; st_n = tt / k.tiles_y;
  }
  auto load_tile = [&](auto& gv, auto& xv, unsigned& xok) {
#ifdef UBR_WGRAD_STAMPS
    if (k.dbg & 16) return;
#endif
    const int tx = nx_tx, ty = nx_ty, n = nx_n;
    nx_tx += st_tx; nx_ty += st_ty; nx_n += st_n;
    if (nx_tx >= k.tiles_x) { nx_tx -= k.tiles_x; nx_ty += 1; }
    if (nx_ty >= k.tiles_y) { nx_ty -= k.tiles_y; nx_n += 1; }
    const int oy0 = ty * k.TH, ox0 = tx * 32;
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc((void*)(k.g + (long)n * k.g_sn), 0, g_img, 0x00020000);
    const int gorg = oy0 * k.g_sy32 + ox0 * k.g_sx32;
    const int grx = k.GW - ox0;          // columns of the gradient grid left from the tile origin
#pragma unroll
    for (int u = 0; u < GS; ++u) {
      int vo = g_col[u] < grx ? g_rel[u] + gorg : -1;
#ifdef UBR_WGRAD_STAMPS
      if (k.dbg & 1) vo = -1;
#endif
      gv[u] = __builtin_amdgcn_raw_buffer_load_b128(gr, vo, 0, 0);
    }
    const int hy0 = oy0 * k.S + k.iy0 + k.dymin, hx0 = ox0 * k.S + k.ix0 + k.dxmin;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)(k.x + (long)n * k.x_sn), 0, x_img, 0x00020000);
    const int xorg = hy0 * k.x_sy32 + hx0 * k.x_sx32;      // negative above / left of the image: the sum wraps out of range
    xok = 0u;
#pragma unroll
    for (int u = 0; u < XS; ++u) {
      const bool okx = (unsigned)(hx0 + (x_rc[u] & 0xffff)) < (unsigned)k.W;
      int vo = okx ? x_rel[u] + xorg : -1;
#ifdef UBR_WGRAD_STAMPS
      if (k.dbg & 1) vo = -1;
#endif
      xv[u] = __builtin_amdgcn_raw_buffer_load_b128(xr, vo, 0, 0);
      if (has_xf) xok |= (okx && x_rel[u] != kOOR && (unsigned)(hy0 + (x_rc[u] >> 16)) < (unsigned)k.H) ? (1u << u) : 0u;   // padding stays zero through the transform
    }
  };
  // LDS side: complete slots (every thread has a pixel) and the partial one are wave-uniform tests
  char* const gl_w = gl + pg0 * k.pixbG + cg * 16;
  char* const xl_w = xl + px0 * k.pixbX + cx * 16;
  const int ginc = PGS * k.pixbG, xinc = PXS_T * k.pixbX;
  const int gfull = npxG / PGS, grem = npxG % PGS, xfull = npxX / PXS_T, xrem = npxX % PXS_T;
  auto xform_tile = [&](auto& xv, unsigned xok) {      // PC producers: the transform alone, in registers
    if (!has_xf) return;
#ifdef UBR_WGRAD_STAMPS
    if (k.dbg & 8) return;
#endif
#pragma unroll
    for (int u = 0; u < XS; ++u) {
      float f[CPU];
      ET<T>::unpack(make_uint4(xv[u].x, xv[u].y, xv[u].z, xv[u].w), f);
      ubr_bnrelu<CPU>(f, xsub, xsc, xsh, xlo);
      const uint4 t4 = ET<T>::pack(f);
      const bool ok = (xok >> u) & 1u;
      xv[u].x = ok ? t4.x : 0u; xv[u].y = ok ? t4.y : 0u; xv[u].z = ok ? t4.z : 0u; xv[u].w = ok ? t4.w : 0u;
    }
  };
  auto store_tile = [&](const auto& gv, const auto& xv, unsigned xok, bool raw = false) {
#ifdef UBR_WGRAD_STAMPS
    if (k.dbg & 4) return;
#endif
#pragma unroll
    for (int u = 0; u < GS; ++u)
      if (u < gfull || (u == gfull && pg0 < grem)) *reinterpret_cast<u32x4_t*>(gl_w + u * ginc) = gv[u];
#pragma unroll
    for (int u = 0; u < XS; ++u) {
      u32x4_t v = xv[u];
      if (has_xf && !raw) {
        float f[CPU];
        ET<T>::unpack(make_uint4(v.x, v.y, v.z, v.w), f);
        ubr_bnrelu<CPU>(f, xsub, xsc, xsh, xlo);
        const uint4 t4 = ET<T>::pack(f);
        const bool ok = (xok >> u) & 1u;
        v.x = ok ? t4.x : 0u; v.y = ok ? t4.y : 0u; v.z = ok ? t4.z : 0u; v.w = ok ? t4.w : 0u;
      }
      if (u < xfull || (u == xfull && px0 < xrem)) *reinterpret_cast<u32x4_t*>(xl_w + u * xinc) = v;
    }
  };

  // per-tap LDS offsets are wave-uniform: hoist them out of the row loop
  int toff[TPG];
#pragma unroll
  for (int t = 0; t < TPG; ++t)
    toff[t] = (t0 + t < k.ntaps) ? ((k.dy[t0 + t] - k.dymin) * k.HW + (k.dx[t0 + t] - k.dxmin)) * k.pixbX : 0;

  auto compute_tile = [&]() {
    // ---- 3x3 layers, 16-bit types: input row i of the halo feeds output rows i, i-1 and i-2 (vertical taps 0, 1, 2), so a wave
    // that owns CONSECUTIVE output rows reads each halo row's three column-shifted fragments once and keeps a three-row window
    // of gradient fragments: (R + 2) * 3 * NBW + R * MA fragment reads for R rows instead of R * (9 * NBW + MA).  The MFMA
    // phase of these kernels is bound by the ds_read_b64_tr_b16 rate, not by the matrix pipe (in-kernel stamps, wide tile at
    // 16x128x128x64: 4.5 k cycles per 8-row tile for 2.3 k cycles of MFMA).  Every accumulator still sums its rows in
    // ascending order, so the N-split kernels produce the same bits as the tap-by-tap loop below. ----
    if constexpr (TPG == 9 && CPU == 8) {
      constexpr int ROWS = NSPLIT ? (TALL ? 8 : 4) : wgrad_kth(NSPLIT, TPG, MA, NB) / 4;
      if (k.rowreuse) {
        const int r0 = NSPLIT ? 0 : wave * ROWS;
        const char* pa = gl + (r0 * 32 + 4 * q + (l16 >> 2)) * k.pixbG + (l16 & 3) * 8;
        int xrow = ((r0 * k.HW) + 4 * q + (l16 >> 2)) * k.pixbX + (l16 & 3) * 8 + nb0 * 32;
        asm volatile("" : "+v"(xrow));
        const int arow = 32 * k.pixbG, brow = k.HW * k.pixbX;
        uint4 Aw[3][MA];
#pragma unroll
        for (int i = 0; i < ROWS + 2; ++i) {
          if (i < ROWS) {
#pragma unroll
            for (int a = 0; a < MA; ++a) Aw[i % 3][a] = tr_frag16(pa + i * arow + a * 32, 16 * k.pixbG);
          }
          uint4 B[3][NBW];
#pragma unroll
          for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int b = 0; b < NBW; ++b) B[dx][b] = tr_frag16(xl + (xrow + i * brow + toff[dx] + b * 32), 16 * k.pixbX);
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            const int r = i - dy;
            if (r >= 0 && r < ROWS) {
#pragma unroll
              for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int b = 0; b < NBW; ++b)
#pragma unroll
                  for (int a = 0; a < MA; ++a) acc[dy * 3 + dx][a][b] = mma_step<T>(acc[dy * 3 + dx][a][b], B[dx][b], Aw[r % 3][a]);
            }
          }
        }
        return;
      }
    }
    // ---- MFMA: K-split: wave takes rows wave, wave+4, ...; N-split: every wave takes every row ----
#ifdef UBR_WGRAD_STAMPS
    const int nrows = (k.dbg & 2) ? 1 : k.TH;
#else
    const int nrows = k.TH;
#endif
    for (int r = NSPLIT ? 0 : wave; r < nrows; r += NSPLIT ? 1 : 4) {
#pragma unroll
      for (int ks = 0; ks < KSR; ++ks) {
        uint4 A[MA];
        if constexpr (CPU == 8) {
          // K (pixel) assignment inside a 32-pixel row: quad q takes pixels 4q..4q+3 (low half) and 16+4q..16+4q+3 (high half),
          // the same for both operands.  A 32-lane half of a ds_read_b64_tr_b16 then covers 8 CONSECUTIVE pixels, which with a
          // pixel stride of 32*odd bytes is conflict-free; round 1's "8q + {0..3}, +4" assignment (pixels {0-3, 8-11} per half)
          // is a 2-way bank conflict for every stride (tools/lds_banks.py).
          const char* p = gl + (r * 32 + 4 * q + (l16 >> 2)) * k.pixbG + (l16 & 3) * 8;
#pragma unroll
          for (int a = 0; a < MA; ++a) A[a] = tr_frag16(p + a * 32, 16 * k.pixbG);
        } else {
          const char* p = gl + (r * 32 + ks * 16 + 4 * q) * k.pixbG + l16 * 4;
#pragma unroll
          for (int a = 0; a < MA; ++a) A[a] = sc_frag32(p + a * 64, k.pixbG);
        }
        // this lane's LDS offset inside row r; kept opaque so that the compiler adds the (scalar) tap offset at each
        // read instead of hoisting one address register per tap out of the row loop (50 VGPRs for 25 taps)
        int xrow;
        if constexpr (CPU == 8) xrow = (r * k.S * k.HW + (4 * q + (l16 >> 2)) * k.S) * k.pixbX + (l16 & 3) * 8;
        else xrow = (r * k.S * k.HW + (ks * 16 + 4 * q) * k.S) * k.pixbX + l16 * 4;
        asm volatile("" : "+v"(xrow));
        // Taps beyond ntaps (toff = 0) are computed and discarded: no branch.  The taps go in chunks of TC: all of a
        // chunk's LDS fragment reads are issued first, then its MFMAs, so a read's latency hides under the
        // previous MFMAs instead of being waited for tap by tap.
        constexpr int TC = (TPG == 25) ? 5 : ((TPG == 9 && NSPLIT) ? UBR_WGRAD_TC9 : 1);   // taps per read-ahead chunk
#pragma unroll
        for (int tc = 0; tc < TPG; tc += TC) {
          uint4 B[TC][NBW];
#pragma unroll
          for (int tt = 0; tt < TC; ++tt)
#pragma unroll
            for (int b = 0; b < NBW; ++b) {
              if constexpr (CPU == 8) {
                B[tt][b] = tr_frag16(xl + (xrow + toff[tc + tt] + (nb0 + b) * 32), 16 * k.S * k.pixbX);
              } else {
                B[tt][b] = sc_frag32(xl + (xrow + toff[tc + tt] + (nb0 + b) * 64), k.S * k.pixbX);
              }
            }
          if constexpr (TC > 1) __builtin_amdgcn_sched_barrier(0);   // keep the reads grouped ahead of the MFMAs (the scheduler would re-pair them)
#pragma unroll
          for (int tt = 0; tt < TC; ++tt)
#pragma unroll
            for (int b = 0; b < NBW; ++b)
#pragma unroll
              for (int a = 0; a < MA; ++a) acc[tc + tt][a][b] = mma_step<T>(acc[tc + tt][a][b], B[tt][b], A[a]);   // D[cin = 4q+r][cout = l16]
          if constexpr (TC > 1) __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  };
  const int G = gridDim.x;
  int cur = blockIdx.x;
#ifdef UBR_WGRAD_STAMPS
  unsigned long long tS = 0, tL = 0, tC = 0, tB = 0, tW = 0, t_;
#define UBR_STAMP(acc_) do { unsigned long long n_ = __builtin_amdgcn_s_memtime(); acc_ += n_ - t_; t_ = n_; } while (0)
  const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
  t_ = t_begin;
#else
#define UBR_STAMP(acc_) do { } while (0)
#endif
  if constexpr (PC) {
    if (producer) {
      if (cur < k.ntiles) load_tile(gvA, xvA, xokA);
      if (cur + G < k.ntiles) load_tile(gvB, xvB, xokB);
      UBR_STAMP(tL);
      while (cur < k.ntiles) {
        xform_tile(xvA, xokA);
        UBR_STAMP(tS);
        __syncthreads();     // the consumers are done with the previous tile
        UBR_STAMP(tB);
        store_tile(gvA, xvA, xokA, true);
        UBR_STAMP(tW);
        __syncthreads();     // tile complete in LDS
        UBR_STAMP(tB);
        if (cur + 2 * G < k.ntiles) load_tile(gvA, xvA, xokA);
        UBR_STAMP(tL);
        cur += G;
        if (cur >= k.ntiles) break;
        xform_tile(xvB, xokB);
        UBR_STAMP(tS);
        __syncthreads();
        UBR_STAMP(tB);
        store_tile(gvB, xvB, xokB, true);
        UBR_STAMP(tW);
        __syncthreads();
        UBR_STAMP(tB);
        if (cur + 2 * G < k.ntiles) load_tile(gvB, xvB, xokB);
        UBR_STAMP(tL);
        cur += G;
      }
#ifdef UBR_WGRAD_STAMPS
      if (k.stamps != nullptr && threadIdx.x == 256) {     // producer wave 4: transform (incl. waiting for the loads), barriers, LDS writes, load issue
        unsigned long long* o = k.stamps + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16 + 8;
        o[0] = tS; o[1] = tL; o[2] = tW; o[3] = tB; o[4] = t_ - t_begin;
      }
#endif
      return;                // (the slab is written by the consumers; no barrier follows the tile loop)
    }
    while (cur < k.ntiles) {
      __syncthreads();
      __syncthreads();
      UBR_STAMP(tB);
      compute_tile();
      UBR_STAMP(tC);
      cur += G;
    }
  } else {
  if (cur < k.ntiles) load_tile(gvA, xvA, xokA);
  if constexpr (DEEP) {
    if (cur + G < k.ntiles) load_tile(gvB, xvB, xokB);
    UBR_STAMP(tL);
    while (cur < k.ntiles) {
      __syncthreads();     // previous tile fully consumed
#ifdef UBR_WGRAD_STAMPS
      UBR_STAMP(tB);
      asm volatile("s_waitcnt vmcnt(9)" ::: "memory");        // (GS + XS loads of the other register set may stay in flight)
      UBR_STAMP(tW);
#endif
      store_tile(gvA, xvA, xokA);
      UBR_STAMP(tS);
      __syncthreads();
      UBR_STAMP(tB);
      if (cur + 2 * G < k.ntiles) load_tile(gvA, xvA, xokA);      // in flight during two MFMA phases
      UBR_STAMP(tL);
      compute_tile();
      UBR_STAMP(tC);
      cur += G;
      if (cur >= k.ntiles) break;
      __syncthreads();
#ifdef UBR_WGRAD_STAMPS
      UBR_STAMP(tB);
      asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
      UBR_STAMP(tW);
#endif
      store_tile(gvB, xvB, xokB);
      UBR_STAMP(tS);
      __syncthreads();
      UBR_STAMP(tB);
      if (cur + 2 * G < k.ntiles) load_tile(gvB, xvB, xokB);
      UBR_STAMP(tL);
      compute_tile();
      UBR_STAMP(tC);
      cur += G;
    }
  } else {
    while (cur < k.ntiles) {
      __syncthreads();     // previous tile fully consumed
      store_tile(gvA, xvA, xokA);
      __syncthreads();
      const int next = cur + G;
      if (next < k.ntiles) load_tile(gvA, xvA, xokA);      // in flight during the MFMA phase below
      cur = next;
      compute_tile();
    }
  }
  }

  float* slab = k.slabs + (long)blockIdx.x * k.ntaps * k.Cout_pad * k.Cin;
  // The accumulators hold dW TRANSPOSED (rows = 4 consecutive input channels per lane, columns = output channels across the
  // 16 lanes of a quad): the slab is [tap][cout][cin] with cin contiguous, so every lane owns 16 contiguous bytes per tile and
  // the epilogue is one 16-byte store per accumulator instead of four scattered 4-byte ones (the slab write was ~40 % of the
  // wide 9-tap kernel: 74 KB per workgroup behind only 16 pixel tiles of MFMA work).
  if constexpr (NSPLIT) {
    // every wave owns its cin fragments outright: write the slab straight from the accumulators
#pragma unroll
    for (int t = 0; t < TPG; ++t) {
      if (t0 + t < k.ntaps) {
#pragma unroll
        for (int a = 0; a < MA; ++a)
#pragma unroll
          for (int b = 0; b < NBW; ++b) {
            const int co = co0 + a * 16 + l16, ci = ci0 + (nb0 + b) * 16 + 4 * q;
            *reinterpret_cast<float4*>(&slab[((long)(t0 + t) * k.Cout_pad + co) * k.Cin + ci]) =
                make_float4(acc[t][a][b][0], acc[t][a][b][1], acc[t][a][b][2], acc[t][a][b][3]);
          }
      }
    }
  } else {
    // ---- cross-wave sum in fixed order (wave 0,1,2,3) through LDS, then one slab write ----
    float* red = reinterpret_cast<float*>(smem);
    for (int w = 0; w < 4; ++w) {
      __syncthreads();
      if (wave == w) {
#pragma unroll
        for (int t = 0; t < TPG; ++t)
#pragma unroll
          for (int a = 0; a < MA; ++a)
#pragma unroll
            for (int b = 0; b < NBW; ++b) {
              float4* p = reinterpret_cast<float4*>(red + (((t * MA + a) * NB + b) * 64 + lane) * 4);
              const float4 v = make_float4(acc[t][a][b][0], acc[t][a][b][1], acc[t][a][b][2], acc[t][a][b][3]);
              if (w == 0) *p = v;
              else { const float4 o = *p; *p = make_float4(o.x + v.x, o.y + v.y, o.z + v.z, o.w + v.w); }
            }
      }
    }
    __syncthreads();
    for (int s = wave; s < TPG * MA * NB; s += 4) {
      const int b = s % NB, a = (s / NB) % MA, t = s / (NB * MA);
      if (t0 + t < k.ntaps) {
        const int co = co0 + a * 16 + l16, ci = ci0 + b * 16 + 4 * q;
        *reinterpret_cast<float4*>(&slab[((long)(t0 + t) * k.Cout_pad + co) * k.Cin + ci]) = *reinterpret_cast<const float4*>(red + (s * 64 + lane) * 4);
      }
    }
  }
#ifdef UBR_WGRAD_STAMPS
  if (k.stamps != nullptr && tid == 0) {
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    unsigned long long* o = k.stamps + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (PC ? 16 : 8);
    o[0] = tS; o[1] = tL; o[2] = tC; o[3] = t_end - t_; o[4] = t_end - t_begin; o[5] = t_begin; o[6] = tB; o[7] = tW;
  }
#endif
}

struct WPlan { int MA, NB, TPG, TH, HH, HW, pixbG, pixbX, x_off, tiles_x, tiles_y, ntiles, nsplit, gy, gz, nsplit_mode, bigx; size_t lds; };

static int wgrad_plan(const ubr_wgrad_desc* d, WPlan* p) {
  UBR_CHECK(d != nullptr, "ubr_wgrad: null descriptor");
  UBR_CHECK(ubr_dtype_ok(d->dtype), "ubr_wgrad: bad dtype");
  const int esz = ubr_esize(d->dtype);
  UBR_CHECK(d->N > 0 && d->H > 0 && d->W > 0 && d->GH > 0 && d->GW > 0, "ubr_wgrad: empty extent");
  UBR_CHECK(d->Cin > 0 && d->Cin % 16 == 0 && d->Cout > 0 && d->Cout % 16 == 0, "ubr_wgrad: channels must be multiples of 16 (Cin=%d Cout=%d)", d->Cin, d->Cout);
  UBR_CHECK(d->ntaps >= 1 && d->ntaps <= UBR_MAX_TAPS, "ubr_wgrad: ntaps out of range");
  UBR_CHECK(d->S == 1 || d->S == 2, "ubr_wgrad: S unsupported");
  int dymin = 127, dymax = -128, dxmin = 127, dxmax = -128;
  for (int t = 0; t < d->ntaps; ++t) {
    dymin = d->dy[t] < dymin ? d->dy[t] : dymin; dymax = d->dy[t] > dymax ? d->dy[t] : dymax;
    dxmin = d->dx[t] < dxmin ? d->dx[t] : dxmin; dxmax = d->dx[t] > dxmax ? d->dx[t] : dxmax;
  }
  int TPG = d->ntaps <= 1 ? 1 : d->ntaps <= 4 ? 4 : d->ntaps <= 9 ? 9 : 25;
  int MA = (d->Cout % 32 == 0) ? 2 : 1, NB = (d->Cin % 32 == 0) ? 2 : 1;
  if (TPG == 25) { MA = 1; NB = 1; }
  if (TPG == 9 && MA == 2 && NB == 2) NB = 1;   // the 32 x 32 x 9-tap tile needs 320 VGPRs (one wave per SIMD); 32 x 16 fits two: 89 -> 64 us
  p->nsplit_mode = 0;
  p->TH = d->S == 1 ? 8 : 4;   // rows beyond GH are zero-filled, so small grids stay correct
  if (wgrad_kth(false, TPG, MA, NB) == 16 && d->S == 1 && d->GH >= 16) p->TH = 16;
  if (TPG <= 9 && d->Cout % 64 == 0 && d->Cin % 64 == 0) {   // wide layers: 64 x 64 channel tile, waves split cin
    static const int ma9 = [] { const char* e = getenv("UBR_WGRAD_MA9"); return e ? atoi(e) : 2; }();
    MA = (TPG == 9) ? ma9 : 4; NB = 4; p->nsplit_mode = 1;   // 9 taps: 32 x 64 tile keeps the kernel under 256 VGPRs (2 waves/SIMD)
    p->TH = d->S == 1 ? (wgrad_tall(true, TPG, MA) ? 8 : 4) : 2;
  }
  {
    // the pipelined loop keeps one tile in registers: shrink the tile height, then the channel tile, until the
    // input halo fits the per-thread staging slots (large dilations / stride 2 / fp32 have the biggest halos)
    const int cpu = ubr_cpu(d->dtype);
    const int th0 = p->TH;
    p->bigx = 0;
    for (;;) {
      const int xs = wgrad_xslots(p->nsplit_mode != 0, TPG, NB, cpu, p->bigx != 0, MA);
      const int ux = NB * 16 / cpu;
      p->HH = (p->TH - 1) * d->S + 1 + (dymax - dymin);
      p->HW = 31 * d->S + 1 + (dxmax - dxmin);
      if ((long)p->HH * p->HW * ux <= 256L * xs) break;
      // more slots before a shorter tile, where the wide-slot variant exists
      // (an exclusive launch prefers a shorter tile on the narrow-slot variant: 158 instead of 269 registers, three waves per SIMD)
      if (!d->exclusive && !p->bigx && !p->nsplit_mode && MA == 1 && TPG == 9) { p->bigx = 1; continue; }
      if (p->TH > 1) { p->TH /= 2; continue; }
      if (p->nsplit_mode) { p->nsplit_mode = 0; MA = 2; NB = 2; p->TH = d->S == 1 ? 8 : 4; continue; }
      if (NB > 1) { NB = 1; p->bigx = 0; p->TH = th0 > 4 ? 4 : th0; continue; }
      if (MA > 1 && TPG == 9) { MA = 1; p->TH = th0 > 4 ? 4 : th0; continue; }
      ubr_set_error("ubr_wgrad: halo of %d x %d pixels does not fit the staging registers", p->HH, p->HW);
      return UBR_EINVAL;
    }
  }
  p->MA = MA; p->NB = NB; p->TPG = TPG;
  p->pixbG = MA * 16 * esz + 16;
  p->pixbX = NB * 16 * esz + 16;
  if (esz == 2) {
    // conflict-free strides of the transposed fragment reads (see wgrad_kernel): 32 * odd bytes where consecutive K pixels are
    // one pixel apart (the gradient tile, and the input tile at stride 1); the padded stride already is conflict-free at stride 2
    auto odd32 = [](int bytes) { int k = (bytes + 31) / 32; if (k % 2 == 0) ++k; return 32 * k; };
    p->pixbG = odd32(MA * 32);
    if (d->S == 1) p->pixbX = odd32(NB * 32);
  }
  size_t gbytes = (size_t)p->TH * 32 * p->pixbG;
  gbytes = (gbytes + 15) & ~(size_t)15;
  p->x_off = (int)gbytes;
  size_t stage = gbytes + (size_t)p->HH * p->HW * p->pixbX;
  size_t redb = p->nsplit_mode ? 0 : (size_t)TPG * MA * NB * 4 * 64 * sizeof(float);
  p->lds = stage > redb ? stage : redb;
  UBR_CHECK(p->lds <= 160 * 1024, "ubr_wgrad: LDS need %zu exceeds 160 KiB", p->lds);
  p->tiles_x = ubr_cdiv(d->GW, 32); p->tiles_y = ubr_cdiv(d->GH, p->TH);
  p->ntiles = p->tiles_x * p->tiles_y * d->N;
  p->gy = (d->Cout / (MA * 16)) * (d->Cin / (NB * 16));
  p->gz = ubr_cdiv(d->ntaps, TPG);
  // One workgroup per CU.  More would finish a lone weight gradient sooner, but these kernels run on the side stream
  // beside the dgrad chain and their long-lived workgroups (200+ VGPRs, grid-stride over tiles) must leave register
  // file and LDS for the compute stream's kernels: 512 / 1024 workgroups cost the step 5 % (1077 vs 1136 img/s).
  // (round 3: the wide, channel-split kernels -- producer/consumer, eight waves, all of a CU's registers -- on HALF the CUs: 128
  // workgroups leave the other 128 CUs to the compute stream outright instead of time-slicing all 256, and halve the slabs:
  // 11.74 -> 11.39 ms/step; 96: 11.77, 160: 11.49, 64: 12.5.  The K-split kernels of the thin layers stay at 256: fewer is slower.)
  static const int tgt_n = [] { const char* e = getenv("UBR_WGRAD_TARGET_N"); return e ? atoi(e) : 128; }();
  static const int tgt_k = [] { const char* e = getenv("UBR_WGRAD_TARGET_K"); return e ? atoi(e) : 256; }();
  // an exclusive launch (the stem's weight gradient closes the backward pass: the compute stream is idle by then) takes three
  // workgroups per CU: 188 -> 88 us for the 7-tap 16-channel layer at 16x512x512
  int target = (d->exclusive ? 768 : (p->nsplit_mode ? tgt_n : tgt_k)) / (p->gy * p->gz);
  if (target < 1) target = 1;
  // bound slab memory: at most 64 MiB of partials per launch
  const size_t slab_bytes = (size_t)d->ntaps * d->Cout * d->Cin * sizeof(float);
  size_t cap = (64u << 20) / (slab_bytes ? slab_bytes : 1);
  if (cap < 1) cap = 1;
  if ((size_t)target > cap) target = (int)cap;
  p->nsplit = p->ntiles < target ? p->ntiles : target;
  return UBR_OK;
}

template <typename T, int MA, int NB, int TPG, bool NSPLIT, bool BIGX = false, bool PC = false>
int wlaunch(const WgK& k, const WPlan& p, hipStream_t st) {
  auto fn = wgrad_kernel<T, MA, NB, TPG, NSPLIT, BIGX, PC>;
  if (p.lds > 64 * 1024) {
    static thread_local size_t maxset = 0;
    if (p.lds > maxset) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds);
      if (e != hipSuccess) { ubr_set_error("ubr_wgrad: cannot raise LDS limit: %s", hipGetErrorString(e)); return UBR_ELAUNCH; }
      maxset = p.lds;
    }
  }
  ubr_launch(fn, dim3(p.nsplit, p.gy, p.gz), dim3(PC ? 512 : 256), p.lds, st, k);
  UBR_LAUNCH_CHECK("ubr_wgrad");
  return UBR_OK;
}

static thread_local int g_last_wgrad_cfg[6] = {0, 0, 0, 0, 0, 0};   // MA, NB, TPG, N-split, wide staging, producer/consumer of this thread's last launch

template <typename T>
int wdispatch(const WgK& k, const WPlan& p, hipStream_t st) {
  if (p.bigx && !p.nsplit_mode && p.MA == 1 && p.TPG == 9) {
    if (p.NB == 1) return wlaunch<T, 1, 1, 9, false, true>(k, p, st);
    if (p.NB == 2) return wlaunch<T, 1, 2, 9, false, true>(k, p, st);
  }
  if constexpr (sizeof(T) == 2) {
    static const int pc = [] { const char* e = getenv("UBR_WGRAD_PC"); return e ? atoi(e) : 1; }();
    if (pc && !p.bigx && p.nsplit_mode && p.MA == 2 && p.NB == 4 && p.TPG == 9) { g_last_wgrad_cfg[5] = 1; return wlaunch<T, 2, 4, 9, true, false, true>(k, p, st); }
  }
#define UBR_WCASE(ma, nb, tpg) if (!p.bigx && !p.nsplit_mode && p.MA == ma && p.NB == nb && p.TPG == tpg) return wlaunch<T, ma, nb, tpg, false>(k, p, st);
#define UBR_NCASE(ma, tpg) if (!p.bigx && p.nsplit_mode && p.MA == ma && p.NB == 4 && p.TPG == tpg) return wlaunch<T, ma, 4, tpg, true>(k, p, st);
  UBR_NCASE(4, 1) UBR_NCASE(4, 4) UBR_NCASE(2, 9) UBR_NCASE(4, 9)
#undef UBR_NCASE
  UBR_WCASE(1, 1, 1) UBR_WCASE(1, 2, 1) UBR_WCASE(2, 1, 1) UBR_WCASE(2, 2, 1)
  UBR_WCASE(1, 1, 4) UBR_WCASE(1, 2, 4) UBR_WCASE(2, 1, 4) UBR_WCASE(2, 2, 4)
  UBR_WCASE(1, 1, 9) UBR_WCASE(1, 2, 9) UBR_WCASE(2, 1, 9) UBR_WCASE(2, 2, 9)
  UBR_WCASE(1, 1, 25)
#undef UBR_WCASE
  ubr_set_error("ubr_wgrad: no kernel for MA=%d NB=%d TPG=%d", p.MA, p.NB, p.TPG);
  return UBR_EINVAL;
}

struct RedK {
  const float* slabs; float* dst;
  int nsplit, ntaps, Cout_pad, Cin, Cout_valid, Cin_valid, accumulate;
  long sm, sk, slab_stride;
  int tapidx[UBR_MAX_TAPS];
};
// One launch per weight gradient, whatever the number of slabs: a block owns 64 consecutive elements; its four waves each sum
// a contiguous quarter of the slabs (eight loads in flight, added in slab order) and the four partials meet in LDS in a fixed
// order -- bitwise reproducible, and no thread's serial chain is longer than nsplit / 4 loads.  (The first form took two launches
// when there were more than 32 slabs: 35 extra launches per step on the weight-gradient stream.)
constexpr int kRedMaxSplit = 1024;
// few slabs (the wide layers: 2-8 slabs of up to 9.4 MB): one thread per element, every slab load in flight at once
__global__ __launch_bounds__(256) void wgrad_reduce_flat_kernel(const RedK k) {
  const long per = (long)k.ntaps * k.Cout_pad * k.Cin;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < per; i += (long)gridDim.x * 256) {
    const int ci = (int)(i % k.Cin);
    long r = i / k.Cin;
    const int co = (int)(r % k.Cout_pad);
    const int t = (int)(r / k.Cout_pad);
    if (co >= k.Cout_valid || ci >= k.Cin_valid) continue;
    float s = 0.f;
#pragma unroll 8
    for (int sp = 0; sp < k.nsplit; ++sp) s += k.slabs[(long)sp * k.slab_stride + i];
    float* d = k.dst + (long)co * k.sm + (long)ci * k.sk + k.tapidx[t];
    *d = k.accumulate ? (*d + s) : s;
  }
}
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const RedK k) {
  __shared__ float part[4][64];
  const long per = (long)k.ntaps * k.Cout_pad * k.Cin;
  const int e = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const int q = (k.nsplit + 3) / 4;
  const int s0 = sg * q, s1 = min(s0 + q, k.nsplit);
  for (long base = (long)blockIdx.x * 64; base < per; base += (long)gridDim.x * 64) {
    const long i = base + e;
    float s = 0.f;
    if (i < per) {
      int sp = s0;
      for (; sp + 8 <= s1; sp += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = k.slabs[(long)(sp + u) * k.slab_stride + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
      }
      for (; sp < s1; ++sp) s += k.slabs[(long)sp * k.slab_stride + i];
    }
    part[sg][e] = s;
    __syncthreads();
    if (sg == 0 && i < per) {
      const int ci = (int)(i % k.Cin);
      long r = i / k.Cin;
      const int co = (int)(r % k.Cout_pad);
      const int t = (int)(r / k.Cout_pad);
      if (co < k.Cout_valid && ci < k.Cin_valid) {
        const float tot = (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
        float* d = k.dst + (long)co * k.sm + (long)ci * k.sk + k.tapidx[t];
        *d = k.accumulate ? (*d + tot) : tot;
      }
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" int ubr_wgrad_plan(const ubr_wgrad_desc* d, int32_t* nsplit, int64_t* workspace_bytes) {
  WPlan p{};
  int rc = wgrad_plan(d, &p);
  if (rc != UBR_OK) return rc;
  if (nsplit) *nsplit = p.nsplit;
  if (workspace_bytes) *workspace_bytes = (int64_t)p.nsplit * d->ntaps * d->Cout * d->Cin * (int64_t)sizeof(float);
  return UBR_OK;
}

extern "C" int ubr_wgrad_last_pc(void) { return g_last_wgrad_cfg[5]; }     // 1: the last launch was the producer/consumer variant (7th template argument)
extern "C" void ubr_wgrad_last_config(int* ma, int* nb, int* tpg, int* nsplit_mode, int* bigx) {
  if (ma) *ma = g_last_wgrad_cfg[0];
  if (nb) *nb = g_last_wgrad_cfg[1];
  if (tpg) *tpg = g_last_wgrad_cfg[2];
  if (nsplit_mode) *nsplit_mode = g_last_wgrad_cfg[3];
  if (bigx) *bigx = g_last_wgrad_cfg[4];
}

extern "C" int ubr_wgrad(const ubr_wgrad_desc* d, void* stream) {
  WPlan p{};
  int rc = wgrad_plan(d, &p);
  if (rc != UBR_OK) return rc;
  g_last_wgrad_cfg[0] = p.MA; g_last_wgrad_cfg[1] = p.NB; g_last_wgrad_cfg[2] = p.TPG; g_last_wgrad_cfg[3] = p.nsplit_mode; g_last_wgrad_cfg[4] = p.bigx; g_last_wgrad_cfg[5] = 0;
  const int esz = ubr_esize(d->dtype);
  UBR_CHECK(d->x.p && d->g.p && d->slabs, "ubr_wgrad: null tensor");
  UBR_CHECK(d->nsplit == p.nsplit, "ubr_wgrad: nsplit %d does not match plan %d", d->nsplit, p.nsplit);
  UBR_CHECK(ubr_aligned16(d->x.p) && ubr_aligned16(d->g.p), "ubr_wgrad: x/g must be 16-byte aligned");
  UBR_CHECK((d->x.sx * esz) % 16 == 0 && (d->x.sy * esz) % 16 == 0 && (d->x.sn * esz) % 16 == 0 &&
            (d->g.sx * esz) % 16 == 0 && (d->g.sy * esz) % 16 == 0 && (d->g.sn * esz) % 16 == 0,
            "ubr_wgrad: strides must keep 16-byte alignment");
  UBR_CHECK(d->x.sx >= d->Cin && d->g.sx >= d->Cout, "ubr_wgrad: pixel stride smaller than channel count");
  const bool xf = d->xf.scale != nullptr;
  UBR_CHECK(xf == (d->xf.shift != nullptr) && xf == (d->xf.lo != nullptr) && xf == (d->xf.sub != nullptr), "ubr_wgrad: xf needs sub, scale, shift and lo together");
  int dymin = 127, dxmin = 127;
  for (int t = 0; t < d->ntaps; ++t) { dymin = d->dy[t] < dymin ? d->dy[t] : dymin; dxmin = d->dx[t] < dxmin ? d->dx[t] : dxmin; }
  WgK k{};
  k.x = (const char*)d->x.p; k.x_sn = d->x.sn * esz; k.x_sy = d->x.sy * esz; k.x_sx = d->x.sx * esz;
  k.in_sub = d->xf.sub; k.in_scale = d->xf.scale; k.in_shift = d->xf.shift; k.in_lo = d->xf.lo;
  k.g = (const char*)d->g.p; k.g_sn = d->g.sn * esz; k.g_sy = d->g.sy * esz; k.g_sx = d->g.sx * esz;
  k.slabs = d->slabs;
  k.N = d->N; k.H = d->H; k.W = d->W; k.GH = d->GH; k.GW = d->GW; k.Cin = d->Cin; k.Cout_pad = d->Cout;
  k.ntaps = d->ntaps; k.S = d->S; k.iy0 = d->iy0; k.ix0 = d->ix0; k.dymin = dymin; k.dxmin = dxmin; k.HH = p.HH; k.HW = p.HW;
  k.TH = p.TH; k.tiles_x = p.tiles_x; k.tiles_y = p.tiles_y; k.ntiles = p.ntiles;
  k.pixbG = p.pixbG; k.pixbX = p.pixbX; k.x_off = p.x_off;
  k.hw_magic = (unsigned)((0x100000000ull + (unsigned)p.HW - 1) / (unsigned)p.HW);
  UBR_CHECK((long)d->H * k.x_sy < (1L << 31) && (long)d->GH * k.g_sy < (1L << 31), "ubr_wgrad: image too large for 32-bit offsets");
  k.x_sy32 = (int)k.x_sy; k.x_sx32 = (int)k.x_sx; k.g_sy32 = (int)k.g_sy; k.g_sx32 = (int)k.g_sx;
  k.n_cot = d->Cout / (p.MA * 16);
  {
    static const int rr = [] { const char* e = getenv("UBR_WGRAD_ROWREUSE"); return e ? atoi(e) : 1; }();
    const int rows_full = p.nsplit_mode ? (wgrad_tall(true, p.TPG, p.MA) ? 8 : 4) : wgrad_kth(false, p.TPG, p.MA, p.NB);
    bool grid3 = rr && d->ntaps == 9 && d->S == 1 && esz == 2 && p.TH == rows_full;
    for (int t = 0; grid3 && t < 9; ++t) grid3 = d->dy[t] == dymin + t / 3 && d->dx[t] == dxmin + t % 3;
    k.rowreuse = grid3 ? 1 : 0;
  }
#ifdef UBR_WGRAD_STAMPS     // diagnostic builds only
  { static const int dbg = [] { const char* e = getenv("UBR_WGRAD_DBG"); return e ? atoi(e) : 0; }(); k.dbg = dbg; }
  { static const char* sp = getenv("UBR_WGRAD_STAMP_PTR"); k.stamps = sp ? (unsigned long long*)strtoull(sp, nullptr, 0) : nullptr; }
#else
  k.dbg = 0; k.stamps = nullptr;
#endif
  for (int t = 0; t < d->ntaps; ++t) { k.dy[t] = d->dy[t]; k.dx[t] = d->dx[t]; }
  hipStream_t st = (hipStream_t)stream;
  switch (d->dtype) {
    case UBR_F32: return wdispatch<float>(k, p, st);
    case UBR_BF16: return wdispatch<bf16_t>(k, p, st);
    default: return wdispatch<f16_t>(k, p, st);
  }
}

// ---- batched form: the items travel in the kernel arguments (no device table to upload or keep alive; a launch tape stores them by value)
struct RedItem {
  const float* slabs; float* dst;
  int nsplit, ntaps, Cout_pad, Cin, Cout_valid, Cin_valid, accumulate, blocks;
  long sm, sk;
  uint8_t tapidx[UBR_MAX_TAPS];
};
struct RedBatch { int n, pad_; RedItem it[UBR_REDUCE_BATCH]; };
static_assert(sizeof(RedBatch) <= 3072, "kernel argument budget");

__global__ __launch_bounds__(256) void wgrad_reduce_batched_kernel(const RedBatch b) {
  __shared__ float part[4][64];
  int it = 0, blk = (int)blockIdx.x;
  while (it + 1 < b.n && blk >= b.it[it].blocks) { blk -= b.it[it].blocks; ++it; }     // (uniform: scalar loads from the argument segment)
  const RedItem& k = b.it[it];
  const long per = (long)k.ntaps * k.Cout_pad * k.Cin;
  if (k.nsplit <= 8) {
    for (long i = (long)blk * 256 + threadIdx.x; i < per; i += (long)k.blocks * 256) {
      const int ci = (int)(i % k.Cin);
      long r = i / k.Cin;
      const int co = (int)(r % k.Cout_pad);
      const int t = (int)(r / k.Cout_pad);
      if (co >= k.Cout_valid || ci >= k.Cin_valid) continue;
      float s = 0.f;
#pragma unroll 8
      for (int sp = 0; sp < k.nsplit; ++sp) s += k.slabs[(long)sp * per + i];
      float* d = k.dst + (long)co * k.sm + (long)ci * k.sk + k.tapidx[t];
      *d = k.accumulate ? (*d + s) : s;
    }
    return;
  }
  const int e = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const int q = (k.nsplit + 3) / 4;
  const int s0 = sg * q, s1 = min(s0 + q, k.nsplit);
  for (long base = (long)blk * 64; base < per; base += (long)k.blocks * 64) {
    const long i = base + e;
    float s = 0.f;
    if (i < per) {
      int sp = s0;
      for (; sp + 8 <= s1; sp += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = k.slabs[(long)(sp + u) * per + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
      }
      for (; sp < s1; ++sp) s += k.slabs[(long)sp * per + i];
    }
    part[sg][e] = s;
    __syncthreads();
    if (sg == 0 && i < per) {
      const int ci = (int)(i % k.Cin);
      long r = i / k.Cin;
      const int co = (int)(r % k.Cout_pad);
      const int t = (int)(r / k.Cout_pad);
      if (co < k.Cout_valid && ci < k.Cin_valid) {
        const float tot = (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
        float* d = k.dst + (long)co * k.sm + (long)ci * k.sk + k.tapidx[t];
        *d = k.accumulate ? (*d + tot) : tot;
      }
    }
    __syncthreads();
  }
}

extern "C" int ubr_wgrad_reduce_batched(const ubr_wgrad_reduce_item* items, int nitems, void* stream) {
  UBR_CHECK(items != nullptr && nitems >= 1 && nitems <= UBR_REDUCE_BATCH, "ubr_wgrad_reduce_batched: 1..%d items", UBR_REDUCE_BATCH);
  RedBatch b{};
  b.n = nitems;
  long total = 0;
  for (int i = 0; i < nitems; ++i) {
    const ubr_wgrad_reduce_item& s = items[i];
    UBR_CHECK(s.slabs && s.dst, "ubr_wgrad_reduce_batched: null pointer (item %d)", i);
    UBR_CHECK(s.nsplit >= 1 && s.nsplit <= kRedMaxSplit && s.ntaps >= 1 && s.ntaps <= UBR_MAX_TAPS && s.Cout_pad > 0 && s.Cin > 0 &&
              s.Cout_valid > 0 && s.Cout_valid <= s.Cout_pad && s.Cin_valid > 0 && s.Cin_valid <= s.Cin, "ubr_wgrad_reduce_batched: bad extents (item %d)", i);
    RedItem& r = b.it[i];
    r.slabs = s.slabs; r.dst = s.dst; r.nsplit = s.nsplit; r.ntaps = s.ntaps; r.Cout_pad = s.Cout_pad; r.Cin = s.Cin;
    r.Cout_valid = s.Cout_valid; r.Cin_valid = s.Cin_valid; r.accumulate = s.accumulate; r.sm = s.sm; r.sk = s.sk;
    for (int t = 0; t < s.ntaps; ++t) {
      UBR_CHECK(s.tapidx[t] >= 0 && s.tapidx[t] < 256, "ubr_wgrad_reduce_batched: tap index %d out of range", s.tapidx[t]);
      r.tapidx[t] = (uint8_t)s.tapidx[t];
    }
    const long per = (long)s.ntaps * s.Cout_pad * s.Cin;
    const bool flat = s.nsplit <= 8;
    long blocks = (per + (flat ? 255 : 63)) / (flat ? 256 : 64);      // (same grids as the single-item launches)
    if (blocks > 8192) blocks = 8192;
    r.blocks = (int)blocks;
    total += blocks;
  }
  ubr_launch(wgrad_reduce_batched_kernel, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, b);
  UBR_LAUNCH_CHECK("ubr_wgrad_reduce_batched");
  return UBR_OK;
}

extern "C" int ubr_wgrad_reduce(float* slabs, int nsplit, int ntaps, int Cout_pad, int Cin,
                                int Cout_valid, int Cin_valid, float* dst, int64_t sm, int64_t sk,
                                const int32_t* tapidx_host, int accumulate, void* stream) {
  UBR_CHECK(slabs && dst && tapidx_host, "ubr_wgrad_reduce: null pointer");
  UBR_CHECK(nsplit >= 1 && nsplit <= kRedMaxSplit && ntaps >= 1 && ntaps <= UBR_MAX_TAPS && Cout_pad > 0 && Cin > 0 &&
            Cout_valid > 0 && Cout_valid <= Cout_pad && Cin_valid > 0 && Cin_valid <= Cin, "ubr_wgrad_reduce: bad extents");
  const long per = (long)ntaps * Cout_pad * Cin;
  const bool flat = nsplit <= 8;
  int blocks = (int)((per + (flat ? 255 : 63)) / (flat ? 256 : 64));
  if (blocks > 8192) blocks = 8192;
  hipStream_t st = (hipStream_t)stream;
  RedK k{};
  k.slabs = slabs; k.dst = dst; k.nsplit = nsplit; k.ntaps = ntaps; k.Cout_pad = Cout_pad; k.Cin = Cin;
  k.Cout_valid = Cout_valid; k.Cin_valid = Cin_valid; k.accumulate = accumulate; k.sm = sm; k.sk = sk; k.slab_stride = per;
  for (int t = 0; t < ntaps; ++t) k.tapidx[t] = tapidx_host[t];
  if (flat) ubr_launch(wgrad_reduce_flat_kernel, dim3(blocks), dim3(256), 0, st, k);
  else ubr_launch(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, k);
  UBR_LAUNCH_CHECK("ubr_wgrad_reduce");
  return UBR_OK;
}
