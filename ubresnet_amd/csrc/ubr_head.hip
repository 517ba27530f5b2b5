// Boundary kernels of the U-ResNet path: the NCHW fp32 stem convolution (conv1 7x7, Cin 1..4,
// models/ub_uresnet.py:41,94) and its weight gradient, LogSoftmax backward with NCHW->NHWC
// re-layout (models/ub_uresnet.py:143), PixelWiseNLLLoss forward/backward
// (training/pixelwise_nllloss.py:41-61) and the one-pass confusion matrix behind accuracy()
// (training/train_ubresnet2018_wlarcv2.py:509-566).
#include <math.h>
#include "ubr_common.h"
#include "ubr_host.h"
#ifndef UBR_MAX_TILES
#define UBR_MAX_TILES 64
#endif

namespace {

// ------------------------------------------------------------------------------------------
// stem forward: one thread per output pixel of a 16x16 tile, all Cout channels in 16-ch chunks.
// The only layer whose input is sparse (1-3 % non-zero LArTPC crops): an all-zero halo tile
// contributes only the bias, so it skips the 49-tap loop.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* x, int N, int Cin, int H, int W, const float* wgt, const float* bias,
                                                       int Cout, char* y, long y_sn, long y_sy, long y_sx, double* stats,
                                                       int tiles_x, int tiles_y) {
  constexpr int ESZ = 16 / ET<T>::CPU;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* xt = reinterpret_cast<float*>(smem);                 // [Cin][22][22]
  float* wl = xt + Cin * 484;                                  // [Cin*49][Cout]
  float* red = wl + Cin * 49 * Cout;                           // [4][Cout][2]
  int& any_nz = *reinterpret_cast<int*>(red + 4 * Cout * 2);   // all LDS in the dynamic region (keeps 16-B alignment)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int t = blockIdx.x;
  const int tx0 = (t % tiles_x) * 16; t /= tiles_x;
  const int ty0 = (t % tiles_y) * 16;
  const int n = t / tiles_y;
  if (tid == 0) any_nz = 0;
  __syncthreads();
  int nz = 0;
  for (int i = tid; i < Cin * 484; i += 256) {
    const int ci = i / 484, r = i % 484;
    const int iy = ty0 - 3 + r / 22, ix = tx0 - 3 + r % 22;
    float v = 0.f;
    if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = x[(((long)n * Cin + ci) * H + iy) * W + ix];
    xt[i] = v;
    nz |= (v != 0.f);
  }
  if (nz) any_nz = 1;
  for (int i = tid; i < Cin * 49 * Cout; i += 256) {
    const int co = i % Cout, r = i / Cout;   // r = ci*49 + tap
    wl[i] = wgt[(long)co * Cin * 49 + r];
  }
  __syncthreads();
  const bool dense = any_nz != 0;
  const int py = tid >> 4, px = tid & 15;
  const int oy = ty0 + py, ox = tx0 + px;
  const bool valid = oy < H && ox < W;
  for (int c0 = 0; c0 < Cout; c0 += 16) {
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = bias ? bias[c0 + c] : 0.f;
    if (dense) {
      for (int ci = 0; ci < Cin; ++ci) {
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
          for (int kx = 0; kx < 7; ++kx) {
            const float v = xt[ci * 484 + (py + ky) * 22 + px + kx];
            const float4* wp = reinterpret_cast<const float4*>(wl + ((ci * 49 + ky * 7 + kx) * Cout + c0));
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
              const float4 w4 = wp[c4];
              acc[c4 * 4 + 0] = fmaf(v, w4.x, acc[c4 * 4 + 0]);
              acc[c4 * 4 + 1] = fmaf(v, w4.y, acc[c4 * 4 + 1]);
              acc[c4 * 4 + 2] = fmaf(v, w4.z, acc[c4 * 4 + 2]);
              acc[c4 * 4 + 3] = fmaf(v, w4.w, acc[c4 * 4 + 3]);
            }
          }
      }
    }
    if (valid) {
      char* dst = y + (long)n * y_sn + (long)oy * y_sy + (long)ox * y_sx + (long)c0 * ESZ;
#pragma unroll
      for (int u = 0; u < 16 / ET<T>::CPU; ++u) stg16(dst + u * 16, ET<T>::pack(acc + u * ET<T>::CPU));
    }
    if (stats != nullptr) {
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const float v = valid ? acc[c] : 0.f;
        const float a = wave_sum64(v), b = wave_sum64(v * v);
        if (lane == 0) { red[(wave * Cout + c0 + c) * 2] = a; red[(wave * Cout + c0 + c) * 2 + 1] = b; }
      }
    }
  }
  if (stats != nullptr) {
    __syncthreads();
    if (tid < Cout) {
      double a = 0.0, b = 0.0;
      for (int w = 0; w < 4; ++w) { a += (double)red[(w * Cout + tid) * 2]; b += (double)red[(w * Cout + tid) * 2 + 1]; }
      double* st = stats + (size_t)(blockIdx.x % UBR_STAT_SLOTS) * 2 * Cout;
      atomicAdd(&st[tid], a);
      atomicAdd(&st[Cout + tid], b);
    }
  }
}

// ------------------------------------------------------------------------------------------
// stem weight gradient: thread = (cout, tap slot); loops the pixels of a 16x16 tile held in LDS.
// ------------------------------------------------------------------------------------------
template <typename T, int COUT>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* x, int N, int Cin, int H, int W, const char* g, long g_sn, long g_sy,
                                                         long g_sx, float* partial, int tiles_x, int tiles_y, int ntiles) {
  constexpr int CPU = ET<T>::CPU;
  constexpr int ESZ = 16 / CPU;
  constexpr int NSLOT = 256 / COUT;
  constexpr int TPS = (49 + NSLOT - 1) / NSLOT;
  constexpr int MAXCIN = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* gl = reinterpret_cast<float*>(smem);   // [256][COUT]
  float* xt = gl + 256 * COUT;                  // [Cin][22][22]
  const int tid = threadIdx.x;
  const int co = tid % COUT, slot = tid / COUT;
  float acc[MAXCIN][TPS];
  float accb = 0.f;
#pragma unroll
  for (int ci = 0; ci < MAXCIN; ++ci)
#pragma unroll
    for (int j = 0; j < TPS; ++j) acc[ci][j] = 0.f;

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int t = tile;
    const int tx0 = (t % tiles_x) * 16; t /= tiles_x;
    const int ty0 = (t % tiles_y) * 16;
    const int n = t / tiles_y;
    __syncthreads();
    for (int i = tid; i < 256 * (COUT / CPU); i += 256) {
      const int u = i % (COUT / CPU), p = i / (COUT / CPU);
      const int oy = ty0 + (p >> 4), ox = tx0 + (p & 15);
      float f[CPU];
      if (oy < H && ox < W) {
        ET<T>::unpack(ldg16(g + (long)n * g_sn + (long)oy * g_sy + (long)ox * g_sx + (long)u * 16), f);
      } else {
#pragma unroll
        for (int e = 0; e < CPU; ++e) f[e] = 0.f;
      }
#pragma unroll
      for (int e = 0; e < CPU; ++e) gl[p * COUT + u * CPU + e] = f[e];
    }
    for (int i = tid; i < Cin * 484; i += 256) {
      const int ci = i / 484, r = i % 484;
      const int iy = ty0 - 3 + r / 22, ix = tx0 - 3 + r % 22;
      float v = 0.f;
      if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = x[(((long)n * Cin + ci) * H + iy) * W + ix];
      xt[i] = v;
    }
    __syncthreads();
    for (int p = 0; p < 256; ++p) {
      const float gv = gl[p * COUT + co];
      const int py = p >> 4, px = p & 15;
      if (slot == 0) accb += gv;
#pragma unroll
      for (int j = 0; j < TPS; ++j) {
        const int tap = slot + j * NSLOT;
        if (tap < 49) {
          const int off = (py + tap / 7) * 22 + px + tap % 7;
#pragma unroll
          for (int ci = 0; ci < MAXCIN; ++ci)
            if (ci < Cin) acc[ci][j] = fmaf(gv, xt[ci * 484 + off], acc[ci][j]);
        }
      }
    }
  }
  // partial[wg][co][ci][tap] and partial bias at [wg][COUT*Cin*49 + co]
  float* out = partial + (long)blockIdx.x * (COUT * Cin * 49 + COUT);
#pragma unroll
  for (int j = 0; j < TPS; ++j) {
    const int tap = slot + j * NSLOT;
    if (tap < 49) {
#pragma unroll
      for (int ci = 0; ci < MAXCIN; ++ci)
        if (ci < Cin) out[(co * Cin + ci) * 49 + tap] = acc[ci][j];
    }
  }
  if (slot == 0) out[COUT * Cin * 49 + co] = accb;
}

__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float* partial, int nwg, int nw, int Cout, float* dweight, float* dbias, int accumulate) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nw + Cout) return;
  float s = 0.f;
  for (int w = 0; w < nwg; ++w) s += partial[(long)w * (nw + Cout) + i];
  if (i < nw) dweight[i] = accumulate ? dweight[i] + s : s;
  else if (dbias != nullptr) dbias[i - nw] = accumulate ? dbias[i - nw] + s : s;
}

// ------------------------------------------------------------------------------------------
// stem expansion: NCHW fp32 image -> NHWC T with 16 channels per image plane, channel kx (0..6)
// holding the image shifted by kx-3 columns.  A 7x7 convolution over Cin planes then IS a
// 7-tap (vertical) convolution over 16*Cin channels, which runs on the MFMA implicit-GEMM and
// weight-gradient kernels (K = 7*16 per plane instead of 49; the zero channels cost no HBM).
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void stem_expand_kernel(const float* x, int N, int Cin, int H, int W, char* out, long out_ps) {
  constexpr int CPU = ET<T>::CPU;
  constexpr int ESZ = 16 / CPU;
  const long hw = (long)H * W;
  const long total = (long)N * Cin * hw;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ci = (int)(i % Cin);
    const long p = i / Cin;                 // pixel index over N*H*W
    const long n = p / hw, r = p - n * hw;
    const int xx = (int)(r % W);
    const float* row = x + ((n * Cin + ci) * hw + (r - xx));
    float f[16];
#pragma unroll
    for (int kx = 0; kx < 7; ++kx) {
      const int sx = xx + kx - 3;
      f[kx] = ((unsigned)sx < (unsigned)W) ? row[sx] : 0.f;
    }
#pragma unroll
    for (int c = 7; c < 16; ++c) f[c] = 0.f;
    char* dst = out + (p * out_ps + (long)ci * 16) * ESZ;
#pragma unroll
    for (int u = 0; u < 16 / CPU; ++u) stg16(dst + u * 16, ET<T>::pack(f + u * CPU));
  }
}

// ------------------------------------------------------------------------------------------
// LogSoftmax backward + re-layout
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void logsoftmax_bwd_kernel(long npix_img, int N, int C, const float* g, const float* lp, char* out, long out_ps) {
  constexpr int CPU = ET<T>::CPU;
  constexpr int ESZ = 16 / CPU;
  const long total = (long)N * npix_img;
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < total; p += (long)gridDim.x * 256) {
    const long n = p / npix_img, r = p - n * npix_img;
    float gv[16], s = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      gv[c] = c < C ? g[((long)n * C + c) * npix_img + r] : 0.f;
      s += gv[c];
    }
    float o[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) o[c] = c < C ? gv[c] - expf(lp[((long)n * C + c) * npix_img + r]) * s : 0.f;
    char* dst = out + p * out_ps * ESZ;
#pragma unroll
    for (int u = 0; u < 16 / CPU; ++u) stg16(dst + u * 16, ET<T>::pack(o + u * CPU));
  }
}

// ------------------------------------------------------------------------------------------
// PixelWiseNLLLoss
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nll_fwd_kernel(const float* pred, const long long* target, const float* pw, const float* cw,
                                                      int N, int C, long hw, long long ignore_index, double* acc, unsigned long long* bad) {
  __shared__ double part[4];
  const long total = (long)N * hw;
  double s = 0.0;
  unsigned nbad = 0u;
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < total; p += (long)gridDim.x * 256) {
    const long long t = target[p];
    if (t != ignore_index && (t < 0 || t >= C)) ++nbad;      // F.nll_loss device-asserts on these (reference behaviour)
    if (t != ignore_index && t >= 0 && t < C) {
      const long n = p / hw, r = p - n * hw;
      const float w = cw ? cw[t] : 1.f;
      // F.nll_loss(reduction='none') gives -x*w (fp32), then * pixelweights (fp32)
      s += (double)((-pred[((long)n * C + t) * hw + r] * w) * pw[p]);
    }
  }
  s = wave_sum64d(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(acc + (blockIdx.x % UBR_STAT_SLOTS), part[0] + part[1] + part[2] + part[3]);
  if (bad != nullptr && nbad) atomicAdd(bad, (unsigned long long)nbad);
}

__global__ __launch_bounds__(256) void nll_bwd_kernel(const float* gloss, const long long* target, const float* pw, const float* cw,
                                                      int N, int C, long hw, long long ignore_index, float* gpred) {
  const long total = (long)N * hw;
  const float gl = *gloss / (float)total;
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < total; p += (long)gridDim.x * 256) {
    const long long t = target[p];
    const long n = p / hw, r = p - n * hw;
    const bool ok = t != ignore_index && t >= 0 && t < C;
    const float gv = ok ? -gl * pw[p] * (cw ? cw[t] : 1.f) : 0.f;
    for (int c = 0; c < C; ++c) gpred[((long)n * C + c) * hw + r] = (ok && c == (int)t) ? gv : 0.f;
  }
}

__global__ __launch_bounds__(256) void confusion_kernel(const float* lp, const long long* target, int N, int C, long hw, unsigned long long* cm) {
  extern __shared__ unsigned int lcm[];
  for (int i = threadIdx.x; i < C * C; i += 256) lcm[i] = 0u;
  __syncthreads();
  const long total = (long)N * hw;
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < total; p += (long)gridDim.x * 256) {
    const long n = p / hw, r = p - n * hw;
    int best = 0; float bv = lp[((long)n * C) * hw + r];
    for (int c = 1; c < C; ++c) { const float v = lp[((long)n * C + c) * hw + r]; if (v > bv) { bv = v; best = c; } }
    const long long t = target[p];
    if (t >= 0 && t < C) atomicAdd(&lcm[(int)t * C + best], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C * C; i += 256) if (lcm[i]) atomicAdd(&cm[i], (unsigned long long)lcm[i]);
}

}  // namespace

#define UBR_DT_SWITCH(dtype, CALL)                    \
  switch (dtype) {                                    \
    case UBR_F32: { typedef float TT; CALL; } break;  \
    case UBR_BF16: { typedef bf16_t TT; CALL; } break;\
    default: { typedef f16_t TT; CALL; } break;       \
  }

extern "C" int ubr_stem_forward(int dtype, const float* x_nchw, int N, int Cin, int H, int W,
                                const float* weight, const float* bias, int Cout,
                                ubr_tensor y, double* stats, void* stream) {
  UBR_CHECK(ubr_dtype_ok(dtype), "ubr_stem_forward: bad dtype");
  UBR_CHECK(x_nchw && weight && y.p, "ubr_stem_forward: null pointer");
  UBR_CHECK(N > 0 && H > 0 && W > 0 && Cin >= 1 && Cin <= 4, "ubr_stem_forward: bad extents (Cin must be 1..4, got %d)", Cin);
  UBR_CHECK(Cout % 16 == 0 && Cout >= 16 && Cout <= 64, "ubr_stem_forward: Cout=%d must be 16, 32, 48 or 64", Cout);
  const int esz = ubr_esize(dtype);
  UBR_CHECK(ubr_aligned16(y.p) && (y.sx * esz) % 16 == 0 && (y.sy * esz) % 16 == 0 && (y.sn * esz) % 16 == 0 && y.sx >= Cout,
            "ubr_stem_forward: output view must be 16-byte aligned with pixel stride >= Cout");
  const int tiles_x = ubr_cdiv(W, 16), tiles_y = ubr_cdiv(H, 16);
  const size_t lds = ((size_t)Cin * 484 + (size_t)Cin * 49 * Cout + 4 * Cout * 2) * sizeof(float) + 16;
  UBR_DT_SWITCH(dtype, ubr_launch(stem_fwd_kernel<TT>, dim3(tiles_x * tiles_y * N), dim3(256), lds, (hipStream_t)stream,
                                          x_nchw, N, Cin, H, W, weight, bias, Cout, (char*)y.p, (long)y.sn * esz, (long)y.sy * esz,
                                          (long)y.sx * esz, stats, tiles_x, tiles_y));
  UBR_LAUNCH_CHECK("ubr_stem_forward");
  return UBR_OK;
}

extern "C" int ubr_stem_expand(int dtype, const float* x_nchw, int N, int Cin, int H, int W, void* out, int64_t out_ps, void* stream) {
  UBR_CHECK(ubr_dtype_ok(dtype), "ubr_stem_expand: bad dtype");
  UBR_CHECK(x_nchw && out && N > 0 && H > 0 && W > 0 && Cin >= 1 && Cin <= 8, "ubr_stem_expand: bad arguments");
  UBR_CHECK(ubr_aligned16(out) && out_ps >= 16 * Cin && (out_ps * ubr_esize(dtype)) % 16 == 0, "ubr_stem_expand: output pixel stride must be >= 16*Cin and 16-byte aligned");
  long blocks = ((long)N * Cin * H * W + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  UBR_DT_SWITCH(dtype, ubr_launch(stem_expand_kernel<TT>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x_nchw, N, Cin, H, W,
                                          (char*)out, (long)out_ps));
  UBR_LAUNCH_CHECK("ubr_stem_expand");
  return UBR_OK;
}

static int stem_wgrad_nwg(int N, int H, int W) {
  const int ntiles = ubr_cdiv(W, 16) * ubr_cdiv(H, 16) * N;
  return ntiles < 1024 ? ntiles : 1024;
}
extern "C" int64_t ubr_stem_wgrad_workspace(int N, int Cin, int H, int W, int Cout) {
  return (int64_t)stem_wgrad_nwg(N, H, W) * (Cout * Cin * 49 + Cout) * (int64_t)sizeof(float);
}
extern "C" int ubr_stem_wgrad(int dtype, const float* x_nchw, int N, int Cin, int H, int W, ubr_tensor g, int Cout,
                              float* partial, int64_t partial_bytes, float* dweight, float* dbias,
                              int accumulate, void* stream) {
  UBR_CHECK(ubr_dtype_ok(dtype), "ubr_stem_wgrad: bad dtype");
  UBR_CHECK(x_nchw && g.p && partial && dweight, "ubr_stem_wgrad: null pointer");
  UBR_CHECK(N > 0 && H > 0 && W > 0 && Cin >= 1 && Cin <= 4, "ubr_stem_wgrad: bad extents");
  UBR_CHECK(Cout == 16 || Cout == 32 || Cout == 64, "ubr_stem_wgrad: Cout=%d must be 16, 32 or 64", Cout);
  UBR_CHECK(partial_bytes >= ubr_stem_wgrad_workspace(N, Cin, H, W, Cout), "ubr_stem_wgrad: workspace too small");
  const int esz = ubr_esize(dtype);
  UBR_CHECK(ubr_aligned16(g.p) && (g.sx * esz) % 16 == 0 && (g.sy * esz) % 16 == 0 && (g.sn * esz) % 16 == 0 && g.sx >= Cout,
            "ubr_stem_wgrad: gradient view must be 16-byte aligned with pixel stride >= Cout");
  const int tiles_x = ubr_cdiv(W, 16), tiles_y = ubr_cdiv(H, 16);
  const int ntiles = tiles_x * tiles_y * N;
  const int nwg = stem_wgrad_nwg(N, H, W);
  const size_t lds = ((size_t)256 * Cout + (size_t)Cin * 484) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
#define UBR_SW(CO) UBR_DT_SWITCH(dtype, ubr_launch((stem_wgrad_kernel<TT, CO>), dim3(nwg), dim3(256), lds, st, x_nchw, N, Cin, H, W, \
    (const char*)g.p, (long)g.sn * esz, (long)g.sy * esz, (long)g.sx * esz, partial, tiles_x, tiles_y, ntiles))
  if (lds > 64 * 1024) { ubr_set_error("ubr_stem_wgrad: LDS too large"); return UBR_EINVAL; }
  if (Cout == 16) { UBR_SW(16); } else if (Cout == 32) { UBR_SW(32); } else { UBR_SW(64); }
#undef UBR_SW
  UBR_LAUNCH_CHECK("ubr_stem_wgrad");
  const int nw = Cout * Cin * 49;
  ubr_launch(stem_wgrad_reduce_kernel, dim3(ubr_cdiv(nw + Cout, 256)), dim3(256), 0, st, partial, nwg, nw, Cout, dweight, dbias, accumulate);
  UBR_LAUNCH_CHECK("ubr_stem_wgrad(reduce)");
  return UBR_OK;
}

extern "C" int ubr_logsoftmax_bwd(int dtype, int N, int C, int H, int W, const float* g_logp_nchw, const float* logp_nchw,
                                  void* g_logits, int64_t gl_ps, void* stream) {
  UBR_CHECK(ubr_dtype_ok(dtype), "ubr_logsoftmax_bwd: bad dtype");
  UBR_CHECK(g_logp_nchw && logp_nchw && g_logits, "ubr_logsoftmax_bwd: null pointer");
  UBR_CHECK(N > 0 && H > 0 && W > 0 && C >= 1 && C <= 16, "ubr_logsoftmax_bwd: C=%d must be 1..16", C);
  UBR_CHECK(ubr_aligned16(g_logits) && gl_ps >= 16 && (gl_ps * ubr_esize(dtype)) % 16 == 0, "ubr_logsoftmax_bwd: output pixel stride must be >= 16 and 16-byte aligned");
  const long hw = (long)H * W;
  long blocks = ((long)N * hw + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  UBR_DT_SWITCH(dtype, ubr_launch(logsoftmax_bwd_kernel<TT>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, hw, N, C,
                                          g_logp_nchw, logp_nchw, (char*)g_logits, (long)gl_ps));
  UBR_LAUNCH_CHECK("ubr_logsoftmax_bwd");
  return UBR_OK;
}

extern "C" int ubr_pixelwise_nll_fwd(const float* predict_nchw, const int64_t* target, const float* pixelweights,
                                     const float* classw, int N, int C, int H, int W, int64_t ignore_index,
                                     double* acc, unsigned long long* bad_labels, void* stream) {
  UBR_CHECK(predict_nchw && target && pixelweights && acc, "ubr_pixelwise_nll_fwd: null pointer");
  UBR_CHECK(N > 0 && C > 0 && H > 0 && W > 0, "ubr_pixelwise_nll_fwd: bad extents");
  const long hw = (long)H * W;
  long blocks = ((long)N * hw + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  ubr_launch(nll_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, predict_nchw, (const long long*)target,
                     pixelweights, classw, N, C, hw, (long long)ignore_index, acc, bad_labels);
  UBR_LAUNCH_CHECK("ubr_pixelwise_nll_fwd");
  return UBR_OK;
}
extern "C" int ubr_pixelwise_nll_bwd(const float* g_loss, const int64_t* target, const float* pixelweights,
                                     const float* classw, int N, int C, int H, int W, int64_t ignore_index,
                                     float* g_predict_nchw, void* stream) {
  UBR_CHECK(g_loss && target && pixelweights && g_predict_nchw, "ubr_pixelwise_nll_bwd: null pointer");
  UBR_CHECK(N > 0 && C > 0 && H > 0 && W > 0, "ubr_pixelwise_nll_bwd: bad extents");
  const long hw = (long)H * W;
  long blocks = ((long)N * hw + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  ubr_launch(nll_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g_loss, (const long long*)target,
                     pixelweights, classw, N, C, hw, (long long)ignore_index, g_predict_nchw);
  UBR_LAUNCH_CHECK("ubr_pixelwise_nll_bwd");
  return UBR_OK;
}
extern "C" int ubr_confusion(const float* logp_nchw, const int64_t* target, int N, int C, int H, int W,
                             unsigned long long* cm, void* stream) {
  UBR_CHECK(logp_nchw && target && cm, "ubr_confusion: null pointer");
  UBR_CHECK(N > 0 && C > 0 && C <= 64 && H > 0 && W > 0, "ubr_confusion: bad extents");
  const long hw = (long)H * W;
  long blocks = ((long)N * hw + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  ubr_launch(confusion_kernel, dim3((unsigned)blocks), dim3(256), (size_t)C * C * sizeof(unsigned), (hipStream_t)stream,
                     logp_nchw, (const long long*)target, N, C, hw, cm);
  UBR_LAUNCH_CHECK("ubr_confusion");
  return UBR_OK;
}

// ------------------------------------------------------------------------------------------
// whole-view tiling (shape of deploy/run_ubresnet_wholeview.py:191-277): crop overlapping tiles out of
// the plane images into a batch, and stitch per-tile class scores back, each output pixel taken
// from exactly one tile (its keep-window), so the result does not depend on launch order.
// ------------------------------------------------------------------------------------------
struct TileK {
  int ntiles, th, tw, rows, cols, C;
  int plane[UBR_MAX_TILES], r0[UBR_MAX_TILES], c0[UBR_MAX_TILES];
  int kr0[UBR_MAX_TILES], kr1[UBR_MAX_TILES], kc0[UBR_MAX_TILES], kc1[UBR_MAX_TILES];   // keep window, tile coordinates
};
__global__ __launch_bounds__(256) void crop_tiles_kernel(const float* view, float* out, const TileK k) {
  const int t = blockIdx.y;
  const long n = (long)k.th * k.tw;
  const float* src = view + (long)k.plane[t] * k.rows * k.cols;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int y = (int)(i / k.tw), x = (int)(i % k.tw);
    const int sy = k.r0[t] + y, sx = k.c0[t] + x;
    out[(long)t * n + i] = (sy < k.rows && sx < k.cols) ? src[(long)sy * k.cols + sx] : 0.f;
  }
}
__global__ __launch_bounds__(256) void stitch_tiles_kernel(const float* scores, float* out, const TileK k) {
  const int t = blockIdx.y;
  const int kh = k.kr1[t] - k.kr0[t], kw = k.kc1[t] - k.kc0[t];
  const long n = (long)k.C * kh * kw;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int x = (int)(i % kw);
    const long r = i / kw;
    const int y = (int)(r % kh), c = (int)(r / kh);
    const int ty = k.kr0[t] + y, tx = k.kc0[t] + x;
    const int oy = k.r0[t] + ty, ox = k.c0[t] + tx;
    if (oy < k.rows && ox < k.cols)
      out[(((long)k.plane[t] * k.C + c) * k.rows + oy) * k.cols + ox] = scores[(((long)t * k.C + c) * k.th + ty) * k.tw + tx];
  }
}
static int fill_tilek(TileK& k, const int32_t* desc, int ntiles, int th, int tw, int rows, int cols, int C, int P, bool keep) {
  UBR_CHECK(desc && ntiles >= 1 && ntiles <= UBR_MAX_TILES && th > 0 && tw > 0 && rows > 0 && cols > 0 && C >= 1 && P >= 1,
            "ubr tiles: bad arguments (ntiles=%d)", ntiles);
  k.ntiles = ntiles; k.th = th; k.tw = tw; k.rows = rows; k.cols = cols; k.C = C;
  for (int t = 0; t < ntiles; ++t) {
    const int32_t* d = desc + 7 * t;
    UBR_CHECK(d[0] >= 0 && d[0] < P && d[1] >= 0 && d[2] >= 0 && d[1] < rows && d[2] < cols, "ubr tiles: tile %d origin out of range", t);
    k.plane[t] = d[0]; k.r0[t] = d[1]; k.c0[t] = d[2];
    if (keep) {
      UBR_CHECK(d[3] >= 0 && d[3] <= d[4] && d[4] <= th && d[5] >= 0 && d[5] <= d[6] && d[6] <= tw, "ubr tiles: tile %d keep window out of range", t);
      k.kr0[t] = d[3]; k.kr1[t] = d[4]; k.kc0[t] = d[5]; k.kc1[t] = d[6];
    }
  }
  return UBR_OK;
}
extern "C" int ubr_crop_tiles(const float* view, int P, int rows, int cols, const int32_t* tile_desc_host, int ntiles,
                              int th, int tw, float* out, void* stream) {
  UBR_CHECK(view && out, "ubr_crop_tiles: null pointer");
  TileK k{};
  int rc = fill_tilek(k, tile_desc_host, ntiles, th, tw, rows, cols, 1, P, false);
  if (rc != UBR_OK) return rc;
  ubr_launch(crop_tiles_kernel, dim3(ubr_cdiv(th * tw, 256 * 8), ntiles), dim3(256), 0, (hipStream_t)stream, view, out, k);
  UBR_LAUNCH_CHECK("ubr_crop_tiles");
  return UBR_OK;
}
extern "C" int ubr_stitch_tiles(const float* scores, int C, int th, int tw, const int32_t* tile_desc_host, int ntiles,
                                float* out, int P, int rows, int cols, void* stream) {
  UBR_CHECK(scores && out, "ubr_stitch_tiles: null pointer");
  TileK k{};
  int rc = fill_tilek(k, tile_desc_host, ntiles, th, tw, rows, cols, C, P, true);
  if (rc != UBR_OK) return rc;
  ubr_launch(stitch_tiles_kernel, dim3(ubr_cdiv(C * th * tw, 256 * 8), ntiles), dim3(256), 0, (hipStream_t)stream, scores, out, k);
  UBR_LAUNCH_CHECK("ubr_stitch_tiles");
  return UBR_OK;
}

// ------------------------------------------------------------------------------------------
// Flat optimizers: every parameter of the network lives in one fp32 buffer laid out like the flat gradient buffer
// the backward pass fills, so a whole optimizer step is ONE streaming kernel (reference: torch.optim.Adam(lr 1e-5,
// weight_decay 1e-4), training/train_ubresnet2018_wlarcv2.py:155-157; SGD momentum 0.9, train_ubresnet2018_wlarcv1.py:127-129).
// Arithmetic follows torch.optim's single-tensor formulas in fp32 (L2 weight decay added to the gradient).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n4, float lr, float b1, float b2, float eps, float wd,
                                                   float bc1, float sqrt_bc2, float gscale) {
  const float step_size = lr / bc1;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 P = reinterpret_cast<float4*>(p)[i], G = reinterpret_cast<const float4*>(g)[i];
    float4 M = reinterpret_cast<float4*>(m)[i], V = reinterpret_cast<float4*>(v)[i];
    float pp[4] = {P.x, P.y, P.z, P.w}, gg[4] = {G.x, G.y, G.z, G.w}, mm[4] = {M.x, M.y, M.z, M.w}, vv[4] = {V.x, V.y, V.z, V.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float gr = gg[e] * gscale;
      gr = fmaf(wd, pp[e], gr);                                  // grad.add(param, alpha=weight_decay)
      mm[e] = mm[e] + (1.f - b1) * (gr - mm[e]);                 // exp_avg.lerp_(grad, 1 - beta1)
      vv[e] = b2 * vv[e] + (1.f - b2) * gr * gr;                 // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
      const float denom = sqrtf(vv[e]) / sqrt_bc2 + eps;
      pp[e] = pp[e] - step_size * (mm[e] / denom);               // param.addcdiv_(exp_avg, denom, value=-step_size)
    }
    reinterpret_cast<float4*>(p)[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
    reinterpret_cast<float4*>(m)[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
    reinterpret_cast<float4*>(v)[i] = make_float4(vv[0], vv[1], vv[2], vv[3]);
  }
}
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, long n4,
                                                  float lr, float momentum, float dampening, float wd, int nesterov, int first,
                                                  float gscale) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 P = reinterpret_cast<float4*>(p)[i], G = reinterpret_cast<const float4*>(g)[i];
    float4 B = make_float4(0.f, 0.f, 0.f, 0.f);
    if (buf != nullptr && !first) B = reinterpret_cast<float4*>(buf)[i];
    float pp[4] = {P.x, P.y, P.z, P.w}, gg[4] = {G.x, G.y, G.z, G.w}, bb[4] = {B.x, B.y, B.z, B.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float gr = fmaf(wd, pp[e], gg[e] * gscale);
      if (buf != nullptr) {
        bb[e] = first ? gr : momentum * bb[e] + (1.f - dampening) * gr;   // torch.optim.SGD: first step clones the gradient
        gr = nesterov ? fmaf(momentum, bb[e], gr) : bb[e];
      }
      pp[e] = pp[e] - lr * gr;
    }
    reinterpret_cast<float4*>(p)[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
    if (buf != nullptr) reinterpret_cast<float4*>(buf)[i] = make_float4(bb[0], bb[1], bb[2], bb[3]);
  }
}
static int opt_blocks(long n4) {
  long b = (n4 + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}
extern "C" int ubr_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int64_t step, float grad_scale, void* stream) {
  UBR_CHECK(param && grad && exp_avg && exp_avg_sq && n > 0 && n % 4 == 0 && step >= 1, "ubr_adam_step: bad arguments (n must be a multiple of 4)");
  UBR_CHECK(ubr_aligned16(param) && ubr_aligned16(grad) && ubr_aligned16(exp_avg) && ubr_aligned16(exp_avg_sq), "ubr_adam_step: buffers must be 16-byte aligned");
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  ubr_launch(adam_kernel, dim3(opt_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, (long)(n / 4),
                     lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), grad_scale);
  UBR_LAUNCH_CHECK("ubr_adam_step");
  return UBR_OK;
}
extern "C" int ubr_sgd_step(float* param, const float* grad, float* momentum_buf, int64_t n, float lr, float momentum, float dampening,
                            float weight_decay, int nesterov, int first_step, float grad_scale, void* stream) {
  UBR_CHECK(param && grad && n > 0 && n % 4 == 0, "ubr_sgd_step: bad arguments (n must be a multiple of 4)");
  UBR_CHECK((momentum == 0.f) == (momentum_buf == nullptr), "ubr_sgd_step: momentum buffer iff momentum != 0");
  UBR_CHECK(ubr_aligned16(param) && ubr_aligned16(grad) && (!momentum_buf || ubr_aligned16(momentum_buf)), "ubr_sgd_step: buffers must be 16-byte aligned");
  ubr_launch(sgd_kernel, dim3(opt_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, param, grad, momentum_buf, (long)(n / 4), lr, momentum,
                     dampening, weight_decay, nesterov, first_step, grad_scale);
  UBR_LAUNCH_CHECK("ubr_sgd_step");
  return UBR_OK;
}

// ------------------------------------------------------------------------------------------
// error string / version
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
extern "C" void ubr_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* ubr_last_error(void) { return g_err; }
extern "C" int ubr_version(void) { return 1; }
