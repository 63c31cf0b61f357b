// libaomhip -- the RD helpers of SURVEY 8(f)-3 as batched device calls, gfx950:
//   aom_sse / aom_highbd_sse                              aom_dsp/sse.c:19-53
//   aom_hadamard_{4x4,8x8,16x16,32x32}, aom_hadamard_lp_{8x8,16x16},
//   aom_highbd_hadamard_{8x8,16x16,32x32}                 aom_dsp/avg.c:110-514
//   aom_satd / aom_satd_lp                                aom_dsp/avg.c:517-533
//   av1_txb_init_levels                                   av1/encoder/encodetxb.c:238-254
// One wavefront per block / candidate.  These are small integer kernels whose cost is the launch; what matters is
// that whole lists go through in one call and that every intermediate keeps the reference's storage width (the
// Hadamard forms compute in int16_t and wrap exactly like the C code).
#include "common.h"

namespace aomhip {

__device__ __forceinline__ int64_t wave_sum64(int64_t v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor((long long)v, m, 64);
  return v;
}

template <typename T>
__global__ __launch_bounds__(256) void sse_kernel(PlaneView<T> a, PlaneView<T> b, int frame, int w, int h,
                                                  const aomhip_sad_cand *__restrict__ cands, int n, int64_t *__restrict__ out) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int ci = blockIdx.x * 4 + wave;
  if (ci >= n) return;
  const aomhip_sad_cand c = cands[ci];
  const T *ap = a.origin + (int64_t)frame * a.frame_stride + (int64_t)c.sy * a.stride + c.sx;
  const T *bp = b.origin + (int64_t)frame * b.frame_stride + (int64_t)c.ry * b.stride + c.rx;
  int64_t acc = 0;
  for (int i = lane; i < w * h; i += 64) {
    const int r = i / w, x = i - r * w;
    const int d = (int)ap[(int64_t)r * a.stride + x] - (int)bp[(int64_t)r * b.stride + x];
    acc += __mul24(d, d);
  }
  acc = wave_sum64(acc);
  if (lane == 0) out[ci] = acc;
}

// hadamard_col8 (avg.c:155-183); WIDE = hadamard_highbd_col8_second_pass (:391-422), 32-bit throughout
template <bool WIDE> __device__ __forceinline__ void col8(const int (&in)[8], int (&out)[8]) {
  auto w = [](int v) { return WIDE ? v : (int)(int16_t)v; };
  const int b0 = w(in[0] + in[1]), b1 = w(in[0] - in[1]), b2 = w(in[2] + in[3]), b3 = w(in[2] - in[3]);
  const int b4 = w(in[4] + in[5]), b5 = w(in[4] - in[5]), b6 = w(in[6] + in[7]), b7 = w(in[6] - in[7]);
  const int c0 = w(b0 + b2), c1 = w(b1 + b3), c2 = w(b0 - b2), c3 = w(b1 - b3);
  const int c4 = w(b4 + b6), c5 = w(b5 + b7), c6 = w(b4 - b6), c7 = w(b5 - b7);
  out[0] = w(c0 + c4); out[7] = w(c1 + c5); out[3] = w(c2 + c6); out[4] = w(c3 + c7);
  out[2] = w(c0 - c4); out[6] = w(c1 - c5); out[1] = w(c2 - c6); out[5] = w(c3 - c7);
}

enum { kHadPlain = 0, kHadLp = 1, kHadHighbd = 2 };

// N = 8, 16, 32.  The 8x8 sub-blocks are numbered the way the reference nests them (quadrant order at every level,
// avg.c:258-263,329-334), sub-block s landing at coeff + 64 s; 8 lanes own one sub-block (one column each).
template <int N, int FLAVOUR>
__global__ __launch_bounds__(256) void hadamard_kernel(const int16_t *__restrict__ residual, int stride,
                                                       const aomhip_txb *__restrict__ blocks, int n_blocks, void *__restrict__ coeff_out,
                                                       int32_t *__restrict__ satd_out) {
  constexpr int NSUB = (N / 8) * (N / 8);
  __shared__ int lds[4][2][N * N];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int bi = blockIdx.x * 4 + wave;
  const bool live = bi < n_blocks;
  const aomhip_txb blk = blocks[live ? bi : n_blocks - 1];
  const int16_t *src = residual + (int64_t)blk.y * stride + blk.x;
  int *buf = lds[wave][0], *co = lds[wave][1];
  for (int task = lane; task < NSUB * 8; task += 64) {
    const int s = task >> 3, c = task & 7;
    const int i16 = s >> 2, i8 = s & 3;
    const int row0 = N == 8 ? 0 : N == 16 ? (s >> 1) * 8 : (i16 >> 1) * 16 + (i8 >> 1) * 8;
    const int col0 = N == 8 ? 0 : N == 16 ? (s & 1) * 8 : (i16 & 1) * 16 + (i8 & 1) * 8;
    int in[8], out[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) in[r] = src[(int64_t)(row0 + r) * stride + col0 + c];
    col8<false>(in, out);
#pragma unroll
    for (int k = 0; k < 8; ++k) buf[s * 64 + c * 8 + k] = out[k];
  }
  __syncthreads();
  for (int task = lane; task < NSUB * 8; task += 64) {
    const int s = task >> 3, c = task & 7;
    int in[8], out[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) in[r] = buf[s * 64 + r * 8 + c];
    col8<FLAVOUR == kHadHighbd>(in, out);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      // the plain / lp forms end with a transpose "to match SSE2 behavior" (avg.c:207-212,237-243); highbd does not (:445)
      if constexpr (FLAVOUR == kHadHighbd) co[s * 64 + c * 8 + k] = out[k];
      else co[s * 64 + k * 8 + c] = out[k];
    }
  }
  __syncthreads();
  if constexpr (N >= 16) {
    constexpr int G = N == 16 ? 1 : 4;
    for (int t = lane; t < G * 64; t += 64) {  // avg.c:265-283 (lp: :305-322, highbd: :461-479)
      int *p = co + (t >> 6) * 256 + (t & 63);
      const int a0 = p[0], a1 = p[64], a2 = p[128], a3 = p[192];
      int b0 = (a0 + a1) >> 1, b1 = (a0 - a1) >> 1, b2 = (a2 + a3) >> 1, b3 = (a2 - a3) >> 1;
      if constexpr (FLAVOUR == kHadLp) { b0 = (int16_t)b0; b1 = (int16_t)b1; b2 = (int16_t)b2; b3 = (int16_t)b3; }
      int o0 = b0 + b2, o1 = b1 + b3, o2 = b0 - b2, o3 = b1 - b3;
      if constexpr (FLAVOUR == kHadLp) { o0 = (int16_t)o0; o1 = (int16_t)o1; o2 = (int16_t)o2; o3 = (int16_t)o3; }
      p[0] = o0; p[64] = o1; p[128] = o2; p[192] = o3;
    }
    __syncthreads();
    if constexpr (FLAVOUR == kHadPlain) {  // "extra shift to match AVX2 output" (avg.c:285-294)
      for (int t = lane; t < G * 64; t += 64) {
        int *p = co + (t >> 6) * 256 + ((t & 63) >> 2) * 16 + (t & 3);
        const int tmp = p[4];
        p[4] = p[8];
        p[8] = tmp;
      }
      __syncthreads();
    }
  }
  if constexpr (N == 32) {
    for (int t = lane; t < 256; t += 64) {  // avg.c:337-354, highbd :493-511
      int *p = co + t;
      const int a0 = p[0], a1 = p[256], a2 = p[512], a3 = p[768];
      const int b0 = (a0 + a1) >> 2, b1 = (a0 - a1) >> 2, b2 = (a2 + a3) >> 2, b3 = (a2 - a3) >> 2;
      p[0] = b0 + b2; p[256] = b1 + b3; p[512] = b0 - b2; p[768] = b1 - b3;
    }
    __syncthreads();
  }
  int satd = 0;
  for (int i = lane; i < N * N; i += 64) {
    const int v = co[i];
    satd += v < 0 ? -v : v;
    if (live && coeff_out) {
      if constexpr (FLAVOUR == kHadLp) static_cast<int16_t *>(coeff_out)[(int64_t)blk.out_offset + i] = (int16_t)v;
      else static_cast<int32_t *>(coeff_out)[(int64_t)blk.out_offset + i] = v;
    }
  }
  satd = (int)wave_sum64(satd);
  if (live && lane == 0 && satd_out) satd_out[bi] = satd;
}

// aom_hadamard_4x4_c (avg.c:110-153): four lanes per block, both passes halve after their first butterfly
__global__ __launch_bounds__(256) void hadamard4_kernel(const int16_t *__restrict__ residual, int stride, const aomhip_txb *__restrict__ blocks,
                                                        int n_blocks, int32_t *__restrict__ coeff_out, int32_t *__restrict__ satd_out) {
  __shared__ int lds[64][16];
  const int g = threadIdx.x >> 2, c = threadIdx.x & 3;
  const int bi = blockIdx.x * 64 + g;
  const bool live = bi < n_blocks;
  const aomhip_txb blk = blocks[live ? bi : n_blocks - 1];
  const int16_t *src = residual + (int64_t)blk.y * stride + blk.x;
  auto col4 = [](const int (&v)[4], int (&o)[4]) {
    const int b0 = (int16_t)((v[0] + v[1]) >> 1), b1 = (int16_t)((v[0] - v[1]) >> 1);
    const int b2 = (int16_t)((v[2] + v[3]) >> 1), b3 = (int16_t)((v[2] - v[3]) >> 1);
    o[0] = (int16_t)(b0 + b2); o[1] = (int16_t)(b1 + b3); o[2] = (int16_t)(b0 - b2); o[3] = (int16_t)(b1 - b3);
  };
  int v[4], o[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = src[(int64_t)r * stride + c];
  col4(v, o);
#pragma unroll
  for (int k = 0; k < 4; ++k) lds[g][c * 4 + k] = o[k];
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = lds[g][r * 4 + c];
  col4(v, o);
  int satd = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    satd += o[k] < 0 ? -o[k] : o[k];
    if (live && coeff_out) coeff_out[(int64_t)blk.out_offset + k * 4 + c] = o[k];  // transposed like the 8x8 form
  }
  satd += __shfl_xor(satd, 1, 64);
  satd += __shfl_xor(satd, 2, 64);
  if (live && c == 0 && satd_out) satd_out[bi] = satd;
}

__global__ __launch_bounds__(256) void txb_levels_kernel(const int32_t *__restrict__ coeff, int width, int height,
                                                         const uint32_t *__restrict__ coeff_offset, int n_blocks, uint8_t *__restrict__ levels,
                                                         int64_t levels_pitch) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int bi = blockIdx.x * 4 + wave;
  if (bi >= n_blocks) return;
  const int32_t *cf = coeff + (coeff_offset ? (int64_t)coeff_offset[bi] : (int64_t)bi * width * height);
  uint8_t *lv = levels + (int64_t)bi * levels_pitch;
  const int stride = height + 4;                         // TX_PAD_HOR (av1/common/enums.h:192)
  const int body = stride * width, total = body + 4 * stride + 16;  // + TX_PAD_BOTTOM rows + TX_PAD_END
  for (int idx = lane; idx < total; idx += 64) {
    int v = 0;
    if (idx < body) {
      const int i = idx / stride, j = idx - i * stride;
      if (j < height) {
        const int c = cf[i * height + j];
        const int a = c < 0 ? -c : c;
        v = a > 127 ? 127 : a;  // clamp(abs(coeff), 0, INT8_MAX)
      }
    }
    lv[idx] = (uint8_t)v;
  }
}


// ---- the wedge-mask helpers of pick_wedge (av1/encoder/compound_type.c): av1_wedge_sse_from_residuals / _sign_from_residuals /
// _compute_delta_squares (av1/encoder/wedge_utils.c:52-125).  One wavefront per (block, mask): the block's residual arrays are N int16
// (N = bw * bh, a multiple of 64), the masks N uint8 each -- 8 elements per lane and step (one 16-byte load of each int16 array, one 8-byte
// load of the mask), the wavefront's sums by shuffles.  HBM-bound streaming: 5 N bytes per (block, mask) for the SSE, 3 N for the sign.
__global__ __launch_bounds__(256) void wedge_sse_kernel(const int16_t *__restrict__ r1, const int16_t *__restrict__ d, const uint8_t *__restrict__ masks, int n,
                                                         int n_blocks, int n_masks, uint64_t *__restrict__ out) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int64_t job = (int64_t)blockIdx.x * 4 + wave;
  if (job >= (int64_t)n_blocks * n_masks) return;
  const int bi = (int)(job / n_masks), mi = (int)(job - (int64_t)bi * n_masks);
  const int16_t *pr = r1 + (int64_t)bi * n, *pd = d + (int64_t)bi * n;
  const uint8_t *pm = masks + (int64_t)mi * n;
  unsigned long long csse = 0;
  for (int i = lane * 8; i < n; i += 512) {
    const uint4 vr = *reinterpret_cast<const uint4 *>(pr + i), vd = *reinterpret_cast<const uint4 *>(pd + i);
    const uint2 vm = *reinterpret_cast<const uint2 *>(pm + i);
    const uint32_t wr[4] = { vr.x, vr.y, vr.z, vr.w }, wd[4] = { vd.x, vd.y, vd.z, vd.w };
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
      uint32_t pair = 0;   // t^2 <= 2^30 (t = -32768): two of them fit 32 bits, more do not
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int e = k + h;
        const int r = (int)(int16_t)(wr[e >> 1] >> (16 * (e & 1))), dd = (int)(int16_t)(wd[e >> 1] >> (16 * (e & 1)));
        const int m = (int)(((e < 4 ? vm.x : vm.y) >> (8 * (e & 3))) & 0xffu);
        int t = 64 * r + m * dd;   // MAX_MASK_VALUE * r1 + m * d
        t = t < -32768 ? -32768 : (t > 32767 ? 32767 : t);
        pair += (uint32_t)(t * t);
      }
      csse += pair;
    }
  }
  for (int msk = 1; msk < 64; msk <<= 1) csse += __shfl_xor(csse, msk, 64);
  if (lane == 0) out[job] = (csse + 2048) >> 12;   // ROUND_POWER_OF_TWO(csse, 2 * WEDGE_WEIGHT_BITS)
}

__global__ __launch_bounds__(256) void wedge_sign_kernel(const int16_t *__restrict__ ds, const uint8_t *__restrict__ masks, int n, int n_blocks, int n_masks,
                                                          const int64_t *__restrict__ limits, int8_t *__restrict__ out) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int64_t job = (int64_t)blockIdx.x * 4 + wave;
  if (job >= (int64_t)n_blocks * n_masks) return;
  const int bi = (int)(job / n_masks), mi = (int)(job - (int64_t)bi * n_masks);
  const int16_t *pd = ds + (int64_t)bi * n;
  const uint8_t *pm = masks + (int64_t)mi * n;
  long long acc = 0;
  for (int i = lane * 8; i < n; i += 512) {
    const uint4 vd = *reinterpret_cast<const uint4 *>(pd + i);
    const uint2 vm = *reinterpret_cast<const uint2 *>(pm + i);
    const uint32_t wd[4] = { vd.x, vd.y, vd.z, vd.w };
    int a = 0;   // 8 x 2^15 x 2^6
#pragma unroll
    for (int k = 0; k < 8; ++k)
      a += (int)(int16_t)(wd[k >> 1] >> (16 * (k & 1))) * (int)(((k < 4 ? vm.x : vm.y) >> (8 * (k & 3))) & 0xffu);
    acc += a;
  }
  for (int msk = 1; msk < 64; msk <<= 1) acc += __shfl_xor(acc, msk, 64);
  if (lane == 0) out[job] = (int8_t)(acc > limits[bi]);
}

__global__ __launch_bounds__(256) void wedge_delta_squares_kernel(const int16_t *__restrict__ a, const int16_t *__restrict__ b, int64_t n_total,
                                                                   int16_t *__restrict__ d) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
  if (i >= n_total) return;
  const uint4 va = *reinterpret_cast<const uint4 *>(a + i), vb = *reinterpret_cast<const uint4 *>(b + i);
  const uint32_t wa[4] = { va.x, va.y, va.z, va.w }, wb[4] = { vb.x, vb.y, vb.z, vb.w };
  uint32_t o[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int v[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int x = (int)(int16_t)(wa[k] >> (16 * h)), y = (int)(int16_t)(wb[k] >> (16 * h));
      const int t = x * x - y * y;
      v[h] = t < -32768 ? -32768 : (t > 32767 ? 32767 : t);
    }
    o[k] = (uint32_t)(uint16_t)v[0] | ((uint32_t)(uint16_t)v[1] << 16);
  }
  *reinterpret_cast<uint4 *>(d + i) = make_uint4(o[0], o[1], o[2], o[3]);
}

// aom_sum_squares_2d_i16 / aom_sum_sse_2d_i16 (aom_dsp/sum_squares.c:16-30,75-90) over a list of width x height blocks of an int16 residual plane: one
// wavefront per block, 64 pixels per step; the squares fit 31 bits, the sums are 64-bit (32-bit for `sum`, as the reference's int).
__global__ __launch_bounds__(256) void sum_sse_i16_kernel(const int16_t *__restrict__ residual, int stride, int width, int height,
                                                          const aomhip_txb *__restrict__ blocks, int n_blocks, int64_t *__restrict__ sse,
                                                          int32_t *__restrict__ sum) {
  const int lane = threadIdx.x & 63;
  const int bi = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (bi >= n_blocks) return;
  const int16_t *p = residual + (int64_t)blocks[bi].y * stride + blocks[bi].x;
  int64_t ss = 0;
  int s = 0;
  for (int i = lane; i < width * height; i += 64) {
    const int r = i / width, c = i - r * width;
    const int v = p[(int64_t)r * stride + c];
    ss += v * v;
    s += v;
  }
  for (int m = 1; m < 64; m <<= 1) {
    ss += __shfl_xor((unsigned long long)ss, m, 64);
    s += __shfl_xor(s, m, 64);
  }
  if (lane == 0) {
    sse[bi] = ss;
    if (sum) sum[bi] = s;
  }
}

}  // namespace aomhip

using namespace aomhip;

extern "C" {

int aomhip_sum_sse_2d_i16_batch(aomhip_ctx *ctx, const int16_t *d_residual, int residual_stride, int width, int height, const aomhip_txb *d_blocks,
                                int n_blocks, int64_t *d_sse, int32_t *d_sum) {
  if (!ctx || !d_residual || residual_stride <= 0 || width <= 0 || height <= 0 || width > 128 || height > 128 || n_blocks < 0 ||
      (n_blocks > 0 && (!d_blocks || !d_sse))) {
    set_error("aomhip_sum_sse_2d_i16_batch: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  if (n_blocks == 0) return AOMHIP_OK;
  hipLaunchKernelGGL(sum_sse_i16_kernel, dim3((n_blocks + 3) / 4), dim3(256), 0, ctx->stream, d_residual, residual_stride, width, height, d_blocks, n_blocks,
                     d_sse, d_sum);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}

int aomhip_sse_batch(aomhip_ctx *ctx, const aomhip_planes *a, const aomhip_planes *b, int frame, int width, int height,
                     const aomhip_sad_cand *d_cands, int n_cands, int64_t *d_out) {
  if (!ctx || !a || !b || !a->base || !b->base || (n_cands > 0 && (!d_cands || !d_out)) || n_cands < 0 || frame < 0 || frame >= a->n_frames ||
      frame >= b->n_frames || width <= 0 || height <= 0 || width > 128 || height > 128 || (a->bit_depth == 8) != (b->bit_depth == 8)) {
    set_error("aomhip_sse_batch: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  if (n_cands == 0) return AOMHIP_OK;
  const dim3 grid((n_cands + 3) / 4), block(256);
  if (a->bit_depth == 8)
    hipLaunchKernelGGL(sse_kernel<uint8_t>, grid, block, 0, ctx->stream, view_of<uint8_t>(*a), view_of<uint8_t>(*b), frame, width, height,
                       d_cands, n_cands, d_out);
  else
    hipLaunchKernelGGL(sse_kernel<uint16_t>, grid, block, 0, ctx->stream, view_of<uint16_t>(*a), view_of<uint16_t>(*b), frame, width, height,
                       d_cands, n_cands, d_out);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}

int aomhip_hadamard_batch(aomhip_ctx *ctx, const int16_t *d_residual, int residual_stride, int n, int flavour, const aomhip_txb *d_blocks,
                          int n_blocks, void *d_coeff, int32_t *d_satd) {
  const bool size_ok = (flavour == AOMHIP_HADAMARD && (n == 4 || n == 8 || n == 16 || n == 32)) ||
                       (flavour == AOMHIP_HADAMARD_LP && (n == 8 || n == 16)) ||
                       (flavour == AOMHIP_HADAMARD_HIGHBD && (n == 8 || n == 16 || n == 32));
  if (!ctx || !d_residual || residual_stride <= 0 || !size_ok || (n_blocks > 0 && !d_blocks) || n_blocks < 0 || (!d_coeff && !d_satd)) {
    set_error("aomhip_hadamard_batch: invalid argument (size %d, flavour %d)", n, flavour);
    return AOMHIP_ERR_INVALID;
  }
  if (n_blocks == 0) return AOMHIP_OK;
  const dim3 block(256);
  if (n == 4) {
    hipLaunchKernelGGL(hadamard4_kernel, dim3((n_blocks + 63) / 64), block, 0, ctx->stream, d_residual, residual_stride, d_blocks, n_blocks,
                       static_cast<int32_t *>(d_coeff), d_satd);
    AOMHIP_LAUNCH_CHECK();
    return AOMHIP_OK;
  }
  const dim3 grid((n_blocks + 3) / 4);
#define LAUNCH(N, F)                                                                                                                \
  if (n == N && flavour == F) {                                                                                                     \
    hipLaunchKernelGGL((hadamard_kernel<N, F>), grid, block, 0, ctx->stream, d_residual, residual_stride, d_blocks, n_blocks, d_coeff, \
                       d_satd);                                                                                                     \
    AOMHIP_LAUNCH_CHECK();                                                                                                          \
    return AOMHIP_OK;                                                                                                               \
  }
  LAUNCH(8, kHadPlain) LAUNCH(16, kHadPlain) LAUNCH(32, kHadPlain) LAUNCH(8, kHadLp) LAUNCH(16, kHadLp)
  LAUNCH(8, kHadHighbd) LAUNCH(16, kHadHighbd) LAUNCH(32, kHadHighbd)
#undef LAUNCH
  return AOMHIP_ERR_INVALID;
}

int aomhip_txb_init_levels_batch(aomhip_ctx *ctx, const int32_t *d_coeff, int width, int height, const uint32_t *d_coeff_offset, int n_blocks,
                                 uint8_t *d_levels, int64_t levels_pitch) {
  if (!ctx || (n_blocks > 0 && (!d_coeff || !d_levels)) || n_blocks < 0 || width < 4 || height < 4 || width > 32 || height > 32 ||
      (width & (width - 1)) || (height & (height - 1)) || levels_pitch < (int64_t)(height + 4) * (width + 4) + 16) {
    set_error("aomhip_txb_init_levels_batch: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  if (n_blocks == 0) return AOMHIP_OK;
  hipLaunchKernelGGL(txb_levels_kernel, dim3((n_blocks + 3) / 4), dim3(256), 0, ctx->stream, d_coeff, width, height, d_coeff_offset, n_blocks,
                     d_levels, levels_pitch);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}

int aomhip_wedge_sse_from_residuals_batch(aomhip_ctx *ctx, const int16_t *d_r1, const int16_t *d_d, const uint8_t *d_masks, int n, int n_blocks, int n_masks,
                                          uint64_t *d_sse) {
  if (!ctx || n_blocks < 0 || n_masks < 0 || n <= 0 || (n & 63) || ((n_blocks > 0 && n_masks > 0) && (!d_r1 || !d_d || !d_masks || !d_sse))) {
    set_error("aomhip_wedge_sse_from_residuals_batch: invalid argument (N is a positive multiple of 64)");
    return AOMHIP_ERR_INVALID;
  }
  const int64_t jobs = (int64_t)n_blocks * n_masks;
  if (jobs == 0) return AOMHIP_OK;
  hipLaunchKernelGGL(wedge_sse_kernel, dim3((unsigned)((jobs + 3) / 4)), dim3(256), 0, ctx->stream, d_r1, d_d, d_masks, n, n_blocks, n_masks, d_sse);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}

int aomhip_wedge_sign_from_residuals_batch(aomhip_ctx *ctx, const int16_t *d_ds, const uint8_t *d_masks, int n, int n_blocks, int n_masks,
                                           const int64_t *d_limits, int8_t *d_sign) {
  if (!ctx || n_blocks < 0 || n_masks < 0 || n <= 0 || (n & 63) || ((n_blocks > 0 && n_masks > 0) && (!d_ds || !d_masks || !d_limits || !d_sign))) {
    set_error("aomhip_wedge_sign_from_residuals_batch: invalid argument (N is a positive multiple of 64)");
    return AOMHIP_ERR_INVALID;
  }
  const int64_t jobs = (int64_t)n_blocks * n_masks;
  if (jobs == 0) return AOMHIP_OK;
  hipLaunchKernelGGL(wedge_sign_kernel, dim3((unsigned)((jobs + 3) / 4)), dim3(256), 0, ctx->stream, d_ds, d_masks, n, n_blocks, n_masks, d_limits, d_sign);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}

int aomhip_wedge_compute_delta_squares_batch(aomhip_ctx *ctx, const int16_t *d_a, const int16_t *d_b, int n, int n_blocks, int16_t *d_d) {
  if (!ctx || n_blocks < 0 || n <= 0 || (n & 63) || (n_blocks > 0 && (!d_a || !d_b || !d_d))) {
    set_error("aomhip_wedge_compute_delta_squares_batch: invalid argument (N is a positive multiple of 64)");
    return AOMHIP_ERR_INVALID;
  }
  const int64_t total = (int64_t)n * n_blocks;
  if (total == 0) return AOMHIP_OK;
  hipLaunchKernelGGL(wedge_delta_squares_kernel, dim3((unsigned)((total / 8 + 255) / 256)), dim3(256), 0, ctx->stream, d_a, d_b, total, d_d);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}

}  // extern "C"
