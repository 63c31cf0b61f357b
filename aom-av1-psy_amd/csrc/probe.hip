// aomhip_strip_read_probe: the TRANSPORT of sad_strip_kernel with everything else removed -- persistent workgroups pull the same
// strip-shaped windows (rows of sb_w + 2 * range reference pixels and sb_w source pixels at the plane's row pitch, sb_h rows per step,
// the strips of one frame side by side on one XCD) out of memory into registers: no LDS, no barriers, no evaluation.  Its launch time is
// the ceiling of that walk on this box, measured by bench.py in the same run as the kernel (roofline.ceiling_GBs / frac_of_ceiling).
// Measurement support, not part of the encoder path: it computes nothing.  (Origin: tools/strip_read_probe.hip, profiles/r02_strip_read_probe.log.)
#include "common.h"

#include <algorithm>
#include <array>
#include <utility>
#include <vector>

namespace aomhip {
namespace {

typedef uint32_t V4 __attribute__((ext_vector_type(4)));
struct ProbeArgs {
  const char *ref, *src;
  int64_t r_fstride, s_fstride;   // bytes between frames
  int frames, strips, sb_w_bytes, sb_h, range_bytes, r_pitch, s_pitch, steps, lanes;
  int cpr, scpr;                  // 16-byte chunks per window row / source row
  unsigned magic_cpr, magic_scpr;
  unsigned *sink;
};
unsigned magic_of(int d) { return (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); }

template <int N, int NS>
__global__ __launch_bounds__(1024) void strip_read_probe_kernel(ProbeArgs a) {
  const int tid = (int)threadIdx.x;
  if (tid >= a.lanes) return;
  const int xcd = (int)(blockIdx.x & 7), wg_j = (int)(blockIdx.x >> 3), wg_n = (int)(gridDim.x >> 3);
  const int frames_here = (a.frames - xcd + 7) >> 3;
  const int n_items = frames_here * a.strips;
  unsigned r_off[N], s_off[NS];
  const int total_r = a.sb_h * a.cpr, total_s = a.sb_h * a.scpr;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const unsigned q = (unsigned)min(tid + i * a.lanes, total_r - 1), row = __umulhi(q, a.magic_cpr), col = q - row * (unsigned)a.cpr;
    r_off[i] = row * (unsigned)a.r_pitch + col * 16u;
  }
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const unsigned q = (unsigned)min(tid + i * a.lanes, total_s - 1), row = __umulhi(q, a.magic_scpr), col = q - row * (unsigned)a.scpr;
    s_off[i] = row * (unsigned)a.s_pitch + col * 16u;
  }
  V4 acc = { 0, 0, 0, 0 };
  for (int item = wg_j; item < n_items; item += wg_n) {
    const int fi = item / a.strips, cx = item - fi * a.strips, f = xcd + 8 * fi;
    // (ref / src point at the first window / source row and column of strip 0; the window starts `range` left of the cell)
    const char *rb = a.ref + f * a.r_fstride + (int64_t)cx * a.sb_w_bytes;
    const char *sb = a.src + f * a.s_fstride + (int64_t)cx * a.sb_w_bytes;
    V4 v[N], s[NS];
    auto load = [&](int cy) {
      const int cyc = min(cy, a.steps - 1);
      const char *r0 = rb + (int64_t)cyc * a.sb_h * a.r_pitch, *s0 = sb + (int64_t)cyc * a.sb_h * a.s_pitch;
#pragma unroll
      for (int i = 0; i < N; ++i) v[i] = *reinterpret_cast<const V4 *>(r0 + r_off[i]);
#pragma unroll
      for (int i = 0; i < NS; ++i) s[i] = *reinterpret_cast<const V4 *>(s0 + s_off[i]);
    };
    load(0);
    for (int cy = 0; cy < a.steps; ++cy) {
#pragma unroll
      for (int i = 0; i < N; ++i) acc ^= v[i];
#pragma unroll
      for (int i = 0; i < NS; ++i) acc ^= s[i];
      load(cy + 1);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) acc ^= v[i];
#pragma unroll
    for (int i = 0; i < NS; ++i) acc ^= s[i];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u && a.sink) a.sink[0] = 1;
}

}  // namespace
}  // namespace aomhip

using namespace aomhip;

extern "C" int aomhip_strip_read_probe(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *ref, int first_frame, int n_frames,
                                       int x0, int x1, int sb_w, int sb_h, int range, int64_t *bytes_requested) {
  if (!ctx || !src || !ref || !src->base || !ref->base || first_frame < 0 || n_frames <= 0 || first_frame + n_frames > src->n_frames ||
      first_frame + n_frames > ref->n_frames || sb_w <= 0 || sb_h <= 0 || range < 0 || x0 < 0 || x1 > src->width || x1 - x0 < sb_w ||
      range > ref->border || src->bit_depth != ref->bit_depth) {
    set_error("aomhip_strip_read_probe: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  const int es = src->bit_depth == 8 ? 1 : 2;
  if ((sb_w * es) % 16 || (range * es) % 16) {
    set_error("aomhip_strip_read_probe: cell width and range must be whole 16-byte chunks");
    return AOMHIP_ERR_INVALID;
  }
  ProbeArgs a;
  a.frames = n_frames; a.strips = (x1 - x0) / sb_w; a.sb_w_bytes = sb_w * es; a.sb_h = sb_h; a.range_bytes = range * es;
  a.r_pitch = ref->stride * es; a.s_pitch = src->stride * es;
  a.steps = src->height / sb_h;   // whole steps only; the window's first row = the first visible row (the halo rows are the same per strip)
  a.cpr = (sb_w + 2 * range) * es / 16; a.scpr = sb_w * es / 16;
  a.magic_cpr = magic_of(a.cpr); a.magic_scpr = magic_of(a.scpr);
  a.r_fstride = ref->frame_stride * es; a.s_fstride = src->frame_stride * es;
  a.ref = static_cast<const char *>(ref->base) + ((int64_t)first_frame * ref->frame_stride + (int64_t)ref->border * ref->stride + ref->border + x0 - range) * es;
  a.src = static_cast<const char *>(src->base) + ((int64_t)first_frame * src->frame_stride + (int64_t)src->border * src->stride + src->border + x0) * es;
  a.sink = nullptr;
  a.lanes = 512;
  const int chunks = sb_h * a.cpr, schunks = sb_h * a.scpr;
  if (bytes_requested) *bytes_requested = (int64_t)n_frames * a.strips * a.steps * (int64_t)(chunks + schunks) * 16;
  const int n = (chunks + 511) / 512, ns = (schunks + 511) / 512;
  const dim3 grid(256), block(1024);
#define RUN(N, NS)                                                                                          \
  if (n <= N && ns <= NS) {                                                                                 \
    hipLaunchKernelGGL((strip_read_probe_kernel<N, NS>), grid, block, 0, ctx->stream, a);                    \
    AOMHIP_LAUNCH_CHECK();                                                                                  \
    return AOMHIP_OK;                                                                                       \
  }
  RUN(2, 1) RUN(2, 2) RUN(3, 2) RUN(4, 2) RUN(4, 3) RUN(6, 4) RUN(8, 6)
#undef RUN
  set_error("aomhip_strip_read_probe: cell %d x %d too large for the probe", sb_w, sb_h);
  return AOMHIP_ERR_INVALID;
}

// ------------------------------------------------------------------------------------------------------------------------------------
// aomhip_valu_issue_probe: the ISSUE RATE of one VALU opcode class on this box, in this run -- the denominator of bench.py's `valu_frac`
// figures.  W wavefronts per SIMD, each with 8 independent dependency chains of the same instruction (128 instructions per loop trip,
// one scalar add / compare / branch beside them), long enough to run at the sustained clock.  Reported: wave-instructions per second per
// SIMD from the HIP-event time of the launch, and the same interval in s_memtime ticks per wave-instruction together with the tick rate
// (against s_memrealtime, 100 MHz), so "how many clocks does a wave64 instruction take" does not rest on an assumed frequency.
// Measurement support, not part of the encoder path: it computes nothing.  (Origin: tools/r03_ubench.hip, profiles/r03_ubench.log.)
namespace aomhip {
namespace {

enum ValuOp {
  OP_ADD_U32 = 0, OP_MAD_I32_I24, OP_MUL_U32_U24, OP_MUL_LO_U32, OP_MUL_HI_U32, OP_MAD_U64_U32, OP_LSHL_ADD_U32, OP_ADD3_U32, OP_BFE_I32,
  OP_MAX_I32, OP_MED3_I32, OP_CNDMASK, OP_SAD_U8, OP_SAD_U16, OP_ALIGNBYTE, OP_PERM_B32, OP_PK_ADD_I16, OP_PK_MAD_I16, OP_PK_MAX_I16,
  OP_DOT2_I32_I16, OP_DOT2_U32_U16, OP_ADD_DPP_ROW_SHR, OP_MOV_DPP_ROW_MIRROR, OP_ADD_F32, OP_FMA_F32, OP_EXP_F32, OP_FMA_F64, OP_MUL_F64,
  OP_ADD_F64, OP_RCP_F64, OP_CVT_F64_I32, OP_COUNT
};
const char *const kValuOpNames[OP_COUNT] = {
  "v_add_u32", "v_mad_i32_i24", "v_mul_u32_u24", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_lshl_add_u32", "v_add3_u32", "v_bfe_i32",
  "v_max_i32", "v_med3_i32", "v_cndmask_b32", "v_sad_u8", "v_sad_u16", "v_alignbyte_b32", "v_perm_b32", "v_pk_add_i16", "v_pk_mad_i16", "v_pk_max_i16",
  "v_dot2_i32_i16", "v_dot2_u32_u16", "v_add_u32_dpp(row_shr:1)", "v_mov_b32_dpp(row_mirror)", "v_add_f32", "v_fma_f32", "v_exp_f32", "v_fma_f64", "v_mul_f64",
  "v_add_f64", "v_rcp_f64", "v_cvt_f64_i32"
};

// one instruction on chain register c (32-bit chains) / d (64-bit chains); a, b: loop-invariant VGPR operands
template <int OP> __device__ __forceinline__ void one32(uint32_t &c, uint32_t a, uint32_t b) {
  if constexpr (OP == OP_ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(c) : "v"(a));
  else if constexpr (OP == OP_MAD_I32_I24) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(c) : "v"(a), "v"(b));
  else if constexpr (OP == OP_MUL_U32_U24) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(c) : "v"(a));
  else if constexpr (OP == OP_MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(c) : "v"(a));
  else if constexpr (OP == OP_MUL_HI_U32) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(c) : "v"(a));
  else if constexpr (OP == OP_LSHL_ADD_U32) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(c) : "v"(a));
  else if constexpr (OP == OP_ADD3_U32) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(c) : "v"(a), "v"(b));
  else if constexpr (OP == OP_BFE_I32) asm volatile("v_bfe_i32 %0, %0, %1, 16" : "+v"(c) : "v"(a));
  else if constexpr (OP == OP_MAX_I32) asm volatile("v_max_i32 %0, %0, %1" : "+v"(c) : "v"(a));
  else if constexpr (OP == OP_MED3_I32) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(c) : "v"(a), "v"(b));
  else if constexpr (OP == OP_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(c) : "v"(a));
  else if constexpr (OP == OP_SAD_U8) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  else if constexpr (OP == OP_SAD_U16) asm volatile("v_sad_u16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  else if constexpr (OP == OP_ALIGNBYTE) asm volatile("v_alignbyte_b32 %0, %0, %1, %2" : "+v"(c) : "v"(a), "v"(b));
  else if constexpr (OP == OP_PERM_B32) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(c) : "v"(a), "v"(b));
  else if constexpr (OP == OP_PK_ADD_I16) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(c) : "v"(a));
  else if constexpr (OP == OP_PK_MAD_I16) asm volatile("v_pk_mad_i16 %0, %0, %1, %2" : "+v"(c) : "v"(a), "v"(b));
  else if constexpr (OP == OP_PK_MAX_I16) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(c) : "v"(a));
  else if constexpr (OP == OP_DOT2_I32_I16) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  else if constexpr (OP == OP_DOT2_U32_U16) asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  else if constexpr (OP == OP_ADD_DPP_ROW_SHR) asm volatile("v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(c) : "v"(a));
  else if constexpr (OP == OP_MOV_DPP_ROW_MIRROR) asm volatile("v_mov_b32_dpp %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(c));
  else if constexpr (OP == OP_ADD_F32) asm volatile("v_add_f32 %0, %0, %1" : "+v"(c) : "v"(a));
  else if constexpr (OP == OP_FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(c) : "v"(a), "v"(b));
  else if constexpr (OP == OP_EXP_F32) asm volatile("v_exp_f32 %0, %0" : "+v"(c));
}
template <int OP> __device__ __forceinline__ void one64(double &d, double a, double b, uint32_t ia) {
  if constexpr (OP == OP_FMA_F64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d) : "v"(a), "v"(b));
  else if constexpr (OP == OP_MUL_F64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d) : "v"(a));
  else if constexpr (OP == OP_ADD_F64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d) : "v"(a));
  else if constexpr (OP == OP_RCP_F64) asm volatile("v_rcp_f64 %0, %0" : "+v"(d));
  else if constexpr (OP == OP_CVT_F64_I32) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d) : "v"(ia));
  else if constexpr (OP == OP_MAD_U64_U32) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d) : "v"(ia), "v"(ia) : "vcc");
}
template <int OP> constexpr bool is64() { return OP == OP_FMA_F64 || OP == OP_MUL_F64 || OP == OP_ADD_F64 || OP == OP_RCP_F64 || OP == OP_CVT_F64_I32 || OP == OP_MAD_U64_U32; }

constexpr int kProbeChains = 8, kProbeReps = 16;   // 128 instructions per loop trip

template <int OP>
__global__ __launch_bounds__(1024) void valu_issue_probe_kernel(int iters, uint32_t a0, uint32_t b0, unsigned long long *stamps, uint32_t *sink) {
  extern __shared__ uint32_t probe_lds_pad[];   // (only its size matters: it limits the workgroups per CU)
  const uint32_t tid = threadIdx.x;
  uint32_t a = a0 + (tid & 1), b = b0 | 1u;
  uint32_t c[kProbeChains];
  double d[kProbeChains];
#pragma unroll
  for (int k = 0; k < kProbeChains; ++k) { c[k] = tid * 2654435761u + k; d[k] = 1.0 + 1e-9 * (double)(tid + k); }
  const double da = 1.0 + 1e-12 * (double)tid, db = 1e-30;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < kProbeReps; ++r) {
#pragma unroll
      for (int k = 0; k < kProbeChains; ++k) {
        if constexpr (is64<OP>()) one64<OP>(d[k], da, db, a); else one32<OP>(c[k], a, b);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  uint32_t x = 0;
#pragma unroll
  for (int k = 0; k < kProbeChains; ++k) x ^= is64<OP>() ? (uint32_t)__double2loint(d[k]) ^ (uint32_t)__double2hiint(d[k]) : c[k];
  if (x == 0x12345678u && sink) sink[0] = x + probe_lds_pad[0];
  if ((tid & 63) == 0 && stamps) {
    unsigned long long *s = stamps + ((size_t)blockIdx.x * (blockDim.x >> 6) + (tid >> 6)) * 2;
    s[0] = t1 - t0;
    s[1] = r1 - r0;
  }
}

typedef void (*ProbeFn)(int, uint32_t, uint32_t, unsigned long long *, uint32_t *);
template <int... I> constexpr std::array<ProbeFn, sizeof...(I)> probe_table(std::integer_sequence<int, I...>) { return {{valu_issue_probe_kernel<I>...}}; }

}  // namespace
}  // namespace aomhip

extern "C" const char *aomhip_valu_issue_probe_name(int op_class) { return op_class >= 0 && op_class < OP_COUNT ? kValuOpNames[op_class] : nullptr; }

extern "C" int aomhip_valu_issue_probe(aomhip_ctx *ctx, int op_class, int waves_per_simd, int iters, aomhip_valu_probe_result *out) {
  if (!ctx || !out || op_class < 0 || op_class >= OP_COUNT || iters <= 0 || (waves_per_simd != 1 && waves_per_simd != 2 && waves_per_simd != 4 && waves_per_simd != 8)) {
    set_error("aomhip_valu_issue_probe: invalid argument (op class 0..%d, 1 / 2 / 4 / 8 wavefronts per SIMD)", OP_COUNT - 1);
    return AOMHIP_ERR_INVALID;
  }
  static const auto table = probe_table(std::make_integer_sequence<int, OP_COUNT>{});
  AOMHIP_TRY(hipSetDevice(ctx->device));
  hipDeviceProp_t prop;
  AOMHIP_TRY(hipGetDeviceProperties(&prop, ctx->device));
  const int cus = prop.multiProcessorCount;
  // a 256-thread workgroup = one wavefront per SIMD; W <= 4: one workgroup of W * 256 threads per CU (its LDS request keeps a second one
  // off the CU), W = 8: two workgroups of 1024 threads per CU (the CU holds 2048 threads: no third)
  const int per_wg = waves_per_simd > 4 ? 4 : waves_per_simd, wgs_per_cu = waves_per_simd / per_wg;
  const dim3 grid(cus * wgs_per_cu), block(256 * per_wg);
  const size_t lds = wgs_per_cu == 1 ? 96 * 1024 : 1024;
  const size_t n_waves = (size_t)grid.x * (block.x / 64);
  unsigned long long *d_stamps = nullptr;
  AOMHIP_TRY(hipMalloc(&d_stamps, n_waves * 2 * sizeof(unsigned long long)));
  ProbeFn fn = table[op_class];
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(fn, grid, block, lds, ctx->stream, iters / 8 + 1, 3u, 5u, d_stamps, (uint32_t *)nullptr);   // (ramp; same code)
    e = hipEventRecord(ctx->ev0, ctx->stream);
  }
  if (e == hipSuccess) {
    hipLaunchKernelGGL(fn, grid, block, lds, ctx->stream, iters, 3u, 5u, d_stamps, (uint32_t *)nullptr);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipEventRecord(ctx->ev1, ctx->stream);
  if (e == hipSuccess) e = hipEventSynchronize(ctx->ev1);
  float ms = 0.f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
  std::vector<unsigned long long> h(n_waves * 2);
  if (e == hipSuccess) e = hipMemcpy(h.data(), d_stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  (void)hipFree(d_stamps);
  if (e != hipSuccess) {
    set_error("aomhip_valu_issue_probe: %s", hipGetErrorString(e));
    return AOMHIP_ERR_HIP;
  }
  const double insts_per_wave = (double)iters * kProbeChains * kProbeReps;
  std::vector<double> ticks(n_waves), hz(n_waves);
  for (size_t i = 0; i < n_waves; ++i) {
    ticks[i] = (double)h[2 * i] / insts_per_wave;
    hz[i] = h[2 * i + 1] ? (double)h[2 * i] / (double)h[2 * i + 1] * 1e8 : 0.0;
  }
  std::nth_element(ticks.begin(), ticks.begin() + n_waves / 2, ticks.end());
  std::nth_element(hz.begin(), hz.begin() + n_waves / 2, hz.end());
  out->launch_ms = ms;
  out->wave_insts_per_s_per_simd = insts_per_wave * (double)n_waves / ((double)cus * 4.0) / ((double)ms * 1e-3);
  out->memtime_ticks_per_wave_inst = ticks[n_waves / 2] / (double)waves_per_simd;   // issue interval of the SIMD (a wave sees W times that)
  out->memtime_hz = hz[n_waves / 2];
  out->waves_per_simd = waves_per_simd;
  out->compute_units = cus;
  return AOMHIP_OK;
}
