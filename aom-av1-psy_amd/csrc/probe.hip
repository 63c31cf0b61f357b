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
// Measurement support, not part of the encoder path: it computes nothing.  (Origin: tools/lds_ubench.hip, profiles/r03_ubench.log.)
namespace aomhip {
namespace {

// X(enum tag, printed name, instructions per chain step, 64-bit chain?, asm)   -- [c] = the chain register, [a] / [b] = loop-invariant VGPRs,
// [t] = a 32-bit scratch VGPR of the 64-bit forms
#define AOMHIP_VALU_OPS(X)                                                                                              \
  X(ADD_U32, "v_add_u32", 1, 0, "v_add_u32 %[c], %[c], %[a]")                                                                  \
  X(ADD_U32_E64, "v_add_u32_e64", 1, 0, "v_add_u32_e64 %[c], %[c], %[a]")                                                  \
  X(SUB_U32, "v_sub_u32", 1, 0, "v_sub_u32 %[c], %[c], %[a]")                                                                  \
  X(MOV_B32, "v_mov_b32", 1, 0, "v_mov_b32 %[c], %[a]")                                                                      \
  X(AND_B32, "v_and_b32", 1, 0, "v_and_b32 %[c], %[c], %[a]")                                                                  \
  X(OR_B32, "v_or_b32", 1, 0, "v_or_b32 %[c], %[c], %[a]")                                                                     \
  X(XOR_B32, "v_xor_b32", 1, 0, "v_xor_b32 %[c], %[c], %[a]")                                                                  \
  X(LSHLREV_B32, "v_lshlrev_b32", 1, 0, "v_lshlrev_b32 %[c], 1, %[c]")                                                       \
  X(LSHRREV_B32, "v_lshrrev_b32", 1, 0, "v_lshrrev_b32 %[c], 1, %[c]")                                                       \
  X(ASHRREV_I32, "v_ashrrev_i32", 1, 0, "v_ashrrev_i32 %[c], 1, %[c]")                                                       \
  X(MAX_I32, "v_max_i32", 1, 0, "v_max_i32 %[c], %[c], %[a]")                                                                  \
  X(MIN_I32, "v_min_i32", 1, 0, "v_min_i32 %[c], %[c], %[a]")                                                                  \
  X(MUL_U32_U24, "v_mul_u32_u24", 1, 0, "v_mul_u32_u24 %[c], %[c], %[a]")                                                      \
  X(MUL_I32_I24, "v_mul_i32_i24", 1, 0, "v_mul_i32_i24 %[c], %[c], %[a]")                                                      \
  X(MAD_I32_I24, "v_mad_i32_i24", 1, 0, "v_mad_i32_i24 %[c], %[c], %[a], %[b]")                                                  \
  X(MUL_LO_U32, "v_mul_lo_u32", 1, 0, "v_mul_lo_u32 %[c], %[c], %[a]")                                                         \
  X(MUL_HI_U32, "v_mul_hi_u32", 1, 0, "v_mul_hi_u32 %[c], %[c], %[a]")                                                         \
  X(LSHL_ADD_U32, "v_lshl_add_u32", 1, 0, "v_lshl_add_u32 %[c], %[c], 1, %[a]")                                                \
  X(ADD3_U32, "v_add3_u32", 1, 0, "v_add3_u32 %[c], %[c], %[a], %[b]")                                                           \
  X(BFE_U32, "v_bfe_u32", 1, 0, "v_bfe_u32 %[c], %[c], %[a], 16")                                                              \
  X(BFE_I32, "v_bfe_i32", 1, 0, "v_bfe_i32 %[c], %[c], %[a], 16")                                                              \
  X(MED3_I32, "v_med3_i32", 1, 0, "v_med3_i32 %[c], %[c], %[a], %[b]")                                                           \
  X(ALIGNBIT, "v_alignbit_b32", 1, 0, "v_alignbit_b32 %[c], %[c], %[a], %[b]")                                                   \
  X(ALIGNBYTE, "v_alignbyte_b32", 1, 0, "v_alignbyte_b32 %[c], %[c], %[a], %[b]")                                                \
  X(PERM_B32, "v_perm_b32", 1, 0, "v_perm_b32 %[c], %[c], %[a], %[b]")                                                           \
  X(CMP_CNDMASK, "v_cmp_gt_i32+v_cndmask_b32", 2, 0, "v_cmp_gt_i32 vcc, %[c], %[a]\n\tv_cndmask_b32 %[c], %[c], %[b], vcc")        \
  X(SAD_U8, "v_sad_u8", 1, 0, "v_sad_u8 %[c], %[a], %[b], %[c]")                                                                 \
  X(SAD_U16, "v_sad_u16", 1, 0, "v_sad_u16 %[c], %[a], %[b], %[c]")                                                              \
  X(ADD_U16, "v_add_u16", 1, 0, "v_add_u16 %[c], %[c], %[a]")                                                                  \
  X(PK_ADD_I16, "v_pk_add_i16", 1, 0, "v_pk_add_i16 %[c], %[c], %[a]")                                                         \
  X(PK_SUB_I16, "v_pk_sub_i16", 1, 0, "v_pk_sub_i16 %[c], %[c], %[a]")                                                         \
  X(PK_MAD_I16, "v_pk_mad_i16", 1, 0, "v_pk_mad_i16 %[c], %[c], %[a], %[b]")                                                     \
  X(PK_MAX_I16, "v_pk_max_i16", 1, 0, "v_pk_max_i16 %[c], %[c], %[a]")                                                         \
  X(DOT2_I32_I16, "v_dot2_i32_i16", 1, 0, "v_dot2_i32_i16 %[c], %[a], %[b], %[c]")                                               \
  X(DOT2C_I32_I16, "v_dot2c_i32_i16", 1, 0, "v_dot2c_i32_i16 %[c], %[a], %[b]")                                                \
  X(DOT2_U32_U16, "v_dot2_u32_u16", 1, 0, "v_dot2_u32_u16 %[c], %[a], %[b], %[c]")                                               \
  X(ADD_DPP_ROW_SHR, "v_add_u32_dpp(row_shr:1)", 1, 0, "v_add_u32_dpp %[c], %[c], %[a] row_shr:1 row_mask:0xf bank_mask:0xf")  \
  X(MOV_DPP_ROW_MIRROR, "v_mov_b32_dpp(row_mirror)", 1, 0, "v_mov_b32_dpp %[c], %[c] row_mirror row_mask:0xf bank_mask:0xf") \
  X(READLANE, "v_readlane_b32", 1, 0, "v_readlane_b32 s20, %[c], 3")                                                       \
  X(ADD_F32, "v_add_f32", 1, 0, "v_add_f32 %[c], %[c], %[a]")                                                                  \
  X(MUL_F32, "v_mul_f32", 1, 0, "v_mul_f32 %[c], %[c], %[a]")                                                                  \
  X(FMA_F32, "v_fma_f32", 1, 0, "v_fma_f32 %[c], %[c], %[a], %[b]")                                                              \
  X(EXP_F32, "v_exp_f32", 1, 0, "v_exp_f32 %[c], %[c]")                                                                      \
  X(CVT_F32_I32, "v_cvt_f32_i32", 1, 0, "v_cvt_f32_i32 %[c], %[c]")                                                          \
  X(LSHL_ADD_U64, "v_lshl_add_u64", 1, 1, "v_lshl_add_u64 %[c], %[c], 1, %[a]")                                                \
  X(LSHRREV_B64, "v_lshrrev_b64", 1, 1, "v_lshrrev_b64 %[c], 1, %[c]")                                                       \
  X(MOV_B64, "v_mov_b64", 1, 1, "v_mov_b64 %[c], %[a]")                                                                      \
  X(MAD_U64_U32, "v_mad_u64_u32", 1, 1, "v_mad_u64_u32 %[c], vcc, %[t], %[t], %[c]")                                             \
  X(MAD_I64_I32, "v_mad_i64_i32", 1, 1, "v_mad_i64_i32 %[c], vcc, %[t], %[t], %[c]")                                             \
  X(FMA_F64, "v_fma_f64", 1, 1, "v_fma_f64 %[c], %[c], %[a], %[b]")                                                              \
  X(MUL_F64, "v_mul_f64", 1, 1, "v_mul_f64 %[c], %[c], %[a]")                                                                  \
  X(ADD_F64, "v_add_f64", 1, 1, "v_add_f64 %[c], %[c], %[a]")                                                                  \
  X(RCP_F64, "v_rcp_f64", 1, 1, "v_rcp_f64 %[c], %[c]")                                                                      \
  X(CVT_F64_I32, "v_cvt_f64_i32", 1, 1, "v_cvt_f64_i32 %[c], %[t]")                                                          \
  X(EXP_F32_VIA_F64, "v_cvt_f32_f64+v_exp_f32+v_cvt_f64_f32", 3, 1, "v_cvt_f32_f64 %[t], %[c]\n\tv_exp_f32 %[t], %[t]\n\tv_cvt_f64_f32 %[c], %[t]")

enum ValuOp {
#define X(tag, name, n, wide, text) OP_##tag,
  AOMHIP_VALU_OPS(X)
#undef X
  OP_COUNT
};
const char *const kValuOpNames[OP_COUNT] = {
#define X(tag, name, n, wide, text) name,
  AOMHIP_VALU_OPS(X)
#undef X
};
const int kValuOpInsts[OP_COUNT] = {
#define X(tag, name, n, wide, text) n,
  AOMHIP_VALU_OPS(X)
#undef X
};
constexpr bool kValuOpWide[OP_COUNT] = {
#define X(tag, name, n, wide, text) wide != 0,
  AOMHIP_VALU_OPS(X)
#undef X
};

// one chain step on chain register c (32-bit chains) / d (64-bit chains); a, b / da, db: loop-invariant VGPR operands; t: a 32-bit temporary
template <int OP> __device__ __forceinline__ void step32(uint32_t &c, uint32_t a, uint32_t b) {
#define X(tag, name, n, wide, text) \
  if constexpr (OP == OP_##tag && !wide) {                                                                                          \
    if constexpr (OP == OP_CMP_CNDMASK || OP == OP_READLANE) asm volatile(text : [c] "+v"(c) : [a] "v"(a), [b] "v"(b), [t] "v"(b) : "vcc", "s20"); \
    else asm volatile(text : [c] "+v"(c) : [a] "v"(a), [b] "v"(b), [t] "v"(b));                                                       \
  }
  AOMHIP_VALU_OPS(X)
#undef X
}
template <int OP> __device__ __forceinline__ void step64(double &d, double da, double db, uint32_t &t) {
#define X(tag, name, n, wide, text) \
  if constexpr (OP == OP_##tag && wide) {                                                                                          \
    if constexpr (OP == OP_MAD_U64_U32 || OP == OP_MAD_I64_I32) asm volatile(text : [c] "+v"(d), [t] "+v"(t) : [a] "v"(da), [b] "v"(db) : "vcc"); \
    else asm volatile(text : [c] "+v"(d), [t] "+v"(t) : [a] "v"(da), [b] "v"(db));                                                    \
  }
  AOMHIP_VALU_OPS(X)
#undef X
}
template <int OP> constexpr bool is64() { return kValuOpWide[OP]; }

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
        if constexpr (is64<OP>()) step64<OP>(d[k], da, db, a); else step32<OP>(c[k], a, b);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  uint32_t x = 0;
#pragma unroll
  for (int k = 0; k < kProbeChains; ++k) x ^= is64<OP>() ? (uint32_t)__double2loint(d[k]) ^ (uint32_t)__double2hiint(d[k]) : c[k];
  if (x == 0x12345678u && sink) sink[0] = x + probe_lds_pad[0];
  if ((tid & 63) == 0 && stamps) {
    unsigned long long *s = stamps + ((size_t)blockIdx.x * (blockDim.x >> 6) + (tid >> 6)) * 4;
    s[0] = t0; s[1] = t1; s[2] = r0; s[3] = r1;
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
  AOMHIP_TRY(hipMalloc(&d_stamps, n_waves * 4 * sizeof(unsigned long long)));
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
  std::vector<unsigned long long> h(n_waves * 4);
  if (e == hipSuccess) e = hipMemcpy(h.data(), d_stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  (void)hipFree(d_stamps);
  if (e != hipSuccess) {
    set_error("aomhip_valu_issue_probe: %s", hipGetErrorString(e));
    return AOMHIP_ERR_HIP;
  }
  const double insts_per_wave = (double)iters * kProbeChains * kProbeReps * kValuOpInsts[op_class];
  // Ticks: the SIMD arbitrates by age (the oldest wavefront that can issue does), so the wavefronts of a SIMD finish one after another and
  // a single wavefront's interval says little; the span from the first start to the last end over the wavefronts of ONE workgroup (one CU,
  // one counter) covers W * insts_per_wave instructions on every SIMD when that workgroup has the CU to itself (W <= 4).  With two
  // workgroups per CU (W = 8) which two share a CU is not known: the figure is then the launch's event time at the measured tick rate.
  const int wpb = (int)block.x / 64;
  std::vector<double> span(grid.x), hzv;
  hzv.reserve(n_waves);
  for (unsigned g = 0; g < grid.x; ++g) {
    unsigned long long t_lo = ~0ull, t_hi = 0;
    for (int w = 0; w < wpb; ++w) {
      const unsigned long long *q = &h[((size_t)g * wpb + w) * 4];
      t_lo = std::min(t_lo, q[0]); t_hi = std::max(t_hi, q[1]);
      if (q[3] > q[2]) hzv.push_back((double)(q[1] - q[0]) / (double)(q[3] - q[2]) * 1e8);
    }
    span[g] = (double)(t_hi - t_lo);
  }
  std::nth_element(span.begin(), span.begin() + span.size() / 2, span.end());
  if (hzv.empty()) hzv.push_back(0.0);
  std::nth_element(hzv.begin(), hzv.begin() + hzv.size() / 2, hzv.end());
  const double hz_med = hzv[hzv.size() / 2];
  const double ticks_per_inst = wgs_per_cu == 1 ? span[span.size() / 2] / (insts_per_wave * (double)waves_per_simd)
                                                : hz_med * (double)ms * 1e-3 / (insts_per_wave * (double)waves_per_simd);
  out->launch_ms = ms;
  out->wave_insts_per_s_per_simd = insts_per_wave * (double)n_waves / ((double)cus * 4.0) / ((double)ms * 1e-3);
  out->memtime_ticks_per_wave_inst = ticks_per_inst;
  out->memtime_hz = hz_med;
  out->waves_per_simd = waves_per_simd;
  out->compute_units = cus;
  return AOMHIP_OK;
}
