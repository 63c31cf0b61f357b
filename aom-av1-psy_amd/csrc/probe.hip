// aomhip_strip_read_probe: the TRANSPORT of sad_strip_kernel with everything else removed -- persistent workgroups pull the same
// strip-shaped windows (rows of sb_w + 2 * range reference pixels and sb_w source pixels at the plane's row pitch, sb_h rows per step,
// the strips of one frame side by side on one XCD) out of memory into registers: no LDS, no barriers, no evaluation.  Its launch time is
// the ceiling of that walk on this box, measured by bench.py in the same run as the kernel (roofline.ceiling_GBs / frac_of_ceiling).
// Measurement support, not part of the encoder path: it computes nothing.  (Origin: tools/strip_read_probe.hip, profiles/r02_strip_read_probe.log.)
#include "common.h"

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
