// CDEF (constrained directional enhancement filter), luma plane, on gfx950.
// Reference: av1/common/cdef_block.c:57-133 (cdef_find_dir), :139-201 (cdef_filter_block_internal),
// :289-293 (adjust_strength), :323-426 (av1_cdef_filter_fb); av1/common/cdef.h:59-67 (constrain);
// per-64x64 driver av1/common/cdef.c:138-345 (cdef_prepare_fb / cdef_fb_col).
//
// The reference filters in place and keeps line / column buffers so that every tap reads PRE-CDEF
// (deblocked) pixels, with CDEF_VERY_LARGE outside the frame.  An out-of-place kernel (read the deblocked
// plane, write the CDEF plane) has exactly those semantics with no buffers.
//
// One workgroup per 64x64 filter block:
//   1. the 68 x 72 footprint (2-row / 4-column halo) is staged in LDS as uint16 with 8-byte loads, 0x4000 outside
//      the frame
//   2. direction search: lane = 8x8 block, each of the four waves evaluates two of the eight directional
//      partial-sum sets for all 64 blocks (pixels in VGPRs, compile-time line indices); 64 lanes then pick the best
//   3. filter: lane = pixel column, each wave owns 16 rows: conflict-free LDS tap reads, one contiguous 64-pixel
//      store per row, no divergent tap code (a disabled strength makes its taps contribute 0)
// Skipped 8x8 blocks (all four 4x4 skip_txfm) and blocks of a zero-strength filter block are copied.
// Algorithmic bytes: one read and one write per pixel.
#include "common.h"

namespace aomhip {

constexpr int kVeryLarge = 0x4000;  // CDEF_VERY_LARGE (cdef_block.h:31)
constexpr int kTW = 72, kTH = 68;   // LDS tile: rows -2..65, cols -4..67

__device__ __forceinline__ int msb_u(unsigned v) { return 31 - __clz((int)v); }  // get_msb, v != 0

__device__ __forceinline__ int constrain_d(int diff, int threshold, int damping) {
  if (!threshold) return 0;
  int shift = damping - msb_u((unsigned)threshold);
  shift = shift < 0 ? 0 : shift;
  const int a = diff < 0 ? -diff : diff;
  int m = threshold - (a >> shift);
  m = m < 0 ? 0 : m;
  m = m > a ? a : m;
  return diff < 0 ? -m : m;
}

// line index of pixel (i, j) for direction D (cdef_block.c:71-87)
template <int D> __device__ __forceinline__ constexpr int dir_line(int i, int j) {
  return D == 0 ? i + j
       : D == 1 ? i + j / 2
       : D == 2 ? i
       : D == 3 ? 3 + i - j / 2
       : D == 4 ? 7 + i - j
       : D == 5 ? 3 - i / 2 + j
       : D == 6 ? j
                : i / 2 + j;
}

template <int D> __device__ __forceinline__ int dir_cost(const int (&x)[64]) {
  int partial[15];
#pragma unroll
  for (int k = 0; k < 15; ++k) partial[k] = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) partial[dir_line<D>(i, j)] += x[i * 8 + j];
  // weights 840 / (pixels on the line): div_table (cdef_block.c:67)
  constexpr int div_table[9] = { 0, 840, 420, 280, 210, 168, 140, 120, 105 };
  // |partial| <= 8 * 128, so partial^2 < 2^21 and every product below fits the 24-bit multiplier (v_mul_i32_i24 / v_mad_i32_i24 run at
  // full rate; the 32-bit v_mul_lo_u32 the plain `*` compiled to is a quarter-rate instruction, 69 of them per wavefront)
  auto sq = [](int v) { return __mul24(v, v); };
  int cost = 0;
  if constexpr (D == 2 || D == 6) {
#pragma unroll
    for (int k = 0; k < 8; ++k) cost += sq(partial[k]);
    cost = (int)__umul24((unsigned)cost, div_table[8]);   // cost <= 8 * 2^20 = 2^23: inside the UNSIGNED 24-bit range (all-zero pixels reach it)
  } else if constexpr (D == 0 || D == 4) {
#pragma unroll
    for (int k = 0; k < 7; ++k) cost += (int)__umul24((unsigned)(sq(partial[k]) + sq(partial[14 - k])), div_table[k + 1]);
    cost += (int)__umul24((unsigned)sq(partial[7]), div_table[8]);
  } else {
#pragma unroll
    for (int k = 0; k < 5; ++k) cost += sq(partial[3 + k]);
    cost = (int)__umul24((unsigned)cost, div_table[8]);
#pragma unroll
    for (int k = 0; k < 3; ++k)
      cost += (int)__umul24((unsigned)(sq(partial[k]) + sq(partial[10 - k])), div_table[2 * k + 2]);
  }
  return cost;
}

// Cdef_Directions (AV1 spec 7.15.3; cdef_block.c:25-48) as LDS offsets dy * kTW + dx for taps k = 0, 1
__device__ constexpr int kDirOff[8][2] = {
  { -1 * kTW + 1, -2 * kTW + 2 }, { 0 * kTW + 1, -1 * kTW + 2 }, { 0 * kTW + 1, 0 * kTW + 2 }, { 0 * kTW + 1, 1 * kTW + 2 },
  { 1 * kTW + 1, 2 * kTW + 2 },   { 1 * kTW + 0, 2 * kTW + 1 },  { 1 * kTW + 0, 2 * kTW + 0 }, { 1 * kTW + 0, 2 * kTW - 1 }
};

// Packed int16 pairs (two pixels per VGPR; clang lowers the element-wise operators to v_pk_* on gfx950)
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 splat2(int v) { return s16x2{ (short)v, (short)v }; }
__device__ __forceinline__ s16x2 pack2(int lo, int hi) { return s16x2{ (short)lo, (short)hi }; }
__device__ __forceinline__ s16x2 pmax(s16x2 a, s16x2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ s16x2 pmin(s16x2 a, s16x2 b) { return __builtin_elementwise_min(a, b); }
// constrain() (av1/common/cdef.h:59-67) on a pair, with shift = max(0, damping - msb(threshold)) hoisted out of it;
// threshold 0 yields 0 by itself (m = -(a >> shift) <= 0)
// sign(d) min(|d|, max(0, thr - (|d| >> shift))) written as a clamp of d to [-lim, lim] with lim = thr -sat (|d| >> shift) (unsigned saturating
// subtract: one v_pk_sub_u16 clamp): 7 packed operations instead of 10 -- the kernel is VALU-bound and runs this 12 times per pixel pair.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 constrain_pk(s16x2 diff, s16x2 thr, s16x2 shift) {
  const s16x2 a = pmax(diff, -diff);
  const s16x2 lim = __builtin_bit_cast(s16x2, __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, thr), __builtin_bit_cast(u16x2, a >> shift)));
  return pmax(pmin(diff, lim), -lim);
}

// constrain() with the shift (damping - msb(threshold), floored at 0) hoisted out; threshold 0 gives 0 by itself
__device__ __forceinline__ int constrain_s(int diff, int threshold, int shift) {
  const int a = diff < 0 ? -diff : diff;
  int m = threshold - (a >> shift);
  m = m < 0 ? 0 : m;
  m = m > a ? a : m;
  return diff < 0 ? -m : m;
}

// SEARCH = false: the filtering pass (dst = CDEF(src) with each filter block's strengths).
// SEARCH = true : the distortion table of av1_cdef_search (av1/encoder/pickcdef.c:401-615 get_filt_error /
//   av1_cdef_mse_calc_block): the footprint is staged and the directions searched ONCE, then every (pri, sec) of the
//   strength list is applied to the filter block and the squared error against the source frame `orig` is summed over
//   the non-skip 8x8 blocks -- nothing is written but n_strengths sums per filter block (fb_pri = the strength pairs,
//   fb_sec unused, dst unused).
struct CdefSearchArgs {
  const void *orig;   // the source frame the reconstruction is compared with, pixel (0, 0), same pixel type
  int orig_stride;
  int n_strengths;
  uint64_t *sse;      // [n_strengths][grid.y * fb_stride + grid.x ...] = [gi][fby * fb_stride + fbx]
};

template <typename PIX, bool SEARCH>
__global__ __launch_bounds__(256) void cdef_luma_kernel(const PIX *__restrict__ src, PIX *__restrict__ dst, int stride,
                                                        int width, int height, const uint8_t *__restrict__ fb_pri,
                                                        const uint8_t *__restrict__ fb_sec, int fb_stride,
                                                        const uint8_t *__restrict__ skip, int damping, int coeff_shift,
                                                        uint8_t *__restrict__ dir_out, int32_t *__restrict__ var_out,
                                                        CdefSearchArgs sa_) {
  __shared__ __attribute__((aligned(16))) uint16_t tile[kTH * kTW];
  __shared__ int32_t scost[8][64];
  __shared__ int8_t sdir[64];
  __shared__ int32_t svar[64];
  __shared__ unsigned long long swave[4];
  const int fbx = blockIdx.x, fby = blockIdx.y;
  const int x0 = fbx * 64, y0 = fby * 64;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = tid & 63;
  const int b8w = width >> 3;

  // 1. stage the 68 x 72 footprint, four pixels per lane and step.  The frame width is a multiple of 8 and a group
  //    starts at a multiple of 4, so a group is entirely inside or entirely outside the frame.
  {
    constexpr int kGroups = kTW / 4;  // 18 per row
    const bool vec_ok = ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)((size_t)stride * sizeof(PIX))) & (4 * sizeof(PIX) - 1)) == 0;
    for (int gi = tid; gi < kTH * kGroups; gi += 256) {
      const int r = gi / kGroups - 2, c = (gi % kGroups) * 4 - 4;
      const int y = y0 + r, x = x0 + c;
      uint32_t lo = kVeryLarge | (kVeryLarge << 16), hi = lo;
      if (y >= 0 && y < height && x >= 0 && x < width) {
        const PIX *p = src + (int64_t)y * stride + x;
        if constexpr (sizeof(PIX) == 2) {
          if (vec_ok) {
            const uint2 v = *reinterpret_cast<const uint2 *>(p);
            lo = v.x;
            hi = v.y;
          } else {
            lo = (uint32_t)p[0] | ((uint32_t)p[1] << 16);
            hi = (uint32_t)p[2] | ((uint32_t)p[3] << 16);
          }
        } else {
          uint32_t v;
          if (vec_ok) v = *reinterpret_cast<const uint32_t *>(p);
          else v = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
          lo = (v & 0xffu) | ((v & 0xff00u) << 8);
          hi = ((v >> 16) & 0xffu) | ((v >> 8) & 0xff0000u);
        }
      }
      *reinterpret_cast<uint2 *>(&tile[gi * 4]) = make_uint2(lo, hi);
    }
  }
  int level = 0, sec = 0;
  if constexpr (!SEARCH) {
    level = fb_pri[fby * fb_stride + fbx];
    sec = fb_sec[fby * fb_stride + fbx];
  }
  __syncthreads();

  // 2. direction search: lane = 8x8 block, wave w evaluates directions 2w and 2w + 1 of all 64 blocks
  {
    const int by = lane >> 3, bx = lane & 7;
    int x[64];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint2 v0 = *reinterpret_cast<const uint2 *>(&tile[(by * 8 + i + 2) * kTW + bx * 8 + 4]);
      const uint2 v1 = *reinterpret_cast<const uint2 *>(&tile[(by * 8 + i + 2) * kTW + bx * 8 + 8]);
      const uint32_t w4[4] = { v0.x, v0.y, v1.x, v1.y };
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        x[i * 8 + 2 * k] = ((int)(w4[k] & 0xffffu) >> coeff_shift) - 128;
        x[i * 8 + 2 * k + 1] = ((int)(w4[k] >> 16) >> coeff_shift) - 128;
      }
    }
    int ca, cb;
    if (wave == 0) { ca = dir_cost<0>(x); cb = dir_cost<1>(x); }
    else if (wave == 1) { ca = dir_cost<2>(x); cb = dir_cost<3>(x); }
    else if (wave == 2) { ca = dir_cost<4>(x); cb = dir_cost<5>(x); }
    else { ca = dir_cost<6>(x); cb = dir_cost<7>(x); }
    scost[2 * wave][lane] = ca;
    scost[2 * wave + 1][lane] = cb;
  }
  __syncthreads();
  if (tid < 64) {
    const int by = tid >> 3, bx = tid & 7;
    const int gy = y0 + by * 8, gx = x0 + bx * 8;
    int d = -1, var = 0;
    // (a zero-strength filter block is still searched when the caller wants the directions: its chroma strengths
    //  may be non-zero, cdef.c:334-345; filtering with zero strengths below is the identity)
    if (gy < height && gx < width && (SEARCH || (level | sec) != 0 || dir_out || var_out) && !skip[(gy >> 3) * b8w + (gx >> 3)]) {
      int cost[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) cost[k] = scost[k][tid];
      int best = 0, best_cost = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (cost[k] > best_cost) {
          best_cost = cost[k];
          best = k;
        }
      int orth = cost[0];
#pragma unroll
      for (int k = 1; k < 8; ++k) orth = (((best + 4) & 7) == k) ? cost[k] : orth;
      var = (best_cost - orth) >> 10;
      d = best;
    }
    sdir[tid] = (int8_t)d;
    svar[tid] = var;
    if (gy < height && gx < width) {
      if (dir_out) dir_out[(gy >> 3) * b8w + (gx >> 3)] = d < 0 ? 0 : (uint8_t)d;
      if (var_out) var_out[(gy >> 3) * b8w + (gx >> 3)] = var;
    }
  }
  __syncthreads();

  // 3. filter: lane = pixel column, wave w owns rows 16w .. 16w + 15 (two rows of 8x8 blocks).  Adjacent lanes read
  //    adjacent LDS elements and write one contiguous 64-pixel row segment per step.  Two vertically adjacent pixels
  //    are processed together as packed int16 pairs (v_pk_* ops: the kernel is VALU-bound, PMC in
  //    profiles/r01_cdef.md) -- every intermediate of cdef_filter_block_internal fits int16, which is also what the
  //    reference computes in.  Disabled taps contribute nothing by themselves (constrain with strength 0 is 0), so the
  //    per-block enables only select whether the final clamp applies: no divergent tap code.
  const int gx = x0 + lane;
  if constexpr (!SEARCH) {
    if (gx >= width) return;
  }
  // does the staged footprint (rows y0 - 2 .. y0 + 65, columns x0 - 4 .. x0 + 67) reach outside the frame (CDEF_VERY_LARGE entries)?
  const bool fb_on_edge = x0 == 0 || y0 == 0 || x0 + 64 + 4 > width || y0 + 64 + 2 > height;
  // SEARCH: this lane's 16 source pixels (rows 16 wave .. 16 wave + 15 of column gx) stay in registers for all strengths
  [[maybe_unused]] int og[16];
  if constexpr (SEARCH) {
    const PIX *op = static_cast<const PIX *>(sa_.orig);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int gy = y0 + wave * 16 + r;
      og[r] = (gx < width && gy < height) ? (int)op[(int64_t)gy * sa_.orig_stride + gx] : 0;
    }
  }
  const int n_iter = SEARCH ? sa_.n_strengths : 1;
#pragma unroll 1
  for (int gi = 0; gi < n_iter; ++gi) {
  if constexpr (SEARCH) {
    level = fb_pri[2 * gi];
    sec = fb_pri[2 * gi + 1];
  }
  [[maybe_unused]] uint32_t err = 0;
  const int pri_strength = level << coeff_shift, sec_strength = sec << coeff_shift;
  const int dmp = damping + coeff_shift;
  const int sec_shift = sec_strength ? max(0, dmp - msb_u((unsigned)sec_strength)) : 0;
  const s16x2 sec_thr = splat2(sec_strength), sec_sh = splat2(sec_shift);
#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
    const int by = wave * 2 + half, bx = lane >> 3, blk = by * 8 + bx;
    if (y0 + by * 8 >= height || gx >= width) break;
    const int d = sdir[blk];
    int t = 0;
    if (d >= 0) {  // adjust_strength (cdef_block.c:289-293)
      const int var = svar[blk];
      const int i = (var >> 6) ? min(msb_u((unsigned)(var >> 6)), 12) : 0;
      t = var ? (pri_strength * (4 + i) + 8) >> 4 : 0;
    }
    const int pri_shift = t ? max(0, dmp - msb_u((unsigned)t)) : 0;
    const int dir = pri_strength ? (d < 0 ? 0 : d) : 0;
    const bool clip = (t != 0) && (sec_strength != 0);
    const s16x2 pt0 = splat2(((t >> coeff_shift) & 1) ? 3 : 4), pt1 = splat2(((t >> coeff_shift) & 1) ? 3 : 2);  // cdef_pri_taps
    const s16x2 pri_thr = splat2(t), pri_sh = splat2(pri_shift);
    const int po0 = kDirOff[dir][0], po1 = kDirOff[dir][1];
    const int s1o0 = kDirOff[(dir + 2) & 7][0], s1o1 = kDirOff[(dir + 2) & 7][1];
    const int s2o0 = kDirOff[(dir + 6) & 7][0], s2o1 = kDirOff[(dir + 6) & 7][1];
#pragma unroll 2
    for (int rr = 0; rr < 8; rr += 2) {
      const int ly = by * 8 + rr;
      const int pos = (ly + 2) * kTW + lane + 4;
      auto ld2 = [&](int off) { return pack2(tile[pos + off], tile[pos + kTW + off]); };  // rows ly and ly + 1
      const s16x2 x = ld2(0);
      s16x2 y = x;
      if (d >= 0) {
        const s16x2 p0 = ld2(po0), p1 = ld2(-po0), p2 = ld2(po1), p3 = ld2(-po1);
        const s16x2 a0 = ld2(s1o0), a1 = ld2(-s1o0), a2 = ld2(s2o0), a3 = ld2(-s2o0);
        const s16x2 c0 = ld2(s1o1), c1 = ld2(-s1o1), c2 = ld2(s2o1), c3 = ld2(-s2o1);
        s16x2 sum = pt0 * (constrain_pk(p0 - x, pri_thr, pri_sh) + constrain_pk(p1 - x, pri_thr, pri_sh)) +
                    pt1 * (constrain_pk(p2 - x, pri_thr, pri_sh) + constrain_pk(p3 - x, pri_thr, pri_sh));
        const s16x2 sa = constrain_pk(a0 - x, sec_thr, sec_sh) + constrain_pk(a1 - x, sec_thr, sec_sh) +
                         constrain_pk(a2 - x, sec_thr, sec_sh) + constrain_pk(a3 - x, sec_thr, sec_sh);
        const s16x2 sc = constrain_pk(c0 - x, sec_thr, sec_sh) + constrain_pk(c1 - x, sec_thr, sec_sh) +
                         constrain_pk(c2 - x, sec_thr, sec_sh) + constrain_pk(c3 - x, sec_thr, sec_sh);
        sum += sa + sa + sc;
        // y = x + ((8 + sum - (sum < 0)) >> 4): (sum < 0) is -(sum >> 15)
        y = x + ((splat2(8) + sum + (sum >> splat2(15))) >> splat2(4));
        if (clip) {
          // CDEF_VERY_LARGE (0x4000) must not enter the maximum: & 0x3fff turns it into 0 and leaves pixels alone.  Only a filter block
          // on the frame's edge has such taps staged; the others (91 % of a 4K frame) skip the twelve ANDs.
          s16x2 mx;
          if (fb_on_edge) {
            const s16x2 k = splat2(0x3fff);
            mx = pmax(pmax(pmax(x, p0 & k), pmax(p1 & k, p2 & k)), pmax(pmax(p3 & k, a0 & k), pmax(a1 & k, a2 & k)));
            mx = pmax(pmax(pmax(mx, a3 & k), pmax(c0 & k, c1 & k)), pmax(c2 & k, c3 & k));
          } else {
            mx = pmax(pmax(pmax(x, p0), pmax(p1, p2)), pmax(pmax(p3, a0), pmax(a1, a2)));
            mx = pmax(pmax(pmax(mx, a3), pmax(c0, c1)), pmax(c2, c3));
          }
          s16x2 mn = pmin(pmin(pmin(x, p0), pmin(p1, p2)), pmin(pmin(p3, a0), pmin(a1, a2)));
          mn = pmin(pmin(pmin(mn, a3), pmin(c0, c1)), pmin(c2, c3));
          y = pmin(pmax(y, mn), mx);
        }
      }
      if constexpr (SEARCH) {
        if (d >= 0) {  // only the blocks of the filter list count (compute_cdef_dist*, pickcdef.c:237-315)
          const int e0 = og[half * 8 + rr] - (int)y.x, e1 = og[half * 8 + rr + 1] - (int)y.y;
          err += (uint32_t)(e0 * e0) + (uint32_t)(e1 * e1);
        }
      } else {
        PIX *o = dst + (int64_t)(y0 + ly) * stride + gx;
        o[0] = (PIX)(uint16_t)y.x;
        o[stride] = (PIX)(uint16_t)y.y;
      }
    }
  }
  if constexpr (SEARCH) {
    unsigned long long tot = err;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) tot += __shfl_xor(tot, m, 64);
    __syncthreads();  // the previous strength's swave has been consumed
    if (lane == 0) swave[wave] = tot;
    __syncthreads();
    if (tid == 0)
      sa_.sse[(int64_t)gi * gridDim.y * fb_stride + fby * fb_stride + fbx] = swave[0] + swave[1] + swave[2] + swave[3];
  }
  }  // strengths
}

// Chroma planes: av1_cdef_filter_fb with pli > 0 (cdef_block.c:323-426).  A luma 8x8 block maps to a
// (8 >> XDEC) x (8 >> YDEC) chroma block; the direction comes from luma (converted for 4:2:2 / 4:4:0, :362-371),
// the primary strength is not variance-adjusted and the damping is one less (:333).  Same structure as the luma
// kernel without the direction search: one workgroup per filter block, footprint in LDS, 4 lanes per block.
__device__ constexpr int8_t kDirDyDx[8][2][2] = { { { -1, 1 }, { -2, 2 } }, { { 0, 1 }, { -1, 2 } }, { { 0, 1 }, { 0, 2 } },
                                                  { { 0, 1 }, { 1, 2 } },   { { 1, 1 }, { 2, 2 } },  { { 1, 0 }, { 2, 1 } },
                                                  { { 1, 0 }, { 2, 0 } },   { { 1, 0 }, { 2, -1 } } };

// SEARCH as in cdef_luma_kernel: fb_pri = the (pri, sec) list, only the per-strength squared-error sums are written.
template <typename PIX, int XDEC, int YDEC, bool SEARCH>
__global__ __launch_bounds__(256) void cdef_chroma_kernel(const PIX *__restrict__ src, PIX *__restrict__ dst, int stride,
                                                          int width, int height, const uint8_t *__restrict__ luma_dir,
                                                          const uint8_t *__restrict__ fb_pri,
                                                          const uint8_t *__restrict__ fb_sec, int fb_stride,
                                                          const uint8_t *__restrict__ skip, int damping,
                                                          int coeff_shift, CdefSearchArgs sa_) {
  constexpr int BW = 8 >> XDEC, BH = 8 >> YDEC;    // chroma block of one luma 8x8
  constexpr int FW = 64 >> XDEC, FH = 64 >> YDEC;  // chroma filter block
  constexpr int TW = FW + 8, TH = FH + 4;          // LDS tile: rows -2..FH+1, cols -4..FW+3
  __shared__ uint16_t tile[TH * TW];
  __shared__ unsigned long long swave[4];
  const int fbx = blockIdx.x, fby = blockIdx.y;
  const int x0 = fbx * FW, y0 = fby * FH;
  const int tid = threadIdx.x;
  const int nbx = width / BW;
  for (int i = tid; i < TH * TW; i += 256) {
    const int r = i / TW - 2, c = i % TW - 4;
    const int y = y0 + r, x = x0 + c;
    int v = kVeryLarge;
    if (y >= 0 && y < height && x >= 0 && x < width) v = src[(int64_t)y * stride + x];
    tile[i] = (uint16_t)v;
  }
  int level = 0, sec = 0;
  if constexpr (!SEARCH) {
    level = fb_pri[fby * fb_stride + fbx];
    sec = fb_sec[fby * fb_stride + fbx];
  }
  __syncthreads();

  // filter: lane = pixel column (two rows of lanes per wavefront when the filter block is 32 wide), two vertically
  // adjacent pixels per lane as packed int16 -- the luma kernel's scheme (profiles/r01_cdef.md)
  constexpr int kLanesPerRow = FW;                       // 64 or 32
  constexpr int kSub = 64 / kLanesPerRow;                // row pairs a wavefront handles per step
  constexpr int kRowsPerWave = FH / 4;                   // 16 or 8
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = tid & 63;
  const int col = lane % kLanesPerRow, rsub = lane / kLanesPerRow;
  const int gx = x0 + col;
  if constexpr (!SEARCH) {
    if (gx >= width) return;
  }
  const int n_iter = SEARCH ? sa_.n_strengths : 1;
#pragma unroll 1
  for (int gi = 0; gi < n_iter; ++gi) {
  if constexpr (SEARCH) {
    level = fb_pri[2 * gi];
    sec = fb_pri[2 * gi + 1];
  }
  [[maybe_unused]] uint32_t err = 0;
  const int pri_strength = level << coeff_shift, sec_strength = sec << coeff_shift;
  const int dmp = damping + coeff_shift - 1;
  const int t = pri_strength;
  const int pri_shift = t ? max(0, dmp - msb_u((unsigned)t)) : 0;
  const int sec_shift = sec_strength ? max(0, dmp - msb_u((unsigned)sec_strength)) : 0;
  const bool clip = (t != 0) && (sec_strength != 0);
  const s16x2 pri_thr = splat2(t), pri_sh = splat2(pri_shift), sec_thr = splat2(sec_strength), sec_sh = splat2(sec_shift);
  const s16x2 pt0 = splat2(((t >> coeff_shift) & 1) ? 3 : 4), pt1 = splat2(((t >> coeff_shift) & 1) ? 3 : 2);
  auto off = [&](int d, int k) { return kDirDyDx[d][k][0] * TW + kDirDyDx[d][k][1]; };
#pragma unroll 1
  for (int step = 0; step < kRowsPerWave / (2 * kSub); ++step) {
    const int ly = wave * kRowsPerWave + (step * kSub + rsub) * 2;   // rows ly, ly + 1 (same chroma block: BH is 4 or 8)
    const int gy = y0 + ly;
    if (gy >= height || gx >= width) continue;
    const int bidx = (gy / BH) * nbx + gx / BW;
    const bool filt = (level | sec) != 0 && !skip[bidx];
    int dir = luma_dir[bidx] & 7;
    if constexpr (XDEC != YDEC) {
      constexpr int conv422[8] = { 7, 0, 2, 4, 5, 6, 6, 6 }, conv440[8] = { 1, 2, 2, 2, 3, 4, 6, 0 };
      int c = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) c = dir == k ? (XDEC ? conv422[k] : conv440[k]) : c;
      dir = c;
    }
    dir = pri_strength ? dir : 0;
    const int po0 = off(dir, 0), po1 = off(dir, 1);
    const int s1o0 = off((dir + 2) & 7, 0), s1o1 = off((dir + 2) & 7, 1);
    const int s2o0 = off((dir + 6) & 7, 0), s2o1 = off((dir + 6) & 7, 1);
    const int pos = (ly + 2) * TW + col + 4;
    auto ld2 = [&](int o) { return pack2(tile[pos + o], tile[pos + TW + o]); };
    const s16x2 x = ld2(0);
    s16x2 y = x;
    if (filt) {
      const s16x2 p0 = ld2(po0), p1 = ld2(-po0), p2 = ld2(po1), p3 = ld2(-po1);
      const s16x2 a0 = ld2(s1o0), a1 = ld2(-s1o0), a2 = ld2(s2o0), a3 = ld2(-s2o0);
      const s16x2 c0 = ld2(s1o1), c1 = ld2(-s1o1), c2 = ld2(s2o1), c3 = ld2(-s2o1);
      s16x2 sum = pt0 * (constrain_pk(p0 - x, pri_thr, pri_sh) + constrain_pk(p1 - x, pri_thr, pri_sh)) +
                  pt1 * (constrain_pk(p2 - x, pri_thr, pri_sh) + constrain_pk(p3 - x, pri_thr, pri_sh));
      const s16x2 sa = constrain_pk(a0 - x, sec_thr, sec_sh) + constrain_pk(a1 - x, sec_thr, sec_sh) +
                       constrain_pk(a2 - x, sec_thr, sec_sh) + constrain_pk(a3 - x, sec_thr, sec_sh);
      const s16x2 sc = constrain_pk(c0 - x, sec_thr, sec_sh) + constrain_pk(c1 - x, sec_thr, sec_sh) +
                       constrain_pk(c2 - x, sec_thr, sec_sh) + constrain_pk(c3 - x, sec_thr, sec_sh);
      sum += sa + sa + sc;
      y = x + ((splat2(8) + sum + (sum >> splat2(15))) >> splat2(4));
      if (clip) {
        const s16x2 k = splat2(0x3fff);
        s16x2 mx = pmax(pmax(pmax(x, p0 & k), pmax(p1 & k, p2 & k)), pmax(pmax(p3 & k, a0 & k), pmax(a1 & k, a2 & k)));
        mx = pmax(pmax(pmax(mx, a3 & k), pmax(c0 & k, c1 & k)), pmax(c2 & k, c3 & k));
        s16x2 mn = pmin(pmin(pmin(x, p0), pmin(p1, p2)), pmin(pmin(p3, a0), pmin(a1, a2)));
        mn = pmin(pmin(pmin(mn, a3), pmin(c0, c1)), pmin(c2, c3));
        y = pmin(pmax(y, mn), mx);
      }
    }
    if constexpr (SEARCH) {
      if (!skip[bidx]) {  // only the units of the filter list count (compute_cdef_dist*, pickcdef.c:237-315)
        const PIX *op = static_cast<const PIX *>(sa_.orig) + (int64_t)gy * sa_.orig_stride + gx;
        const int e0 = (int)op[0] - (int)y.x, e1 = (int)op[sa_.orig_stride] - (int)y.y;
        err += (uint32_t)(e0 * e0) + (uint32_t)(e1 * e1);
      }
    } else {
      PIX *o = dst + (int64_t)gy * stride + gx;
      o[0] = (PIX)(uint16_t)y.x;
      o[stride] = (PIX)(uint16_t)y.y;
    }
  }
  if constexpr (SEARCH) {
    unsigned long long tot = err;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) tot += __shfl_xor(tot, m, 64);
    __syncthreads();
    if (lane == 0) swave[wave] = tot;
    __syncthreads();
    if (tid == 0)
      sa_.sse[(int64_t)gi * gridDim.y * fb_stride + fby * fb_stride + fbx] = swave[0] + swave[1] + swave[2] + swave[3];
  }
  }  // strengths
}

}  // namespace aomhip

using namespace aomhip;


// ---- the reference's block-level primitives on staged blocks (rtcd-signature entry points) ----
// cdef_find_dir_c (cdef_block.c:57-126) for `n` 8x8 blocks (n = 2: cdef_find_dir_dual_c), lane = block.
__global__ __launch_bounds__(64) void cdef_find_dir_kernel(const uint16_t *__restrict__ img, int n, int coeff_shift, int32_t *__restrict__ out) {
  const int b = threadIdx.x;
  if (b >= n) return;
  int x[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) x[i] = (int)(img[b * 64 + i] >> coeff_shift) - 128;
  int cost[8];
  cost[0] = dir_cost<0>(x); cost[1] = dir_cost<1>(x); cost[2] = dir_cost<2>(x); cost[3] = dir_cost<3>(x);
  cost[4] = dir_cost<4>(x); cost[5] = dir_cost<5>(x); cost[6] = dir_cost<6>(x); cost[7] = dir_cost<7>(x);
  int best = 0, best_cost = 0;
#pragma unroll
  for (int d = 0; d < 8; ++d)
    if (cost[d] > best_cost) { best_cost = cost[d]; best = d; }
  int orth = 0;
#pragma unroll
  for (int d = 0; d < 8; ++d) orth = d == ((best + 4) & 7) ? cost[d] : orth;
  out[2 * b] = best;
  out[2 * b + 1] = (best_cost - orth) >> 10;
}

// cdef_filter_block_internal (cdef_block.c:139-281) = cdef_filter_{8,16}_{0..3}_c: `in` is the staged (bh + 4) x (bw + 4)
// neighbourhood (2 pixels on every side, row pitch bw + 4) of the uint16 input whose stride was CDEF_BSTRIDE; lane = pixel.
__global__ __launch_bounds__(64) void cdef_filter_block_kernel(const uint16_t *__restrict__ in, uint16_t *__restrict__ out, int bw, int bh,
                                                               int pri_strength, int sec_strength, int dir, int pri_damping,
                                                               int sec_damping, int coeff_shift, int enable_primary, int enable_secondary) {
  const int t = threadIdx.x;
  if (t >= bw * bh) return;
  const int i = t / bw, j = t % bw, s = bw + 4;
  const uint16_t *c = in + (i + 2) * s + (j + 2);
  constexpr int pri_taps[2][2] = { { 4, 2 }, { 3, 3 } };
  constexpr int sec_taps[2] = { 2, 1 };
  const int ps = (pri_strength >> coeff_shift) & 1;
  const bool clip = enable_primary && enable_secondary;
  const int x = c[0];
  int sum = 0, mx = x, mn = x;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    if (enable_primary) {
      const int o = kDirDyDx[dir][k][0] * s + kDirDyDx[dir][k][1];
      const int p0 = c[o], p1 = c[-o];
      sum += pri_taps[ps][k] * (constrain_d(p0 - x, pri_strength, pri_damping) + constrain_d(p1 - x, pri_strength, pri_damping));
      if (clip) {
        if (p0 != kVeryLarge) mx = max(mx, p0);
        if (p1 != kVeryLarge) mx = max(mx, p1);
        mn = min(mn, min(p0, p1));
      }
    }
    if (enable_secondary) {
      const int o1 = kDirDyDx[(dir + 2) & 7][k][0] * s + kDirDyDx[(dir + 2) & 7][k][1];
      const int o2 = kDirDyDx[(dir + 6) & 7][k][0] * s + kDirDyDx[(dir + 6) & 7][k][1];
      const int q[4] = { c[o1], c[-o1], c[o2], c[-o2] };
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        if (clip) {
          if (q[m] != kVeryLarge) mx = max(mx, q[m]);
          mn = min(mn, q[m]);
        }
        sum += sec_taps[k] * constrain_d(q[m] - x, sec_strength, sec_damping);
      }
    }
  }
  int y = (int)(int16_t)x + ((8 + (int)(int16_t)sum - ((int16_t)sum < 0)) >> 4);
  if (clip) y = y < mn ? mn : (y > mx ? mx : y);
  out[t] = (uint16_t)y;
}

template <typename PIX, bool SEARCH>
static void launch_chroma(hipStream_t st, dim3 grid, int xdec, int ydec, const void *s, void *d, int stride, int w, int h,
                          const uint8_t *dir, const uint8_t *pri, const uint8_t *sec, int fbs, const uint8_t *skip,
                          int damping, int cs, const CdefSearchArgs &sa = CdefSearchArgs{}) {
#define AOMHIP_CDEF_C(X, Y)                                                                                          \
  hipLaunchKernelGGL((cdef_chroma_kernel<PIX, X, Y, SEARCH>), grid, dim3(256), 0, st, static_cast<const PIX *>(s),   \
                     static_cast<PIX *>(d), stride, w, h, dir, pri, sec, fbs, skip, damping, cs, sa)
  if (xdec == 1 && ydec == 1) AOMHIP_CDEF_C(1, 1);
  else if (xdec == 0 && ydec == 0) AOMHIP_CDEF_C(0, 0);
  else if (xdec == 1 && ydec == 0) AOMHIP_CDEF_C(1, 0);
  else AOMHIP_CDEF_C(0, 1);
#undef AOMHIP_CDEF_C
}

extern "C" {

int aomhip_cdef_luma_plane(aomhip_ctx *ctx, const aomhip_planes *src, int src_frame, const aomhip_planes *dst,
                           int dst_frame, const uint8_t *d_fb_pri, const uint8_t *d_fb_sec, int fb_stride,
                           const uint8_t *d_skip8x8, int damping, uint8_t *d_dir_out, int32_t *d_var_out) {
  if (!ctx || !src || !dst || !src->base || !dst->base || !d_fb_pri || !d_fb_sec || !d_skip8x8 || src_frame < 0 ||
      src_frame >= src->n_frames || dst_frame < 0 || dst_frame >= dst->n_frames || src->width != dst->width ||
      src->height != dst->height || src->stride != dst->stride || src->bit_depth != dst->bit_depth ||
      (src->width & 7) || (src->height & 7) || damping < 3 || damping > 6 || fb_stride < (src->width + 63) / 64 ||
      (src->base == dst->base && src_frame == dst_frame)) {
    set_error("aomhip_cdef_luma_plane: invalid argument (dimensions must be multiples of 8, out of place)");
    return AOMHIP_ERR_INVALID;
  }
  const size_t esz = src->bit_depth == 8 ? 1 : 2;
  const char *s = static_cast<const char *>(src->base) +
                  ((size_t)src_frame * src->frame_stride + (size_t)src->border * src->stride + src->border) * esz;
  char *d = static_cast<char *>(dst->base) +
            ((size_t)dst_frame * dst->frame_stride + (size_t)dst->border * dst->stride + dst->border) * esz;
  const dim3 grid((src->width + 63) / 64, (src->height + 63) / 64);
  if (esz == 1)
    hipLaunchKernelGGL((cdef_luma_kernel<uint8_t, false>), grid, dim3(256), 0, ctx->stream, reinterpret_cast<const uint8_t *>(s),
                       reinterpret_cast<uint8_t *>(d), src->stride, src->width, src->height, d_fb_pri, d_fb_sec,
                       fb_stride, d_skip8x8, damping, 0, d_dir_out, d_var_out, CdefSearchArgs{});
  else
    hipLaunchKernelGGL((cdef_luma_kernel<uint16_t, false>), grid, dim3(256), 0, ctx->stream,
                       reinterpret_cast<const uint16_t *>(s), reinterpret_cast<uint16_t *>(d), src->stride, src->width,
                       src->height, d_fb_pri, d_fb_sec, fb_stride, d_skip8x8, damping, src->bit_depth - 8, d_dir_out,
                       d_var_out, CdefSearchArgs{});
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}

int aomhip_cdef_search_sse_luma(aomhip_ctx *ctx, const aomhip_planes *recon, int recon_frame, const aomhip_planes *source,
                                int source_frame, const uint8_t *d_strengths, int n_strengths, const uint8_t *d_skip8x8, int damping,
                                int fb_stride, uint64_t *d_sse, uint8_t *d_dir_out, int32_t *d_var_out) {
  if (!ctx || !recon || !source || !recon->base || !source->base || !d_strengths || n_strengths <= 0 || n_strengths > 64 || !d_skip8x8 ||
      !d_sse || recon_frame < 0 || recon_frame >= recon->n_frames || source_frame < 0 || source_frame >= source->n_frames ||
      recon->width != source->width || recon->height != source->height || recon->bit_depth != source->bit_depth || (recon->width & 7) ||
      (recon->height & 7) || damping < 3 || damping > 6 || fb_stride < (recon->width + 63) / 64) {
    set_error("aomhip_cdef_search_sse_luma: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  const size_t esz = recon->bit_depth == 8 ? 1 : 2;
  const char *s = static_cast<const char *>(recon->base) +
                  ((size_t)recon_frame * recon->frame_stride + (size_t)recon->border * recon->stride + recon->border) * esz;
  const char *o = static_cast<const char *>(source->base) +
                  ((size_t)source_frame * source->frame_stride + (size_t)source->border * source->stride + source->border) * esz;
  const dim3 grid((recon->width + 63) / 64, (recon->height + 63) / 64);
  const CdefSearchArgs sa{ o, source->stride, n_strengths, d_sse };
  if (esz == 1)
    hipLaunchKernelGGL((cdef_luma_kernel<uint8_t, true>), grid, dim3(256), 0, ctx->stream, reinterpret_cast<const uint8_t *>(s),
                       static_cast<uint8_t *>(nullptr), recon->stride, recon->width, recon->height, d_strengths, d_strengths, fb_stride,
                       d_skip8x8, damping, 0, d_dir_out, d_var_out, sa);
  else
    hipLaunchKernelGGL((cdef_luma_kernel<uint16_t, true>), grid, dim3(256), 0, ctx->stream, reinterpret_cast<const uint16_t *>(s),
                       static_cast<uint16_t *>(nullptr), recon->stride, recon->width, recon->height, d_strengths, d_strengths, fb_stride,
                       d_skip8x8, damping, recon->bit_depth - 8, d_dir_out, d_var_out, sa);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}

int aomhip_cdef_chroma_plane(aomhip_ctx *ctx, const aomhip_planes *src, int src_frame, const aomhip_planes *dst,
                             int dst_frame, int xdec, int ydec, const uint8_t *d_luma_dir, const uint8_t *d_fb_uv_pri,
                             const uint8_t *d_fb_uv_sec, int fb_stride, const uint8_t *d_skip8x8, int damping) {
  if (!ctx || !src || !dst || !src->base || !dst->base || !d_luma_dir || !d_fb_uv_pri || !d_fb_uv_sec || !d_skip8x8 ||
      src_frame < 0 || src_frame >= src->n_frames || dst_frame < 0 || dst_frame >= dst->n_frames ||
      src->width != dst->width || src->height != dst->height || src->stride != dst->stride ||
      src->bit_depth != dst->bit_depth || xdec < 0 || xdec > 1 || ydec < 0 || ydec > 1 ||
      (src->width % (8 >> xdec)) || (src->height % (8 >> ydec)) || damping < 3 || damping > 6 ||
      fb_stride < (src->width + (64 >> xdec) - 1) / (64 >> xdec) || (src->base == dst->base && src_frame == dst_frame)) {
    set_error("aomhip_cdef_chroma_plane: invalid argument (chroma size must be whole blocks, out of place)");
    return AOMHIP_ERR_INVALID;
  }
  const size_t esz = src->bit_depth == 8 ? 1 : 2;
  const char *s = static_cast<const char *>(src->base) +
                  ((size_t)src_frame * src->frame_stride + (size_t)src->border * src->stride + src->border) * esz;
  char *d = static_cast<char *>(dst->base) +
            ((size_t)dst_frame * dst->frame_stride + (size_t)dst->border * dst->stride + dst->border) * esz;
  const dim3 grid((src->width + (64 >> xdec) - 1) / (64 >> xdec), (src->height + (64 >> ydec) - 1) / (64 >> ydec));
  if (esz == 1)
    launch_chroma<uint8_t, false>(ctx->stream, grid, xdec, ydec, s, d, src->stride, src->width, src->height, d_luma_dir,
                           d_fb_uv_pri, d_fb_uv_sec, fb_stride, d_skip8x8, damping, 0);
  else
    launch_chroma<uint16_t, false>(ctx->stream, grid, xdec, ydec, s, d, src->stride, src->width, src->height, d_luma_dir,
                            d_fb_uv_pri, d_fb_uv_sec, fb_stride, d_skip8x8, damping, src->bit_depth - 8);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}

int aomhip_cdef_search_sse_chroma(aomhip_ctx *ctx, const aomhip_planes *recon, int recon_frame, const aomhip_planes *source,
                                  int source_frame, int xdec, int ydec, const uint8_t *d_luma_dir, const uint8_t *d_strengths,
                                  int n_strengths, const uint8_t *d_skip8x8, int damping, int fb_stride, uint64_t *d_sse) {
  if (!ctx || !recon || !source || !recon->base || !source->base || !d_luma_dir || !d_strengths || n_strengths <= 0 || n_strengths > 64 ||
      !d_skip8x8 || !d_sse || recon_frame < 0 || recon_frame >= recon->n_frames || source_frame < 0 || source_frame >= source->n_frames ||
      recon->width != source->width || recon->height != source->height || recon->bit_depth != source->bit_depth || xdec < 0 || xdec > 1 ||
      ydec < 0 || ydec > 1 || (recon->width % (8 >> xdec)) || (recon->height % (8 >> ydec)) || damping < 3 || damping > 6 ||
      fb_stride < (recon->width + (64 >> xdec) - 1) / (64 >> xdec)) {
    set_error("aomhip_cdef_search_sse_chroma: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  const size_t esz = recon->bit_depth == 8 ? 1 : 2;
  const char *s = static_cast<const char *>(recon->base) +
                  ((size_t)recon_frame * recon->frame_stride + (size_t)recon->border * recon->stride + recon->border) * esz;
  const char *o = static_cast<const char *>(source->base) +
                  ((size_t)source_frame * source->frame_stride + (size_t)source->border * source->stride + source->border) * esz;
  const dim3 grid((recon->width + (64 >> xdec) - 1) / (64 >> xdec), (recon->height + (64 >> ydec) - 1) / (64 >> ydec));
  const CdefSearchArgs sa{ o, source->stride, n_strengths, d_sse };
  if (esz == 1)
    launch_chroma<uint8_t, true>(ctx->stream, grid, xdec, ydec, s, nullptr, recon->stride, recon->width, recon->height, d_luma_dir, d_strengths,
                                 d_strengths, fb_stride, d_skip8x8, damping, 0, sa);
  else
    launch_chroma<uint16_t, true>(ctx->stream, grid, xdec, ydec, s, nullptr, recon->stride, recon->width, recon->height, d_luma_dir,
                                  d_strengths, d_strengths, fb_stride, d_skip8x8, damping, recon->bit_depth - 8, sa);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}


// cdef_find_dir / cdef_find_dir_dual (av1/common/av1_rtcd_defs.pl:504-506) on host pointers.
int aomhip_cdef_find_dir(const uint16_t *img, int stride, int32_t *var, int coeff_shift) {
  *var = 0;
  aomhip_ctx *ctx = default_ctx();
  if (!ctx) return 0;
  const size_t total = 128 + 16;
  char *h = static_cast<char *>(pinned(ctx, total)), *d = static_cast<char *>(scratch(ctx, total));
  if (!h || !d) { note_failure("aomhip_cdef_find_dir scratch", AOMHIP_ERR_NOMEM); return 0; }
  for (int r = 0; r < 8; ++r) memcpy(h + r * 16, img + (ptrdiff_t)r * stride, 16);
  if (hipMemcpyAsync(d, h, 128, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { note_failure("aomhip_cdef_find_dir H2D"); return 0; }
  hipLaunchKernelGGL(cdef_find_dir_kernel, dim3(1), dim3(64), 0, ctx->stream, reinterpret_cast<const uint16_t *>(d), 1, coeff_shift,
                     reinterpret_cast<int32_t *>(d + 128));
  if (hipGetLastError() != hipSuccess || hipMemcpyAsync(h + 128, d + 128, 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess) { note_failure("aomhip_cdef_find_dir"); return 0; }
  const int32_t *o = reinterpret_cast<const int32_t *>(h + 128);
  *var = o[1];
  return o[0];
}
void aomhip_cdef_find_dir_dual(const uint16_t *img1, const uint16_t *img2, int stride, int32_t *var1, int32_t *var2, int coeff_shift,
                               int *out1, int *out2) {
  *out1 = aomhip_cdef_find_dir(img1, stride, var1, coeff_shift);
  *out2 = aomhip_cdef_find_dir(img2, stride, var2, coeff_shift);
}

// cdef_filter_{8,16}_{0..3} (av1_rtcd_defs.pl:508-519): dst8 is a uint8_t (is_16 = 0) or uint16_t (is_16 = 1) block, `in` points
// into the 16-bit CDEF_BSTRIDE (144) buffer with its 2-pixel borders; variant _0 = primary + secondary, _1 primary only,
// _2 secondary only, _3 neither (cdef_block.c:232-281).
void aomhip_cdef_filter_any(void *dst8, int dstride, const uint16_t *in, int pri_strength, int sec_strength, int dir, int pri_damping,
                            int sec_damping, int coeff_shift, int block_width, int block_height, int is_16, int variant) {
  aomhip_ctx *ctx = default_ctx();
  if (!ctx) return;
  if (block_width < 1 || block_height < 1 || block_width * block_height > 64 || variant < 0 || variant > 3 || dir < 0 || dir > 7) {
    set_error("aomhip_cdef_filter: unsupported block %dx%d / variant %d", block_width, block_height, variant);
    return note_failure("aomhip_cdef_filter", AOMHIP_ERR_INVALID);
  }
  const int enable_primary = variant == 0 || variant == 1, enable_secondary = variant == 0 || variant == 2;
  const int s = block_width + 4, rows = block_height + 4;
  const size_t in_bytes = (size_t)s * rows * 2, out_off = (in_bytes + 15) & ~(size_t)15, total = out_off + 128;
  char *h = static_cast<char *>(pinned(ctx, total)), *d = static_cast<char *>(scratch(ctx, total));
  if (!h || !d) return note_failure("aomhip_cdef_filter scratch", AOMHIP_ERR_NOMEM);
  for (int r = 0; r < rows; ++r) memcpy(h + (size_t)r * s * 2, in + (ptrdiff_t)(r - 2) * 144 - 2, (size_t)s * 2);
  if (hipMemcpyAsync(d, h, in_bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return note_failure("aomhip_cdef_filter H2D");
  hipLaunchKernelGGL(cdef_filter_block_kernel, dim3(1), dim3(64), 0, ctx->stream, reinterpret_cast<const uint16_t *>(d),
                     reinterpret_cast<uint16_t *>(d + out_off), block_width, block_height, pri_strength, sec_strength, dir, pri_damping,
                     sec_damping, coeff_shift, enable_primary, enable_secondary);
  if (hipGetLastError() != hipSuccess || hipMemcpyAsync(h + out_off, d + out_off, 128, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess)
    return note_failure("aomhip_cdef_filter");
  const uint16_t *o = reinterpret_cast<const uint16_t *>(h + out_off);
  for (int i = 0; i < block_height; ++i)
    for (int j = 0; j < block_width; ++j) {
      if (is_16) static_cast<uint16_t *>(dst8)[(ptrdiff_t)i * dstride + j] = o[i * block_width + j];
      else static_cast<uint8_t *>(dst8)[(ptrdiff_t)i * dstride + j] = (uint8_t)o[i * block_width + j];
    }
}

}  // extern "C"
