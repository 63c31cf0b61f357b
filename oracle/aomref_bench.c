/*
 * oracle/aomref_bench.c -- the CPU baseline bench.py times beside the GPU ("cpu_baseline", kind "port").
 * TEST INFRASTRUCTURE ONLY (see aomref.h).  Never linked into libaomhip.
 *
 * Shape of a run (the reference's speed-test convention, test/sad_test.cc:346-362: a fixed work list, many passes,
 * wall clock around the whole thing), threaded the way the reference threads an encode -- a STATIC partition of the work
 * over the threads (tile / row workers, av1/encoder/ethread.c:488-593): thread t owns a contiguous slice of the work
 * list and repeats its own slice; results go to a thread-private accumulator (folded into a checksum at the end so that
 * nothing is optimised away), so no two threads ever write the same cache line.
 *
 * Two kernel flavours, both from scratch:
 *   scalar -- the oracle functions (orc_sad, orc_highbd_sad, orc_fwd_txfm2d, orc_quantize_b), i.e. what the
 *             reference's `generic` target runs (aom_sadWxH_c, ...); gcc -O3 may auto-vectorise them;
 *   avx2   -- hand-written AVX2 intrinsics for the 16x16 SAD / SADx4d (8-bit: psadbw; 16-bit: abs-diff + pmaddwd) and
 *             aom_quantize_b (the shapes of aom_dsp/x86/sad_avx2.c, sad4d_avx2.c, quantize_avx2.c; not copies).
 *             The forward transform stays scalar (the reference has AVX2 transforms; this port does not, and says so).
 */
#include <immintrin.h>
#include <omp.h>
#include <string.h>

#include "aomref.h"

typedef struct { int16_t sx, sy, rx, ry; } orc_cand;            /* == aomhip_sad_cand */
typedef struct { int16_t sx, sy, rx[4], ry[4]; } orc_x4d_group;  /* == aomhip_sad_x4d_cand */

static double now_s(void) { return omp_get_wtime(); }

/* ---- AVX2 16x16 SAD kernels ---- */
static inline uint32_t sad16x16_u8_avx2(const uint8_t *s, int ss, const uint8_t *r, int rs) {
  __m256i acc = _mm256_setzero_si256();
  for (int y = 0; y < 16; y += 2) {
    const __m256i a = _mm256_inserti128_si256(_mm256_castsi128_si256(_mm_loadu_si128((const __m128i *)(s + y * ss))),
                                              _mm_loadu_si128((const __m128i *)(s + (y + 1) * ss)), 1);
    const __m256i b = _mm256_inserti128_si256(_mm256_castsi128_si256(_mm_loadu_si128((const __m128i *)(r + y * rs))),
                                              _mm_loadu_si128((const __m128i *)(r + (y + 1) * rs)), 1);
    acc = _mm256_add_epi64(acc, _mm256_sad_epu8(a, b));
  }
  const __m128i h = _mm_add_epi64(_mm256_castsi256_si128(acc), _mm256_extracti128_si256(acc, 1));
  return (uint32_t)(_mm_cvtsi128_si64(h) + _mm_extract_epi64(h, 1));
}
static inline uint32_t sad16x16_u16_avx2(const uint16_t *s, int ss, const uint16_t *r, int rs) {
  __m256i acc = _mm256_setzero_si256();
  const __m256i one = _mm256_set1_epi16(1);
  for (int y = 0; y < 16; ++y) {
    const __m256i a = _mm256_loadu_si256((const __m256i *)(s + y * ss));
    const __m256i b = _mm256_loadu_si256((const __m256i *)(r + y * rs));
    const __m256i d = _mm256_abs_epi16(_mm256_sub_epi16(a, b)); /* samples <= 4095: the difference fits int16 */
    acc = _mm256_add_epi32(acc, _mm256_madd_epi16(d, one));
  }
  __m128i h = _mm_add_epi32(_mm256_castsi256_si128(acc), _mm256_extracti128_si256(acc, 1));
  h = _mm_add_epi32(h, _mm_shuffle_epi32(h, 0x4e));
  h = _mm_add_epi32(h, _mm_shuffle_epi32(h, 0xb1));
  return (uint32_t)_mm_cvtsi128_si32(h);
}

/*
 * Mode A on F frame pairs: per block one single candidate + one x4d group (5 SADs).  planes: F source origins and F
 * reference origins (pixel (0,0) of bordered planes, common strides).  groups are per frame (F * n entries), the single
 * candidates shared.  Runs passes over the whole ring until `seconds` have elapsed (at least one); returns the
 * candidates evaluated, *elapsed the wall time, *checksum a value depending on every SAD.
 */
long long orc_bench_sad_mode_a(const void *const *src_origins, const void *const *ref_origins, int n_frames, int src_stride,
                               int ref_stride, int elem16, int bd, const orc_cand *c, const orc_x4d_group *g, int n, int threads,
                               int avx2, double seconds, double *elapsed, unsigned long long *checksum) {
  if (threads < 1) threads = 1;
  const int shift = bd == 10 ? 2 : bd == 12 ? 4 : 0;
  unsigned long long sum_all = 0;
  long long passes_done = 0;
  static int shared_stop;
  const double t0 = now_s();
#pragma omp parallel num_threads(threads) reduction(+ : sum_all)
  {
    const int t = omp_get_thread_num(), nt = omp_get_num_threads();
    const int lo = (int)((long long)n * t / nt), hi = (int)((long long)n * (t + 1) / nt);
    unsigned long long acc = 0;
    long long passes = 0;
    int stop = 0;
    while (!stop) {
      for (int f = 0; f < n_frames; ++f) {
        const orc_x4d_group *gf = g + (size_t)f * n;
        if (!elem16) {
          const uint8_t *so = (const uint8_t *)src_origins[f], *ro = (const uint8_t *)ref_origins[f];
          for (int i = lo; i < hi; ++i) {
            const uint8_t *s = so + (ptrdiff_t)c[i].sy * src_stride + c[i].sx;
            const uint8_t *r = ro + (ptrdiff_t)c[i].ry * ref_stride + c[i].rx;
            acc += avx2 ? sad16x16_u8_avx2(s, src_stride, r, ref_stride) : orc_sad(s, src_stride, r, ref_stride, 16, 16);
            const uint8_t *s4 = so + (ptrdiff_t)gf[i].sy * src_stride + gf[i].sx;
            for (int k = 0; k < 4; ++k) {
              const uint8_t *r4 = ro + (ptrdiff_t)gf[i].ry[k] * ref_stride + gf[i].rx[k];
              acc += avx2 ? sad16x16_u8_avx2(s4, src_stride, r4, ref_stride) : orc_sad(s4, src_stride, r4, ref_stride, 16, 16);
            }
          }
        } else {
          const uint16_t *so = (const uint16_t *)src_origins[f], *ro = (const uint16_t *)ref_origins[f];
          for (int i = lo; i < hi; ++i) {
            const uint16_t *s = so + (ptrdiff_t)c[i].sy * src_stride + c[i].sx;
            const uint16_t *r = ro + (ptrdiff_t)c[i].ry * ref_stride + c[i].rx;
            acc += avx2 ? sad16x16_u16_avx2(s, src_stride, r, ref_stride) >> shift : orc_highbd_sad(s, src_stride, r, ref_stride, 16, 16, bd);
            const uint16_t *s4 = so + (ptrdiff_t)gf[i].sy * src_stride + gf[i].sx;
            for (int k = 0; k < 4; ++k) {
              const uint16_t *r4 = ro + (ptrdiff_t)gf[i].ry[k] * ref_stride + gf[i].rx[k];
              acc += avx2 ? sad16x16_u16_avx2(s4, src_stride, r4, ref_stride) >> shift
                          : orc_highbd_sad(s4, src_stride, r4, ref_stride, 16, 16, bd);
            }
          }
        }
      }
      ++passes;
      /* every thread does the same number of passes: thread 0 decides after each one */
#pragma omp barrier
      if (t == 0) { shared_stop = now_s() - t0 >= seconds; passes_done = passes; }
#pragma omp barrier
      stop = shared_stop;
    }
    sum_all += acc;
  }
  *elapsed = now_s() - t0;
  *checksum = sum_all;
  return passes_done * (long long)n_frames * n * 5;
}

/* ---- AVX2 aom_quantize_b (log_scale 0 / 1 / 2, low bit depth: the int16 clamp), same results as orc_quantize_b ---- */
static void quantize_b_avx2(const int32_t *coeff, intptr_t n, const int16_t *zbin, const int16_t *round, const int16_t *quant,
                            const int16_t *quant_shift, int32_t *qcoeff, int32_t *dqcoeff, const int16_t *dequant, uint16_t *eob,
                            const int16_t *iscan, int log_scale) {
  const int zb0 = (zbin[0] + ((1 << log_scale) >> 1)) >> log_scale, zb1 = (zbin[1] + ((1 << log_scale) >> 1)) >> log_scale;
  const int rd0 = (round[0] + ((1 << log_scale) >> 1)) >> log_scale, rd1 = (round[1] + ((1 << log_scale) >> 1)) >> log_scale;
  __m256i vzb = _mm256_set1_epi32(zb1), vrd = _mm256_set1_epi32(rd1), vq = _mm256_set1_epi32(quant[1]);
  __m256i vqs = _mm256_set1_epi32(quant_shift[1]), vdq = _mm256_set1_epi32(dequant[1]);
  /* lane 0 of the first vector is the DC coefficient */
  __m256i vzb0 = _mm256_insert_epi32(vzb, zb0, 0), vrd0 = _mm256_insert_epi32(vrd, rd0, 0), vq0 = _mm256_insert_epi32(vq, quant[0], 0);
  __m256i vqs0 = _mm256_insert_epi32(vqs, quant_shift[0], 0), vdq0 = _mm256_insert_epi32(vdq, dequant[0], 0);
  const __m256i v32767 = _mm256_set1_epi32(32767);
  const __m128i sh_q = _mm_cvtsi32_si128(21 - log_scale), sh_dq = _mm_cvtsi32_si128(log_scale);
  __m256i veob = _mm256_setzero_si256();
  for (intptr_t i = 0; i < n; i += 8) {
    const __m256i c = _mm256_loadu_si256((const __m256i *)(coeff + i));
    const __m256i sign = _mm256_srai_epi32(c, 31);
    const __m256i a = _mm256_abs_epi32(c);
    const __m256i z = i ? vzb : vzb0, r = i ? vrd : vrd0, q = i ? vq : vq0, qs = i ? vqs : vqs0, dq = i ? vdq : vdq0;
    const __m256i keep = _mm256_or_si256(_mm256_cmpgt_epi32(a, z), _mm256_cmpeq_epi32(a, z)); /* |c| * wt >= zbin << 5 */
    const __m256i t = _mm256_min_epi32(_mm256_add_epi32(a, r), v32767);                          /* clamp to int16 (low bit depth) */
    /* tmp32 = ((((32 t) * quant) >> 16) + 32 t) * quant_shift >> (16 - log_scale + 5); (32 t * quant) >> 16 == (t * quant) >> 11 */
    const __m256i t1 = _mm256_add_epi32(_mm256_srai_epi32(_mm256_mullo_epi32(t, q), 11), _mm256_slli_epi32(t, 5));
    /* t1 < 2^22, quant_shift < 2^15 as a signed int16: the product needs up to 37 bits -> 64-bit halves */
    const __m256i lo = _mm256_srl_epi64(_mm256_mul_epi32(t1, qs), sh_q);
    const __m256i hi = _mm256_srl_epi64(_mm256_mul_epi32(_mm256_srli_epi64(t1, 32), _mm256_srli_epi64(qs, 32)), sh_q);
    __m256i qv = _mm256_blend_epi32(lo, _mm256_slli_epi64(hi, 32), 0xaa);
    qv = _mm256_and_si256(qv, keep);
    const __m256i dqv = _mm256_sra_epi32(_mm256_mullo_epi32(qv, dq), sh_dq);
    _mm256_storeu_si256((__m256i *)(qcoeff + i), _mm256_sub_epi32(_mm256_xor_si256(qv, sign), sign));
    _mm256_storeu_si256((__m256i *)(dqcoeff + i), _mm256_sub_epi32(_mm256_xor_si256(dqv, sign), sign));
    const __m256i isc = _mm256_cvtepi16_epi32(_mm_loadu_si128((const __m128i *)(iscan + i)));
    const __m256i nz = _mm256_xor_si256(_mm256_cmpeq_epi32(qv, _mm256_setzero_si256()), _mm256_set1_epi32(-1));
    veob = _mm256_max_epi32(veob, _mm256_and_si256(_mm256_add_epi32(isc, _mm256_set1_epi32(1)), nz));
  }
  __m128i e = _mm_max_epi32(_mm256_castsi256_si128(veob), _mm256_extracti128_si256(veob, 1));
  e = _mm_max_epi32(e, _mm_shuffle_epi32(e, 0x4e));
  e = _mm_max_epi32(e, _mm_shuffle_epi32(e, 0xb1));
  *eob = (uint16_t)_mm_cvtsi128_si32(e);
}

/* self-check of the AVX2 quantiser against the scalar oracle (tests/test_oracle_txfm_quant.py) */
void orc_quantize_b_avx2(const int32_t *coeff, intptr_t n, const int16_t *zbin, const int16_t *round, const int16_t *quant,
                         const int16_t *quant_shift, int32_t *qcoeff, int32_t *dqcoeff, const int16_t *dequant, uint16_t *eob,
                         const int16_t *iscan, int log_scale) {
  quantize_b_avx2(coeff, n, zbin, round, quant, quant_shift, qcoeff, dqcoeff, dequant, eob, iscan, log_scale);
}
uint32_t orc_sad16x16_avx2(const void *s, int ss, const void *r, int rs, int elem16) {
  return elem16 ? sad16x16_u16_avx2((const uint16_t *)s, ss, (const uint16_t *)r, rs)
                : sad16x16_u8_avx2((const uint8_t *)s, ss, (const uint8_t *)r, rs);
}

/*
 * av1_xform_quant over every tx block of 4 sizes (4x4 .. 32x32, DCT_DCT, grid mode) of `n_planes` residual planes: the
 * blocks of each (plane, size) pass are partitioned statically over the threads, outputs are thread-private scratch.
 * Returns blocks processed.
 */
long long orc_bench_txq_bd(const int16_t *const *planes, int n_planes, int width, int height, const int16_t q[5][2], int threads,
                           int avx2_quant, int bd, double seconds, double *elapsed, unsigned long long *checksum);
long long orc_bench_txq(const int16_t *const *planes, int n_planes, int width, int height, const int16_t q[5][2], int threads,
                        int avx2_quant, double seconds, double *elapsed, unsigned long long *checksum) {
  return orc_bench_txq_bd(planes, n_planes, width, height, q, threads, avx2_quant, 8, seconds, elapsed, checksum);
}
/* bd > 8: the high-bit-depth leg -- the forward transform with bd's stage ranges and aom_highbd_quantize_b (64-bit products; the port has
 * no AVX2 form of it, avx2_quant is ignored there). */
long long orc_bench_txq_bd(const int16_t *const *planes, int n_planes, int width, int height, const int16_t q[5][2], int threads,
                           int avx2_quant, int bd, double seconds, double *elapsed, unsigned long long *checksum) {
  if (threads < 1) threads = 1;
  static const int kTx[4] = { 0, 1, 2, 3 }; /* TX_4X4 .. TX_32X32 */
  int16_t scans[4][1024], iscans[4][1024];
  for (int s = 0; s < 4; ++s) orc_get_scan(kTx[s], 0, scans[s], iscans[s]);
  unsigned long long sum_all = 0;
  long long passes_done = 0, per_pass = 0;
  for (int s = 0; s < 4; ++s) per_pass += (long long)(width / (4 << s)) * (height / (4 << s));
  const double t0 = now_s();
  static int shared_stop;
#pragma omp parallel num_threads(threads) reduction(+ : sum_all)
  {
    const int t = omp_get_thread_num(), nt = omp_get_num_threads();
    int32_t full[32 * 32], qc[32 * 32], dq[32 * 32];
    unsigned long long acc = 0;
    long long passes = 0;
    int stop = 0;
    while (!stop) {
      for (int p = 0; p < n_planes; ++p)
        for (int s = 0; s < 4; ++s) {
          const int nside = 4 << s, cols = width / nside, nb = cols * (height / nside), nc = nside * nside;
          const int log_scale = (nc > 256) + (nc > 1024);
          const int lo = (int)((long long)nb * t / nt), hi = (int)((long long)nb * (t + 1) / nt);
          for (int i = lo; i < hi; ++i) {
            uint16_t eob;
            orc_fwd_txfm2d(planes[p] + (ptrdiff_t)(i / cols) * nside * width + (i % cols) * nside, full, width, kTx[s], 0, bd);
            if (bd > 8)
              orc_highbd_quantize_b(full, nc, q[0], q[1], q[2], q[3], qc, dq, q[4], &eob, scans[s], iscans[s], log_scale);
            else if (avx2_quant && nc >= 8)
              quantize_b_avx2(full, nc, q[0], q[1], q[2], q[3], qc, dq, q[4], &eob, iscans[s], log_scale);
            else
              orc_quantize_b(full, nc, q[0], q[1], q[2], q[3], qc, dq, q[4], &eob, scans[s], iscans[s], log_scale);
            acc += eob + (unsigned)qc[0] + (unsigned)dq[nc - 1];
          }
        }
      ++passes;
#pragma omp barrier
      if (t == 0) { shared_stop = now_s() - t0 >= seconds; passes_done = passes; }
#pragma omp barrier
      stop = shared_stop;
    }
    sum_all += acc;
  }
  *elapsed = now_s() - t0;
  *checksum = sum_all;
  return passes_done * per_pass * n_planes;
}
