// Binder for the encoder's per-block-size kernel table (aom_variance_fn_ptr_t, aom_dsp/variance.h:84-103;
// filled by BFP / HIGHBD_BFP at av1/encoder/encoder.c:986-1226 and encoder_utils.h:130-139,572-).  Motion
// search reaches its kernels only through this table (mcomp.c:105,125-132), so overwriting its entries after
// av1_create_primary_compressor is the least invasive way to route sdf / sdsf / vf / svf / sdx4df / sdx3df /
// sdsx4df -- and the compound / masked / OBMC members sdaf, svaf, jsdaf, jsvaf, msdf, msvf, osdf, ovf, osvf -- to the
// GPU.  Each entry is a fixed-size function with the reference's exact signature that forwards to the generic
// rtcd-signature entry points of libaomhip (sad.hip, variance.hip, compound.hip).
#include "aomhip.h"

namespace {

#define AOMHIP_SIZES(X)                                                                                          \
  X(4, 4) X(4, 8) X(8, 4) X(8, 8) X(8, 16) X(16, 8) X(16, 16) X(16, 32) X(32, 16) X(32, 32) X(32, 64) X(64, 32) \
  X(64, 64) X(64, 128) X(128, 64) X(128, 128) X(4, 16) X(16, 4) X(8, 32) X(32, 8) X(16, 64) X(64, 16)

template <int W, int H> struct Fixed {
  static unsigned int sdf(const uint8_t *a, int as, const uint8_t *b, int bs) { return aomhip_sad(a, as, b, bs, W, H); }
  static unsigned int sdsf(const uint8_t *a, int as, const uint8_t *b, int bs) { return aomhip_sad_skip(a, as, b, bs, W, H); }
  static unsigned int vf(const uint8_t *a, int as, const uint8_t *b, int bs, unsigned int *sse) {
    return aomhip_variance(a, as, b, bs, W, H, sse);
  }
  static unsigned int svf(const uint8_t *a, int as, int xo, int yo, const uint8_t *b, int bs, unsigned int *sse) {
    return aomhip_sub_pixel_variance(a, as, xo, yo, b, bs, W, H, sse);
  }
  static void sdx4df(const uint8_t *a, int as, const uint8_t *const b[], int bs, unsigned int *out) {
    aomhip_sad_x4d(a, as, b, bs, out, W, H);
  }
  static void sdsx4df(const uint8_t *a, int as, const uint8_t *const b[], int bs, unsigned int *out) {
    aomhip_sad_skip_x4d(a, as, b, bs, out, W, H);
  }
  // highbd flavours: CONVERT_TO_BYTEPTR pointers in, the _bits{8,10,12} wrappers folded in via BD
  template <int BD> static unsigned int hsdf(const uint8_t *a, int as, const uint8_t *b, int bs) {
    return aomhip_highbd_sad(a, as, b, bs, W, H, BD);
  }
  template <int BD> static unsigned int hvf(const uint8_t *a, int as, const uint8_t *b, int bs, unsigned int *sse) {
    return aomhip_highbd_variance(a, as, b, bs, W, H, BD, sse);
  }
  template <int BD>
  static unsigned int hsvf(const uint8_t *a, int as, int xo, int yo, const uint8_t *b, int bs, unsigned int *sse) {
    return aomhip_highbd_sub_pixel_variance(a, as, xo, yo, b, bs, W, H, BD, sse);
  }
  template <int BD> static void hsdx4df(const uint8_t *a, int as, const uint8_t *const b[], int bs, unsigned int *out) {
    aomhip_highbd_sad_x4d(a, as, b, bs, out, W, H, BD, 0);
  }
  template <int BD> static unsigned int hsdsf(const uint8_t *a, int as, const uint8_t *b, int bs) {
    return aomhip_highbd_sad_skip(a, as, b, bs, W, H, BD);
  }
  template <int BD> static void hsdsx4df(const uint8_t *a, int as, const uint8_t *const b[], int bs, unsigned int *out) {
    aomhip_highbd_sad_x4d(a, as, b, bs, out, W, H, BD, 1);
  }
  // The compound / masked / OBMC members (BD 0 = the 8-bit table; 8 / 10 / 12 = the highbd tables).  Operand roles as
  // in aom_dsp/variance.h:29-82: for the SAD forms `a` is the source and `b` the reference, for the sub-pixel forms `a`
  // is the block that gets interpolated.
  struct Jcp { int use_dist_wtd_comp_avg, fwd_offset, bck_offset; };  // DIST_WTD_COMP_PARAMS, av1/common/blockd.h:558-562
  static aomhip_compound_params params(int kind, int subpel, const void *jcp = nullptr, int mask_stride = 0, int invert = 0) {
    const Jcp *j = static_cast<const Jcp *>(jcp);
    return aomhip_compound_params{ kind, subpel, j ? j->fwd_offset : 0, j ? j->bck_offset : 0, mask_stride, invert };
  }
  template <int BD> static unsigned int sdaf(const uint8_t *a, int as, const uint8_t *b, int bs, const uint8_t *second_pred) {
    const aomhip_compound_params p = params(AOMHIP_COMP_AVG, 0);
    return aomhip_compound(&p, b, bs, 0, 0, a, as, second_pred, nullptr, nullptr, nullptr, W, H, BD, BD != 0, 1, nullptr);
  }
  template <int BD>
  static unsigned int jsdaf(const uint8_t *a, int as, const uint8_t *b, int bs, const uint8_t *second_pred, const void *jcp) {
    const aomhip_compound_params p = params(AOMHIP_COMP_DIST_WTD, 0, jcp);
    return aomhip_compound(&p, b, bs, 0, 0, a, as, second_pred, nullptr, nullptr, nullptr, W, H, BD, BD != 0, 1, nullptr);
  }
  template <int BD>
  static unsigned int svaf(const uint8_t *a, int as, int xo, int yo, const uint8_t *b, int bs, unsigned int *sse,
                           const uint8_t *second_pred) {
    const aomhip_compound_params p = params(AOMHIP_COMP_AVG, 1);
    return aomhip_compound(&p, a, as, xo, yo, b, bs, second_pred, nullptr, nullptr, nullptr, W, H, BD, BD != 0, 0, sse);
  }
  template <int BD>
  static unsigned int jsvaf(const uint8_t *a, int as, int xo, int yo, const uint8_t *b, int bs, unsigned int *sse,
                            const uint8_t *second_pred, const void *jcp) {
    const aomhip_compound_params p = params(AOMHIP_COMP_DIST_WTD, 1, jcp);
    return aomhip_compound(&p, a, as, xo, yo, b, bs, second_pred, nullptr, nullptr, nullptr, W, H, BD, BD != 0, 0, sse);
  }
  template <int BD>
  static unsigned int msdf(const uint8_t *src, int ss, const uint8_t *ref, int rs, const uint8_t *second_pred, const uint8_t *msk,
                           int ms, int invert) {
    const aomhip_compound_params p = params(AOMHIP_COMP_MASK, 0, nullptr, ms, invert);
    return aomhip_compound(&p, ref, rs, 0, 0, src, ss, second_pred, msk, nullptr, nullptr, W, H, BD, BD != 0, 1, nullptr);
  }
  template <int BD>
  static unsigned int msvf(const uint8_t *src, int ss, int xo, int yo, const uint8_t *ref, int rs, const uint8_t *second_pred,
                           const uint8_t *msk, int ms, int invert, unsigned int *sse) {
    const aomhip_compound_params p = params(AOMHIP_COMP_MASK, 1, nullptr, ms, invert);
    return aomhip_compound(&p, src, ss, xo, yo, ref, rs, second_pred, msk, nullptr, nullptr, W, H, BD, BD != 0, 0, sse);
  }
  template <int BD> static unsigned int osdf(const uint8_t *pre, int ps, const int32_t *wsrc, const int32_t *msk) {
    const aomhip_compound_params p = params(AOMHIP_COMP_OBMC, 0);
    return aomhip_compound(&p, pre, ps, 0, 0, nullptr, 0, nullptr, nullptr, wsrc, msk, W, H, BD, BD != 0, 1, nullptr);
  }
  template <int BD> static unsigned int ovf(const uint8_t *pre, int ps, const int32_t *wsrc, const int32_t *msk, unsigned int *sse) {
    const aomhip_compound_params p = params(AOMHIP_COMP_OBMC, 0);
    return aomhip_compound(&p, pre, ps, 0, 0, nullptr, 0, nullptr, nullptr, wsrc, msk, W, H, BD, BD != 0, 0, sse);
  }
  template <int BD>
  static unsigned int osvf(const uint8_t *pre, int ps, int xo, int yo, const int32_t *wsrc, const int32_t *msk, unsigned int *sse) {
    const aomhip_compound_params p = params(AOMHIP_COMP_OBMC, 1);
    return aomhip_compound(&p, pre, ps, xo, yo, nullptr, 0, nullptr, nullptr, wsrc, msk, W, H, BD, BD != 0, 0, sse);
  }
};

template <int W, int H, int BD> void fill_compound(aomhip_variance_vtable *t) {
  using F = Fixed<W, H>;
  t->sdaf = F::template sdaf<BD>; t->jsdaf = F::template jsdaf<BD>; t->svaf = F::template svaf<BD>; t->jsvaf = F::template jsvaf<BD>;
  t->msdf = F::template msdf<BD>; t->msvf = F::template msvf<BD>;
  t->osdf = F::template osdf<BD>; t->ovf = F::template ovf<BD>; t->osvf = F::template osvf<BD>;
}

template <int W, int H> void fill(aomhip_variance_vtable *t, int bd) {
  using F = Fixed<W, H>;
  if (bd == 8) {
    t->sdf = F::sdf; t->sdsf = F::sdsf; t->vf = F::vf; t->svf = F::svf;
    t->sdx4df = F::sdx4df; t->sdx3df = F::sdx4df;  // aom_sadWxHx3d_c forwards to x4d (aom_dsp/sad.c:124-129)
    t->sdsx4df = F::sdsx4df;
    fill_compound<W, H, 0>(t);
  } else if (bd == 10) {
    t->sdf = F::template hsdf<10>; t->vf = F::template hvf<10>; t->svf = F::template hsvf<10>;
    t->sdx4df = F::template hsdx4df<10>; t->sdx3df = F::template hsdx4df<10>;
    t->sdsf = F::template hsdsf<10>; t->sdsx4df = F::template hsdsx4df<10>;
    fill_compound<W, H, 10>(t);
  } else {
    t->sdf = F::template hsdf<12>; t->vf = F::template hvf<12>; t->svf = F::template hsvf<12>;
    t->sdx4df = F::template hsdx4df<12>; t->sdx3df = F::template hsdx4df<12>;
    t->sdsf = F::template hsdsf<12>; t->sdsx4df = F::template hsdsx4df<12>;
    fill_compound<W, H, 12>(t);
  }
}

}  // namespace

extern "C" int aomhip_bind_variance_vtable(aomhip_variance_vtable *table, int bit_depth) {
  if (!table || (bit_depth != 8 && bit_depth != 10 && bit_depth != 12)) return AOMHIP_ERR_INVALID;
  // BLOCK_SIZE order of av1/common/enums.h:99-124
  int i = 0;
#define X(W, H) fill<W, H>(&table[i++], bit_depth);
  AOMHIP_SIZES(X)
#undef X
  return AOMHIP_OK;
}
