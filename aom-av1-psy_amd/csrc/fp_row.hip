// The first pass's chained motion search as ONE launch per frame: a workgroup per block row walks its blocks left to right.
//
// firstpass_inter_prediction (av1/encoder/firstpass.c:690-815) under the raster loop (:1148-1193): best_ref_mv of block (r, c) is block
// (r, c - 1)'s *best_mv and kZeroMv at c == 0 (:1165, :1190) -- rows are independent, columns a chain.  aomhip_first_pass_inter_frame
// computes everything that does not depend on the chain (the three 0,0 errors, the two zero-MV legs) for the whole frame; the leg
// started at best_ref_mv used to run one block COLUMN at a time, a search launch + a decision launch per column: 2 x 240 dependent
// launches of a 4K frame with 135 wavefronts each (profiles/r03z_first_pass_4k_10bit_kernel_stats.csv: 24 ms per frame).  Here the chain
// lives in the registers of the row's wavefronts: list entry (get_fullmv_from_mv(best_ref_mv), av1_set_mv_search_range) ->
// av1_full_pixel_search (the same device function the batched kernel runs, fullpel_search.inc fps_block) -> av1_get_mvpred_sse + MV cost +
// NEW_MV_MODE_PENALTY -> the decision (:722-752, :777-794) -> next block, with no launch and no global round trip between the links, and
// only for the blocks that need it (raw_motion_error above the threshold, best_ref_mv != 0).  A link of the chain is ~45 000 clocks of pure
// latency (8 dependent search rounds with two memory round trips each, profiles/r04_first_pass.md), so the row's wavefronts SPECULATE
// along it (below): 7.6 -> 4.9 ms per 4K 10-bit frame; the golden-frame leg runs beside the kernel on a second stream (tf_search.hip): 3.7 ms.
#include <climits>

#define AOMHIP_FPS_DEVICE_ONLY
#include "fullpel_search.inc"

namespace aomhip {
namespace {

constexpr int kMaxFullPel = 1023;          // MAX_FULL_PEL_VAL (mcomp_structs.h:22)
constexpr int kMvLow = -(1 << 14), kMvUpp = 1 << 14;  // MV_LOW / MV_UPP (entropymv.h:75-76)
__device__ __forceinline__ int rawpel(int x) { return (x + 3 + (x >= 0)) >> 3; }  // GET_MV_RAWPEL (mv.h:28)

// SPEC wavefronts per row, speculating on the chain.  best_ref_mv of block c + 1 is block c's *best_mv, and over most of a frame that is
// the SAME vector block after block (camera motion, static background).  So the row's wavefronts search blocks c .. c + width - 1 side by
// side, ALL started from the last known best_ref_mv; then every wavefront reads the results in order: block c is always right; block c + 1
// is right if block c's *best_mv equals the best_ref_mv it was started from, and so on -- the first block whose result changes the vector
// ends the batch, the blocks behind it are searched again from the new vector in the next one.  Exactly the reference's values (a block's
// result is only kept when its input was the true one), at up to `width` links of the chain per search latency; `width` halves after a
// batch that was cut short and doubles after one that went through (1 .. SPEC), so incoherent content costs what one wavefront per row costs.
struct FpBlockOut { int nrow, ncol, err, mrow, mcol, gf, raw; };

template <typename T, int W, int H, int SPEC, bool NOSKIP>
__global__ __launch_bounds__(SPEC * 64) void fp_row_kernel(PlaneView<T> src, PlaneView<T> last, const aomhip_search_block *__restrict__ blocks,
                                                          const SiteTable *__restrict__ sites, SearchArgs q, FpfLegs L, FpfCost C,
                                                          const int32_t *__restrict__ intra, int rows, int cols, int thr, int skip_zeromv, FpfOut out, CellMap cm) {
  // A batch's blocks are horizontal neighbours searched from ONE vector: a cell, if ever there was one.  Their window (search_window.h) is staged
  // once per batch by the whole workgroup -- the row's workgroup has a CU to itself, so its reach is whatever the launch's LDS allows -- and the
  // steps of a link that stay inside it read LDS: a plane step is 2.2 x a window step on this kernel's body (profiles/r06_fps_nstep.md).
  extern __shared__ uint32_t cell_lds[];
  __shared__ SiteTable sS;
  __shared__ int next_mv[SPEC][2];
  {
    const uint32_t *g = reinterpret_cast<const uint32_t *>(sites);
    uint32_t *d = reinterpret_cast<uint32_t *>(&sS);
    for (int i = threadIdx.x; i < (int)(sizeof(SiteTable) / 4); i += SPEC * 64) d[i] = g[i];
  }
  __syncthreads();
  const int r = blockIdx.x, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (r >= rows) return;
  __builtin_amdgcn_s_setprio(3);   // a chain of latencies: its instructions go first when a throughput kernel shares the SIMD (the golden leg)
  // the list entry of first_pass_motion_search (:261-299): ref_mv = best_ref_mv, start = get_fullmv_from_mv(ref_mv), limits =
  // av1_set_mv_search_range(&x->mv_limits, &ref_mv) (mcomp.c:196-215) on the block's raw limits
  auto list_entry = [&](const aomhip_search_block &b, int brow, int bcol) -> BlockScalars {
    BlockScalars bs = BlockScalars::of(b);
    bs.ref_row = brow; bs.ref_col = bcol;
    bs.start_row = rawpel(brow); bs.start_col = rawpel(bcol);
    int col_min = rawpel(bcol) - kMaxFullPel + ((bcol & 7) ? 1 : 0), row_min = rawpel(brow) - kMaxFullPel + ((brow & 7) ? 1 : 0);
    int col_max = rawpel(bcol) + kMaxFullPel, row_max = rawpel(brow) + kMaxFullPel;
    const int lo = rawpel(kMvLow) + 1, hi = rawpel(kMvUpp) - 1;
    col_min = max(col_min, lo); row_min = max(row_min, lo);
    col_max = min(col_max, hi); row_max = min(row_max, hi);
    bs.col_min = max(bs.col_min, col_min); bs.col_max = min(bs.col_max, col_max);
    bs.row_min = max(bs.row_min, row_min); bs.row_max = min(bs.row_max, row_max);
    return bs;
  };
  // does block c run the chained search from (brow, bcol), and where does it start (pixel position of the clamped start MV's block)?
  auto search_start = [&](int c, int brow, int bcol, int *px, int *py) -> bool {
    const size_t i = (size_t)r * cols + c;
    if ((brow | bcol) == 0 || (int)L.raw[i] <= thr) return false;
    const aomhip_search_block b = blocks[i];
    const BlockScalars bs = list_entry(b, brow, bcol);
    *px = __builtin_amdgcn_readfirstlane((int)b.bx) + min(max(bs.start_col, bs.col_min), bs.col_max);
    *py = __builtin_amdgcn_readfirstlane((int)b.by) + min(max(bs.start_row, bs.row_min), bs.row_max);
    return true;
  };

  // one block of firstpass_inter_prediction with best_ref_mv = (brow, bcol) (1/8 pel); cw: the batch's window (may be empty)
  auto one_block = [&](int c, int brow, int bcol, const CellWin &cw) -> FpBlockOut {
    const size_t i = (size_t)r * cols + c;
    const aomhip_search_block b = blocks[i];
    const int bx = __builtin_amdgcn_readfirstlane((int)b.bx), by = __builtin_amdgcn_readfirstlane((int)b.by);
    const bool moved = (brow | bcol) != 0;
    const int raw = (int)L.raw[i];
    int e1 = INT_MAX, m1r = 0, m1c = 0;
    if (raw > thr) {
      if (moved) {
        const BlockScalars bs = list_entry(b, brow, bcol);
        const T *sp = src.origin + (int64_t)by * src.stride + bx;
        const T *rbase = last.origin + (int64_t)by * last.stride + bx;
        FpsResult fr;
        fps_block<T, W, H, true, true, NOSKIP>(sp, src.stride, rbase, last.stride, bx, by, bs, sS, q, cw, cell_lds, lane, &fr);
        m1r = fr.br; m1c = fr.bc;
        if (fr.var != INT_MAX) {
          // av1_get_mvpred_sse (mcomp.c:3661-3677): the sse of the mse function at the full-pel MV + mv_err_cost, + NEW_MV_MODE_PENALTY (:292-296)
          const T *rp = rbase + (int64_t)m1r * last.stride + m1c;
          unsigned long long sse = 0;
          for (int t = lane; t < W * H; t += 64) {
            const int y = t / W, x = t - y * W;
            const int d = (int)sp[(int64_t)y * src.stride + x] - (int)rp[(int64_t)y * last.stride + x];
            sse += (unsigned)__mul24(d, d);
          }
#pragma unroll
          for (int m = 1; m < 64; m <<= 1) sse += __shfl_xor(sse, m, 64);
          const uint32_t qq = q.bit_depth == 10 ? (uint32_t)((sse + 8) >> 4) : q.bit_depth == 12 ? (uint32_t)((sse + 128) >> 8) : (uint32_t)sse;
          const int mrow = m1r * 8, mcol = m1c * 8;
          int cost;
          if (C.type == kCostEntropy) {
            const int dr = mrow - brow, dc = mcol - bcol;
            const int64_t bits = (int64_t)C.mvjcost[(dc != 0) | ((dr != 0) << 1)] + C.mvcost0[dr] + C.mvcost1[dc];
            cost = (int)((bits * C.error_per_bit + (1 << 13)) >> 14);
          } else {
            const CostCtx cc{ C.type, brow, bcol };
            cost = cc.var_cost(mrow, mcol);
          }
          e1 = (int32_t)(qq + (uint32_t)cost + 32u);
        }
      } else {
        e1 = L.zerr[i]; m1r = L.zmv[2 * i]; m1c = L.zmv[2 * i + 1];
      }
    }
    // the decision (:722-752, :777-794), every lane the same values
    FpBlockOut o;
    int err = (int)L.err0[i], mrow = 0, mcol = 0, gf;
    gf = err;
    if (raw > thr) {
      if (e1 < err) { err = e1; mrow = m1r; mcol = m1c; }
      if (!skip_zeromv && moved) {
        const int e0 = L.zerr[i];
        if (e0 < err) { err = e0; mrow = L.zmv[2 * i]; mcol = L.zmv[2 * i + 1]; }
      }
      gf = err;
      if (L.gerr) { gf = (int)L.gf0[i]; if (L.gerr[i] < gf) gf = L.gerr[i]; }
    }
    int nrow = 0, ncol = 0;
    if (err <= intra[i]) { nrow = mrow * 8; ncol = mcol * 8; }
    o.nrow = __builtin_amdgcn_readfirstlane(nrow); o.ncol = __builtin_amdgcn_readfirstlane(ncol);
    o.err = err; o.mrow = mrow; o.mcol = mcol; o.gf = gf; o.raw = raw;
    return o;
  };
  auto store = [&](int c, const FpBlockOut &o) {
    const size_t i = (size_t)r * cols + c;
    out.best_mv[2 * i] = (int16_t)o.nrow; out.best_mv[2 * i + 1] = (int16_t)o.ncol;
    if (out.full_mv) { out.full_mv[2 * i] = (int16_t)o.mrow; out.full_mv[2 * i + 1] = (int16_t)o.mcol; }
    out.motion_error[i] = o.err;
    if (out.gf_motion_error) out.gf_motion_error[i] = o.gf;
    if (out.raw_motion_error) out.raw_motion_error[i] = o.raw;
  };

  int brow = 0, bcol = 0;   // MV best_ref_mv = kZeroMv at the start of every row (:1165), in 1/8 pel
  if constexpr (SPEC == 1) {
    for (int c = 0; c < cols; ++c) {
      int px = 0, py = 0;
      const bool need = search_start(c, brow, bcol, &px, &py);
      CellWin cw{ 0, 0, 0, 0, 0 };
      if (cm.win_r >= 0) cw = stage_cell_window<T, W, H, SPEC>(cm, last.origin, last.stride, need, px, py, 0, cell_lds);
      const FpBlockOut o = one_block(c, brow, bcol, cw);
      brow = o.nrow; bcol = o.ncol;
      if (lane == 0) store(c, o);
    }
  } else {
    int width = SPEC;
    for (int c = 0; c < cols;) {
      const int lim = min(width, cols - c);
      const bool mine = wave < lim;
      int px = 0, py = 0;
      const bool need = mine && search_start(c + wave, brow, bcol, &px, &py);
      CellWin cw{ 0, 0, 0, 0, 0 };
      if (cm.win_r >= 0) cw = stage_cell_window<T, W, H, SPEC>(cm, last.origin, last.stride, need, px, py, wave, cell_lds);   // (workgroup barriers inside)
      FpBlockOut o{};
      if (mine) {
        o = one_block(c + wave, brow, bcol, cw);
        if (lane == 0) { next_mv[wave][0] = o.nrow; next_mv[wave][1] = o.ncol; }
      }
      __syncthreads();
      int valid = 0, nr = brow, nc = bcol;
      bool same = true;
      while (valid < lim && same) {   // block c + valid was searched from the true best_ref_mv
        nr = __builtin_amdgcn_readfirstlane(next_mv[valid][0]);
        nc = __builtin_amdgcn_readfirstlane(next_mv[valid][1]);
        same = nr == brow && nc == bcol;
        ++valid;
      }
      if (mine && wave < valid && lane == 0) store(c + wave, o);
      width = valid == lim ? min(2 * width, SPEC) : max(width >> 1, 1);
      brow = nr; bcol = nc;
      c += valid;
      __syncthreads();   // (next_mv is rewritten by the next batch)
    }
  }
}

}  // namespace

bool fp_rows_supported(int bw, int bh) { return (bw == 16 && bh == 16) || (bw == 8 && bh == 8); }
// the row kernel carries the lean search body (fullpel_search.inc): the diamond / n-step family -- what the first pass uses (NSTEP on the
// first-pass site table); any other method takes the column-by-column form with the general kernel
bool fp_rows_supported(int bw, int bh, int method) { return fp_rows_supported(bw, bh) && (method < kHex || method == kNstepFpf); }

int launch_fp_rows(aomhip_ctx *ctx, const aomhip_planes *src1, const aomhip_planes *last1, int bw, int bh, const aomhip_search_params *p,
                   const int32_t *d_mvjcost, const int32_t *d_mvcost_row, const int32_t *d_mvcost_col, const aomhip_search_block *d_blocks,
                   const FpfLegs &L, const int32_t *d_intra, int rows, int cols, int thr, int skip_zeromv, const FpfOut &out) {
  if (!fp_rows_supported(bw, bh, p->search_method)) return AOMHIP_ERR_INVALID;
  const SiteTable *d_sites = fps_device_sites(ctx->device, p->search_method);
  if (!d_sites) {
    set_error("first pass: could not place the site table on device %d", ctx->device);
    return AOMHIP_ERR_HIP;
  }
  const SearchArgs q = fps_search_args(p, d_mvjcost, d_mvcost_row, d_mvcost_col, src1->bit_depth, false);
  const FpfCost C{ p->mv_cost_type, p->error_per_bit, d_mvjcost, d_mvcost_row, d_mvcost_col };
  const int spec = [] { const char *e = getenv("AOMHIP_FP_ROW_WAVES"); const int v = e ? atoi(e) : 16; return v == 1 || v == 4 || v == 8 ? v : 16; }();   // (A/B, tests)
  // the batch's window: reach AOMHIP_FP_ROW_R (default 64; < 0: none) around the common start, in whatever LDS one workgroup per CU may take.
  // 4K 10-bit, ms per frame (same box): 8 wavefronts per row without a window 3.10, with reach 32 / 64 / 96: 2.96 / 2.92 / 2.97; 16 wavefronts: 2.97
  // without, 2.93 / 2.87 / 2.88 / 2.93 with reach 56 / 64 / 72 / 80
  const int win_r = [] { const char *e = getenv("AOMHIP_FP_ROW_R"); return e ? atoi(e) : 64; }();   // (read per launch: A/B, tests)
  const int es = src1->bit_depth == 8 ? 1 : 2;
  CellMap cm{ rows, win_r, 0, -last1->border, -last1->border, last1->stride - last1->border, last1->height + last1->border };
  int lds = 0;
  if (win_r >= 0) {
    int64_t pitch = ((int64_t)spec * bw + 2 * win_r) * es + 16 + 16;   // (+ 16: the left edge is rounded down to a 16-byte boundary)
    pitch += ((7 - (pitch >> 2)) & 31) << 2;
    lds = (int)std::min<int64_t>(pitch * (bh + 2 * win_r), 120 * 1024);
    cm.lds_bytes = lds;
  }
  // (NOSKIP: the full-SAD form -- fps_block without the row-skipping SAD and its re-check, 75 instead of 112 VGPRs in the batched kernel)
#define XP(T, W, H, P)                                                                                                                   \
  {                                                                                                                                     \
    auto k = q.skip_sad ? fp_row_kernel<T, W, H, P, false> : fp_row_kernel<T, W, H, P, true>;                                           \
    if (lds > 48 * 1024) AOMHIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
    hipLaunchKernelGGL(k, dim3(rows), dim3(P * 64), (size_t)lds, ctx->stream, view_of<T>(*src1), view_of<T>(*last1),                     \
                       d_blocks, d_sites, q, L, C, d_intra, rows, cols, thr, skip_zeromv, out, cm);                                     \
  }
#define X(T, W, H)                                                                                                                      \
  {                                                                                                                                     \
    if (spec == 1) XP(T, W, H, 1) else if (spec == 4) XP(T, W, H, 4) else if (spec == 8) XP(T, W, H, 8) else XP(T, W, H, 16)            \
  }
  if (src1->bit_depth == 8) {
    if (bw == 16) X(uint8_t, 16, 16) else X(uint8_t, 8, 8)
  } else {
    if (bw == 16) X(uint16_t, 16, 16) else X(uint16_t, 8, 8)
  }
#undef XP
#undef X
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}

}  // namespace aomhip
