// Probe: how fast can 256 persistent workgroups pull strip-shaped windows (rows of `seg` bytes at the plane's row pitch, `sb_h`
// rows per step, one strip per workgroup, strips of one frame side by side on one XCD) out of HBM into registers, as a function of
// how many loader lanes a workgroup uses and how many batches each keeps in flight?  It is the transport of sad_strip_kernel with
// everything else removed (no LDS, no barriers, no evaluation), to tell a memory-system limit of the access pattern from a limit of
// the kernel's step structure.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/strip_read_probe.hip -o build/strip_read_probe && build/strip_read_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

typedef uint32_t V4 __attribute__((ext_vector_type(4)));

struct Args {
  const char *ref, *src;
  int64_t fstride;
  int frames, strips, sb_w, sb_h, range, pitch, height, lanes;
  int cpr, scpr;           // 16-byte chunks per ring row / source row
  unsigned magic_cpr, magic_scpr;
  unsigned *sink;
};

static unsigned magic_of(int d) { return (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); }

template <int N, int NS, int D>
__global__ __launch_bounds__(1024) void probe(Args a) {
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
    r_off[i] = row * (unsigned)a.pitch + col * 16u;
  }
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const unsigned q = (unsigned)min(tid + i * a.lanes, total_s - 1), row = __umulhi(q, a.magic_scpr), col = q - row * (unsigned)a.scpr;
    s_off[i] = row * (unsigned)a.pitch + col * 16u;
  }
  V4 acc = { 0, 0, 0, 0 };
  const int steps = a.height / a.sb_h;  // (whole steps only)
  for (int item = wg_j; item < n_items; item += wg_n) {
    const int fi = item / a.strips, cx = item - fi * a.strips, f = xcd + 8 * fi;
    // border 160: visible origin at (160, 160); the window starts `range` to the left of the cell
    const char *rb = a.ref + f * a.fstride + (int64_t)(160 + a.range) * a.pitch + 160 + cx * a.sb_w - a.range;
    const char *sb = a.src + f * a.fstride + (int64_t)160 * a.pitch + 160 + cx * a.sb_w;
    V4 v[D][N], s[D][NS];
    auto load = [&](int d, int cy) {
      const int cyc = min(cy, steps - 1);
      const char *r0 = rb + (int64_t)cyc * a.sb_h * a.pitch, *s0 = sb + (int64_t)cyc * a.sb_h * a.pitch;
#pragma unroll
      for (int i = 0; i < N; ++i) v[d][i] = *reinterpret_cast<const V4 *>(r0 + r_off[i]);
#pragma unroll
      for (int i = 0; i < NS; ++i) s[d][i] = *reinterpret_cast<const V4 *>(s0 + s_off[i]);
    };
#pragma unroll
    for (int d = 0; d < D; ++d) load(d, d);
    for (int cy = 0; cy < steps; cy += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
#pragma unroll
        for (int i = 0; i < N; ++i) acc ^= v[d][i];
#pragma unroll
        for (int i = 0; i < NS; ++i) acc ^= s[d][i];
        load(d, cy + d + D);
      }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
#pragma unroll
      for (int i = 0; i < N; ++i) acc ^= v[d][i];
#pragma unroll
      for (int i = 0; i < NS; ++i) acc ^= s[d][i];
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) a.sink[0] = 1;
}

template <int N, int NS, int D>
static float run(const Args &a, int grid) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((probe<N, NS, D>), dim3(grid), dim3(1024), 0, 0, a);
  hipEventRecord(e0, 0);
  const int reps = 5;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((probe<N, NS, D>), dim3(grid), dim3(1024), 0, 0, a);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main(int argc, char **argv) {
  const int W = 3840, H = 2160, border = 160, pitch = 4160, rows = H + 2 * border, frames = 64;
  const int64_t fstride = (int64_t)pitch * rows;
  char *ref, *src;
  unsigned *sink;
  if (hipMalloc(&ref, fstride * frames) != hipSuccess || hipMalloc(&src, fstride * frames) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&sink, 4);
  hipMemset(ref, 1, fstride * frames); hipMemset(src, 2, fstride * frames);
  hipDeviceSynchronize();
  struct Cell { int w, h; };
  const Cell cells[] = { { 240, 48 }, { 480, 32 }, { 960, 16 }, { 240, 16 }, { 128, 64 } };
  for (const Cell &c : cells) {
    Args a;
    a.ref = ref; a.src = src; a.fstride = fstride; a.frames = frames; a.strips = W / c.w; a.sb_w = c.w; a.sb_h = c.h; a.range = 64;
    a.pitch = pitch; a.height = H - 2 * 64; a.sink = sink;
    a.cpr = (c.w + 128) / 16; a.scpr = c.w / 16;
    a.magic_cpr = magic_of(a.cpr); a.magic_scpr = magic_of(a.scpr);
    const double bytes = (double)frames * a.strips * (a.height / c.h) * c.h * (double)(a.cpr + a.scpr) * 16;
    const double visible = (double)frames * 2 * W * a.height;
    const int chunks = c.h * a.cpr, schunks = c.h * a.scpr;
    printf("cell %dx%d: %d + %d chunks per step, requested %.3f GB (visible %.3f GB)\n", c.w, c.h, chunks, schunks, bytes / 1e9, visible / 1e9);
#define RUN(LANES, N, NS, D)                                                                                                 \
  if ((LANES) * (N) >= chunks && (LANES) * ((N) - 1) < chunks && (LANES) * (NS) >= schunks) {                                   \
    a.lanes = LANES;                                                                                                        \
    const float ms = run<N, NS, D>(a, 256);                                                                                 \
    printf("  lanes %4d x (%d+%d chunks) x depth %d = %3d KB in flight per CU: %.3f ms  requested %.2f TB/s  visible %.2f TB/s\n", \
           LANES, N, NS, D, (LANES) * ((N) + (NS)) * (D) / 64, ms, bytes / ms / 1e9, visible / ms / 1e9);                     \
  }
#define RUNS(LANES, N, NS) RUN(LANES, N, NS, 1) RUN(LANES, N, NS, 2) RUN(LANES, N, NS, 3) RUN(LANES, N, NS, 4)
    RUNS(256, 1, 1) RUNS(256, 2, 1) RUNS(256, 2, 2) RUNS(256, 3, 2) RUNS(256, 4, 2) RUNS(256, 4, 3) RUNS(256, 4, 4) RUNS(256, 5, 3) RUNS(256, 5, 4) RUNS(256, 6, 4)
    RUNS(512, 1, 1) RUNS(512, 2, 1) RUNS(512, 2, 2) RUNS(512, 3, 2) RUNS(512, 4, 2)
    RUNS(1024, 1, 1) RUNS(1024, 2, 1) RUNS(1024, 2, 2)
  }
  return 0;
}
