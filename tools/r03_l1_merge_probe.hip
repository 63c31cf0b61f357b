// Micro-probe (not part of the library): how does the L1 (TCP) of gfx950 charge a 64-lane global_load_dwordx4 whose lanes fall into FEWER
// cache lines?  The search kernels are bound by L1 line look-ups (profiles/r02_search_bound.md: 0.8 per CU per clock, 41 per load
// instruction with 2 lanes per 128-byte line); if k consecutive lanes inside one line cost one look-up, re-mapping lanes so that the sites
// of a diamond step that share reference rows sit next to each other cuts the look-ups without moving any data.
//   hipcc --offload-arch=gfx950 -O3 tools/r03_l1_merge_probe.hip -o build/r03_l1_merge_probe && build/r03_l1_merge_probe
// Patterns (16-byte load per lane, rows 8 KB apart, a 2-byte-aligned start like a 10-bit search):
//   pairs      lanes (2k, 2k+1) = the two halves of one 32-byte row segment; 32 different rows per instruction      (today)
//   quad_same  4 consecutive lanes = two overlapping 32-byte segments of ONE row (2 sites 4 pixels apart)          16 rows
//   six_same   6 consecutive lanes (+2 idle) = three overlapping segments of one row (sites at -r, 0, +r; r = 4)   8 rows
//   sixteen3   16 consecutive lanes = 3 + 2 + 3 sites on three rows (the whole diamond step's sites of one block row)  12 rows
//   sixteen3_r16  the same with r = 16 (segments of one row up to 64 bytes apart)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef uint32_t V4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(2))) U128 { uint32_t v[4]; };
constexpr int kPitch = 8192, kRows = 256 + 64;  // a 2.6 MB region: L2 resident, far larger than the 32 KB L1

__global__ __launch_bounds__(256) void probe(const unsigned char *base, const int *lane_row, const int *lane_off, int iters, uint32_t *out) {
  const int lane = threadIdx.x & 63;
  const int lr = lane_row[lane], lo = lane_off[lane];
  uint32_t acc = 0;
  uint32_t s = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2654435761u + 12345u;
  s = __builtin_amdgcn_readfirstlane(s);
  for (int it = 0; it < iters; ++it) {
    s = s * 1664525u + 1013904223u;
    const int by = (s >> 8) % 192, bx = ((s >> 16) % 240) * 32 + 64;   // block position: row by, byte column bx (16 pixels x 2 bytes apart)
    const unsigned char *p = base + (size_t)(by + lr) * kPitch + bx + lo;
    const U128 v = *reinterpret_cast<const U128 *>(p);
    acc += v.v[0] ^ v.v[1] ^ v.v[2] ^ v.v[3];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
  unsigned char *d;
  CK(hipMalloc(&d, (size_t)kPitch * kRows));
  CK(hipMemset(d, 1, (size_t)kPitch * kRows));
  uint32_t *out;
  CK(hipMalloc(&out, 4096 * 256 * 4));
  int *d_row, *d_off;
  CK(hipMalloc(&d_row, 256)); CK(hipMalloc(&d_off, 256));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct Pat { const char *name; int rows_per_instr; std::vector<int> row, off; };
  std::vector<Pat> pats;
  auto mk = [&](const char *name, int rows, auto f) { Pat p{ name, rows, std::vector<int>(64), std::vector<int>(64) }; for (int l = 0; l < 64; ++l) f(l, p.row[l], p.off[l]); pats.push_back(p); };
  mk("pairs (today)", 32, [](int l, int &r, int &o) { r = l / 2; o = (l & 1) * 16 + 2 * ((l / 2) % 3); });
  mk("quad_same", 16, [](int l, int &r, int &o) { r = l / 4; o = ((l >> 1) & 1) * 8 + (l & 1) * 16; });
  mk("six_same(+2 idle)", 8, [](int l, int &r, int &o) { r = l / 8; const int u = l & 7; o = u < 6 ? (u / 2) * 8 + (u & 1) * 16 : 0; });
  auto step = [](int rr) { return [rr](int l, int &r, int &o) {
    const int u = l & 15, i = l / 16;               // block row i of this instruction; unit u: sites T.A T.B T.C | M.D M.E | B.F B.G B.H, two halves each
    const int site = u / 2, half = u & 1;
    const int rowset = site < 3 ? 0 : site < 5 ? 1 : 2;
    const int dc = site < 3 ? site - 1 : site < 5 ? (site == 3 ? -1 : 1) : site - 6;
    r = 32 + i + (rowset - 1) * rr; o = dc * rr * 2 + half * 16;
  }; };
  mk("sixteen3 r=4", 12, step(4));
  mk("sixteen3 r=8", 12, step(8));
  mk("sixteen3 r=16", 12, step(16));
  mk("sixteen3 r=1", 12, step(1));
  // the same 8 sites of a step in today's mapping (8 lanes per site, 4 rows x 2 halves per instruction), for the same r
  auto today = [](int rr) { return [rr](int l, int &r, int &o) {
    const int g = l / 8, k = l & 7;
    const int dr = (g == 0 || g == 4 || g == 6) ? -1 : (g == 1 || g == 5 || g == 7) ? 1 : 0;
    const int dc = (g == 2 || g == 4 || g == 7) ? -1 : (g == 3 || g == 5 || g == 6) ? 1 : 0;
    r = 32 + k / 2 + dr * rr; o = dc * rr * 2 + (k & 1) * 16;
  }; };
  mk("today's 8 x 8 mapping r=4", 32, today(4));
  mk("today's 8 x 8 mapping r=16", 32, today(16));
  // alignment of the 16-byte load by itself: today's mapping at r = 4 with every lane's address moved by 0 / 2 / 4 / 6 / 8 / 16 bytes
  for (int sh : { 0, 2, 4, 6, 8, 16 }) {
    static char names[8][64];
    static int ni = 0;
    snprintf(names[ni], 64, "today r=4, all lanes +%d bytes", sh);
    auto f = today(4);
    mk(names[ni++], 32, [f, sh](int l, int &r, int &o) { f(l, r, o); o += sh; });
  }
  // distinct lines per instruction, dword-aligned: 32 / 16 / 8 / 4 rows
  mk("32 rows, aligned", 32, [](int l, int &r, int &o) { r = l / 2; o = (l & 1) * 16; });
  mk("16 rows, aligned", 16, [](int l, int &r, int &o) { r = l / 4; o = (l & 3) * 16; });
  mk("8 rows, aligned", 8, [](int l, int &r, int &o) { r = l / 8; o = (l & 7) * 16; });
  mk("8 rows, aligned, 2 lines each", 8, [](int l, int &r, int &o) { r = l / 8; o = (l & 7) * 32; });
  mk("4 rows x 256 B", 4, [](int l, int &r, int &o) { r = l / 16; o = (l & 15) * 16; });
  const int iters = 2000, grid = 256 * 8;
  for (auto &p : pats) {
    CK(hipMemcpy(d_row, p.row.data(), 256, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_off, p.off.data(), 256, hipMemcpyHostToDevice));
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 0, 0, d + 64 * kPitch, d_row, d_off, iters, out);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double instr_per_cu = (double)grid * 4 * iters / 256;
    printf("%-30s rows/instr %2d  %8.3f ms  %6.2f ns per load instruction per CU  (%.1f clocks at 2.1 GHz)\n", p.name, p.rows_per_instr, ms, ms * 1e6 / instr_per_cu,
           ms * 1e6 / instr_per_cu * 2.1);
  }
  return 0;
}
