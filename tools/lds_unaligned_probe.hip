// Micro-probe (not part of the library): cost of ds_read_b128 by address pattern on gfx950.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_unaligned_probe.hip -o /tmp/lds_probe && /tmp/lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
struct __attribute__((packed, aligned(1))) U128 { uint32_t v[4]; };
template <bool ALIGNED>
__global__ void probe(int lane_stride, int misalign, int iters, uint32_t *out) {
  extern __shared__ uint32_t lds[];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i * 2654435761u;
  __syncthreads();
  uint32_t acc = 0;
  int off = (threadIdx.x & 63) * lane_stride + misalign;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      uint32_t a, b, c, d;
      if constexpr (ALIGNED) {
        const uint4 t = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(lds) + off + r * 608);
        a = t.x; b = t.y; c = t.z; d = t.w;
      } else {
        const U128 t = *reinterpret_cast<const U128 *>(reinterpret_cast<const char *>(lds) + off + r * 608);
        a = t.v[0]; b = t.v[1]; c = t.v[2]; d = t.v[3];
      }
      acc += a ^ b ^ c ^ d;
    }
    off = (off + 32) & 8191;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
  uint32_t *d;
  hipMalloc(&d, 1024 * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000, grid = 1024;
  struct { const char *name; bool aligned; int stride, mis; } cases[] = {
    { "aligned type, lane stride 16 B", true, 16, 0 },   { "packed type, lane stride 16 B, offset 0", false, 16, 0 },
    { "packed type, lane stride 16 B, offset 2", false, 16, 2 }, { "packed type, lane stride 2 B (adjacent columns), offset 0", false, 2, 0 },
    { "packed type, lane stride 2 B, offset 6", false, 2, 6 },   { "packed type, lane stride 1 B, offset 3", false, 1, 3 },
    { "aligned type, all lanes same address (broadcast)", true, 0, 0 }, { "packed type, lane stride 32 B, offset 4", false, 32, 4 },
  };
  for (auto &c : cases) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (c.aligned) hipLaunchKernelGGL(probe<true>, dim3(grid), dim3(256), 65536, 0, c.stride, c.mis, iters, d);
      else hipLaunchKernelGGL(probe<false>, dim3(grid), dim3(256), 65536, 0, c.stride, c.mis, iters, d);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("%-62s %8.3f ms  %7.1f G ds_read_b128 lanes/s\n", c.name, ms, (double)grid * 256 * iters * 16 / ms / 1e6);
    }
  }
  return 0;
}
