// Micro-probes for the round-3 SAD evaluation redesign (not part of the library):
//   (1) VALU issue rate of the byte-SAD family on gfx950 (v_sad_u8, v_alignbyte, v_qsad_pk_u16_u8, v_mqsad_pk_u16_u8, v_mqsad_u32_u8,
//       v_sad_u16, v_perm_b32) against v_add_u32;
//   (2) LDS read cost and RESULT of ds_read_b32 / ds_read2_b32 / ds_read_b64 / ds_read_b96 / ds_read_b128 by address alignment, with the
//       strip kernel's lane pattern (8 lanes = 8 rows of one block at a 464-byte pitch, blocks at random columns).
//   hipcc --offload-arch=gfx950 -O3 tools/lds_ubench.hip -o build/r03_ubench && build/r03_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// ---- (1) VALU
enum { OP_ADD, OP_SAD8, OP_ALIGN, OP_QSAD, OP_MQSAD_PK, OP_MQSAD_U32, OP_SAD16, OP_PERM, OP_ALIGN_SAD, OP_N };
static const char *op_name[OP_N] = { "v_add_u32", "v_sad_u8", "v_alignbyte_b32", "v_qsad_pk_u16_u8", "v_mqsad_pk_u16_u8", "v_mqsad_u32_u8",
                                     "v_sad_u16", "v_perm_b32", "alignbyte+sad_u8 pair" };
template <int OP> __global__ __launch_bounds__(256) void valu_probe(int iters, uint32_t *out) {
  uint32_t a0 = threadIdx.x * 2654435761u, a1 = a0 ^ 0x12345678u, s = (threadIdx.x * 7u) | 1u;
  uint32_t acc[8] = { 1, 2, 3, 4, 5, 6, 7, 8 };
  unsigned long long q[4] = { 1, 2, 3, 4 };
  typedef uint32_t V4 __attribute__((ext_vector_type(4)));
  V4 m[2] = { { 1, 2, 3, 4 }, { 5, 6, 7, 8 } };
  const unsigned long long pair = ((unsigned long long)a1 << 32) | a0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if constexpr (OP == OP_ADD) { asm volatile("v_add_u32 %0, %1, %0" : "+v"(acc[r]) : "v"(a0)); }
      if constexpr (OP == OP_SAD8) { asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(acc[r]) : "v"(a0), "v"(a1)); }
      if constexpr (OP == OP_ALIGN) { asm volatile("v_alignbyte_b32 %0, %1, %0, %2" : "+v"(acc[r]) : "v"(a0), "v"(s)); }
      if constexpr (OP == OP_QSAD) { asm volatile("v_qsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(q[r & 3]) : "v"(pair), "v"(a1)); }
      if constexpr (OP == OP_MQSAD_PK) { asm volatile("v_mqsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(q[r & 3]) : "v"(pair), "v"(a1)); }
      if constexpr (OP == OP_MQSAD_U32) { asm volatile("v_mqsad_u32_u8 %0, %1, %2, %0" : "+v"(m[r & 1]) : "v"(pair), "v"(a1)); }
      if constexpr (OP == OP_SAD16) { asm volatile("v_sad_u16 %0, %1, %2, %0" : "+v"(acc[r]) : "v"(a0), "v"(a1)); }
      if constexpr (OP == OP_PERM) { asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(acc[r]) : "v"(a0), "v"(s)); }
      if constexpr (OP == OP_ALIGN_SAD) {
        uint32_t t;
        asm volatile("v_alignbyte_b32 %0, %1, %2, %3" : "=v"(t) : "v"(a0), "v"(a1), "v"(s));
        asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(acc[r]) : "v"(t), "v"(a1));
      }
    }
  }
  uint32_t r = 0;
  for (int i = 0; i < 8; ++i) r ^= acc[i];
  for (int i = 0; i < 4; ++i) r ^= (uint32_t)q[i] ^ (uint32_t)(q[i] >> 32);
  r ^= m[0].x ^ m[0].y ^ m[0].z ^ m[0].w ^ m[1].x ^ m[1].y ^ m[1].z ^ m[1].w;
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// semantics check of the quad-SAD family on one lane pattern
__global__ void qsad_semantics(const uint32_t *in, uint32_t *out) {
  const uint32_t lo = in[0], hi = in[1], ref = in[2];
  const unsigned long long pair = ((unsigned long long)hi << 32) | lo;
  unsigned long long q = 0, mq = 0;
  typedef uint32_t V4 __attribute__((ext_vector_type(4)));
  V4 m = { 0, 0, 0, 0 };
  asm volatile("v_qsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(q) : "v"(pair), "v"(ref));
  asm volatile("v_mqsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(mq) : "v"(pair), "v"(ref));
  asm volatile("v_mqsad_u32_u8 %0, %1, %2, %0" : "+v"(m) : "v"(pair), "v"(ref));
  out[0] = (uint32_t)q; out[1] = (uint32_t)(q >> 32); out[2] = (uint32_t)mq; out[3] = (uint32_t)(mq >> 32);
  out[4] = m.x; out[5] = m.y; out[6] = m.z; out[7] = m.w;
}

// ---- (2) LDS
enum { L_B32, L_2B32, L_B64, L_B96, L_B128, L_3B64, L_N };
static const char *l_name[L_N] = { "5 x ds_read_b32", "2 x ds_read2_b32 + ds_read_b32", "ds_read_b64 x2 (+b32)", "ds_read_b96 + b64", "ds_read_b128 + b32",
                                   "3 x ds_read_b64 (8-aligned cover)" };
constexpr int kPitch = 464, kLdsBytes = 128 * 1024;
// lane l of a wavefront: block l / 8 of the wavefront, row l % 8; the block's column comes from a per-(wave, iteration) table
template <int MODE> __global__ __launch_bounds__(256) void lds_probe(int iters, int mis_mask, int mis_add, const uint32_t *cols, uint32_t *out, int check) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  for (int i = threadIdx.x; i < kLdsBytes / 4; i += blockDim.x) reinterpret_cast<uint32_t *>(lds)[i] = (uint32_t)i * 2654435761u + 12345u;
  __syncthreads();
  const int lane = threadIdx.x & 63, blk = (threadIdx.x >> 3);
  uint32_t acc = 0, chk = 0;
  for (int it = 0; it < iters; ++it) {
    const uint32_t c = cols[(blk * 61 + it * 17) & 1023];  // column of this block in [0, 384)
    const uint32_t col = (c & ~(uint32_t)mis_mask) + mis_add;
#pragma unroll
    for (int r = 0; r < 8; ++r) {  // 8 row groups below each other: rows (lane & 7) + 8 r
      const uint32_t addr = (uint32_t)(((lane & 7) + 8 * r) * kPitch) + col;
      uint32_t d0 = 0, d1 = 0, d2 = 0, d3 = 0, d4 = 0, d5 = 0;
      if constexpr (MODE == L_B32) {
        asm volatile("ds_read_b32 %0, %5\n ds_read_b32 %1, %5 offset:4\n ds_read_b32 %2, %5 offset:8\n ds_read_b32 %3, %5 offset:12\n ds_read_b32 %4, %5 offset:16\n s_waitcnt lgkmcnt(0)"
                     : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4) : "v"(addr));
      }
      if constexpr (MODE == L_2B32) {
        unsigned long long p0, p1;
        asm volatile("ds_read2_b32 %0, %3 offset1:1\n ds_read2_b32 %1, %3 offset0:2 offset1:3\n ds_read_b32 %2, %3 offset:16\n s_waitcnt lgkmcnt(0)"
                     : "=&v"(p0), "=&v"(p1), "=&v"(d4) : "v"(addr));
        d0 = (uint32_t)p0; d1 = (uint32_t)(p0 >> 32); d2 = (uint32_t)p1; d3 = (uint32_t)(p1 >> 32);
      }
      if constexpr (MODE == L_B64) {
        unsigned long long p0, p1;
        asm volatile("ds_read_b64 %0, %3\n ds_read_b64 %1, %3 offset:8\n ds_read_b32 %2, %3 offset:16\n s_waitcnt lgkmcnt(0)"
                     : "=&v"(p0), "=&v"(p1), "=&v"(d4) : "v"(addr));
        d0 = (uint32_t)p0; d1 = (uint32_t)(p0 >> 32); d2 = (uint32_t)p1; d3 = (uint32_t)(p1 >> 32);
      }
      if constexpr (MODE == L_B96) {
        typedef uint32_t V3 __attribute__((ext_vector_type(3)));
        V3 p0; unsigned long long p1;
        asm volatile("ds_read_b96 %0, %2\n ds_read_b64 %1, %2 offset:12\n s_waitcnt lgkmcnt(0)" : "=&v"(p0), "=&v"(p1) : "v"(addr));
        d0 = p0.x; d1 = p0.y; d2 = p0.z; d3 = (uint32_t)p1; d4 = (uint32_t)(p1 >> 32);
      }
      if constexpr (MODE == L_B128) {
        typedef uint32_t V4 __attribute__((ext_vector_type(4)));
        V4 p0;
        asm volatile("ds_read_b128 %0, %2\n ds_read_b32 %1, %2 offset:16\n s_waitcnt lgkmcnt(0)" : "=&v"(p0), "=&v"(d4) : "v"(addr));
        d0 = p0.x; d1 = p0.y; d2 = p0.z; d3 = p0.w;
      }
      if constexpr (MODE == L_3B64) {
        unsigned long long p0, p1, p2;
        const uint32_t a8 = addr & ~7u;
        asm volatile("ds_read_b64 %0, %3\n ds_read_b64 %1, %3 offset:8\n ds_read_b64 %2, %3 offset:16\n s_waitcnt lgkmcnt(0)"
                     : "=&v"(p0), "=&v"(p1), "=&v"(p2) : "v"(a8));
        d0 = (uint32_t)p0; d1 = (uint32_t)(p0 >> 32); d2 = (uint32_t)p1; d3 = (uint32_t)(p1 >> 32); d4 = (uint32_t)p2; d5 = (uint32_t)(p2 >> 32);
      }
      acc += d0 ^ d1 ^ d2 ^ d3 ^ d4 ^ d5;
      if (check && MODE != L_3B64) {  // what the bytes at addr .. addr + 19 are
        uint32_t e[5];
        for (int k = 0; k < 5; ++k) {
          uint32_t v = 0;
          for (int b = 0; b < 4; ++b) v |= (uint32_t)lds[addr + 4 * k + b] << (8 * b);
          e[k] = v;
        }
        chk |= (e[0] != d0) | (e[1] != d1) | (e[2] != d2) | (e[3] != d3) | (e[4] != d4);
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = check ? chk : acc;
}

template <int OP> static int run_valu(uint32_t *d, hipEvent_t e0, hipEvent_t e1, double *base) {
  const int iters = 4000, grid = 256 * 8;  // 8 blocks of 4 wavefronts per CU = 8 wavefronts per SIMD
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(valu_probe<OP>, dim3(grid), dim3(256), 0, 0, iters, d);
    hipEventRecord(e1);
    CK(hipEventSynchronize(e1));
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double n = (double)grid * 4 * iters * 8 * (OP == OP_ALIGN_SAD ? 2 : 1);  // wave-instructions
  const double rate = n / (ms * 1e-3) / 1024;                                    // per SIMD per second
  if (OP == OP_ADD) *base = rate;
  printf("VALU %-24s %8.3f ms  %7.3f G wave-instr/s/SIMD  (%.2f x v_add_u32 time)\n", op_name[OP], ms, rate / 1e9, *base / rate);
  return 0;
}

template <int MODE> static int run_lds(uint32_t *d, const uint32_t *cols, hipEvent_t e0, hipEvent_t e1, int mask, int add, const char *what) {
  const int iters = 500, grid = 256;  // one 256-lane workgroup per CU ... x4 below
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(lds_probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
  // correctness first
  hipLaunchKernelGGL(lds_probe<MODE>, dim3(8), dim3(256), kLdsBytes, 0, 4, mask, add, cols, d, 1);
  std::vector<uint32_t> h(8 * 256);
  CK(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
  int bad = 0;
  for (auto v : h) bad += v != 0;
  float ms = 0;
  for (int thr = 256; thr <= 1024; thr *= 2) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(lds_probe<MODE>, dim3(grid), dim3(thr), kLdsBytes, 0, iters, mask, add, cols, d, 0);
      hipEventRecord(e1);
      CK(hipEventSynchronize(e1));
      hipEventElapsedTime(&ms, e0, e1);
    }
    const double rows = (double)grid * (thr / 64) * iters * 8;  // wavefront row-reads (64 lanes x 20 bytes each)
    printf("LDS  %-34s %-28s waves/CU %2d  %8.3f ms  %7.2f ns per wavefront row-read per CU  %s\n", l_name[MODE], what, thr / 64, ms, ms * 1e6 / (rows / 256),
           MODE == L_3B64 ? "" : bad ? "RESULT WRONG" : "result ok");
  }
  return 0;
}

int main() {
  uint32_t *d, *cols;
  CK(hipMalloc(&d, 1024 * 2048 * 4));
  std::vector<uint32_t> hc(1024);
  uint32_t s = 1;
  for (auto &c : hc) { s = s * 1664525u + 1013904223u; c = (s >> 8) % 384; }
  CK(hipMalloc(&cols, 4096));
  CK(hipMemcpy(cols, hc.data(), 4096, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  double base = 1;
  run_valu<OP_ADD>(d, e0, e1, &base); run_valu<OP_SAD8>(d, e0, e1, &base); run_valu<OP_ALIGN>(d, e0, e1, &base);
  run_valu<OP_QSAD>(d, e0, e1, &base); run_valu<OP_MQSAD_PK>(d, e0, e1, &base); run_valu<OP_MQSAD_U32>(d, e0, e1, &base);
  run_valu<OP_SAD16>(d, e0, e1, &base); run_valu<OP_PERM>(d, e0, e1, &base); run_valu<OP_ALIGN_SAD>(d, e0, e1, &base);
  {
    const uint32_t in[3] = { 0x04030201u, 0x08070605u, 0x05040302u };
    uint32_t *di, *dout, o[8];
    CK(hipMalloc(&di, 12)); CK(hipMalloc(&dout, 32));
    CK(hipMemcpy(di, in, 12, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(qsad_semantics, dim3(1), dim3(1), 0, 0, di, dout);
    CK(hipMemcpy(o, dout, 32, hipMemcpyDeviceToHost));
    printf("qsad semantics: S0 = 08070605_04030201, S1 = 05040302: qsad_pk = %08x_%08x  mqsad_pk = %08x_%08x  mqsad_u32 = %u %u %u %u\n", o[1], o[0], o[3], o[2], o[4], o[5], o[6], o[7]);
  }
  struct { int mask, add; const char *what; } al[] = {
    { 15, 0, "16-byte aligned" }, { 7, 0, "8-byte aligned" }, { 3, 0, "4-byte aligned (random dword)" }, { 0, 0, "random byte" },
    { 3, 1, "dword + 1" }, { 3, 2, "dword + 2" }, { 3, 3, "dword + 3" },
  };
  for (auto &a : al) {
    run_lds<L_B32>(d, cols, e0, e1, a.mask, a.add, a.what);
    run_lds<L_2B32>(d, cols, e0, e1, a.mask, a.add, a.what);
    run_lds<L_B64>(d, cols, e0, e1, a.mask, a.add, a.what);
    run_lds<L_B96>(d, cols, e0, e1, a.mask, a.add, a.what);
    run_lds<L_B128>(d, cols, e0, e1, a.mask, a.add, a.what);
    run_lds<L_3B64>(d, cols, e0, e1, a.mask, a.add, a.what);
  }
  return 0;
}
