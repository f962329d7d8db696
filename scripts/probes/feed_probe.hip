// Stand-alone probe: how fast can ONE compute unit pull the operand tiles of a 256 x 256 x 64 bf16 GEMM step out of L2 / HBM,
// by which path?  (round 4: the ping-pong kernel, the 4-wave kernel and the library's assembly kernel all end at ~23-24 B/clk/CU
// of operand feed; is that a property of the LDS-DMA path?)
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/feed scripts/probes/feed_probe.hip && /tmp/feed
//
// The access pattern is the GEMM's: 8192^3 NT, 1024 tiles of 256 x 256, 128 K tiles of 64 (whole 128-byte lines), every
// workgroup walks its K range fetching 256 rows of A and 256 rows of B per step = 64 KiB.  No MFMA, no result: only the feed.
//   mode 0: buffer_load ... lds (LDS-DMA, 16 B per lane), 8 waves
//   mode 1: buffer_load_dwordx4 into VGPRs (discarded), 8 waves
//   mode 2: half of each (A by DMA, B into VGPRs)
//   mode 3: mode 0 plus the fragment-read traffic of the ping-pong kernel (24 ds_read_b128 per wave per K tile)
//   mode 4: mode 0 with 4 waves (16 DMA instructions per wave per K tile)
//   mode 5: mode 1 + ds_write_b128 of what was loaded (the register-staged feed)
// D = K tiles in flight (vmcnt throttle); BAR = one s_barrier per K tile.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 2; } } while (0)

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
constexpr int DIM = 8192, NKT = DIM / 64, TILES = (DIM / 256) * (DIM / 256);

__device__ __forceinline__ u32x4_t raw_rsrc(const void* p) {
  const uint64_t a = (uint64_t)(uintptr_t)p;
  u32x4_t r = {(uint32_t)a, (uint32_t)(a >> 32) & 0xffffu, 0x7fffffffu, 0x00020000u};
  for (int i = 0; i < 4; ++i) r[i] = __builtin_amdgcn_readfirstlane(r[i]);
  return r;
}
// a load the compiler does not track: the probe throttles with its own counted vmcnt and never looks at the data
__device__ __forceinline__ u32x4_t load_untracked(u32x4_t rs, uint32_t vo, int soff) {
  u32x4_t v;
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(vo), "s"(rs), "s"(soff) : "memory");
  return v;
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int MODE, int NW, int D, bool BAR>
__global__ __launch_bounds__(NW * 64) void feed_k(const char* __restrict__ A, const char* __restrict__ B, uint64_t* clk, uint32_t* sink) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // the product kernel's tile order: block ids dealt round-robin to the XCDs, every XCD walks a contiguous run, 4 row tiles per group
  int bid = blockIdx.x;
  bid = (bid & 7) * (TILES / 8) + (bid >> 3);
  const int grp = bid / (4 * 32), in = bid % (4 * 32);
  const int tm = grp * 4 + (in & 3), tn = in >> 2;
  const uint32_t ld2 = DIM * 2;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(A), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(B), 0, 0x7fffffff, 0x00020000);
  const u32x4_t qA = raw_rsrc(A), qB = raw_rsrc(B);
  constexpr int PER = 32 / NW;                     // 1 KiB instructions per operand per wave per K tile (32 per operand)
  uint32_t voA[PER], voB[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int row = (wave * PER + j) * 8 + (lane >> 3);
    voA[j] = (uint32_t)(tm * 256 + row) * ld2 + (uint32_t)((lane & 7) << 4);
    voB[j] = (uint32_t)(tn * 256 + row) * ld2 + (uint32_t)((lane & 7) << 4);
  }
  uint64_t c0 = 0, w0 = 0;
  if (blockIdx.x == 0 && tid == 0) { c0 = __builtin_readcyclecounter(); w0 = wall_clock64(); }
  u32x4_t acc = {0u, 0u, 0u, 0u};
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  for (int t = 0; t < NKT; ++t) {
    char* buf = smem + (t & 1) * 65536;
    const int soff = t * 128;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      if constexpr (MODE == 0 || MODE == 3 || MODE == 4 || MODE == 2) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_void_t*)(buf + (wave * PER + j) * 1024), 16, voA[j], soff, 0, 0);
      } else {
        u32x4_t v = load_untracked(qA, voA[j], soff);
        if constexpr (MODE == 5) {
          asm volatile("s_waitcnt vmcnt(0)\n ds_write_b128 %0, %1" ::"v"(lds0 + (uint32_t)((t & 1) * 65536 + (wave * PER + j) * 1024 + lane * 16)), "v"(v) : "memory");
        }
      }
      if constexpr (MODE == 0 || MODE == 3 || MODE == 4) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_void_t*)(buf + 32768 + (wave * PER + j) * 1024), 16, voB[j], soff, 0, 0);
      } else {
        u32x4_t v = load_untracked(qB, voB[j], soff);
        if constexpr (MODE == 5) {
          asm volatile("s_waitcnt vmcnt(0)\n ds_write_b128 %0, %1" ::"v"(lds0 + (uint32_t)((t & 1) * 65536 + 32768 + (wave * PER + j) * 1024 + lane * 16)), "v"(v) : "memory");
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (MODE != 5) wait_vm<(D - 1) * 2 * PER < 63 ? (D - 1) * 2 * PER : 63>();
    if constexpr (MODE == 3) {
#pragma unroll
      for (int r = 0; r < 24; ++r) {
        u32x4_t v;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(lds0 + (uint32_t)(((t & 1) ^ 1) * 65536 + lane * 16 + (wave & 3) * 1024)), "n"(r * 2048) : "memory");
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        acc ^= v;
      }
    }
    if constexpr (BAR) __builtin_amdgcn_s_barrier();
  }
  wait_vm<0>();
  if (acc[0] == 0x12345678u) sink[tid] = acc[1];
  if (blockIdx.x == 0 && tid == 0) { clk[0] = __builtin_readcyclecounter() - c0; clk[1] = wall_clock64() - w0; }
#endif
}

template <int MODE, int NW, int D, bool BAR>
int run(const char* name, const char* A, const char* B, uint64_t* clk, uint32_t* sink) {
  auto k = feed_k<MODE, NW, D, BAR>;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k, dim3(TILES), dim3(NW * 64), 131072, 0, A, B, clk, sink);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  uint64_t h[2]; CHECK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
  const double bytes = (double)TILES * NKT * 65536.0, ghz = h[1] ? (double)h[0] / ((double)h[1] * 10.0) : 0.0;   // wall clock: 100 MHz
  printf("%-44s %8.1f us  %6.2f TB/s  %5.1f GB/s/CU  cycle counter / wall = %.3f GHz\n", name, best * 1e3, bytes / best / 1e9,
         bytes / best / 1e6 / 256.0, ghz);
  fflush(stdout);
  return 0;
}

int main() {
  char *A, *B; uint64_t* clk; uint32_t* sink;
  const size_t n = (size_t)DIM * DIM * 2;
  CHECK(hipMalloc(&A, n)); CHECK(hipMalloc(&B, n)); CHECK(hipMalloc(&clk, 16)); CHECK(hipMalloc(&sink, 4096));
  CHECK(hipMemset(A, 0x3c, n)); CHECK(hipMemset(B, 0x3d, n));
  if (run<0, 8, 2, false>("LDS-DMA 8 waves, 2 K tiles in flight", A, B, clk, sink)) return 2;
  if (run<0, 8, 3, false>("LDS-DMA 8 waves, 3 in flight", A, B, clk, sink)) return 2;
  if (run<0, 8, 4, false>("LDS-DMA 8 waves, 4 in flight", A, B, clk, sink)) return 2;
  if (run<0, 8, 2, true>("LDS-DMA 8 waves, 2 in flight, barrier", A, B, clk, sink)) return 2;
  if (run<4, 4, 2, false>("LDS-DMA 4 waves, 2 in flight", A, B, clk, sink)) return 2;
  if (run<1, 8, 2, false>("VGPR loads 8 waves, 2 in flight", A, B, clk, sink)) return 2;
  if (run<1, 8, 4, false>("VGPR loads 8 waves, 4 in flight", A, B, clk, sink)) return 2;
  if (run<2, 8, 2, false>("A by LDS-DMA + B into VGPRs, 2 in flight", A, B, clk, sink)) return 2;
  if (run<2, 8, 4, false>("A by LDS-DMA + B into VGPRs, 4 in flight", A, B, clk, sink)) return 2;
  if (run<5, 8, 2, false>("VGPR loads + ds_write_b128 (no overlap)", A, B, clk, sink)) return 2;
  if (run<3, 8, 2, false>("LDS-DMA + 24 ds_read_b128 / wave / K tile", A, B, clk, sink)) return 2;
  if (run<3, 8, 3, false>("LDS-DMA + reads, 3 in flight", A, B, clk, sink)) return 2;
  return 0;
}
