// Probe: semantics of ds_read_b64_tr_b16 on gfx950.  LDS[i] = i (u16); lane l passes byte address 8*perm(l).
// Prints, for each lane, the 4 u16 it received.  Build: hipcc --offload-arch=gfx950 -O2 tr_read_probe.hip -o tr_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ void probe(uint16_t* out, int mode) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int l = threadIdx.x;
  uint32_t addr;
  if (mode == 0) addr = 8 * l;                                     // lane l -> elements 4l..4l+3
  else addr = ((l & 15) >> 2) * 256 + (l & 3) * 8 + (l >> 4) * 32;   // [4 rows][16 cols] block per 16 lanes, row stride 128 elems
  addr += (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)lds;
  uint2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[l * 4 + 0] = v.x & 0xffff; out[l * 4 + 1] = v.x >> 16; out[l * 4 + 2] = v.y & 0xffff; out[l * 4 + 3] = v.y >> 16;
}
int main() {
  uint16_t* d; hipMalloc(&d, 256 * 2);
  uint16_t h[256];
  for (int mode = 0; mode < 2; ++mode) {
    probe<<<1, 64>>>(d, mode);
    hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  }
  return 0;
}
