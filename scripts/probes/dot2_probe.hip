// does v_dot2c_f32_bf16 compute a.x*b.x + a.y*b.y + c on gfx950, and does the halving butterfly of decode_fused.hip land row r on lane r?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__global__ void k(const uint32_t* a, const uint32_t* b, float* o, float* o2) {
  const int lane = threadIdx.x;
  o[lane] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a[lane]), __builtin_bit_cast(bf16x2_t, b[lane]), 1.0f, false);
  float acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = (float)(i * 100 + lane);      // row i, lane partial
  float v8[8], v4[4], v2[2];
  { const bool hi = (lane & 32) != 0; for (int i = 0; i < 8; ++i) { const float keep = hi ? acc[8 + i] : acc[i], send = hi ? acc[i] : acc[8 + i]; v8[i] = keep + __shfl_xor(send, 32, 64); } }
  { const bool hi = (lane & 16) != 0; for (int i = 0; i < 4; ++i) { const float keep = hi ? v8[4 + i] : v8[i], send = hi ? v8[i] : v8[4 + i]; v4[i] = keep + __shfl_xor(send, 16, 64); } }
  { const bool hi = (lane & 8) != 0; for (int i = 0; i < 2; ++i) { const float keep = hi ? v4[2 + i] : v4[i], send = hi ? v4[i] : v4[2 + i]; v2[i] = keep + __shfl_xor(send, 8, 64); } }
  float v1; { const bool hi = (lane & 4) != 0; const float keep = hi ? v2[1] : v2[0], send = hi ? v2[0] : v2[1]; v1 = keep + __shfl_xor(send, 4, 64); }
  v1 += __shfl_xor(v1, 2, 64); v1 += __shfl_xor(v1, 1, 64);
  o2[lane] = v1;
}
static uint32_t bf(float f) { uint32_t u; memcpy(&u, &f, 4); return u >> 16; }
int main() {
  uint32_t ha[64], hb[64]; float ho[64], ho2[64];
  for (int i = 0; i < 64; ++i) { ha[i] = bf(1.0f + i) | (bf(2.0f) << 16); hb[i] = bf(3.0f) | (bf(0.5f * i) << 16); }
  uint32_t *a, *b; float *o, *o2;
  hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&o, 256); hipMalloc(&o2, 256);
  hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
  k<<<1, 64>>>(a, b, o, o2);
  hipMemcpy(ho, o, 256, hipMemcpyDeviceToHost); hipMemcpy(ho2, o2, 256, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; ++i) { float want = (1.0f + i) * 3.0f + 2.0f * 0.5f * i + 1.0f; if (ho[i] != want) { if (bad < 4) printf("dot2 lane %d got %f want %f\n", i, ho[i], want); ++bad; } }
  printf("dot2 mismatches: %d\n", bad);
  bad = 0;
  for (int l = 0; l < 64; ++l) { int row = ((l >> 5) & 1) * 8 + ((l >> 4) & 1) * 4 + ((l >> 3) & 1) * 2 + ((l >> 2) & 1); float want = 64.0f * row * 100 + 2016.0f; if (ho2[l] != want) { if (bad < 4) printf("butterfly lane %d row %d got %f want %f\n", l, row, ho2[l], want); ++bad; } }
  printf("butterfly mismatches: %d\n", bad);
  return 0;
}
