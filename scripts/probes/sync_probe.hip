// Stand-alone probe: what one dependent phase of the persistent DiT sampler (csrc/dit_fused.hip) costs, piece by piece, and what
// other exchange protocols would cost on the same grid (192 workgroups x 512 threads, one per CU, spread over the 8 XCDs).
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/sync_probe scripts/probes/sync_probe.hip && /tmp/sync_probe
//
// modes (time per iteration, 2000 iterations in one launch):
//   0  store + ack: one write-through (sc1) store, s_waitcnt vmcnt(0)
//   1  dependent agent-scope load chain (sc1 load round trip)
//   2  ping-pong of a flag between two workgroups on DIFFERENT XCDs (blockIdx 0 and 1): one-way hop = half the figure
//   3  the same between two workgroups of the SAME XCD (blockIdx 0 and 8)
//   4  the shipped barrier: atomic arrival counter + epoch flag published by the last arrival
//   5  flag-array barrier: every workgroup stores its own epoch word, wave 0 of every workgroup polls all of them (no atomic)
//   6  phase emulation, shipped barrier: 2 KB of write-through stores -> vmcnt(0) -> barrier 4 -> 104 KB of sc1 loads per workgroup
//   7  phase emulation, flag-array barrier (5) instead
//   8  phase emulation, flags in the data (LL): every value travels as an 8-byte (value, epoch) pair, consumers poll the pairs
//   9  like 6 with the activation loads through the XCD's L2 (no sc1) — wrong for coherence, shows what the bypass costs
//  10  operand burst, round-3/4 pattern: every wave 24 x dwordx4 loads whose 64 lanes touch 16 rows x 64 bytes (row stride 3 KB),
//      192 KB per workgroup per iteration out of a 1 GB buffer (cold), all workgroups at once, __syncthreads between iterations
//  11  the same bytes as 1 KB-contiguous wave loads (what tile-major operands would give)
//  12  shared operand, write-once addresses: after a device-wide barrier EVERY workgroup reads the same 104 KB region, a region never
//      touched before in this launch (iteration i reads region i of the 1 GB buffer), plain cached loads, 16 rows x 64 B per wave load
//  13  the same with agent-scope (sc1) loads: what dit_fused.hip does today on re-used addresses
//  14 / 15  like 12 / 13 with 1 KB-contiguous wave loads
//  16  barrier: EIGHT arrival counters 4 KB apart (workgroup b arrives at counter b % 8 with a no-return atomic), lanes 0-7 of wave 0
//      poll one counter each — no single address takes more than 1/8 of the atomics, no second dependent round trip
//  17  barrier 16 inside the phase emulation of mode 6
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 2; } } while (0)

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
constexpr int SC1 = 16;
constexpr int ROWS = 34, KW = 768;            // the activation every workgroup reads in a phase: 34 x 768 fp32 = 104 KB
constexpr unsigned SPIN = 1u << 22;      // per-launch budget of polls per thread: a protocol error ends in seconds, not in a hang
__device__ unsigned g_timeouts;

__device__ __forceinline__ void bar_atomic(unsigned* bar, unsigned nblk, unsigned& epoch, unsigned& spins) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    epoch += 1;
    const unsigned target = epoch * nblk;
    const unsigned arrived = __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    unsigned* flag = bar + 32;
    if (arrived == target) __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch && ++spins < SPIN) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}

// every workgroup owns one word of `flags` (64-byte apart would cost 192 lines per poll: packed, 4 bytes each, 768 bytes in all)
__device__ __forceinline__ void bar_flags(unsigned* flags, unsigned nblk, unsigned& epoch, unsigned& spins) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  epoch += 1;
  if (threadIdx.x < 64) {
    if (threadIdx.x == 0) __hip_atomic_store(flags + blockIdx.x, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(flags, 0, (int)(nblk * 4), 0x00020000);
    while (true) {
      const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(threadIdx.x * 16), 0, SC1);   // out of range: zeros
      const unsigned base = threadIdx.x * 4;
      const bool ok = (base + 0 >= nblk || v[0] >= epoch) && (base + 1 >= nblk || v[1] >= epoch) && (base + 2 >= nblk || v[2] >= epoch) &&
                      (base + 3 >= nblk || v[3] >= epoch);
      if (__builtin_amdgcn_ballot_w64(!ok) == 0ull || ++spins >= SPIN) break;
    }
  }
  __syncthreads();
}

// eight counters, 4 KB apart (1024 words): counter g counts the arrivals of the workgroups b with b % 8 == g
__device__ int g_nc = 8;      // number of arrival counters of modes 16 / 17 (argv[5])
__device__ __forceinline__ void bar_split8(unsigned* ctr, unsigned nblk, unsigned& epoch, unsigned& spins) {
  const unsigned NC = (unsigned)g_nc;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  epoch += 1;
  if (threadIdx.x < 64) {
    if (threadIdx.x == 0) (void)__hip_atomic_fetch_add(ctr + (blockIdx.x % NC) * 1024, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned g = threadIdx.x % NC;
    const unsigned target = epoch * ((nblk + NC - 1u - g) / NC);
    while (true) {
      const unsigned v = threadIdx.x < NC ? __hip_atomic_load(ctr + g * 1024, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : target;
      if (__builtin_amdgcn_ballot_w64(v < target) == 0ull || ++spins >= SPIN) break;
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(512) void probe_k(unsigned* bar, unsigned* flags, float* act, u32x2_t* ll, float* sink, int mode, int iters, const float* big, size_t big_floats) {
  const unsigned nblk = gridDim.x;
  const int tid = threadIdx.x;
  unsigned epoch = 0, spins = 0;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(act, 0, ROWS * KW * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(ll, 0, ROWS * KW * 8, 0x00020000);
  float acc = 0.f;
  if (mode == 0) {
    if (tid == 0)
      for (int it = 0; it < iters; ++it) {
        __builtin_amdgcn_raw_buffer_store_b32((unsigned)it, ra, (int)(blockIdx.x * 256), 0, SC1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
  } else if (mode == 1) {
    if (tid == 0) {
      unsigned off = blockIdx.x * 256;
      for (int it = 0; it < iters; ++it) {
        const unsigned v = __builtin_amdgcn_raw_buffer_load_b32(ra, (int)off, 0, SC1);
        off = (off + (v & 1u) * 4u) % (ROWS * KW * 4);
      }
      acc = (float)off;
    }
  } else if (mode == 2 || mode == 3) {
    const unsigned partner = mode == 2 ? 1u : 8u;
    if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == partner)) {
      unsigned* mine = flags + (blockIdx.x == 0 ? 0 : 64);
      unsigned* theirs = flags + (blockIdx.x == 0 ? 64 : 0);
      for (int it = 1; it <= iters; ++it) {
        if (blockIdx.x == 0) {
          __hip_atomic_store(mine, (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          while (__hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it && ++spins < SPIN) {}
        } else {
          while (__hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it && ++spins < SPIN) {}
          __hip_atomic_store(mine, (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
  } else if (mode == 4) {
    for (int it = 0; it < iters; ++it) bar_atomic(bar, nblk, epoch, spins);
  } else if (mode == 5) {
    for (int it = 0; it < iters; ++it) bar_flags(flags, nblk, epoch, spins);
  } else if (mode == 16) {
    for (int it = 0; it < iters; ++it) bar_split8(flags, nblk, epoch, spins);
  } else if (mode == 17) {
    for (int it = 0; it < iters; ++it) {
      if (tid < ROWS) {
        const u32x4_t v = {__float_as_uint(acc + 1.f), __float_as_uint(acc + 2.f), __float_as_uint(acc + 3.f), __float_as_uint(acc)};
        __builtin_amdgcn_raw_buffer_store_b128(v, ra, (int)((tid * KW + (blockIdx.x % 192) * 4) * 4), 0, SC1);
      }
      bar_split8(flags, nblk, epoch, spins);
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 13; ++j) {
        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(ra, (j * 512 + tid) * 16, 0, SC1);
        s += __uint_as_float(v[0]) + __uint_as_float(v[3]);
      }
      acc = s * 1e-30f;
    }
  } else if (mode == 6 || mode == 7 || mode == 9) {
    // a workgroup produces 4 columns of the 34 x 768 activation (192 x 4 = 768), then reads all of it
    for (int it = 0; it < iters; ++it) {
      if (tid < ROWS) {
        const u32x4_t v = {__float_as_uint(acc + 1.f), __float_as_uint(acc + 2.f), __float_as_uint(acc + 3.f), __float_as_uint(acc)};
        __builtin_amdgcn_raw_buffer_store_b128(v, ra, (int)((tid * KW + (blockIdx.x % 192) * 4) * 4), 0, SC1);
      }
      if (mode == 7) bar_flags(flags, nblk, epoch, spins); else bar_atomic(bar, nblk, epoch, spins);
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 13; ++j) {                   // 512 threads x 13 x 16 B = 104 KB
        const int idx = (j * 512 + tid) * 16;
        const u32x4_t v = mode == 9 ? __builtin_amdgcn_raw_buffer_load_b128(ra, idx, 0, 0) : __builtin_amdgcn_raw_buffer_load_b128(ra, idx, 0, SC1);
        s += __uint_as_float(v[0]) + __uint_as_float(v[3]);
      }
      acc = s * 1e-30f;
    }
  } else if (mode == 8) {
    for (int it = 1; it <= iters; ++it) {
      if (tid < ROWS * 4) {                            // 34 rows x 4 columns, one 8-byte pair per lane
        const int row = tid >> 2, col = (blockIdx.x % 192) * 4 + (tid & 3);
        const u32x2_t v = {__float_as_uint(acc + (float)col), (unsigned)it};
        __builtin_amdgcn_raw_buffer_store_b64(v, rl, (int)((row * KW + col) * 8), 0, SC1);
      }
      float s = 0.f;
      u32x4_t v[26];                                   // 512 threads x 26 x 16 B = 208 KB of pairs, all requested at once; the whole
      while (true) {                                   // set is requested again while any pair still carries an older epoch
        bool ok = true;
#pragma unroll
        for (int j = 0; j < 26; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rl, (j * 512 + tid) * 16, 0, SC1);
#pragma unroll
        for (int j = 0; j < 26; ++j) ok = ok && v[j][1] >= (unsigned)it && v[j][3] >= (unsigned)it;
        if (ok || ++spins >= SPIN) break;
      }
#pragma unroll
      for (int j = 0; j < 26; ++j) s += __uint_as_float(v[j][0]) + __uint_as_float(v[j][2]);
      acc = s * 1e-30f;
      __syncthreads();
    }
  }
  if (mode == 10 || mode == 11) {
    const int lane = tid & 63, wave = tid >> 6;
    const size_t wg_bytes = 8 * 24 * 1024, total = big_floats * 4;
    for (int it = 0; it < iters; ++it) {
      const size_t chunk = ((size_t)it * gridDim.x + blockIdx.x) * wg_bytes % (total - wg_bytes);
      const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(big) + chunk / 4, 0, (int)wg_bytes, 0x00020000);
      u32x4_t v[24];
#pragma unroll
      for (int i = 0; i < 24; ++i) {
        // 10: wave's 16 rows of 3072 B = 48 KB region per TWO waves... keep it simple: rows (wave % 4) * 16 + l16 of a [64 rows][3072 B] panel
        //     (192 KB), instruction i covers bytes (wave / 4) * 1536 + i * 64 .. + 63 of each row
        const unsigned off = mode == 10 ? (unsigned)((((wave & 3) * 16 + (lane & 15)) * 3072) + (wave >> 2) * 1536 + i * 64 + (lane >> 4) * 16)
                                        : (unsigned)(wave * 24576 + i * 1024 + lane * 16);
        v[i] = __builtin_amdgcn_raw_buffer_load_b128(rb, (int)off, 0, 0);
      }
      float sacc = 0.f;
#pragma unroll
      for (int i = 0; i < 24; ++i) sacc += __uint_as_float(v[i][0]) + __uint_as_float(v[i][3]);
      acc += sacc * 1e-30f;
      __syncthreads();
    }
  }
  if (mode >= 12 && mode <= 15) {
    const int lane = tid & 63, wave = tid >> 6;
    for (int it = 0; it < iters; ++it) {
      bar_atomic(bar, nblk, epoch, spins);
      const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(big) + (size_t)it * (ROWS * KW), 0, ROWS * KW * 4, 0x00020000);
      u32x4_t v[13];
#pragma unroll
      for (int i = 0; i < 13; ++i) {
        // strided: wave load = 16 rows x 64 B of the [34][768] fp32 matrix (rows 16 (i % 3) + l16; out of range past row 33: zeros)
        const int row = 16 * (i % 3) + (lane & 15), col16 = (wave * 13 + i) / 3 % 48;
        const unsigned off = (mode & 2) ? (unsigned)((wave * 13 + i) * 1024 + lane * 16)
                                        : (row < ROWS ? (unsigned)(row * KW * 4 + col16 * 64 + (lane >> 4) * 16) : 0x80000000u);
        v[i] = (mode & 1) ? __builtin_amdgcn_raw_buffer_load_b128(rb, (int)off, 0, SC1) : __builtin_amdgcn_raw_buffer_load_b128(rb, (int)off, 0, 0);
      }
      float sacc = 0.f;
#pragma unroll
      for (int i = 0; i < 13; ++i) sacc += __uint_as_float(v[i][0]) + __uint_as_float(v[i][3]);
      acc += sacc * 1e-30f;
    }
  }
  if (spins >= SPIN) atomicAdd(&g_timeouts, 1u);
  if (acc == 12345.678f) sink[0] = acc;
}

int main(int argc, char** argv) {
  const int grid = argc > 1 ? atoi(argv[1]) : 192, iters = argc > 2 ? atoi(argv[2]) : 2000;
  unsigned *bar, *flags;
  float *act, *sink;
  u32x2_t* ll;
  CHECK(hipMalloc(&bar, 1024));
  CHECK(hipMalloc(&flags, 65536 * 4));
  { const int nc = argc > 5 ? atoi(argv[5]) : 8; CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_nc), &nc, sizeof(nc))); }
  CHECK(hipMalloc(&act, ROWS * KW * 4));
  CHECK(hipMalloc(&ll, ROWS * KW * 8));
  CHECK(hipMalloc(&sink, 64));
  float* big;
  const size_t big_floats = (size_t)1 << 28;      // 1 GB
  CHECK(hipMalloc(&big, big_floats * 4));
  CHECK(hipMemset(big, 0, big_floats * 4));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const char* names[18] = {"store + ack", "sc1 load chain", "flag ping-pong, different XCDs (2 hops)", "flag ping-pong, same XCD (2 hops)",
                           "barrier: atomic counter + flag (shipped)", "barrier: flag array, no atomics", "phase: store, shipped barrier, 104 KB sc1 loads",
                           "phase: store, flag-array barrier, 104 KB sc1 loads", "phase: (value, epoch) pairs, poll the data (208 KB)",
                           "phase: store, shipped barrier, 104 KB loads through L2",
                           "operand burst 192 KB / workgroup, 16 rows x 64 B per wave load", "operand burst 192 KB / workgroup, 1 KB contiguous per wave load",
                           "barrier + shared 104 KB, fresh addresses, cached, 16 x 64 B", "barrier + shared 104 KB, fresh addresses, sc1, 16 x 64 B",
                           "barrier + shared 104 KB, fresh addresses, cached, 1 KB contiguous", "barrier + shared 104 KB, fresh addresses, sc1, 1 KB contiguous",
                           "barrier: 8 arrival counters 4 KB apart, 8 lanes poll", "phase: store, 8-counter barrier, 104 KB sc1 loads"};
  for (int mode = (argc > 3 ? atoi(argv[3]) : 0); mode < (argc > 4 ? atoi(argv[4]) + 1 : 18); ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      CHECK(hipMemset(bar, 0, 1024));
      CHECK(hipMemset(flags, 0, 65536 * 4));
      CHECK(hipMemset(ll, 0, ROWS * KW * 8));
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(probe_k, dim3(grid), dim3(512), 0, 0, bar, flags, act, ll, sink, mode, mode >= 10 && mode < 12 ? 200 : iters, big, big_floats);
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipEventSynchronize(e1));
      float ms = 0.f;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      unsigned to = 0;
      CHECK(hipMemcpyFromSymbol(&to, HIP_SYMBOL(g_timeouts), sizeof(to)));
      if (rep == 1) printf("mode %d  %-62s %7.3f us per iteration%s\n", mode, names[mode], ms * 1e3f / (mode >= 10 && mode < 12 ? 200 : iters), to ? "  (POLL BUDGET EXHAUSTED)" : "");
    }
  }
  return 0;
}
