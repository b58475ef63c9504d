// tools/row_gather_rate.hip -- how fast can a CU pull random rows of a table that lives in L2 / Infinity Cache?
// The sparse null kernel's mask-row loads are exactly this pattern: wave-uniform random row, lane*W bytes inside it.
//   hipcc --offload-arch=gfx950 -O3 tools/row_gather_rate.hip -o tools/row_gather_rate && tools/row_gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32;
typedef u32 __attribute__((ext_vector_type(2))) u32x2;
typedef u32 __attribute__((ext_vector_type(4))) u32x4;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// each wave: ITERS blocks of 16 random rows; row = 64 lanes x W dwords; xor-reduce so nothing is dead
template <int W>
__global__ __launch_bounds__(256) void k_rows(const u32* table, u32 nrows, u32* out, int iters, u32 seed) {
  const int lane = threadIdx.x & 63;
  const u32 wave = __builtin_amdgcn_readfirstlane((blockIdx.x * 256 + threadIdx.x) >> 6);
  u32 s = seed ^ (wave * 2654435761u);
  u32 acc[W];
  for (int w = 0; w < W; w++) acc[w] = 0;
  for (int it = 0; it < iters; it++) {
    u32 rows[16];
#pragma unroll
    for (int j = 0; j < 16; j++) { s = s * 1664525u + 1013904223u; rows[j] = __builtin_amdgcn_readfirstlane((s >> 8) % nrows); }
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const u32* p = table + (size_t)rows[j] * 64 * W + lane * W;
      if (W == 1) acc[0] ^= p[0];
      else if (W == 2) { u32x2 v = *(const u32x2*)p; acc[0] ^= v.x; acc[1] ^= v.y; }
      else { u32x4 v = *(const u32x4*)p; acc[0] ^= v.x; acc[1] ^= v.y; acc[2] ^= v.z; acc[3] ^= v.w; }
    }
  }
  u32 r = 0;
  for (int w = 0; w < W; w++) r ^= acc[w];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

// QUAD: one dwordx4 load fetches FOUR random 256-byte rows, one per group of 16 lanes (lane = 16 B of its group's row):
// the access pattern of a wave that works on four joined paths at once.  ACT = lane groups that take part (exec mask).
template <int ACT>
__global__ __launch_bounds__(256) void k_rows_quad(const u32* table, u32 nrows, u32* out, int iters, u32 seed) {
  const int lane = threadIdx.x & 63;
  const u32 wave = __builtin_amdgcn_readfirstlane((blockIdx.x * 256 + threadIdx.x) >> 6);
  u32 s = seed ^ (wave * 2654435761u) ^ ((u32)(lane >> 4) * 0x9e3779b9u);
  u32 acc[4] = {0, 0, 0, 0};
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)table, 0, 0x7fffffff, 0x00020000);
  for (int it = 0; it < iters; it++) {
    u32 offs[16];
#pragma unroll
    for (int j = 0; j < 16; j++) { s = s * 1664525u + 1013904223u; offs[j] = __umulhi(s, nrows) * 256u + (u32)(lane & 15) * 16u; }
    if ((lane >> 4) < ACT) {
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, offs[j], 0, 0);
        acc[0] ^= v.x; acc[1] ^= v.y; acc[2] ^= v.z; acc[3] ^= v.w;
      }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
}

// MASKED: the dword form (one 256-byte row per wave load) with only `act` lanes taking part: what the bound filter's
// uncertain lanes issue.  Is an exec-masked wave load cheaper for the vector-memory pipe than a full one?
__global__ __launch_bounds__(256) void k_rows_masked(const u32* table, u32 nrows, u32* out, int iters, u32 seed, int act) {
  const int lane = threadIdx.x & 63;
  const u32 wave = __builtin_amdgcn_readfirstlane((blockIdx.x * 256 + threadIdx.x) >> 6);
  u32 s = seed ^ (wave * 2654435761u);
  u32 acc = 0;
  const bool on = ((lane * 37 + 11) & 63) < act;   // scattered lanes
  for (int it = 0; it < iters; it++) {
    u32 rows[16];
#pragma unroll
    for (int j = 0; j < 16; j++) { s = s * 1664525u + 1013904223u; rows[j] = __builtin_amdgcn_readfirstlane((s >> 8) % nrows); }
    if (on) {
#pragma unroll
      for (int j = 0; j < 16; j++) acc ^= table[(size_t)rows[j] * 64 + lane];
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int run_masked(u32 nrows, int blocks_per_cu, int iters, int act) {
  const size_t bytes = (size_t)nrows * 256;
  u32 *table, *out;
  CHECK(hipMalloc(&table, bytes)); CHECK(hipMemset(table, 1, bytes));
  const int blocks = 256 * blocks_per_cu;
  CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  float ms = 0;
  for (int rep = 0; rep < 3; rep++) {
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(k_rows_masked, dim3(blocks), dim3(256), 0, 0, table, nrows, out, iters, 12345u + rep, act);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    CHECK(hipEventElapsedTime(&ms, a, b));
  }
  const double loads = (double)blocks * 4 * iters * 16;
  printf("masked dword rows, %2d of 64 lanes  table %6.1f MB  %2d waves/CU : %.3f ms  %.2f Gload/s  %.1f clk/load/CU @2.3GHz\n", act,
         bytes / 1e6, blocks_per_cu * 4, ms, loads / ms / 1e6, ms * 1e-3 * 2.3e9 * 256 / loads);
  hipFree(table); hipFree(out);
  return 0;
}

template <int ACT>
int run_quad(u32 nrows, int blocks_per_cu, int iters) {
  const size_t bytes = (size_t)nrows * 256;
  u32 *table, *out;
  CHECK(hipMalloc(&table, bytes)); CHECK(hipMemset(table, 1, bytes));
  const int blocks = 256 * blocks_per_cu;
  CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  float ms = 0;
  for (int rep = 0; rep < 3; rep++) {
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(k_rows_quad<ACT>, dim3(blocks), dim3(256), 0, 0, table, nrows, out, iters, 12345u + rep);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    CHECK(hipEventElapsedTime(&ms, a, b));
  }
  const double loads = (double)blocks * 4 * iters * 16;
  const double gb = loads * ACT * 256 / 1e9;
  printf("quad x4 (%d of 4 groups active, 256-B rows)  table %6.1f MB  %2d waves/CU : %.3f ms  %.2f Gload/s  %.1f TB/s  %.1f clk/load/CU @2.3GHz\n", ACT,
         bytes / 1e6, blocks_per_cu * 4, ms, loads / ms / 1e6, gb / ms, ms * 1e-3 * 2.3e9 * 256 / loads);
  hipFree(table); hipFree(out);
  return 0;
}

template <int W>
int run(u32 nrows, int blocks_per_cu, int iters) {
  const size_t bytes = (size_t)nrows * 64 * W * 4;
  u32 *table, *out;
  CHECK(hipMalloc(&table, bytes)); CHECK(hipMemset(table, 1, bytes));
  const int blocks = 256 * blocks_per_cu;
  CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  float ms = 0;
  for (int rep = 0; rep < 3; rep++) {
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(k_rows<W>, dim3(blocks), dim3(256), 0, 0, table, nrows, out, iters, 12345u + rep);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    CHECK(hipEventElapsedTime(&ms, a, b));
  }
  const double loads = (double)blocks * 4 * iters * 16;
  const double gb = loads * 64 * W * 4 / 1e9;
  printf("row %4d B  table %6.1f MB  %2d waves/CU : %.3f ms  %.2f Gload/s  %.1f TB/s  %.1f B/clk/CU @2.3GHz\n", 64 * W * 4,
         bytes / 1e6, blocks_per_cu * 4, ms, loads / ms / 1e6, gb / ms, gb / ms * 1e12 / 256 / 2.3e9 / 1e3 * 1e0);
  hipFree(table); hipFree(out);
  return 0;
}

int main() {
  for (int bpc : {2, 3, 5, 8}) {
    for (u32 nrows : {5121u, 25605u}) {      // one tile (1.3 MB) / five tiles (6.5 MB) of 256-B rows
      if (run<1>(nrows, bpc, 256)) return 1;
      if (run<2>(nrows / 2, bpc, 256)) return 1;
      if (run<4>(nrows / 4, bpc, 256)) return 1;
      if (run_quad<4>(nrows, bpc, 256)) return 1;
      if (run_quad<2>(nrows, bpc, 256)) return 1;
    }
    for (int act : {1, 2, 4, 16, 64})
      if (run_masked(5121u, bpc, 256, act)) return 1;
  }
  return 0;
}
