// tools/filter_rate.hip -- what bounds the filter pass of k_null_ie_q (gcre_ieq.hip), measured on its own: four sets of
// base counters in registers, one 2-KB plane load per position (L2-resident rows, prefetched one position ahead), one
// add + two bit-sliced comparisons against wave-uniform bounds per path.  Variants of how the bounds' bits reach the
// comparison:
//   0  s_bfe_i32 per bit and bound, v_bitop3 with a scalar operand          (the shipped loop)
//   1  the 2 x L masks of a bound pair loaded from a table with two s_load_dwordx16 one path ahead (no s_bfe)
//   2  masks already in vector registers (what the loop would cost if the bounds were free)
//   3  variant 0 with the low two planes not compared (bounds rounded to multiples of 4)
//   4  carry chain only: X + Y >= 0 for two pre-shifted operand pairs (no sums, no bounds)
//   5  the packed bounds broadcast into a vector register, v_bfe_i32 per bit and bound, v_bitop3 with vector operands only
// Diagnostic only.   hipcc --offload-arch=gfx950 -O3 tools/filter_rate.hip -o tools/filter_rate && tools/filter_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef uint32_t u32;
typedef uint64_t u64;
typedef u32 __attribute__((ext_vector_type(4))) u32x4;
typedef u32 __attribute__((ext_vector_type(16))) u32x16;
#define CONSTANT __attribute__((address_space(4)))

__device__ __forceinline__ u32 xor3(u32 a, u32 b, u32 c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
__device__ __forceinline__ u32 majority(u32 a, u32 b, u32 c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8); }
__device__ __forceinline__ u32 borrow3(u32 a, u32 b, u32 c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x8E); }
__device__ __forceinline__ u32 rdlane(u32 v, u32 t) { return (u32)__builtin_amdgcn_readlane((int)v, (int)t); }

constexpr int L = 10, LZ = 8, NSEG = 4;

template <int V, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_filter(const u32* base, const u32* planes, const u32* bounds,
                                                                                 const u32* table, u32* out, int nquads, int npaths) {
  const int lane = threadIdx.x & 63;
  const u32 lane16 = (u32)lane * 16u;
  u32 B[NSEG][L];
#pragma unroll
  for (int g = 0; g < NSEG; g++)
#pragma unroll
    for (int l = 0; l < L; l++) B[g][l] = base[((blockIdx.x * 4 + g) * L + l) * 64 + lane];
  u32 KV[NSEG][V == 2 ? 2 * L : 1];
  if (V == 2) {
#pragma unroll
    for (int g = 0; g < NSEG; g++)
#pragma unroll
      for (int l = 0; l < 2 * L; l++) KV[g][V == 2 ? l : 0] = base[(g * 2 * L + l) * 64 + lane] & 1u ? 0xffffffffu : 0u;
  }
  u64 todo_all = 0ull;
  const u32 valid = 0xffffffffu;
  for (int q = 0; q < nquads; q++) {
    // lane t <-> path t: packed bounds hi << 16 | lo, and the row of the added planes
    u32 lfv[NSEG];
#pragma unroll
    for (int g = 0; g < NSEG; g++) lfv[g] = bounds[((size_t)(blockIdx.x * 7 + q) * NSEG + g) % 4096 * 64 + lane];
    const u32 zrow = (lfv[0] * 2654435761u) >> 26;   // 64 rows of 2 KB: L2-resident
    u64 todo[NSEG] = {0ull, 0ull, 0ull, 0ull};
    auto issue = [&](u32 t, u32 (&ZZ)[LZ]) {
      __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)planes + (u64)rdlane(zrow, t) * 2048u), 0, 0x7fffffff, 0x00020000);
#pragma unroll
      for (int j = 0; j < LZ / 4; j++) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rz, lane16 + (u32)j * 1024u, 0, 0);
        ZZ[4 * j + 0] = v.x; ZZ[4 * j + 1] = v.y; ZZ[4 * j + 2] = v.z; ZZ[4 * j + 3] = v.w;
      }
    };
    auto filter_f = [&](int g, u32 t, const u32 (&Bg)[L], const u32 (&Z)[LZ], const u32x16& mlo, const u32x16& mhi) {
      const u32 lf = rdlane(lfv[g], t);
      u32 lfb = lf;
      if (V == 5) asm volatile("v_mov_b32 %0, %1" : "=v"(lfb) : "s"(lf));   // a vector copy the compiler cannot fold back
      u32 cy = 0u, blo = 0u, bhi = 0u;
      if (V == 4) {
        // two carry chains (lower and upper test as sign tests of pre-shifted sums): 2 x (L + 1) instructions
        u32 c2 = 0u;
#pragma unroll
        for (int l = 0; l < L; l++) {
          const u32 z = l < LZ ? Z[l < LZ ? l : 0] : 0u;
          cy = majority(Bg[l], z, cy);
          c2 = majority(~Bg[l], z, c2);
        }
        blo = cy; bhi = c2 ^ lf;
      } else {
#pragma unroll
        for (int l = 0; l < L; l++) {
          u32 w;
          if (l < LZ) {
            w = xor3(Bg[l], Z[l < LZ ? l : 0], cy);
            cy = majority(Bg[l], Z[l < LZ ? l : 0], cy);
          } else {
            w = Bg[l] ^ cy;
            cy = Bg[l] & cy;
          }
          if (V == 3 && l < 2) continue;
          u32 kl, kh;
          if (V == 0 || V == 3) {
            kl = (u32)__builtin_amdgcn_sbfe((int)lf, l, 1);
            kh = (u32)__builtin_amdgcn_sbfe((int)lf, 16 + l, 1);
          } else if (V == 5) {
            kl = (u32)__builtin_amdgcn_sbfe((int)lfb, l, 1);
            kh = (u32)__builtin_amdgcn_sbfe((int)lfb, 16 + l, 1);
          } else if (V == 1) {
            kl = mlo[l]; kh = mhi[l];
          } else {
            kl = KV[g][V == 2 ? l : 0]; kh = KV[g][V == 2 ? L + l : 0];
          }
          blo = borrow3(w, kl, blo);
          bhi = borrow3(kh, w, bhi);
        }
      }
      if (__builtin_amdgcn_ballot_w64(((blo | bhi) & valid) != 0u) != 0ull) todo[g] |= 1ull << t;
    };
    auto masks = [&](int g, u32 t, u32x16& mlo, u32x16& mhi) {
      if (V == 1) {
        const u32 lf = rdlane(lfv[g], t);
        mlo = *(const u32x16 CONSTANT*)((const char CONSTANT*)table + (u64)((lf & 0x3ffu) * 64u));
        mhi = *(const u32x16 CONSTANT*)((const char CONSTANT*)table + (u64)(((lf >> 16) & 0x3ffu) * 64u));
      }
    };
    u32 ZA[LZ], ZB[LZ];
    // masks of the next path in flight while this one is tested: two sets, alternating along the (position, segment) order
    u32x16 mlP, mhP, mlQ, mhQ;
    mlP = mhP = mlQ = mhQ = u32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const u32 last = (u32)npaths - 1u;
    auto at = [&](u32 t) -> u32 { return t < last ? t : last; };
    issue(0u, ZA);
    masks(0, 0u, mlP, mhP);
    for (u32 t = 0; t < (u32)npaths; t += 2) {
      issue(at(t + 1), ZB);
      masks(1, t, mlQ, mhQ);           filter_f(0, t, B[0], ZA, mlP, mhP);
      masks(2, t, mlP, mhP);           filter_f(1, t, B[1], ZA, mlQ, mhQ);
      masks(3, t, mlQ, mhQ);           filter_f(2, t, B[2], ZA, mlP, mhP);
      masks(0, at(t + 1), mlP, mhP);   filter_f(3, t, B[3], ZA, mlQ, mhQ);
      issue(at(t + 2), ZA);
      masks(1, at(t + 1), mlQ, mhQ);   filter_f(0, t + 1, B[0], ZB, mlP, mhP);
      masks(2, at(t + 1), mlP, mhP);   filter_f(1, t + 1, B[1], ZB, mlQ, mhQ);
      masks(3, at(t + 1), mlQ, mhQ);   filter_f(2, t + 1, B[2], ZB, mlP, mhP);
      masks(0, at(t + 2), mlP, mhP);   filter_f(3, t + 1, B[3], ZB, mlQ, mhQ);
    }
#pragma unroll
    for (int g = 0; g < NSEG; g++) todo_all += (u64)__builtin_popcountll(todo[g]);
  }
  if (lane == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = (u32)todo_all;
}

template <int V, int WPE>
int run(const char* name, int blocks_per_cu) {
  const int blocks = 256 * blocks_per_cu, nquads = 64, npaths = 16;
  u32 *base, *planes, *bounds, *table, *out;
  CHECK(hipMalloc(&base, (size_t)blocks * 4 * L * 64 * 4 + (1 << 20)));
  CHECK(hipMalloc(&planes, 64 * 2048));
  CHECK(hipMalloc(&bounds, 4096 * 64 * 4));
  CHECK(hipMalloc(&table, 1024 * 64));
  CHECK(hipMalloc(&out, (size_t)blocks * 16));
  std::vector<u32> h((size_t)blocks * 4 * L * 64 + (1 << 18));
  for (size_t i = 0; i < h.size(); i++) h[i] = (u32)(i * 2654435761u) ^ (u32)(i >> 3) * 0x9e3779b9u;
  CHECK(hipMemcpy(base, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(planes, h.data(), 64 * 2048, hipMemcpyHostToDevice));
  std::vector<u32> hb(4096 * 64);
  for (size_t i = 0; i < hb.size(); i++) { const u32 r = (u32)(i * 2246822519u) >> 8; hb[i] = (r & 0x3ffu) | (((r >> 10) & 0x3ffu) << 16); }
  CHECK(hipMemcpy(bounds, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  std::vector<u32> ht(1024 * 16, 0u);
  for (int v = 0; v < 1024; v++) for (int l = 0; l < L; l++) ht[(size_t)v * 16 + l] = (v >> l) & 1 ? 0xffffffffu : 0u;
  CHECK(hipMemcpy(table, ht.data(), ht.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  // LDS per block caps the blocks per CU (160 KB per CU)
  const size_t lds = (size_t)(160 * 1024 / blocks_per_cu) - 1024;
  CHECK(hipFuncSetAttribute((const void*)k_filter<V, WPE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  float ms = 0;
  for (int rep = 0; rep < 3; rep++) {
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((k_filter<V, WPE>), dim3(blocks), dim3(256), lds, 0, base, planes, bounds, table, out, nquads, npaths);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    CHECK(hipEventElapsedTime(&ms, a, b));
  }
  const double path_tiles = (double)blocks * 4 * nquads * npaths * NSEG;
  const double clk_per_pt_simd = ms * 1e-3 * 2.4e9 / (path_tiles / 1024.0);
  printf("%-28s waves/SIMD %d  %.3f ms  %.1f G path-tiles/s  %.0f clocks per path-tile per SIMD (at 2.4 GHz)\n", name, blocks_per_cu, ms,
         path_tiles / (ms * 1e-3) / 1e9, clk_per_pt_simd);
  hipFree(base); hipFree(planes); hipFree(bounds); hipFree(table); hipFree(out);
  return 0;
}

int main() {
  if (run<0, 3>("sbfe + scalar bitop3", 3)) return 1;
  if (run<0, 4>("sbfe + scalar bitop3", 4)) return 1;
  if (run<1, 3>("mask table (s_load x16)", 3)) return 1;
  if (run<1, 4>("mask table (s_load x16)", 4)) return 1;
  if (run<2, 3>("masks in VGPRs (bound free)", 3)) return 1;
  if (run<3, 3>("sbfe, low 2 planes skipped", 3)) return 1;
  if (run<3, 4>("sbfe, low 2 planes skipped", 4)) return 1;
  if (run<5, 3>("v_bfe on a vector copy", 3)) return 1;
  if (run<5, 4>("v_bfe on a vector copy", 4)) return 1;
  if (run<4, 3>("two carry chains only", 3)) return 1;
  if (run<4, 4>("two carry chains only", 4)) return 1;
  return 0;
}
