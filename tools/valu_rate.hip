// tools/valu_rate.hip -- measured issue rates of the integer VALU ops the null kernel is built from
// (v_and_b32 with a scalar operand, v_bcnt_u32_b32 with accumulate) and the clock the chip holds while
// running them.  Diagnostic only: gives the ceiling k_null is priced against in DESIGN.md.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ACC = 32;
constexpr int ITERS = 2048;

// MODE 0: and+bcnt pairs (the k_null inner loop), 1: bcnt only, 2: and only (xor-accumulated), 3: add only,
// 4: v_bitop3_b32 (xor3 into an accumulator, independent chains), 5: full-adder pairs xor3 + majority on a carry chain
// of 8 planes (the k_null_ie arithmetic: every second instruction depends on the previous pair), 6: mode 5 with one
// s_bfe/s_lshl scalar pair and one v_readlane per four VALU (the instruction mix of the pruned kernel's path loop)
template <int MODE>
__global__ __launch_bounds__(256) void k_rate(const uint32_t* in, uint32_t* out, uint64_t* clk) {
  uint32_t acc[ACC], m[ACC];
#pragma unroll
  for (int i = 0; i < ACC; i++) { acc[i] = 0; m[i] = in[threadIdx.x + 256 * i]; }
  uint32_t s = __builtin_amdgcn_readfirstlane(in[blockIdx.x & 255]);
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < ACC; i++) {
      if (MODE == 0) acc[i] += __builtin_popcount(s & m[i]);
      else if (MODE == 1) acc[i] += __builtin_popcount(m[i]);
      else if (MODE == 2) acc[i] ^= (s & m[i]);
      else if (MODE == 3) acc[i] += m[i];
      else if (MODE == 4) acc[i] = __builtin_amdgcn_bitop3_b32(acc[i], m[i], s, 0x96);
    }
    if (MODE == 5 || MODE == 6) {
      // four independent ripple adders of 8 planes each: acc[8c..8c+7] += m[8c..8c+7]  (2 bitop3 per plane)
#pragma unroll
      for (int c = 0; c < 4; c++) {
        uint32_t cy = s;
#pragma unroll
        for (int l = 0; l < 8; l++) {
          const uint32_t a = acc[8 * c + l], b = m[8 * c + l];
          acc[8 * c + l] = __builtin_amdgcn_bitop3_b32(a, b, cy, 0x96);
          cy = __builtin_amdgcn_bitop3_b32(a, b, cy, 0xE8);
          if (MODE == 6 && (l & 1)) {
            const uint32_t k = (uint32_t)__builtin_amdgcn_sbfe((int)s, l, 1);
            const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)acc[8 * c + l], (l * 7 + c) & 63);
            s = (s ^ k) + r;
          }
        }
        m[8 * c] ^= cy;
      }
    }
    s = s * 1664525u + 1013904223u;   // scalar, keeps the compiler from hoisting s & m[i]
    if (MODE == 1 || MODE == 3) {
#pragma unroll
      for (int i = 0; i < ACC; i++) asm volatile("" : "+v"(m[i]));
    }
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
  uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < ACC; i++) r += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int MODE>
int run(const char* name, int blocks, double valu_per_iter_per_acc) {
  uint32_t *in, *out; uint64_t* clk;
  CHECK(hipMalloc(&in, 256 * ACC * 4)); CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4)); CHECK(hipMalloc(&clk, (size_t)blocks * 16));
  std::vector<uint32_t> h(256 * ACC);
  for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u) ^ 0x9e3779b9u;
  CHECK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  for (int rep = 0; rep < 3; rep++) {
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(256), 0, 0, in, out, clk);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
  }
  float ms; CHECK(hipEventElapsedTime(&ms, a, b));
  std::vector<uint64_t> hc(2 * (size_t)blocks);
  CHECK(hipMemcpy(hc.data(), clk, hc.size() * 8, hipMemcpyDeviceToHost));
  double mhz = 0; for (int i = 0; i < blocks; i++) mhz += (double)hc[2 * i] / (double)hc[2 * i + 1] * 100.0;
  mhz /= blocks;
  const double lane_ops = (double)blocks * 256 * ITERS * ACC * valu_per_iter_per_acc;
  const double rate = lane_ops / (ms * 1e-3);
  printf("%-10s blocks %5d  %.3f ms  %.2f Tlane-op/s  clock %.0f MHz  -> %.1f lane-ops/clk/CU (peak 128)\n", name, blocks, ms,
         rate / 1e12, mhz, rate / (mhz * 1e6) / 256.0);
  hipFree(in); hipFree(out); hipFree(clk);
  return 0;
}

int main() {
  for (int bpc : {1, 2, 4, 8}) {
    const int blocks = 256 * bpc;
    if (run<0>("and+bcnt", blocks, 2)) return 1;
    if (run<1>("bcnt", blocks, 1)) return 1;
    if (run<2>("and+xor", blocks, 2)) return 1;
    if (run<3>("add", blocks, 1)) return 1;
    if (run<4>("bitop3", blocks, 1)) return 1;
    if (run<5>("fulladd", blocks, 100.0 / 32)) return 1;   // 64 bitop3 + 32 v_mov + 4 v_xor per iteration (ISA checked)
    if (run<6>("fulladd+s", blocks, 116.0 / 32)) return 1; // + 16 v_readlane; 51 SALU ride along
  }
  return 0;
}
