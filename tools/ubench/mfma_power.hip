// What the bf16 matrix pipe SUSTAINS on this part, as a function of what else the CU does and of the operand data (development aid,
// not part of libivg).  Build: make -C tools/ubench ; run on the GPU box:  bin/mfma_power [ms per arm]
//
// Question behind it (round 5): every structure of the 3x3 convolution -- 256- or 512-pixel workgroups, one or two per CU, halo
// fragments prefetched or not -- ends at the same 1.30-1.45 PFLOP/s while its MFMA-busy share moves between 50 and 74 %: the cycle
// count of the launch (GRBM_GUI_ACTIVE) and its wall time give 1.82-2.17 GHz, the busier the kernel the lower.  This program measures
// the rate (wall clock from events; tools/sessions/r05_s5.sh samples socket power and sclk beside it) of
//   MFMA only, all-zero operands                     (the "peak" a datasheet-style micro-benchmark reports)
//   MFMA only, random bf16 operands                  (same instruction stream, data that toggles the multipliers)
//   random + 6 / 8 / 16 ds_read_b128 per 16 MFMAs    (the LDS fragment traffic of conv3x3w: 12 per 32; of conv3x3.hip: 8 per 16), the
//                                                    reads software-pipelined by a whole iteration: no MFMA ever waits for one
//   zero operands + 8 reads per 16 MFMAs             (separates the cost of the reads from the cost of the data)
// 256 CUs x 8 waves (2 per SIMD) x 256 registers, 64 independent accumulators per wave: the matrix pipe is never starved.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int LDS_PER_16>   // ds_read_b128 per 16 MFMAs: 0, 6 (conv3x3w: 12 per 32), 8 (conv3x3.hip), 16
__global__ __launch_bounds__(512, 2) void mfma_kernel(const u32x4* __restrict__ ops, int iters, float* __restrict__ out, long long* __restrict__ clk) {
  __shared__ u32x4 lds[512 * 4];
  const int tid = threadIdx.x;
  for (int i = 0; i < 4; ++i) lds[i * 512 + tid] = ops[(i * 512 + tid) & 4095];
  __syncthreads();
  u32x4 a[4], b[4], nb[4];
  for (int i = 0; i < 4; ++i) { a[i] = ops[(tid * 4 + i) & 4095]; b[i] = ops[(tid * 4 + i + 2048) & 4095]; nb[i] = b[i]; }
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      // fragment reads whose results feed the NEXT iteration's MFMAs (software-pipelined by 16 MFMAs: no wait on their latency)
      const bool rd = LDS_PER_16 == 16 || (LDS_PER_16 == 8 && (i & 1) == 0) || (LDS_PER_16 == 6 && (i == 0 || i == 3 || i == 5 || i == 8 || i == 11 || i == 13));
      if (rd) nb[i & 3] = lds[((it * 16 + i) * 64 + tid) & 2047];
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i & 3]), "v"(b[(i + 1) & 3]));
    }
    if (LDS_PER_16) {
#pragma unroll
      for (int i = 0; i < 4; ++i) b[i] = nb[i];
    }
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 512 + tid] = s;
  if (tid == 0) clk[blockIdx.x] = t1 - t0;
}

// the same stream with v_mfma_f32_32x32x16_bf16: twice the MACs per operand element read from the register file (16 per element
// against 8) -- does the bigger tile sustain more under the power cap?  8 independent 32x32 accumulators (128 registers) per wave.
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int LDS_PER_16>
__global__ __launch_bounds__(512, 2) void mfma32_kernel(const u32x4* __restrict__ ops, int iters, float* __restrict__ out, long long* __restrict__ clk) {
  __shared__ u32x4 lds[512 * 4];
  const int tid = threadIdx.x;
  for (int i = 0; i < 4; ++i) lds[i * 512 + tid] = ops[(i * 512 + tid) & 4095];
  __syncthreads();
  u32x4 a[4], b[4], nb[4];
  for (int i = 0; i < 4; ++i) { a[i] = ops[(tid * 4 + i) & 4095]; b[i] = ops[(tid * 4 + i + 2048) & 4095]; nb[i] = b[i]; }
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {   // 8 MFMAs of 32x32x16 = the FLOPs of 16 MFMAs of 16x16x32
      const bool rd = LDS_PER_16 == 8 || (LDS_PER_16 == 6 && i != 2 && i != 6);
      if (rd) nb[i & 3] = lds[((it * 8 + i) * 64 + tid) & 2047];
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i & 3]), "v"(b[(i + 1) & 3]));
    }
    if (LDS_PER_16) {
#pragma unroll
      for (int i = 0; i < 4; ++i) b[i] = nb[i];
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 512 + tid] = s;
  if (tid == 0) clk[blockIdx.x] = 0;
}

// operand reuse between CONSECUTIVE MFMAs (random operands, no LDS): does the order in which a 4 x 4 fragment tile is walked change
// what the pipe sustains?  PAT 0: both operands change at every MFMA; 1: row-major over (b, a) -- the B operand stays for four MFMAs,
// both change at a row end (conv3x3's order); 2: snake -- exactly one operand changes at every MFMA; 3: the same two fragments always.
template <int PAT>
__global__ __launch_bounds__(512, 2) void mfma_order_kernel(const u32x4* __restrict__ ops, int iters, float* __restrict__ out, long long* __restrict__ clk) {
  const int tid = threadIdx.x;
  u32x4 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = ops[(tid * 4 + i) & 4095]; b[i] = ops[(tid * 4 + i + 2048) & 4095]; }
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      int ia, ib;
      if (PAT == 0) { ia = i & 3; ib = (i + 1 + (i >> 2)) & 3; }
      else if (PAT == 1) { ib = i >> 2; ia = i & 3; }
      else if (PAT == 2) { ib = i >> 2; ia = (ib & 1) ? 3 - (i & 3) : (i & 3); }
      else { ia = 0; ib = 0; }
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[ib * 4 + ia]) : "v"(a[ia]), "v"(b[ib]));
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 512 + tid] = s;
  if (tid == 0) clk[blockIdx.x] = 0;
}

static double now_s() { timespec ts; clock_gettime(CLOCK_REALTIME, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }
template <int L, bool BIG = false, int ORDER = -1>
static void run(const char* name, const u32x4* ops, int iters, float* out, long long* clk, int grid, double target_ms) {
  const double t_begin = now_s();
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  if (ORDER >= 0) hipLaunchKernelGGL(mfma_order_kernel<(ORDER < 0 ? 0 : ORDER)>, dim3(grid), dim3(512), 0, 0, ops, 1000, out, clk);
  else if (BIG) hipLaunchKernelGGL(mfma32_kernel<L>, dim3(grid), dim3(512), 0, 0, ops, 1000, out, clk);
  else hipLaunchKernelGGL(mfma_kernel<L>, dim3(grid), dim3(512), 0, 0, ops, 1000, out, clk);   // warm-up
  CK(hipDeviceSynchronize());
  // size the launch for ~target_ms
  CK(hipEventRecord(e0));
  if (ORDER >= 0) hipLaunchKernelGGL(mfma_order_kernel<(ORDER < 0 ? 0 : ORDER)>, dim3(grid), dim3(512), 0, 0, ops, iters, out, clk);
  else if (BIG) hipLaunchKernelGGL(mfma32_kernel<L>, dim3(grid), dim3(512), 0, 0, ops, iters, out, clk);
  else hipLaunchKernelGGL(mfma_kernel<L>, dim3(grid), dim3(512), 0, 0, ops, iters, out, clk);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  const int it2 = (int)(iters * target_ms / ms);
  double best = 0, best_ms = 0; long long cyc = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    if (ORDER >= 0) hipLaunchKernelGGL(mfma_order_kernel<(ORDER < 0 ? 0 : ORDER)>, dim3(grid), dim3(512), 0, 0, ops, it2, out, clk);
    else if (BIG) hipLaunchKernelGGL(mfma32_kernel<L>, dim3(grid), dim3(512), 0, 0, ops, it2, out, clk);
    else hipLaunchKernelGGL(mfma_kernel<L>, dim3(grid), dim3(512), 0, 0, ops, it2, out, clk);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double flop = 2.0 * 16 * 16 * 32 * 16.0 * it2 * 8.0 * grid;
    std::vector<long long> h(grid);
    CK(hipMemcpy(h.data(), clk, grid * sizeof(long long), hipMemcpyDeviceToHost));
    long long mx = 0; for (auto v : h) mx = v > mx ? v : mx;
    if (flop / ms > best) { best = flop / ms; best_ms = ms; cyc = mx; }
  }
  // per SIMD: 2 waves x 16 MFMAs x 16 clocks per iteration if the pipe never idles
  const double mfma_clk = 2.0 * 16 * 16 * it2;
  (void)cyc;
  printf("%-50s %8.2f ms  %7.1f TFLOP/s  = %.3f of 2.5 PF | busy x clock >= %.2f GHz | wall %.2f .. %.2f s\n",
         name, best_ms, best / 1e9, best / 1e9 / 2500.0, mfma_clk / (best_ms * 1e6), t_begin, now_s());
  fflush(stdout);
}

int main(int argc, char** argv) {
  const double target_ms = argc > 1 ? atof(argv[1]) : 20.0;
  const int grid = 256;
  u32x4 *zeros, *rnd; float* out; long long* clk;
  CK(hipMalloc(&zeros, 4096 * 16)); CK(hipMalloc(&rnd, 4096 * 16)); CK(hipMalloc(&out, grid * 512 * 4)); CK(hipMalloc(&clk, grid * 8));
  CK(hipMemset(zeros, 0, 4096 * 16));
  std::vector<unsigned short> h(4096 * 8);
  unsigned s = 12345u;
  for (auto& v : h) {   // random bf16 in (-2, 2): sign, exponent 125..127, random mantissa -- finite products, toggling bits
    s = s * 1664525u + 1013904223u;
    const unsigned sign = (s >> 31) & 1, ex = 125 + ((s >> 24) & 3) % 3, man = (s >> 8) & 0x7f;
    v = (unsigned short)((sign << 15) | (ex << 7) | man);
  }
  CK(hipMemcpy(rnd, h.data(), 4096 * 16, hipMemcpyHostToDevice));
  printf("# bf16 v_mfma_f32_16x16x32_bf16, 256 workgroups x 8 waves, %g ms per arm (3 repeats, best)\n", target_ms);
  run<0>("MFMA only, zero operands", zeros, 20000, out, clk, grid, target_ms);
  run<0>("MFMA only, random operands", rnd, 20000, out, clk, grid, target_ms);
  run<6>("random + 6 ds_read_b128 per 16 MFMAs (conv3x3w)", rnd, 20000, out, clk, grid, target_ms);
  run<8>("random + 8 ds_read_b128 per 16 MFMAs (conv3x3)", rnd, 20000, out, clk, grid, target_ms);
  run<16>("random + 16 ds_read_b128 per 16 MFMAs", rnd, 20000, out, clk, grid, target_ms);
  run<8>("ZERO operands + 8 ds_read_b128 per 16 MFMAs", zeros, 20000, out, clk, grid, target_ms);
  run<0, true>("32x32x16: MFMA only, zero operands", zeros, 20000, out, clk, grid, target_ms);
  run<0, true>("32x32x16: MFMA only, random operands", rnd, 20000, out, clk, grid, target_ms);
  run<6, true>("32x32x16: random + 6 ds_read_b128 per 16-MFMA equiv", rnd, 20000, out, clk, grid, target_ms);
  run<8, true>("32x32x16: random + 8 ds_read_b128 per 16-MFMA equiv", rnd, 20000, out, clk, grid, target_ms);
  run<0, false, 0>("order: both operands change at every MFMA", rnd, 20000, out, clk, grid, target_ms);
  run<0, false, 1>("order: row-major (b outer, a inner; conv3x3)", rnd, 20000, out, clk, grid, target_ms);
  run<0, false, 2>("order: snake (one operand changes per MFMA)", rnd, 20000, out, clk, grid, target_ms);
  run<0, false, 3>("order: the same two fragments always", rnd, 20000, out, clk, grid, target_ms);
  run<0>("MFMA only, zero operands (again, warm chip)", zeros, 20000, out, clk, grid, target_ms);
  return 0;
}
