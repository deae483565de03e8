// Decode-step micro-benchmarks (development aid, not part of libivg): what a dependent kernel boundary, a single-burst
// GEMM-shaped load and two concurrent graph branches cost on this box.  Build: make -C tools/ubench ; run on the GPU box.
//   T1  boundary      : chain of empty kernels inside one hipGraph -> us per launch
//   T2  burst         : every lane requests U x 16 B up front (fragment-shaped 16 rows x 64 B, or 1 KiB lines; from one small
//                       L2-resident buffer or from unique HBM), LDS reduce, one store -> us per launch, B/clk/CU
//   T3  concurrency   : an HBM-streaming kernel (decode-attention-like) next to a chain of T2 kernels, on two streams and as
//                       two branches of one graph -> do they overlap?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void empty_kernel(int* p) { if (p && threadIdx.x == 9999) p[0] = 1; }

// MODE 0: fragment-shaped (lane (lr = l & 15, lg = l >> 4) reads row lr, 16-byte chunk lg of a 64-byte segment; rows `ld` apart)
// MODE 1: line-shaped (a wave reads 1 KiB contiguous)
// SRC  0: every workgroup reads the SAME small buffer (activation-like, L2 resident)   1: unique bytes per workgroup (weights)
template <int U, int MODE, int NT>
__global__ __launch_bounds__(512) void burst_kernel(const char* __restrict__ src, long wg_stride, int ld, float* __restrict__ out) {
  __shared__ float red[8];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const char* base = src + (long)blockIdx.x * wg_stride;
  u32x4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    long off;
    if (MODE == 0) off = ((long)(wave * 16 + (lane & 15))) * ld + u * 64 + (lane >> 4) * 16;   // 16 rows x 64 B per instruction; a wave owns 16 rows of U*64 B
    else off = ((long)(wave * U + u) * 64 + lane) * 16;
    const u32x4* p = (const u32x4*)(base + off);
    v[u] = NT ? __builtin_nontemporal_load(p) : *p;
  }
  unsigned acc = 0;
#pragma unroll
  for (int u = 0; u < U; ++u) acc ^= v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
  float f = (float)(acc & 0xff);
  for (int o = 32; o > 0; o >>= 1) f += __shfl_xor(f, o, 64);
  if (lane == 0) red[wave] = f;
  __syncthreads();
  if (tid == 0) { float s = 0; for (int w = 0; w < 8; ++w) s += red[w]; out[blockIdx.x] = s; }
}

// HBM streamer: each workgroup reads `bytes_per_wg` contiguous bytes (8 x 16 B in flight per lane), one store
template <int NT>
__global__ __launch_bounds__(256) void stream_kernel(const char* __restrict__ src, long bytes_per_wg, float* __restrict__ out) {
  const char* base = src + (long)blockIdx.x * bytes_per_wg;
  unsigned acc = 0;
  for (long o = (long)threadIdx.x * 16; o < bytes_per_wg; o += 256L * 16 * 8) {
    u32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long oo = o + (long)u * 256 * 16;
      const u32x4* p = (const u32x4*)(base + (oo < bytes_per_wg ? oo : 0));
      v[u] = NT ? __builtin_nontemporal_load(p) : *p;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc ^= v[u][0] ^ v[u][3];
  }
  if (acc == 0x12345678u) out[blockIdx.x] = 1.f;
}

static double time_graph(hipGraphExec_t ex, hipStream_t st, int reps) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  CK(hipGraphLaunch(ex, st)); CK(hipStreamSynchronize(st));
  std::vector<float> t;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(a, st)); CK(hipGraphLaunch(ex, st)); CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms);
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2] * 1e3;   // us
}

template <typename F>
static hipGraphExec_t capture(hipStream_t st, F&& f) {
  hipGraph_t g; hipGraphExec_t ex;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  f();
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
  CK(hipGraphDestroy(g));
  return ex;
}

static char* g_hbm; static size_t g_hbm_bytes; static float* g_out; static char* g_small;
static double g_clk_ghz = 2.1;

template <int U, int MODE, int NT>
static void run_burst(hipStream_t st, int grid, int src, const char* tag) {
  const int N = 48;
  const long per_wg = 8L * U * 1024;                 // bytes one workgroup requests
  const int ld = U * 64;                             // MODE 0: a wave's tile is 16 rows of U * 64 B (like 16 weight rows of its K slice)
  const long foot = per_wg;
  int launch = 0;
  hipGraphExec_t ex = capture(st, [&] {
    for (int i = 0; i < N; ++i) {
      const char* s; long stride;
      if (src == 0) { s = g_small; stride = 0; }
      else { const long span = (long)grid * foot; const long slots = (long)(g_hbm_bytes / span); s = g_hbm + (launch++ % slots) * span; stride = foot; }
      hipLaunchKernelGGL((burst_kernel<U, MODE, NT>), dim3(grid), dim3(512), 0, st, s, stride, ld, g_out);
    }
  });
  const double us = time_graph(ex, st, 7) / N;
  printf("T2 burst  %-28s grid %4d  U %2d  %6.1f KB/wg  %6.2f us/launch  (minus 1.4 us boundary: %5.1f B/clk/CU at %.1f GHz)\n", tag, grid, U,
         per_wg / 1024.0, us, per_wg / ((us - 1.4) * 1e-6) / (g_clk_ghz * 1e9), g_clk_ghz);
  CK(hipGraphExecDestroy(ex));
}

int main(int argc, char** argv) {
  CK(hipSetDevice(0));
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  printf("device %s, %d CUs, clock %.2f GHz\n", pr.name, pr.multiProcessorCount, pr.clockRate * 1e-6);
  hipStream_t st, st2;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
  g_hbm_bytes = 3ull << 30;
  CK(hipMalloc((void**)&g_hbm, g_hbm_bytes)); CK(hipMemset(g_hbm, 1, g_hbm_bytes));
  CK(hipMalloc((void**)&g_small, 8 << 20)); CK(hipMemset(g_small, 2, 8 << 20));
  CK(hipMalloc((void**)&g_out, 1 << 20));
  CK(hipDeviceSynchronize());

  // ---- T1
  for (int grid : {64, 256, 768}) {
    const int N = 64;
    hipGraphExec_t ex = capture(st, [&] { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(256), 0, st, (int*)nullptr); });
    printf("T1 boundary: empty kernel grid %4d: %.2f us per launch (graph of %d)\n", grid, time_graph(ex, st, 9) / N, N);
    CK(hipGraphExecDestroy(ex));
  }
  if (argc > 1 && argv[1][0] == 'b') {   // boundary only (A/B of runtime settings)
    const int N = 64;
    hipGraphExec_t ex = capture(st, [&] {
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL((burst_kernel<6, 1, 0>), dim3(192), dim3(512), 0, st, g_small, 0L, 384, g_out);
    });
    printf("T2 burst line L2-shared U 6 grid 192: %.2f us per launch\n", time_graph(ex, st, 9) / N);
    return 0;
  }
  // ---- T2
  for (int grid : {192, 256}) {
    run_burst<6, 0, 0>(st, grid, 0, "frag  L2-shared");
    run_burst<6, 1, 0>(st, grid, 0, "line  L2-shared");
    run_burst<12, 0, 0>(st, grid, 0, "frag  L2-shared");
    run_burst<12, 1, 0>(st, grid, 0, "line  L2-shared");
    run_burst<24, 0, 0>(st, grid, 0, "frag  L2-shared");
    run_burst<24, 1, 0>(st, grid, 0, "line  L2-shared");
    run_burst<6, 0, 0>(st, grid, 1, "frag  HBM-unique");
    run_burst<6, 1, 0>(st, grid, 1, "line  HBM-unique");
    run_burst<12, 0, 0>(st, grid, 1, "frag  HBM-unique");
    run_burst<12, 1, 0>(st, grid, 1, "line  HBM-unique");
    run_burst<12, 1, 1>(st, grid, 1, "line  HBM-unique nt");
    run_burst<24, 0, 0>(st, grid, 1, "frag  HBM-unique");
    run_burst<24, 1, 0>(st, grid, 1, "line  HBM-unique");
    run_burst<24, 1, 1>(st, grid, 1, "line  HBM-unique nt");
  }
  // ---- T3: streamer (768 wgs x 162 KB = 124 MB) alone, GEMM-like chain alone, both
  {
    const long per = 162L * 1024; const int G = 768;
    auto streamer = [&](hipStream_t s, int rot) { hipLaunchKernelGGL((stream_kernel<0>), dim3(G), dim3(256), 0, s, g_hbm + (size_t)(rot % 8) * G * per, per, g_out); };
    auto streamer_nt = [&](hipStream_t s, int rot) { hipLaunchKernelGGL((stream_kernel<1>), dim3(G), dim3(256), 0, s, g_hbm + (size_t)(rot % 8) * G * per, per, g_out); };
    auto chain = [&](hipStream_t s, int n, int rot) {
      for (int i = 0; i < n; ++i)
        hipLaunchKernelGGL((burst_kernel<12, 1, 0>), dim3(192), dim3(512), 0, s, g_hbm + (1ull << 30) + (size_t)((rot * n + i) % 64) * 192 * 96 * 1024, 96L * 1024, 128, g_out + 4096);
    };
    const int L = 12;
    hipGraphExec_t a = capture(st, [&] { for (int l = 0; l < L; ++l) streamer(st, l); });
    { double us = time_graph(a, st, 5) / L; printf("T3 streamer alone        : %.2f us per launch = %.2f TB/s\n", us, G * per / us * 1e-6); }
    hipGraphExec_t ant = capture(st, [&] { for (int l = 0; l < L; ++l) streamer_nt(st, l); });
    { double us = time_graph(ant, st, 5) / L; printf("T3 streamer alone (nt)   : %.2f us per launch = %.2f TB/s\n", us, G * per / us * 1e-6); }
    hipGraphExec_t b = capture(st, [&] { for (int l = 0; l < L; ++l) chain(st, 4, l); });
    printf("T3 4-GEMM chain alone    : %.2f us per layer\n", time_graph(b, st, 5) / L);
    hipGraphExec_t c = capture(st, [&] { for (int l = 0; l < L; ++l) { streamer(st, l); chain(st, 4, l); } });
    printf("T3 serial (one chain)    : %.2f us per layer\n", time_graph(c, st, 5) / L);
    // two branches of ONE graph: fork at the start, join at the end
    hipEvent_t fork, join; CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    hipGraphExec_t d = capture(st, [&] {
      CK(hipEventRecord(fork, st)); CK(hipStreamWaitEvent(st2, fork, 0));
      for (int l = 0; l < L; ++l) streamer(st, l);
      for (int l = 0; l < L; ++l) chain(st2, 4, l);
      CK(hipEventRecord(join, st2)); CK(hipStreamWaitEvent(st, join, 0));
    });
    printf("T3 two graph branches    : %.2f us per layer (streamer branch || GEMM-chain branch)\n", time_graph(d, st, 5) / L);
    // interleaved half-batch schedule inside one graph: branch A = [stream, chain] x L, branch B = [chain, stream] x L (half-size work each)
    auto half_streamer = [&](hipStream_t s, int rot) { hipLaunchKernelGGL((stream_kernel<0>), dim3(G / 2), dim3(256), 0, s, g_hbm + (size_t)(rot % 16) * (G / 2) * per, per, g_out); };
    hipGraphExec_t e2 = capture(st, [&] {
      CK(hipEventRecord(fork, st)); CK(hipStreamWaitEvent(st2, fork, 0));
      for (int l = 0; l < L; ++l) { half_streamer(st, 2 * l); chain(st, 4, 2 * l); }
      for (int l = 0; l < L; ++l) { chain(st2, 4, 2 * l + 1); half_streamer(st2, 2 * l + 1); }
      CK(hipEventRecord(join, st2)); CK(hipStreamWaitEvent(st, join, 0));
    });
    printf("T3 two half-batch chains : %.2f us per layer (A: stream,gemms | B: gemms,stream; half-size streamers)\n", time_graph(e2, st, 5) / L);
    hipGraphExec_t e1 = capture(st, [&] { for (int l = 0; l < L; ++l) { half_streamer(st, 2 * l); chain(st, 4, 2 * l); chain(st, 4, 2 * l + 1); half_streamer(st, 2 * l + 1); } });
    printf("T3 same work, one chain  : %.2f us per layer\n", time_graph(e1, st, 5) / L);
    // two plain streams, no graph
    hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(t0, st));
    CK(hipEventRecord(fork, st)); CK(hipStreamWaitEvent(st2, fork, 0));
    for (int l = 0; l < L; ++l) { half_streamer(st, 2 * l); chain(st, 4, 2 * l); }
    for (int l = 0; l < L; ++l) { chain(st2, 4, 2 * l + 1); half_streamer(st2, 2 * l + 1); }
    CK(hipEventRecord(join, st2)); CK(hipStreamWaitEvent(st, join, 0));
    CK(hipEventRecord(t1, st)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, t0, t1));
    printf("T3 two plain streams     : %.2f us per layer (host-launched)\n", ms * 1e3 / L);
  }
  return 0;
}
