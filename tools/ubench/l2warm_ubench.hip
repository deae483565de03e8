// Does a weight tile requested by launch A still sit in the XCD-local L2 (or the Infinity Cache) when the dependent launch B asks
// for it?  (development aid, not part of libivg).  Build: make -C tools/ubench ; run on the GPU box.
//   consume : grid workgroups, each pulls its own `kb` KiB (all 16-byte loads of a lane issued up front, whole lines) -- the
//             weight stream of a decode GEMM.  Slots of a 3 GiB buffer are cycled so that an un-warmed consume reads cold HBM.
//   warm    : the same bytes requested one launch earlier by a workgroup with the SAME block index (-> same XCD, observed
//             block b -> XCD b % 8), or by block b + 1 (-> another XCD: whatever still helps then is the Infinity Cache)
//   stream  : a 124 MB non-temporal read (decode-attention-like) between warm and consume
// Prints us per consume launch for every arm (a consume arm's cost = chain - chain without it).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// U x 16 B per lane, 512 threads: U * 8 KiB per workgroup.  shift: block b reads the region of block (b + shift) % grid
template <int U, int NT>
__global__ __launch_bounds__(512) void pull_kernel(const char* __restrict__ src, long wg_stride, int shift, float* __restrict__ out) {
  __shared__ float red[8];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int blk = (int)((blockIdx.x + shift) % gridDim.x);
  const char* base = src + (long)blk * wg_stride;
  u32x4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const u32x4* p = (const u32x4*)(base + ((long)(wave * U + u) * 64 + lane) * 16);
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

__global__ __launch_bounds__(256) void stream_nt_kernel(const char* __restrict__ src, long bytes_per_wg, float* __restrict__ out) {
  const char* base = src + (long)blockIdx.x * bytes_per_wg;
  unsigned acc = 0;
  for (long o = (long)threadIdx.x * 16; o < bytes_per_wg; o += 256L * 16 * 8) {
    u32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long oo = o + (long)u * 256 * 16;
      v[u] = __builtin_nontemporal_load((const u32x4*)(base + (oo < bytes_per_wg ? oo : 0)));
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc ^= v[u][0] ^ v[u][3];
  }
  if (acc == 0x12345678u) out[blockIdx.x] = 1.f;
}

static char* g_hbm; static size_t g_bytes; static float* g_out;
static hipStream_t st;
static hipEvent_t e0, e1;

template <typename F>
static double time_chain(F&& body, int n) {   // us per iteration of body(i), eager launches on one stream
  for (int i = 0; i < 4; ++i) body(i);
  CK(hipStreamSynchronize(st));
  std::vector<float> t;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < n; ++i) body(rep * n + i + 4);
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms * 1e3f / n);
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

template <int U>
static void run(int grid) {
  const long per = 8L * U * 1024, span = (long)grid * per;
  const long slots = (long)((g_bytes - (2ull << 30)) / span);          // the last 2 GiB feed the streamer
  char* kv = g_hbm + (g_bytes - (2ull << 30));
  const long kv_per = 162L * 1024; const int G = 768;
  auto slot = [&](int i) { return g_hbm + (size_t)(i % slots) * span; };
  auto consume = [&](int i, int nt) {
    if (nt) hipLaunchKernelGGL((pull_kernel<U, 1>), dim3(grid), dim3(512), 0, st, slot(i), per, 0, g_out);
    else hipLaunchKernelGGL((pull_kernel<U, 0>), dim3(grid), dim3(512), 0, st, slot(i), per, 0, g_out);
  };
  auto warm = [&](int i, int shift) { hipLaunchKernelGGL((pull_kernel<U, 0>), dim3(grid), dim3(512), 0, st, slot(i), per, shift, g_out + 2048); };
  auto stream = [&](int i) { hipLaunchKernelGGL(stream_nt_kernel, dim3(G), dim3(256), 0, st, kv + (size_t)(i % 16) * G * kv_per, kv_per, g_out + 4096); };
  const int n = 64;
  const double cold = time_chain([&](int i) { consume(i, 0); }, n);
  const double cold_nt = time_chain([&](int i) { consume(i, 1); }, n);
  const double w_only = time_chain([&](int i) { warm(i, 0); }, n);
  const double w_same = time_chain([&](int i) { warm(i, 0); consume(i, 0); }, n);
  const double w_same_nt = time_chain([&](int i) { warm(i, 0); consume(i, 1); }, n);
  const double w_other = time_chain([&](int i) { warm(i, 1); consume(i, 0); }, n);
  const double s_only = time_chain([&](int i) { stream(i); }, n);
  const double w_s = time_chain([&](int i) { warm(i, 0); stream(i); }, n);
  const double w_s_c = time_chain([&](int i) { warm(i, 0); stream(i); consume(i, 0); }, n);
  const double s_c = time_chain([&](int i) { stream(i); consume(i, 0); }, n);
  // re-read of a working set that fits the Infinity Cache: 24 slots cycled (24 x span bytes), consumed every 24 launches
  const long ring = std::max(2L, std::min(slots, (200L << 20) / span));
  const double ring_t = time_chain([&](int i) { hipLaunchKernelGGL((pull_kernel<U, 0>), dim3(grid), dim3(512), 0, st, g_hbm + (size_t)(i % ring) * span, per, 0, g_out); }, n);
  const double ring_nt = time_chain([&](int i) { hipLaunchKernelGGL((pull_kernel<U, 1>), dim3(grid), dim3(512), 0, st, g_hbm + (size_t)(i % ring) * span, per, 0, g_out); }, n);
  printf("grid %3d x %3ld KiB (%5.1f MB per launch): cold %.2f (nt %.2f) | warm launch alone %.2f | consume after same-XCD warm %.2f (nt consume %.2f) | "
         "after other-XCD warm %.2f | stream alone %.2f, consume after stream %.2f, after warm+stream %.2f | %ld-slot ring (%.0f MB) re-read %.2f (nt %.2f)\n",
         grid, per >> 10, span / 1e6, cold, cold_nt, w_only, w_same - w_only, w_same_nt - w_only, w_other - w_only, s_only, s_c - s_only, w_s_c - w_s, ring,
         ring * span / 1e6, ring_t, ring_nt);
}

int main() {
  CK(hipSetDevice(0));
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  g_bytes = 5ull << 30;
  CK(hipMalloc((void**)&g_hbm, g_bytes)); CK(hipMemset(g_hbm, 1, g_bytes));
  CK(hipMalloc((void**)&g_out, 1 << 20));
  CK(hipDeviceSynchronize());
  run<3>(192);    //  24 KiB per workgroup:  4.7 MB (down / q,k,v sized)
  run<6>(192);    //  48 KiB:                9.4 MB (gate/up)
  run<2>(256);    //  16 KiB on every CU
  run<6>(256);
  run<12>(256);   //  96 KiB: 25 MB (lm_head)
  return 0;
}
