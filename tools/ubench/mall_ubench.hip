// Can the NEXT layer's K / V rows be pulled into the memory-side Infinity Cache while the decode GEMMs of a layer run (HBM idle),
// so that the decode attention then streams from the cache instead of HBM?  (development aid, round 6; not part of libivg).
// Build: make -C tools/ubench ; run on the GPU box.
//   M1  stream (nt) over 12 rotating 124 MB regions          -> cold HBM, us per launch
//   M2  stream (nt) over ONE region again and again          -> do non-temporal reads hit / allocate in the Infinity Cache?
//   M3  stream (default policy) over ONE region
//   M4  prefetch (default policy, P workgroups) of region r, then stream (nt) of region r, rotating -> the consumer's time on hits
//   M5  the rollout's layer loop: [4 GEMM-like bursts, stream(region l)] x 12 on one stream, eager launches, with and without a side
//       stream that prefetches region l + 1 once stream(l) is done (event) -> us per layer
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

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

// prefetch: gridDim.x workgroups of 256 threads walk `bytes` contiguous bytes, one 4-byte request per 128-byte line and lane (the
// line lands in the caches; 1/32 of the register traffic of a real read), 8 requests in flight per lane
__global__ __launch_bounds__(256) void prefetch_kernel(const char* __restrict__ src, long bytes, float* __restrict__ out) {
  const long lines = bytes >> 7;
  unsigned acc = 0;
  for (long l0 = (long)blockIdx.x * 256 + threadIdx.x; l0 < lines; l0 += (long)gridDim.x * 256 * 8) {
    unsigned v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long l = l0 + (long)u * gridDim.x * 256;
      v[u] = *(const unsigned*)(src + ((l < lines ? l : 0) << 7));
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc ^= v[u];
  }
  if (acc == 0x12345678u) out[blockIdx.x] = 1.f;
}

template <int U, int NT>
__global__ __launch_bounds__(512) void burst_kernel(const char* __restrict__ src, long wg_stride, float* __restrict__ out) {
  __shared__ float red[8];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const char* base = src + (long)blockIdx.x * wg_stride;
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

static hipStream_t st, st2;
static hipEvent_t e0, e1;

template <typename F>
static double time_loop(F&& body, int n, int reps = 5) {   // us per iteration of body(i), eager launches
  for (int i = 0; i < n; ++i) body(i);
  CK(hipDeviceSynchronize());
  std::vector<float> t;
  for (int rep = 0; rep < reps; ++rep) {
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < n; ++i) body(i);
    CK(hipEventRecord(e1, st)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms * 1e3f / n);
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

int main(int argc, char** argv) {
  CK(hipSetDevice(0));
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const long per = 162L * 1024; const int G = 768; const long region = G * per;   // 124 MB: one layer's K + V rows at B = 64, 632 keys
  const int L = 12;
  char* kv; float* out; char* wts;
  CK(hipMalloc((void**)&kv, (size_t)region * L)); CK(hipMemset(kv, 1, (size_t)region * L));
  const long wper = 96L * 1024; const int WG = 192; const long wsz = WG * wper;   // 18.9 MB "weights" per GEMM-like launch (4 per layer: more than the model's)
  CK(hipMalloc((void**)&wts, (size_t)wsz * 4 * L)); CK(hipMemset(wts, 2, (size_t)wsz * 4 * L));
  CK(hipMalloc((void**)&out, 1 << 20));
  CK(hipDeviceSynchronize());
  auto tbs = [&](double us) { return region / us * 1e-6; };

  { double us = time_loop([&](int i) { hipLaunchKernelGGL((stream_kernel<1>), dim3(G), dim3(256), 0, st, kv + (size_t)(i % L) * region, per, out); }, 24);
    printf("M1 stream nt, 12 rotating regions : %6.2f us = %5.2f TB/s\n", us, tbs(us)); }
  { double us = time_loop([&](int i) { hipLaunchKernelGGL((stream_kernel<0>), dim3(G), dim3(256), 0, st, kv + (size_t)(i % L) * region, per, out); }, 24);
    printf("M1 stream default, 12 rotating     : %6.2f us = %5.2f TB/s\n", us, tbs(us)); }
  { double us = time_loop([&](int i) { hipLaunchKernelGGL((stream_kernel<1>), dim3(G), dim3(256), 0, st, kv, per, out); }, 24);
    printf("M2 stream nt, ONE region          : %6.2f us = %5.2f TB/s\n", us, tbs(us)); }
  { double us = time_loop([&](int i) { hipLaunchKernelGGL((stream_kernel<0>), dim3(G), dim3(256), 0, st, kv, per, out); }, 24);
    printf("M3 stream default, ONE region     : %6.2f us = %5.2f TB/s\n", us, tbs(us)); }
  { double us = time_loop([&](int i) { hipLaunchKernelGGL((stream_kernel<0>), dim3(G), dim3(256), 0, st, kv + (size_t)(i % 2) * region, per, out); }, 24);
    printf("M3 stream default, TWO regions    : %6.2f us = %5.2f TB/s  (248 MB of a 256 MB cache)\n", us, tbs(us)); }
  for (int P : {64, 128, 256, 512, 1024}) {
    double pf = time_loop([&](int i) { hipLaunchKernelGGL(prefetch_kernel, dim3(P), dim3(256), 0, st, kv + (size_t)(i % L) * region, region, out); }, 24);
    double both = time_loop([&](int i) {
      hipLaunchKernelGGL(prefetch_kernel, dim3(P), dim3(256), 0, st, kv + (size_t)(i % L) * region, region, out);
      hipLaunchKernelGGL((stream_kernel<1>), dim3(G), dim3(256), 0, st, kv + (size_t)(i % L) * region, per, out); }, 24);
    double bothd = time_loop([&](int i) {
      hipLaunchKernelGGL(prefetch_kernel, dim3(P), dim3(256), 0, st, kv + (size_t)(i % L) * region, region, out);
      hipLaunchKernelGGL((stream_kernel<0>), dim3(G), dim3(256), 0, st, kv + (size_t)(i % L) * region, per, out); }, 24);
    printf("M4 prefetch P=%4d alone %6.2f us (%5.2f TB/s) | + stream nt %6.2f -> consumer %6.2f us = %5.2f TB/s | + stream default -> consumer %6.2f us\n",
           P, pf, tbs(pf), both, both - pf, tbs(both - pf), bothd - pf);
  }
  // ---- M5: the layer loop
  hipEvent_t ev[L];
  for (int l = 0; l < L; ++l) CK(hipEventCreateWithFlags(&ev[l], hipEventDisableTiming));
  auto chain = [&](int l, int nt) {
    for (int k = 0; k < 4; ++k) {
      const char* w = wts + ((size_t)l * 4 + k) * wsz;
      if (nt) hipLaunchKernelGGL((burst_kernel<12, 1>), dim3(WG), dim3(512), 0, st, w, wper, out + 4096);
      else hipLaunchKernelGGL((burst_kernel<12, 0>), dim3(WG), dim3(512), 0, st, w, wper, out + 4096);
    }
  };
  { double us = time_loop([&](int i) { chain(i % L, 1); }, 24); printf("M5 4-burst chain alone (nt weights): %6.2f us per layer\n", us); }
  for (int nt : {1, 0}) {
    double base = time_loop([&](int i) { const int l = i % L; chain(l, nt); hipLaunchKernelGGL((stream_kernel<1>), dim3(G), dim3(256), 0, st, kv + (size_t)l * region, per, out); }, 48);
    printf("M5 serial loop (weights %s)          : %6.2f us per layer\n", nt ? "nt" : "default", base);
    for (int P : {64, 128, 256, 512}) {
      for (int frac : {100, 50}) {
        double us = time_loop([&](int i) {
          const int l = i % L, nl = (l + 1) % L;
          chain(l, nt);
          hipLaunchKernelGGL((stream_kernel<1>), dim3(G), dim3(256), 0, st, kv + (size_t)l * region, per, out);
          CK(hipEventRecord(ev[l], st)); CK(hipStreamWaitEvent(st2, ev[l], 0));
          hipLaunchKernelGGL(prefetch_kernel, dim3(P), dim3(256), 0, st2, kv + (size_t)nl * region, region * frac / 100, out + 8192);
        }, 48);
        printf("M5 + side-stream prefetch P=%4d of %3d %% of the next region: %6.2f us per layer\n", P, frac, us);
      }
    }
    {   // the event traffic alone (an empty prefetch)
      double us = time_loop([&](int i) {
        const int l = i % L;
        chain(l, nt);
        hipLaunchKernelGGL((stream_kernel<1>), dim3(G), dim3(256), 0, st, kv + (size_t)l * region, per, out);
        CK(hipEventRecord(ev[l], st)); CK(hipStreamWaitEvent(st2, ev[l], 0));
        hipLaunchKernelGGL(prefetch_kernel, dim3(1), dim3(256), 0, st2, kv, 0L, out + 8192);
      }, 48);
      printf("M5 + events and an EMPTY side launch: %6.2f us per layer\n", us);
    }
  }
  return 0;
}
