// Decode-layer GEMM chain with per-phase stamps (development aid, not part of libivg).  Build: make -C tools/ubench ; run on the GPU box.
// Compiles the product's dgemm.hip into this binary and runs the five launches of a decode layer of the small / medium
// transformer (q/k/v, an HBM streamer standing in for the decode attention, o-proj, gate/up, down) as a chain of `layers` layers,
// eagerly on one stream, exactly as transformer.cpp issues them:
//   * us per layer and per GEMM class (HIP events around the whole chain; per-class by leaving one class out);
//   * with the kernel's development stamps on (SkinnyArgs::dbg): where a launch's time goes -- entry -> DMA issued -> first line
//     group landed -> last group landed -> MFMAs done -> combine -> stores issued, per wave, min / median / max over the
//     workgroups, plus the spread of the workgroups' start and end times on the 100 MHz wall clock.
// Usage: dgemm_phase [small|medium] [M]      (IVG_DG3 / IVG_DG3_WARM / IVG_DECODE_LDS_KB act as in the product: csrc/switches.h)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <chrono>
#include <thread>

#include "../../ivideogpt_amd/csrc/switches.cpp"
#include "../../ivideogpt_amd/csrc/dgemm.hip"
#include "../../ivideogpt_amd/csrc/dgemm3.hip"

#define CKH(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

using namespace ivg;

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void stream_nt_kernel(const char* __restrict__ src, long bytes_per_wg, float* __restrict__ out) {
  const char* base = src + (long)blockIdx.x * bytes_per_wg;
  unsigned acc = 0;
  for (long o = (long)threadIdx.x * 16; o < bytes_per_wg; o += 256L * 16 * 8) {
    u32x4_t v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long oo = o + (long)u * 256 * 16;
      v[u] = __builtin_nontemporal_load((const u32x4_t*)(base + (oo < bytes_per_wg ? oo : 0)));
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc ^= v[u][0] ^ v[u][3];
  }
  if (acc == 0x12345678u) out[blockIdx.x] = 1.f;
}

__global__ void fill_bf16(bf16_t* p, long n, unsigned seed, float scale) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (bf16_t)(((int)(h & 0xffff) - 32768) / 32768.0f * scale);
  }
}

struct Layer { bf16_t *wqkv, *wo, *wgu, *wdown; };

int main(int argc, char** argv) {
  const bool medium = argc > 1 && !strcmp(argv[1], "medium");
  const int M = argc > 2 ? atoi(argv[2]) : 64;
  const int H = medium ? 1024 : 768, I = 4 * H, layers = medium ? 24 : 12;
  CKH(hipSetDevice(0));
  hipStream_t st;
  CKH(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  auto alloc = [&](long n, unsigned seed, float scale) {
    bf16_t* p; CKH(hipMalloc((void**)&p, n * 2));
    hipLaunchKernelGGL(fill_bf16, dim3(1024), dim3(256), 0, st, p, n, seed, scale);
    return p;
  };
  std::vector<Layer> L(layers);
  for (int l = 0; l < layers; ++l)
    L[l] = Layer{alloc(3L * H * H, 11 + l, 0.03f), alloc((long)H * H, 211 + l, 0.03f), alloc(2L * I * H, 411 + l, 0.03f), alloc((long)H * I, 611 + l, 0.02f)};
  bf16_t* x = alloc((long)128 * H, 1, 1.0f);
  bf16_t* qkv = alloc((long)128 * 3 * H, 2, 1.0f);
  bf16_t* attn = alloc((long)128 * H, 3, 1.0f);
  bf16_t* act = alloc((long)128 * I, 4, 1.0f);
  const int heads = H / 64;
  const long kv_per_wg = 2L * 632 * 64 * 2;           // K and V rows of one (trajectory, head) at the mean cache length of config 2
  const int G = M * heads;
  const int kv_slots = 16;   // (x 2 half-size slots in the chains2 mode)
  char* kvbuf; CKH(hipMalloc((void**)&kvbuf, (size_t)kv_slots * G * kv_per_wg)); CKH(hipMemsetAsync(kvbuf, 1, (size_t)kv_slots * G * kv_per_wg, st));
  float* sink; CKH(hipMalloc((void**)&sink, 1 << 20));
  const size_t dbg_per = (size_t)1024 * 16 * 16;       // workgroups x waves x stamps
  long long* dbg; CKH(hipMalloc((void**)&dbg, 4 * dbg_per * 8)); CKH(hipMemsetAsync(dbg, 0, 4 * dbg_per * 8, st));
  CKH(hipStreamSynchronize(st));

  const int gen = getenv("GEN") ? atoi(getenv("GEN")) : 3;
  const bool warm = !(getenv("WARM") && getenv("WARM")[0] == '0');
  auto gemm = [&](int which, int l, long long* d) {
    const Layer& w = L[l];
    SkinnyArgs s;
    s.M = M; s.eps = 1e-6f; s.dbg = d;
    if (which == 0) { s.X = x; s.W = w.wqkv; s.Y = qkv; s.N = 3 * H; s.K = H; s.ldx = H; s.ldw = H; s.ldy = 3 * H; s.flags = SK_NORM; }
    if (which == 1) { s.X = attn; s.W = w.wo; s.Y = x; s.N = H; s.K = H; s.ldx = H; s.ldw = H; s.ldy = H; s.flags = IG_RESIDUAL; }
    if (which == 2) { s.X = x; s.W = w.wgu; s.Y = act; s.N = 2 * I; s.K = H; s.ldx = H; s.ldw = H; s.ldy = I; s.flags = IG_GLU | SK_NORM; }
    if (which == 3) { s.X = act; s.W = w.wdown; s.Y = x; s.N = H; s.K = I; s.ldx = I; s.ldw = I; s.ldy = H; s.flags = IG_RESIDUAL; }
    if (warm) {   // the next launch of the chain: o-proj, gate/up, down, q/k/v of the next layer
      SkinnyArgs n;
      n.M = M;
      const Layer& wn = which == 3 ? L[(l + 1) % layers] : w;
      if (which == 0) { n.W = wn.wo; n.N = H; n.K = H; n.ldw = H; n.flags = IG_RESIDUAL; n.ldy = H; n.Y = x; n.X = attn; n.ldx = H; }
      if (which == 1) { n.W = wn.wgu; n.N = 2 * I; n.K = H; n.ldw = H; n.flags = IG_GLU | SK_NORM; n.ldy = I; n.Y = act; n.X = x; n.ldx = H; }
      if (which == 2) { n.W = wn.wdown; n.N = H; n.K = I; n.ldw = I; n.flags = IG_RESIDUAL; n.ldy = H; n.Y = x; n.X = act; n.ldx = I; }
      if (which == 3) { n.W = wn.wqkv; n.N = 3 * H; n.K = H; n.ldw = H; n.flags = SK_NORM; n.ldy = 3 * H; n.Y = qkv; n.X = x; n.ldx = H; }
      int rows = dgemm3_w_rows_per_block(n, BF16);
      if (rows <= 0) rows = dgemm_w_rows_per_block(n, BF16);
      if (rows > 0) { s.next_W = n.W; s.next_tile_bytes = (long)rows * n.K * 2; s.next_tiles = n.N / rows; }
    }
    const char* gm = getenv("GENMASK");   // per-GEMM generation, e.g. GENMASK=3323: gate/up on the second-generation kernel
    const int g_this = gm && strlen(gm) == 4 ? gm[which] - '0' : gen;
    int rc = g_this == 3 ? launch_dgemm3(s, BF16, st) : -1;   // (-1: not covered / on the skip list -> second generation, as launch_skinny does)
    if (rc == -1) rc = launch_dgemm(s, BF16, st);
    if (rc != 0) { fprintf(stderr, "launch (generation %d, GEMM %d) -> %d\n", g_this, which, rc); exit(1); }
  };
  int rot = 0;
  auto layer = [&](int l, unsigned mask, bool stream, long long* d) {
    if (mask & 1) gemm(0, l, d ? d + 0 * dbg_per : nullptr);
    if (stream) hipLaunchKernelGGL(stream_nt_kernel, dim3(G), dim3(256), 0, st, kvbuf + (size_t)(rot++ % kv_slots) * G * kv_per_wg, kv_per_wg, sink);
    if (mask & 2) gemm(1, l, d ? d + 1 * dbg_per : nullptr);
    if (mask & 4) gemm(2, l, d ? d + 2 * dbg_per : nullptr);
    if (mask & 8) gemm(3, l, d ? d + 3 * dbg_per : nullptr);
  };
  if (argc > 3 && !strcmp(argv[3], "chains2")) {
    // Two half-batch chains on CU-MASKED streams (hipExtStreamCreateWithCUMask: each chain owns half of the CUs), launched eagerly by
    // two host threads: does the HBM-bound attention of one chain run beside the ingest-bound GEMMs of the other?
    const int Mh = M / 2, Gh = Mh * heads;
    const int layout = argc > 4 ? atoi(argv[4]) : 0;   // 0: CUs [0,128) | [128,256)   1: even | odd CUs   2: no masks (plain streams)
    hipStream_t cs[2];
    for (int c = 0; c < 2; ++c) {
      uint32_t mask[8];
      for (int w = 0; w < 8; ++w) mask[w] = layout == 0 ? ((w < 4) == (c == 0) ? 0xffffffffu : 0u) : (c == 0 ? 0x55555555u : 0xaaaaaaaau);
      if (layout == 2) CKH(hipStreamCreateWithFlags(&cs[c], hipStreamNonBlocking));
      else CKH(hipExtStreamCreateWithCUMask(&cs[c], 8, mask));
    }
    bf16_t* cx[2] = {alloc((long)64 * H, 21, 1.0f), alloc((long)64 * H, 22, 1.0f)};
    bf16_t* cqkv[2] = {alloc((long)64 * 3 * H, 23, 1.0f), alloc((long)64 * 3 * H, 24, 1.0f)};
    bf16_t* cattn[2] = {alloc((long)64 * H, 25, 1.0f), alloc((long)64 * H, 26, 1.0f)};
    bf16_t* cact[2] = {alloc((long)64 * I, 27, 1.0f), alloc((long)64 * I, 28, 1.0f)};
    CKH(hipStreamSynchronize(st));
    auto cgemm = [&](int c, int which, int l) {
      const Layer& w = L[l];
      SkinnyArgs a;
      a.M = Mh; a.eps = 1e-6f;
      if (which == 0) { a.X = cx[c]; a.W = w.wqkv; a.Y = cqkv[c]; a.N = 3 * H; a.K = H; a.ldx = H; a.ldw = H; a.ldy = 3 * H; a.flags = SK_NORM; }
      if (which == 1) { a.X = cattn[c]; a.W = w.wo; a.Y = cx[c]; a.N = H; a.K = H; a.ldx = H; a.ldw = H; a.ldy = H; a.flags = IG_RESIDUAL; }
      if (which == 2) { a.X = cx[c]; a.W = w.wgu; a.Y = cact[c]; a.N = 2 * I; a.K = H; a.ldx = H; a.ldw = H; a.ldy = I; a.flags = IG_GLU | SK_NORM; }
      if (which == 3) { a.X = cact[c]; a.W = w.wdown; a.Y = cx[c]; a.N = H; a.K = I; a.ldx = I; a.ldw = I; a.ldy = H; a.flags = IG_RESIDUAL; }
      int rc = gen == 3 ? launch_dgemm3(a, BF16, cs[c]) : -1;
      if (rc == -1) rc = launch_dgemm(a, BF16, cs[c]);
      if (rc != 0) { fprintf(stderr, "chain launch -> %d\n", rc); exit(1); }
    };
    auto chain_layer = [&](int c, int l, int r, bool gemms, bool stream) {
      if (gemms) cgemm(c, 0, l);
      if (stream) hipLaunchKernelGGL(stream_nt_kernel, dim3(Gh), dim3(256), 0, cs[c], kvbuf + (size_t)((2 * r + c) % (2 * kv_slots)) * Gh * kv_per_wg, kv_per_wg, sink + 1024 * c);
      if (gemms) { cgemm(c, 1, l); cgemm(c, 2, l); cgemm(c, 3, l); }
    };
    auto run = [&](bool two, bool gemms, bool stream, int steps) {
      auto body = [&](int c) { CKH(hipSetDevice(0)); int r = 0; for (int s = 0; s < steps; ++s) for (int l = 0; l < layers; ++l) chain_layer(c, l, r++, gemms, stream); };
      body(0); if (two) body(1);                               // warm-up (attributes)
      CKH(hipDeviceSynchronize());
      std::vector<double> t;
      for (int rep = 0; rep < 3; ++rep) {
        const auto t0 = std::chrono::steady_clock::now();
        std::thread th0(body, 0);
        if (two) { std::thread th1(body, 1); th1.join(); }
        th0.join();
        CKH(hipDeviceSynchronize());
        t.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (steps * layers));
      }
      std::sort(t.begin(), t.end());
      return t[1];
    };
    const int steps = 16;
    printf("chains2 (generation %d, %d rows per chain, CU mask layout %d), us per layer (host wall clock, %d steps x %d layers):\n", gen, Mh, layout, steps, layers);
    printf("  one chain alone : streamer %.2f | GEMMs %.2f | both %.2f\n", run(false, false, true, steps), run(false, true, false, steps), run(false, true, true, steps));
    printf("  two chains      : streamers %.2f | GEMMs %.2f | both %.2f   (full batch on one unmasked chain: see the default mode)\n",
           run(true, false, true, steps), run(true, true, false, steps), run(true, true, true, steps));
    return 0;
  }
  hipEvent_t e0, e1; CKH(hipEventCreate(&e0)); CKH(hipEventCreate(&e1));
  auto time_chain = [&](unsigned mask, bool stream, int steps) {
    for (int l = 0; l < layers; ++l) layer(l, mask, stream, nullptr);   // warm-up (attributes, code upload)
    CKH(hipStreamSynchronize(st));
    std::vector<float> t;
    for (int rep = 0; rep < 5; ++rep) {
      CKH(hipEventRecord(e0, st));
      for (int s = 0; s < steps; ++s) for (int l = 0; l < layers; ++l) layer(l, mask, stream, nullptr);
      CKH(hipEventRecord(e1, st)); CKH(hipStreamSynchronize(st));
      float ms; CKH(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms * 1e3f / (steps * layers));
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
  };
  const char* names[4] = {"q/k/v", "o-proj", "gate/up", "down"};
  const float all_s = time_chain(15, true, 8), all = time_chain(15, false, 8), str = time_chain(0, true, 8);
  printf("GENMASK=%s generation %d%s, IVG_DECODE_LDS_KB=%s: ", getenv("GENMASK") ? getenv("GENMASK") : "-", gen, gen == 3 ? (warm ? " + L2 warm-up" : ", no warm-up") : "", getenv("IVG_DECODE_LDS_KB") ? getenv("IVG_DECODE_LDS_KB") : "160");
  printf("%s transformer, M = %d: layer chain %.2f us (4 GEMMs + streamer), streamer alone %.2f, 4 GEMMs alone %.2f us per layer\n",
         medium ? "medium" : "small", M, all_s, str, all);
  for (int k = 0; k < 4; ++k) {
    const float without = time_chain(15 & ~(1u << k), true, 8);
    printf("  %-8s %.2f us per launch inside the chain (chain without it %.2f)\n", names[k], all_s - without, without);
  }
  // ---- phase stamps of the last layer of a chain
  for (int s = 0; s < 3; ++s) for (int l = 0; l < layers; ++l) layer(l, 15, true, l == layers - 1 && s == 2 ? dbg : nullptr);
  CKH(hipStreamSynchronize(st));
  std::vector<long long> h(4 * dbg_per);
  CKH(hipMemcpy(h.data(), dbg, 4 * dbg_per * 8, hipMemcpyDeviceToHost));
  const char* ph[8] = {"entry", "dma issued", "first line", "last line", "mfma done", gen == 3 ? "parked" : "barrier 1", gen == 3 ? "barrier" : "barrier 2", "stores issued"};
  for (int k = 0; k < 4; ++k) {
    const long long* d = h.data() + k * dbg_per;
    std::vector<int> wgs;
    int maxw = 0;
    for (int wg = 0; wg < 1024; ++wg) if (d[((size_t)wg * 16) * 16 + 8] != 0) { wgs.push_back(wg); for (int w = 0; w < 16; ++w) if (d[((size_t)wg * 16 + w) * 16 + 8] != 0) maxw = std::max(maxw, w + 1); }
    if (wgs.empty()) { printf("%s: no stamps\n", names[k]); continue; }
    long long w0 = 1LL << 62, w1 = 0, s_last = 0, e_first = 1LL << 62;
    for (int wg : wgs) for (int w = 0; w < maxw; ++w) {
      const long long* r = d + ((size_t)wg * 16 + w) * 16;
      w0 = std::min(w0, r[8]); w1 = std::max(w1, r[9]); s_last = std::max(s_last, r[8]); e_first = std::min(e_first, r[9]);
    }
    printf("%s: %zu workgroups x %d waves; launch window %.2f us (first start -> last end), starts spread %.2f us, ends spread %.2f us\n", names[k], wgs.size(),
           maxw, (w1 - w0) * 0.01, (s_last - w0) * 0.01, (w1 - e_first) * 0.01);
    {   // does block b run on XCD b % 8 (what the L2 warm-up assumes)?  dbg[10] = HW_REG_XCC_ID (third generation only)
      int match = 0, seen = 0;
      for (int wg : wgs) { const long long* r = d + ((size_t)wg * 16) * 16; if (r[8]) { ++seen; match += (int)(r[10] == (wg & 7)); } }
      printf("    workgroups whose XCC_ID equals block %% 8: %d of %d\n", match, seen);
    }
    printf("    cycles from the wave's entry (min / median / max over workgroups and waves):\n");
    for (int i = 1; i < 8; ++i) {
      std::vector<long long> v;
      for (int wg : wgs) for (int w = 0; w < maxw; ++w) { const long long* r = d + ((size_t)wg * 16 + w) * 16; if (r[i] && r[0]) v.push_back(r[i] - r[0]); }
      if (v.empty()) continue;
      std::sort(v.begin(), v.end());
      printf("      %-14s %7lld %7lld %7lld\n", ph[i], v.front(), v[v.size() / 2], v.back());
    }
  }
  return 0;
}
