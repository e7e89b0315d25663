// Development probe, not part of the library: where does a workgroup of the 8-wave large-tile GEMM (csrc/gemm_w8.hip) spend its
// life?  Same stamps as gemm_timing.hip (entry / first tile in LDS / k-step 4 / end of the k loop / epilogue phases / exit).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DEA_GEMM_TIMING -Iespresso_amd/csrc -Iinclude tools/probes/gemm_w8_timing.hip -o tools/probes/gemm_w8_timing
#include "../../espresso_amd/csrc/gemm_w8.hip"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static void* dalloc(size_t bytes, int fill) {
  void* p;
  CK(hipMalloc(&p, bytes));
  CK(hipMemset(p, fill, bytes));
  return p;
}

struct Case { const char* name; int N, K; int epi; int cfg; int pad_a, pad_b, pad_c; };  // epi: 0 plain, 1 bias, 2 W1, 3 W2, 4 W2 dgrad

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 6240;
  const int MAXN = 5056 + 64, MAXK = 2560 + 64;
  void* A = dalloc((size_t)M * MAXK * 2, 0x3c);
  void* W = dalloc((size_t)MAXN * MAXK * 2, 0x3c);
  void* C = dalloc((size_t)M * MAXN * 2, 0);
  void* C2 = dalloc((size_t)M * MAXN * 2, 0);
  void* R = dalloc((size_t)M * MAXN * 2, 0x3c);
  float* bias = (float*)dalloc((size_t)MAXN * 4, 0);
  const size_t max_wg = 1 << 16;
  unsigned long long* tbuf = (unsigned long long*)dalloc(max_wg * 8 * 8, 0);
  std::vector<unsigned long long> host(max_wg * 8);
  unsigned long long* sbuf = (unsigned long long*)dalloc(4096 * 10 * 8, 0);
  std::vector<unsigned long long> hseg(4096 * 10);
  const Case cases[] = {
      {"W1 fwd 2048x512 bias+silu+drop+2out  256sq", 2048, 512, 2, 2},
      {"W1 shape plain 2048x512              256sq", 2048, 512, 0, 2},
      {"W1 shape plain 2048x512            256x128", 2048, 512, 0, 4},
      {"W2 dgrad 2048x512 drop*silu'(aux)    256sq", 2048, 512, 4, 2},
      {"qkv 1536x512 bias                    256sq", 1536, 512, 1, 2},
      {"pw1 1024x512 plain                   256sq", 1024, 512, 0, 2},
      {"pw1 1024x512 plain                 256x128", 1024, 512, 0, 4},
      {"W2 fwd 512x2048 bias+drop+resid    128sq/4", 512, 2048, 3, 3},
      {"W1 dgrad 512x2048 plain            128sq/4", 512, 2048, 0, 3},
      {"out_proj 512x512 plain             128sq/4", 512, 512, 0, 3},
  };
  const int only = argc > 2 ? atoi(argv[2]) : -1;  // case index (PMC runs: one case per process)
  int ci = -1;
  for (const Case& c : cases) {
    if (++ci != only && only >= 0) continue;
    EaGemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.B = W; p.C = C;
    p.M = M; p.N = c.N; p.K = c.K; p.batch = 1; p.zdiv = 1; p.splitk = 1; p.kchunk = c.K;
    p.lda = c.K + c.pad_a; p.ldb = c.K + c.pad_b; p.ldc = c.N + c.pad_c;
    p.alpha = 1.f; p.out_scale = 1.f;
    if (c.epi >= 1) p.bias = bias;
    if (c.epi == 2) { p.act = 2; p.C2 = C2; p.ldc2 = c.N; p.drop_seed = 7; p.drop_thr = 429496729u; p.drop_scale = 1.f / 0.9f; }
    if (c.epi == 4) { p.act = 2; p.aux = R; p.ldaux = c.N; p.drop_seed = 7; p.drop_thr = 429496729u; p.drop_scale = 1.f / 0.9f; }
    if (c.epi == 3) { p.resid = R; p.ldr = c.N; p.out_scale = 0.5f; p.drop_seed = 9; p.drop_thr = 429496729u; p.drop_scale = 1.f / 0.9f; }
    ea_set_gemm_w8(c.cfg);
    unsigned long long* null_ptr = nullptr;
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_ea_timing), &null_ptr, sizeof(null_ptr)));
    int cfg = 0;
    for (int i = 0; i < 5; ++i) if (!ea_gemm_w8_try(p, 0, 0, &cfg)) { printf("launch refused\n"); return 1; }
    CK(hipDeviceSynchronize());
    hipEvent_t e0, ev1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&ev1));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 20; ++i) ea_gemm_w8_try(p, 0, 0, &cfg);
    CK(hipEventRecord(ev1, 0));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, ev1));
    CK(hipMemset(tbuf, 0, max_wg * 64));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_ea_timing), &tbuf, sizeof(tbuf)));
    CK(hipMemset(sbuf, 0, 4096 * 80));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_ea_seg), &sbuf, sizeof(sbuf)));
    for (int i = 0; i < 3; ++i) ea_gemm_w8_try(p, 0, 0, &cfg);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(host.data(), tbuf, max_wg * 64, hipMemcpyDeviceToHost));
    size_t nwg = 0;
    while (nwg < max_wg && host[nwg * 8 + 3]) ++nwg;
    unsigned long long t_min = ~0ull, t_max = 0;
    for (size_t w = 0; w < nwg; ++w) { t_min = std::min(t_min, host[w * 8]); t_max = std::max(t_max, host[w * 8 + 3]); }
    double life = 0, ph1 = 0, ph2a = 0, ph2b = 0, e1 = 0, e2 = 0, e3 = 0;
    std::vector<double> starts, lifes;
    std::map<unsigned long long, int> per_cu;
    for (size_t w = 0; w < nwg; ++w) {
      const unsigned long long* t = &host[w * 8];
      life += (t[3] - t[0]); ph1 += (t[1] - t[0]); ph2a += (t[7] - t[1]); ph2b += (t[2] - t[7]);
      if (t[5]) { e1 += (t[5] - t[2]); e2 += (t[6] - t[5]); e3 += (t[3] - t[6]); } else { e3 += (t[3] - t[2]); }  // (register epilogue: one figure)
      starts.push_back((t[0] - t_min) * 0.01);
      lifes.push_back((t[3] - t[0]) * 0.01);
      const unsigned hw = (unsigned)t[4], xcc = (unsigned)(t[4] >> 32) & 15;
      per_cu[((unsigned long long)xcc << 16) | (hw & 0xff00)]++;
    }
    std::sort(starts.begin(), starts.end());
    std::sort(lifes.begin(), lifes.end());
    int cu_max = 0;
    for (auto& kv : per_cu) cu_max = std::max(cu_max, kv.second);
    const double span = (t_max - t_min) * 0.01, n = (double)nwg;
    printf("%-46s %5.1f us/launch | span %5.1f us, %4zu WGs on %3zu CUs (max %d per CU) | life mean %5.2f p50 %5.2f p95 %5.2f us = first tile %5.2f + "
           "k-steps 0-3 %5.2f + k-steps 4.. %5.2f + epilogue (to barrier %4.2f, first slab -> LDS %4.2f, rest %5.2f) | starts p50 %5.2f max %5.2f us\n",
           c.name, ms * 1e3 / 20, span, nwg, per_cu.size(), cu_max, life * 0.01 / n, lifes[nwg / 2], lifes[nwg * 95 / 100], ph1 * 0.01 / n,
           ph2a * 0.01 / n, ph2b * 0.01 / n, e1 * 0.01 / n, e2 * 0.01 / n, e3 * 0.01 / n, starts[nwg / 2], starts[nwg - 1]);
    CK(hipMemcpy(hseg.data(), sbuf, 4096 * 80, hipMemcpyDeviceToHost));
    if (hseg[0] | hseg[1] | hseg[2]) {
      for (int h = 0; h < 2; ++h) {
        double sg[5] = {0, 0, 0, 0, 0};
        for (size_t w = 0; w < nwg && w < 4096; ++w) for (int k = 0; k < 5; ++k) sg[k] += hseg[(w * 2 + h) * 5 + k];
        printf("      half %d, shader cycles per workgroup: read fragments (issue) %7.0f | request k-tile %7.0f | wait lds + loads %7.0f | MFMAs (issue) %7.0f | barrier %7.0f  (k-tiles %d)\n", h,
               sg[0] / n, sg[1] / n, sg[2] / n, sg[3] / n, sg[4] / n, c.K / 64);
      }
    }
  }
  return 0;
}
