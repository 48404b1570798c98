// gemm_probe -- stand-alone micro-benchmark / race screen for the GEMM kernels of libwjhip.so, through the C ABI only
// (wj_k_gemm, wj_k_gemm_timed, wj_tune).  No Python, no torch: a gpurun call with it costs seconds.
//
//   hipcc --offload-arch=gfx950 -O2 scripts/gemm_probe.hip -o whisperjav_amd/csrc/gemm_probe -Iinclude -ldl
//   (scripts/build_probe.sh also builds whisperjav_amd/csrc/libwjhip_ref.so from the sources of a given commit; when that
//    file sits next to the probe its output is the reference, otherwise the library's own reference variant is)
//   whisperjav_amd/csrc/gemm_probe [enc|dec|all] [reps] > gpurun_out/gemm_probe.jsonl
//
// Every candidate variant is compared BIT FOR BIT with the reference variant of the same shape (all of them accumulate
// k in ascending blocks of 32 into fp32, so equal inputs must give equal bits), `screen` times in a row with fresh
// output buffers (a race shows up as a mismatch that comes and goes), then timed with wj_k_gemm_timed.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <dlfcn.h>

#include "wjhip.h"

// the library under test and (optional) a build of the previous commit as the bit-exact reference
struct Api {
  void* h = nullptr;
  decltype(&wj_init) init;
  decltype(&wj_shutdown) shutdown;
  decltype(&wj_sync) sync;
  decltype(&wj_tune) tune;
  decltype(&wj_k_gemm) k_gemm;
  decltype(&wj_k_gemm_timed) k_gemm_timed;
  decltype(&wj_last_error) last_error;
  wj_ctx* ctx = nullptr;
  bool load(const char* path) {
    h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen(%s): %s\n", path, dlerror()); return false; }
#define SYM(f, name) f = (decltype(f))dlsym(h, name); if (!f) { fprintf(stderr, "missing %s in %s\n", name, path); return false; }
    SYM(init, "wj_init") SYM(shutdown, "wj_shutdown") SYM(sync, "wj_sync") SYM(tune, "wj_tune") SYM(k_gemm, "wj_k_gemm")
    SYM(k_gemm_timed, "wj_k_gemm_timed") SYM(last_error, "wj_last_error")
#undef SYM
    return init(0, &ctx) == 0;
  }
};
static Api L, R;   // under test, reference

#define CK(x)                                                                                   \
  do {                                                                                          \
    hipError_t e_ = (x);                                                                        \
    if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } \
  } while (0)
#define WJ(x)                                                                                   \
  do {                                                                                          \
    int rc_ = (x);                                                                              \
    if (rc_ != 0) { fprintf(stderr, "wj error %d (%s) at %s:%d\n", rc_, L.last_error(), __FILE__, __LINE__); exit(3); } \
  } while (0)

__global__ void fill_f16(_Float16* p, int64_t n, uint32_t seed, float scale) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u ^ seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    p[i] = (_Float16)(((int)(h & 0xffff) - 32768) * (scale / 32768.0f));
  }
}
__global__ void fill_f32(float* p, int64_t n, uint32_t seed, float scale) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u ^ seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = ((int)(h & 0xffff) - 32768) * (scale / 32768.0f);
  }
}
__global__ void count_diff(const uint32_t* a, const uint32_t* b, int64_t nwords, unsigned long long* out) {
  unsigned long long c = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nwords; i += (int64_t)gridDim.x * blockDim.x)
    c += a[i] != b[i];
  if (c) atomicAdd(out, c);
}


// ---- LDS-DMA streaming micro-benchmark: the global -> LDS traffic of a 256 x 256 GEMM tile sweep without the GEMM ----------
// Same tile order as the library's 256-tile kernels.  SEG = contiguous bytes fetched per row and request group:
//   64  : 16 rows x 64 B per wave request, one 32-wide k-stage per step          (the ping-pong kernel's pattern)
//   65  : as 64, but the two 64-byte halves of a 128-byte line are requested back to back (two stages per step)
//   128 : 8 rows x 128 B per wave request, two stages (64 k) per step
//   256 : 4 rows x 256 B per wave request, four stages (128 k) per step
// DEPTH = steps kept in flight (ring slots - 1).  Every step moves (SEG == 64 ? 32 : SEG == 256 ? 128 : 64) KiB per workgroup.
template <int SEG, int DEPTH>
__global__ __launch_bounds__(512) void dma_stream_kernel(const _Float16* __restrict__ A, const _Float16* __restrict__ W, int M,
                                                         int N, int K, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nx = gridDim.x, ntiles = gridDim.x * gridDim.y;
  const int lin = blockIdx.y * nx + blockIdx.x;
  const int q8 = ntiles >> 3, r8 = ntiles & 7, xcd = lin & 7;
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (lin >> 3);
  constexpr int GM = 8;
  const int per_group = GM * nx, group = tile / per_group, first_m = group * GM;
  const int gsz = min(GM, (int)gridDim.y - first_m), in_group = tile - group * per_group;
  const int m0 = (first_m + in_group % gsz) * 256, n0 = (in_group / gsz) * 256;
  constexpr int KSTEP = SEG == 64 ? 32 : (SEG == 256 ? 128 : 64);     // k elements per step
  constexpr int PIECES = KSTEP / 32 * 4;                               // wave requests per step (1 KiB each)
  constexpr int STEP_EL = 2 * 256 * KSTEP;                             // LDS elements per step
  constexpr int LPR = (SEG == 65 ? 64 : SEG) / 16;                     // lanes per row
  constexpr int RPP = 64 / LPR;                                        // rows per request
  const int nsteps = K / KSTEP;
  auto issue = [&](int step, int slot) {
    _Float16* base = lds + (size_t)slot * STEP_EL;
#pragma unroll
    for (int q = 0; q < PIECES / 2; ++q) {
      // piece q of this wave: rows and k offset
      int row, koff;
      if constexpr (SEG == 65) {          // q = half * (PIECES / 4) + p : all first halves, then all second halves
        const int half = q / (PIECES / 4), p = q % (PIECES / 4);
        row = (wave * (PIECES / 4) + p) * RPP + lane / LPR;
        koff = half * 32 + (lane % LPR) * 8;
      } else {
        row = (wave * (PIECES / 2) + q) * RPP + lane / LPR;
        koff = (lane % LPR) * 8;
      }
      const _Float16* ga = A + (int64_t)min(m0 + row, M - 1) * K + (int64_t)step * KSTEP + koff;
      const _Float16* gw = W + (int64_t)min(n0 + row, N - 1) * K + (int64_t)step * KSTEP + koff;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ga,
          (__attribute__((address_space(3))) void*)(base + (wave * (PIECES / 2) + q) * 512), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gw,
          (__attribute__((address_space(3))) void*)(base + 256 * KSTEP + (wave * (PIECES / 2) + q) * 512), 16, 0, 0);
    }
  };
#pragma unroll
  for (int st = 0; st < DEPTH; ++st) issue(st, st);
  int slot = DEPTH;
  for (int kt = 0; kt < nsteps; ++kt) {
    if (kt + DEPTH < nsteps) {
      issue(kt + DEPTH, slot);
      // the oldest step has landed; DEPTH steps stay in flight
      if constexpr (PIECES * DEPTH == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if constexpr (PIECES * DEPTH == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if constexpr (PIECES * DEPTH == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else if constexpr (PIECES * DEPTH == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if constexpr (PIECES * DEPTH == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
      else if constexpr (PIECES * DEPTH == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    slot = slot + 1 == DEPTH + 1 ? 0 : slot + 1;
  }
  if (lds[tid] == (_Float16)12345.0f) sink[0] = 1;   // keep the LDS contents observable
}

template <int SEG, int DEPTH>
static void run_dma(const char* name, const _Float16* A, const _Float16* W, int M, int N, int K, unsigned* sink, int reps) {
  constexpr int KSTEP = SEG == 64 ? 32 : (SEG == 256 ? 128 : 64);
  constexpr size_t smem = (size_t)(DEPTH + 1) * 2 * 256 * KSTEP * 2;
  if (smem > 163840) return;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dma_stream_kernel<SEG, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  dim3 grid(N / 256, M / 256);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  dma_stream_kernel<SEG, DEPTH><<<grid, 512, smem>>>(A, W, M, N, K, sink);
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) dma_stream_kernel<SEG, DEPTH><<<grid, 512, smem>>>(A, W, M, N, K, sink);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double bytes = (double)grid.x * grid.y * (double)K * 512 * 2;   // A + W tile bytes through the DMA
  printf("{\"dma\": \"%s\", \"M\": %d, \"N\": %d, \"K\": %d, \"seg\": %d, \"depth\": %d, \"lds_kib\": %d, \"ms\": %.4f, \"dma_tb_s\": %.2f, "
         "\"tflops_if_hidden\": %.0f}\n", name, M, N, K, SEG, DEPTH, (int)(smem >> 10), ms, bytes / ms * 1e-9,
         2.0 * M * N * (double)K / ms * 1e-9);
  fflush(stdout);
}

struct Shape { const char* name; int M, N, K, gelu, f32; };

int main(int argc, char** argv) {
  const std::string what = argc > 1 ? argv[1] : "all";
  const int reps = argc > 2 ? atoi(argv[2]) : 5;
  const int screen = argc > 3 ? atoi(argv[3]) : 6;
  std::string dir = argv[0];
  dir = dir.find('/') == std::string::npos ? "." : dir.substr(0, dir.rfind('/'));
  if (!L.load((dir + "/libwjhip.so").c_str())) return 4;
  const bool have_ref = R.load((dir + "/libwjhip_ref.so").c_str());
  fprintf(stderr, "reference library: %s\n", have_ref ? "libwjhip_ref.so (previous commit)" : "none -- the library's own reference variant");
  wj_ctx* ctx = L.ctx;
  unsigned long long* d_cnt;
  CK(hipMalloc(&d_cnt, 8));

  auto run_set = [&](const std::vector<Shape>& shapes, int ref_variant, const std::vector<int>& variants,
                     const std::vector<std::pair<const char*, int>>& tune_ab) {
    for (const Shape& sh : shapes) {
      const int64_t na = (int64_t)sh.M * sh.K, nw = (int64_t)sh.N * sh.K, nc = (int64_t)sh.M * sh.N;
      const int64_t cbytes = nc * (sh.f32 ? 4 : 2);
      _Float16 *A, *W;
      float* bias;
      void *Cref, *C;
      CK(hipMalloc(&A, na * 2)); CK(hipMalloc(&W, nw * 2)); CK(hipMalloc(&bias, sh.N * 4));
      CK(hipMalloc(&Cref, cbytes)); CK(hipMalloc(&C, cbytes));
      fill_f16<<<2048, 256>>>(A, na, 0x1234u, 1.0f);
      fill_f16<<<2048, 256>>>(W, nw, 0x9876u, 0.05f);
      fill_f32<<<64, 256>>>(bias, sh.N, 0x5555u, 0.5f);
      CK(hipDeviceSynchronize());
      const double flops = 2.0 * sh.M * sh.N * (double)sh.K;
      if (have_ref) {
        CK(hipMemset(Cref, 0xee, cbytes)); CK(hipDeviceSynchronize());
        if (R.k_gemm(R.ctx, WJ_F16, A, W, bias, Cref, sh.M, sh.N, sh.K, sh.gelu, sh.f32, ref_variant, nullptr)) {
          fprintf(stderr, "reference library failed: %s\n", R.last_error());
          exit(5);
        }
        R.sync(R.ctx);
        float ms = 0.f;
        R.k_gemm_timed(R.ctx, WJ_F16, A, W, bias, C, sh.M, sh.N, sh.K, sh.gelu, sh.f32, ref_variant, reps, &ms);
        printf("{\"shape\": \"%s\", \"M\": %d, \"N\": %d, \"K\": %d, \"variant\": %d, \"lib\": \"previous commit\", \"ms\": %.4f, "
               "\"tflops\": %.1f}\n", sh.name, sh.M, sh.N, sh.K, ref_variant, ms, flops / ms * 1e-9);
      }
      auto one = [&](int variant, const char* tkey, int tval) {
        if (tkey) WJ(L.tune(tkey, tval));
        const bool is_ref = !have_ref && variant == ref_variant && !tkey;
        CK(hipMemset(C, 0xff, cbytes)); CK(hipDeviceSynchronize());
        int rc = L.k_gemm(ctx, WJ_F16, A, W, bias, is_ref ? Cref : C, sh.M, sh.N, sh.K, sh.gelu, sh.f32, variant, nullptr);
        if (rc) {
          printf("{\"shape\": \"%s\", \"variant\": %d, \"error\": \"%s\"}\n", sh.name, variant, L.last_error());
          return;
        }
        WJ(L.sync(ctx));
        unsigned long long worst = 0;
        int bad_runs = 0;
        if (!is_ref) {
          for (int r = 0; r < screen; ++r) {
            if (r) {
              CK(hipMemset(C, 0xff, cbytes)); CK(hipDeviceSynchronize());
              WJ(L.k_gemm(ctx, WJ_F16, A, W, bias, C, sh.M, sh.N, sh.K, sh.gelu, sh.f32, variant, nullptr));
              WJ(L.sync(ctx));
            }
            CK(hipMemset(d_cnt, 0, 8)); CK(hipDeviceSynchronize());
            count_diff<<<2048, 256>>>((const uint32_t*)Cref, (const uint32_t*)C, cbytes / 4, d_cnt);
            unsigned long long h = 0;
            CK(hipMemcpy(&h, d_cnt, 8, hipMemcpyDeviceToHost));
            if (h) ++bad_runs;
            if (h > worst) worst = h;
          }
        }
        float ms = 0.f;
        WJ(L.k_gemm_timed(ctx, WJ_F16, A, W, bias, C, sh.M, sh.N, sh.K, sh.gelu, sh.f32, variant, reps, &ms));
        printf("{\"shape\": \"%s\", \"M\": %d, \"N\": %d, \"K\": %d, \"gelu\": %d, \"f32\": %d, \"variant\": %d, \"tune\": \"%s=%d\", "
               "\"ms\": %.4f, \"tflops\": %.1f, \"mismatch_words\": %llu, \"bad_runs\": %d, \"screen\": %d}\n",
               sh.name, sh.M, sh.N, sh.K, sh.gelu, sh.f32, variant, tkey ? tkey : "", tkey ? tval : 0, ms, flops / ms * 1e-9,
               worst, bad_runs, screen);
        fflush(stdout);
      };
      one(ref_variant, nullptr, 0);
      for (auto& t : tune_ab) { one(ref_variant, t.first, t.second); }
      for (auto& t : tune_ab) WJ(L.tune(t.first, 1));   // back to the defaults (all A/B keys used here default to 1)
      for (int v : variants) one(v, nullptr, 0);
      CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(bias)); CK(hipFree(Cref)); CK(hipFree(C));
    }
  };

  if (what == "enc" || what == "all") {
    // encoder GEMMs of one 96-window slice of a 384-window batch (M = 96 * 1500 rounded to 256-row tiles)
    const int M = 144128;   // 563 row tiles
    std::vector<Shape> enc = {
        {"enc_fc1", M, 5120, 1280, 1, 0}, {"enc_fc2", M, 1280, 5120, 0, 0}, {"enc_qk", M, 2560, 1280, 0, 0},
        {"enc_out", M, 1280, 1280, 0, 0}, {"enc_out_f32", M, 1280, 1280, 0, 1}};
    run_set(enc, 6, {86, 87}, {});
  }
  if (what == "dma") {
    for (auto sh : {Shape{"fc2", 144128, 1280, 5120, 0, 0}, Shape{"qk", 144128, 2560, 1280, 0, 0}, Shape{"fc1", 144128, 5120, 1280, 0, 0}}) {
      _Float16 *A, *W;
      unsigned* sink;
      CK(hipMalloc(&A, (int64_t)sh.M * sh.K * 2)); CK(hipMalloc(&W, (int64_t)sh.N * sh.K * 2)); CK(hipMalloc(&sink, 4));
      fill_f16<<<2048, 256>>>(A, (int64_t)sh.M * sh.K, 1u, 1.0f);
      fill_f16<<<2048, 256>>>(W, (int64_t)sh.N * sh.K, 2u, 1.0f);
      CK(hipDeviceSynchronize());
      run_dma<64, 1>(sh.name, A, W, sh.M, sh.N, sh.K, sink, reps);
      run_dma<64, 2>(sh.name, A, W, sh.M, sh.N, sh.K, sink, reps);
      run_dma<64, 4>(sh.name, A, W, sh.M, sh.N, sh.K, sink, reps);
      run_dma<65, 1>(sh.name, A, W, sh.M, sh.N, sh.K, sink, reps);
      run_dma<128, 1>(sh.name, A, W, sh.M, sh.N, sh.K, sink, reps);
      run_dma<256, 0>(sh.name, A, W, sh.M, sh.N, sh.K, sink, reps);
      CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(sink));
    }
  }
  if (what == "abl") {   // timing ablations of the ping-pong kernel (their results are wrong by design: ignore the mismatch columns)
    std::vector<Shape> enc = {{"abl_fc2", 144128, 1280, 5120, 0, 0}, {"abl_qk", 144128, 2560, 1280, 0, 0}};
    run_set(enc, 6, {83, 113, 123, 133, 143, 153, 163, 183, 193, 203, 223, 253}, {});
  }
  if (what == "dec" || what == "all") {
    // decode-step GEMMs of a 384-window beam-5 batch
    std::vector<Shape> dec = {{"dec_qkv", 1920, 3840, 1280, 0, 0}, {"dec_fc1", 1920, 5120, 1280, 1, 0},
                              {"dec_out", 1920, 1280, 1280, 0, 1}, {"dec_fc2", 1920, 1280, 5120, 0, 1}};
    run_set(dec, 3, {4, 73, 74, 75}, {{"epi_wide", 0}});
  }
  WJ(L.shutdown(ctx));
  if (have_ref) R.shutdown(R.ctx);
  return 0;
}
