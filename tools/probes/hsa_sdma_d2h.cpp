// PROBE: device-to-host copies through the SDMA engines (hsa_amd_memory_async_copy[_on_engine]) instead of the HIP runtime's
// shader copies: rate alone, and beside an HBM-streaming kernel (does either slow the other?).
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/hsa_sdma_d2h.cpp -lhsa-runtime64 -o tools/probes/hsa_sdma_d2h
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define HK(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { printf("HSA error %d line %d\n", (int)s_, __LINE__); return 1; } } while (0)

__global__ void stream_kernel(const double2* a, double2* b, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    double2 v = a[i];
    v.x += 1.0;
    b[i] = v;
  }
}
static hsa_agent_t g_gpu, g_cpu;
static bool have_gpu = false, have_cpu = false;
static hsa_status_t agent_cb(hsa_agent_t ag, void*) {
  hsa_device_type_t t;
  hsa_agent_get_info(ag, HSA_AGENT_INFO_DEVICE, &t);
  if (t == HSA_DEVICE_TYPE_GPU && !have_gpu) g_gpu = ag, have_gpu = true;
  if (t == HSA_DEVICE_TYPE_CPU && !have_cpu) g_cpu = ag, have_cpu = true;
  return HSA_STATUS_SUCCESS;
}
int main() {
  const size_t bytes = 456u << 20, piece = bytes / 10;
  char *d_src, *h_dst;
  double2 *A, *B;
  const long long n = 1LL << 28;  // 4 GB each
  CK(hipMalloc(&d_src, bytes));
  CK(hipHostMalloc(&h_dst, bytes, hipHostMallocDefault));
  CK(hipMalloc(&A, n * 16));
  CK(hipMalloc(&B, n * 16));
  CK(hipMemset(d_src, 1, bytes));
  CK(hipMemset(A, 0, n * 16));
  HK(hsa_init());
  HK(hsa_iterate_agents(agent_cb, nullptr));
  if (!have_gpu || !have_cpu) { printf("agents missing\n"); return 1; }
  uint32_t mask = 0;
  hsa_status_t st = hsa_amd_memory_copy_engine_status(g_cpu, g_gpu, &mask);
  printf("copy engine status (dst cpu, src gpu): status %d mask 0x%x\n", (int)st, mask);
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  auto sdma_copy = [&](int engine_bit) -> double {
    std::vector<hsa_signal_t> sig(10);
    for (auto& x : sig) hsa_signal_create(1, 0, nullptr, &x);
    auto t0 = now();
    for (int k = 0; k < 10; ++k) {
      hsa_status_t r = engine_bit < 0 ? hsa_amd_memory_async_copy(h_dst + k * piece, g_cpu, d_src + k * piece, g_gpu, piece, 0, nullptr, sig[k])
                                      : hsa_amd_memory_async_copy_on_engine(h_dst + k * piece, g_cpu, d_src + k * piece, g_gpu, piece, 0, nullptr, sig[k],
                                                                            (hsa_amd_sdma_engine_id_t)(1u << engine_bit), true);
      if (r != HSA_STATUS_SUCCESS) { printf("copy failed %d\n", (int)r); return -1.0; }
    }
    for (auto& x : sig) hsa_signal_wait_scacquire(x, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED);
    double t = ms(t0, now());
    for (auto& x : sig) hsa_signal_destroy(x);
    return t;
  };
  for (int rep = 0; rep < 2; ++rep) {
    auto t0 = now();
    CK(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    printf("hipMemcpyAsync D2H alone: %.2f ms (%.1f GB/s)\n", ms(t0, now()), bytes / ms(t0, now()) / 1e6);
    double t = sdma_copy(-1);
    printf("hsa_amd_memory_async_copy D2H alone: %.2f ms (%.1f GB/s)\n", t, bytes / t / 1e6);
    for (int bit = 0; bit < 4; ++bit)
      if (mask & (1u << bit)) {
        t = sdma_copy(bit);
        if (t > 0) printf("  on engine bit %d: %.2f ms (%.1f GB/s)\n", bit, t, bytes / t / 1e6);
      }
    // the streaming kernel alone
    t0 = now();
    hipLaunchKernelGGL(stream_kernel, dim3(4096), dim3(256), 0, s, A, B, n);
    CK(hipStreamSynchronize(s));
    const double tk = ms(t0, now());
    printf("stream kernel alone (8.6 GB): %.2f ms (%.2f TB/s)\n", tk, 2.0 * n * 16 / tk / 1e9);
    // both: kernel + HIP copy on another stream
    hipStream_t s2;
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    t0 = now();
    hipLaunchKernelGGL(stream_kernel, dim3(4096), dim3(256), 0, s, A, B, n);
    hipLaunchKernelGGL(stream_kernel, dim3(4096), dim3(256), 0, s, A, B, n);
    CK(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, s2));
    CK(hipStreamSynchronize(s));
    const double tk2 = ms(t0, now());
    CK(hipStreamSynchronize(s2));
    printf("two stream kernels beside hipMemcpyAsync D2H: kernels %.2f ms, all done %.2f ms\n", tk2, ms(t0, now()));
    // both: kernel + SDMA copy
    t0 = now();
    hipLaunchKernelGGL(stream_kernel, dim3(4096), dim3(256), 0, s, A, B, n);
    hipLaunchKernelGGL(stream_kernel, dim3(4096), dim3(256), 0, s, A, B, n);
    t = sdma_copy(-1);
    const double tc = ms(t0, now());
    CK(hipStreamSynchronize(s));
    printf("two stream kernels beside hsa copy D2H: copy done %.2f ms, all done %.2f ms\n", tc, ms(t0, now()));
    CK(hipStreamDestroy(s2));
  }
  return 0;
}
