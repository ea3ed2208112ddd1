// Micro-benchmark behind the resident step kernel ("Step outputs and the host boundary"): a kernel that stays on the GPU and
// is told to run a step through a word in pinned host memory, against launching a kernel per step. Same stand-in work and the same
// mapped host buffers as host_io.hip.   hipcc --offload-arch=gfx950 -O3 -o resident resident.hip && ./resident
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

constexpr int N = 8192, OBS = 15;
__device__ __forceinline__ float spin(float x, int iters) { for (int i = 0; i < iters; ++i) x = x * 1.0000001f + 1e-9f; return x; }

__device__ __forceinline__ void body(const float4* act, float* obs, float* rew, int iters, float* L) {
  int n = blockIdx.x * 64 + threadIdx.x;
  float4 a = act[n];
  float x = spin(a.x + a.y + a.z + a.w, iters);
  for (int k = 0; k < OBS; ++k) L[threadIdx.x * OBS + k] = x + k;
  __syncthreads();
  float4* o4 = reinterpret_cast<float4*>(obs + (size_t)blockIdx.x * 64 * OBS);
  const float4* l4 = reinterpret_cast<const float4*>(L);
  for (int i = threadIdx.x; i < 64 * OBS / 4; i += 64) o4[i] = l4[i];
  rew[n] = x;
}
__global__ void k_step(const float4* act, float* obs, float* rew, int iters) {
  __shared__ float L[64 * OBS];
  body(act, obs, rew, iters, L);
}
struct Mail { unsigned long long cmd; char p0[56]; unsigned long long done; char p1[56]; unsigned long long stamp[8]; };
// cmd = (seq << 1) | exit.  Every workgroup polls the word itself; the last one to finish a step (device-scope counter) reports it.
template <int V>
__global__ void k_resident(const float4* act, float* obs, float* rew, int iters, Mail* mail, unsigned long long* counter, unsigned long long idle_ticks) {
  __shared__ float L[64 * OBS];
  __shared__ unsigned long long s_cmd;
  unsigned long long seq = 0;
  for (;;) {
    if (threadIdx.x == 0) {
      const unsigned long long t0 = wall_clock64();
      unsigned long long w;
      for (;;) {
        w = __hip_atomic_load(&mail->cmd, V == 0 ? __ATOMIC_ACQUIRE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((w >> 1) == seq + 1 || (w & 1)) break;
        if (wall_clock64() - t0 > idle_ticks) { w = 1; break; }
        __builtin_amdgcn_s_sleep(4);
      }
      if (V >= 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
      s_cmd = w;
    }
    __syncthreads();
    const unsigned long long w = s_cmd;
    if (w & 1) break;
    seq += 1;
    const unsigned long long c0 = wall_clock64();
    body(act, obs, rew, iters, L);
    const unsigned long long c1 = wall_clock64();
    if (V == 0) __threadfence_system();
    else if (V == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    else __builtin_amdgcn_s_waitcnt(0);     // V == 2: the stores to (uncached) host memory have been acknowledged, nothing else
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) { mail->stamp[0] += c1 - c0; mail->stamp[1] += wall_clock64() - c1; }
    if (threadIdx.x == 0) {
      const unsigned long long old = __hip_atomic_fetch_add(counter, 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (old + 1 == seq * gridDim.x) __hip_atomic_store(&mail->done, seq, V == 2 ? __ATOMIC_RELAXED : __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

int main() {
  hipStream_t st; OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  float4* act; float *obs, *rew; Mail* mail; unsigned long long* counter;
  OK(hipHostMalloc((void**)&act, sizeof(float4) * N, hipHostMallocDefault));
  OK(hipHostMalloc((void**)&obs, sizeof(float) * N * OBS, hipHostMallocDefault));
  OK(hipHostMalloc((void**)&rew, sizeof(float) * N, hipHostMallocDefault));
  OK(hipHostMalloc((void**)&mail, sizeof(Mail), hipHostMallocDefault));
  OK(hipMalloc(&counter, 8)); OK(hipMemset(counter, 0, 8));
  memset(act, 0, sizeof(float4) * N); memset(mail, 0, sizeof(Mail));
  const int iters = 9000, K = 3000;   // ~17 us of dependent FMAs
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_step, dim3(N / 64), dim3(64), 0, st, act, obs, rew, iters);
  OK(hipStreamSynchronize(st));
  double t0 = now();
  for (int i = 0; i < K; ++i) { act[i & 1023].x = (float)i; hipLaunchKernelGGL(k_step, dim3(N / 64), dim3(64), 0, st, act, obs, rew, iters); OK(hipStreamSynchronize(st)); }
  printf("launch per step + hipStreamSynchronize: %.2f us/step\n", (now() - t0) / K * 1e6);
  for (int coherent = 0; coherent < 2; ++coherent) {
    if (coherent) {   // the same buffers as fine-grained (uncached on the GPU) host memory
      OK(hipHostMalloc((void**)&act, sizeof(float4) * N, hipHostMallocCoherent)); OK(hipHostMalloc((void**)&obs, sizeof(float) * N * OBS, hipHostMallocCoherent));
      OK(hipHostMalloc((void**)&rew, sizeof(float) * N, hipHostMallocCoherent)); OK(hipHostMalloc((void**)&mail, sizeof(Mail), hipHostMallocCoherent));
      memset(act, 0, sizeof(float4) * N);
    }
    for (int v = 0; v < 3; ++v) for (int its = 0; its < 2; ++its) {
      const int it = its ? iters : 0;
      memset(mail, 0, sizeof(Mail)); OK(hipMemset(counter, 0, 8));
      if (v == 0) hipLaunchKernelGGL(k_resident<0>, dim3(N / 64), dim3(64), 0, st, act, obs, rew, it, mail, counter, 200000000ull);
      if (v == 1) hipLaunchKernelGGL(k_resident<1>, dim3(N / 64), dim3(64), 0, st, act, obs, rew, it, mail, counter, 200000000ull);
      if (v == 2) hipLaunchKernelGGL(k_resident<2>, dim3(N / 64), dim3(64), 0, st, act, obs, rew, it, mail, counter, 200000000ull);
      OK(hipGetLastError());
      unsigned long long seq = 0;
      auto step = [&]() {
        seq += 1;
        __atomic_store_n(&mail->cmd, seq << 1, __ATOMIC_RELEASE);
        const double w0 = now();
        while (__atomic_load_n(&mail->done, __ATOMIC_ACQUIRE) != seq) { __builtin_ia32_pause(); if (now() - w0 > 2.0) { printf("resident kernel did not answer\n"); exit(1); } }
      };
      for (int i = 0; i < 50; ++i) step();
      t0 = now();
      for (int i = 0; i < K; ++i) { act[i & 1023].x = (float)i; step(); }
      const double per = (now() - t0) / K * 1e6;
      printf("%s host memory, protocol %d, %s body: %.2f us/step   (GPU: body %.2f us, fence + barrier %.2f us)  rew[5] = %g\n", coherent ? "coherent" : "default ", v, its ? "17 us" : "empty",
             per, mail->stamp[0] / 100.0 / (K + 50), mail->stamp[1] / 100.0 / (K + 50), rew[5]);
      __atomic_store_n(&mail->cmd, 1ull, __ATOMIC_RELEASE);
      OK(hipStreamSynchronize(st));
    }
  }
  // idle exit: start it again and send nothing
  memset(mail, 0, sizeof(Mail)); OK(hipMemset(counter, 0, 8));
  hipLaunchKernelGGL(k_resident<1>, dim3(N / 64), dim3(64), 0, st, act, obs, rew, iters, mail, counter, 20000000ull /* 0.2 s */);
  t0 = now(); OK(hipStreamSynchronize(st));
  printf("idle exit after %.3f s\n", now() - t0);
  return 0;
}
