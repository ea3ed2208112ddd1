// Micro-benchmark behind the host-boundary design of ac_step_host: what does it cost a kernel to read its actions from
// and write its outputs to pinned, device-mapped host memory, against separate copy-engine transfers, and how is completion seen
// soonest?   hipcc --offload-arch=gfx950 -O3 -o host_io host_io.hip && ./host_io
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <vector>
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

constexpr int N = 8192, OBS = 15, ACT = 4;
// stand-in for the step kernel's compute: a dependent FMA chain of ~17 us
__device__ __forceinline__ float spin(float x, int iters) { for (int i = 0; i < iters; ++i) x = x * 1.0000001f + 1e-9f; return x; }

__global__ void k_scattered(const float* act, float* obs, float* rew, unsigned char* done, int iters) {
  int n = blockIdx.x * 64 + threadIdx.x;
  const float* a = act + n * ACT;
  float x = a[0] + a[1] + a[2] + a[3];
  x = spin(x, iters);
  for (int k = 0; k < OBS; ++k) obs[n * OBS + k] = x + k;
  rew[n] = x; done[n] = x > 1e30f;
}
__global__ void k_coalesced(const float4* act, float* obs, float* rew, unsigned char* done, int iters) {
  __shared__ float L[64 * OBS];
  int n = blockIdx.x * 64 + threadIdx.x;
  float4 a = act[n];
  float x = a.x + a.y + a.z + a.w;
  x = spin(x, iters);
  for (int k = 0; k < OBS; ++k) L[threadIdx.x * OBS + k] = x + k;
  __syncthreads();
  float4* o4 = reinterpret_cast<float4*>(obs + (size_t)blockIdx.x * 64 * OBS);
  const float4* l4 = reinterpret_cast<const float4*>(L);
  for (int i = threadIdx.x; i < 64 * OBS / 4; i += 64) o4[i] = l4[i];
  rew[n] = x;
  unsigned long long b = __ballot(x > 1e30f);
  if (threadIdx.x < 16) reinterpret_cast<unsigned*>(done + blockIdx.x * 64)[threadIdx.x] =
      ((b >> (4 * threadIdx.x)) & 1) | (((b >> (4 * threadIdx.x + 1)) & 1) << 8) | (((b >> (4 * threadIdx.x + 2)) & 1) << 16) | (((b >> (4 * threadIdx.x + 3)) & 1) << 24);
}

// the same with non-temporal stores (do they leave the GPU as larger PCIe writes?)
__global__ void k_coalesced_nt(const float4* act, float* obs, float* rew, unsigned char* done, int iters) {
  __shared__ float L[64 * OBS];
  int n = blockIdx.x * 64 + threadIdx.x;
  float4 a = act[n];
  float x = a.x + a.y + a.z + a.w;
  x = spin(x, iters);
  for (int k = 0; k < OBS; ++k) L[threadIdx.x * OBS + k] = x + k;
  __syncthreads();
  typedef float fx4 __attribute__((ext_vector_type(4)));
  fx4* o4 = reinterpret_cast<fx4*>(obs + (size_t)blockIdx.x * 64 * OBS);
  const fx4* l4 = reinterpret_cast<const fx4*>(L);
  for (int i = threadIdx.x; i < 64 * OBS / 4; i += 64) __builtin_nontemporal_store(l4[i], &o4[i]);
  __builtin_nontemporal_store(x, &rew[n]);
  unsigned long long b = __ballot(x > 1e30f);
  if (threadIdx.x < 16) reinterpret_cast<unsigned*>(done + blockIdx.x * 64)[threadIdx.x] = (unsigned)((b >> (4 * threadIdx.x)) & 1);
}

// completion flag in mapped host memory: every workgroup fences its output stores at system scope and bumps a device counter; the last one
// to arrive writes the launch's sequence number where the host is polling
__global__ void k_flag(const float4* act, float* obs, float* rew, unsigned char* done, int iters, unsigned* counter, volatile unsigned* flag, unsigned seq) {
  __shared__ float L[64 * OBS];
  int n = blockIdx.x * 64 + threadIdx.x;
  float4 a = act[n];
  float x = a.x + a.y + a.z + a.w;
  x = spin(x, iters);
  for (int k = 0; k < OBS; ++k) L[threadIdx.x * OBS + k] = x + k;
  __syncthreads();
  float4* o4 = reinterpret_cast<float4*>(obs + (size_t)blockIdx.x * 64 * OBS);
  const float4* l4 = reinterpret_cast<const float4*>(L);
  for (int i = threadIdx.x; i < 64 * OBS / 4; i += 64) o4[i] = l4[i];
  rew[n] = x;
  unsigned long long b = __ballot(x > 1e30f);
  if (threadIdx.x < 16) reinterpret_cast<unsigned*>(done + blockIdx.x * 64)[threadIdx.x] = (unsigned)((b >> (4 * threadIdx.x)) & 1);
  __threadfence_system();
  if (threadIdx.x == 0) {
    const unsigned old = atomicAdd(counter, 1u);
    if (old == gridDim.x - 1) { *counter = 0; __threadfence_system(); *flag = seq; }
  }
}

int main(int argc, char** argv) {
  int iters = argc > 1 ? atoi(argv[1]) : 9000;
  hipStream_t s; OK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  float *d_act, *d_obs, *d_rew; unsigned char* d_done;
  OK(hipMalloc(&d_act, N * ACT * 4)); OK(hipMalloc(&d_obs, N * OBS * 4)); OK(hipMalloc(&d_rew, N * 4)); OK(hipMalloc(&d_done, N));
  float *h_act, *h_obs, *h_rew; unsigned char* h_done;
  for (int mode = 0; mode < 2; ++mode) {
    unsigned flags = mode == 0 ? hipHostMallocDefault : (hipHostMallocMapped | hipHostMallocNonCoherent);
    OK(hipHostMalloc(&h_act, N * ACT * 4, flags)); OK(hipHostMalloc(&h_obs, N * OBS * 4, flags));
    OK(hipHostMalloc(&h_rew, N * 4, flags)); OK(hipHostMalloc(&h_done, N, flags));
    for (int i = 0; i < N * ACT; ++i) h_act[i] = (float)(i % 41);
    float *m_act, *m_obs, *m_rew; unsigned char* m_done;
    OK(hipHostGetDevicePointer((void**)&m_act, h_act, 0)); OK(hipHostGetDevicePointer((void**)&m_obs, h_obs, 0));
    OK(hipHostGetDevicePointer((void**)&m_rew, h_rew, 0)); OK(hipHostGetDevicePointer((void**)&m_done, h_done, 0));
    printf("---- host memory: %s\n", mode == 0 ? "hipHostMallocDefault (coherent)" : "Mapped | NonCoherent");
    const int R = 300;
    auto wait_spin = [&]() { while (hipStreamQuery(s) == hipErrorNotReady) {} };
    auto wait_sync = [&]() { OK(hipStreamSynchronize(s)); };
    for (int w = 0; w < 2; ++w) {
      auto wait = [&]() { if (w) wait_spin(); else wait_sync(); };
      const char* wn = w ? "spin on hipStreamQuery" : "hipStreamSynchronize";
      // (a) copy-engine path: H2D, kernel on device buffers, 3 D2H
      for (int r = 0; r < 20; ++r) { OK(hipMemcpyAsync(d_act, h_act, N * ACT * 4, hipMemcpyHostToDevice, s)); hipLaunchKernelGGL(k_scattered, dim3(N / 64), dim3(64), 0, s, d_act, d_obs, d_rew, d_done, iters); wait(); }
      double t0 = now();
      for (int r = 0; r < R; ++r) {
        OK(hipMemcpyAsync(d_act, h_act, N * ACT * 4, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_scattered, dim3(N / 64), dim3(64), 0, s, d_act, d_obs, d_rew, d_done, iters);
        OK(hipMemcpyAsync(h_obs, d_obs, N * OBS * 4, hipMemcpyDeviceToHost, s));
        OK(hipMemcpyAsync(h_rew, d_rew, N * 4, hipMemcpyDeviceToHost, s));
        OK(hipMemcpyAsync(h_done, d_done, N, hipMemcpyDeviceToHost, s));
        wait();
      }
      printf("%-24s copies (H2D + kernel + 3 D2H)            %7.1f us/step\n", wn, (now() - t0) / R * 1e6);
      // (b) kernel alone on device buffers
      t0 = now();
      for (int r = 0; r < R; ++r) { hipLaunchKernelGGL(k_scattered, dim3(N / 64), dim3(64), 0, s, d_act, d_obs, d_rew, d_done, iters); wait(); }
      printf("%-24s kernel only, device buffers             %7.1f us/step\n", wn, (now() - t0) / R * 1e6);
      t0 = now();
      for (int r = 0; r < R; ++r) { hipLaunchKernelGGL(k_coalesced, dim3(N / 64), dim3(64), 0, s, (const float4*)d_act, d_obs, d_rew, d_done, iters); wait(); }
      printf("%-24s kernel only, device buffers, coalesced  %7.1f us/step\n", wn, (now() - t0) / R * 1e6);
      // (c) kernel reads / writes mapped host memory directly
      t0 = now();
      for (int r = 0; r < R; ++r) { hipLaunchKernelGGL(k_scattered, dim3(N / 64), dim3(64), 0, s, m_act, m_obs, m_rew, m_done, iters); wait(); }
      printf("%-24s zero-copy, scattered dword stores       %7.1f us/step\n", wn, (now() - t0) / R * 1e6);
      t0 = now();
      for (int r = 0; r < R; ++r) { hipLaunchKernelGGL(k_coalesced, dim3(N / 64), dim3(64), 0, s, (const float4*)m_act, m_obs, m_rew, m_done, iters); wait(); }
      printf("%-24s zero-copy, LDS-transposed 16 B stores   %7.1f us/step\n", wn, (now() - t0) / R * 1e6);
      t0 = now();
      for (int r = 0; r < R; ++r) { hipLaunchKernelGGL(k_coalesced, dim3(N / 64), dim3(64), 0, s, (const float4*)m_act, d_obs, d_rew, d_done, iters); wait(); }
      printf("%-24s actions from host, outputs to device    %7.1f us/step\n", wn, (now() - t0) / R * 1e6);
      t0 = now();
      for (int r = 0; r < R; ++r) { hipLaunchKernelGGL(k_coalesced, dim3(N / 64), dim3(64), 0, s, (const float4*)d_act, m_obs, m_rew, m_done, iters); wait(); }
      printf("%-24s actions from device, outputs to host    %7.1f us/step\n", wn, (now() - t0) / R * 1e6);
      t0 = now();
      for (int r = 0; r < R; ++r) { hipLaunchKernelGGL(k_coalesced_nt, dim3(N / 64), dim3(64), 0, s, (const float4*)d_act, m_obs, m_rew, m_done, iters); wait(); }
      printf("%-24s the same, non-temporal stores           %7.1f us/step\n", wn, (now() - t0) / R * 1e6);
      // (d) one packed D2H after the kernel
      t0 = now();
      for (int r = 0; r < R; ++r) {
        hipLaunchKernelGGL(k_coalesced, dim3(N / 64), dim3(64), 0, s, (const float4*)m_act, d_obs, d_rew, d_done, iters);
        OK(hipMemcpyAsync(h_obs, d_obs, N * OBS * 4, hipMemcpyDeviceToHost, s));
        wait();
      }
      printf("%-24s actions from host + ONE D2H copy        %7.1f us/step\n", wn, (now() - t0) / R * 1e6);
    }
    float chk = 0; for (int i = 0; i < N * OBS; ++i) chk += h_obs[i];
    printf("checksum %g\n", chk);
    OK(hipHostFree(h_act)); OK(hipHostFree(h_obs)); OK(hipHostFree(h_rew)); OK(hipHostFree(h_done));
  }
  {   // completion flag polled in host memory against hipStreamSynchronize
    float *h_act, *h_obs, *h_rew; unsigned char* h_done; unsigned* h_flag; unsigned* d_counter;
    OK(hipHostMalloc(&h_act, N * ACT * 4, hipHostMallocDefault)); OK(hipHostMalloc(&h_obs, N * OBS * 4, hipHostMallocDefault));
    OK(hipHostMalloc(&h_rew, N * 4, hipHostMallocDefault)); OK(hipHostMalloc(&h_done, N, hipHostMallocDefault));
    OK(hipHostMalloc(&h_flag, 64, hipHostMallocDefault)); OK(hipMalloc(&d_counter, 4)); OK(hipMemset(d_counter, 0, 4));
    for (int i = 0; i < N * ACT; ++i) h_act[i] = (float)(i % 41);
    *h_flag = 0;
    const int R = 300;
    unsigned seq = 0;
    double t_flag = 0, t_sync = 0, t_only = 0;
    for (int r = 0; r < R + 20; ++r) {
      ++seq;
      double t0 = now();
      hipLaunchKernelGGL(k_flag, dim3(N / 64), dim3(64), 0, s, (const float4*)h_act, h_obs, h_rew, h_done, iters, d_counter, (volatile unsigned*)h_flag, seq);
      while (*(volatile unsigned*)h_flag != seq) {}
      double t1 = now();
      OK(hipStreamSynchronize(s));
      double t2 = now();
      if (r >= 20) { t_flag += t1 - t0; t_sync += t2 - t0; }
    }
    for (int r = 0; r < R + 20; ++r) {
      ++seq;
      double t0 = now();
      hipLaunchKernelGGL(k_flag, dim3(N / 64), dim3(64), 0, s, (const float4*)h_act, h_obs, h_rew, h_done, iters, d_counter, (volatile unsigned*)h_flag, seq);
      while (*(volatile unsigned*)h_flag != seq) {}
      double t1 = now();
      if (r >= 20) t_only += t1 - t0;     // no hipStreamSynchronize at all between launches
    }
    OK(hipStreamSynchronize(s));
    float chk = 0; for (int i = 0; i < N * OBS; ++i) chk += h_obs[i];
    printf("zero-copy + completion flag in host memory: flag seen %.1f us, hipStreamSynchronize returned %.1f us after the launch call; flag only, back to back %.1f us/step (checksum %g)\n",
           t_flag / R * 1e6, t_sync / R * 1e6, t_only / R * 1e6, chk);
  }
  // empty-kernel launch + wait latency
  for (int w = 0; w < 2; ++w) {
    double t0 = now();
    for (int r = 0; r < 1000; ++r) { hipLaunchKernelGGL(k_scattered, dim3(1), dim3(64), 0, s, d_act, d_obs, d_rew, d_done, 0); if (w) { while (hipStreamQuery(s) == hipErrorNotReady) {} } else OK(hipStreamSynchronize(s)); }
    printf("launch + wait of a trivial kernel (%s): %.1f us\n", w ? "spin" : "sync", (now() - t0) / 1000 * 1e6);
  }
  return 0;
}
