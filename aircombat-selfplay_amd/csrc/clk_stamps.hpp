// Cycle stamps for scratch builds (tools/build_clk_variant.sh -> variants/libclk.so, read by tools/diag/clk_*.py): AC_CLK(i) stamps
// slot i from the first wave of workgroup 0, AC_CLKW(w, i) from its wave w. The product build compiles them to nothing.
#pragma once
#ifdef AC_SPLIT_TIMING
#ifndef AC_CLK_BLOCK
#define AC_CLK_BLOCK 0   // the workgroup that stamps (-DAC_CLK_BLOCK=n for another one)
#endif
__device__ unsigned long long g_clk[256];
#define AC_CLK(i) do { if (blockIdx.x == AC_CLK_BLOCK && threadIdx.x == 0) g_clk[i] = __builtin_readcyclecounter(); } while (0)
#define AC_CLKW(w, i) do { if (blockIdx.x == AC_CLK_BLOCK && threadIdx.x == 64 * (w)) g_clk[i] = __builtin_readcyclecounter(); } while (0)   // wave w of workgroup 0
extern "C" void ac_debug_clocks(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_clk), sizeof g_clk); }
#else
#define AC_CLK(i) do {} while (0)
#define AC_CLKW(w, i) do {} while (0)
#endif
