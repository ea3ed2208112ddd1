#!/bin/bash
# Cycle stamps of workgroup 0 of the headline kernel and of the controller kernel in one pass (variants/libclk.so from
# tools/build_clk_variant.sh) -> gpurun_out/<tag>_cycle_stamps.txt (copy to profiles/).  usage (inside gpurun): bash tools/stamps_round.sh <tag>
tag=${1:-round}
out=gpurun_out/${tag}_cycle_stamps.txt
{
  echo "# Cycle stamps behind DESIGN.md section 5 ($tag; workgroup 0 of each kernel, variants/libclk.so = the product sources + -DAC_SPLIT_TIMING)"
  echo
  echo "## tools/diag/clk_prologue.py -- three-wave singlecombat kernel, 4096 envs: the prologue piece by piece, host-boundary and device-resident steps"
  python3 tools/diag/clk_prologue.py 2>&1 | grep -v amdgpu.ids
  echo
  echo "## tools/diag/clk_split.py -- the same kernel tick by tick (steps through VecEnv.step: the action row crosses PCIe, tick 0's wait at B2 is the systems wave waiting for it)"
  python3 tools/diag/clk_split.py 2>&1 | grep -v amdgpu.ids | tail -17
  echo
  echo "## tools/diag/clk_controller.py -- controller8_kernel, 32 aircraft per workgroup (8192 aircraft), device-resident and host-boundary steps"
  AIRCOMBAT_CTL_ROWS=32 python3 tools/diag/clk_controller.py 4096 2>&1 | grep -v amdgpu.ids | tail -3
  AC_DEVICE_RESIDENT=0 AIRCOMBAT_CTL_ROWS=32 python3 tools/diag/clk_controller.py 4096 2>&1 | grep -v amdgpu.ids | tail -3
  echo
  echo "## the same, 64 aircraft per workgroup (32 768 aircraft)"
  AIRCOMBAT_CTL_ROWS=64 python3 tools/diag/clk_controller.py 16384 2>&1 | grep -v amdgpu.ids | tail -3
} > $out 2>&1
cat $out
