#!/bin/bash
# Profile bench.py on the GPU box: kernel trace + PMC passes (each in its own run).
# usage: tools/prof_bench.sh <tag> [extra bench args]
set -o pipefail
tag=${1:-r05}; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
# The library sets this itself when it is loaded -- but under rocprofv3 the profiler's own library has initialised the HIP runtime before
# python starts, and the runtime reads the variable then: without the export the profiled kernels fetch their arguments from host memory
# (+1.6 us on the frames kernel) and the trace is not the product's.
export HIP_FORCE_DEV_KERNARG=1
# the default bench.py command (steps 2000, warmup 200, HIP-event bracketing on), minus the CPU baseline leg
B="$GRAFT_REPO_ROOT/bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-extra-configs --no-placement-compare $@"
# the counter passes slow the placement probe down, the arena then calls the buffer slow-class and the library would pick the tile-major
# kernel for it: the counters are wanted for the kernel the trace pass (and the bench) runs, so the PMC passes name it
T="$B"
B="$B --frames-kernel 2"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $T > $out/trace.log 2>&1 || { echo trace failed; tail -5 $out/trace.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $B > $out/pmc_fetch.log 2>&1 || { echo fetch failed; tail -5 $out/pmc_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $out/pmc_write -- python3 $B > $out/pmc_write.log 2>&1 || { echo write failed; tail -5 $out/pmc_write.log; exit 1; }
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $out/pmc_sq1 -- python3 $B > $out/pmc_sq1.log 2>&1 || { echo sq1 failed; tail -5 $out/pmc_sq1.log; exit 1; }
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS --output-format csv -d $out/pmc_sq2 -- python3 $B > $out/pmc_sq2.log 2>&1 || { echo sq2 failed; tail -5 $out/pmc_sq2.log; exit 1; }
rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_misc -- python3 $B > $out/pmc_misc.log 2>&1 || { echo misc failed; tail -5 $out/pmc_misc.log; exit 1; }
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CU_CYCLES --output-format csv -d $out/pmc_mfma -- python3 $B > $out/pmc_mfma.log 2>&1 || { echo mfma failed; tail -5 $out/pmc_mfma.log; exit 1; }
find $out -name '*.csv' | tail -40
du -sh $out
