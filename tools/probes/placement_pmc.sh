#!/bin/bash
# tools/probes/placement_pmc.hip on the GPU box: one plain run, then PMC passes (each in its own run, --pmc only).
set -o pipefail
out=$GRAFT_REPO_ROOT/gpurun_out/placement_pmc
mkdir -p $out
P=$GRAFT_REPO_ROOT/tools/probes/placement_pmc
cd /tmp && export TMPDIR=/tmp
$P 24 > $out/plain.log 2>&1 || { echo plain run failed; tail -5 $out/plain.log; exit 1; }
cat $out/plain.log
pass() {
    name=$1; shift
    rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- $P 12 > $out/$name.log 2>&1 || { echo $name failed; tail -5 $out/$name.log; return 1; }
    echo pass $name ok
}
pass utcl1a TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum &&
pass utcl1b TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum &&
pass utcl1c TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_LFIFO_FULL_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum &&
pass tcc1 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum &&
pass tcc2 TCC_TAG_STALL_sum TCC_IB_STALL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_BUSY_sum &&
pass grbm GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY &&
pass sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR
du -sh $out
