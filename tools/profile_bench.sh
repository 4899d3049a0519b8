#!/bin/bash
# GPU box: rocprofv3 passes of `python bench.py <args>` into gpurun_out/<prefix>_{kt,pmc1..5}; every pass under its own `timeout`
#   bash tools/profile_bench.sh <prefix> [bench.py args...]
# --kernel-trace --stats in its own pass; the PMC counters in separate passes (never combined with tracing).
# Reduce with: python tools/summarize_profile.py <prefix> <round> <name> [kernel-substring]
P=$1; shift
ONLY=${ONLY:-kt pmc1 pmc2 pmc3 pmc4 pmc5}   # subset of passes to run; STEPS / WARMUP: launches per pass (short kernels: more, so that
                                            # the first, cold launch does not carry the average)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps ${STEPS:-3} --warmup ${WARMUP:-1} --no-cpu-baseline --no-extras --no-parity $*"
for p in $ONLY; do rm -rf $O/${P}_$p; done
[[ " $ONLY " == *" kt "* ]] && timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${P}_kt -o kt -- $B > $O/${P}_kt.log 2>&1
[[ " $ONLY " == *" pmc1 "* ]] && timeout 420 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/${P}_pmc1 -o p -- $B > $O/${P}_pmc1.log 2>&1
# TCC has 4 counter slots: FETCH_SIZE costs 3 and WRITE_SIZE 2 (MI355X_MICROARCH.md), so one pass each -- an
# over-subscribed pass fails with "Request exceeds the capabilities of the hardware" and rocprofv3 then hangs
[[ " $ONLY " == *" pmc2 "* ]] && timeout 420 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${P}_pmc2 -o p -- $B > $O/${P}_pmc2.log 2>&1
[[ " $ONLY " == *" pmc5 "* ]] && timeout 420 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${P}_pmc5 -o p -- $B > $O/${P}_pmc5.log 2>&1
[[ " $ONLY " == *" pmc4 "* ]] && timeout 420 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/${P}_pmc4 -o p -- $B > $O/${P}_pmc4.log 2>&1
[[ " $ONLY " == *" pmc3 "* ]] && timeout 420 rocprofv3 --pmc SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU --output-format csv -d $O/${P}_pmc3 -o p -- $B > $O/${P}_pmc3.log 2>&1
tail -n 1 $O/${P}_kt.log
