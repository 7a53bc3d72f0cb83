#!/bin/bash
# the persistent double-buffered radix-256 pass (k_ntt_pass_pers, MI_NTT_PERSISTENT=1) under the same two rocprofv3
# passes as tools/pmc_ntt.sh (kernel trace; SQ counters), beside the default build, for profiles/r04_pmc_ntt_persistent.txt.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/pmc_ntt_pers
mkdir -p $OUT
BENCH="python3 bench.py --steps 2 --warmup 1 --no-verify --no-cpu-baseline --no-genproof --pcie-steps 0"
for p in 0 1; do
    export MI_NTT_PERSISTENT=$p
    timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $OUT/p${p}_trace -o p${p}_trace --output-format csv -- $BENCH > $OUT/p${p}_trace.log 2>&1 || { echo "trace $p failed" >> $OUT/status.txt; exit 1; }
    timeout -k 10 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE \
        -d $OUT/p${p}_sq1 -o p${p}_sq1 --output-format csv -- $BENCH > $OUT/p${p}_sq1.log 2>&1 || { echo "sq1 $p failed" >> $OUT/status.txt; exit 1; }
done
echo done >> $OUT/status.txt
