#!/bin/bash
# HBM traffic and SQ counters of the dominant kernel (k_linear_hash_rows_lines) of bench.py's default workload: three separate rocprofv3
# --pmc passes (SQ, FETCH_SIZE, WRITE_SIZE: MI355X_MICROARCH.md "HBM / rocprofv3"), summarised by tools/pmc_leaf_json.py into
# profiles/r04_pmc_leaf.json, which bench.py reads for roofline.traffic and the valu roofline.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/pmc_leaf
mkdir -p $OUT
BENCH="python3 bench.py --steps 2 --warmup 1 --no-verify --no-cpu-baseline --no-genproof --pcie-steps 0"
run() { local name=$1; shift; timeout -k 10 300 rocprofv3 "$@" -d $OUT/$name -o $name --output-format csv -- $BENCH > $OUT/$name.log 2>&1 || { echo "$name failed" >> $OUT/status.txt; exit 1; }; }
run sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run trace --kernel-trace --stats
echo done >> $OUT/status.txt
