#!/bin/bash
# SQ counters and HBM traffic of the generated constraint kernels (chelpers_chunk) and their operand copies in ONE Starks::genProof at zkEVM
# size: three separate rocprofv3 --pmc passes of bench_starks.py (SQ, FETCH_SIZE, WRITE_SIZE: MI355X_MICROARCH.md "HBM / rocprofv3"),
# summarised per step by tools/pmc_chunk_table.py into profiles/r04_pmc_chelpers.txt.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/pmc_chunk
mkdir -p $OUT
BENCH="python3 bench_starks.py --proofs 1 --check-rows 0"
run() { local name=$1; shift; timeout -k 10 400 rocprofv3 "$@" -d $OUT/$name -o $name --output-format csv -- $BENCH > $OUT/$name.log 2>&1 || { echo "$name failed" >> $OUT/status.txt; exit 1; }; }
run sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
echo done >> $OUT/status.txt
