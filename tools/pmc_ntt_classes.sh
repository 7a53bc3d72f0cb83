#!/bin/bash
# Where the NTT's VALU instructions go, class by class: the passes of bench.py's default workload with the shipped
# library and with five diagnosis builds, each with ONE class of the arithmetic compiled out (csrc/ntt_math.h: MI_NTT_AB_*; the no-arith
# build of tools/pmc_ntt.sh removes all of it).  SQ_INSTS_VALU per launch of every build; the differences are the classes' counts.
# Summarised by tools/pmc_ntt_classes_table.py into profiles/r05_ntt_valu_breakdown.{txt,json}.  Run on the GPU box from the repo root.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/pmc_ntt_classes
mkdir -p $OUT
: > $OUT/status.txt
make -C merlin-zkevm-prover_amd/csrc ab-noarith ab-ntt-classes > $OUT/ab_build.log 2>&1 || { echo "ab build failed" >> $OUT/status.txt; exit 1; }
BENCH="python3 bench.py --steps 2 --warmup 1 --no-verify --no-cpu-baseline --pcie-steps 0 --no-genproof"
run() { # name, lib, rocprof args...
    local name=$1 lib=$2; shift 2
    if [ -n "$lib" ]; then export MI_STARK_LIB=$PWD/$lib; else unset MI_STARK_LIB; fi
    timeout -k 10 200 rocprofv3 "$@" -d $OUT/$name -o $name --output-format csv -- $BENCH > $OUT/$name.log 2>&1
    local rc=$?
    echo "$name rc=$rc" >> $OUT/status.txt
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout in $name: stopping" >> $OUT/status.txt; exit 1; fi
}
for v in shipped noarith NOCANON NOADDSUB NOPOW2 NOMULW; do
    lib=ab_libs/libmi_stark_$v.so; [ $v = shipped ] && lib=""
    run ${v}_sq "$lib" --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES
    run ${v}_trace "$lib" --kernel-trace --stats
done
echo done >> $OUT/status.txt
