#!/bin/bash
# LDE diagnosis: the NTT passes of bench.py's default workload under rocprofv3 PMC passes, with the shipped
# library and with the MI_NTT_NO_ARITH build (ab_libs/libmi_stark_noarith.so: same loads, LDS round trips, barriers and stores,
# no field arithmetic).  Output: gpurun_out/pmc_ntt/<lib>_<pass>/...counter_collection.csv + kernel stats; summarised by
# tools/pmc_ntt_table.py into profiles/r03_pmc_ntt.txt.  Run on the GPU box from the repo root.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/pmc_ntt
mkdir -p $OUT
# the A/B library is scratch (ab_libs/ is not tracked): build it when it is not there
[ -f ab_libs/libmi_stark_noarith.so ] || make -C merlin-zkevm-prover_amd/csrc ab-noarith > $OUT/ab_build.log 2>&1 || { echo "ab-noarith build failed" >> $OUT/status.txt; exit 1; }
BENCH="python3 bench.py --steps 2 --warmup 1 --no-verify --no-cpu-baseline --pcie-steps 0"
python3 -c "import torch; f,t=torch.cuda.mem_get_info(); print('hbm free/total bytes', f, t)" > $OUT/meminfo.txt 2>&1
rocprofv3 -L > $OUT/counters_list.txt 2>&1
run() { # name, lib, rocprof args...
    local name=$1 lib=$2; shift 2
    if [ -n "$lib" ]; then export MI_STARK_LIB=$PWD/$lib; else unset MI_STARK_LIB; fi
    timeout -k 10 240 rocprofv3 "$@" -d $OUT/$name -o $name --output-format csv -- $BENCH > $OUT/$name.log 2>&1
    local rc=$?
    echo "$name rc=$rc" >> $OUT/status.txt
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout in $name: stopping" >> $OUT/status.txt; exit 1; fi
}
for v in arith noarith; do
    lib=""; [ $v = noarith ] && lib=ab_libs/libmi_stark_noarith.so
    run ${v}_trace "$lib" --kernel-trace --stats
    run ${v}_sq1 "$lib" --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE
    run ${v}_sq2 "$lib" --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU
    run ${v}_tcc1 "$lib" --pmc FETCH_SIZE TCC_EA0_RDREQ_sum
    run ${v}_tcc2 "$lib" --pmc WRITE_SIZE TCC_EA0_WRREQ_sum
done
find $OUT -name "*.csv" | head -50 > $OUT/files.txt
echo done >> $OUT/status.txt
