#!/bin/bash
# A/B of the image layout at zkEVM size, same box, same session: the wide extended sections and the resident constants tile-major
# (default) against row-major (MI_STARK_TILED_EXT=0 MI_STARK_TILED_CONSTS=0: the layout of the start of round 5), three proofs each (the
# last one is reported), then a soak of eight proofs with the default.
#   gpurun -- 'bash tools/tiled_ext_ab.sh > gpurun_out/r05_tiled_ext_ab.txt 2>&1'
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
show() { python3 - "$1" "$2" <<'P'
import json, sys
o = json.load(open(sys.argv[2]))
ph = o["phase_ms"]
keep = ("STARK_STEP_1", "STARK_STEP_2_CALCULATE_EXPS", "STARK_STEP_3_CALCULATE_EXPS", "STARK_STEP_3_CALCULATE_EXPS_2", "STARK_STEP_4_INIT", "STARK_STEP_2_LDE_AND_MERKLETREE", "STARK_STEP_3_LDE_AND_MERKLETREE", "STARK_STEP_4_CALCULATE_EXPS_2NS", "STARK_STEP_5_EVMAP", "STARK_STEP_5_CALCULATE_EXPS")
print("%-28s genproof %.1f ms  peak HBM %.1f GB  %s  zkin_sha256 %s" % (sys.argv[1], o["value"], o["hbm"]["peak_hbm_gb"], o["checks"], o.get("zkin_sha256", "")[:16]))
print("    " + "  ".join("%s %.1f" % (k.replace("STARK_STEP_", ""), ph[k]) for k in keep))
P
}
for rep in 1 2; do
    timeout -k 10 400 python3 bench_starks.py --proofs 3 > gpurun_out/ab_tiled_$rep.json 2> gpurun_out/ab_tiled_$rep.err || exit 1
    show "tile-major (run $rep)" gpurun_out/ab_tiled_$rep.json
    MI_STARK_TILED_EXT=0 MI_STARK_TILED_CONSTS=0 timeout -k 10 400 python3 bench_starks.py --proofs 3 > gpurun_out/ab_rowmajor_$rep.json 2> gpurun_out/ab_rowmajor_$rep.err || exit 1
    show "row-major (run $rep)" gpurun_out/ab_rowmajor_$rep.json
done
timeout -k 10 400 python3 bench_starks.py --proofs 8 --check-rows 2 > gpurun_out/r05_starks_soak_n1.json 2> gpurun_out/soak.err || exit 1
show "soak, 8 proofs" gpurun_out/r05_starks_soak_n1.json
