#!/bin/bash
# A/B of the stage-1 upload out of a PAGE-LOCKED host trace: every k-th chunk read in place by the DMA engines (strided 2-D copy) beside
# the host-packed ones (MI_UPLOAD_STRIDED_EVERY = k; 0 = all packed, the r04 form).  bench.py's PCIe-inclusive leg at full size.
#   gpurun -- 'bash tools/upload_mix_ab.sh > gpurun_out/r05_upload_mix_ab.txt 2>&1'
cd "${GRAFT_REPO_ROOT:-.}"
for k in 0 4 3 2 0 4; do
    MI_UPLOAD_STRIDED_EVERY=$k timeout -k 10 200 python3 bench.py --steps 1 --warmup 1 --no-verify --no-cpu-baseline --no-genproof --pcie-steps 3 2> /dev/null | python3 -c "
import json, sys
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
p = j['pcie_inclusive']
print('MI_UPLOAD_STRIDED_EVERY=$k  pcie-inclusive %.1f ms per step (%.2f G elements/s), resident %.1f ms, root matches: %s' % (p['ms_per_step'], p['value'] / 1e9, j['ms_per_step'], p['root_matches']))"
done
