#!/bin/bash
# h2d_probe.sh -- tools/h2d_probe.hip under the runtime's copy-engine switches; for each: the probe's own timings and, from a
# `rocprofv3 --kernel-trace --stats` run, whether the upload shows up as `__amd_rocclr_copyBuffer` blit kernels.
#   gpurun -- 'bash tools/h2d_probe.sh > gpurun_out/h2d_probe.txt 2>&1'
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/h2d_probe
mkdir -p "$OUT"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/h2d_probe.hip -o "$OUT/h2d_probe" || exit 1
echo "== host: $(nproc) threads, $(free -g | awk '/Mem:/ {print $2}') GiB RAM ($(free -g | awk '/Mem:/ {print $7}') available)"
grep -m1 'model name' /proc/cpuinfo
cat /sys/fs/cgroup/cpu.max 2>/dev/null
cat /sys/fs/cgroup/memory.max 2>/dev/null
run() { # name, env assignments...
    local name=$1; shift
    echo
    echo "== $name: $*"
    env "$@" "$OUT/h2d_probe" 4
    # the profiler run: the program itself after `--` (no env/bash hop), switches exported for this subshell only
    ( for kv in "$@"; do export "$kv"; done
      cd /tmp && export TMPDIR=/tmp && timeout -k 10 120 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/$name" -o p -- "$OLDPWD/$OUT/h2d_probe" 2 > /dev/null 2>&1 )
    local f; f=$(find "$OUT/$name" -name '*kernel_stats.csv' | head -1)
    if [ -n "$f" ]; then grep -E 'copyBuffer|k_valu|Name' "$f" | cut -d, -f1-4; else echo "(no kernel stats)"; fi
}
run default MI_DUMMY=1
run blit_engine_dma GPU_BLIT_ENGINE_TYPE=2
run blit_engine_kernel GPU_BLIT_ENGINE_TYPE=3
run force_blit_0 GPU_FORCE_BLIT_COPY_SIZE=0
run limit_blit_wg DEBUG_CLR_LIMIT_BLIT_WG=16
run hsa_sdma_off HSA_ENABLE_SDMA=0
echo
echo "== AMD_LOG_LEVEL=4: the copy path the runtime logs for one upload"
AMD_LOG_LEVEL=4 "$OUT/h2d_probe" 1 2>&1 | grep -i -E "HSA Copy|blit|staging|sdma" | sort | uniq -c | sort -rn | head -12
