#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path on MI355X.

Workload (BASELINE.json configs[2], the configuration the metric "field-elements/s through NTT+Merkleize at
2^23 rows" is quoted on; it fits one GPU): synthetic 2^23-row x 665-column Goldilocks trace, already resident
in HBM -> LDE to 2^24 rows (blow-up 2) -> Poseidon Merkle tree of the extended trace.  One "step" = one full
pass over that trace.  value = trace field elements (rows x cols of the INPUT) per second, whole job.

N > 1 (one process per GPU, torch.distributed over RCCL): the columns are dealt to the ranks in rounds of equal-width
tiles (665 columns over 8 ranks: 32 + 32 + 24 per rank); each round a rank extends one tile, ships every peer its rows
of it (batched isend/irecv) and absorbs the previous round's columns of its own rows into the leaf sponges, so exchange
and arithmetic overlap; subtree roots are all-gathered (merlin-zkevm-prover_amd/shard.py).  Total work is fixed ->
"scaling": "strong".  `python bench.py --gpus N` starts the N ranks itself (torch.distributed.run from a parent that
never touches a GPU); under an external launcher (WORLD_SIZE set) it is one of the ranks.

At N = 1 the result of the last step is verified against the CPU oracle at full size after the timed region
(tests/verify_full.py: whole sampled columns, a random combination of all columns at all rows, sampled leaves, every tree
level above the leaf digests) -> "root_verified_against_oracle".

Prints ONE JSON line (rank 0) with the driver's contract fields plus "roofline" and "cpu_baseline".
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "merlin-zkevm-prover_amd"))

# Regression constant only: the Merkle root of the default workload (2^23 x 665, seed 0x5EED0003).  It is what the
# oracle-verified N = 1 run (root_verified_against_oracle, tests/test_gpu_fullsize.py) produces; N > 1 runs, which do not
# re-run the oracle checks, must reproduce it.
ROOT_2P23_X665 = [17877856175459861405, 3297257765296605804, 13052643778398375791, 11912701812281293778]

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s peak (6.3 TB/s achievable)
VALU_LANE_OPS_PER_S = 256 * 4 * 32 * 2.4e9   # 256 CUs x 4 SIMD-32 x 2.4 GHz = 78.6 T full-rate lane-ops/s


def valu_roofline(perms, leaf_ms, pmc):
    """Issue-rate roofline of the Poseidon leaf kernel (the kernel is VALU-bound, not HBM-bound).
    Instructions per permutation: MEASURED (rocprofv3 --pmc SQ_INSTS_VALU over the launch / wave-permutations,
    profiles/r03_pmc_leaf.json).  The kernel's loop-weighted static mix of the two issue classes comes from
    tools/valu_mix.py (2-clk: v_mov_b32 and plain 32-bit add/sub/logic; 4-clk: everything else, v_cndmask and compares
    included -- profiles/r02_ubench_int3_issue_rates.txt).  peak = 1 / (f2 / R2 + f4 / R4) wave-instructions per second for
    the whole chip, quoted at the nominal 2.4 GHz clock (R = 1024 SIMDs x 2.4 GHz / 2 resp. 4) -- the PMC capture puts
    the kernel's effective clock at 2.37 GHz -- and, beside it, at the class rates the microbenchmarks reach."""
    instrs = float(pmc["valu_instructions_per_permutation"])
    f2, f4 = pmc["issue_mix"]["frac_2clk"], pmc["issue_mix"]["frac_4clk"]
    ach = perms / (leaf_ms * 1e-3) if leaf_ms > 0 else 0.0
    out = {"kernel": "k_linear_hash_rows_lines", "bound": "valu-issue", "perms_per_launch": perms, "achieved": ach,
           "unit": "permutations/s", "valu_instructions_per_permutation": instrs, "issue_mix": {"frac_2clk": f2, "frac_4clk": f4},
           "achieved_wave_instr_per_s": ach / 64.0 * instrs, "effective_clock_ghz_pmc": pmc.get("effective_clock_ghz"),
           "wave_time_split_pmc": pmc.get("wave_time_split"),
           "source": "profiles/r0x_pmc_leaf.json, the newest (SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, SQ_WAIT_*, GRBM_GUI_ACTIVE: tools/pmc_leaf.sh), tools/valu_mix.py, tools/ubench_int3.hip"}
    # what the ALGORITHM needs (tools/valu_floor.py: 472 modular multiplies at 4 partial products + a 5-instruction reduction, 8 full-round MDS
    # layers on 32-bit halves, the grouped partial rounds' 737 dot terms + 44 closings; no data movement) against what the compiler emitted
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import valu_floor
    fl = valu_floor.poseidon_floor()
    out["algorithmic_floor_instr"] = fl["floor_instr_per_permutation"]
    out["frac_of_floor"] = fl["floor_instr_per_permutation"] / instrs
    out["floor_breakdown"] = {k: fl[k] for k in ("modular_multiplies", "sbox_instr", "full_round_mds_instr", "partial_round_linear_instr", "constant_adds_instr")}
    out["floor_note"] = ("floor = instructions one permutation NEEDS if every 32x32 partial product is one v_mad_u64_u32, a 128-bit reduction is 5 instructions and no "
                         "register-pair data movement exists (tools/valu_floor.py); frac_of_floor = floor / measured count; the kernel's time at the floor count and the "
                         "same issue rate would be avg_launch_ms * frac_of_floor")
    for key, rates in pmc["issue_rates_wave_instr_per_s"].items():
        blended = 1.0 / (f2 / rates["2clk"] + f4 / rates["4clk"])
        peak_perms = blended * 64.0 / instrs
        if key.startswith("nominal"):
            out.update({"peak": peak_perms, "frac": ach / peak_perms, "peak_wave_instr_per_s": blended, "peak_basis": key})
        else:
            out["vs_microbenchmark_rates"] = {"peak": peak_perms, "frac": ach / peak_perms, "peak_wave_instr_per_s": blended,
                                              "note": "the dependent-chain microbenchmarks reach 92 % / 76 % of the nominal class rates; the kernel's mixed stream does better"}
    return out


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


T_PROCESS0 = time.perf_counter()


def run_child(cmd, timeout_s, env=None):
    """A child leg under a HARD wall limit: its own process group, killed as a group at the limit, and never waited for longer than a few
    seconds after that (a process stuck in the driver does not die on SIGKILL until the GPU lets go of it: the contract line must not wait
    for it).  -> (returncode or None, stdout, stderr, note)"""
    import signal
    import subprocess
    import tempfile
    if timeout_s < 5:
        return None, "", "", "skipped: %.0f s left of the leg budget" % timeout_s
    fo, fe = tempfile.TemporaryFile(mode="w+"), tempfile.TemporaryFile(mode="w+")   # (files, not pipes: nothing to drain while polling)
    p = subprocess.Popen(cmd, stdout=fo, stderr=fe, env=env, start_new_session=True)
    t0 = time.perf_counter()
    note = ""
    while p.poll() is None:
        if time.perf_counter() - t0 > timeout_s:
            note = "killed at its %.0f s limit" % timeout_s
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except OSError:
                pass
            t_k = time.perf_counter()
            while p.poll() is None and time.perf_counter() - t_k < 10:
                time.sleep(0.2)
            if p.poll() is None:
                note += " and still not gone 10 s later (abandoned)"
            break
        time.sleep(0.2)
    fo.seek(0), fe.seek(0)
    return p.returncode, fo.read(), fe.read(), note


def cpu_baseline(log_n, ncols, trace_host=None):
    """CPU baseline on this node's host cores, on a bounded sample of the same workload (LDE + Merkle tree of a 2^log_n x ncols trace).
    The reference's CPU path (src/goldilocks: AVX2 + OpenMP) is an absent submodule, so what is timed is this repo's restatement of
    it, labelled as such:
      value           oracle/cpu_baseline_avx2.c -- hand-vectorised (4 sponges per __m256i, MDS as 32-bit multiply-adds, vector butterflies on
                      cache-sized column blocks), OpenMP; itself checked bit for bit against the checker (tests/test_cpu_baseline.py);
      naive           oracle/gl_oracle.c built -O3 -mavx2 -- the CHECKER's scalar code (30 x 144 128-bit multiply-adds per permutation, a
                      layer-by-layer NTT over the whole matrix), on a sample 4x smaller: kept for continuity with rounds 1-2, not a baseline
                      anyone should quote.
    Phase names mirror the reference's timers (starks.cpp:50-57)."""
    share = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:                                                    # a container's CPU quota (the GPU box: 16 cores per GPU) is not in the affinity mask
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            share = min(share, max(1, -(-int(q) // int(per))))
    except (OSError, ValueError):
        pass
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes
    import glo
    u64 = ctypes.c_uint64
    threads = max(1, min(share, 64))

    def synthetic(n):                                       # glo.splitmix64 in pieces (its temporaries are several times the result)
        out = np.empty(n * ncols, dtype=np.uint64)
        piece = 1 << 26
        for o in range(0, n * ncols, piece):
            k = min(piece, n * ncols - o)
            idx = np.arange(o + 1, o + k + 1, dtype=np.uint64)
            with np.errstate(over="ignore"):
                z = np.uint64(0x5EED0003) + idx * np.uint64(0x9E3779B97F4A7C15)
                z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
                z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
                z = z ^ (z >> np.uint64(31))
            out[o:o + k] = np.where(z >= np.uint64(glo.P), z - np.uint64(glo.P), z)
        return out.reshape(n, ncols)

    def run(L, extend, tree, n, label):
        n_ext = 2 * n
        # the benchmark's own trace when the caller still holds it on the host (the PCIe leg's copy: same seed, same values), else generated
        trace = trace_host if (trace_host is not None and trace_host.shape == (n, ncols)) else synthetic(n)
        L.glo_set_num_threads(threads)
        cores = int(L.glo_num_threads())
        t0 = time.perf_counter()
        ext = extend(trace, n_ext, n)
        t1 = time.perf_counter()
        nodes = tree(ext, n_ext)
        t2 = time.perf_counter()
        L.glo_set_num_threads(1)                           # 1 thread: 32 of the columns over at most 2^20 rows, tree over 2^13 extended rows
        c1, h1, n1 = min(32, ncols), min(n_ext, 1 << 13), min(n, 1 << 20)
        sub = np.ascontiguousarray(trace[:n1, :c1])
        u0 = time.perf_counter()
        extend(sub, 2 * n1, n1)
        u1 = time.perf_counter()
        tree(np.ascontiguousarray(ext[:h1]), h1)
        u2 = time.perf_counter()
        L.glo_set_num_threads(cores)
        lde_1t, mrk_1t = n1 * c1 / (u1 - u0), (h1 / 2) * ncols / (u2 - u1)
        perms = n_ext * ((ncols + 7) // 8) + n_ext - 1
        return {"value": n * ncols / (t2 - t0), "unit": "field-elements/s", "cores": cores, "kind": "port", "label": label,
                "sample": f"2^{n.bit_length() - 1} rows x {ncols} cols -> LDE 2^{n.bit_length()} + Poseidon Merkle tree, {cores} threads; "
                          f"STARK_STEP_1_LDE {t1 - t0:.2f} s, STARK_STEP_1_MERKLETREE {t2 - t1:.2f} s",
                "phase_s": {"STARK_STEP_1_LDE": t1 - t0, "STARK_STEP_1_MERKLETREE": t2 - t1},
                "permutations_per_s_per_thread": perms / (t2 - t1) / cores,
                "one_thread": {"value": 1.0 / (1.0 / lde_1t + 1.0 / mrk_1t), "unit": "field-elements/s", "lde_elements_per_s": lde_1t,
                               "merkle_elements_per_s": mrk_1t, "permutations_per_s": (h1 * ((ncols + 7) // 8) + h1 - 1) / (u2 - u1)},
                "root": [int(v) for v in nodes[-4:]]}

    Lb = glo.lib("baseline")

    def extend_fast(trace, n_ext, n):
        c = trace.shape[1]
        out = np.empty(n_ext * c, dtype=np.uint64)
        Lb.glb_extend_pol(glo.ptr(out), glo.ptr(np.ascontiguousarray(trace).reshape(-1)), u64(n_ext), u64(n), u64(c))
        return out.reshape(n_ext, c)

    def tree_fast(ext, rows):
        nodes = np.zeros((2 * rows - 1) * 4, dtype=np.uint64)
        Lb.glb_merkletree(glo.ptr(nodes), glo.ptr(np.ascontiguousarray(ext).reshape(-1)), u64(ext.shape[1]), u64(rows))
        return nodes

    out = run(Lb, extend_fast, tree_fast, 1 << log_n,
              "restatement, not upstream (oracle/cpu_baseline_avx2.c: hand-vectorised AVX2 + OpenMP; the reference's src/goldilocks library is absent)")
    out["cpu_model"], out["nproc"], out["omp_threads"], out["cgroup_cpu_share"] = cpu_model(), os.cpu_count(), out["cores"], share
    Ln = glo.lib("avx2")
    out["naive"] = run(Ln, lambda t, ne, n: glo.extend_pol(t, ne, n, t.shape[1], flavour="avx2"),
                       lambda e, rows: glo.merkletree(e, e.shape[1], rows, flavour="avx2"), 1 << max(min(log_n - 2, 16), 8),
                       "the CHECKER's scalar restatement (oracle/gl_oracle.c, gcc -O3 -mavx2 -fopenmp): not an optimised CPU path")
    return out


def proof_kernel_rooflines(stats_csv, proofs_in_profile, zk):
    """Roofline entries for the dominant kernels of ONE Starks::genProof at the zkEVM shape: average time per proof from the committed
    rocprofv3 --kernel-trace --stats summary of the same command (bench_starks.py; a profiler cannot run inside this process), algorithmic
    bytes per proof from the shape (Merkle 8 h w + 32 (2h - 1), LDE 8 N c + 8 Next c, NTT 16 n c; a constraint program
    reads each declared section once per row).  zk: the shape's sizes."""
    import csv
    import re
    N, NE = zk["n"], zk["n_ext"]
    w = zk["widths"]                                        # cm1, cm2, cm3, cm4_2ns, tmpExp, constants
    fam = {"leaf_hashing": ("k_linear_hash_rows_lines",), "ntt_passes": ("k_ntt_pass", "k_lde_mid"), "constraint_kernels": ("chelpers_chunk",),
           "operand_transpose": ("k_chp_transpose",), "linear_kernel": ("k_chp_linear",), "evmap": ("k_evmap_partial", "k_evmap_partial_tiled", "k_evmap_reduce")}
    ns = {k: 0.0 for k in fam}
    calls = {k: 0 for k in fam}
    for row in csv.DictReader(open(stats_csv)):
        for k, pats in fam.items():
            # the demangled name without its return type and namespaces, up to the template / argument list: matched EXACTLY (a prefix match
            # would count k_ntt_pass_pers -- the opt-in persistent pass -- in k_ntt_pass's family)
            base = re.split(r"[<(]", row["Name"].removeprefix("void ").strip(), maxsplit=1)[0].split("::")[-1].strip()
            if base in pats:
                ns[k] += float(row["TotalDurationNs"])
                calls[k] += int(row["Calls"])
    committed = [w["cm1"], w["cm2"], w["cm3"], w["cm4"]]
    ext_all = w["cm1"] + w["cm2"] + w["cm3"]
    alg = {
        "leaf_hashing": sum((8.0 * c + 32.0) * NE for c in committed),
        "ntt_passes": (8.0 * N + 8.0 * NE) * (ext_all + w["const"]) + 16.0 * NE * 3 + 16.0 * NE * w["cm4"] + 2 * 16.0 * N * 3,
        # step2prev / step3prev / step3 over N rows (cm1_n .. tmpExp_n + constants), step42ns over NE rows (cm1..3_2ns + constants); stores counted once
        "constraint_kernels": 3 * 8.0 * N * (ext_all + w["tmpexp"] + w["const"]) + 8.0 * NE * (ext_all + w["const"]) + 24.0 * NE,
        "operand_transpose": 0.0,
        "linear_kernel": 8.0 * NE * (ext_all + w["cm4"] + w["const"]) + 72.0 * NE,
        "evmap": 8.0 * N * zk["n_evals"],
    }
    bound = {"leaf_hashing": "hbm (VALU-issue-bound in fact: see valu)", "ntt_passes": "hbm", "constraint_kernels": "hbm (co-limited by integer issue)",
             "operand_transpose": "hbm: pure data movement the algorithm does not ask for (tile-major copies, per batch of rows, of the operands that are still row-major: the computed base-domain sections and the extended constants; the witness is written tile-major ONCE, behind its upload, by the same kernel -- 22 of these launches --, the wide extended sections by the leaf kernel while it absorbs them)", "linear_kernel": "hbm",
             "evmap": "hbm"}
    out = {}
    for k in fam:
        ms = ns[k] / 1e6 / max(proofs_in_profile, 1)
        if ms <= 0:
            continue
        ach = alg[k] / (ms * 1e-3) / 1e9
        peak = HBM_PEAK_GBS
        out[k] = {"kernels": list(fam[k]), "bound": bound[k], "ms_per_proof": ms, "launches_per_proof": calls[k] / max(proofs_in_profile, 1),
                  "algorithmic_bytes_per_proof": alg[k], "achieved": ach, "peak": peak, "unit": "GB/s", "frac": ach / peak}
    return out


def shape_is_zkevm_default(workload):
    """the committed kernel profile is of bench_starks.py's default (zkEVM-shaped, 2^23 rows) proof: rooflines only for that workload"""
    return workload.startswith("2^23 rows, sections [665, 128, 371]")


def genproof_leg(shape):
    """BASELINE's 'batch-proof wall time' and north_star's NTT + Merkle + FRI under the SAME command the driver times: after this process has
    released its HBM, a child runs bench_starks.py -- Starks::genProof of the product class (host/starks.hpp) over the synthetic zkEVM-shaped
    STARK at full size, then (shape 'batch') c12a's and recursive1's shapes, as genBatchProof's three calls (prover.cpp:541,577,611) -- and its
    JSON is merged into this line.  A child: one proof's plan is 273 of the device's 309 GB, and the arena is a process-wide singleton."""
    cmd = [sys.executable, os.path.join(ROOT, "bench_starks.py"), "--shape", shape, "--proofs", "2", "--check-rows", "4"]
    # the whole default command must print its line inside MI_BENCH_WALL_BUDGET_S (default 560 s) of its start: the child gets what is left
    limit = min(400.0, float(os.environ.get("MI_BENCH_WALL_BUDGET_S", "560")) - (time.perf_counter() - T_PROCESS0) - 5.0)
    t0 = time.perf_counter()
    rc, so, se, note = run_child(cmd, limit)
    wall = time.perf_counter() - t0
    if rc != 0:
        return {"error": "bench_starks.py: %s" % (note or "exited %s" % rc), "timeout_s": limit, "stderr_tail": se[-1500:]}
    j = json.loads(so.strip().splitlines()[-1])
    zk = j["starks"]["zkevm"] if "starks" in j else {"genproof_ms": j["value"], "phase_ms": j["phase_ms"], "hbm": j["hbm"], "checks": j["checks"],
                                                     "genproof_wall_ms": j["genproof_wall_ms"], "setup_s": j["setup_s"], "workload": j["config"]["workload"],
                                                     "field_ops_per_row": j["config"]["field_ops_per_row"], "flow": j["flow"], "peak_hbm_gb": j["hbm"]["peak_hbm_gb"]}
    ph = zk["phase_ms"]
    out = {"metric": "Starks::genProof wall time, one MI355X, synthetic zkEVM-shaped STARK at full size (BASELINE config 4 substitute; its inputs are absent from the reference tree)",
           "ms": zk["genproof_ms"], "unit": "ms", "flow": zk["flow"], "workload": zk["workload"], "field_ops_per_row": zk["field_ops_per_row"],
           "genproof_wall_ms_all_runs": zk["genproof_wall_ms"], "phase_ms": ph, "fri_ms": ph.get("STARK_STEP_FRI"),
           "ntt_merkle_fri_ms": {"lde_and_merkle": sum(ph.get(k, 0.0) for k in ("STARK_STEP_1_LDE_AND_MERKLETREE", "STARK_STEP_2_LDE_AND_MERKLETREE",
                                                                                  "STARK_STEP_3_LDE_AND_MERKLETREE", "STARK_STEP_4_MERKLETREE")),
                                 "fri": ph.get("STARK_STEP_FRI")},
           "peak_hbm_gb": zk["peak_hbm_gb"], "hbm": zk["hbm"], "checks": zk["checks"], "setup_s": zk["setup_s"], "child_wall_s": wall,
           "checks_note": "after the clock: openings climb to their roots at the replayed transcript's indices, folds land on the next layer, sampled q_2ns / f_2ns rows "
                          "equal the oracle interpreters'; whole-proof BYTE parity with an oracle prover is tests/test_genproof_parity.py (2^10 .. 2^14 rows)"}
    if "starks" in j:
        out["batch_ms"] = j["value"]
        out["batch"] = {sh: {"genproof_ms": o["genproof_ms"], "workload": o["workload"], "peak_hbm_gb": o["peak_hbm_gb"], "checks": o["checks"],
                             "fri_ms": o["phase_ms"].get("STARK_STEP_FRI")} for sh, o in j["starks"].items()}
        out["batch_note"] = "the three Starks::genProof calls of genBatchProof (zkEVM, c12a, recursive1 shapes) in one process sharing the HBM arena; c12a's starkStruct is assumed (DESIGN.md)"
    # per-kernel rooflines of the proof: from the NEWEST committed rocprofv3 summary of the same command (a profiler cannot run inside this
    # process); stamped with the file and its round so that a reader sees how old the kernel times are -- the proof's own wall time above is live
    import glob
    import re as _re
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_starks_genproof_kernel_stats.csv")))
    if cands and shape_is_zkevm_default(zk["workload"]):
        stats = cands[-1]
        out["kernel_rooflines"] = proof_kernel_rooflines(stats, 2, {"n": 1 << 23, "n_ext": 1 << 24, "n_evals": 1768,
                                                                    "widths": {"cm1": 665, "cm2": 128, "cm3": 371, "cm4": 6, "tmpexp": 265, "const": 218}})
        rnd = _re.search(r"r(\d\d)_", os.path.basename(stats)).group(1)
        out["kernel_rooflines_source"] = {"file": "profiles/" + os.path.basename(stats), "recorded_in_round": int(rnd),
                                          "command": "rocprofv3 --kernel-trace --stats -- python3 bench_starks.py --proofs 2 --check-rows 0 (tools/prof_starks.sh): the same command as this leg's zkEVM proof, two proofs, per-proof averages",
                                          "note": "kernel TIMES are the profile's, not this run's (compare ms above with the profile's own proof time); algorithmic bytes follow the shape"}
        cbw = os.path.join(ROOT, "profiles", "r%s_copybuffer_where.json" % rnd)
        if os.path.exists(cbw):
            cb = json.load(open(cbw))
            out["copybuffer_launches"] = {"what": "`__amd_rocclr_copyBuffer`, second line of that profile (%d launches, %.0f ms over the run): the runtime's staging copies of bench_starks.py's INPUT GENERATION "
                                                  "(the synthetic witness is made on the device and copied to pageable host memory; the constants' upload in the Starks constructor) -- %d launches / %.0f ms fall before the "
                                                  "first proof, %.1f ms inside the proofs.  Stage 1's upload is page-locked staging moved by the SDMA engines (tools/h2d_probe.hip: 57 GB/s, no kernel), not on the CUs."
                                                  % (cb["copyBuffer_launches"], cb["copyBuffer_ms_total"], cb["before_the_first_proof"]["launches"], cb["before_the_first_proof"]["ms"],
                                                     sum(x["ms"] for x in cb["inside_proofs"])),
                                          "source": "profiles/" + os.path.basename(cbw) + " (tools/copybuffer_where.py over the kernel trace)"}
    return out


def single_process_main(args):
    """`--single-process`: the N-shard step from ONE process through the C ABI (csrc/multi.hip: mi_multi_commit) -- the form a one-process
    Prover links -- instead of N torch.distributed ranks.  Shard g runs on device g (or, with --logical-shards, all on device 0: how a
    one-GPU box rehearses the path).  Timed with the trace resident in HBM on device 0 (tiles cross xGMI to their shards); the
    PCIe-inclusive variant (trace in pageable host memory, every shard's tiles over its own PCIe link) is reported beside it."""
    import torch
    import mi_stark
    G = args.gpus
    have = torch.cuda.device_count()
    devices = [0] * G if args.logical_shards else list(range(G))
    if not args.logical_shards and have < G:
        sys.stderr.write(f"bench.py: --gpus {G} --single-process needs {G} GPUs, this node shows {have} (add --logical-shards to rehearse on one)\n")
        return 2
    n, n_ext, ncols = 1 << args.log_n, 2 << args.log_n, args.cols
    ctx = mi_stark.Context(0)
    trace = ctx.empty(n * ncols)
    ctx.fill_synthetic_2d(trace, n, ncols, ncols, 0, 0x5EED0003)
    ctx.sync()
    m = mi_stark.Multi(devices)

    def run(src_ptr, src_device, steps, warmup):
        root, st = None, None
        for it in range(warmup + steps):
            if it == warmup:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            t = m.commit(src_ptr, n, n_ext, ncols, src_device=src_device)
            root = [int(v) for v in t.root]
            st = m.last_stats()
            t.free()
        return (time.perf_counter() - t0) / max(steps, 1), root, st
    dt, root, st = run(trace.data_ptr(), 0, args.steps, args.warmup)
    pcie = None
    if args.pcie_steps > 0:
        # the host trace page-locked (what Starks does with the witness section of pAddress when several devices are configured): each
        # shard's DMA engines read its tiles in place -- "strided" --, then the same buffer through the host-packed form (what a pageable
        # trace takes) for the A/B
        host = torch.empty(n * ncols, dtype=torch.int64, pin_memory=not args.pcie_pageable)
        host.copy_(trace)
        torch.cuda.synchronize()
        legs = {}
        for name, mode in (("auto", 1 if (args.logical_shards and not args.pcie_pageable) else -1), ("packed", 0)):  # (logical shards share a link: auto would pack)
            m.set_upload_mode(mode)
            dt_h, root_h, st_h = run(host.data_ptr(), -1, args.pcie_steps, 1)
            legs[name] = {"upload": m.last_upload_mode(), "ms_per_step": 1e3 * dt_h, "value": n * ncols / dt_h, "unit": "field-elements/s", "root_matches": root_h == root,
                          "per_shard": st_h["per_shard"]}
            if legs[name]["upload"] == "packed":
                break
        m.set_upload_mode(-1)
        pcie = dict(legs["auto"])
        pcie["host_memory"] = "pageable" if args.pcie_pageable else "page-locked"
        pcie["path"] = ("strided: every shard's DMA engines read its column tiles out of the page-locked trace over that shard's own PCIe link (no host thread in the data path)"
                        if pcie["upload"] == "strided" else "packed: every shard's tiles packed by host threads into page-locked staging and sent over that shard's own PCIe link")
        if "packed" in legs:
            pcie["packed_ab"] = {k: legs["packed"][k] for k in ("ms_per_step", "value", "root_matches")}
            pcie["packed_ab"]["host_pack_ms_per_shard"] = [s_["host_pack_ms"] for s_ in legs["packed"]["per_shard"]]
        del host
    out = {"metric": "goldilocks_field_elements_per_s_lde_merkleize_2^%d_rows" % args.log_n, "value": n * ncols / dt, "unit": "field-elements/s", "n_gpus": G,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt, "higher_is_better": True, "scaling": "strong" if G > 1 else "none", "vs_baseline": None,
           "dtype": "u64 (Goldilocks mod 2^64-2^32+1, 32-bit integer VALU)", "data": "synthetic",
           "config": {"workload": "BASELINE configs[2]: 2^%d-row x %d-col trace -> LDE blow-up 2 -> Poseidon Merkle tree" % (args.log_n, ncols), "rows": n, "cols": ncols,
                      "rows_ext": n_ext, "parallelism": "single process, %d shards on devices %s: column-tile LDE -> hipMemcpyPeerAsync exchange -> row-shard streaming Merkle" % (G, devices)},
           "root": root, "root_matches_regression_constant": (root == ROOT_2P23_X665) if (args.log_n == 23 and ncols == 665) else None,
           "comm": {"backend": "hipMemcpyPeerAsync / hipMemcpyAsync inside one process (csrc/multi.hip)", "shards": G, "devices": devices, "logical_shards_on_one_device": bool(args.logical_shards)},
           "peer_access": m.peer_access(),
           "per_rank": [{"rank": s["shard"], "device": s["device"], "lde_ms": s["lde_ms"], "absorb_ms": s["absorb_ms"], "exchange_wait_ms": s["exchange_wait_ms"],
                         "bytes_sent_to_peer": s["bytes_sent_to_shard"]} for s in st["per_shard"]],
           "pcie_inclusive": pcie, "value_pcie_inclusive": (pcie or {}).get("value"), "roofline": None, "cpu_baseline": None}
    print(json.dumps(out), flush=True)
    m.close()
    ctx.close()
    return 0


def launch_ranks(args):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks.  This parent never initialises a GPU
    (device_count() does not on this stack); the ranks are children of torch.distributed.run."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} needs {args.gpus} GPUs, this node shows {have}\n")
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log-n", type=int, default=23, help="log2 of trace rows (BASELINE: 23)")
    ap.add_argument("--cols", type=int, default=665, help="committed columns (BASELINE: 665)")
    ap.add_argument("--workspace-gib", type=float, default=32.0)
    ap.add_argument("--poseidon-variant", type=int, default=2)
    ap.add_argument("--cpu-log-n", type=int, default=23,
                    help="log2 rows of the CPU-baseline sample: 23 = the benchmark's own size (134 GB of host memory, about two minutes on 16 threads); "
                         "the checker's scalar code runs on 2^16 rows at most")
    ap.add_argument("--no-genproof", action="store_true",
                    help="N = 1: skip the batch-proof leg (bench_starks.py --shape batch in a child process after this process has released its HBM)")
    ap.add_argument("--genproof-shape", default="batch", help="batch (zkEVM + c12a + recursive1 shapes, as genBatchProof) or zkevm")
    ap.add_argument("--pcie-steps", type=int, default=2, help="N = 1: steps of the PCIe-inclusive leg (host trace streamed in by mi_lde_merkle_host); 0 = skip")
    ap.add_argument("--pack-threads", type=int, default=-1,
                    help="PCIe-inclusive leg: host threads packing column chunks into page-locked staging (0 = strided 2-D copies; "
                         "-1 = the library's default, min(16, hardware threads))")
    ap.add_argument("--pcie-pageable", action="store_true", help="PCIe-inclusive leg: the host trace in pageable memory (as a mapped file would be)")
    ap.add_argument("--no-verify", action="store_true", help="skip the full-size oracle verification after the timed region (N = 1)")
    ap.add_argument("--verify-sharded", action="store_true",
                    help="tile-by-tile / N > 1 path: gather the row-sharded extension and leaf digests on rank 0 and verify them against the oracle "
                         "(for sizes one GPU holds whole, e.g. --log-n 18: about half a minute of CPU)")
    ap.add_argument("--ntt-log-b", type=int, default=5)
    ap.add_argument("--lde-fuse", type=int, default=1, help="0 = separate last-INTT / first-NTT passes (A/B only)")
    ap.add_argument("--leaf-mode", type=int, default=1, help="1 = line-aligned leaf fetch (default), 0 = per-block loads")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-process-leg", action="store_true", help="N > 1: skip the extra leg that runs the same step once more from ONE process over the same devices")
    ap.add_argument("--force-exchange", action="store_true",
                    help="N = 1 only: run the N > 1 code path (tile-by-tile LDE, streaming leaf absorption, subtree) on a one-rank "
                         "communicator, to rehearse it at full size on a single GPU")
    ap.add_argument("--single-process", action="store_true",
                    help="N > 1 from ONE process through the C ABI (mi_multi_commit: what a one-process Prover links) instead of torch.distributed ranks")
    ap.add_argument("--logical-shards", action="store_true", help="--single-process: all shards on device 0 (rehearsal on a one-GPU box)")
    args = ap.parse_args()
    if args.single_process:
        sys.exit(single_process_main(args))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))

    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a version banner on
    # communicator creation), so keep a private handle on the real stdout and point fd 1 at stderr for everything else.
    real_stdout = os.fdopen(os.dup(1), "w")
    sys.stdout.flush()
    os.dup2(2, 1)

    import torch
    import mi_stark
    from shard import ShardPlan, lde_merkle_sharded

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world          # under a launcher the launcher's world size wins
    dist = None
    if world > 1 or args.force_exchange:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == world

    ctx = mi_stark.Context(local_rank, workspace_limit=int(args.workspace_gib * (1 << 30)))
    ctx.set_poseidon_variant(args.poseidon_variant)
    ctx.set_ntt_tile(args.ntt_log_b)
    ctx.set_lde_fuse(args.lde_fuse)
    ctx.set_leaf_mode(args.leaf_mode)
    n, n_ext, ncols = 1 << args.log_n, 2 << args.log_n, args.cols
    plan = ShardPlan(n=n, n_ext=n_ext, ncols=ncols, world=world, rank=rank)

    # ---- device-resident buffers (HBM layout: DESIGN.md "Data layout")
    exchange = world > 1 or args.force_exchange
    if exchange:
        # my tiles of the synthetic trace (tile t = columns [32 t, 32 t + 32), dealt round-robin), packed side by side
        trace = ctx.empty(n * max(plan.my_cols, 1))
        for k, (c0, w) in enumerate(plan.my_tile_cols()):
            ctx.fill_synthetic_2d(trace, n, w, ncols, c0, 0x5EED0003, out_pitch=plan.my_cols, out_off=plan.local_col(k))
        bufs = {"ext": ctx.empty(plan.ext_elems()), "recv": ctx.empty(plan.recv_elems()),
                "nodes": ctx.empty((2 * plan.rows_per_rank - 1) * 4), "roots": ctx.empty((2 * world - 1) * 4)}
    else:
        trace = ctx.empty(n * ncols)
        ctx.fill_synthetic_2d(trace, n, ncols, ncols, 0, 0x5EED0003)
        bufs = {"ext": ctx.empty(n_ext * ncols), "nodes": ctx.empty((2 * n_ext - 1) * 4)}

    # HIP-event timer slots: 0..2 the three phases of the single-GPU path; 8+k / 36+k the LDE / absorb of pipeline round k
    LDE_SLOT0, ABSORB_SLOT0 = 8, 36
    assert plan.n_rounds <= 28
    comm = {"backend": dist.get_backend(), "rccl_ranks": dist.get_world_size()} if dist is not None else None

    from shard import device_ops

    class Ops(device_ops(ctx)):
        @staticmethod
        def merkle_build(nodes, src, c, rows):
            # same two launches as mi_merkle_build_dev, split so the dominant kernel can be timed alone
            ctx.timer_start(1)
            ctx.linear_hash_rows(nodes, src, c, rows)
            ctx.timer_stop(1)
            ctx.timer_start(2)
            ctx.merkle_levels(nodes, rows)
            ctx.timer_stop(2)

    class OpsTimed(Ops):
        @staticmethod
        def lde(out, inp, ne, nn, c, out_pitch=None, in_pitch=None, out_off=0, in_off=0, chunk=0):
            slot = LDE_SLOT0 + chunk if exchange else 0
            ctx.timer_start(slot)
            ctx.lde(out, inp, ne, nn, c, out_pitch=out_pitch, in_pitch=in_pitch, out_off=out_off, in_off=in_off)
            ctx.timer_stop(slot)

        @staticmethod
        def absorb(digests, windows, nrows, first, final, chunk=0):
            ctx.timer_start(ABSORB_SLOT0 + chunk)
            ctx.linear_hash_absorb(digests, windows, nrows, first, final)
            ctx.timer_stop(ABSORB_SLOT0 + chunk)

        wait_events = []

        @staticmethod
        def wait_begin(k):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            OpsTimed.wait_events.append([e, None])

        @staticmethod
        def wait_end(k):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            OpsTimed.wait_events[-1][1] = e

        @staticmethod
        def merkle_levels(nodes, nleaves):
            if nleaves == plan.rows_per_rank:        # my subtree (the top levels over the G roots are not timed apart)
                ctx.timer_start(2)
                ctx.merkle_levels(nodes, nleaves)
                ctx.timer_stop(2)
            else:
                ctx.merkle_levels(nodes, nleaves)

    lde_rounds = [k for k in range(plan.n_rounds) if plan.width(k, rank)] if exchange else []
    absorb_rounds = list(range(plan.n_rounds)) if exchange else []

    t_lde = t_leaf = t_lvls = t_wait = 0.0

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    root = None
    for _ in range(args.warmup):
        root = lde_merkle_sharded(plan, Ops, dist, trace, bufs, always_exchange=exchange)
    import gc
    gc.collect()
    gc.disable()     # no collector pauses inside the timed region (an earlier script met a 35-55 ms one)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        root = lde_merkle_sharded(plan, OpsTimed, dist, trace, bufs, always_exchange=exchange)
        # HIP-event readings need the stop events complete; reading them after the loop would only see the
        # last step, so accumulate per step (the sync this implies is inside the timed region on purpose:
        # it costs microseconds against a ~1 s step and keeps per-kernel times honest)
        if exchange:
            t_lde += sum(ctx.timer_ms(LDE_SLOT0 + k) for k in lde_rounds)
            t_leaf += sum(ctx.timer_ms(ABSORB_SLOT0 + k) for k in absorb_rounds)
        else:
            t_lde += ctx.timer_ms(0)
            t_leaf += ctx.timer_ms(1)
        t_lvls += ctx.timer_ms(2)
        if exchange:
            torch.cuda.synchronize()
            t_wait += sum(a.elapsed_time(b) for a, b in OpsTimed.wait_events if b is not None)
            OpsTimed.wait_events.clear()
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=ctx.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    root_host = [int(v) for v in ctx.to_host(root)]

    # ---- per-rank breakdown (every N): a first run on real hardware must be able to say WHERE it went wrong or slow
    from shard import exchange_bytes_to_peers, gather_sharded_result
    K_ = max(args.steps, 1)
    mine = [t_lde / K_, t_leaf / K_, t_lvls / K_, t_wait / K_] + [float(b) for b in (exchange_bytes_to_peers(plan) if exchange else [0] * world)]
    per_rank = None
    if dist is not None and world > 1:
        tt = torch.tensor(mine, dtype=torch.float64, device=ctx.device)
        allt = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(allt, tt)
        rows_ = [[float(v) for v in t_.cpu()] for t_ in allt]
    else:
        rows_ = [mine]
    per_rank = [{"rank": r, "lde_ms": v[0], "absorb_ms": v[1], "levels_ms": v[2], "exchange_wait_ms": v[3], "bytes_sent_to_peer": [int(b) for b in v[4:]]}
                for r, v in enumerate(rows_)]
    sharded_verify = None
    if exchange and args.verify_sharded:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from verify_full import verify_gathered
        ext_full, dig_full = gather_sharded_result(plan, OpsTimed, dist if world > 1 else None, bufs)
        if rank == 0:
            sharded_verify = verify_gathered(ctx, ext_full, dig_full, root_host, n, n_ext, ncols, seed=0x5EED0003,
                                             log=lambda m: print(m, file=sys.stderr, flush=True))

    verify = None
    ext_windows = None
    if world == 1 and not args.no_verify:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from verify_full import verify_lde_merkle
        if exchange:
            ext_windows = [(bufs[name], off - plan.row0 * w, w, pitch) for (name, off, w, pitch) in plan.row_windows()]
        else:
            ext_windows = [(bufs["ext"], 0, ncols, ncols)]
        verify = verify_lde_merkle(ctx, [(trace, 0, ncols, ncols)], ext_windows, bufs["nodes"], n, n_ext, ncols,
                                   log=lambda m: print(m, file=sys.stderr, flush=True))
        assert verify["root"] == root_host

    # ---- PCIe-inclusive leg (never `value`): the same step with the trace in (page-locked) HOST memory, streamed up in column
    # chunks behind the kernels by mi_lde_merkle_host; the extension and the tree stay resident, the root comes back
    pcie = None
    host_keep = None
    if world == 1 and not exchange and args.pcie_steps > 0:
        try:
            t_a = time.perf_counter()
            host_trace = torch.empty(n * ncols, dtype=torch.int64, pin_memory=not args.pcie_pageable)
            t_b = time.perf_counter()
            host_trace.copy_(trace)
            torch.cuda.synchronize()
            t_c = time.perf_counter()
            ctx.set_host_pack_threads(args.pack_threads)
            pack_threads = ctx.host_pack_threads()                # the count in effect (the default resolves against this host's threads)
            ctx.lde_merkle_host(bufs["nodes"], bufs["ext"], host_trace.data_ptr(), n, n_ext, ncols)     # warm-up (staging buffers)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.pcie_steps):
                ctx.lde_merkle_host(bufs["nodes"], bufs["ext"], host_trace.data_ptr(), n, n_ext, ncols)
                root_p = ctx.to_host(bufs["nodes"][(2 * n_ext - 2) * 4:(2 * n_ext - 1) * 4])
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.pcie_steps
            # the upload alone, same chunking, for the achieved H2D rate
            import ctypes as _ct
            pcie = {"ms_per_step": 1e3 * dt, "value": n * ncols / dt, "unit": "field-elements/s", "steps": args.pcie_steps,
                    "ratio_to_device_resident_step": dt / (elapsed / max(args.steps, 1)),
                    "host_memory": "%s, %.1f GB, allocated in %.1f s; D2H fill %.1f GB/s" % ("pageable" if args.pcie_pageable else "page-locked (hipHostMalloc)", n * ncols * 8 / 1e9, t_b - t_a, n * ncols * 8 / 1e9 / (t_c - t_b)),
                    "h2d_bytes_per_step": n * ncols * 8, "h2d_floor_ms_at_55GBps": 1e3 * n * ncols * 8 / 55e9,
                    "root_matches": [int(v) for v in root_p] == root_host,
                    "host_pack_threads": pack_threads,
                    "path": ("mi_lde_merkle_host: 32-column chunks packed by %d host threads into page-locked staging, one contiguous H2D copy each, "
                             "|| LDE + leaf absorption of the chunk before; extension + tree stay in HBM" % pack_threads) if pack_threads else
                            "mi_lde_merkle_host: column chunks of 32, 64, 64 ... columns, H2D of chunk k+1 (two copy streams) || LDE + leaf absorption of chunk k; extension + tree stay in HBM",
                    "h2d_note": "strided 2-D copies out of the row-major host trace run at 39 / 49 / 53 GB/s for 32 / 64 / 128 columns against 57 GB/s contiguous "
                                "(profiles/r02_pcie_chunk_sweep.json): hence the host-side packing"}
            if args.cpu_log_n == args.log_n and not args.no_cpu_baseline:
                host_keep = host_trace          # the CPU baseline runs on this very trace (saves generating 44.6 GB a second time)
            del host_trace
        except Exception as e:  # a box without enough page-lockable host memory must not lose the headline number
            pcie = {"error": repr(e)}

    # ---- N > 1: the same step ONCE MORE from ONE process over the same N devices (csrc/multi.hip: mi_multi_commit, the form a one-process
    # Prover links; bench.py --single-process) -- so that a run on a real multi-GPU node measures both forms of the plan.  Never `value`.
    # The ranks give their buffers back and wait on the CPU (a gloo group: an RCCL barrier would spin on the GPUs the child is using); the
    # child has a hard time limit; whatever goes wrong here becomes an "error" string and nothing else.
    single_process = None
    if dist is not None and (world > 1 or args.force_exchange) and not args.no_single_process_leg:
        cpu_group = None
        try:
            import datetime
            import socket
            try:
                socket.gethostbyname(socket.gethostname())
            except Exception:   # a container whose hostname does not resolve: gloo finds its interface by that name; this is one node, loopback will do
                os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            cpu_group = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=900))
        except Exception as e:
            single_process = {"error": "gloo group: " + repr(e)}
        if cpu_group is not None:
            try:
                trace = root = None
                ext_windows = None   # (a one-rank rehearsal verified the result above: its window list holds the extension)
                bufs.clear()
                gc.collect()
                torch.cuda.empty_cache()
                ctx.close()      # the library's own workspace and staging too (the proof below plans 284 of device 0's 309 GB)
                ctx = None
                torch.cuda.synchronize()
                dist.barrier(group=cpu_group)
            except Exception as e:
                single_process = {"error": "before the child: " + repr(e)}
            if rank == 0 and single_process is None:
                # Both legs together get MI_BENCH_LEG_BUDGET_S (default 240 s; the commit at most 40 % of it) of wall time, each child what is left of it, killed as a process
                # group at its limit and never waited for beyond that (run_child): the contract line below is printed inside the driver's
                # limit whatever these never-yet-run-on-hardware paths do on first contact.
                leg_budget = float(os.environ.get("MI_BENCH_LEG_BUDGET_S", "240"))
                t_legs0 = time.perf_counter()
                free_b, total_b = torch.cuda.mem_get_info()
                while free_b < 0.97 * total_b and time.perf_counter() - t_legs0 < 20:   # released memory is handed back (and wiped) asynchronously
                    time.sleep(0.5)
                    gc.collect()
                    torch.cuda.empty_cache()
                    free_b, total_b = torch.cuda.mem_get_info()
                left = lambda: leg_budget - (time.perf_counter() - t_legs0)
                legs = {"budget_s": leg_budget, "hbm_free_before_gb": free_b / 1e9}
                try:
                    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(world), "--single-process", "--steps", "2", "--warmup", "1", "--pcie-steps", "0",
                           "--log-n", str(args.log_n), "--cols", str(ncols)]
                    t_c0 = time.perf_counter()
                    legs["commit_timeout_s"] = lim = min(left(), 0.4 * leg_budget)
                    rc, so, se, note = run_child(cmd, lim)
                    if rc != 0:
                        single_process = {"error": "child: %s" % (note or "exited %s" % rc), "stderr_tail": se[-800:]}
                    else:
                        j = json.loads(so.strip().splitlines()[-1])
                        single_process = {"what": "the same step from ONE process over the same devices (mi_multi_commit; python bench.py --gpus %d --single-process): what a one-process Prover links" % world,
                                          "value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"], "root_matches": j["root"] == root_host,
                                          "per_shard": j["per_rank"], "comm": j["comm"], "peer_access": j.get("peer_access"), "child_wall_s": time.perf_counter() - t_c0}
                except Exception as e:
                    single_process = {"error": repr(e)}
                # ... and one Starks::genProof at the same size with MI_STARK_DEVICES naming the same devices: sharded commits, row-sharded
                # step42ns / step52ns / evmap (host/starks.hpp) -- the batch-proof path on N devices, with its oracle checks
                try:
                    devs = os.environ.get("MI_BENCH_LEG_DEVICES") or ",".join(str(d_) for d_ in range(world))   # (the override: a one-GPU box rehearsing the plumbing)
                    if len(devs.split(",")) > 1:
                        env = dict(os.environ, MI_STARK_DEVICES=devs)
                        if len(set(devs.split(","))) == 1:   # logical shards of one GPU: a device group, so that the full size fits (host/starks.hpp)
                            env.update(MI_STARK_ROW_SHARDED="1", MI_MULTI_GROUP_SAME_DEVICE="1", MI_CHELPERS_BATCH_GIB="2")
                        else:
                            # real devices, first contact: the images as plain allocations (hipMalloc + hipDeviceEnablePeerAccess: the path every
                            # runtime has walked) rather than address ranges with per-piece peer access (hipMemSetAccess for another device has
                            # only ever run on one GPU here); 288 GB a device holds the dense form.  MI_BENCH_LEG_SPARSE=1 asks for the sparse one.
                            env.setdefault("MI_STARK_SPARSE_IMAGE", "1" if os.environ.get("MI_BENCH_LEG_SPARSE") == "1" else "0")
                        cmd = [sys.executable, os.path.join(ROOT, "bench_starks.py"), "--log-n", str(args.log_n), "--proofs", "2", "--check-rows", "8"]
                        t_c0 = time.perf_counter()
                        legs["genproof_timeout_s"] = lim = left()
                        rc, so, se, note = run_child(cmd, lim, env=env)
                        if rc != 0:
                            gp = {"error": "child: %s" % (note or "exited %s" % rc), "stderr_tail": se[-800:]}
                        else:
                            j = json.loads(so.strip().splitlines()[-1])
                            gp = {"what": "Starks::genProof (synthetic zkEVM-shaped STARK, 2^%d rows) with MI_STARK_DEVICES=%s: stage commits sharded, step42ns / step52ns / evmap row-sharded" % (args.log_n, devs),
                                  "ms": j["value"], "genproof_wall_ms": j["genproof_wall_ms"], "phase_ms": j["phase_ms"], "checks": j["checks"], "hbm_device0": j["hbm"],
                                  "peer_access": j.get("peer_access"), "child_wall_s": time.perf_counter() - t_c0}
                        single_process = dict(single_process or {}, genproof=gp)
                except Exception as e:
                    single_process = dict(single_process or {}, genproof={"error": repr(e)})
                legs["wall_s"] = time.perf_counter() - t_legs0
                single_process = dict(single_process or {}, legs=legs)
            try:
                dist.barrier(group=cpu_group)
            except Exception as e:
                if rank == 0 and single_process is not None:
                    single_process["after_the_child"] = repr(e)

    if rank == 0:
        K = max(args.steps, 1)
        ms_per_step = 1e3 * elapsed / K
        value = n * ncols * K / elapsed
        rows_local = plan.rows_per_rank
        # dominant kernel: k_linear_hash_rows -- algorithmic bytes per launch = 8*h*w read + 32*h written
        leaf_bytes = 8.0 * rows_local * ncols + 32.0 * rows_local
        leaf_ms = t_leaf / K
        perms = rows_local * ((ncols + 7) // 8 if ncols > 4 else 0)
        my_cols = plan.my_cols if exchange else ncols
        lde_bytes = 8.0 * n * my_cols + 8.0 * n_ext * my_cols
        lde_ms = t_lde / K
        achieved = leaf_bytes / (leaf_ms * 1e-3) / 1e9 if leaf_ms > 0 else 0.0
        out = {
            "metric": "goldilocks_field_elements_per_s_lde_merkleize_2^%d_rows" % args.log_n,
            "value": value, "unit": "field-elements/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if world > 1 else "none",
            "vs_baseline": None, "dtype": "u64 (Goldilocks mod 2^64-2^32+1, 32-bit integer VALU)", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: 2^%d-row x %d-col trace -> LDE blow-up 2 -> Poseidon Merkle tree"
                                   % (args.log_n, ncols),
                       "rows": n, "cols": ncols, "rows_ext": n_ext, "parallelism": "tile-cyclic column-shard LDE -> pipelined point-to-point exchange -> row-shard streaming Merkle x%d" % world,
                       "poseidon_variant": args.poseidon_variant},
            "root": root_host,
            "root_verified_against_oracle": bool(verify and verify.get("tree_levels_match_oracle") and verify.get("all_columns_all_rows_lincomb")),
            "leaf_rows_checked": (verify or {}).get("leaf_rows_checked", 0),
            "verify": verify,
            "root_matches_regression_constant": (root_host == ROOT_2P23_X665) if (args.log_n == 23 and ncols == 665) else None,
            "comm": comm, "per_rank": per_rank, "sharded_verify": sharded_verify, "single_process": single_process,
            "pcie_inclusive": pcie,
            "value_pcie_inclusive": (pcie or {}).get("value"),
            "roofline": {"kernel": "k_linear_hash_rows_lines" if args.leaf_mode else "k_linear_hash_rows", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_launch": leaf_bytes, "avg_launch_ms": leaf_ms,
                         "note": "VALU-bound kernel (~1.7e4 integer instructions per 64 B absorbed): see valu + DESIGN.md"},
            "roofline_lde": {"kernel": "k_ntt_pass x4 + k_lde_mid per column chunk (96 columns at the default workspace)", "bound": "hbm",
                             "achieved": lde_bytes / (lde_ms * 1e-3) / 1e9 if lde_ms > 0 else 0.0, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "algorithmic_bytes": lde_bytes, "avg_ms": lde_ms},
            "phase_ms": {"STARK_STEP_1_LDE": lde_ms, "STARK_STEP_1_MERKLETREE_leaves": leaf_ms, "STARK_STEP_1_MERKLETREE_levels": t_lvls / K},
        }
        out["roofline_lde"]["frac"] = out["roofline_lde"]["achieved"] / HBM_PEAK_GBS
        # HBM traffic of the dominant kernel comes from separate rocprofv3 --pmc passes (they cannot run inside the
        # timed process); the committed summary of the same command is read back here when the workload matches
        pmc_path = next((q for q in (os.path.join(ROOT, "profiles", f) for f in ("r04_pmc_leaf.json", "r03_pmc_leaf.json")) if os.path.exists(q)), "")
        pmc_name = os.path.basename(pmc_path)
        if (world == 1 and args.log_n == 23 and ncols == 665 and args.leaf_mode == 1 and args.poseidon_variant == 2
                and os.path.exists(pmc_path)):
            pmc = json.load(open(pmc_path))["kernels"]["k_linear_hash_rows_lines"]
            out["roofline"]["traffic"] = pmc["hbm_bytes_per_launch"]
            out["roofline"]["traffic_source"] = "profiles/%s (separate rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE passes of this command, gfx950 x2 read correction: tools/pmc_leaf.sh)" % pmc_name
            out["valu"] = valu_roofline(perms, leaf_ms, pmc)
            nb = os.path.join(ROOT, "profiles", "r05_ntt_valu_breakdown.json")
            if os.path.exists(nb):   # where the NTT passes' VALU instructions go, class by class (tools/pmc_ntt_classes.sh: builds with one class compiled out)
                out["valu"]["ntt_breakdown"] = json.load(open(nb))
        if world == 1 and not args.no_cpu_baseline:
            th = host_keep.numpy().view(np.uint64).reshape(n, ncols) if host_keep is not None else None
            out["cpu_baseline"] = cpu_baseline(args.cpu_log_n, ncols, th)
            th = host_keep = None
        else:
            out["cpu_baseline"] = None
        if world == 1 and not exchange and not args.no_genproof:
            # everything this process holds in HBM goes first: the child's proof plans 273 of the device's 309 GB
            del trace, root
            ext_windows = verify = None
            bufs.clear()
            gc.collect()
            torch.cuda.empty_cache()
            ctx.close()
            ctx = None
            torch.cuda.synchronize()
            free_b, total_b = torch.cuda.mem_get_info()
            t_wait0 = time.perf_counter()
            while free_b < 0.97 * total_b and time.perf_counter() - t_wait0 < 30:   # released memory is handed back (and wiped) asynchronously
                time.sleep(0.5)
                gc.collect()
                torch.cuda.empty_cache()
                free_b, total_b = torch.cuda.mem_get_info()
            print("bench.py: %.1f of %.1f GB of HBM free before the batch-proof leg" % (free_b / 1e9, total_b / 1e9), file=sys.stderr, flush=True)
            out["genproof"] = genproof_leg(args.genproof_shape)
            out["genproof"]["hbm_free_before_child_gb"] = free_b / 1e9
        print(json.dumps(out), file=real_stdout, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if ctx is not None:
        ctx.close()


if __name__ == "__main__":
    main()
