"""Turn the rocprofv3 CSVs of scripts/profile_round.sh into traffic.json (HBM bytes per launch of the bench kernel).

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-byte read requests at 64 bytes and is doubled
(MI355X_MICROARCH.md, HBM / rocprofv3 section).  Only the dispatches of the workload's sketch kernel are averaged.
"""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (WORKLOADS table)

# the sketch kernel family each workload's plan lands on (only used to pick that kernel's rows out of the CSVs; the full
# instantiated name is read from the rows themselves)
FAMILY = {"min": "k_minimizer_pk<", "nt": "k_nthash_fast", "syn": "k_syncmer_pf<", "pmin": "k_prot_minimizer_fast", "kmer": "k_nthash_fast",
          "phash": "k_prot_hash_fast", "sim": "k_simhash_fast"}

out_dir, workloads = sys.argv[1], sys.argv[2].split()
VALU_MODEL = {}
for cand in (os.path.join(out_dir, "valu_model.json"), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "valu_model.json")):
    if os.path.exists(cand):
        VALU_MODEL = json.load(open(cand)).get("kernels", {})
        break
entries = []
for w in workloads:
    kind, n_reads = bench.WORKLOADS[w][0], bench.WORKLOADS[w][1]
    kname = "k_minimizer_ring<" if w == "minimizer250" else "k_minimizer_pkd<" if w == "minimizer400" else "k_syncmer_pfl<" if w == "syncmer250" else FAMILY[kind]  # (reads of ~160-280 bases are planned on the unit-row kernel; syncmers of 190+ bases on the long packed plan)
    full = {"name": None}

    def mean_counter(path, counter):
        vals = []
        if not os.path.exists(path):
            return None
        for row in csv.DictReader(open(path)):
            if kname in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                vals.append(float(row["Counter_Value"]))
                full["name"] = row["Kernel_Name"].replace("void bsk::", "").replace("(bsk::KArgs)", "").replace(", ", ",")
        return sum(vals) / len(vals) if vals else None

    f = mean_counter(os.path.join(out_dir, f"bench_{w}_pmc_fetch.csv"), "FETCH_SIZE")
    wr = mean_counter(os.path.join(out_dir, f"bench_{w}_pmc_write.csv"), "WRITE_SIZE")
    kms = None
    sp = os.path.join(out_dir, f"bench_{w}_kernel_stats.csv")
    if os.path.exists(sp):
        for row in csv.DictReader(open(sp)):
            if kname in row.get("Name", ""):
                kms = float(row["AverageNs"]) / 1e6
    if f is None or wr is None:
        print("no counters for", w)
        continue
    # VALU utilisation: wave-instructions issued to the VALU (one quad-cycle each) over the SIMD-cycles of the dispatch
    # (256 CUs x 4 SIMDs x the shader clock cycles GRBM_GUI_ACTIVE counts per XCD; 8 XCDs are summed in the CSV)
    sqp = os.path.join(out_dir, f"bench_{w}_pmc_sq.csv")
    va, gui, vi = mean_counter(sqp, "SQ_ACTIVE_INST_VALU"), mean_counter(sqp, "GRBM_GUI_ACTIVE"), mean_counter(sqp, "SQ_INSTS_VALU")
    # VALU-issue roofline: wave-instructions issued (SQ_INSTS_VALU) x the mean issue cost of the kernel's inner-loop instruction mix
    # (scripts/valu_model.py: 2.0 cycles for VOP1/VOP2/v_bitop3, 3.5 for VOP3 forms, 4.0 for 64-bit / multiplier ops -- the rates
    # measured in scripts/ubench) over the SIMD-cycles of the dispatch (1024 SIMDs x the shader cycles GRBM_GUI_ACTIVE counts per XCD)
    valu = None
    vm = VALU_MODEL.get(w, {})
    if vi and gui and "cycles_per_inst" in vm:
        simd_cycles = gui / 8 * 1024
        units = n_reads * bench.WORKLOADS[w][2]
        valu = {"wave_insts": vi, "cycles_per_inst": vm["cycles_per_inst"], "vop3_frac": vm["vop3_frac"], "issue_cycles": vi * vm["cycles_per_inst"],
                "simd_cycles": simd_cycles, "frac": round(vi * vm["cycles_per_inst"] / simd_cycles, 4),
                "issue_cycles_per_unit": round(vi * vm["cycles_per_inst"] / units, 5), "insts_per_unit": round(vi / units, 4),
                "shader_clock_GHz": round(gui / 8 / (kms * 1e-3) / 1e9, 3) if kms else None}
    entries.append({"workload": w, "reads_per_gpu": n_reads, "kernel": full["name"], "valu": valu, "commit": os.environ.get("BSK_BENCH_COMMIT"),
                    "SQ_ACTIVE_INST_VALU": va, "GRBM_GUI_ACTIVE_sum_over_xcds": gui, "FETCH_SIZE_KiB_mean": round(f), "WRITE_SIZE_KiB_mean": round(wr),
                    "fetch_bytes_corrected": int(f * 1024 * 2), "write_bytes": int(wr * 1024), "hbm_bytes_per_launch": int(f * 2048 + wr * 1024),
                    "rocprof_kernel_ms_avg": kms})
json.dump({"note": "HBM traffic of one launch of the bench kernel: rocprofv3 --pmc passes on `python bench.py --workload W --steps 3 --warmup 1` "
                   "(separate FETCH_SIZE and WRITE_SIZE passes; KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md: gfx950 tallies 128-byte read "
                   "requests at 64 bytes). Raw CSVs beside this file.", "entries": entries}, open(os.path.join(out_dir, "traffic.json"), "w"), indent=1)
print(json.dumps(entries, indent=1))
