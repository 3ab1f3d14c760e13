"""PMC passes (rocprofv3, one counter set per pass) over each GEMM launch shape of the DiT block, summarised to JSON:
L2->fabric bytes (FETCH_SIZE / WRITE_SIZE, gfx950-corrected), MFMA-busy, wait split, effective clock.
usage: python scripts/pmc_gemm.py OUTDIR OUT.json    (run on the GPU box; rocprofv3 wraps `python scripts/prof_gemm.py`)"""
import csv, glob, json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
outdir, outjson = sys.argv[1], sys.argv[2]
SHAPES = {"ff1_gelu": (2560, 16384, 4096), "ff2_gate": (2560, 4096, 16384), "qk_sumsq": (2560, 8192, 4096), "v_transposed": (2560, 4096, 4096), "out_gate": (2560, 4096, 4096),
          "o2_res": (2560, 4096, 4096), "q2": (2560, 4096, 4096), "ctx_kv_split": (2048, 8192, 4096)}
PASSES = {"fetch": ["FETCH_SIZE"], "write": ["WRITE_SIZE"],
          "sq1": ["GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU"],
          "sq2": ["SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_SALU", "SQ_INST_CYCLES_VMEM"]}
env = dict(os.environ, TMPDIR="/tmp")
res = {}
for shape, (M, N, K) in SHAPES.items():
    vals = {}
    for pname, counters in PASSES.items():
        d = os.path.join(outdir, f"{shape}_{pname}")
        subprocess.run(["rocprofv3", "--pmc", *counters, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(root, "scripts", "prof_gemm.py"), "3", shape],
                       env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        acc = {}
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f, newline="")):
                if "gemm_bf16_" in row["Kernel_Name"]:
                    acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        for c, v in acc.items():
            v = v[3:] if len(v) > 3 else v          # skip the warm-up launches
            vals[c] = sum(v) / len(v)
    d = os.path.join(outdir, f"{shape}_trace")
    subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(root, "scripts", "prof_gemm.py"), "6", shape],
                   env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
        for row in csv.DictReader(open(f, newline="")):
            if "gemm_bf16_" in row["Name"]:
                vals["avg_ns"], vals["min_ns"], vals["kernel"] = float(row["AverageNs"]), float(row["MinNs"]), row["Name"]
    e = {"M": M, "N": N, "K": K, "kernel": vals.get("kernel"), "counters": vals}
    if "FETCH_SIZE" in vals:
        e["FETCH_SIZE_KiB"], e["WRITE_SIZE_KiB"] = vals["FETCH_SIZE"], vals.get("WRITE_SIZE")
        e["read_bytes_corrected"] = 2 * 1024 * vals["FETCH_SIZE"]        # gfx950: 64 B tallied per 128-B request (MI355X_MICROARCH.md, HBM)
        e["write_bytes"] = 1024 * vals.get("WRITE_SIZE", 0.0)
        e["algorithmic_bytes"] = 2 * (M * K + N * K + M * N)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and "min_ns" in vals:
        simds = 1024
        cyc = vals["GRBM_GUI_ACTIVE"] / 8.0                                  # summed over the 8 XCDs
        e["clock_GHz_profiled"] = cyc / vals["avg_ns"] if vals.get("avg_ns") else None
        e["mfma_busy_frac_of_simd_cycles"] = vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (simds * cyc)
        w = vals["SQ_WAVE_CYCLES"]
        e["wave_cycle_split"] = {"parked_WAIT_ANY": vals["SQ_WAIT_ANY"] / w, "issue_stalled_WAIT_INST_ANY": vals["SQ_WAIT_INST_ANY"] / w,
                                 "issuing_ACTIVE_INST_ANY": vals["SQ_ACTIVE_INST_ANY"] / w}
        e["tflops_by_trace_min"] = 2.0 * M * N * K / vals["min_ns"] / 1e3
        e["tflops_by_trace_avg"] = 2.0 * M * N * K / vals["avg_ns"] / 1e3
    res[shape] = e
    print(shape, json.dumps({k: v for k, v in e.items() if k != "counters"}), flush=True)
from bench import source_sha
names = {"ff1_gelu": "gemm_bf16_big_kernel<GELU> FF1 M=2560 N=16384 K=4096"}
out = {"source": "rocprofv3 --pmc (one pass per counter set) -- python scripts/prof_gemm.py 3 <shape>; kernel-trace of the same driver; MI355X",
       "correction": "FETCH_SIZE counts 64 B per 128-B L2 read request on gfx950: bytes = 2*FETCH_SIZE KiB; WRITE_SIZE exact; Infinity-Cache hits are counted (L2->fabric traffic)",
       "source_sha": source_sha(),
       "kernels": {names.get(k, k): v for k, v in res.items()}}
json.dump(out, open(outjson, "w"), indent=1)
