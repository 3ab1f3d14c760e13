"""PMC passes (rocprofv3, one counter set per pass - FETCH_SIZE and WRITE_SIZE cannot share one, MI355X_MICROARCH.md "rocprofv3
PMC slots") over one launch shape of every kernel family the bench line reports, summarised to one JSON that bench.py quotes
as `roofline_by_family.<family>.traffic` while the kernel sources still hash to `source_sha`:
  L2->fabric bytes per launch (FETCH_SIZE x 2 on gfx950 - 128-B requests are tallied at 64 B; WRITE_SIZE exact; Infinity-Cache
  hits are counted), SQ_VALU_MFMA_BUSY_CYCLES as a fraction of SIMD cycles, the clock held (GRBM_GUI_ACTIVE / 8 XCDs / duration),
  the wave-cycle split and the VALU / MFMA / LDS instruction counts, plus the kernel-trace duration of the same driver.
usage (on the GPU box): python3 scripts/pmc_families.py OUTDIR OUT.json [OUT.txt] [family-prefix ...]"""
import csv, glob, json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
outdir, outjson = sys.argv[1], sys.argv[2]
outtxt = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3].endswith(".txt") else None
only = [a for a in sys.argv[3:] if not a.endswith(".txt")]
D = 4096


def attn_case(Tq, Tk, B=2, H=32):
    tkp = (Tk + 63) // 64 * 64
    return {"family": "flash_attn", "driver": ["prof_family.py", f"attn_{Tq}x{Tk}" + ("_b1" if B == 1 else "")], "match": "flash_attn16_",
            "flops": 4.0 * B * H * Tq * Tk * 128, "algorithmic_bytes": 2.0 * B * D * (2 * Tq + Tk + tkp),
            "shape": f"B={B} H={H} Tq={Tq} Tk={Tk} dh=128"}


GEMM = {"ff1_gelu": (2560, 16384, 4096), "ff2_gate": (2560, 4096, 16384), "qk_sumsq": (2560, 8192, 4096), "v_transposed": (2560, 4096, 4096),
        "out_gate": (2560, 4096, 4096), "o2_res": (2560, 4096, 4096), "q2": (2560, 4096, 4096), "ctx_kv_split": (2048, 8192, 4096)}
CASES = {f"gemm_{k}": {"family": "gemm_bf16", "driver": ["prof_gemm.py", k], "match": "gemm_bf16_", "flops": 2.0 * M * N * K,
                       "algorithmic_bytes": 2.0 * (M * K + N * K + M * N), "shape": f"M={M} N={N} K={K}", "driver_iters_first": True}
         for k, (M, N, K) in GEMM.items()}
CASES.update({
    "attn_1280x1280": attn_case(1280, 1280), "attn_1280x1024": attn_case(1280, 1024), "attn_5184x5184": attn_case(5184, 5184),
    "conv128": {"family": "conv3d_k3", "driver": ["prof_family.py", "conv128"], "match": "conv3d_k3", "flops": 2.0 * 27 * 128 * 128 * 33 * 128 * 128,
                "algorithmic_bytes": 2.0 * (2 * 33 * 128 * 128 * 128 + 27 * 128 * 128), "shape": "33x128x128 voxels, Cin=Cout=128 (kw-reuse kernel)"},
    "conv256": {"family": "conv3d_k3", "driver": ["prof_family.py", "conv256"], "match": "conv3d_k3", "flops": 2.0 * 27 * 256 * 256 * 17 * 64 * 64,
                "algorithmic_bytes": 2.0 * (2 * 17 * 64 * 64 * 256 + 27 * 256 * 256), "shape": "17x64x64 voxels, Cin=Cout=256"},
    "norm_mod": {"family": "rmsnorm_modulate", "driver": ["prof_family.py", "norm_mod"], "match": "norm_", "flops": 0.0,
                 "algorithmic_bytes": 4.0 * 2560 * D, "shape": "M=2560 D=4096, carried row statistics, (1+scale), shift"},
    "norm_plain": {"family": "rmsnorm_modulate", "driver": ["prof_family.py", "norm_plain"], "match": "norm_", "flops": 0.0,
                   "algorithmic_bytes": 4.0 * 2560 * D, "shape": "M=2560 D=4096, carried row statistics, no modulation"},
    "qknorm_k": {"family": "qknorm_rope", "driver": ["prof_family.py", "qknorm_k"], "match": "qknorm_rope", "flops": 0.0,
                 "algorithmic_bytes": 4.0 * 2560 * D + 8.0 * 2560 * D // 2, "shape": "k in place, M=2560 D=4096, cos/sin (32,1280,64) fp32"},
    "pixelnorm128": {"family": "pixelnorm_act", "driver": ["prof_family.py", "pixelnorm128"], "match": "pixelnorm_act", "flops": 0.0,
                     "algorithmic_bytes": 4.0 * 33 * 128 * 128 * 128, "shape": "33x128x128 voxels x 128 channels, PixelNorm + SiLU"},
})
PASSES = {"fetch": ["FETCH_SIZE"], "write": ["WRITE_SIZE"],
          "sq1": ["GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU"],
          "sq2": ["SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_SALU", "SQ_INST_CYCLES_VMEM"]}
env = dict(os.environ, TMPDIR="/tmp")
WARM = 3
res, txt = {}, []
for name, c in CASES.items():
    if only and not any(name.startswith(p) for p in only):
        continue
    def cmd(iters):
        drv = os.path.join(root, "scripts", c["driver"][0])
        return [sys.executable, drv, str(iters), c["driver"][1]] if c.get("driver_iters_first") else [sys.executable, drv, c["driver"][1], str(iters)]
    vals = {}
    # kernel trace first: the family's dominant kernel of this driver (a launch may come with a small tail launch of the same
    # template); the counters below are taken from that kernel's largest grid only
    d = os.path.join(outdir, f"{name}_trace")
    subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--", *cmd(WARM + 6)], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    tot = 0.0
    for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
        for row in csv.DictReader(open(f, newline="")):
            if c["match"] in row["Name"] and float(row["TotalDurationNs"]) > tot:
                tot, vals["avg_ns"], vals["min_ns"], vals["kernel"] = float(row["TotalDurationNs"]), float(row["AverageNs"]), float(row["MinNs"]), row["Name"]
    for pname, counters in PASSES.items():
        d = os.path.join(outdir, f"{name}_{pname}")
        subprocess.run(["rocprofv3", "--pmc", *counters, "--output-format", "csv", "-d", d, "--", *cmd(WARM + 3)], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        rows = []
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            allr = [r for r in csv.DictReader(open(f, newline="")) if c["match"] in r["Kernel_Name"]]
            exact = [r for r in allr if r["Kernel_Name"] == vals.get("kernel")]
            rows += exact if exact else allr          # (the two files name a kernel the same way; if not, the largest grid decides)
        grid = max((int(r["Grid_Size"]) for r in rows), default=0)
        acc = {}
        for r in rows:
            if int(r["Grid_Size"]) == grid:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for cn, v in acc.items():
            v = v[WARM:] if len(v) > WARM else v
            vals[cn] = sum(v) / len(v)
    e = {"family": c["family"], "shape": c["shape"], "kernel": vals.get("kernel"), "counters": vals, "algorithmic_bytes": c["algorithmic_bytes"]}
    if "FETCH_SIZE" in vals:
        e["FETCH_SIZE_KiB"], e["WRITE_SIZE_KiB"] = vals["FETCH_SIZE"], vals.get("WRITE_SIZE")
        e["read_bytes_corrected"] = 2 * 1024 * vals["FETCH_SIZE"]        # gfx950: 64 B tallied per 128-B request (MI355X_MICROARCH.md, HBM)
        e["write_bytes"] = 1024 * vals.get("WRITE_SIZE", 0.0)
        e["bytes_per_launch"] = e["read_bytes_corrected"] + e["write_bytes"]
        e["traffic_over_algorithmic"] = e["bytes_per_launch"] / c["algorithmic_bytes"]
    if "GRBM_GUI_ACTIVE" in vals and vals.get("avg_ns"):
        cyc = vals["GRBM_GUI_ACTIVE"] / 8.0                                  # summed over the 8 XCDs
        if vals["avg_ns"] >= 30000:          # (short launches: the un-profiled trace duration and the profiled cycle count do not describe the same run)
            e["clock_GHz_profiled"] = cyc / vals["avg_ns"]
        e["mfma_busy_frac_of_simd_cycles"] = vals.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * cyc)
        w = vals["SQ_WAVE_CYCLES"]
        e["wave_cycle_split"] = {"parked_WAIT_ANY": vals["SQ_WAIT_ANY"] / w, "issue_stalled_WAIT_INST_ANY": vals["SQ_WAIT_INST_ANY"] / w,
                                 "issuing_ACTIVE_INST_ANY": vals["SQ_ACTIVE_INST_ANY"] / w}
        if vals.get("SQ_INSTS_MFMA"):
            e["valu_per_mfma"] = vals["SQ_INSTS_VALU"] / vals["SQ_INSTS_MFMA"]
            e["lds_insts_per_mfma"] = vals.get("SQ_INSTS_LDS", 0.0) / vals["SQ_INSTS_MFMA"]
        if c["flops"]:
            e["tflops_by_trace_avg"], e["tflops_by_trace_min"] = c["flops"] / vals["avg_ns"] / 1e3, c["flops"] / vals["min_ns"] / 1e3
        e["GBs_algorithmic_by_trace_avg"] = c["algorithmic_bytes"] / vals["avg_ns"]
        if "bytes_per_launch" in e:
            e["GBs_fabric_by_trace_avg"] = e["bytes_per_launch"] / vals["avg_ns"]
    res[name] = e
    line = name + " " + json.dumps({k: v for k, v in e.items() if k != "counters"})
    print(line, flush=True)
    txt.append(f"{name}  [{c['shape']}]  {vals.get('kernel')}")
    for cn in sorted(vals):
        if isinstance(vals[cn], float):
            txt.append(f"   {cn:34s} {vals[cn]:18.1f}")
    for k in ("bytes_per_launch", "traffic_over_algorithmic", "clock_GHz_profiled", "mfma_busy_frac_of_simd_cycles", "valu_per_mfma", "lds_insts_per_mfma",
              "tflops_by_trace_avg", "GBs_algorithmic_by_trace_avg", "GBs_fabric_by_trace_avg"):
        if k in e:
            txt.append(f"   -> {k:32s} {e[k]:18.4f}")
from bench import source_sha
out = {"source": "rocprofv3 --pmc (one pass per counter set) -- python3 scripts/prof_gemm.py | scripts/prof_family.py; kernel-trace of the same driver; MI355X",
       "correction": "FETCH_SIZE counts 64 B per 128-B L2 read request on gfx950: bytes = 2*FETCH_SIZE KiB; WRITE_SIZE exact; Infinity-Cache hits are counted (L2->fabric traffic)",
       "source_sha": source_sha(), "kernels": res}
if os.path.exists(outjson) and only:            # partial run: merge into the existing file if it was measured on the same sources
    old = json.load(open(outjson))
    if old.get("source_sha") == out["source_sha"]:
        old["kernels"].update(res)
        out = old
json.dump(out, open(outjson, "w"), indent=1)
if outtxt:
    open(outtxt, "w").write(f"source_sha {out['source_sha']}\n" + "\n".join(txt) + "\n")
