#!/usr/bin/env python3
"""Copies the summaries that tools/gpu_profile.sh left under gpurun_out/prof into profiles/
(round tag as argument, default r01) and rewrites profiles/traffic.json from the PMC passes."""
import json, os, re, shutil, sqlite3, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(R, "gpurun_out", "prof")
cur = sqlite3.connect(os.path.join(src, "trace", "bench_results.db")).cursor()
rows = cur.execute("select name, grid_x/workgroup_x, count(*), avg(end-start)/1e6, min(end-start)/1e6, max(end-start)/1e6 "
                   "from kernels where name like '%gemm_kernel%' and (end-start) > 2e7 group by name, grid_x order by 4 desc").fetchall()
b = json.load(open(os.path.join(src, "bench_under_trace.json")))
with open(os.path.join(R, "profiles", f"{tag}_kernel_trace_stats_bench.txt"), "w") as f:
    f.write("rocprofv3 --kernel-trace --stats -- python3 bench.py     (default flags: N=1, 10 steps, 2 warm-up, extras and CPU baseline on)\n")
    f.write("summary of the rocpd database by tools/rocpd_summary.py; gemm_kernel<1,...> also counts the launches of the LU / TRSM / SYRK legs\n\n")
    f.write(open(os.path.join(src, "kernel_trace_stats_bench.txt")).read().split("\n", 1)[1])
    f.write("\nheadline launches only (duration > 20 ms), grouped by kernel and grid:\n")
    f.write(" workgroups  calls     avg_ms     min_ms     max_ms  kernel\n")
    for n, g, c, a, mn, mx in rows:
        f.write(f" {g:10d} {c:6d} {a:10.3f} {mn:10.3f} {mx:10.3f}  {n}\n")
    # the timed launches: f64 kernel, one workgroup per 256 x 128 tile of a 16384^2 C, full reduction length (the SYRK leg
    # and TRMM legs launch the same grid, but skip tiles or k-tiles: they are told apart by their duration)
    wg = (16384 // 256) * (16384 // 128)
    durs = [d for (d,) in cur.execute("select (end-start)/1e6 from kernels where name like '%gemm_kernel<1%' and grid_x/workgroup_x = ?", (wg,))]
    head = [d for d in durs if d > 0.9 * max(durs)] if durs else []
    if head:
        f.write(f"\nbench.py's own HIP-event average of the timed dgemm 16384^3 launches in the same run: {b['roofline']['avg_launch_ms']} ms "
                f"(trace: {sum(head) / len(head):.3f} ms over the {len(head)} launches of that product, warm-up included; the other "
                f"{len(durs) - len(head)} launches of the same grid are the SYRK and TRMM legs); value {b['value']} TFLOP/s\n")
# per-variant table from the trace: kernel instantiation <kind, A k-contiguous, B k-contiguous> = operand forms
# (A k-contiguous: 'T' or 'C' on A; B k-contiguous: 'N' on B; conjugation is a run-time flag of the same kernel),
# full launches of the configured products only (within 10 % of the longest launch of that instantiation and grid)
with open(os.path.join(R, "profiles", f"{tag}_kernel_trace_stats_bench.txt"), "a") as f:
    f.write("\nper variant, from the trace (configured sizes; TFLOP/s = flops / average launch duration):\n")
    f.write(" product            forms (A, B)      launches     avg_ms   TFLOP/s\n")
    spec = [(1, "dgemm 16384^3", (16384 // 256) * (16384 // 128), 2.0 * 16384 ** 3),
            (0, "sgemm 16384^3", (16384 // 256) * (16384 // 256), 2.0 * 16384 ** 3),
            (3, "zgemm 8192^3", (8192 // 128) * (8192 // 128), 8.0 * 8192 ** 3),
            (2, "cgemm 8192^3", (8192 // 128) * (8192 // 256), 8.0 * 8192 ** 3)]
    for kind, label, wg, flops in spec:
        for ak in ("false", "true"):
            for bk in ("true", "false"):
                name = f"%gemm_kernel<{kind}, {ak}, {bk}, false>%"
                d = [x for (x,) in cur.execute("select (end-start)/1e6 from kernels where name like ? and grid_x/workgroup_x = ?", (name, wg))]
                full = [x for x in d if x > 0.9 * max(d)] if d else []
                if not full:
                    continue
                forms = ("T|C" if ak == "true" else "N") + ", " + ("N" if bk == "true" else "T|C")
                avg = sum(full) / len(full)
                f.write(f" {label:18s} {forms:16s} {len(full):9d} {avg:10.3f} {flops / avg / 1e9:9.2f}\n")
json.dump(b, open(os.path.join(R, "profiles", f"{tag}_bench_n1.json"), "w"), indent=1)
pmc = open(os.path.join(src, "pmc_gemm16384.txt")).read()
open(os.path.join(R, "profiles", f"{tag}_pmc_gemm16384.txt"), "w").write(pmc)
traffic = {}
for dt in ("f64", "f32"):
    sec = pmc.split(f"gemm_check {dt} ")[1].split("== build")[0]
    fetch = float(re.search(r"FETCH_SIZE\s+([0-9.e+]+)", sec).group(1))
    write = float(re.search(r"WRITE_SIZE\s+([0-9.e+]+)", sec).group(1))
    traffic[f"{dt}_16384"] = {
        "bytes_per_launch": float(f"{fetch * 1024 * 2 + write * 1024:.4e}"),
        "source": f"profiles/{tag}_pmc_gemm16384.txt: FETCH_SIZE {fetch:.3e} KB x 1024 x 2 (gfx950 128-B request correction) + "
                  f"WRITE_SIZE {write:.3e} KB x 1024; L2<->fabric traffic, Infinity-Cache hits included"}
json.dump(traffic, open(os.path.join(R, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
