#!/bin/bash
# rocprofv3 passes for the headline kernels.  Each pass is its own run; counter passes
# use --pmc only (never with a trace domain).  Summaries land in gpurun_out/prof/*.txt.
set -o pipefail
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT


timeout -k 10 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python3 bench.py > $OUT/trace_bench.log 2>&1
echo "profiler exit status $?" | tee $OUT/trace_exit_status.txt
grep '^{' $OUT/trace_bench.log > $OUT/bench_under_trace.json || { tail -5 $OUT/trace_bench.log; exit 1; }
python3 tools/rocpd_summary.py $(find $OUT/trace -name "*.db") > $OUT/kernel_trace_stats_bench.txt
echo "trace done"
for dt in f64 f32; do
  for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_LDS" "GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "TCC_HIT_sum TCC_MISS_sum"; do
    tag=${dt}_$(echo $pass | tr ' ' '_' | cut -c1-24)
    timeout -s KILL 120 rocprofv3 --pmc $pass -d $OUT/pmc_$tag -o r -- ./build/gemm_check $dt N N 16384 16384 16384 2 > $OUT/pmc_$tag.log 2>&1 || { echo "pass $tag failed"; tail -3 $OUT/pmc_$tag.log; }
    echo "pass $tag done"
  done
done
python3 - <<'PY' > $OUT/pmc_gemm16384.txt
import sqlite3, glob, os
out = "gpurun_out/prof"
for dt in ("f64", "f32"):
    print(f"== build/gemm_check {dt} N N 16384 16384 16384 2   (separate rocprofv3 --pmc passes; averages per launch of gemm_kernel)")
    vals = {}
    for db in sorted(glob.glob(f"{out}/pmc_{dt}_*/**/*.db", recursive=True)):
        cur = sqlite3.connect(db).cursor()
        ms = cur.execute("select avg(end-start)/1e6 from kernels where name like '%gemm_kernel%'").fetchone()[0]
        for c, v in cur.execute("select counter_name, avg(value) from counters_collection where kernel_name like '%gemm_kernel%' group by counter_name"):
            vals[c] = v
            print(f"  {c:32s} {v:16.6e}   (kernel avg {ms:.2f} ms in this pass)")
    if "GRBM_GUI_ACTIVE" in vals and "SQ_VALU_MFMA_BUSY_CYCLES" in vals:
        cyc = vals["GRBM_GUI_ACTIVE"] / 8
        print(f"  -> cycles per XCD {cyc:.4e}; MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / cycles = {100 * vals['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc:.1f} %")
    if "FETCH_SIZE" in vals:
        f = vals["FETCH_SIZE"] * 1024 * 2; w = vals.get("WRITE_SIZE", 0) * 1024
        print(f"  -> L2<->fabric bytes per launch: FETCH_SIZE x 1024 x 2 (gfx950 correction) = {f:.4e}, WRITE_SIZE x 1024 = {w:.4e}, total {f + w:.4e}")
    if "TCC_HIT_sum" in vals:
        print(f"  -> L2 hit rate {100 * vals['TCC_HIT_sum'] / (vals['TCC_HIT_sum'] + vals['TCC_MISS_sum']):.1f} %")
PY
cat $OUT/pmc_gemm16384.txt
