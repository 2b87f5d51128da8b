#!/bin/bash
# MFMA utilisation and clock of kernel variants:  name:dtype:size ...
set -o pipefail
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_util
rm -rf $OUT; mkdir -p $OUT
for v in "$@"; do
  IFS=: read name dt n <<< "$v"
  export LD_LIBRARY_PATH=$PWD/build/variants/$name
  for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU" "GRBM_GUI_ACTIVE"; do
    tag=${name}-${dt}-${n}_$(echo $pass | cut -c1-8)
    timeout -s KILL 120 rocprofv3 --pmc $pass -d $OUT/$tag -o r -- ./build/gemm_check $dt N N $n $n $n 1 > $OUT/$tag.log 2>&1 || { echo "pass $tag failed"; tail -3 $OUT/$tag.log; }
  done
  echo "$name $dt: $(grep -o 'best=.*TFLOP/s' $OUT/$tag.log)"
done
python3 - <<'PY'
import sqlite3, glob, os
out = os.environ.get("OUT", "gpurun_out/prof_util")
res = {}
for db in sorted(glob.glob(out + "/*/r_results.db")):
    tag = os.path.basename(os.path.dirname(db))
    cur = sqlite3.connect(db).cursor()
    k = cur.execute("select avg(end-start) from kernels where name like '%gemm_kernel%'").fetchone()[0]
    rows = cur.execute("select counter_name, avg(value) from counters_collection where kernel_name like '%gemm_kernel%' group by counter_name").fetchall()
    base = tag.rsplit("_", 2)[0] if tag.endswith("GRBM_GUI") or tag.endswith("SQ_VALU_") else tag
    key = tag.replace("_GRBM_GUI", "").replace("_SQ_VALU_", "")
    res.setdefault(key, {})["ms_" + ("g" if "GRBM" in tag else "s")] = k / 1e6
    for c, v in rows: res[key][c] = v
for key, d in res.items():
    if "GRBM_GUI_ACTIVE" in d and "SQ_VALU_MFMA_BUSY_CYCLES" in d:
        cyc = d["GRBM_GUI_ACTIVE"] / 8
        clk = cyc / (d["ms_g"] * 1e-3) / 1e9
        util = d["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc
        print(f"{key:28s} {d['ms_g']:8.2f} ms  clock {clk:.3f} GHz  MfmaUtil {100*util:5.1f}%  wait_any/wave {d['SQ_WAIT_ANY']/d['SQ_WAVE_CYCLES']:.3f}  "
              f"wait_lds/wave {d['SQ_WAIT_INST_LDS']/d['SQ_WAVE_CYCLES']:.4f}  active/wave {d['SQ_ACTIVE_INST_ANY']/d['SQ_WAVE_CYCLES']:.3f}  lds_insts {d['SQ_INSTS_LDS']:.3e} valu {d['SQ_INSTS_VALU']:.3e}")
PY
