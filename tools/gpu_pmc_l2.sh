#!/bin/bash
# L2 hit rate and L2 <-> fabric read traffic of the f64 16384^3 product for several tile maps:
# arguments "group_m xcd_remap dtype n".
set -o pipefail
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_l2
rm -rf $OUT; mkdir -p $OUT
for cfg in "$@"; do
  set -- $cfg
  export EIGAMD_GROUP_M=$1 EIGAMD_XCD_REMAP=$2
  dt=${3:-f64}; n=${4:-16384}
  for pass in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    tag=${dt}_gm$1_x$2_$(echo $pass | tr ' ' '_' | cut -c1-12)
    timeout -s KILL 120 rocprofv3 --pmc $pass -d $OUT/$tag -o r -- ./build/gemm_check $dt N N $n $n $n 1 > $OUT/$tag.log 2>&1 || { echo "pass $tag failed"; tail -3 $OUT/$tag.log; }
  done
  echo "cfg $dt $n gm=$1 xcd=$2: $(grep -o 'best=.*TFLOP/s' $OUT/$tag.log)"
done
python3 - <<'PY'
import sqlite3, glob
for db in sorted(glob.glob("gpurun_out/prof_l2/*/**/*.db", recursive=True)):
    cur = sqlite3.connect(db).cursor()
    vals = dict(cur.execute("select counter_name, avg(value) from counters_collection where kernel_name like '%gemm_kernel%' group by counter_name"))
    print(db.split("/")[2], {k: f"{v:.4e}" for k, v in vals.items()},
          ("hit %.1f %%" % (100 * vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"]))) if "TCC_HIT_sum" in vals else "")
PY
