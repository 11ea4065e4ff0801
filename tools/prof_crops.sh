R=$PWD; cd /tmp && export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3_crops2 -o runc --output-format csv -- python3 $R/tools/bench_crops.py > $R/gpurun_out/r3_crops2.log 2>&1; cd $R
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/r3_crops2/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:6]:
    print("%-60s calls %5s avg %8.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
grep "device builder" gpurun_out/r3_crops2.log
python -m pytest tests/test_crops.py -m gpu -x -q 2>&1 | tail -2
