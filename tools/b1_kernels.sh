#!/bin/bash
# per-kernel times of the one-crop whole-forward graph replays (rocprofv3 kernel trace of tools/profile_stream.py), top 30
export TMPDIR=/tmp
cd "$(dirname "$0")/.." || exit 1
T=${1:-b1k}
rm -rf gpurun_out/$T
rocprofv3 --kernel-trace --stats -d gpurun_out/$T -o runc --output-format csv -- python3 tools/profile_stream.py ${2:-1} > gpurun_out/$T.log 2>&1
python3 - "$T" <<'P'
import csv, glob, re, sys
f = glob.glob("gpurun_out/%s/**/*kernel_stats.csv" % sys.argv[1], recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:30]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]).replace("void ", "")[:80]
    print("%-80s %5s %8.1f us %5.1f%%" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
P
