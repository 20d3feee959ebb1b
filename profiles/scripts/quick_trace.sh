# quick kernel statistics of the bench step (evaluation only):  bash profiles/scripts/quick_trace.sh [extra bench args]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/qt; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R:$R/tests
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o trace -- python $R/bench.py --no-scipy-mode --no-lsmr-mode --steps 200 --warmup 20 --no-cpu-baseline --no-solve "$@" > $O/bench.json 2> $O/bench.err
python - <<PY
import csv, glob
f = glob.glob("$O/trace/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
  print("%-60s calls %6s avg %8.2f us min %8.2f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*.db" -delete
