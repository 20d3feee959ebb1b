# kernel statistics of any command:  bash profiles/scripts/quick_trace_cmd.sh <tag> python profiles/scripts/prof_workspace.py cfg3 --solver lsmr
R=$GRAFT_REPO_ROOT; TAG=$1; shift; O=$R/gpurun_out/qtc_$TAG; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp PYTHONPATH=$R:$R/tests
cd $R && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o trace -- "$@" > $O/run.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("$O/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("$TAG: total kernel time %.2f ms" % (tot / 1e6))
for r in rows[:14]:
  print("%-58s calls %6s avg %8.2f us  %5.1f%%" % (r["Name"][:58], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*.db" -delete
