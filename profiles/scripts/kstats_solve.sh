# per-kernel average durations of the solver loop: bash profiles/scripts/kstats_solve.sh <tag> [cfg]
R=$GRAFT_REPO_ROOT; T=$1; CFG=${2:-cfg3}; O=$R/gpurun_out/kss_$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PYTHONPATH=$R:$R/tests rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o trace -- python $R/profiles/scripts/prof_cfg.py $CFG > $O/solve.log 2>&1
grep "^$CFG" $O/solve.log
python - <<PY
import csv, glob
f = glob.glob("$O/trace/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
  n = r["Name"]
  if "mcba" in n and int(r["Calls"]) > 20: print("$T", n.split("(")[0][-40:], r["Calls"], round(float(r["AverageNs"]) / 1e3, 2))
PY
cd $R
