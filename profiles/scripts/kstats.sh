# kernel statistics of a long solve: bash profiles/scripts/kstats.sh <tag> <cfg> [ENV=VAL ...]   -> gpurun_out/ks/<tag>.txt
R=$GRAFT_REPO_ROOT; TAG=$1; CFG=$2; shift 2
O=$R/gpurun_out/ks; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
env "$@" PYTHONPATH=$R:$R/tests rocprofv3 --kernel-trace --stats --output-format csv -d $O/$TAG -o solve -- python $R/profiles/scripts/prof_cfg.py $CFG > $O/$TAG.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("$O/$TAG/**/solve_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open("$O/$TAG.txt", "w") as out:
    for r in rows[:22]:
        line = "%-60s calls %5s avg %8.1f us" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3)
        print(line); out.write(line + "\n")
PY
