# Round-4 evidence, ALL of it from the commit that is checked out:  bash profiles/scripts/collect_r04.sh   (on the GPU box)
# Raw rocprofv3 output goes to gpurun_out/r4p (scratch); `python profiles/make_summary.py r04 gpurun_out/r4p` condenses it into the
# tracked profiles/r04_* files (kernel statistics of the bench and of the solves of cfg2 / cfg3 / cfg4, PMC averages of every
# kernel of the evaluation and of the LM iteration, the LM timeline, phase stamps, host-phase breakdown of Workspace.calibrate).
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-scipy-mode --no-lsmr-mode"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o trace -- $B --steps 200 --warmup 20 > $O/bench_traced.json 2> $O/bench_traced.err
# PMC: counters in their own passes, kernel trace only (no other trace domain)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc -o pmc_fetch -- $B --steps 30 --warmup 5 --repeats 1 --no-cpu-baseline --no-solve > /dev/null 2> $O/pmc1.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc -o pmc_write -- $B --steps 30 --warmup 5 --repeats 1 --no-cpu-baseline --no-solve > /dev/null 2> $O/pmc2.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/pmc -o pmc_mfma -- $B --steps 30 --warmup 5 --repeats 1 --no-cpu-baseline --no-solve > /dev/null 2> $O/pmc3.err
for CFG in cfg3 cfg4 cfg2; do
  PYTHONPATH=$R:$R/tests rocprofv3 --kernel-trace --stats --output-format csv -d $O/solve_$CFG -o solve -- python $R/profiles/scripts/prof_cfg.py $CFG > $O/solve_$CFG.log 2>&1
done
for C in FETCH_SIZE WRITE_SIZE; do
  PYTHONPATH=$R:$R/tests rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/spmc_$C -o pmc -- python $R/profiles/scripts/prof_cfg.py cfg3 > $O/spmc_$C.log 2>&1
done
cd $R
python bench.py > $O/bench.json 2> $O/bench.err
tail -c 300 $O/bench.json
python profiles/scripts/prof_linearize.py > $O/lin_phases.log 2>&1; tail -12 $O/lin_phases.log
for CFG in cfg3 cfg4 cfg2; do MCBA_TIMING=1 python profiles/scripts/prof_workspace.py $CFG > $O/workspace_$CFG.log 2>&1; grep "calibrate ms" $O/workspace_$CFG.log; done
MCBA_TIMING=1 python profiles/scripts/prof_workspace.py cfg3 --float32 > $O/workspace_cfg3_f32.log 2>&1; grep "calibrate ms" $O/workspace_cfg3_f32.log
python profiles/scripts/prof_lin_cfgs.py cfg2 cfg3 cfg4 cfg5 > $O/lin_cfgs.log 2>&1; cat $O/lin_cfgs.log
python profiles/scripts/prof_r4.py frames cfg3 cfg4 > $O/frame_groups.log 2>&1; cat $O/frame_groups.log
python profiles/scripts/prof_chol_phases.py > $O/chol_phases.log 2>&1; tail -4 $O/chol_phases.log
python profiles/scripts/prof_chol.py > $O/chol_paths.log 2>&1; cat $O/chol_paths.log
python profiles/scripts/prof_init.py cfg2 cfg3 cfg4 > $O/init.log 2>&1; cat $O/init.log
python profiles/scripts/prof_lsmr.py > $O/lsmr_mode.md 2> $O/lsmr_mode.err; tail -5 $O/lsmr_mode.md
python profiles/scripts/prof_parity_table.py > $O/parity_table.md 2> $O/parity_table.err; cp gpurun_out/parity_table.json $O/ 2>/dev/null
ls $O
