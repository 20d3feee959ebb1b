set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o trace -- python $R/bench.py --steps 200 --warmup 20 > $O/bench_traced.json 2> $O/bench_traced.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc -o pmc_fetch -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-solve > /dev/null 2> $O/pmc1.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc -o pmc_write -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-solve > /dev/null 2> $O/pmc2.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/pmc -o pmc_mfma -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-solve > /dev/null 2> $O/pmc3.err
PYTHONPATH=$R:$R/tests rocprofv3 --kernel-trace --stats --output-format csv -d $O/solve -o solve -- python $R/profiles/scripts/prof_cfg.py cfg3 > $O/solve.log 2>&1
cd $R
python bench.py > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.json
python profiles/scripts/prof_linearize.py > $O/lin_phases.log 2>&1
MCBA_TIMING=1 python profiles/scripts/prof_workspace.py cfg3 > $O/workspace_cfg3.log 2>&1; grep "calibrate ms" $O/workspace_cfg3.log
python profiles/scripts/prof_lin_cfgs.py cfg2 cfg3 cfg4 cfg5 > $O/lin_cfgs.log 2>&1; cat $O/lin_cfgs.log
python profiles/scripts/prof_scale.py > $O/lin_scale.log 2>&1; tail -8 $O/lin_scale.log
python profiles/scripts/prof_init.py cfg2 cfg3 cfg4 > $O/init.log 2>&1; cat $O/init.log
# (round 2 script, kept for the record: since round 5 the product library reads MCBA_* switches only through mcba_debug_set_switch;
#  MCBA_FUSED is honoured by MCBA_BUILD_VARIANT builds alone)
MCBA_FUSED=1 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_fused.json 2>/dev/null; grep -o '"ms_per_step": [0-9.]*' $O/bench_fused.json | head -1
MCBA_TIMING=1 python profiles/scripts/prof_workspace.py cfg4 2>&1 | grep 'calibrate ms'
MCBA_TIMING=1 python profiles/scripts/prof_workspace.py cfg2 2>&1 | grep 'calibrate ms'
python profiles/scripts/prof_chol_phases.py > $O/chol_phases.log 2>&1; tail -9 $O/chol_phases.log
python profiles/scripts/prof_dispatch.py > $O/dispatch.log 2>&1; tail -16 $O/dispatch.log
python profiles/scripts/prof_solve_repeat.py cfg3 > $O/solve_repeat.log 2>&1; head -4 $O/solve_repeat.log
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; grep -E 'passed|failed' $O/pytest_gpu.log
ls $O $O/pmc
