cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
cd $R
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d gpurun_out/pmc/a -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc/a.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS -d gpurun_out/pmc/b -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc/b.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE -d gpurun_out/pmc/c -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc/c.log 2>&1
for d in a b c; do f=$(find gpurun_out/pmc/$d -name "*.db" | head -1); echo "== $d $f"; python tools/pmc_summary.py $f 2>&1 | grep -E "match_grp|entropy" ; done
find gpurun_out/pmc -name "*.db" -delete
