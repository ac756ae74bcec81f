#!/bin/bash
# PMC passes over the forward blend kernel (tools/run_raster.py, C3, FAST): each counter group in its own timeout 500 rocprofv3 run
# (--pmc only, no trace domains).  Usage on the GPU box: bash tools/pmc_fwd.sh TAG  -> gpurun_out/pmc_TAG.txt
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG.txt
: > $OUT
cd /tmp && export TMPDIR=/tmp
for G in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_INST_ANY"; do
  D=/tmp/pmc_$RANDOM
  timeout 500 rocprofv3 --pmc $G -d $D -o pmc --output-format csv -- python $ROOT/tools/run_raster.py ${RASTER_ARGS:---config C3 --iters 3 --sparse 2} > /tmp/pmc_run.log 2>&1
  echo "## $G" >> $OUT
  python $ROOT/tools/pmc_summary.py $D ${KERNEL:-k_render_fwd} >> $OUT 2>&1 || tail -3 /tmp/pmc_run.log >> $OUT
  rm -rf $D
done
cat $OUT
