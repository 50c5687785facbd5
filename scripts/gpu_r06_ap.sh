#!/bin/bash
# pack2_kernel at six / seven workgroups per CU (PACK2_WAVES) against the tree's five: same box, rotating, -8 and -5
TAG=$1
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash scripts/gpu_abn.sh 3 --no-api --no-decode --no-clock --frames 65536 2>&1 | grep -v amdgpu.ids | tee $OUT/abn_level8.txt
bash scripts/gpu_abn.sh 3 --no-api --no-decode --no-clock --frames 65536 --level 5 2>&1 | grep -v amdgpu.ids | tee $OUT/abn_level5.txt
