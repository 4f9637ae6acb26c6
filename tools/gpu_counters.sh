#!/bin/bash
# usage: gpu_counters.sh <tag> "<bench args>"   (through gpurun)
# SQ / LDS / MFMA counter passes of one bench.py workload (separate rocprofv3 --pmc runs, kernel-trace
# only beside them), condensed per kernel by tools/summarise_counters.py into
# gpurun_out/ctr_<tag>/<tag>_counters.json (copy to profiles/).
export TMPDIR=/tmp
tag=$1; args=$2
out=gpurun_out/ctr_$tag
rm -rf $out; mkdir -p $out
passA="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
passB="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
passC="SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_BUSY_CU_CYCLES SQ_INSTS_SALU"
i=0
for ctrs in "$passA" "$passB" "$passC"; do
  i=$((i+1))
  timeout -k 10 600 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out/pass$i -- python3 bench.py --no-cpu --steps 4 --warmup 1 --repeats 1 --both-geometries 0 --traffic none $args > $out/pass$i.log 2>&1 || { tail -5 $out/pass$i.log; exit 1; }
  echo "counter pass $i done"
done
python3 tools/summarise_counters.py $tag "$args"
