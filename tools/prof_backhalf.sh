#!/bin/bash
# usage: tools/prof_backhalf.sh <tag> [bench args...]
# Stall attribution of the HBM-bound back half (LDS reduce, key-major writer): six counter passes over bench.py, one block of
# counters each (SQ wait buckets / vector-memory issue / LDS queue / TCP / TCC), merged per kernel into gpurun_out/<tag>_backhalf_sq.json.
# A pass whose counters the profiler refuses is skipped (its log stays), the others still count.
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_LDS"
P2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
P3="SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
P4="TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum"
P5="TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum"
P6="TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_LEVEL_sum"
P7="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_WRREQ_LEVEL_sum"
i=0
CSVS=""
for P in "$P1" "$P2" "$P3" "$P4" "$P5" "$P6" "$P7"; do
  i=$((i+1))
  rm -rf $OUT/${TAG}_bh$i
  if rocprofv3 --pmc $P --output-format csv -d $OUT/${TAG}_bh$i -- python3 $ROOT/bench.py --steps 2 --warmup 1 --cpu-groups 0 --e2e 0 "$@" > $OUT/${TAG}_bh$i.log 2>&1; then
    CSVS="$CSVS $(find $OUT/${TAG}_bh$i -name '*counter_collection.csv')"
    echo "pass $i ok"
  else
    echo "pass $i FAILED ($P)"; tail -3 $OUT/${TAG}_bh$i.log
  fi
done
python3 $ROOT/tools/sq_counters.py $OUT/${TAG}_backhalf_sq.json $CSVS
for j in 1 2 3 4 5 6 7; do rm -rf $OUT/${TAG}_bh$j; done
