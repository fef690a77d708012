#!/bin/bash
# rocprofv3 counter passes over tools/gemm_pmc_probe.py (separate --pmc runs, kernel-trace only); results under gpurun_out/pmc_gemm/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SHAPE=${SHAPE:-"2048 3840 1280 0"}
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
           "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum TCP_TCP_LATENCY_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F16" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmc_gemm/p$i -o p --output-format csv -- python3 $R/tools/gemm_pmc_probe.py $SHAPE 10 > $R/gpurun_out/pmc_gemm/p$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
tot=collections.defaultdict(float); n=collections.defaultdict(int)
for f in glob.glob(R+"/gpurun_out/pmc_gemm/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_f16" in r["Kernel_Name"]:
            tot[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
for k in sorted(tot): print(f"{k:36s} {tot[k]/max(1,n[k]):16.1f} per launch ({n[k]} rows)")
PY
