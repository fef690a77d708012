#!/bin/bash
# rocprofv3 counter passes over tools/attn_pmc_probe.py (separate --pmc runs, kernel-trace only); results under gpurun_out/pmc_attn/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SHAPE=${SHAPE:-"2 10 4096"}
mkdir -p $R/gpurun_out/pmc_attn
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VALU" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmc_attn/p$i -o p --output-format csv -- python3 $R/tools/attn_pmc_probe.py $SHAPE 10 > $R/gpurun_out/pmc_attn/p$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
tot=collections.defaultdict(float); n=collections.defaultdict(int)
for f in glob.glob(R+"/gpurun_out/pmc_attn/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attention_f16" in r["Kernel_Name"]:
            tot[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
for k in sorted(tot): print(f"{k:36s} {tot[k]/max(1,n[k]):16.1f} per launch ({n[k]} rows)")
PY
find $R/gpurun_out/pmc_attn -name "*counter_collection.csv" -delete; find $R/gpurun_out/pmc_attn -name "*kernel_trace.csv" -delete
