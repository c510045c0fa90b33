#!/bin/bash
# SQ counters of one training step (tools/train_step_bench.py) -> gpurun_out/train_pmc/summary.txt (largest launch per kernel)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/train_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
T="$R/tools/train_step_bench.py --steps 1 --warmup 1"
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES"
Bc="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_TRANS_F32"
Cc="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR GRBM_GUI_ACTIVE"
i=0
for set in "$A" "$Bc" "$Cc"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- python3 $T > $O/p$i.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, os, collections
O=os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out/train_pmc")
best=collections.defaultdict(dict)
for f in glob.glob(O+"/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name=r["Kernel_Name"].split("(")[0][-60:]
        c=r["Counter_Name"]; v=float(r["Counter_Value"])
        key=(name, r.get("Grid_Size","") if "Grid_Size" in r else "")
        if v>best[key].get(c,0): best[key][c]=v
rows=sorted(best.items(), key=lambda kv:-kv[1].get("SQ_WAVE_CYCLES",0))[:24]
with open(O+"/summary.txt","w") as out:
    for (name,grid),c in rows:
        wc=c.get("SQ_WAVE_CYCLES",1)
        out.write(f"{name} grid={grid} wave_cyc={wc:.3g} wait_any={c.get('SQ_WAIT_ANY',0)/wc:.2f} wait_inst={c.get('SQ_WAIT_INST_ANY',0)/wc:.2f} valu={c.get('SQ_ACTIVE_INST_VALU',0)/wc:.2f} lds={c.get('SQ_ACTIVE_INST_LDS',0)/wc:.2f} "
                  f"insts_valu={c.get('SQ_INSTS_VALU',0):.3g} mfma={c.get('SQ_INSTS_MFMA',0):.3g} lds_i={c.get('SQ_INSTS_LDS',0):.3g} vmem={c.get('SQ_INSTS_VMEM',0):.3g} mfma_busy={c.get('SQ_VALU_MFMA_BUSY_CYCLES',0):.3g} "
                  f"bank_conf={c.get('SQ_LDS_BANK_CONFLICT',0):.3g} lds_active={c.get('SQ_LDS_IDX_ACTIVE',0):.3g} wait_lds={c.get('SQ_WAIT_INST_LDS',0):.3g} waves={c.get('SQ_WAVES',0):.3g}\n")
PY
rm -rf $O/p1 $O/p2 $O/p3
cat $O/summary.txt | cut -c1-420
