#!/bin/bash
# Collect the round's profiler evidence on the GPU box (run from the repo root; writes gpurun_out/$ROUND/, default r04):
#   kernel-trace stats of the bench command (1 stream and default), SQ / TCC counter passes of the same command (--streams 1)
#   and of the step-granular SDE step at 786 432 rows.  Counter passes are separate runs with --kernel-trace only
#   (MI355X_MICROARCH.md: 8 SQ slots per pass; FETCH_SIZE and WRITE_SIZE do not fit one pass).
#   bash tools/collect_counters.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${ROUND:-r04}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --steps 20 --warmup 3 --windows 1 --no-cpu-baseline --no-train-step --no-secondary"
P="$R/bench.py --steps 2 --warmup 1 --windows 1 --no-cpu-baseline --no-train-step --no-secondary --streams 1"
S="$R/tools/sde_step_bench.py 786432 30"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks1 -- python3 $B --streams 1 > $O/ks1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks3 -- python3 $B > $O/ks3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kss -- python3 $S > $O/kss.log 2>&1
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES"
Bc="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_TRANS_F32"
Cc="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR GRBM_GUI_ACTIVE"
i=0
for set in "$A" "$Bc" "$Cc" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_fwd_$i -- python3 $P > $O/pmc_fwd_$i.log 2>&1
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_sde_$i -- python3 $S > $O/pmc_sde_$i.log 2>&1
done
# BASELINE configs[4] (stress shape, bf16 hidden state): kernel stats + HBM counters of tools/stress_forward.py
T="$R/tools/stress_forward.py --storage bf16 --iters 3"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kst -- python3 $T > $O/kst.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_stress_4 -- python3 $T > $O/pmc_stress_4.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_stress_5 -- python3 $T > $O/pmc_stress_5.log 2>&1
python3 $R/tools/stress_forward.py --check > $O/stress_forward.json 2> $O/stress_forward.err
python3 $R/tools/train_step_bench.py > $O/train_step.log 2>&1
python3 $R/tools/train_step_bench.py --config config4 > $O/train_step_config4.log 2>&1
cd $R
cp $(find $O/kst -name "*kernel_stats.csv" | head -1) $O/kernel_stats_stress_bf16.csv
cp $(find $O/ks1 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_streams1.csv
cp $(find $O/ks3 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_default_streams3.csv
cp $(find $O/kss -name "*kernel_stats.csv" | head -1) $O/kernel_stats_sde_step.csv
python3 tools/counter_summary.py $O $O/sq_counters.md $O/traffic.json
rm -rf $O/ks1 $O/ks3 $O/kss $O/kst
find $O -name "*kernel_trace.csv" -delete
find $O -name "*agent_info.csv" -delete
tail -3 $O/sq_counters.md
