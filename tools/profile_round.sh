# usage (on the GPU box, from the repo root): bash tools/profile_round.sh TAG
# kernel stats + step sequence + HBM traffic (two PMC passes) + MFMA busy (one PMC pass) of the default bench workload
set -e
TAG=$1
R=$PWD; cd /tmp; export TMPDIR=/tmp; cd $R
B="python bench.py --no-cpu-baseline --no-kernel-timing --no-extra"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- $B --steps 10 --warmup 3 > gpurun_out/prof_$TAG.log 2>&1
cp gpurun_out/prof_$TAG/*/*_kernel_stats.csv gpurun_out/${TAG}_kernel_stats.csv
python tools/step_trace.py gpurun_out/prof_$TAG/*/*_kernel_trace.csv --seq > gpurun_out/${TAG}_step_sequence.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- $B --steps 5 --warmup 2 > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- $B --steps 5 --warmup 2 > gpurun_out/pmc_write.log 2>&1
python tools/collect_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write > gpurun_out/${TAG}_traffic.txt
cp profiles/traffic.json gpurun_out/traffic.json
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_mfma -- $B --steps 5 --warmup 2 > gpurun_out/pmc_mfma.log 2>&1
python tools/collect_mfma.py gpurun_out/pmc_mfma > gpurun_out/${TAG}_mfma.txt
cp profiles/mfma_util.json gpurun_out/mfma_util.json
tail -3 gpurun_out/${TAG}_step_sequence.txt; head -12 gpurun_out/${TAG}_traffic.txt; head -14 gpurun_out/${TAG}_mfma.txt
