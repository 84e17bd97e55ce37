# usage: bash tools/pmc_lds.sh "<python script and args>"  -- LDS bank-conflict cycles per LDS cycle, per kernel (one PMC pass)
R=$PWD; cd /tmp; export TMPDIR=/tmp; cd $R
rm -rf gpurun_out/pmc_lds
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_lds -- python $1 > gpurun_out/pmc_lds.log 2>&1
python - <<'P'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_lds/*/*_counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:44]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
for k, c in acc.items():
    if c["SQ_ACTIVE_INST_LDS"] > 0:
        print(f"{k:44s} n={n[k]:4d} conflict/idx_active={c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_LDS_IDX_ACTIVE'], 1):5.2f} conflict/inst_lds={c['SQ_LDS_BANK_CONFLICT'] / c['SQ_ACTIVE_INST_LDS']:5.2f} wait_lds/wave={c['SQ_WAIT_INST_LDS'] / c['SQ_WAVE_CYCLES']:5.2f} wait_inst/wave={c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']:5.2f}")
P
