R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rm -rf $R/gpurun_out/gpmc_$tag
  timeout -k 10 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/gpmc_$tag -- python3 $R/tools/gemm_one.py $GEMM_ARGS > $R/gpurun_out/gpmc_$tag.log 2>&1 || echo "group failed: $grp"
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob('$R/gpurun_out/gpmc_*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if 'gemm' in row['Kernel_Name']:
                acc[row['Counter_Name']].append(float(row['Counter_Value']))
        for k, v in acc.items():
            print('%-32s per launch %.4g  (launches %d)' % (k, sum(v) / len(v), len(v)))
PY
