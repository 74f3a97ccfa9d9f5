set -e
python -m pytest tests -x -q -m gpu -k "tsrm or fused_train_step or encoder" 2>&1 | tail -3
bash tools/ab_env.sh ECHR_TSRM_HEADS 0 1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/headprof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu --no-native --no-roofline > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,os
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/headprof/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'head' in r['Name'] or 'softmax' in r['Name'] or 'gemm_f32' in r['Name']: print(r['Name'][:60], r['Calls'], r['AverageNs'])
PY
