set -o pipefail
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/gpu_tests.log
python bench.py --no-cpu-baseline > gpurun_out/b_n4.json 2> gpurun_out/b_n4.err; python -c "
import json;d=json.load(open('gpurun_out/b_n4.json'));print(round(d['ms_per_step']*1e3,2), round(d['with_prune']['ms_per_step']*1e3,2), round(d['with_cached_trees']['ms_per_step']*1e3,2))"; tail -3 gpurun_out/b_n4.err
