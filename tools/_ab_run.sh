set -o pipefail
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/gpu_tests.log
show() { python -c "
import json,sys;d=json.load(open(sys.argv[1]));print(sys.argv[1], round(d['ms_per_step']*1e3,2),'us', round(d['with_prune']['ms_per_step']*1e3,2), {k:round(v['avg_us'],2) for k,v in d['kernels'].items()}, d['roofline']['kernel'], round(d['roofline']['frac'],4))" $1; }
python bench.py --no-cpu-baseline > gpurun_out/b1.json 2> gpurun_out/b1.err; show gpurun_out/b1.json
python bench.py --no-cpu-baseline > gpurun_out/b2.json 2> gpurun_out/b2.err; show gpurun_out/b2.json
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
