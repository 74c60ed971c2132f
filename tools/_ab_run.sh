set -o pipefail
show() { python -c "
import json,sys;d=json.load(open(sys.argv[1]));print(sys.argv[1], round(d['ms_per_step']*1e3,2),'us', round(d['with_prune']['ms_per_step']*1e3,2), {k:round(v['avg_us'],2) for k,v in d['kernels'].items()}, d['roofline']['kernel'], round(d['roofline']['frac'],4))" $1; }
C=$PWD/gcn-over-pruned-trees_amd/csrc
python bench.py --no-cpu-baseline > gpurun_out/b_base.json 2> gpurun_out/b_base.err; show gpurun_out/b_base.json
GCNPT_LIB=$C/libgcnpt_share.so python bench.py --no-cpu-baseline > gpurun_out/b_share.json 2> gpurun_out/b_share.err; show gpurun_out/b_share.json
GCNPT_LIB=$C/libgcnpt_e1.so python bench.py --no-cpu-baseline > gpurun_out/b_e1.json 2> gpurun_out/b_e1.err; show gpurun_out/b_e1.json
python bench.py --no-cpu-baseline --drop 0 > gpurun_out/b_drop0.json 2> gpurun_out/b_drop0.err; show gpurun_out/b_drop0.json
python bench.py --no-cpu-baseline > gpurun_out/b_base2.json 2> gpurun_out/b_base2.err; show gpurun_out/b_base2.json
GCNPT_LIB=$C/libgcnpt_e1.so python bench.py --no-cpu-baseline > gpurun_out/b_e1b.json 2> gpurun_out/b_e1b.err; show gpurun_out/b_e1b.json
