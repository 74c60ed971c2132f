#!/bin/bash
# Developer aid (run on the GPU box through gpurun): A/B of library builds on the bench step.  Build a variant with other compile-time
# switches into csrc/libgcnpt_<name>.so (copy csrc/ + include/ to a scratch directory, `make CXXFLAGS="... -DGCNPT_A_AHEAD=3"`, copy the
# .so back), then
#     LIBS="libgcnpt.so libgcnpt_<name>.so" BARGS="--batch 128 ..." bash tools/ab_libs.sh
# runs bench.py on each library in turn (GCNPT_LIB), twice, and prints the step and per-launch times.  Variant libraries are not
# committed (*.so is ignored).
set -o pipefail
LIBS="${LIBS:-libgcnpt.so}"
for rep in 1 2; do
for lib in $LIBS; do
  GCNPT_LIB=$PWD/gcn-over-pruned-trees_amd/csrc/$lib timeout -k 10 200 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-pooled-only --no-secondary --no-secondary-shapes --no-floor $BARGS > gpurun_out/b.json 2>gpurun_out/b.err || { tail -3 gpurun_out/b.err; exit 1; }
  python -c "
import json;d=json.loads(open('gpurun_out/b.json').read().strip().split('\n')[-1]);print('$lib', round(d['ms_per_step']*1e3,2), {k:round(v['avg_us'],2) for k,v in d['kernels'].items()})"
done
done
