#!/bin/bash
# usage (gpurun): tools/variants/run.sh [reps]  -- bench.py's kernel timings with the product library and every variant in
# this directory, interleaved `reps` times so that clock / box differences show up as spread
cd $GRAFT_REPO_ROOT
one() { python bench.py --steps 40 --no-extras --no-cpu-baseline --no-live-pmc --no-config5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-40s value %.0f  step %.4f ms  pixel %.4f  prep %.4f  lift %.4f' % (sys.argv[1], d['value'], d['ms_per_step'], d['kernels_ms']['pixel'], d['kernels_ms']['prep'], d['kernels_ms']['lift']))" "$1"; }
for rep in $(seq 1 ${1:-2}); do
one product
for f in tools/variants/lib_*.so; do CELESTE_MI355X_LIB=$PWD/$f one $f; done
for g in ${GROUPS_TO_TRY:-}; do CELESTE_CHUNK_GROUP=$g one "product G=$g"; done
done
