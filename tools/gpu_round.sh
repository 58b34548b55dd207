#!/bin/bash
# one GPU-box visit: GPU test-suite, bench line, fixture identity table -> gpurun_out/
tag=${1:-r02_a}
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x --timeout=1200 > gpurun_out/${tag}_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/${tag}_tests.log
tail -5 gpurun_out/${tag}_tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -c 3000 gpurun_out/${tag}_bench.json
python tools/fixture_identity.py > gpurun_out/${tag}_identity.json 2> gpurun_out/${tag}_identity.err
