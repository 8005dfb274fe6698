#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/r2d; mkdir -p $o
timeout 900 python -m pytest -q -p no:cacheprovider -m gpu tests/test_gpu_binning.py tests/test_gpu_raster.py -x > $o/tests1.log 2>&1
echo "rc=$?" >> $o/tests1.log; tail -15 $o/tests1.log
timeout 600 python bench.py --gpus 1 --steps 100 --warmup 10 --no-secondary --no-cpu-baseline > $o/bench_bucket.json 2> $o/bench_bucket.err
ADGS_BINNING=sort timeout 600 python bench.py --gpus 1 --steps 100 --warmup 10 --no-secondary --no-cpu-baseline > $o/bench_sort.json 2> $o/bench_sort.err
ADGS_V2_PPL=2 timeout 600 python bench.py --gpus 1 --steps 100 --warmup 10 --no-secondary --no-cpu-baseline > $o/bench_ppl2.json 2> $o/bench_ppl2.err
for ct in 6 8; do ADGS_CELL_TILES=$ct timeout 600 python bench.py --gpus 1 --steps 100 --warmup 10 --no-secondary --no-cpu-baseline > $o/bench_bucket_ct$ct.json 2> $o/bench_bucket_ct$ct.err; done
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --config C5 > $o/bench_c5.json 2> $o/bench_c5.err
for f in $o/*.err; do tail -2 $f; done
