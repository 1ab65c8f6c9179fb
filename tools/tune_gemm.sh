#!/bin/bash
# Experiment: PyTorch TunableOp over the GEMM shapes of the bench step (library ceiling for the dense stream).
OUT=gpurun_out/tune
mkdir -p $OUT
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_VERBOSE=1
export PYTORCH_TUNABLEOP_FILENAME=$OUT/tunableop_results.csv
export PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=15 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=5
( time timeout 1300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --fresh-tuning ) > $OUT/tune_run.log 2>&1
tail -3 $OUT/tune_run.log | cut -c1-600
ls -la $OUT; wc -l $OUT/tunableop_results*.csv
export PYTORCH_TUNABLEOP_TUNING=0
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --fresh-tuning > $OUT/bench_tuned.log 2>&1; tail -1 $OUT/bench_tuned.log | cut -c1-400
