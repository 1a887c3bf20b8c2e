set -e
cd /root/repo; mkdir -p gpurun_out
[ -n "$SKIP_TESTS" ] || timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/dig_tests.log 2>&1 || { tail -30 gpurun_out/dig_tests.log; exit 1; }
[ -n "$SKIP_TESTS" ] || tail -3 gpurun_out/dig_tests.log
for w in c2 c4full c5 c3; do timeout -k 10 200 python bench.py --config $w --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/dig_bench_$w.json 2>gpurun_out/dig_bench_$w.err || { tail -5 gpurun_out/dig_bench_$w.err; exit 1; }; cat gpurun_out/dig_bench_$w.json; done
