for rep in 1 2; do
echo "--- new"; python scripts/bench_rnn.py 2>&1 | tail -7
echo "--- old"; CLSR_LIB=$PWD/build/abl/lib_rnn_old.so python scripts/bench_rnn.py 2>&1 | tail -7
done
for rep in 1 2; do
for lib in "" $PWD/build/abl/lib_rnn_old.so; do echo -n "lib=$lib "; CLSR_LIB=$lib python bench.py --no-cpu-baseline --no-catalogue --no-extra --steps 40 2>/dev/null | grep '^{"metric' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step")'; done
done
