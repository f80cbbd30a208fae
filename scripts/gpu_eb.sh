python -m pytest tests/test_kernels_gpu.py -q -k "fused_encoder" 2>&1 | tail -2
python scripts/bench_encbwd.py 2>&1 | tail -1
for v in NOMFMA NODH NOFETCH; do echo -n "$v: "; CLSR_LIB=$PWD/build/abl/lib_eb_$v.so python scripts/bench_encbwd.py 2>&1 | tail -1; done
python -m pytest tests/test_step_gpu.py -q -x 2>&1 | tail -2
bash scripts/r3_ab.sh CLSR_NO_ENC_BWD_FUSED "" 1
bash scripts/prof_step.sh r3eb > /dev/null 2>&1
sed -n '/rnn_multi_bwd/,$p' gpurun_out/r3eb_timeline.txt | cut -c1-110
