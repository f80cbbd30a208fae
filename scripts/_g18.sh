mkdir -p gpurun_out/r5
timeout 1200 python -m pytest tests/test_att_bwd_x3_gpu.py tests/test_step_gpu.py tests/test_kernels_gpu.py tests/test_bf16_gpu.py -q > gpurun_out/r5/x3tests.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/r5/x3tests.log
timeout 300 python scripts/bench_att_bwd.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/bench_att_bwd6.txt | grep "pass 1"
CLSR_LIB=$PWD/build/abl/lib_l1p1occ1.so timeout 300 python scripts/bench_att_bwd.py 2>&1 | grep "x3    l1 pass 1" | sed 's/^/occ1 /'
B="python bench.py --no-cpu-baseline --no-catalogue --no-extra --steps 40"
for rep in 1 2 3; do
echo "default(occ2) $($B 2>&1 | grep -E timed)"
echo "occ1          $(CLSR_LIB=$PWD/build/abl/lib_l1p1occ1.so $B 2>&1 | grep -E timed)"
done
bash scripts/prof_step.sh r05k_fp32
grep "att_out_fwd\|att_l1_bwd_x3_kernelILi5ELi3ELb0" gpurun_out/r05k_fp32_timeline.txt
