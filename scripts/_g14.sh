B="python bench.py --no-cpu-baseline --no-catalogue --no-extra --steps 40"
for rep in 1 2 3; do
echo "default         $($B 2>&1 | grep -E timed)"
echo "aux->dw0        $(CLSR_AUX_ALIAS=@dw0 $B 2>&1 | grep -E timed)"
echo "aux,g2->dw0     $(CLSR_AUX_ALIAS=@dw0 CLSR_G2_STREAM=@dw0 $B 2>&1 | grep -E timed)"
echo "g2->dw0         $(CLSR_G2_STREAM=@dw0 $B 2>&1 | grep -E timed)"
done
CLSR_AUX_ALIAS=@dw0 bash scripts/prof_step.sh r05j_fp32_auxdw0
echo "bf16 default    $($B --precision bf16 2>&1 | grep -E timed)"
echo "bf16 aux->dw0   $(CLSR_AUX_ALIAS=@dw0 $B --precision bf16 2>&1 | grep -E timed)"
