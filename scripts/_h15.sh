mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_det_grads_gpu.py tests/test_fullsize_gpu.py tests/test_step_gpu.py tests/test_kernels_gpu.py -x -q 2>&1 | tail -3
echo "--- item site, with hint"; EMBED_SITES=item python scripts/prof_kernels.py embed 2>&1 | grep "segmented"
echo "--- item site, no hint";   EMBED_SITES=item EMBED_NO_WCH=1 python scripts/prof_kernels.py embed 2>&1 | grep "segmented"
echo "--- both sites, with hint"; python scripts/prof_kernels.py embed 2>&1 | grep "segmented"
echo "--- both sites, no hint";   EMBED_NO_WCH=1 python scripts/prof_kernels.py embed 2>&1 | grep "segmented"
EMBED_SITES=item bash scripts/prof_embed.sh r05g_item
grep "ss_chunks\|ss_borders" gpurun_out/r05g_item_embed_kernel_trace.md
bash scripts/prof_embed.sh r05g
grep "ss_chunks\|ss_borders" gpurun_out/r05g_embed_kernel_trace.md
