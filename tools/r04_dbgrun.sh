cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=8
AMD_LOG_LEVEL=3 timeout -k 5 200 python bench.py --leg configs_4_share --tail-ring 128 > gpurun_out/r04_fault2.out 2> gpurun_out/r04_fault2.err
echo "exit=$?"
grep -n "Memory access fault" gpurun_out/r04_fault2.err | head -3
grep -n "ShaderName" gpurun_out/r04_fault2.err | tail -6 | cut -c1-220
tail -c 400 gpurun_out/r04_fault2.out
ls -la gpucore* 2>/dev/null | head -3
