cd $GRAFT_REPO_ROOT
O=gpurun_out/r05bn; mkdir -p $O
IRX_BN_SLICE_BYTES=0 IRX_BN_LASTBLOCK=0 timeout 300 python tools/bn_microbench.py > $O/old.txt 2>&1
IRX_BN_SLICE_BYTES=0 timeout 300 python tools/bn_microbench.py > $O/lastblock.txt 2>&1
IRX_BN_SLICE_BYTES=100000000 timeout 300 python tools/bn_microbench.py > $O/slice_all.txt 2>&1
paste -d'\n' $O/old.txt $O/lastblock.txt $O/slice_all.txt | grep -v amdgpu
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_bf16_gpu.py -x -q 2>&1 | tail -5
