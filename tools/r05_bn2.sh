cd $GRAFT_REPO_ROOT
O=gpurun_out/r05bn2; mkdir -p $O
IRX_BN_SLICE_BYTES=0 IRX_BN_LASTBLOCK=0 timeout 300 python tools/bn_microbench.py > $O/old.txt 2>&1
IRX_BN_SLICE_BYTES=0 IRX_BN_LASTBLOCK=1 timeout 300 python tools/bn_microbench.py > $O/lastblock.txt 2>&1
paste -d'\n' $O/old.txt $O/lastblock.txt | grep -v amdgpu | grep bf16
timeout 200 python tools/mlp_microbench.py 2>&1 | grep -v amdgpu | tee $O/mlp.txt
IRX_BN_LASTBLOCK=1 timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_bf16_gpu.py -x -q -k "batchnorm or bn or executor or layer_by_layer or mlp" 2>&1 | tail -3
