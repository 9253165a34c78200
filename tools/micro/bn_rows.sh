for r in 0 512 1024 2048; do echo "IRX_BN_ROWS=$r"; IRX_BN_ROWS=$r timeout 200 python tools/bn_microbench.py 2>&1 | grep "n=" | sed 's/| apply .*| bwd /| bwd /' | cut -c1-170; done
