# dev: how much is a resident workgroup worth? extra dynamic LDS per k_spconv2 launch -> fewer workgroups per CU
for e in 0 12000 40000 90000; do echo "IRX_S2_EXTRA_LDS=$e"; IRX_S2_EXTRA_LDS=$e python tools/conv_microbench.py 16 2>/dev/null | grep "stride  [24]" | cut -c1-75; done
