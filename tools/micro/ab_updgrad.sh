# dev A/B: stride-2 data-gradient tiled by parent rows (k_updgrad) vs through k_spconv2 over the transposed child table
for v in 1 0; do echo "IRX_UPDGRAD=$v"; IRX_UPDGRAD=$v IRX_BENCH_LAYERS=1 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-alt-dtype 2>&1 >/dev/null | grep -E "^dgrad +[0-9]+ +8 "; done
bash tools/micro/ab_env.sh IRX_UPDGRAD 0 1
