for eng in queue lane; do for d in 0 1 5; do
export KYHIP_ENGINE=$eng
bash tools/pmc.sh e${eng}d$d q3p0 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY" --spp 64 --steps 2 --warmup 1 --depth $d | grep "SQ_\|render_kernel.*calls\|avg_us" | cut -c1-24,50-140
done; done
