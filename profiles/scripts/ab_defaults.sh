run() { python bench.py --precision fp32 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --also "" 2>&1 | tail -1 | cut -c79-90; }
for rep in 1 2; do
echo "rep$rep default $(run)"
for kv in PSEG_CONV_NARROW=0 PSEG_CONV_DMA32=0 PSEG_BN_MASK=0 PSEG_FUSE_CE_UPSAMPLE=0 PSEG_CONV_NOBAND=1 PSEG_DEFER_SLABS=1 PSEG_GRAPH=1; do
echo "rep$rep $kv $(env $kv python bench.py --precision fp32 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --also "" 2>&1 | tail -1 | cut -c79-90)"
done; done
