mkdir -p gpurun_out/r4o
for cfg in "easy 0" "easy 3000" "hg19like 3000"; do
  set -- $cfg
  timeout -k 10 400 python bench.py --genome $1 --contigs $2 --steps 5 --warmup 2 --no-extra --ref-sample 0 > gpurun_out/r4o/se_$1_$2.json 2> gpurun_out/r4o/se_$1_$2.err || { echo "failed $cfg"; exit 1; }
  grep -h "heavy pass" gpurun_out/r4o/se_$1_$2.err
done
timeout -k 10 400 python bench.py --mode pe --genome easy --contigs 3000 --steps 3 --warmup 1 --no-extra > gpurun_out/r4o/pe_easy_3000.json 2> gpurun_out/r4o/pe_easy_3000.err || echo "pe failed"
grep -h "ms/step" gpurun_out/r4o/pe_easy_3000.err
