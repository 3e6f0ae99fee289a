#!/bin/bash
# Timeline of bin/walt on a RAM-backed FASTQ (GPU box): builds the hg19-like index, writes it and 20 M reads to /dev/shm,
# then runs bin/walt -v under `time` twice (cold/warm page cache is the same on tmpfs) with strace -c when available.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
OUT=gpurun_out/e2e_probe; mkdir -p $OUT
python3 - <<'PY'
import os, sys, time
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import torch, synth, walt_amd, bench
dev = torch.device("cuda", 0)
g, lens, names = synth.make_genome(torch, dev, 1.0, seed=2, kind="hg19like")
idx = walt_amd.Index.build_device(g.data_ptr(), lens, names, device=0, strands=walt_amd.STRANDS_CT)
os.makedirs("/dev/shm/walt_probe", exist_ok=True)
t0 = time.time(); idx.write("/dev/shm/walt_probe/hg.dbindex"); print("index written %.1f s" % (time.time() - t0))
for s in ("_GA10", "_GA11"): open("/dev/shm/walt_probe/hg.dbindex" + s, "wb").close()
b, _ = synth.make_reads(torch, dev, g, 20_000_000, 100, seed=1000)
bench.write_fastq("/dev/shm/walt_probe/r.fastq", b.cpu().numpy(), 20_000_000, 100)
idx.close()
PY
for i in 1 2; do
  time walt_amd/bin/walt -i /dev/shm/walt_probe/hg.dbindex -r /dev/shm/walt_probe/r.fastq -o /dev/shm/walt_probe/out.sam -m 6 -b 5000 -a -u -sam -t 16 -N 10000000 -v > $OUT/run$i.log 2>&1
  grep -E "walt_amd" $OUT/run$i.log
  rm -f /dev/shm/walt_probe/out.sam*
done
rm -rf /dev/shm/walt_probe
