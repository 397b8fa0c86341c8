#!/bin/bash
# dev helper (GPU box): the config-5 FASTA (40 Gbase, 19 volumes) through the CLI's split, host threads against MECAT_HIP_SPLIT=gpu
# (decide mhip_volume_pack by measurement at the scale it was meant for).  Only `-j 0` up to the first cell matters here:
# the split's own timer lines are what is compared, the run is killed once they are printed.
D=/dev/shm/split_c5; rm -rf $D; mkdir -p $D
mecat_amd/bin/synth_reads $D/reads.fa ${1:-2000000} 20000 0.12 1300000000 5 1 > /dev/null 2>&1
ls -la $D/reads.fa
for mode in host gpu host gpu; do
  rm -rf $D/w $D/out.txt; sleep 2
  if [ $mode = gpu ]; then export MECAT_HIP_SPLIT=gpu; else unset MECAT_HIP_SPLIT; fi
  MECAT_TRACE=1 timeout 300 mecat_amd/bin/mecat2pw -j 0 -x 1 -d $D/reads.fa -o $D/out.txt -w $D/w -t 32 > $D/log 2> $D/err &
  pid=$!
  while kill -0 $pid 2>/dev/null && ! grep -q "split_raw_dataset\] takes" $D/err $D/log 2>/dev/null; do sleep 0.2; done
  kill $pid 2>/dev/null; wait $pid 2>/dev/null
  echo "== $mode: $(grep -h "split_raw_dataset\] takes" $D/err $D/log) ; pack $(grep -h 'split: pack' $D/err | awk '{s+=$(NF-1)} END{print s}') s, dump $(grep -h 'split: dump' $D/err | awk '{s+=$(NF-1)} END{print s}') s, scan $(grep -h 'split: scan' $D/err | awk '{s+=$(NF-1)} END{print s}') s"
done
rm -rf $D
