#!/bin/bash
# dev helper (GPU box): the 2.37 Gbase / two-volume input of profiles/r01_e2e_two_volumes.txt, ours only; the md5 sums recorded
# there for the reference's output identify identical results without re-running the reference (8 and 10 minutes).
D=/tmp/e2e_2v; rm -rf $D; mkdir -p $D
mecat_amd/bin/synth_reads $D/reads.fa 150000 15000 0.15 75000000 3 > /dev/null 2>&1
sync
for T in 0 1; do
  rm -rf $D/w
  t=$(date +%s%N)
  mecat_amd/bin/mecat2pw -j $T -d $D/reads.fa -o $D/out$T.txt -w $D/w -t 32 > $D/log$T 2> $D/err$T
  echo "-j $T: rc=$? wall $(( ($(date +%s%N) - t) / 1000000 )) ms, $(wc -l < $D/out$T.txt) lines, md5 of the sorted output $(sort $D/out$T.txt | md5sum | cut -d' ' -f1)"
  grep "takes" $D/err$T $D/log$T | sed 's/^[^:]*://' | tr '\n' ' '; echo
done
