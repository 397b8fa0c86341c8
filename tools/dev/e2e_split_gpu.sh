#!/bin/bash
# dev helper (GPU box): the split stage of a config-2 sized FASTA through the CLI, host packer against device packer (MECAT_HIP_SPLIT=gpu)
N=${1:-100000}
D=/tmp/e2e_sp; rm -rf $D; mkdir -p $D
mecat_amd/bin/synth_reads $D/reads.fa $N 15000 0.15 50000000 1 > /dev/null 2>&1
sync
for mode in host gpu host gpu; do
  rm -rf $D/w $D/out.*
  t=$(date +%s%N)
  if [ $mode = gpu ]; then export MECAT_HIP_SPLIT=gpu; else unset MECAT_HIP_SPLIT; fi
  MECAT_TRACE=1 mecat_amd/bin/mecat2pw -j 0 -d $D/reads.fa -o $D/out.txt -w $D/w -t 32 > $D/log 2> $D/err
  echo "== $mode rc=$? wall $(( ($(date +%s%N) - t) / 1000000 )) ms, $(wc -l < $D/out.txt) lines, vol0 sha $(sha256sum $D/w/vol0 | cut -c1-12)"
  grep "\[trace\] split\|split_raw\|ctx_create" $D/err | tr '\n' ';'; echo
done
