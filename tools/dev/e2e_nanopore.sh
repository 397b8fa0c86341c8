#!/bin/bash
# dev helper (GPU box): nanopore mode end to end, ours vs the unmodified reference on the same FASTA.
#   bash tools/dev/e2e_nanopore.sh <reads> <len> <genome> <out.txt>
set -u
N=${1:-40000}; L=${2:-10000}; G=${3:-13000000}; OUT=${4:-gpurun_out/e2e_nanopore.txt}
D=/tmp/e2e_ont; rm -rf $D; mkdir -p $D
mecat_amd/bin/synth_reads $D/reads.fa $N $L 0.12 $G 11 1 > /dev/null 2>&1
sync
ms() { echo $(( ($(date +%s%N) - $1) / 1000000 )); }
{
echo "# nanopore mode (-x 1 -j 1 -g 1) end to end, $N reads x $L bp @ 12 %, genome $G, $(stat -c %s $D/reads.fa) byte FASTA, host $(nproc) threads"
t=$(date +%s%N); mecat_amd/bin/mecat2pw -j 1 -x 1 -g 1 -d $D/reads.fa -o $D/hip.m4 -w $D/w1 -t 32 > $D/hip.log 2> $D/hip.err; echo "mecat_amd/bin/mecat2pw rc=$? wall $(ms $t) ms"
grep "takes" $D/hip.err $D/hip.log | sed 's/^[^:]*://' | tr '\n' ' '; echo
t=$(date +%s%N); timeout 1500 oracle/_ref/mecat2pw -j 1 -x 1 -g 1 -d $D/reads.fa -o $D/ref.m4 -w $D/w2 -t $(nproc) > $D/ref.log 2> $D/ref.err; echo "oracle/_ref/mecat2pw rc=$? wall $(ms $t) ms"
grep "takes" $D/ref.err $D/ref.log | sed 's/^[^:]*://' | tr '\n' ' '; echo
echo "lines: $(wc -l < $D/hip.m4) / $(wc -l < $D/ref.m4); md5 of the sorted output: $(sort $D/hip.m4 | md5sum | cut -d' ' -f1) / $(sort $D/ref.m4 | md5sum | cut -d' ' -f1)"
} > $OUT 2>&1
cat $OUT
