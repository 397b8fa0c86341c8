#!/bin/bash
# development: stage times of the config-3 run (500 000 reads x 12 kb, three volumes, -j 0) through the CLI
D=/dev/shm/c3; rm -rf $D; mkdir -p $D
mecat_amd/bin/synth_reads $D/reads.fa 500000 12000 0.15 200000000 3 0 > /dev/null 2>&1
ls -la $D/reads.fa | awk '{print "fasta bytes", $5}'
t=$(date +%s%N)
MECAT_TRACE=1 mecat_amd/bin/mecat2pw -j ${TASK:-0} -d $D/reads.fa -o $D/out.txt -w $D/w -t 64 > $D/log 2> $D/err
echo "rc=$? wall $(( ($(date +%s%N) - t) / 1000000 )) ms, $(wc -l < $D/out.txt) lines"
grep -E "trace\]|takes" $D/err | grep -v "idx trace\|dw trace" | sed 's/\[trace\] //' | head -60
rm -rf $D
