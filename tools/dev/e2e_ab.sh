#!/bin/bash
# development: wall clock of the CLI at config 2, runs spaced by 3 s, with the trace lines of each
D=/dev/shm/e2e_ab; rm -rf $D; mkdir -p $D
mecat_amd/bin/synth_reads $D/reads.fa 100000 15000 0.15 50000000 2 > /dev/null 2>&1
for rep in 1 2 3; do
rm -rf $D/w $D/out.*
sleep 3
t=$(date +%s%N)
MECAT_TRACE=1 mecat_amd/bin/mecat2pw -j ${TASK:-0} -d $D/reads.fa -o $D/out.txt -w $D/w -t 32 > $D/log 2> $D/err
echo "rc=$? wall $(( ($(date +%s%N) - t) / 1000000 )) ms: $(grep -E 'trace|takes' $D/err | sed 's/\[trace\] //' | tr '\n' ';' | tr -s ' ')"
done
rm -rf $D
