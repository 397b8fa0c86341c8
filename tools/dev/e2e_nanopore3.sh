#!/bin/bash
# dev helper (GPU box): the first three volumes of config 5 (320 000 ONT-style reads) through the CLI, -j 0 -x 1, with the trace lines per cell
D=/dev/shm/e2e_n3; rm -rf $D; mkdir -p $D
mecat_amd/bin/synth_reads $D/reads.fa ${1:-320000} 20000 0.12 1300000000 5 1 > /dev/null 2>&1
sleep 2
t=$(date +%s%N)
MECAT_TRACE=1 mecat_amd/bin/mecat2pw -j ${2:-0} -x 1 -g 1 -d $D/reads.fa -o $D/out.txt -w $D/w -t 32 > $D/log 2> $D/err
echo "rc=$? wall $(( ($(date +%s%N) - t) / 1000000 )) ms, $(wc -l < $D/out.txt) lines"
grep "takes\|\[trace\]" $D/err $D/log | sed 's/^[^:]*://' | grep -v "^\[mecat_hip\] [a-z_0-9]* " | head -80
rm -rf $D
