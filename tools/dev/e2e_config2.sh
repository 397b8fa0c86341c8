#!/bin/bash
# dev helper (GPU box): config-2 sized FASTA through the CLI with MECAT_TRACE timers.   bash tools/dev/e2e_config2.sh [task] [reads]
T=${1:-1}; N=${2:-100000}
D=/tmp/e2e_c2; rm -rf $D; mkdir -p $D
mecat_amd/bin/synth_reads $D/reads.fa $N 15000 0.15 50000000 1 > /dev/null 2>&1
sync
for rep in 1 2; do
rm -rf $D/w $D/out.*
t=$(date +%s%N)
MECAT_TRACE=1 mecat_amd/bin/mecat2pw -j $T -g 1 -d $D/reads.fa -o $D/out.txt -w $D/w -t 32 > $D/log 2> $D/err
echo "rc=$? wall $(( ($(date +%s%N) - t) / 1000000 )) ms, $(wc -l < $D/out.txt) lines, $(stat -c %s $D/out.txt) bytes"
done
grep "takes\|\[trace\]" $D/err $D/log | sed 's/^[^:]*://' | grep -v "^\[mecat_hip\] [a-z_0-9]* " | awk '{a[$0]++} END{for(k in a) print a[k]"x "k}' | sort -k2 | head -60
