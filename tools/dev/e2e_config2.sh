#!/bin/bash
# dev helper (GPU box): config-2 sized FASTA through the CLI with MECAT_TRACE timers, every run on a device that has had time to take its
# memory back from the run before.   bash tools/dev/e2e_config2.sh [tasks="1 0"] [reads]
TASKS=${1:-"1 0"}; N=${2:-100000}
D=/dev/shm/e2e_c2; rm -rf $D; mkdir -p $D
mecat_amd/bin/synth_reads $D/reads.fa $N 15000 0.15 50000000 2 > /dev/null 2>&1
for T in $TASKS; do for rep in 1 2; do
sleep 4
rm -rf $D/w $D/out.*
t=$(date +%s%N)
MECAT_TRACE=1 mecat_amd/bin/mecat2pw -j $T -g 1 -d $D/reads.fa -o $D/out.txt -w $D/w -t 32 > $D/log 2> $D/err
echo "== -j $T run $rep: rc=$? wall $(( ($(date +%s%N) - t) / 1000000 )) ms, $(wc -l < $D/out.txt) lines"
grep "takes\|\[trace\]" $D/err $D/log | sed 's/^[^:]*://' | grep -v "^\[mecat_hip\] [a-z_0-9]* " | tr '\n' ';' | cut -c1-1500; echo
done; done
rm -rf $D
