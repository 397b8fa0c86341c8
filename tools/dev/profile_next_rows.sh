#!/bin/bash
# dev helper (GPU box): rocprofv3 passes of the kernels behind the "next" rows that bench.py only times as extras:
#   cns_extend (row N1: mecat2cns re-aligner, tools/dev/bench_cns.py: 20 000 reads x 15 kb, ~442 k candidates)
#   asm_seed / asm_extend (row N3: the mecat2asmpw drop-in through its own command line, tools/dev/asmpw_scale.py, device leg only)
# kernel trace + stats, FETCH_SIZE / WRITE_SIZE, SQ instruction counts -> gpurun_out/profiles_out/<tag>_{cns,asm}_*.csv
TAG=${1:-r04}
R=$(pwd); O=$R/gpurun_out/prof_${TAG}_next; P=$R/gpurun_out/profiles_out
rm -rf $O; mkdir -p $O $P
cd /tmp && export TMPDIR=/tmp
run() {   # name, command...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name.k -- "$@" > $O/$name.k.log 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/$name.f -- "$@" > $O/$name.f.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/$name.w -- "$@" > $O/$name.w.log 2>&1
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $O/$name.s -- "$@" > $O/$name.s.log 2>&1
}
run cns python $R/tools/dev/bench_cns.py 20000
ASMPW_SCALE_NOREF=1 run asm python $R/tools/dev/asmpw_scale.py 20000 8000 5000000 2 mecat2asmpw 32 1
cd $R
python3 - <<PY
import collections, csv, glob, os
O, P, TAG = "$O", "$P", "$TAG"
for name, keep in (("cns", ("cns_",)), ("asm", ("asm_", "cns_extend", "ix_"))):
    stats = {}
    for f in glob.glob(os.path.join(O, name + ".k", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            if any(x in k for x in keep):
                s = stats.setdefault(k, [0, 0.0])
                s[0] += int(r["Calls"]); s[1] += float(r["TotalDurationNs"])
    ctr = collections.defaultdict(lambda: collections.defaultdict(float))
    for suf in ("f", "w", "s"):
        for f in glob.glob(os.path.join(O, name + "." + suf, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
                if any(x in k for x in keep):
                    ctr[k][r["Counter_Name"]] += float(r["Counter_Value"])
    with open(os.path.join(P, "%s_%s_kernels.md" % (TAG, name)), "w") as out:
        out.write("# %s - %s kernels (rocprofv3 --kernel-trace --stats and --pmc passes, tools/dev/profile_next_rows.sh)\n\n" % (TAG, name))
        out.write("FETCH_SIZE / WRITE_SIZE in KiB x 1024, summed over the run's launches; instruction counts are wave-level.\n\n")
        out.write("| kernel | launches | total ms | fetch GB | write GB | (fetch + write) / time TB/s | VALU | SALU | LDS |\n|---|---|---|---|---|---|---|---|---|\n")
        for k, (n, ns) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
            c = ctr.get(k, {})
            fb, wb = c.get("FETCH_SIZE", 0) * 1024, c.get("WRITE_SIZE", 0) * 1024
            out.write("| %s | %d | %.2f | %.2f | %.2f | %.2f | %.3g | %.3g | %.3g |\n" % (k[:60], n, ns / 1e6, fb / 1e9, wb / 1e9, (fb + wb) / 1e12 / (ns / 1e9) if ns else 0,
                      c.get("SQ_INSTS_VALU", 0), c.get("SQ_INSTS_SALU", 0), c.get("SQ_INSTS_LDS", 0)))
    log = open(os.path.join(O, name + ".k.log")).read().splitlines()
    open(os.path.join(P, "%s_%s_run.txt" % (TAG, name)), "w").write("\n".join(l for l in log if not l.startswith(("E2", "W2", "I2")))[-3000:] + "\n")
PY
find $O -name "*.db" -delete 2>/dev/null; find $O -type f -size +20M -delete 2>/dev/null
cat $P/${TAG}_cns_kernels.md $P/${TAG}_asm_kernels.md
