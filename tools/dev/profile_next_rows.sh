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
# config 4 itself (bench.py --workload config4, 23 700 templates of it (four forward launches of 262 144 jobs, as in the whole run): the same kernels on the same kind of slices), HBM counters only:
# what bench_config4.py quotes as roofline.traffic for its dominant kernel
MECAT_BENCH_TEMPLATES=23700 timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/c4.f -- python $R/bench.py --workload config4 --no-cpu --steps 1 --warmup 0 > $O/c4.f.log 2>&1
MECAT_BENCH_TEMPLATES=23700 timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/c4.w -- python $R/bench.py --workload config4 --no-cpu --steps 1 --warmup 0 > $O/c4.w.log 2>&1
cd $R
python3 - <<PY
import collections as _c, csv as _csv, glob as _g, json as _j, os as _o
_t = _c.defaultdict(lambda: {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "launches": 0})
for _suf, _ctr in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
    for _f in _g.glob(_o.path.join("$O", "c4." + _suf, "**", "*counter_collection.csv"), recursive=True):
        for _r in _csv.DictReader(open(_f)):
            _n = _r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
            _k = (_n.split(">(", 1)[0] + ">") if "dw_extend2<" in _n else _n.split("(", 1)[0]
            if _k.startswith("cns_") or _k == "dw_extend2<true>":
                if _r["Counter_Name"] == _ctr:
                    _t[_k][_ctr] += float(_r["Counter_Value"]) * 1024
                    if _ctr == "FETCH_SIZE": _t[_k]["launches"] += 1
_out = {k: {"fetch_bytes_per_launch": v["FETCH_SIZE"] / max(1, v["launches"]), "write_bytes_per_launch": v["WRITE_SIZE"] / max(1, v["launches"]), "launches": v["launches"]} for k, v in _t.items()}
_out["_what"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KiB x 1024) of `MECAT_BENCH_TEMPLATES=23700 bench.py --workload config4 --no-cpu --steps 1 --warmup 0`, per launch"
_j.dump(_out, open(_o.path.join("$P", "${TAG}_config4_hbm_traffic.json"), "w"), indent=1)
PY
python3 - <<PY
import collections, csv, glob, os
O, P, TAG = "$O", "$P", "$TAG"
# (the forward pass of the re-aligner is the template instantiation dw_extend2<true>: its name has no "cns_" in it — VERDICT r05 missing 5)
for name, keep in (("cns", ("cns_", "dw_extend2<true>", "dw_extend2<(bool)1>")), ("asm", ("asm_", "cns_extend", "ix_"))):
    stats = {}
    for f in glob.glob(os.path.join(O, name + ".k", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(", 1)[0] if "dw_extend2<" not in r["Name"] else r["Name"].replace("void ", "").split(">(", 1)[0] + ">"
            if any(x in k for x in keep):
                s = stats.setdefault(k, [0, 0.0])
                s[0] += int(r["Calls"]); s[1] += float(r["TotalDurationNs"])
    ctr = collections.defaultdict(lambda: collections.defaultdict(float))
    for suf in ("f", "w", "s"):
        for f in glob.glob(os.path.join(O, name + "." + suf, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(", 1)[0] if "dw_extend2<" not in r["Kernel_Name"] else r["Kernel_Name"].replace("void ", "").split(">(", 1)[0] + ">"
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
