"""Dev helper: hammer the two-process mode of the driver (both ranks on GPU 0) and print what a failing rank said."""
import json, os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))      # tests/helpers.py
import helpers as H
BIN = os.path.join(H.ROOT, "mecat_amd", "bin", "mecat2pw")
G = json.load(open(os.path.join(H.GOLDEN, "golden.json")))
g = G["sets"]["tiny"]["gen"]
codes, lens = H.synth_reads(g["nreads"], g["L"], g["err"], g["genome"], g["seed"], g.get("ont", 0))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
for it in range(n):
    for task in ("0", "1"):
        with tempfile.TemporaryDirectory() as d:
            fa = os.path.join(d, "t.fa")
            H.write_fasta(fa, codes, lens)
            env = dict(os.environ, MECAT_HIP_MCS="250000", MECAT_TRACE="1")
            procs = []
            for rank in (1, 0):
                e = dict(env, MECAT_HIP_WORLD="2", MECAT_HIP_RANK=str(rank), MECAT_HIP_DEVICE="0")
                procs.append((rank, subprocess.Popen([BIN, "-j", task, "-d", fa, "-o", os.path.join(d, "two.out"), "-w", os.path.join(d, "w"), "-t", "4"],
                                                     stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e)))
            t0 = time.time()
            for rank, p in procs:
                try:
                    out, err = p.communicate(timeout=40)
                    if p.returncode != 0:
                        bad += 1
                        print("iteration %d task %s rank %d rc %d\n%s" % (it, task, rank, p.returncode, err[-1500:]), flush=True)
                except subprocess.TimeoutExpired:
                    bad += 1
                    p.kill()
                    out, err = p.communicate()
                    print("iteration %d task %s rank %d TIMEOUT\n%s\n--- files: %s" % (it, task, rank, err[-1500:], sorted(os.listdir(os.path.join(d, "w"))) if os.path.isdir(os.path.join(d, "w")) else None), flush=True)
print("done: %d iterations, %d bad" % (n, bad))
